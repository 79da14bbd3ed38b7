#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float xor32_sum(float x) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm volatile("" : "+v"(b));
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float xor16_sum(float x) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm volatile("" : "+v"(b));
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__global__ void k(float* o){
  float x = (float)threadIdx.x;
  o[threadIdx.x] = xor32_sum(x);
  o[64 + threadIdx.x] = xor16_sum(x);
}
int main(){ float* d; hipMalloc(&d, 256*4); k<<<1,64>>>(d); float h[256]; hipMemcpy(h,d,1024,hipMemcpyDeviceToHost);
  for(int j=0;j<2;j++){ printf("%d:",j); for(int i=0;i<64;i+=1) printf(" %g",h[j*64+i]); printf("\n"); } return 0; }
