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
__device__ __forceinline__ float xor8_sum(float x) {
  int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false);
  return x + __builtin_bit_cast(float, t);
}
__global__ void k(float* o){
  float x = (float)(1 << (threadIdx.x % 20)) + threadIdx.x * 0.001f;
  o[threadIdx.x] = xor32_sum(x);
  o[64 + threadIdx.x] = xor16_sum(x);
  o[128 + threadIdx.x] = xor8_sum(x);
  o[192 + threadIdx.x] = x;
}
int main(){ float* d; hipMalloc(&d, 256*4); k<<<1,64>>>(d); float h[256]; hipMemcpy(h,d,1024,hipMemcpyDeviceToHost);
  int bad32=0,bad16=0,bad8=0; for(int i=0;i<64;i++){ float x=h[192+i]; if(h[i]!=x+h[192+(i^32)]) bad32++; if(h[64+i]!=x+h[192+(i^16)]) bad16++; if(h[128+i]!=x+h[192+(i^8)]) bad8++; }
  printf("bad32 %d bad16 %d bad8 %d\n",bad32,bad16,bad8); return 0; }
