#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float xor32_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float xor16_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k(float* o){
  float x = (float)(1 << (threadIdx.x % 20)) + threadIdx.x * 0.001f;
  o[threadIdx.x] = xor32_sum(x);
  o[64 + threadIdx.x] = xor16_sum(x);
  o[128 + threadIdx.x] = xor16_sum(xor32_sum(x));
  o[192 + threadIdx.x] = x;
}
int main(){ float* d; hipMalloc(&d, 256*4); k<<<1,64>>>(d); float h[256]; hipMemcpy(h,d,1024,hipMemcpyDeviceToHost);
  int bad32=0,bad16=0,bad=0; for(int i=0;i<64;i++){ float x=h[192+i]; float e32 = h[192+(i&31)] + h[192+(i&31)+32]; float e16 = h[192+(i&~16)] + h[192+(i|16)];
    if(h[i]!=e32) bad32++; if(h[64+i]!=e16) bad16++; }
  printf("bad32 %d bad16 %d\n",bad32,bad16); return 0; }
