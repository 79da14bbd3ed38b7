#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o){
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64+threadIdx.x] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[128+threadIdx.x] = q[0]; o[192+threadIdx.x] = q[1];
}
int main(){ unsigned* d; hipMalloc(&d, 256*4); k<<<1,64>>>(d); unsigned h[256]; hipMemcpy(h,d,1024,hipMemcpyDeviceToHost);
  for(int j=0;j<4;j++){ printf("%d:",j); for(int i=0;i<64;i+=1) printf(" %u",h[j*64+i]); printf("\n"); } return 0; }
