// Issue cost of v_pk_fma_f16 / v_exp_f16 / v_rcp_f16 against v_fma_f32 / v_exp_f32 on gfx950: one wave per SIMD, dependent-free streams.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters) {
    float a[8];
    half2_t h[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; h[i] = half2_t{(_Float16)(0.5f + i * 0.01f), (_Float16)(0.25f)}; }
    const half2_t hm = half2_t{(_Float16)0.999f, (_Float16)1.001f}, hc = half2_t{(_Float16)0.001f, (_Float16)0.002f};
    long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
            if (MODE == 1) h[i] = h[i] * hm + hc;                                   // v_pk_fma_f16
            if (MODE == 2) a[i] = __builtin_amdgcn_exp2f(a[i] * 0.01f);
        }
    }
    long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)h[i][0] + (float)h[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 8);
}
int main() {
    float* d; hipMalloc(&d, 1 << 20);
    float h;
    const char* names[] = {"v_fma_f32", "v_pk_fma_f16", "v_mul+v_exp_f32"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int waves = 1; waves <= 3; waves += 2) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256 * waves), 0, 0, d, 4096);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256 * waves), 0, 0, d, 4096);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256 * waves), 0, 0, d, 4096);
            hipDeviceSynchronize();
            hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
            printf("%-16s %d wave(s)/SIMD: %.2f cycles per instruction (per wave)\n", names[mode], waves, h);
        }
    }
    return 0;
}
