// Latency probe: dependent 16-B-per-lane (1 KiB per wave) loads from buffers of different sizes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void chase(const f32x4* buf, size_t n_vec, int iters, unsigned stride, unsigned long long* out, float* sink) {
    const int lane = threadIdx.x & 63;
    size_t idx = ((size_t)blockIdx.x * 977 + (threadIdx.x >> 6) * 131) * 64 % n_vec;
    f32x4 acc = {0, 0, 0, 0};
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        f32x4 v = buf[(idx + lane) % n_vec];
        acc += v;
        // dependent: next index depends on loaded value (always +stride since buffer holds zeros)
        idx = (idx + stride + (size_t)(v[0] != 0.f)) % n_vec;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (lane == 0 && blockIdx.x == 0) out[4000] = r1 - r0;
    if (acc[0] == 123.f) sink[0] = acc[1];
}

int main() {
    const size_t sizes[] = {256 << 10, 2 << 20, 32 << 20, 200 << 20, (size_t)2 << 30};
    for (size_t sz : sizes) {
        f32x4* buf;
        hipMalloc(&buf, sz);
        hipMemset(buf, 0, sz);
        unsigned long long* out;
        float* sink;
        hipMalloc(&out, 8 * 4096);
        hipMalloc(&sink, 16);
        for (int blocks : {1, 256, 1024}) {
            const int iters = 200;
            size_t n_vec = sz / 16;
            unsigned stride = 64 * 37;  // jump 37 KiB
            chase<<<blocks, 64>>>(buf, n_vec, iters, stride, out, sink);  // warm
            hipDeviceSynchronize();
            chase<<<blocks, 64>>>(buf, n_vec, iters, stride, out, sink);
            hipDeviceSynchronize();
            unsigned long long h[4096];
            hipMemcpy(h, out, 8 * blocks, hipMemcpyDeviceToHost);
            double s = 0;
            for (int i = 0; i < blocks; ++i) s += h[i];
            unsigned long long rt; hipMemcpy(&rt, out + 4000, 8, hipMemcpyDeviceToHost);
            printf("buf %8zu KiB  waves %5d : %8.1f shader-ticks per dependent 1KiB load; block0: %llu realtime ticks (10ns) for %llu shader ticks => %.0f MHz\n", sz >> 10, blocks, s / blocks / iters, rt, h[0], (double)h[0] / (rt * 10e-9) / 1e6);
        }
        hipFree(buf);
        hipFree(out);
        hipFree(sink);
    }
    return 0;
}
