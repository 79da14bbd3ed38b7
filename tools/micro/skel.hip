// GEMM-skeleton probe for the Swin block kernel (round 3): how fast can the block's GEMM stages run in each workgroup
// shape, with no LayerNorm / softmax / GELU at all?  A "window" = 64 tokens; its block is modelled as NSLOT = 48 uniform
// steps (QKV 18 + proj 6 + fc1 12 + fc2 12), each a [64 tokens] x [192 columns] x [32 k] product = 12 MFMAs per wave
// of a 4-wave team (2,304 of the block's 2,496 MFMAs; attention is left out).  Activations come from an LDS image
// ([k-group][token] 16-B cells, conflict-free), weights from a fragment-ordered stream of 12 KiB slots.
//   CUR : one window per 4-wave workgroup, 3 workgroups per CU, weight fragments global -> registers (shipped structure)
//   G   : two windows per 4-wave workgroup (128 x 48 tiles), weight fragments global -> registers, one workgroup per CU
//   A   : two windows per 8-wave workgroup, weights through a 4-slot LDS ring filled by LDS-DMA, one barrier per step
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/skel.hip -o tools/micro/bin/skel
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <functional>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
struct __attribute__((aligned(16))) Frag { bf16x8 v; };
#define DEV __device__ __forceinline__

constexpr int NSLOT = 48, KC = 6;

DEV void mma(const Frag& x, const Frag& y, f32x4& c) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.v, y.v, c, 0, 0, 0); }

DEV void fill_image(Frag* img, const Frag* src, int cells, int tid, int nthr) {
    for (int i = tid; i < cells; i += nthr) img[i] = src[i];
}

// ------------------------------------------------------------------------------------------------ CUR
template <int DIST>
__global__ __launch_bounds__(256, 3) void k_cur(const Frag* W, const Frag* Ain, float* out, int nslot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag* Aimg = reinterpret_cast<Frag*>(smem);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ar = lane & 15, ag = lane >> 4;
    fill_image(Aimg, Ain, 24 * 64, threadIdx.x, 256);
    __syncthreads();
    f32x4 acc[4][3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = (f32x4)(0.f);
    Frag ring[DIST + 1][3];
    auto loadb = [&](int s, Frag (&b)[3]) {
        const Frag* f = W + ((size_t)s * 12 + 3 * w) * 64 + lane;
#pragma unroll
        for (int n = 0; n < 3; ++n) b[n] = f[n * 64];
    };
    auto loada = [&](int s, int h, Frag (&a)[2]) {
        const Frag* p = Aimg + ((s % KC) * 4 + ag) * 64 + h * 32 + ar;
        a[0] = p[0];
        a[1] = p[16];
    };
#pragma unroll
    for (int s = 0; s < DIST; ++s) loadb(s, ring[s]);
    Frag a0[2], a1[2];
    loada(0, 0, a0);
    for (int s0 = 0; s0 < nslot; s0 += (DIST + 1)) {
#pragma unroll
        for (int u = 0; u < DIST + 1; ++u) {
            const int s = s0 + u;
            if (s + DIST < nslot) loadb(s + DIST, ring[(u + DIST) % (DIST + 1)]);
            loada(s, 1, a1);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) mma(ring[u][n], a0[m], acc[m][n]);
            loada(s + 1, 0, a0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) mma(ring[u][n], a1[m], acc[2 + m][n]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x4 t = (f32x4)(0.f);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) t += acc[m][n];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
}

// CUR + synthetic non-GEMM work: V independent v_fma per step, either after every step (BURST = 1) or as one burst of
// BURST x V after every BURST-th step (a "stage epilogue": LayerNorm / softmax / GELU), optionally followed by a workgroup barrier.
template <int V, int BURST, bool BAR>
__global__ __launch_bounds__(256, 3) void k_curv(const Frag* W, const Frag* Ain, float* out, int nslot) {
    constexpr int DIST = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag* Aimg = reinterpret_cast<Frag*>(smem);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ar = lane & 15, ag = lane >> 4;
    fill_image(Aimg, Ain, 24 * 64, threadIdx.x, 256);
    __syncthreads();
    f32x4 acc[4][3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = (f32x4)(0.f);
    float vx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vx[i] = (float)(lane + i);
    Frag ring[DIST + 1][3];
    auto loadb = [&](int s, Frag (&b)[3]) {
        const Frag* f = W + ((size_t)s * 12 + 3 * w) * 64 + lane;
#pragma unroll
        for (int n = 0; n < 3; ++n) b[n] = f[n * 64];
    };
    auto loada = [&](int s, int h, Frag (&a)[2]) {
        const Frag* p = Aimg + ((s % KC) * 4 + ag) * 64 + h * 32 + ar;
        a[0] = p[0];
        a[1] = p[16];
    };
#pragma unroll
    for (int s = 0; s < DIST; ++s) loadb(s, ring[s]);
    Frag a0[2], a1[2];
    loada(0, 0, a0);
    for (int s0 = 0; s0 < nslot; s0 += 12) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int s = s0 + u;
            if (s + DIST < nslot) loadb(s + DIST, ring[(u + DIST) % (DIST + 1)]);
            loada(s, 1, a1);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) mma(ring[u % (DIST + 1)][n], a0[m], acc[m][n]);
            loada(s + 1, 0, a0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) mma(ring[u % (DIST + 1)][n], a1[m], acc[2 + m][n]);
            __builtin_amdgcn_sched_barrier(0);
            if ((u + 1) % BURST == 0) {
#pragma unroll
                for (int i = 0; i < V * BURST; ++i) vx[i & 7] = __builtin_fmaf(vx[i & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
                if (BAR) __syncthreads();
            }
        }
    }
    f32x4 t = (f32x4)(0.f);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) t += acc[m][n];
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) vs += vx[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3] + vs;
}

// CUR's work with v_mfma_f32_32x32x16_bf16 (round 4): the same [64 tokens] x [192 columns] x [32 k] step as 6 MFMAs of twice the FLOPs --
// an MFMA holds the SIMD's vector issue for 8 cycles whatever its shape (MI355X_MICROARCH.md), so the step costs 48 instead of 96 issue cycles beside the V
// filler instructions.  Wave w: tokens [32 (w & 1), +32) x columns [96 (w >> 1), +96) = 3 accumulator tiles of 32 x 32; per step 2 activation fragments
// (LDS) and 6 weight fragments (global: each fragment is read by the two waves of a column half -- twice CUR's L2 -> CU weight bytes).
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int V, int BURST, bool BAR>
__global__ __launch_bounds__(256, 3) void k_c32(const Frag* W, const Frag* Ain, float* out, int nslot) {
    constexpr int DIST = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag* Aimg = reinterpret_cast<Frag*>(smem);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    fill_image(Aimg, Ain, 24 * 64, threadIdx.x, 256);
    __syncthreads();
    f32x16 acc[3];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    float vx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vx[i] = (float)(lane + i);
    Frag ring[DIST + 1][6];
    auto loadb = [&](int s, Frag (&b)[6]) {
        const Frag* f = W + ((size_t)s * 12 + 6 * (w >> 1)) * 64 + lane;
#pragma unroll
        for (int n = 0; n < 6; ++n) b[n] = f[n * 64];
    };
    auto loada = [&](int s, Frag (&a)[2]) {  // k16 substep j: k-groups 2 j, 2 j + 1 of the step's chunk; lane = (k-group half, token)
        const Frag* p = Aimg + ((s % KC) * 4 + (lane >> 5)) * 64 + 32 * (w & 1) + (lane & 31);
        a[0] = p[0];
        a[1] = p[2 * 64];
    };
#pragma unroll
    for (int s = 0; s < DIST; ++s) loadb(s, ring[s]);
    Frag a0[2], a1[2];
    loada(0, a0);
    for (int s0 = 0; s0 < nslot; s0 += 12) {
#pragma unroll
        for (int u = 0; u < 12; u += 2) {
            const int s = s0 + u;
            if (s + DIST < nslot) loadb(s + DIST, ring[(u + DIST) % (DIST + 1)]);
            loada(s + 1, a1);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int n = 0; n < 3; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[u % (DIST + 1)][2 * n + j].v, a0[j].v, acc[n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if ((u + 1) % BURST == 0) {
#pragma unroll
                for (int i = 0; i < V * BURST; ++i) vx[i & 7] = __builtin_fmaf(vx[i & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
                if (BAR) __syncthreads();
            }
            if (s + 1 + DIST < nslot) loadb(s + 1 + DIST, ring[(u + 1 + DIST) % (DIST + 1)]);
            loada(s + 2, a0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int n = 0; n < 3; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[(u + 1) % (DIST + 1)][2 * n + j].v, a1[j].v, acc[n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if ((u + 2) % BURST == 0) {
#pragma unroll
                for (int i = 0; i < V * BURST; ++i) vx[i & 7] = __builtin_fmaf(vx[i & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
                if (BAR) __syncthreads();
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) t += acc[n][i];
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) vs += vx[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = t + vs;
}

// CUR with every weight load reading slot 0 (W0 = true: L1-resident, no L2 -> L1 stream) -- how much of CUR's time is the weight stream?
// and T12: ONE 768-thread workgroup = three 4-wave window teams kept in phase by a barrier every SYNC steps, so that a weight
// line fetched by one team is an L1 hit for the other two (L2 -> L1 traffic / 3).
template <bool W0, int NTEAM, int SYNC>
__global__ __launch_bounds__(256 * NTEAM, NTEAM == 1 ? 3 : 3) void k_team(const Frag* W, const Frag* Ain, float* out, int nslot) {
    constexpr int DIST = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int team = wv >> 2, w = wv & 3;
    Frag* Aimg = reinterpret_cast<Frag*>(smem) + team * (50 * 1024 / 16);
    const int ar = lane & 15, ag = lane >> 4;
    for (int i = threadIdx.x & 255; i < 24 * 64; i += 256) Aimg[i] = Ain[i];
    __syncthreads();
    f32x4 acc[4][3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = (f32x4)(0.f);
    Frag ring[DIST + 1][3];
    auto loadb = [&](int s, Frag (&b)[3]) {
        const Frag* f = W + ((size_t)(W0 ? 0 : s) * 12 + 3 * w) * 64 + lane;
#pragma unroll
        for (int n = 0; n < 3; ++n) b[n] = f[n * 64];
    };
    auto loada = [&](int s, int h, Frag (&a)[2]) {
        const Frag* p = Aimg + ((s % KC) * 4 + ag) * 64 + h * 32 + ar;
        a[0] = p[0];
        a[1] = p[16];
    };
#pragma unroll
    for (int s = 0; s < DIST; ++s) loadb(s, ring[s]);
    Frag a0[2], a1[2];
    loada(0, 0, a0);
    for (int s0 = 0; s0 < nslot; s0 += 12) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int s = s0 + u;
            if (s + DIST < nslot) loadb(s + DIST, ring[(u + DIST) % (DIST + 1)]);
            loada(s, 1, a1);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) mma(ring[u % (DIST + 1)][n], a0[m], acc[m][n]);
            loada(s + 1, 0, a0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) mma(ring[u % (DIST + 1)][n], a1[m], acc[2 + m][n]);
            __builtin_amdgcn_sched_barrier(0);
            if (NTEAM > 1 && SYNC > 0 && (u + 1) % SYNC == 0) __builtin_amdgcn_s_barrier();
        }
    }
    f32x4 t = (f32x4)(0.f);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) t += acc[m][n];
    out[(size_t)blockIdx.x * 256 * NTEAM + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
}

// ------------------------------------------------------------------------------------------------ G
// 4 waves, two windows: per step and wave 8 m-tiles x 3 n-tiles = 24 MFMAs per 3 weight fragments.
template <int DIST, int WGS = 1, int V = 0, int BURST = 1, bool BAR = false>  // V / BURST / BAR: synthetic non-GEMM work as k_curv (V v_fma per step per WINDOW).  WGS = 2 (round 5): the same tile at TWO workgroups per CU (48 KiB of LDS each): four windows resident per CU, weights fetched once per two
__global__ __launch_bounds__(256, WGS) void k_g(const Frag* W, const Frag* Ain, float* out, int nslot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag* Aimg = reinterpret_cast<Frag*>(smem);  // [24][128]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ar = lane & 15, ag = lane >> 4;
    for (int i = threadIdx.x; i < 24 * 128; i += 256) Aimg[i] = Ain[(i >> 7) * 64 + (i & 63)];
    __syncthreads();
    f32x4 acc[8][3];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = (f32x4)(0.f);
    Frag ring[DIST + 1][3];
    auto loadb = [&](int s, Frag (&b)[3]) {
        const Frag* f = W + ((size_t)s * 12 + 3 * w) * 64 + lane;
#pragma unroll
        for (int n = 0; n < 3; ++n) b[n] = f[n * 64];
    };
    auto loada = [&](int s, int q, Frag (&a)[2]) {  // quarter q: m-tiles 2q, 2q+1
        const Frag* p = Aimg + ((s % KC) * 4 + ag) * 128 + q * 32 + ar;
        a[0] = p[0];
        a[1] = p[16];
    };
#pragma unroll
    for (int s = 0; s < DIST; ++s) loadb(s, ring[s]);
    Frag aq[2][2];
    loada(0, 0, aq[0]);
    float vx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vx[i] = (float)(lane + i);
    for (int s0 = 0; s0 < nslot; s0 += (DIST + 1)) {
#pragma unroll
        for (int u = 0; u < DIST + 1; ++u) {
            const int s = s0 + u;
            if (V > 0 && s % BURST == 0 && s > 0) {  // a stage epilogue of both windows
#pragma unroll
                for (int i = 0; i < 2 * V * BURST; ++i) vx[i & 7] = __builtin_fmaf(vx[i & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
                if (BAR) __syncthreads();
            }
            if (s + DIST < nslot) loadb(s + DIST, ring[(u + DIST) % (DIST + 1)]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < 3)
                    loada(s, q + 1, aq[(q + 1) & 1]);
                else
                    loada(s + 1, 0, aq[0]);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 3; ++n) mma(ring[u][n], aq[q & 1][m], acc[2 * q + m][n]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x4 t = (f32x4)(0.f);
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) t += acc[m][n];
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) vs += vx[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3] + (V > 0 ? vs : 0.f);
}

// ------------------------------------------------------------------------------------------------ A
// 8 waves = 2 windows x 4 column slices; weights: global -> LDS ring (LDS-DMA, 1 KiB pieces) -> registers.
DEV void dma16(const void* gsrc, unsigned lds_dst) {  // wave-uniform LDS byte address, per-lane global address; lane l lands at lds_dst + 16 l
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define BARRIER() asm volatile("s_barrier" ::: "memory")

template <int MODE>  // MODE 0: per-step barrier, ring of 4 slots.  MODE 1: same, MFMAs issued before the next step's LDS reads
__global__ __launch_bounds__(512, 2) void k_a(const Frag* W, const Frag* Ain, float* out, int nslot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int R = 4;
    Frag* Aimg = reinterpret_cast<Frag*>(smem);          // [2 groups][24][64]
    Frag* ring = Aimg + 2 * 24 * 64;                      // [R][12][64]
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(ring);  // LDS byte address of the ring
    const int lane = threadIdx.x & 63, W8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = W8 >> 2, w = W8 & 3;
    const int ar = lane & 15, ag = lane >> 4;
    for (int i = threadIdx.x; i < 2 * 24 * 64; i += 512) Aimg[i] = Ain[i % (24 * 64)];
    __syncthreads();
    f32x4 acc[4][3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = (f32x4)(0.f);
    const Frag* Ag = Aimg + g * 24 * 64;
    auto issue = [&](int s) {  // this wave's pieces of slot s: fragment W8, and fragment 8 + w when its group's turn
        const unsigned base = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)((s % R) * 12 * 1024));
        dma16(W + ((size_t)s * 12 + W8) * 64 + lane, base + W8 * 1024);
        if (g == (s & 1)) dma16(W + ((size_t)s * 12 + 8 + w) * 64 + lane, base + (8 + w) * 1024);
    };
    auto readstep = [&](int s, Frag (&a)[4], Frag (&b)[3]) {
        const Frag* p = Ag + ((s % KC) * 4 + ag) * 64 + ar;
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = p[m * 16];
        const Frag* f = ring + ((s % R) * 12 + 3 * w) * 64 + lane;
#pragma unroll
        for (int n = 0; n < 3; ++n) b[n] = f[n * 64];
    };
#pragma unroll
    for (int s = 0; s < R; ++s) issue(s);
    WAIT_VM(4);
    BARRIER();
    Frag a[2][4], b[2][3];
    readstep(0, a[0], b[0]);
    for (int s0 = 0; s0 < nslot; s0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = s0 + u;
            WAIT_VM(3);
            WAIT_LGKM0();
            BARRIER();
            if (s + R < nslot) issue(s + R);
            if (MODE == 0) {
                if (s + 1 < nslot) readstep(s + 1, a[u ^ 1], b[u ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    mma(b[u][n], a[u][m], acc[m][n]);
                    if (MODE == 1 && m == 0 && n == 0 && s + 1 < nslot) readstep(s + 1, a[u ^ 1], b[u ^ 1]);
                }
        }
    }
    WAIT_VM(0);
    f32x4 t = (f32x4)(0.f);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) t += acc[m][n];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
}

// ------------------------------------------------------------------------------------------------ host
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <typename F>
float time_us(F&& launch, int iters = 30) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
    const size_t wbytes = (size_t)NSLOT * 12 * 1024;
    std::vector<unsigned short> hw(wbytes / 2), ha(24 * 64 * 8);
    srand(1);
    auto rbf = []() {  // random bf16 in about [-1, 1)
        float f = (float)rand() / RAND_MAX * 2.f - 1.f;
        unsigned u;
        memcpy(&u, &f, 4);
        return (unsigned short)(u >> 16);
    };
    for (auto& v : hw) v = rbf();
    for (auto& v : ha) v = rbf();
    Frag *W, *A;
    float* out;
    CK(hipMalloc(&W, wbytes));
    CK(hipMalloc(&A, ha.size() * 2));
    CK(hipMalloc(&out, 4096 * 512 * 4));
    CK(hipMemcpy(W, hw.data(), wbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    const int lds_cur = 50 * 1024, lds_g = 100 * 1024, lds_a = (2 * 24 + 48) * 1024 + 1024;
    CK(hipFuncSetAttribute((const void*)k_cur<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_cur));
    CK(hipFuncSetAttribute((const void*)k_cur<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_cur));
    CK(hipFuncSetAttribute((const void*)k_g<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_g));
    CK(hipFuncSetAttribute((const void*)k_g<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_g));
    CK(hipFuncSetAttribute((const void*)k_a<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
    CK(hipFuncSetAttribute((const void*)k_a<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
#define CURV(V, B, R) do { auto kf = k_curv<V, B, R>; CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, lds_cur)); \
        char nm[64]; snprintf(nm, 64, "v%d b%d %s", V, B, R ? "bar" : "-"); \
        rows.push_back({nm, [=](int nw) { hipLaunchKernelGGL(kf, dim3(nw), dim3(256), lds_cur, 0, W, A, out, NSLOT); }}); } while (0)
    std::vector<std::pair<std::string, std::function<void(int)>>> rows;
    const int wins[] = {81, 486, 648, 768, 1296, 1944};
    printf("# us per launch; TF/s counts 2304 MFMAs x 16384 FLOP per window\n");
    printf("%-10s", "windows");
    for (int nw : wins) printf("%9d", nw);
    printf("\n");
    auto row = [&](const char* name, auto&& fn) {
        printf("%-10s", name);
        for (int nw : wins) {
            float us = time_us([&] { fn(nw); });
            printf("%9.1f", us);
        }
        printf("   | TF/s:");
        for (int nw : wins) {
            float us = time_us([&] { fn(nw); }, 10);
            printf(" %6.0f", nw * 2304.0 * 16384 / us * 1e-6);
        }
        printf("\n");
        fflush(stdout);
    };
    row("cur d3", [&](int nw) { hipLaunchKernelGGL(k_cur<3>, dim3(nw), dim3(256), lds_cur, 0, W, A, out, NSLOT); });
    row("G d3", [&](int nw) { hipLaunchKernelGGL(k_g<3>, dim3(nw / 2), dim3(256), lds_g, 0, W, A, out, NSLOT); });
    row("A m0", [&](int nw) { hipLaunchKernelGGL(k_a<0>, dim3(nw / 2), dim3(512), lds_a, 0, W, A, out, NSLOT); });
    {
        auto kg2 = k_g<3, 2>;
        const int lds_g2 = 24 * 128 * 16 + 1024;
        CK(hipFuncSetAttribute((const void*)kg2, hipFuncAttributeMaxDynamicSharedMemorySize, lds_g2));
        row("G2 d3", [&](int nw) { hipLaunchKernelGGL(kg2, dim3(nw / 2), dim3(256), lds_g2, 0, W, A, out, NSLOT); });
        auto kg22 = k_g<2, 2>;
        CK(hipFuncSetAttribute((const void*)kg22, hipFuncAttributeMaxDynamicSharedMemorySize, lds_g2));
        row("G2 d2", [&](int nw) { hipLaunchKernelGGL(kg22, dim3(nw / 2), dim3(256), lds_g2, 0, W, A, out, NSLOT); });
        auto kg2v = k_g<3, 2, 40, 6, true>;  // + the synthetic VALU / barrier load of 'v40 b6 bar' (per window)
        CK(hipFuncSetAttribute((const void*)kg2v, hipFuncAttributeMaxDynamicSharedMemorySize, lds_g2));
        row("G2 v40b6", [&](int nw) { hipLaunchKernelGGL(kg2v, dim3(nw / 2), dim3(256), lds_g2, 0, W, A, out, NSLOT); });
        auto kg2w = k_g<2, 2, 40, 6, true>;
        CK(hipFuncSetAttribute((const void*)kg2w, hipFuncAttributeMaxDynamicSharedMemorySize, lds_g2));
        row("G2d2 v40b6", [&](int nw) { hipLaunchKernelGGL(kg2w, dim3(nw / 2), dim3(256), lds_g2, 0, W, A, out, NSLOT); });
    }
#define TEAM(W0, NT, SY) do { auto kf = k_team<W0, NT, SY>; const int lds = NT * 50 * 1024; CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        char nm[64]; snprintf(nm, 64, "t%d %s s%d", NT, W0 ? "w0" : "ws", SY); \
        rows.push_back({nm, [=](int nw) { hipLaunchKernelGGL(kf, dim3(nw / NT), dim3(256 * NT), lds, 0, W, A, out, NSLOT); }}); } while (0)
    TEAM(false, 1, 0); TEAM(true, 1, 0); TEAM(false, 3, 0); TEAM(false, 3, 1); TEAM(false, 3, 2); TEAM(false, 3, 6); TEAM(true, 3, 6); TEAM(false, 2, 1); TEAM(false, 2, 2);
    CURV(0, 1, false); CURV(0, 6, true);
    CURV(40, 6, true);
#define C32(V, B, R) do { auto kf = k_c32<V, B, R>; CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, lds_cur)); \
        char nm[64]; snprintf(nm, 64, "c32 v%d b%d %s", V, B, R ? "bar" : "-"); \
        rows.push_back({nm, [=](int nw) { hipLaunchKernelGGL(kf, dim3(nw), dim3(256), lds_cur, 0, W, A, out, NSLOT); }}); } while (0)
    if (argc > 1 && !strcmp(argv[1], "c32")) {  // round 4: MFMA shape probe (same work, 16x16x32 vs 32x32x16) with 0 / 20 / 40 / 60 filler VALU per step
        rows.clear();
        CURV(0, 1, false); C32(0, 1, false); CURV(20, 1, false); C32(20, 1, false); CURV(40, 1, false); C32(40, 1, false); CURV(60, 1, false); C32(60, 1, false);
        CURV(40, 6, true); C32(40, 6, true);
    }
    for (auto& r : rows) row(r.first.c_str(), r.second);
    CK(hipDeviceSynchronize());
    return 0;
}
