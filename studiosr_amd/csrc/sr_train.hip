// Training engine (BASELINE config 5): the kernels behind studiosr_amd/autograd.py.
//
// Everything a training step of SwinIR / HAT / EDSR / RCAN needs beyond the inference kernels is expressed with
//   * ONE strided, batched, split-K GEMM on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain, so
//     forward values and gradients agree with the fp32 reference to rounding) -- nn.Linear / 3x3 conv (over an im2col buffer) /
//     q k^T / P v forward, their data gradients (transposed strides) and their weight gradients (contraction over tokens, split
//     over workgroups, fp32 atomics).  The fp32 MFMA operand is ONE float per lane (A[i = lane & 15][k = lane >> 4]), so every
//     transpose is just a stride: no packing, no LDS, parameters are read and gradients written in the reference's own layouts
//     (studiosr/models/*.py state_dict shapes), which is what lets torch.optim / DistributedDataParallel work unchanged
//     (studiosr/engine/trainer.py:89-109);
//   * row kernels (LayerNorm forward / backward, softmax(+bias+mask) forward / backward), index-map copies (window partition +
//     roll, OCA unfold, PixelShuffle, im2col / col2im -- gathers in both directions, never scatter-atomics) and flat elementwise ops.
// All tensors are fp32 and UNPADDED (C = 180 stays 180): the reference's bf16 autocast contract (fp32 master weights, fp32 grads)
// is exceeded, not approximated.  Speed-of-light kernels for this path are later rounds' work; the bar here is gradient parity.
#include "sr_common.h"
#include "sr_host.h"

#include <type_traits>
#include <algorithm>

namespace {

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte global access that only needs dword alignment

// ----------------------------------------------------------------------------- strided batched GEMM (fp32 MFMA)
__global__ __launch_bounds__(256) void sr_bgemm_kernel(SrBgemm g) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wm = w >> 1, wn = w & 1;
    const int li = lane & 15, lg = lane >> 4;
    int z = blockIdx.z;
    const int ks = z % g.ksplit;
    z /= g.ksplit;
    const int b2 = z % g.nb2, b1 = z / g.nb2;
    const float* A = g.A + (long long)b1 * g.sa_b1 + (long long)b2 * g.sa_b2;
    const float* B = g.B + (long long)b1 * g.sb_b1 + (long long)b2 * g.sb_b2;
    float* C = g.C + (long long)b1 * g.sc_b1 + (long long)b2 * g.sc_b2;
    const int m0 = blockIdx.y * 64 + wm * 32, n0 = blockIdx.x * 64 + wn * 32;
    const int kchunk = (((g.K + g.ksplit - 1) / g.ksplit) + 15) & ~15;
    const int kbeg = ks * kchunk, kend = min(g.K, kbeg + kchunk);

    const float* ap[2];
    const float* bp[2];
    bool aok[2], bok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = m0 + 16 * t + li, n = n0 + 16 * t + li;
        aok[t] = m < g.M;
        bok[t] = n < g.N;
        ap[t] = A + (long long)min(m, g.M - 1) * g.sa_m;
        bp[t] = B + (long long)min(n, g.N - 1) * g.sb_n;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i][0] = acc[i][1] = (f32x4)(0.0f);
    for (int k = kbeg; k < kend; k += 16) {
        float av[2][4], bv[2][4];
        // MFMA step j contracts k = {k + j, k + 4 + j, k + 8 + j, k + 12 + j}: lane group lg supplies k + 4 lg + j for both operands,
        // so a k-contiguous operand is read as four consecutive floats per lane
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = k + 4 * lg + j;
            const bool ok = kk < kend;
            const long long ka = (long long)min(kk, kend - 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float a = ap[t][ka * g.sa_k], b = bp[t][ka * g.sb_k];
                av[t][j] = (ok && aok[t]) ? a : 0.0f;
                bv[t][j] = (ok && bok[t]) ? b : 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][j], bv[nt][j], acc[mt][nt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + 16 * nt + li;
            if (n >= g.N) continue;
            const float bias = (g.bias && ks == 0) ? g.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * mt + 4 * lg + r;
                if (m >= g.M) continue;
                float* c = C + (long long)m * g.sc_m + (long long)n * g.sc_n;
                const float v = g.alpha * acc[mt][nt][r] + bias;
                if (g.ksplit > 1)
                    atomicAdd(c, v);
                else if (g.accumulate)
                    *c += v;
                else
                    *c = v;
            }
        }
}

// ----------------------------------------------------------------------------- the same GEMM, LDS-tiled (large M, N)
// 128 x 128 output tile per workgroup, 4 waves of 64 x 64 (16 accumulator tiles each); K in steps of 16.  Both operand tiles are
// staged [row][16 k] (row stride 20 floats: the ds_read_b128 of 16 consecutive rows is bank-conflict free) through registers, the
// global loads of step t+1 issued before the MFMAs of step t.  A lane's fragment of a 16-row tile is one ds_read_b128 = its four
// k values 4 lg .. 4 lg + 3, consumed by MFMA steps j = 0..3 (same k permutation on both operands as in the direct kernel).
// Staging vectorises along whichever axis is contiguous: float4 along k (stride_k == 1) or float4 along rows (stride_row == 1).
constexpr int BT = 128, BK = 16, BLD = 20;
static_assert(16 * (BT + 4) <= BT * BLD, "the [k][row] tile must fit the [row][k] tile");

// LDS tile of one operand: k-contiguous and generic operands are staged [row][16 k] (row stride BLD, fragment = one ds_read_b128);
// row-contiguous operands are staged [16 k][row] (row stride KLD: the 16-byte stores of 4 consecutive rows are conflict-free, the
// fragment is four ds_read_b32 whose 16 lanes read 16 consecutive rows).
constexpr int KLD = BT + 4;
constexpr int TILE_FLOATS = BT * BLD;  // >= 16 * KLD

template <int MODE>  // 0: k contiguous, 1: row contiguous, 2: generic
SR_DEV void bg_load(const float* __restrict__ base, long long s_row, long long s_k, int row0, int nrows, int k0, int kend, float (&r)[8]) {
    const int t = threadIdx.x;
    if (MODE == 0) {  // thread -> (row = t / 4 + 64 i, k quad = t % 4)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = row0 + (t >> 2) + 64 * i, k = k0 + 4 * (t & 3);
            if (row < nrows && k + 4 <= kend) {  // whole quad in range: one 16-byte load
                const f32x4u v = *reinterpret_cast<const f32x4u*>(base + (long long)row * s_row + k);
                r[4 * i] = v[0]; r[4 * i + 1] = v[1]; r[4 * i + 2] = v[2]; r[4 * i + 3] = v[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) r[4 * i + j] = (row < nrows && k + j < kend) ? base[(long long)row * s_row + k + j] : 0.f;
            }
        }
    } else if (MODE == 1) {  // thread -> (row quad = t % 32, k = t / 32 + 8 i)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = row0 + 4 * (t & 31), k = k0 + (t >> 5) + 8 * i;
            if (k < kend && row + 4 <= nrows) {
                const f32x4u v = *reinterpret_cast<const f32x4u*>(base + (long long)k * s_k + row);
                r[4 * i] = v[0]; r[4 * i + 1] = v[1]; r[4 * i + 2] = v[2]; r[4 * i + 3] = v[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) r[4 * i + j] = (k < kend && row + j < nrows) ? base[(long long)k * s_k + row + j] : 0.f;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // element e = t + 256 i -> (row = e / 16, k = e % 16)
            const int e = t + 256 * i, row = row0 + (e >> 4), k = k0 + (e & 15);
            const float v = base[(long long)min(row, nrows - 1) * s_row + (long long)min(k, kend - 1) * s_k];
            r[i] = (row < nrows && k < kend) ? v : 0.f;
        }
    }
}
template <int MODE>
SR_DEV void bg_store(float* __restrict__ tile, const float (&r)[8]) {
    const int t = threadIdx.x;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(tile + ((t >> 2) + 64 * i) * BLD + 4 * (t & 3)) = f32x4{r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
    } else if (MODE == 1) {  // [k][row]
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(tile + ((t >> 5) + 8 * i) * KLD + 4 * (t & 31)) = f32x4{r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = t + 256 * i;
            tile[(e >> 4) * BLD + (e & 15)] = r[i];
        }
    }
}
// the four k values 4 lg .. 4 lg + 3 of row `row` of a staged tile
template <int MODE>
SR_DEV f32x4 bg_frag(const float* __restrict__ tile, int row, int lg) {
    if (MODE == 1) {
        const float* p = tile + (4 * lg) * KLD + row;
        return f32x4{p[0], p[KLD], p[2 * KLD], p[3 * KLD]};
    }
    return *reinterpret_cast<const f32x4*>(tile + row * BLD + 4 * lg);
}

template <int MA, int MB>
__global__ __launch_bounds__(256) void sr_bgemm_tiled_kernel(SrBgemm g) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TILE_FLOATS];  // [buffer][A | B][tile]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wm = w >> 1, wn = w & 1;
    const int li = lane & 15, lg = lane >> 4;
    int z = blockIdx.z;
    const int ks = z % g.ksplit;
    z /= g.ksplit;
    const int b2 = z % g.nb2, b1 = z / g.nb2;
    const float* A = g.A + (long long)b1 * g.sa_b1 + (long long)b2 * g.sa_b2;
    const float* B = g.B + (long long)b1 * g.sb_b1 + (long long)b2 * g.sb_b2;
    float* C = g.C + (long long)b1 * g.sc_b1 + (long long)b2 * g.sc_b2;
    const int m0 = blockIdx.y * BT, n0 = blockIdx.x * BT;
    const int kchunk = (((g.K + g.ksplit - 1) / g.ksplit) + BK - 1) / BK * BK;
    const int kbeg = ks * kchunk, kend = min(g.K, kbeg + kchunk);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.0f);
    if (kbeg < kend) {
        float ra[8], rb[8];
        bg_load<MA>(A, g.sa_m, g.sa_k, m0, g.M, kbeg, kend, ra);
        bg_load<MB>(B, g.sb_n, g.sb_k, n0, g.N, kbeg, kend, rb);
        bg_store<MA>(lds[0][0], ra);
        bg_store<MB>(lds[0][1], rb);
        __syncthreads();
        int cur = 0;
        for (int k = kbeg; k < kend; k += BK) {
            const bool more = k + BK < kend;
            if (more) {
                bg_load<MA>(A, g.sa_m, g.sa_k, m0, g.M, k + BK, kend, ra);
                bg_load<MB>(B, g.sb_n, g.sb_k, n0, g.N, k + BK, kend, rb);
            }
            f32x4 av[4], bv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                av[t] = bg_frag<MA>(lds[cur][0], wm * 64 + t * 16 + li, lg);
                bv[t] = bg_frag<MB>(lds[cur][1], wn * 64 + t * 16 + li, lg);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][j], bv[nt][j], acc[mt][nt], 0, 0, 0);
            if (more) {
                bg_store<MA>(lds[cur ^ 1][0], ra);
                bg_store<MB>(lds[cur ^ 1][1], rb);
            }
            __syncthreads();
            cur ^= 1;
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + 16 * nt + li;
            if (n >= g.N) continue;
            const float bias = (g.bias && ks == 0) ? g.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + 16 * mt + 4 * lg + r;
                if (m >= g.M) continue;
                float* c = C + (long long)m * g.sc_m + (long long)n * g.sc_n;
                const float v = g.alpha * acc[mt][nt][r] + bias;
                if (g.ksplit > 1)
                    atomicAdd(c, v);
                else if (g.accumulate)
                    *c += v;
                else
                    *c = v;
            }
        }
}

template <int MA>
int launch_tiled_b(const SrBgemm& g, int mb, dim3 grid, hipStream_t st) {
    if (mb == 0)
        hipLaunchKernelGGL((sr_bgemm_tiled_kernel<MA, 0>), grid, dim3(256), 0, st, g);
    else if (mb == 1)
        hipLaunchKernelGGL((sr_bgemm_tiled_kernel<MA, 1>), grid, dim3(256), 0, st, g);
    else
        hipLaunchKernelGGL((sr_bgemm_tiled_kernel<MA, 2>), grid, dim3(256), 0, st, g);
    return 0;
}

// ----------------------------------------------------------------------------- the tiled GEMM with bf16 operands (autocast contract)
// Same 128 x 128 tile / 4 waves of 64 x 64, but the fp32 operands are rounded to bf16 while they are staged (what torch.autocast does to
// the reference's matmuls, studiosr/engine/trainer.py:80,102), K advances 32 per step and the MFMA is v_mfma_f32_16x16x32_bf16 (16x the
// fp32 matrix rate), fp32 accumulate.  LDS tile [row][32 k] bf16 with an 80-byte row stride: the fragment (8 consecutive k of one row)
// is one conflict-free ds_read_b128.  Staging: a k-contiguous operand gives each thread 8 consecutive k of one row (two 16-byte
// loads); a row-contiguous operand gives each thread 16 k of ONE row (16 loads, each a coalesced 256 B per wave) so that the LDS
// writes are 16-byte rows again instead of transposed 2-byte scatters.
constexpr int HK = 32, HLD = 40;  // bf16 elements per row incl. pad

// ROWS x 32 fp32 operand slab -> registers: NV = ROWS / 8 floats per thread
template <int MODE, int ROWS>
SR_DEV void hg_load(const float* __restrict__ base, long long s_row, long long s_k, int row0, int nrows, int k0, int kend, float (&v)[ROWS / 8]) {
    const int t = threadIdx.x;
    if (MODE == 0) {  // thread -> (row = t / 4 + 64 i, k octet = t % 4): v[8 i .. 8 i + 7]
#pragma unroll
        for (int i = 0; i < ROWS / 64; ++i) {
            const int row = row0 + (t >> 2) + 64 * i, k = k0 + 8 * (t & 3);
            if (row < nrows && k + 8 <= kend) {
                const f32x4u a = *reinterpret_cast<const f32x4u*>(base + (long long)row * s_row + k), b = *reinterpret_cast<const f32x4u*>(base + (long long)row * s_row + k + 4);
                v[8 * i] = a[0]; v[8 * i + 1] = a[1]; v[8 * i + 2] = a[2]; v[8 * i + 3] = a[3];
                v[8 * i + 4] = b[0]; v[8 * i + 5] = b[1]; v[8 * i + 6] = b[2]; v[8 * i + 7] = b[3];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[8 * i + j] = (row < nrows && k + j < kend) ? base[(long long)row * s_row + k + j] : 0.f;
            }
        }
    } else if (MODE == 1) {
        // row-contiguous (unit row stride, 16-byte aligned k-rows): thread -> 4 consecutive rows x KP consecutive k, one 16-byte load per k;
        // v[4 p + j] = (row 4 rg + j, k = KP kq + p)
        constexpr int RG = ROWS / 4, KP = 32 / (256 / RG);
        const int rg = t % RG, kq = t / RG;
        const int row = row0 + 4 * rg;
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            const int k = k0 + KP * kq + p;
            const float* src = base + row + (long long)k * s_k;
            if (row + 4 <= nrows && k < kend) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(src);
                v[4 * p] = a[0]; v[4 * p + 1] = a[1]; v[4 * p + 2] = a[2]; v[4 * p + 3] = a[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * p + j] = (row + j < nrows && k < kend) ? src[j] : 0.f;
            }
        }
    } else {  // generic strides: thread -> (row = t % ROWS, NV consecutive k from NV * (t / ROWS))
        constexpr int NV = ROWS / 8;
        const int row = row0 + (t & (ROWS - 1)), kb = k0 + NV * (t / ROWS);
        if (row < nrows && kb + NV <= kend) {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = base[(long long)row * s_row + (long long)(kb + j) * s_k];
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const float x = base[(long long)min(row, nrows - 1) * s_row + (long long)min(kb + j, kend - 1) * s_k];
                v[j] = (row < nrows && kb + j < kend) ? x : 0.f;
            }
        }
    }
}
template <int MODE, int ROWS>
SR_DEV void hg_store(bf16* __restrict__ tile, const float (&v)[ROWS / 8]) {
    const int t = threadIdx.x;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < ROWS / 64; ++i) *reinterpret_cast<Frag<bf16>*>(tile + ((t >> 2) + 64 * i) * HLD + 8 * (t & 3)) = frag_from8(v + 8 * i);
    } else if (MODE == 1) {
        constexpr int RG = ROWS / 4, KP = 32 / (256 / RG);
        const int rg = t % RG, kq = t / RG;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16* p = tile + (4 * rg + j) * HLD + KP * kq;
            if constexpr (KP == 4) {
                bf16x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (bf16)v[4 * q + j];
                *reinterpret_cast<bf16x4*>(p) = o;
            } else {
                typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                bf16x2_t o;
                o[0] = (bf16)v[j];
                o[1] = (bf16)v[4 + j];
                *reinterpret_cast<bf16x2_t*>(p) = o;
            }
        }
    } else {
        constexpr int NV = ROWS / 8;
        bf16* p = tile + (t & (ROWS - 1)) * HLD + NV * (t / ROWS);
#pragma unroll
        for (int i = 0; i < NV / 8; ++i) *reinterpret_cast<Frag<bf16>*>(p + 8 * i) = frag_from8(v + 8 * i);
    }
}

// bf16-operand GEMM tile 128 x BN (BN = 128 or 64), K steps of 32.  What the first version (one K step of prefetch, 128 x 128 only,
// element-wise stores) showed at the Linear shapes of a training step (tools/bgemm_bench.py: 50-140 TF/s, 1-2 TB/s): the launch is a
// few hundred workgroups that all sit in the same latency chain -- global load -> convert -> LDS -> 16 MFMAs -- once per K step.  So:
//   * TWO K steps of operands in flight in registers (the loads of step t + 2 are issued before the MFMAs of step t);
//   * BN = 64 where that gives more / fuller workgroups (N = 180, 360, 540: 6 % padding instead of 18-42 %);
//   * result tiles computed transposed (MFMA operands swapped) when C is row-major, so a lane owns 4 consecutive n: 16-byte stores.
// XCD-aware workgroup order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): in the natural (x fastest) order
// the n-tiles that re-read one A panel -- or the 15 tiles that share one split-K token slab -- land on 8 different XCDs and every panel
// crosses the fabric up to 8 times (all Linear / conv GEMMs of a training step sat at ~5 TB/s of such traffic whatever their shape).
// Here the workgroups of one XCD (linear id = xcd mod 8) own whole sharing groups: all n-tiles of an m-tile, or all tiles of a (batch, K slice).
struct BgemmGrid {
    int gx, gy, nz, natural;
    SR_DEV bool decode(int& bx, int& by, int& bz) const {
        if (natural) {
            int t = blockIdx.x;
            bx = t % gx, t /= gx;
            by = t % gy, bz = t / gy;
            return bz < nz;
        }
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        if (nz > 1) {
            const int inner = gx * gy, outer = (j / inner) * 8 + xcd, t = j % inner;
            bx = t % gx, by = t / gx, bz = outer;
            return outer < nz;
        }
        bx = j % gx, by = (j / gx) * 8 + xcd, bz = 0;
        return by < gy;
    }
    unsigned blocks() const { return natural ? (unsigned)gx * gy * nz : nz > 1 ? (unsigned)((nz + 7) / 8 * 8) * gx * gy : (unsigned)((gy + 7) / 8 * 8) * gx; }
};

template <int MA, int MB, int BN, bool TR>
__global__ __launch_bounds__(256, BN == 64 ? 4 : 2) void sr_bgemm_bf16_kernel(SrBgemm g, BgemmGrid grid) {
    constexpr int NTW = BN / 32;  // n tiles per wave (waves 2 x 2: 64 rows x BN / 2 columns each)
    __shared__ __attribute__((aligned(16))) bf16 lds[2][(BT + BN) * HLD];  // [buffer][A rows | B rows][k]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wm = w >> 1, wn = w & 1;
    const int li = lane & 15, lg = lane >> 4;
    int bx, by, z;
    if (!grid.decode(bx, by, z)) return;
    const int ks = z % g.ksplit;
    z /= g.ksplit;
    const int b2 = z % g.nb2, b1 = z / g.nb2;
    const float* A = g.A + (long long)b1 * g.sa_b1 + (long long)b2 * g.sa_b2;
    const float* B = g.B + (long long)b1 * g.sb_b1 + (long long)b2 * g.sb_b2;
    float* C = g.C + (long long)b1 * g.sc_b1 + (long long)b2 * g.sc_b2;
    const int m0 = by * BT, n0 = bx * BN;
    const int kchunk = (((g.K + g.ksplit - 1) / g.ksplit) + HK - 1) / HK * HK;
    const int kbeg = ks * kchunk, kend = min(g.K, kbeg + kchunk);
    const int nsteps = kbeg < kend ? (kend - kbeg + HK - 1) / HK : 0;
    // row-major C (unit column stride, 16-byte aligned rows), no split-K atomics: lane = row m, registers = 4 consecutive columns
    constexpr bool tr = TR;
    f32x4 acc[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4)(0.0f);
    float ra[2][BT / 8], rb[2][BN / 8];
    auto load = [&](auto q, int t) {
        constexpr int Q = decltype(q)::value;
        hg_load<MA, BT>(A, g.sa_m, g.sa_k, m0, g.M, kbeg + t * HK, kend, ra[Q]);
        hg_load<MB, BN>(B, g.sb_n, g.sb_k, n0, g.N, kbeg + t * HK, kend, rb[Q]);
    };
    auto store = [&](auto q, int buf) {
        constexpr int Q = decltype(q)::value;
        hg_store<MA, BT>(lds[buf], ra[Q]);
        hg_store<MB, BN>(lds[buf] + BT * HLD, rb[Q]);
    };
    auto step = [&](auto q, int t) {  // registers Q hold step t + 1 after this (loaded two steps ago), and are refilled with t + 2 ... see below
        constexpr int Q = decltype(q)::value;
        using QN = std::integral_constant<int, 1 - Q>;
        // registers Q held step t (already in LDS): refill them with step t + 2; registers 1 - Q hold step t + 1
        if (t + 2 < nsteps) load(q, t + 2);
        const bf16* At = lds[t & 1];
        const bf16* Bt = At + BT * HLD;
        Frag<bf16> av[4], bv[NTW];
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const Frag<bf16>*>(At + (wm * 64 + i * 16 + li) * HLD + 8 * lg);
#pragma unroll
        for (int j = 0; j < NTW; ++j) bv[j] = *reinterpret_cast<const Frag<bf16>*>(Bt + (wn * (BN / 2) + j * 16 + li) * HLD + 8 * lg);
        if constexpr (tr) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[nt].v, av[mt].v, acc[mt][nt], 0, 0, 0);
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mt].v, bv[nt].v, acc[mt][nt], 0, 0, 0);
        }
        if (t + 1 < nsteps) store(QN{}, (t + 1) & 1);
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if (nsteps > 0) {
        load(I0{}, 0);
        if (nsteps > 1) load(I1{}, 1);
        store(I0{}, 0);
        __syncthreads();
        for (int t = 0; t < nsteps; t += 2) {
            step(I0{}, t);
            if (t + 1 < nsteps) step(I1{}, t + 1);
        }
    }
    if constexpr (tr) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + wm * 64 + 16 * mt + li;
            if (m >= g.M) continue;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int n = n0 + wn * (BN / 2) + 16 * nt + 4 * lg;
                if (n >= g.N) continue;
                float* c = C + (long long)m * g.sc_m + n;
                f32x4 v = g.alpha * acc[mt][nt];
                if (g.bias && ks == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < g.N) v[r] += g.bias[n + r];
                }
                if (g.ksplit > 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < g.N) atomicAdd(c + r, v[r]);
                } else if (n + 4 <= g.N) {
                    if (g.accumulate) v += *reinterpret_cast<const f32x4*>(c);
                    *reinterpret_cast<f32x4*>(c) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < g.N) c[r] = g.accumulate ? c[r] + v[r] : v[r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n = n0 + wn * (BN / 2) + 16 * nt + li;
            if (n >= g.N) continue;
            const float bias = (g.bias && ks == 0) ? g.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + 16 * mt + 4 * lg + r;
                if (m >= g.M) continue;
                float* c = C + (long long)m * g.sc_m + (long long)n * g.sc_n;
                const float v = g.alpha * acc[mt][nt][r] + bias;
                if (g.ksplit > 1)
                    atomicAdd(c, v);
                else if (g.accumulate)
                    *c += v;
                else
                    *c = v;
            }
        }
}

// row-major C (unit column stride, 16-byte aligned rows) and no split-K atomics: result tiles are computed transposed, 16-byte stores
bool bgemm_row_major_out(const SrBgemm& g) {
    return g.ksplit == 1 && g.sc_n == 1 && (g.sc_m & 3) == 0 && ((g.sc_b1 | g.sc_b2) & 3) == 0 && (reinterpret_cast<size_t>(g.C) & 15) == 0;
}

template <int MA, int MB>
void launch_bf16(const SrBgemm& g, long long nz, hipStream_t st) {
    // BN = 64 where it pads N less or where 128-wide tiles leave the chip under-filled
    const int force = 0;  // (experiments forced the column tile here)
    const long long t128 = (long long)((g.N + 127) / 128) * ((g.M + BT - 1) / BT) * nz;
    const bool pad64 = ((g.N + 63) / 64) * 64 < ((g.N + 127) / 128) * 128;
    const bool bn64 = force ? force == 64 : (pad64 || t128 < 1024);
    // measured in the HAT / SwinIR training steps: the remap pays for plain GEMMs (one A panel per m-tile: -20..25 % per launch), not for
    // batched / split-K launches, which keep the x-fastest order.  SR_BGEMM_NATURAL=1 switches it off everywhere (A/B knob).
    const int natural = 2;
    const int nat = natural == 1 || (natural == 2 && nz > 1);
    const BgemmGrid g64{(g.N + 63) / 64, (g.M + BT - 1) / BT, (int)nz, nat}, g128{(g.N + 127) / 128, (g.M + BT - 1) / BT, (int)nz, nat};
    if (bgemm_row_major_out(g)) {
        if (bn64)
            hipLaunchKernelGGL((sr_bgemm_bf16_kernel<MA, MB, 64, true>), dim3(g64.blocks()), dim3(256), 0, st, g, g64);
        else
            hipLaunchKernelGGL((sr_bgemm_bf16_kernel<MA, MB, 128, true>), dim3(g128.blocks()), dim3(256), 0, st, g, g128);
    } else {
        if (bn64)
            hipLaunchKernelGGL((sr_bgemm_bf16_kernel<MA, MB, 64, false>), dim3(g64.blocks()), dim3(256), 0, st, g, g64);
        else
            hipLaunchKernelGGL((sr_bgemm_bf16_kernel<MA, MB, 128, false>), dim3(g128.blocks()), dim3(256), 0, st, g, g128);
    }
}

// ----------------------------------------------------------------------------- im2col / col2im, column order (tap, c)
// col[m][tap*C + c] = x[b, y + tap/3 - 1, x + tap%3 - 1, c]: every read and write is contiguous over c (the caller permutes the small
// OIHW weight to [O][tap][C] instead of making the big buffer follow OIHW's (c, tap) order)
template <int V>  // V channels per thread (4: C % 4 == 0 and unit channel stride)
__global__ void sr_im2col3x3_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int H, int W, int C, long long sb, long long sy, long long sx, long long sc) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int CV = C / V;
    const long long total = (long long)B * H * W * 9 * CV;
    if (idx >= total) return;
    const int c = (int)(idx % CV) * V;
    const int t = (int)((idx / CV) % 9);
    const long long m = idx / (9LL * CV);
    const int xx = (int)(m % W);
    const int yy = (int)((m / W) % H);
    const long long b = m / ((long long)W * H);
    const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
    const bool in = (unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W;
    float* dst = col + m * (9LL * C) + t * C + c;
    if (V == 4) {
        *reinterpret_cast<f32x4u*>(dst) = in ? *reinterpret_cast<const f32x4u*>(x + b * sb + y2 * sy + x2 * sx + c) : (f32x4u)(0.0f);
    } else {
        dst[0] = in ? x[b * sb + y2 * sy + x2 * sx + c * sc] : 0.0f;
    }
}
// dx[b,y,x,c] = sum_tap dcol[(b, y - dy, x - dx)][tap*C + c]  (the adjoint of im2col written as a gather)
__global__ void sr_col2im3x3_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int H, int W, int C) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * H * W * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long m = idx / C;
    const int xx = (int)(m % W);
    const int yy = (int)((m / W) % H);
    const long long b = m / ((long long)W * H);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int y2 = yy - (t / 3 - 1), x2 = xx - (t % 3 - 1);  // the pixel whose tap t looked at (yy, xx)
        if ((unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W) s += dcol[((b * H + y2) * W + x2) * (9LL * C) + t * C + c];
    }
    dx[idx] = s;
}

// ----------------------------------------------------------------------------- softmax rows with bias and shift mask
// rows r = ((bw * heads + h) * Nq + i); P = softmax(S + bias[h,i,:] + mask[bw % nW, i, :]) in place (swinir.py:92-100, hat.py:97-106)
__global__ void sr_softmax_fwd_kernel(float* __restrict__ S, const float* __restrict__ bias, const float* __restrict__ mask, long long rows, int heads, int Nq, int Nk, int nW) {
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(r % Nq);
    const int h = (int)((r / Nq) % heads);
    const long long bw = r / ((long long)Nq * heads);
    float* s = S + r * Nk;
    const float* bp = bias ? bias + ((long long)h * Nq + i) * Nk : nullptr;
    const float* mp = mask ? mask + ((bw % nW) * Nq + i) * Nk : nullptr;
    float mx = -3.0e38f;
    for (int j = lane; j < Nk; j += 64) {
        float v = s[j];
        if (bp) v += bp[j];
        if (mp) v += mp[j];
        s[j] = v;
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = lane; j < Nk; j += 64) {
        const float e = expf(s[j] - mx);
        s[j] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    for (int j = lane; j < Nk; j += 64) s[j] *= inv;
}
// dS = P * (dP - sum_j dP_j P_j), written over dP
__global__ void sr_softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, long long rows, int Nk) {
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = P + r * Nk;
    float* d = dP + r * Nk;
    float dot = 0.f;
    for (int j = lane; j < Nk; j += 64) dot += p[j] * d[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    for (int j = lane; j < Nk; j += 64) d[j] = p[j] * (d[j] - dot);
}

// Register-resident forms (Nk <= 64 NV): one read and one write of the row instead of three each; same arithmetic, same summation order.
template <int NV>
__global__ void sr_softmax_fwd_reg_kernel(float* __restrict__ S, const float* __restrict__ bias, const float* __restrict__ mask, long long rows, int heads, int Nq, int Nk, int nW) {
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(r % Nq);
    const int h = (int)((r / Nq) % heads);
    const long long bw = r / ((long long)Nq * heads);
    float* s = S + r * Nk;
    const float* bp = bias ? bias + ((long long)h * Nq + i) * Nk : nullptr;
    const float* mp = mask ? mask + ((bw % nW) * Nq + i) * Nk : nullptr;
    float v[NV];
    float mx = -3.0e38f;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int j = lane + 64 * q;
        v[q] = -3.0e38f;
        if (j < Nk) {
            float t = s[j];
            if (bp) t += bp[j];
            if (mp) t += mp[j];
            v[q] = t;
            mx = fmaxf(mx, t);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q)
        if (lane + 64 * q < Nk) {
            v[q] = expf(v[q] - mx);
            sum += v[q];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int q = 0; q < NV; ++q)
        if (lane + 64 * q < Nk) s[lane + 64 * q] = v[q] * inv;
}
template <int NV>
__global__ void sr_softmax_bwd_reg_kernel(const float* __restrict__ P, float* __restrict__ dP, long long rows, int Nk) {
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = P + r * Nk;
    float* d = dP + r * Nk;
    float pv[NV], dv[NV];
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int j = lane + 64 * q;
        pv[q] = dv[q] = 0.f;
        if (j < Nk) {
            pv[q] = p[j];
            dv[q] = d[j];
            dot += pv[q] * dv[q];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
#pragma unroll
    for (int q = 0; q < NV; ++q)
        if (lane + 64 * q < Nk) d[lane + 64 * q] = pv[q] * (dv[q] - dot);
}

// ----------------------------------------------------------------------------- LayerNorm forward (saving mean / rstd) and backward
__global__ void sr_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ stats,
                                 long long M, int C, float eps) {
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= M) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + r * C;
    float s1 = 0.f;
    for (int c = lane; c < C; c += 64) s1 += xr[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
    const float mean = s1 / C;
    float s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = xr[c] - mean;
        s2 += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rstd = rsqrtf(s2 / C + eps);
    if (lane == 0) {
        stats[2 * r] = mean;
        stats[2 * r + 1] = rstd;
    }
    for (int c = lane; c < C; c += 64) y[r * C + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; dgamma += sum_rows dy * xhat; dbeta += sum_rows dy
// One workgroup = 16 rows, 4 per wave, all of a wave's rows in flight at once (the first version walked 16 rows per wave one after the
// other: 1,024 waves of serial load -> shuffle -> reload chains, 67 us for 35 MB); the workgroup's partial dgamma / dbeta are summed
// through LDS, so the atomic count stays one per column per 16 rows.
constexpr int LN_ROWS_PER_WAVE = 4, LN_ROWS_PER_WG = 16, LN_MAX_COLS_PER_LANE = 8;  // C <= 512
__global__ __launch_bounds__(256) void sr_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ dy,
                                                        float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, long long M, int C) {
    __shared__ float red[2][4][64 * LN_MAX_COLS_PER_LANE];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long r0 = (long long)blockIdx.x * LN_ROWS_PER_WG + w * LN_ROWS_PER_WAVE;
    float xh[LN_ROWS_PER_WAVE][LN_MAX_COLS_PER_LANE], dv[LN_ROWS_PER_WAVE][LN_MAX_COLS_PER_LANE], gm[LN_MAX_COLS_PER_LANE];
    float mean[LN_ROWS_PER_WAVE], rstd[LN_ROWS_PER_WAVE];
#pragma unroll
    for (int q = 0; q < LN_MAX_COLS_PER_LANE; ++q) gm[q] = (lane + 64 * q < C) ? gamma[lane + 64 * q] : 0.f;
#pragma unroll
    for (int rr = 0; rr < LN_ROWS_PER_WAVE; ++rr) {
        const long long r = r0 + rr;
        const bool ok = r < M;
        mean[rr] = ok ? stats[2 * r] : 0.f;
        rstd[rr] = ok ? stats[2 * r + 1] : 0.f;
#pragma unroll
        for (int q = 0; q < LN_MAX_COLS_PER_LANE; ++q) {
            const int c = lane + 64 * q;
            const bool in = ok && c < C;
            xh[rr][q] = in ? x[r * C + c] : 0.f;
            dv[rr][q] = in ? dy[r * C + c] : 0.f;
        }
    }
    float pg[LN_MAX_COLS_PER_LANE], pb[LN_MAX_COLS_PER_LANE];
#pragma unroll
    for (int q = 0; q < LN_MAX_COLS_PER_LANE; ++q) pg[q] = pb[q] = 0.f;
#pragma unroll
    for (int rr = 0; rr < LN_ROWS_PER_WAVE; ++rr) {
        const long long r = r0 + rr;
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int q = 0; q < LN_MAX_COLS_PER_LANE; ++q) {
            const int c = lane + 64 * q;
            if (c < C) {
                xh[rr][q] = (xh[rr][q] - mean[rr]) * rstd[rr];
                const float g = dv[rr][q] * gm[q];
                m1 += g;
                m2 += g * xh[rr][q];
                pg[q] += dv[rr][q] * xh[rr][q];
                pb[q] += dv[rr][q];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m1 += __shfl_xor(m1, o, 64);
            m2 += __shfl_xor(m2, o, 64);
        }
        m1 /= C;
        m2 /= C;
        if (r < M) {
#pragma unroll
            for (int q = 0; q < LN_MAX_COLS_PER_LANE; ++q) {
                const int c = lane + 64 * q;
                if (c < C) dx[r * C + c] = rstd[rr] * (dv[rr][q] * gm[q] - m1 - xh[rr][q] * m2);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < LN_MAX_COLS_PER_LANE; ++q) {
        red[0][w][lane + 64 * q] = pg[q];
        red[1][w][lane + 64 * q] = pb[q];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(dgamma + c, ((red[0][0][c] + red[0][1][c]) + red[0][2][c]) + red[0][3][c]);
        atomicAdd(dbeta + c, ((red[1][0][c] + red[1][1][c]) + red[1][2][c]) + red[1][3][c]);
    }
}

// out[b][c] += alpha * sum_{p in chunk} x[b][p][c]   (bias gradients, pooling; rows p, columns c contiguous)
constexpr int COLSUM_ROWS = 512;
__global__ __launch_bounds__(256) void sr_colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long long P, int C, float alpha) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const long long b = blockIdx.z;
    const long long p0 = (long long)blockIdx.x * COLSUM_ROWS, p1 = min(P, p0 + COLSUM_ROWS);
    const float* xb = x + b * P * C;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        long long p = p0 + rg;
        for (; p + 4 < p1; p += 8) {
            s0 += xb[p * C + c];
            s1 += xb[(p + 4) * C + c];
        }
        if (p < p1) s0 += xb[p * C + c];
    }
    red[rg][cl] = s0 + s1;
    __syncthreads();
    if (rg == 0 && c < C) atomicAdd(out + b * C + c, alpha * (red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]));
}
// C % 4 == 0, 16-byte aligned rows: a thread owns 4 consecutive columns (one 16-byte load per row) of every R-th row of its workgroup's
// row block; the 256 threads are laid out as R rows x QB column quads (QB <= 64: all lanes busy also for C = 60 or 180), 16 independent loads
// in flight per thread.  rows_wg is chosen by the launcher: every workgroup of a column block ends in atomicAdds on the SAME addresses, which
// serialise in L2, so the row blocks are few and long.
__global__ __launch_bounds__(256) void sr_colsum4_kernel(const float* __restrict__ x, float* __restrict__ out, long long P, int C, float alpha, int QB, int rows_wg) {
    __shared__ f32x4 red[256];
    const int q = C >> 2;
    const int R = 256 / QB;
    const int cq = threadIdx.x % QB, r = threadIdx.x / QB;
    const int quad = blockIdx.y * QB + cq;
    const long long b = blockIdx.z;
    const long long p0 = (long long)blockIdx.x * rows_wg, p1 = min(P, p0 + rows_wg);
    const float* xb = x + b * P * C + 4 * quad;
    f32x4 acc[4] = {(f32x4)(0.0f), (f32x4)(0.0f), (f32x4)(0.0f), (f32x4)(0.0f)};
    const bool live = r < R && quad < q;
    if (live) {
        for (long long pc = p0 + r; pc < p1; pc += 16LL * R) {
            f32x4 v[16];  // unconditional loads from clamped addresses: a branch per load would serialise them (load, wait, add, next)
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const f32x4*>(xb + min(pc + (long long)k * R, p1 - 1) * C);
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 3] += (pc + (long long)k * R < p1) ? v[k] : (f32x4)(0.0f);
        }
    }
    red[threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (r == 0 && quad < q) {
        f32x4 t = red[cq];
        for (int i = 1; i < R; ++i) t += red[i * QB + cq];
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(out + b * C + 4 * quad + j, alpha * t[j]);
    }
}
// out[i] = sum_b x[b][i]  (relative-position-bias gradient: sum of dS over windows; deterministic, no atomics)
__global__ void sr_batch_sum_kernel(const float* __restrict__ x, float* __restrict__ out, long long nb, long long n, long long stride_b) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (long long b = 0; b < nb; ++b) s += x[b * stride_b + i];
    out[i] = s;
}
// n, stride_b multiples of 4 and 16-byte aligned pointers: 4 elements per thread, 8 batch entries in flight; same summation order (b ascending)
__global__ void sr_batch_sum4_kernel(const float* __restrict__ x, float* __restrict__ out, long long nb, long long n4, long long stride_b) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float* p = x + 4 * i;
    f32x4 s = (f32x4)(0.0f);
    long long b = 0;
    for (; b + 8 <= nb; b += 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(p + (b + k) * stride_b);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; b < nb; ++b) s += *reinterpret_cast<const f32x4*>(p + b * stride_b);
    *reinterpret_cast<f32x4*>(out + 4 * i) = s;
}

// ----------------------------------------------------------------------------- flat elementwise ops
enum {
    EW_GELU_FWD = 0, EW_GELU_BWD = 1, EW_RELU_FWD = 2, EW_RELU_BWD = 3, EW_LRELU_FWD = 4, EW_LRELU_BWD = 5, EW_AXPBY = 6, EW_MUL = 7,
    EW_SIGMOID_FWD = 8, EW_SIGMOID_BWD = 9, EW_SCALE_SAMPLE = 10, EW_MUL_BC = 11, EW_BCAST_BC = 12, EW_AFFINE_C = 13,
};
// x, y: inputs (y optional), s: small side input; inner = elements per sample (SCALE_SAMPLE), P*C per sample and C (MUL_BC, BCAST_BC), C (AFFINE_C)
__global__ void sr_eltwise_kernel(int op, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ s, float* __restrict__ out, long long n, long long inner,
                                  int C, float a, float b) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v;
    switch (op) {
        case EW_GELU_FWD: v = 0.5f * x[i] * (1.0f + erff(x[i] * 0.70710678118654752440f)); break;
        case EW_GELU_BWD: {  // x = dy, y = forward input
            const float t = y[i];
            v = x[i] * (0.5f * (1.0f + erff(t * 0.70710678118654752440f)) + t * 0.3989422804014327f * expf(-0.5f * t * t));
            break;
        }
        case EW_RELU_FWD: v = x[i] > 0.f ? x[i] : 0.f; break;
        case EW_RELU_BWD: v = y[i] > 0.f ? x[i] : 0.f; break;  // x = dy, y = forward output (or input: same sign test)
        case EW_LRELU_FWD: v = x[i] > 0.f ? x[i] : a * x[i]; break;
        case EW_LRELU_BWD: v = y[i] > 0.f ? x[i] : a * x[i]; break;
        case EW_AXPBY: v = a * x[i] + (y ? b * y[i] : 0.f); break;
        case EW_MUL: v = x[i] * y[i]; break;
        case EW_SIGMOID_FWD: v = 1.0f / (1.0f + expf(-x[i])); break;
        case EW_SIGMOID_BWD: v = x[i] * y[i] * (1.0f - y[i]); break;  // x = dy, y = sigmoid output
        case EW_SCALE_SAMPLE: v = x[i] * s[i / inner]; break;
        case EW_MUL_BC: v = x[i] * s[(i / inner) * C + (i % C)]; break;       // x [B,P,C] * s [B,C]
        case EW_BCAST_BC: v = a * s[(i / inner) * C + (i % C)]; break;        // s [B,C] -> [B,P,C]
        case EW_AFFINE_C: v = x[i] * s[i % C] + (y ? y[i % C] : 0.f); break;  // per-channel scale (s) and shift (y)
        default: v = 0.f;
    }
    out[i] = v;
}

// ----------------------------------------------------------------------------- index-map copies (gathers in both directions)
// window_partition(roll(x, -shift)) : tokens [B*nW, ws*ws, C] <-> image [B,H,W,C]   (swinir.py:154-158,164-168; common.py:236-247)
__global__ void sr_window_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int C, int ws, int shift, int to_windows) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * H * W * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long t = idx / C;  // window-order token index
    const int N = ws * ws, nwx = W / ws, nW = (H / ws) * nwx;
    const int tok = (int)(t % N);
    const int win = (int)((t / N) % nW);
    const long long b = t / ((long long)N * nW);
    int y = (win / nwx) * ws + tok / ws + shift, x = (win % nwx) * ws + tok % ws + shift;
    if (y >= H) y -= H;
    if (x >= W) x -= W;
    const long long img = ((b * H + y) * W + x) * C + c;
    if (to_windows)
        dst[idx] = src[img];
    else
        dst[img] = src[idx];
}
// OCAB key / value windows: nn.Unfold(kernel = wse, stride = ws, padding = (wse - ws) / 2), zero padded (hat.py:217-221,255-263)
__global__ void sr_oca_unfold_kernel(const float* __restrict__ img, float* __restrict__ win, int B, int H, int W, int C, int ws, int wse) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nwx = W / ws, nW = (H / ws) * nwx, Nk = wse * wse, pad = (wse - ws) / 2;
    const long long total = (long long)B * nW * Nk * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long t = idx / C;
    const int tok = (int)(t % Nk);
    const int wi = (int)((t / Nk) % nW);
    const long long b = t / ((long long)Nk * nW);
    const int y = (wi / nwx) * ws - pad + tok / wse, x = (wi % nwx) * ws - pad + tok % wse;
    win[idx] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? img[((b * H + y) * W + x) * C + c] : 0.0f;
}
// adjoint: every pixel sums the (up to 4) window slots that looked at it
__global__ void sr_oca_fold_kernel(const float* __restrict__ dwin, float* __restrict__ dimg, int B, int H, int W, int C, int ws, int wse) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * H * W * C;
    if (idx >= total) return;
    const int nwx = W / ws, nwy = H / ws, Nk = wse * wse, pad = (wse - ws) / 2;
    const int c = (int)(idx % C);
    const long long m = idx / C;
    const int x = (int)(m % W), y = (int)((m / W) % H);
    const long long b = m / ((long long)W * H);
    float s = 0.f;
    for (int wy = 0; wy < nwy; ++wy) {
        const int i = y - (wy * ws - pad);
        if (i < 0 || i >= wse) continue;
        for (int wx = 0; wx < nwx; ++wx) {
            const int j = x - (wx * ws - pad);
            if (j < 0 || j >= wse) continue;
            s += dwin[(((b * nwy + wy) * nwx + wx) * Nk + i * wse + j) * (long long)C + c];
        }
    }
    dimg[idx] = s;
}
// nn.PixelShuffle(r) on NHWC: out[b, y*r+i, x*r+j, c] = in[b, y, x, c*r*r + i*r + j]  (common.py:129,133,136)
__global__ void sr_pixel_shuffle_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int C, int r, int forward) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // index into the shuffled tensor [B, H*r, W*r, C]
    const long long total = (long long)B * H * r * W * r * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long t = idx / C;
    const int X = (int)(t % (W * r)), Y = (int)((t / (W * r)) % (H * r));
    const long long b = t / ((long long)W * r * H * r);
    const long long in = ((b * H + Y / r) * W + X / r) * ((long long)C * r * r) + (long long)c * r * r + (Y % r) * r + (X % r);
    if (forward)
        dst[idx] = src[in];
    else
        dst[in] = src[idx];
}
// relative-position bias: bias[h][ij] = table[rpi[ij] (negative wraps)][h]; adjoint scatter-adds (few adders per address)
__global__ void sr_bias_gather_kernel(const float* __restrict__ table, const long long* __restrict__ rpi, float* __restrict__ bias, float* __restrict__ dtable, int T, int heads, long long NN,
                                      int forward) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NN * heads) return;
    const int h = (int)(idx / NN);
    const long long ij = idx % NN;
    long long t = rpi[ij];
    if (t < 0) t += T;
    if (forward)
        bias[idx] = table[t * heads + h];
    else
        atomicAdd(dtable + t * heads + h, bias[idx]);
}
// the model's output: y [B,Hs,Ws,C] NHWC -> out [B,C,Ho,Wo] NCHW cropped, out = y * scale[c] + shift[c]; adjoint zero-fills the cropped border
__global__ void sr_nhwc_out_kernel(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ scale, const float* __restrict__ shift, int B, int Hs, int Ws, int C,
                                   int Ho, int Wo, int forward) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (forward) {
        const long long total = (long long)B * C * Ho * Wo;
        if (idx >= total) return;
        const int x = (int)(idx % Wo), y = (int)((idx / Wo) % Ho), c = (int)((idx / ((long long)Wo * Ho)) % C);
        const long long b = idx / ((long long)Wo * Ho * C);
        dst[idx] = src[((b * Hs + y) * Ws + x) * C + c] * scale[c] + shift[c];
    } else {
        const long long total = (long long)B * Hs * Ws * C;
        if (idx >= total) return;
        const int c = (int)(idx % C), x = (int)((idx / C) % Ws), y = (int)((idx / ((long long)C * Ws)) % Hs);
        const long long b = idx / ((long long)C * Ws * Hs);
        dst[idx] = (y < Ho && x < Wo) ? src[((b * C + c) * Ho + y) * (long long)Wo + x] * scale[c] : 0.0f;
    }
}

// dst[r, off_d + c] = src[r, off_s + c], c < n  (channel concat / split as strided column copies)
__global__ void sr_copy_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, long long rows, int n, int ld_s, int off_s, int ld_d, int off_d, int accumulate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * n) return;
    const long long r = i / n;
    const int c = (int)(i - r * n);
    float* d = dst + r * ld_d + off_d + c;
    const float v = src[r * ld_s + off_s + c];
    *d = accumulate ? *d + v : v;
}

// nn.Conv3d(1, 1, 3, padding=1) over the (C, H, W) volume of an NHWC tensor (HAN's CSAM, han.py:37-53): 27 taps w[dc][dy][dx].
// flip = 1 evaluates the adjoint (data gradient): taps mirrored.
__global__ void sr_conv3d27_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out, int B, int H, int W, int C, int flip) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * H * W * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long m = idx / C;
    const int xx = (int)(m % W), yy = (int)((m / W) % H);
    const long long b = m / ((long long)W * H);
    float s = (bias && !flip) ? bias[0] : 0.f;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        const int dc = t / 9 - 1, dy = (t / 3) % 3 - 1, dx = t % 3 - 1;
        const int c2 = c + dc, y2 = yy + dy, x2 = xx + dx;
        if ((unsigned)c2 < (unsigned)C && (unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W) s += w[flip ? 26 - t : t] * x[((b * H + y2) * W + x2) * C + c2];
    }
    out[idx] = s;
}
// dw[t] += sum x[shifted by tap t] * dy ; db += sum dy   (28 block-reduced atomics per workgroup)
__global__ void sr_conv3d27_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, float* __restrict__ db, int B, int H, int W, int C) {
    __shared__ float red[4][28];
    const long long total = (long long)B * H * W * C;
    float acc[28];
#pragma unroll
    for (int t = 0; t < 28; ++t) acc[t] = 0.f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const long long m = idx / C;
        const int xx = (int)(m % W), yy = (int)((m / W) % H);
        const long long b = m / ((long long)W * H);
        const float d = dy[idx];
        acc[27] += d;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int c2 = c + t / 9 - 1, y2 = yy + (t / 3) % 3 - 1, x2 = xx + t % 3 - 1;
            if ((unsigned)c2 < (unsigned)C && (unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W) acc[t] += d * x[((b * H + y2) * W + x2) * C + c2];
        }
    }
#pragma unroll
    for (int t = 0; t < 28; ++t) {
        float v = acc[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 28) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(threadIdx.x < 27 ? dw + threadIdx.x : db, v);
    }
}

inline dim3 flat_grid(long long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int sr_bgemm(const SrBgemm* p, void* stream) {
    SR_REQUIRE(p && p->A && p->B && p->C, "sr_bgemm: null pointer");
    SrBgemm g = *p;
    SR_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.nb1 > 0 && g.nb2 > 0 && g.ksplit > 0, "sr_bgemm: bad sizes M=%d N=%d K=%d nb=%dx%d ksplit=%d", g.M, g.N, g.K, g.nb1, g.nb2, g.ksplit);
    const long long nz = (long long)g.nb1 * g.nb2 * g.ksplit;
    SR_REQUIRE(nz < (1ll << 31), "sr_bgemm: too many batches x ksplit (%lld)", nz);
    const bool no_tiled = false;  // (true: the direct-from-global form everywhere)
    if (g.M < 96 && g.N >= 96 && !g.bias) {
        // short-and-wide (weight gradients of convs with few output channels: M = 3 or 60, N = 9 Cin): run the transposed problem
        // C^T = B^T A^T so that the long axis is the 128-row tile axis -- same products, same k order
        std::swap(g.A, g.B);
        std::swap(g.M, g.N);
        const long long am = g.sa_m, ak = g.sa_k;
        g.sa_m = g.sb_n, g.sa_k = g.sb_k, g.sb_n = am, g.sb_k = ak;
        std::swap(g.sa_b1, g.sb_b1);
        std::swap(g.sa_b2, g.sb_b2);
        std::swap(g.sc_m, g.sc_n);
    }
    // bf16 operands: any N and K once M fills a 128-row tile (N = 3: the RGB tail conv over im2col -- memory-bound, the tile's idle
    // columns cost nothing; K = 3: its data gradient)
    if (g.compute_dtype == SR_BF16 && g.M >= 96 && (g.M + BT - 1) / BT <= 65535) {
        // staging mode per operand: 0 = contraction axis contiguous, 1 = row axis contiguous with 16-byte aligned k-rows, 2 = anything else
        auto vec_rows = [](const float* p, long long s_row, long long s_k, long long sb1, long long sb2) {
            return s_row == 1 && (s_k & 3) == 0 && (sb1 & 3) == 0 && (sb2 & 3) == 0 && (reinterpret_cast<size_t>(p) & 15) == 0;
        };
        const int ma = g.sa_k == 1 ? 0 : (vec_rows(g.A, g.sa_m, g.sa_k, g.sa_b1, g.sa_b2) ? 1 : 2);
        const int mb = g.sb_k == 1 ? 0 : (vec_rows(g.B, g.sb_n, g.sb_k, g.sb_b1, g.sb_b2) ? 1 : 2);
        if (ma == 2 || mb == 2) {  // rare: generic strides on either side -> both operands through the scalar staging, 128 x 128 tiles
            const BgemmGrid gg{(g.N + 127) / 128, (g.M + BT - 1) / BT, (int)nz, nz > 1};
            if (ma == 0)
                hipLaunchKernelGGL((sr_bgemm_bf16_kernel<0, 2, 128, false>), dim3(gg.blocks()), dim3(256), 0, ST, g, gg);
            else if (mb == 0)
                hipLaunchKernelGGL((sr_bgemm_bf16_kernel<2, 0, 128, false>), dim3(gg.blocks()), dim3(256), 0, ST, g, gg);
            else
                hipLaunchKernelGGL((sr_bgemm_bf16_kernel<2, 2, 128, false>), dim3(gg.blocks()), dim3(256), 0, ST, g, gg);
        } else if (ma == 0 && mb == 0)
            launch_bf16<0, 0>(g, nz, ST);
        else if (ma == 0)
            launch_bf16<0, 1>(g, nz, ST);
        else if (mb == 0)
            launch_bf16<1, 0>(g, nz, ST);
        else
            launch_bf16<1, 1>(g, nz, ST);
        SR_CHECK_LAUNCH("sr_bgemm");
        return SR_OK;
    }
    // the fp32 kernels below put (batch x ksplit) on grid.z (the bf16 kernels above use a flat x grid and have no such limit:
    // batch 64 of 128 x 128 patches at ws 8 / 6 heads is nb = 16,384 x 6); autograd.bgemm chunks the first batch level beyond it
    SR_REQUIRE(nz <= 65535 && (g.M + 63) / 64 <= 65535, "sr_bgemm: grid too large for the fp32 kernels (batches x ksplit = %lld > 65535)", nz);
    if (g.M >= 96 && g.N >= 24 && g.K >= 16 && !no_tiled && (g.M + BT - 1) / BT <= 65535) {
        // 128 x 128 LDS-tiled kernel; the staging mode of each operand follows which of its axes is contiguous
        const int ma = g.sa_k == 1 ? 0 : (g.sa_m == 1 ? 1 : 2), mb = g.sb_k == 1 ? 0 : (g.sb_n == 1 ? 1 : 2);
        const dim3 grid((g.N + BT - 1) / BT, (g.M + BT - 1) / BT, (unsigned)nz);
        if (ma == 0)
            launch_tiled_b<0>(g, mb, grid, ST);
        else if (ma == 1)
            launch_tiled_b<1>(g, mb, grid, ST);
        else
            launch_tiled_b<2>(g, mb, grid, ST);
    } else {
        hipLaunchKernelGGL(sr_bgemm_kernel, dim3((g.N + 63) / 64, (g.M + 63) / 64, (unsigned)nz), dim3(256), 0, ST, g);
    }
    SR_CHECK_LAUNCH("sr_bgemm");
    return SR_OK;
}

extern "C" int sr_im2col3x3(const float* x, float* col, int B, int H, int W, int C, long long sb, long long sy, long long sx, long long sc, void* stream) {
    SR_REQUIRE(x && col && B > 0 && H > 0 && W > 0 && C > 0, "sr_im2col3x3: bad arguments");
    if (C % 4 == 0 && sc == 1)
        hipLaunchKernelGGL(sr_im2col3x3_kernel<4>, flat_grid((long long)B * H * W * 9 * (C / 4)), dim3(256), 0, ST, x, col, B, H, W, C, sb, sy, sx, sc);
    else
        hipLaunchKernelGGL(sr_im2col3x3_kernel<1>, flat_grid((long long)B * H * W * 9 * C), dim3(256), 0, ST, x, col, B, H, W, C, sb, sy, sx, sc);
    SR_CHECK_LAUNCH("sr_im2col3x3");
    return SR_OK;
}
extern "C" int sr_col2im3x3(const float* dcol, float* dx, int B, int H, int W, int C, void* stream) {
    SR_REQUIRE(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0, "sr_col2im3x3: bad arguments");
    hipLaunchKernelGGL(sr_col2im3x3_kernel, flat_grid((long long)B * H * W * C), dim3(256), 0, ST, dcol, dx, B, H, W, C);
    SR_CHECK_LAUNCH("sr_col2im3x3");
    return SR_OK;
}
extern "C" int sr_softmax_fwd(float* S, const float* bias, const float* mask, long long rows, int heads, int Nq, int Nk, int nW, void* stream) {
    SR_REQUIRE(S && rows > 0 && heads > 0 && Nq > 0 && Nk > 0 && (!mask || nW > 0), "sr_softmax_fwd: bad arguments");
    const dim3 grid((unsigned)((rows + 3) / 4));
    const int nw = nW > 0 ? nW : 1;
    if (Nk <= 256)
        hipLaunchKernelGGL(sr_softmax_fwd_reg_kernel<4>, grid, dim3(256), 0, ST, S, bias, mask, rows, heads, Nq, Nk, nw);
    else if (Nk <= 640)
        hipLaunchKernelGGL(sr_softmax_fwd_reg_kernel<10>, grid, dim3(256), 0, ST, S, bias, mask, rows, heads, Nq, Nk, nw);
    else
        hipLaunchKernelGGL(sr_softmax_fwd_kernel, grid, dim3(256), 0, ST, S, bias, mask, rows, heads, Nq, Nk, nw);
    SR_CHECK_LAUNCH("sr_softmax_fwd");
    return SR_OK;
}
extern "C" int sr_softmax_bwd(const float* P, float* dP, long long rows, int Nk, void* stream) {
    SR_REQUIRE(P && dP && rows > 0 && Nk > 0, "sr_softmax_bwd: bad arguments");
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (Nk <= 256)
        hipLaunchKernelGGL(sr_softmax_bwd_reg_kernel<4>, grid, dim3(256), 0, ST, P, dP, rows, Nk);
    else if (Nk <= 640)
        hipLaunchKernelGGL(sr_softmax_bwd_reg_kernel<10>, grid, dim3(256), 0, ST, P, dP, rows, Nk);
    else
        hipLaunchKernelGGL(sr_softmax_bwd_kernel, grid, dim3(256), 0, ST, P, dP, rows, Nk);
    SR_CHECK_LAUNCH("sr_softmax_bwd");
    return SR_OK;
}
extern "C" int sr_layernorm_fwd_train(const float* x, const float* gamma, const float* beta, float* y, float* stats, long long M, int C, float eps, void* stream) {
    SR_REQUIRE(x && gamma && beta && y && stats && M > 0 && C > 0, "sr_layernorm_fwd_train: bad arguments");
    hipLaunchKernelGGL(sr_ln_fwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, ST, x, gamma, beta, y, stats, M, C, eps);
    SR_CHECK_LAUNCH("sr_layernorm_fwd_train");
    return SR_OK;
}
extern "C" int sr_layernorm_bwd(const float* x, const float* stats, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, long long M, int C, void* stream) {
    SR_REQUIRE(x && stats && gamma && dy && dx && dgamma && dbeta && M > 0 && C > 0 && C <= 64 * LN_MAX_COLS_PER_LANE, "sr_layernorm_bwd: bad arguments (C <= 512)");
    hipLaunchKernelGGL(sr_ln_bwd_kernel, dim3((unsigned)((M + LN_ROWS_PER_WG - 1) / LN_ROWS_PER_WG)), dim3(256), 0, ST, x, stats, gamma, dy, dx, dgamma, dbeta, M, C);
    SR_CHECK_LAUNCH("sr_layernorm_bwd");
    return SR_OK;
}
extern "C" int sr_colsum(const float* x, float* out, int nb, long long P, int C, float alpha, void* stream) {
    SR_REQUIRE(x && out && nb > 0 && nb <= 65535 && P > 0 && C > 0, "sr_colsum: bad arguments");
    if ((C & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0 && P >= 64) {
        const int q = C / 4, ny = (q + 63) / 64, QB = (q + ny - 1) / ny, R = 256 / QB;
        const int env_blocks = 0;  // (> 0 forced the block count in a sweep)
        // row blocks per column block: enough workgroups to stream (a workgroup moves ~30-60 GB/s), few enough that the same-address
        // atomics at the end do not dominate (tools/colsum_bench.py: 16,384 rows 64 blocks, 262,144 rows 128; 256 blocks lose 2x)
        const long long target = env_blocks > 0 ? env_blocks : std::min<long long>(128, std::max<long long>(32, P / 256));
        long long rows_wg = (P + target - 1) / target;
        rows_wg = (rows_wg + 16LL * R - 1) / (16LL * R) * (16LL * R);  // whole 16-load rounds
        hipLaunchKernelGGL(sr_colsum4_kernel, dim3((unsigned)((P + rows_wg - 1) / rows_wg), ny, nb), dim3(256), 0, ST, x, out, P, C, alpha, QB, (int)rows_wg);
    } else
        hipLaunchKernelGGL(sr_colsum_kernel, dim3((unsigned)((P + COLSUM_ROWS - 1) / COLSUM_ROWS), (C + 63) / 64, nb), dim3(256), 0, ST, x, out, P, C, alpha);
    SR_CHECK_LAUNCH("sr_colsum");
    return SR_OK;
}
extern "C" int sr_batch_sum(const float* x, float* out, long long nb, long long n, long long stride_b, void* stream) {
    SR_REQUIRE(x && out && nb > 0 && n > 0, "sr_batch_sum: bad arguments");
    if ((n & 3) == 0 && (stride_b & 3) == 0 && ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(out)) & 15) == 0)
        hipLaunchKernelGGL(sr_batch_sum4_kernel, flat_grid(n / 4), dim3(256), 0, ST, x, out, nb, n / 4, stride_b);
    else
        hipLaunchKernelGGL(sr_batch_sum_kernel, flat_grid(n), dim3(256), 0, ST, x, out, nb, n, stride_b);
    SR_CHECK_LAUNCH("sr_batch_sum");
    return SR_OK;
}
extern "C" int sr_eltwise(int op, const float* x, const float* y, const float* s, float* out, long long n, long long inner, int C, float a, float b, void* stream) {
    SR_REQUIRE(x || op == EW_BCAST_BC, "sr_eltwise: null input");
    SR_REQUIRE(out && n > 0 && op >= 0 && op <= EW_AFFINE_C, "sr_eltwise: bad arguments (op %d)", op);
    hipLaunchKernelGGL(sr_eltwise_kernel, flat_grid(n), dim3(256), 0, ST, op, x, y, s, out, n, inner > 0 ? inner : 1, C > 0 ? C : 1, a, b);
    SR_CHECK_LAUNCH("sr_eltwise");
    return SR_OK;
}
extern "C" int sr_window_copy(const float* src, float* dst, int B, int H, int W, int C, int ws, int shift, int to_windows, void* stream) {
    SR_REQUIRE(src && dst && B > 0 && C > 0 && ws > 0 && H % ws == 0 && W % ws == 0 && shift >= 0 && shift < ws, "sr_window_copy: bad geometry");
    hipLaunchKernelGGL(sr_window_copy_kernel, flat_grid((long long)B * H * W * C), dim3(256), 0, ST, src, dst, B, H, W, C, ws, shift, to_windows);
    SR_CHECK_LAUNCH("sr_window_copy");
    return SR_OK;
}
extern "C" int sr_oca_unfold(const float* img, float* win, int B, int H, int W, int C, int ws, int wse, int forward, void* stream) {
    SR_REQUIRE(img && win && B > 0 && C > 0 && ws > 0 && wse >= ws && (wse - ws) % 2 == 0 && H % ws == 0 && W % ws == 0, "sr_oca_unfold: bad geometry");
    if (forward) {
        const long long total = (long long)B * (H / ws) * (W / ws) * wse * wse * C;
        hipLaunchKernelGGL(sr_oca_unfold_kernel, flat_grid(total), dim3(256), 0, ST, img, win, B, H, W, C, ws, wse);
    } else {  // img = d(image) [out], win = d(windows) [in]
        hipLaunchKernelGGL(sr_oca_fold_kernel, flat_grid((long long)B * H * W * C), dim3(256), 0, ST, win, const_cast<float*>(img), B, H, W, C, ws, wse);
    }
    SR_CHECK_LAUNCH("sr_oca_unfold");
    return SR_OK;
}
extern "C" int sr_pixel_shuffle_nhwc(const float* src, float* dst, int B, int H, int W, int C, int r, int forward, void* stream) {
    SR_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && r > 0, "sr_pixel_shuffle_nhwc: bad arguments");
    hipLaunchKernelGGL(sr_pixel_shuffle_nhwc_kernel, flat_grid((long long)B * H * r * W * r * C), dim3(256), 0, ST, src, dst, B, H, W, C, r, forward);
    SR_CHECK_LAUNCH("sr_pixel_shuffle_nhwc");
    return SR_OK;
}
extern "C" int sr_bias_gather(const float* table, const long long* rpi, float* bias, float* dtable, int T, int heads, long long NN, int forward, void* stream) {
    SR_REQUIRE(rpi && bias && T > 0 && heads > 0 && NN > 0 && (forward ? table != nullptr : dtable != nullptr), "sr_bias_gather: bad arguments");
    hipLaunchKernelGGL(sr_bias_gather_kernel, flat_grid(NN * heads), dim3(256), 0, ST, table, rpi, bias, dtable, T, heads, NN, forward);
    SR_CHECK_LAUNCH("sr_bias_gather");
    return SR_OK;
}
extern "C" int sr_nhwc_out(const float* src, float* dst, const float* scale, const float* shift, int B, int Hs, int Ws, int C, int Ho, int Wo, int forward, void* stream) {
    SR_REQUIRE(src && dst && scale && shift && B > 0 && C > 0 && Ho > 0 && Wo > 0 && Ho <= Hs && Wo <= Ws, "sr_nhwc_out: bad arguments");
    const long long total = forward ? (long long)B * C * Ho * Wo : (long long)B * Hs * Ws * C;
    hipLaunchKernelGGL(sr_nhwc_out_kernel, flat_grid(total), dim3(256), 0, ST, src, dst, scale, shift, B, Hs, Ws, C, Ho, Wo, forward);
    SR_CHECK_LAUNCH("sr_nhwc_out");
    return SR_OK;
}

extern "C" int sr_copy_cols(const float* src, float* dst, long long rows, int n, int ld_s, int off_s, int ld_d, int off_d, int accumulate, void* stream) {
    SR_REQUIRE(src && dst && rows > 0 && n > 0 && off_s >= 0 && off_d >= 0 && off_s + n <= ld_s && off_d + n <= ld_d, "sr_copy_cols: bad arguments");
    hipLaunchKernelGGL(sr_copy_cols_kernel, flat_grid(rows * n), dim3(256), 0, ST, src, dst, rows, n, ld_s, off_s, ld_d, off_d, accumulate);
    SR_CHECK_LAUNCH("sr_copy_cols");
    return SR_OK;
}
extern "C" int sr_conv3d27(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int C, int flip, void* stream) {
    SR_REQUIRE(x && w && out && B > 0 && H > 0 && W > 0 && C > 0, "sr_conv3d27: bad arguments");
    hipLaunchKernelGGL(sr_conv3d27_kernel, flat_grid((long long)B * H * W * C), dim3(256), 0, ST, x, w, bias, out, B, H, W, C, flip);
    SR_CHECK_LAUNCH("sr_conv3d27");
    return SR_OK;
}
extern "C" int sr_conv3d27_wgrad(const float* x, const float* dy, float* dw, float* db, int B, int H, int W, int C, void* stream) {
    SR_REQUIRE(x && dy && dw && db && B > 0 && H > 0 && W > 0 && C > 0, "sr_conv3d27_wgrad: bad arguments");
    const long long total = (long long)B * H * W * C;
    const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(sr_conv3d27_wgrad_kernel, dim3(blocks), dim3(256), 0, ST, x, dy, dw, db, B, H, W, C);
    SR_CHECK_LAUNCH("sr_conv3d27_wgrad");
    return SR_OK;
}
