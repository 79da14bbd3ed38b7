// Fused transformer MLP:  x += fc2(GELU(fc1(LayerNorm(x))))   (common.py:173-195 + swinir.py:172).
//
// The un-fused pair of row GEMMs moves the fp32 stream (32 MB at the bench shape) and the
// [rows x hidden] intermediate through HBM / Infinity Cache five times; here one workgroup owns
// 64 rows end to end and the stream is read once and written once:
//   t0  issue fc1 weight fragments (ring of 3 K-chunks) + this wave's 16 rows (fp32) back to back
//   t1  LayerNorm in registers -> bf16 K-group-major image in LDS (24 KiB)             | barrier
//   t2  fc1: 6 K-chunks x 24 MFMAs per wave (waves split the 384 hidden columns), weights streamed
//       straight from L2 into the register ring, no barrier
//   t3  issue fc2 weight ring + the residual tile (acc2 = x) ; bias + exact-erf GELU on the fc1
//       accumulators -> bf16 -> hidden image in LDS (48 KiB)                            | barrier
//   t4  fc2: 12 K-chunks x 12 MFMAs per wave on top of the residual
//   t5  + bias, 16-byte fp32 stores
// 72 KiB of LDS -> two workgroups per CU, so one workgroup's VALU phases (LN, GELU) run beside
// the other's MFMA phases.
#include "sr_common.cuh"
#include "sr_host.h"

namespace {

// KC1 = Cp/32 (K chunks of fc1), KC2 = Hp/32 (K chunks of fc2 = hidden / 32)
//
// 512 threads = 8 waves as 2 (row halves) x 4 (column quarters).  The kernel is latency-bound, not
// MFMA-bound (rocprof: waves spend >60 % of their life in s_waitcnt / issue stalls at 2 waves per
// SIMD), so the tile is cut for OCCUPANCY: every wave keeps <= 128 VGPRs and two workgroups (72 KiB
// LDS each) put 4 waves on every SIMD.  The two row-halves fetch the same weight fragments a few
// cycles apart (second request hits L1).  fc1 runs as two N-halves of 3 n-tiles so that only 6
// accumulator tiles are live; weights stream through a 3-chunk register ring pinned with
// sched_barriers (the compiler otherwise sinks the loads to their uses).
template <int KC1, int KC2>
__global__ __launch_bounds__(512, 2) void sr_mlp_kernel(SrMlp a) {
    constexpr int M_T = 64, MW = 2;   // 16-row tiles per wave
    constexpr int NT1 = KC2 * 2 / 4;  // fc1 n-tiles per wave column (hidden/16/4) = 6
    constexpr int NH = NT1 / 2;       // per half = 3
    constexpr int NT2 = KC1 * 2 / 4;  // fc2 n-tiles per wave column (Cp/16/4) = 3
    constexpr int RING = 3;
    constexpr int KI = KC1 / 2;
    static_assert(NT1 % 2 == 0 && KC1 >= RING && KC2 >= RING, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<bf16>* Aimg = reinterpret_cast<Frag<bf16>*>(smem);  // [KC1*4][M_T]
    Frag<bf16>* Himg = Aimg + KC1 * 4 * M_T;                 // [KC2*4][M_T]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ar = lane & 15, ag = lane >> 4;
    const int r8 = lane & 7, kq = lane >> 3;
    const int m0 = blockIdx.x * M_T;

    const Frag<bf16>* W1 = reinterpret_cast<const Frag<bf16>*>(a.w1p) + (size_t)(wn * NT1) * KC1 * 64 + lane;
    const Frag<bf16>* W2 = reinterpret_cast<const Frag<bf16>*>(a.w2p) + (size_t)(wn * NT2) * KC2 * 64 + lane;

    // ---- P0: first fc1 weight chunks + this wave's 8 rows -> LayerNorm -> LDS
    Frag<bf16> wr[RING][NH];
#pragma unroll
    for (int c = 0; c < RING; ++c)
#pragma unroll
        for (int n = 0; n < NH; ++n) wr[c][n] = W1[((size_t)n * KC1 + c) * 64];
    {
        f32x4 rlo[KI], rhi[KI];
        const int row = m0 + wave * 8 + r8;
        const float* src = a.x + (size_t)(row < a.M ? row : a.M - 1) * a.ldx + kq * 8;  // clamp: rows >= M never stored
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            rlo[i] = load4(src + i * 64);
            rhi[i] = load4(src + i * 64 + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float inv = 1.0f / (float)a.C;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < KI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += rlo[i][j] + rhi[i][j];
        s = wave_sum_xor(s, 8);
        s = wave_sum_xor(s, 16);
        s = wave_sum_xor(s, 32);
        const float mean = s * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int c0 = (kq + 8 * i) * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d0 = (c0 + j < a.C) ? rlo[i][j] - mean : 0.f;
                const float d1 = (c0 + 4 + j < a.C) ? rhi[i][j] - mean : 0.f;
                q += d0 * d0 + d1 * d1;
            }
        }
        q = wave_sum_xor(q, 8);
        q = wave_sum_xor(q, 16);
        q = wave_sum_xor(q, 32);
        const float rstd = rsqrtf(q * inv + a.eps);
        Frag<bf16>* dst = Aimg + wave * 8 + r8;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int kg = kq + 8 * i;
            f32x4 o0 = (rlo[i] - mean) * rstd;
            f32x4 o1 = (rhi[i] - mean) * rstd;
            if (a.ln_gamma) {  // un-folded affine (folded: gamma/beta live in w1p / b1)
                o0 = o0 * load4(a.ln_gamma + kg * 8) + load4(a.ln_beta + kg * 8);
                o1 = o1 * load4(a.ln_gamma + kg * 8 + 4) + load4(a.ln_beta + kg * 8 + 4);
            }
            Frag<bf16> f;
            f.v[0] = (bf16)o0[0]; f.v[1] = (bf16)o0[1]; f.v[2] = (bf16)o0[2]; f.v[3] = (bf16)o0[3];
            f.v[4] = (bf16)o1[0]; f.v[5] = (bf16)o1[1]; f.v[6] = (bf16)o1[2]; f.v[7] = (bf16)o1[3];
            dst[kg * M_T] = f;
        }
    }
    __syncthreads();

    // ---- P1: fc1 in two N-halves; the weight stream runs continuously through both halves and on
    //          into the first fc2 chunks (stream position t: t < 2*KC1 -> fc1, else fc2 chunk t - 2*KC1)
    auto stream_load = [&](int t, int slot) {
        if (a.debug_flags & 1) return;  // timing-only ablation: no weight streaming
        if (t < KC1) {
#pragma unroll
            for (int n = 0; n < NH; ++n) wr[slot][n] = W1[((size_t)n * KC1 + t) * 64];
        } else if (t < 2 * KC1) {
#pragma unroll
            for (int n = 0; n < NH; ++n) wr[slot][n] = W1[((size_t)(NH + n) * KC1 + (t - KC1)) * 64];
        } else if (t < 2 * KC1 + KC2) {
#pragma unroll
            for (int n = 0; n < NT2; ++n) wr[slot][n] = W2[((size_t)n * KC2 + (t - 2 * KC1)) * 64];
        }
    };
    static_assert(NH == NT2, "the ring is shared by fc1 halves and fc2");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4 acc1[MW][NH];
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < NH; ++n) acc1[m][n] = (f32x4)(0.0f);
#pragma unroll
        for (int c = 0; c < KC1; ++c) {
            const int t = half * KC1 + c;
            const int slot = t % RING;
            const Frag<bf16>* arow = Aimg + (c * 4 + ag) * M_T + wm * 32 + ar;
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const Frag<bf16> av = arow[m * 16];
#pragma unroll
                for (int n = 0; n < NH; ++n)
                    if (!(a.debug_flags & 4)) mma(wr[slot][n], av, acc1[m][n]);
            }
            stream_load(t + RING, slot);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            const int col = (wn * NT1 + half * NH + n) * 16 + ag * 4;
            const f32x4 bias = load4(a.b1 + col);
            char* hbase = reinterpret_cast<char*>(Himg + (col >> 3) * M_T + wm * 32 + ar) + (ag & 1) * 8;
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const f32x4 v = acc1[m][n] + bias;
                bf16x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (bf16)((a.debug_flags & 2) ? v[r] : gelu_fast(v[r]));
                *reinterpret_cast<bf16x4*>(hbase + m * 16 * sizeof(Frag<bf16>)) = h;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();

    // ---- P2: fc2 + residual
    f32x4 res[MW][NT2];
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int row = m0 + wm * 32 + m * 16 + ar;
#pragma unroll
        for (int n = 0; n < NT2; ++n) res[m][n] = load4(a.x + (size_t)(row < a.M ? row : a.M - 1) * a.ldx + (wn * NT2 + n) * 16 + ag * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc2[MW][NT2];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NT2; ++n) acc2[m][n] = (f32x4)(0.0f);
#pragma unroll
    for (int c = 0; c < KC2; ++c) {
        const int t = 2 * KC1 + c;
        const int slot = t % RING;
        const Frag<bf16>* hrow = Himg + (c * 4 + ag) * M_T + wm * 32 + ar;
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const Frag<bf16> hv = hrow[m * 16];
#pragma unroll
            for (int n = 0; n < NT2; ++n)
                if (!(a.debug_flags & 4)) mma(wr[slot][n], hv, acc2[m][n]);
        }
        stream_load(t + RING, slot);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
        const int col = (wn * NT2 + n) * 16 + ag * 4;
        const f32x4 bias = load4(a.b2 + col);
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int row = m0 + wm * 32 + m * 16 + ar;
            if (row < a.M) store4(a.out + (size_t)row * a.ldx + col, acc2[m][n] + res[m][n] + bias);
        }
    }
}

}  // namespace

extern "C" int sr_mlp_fused_supported(int Cp, int Hp, int compute_dtype) {
    return (compute_dtype == SR_BF16 && Cp == 192 && Hp == 384) ? 1 : 0;
}

extern "C" int sr_mlp_fused(const SrMlp* p, void* stream) {
    SR_REQUIRE(p && p->x && p->out && (!p->ln_gamma == !p->ln_beta) && p->w1p && p->b1 && p->w2p && p->b2, "sr_mlp_fused: null pointer");
    const SrMlp& a = *p;
    SR_REQUIRE(a.M > 0 && a.C > 0 && a.C <= a.Cp && a.ldx >= a.Cp, "sr_mlp_fused: bad geometry");
    SR_REQUIRE(sr_mlp_fused_supported(a.Cp, a.Hp, SR_BF16), "sr_mlp_fused: unsupported Cp=%d Hp=%d (use two sr_gemm calls)", a.Cp, a.Hp);
    constexpr int lds = (6 + 12) * 4 * 64 * 16;  // 72 KiB
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = sr_allow_lds(sr_mlp_kernel<6, 12>, lds);
        SR_REQUIRE(e == hipSuccess, "sr_mlp_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL((sr_mlp_kernel<6, 12>), dim3((a.M + 63) / 64), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), a);
    SR_CHECK_LAUNCH("sr_mlp_fused");
    return SR_OK;
}
