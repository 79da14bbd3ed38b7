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
#include "sr_common.h"
#include "sr_host.h"

#include <cstdlib>

namespace {

__device__ unsigned long long sr_dbg_mlp[16];
#define STAMP(i) SR_STAMP(sr_dbg_mlp, i)

// KC1 = Cp/32 (K chunks of fc1), KC2 = Hp/32 (K chunks of fc2 = hidden / 32)
//
// 256 threads = 4 waves that split the output columns; every wave keeps 4 row tiles (all 64 rows) so that one
// weight fragment feeds 4 MFMAs and a 4-slot register ring is 3 chunks x 12 MFMAs = 576 cycles ahead of its use
// (measured L2 latency ~450-600 cycles).  What the in-kernel cycle stamps showed and this layout answers:
//   * every global load that can be issued at kernel entry IS issued there (weights ring, rows, residual tile,
//     both bias vectors) -- a load placed after a store or in an epilogue costs a full exposed L2 round trip;
//   * fc1 runs as two N-halves of 3 n-tiles: 12 live accumulator tiles instead of 24;
//   * one sched_barrier per K-chunk keeps hipcc from sinking the ring loads or hoisting 50 of them at once;
//   * GELU is the sigmoid form (sr_common.h gelu_bf16): with the erf polynomial the VALU time of this kernel
//     exceeded its MFMA time.
template <int KC1, int KC2, int M_T>
__global__ __launch_bounds__(256, M_T == 128 ? 1 : (M_T == 64 ? 2 : 4)) void sr_mlp_kernel(SrMlp a) {
    constexpr int MT = M_T / 16;
    constexpr int NT1 = KC2 * 2 / 4;  // fc1 n-tiles per wave (hidden/16/4) = 6
    constexpr int NH = NT1 / 2;       // per half = 3
    constexpr int NT2 = KC1 * 2 / 4;  // fc2 n-tiles per wave (Cp/16/4) = 3
    constexpr int RING = 4;
    constexpr int NP = M_T / 32;
    constexpr int KI = KC1 / 2;
    static_assert(NT1 % 2 == 0 && NH == NT2 && KC1 >= RING && KC2 >= RING, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<bf16>* Aimg = reinterpret_cast<Frag<bf16>*>(smem);  // [KC1*4][M_T]
    Frag<bf16>* Himg = Aimg + KC1 * 4 * M_T;                 // [KC2*4][M_T]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ar = lane & 15, ag = lane >> 4;
    const int r8 = lane & 7, kq = lane >> 3;
    const int m0 = blockIdx.x * M_T;
    STAMP(0);

    const Frag<bf16>* W1 = reinterpret_cast<const Frag<bf16>*>(a.w1p) + (size_t)(wave * NT1) * KC1 * 64 + lane;
    const Frag<bf16>* W2 = reinterpret_cast<const Frag<bf16>*>(a.w2p) + (size_t)(wave * NT2) * KC2 * 64 + lane;

    // weight stream position t: [0,KC1) fc1 half 0, [KC1,2KC1) fc1 half 1, [2KC1, 2KC1+KC2) fc2; slot = t % RING
    Frag<bf16> wr[RING][NH];
    auto stream_load = [&](int t, int slot) {
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

    // ---- P0: every load that does not depend on computed data
#pragma unroll
    for (int c = 0; c < RING - 1; ++c) stream_load(c, c);
    f32x4 rlo[NP][KI], rhi[NP][KI];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int row = m0 + wave * (M_T / 4) + p * 8 + r8;
        const float* src = a.x + (size_t)(row < a.M ? row : a.M - 1) * a.ldx + kq * 8;  // clamp: rows >= M never stored
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            rlo[p][i] = load4(src + i * 64);
            rhi[p][i] = load4(src + i * 64 + 4);
        }
    }
    f32x4 res[MT][NT2];  // residual tile = initial fc2 accumulator (same lines as above: served by L1/L2)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = m0 + m * 16 + ar;
#pragma unroll
        for (int n = 0; n < NT2; ++n) res[m][n] = load4(a.x + (size_t)(row < a.M ? row : a.M - 1) * a.ldx + (wave * NT2 + n) * 16 + ag * 4);
    }
    f32x4 b1r[NT1], b2r[NT2];
#pragma unroll
    for (int n = 0; n < NT1; ++n) b1r[n] = load4(a.b1 + (wave * NT1 + n) * 16 + ag * 4);
#pragma unroll
    for (int n = 0; n < NT2; ++n) b2r[n] = load4(a.b2 + (wave * NT2 + n) * 16 + ag * 4);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);

    // ---- P1: LayerNorm -> LDS
    {
        const float inv = 1.0f / (float)a.C;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < KI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s += rlo[p][i][j] + rhi[p][i][j];
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
                    const float d0 = (c0 + j < a.C) ? rlo[p][i][j] - mean : 0.f;
                    const float d1 = (c0 + 4 + j < a.C) ? rhi[p][i][j] - mean : 0.f;
                    q += d0 * d0 + d1 * d1;
                }
            }
            q = wave_sum_xor(q, 8);
            q = wave_sum_xor(q, 16);
            q = wave_sum_xor(q, 32);
            const float rstd = rsqrtf(q * inv + a.eps);
            Frag<bf16>* dst = Aimg + wave * (M_T / 4) + p * 8 + r8;
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                const int kg = kq + 8 * i;
                f32x4 o0 = (rlo[p][i] - mean) * rstd;
                f32x4 o1 = (rhi[p][i] - mean) * rstd;
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
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);

    // ---- P2: fc1 in two N-halves, GELU -> hidden image
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4 acc1[MT][NH];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NH; ++n) acc1[m][n] = (f32x4)(0.0f);
#pragma unroll
        for (int c = 0; c < KC1; ++c) {
            const int t = half * KC1 + c;
            stream_load(t + RING - 1, (t + RING - 1) % RING);
            const Frag<bf16>* arow = Aimg + (c * 4 + ag) * M_T + ar;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const Frag<bf16> av = arow[m * 16];
#pragma unroll
                for (int n = 0; n < NH; ++n) mma(wr[t % RING][n], av, acc1[m][n]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(4 + 2 * half);
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            const int col = (wave * NT1 + half * NH + n) * 16 + ag * 4;
            char* hbase = reinterpret_cast<char*>(Himg + (col >> 3) * M_T + ar) + (ag & 1) * 8;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 v = acc1[m][n] + b1r[half * NH + n];
                bf16x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (bf16)gelu_bf16(v[r]);
                *reinterpret_cast<bf16x4*>(hbase + m * 16 * sizeof(Frag<bf16>)) = h;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5 + 2 * half);
    }
    __syncthreads();
    STAMP(8);

    // ---- P3: fc2 on top of the residual
#pragma unroll
    for (int c = 0; c < KC2; ++c) {
        const int t = 2 * KC1 + c;
        stream_load(t + RING - 1, (t + RING - 1) % RING);
        const Frag<bf16>* hrow = Himg + (c * 4 + ag) * M_T + ar;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const Frag<bf16> hv = hrow[m * 16];
#pragma unroll
            for (int n = 0; n < NT2; ++n) mma(wr[t % RING][n], hv, res[m][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(9);
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
        const int col = (wave * NT2 + n) * 16 + ag * 4;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = m0 + m * 16 + ar;
            if (row < a.M) store4(a.out + (size_t)row * a.ldx + col, res[m][n] + b2r[n]);
        }
    }
    STAMP(10);
}

}  // namespace

extern "C" int sr_debug_mlp_stamps(unsigned long long* host16) {
    return hipMemcpyFromSymbol(host16, HIP_SYMBOL(sr_dbg_mlp), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}

extern "C" int sr_mlp_fused_supported(int Cp, int Hp, int compute_dtype) {
    return (compute_dtype == SR_BF16 && Cp == 192 && Hp == 384) ? 1 : 0;
}

extern "C" int sr_mlp_fused(const SrMlp* p, void* stream) {
    SR_REQUIRE(p && p->x && p->out && (!p->ln_gamma == !p->ln_beta) && p->w1p && p->b1 && p->w2p && p->b2, "sr_mlp_fused: null pointer");
    const SrMlp& a = *p;
    SR_REQUIRE(a.M > 0 && a.C > 0 && a.C <= a.Cp && a.ldx >= a.Cp, "sr_mlp_fused: bad geometry");
    SR_REQUIRE(sr_mlp_fused_supported(a.Cp, a.Hp, SR_BF16), "sr_mlp_fused: unsupported Cp=%d Hp=%d (use two sr_gemm calls)", a.Cp, a.Hp);
    const int rows = 64;  // experiments: 128 rows per workgroup (8 MFMAs per weight fragment), 32 rows (2)
    if (rows == 32) {
        constexpr int lds = (6 + 12) * 4 * 32 * 16;  // 36 KiB
        static SrDeviceOnce attr_once32;
        const hipError_t e = sr_once_per_device(attr_once32, [&] { return sr_allow_lds(sr_mlp_kernel<6, 12, 32>, lds); });
        SR_REQUIRE(e == hipSuccess, "sr_mlp_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL((sr_mlp_kernel<6, 12, 32>), dim3((a.M + 31) / 32), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a);
        SR_CHECK_LAUNCH("sr_mlp_fused");
        return SR_OK;
    }
    if (rows == 128) {
        constexpr int lds = (6 + 12) * 4 * 128 * 16;  // 144 KiB
        static SrDeviceOnce attr_once128;
        const hipError_t e = sr_once_per_device(attr_once128, [&] { return sr_allow_lds(sr_mlp_kernel<6, 12, 128>, lds); });
        SR_REQUIRE(e == hipSuccess, "sr_mlp_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL((sr_mlp_kernel<6, 12, 128>), dim3((a.M + 127) / 128), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a);
        SR_CHECK_LAUNCH("sr_mlp_fused");
        return SR_OK;
    }
    constexpr int lds = (6 + 12) * 4 * 64 * 16;  // 72 KiB
    static SrDeviceOnce attr_once;  // one flag per template instantiation, one bit per device
    {
        const hipError_t e = sr_once_per_device(attr_once, [&] { return sr_allow_lds(sr_mlp_kernel<6, 12, 64>, lds); });
        SR_REQUIRE(e == hipSuccess, "sr_mlp_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL((sr_mlp_kernel<6, 12, 64>), dim3((a.M + 63) / 64), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a);
    SR_CHECK_LAUNCH("sr_mlp_fused");
    return SR_OK;
}
