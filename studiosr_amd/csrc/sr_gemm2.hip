// Latency-engineered row GEMM for the hot SwinIR / HAT shapes (bf16 operands, K = 192 or 384,
// N a multiple of 192).  Same math, layouts and epilogues as sr_gemm.hip (which stays the generic
// and exact-fp32 path); what changes is WHEN memory is touched:
//
//   t0  every wave issues its weight-fragment loads for the first 6 K-chunks (18 x 1 KiB) ...
//   t0  ... and ALL of its activation-row loads (12 x 16/32 B per lane) back to back, and, for the
//       residual epilogues, the residual tile itself straight into the accumulator registers
//       (acc = skip + bias, the MFMAs then accumulate on top) -- one exposed L2 latency per
//       workgroup instead of one per row pass / K-chunk / output tile;
//   t1  LayerNorm in registers (two-pass statistics across the 8 lanes of a row), bf16 -> LDS;
//   t2  one barrier, then a fully unrolled MFMA loop that only reads LDS; for K = 384 the second
//       half of the weight fragments is re-loaded into the ring slots as they retire (6 chunks =
//       72+ MFMAs ahead of use);
//   t3  stores only.
// No integer division anywhere: window geometry uses shifts (ws, ntok, hd_p are powers of two)
// and host-computed multiply-shift constants (sr_common.h FastDiv).
// Two workgroups fit per CU (<= 48 KiB LDS, <= 256 VGPRs), so one's MFMA phase hides the
// other's load / LayerNorm / store phases.
#include "sr_common.h"
#include "sr_host.h"

namespace {

struct Gemm2 {
    SrGemm g;
    FastDiv div_hw, div_nwx, div_w;
    int ws_log2, ntok_log2, hdp_log2;
    int acc_from_skip;  // acc = skip + bias before the MFMA loop (act NONE, scale 1)
};

SR_DEV int win_map(const Gemm2& a, int row) {
    uint32_t b, rem;
    a.div_hw.divmod((uint32_t)row, b, rem);
    const uint32_t win = rem >> a.ntok_log2, tok = rem & ((1u << a.ntok_log2) - 1);
    uint32_t wy, wx;
    a.div_nwx.divmod(win, wy, wx);
    const int i = tok >> a.ws_log2, j = tok & ((1 << a.ws_log2) - 1);
    int y = (wy << a.ws_log2) + i + (a.g.y_mode == SR_Y_ROLL ? a.g.shift : 0);
    int x = (wx << a.ws_log2) + j + a.g.shift;
    if (y >= a.g.H) y -= a.g.H;
    if (x >= a.g.W) x -= a.g.W;
    return ((int)b * a.g.H + y) * a.g.W + x;
}

template <typename TIn>
struct Raw;  // one K-group as loaded from global memory
template <>
struct Raw<float> {
    f32x4 lo, hi;
};
template <>
struct Raw<bf16> {
    bf16x8 v;
};
SR_DEV Raw<float> raw_load(const float* p) {
    Raw<float> r;
    r.lo = *reinterpret_cast<const f32x4*>(p);
    r.hi = *reinterpret_cast<const f32x4*>(p + 4);
    return r;
}
SR_DEV Raw<bf16> raw_load(const bf16* p) {
    Raw<bf16> r;
    r.v = *reinterpret_cast<const bf16x8*>(p);
    return r;
}
SR_DEV Raw<float> raw_zero(const float*) {
    Raw<float> r;
    r.lo = (f32x4)(0.f);
    r.hi = (f32x4)(0.f);
    return r;
}
SR_DEV Raw<bf16> raw_zero(const bf16*) {
    Raw<bf16> r;
    r.v = (bf16x8)(0.f);
    return r;
}
SR_DEV Frag<bf16> raw_to_frag(const Raw<bf16>& r) {
    Frag<bf16> f;
    f.v = r.v;
    return f;
}
SR_DEV Frag<bf16> raw_to_frag(const Raw<float>& r) {
    Frag<bf16> f;
    f.v[0] = (bf16)r.lo[0]; f.v[1] = (bf16)r.lo[1]; f.v[2] = (bf16)r.lo[2]; f.v[3] = (bf16)r.lo[3];
    f.v[4] = (bf16)r.hi[0]; f.v[5] = (bf16)r.hi[1]; f.v[6] = (bf16)r.hi[2]; f.v[7] = (bf16)r.hi[3];
    return f;
}

// MT: 16-row tiles per workgroup; KC: K/32.  NW = 3 (N tile 192), 4 waves split N.
// STAGE = false: the activation tile is already in LDS (single-pass QKV: the q, k and v column slices reuse one staged tile)
// sum over the 8 adjacent lanes 8k .. 8k+7 (the K-groups of one activation row), result in all of them: DPP quad_perm xor 1, xor 2, then
// row_half_mirror (lane i <- lane 7 - i of its group of 8: after the quad sums every lane of a quad holds the same value)
SR_DEV float row8_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    return v;
}

template <typename TIn, int MT, int KC, bool SWAPPED, bool STAGE = true>
SR_DEV void gemm2_body(const Gemm2& a, Frag<bf16>* As, int lane, int wave, int yblk) {
    constexpr int NW = 3;
    constexpr int M_T = MT * 16;
    constexpr int MS = M_T + 1;  // row stride of the K-group-major LDS image in cells: odd, so that the staging writes of ONE row's 8 K-groups (8 adjacent lanes) hit 8 bank groups
    constexpr int RING = KC < 6 ? KC : 6;
    constexpr int NP = M_T / 32;  // row passes per wave (8 rows each)
    constexpr int KI = KC / 2;    // K-groups per lane per row (8 lanes share a row)
    const SrGemm& g = a.g;
    const int m0 = blockIdx.x * M_T;
    const int ar = lane & 15, ag = lane >> 4;
    // activation staging: the 8 lanes that share a row are ADJACENT lanes (K-group on the fast lane axis): they read its 128 / 256 contiguous
    // bytes; row-fastest lanes made every quad of lanes touch four cache lines (four TA tag cycles per quad)
    const int kq = lane & 7, r8 = lane >> 3;
    const int ntile0 = yblk * (4 * NW) + wave * NW;

    // ---- t0: weight fragments for the first RING chunks
    const Frag<bf16>* Bp = reinterpret_cast<const Frag<bf16>*>(g.Wp) + (size_t)ntile0 * KC * 64 + lane;
    Frag<bf16> bfr[RING][NW];
#pragma unroll
    for (int c = 0; c < RING; ++c)
#pragma unroll
        for (int n = 0; n < NW; ++n) bfr[c][n] = Bp[((size_t)n * KC + c) * 64];

    // ---- t0: all activation rows of this wave
    Raw<TIn> raw[STAGE ? NP : 1][KI];
    bool rvalid[NP];
    if constexpr (STAGE) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int row = m0 + wave * (M_T / 4) + p * 8 + r8;
        rvalid[p] = row < g.M;
        const int crow = rvalid[p] ? row : g.M - 1;  // clamp instead of branching: rows >= M are never stored
        const int srow = g.a_map == SR_MAP_WINDOW ? win_map(a, crow) : crow;
        const TIn* src = reinterpret_cast<const TIn*>(g.A) + (size_t)srow * g.lda + kq * 8;
#pragma unroll
        for (int i = 0; i < KI; ++i) raw[p][i] = raw_load(src + i * 64);
    }
    }

    // ---- t0: accumulators (optionally pre-loaded with residual + bias)
    f32x4 acc[MT][NW];
    int orow[MT];
    if constexpr (SWAPPED) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = m0 + m * 16 + ar;
            orow[m] = row < g.M ? (g.o_map == SR_MAP_WINDOW ? win_map(a, row) : row) : -1;
        }
    }
    // bf16 activations leave room to fetch the residual at t0; fp32 activations (96 live registers
    // of raw rows) take it after the LDS writes instead.
    constexpr bool EARLY_ACC = sizeof(TIn) == 2;
    auto init_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                acc[m][n] = (f32x4)(0.0f);
                if constexpr (SWAPPED) {
                    if (a.acc_from_skip && orow[m] >= 0) acc[m][n] = load4(g.skip + (size_t)orow[m] * g.ldskip + (ntile0 + n) * 16 + ag * 4);
                }
            }
    };
    if constexpr (EARLY_ACC) init_acc();

    __builtin_amdgcn_sched_barrier(0);  // every global load above is issued before any LayerNorm math
    // ---- t1: (LayerNorm) -> bf16 -> LDS
    if constexpr (STAGE) {
        const bool ln = g.ln_gamma != nullptr || g.ln_norm_only;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            Frag<bf16>* dst = As + wave * (M_T / 4) + p * 8 + r8;
            if constexpr (sizeof(TIn) == 4) {
                if (ln) {
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < KI; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) s += raw[p][i].lo[j] + raw[p][i].hi[j];
                    s = row8_sum(s);
                    const float inv = 1.0f / (float)g.k_real;
                    const float mean = s * inv;
                    float q = 0.f;
#pragma unroll
                    for (int i = 0; i < KI; ++i) {
                        const int c0 = (kq + 8 * i) * 8;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float d0 = (c0 + j < g.k_real) ? raw[p][i].lo[j] - mean : 0.f;
                            const float d1 = (c0 + 4 + j < g.k_real) ? raw[p][i].hi[j] - mean : 0.f;
                            q += d0 * d0 + d1 * d1;
                        }
                    }
                    q = row8_sum(q);
                    const float rstd = rsqrtf(q * inv + g.ln_eps);
#pragma unroll
                    for (int i = 0; i < KI; ++i) {
                        const int kg = kq + 8 * i;
                        Raw<float> o;
                        if (g.ln_norm_only) {  // gamma/beta live in the packed weights; pad lanes meet zero weight rows
                            o.lo = (raw[p][i].lo - mean) * rstd;
                            o.hi = (raw[p][i].hi - mean) * rstd;
                        } else {
                            const f32x4 g0 = load4(g.ln_gamma + kg * 8), g1 = load4(g.ln_gamma + kg * 8 + 4);
                            const f32x4 b0 = load4(g.ln_beta + kg * 8), b1 = load4(g.ln_beta + kg * 8 + 4);
                            o.lo = (raw[p][i].lo - mean) * rstd * g0 + b0;
                            o.hi = (raw[p][i].hi - mean) * rstd * g1 + b1;
                        }
                        dst[kg * MS] = raw_to_frag(o);
                    }
                    continue;
                }
            }
#pragma unroll
            for (int i = 0; i < KI; ++i) dst[(kq + 8 * i) * MS] = raw_to_frag(raw[p][i]);
        }
    }
    if constexpr (!EARLY_ACC) init_acc();
    if constexpr (STAGE) __syncthreads();

    // ---- t2: MFMA loop (LDS + registers only; weight ring refilled for K > 6 chunks)
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const int slot = c % RING;
        const Frag<bf16>* arow = As + (c * 4 + ag) * MS + ar;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const Frag<bf16> av = arow[m * 16];
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                if constexpr (SWAPPED)
                    mma(bfr[slot][n], av, acc[m][n]);
                else
                    mma(av, bfr[slot][n], acc[m][n]);
            }
        }
        if (c + RING < KC) {
#pragma unroll
            for (int n = 0; n < NW; ++n) bfr[slot][n] = Bp[((size_t)n * KC + c + RING) * 64];
            __builtin_amdgcn_sched_barrier(0);  // keep the refill here: RING chunks ahead of its use
        }
    }

    // ---- t3: epilogue
    if constexpr (SWAPPED) {
        const int HP = g.heads * g.hd_p;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int col = (ntile0 + n) * 16 + ag * 4;
            f32x4 bias = (f32x4)(0.0f);
            if (g.bias) bias = load4(g.bias + col);
            int part = 0, head = 0, d0 = 0;
            if (g.epi != SR_EPI_STD) {
                part = (col >= HP) ? 1 : 0;
                const int rem = col - part * HP;
                head = rem >> a.hdp_log2;
                d0 = rem & (g.hd_p - 1);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (orow[m] < 0) continue;
                f32x4 v = acc[m][n] + bias;
                if (g.act != SR_ACT_NONE) {  // wave-uniform; erff-free (compile-time variants only)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = g.act == SR_ACT_GELU ? gelu_fast(v[r]) : (v[r] > 0.f ? v[r] : (g.act == SR_ACT_LRELU ? 0.01f * v[r] : 0.f));
                }
                if (g.out_scale != 1.0f) v *= g.out_scale;
                if (g.epi != SR_EPI_STD) {
                    const int row = m0 + m * 16 + ar;
                    const int bwin = row >> a.ntok_log2, tok = row & (g.ntok - 1);
                    size_t off = ((((size_t)bwin * g.heads + head) << a.ntok_log2) + tok) * g.hd_p + d0;
                    if (g.epi == SR_EPI_QKV_OCA && part == 1) {  // k -> zero-bordered image order (hat.py:255: the border is Unfold's padding)
                        uint32_t bb, rem2, yy, xx;
                        a.div_hw.divmod((uint32_t)win_map(a, row), bb, rem2);
                        a.div_w.divmod(rem2, yy, xx);
                        off = ((((size_t)bb * (g.H + 2 * g.oca_pad) + yy + g.oca_pad) * (g.W + 2 * g.oca_pad) + xx + g.oca_pad) * g.heads + head) * g.hd_p + d0;
                    }
                    store4(reinterpret_cast<bf16*>(part == 0 ? g.out : g.out_k) + off, v);
                } else {
                    if (g.skip && !a.acc_from_skip) v += load4(g.skip + (size_t)orow[m] * g.ldskip + col);
                    if (g.skip2) {  // gated second residual (HAT: + conv_scale * CAB): one fma per element
                        const size_t o2 = (size_t)orow[m] * g.ldskip2 + col;
                        const f32x4 y2 = g.skip2_dtype == SR_BF16 ? load4(reinterpret_cast<const bf16*>(g.skip2) + o2) : load4(reinterpret_cast<const float*>(g.skip2) + o2);
                        const f32x4 gt = load4(g.skip2_gate + (size_t)(orow[m] / g.gate_rows) * g.ld_gate + col);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(y2[r], gt[r], v[r]);
                    }
                    const size_t off = (size_t)orow[m] * g.ldo + col;
                    if (g.out_dtype == SR_BF16)
                        store4(reinterpret_cast<bf16*>(g.out) + off, v);
                    else
                        store4(reinterpret_cast<float*>(g.out) + off, v);
                }
            }
        }
    } else {
        // V third of the QKV projection, stored transposed: lane = feature, registers = 4 consecutive tokens
        const int HP = g.heads * g.hd_p;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int col = (ntile0 + n) * 16 + ar;
            const float bias = g.bias ? g.bias[col] : 0.f;
            const int rem = col - 2 * HP;
            const int head = rem >> a.hdp_log2, d = rem & (g.hd_p - 1);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row0 = m0 + m * 16 + ag * 4;
                if (row0 >= g.M) continue;
                const int bwin = row0 >> a.ntok_log2, tok0 = row0 & (g.ntok - 1);
                size_t off = ((((size_t)bwin * g.heads + head) * g.hd_p + d) << a.ntok_log2) + tok0;
                if (g.epi == SR_EPI_QKV_OCA) {  // v -> transposed zero-bordered planes (4 consecutive tokens = 4 consecutive x)
                    uint32_t bb, rem2, yy, xx;
                    a.div_hw.divmod((uint32_t)win_map(a, row0), bb, rem2);
                    a.div_w.divmod(rem2, yy, xx);
                    const size_t plane = (size_t)(g.H + 2 * g.oca_pad) * (g.W + 2 * g.oca_pad);
                    off = (((size_t)bb * g.heads + head) * g.hd_p + d) * plane + (size_t)(yy + g.oca_pad) * (g.W + 2 * g.oca_pad) + xx + g.oca_pad;
                }
                store4(reinterpret_cast<bf16*>(g.out_vt) + off, acc[m][n] + bias);
            }
        }
    }
}

template <typename TIn, int MT, int KC>
__global__ __launch_bounds__(256, 2) void sr_gemm2_kernel(Gemm2 a) {
    __shared__ __attribute__((aligned(16))) Frag<bf16> As[KC * 4 * (MT * 16 + 1)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool vpart = (a.g.epi != SR_EPI_STD) && ((int)blockIdx.y * 192 >= 2 * a.g.heads * a.g.hd_p);
    if (vpart)
        gemm2_body<TIn, MT, KC, false>(a, As, lane, wave, blockIdx.y);
    else
        gemm2_body<TIn, MT, KC, true>(a, As, lane, wave, blockIdx.y);
}

// QKV projection with heads * hd_p == 192 (N = 576): the q, k and v column slices in ONE workgroup, so the 128-row tile
// is fetched, normalised and staged once instead of three times (the projection is bound by that, not by its MFMAs).
template <typename TIn, int MT, int KC>
__global__ __launch_bounds__(256, 2) void sr_gemm2_qkv_kernel(Gemm2 a) {
    __shared__ __attribute__((aligned(16))) Frag<bf16> As[KC * 4 * (MT * 16 + 1)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gemm2_body<TIn, MT, KC, true, true>(a, As, lane, wave, 0);
    gemm2_body<TIn, MT, KC, true, false>(a, As, lane, wave, 1);
    gemm2_body<TIn, MT, KC, false, false>(a, As, lane, wave, 2);
}

int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

template <typename TIn, int MT, int KC>
int launch2(const Gemm2& a, hipStream_t st) {
    dim3 grid((a.g.M + MT * 16 - 1) / (MT * 16), a.g.N / 192);
    if (a.g.epi != SR_EPI_STD && a.g.N == 576 && a.g.heads * a.g.hd_p == 192 && grid.x >= 512) {  // fewer row tiles than CUs: three column blocks in parallel win
        grid.y = 1;
        hipLaunchKernelGGL((sr_gemm2_qkv_kernel<TIn, MT, KC>), grid, dim3(256), 0, st, a);
        SR_CHECK_LAUNCH("sr_gemm(v2 qkv)");
        return SR_OK;
    }
    hipLaunchKernelGGL((sr_gemm2_kernel<TIn, MT, KC>), grid, dim3(256), 0, st, a);
    SR_CHECK_LAUNCH("sr_gemm(v2)");
    return SR_OK;
}

}  // namespace

// returns 1 when the shape is not one of the specialised ones (caller falls back to the generic kernel)
int sr_gemm_v2_try(const SrGemm& g, hipStream_t st) {
    if (g.compute_dtype != SR_BF16 || g.N % 192 != 0) return 1;
    if (!(g.K == 192 || (g.K == 384 && g.a_dtype == SR_BF16))) return 1;
    Gemm2 a;
    a.g = g;
    a.ws_log2 = a.ntok_log2 = a.hdp_log2 = 0;
    a.div_hw = make_fastdiv(1);
    a.div_nwx = make_fastdiv(1);
    a.div_w = make_fastdiv(1);
    const bool windowed = g.a_map == SR_MAP_WINDOW || g.o_map == SR_MAP_WINDOW;
    if (windowed) {
        a.ws_log2 = ilog2_exact(g.ws);
        if (a.ws_log2 < 0) return 1;
        a.ntok_log2 = 2 * a.ws_log2;
        a.div_hw = make_fastdiv((uint32_t)(g.H * g.W));
        a.div_nwx = make_fastdiv((uint32_t)(g.W / g.ws));
        a.div_w = make_fastdiv((uint32_t)g.W);
    }
    if (g.epi == SR_EPI_QKV_OCA && (g.a_map != SR_MAP_WINDOW || g.shift != 0 || g.oca_pad % 4 != 0)) return 1;
    if (g.epi != SR_EPI_STD) {
        a.hdp_log2 = ilog2_exact(g.hd_p);
        a.ntok_log2 = ilog2_exact(g.ntok);
        if (a.hdp_log2 < 0 || a.ntok_log2 < 0 || (g.heads * g.hd_p) % 192 != 0 || g.out_dtype != SR_BF16) return 1;
    }
    a.acc_from_skip = (g.skip != nullptr && g.act == SR_ACT_NONE && g.out_scale == 1.0f && g.epi == SR_EPI_STD) ? 1 : 0;
    if (g.K == 192) {
        // 128-row tiles; 64-row tiles when those would leave CUs idle (small batches: every launch is one latency chain per workgroup)
        const bool small = (long long)((g.M + 127) / 128) * (g.N / 192) < 512;
        if (g.a_dtype == SR_F32) return small ? launch2<float, 4, 6>(a, st) : launch2<float, 8, 6>(a, st);
        return small ? launch2<bf16, 4, 6>(a, st) : launch2<bf16, 8, 6>(a, st);
    }
    return launch2<bf16, 4, 12>(a, st);
}
