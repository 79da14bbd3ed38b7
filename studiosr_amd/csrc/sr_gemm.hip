// Row GEMM for the transformer half of SwinIR / HAT: y = epi(pro(A) @ W^T).
//
// One workgroup (4 waves) owns M_T = 128 rows (bf16) / 64 rows (fp32) and N_T = 64*NW columns.
//   * the whole [M_T x K] activation tile is staged ONCE into LDS (K <= 384), optionally through a
//     fused LayerNorm (two-pass statistics in registers) and through the window gather
//     (cyclic shift + window partition are pure addressing);
//   * the K loop has no barrier: every wave reads all A fragments from LDS (conflict-free
//     K-group-major image) and streams its own NW weight fragments straight from L2 into
//     registers (one coalesced 1 KiB load per fragment, double buffered);
//   * waves split N, so each weight element is fetched once per workgroup;
//   * the accumulators are produced "swapped" (features on registers, rows on lanes) so that
//     each lane owns 4 consecutive output features of one row -> 8/16-byte stores.  The V third
//     of a QKV projection is produced un-swapped instead and stored transposed ([d][token]) so
//     that the attention kernel can use it as an MFMA operand without any transpose.
#include "sr_common.h"
#include "sr_host.h"

namespace {

template <typename TC>
struct GemmCfg {
    static constexpr int MT = sizeof(TC) == 2 ? 8 : 4;  // 16-row tiles per workgroup
    static constexpr int M_T = MT * 16;
};

template <typename TC, typename TIn>
SR_DEV void stage_rows(const SrGemm& g, Frag<TC>* As, int m0, const WinMap& wm, int lane, int wave) {
    constexpr int M_T = GemmCfg<TC>::M_T;
    const int KG = g.K >> 3;
    const int r8 = lane & 7, kq = lane >> 3;
    for (int rb = wave * 8; rb < M_T; rb += 32) {
        const int row = m0 + rb + r8;
        const bool valid = row < g.M;
        const int srow = valid ? (g.a_map == SR_MAP_WINDOW ? wm(row) : row) : 0;
        const TIn* src = reinterpret_cast<const TIn*>(g.A) + (size_t)srow * g.lda;
        Frag<TC>* dst = As + rb + r8;
        if constexpr (sizeof(TIn) == 4) {
            if (g.ln_gamma != nullptr || g.ln_norm_only) {
                // fused LayerNorm: 8 lanes share one row, lane kq owns K-groups kq, kq+8, ...
                float v[6][8];
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int kg = kq + 8 * i;
                    if (kg < KG && valid) {
                        load8f(reinterpret_cast<const float*>(src) + kg * 8, v[i]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) s += v[i][j];
                }
                s = wave_sum_xor(s, 8);
                s = wave_sum_xor(s, 16);
                s = wave_sum_xor(s, 32);
                const float inv = 1.0f / (float)g.k_real;
                const float mean = s * inv;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int kg = kq + 8 * i;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float d = (kg * 8 + j < g.k_real) ? v[i][j] - mean : 0.f;
                        q += d * d;
                    }
                }
                q = wave_sum_xor(q, 8);
                q = wave_sum_xor(q, 16);
                q = wave_sum_xor(q, 32);
                const float rstd = rsqrtf(q * inv + g.ln_eps);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int kg = kq + 8 * i;
                    if (kg < KG) {
                        float gm[8], bt[8], o[8];
                        if (g.ln_norm_only) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) gm[j] = (kg * 8 + j < g.k_real) ? 1.f : 0.f, bt[j] = 0.f;
                        } else {
                            load8f(g.ln_gamma + kg * 8, gm);
                            load8f(g.ln_beta + kg * 8, bt);
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * gm[j] + bt[j];
                        dst[kg * M_T] = frag_make<TC>(o);
                    }
                }
                continue;
            }
        }
        for (int kg = kq; kg < KG; kg += 8) {
            Frag<TC> f;
            if (valid)
                f = load_group<TC, TIn>(src + kg * 8);
            else
                frag_zero(f);
            dst[kg * M_T] = f;
        }
    }
}

template <typename TC, int NW, bool SWAPPED>
SR_DEV void gemm_body(const SrGemm& g, const Frag<TC>* As, int m0, const WinMap& wm, int lane, int wave) {
    constexpr int MT = GemmCfg<TC>::MT;
    constexpr int M_T = GemmCfg<TC>::M_T;
    const int KC = g.K >> 5;
    const int ar = lane & 15, ag = lane >> 4;
    const int ntile0 = blockIdx.y * (4 * NW) + wave * NW;
    const Frag<TC>* Bp = reinterpret_cast<const Frag<TC>*>(g.Wp) + (size_t)ntile0 * KC * 64 + lane;

    f32x4 acc[MT][NW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4)(0.0f);

    Frag<TC> bc[NW], bn[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) bc[n] = Bp[(size_t)n * KC * 64];

    for (int kc = 0; kc < KC; ++kc) {
        const int kn = (kc + 1 < KC) ? kc + 1 : kc;
#pragma unroll
        for (int n = 0; n < NW; ++n) bn[n] = Bp[((size_t)n * KC + kn) * 64];
        const Frag<TC>* arow = As + (kc * 4 + ag) * M_T + ar;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const Frag<TC> a = arow[m * 16];
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                if constexpr (SWAPPED)
                    mma(bc[n], a, acc[m][n]);
                else
                    mma(a, bc[n], acc[m][n]);
            }
        }
#pragma unroll
        for (int n = 0; n < NW; ++n) bc[n] = bn[n];
    }

    // ------------------------------------------------------------------ epilogue
    if constexpr (SWAPPED) {
        const int HP = g.heads * g.hd_p;
        act_dispatch(g.act, [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = m0 + m * 16 + ar;
            if (row >= g.M) continue;
            const int orow = (g.o_map == SR_MAP_WINDOW) ? wm(row) : row;
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const int col = (ntile0 + n) * 16 + ag * 4;
                f32x4 v = acc[m][n];
                if (g.bias) v += load4(g.bias + col);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = act_ct<ACT>(v[r]) * g.out_scale;
                if (g.epi != SR_EPI_STD) {
                    const int part = col / HP;
                    const int rem = col - part * HP;
                    const int head = rem / g.hd_p, d0 = rem - head * g.hd_p;
                    const int bwin = row / g.ntok, tok = row - bwin * g.ntok;
                    size_t off = (((size_t)bwin * g.heads + head) * g.ntok + tok) * g.hd_p + d0;
                    if (g.epi == SR_EPI_QKV_OCA && part == 1) {  // k -> zero-bordered image order
                        const int pimg = wm(row);
                        const int bb = pimg / wm.hw, rem2 = pimg - bb * wm.hw;
                        const int yy = rem2 / g.W, xx = rem2 - yy * g.W;
                        off = ((((size_t)bb * (g.H + 2 * g.oca_pad) + yy + g.oca_pad) * (g.W + 2 * g.oca_pad) + xx + g.oca_pad) * g.heads + head) * g.hd_p + d0;
                    }
                    void* base = part == 0 ? g.out : g.out_k;
                    if (g.out_dtype == SR_BF16)
                        store4(reinterpret_cast<bf16*>(base) + off, v);
                    else
                        store4(reinterpret_cast<float*>(base) + off, v);
                } else {
                    if (g.skip) v += load4(g.skip + (size_t)orow * g.ldskip + col);
                    if (g.skip2) {  // gated second residual (HAT: + conv_scale * CAB): one fma per element
                        const size_t o2 = (size_t)orow * g.ldskip2 + col;
                        const f32x4 y2 = g.skip2_dtype == SR_BF16 ? load4(reinterpret_cast<const bf16*>(g.skip2) + o2) : load4(reinterpret_cast<const float*>(g.skip2) + o2);
                        const f32x4 gt = load4(g.skip2_gate + (size_t)(orow / g.gate_rows) * g.ld_gate + col);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(y2[r], gt[r], v[r]);
                    }
                    const size_t off = (size_t)orow * g.ldo + col;
                    if (g.out_dtype == SR_BF16)
                        store4(reinterpret_cast<bf16*>(g.out) + off, v);
                    else
                        store4(reinterpret_cast<float*>(g.out) + off, v);
                }
            }
        }
        });
    } else {
        // V third of a QKV projection: lane = feature (ar), registers = 4 consecutive rows.
        const int HP = g.heads * g.hd_p;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row0 = m0 + m * 16 + ag * 4;
            if (row0 >= g.M) continue;
            const int bwin = row0 / g.ntok, tok0 = row0 - bwin * g.ntok;
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const int col = (ntile0 + n) * 16 + ar;
                f32x4 v = acc[m][n];
                if (g.bias) v += g.bias[col];
                const int rem = col - 2 * HP;
                const int head = rem / g.hd_p, d = rem - head * g.hd_p;
                size_t off = (((size_t)bwin * g.heads + head) * g.hd_p + d) * g.ntok + tok0;
                if (g.epi == SR_EPI_QKV_OCA) {  // v -> transposed zero-bordered image (4 consecutive tokens = 4 consecutive x)
                    const int pimg = wm(row0);
                    const int bb = pimg / wm.hw, rem2 = pimg - bb * wm.hw;
                    const int yy = rem2 / g.W, xx = rem2 - yy * g.W;
                    const size_t plane = (size_t)(g.H + 2 * g.oca_pad) * (g.W + 2 * g.oca_pad);
                    off = (((size_t)bb * g.heads + head) * g.hd_p + d) * plane + (size_t)(yy + g.oca_pad) * (g.W + 2 * g.oca_pad) + xx + g.oca_pad;
                }
                if (g.out_dtype == SR_BF16)
                    store4(reinterpret_cast<bf16*>(g.out_vt) + off, v);
                else
                    store4(reinterpret_cast<float*>(g.out_vt) + off, v);
            }
        }
    }
}

template <typename TC, typename TIn, int NW>
__global__ __launch_bounds__(256) void sr_gemm_kernel(SrGemm g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<TC>* As = reinterpret_cast<Frag<TC>*>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * GemmCfg<TC>::M_T;
    WinMap wm;
    wm.H = g.H; wm.W = g.W; wm.ws = g.ws; wm.shift = g.shift; wm.shift_y = g.y_mode == SR_Y_ROLL ? g.shift : 0;
    wm.nwx = g.ws > 0 ? g.W / g.ws : 1;
    wm.ntok = g.ws * g.ws;
    wm.hw = g.H * g.W;

    stage_rows<TC, TIn>(g, As, m0, wm, lane, wave);
    __syncthreads();

    const bool vpart = (g.epi != SR_EPI_STD) && ((int)blockIdx.y * 64 * NW >= 2 * g.heads * g.hd_p);
    if (vpart)
        gemm_body<TC, NW, false>(g, As, m0, wm, lane, wave);
    else
        gemm_body<TC, NW, true>(g, As, m0, wm, lane, wave);
}

template <typename TC, typename TIn, int NW>
int launch_gemm(const SrGemm& g, hipStream_t st) {
    constexpr int M_T = GemmCfg<TC>::M_T;
    const int lds = g.K * M_T * (int)sizeof(TC);
    SR_REQUIRE(lds <= 160 * 1024, "sr_gemm: K=%d needs %d B of LDS", g.K, lds);
    static SrDeviceOnce attr_once;  // one flag per template instantiation, one bit per device
    {
        const hipError_t e = sr_once_per_device(attr_once, [&] { return sr_allow_lds(sr_gemm_kernel<TC, TIn, NW>, 160 * 1024); });
        SR_REQUIRE(e == hipSuccess, "sr_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    dim3 grid((g.M + M_T - 1) / M_T, g.N / (64 * NW));
    hipLaunchKernelGGL((sr_gemm_kernel<TC, TIn, NW>), grid, dim3(256), lds, st, g);
    SR_CHECK_LAUNCH("sr_gemm");
    return SR_OK;
}

template <typename TC, typename TIn>
int dispatch_nw(const SrGemm& g, hipStream_t st) {
    int unit = g.N;
    if (g.epi != SR_EPI_STD) unit = g.heads * g.hd_p;
    if (unit % 192 == 0) return launch_gemm<TC, TIn, 3>(g, st);
    if (unit % 128 == 0) return launch_gemm<TC, TIn, 2>(g, st);
    return launch_gemm<TC, TIn, 1>(g, st);
}

}  // namespace

extern "C" int sr_gemm(const SrGemm* a, void* stream) {
    SR_REQUIRE(a && a->A && a->Wp && a->out, "sr_gemm: null pointer");
    const SrGemm& g = *a;
    SR_REQUIRE(g.M > 0 && g.K > 0 && g.K % 32 == 0 && g.N > 0 && g.N % 64 == 0, "sr_gemm: bad M/K/N %d/%d/%d", g.M, g.K, g.N);
    SR_REQUIRE(g.lda >= g.K && g.lda % 8 == 0, "sr_gemm: lda %d", g.lda);
    if (g.ln_gamma || g.ln_norm_only) {
        SR_REQUIRE(g.a_dtype == SR_F32 && (g.ln_norm_only || g.ln_beta) && g.K <= 384 && g.k_real > 0 && g.k_real <= g.K, "sr_gemm: LayerNorm prologue needs fp32 A and K <= 384");
    }
    if (g.a_map == SR_MAP_WINDOW || g.o_map == SR_MAP_WINDOW) {
        SR_REQUIRE(g.ws > 0 && g.H % g.ws == 0 && g.W % g.ws == 0 && g.shift >= 0 && g.shift < g.ws && g.M % (g.H * g.W) == 0,
                   "sr_gemm: bad window geometry H=%d W=%d ws=%d shift=%d M=%d", g.H, g.W, g.ws, g.shift, g.M);
    }
    if (g.epi != SR_EPI_STD) {
        if (g.epi == SR_EPI_QKV_OCA)
            SR_REQUIRE(g.a_map == SR_MAP_WINDOW && g.shift == 0 && g.oca_pad >= 0 && g.ws % 4 == 0 && g.ntok == g.ws * g.ws, "sr_gemm: bad OCA epilogue geometry");
        SR_REQUIRE(g.out_k && g.out_vt && g.heads > 0 && g.hd_p % 16 == 0 && g.N == 3 * g.heads * g.hd_p && g.ntok % 16 == 0 && g.M % g.ntok == 0 &&
                       (g.heads * g.hd_p) % 64 == 0,
                   "sr_gemm: bad QKV epilogue geometry");
        SR_REQUIRE(g.skip == nullptr && g.skip2 == nullptr, "sr_gemm: QKV epilogue takes no residual");
    } else {
        SR_REQUIRE(g.ldo >= g.N, "sr_gemm: ldo %d < N %d", g.ldo, g.N);
        SR_REQUIRE(!g.skip || g.ldskip >= g.N, "sr_gemm: ldskip");
        SR_REQUIRE(!g.skip2 || (g.skip2_gate && g.ldskip2 >= g.N && g.ld_gate >= g.N && g.gate_rows > 0), "sr_gemm: gated second residual: gate, ldskip2, ld_gate >= N, gate_rows > 0");
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    {
        const int r = sr_gemm_v2_try(g, st);
        if (r != 1) return r;
    }
    if (g.compute_dtype == SR_BF16) {
        if (g.a_dtype == SR_F32) return dispatch_nw<bf16, float>(g, st);
        return dispatch_nw<bf16, bf16>(g, st);
    }
    SR_REQUIRE(g.a_dtype == SR_F32, "sr_gemm: fp32 / bf16x3 compute needs fp32 A");
    if (g.compute_dtype == SR_BF16X3) return dispatch_nw<bf3, float>(g, st);
    return dispatch_nw<float, float>(g, st);
}
