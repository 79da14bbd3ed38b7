// Body of sr_cab_kernel (sr_cab.hip: HAT's CAB convs in one launch) as a device function, shared with sr_hab_mid.hip (the CAB and the window attention of
// a HAB as ONE launch).  SR_CAB_PH (K phases of conv1: 1 = 88 KiB of LDS, 2 = 52 KiB) is fixed per translation unit before this header is included.
// The header can be included MORE THAN ONCE per translation unit with different SR_CAB_TOH (tile heights) inside different namespaces: the includer defines
// SR_CAB_NS_BEGIN / SR_CAB_NS_END (default: the anonymous namespace) and SR_CAB_TOH before each inclusion (sr_hab_mid.hip: 6 rows for small launches, 8 for large).
#include "sr_common.h"
#include "sr_host.h"

#ifndef SR_CAB_NS_BEGIN
#define SR_CAB_NS_BEGIN namespace {
#define SR_CAB_NS_END }
#endif
SR_CAB_NS_BEGIN

constexpr int CI = 192, CM = 64, CO = 192;
constexpr int KG_IN = CI / 8, KC_IN = CI / 32, KCT1 = 9 * KC_IN;   // 24 K-groups, 6 chunks per tap, 54 steps
constexpr int KG_MID = CM / 8, KC_MID = CM / 32, KCT2 = 9 * KC_MID;  // 8 K-groups, 2 chunks per tap, 18 steps
#ifndef SR_CAB_TOH
#define SR_CAB_TOH 6
#endif
constexpr int TOW = 14, TOH = SR_CAB_TOH;  // output tile; rows 4 (70 KiB of LDS = two workgroups per CU) / 6 / 8: HAT x4 b4 3.49 / 3.41 / 3.42 ms, b16 9.66 / 9.42 / 9.24 ms, b1 2.48 / 2.48 / 2.64 ms
constexpr int TIW = 16, TIH = TOH + 2;   // intermediate tile (one MFMA row tile per row)
constexpr int TINW = 18, TINH = TOH + 4; // input halo
constexpr int IN_ROWS = (TINW * TINH + 7) / 8 * 8;  // halo pixels (18 x 10 = 180 -> 184), padded to a multiple of 8
constexpr int IN_RS = IN_ROWS + 1;    // odd cell stride: the 8 K-groups of one pixel (8 adjacent lanes of the staging writes) hit 8 bank groups
constexpr int MID_ROWS = (2 + TIW * TIH + 7) / 8 * 8;  // 1 margin cell + the intermediate pixels + 1 margin cell, padded to a multiple of 8
constexpr int SR_CAB_NS_RING1 = 6, SR_CAB_NS_RING2 = 4;  // weight rings (bf16; the split-operand form: 3 / 3)
// K phases of conv1: PH = 2 keeps only 96 of the 192 input channels of the halo resident (52 KiB of LDS instead of 88: two workgroups per CU
// where the launch is several residency rounds, e.g. HAT x4 b16: 880 workgroups); the second phase's halo loads fly under the first phase's MFMAs
#ifndef SR_CAB_PH
#define SR_CAB_PH 1
#endif
constexpr int PH = SR_CAB_PH;
constexpr int KG_RES = KG_IN / PH, KC_PH = KC_IN / PH;   // resident K-groups, chunks per tap and phase
constexpr int LP = KG_RES / 3, PPI = 64 / LP;            // staging: lanes per pixel (8 / 4), pixels per wave instruction (8 / 16); 3 passes per pixel group
constexpr int UNITS = (IN_ROWS + PPI - 1) / PPI * 3, NU = (UNITS + 3) / 4;  // (pixel group, pass) units; per wave
static_assert(KG_IN % PH == 0 && KG_RES % 3 == 0 && (LP == 8 || LP == 4), "phase geometry");
constexpr int LDS_BYTES = (KG_RES * IN_RS + KG_MID * MID_ROWS) * 16;
constexpr int LDS_BYTES_X3 = (KG_RES * IN_RS + KG_MID * MID_ROWS) * 32;  // split-operand images
constexpr int R1 = TIH / 2;           // intermediate rows per wave pair in conv1
static_assert(TINW * TINH <= IN_ROWS && 2 + TIW * TIH <= MID_ROWS && TIH % 2 == 0, "image sizes");

#ifndef SR_CAB_WGS
#define SR_CAB_WGS (SR_CAB_PH == 1 ? 1 : 2)
#endif
// the work of one workgroup (256 threads, LDS_BYTES of dynamic LDS at `smem`): tile `block_id` of B * tiles_y * tiles_x
// TC = bf16 (x, y bf16) or bf3 (round 5, compute type SR_BF16X3 = precision "fp32x3": every operand a hi + lo bf16 pair, 32-byte image cells -- LDS_BYTES_X3, which needs the
// two K phases; x, y fp32; erf GELU; shorter weight rings of the wider fragments)
template <typename TC = bf16, typename TIn = bf16, typename TOut = bf16>
SR_DEV void cab_block(const SrCab& c, const int block_id, char* smem) {
    constexpr bool X3 = sizeof(Frag<TC>) == 32;
    constexpr int RING1 = X3 ? 3 : SR_CAB_NS_RING1, RING2 = X3 ? 3 : SR_CAB_NS_RING2;
    Frag<TC>* Ain = reinterpret_cast<Frag<TC>*>(smem);   // [KG_RES][IN_RS]
    Frag<TC>* Amid = Ain + KG_RES * IN_RS;                   // [KG_MID][MID_ROWS], pixel p of the 16 x 8 tile at cell 1 + p

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ar = lane & 15, ag = lane >> 4;
    const int tiles_x = (c.W + TOW - 1) / TOW, tiles_y = (c.H + TOH - 1) / TOH;
    int t = block_id;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * TOW, y0 = ty * TOH;  // output tile origin; intermediate origin (y0 - 1, x0 - 1), halo origin (y0 - 2, x0 - 2)

    const Frag<TC>* W1 = reinterpret_cast<const Frag<TC>*>(c.w1p) + (size_t)(wn * 2) * KCT1 * 64 + lane;
    const Frag<TC>* W2 = reinterpret_cast<const Frag<TC>*>(c.w2p) + (size_t)(wave * 3) * KCT2 * 64 + lane;
    // conv1 walks K phase-major: step t = (phase, tap, chunk in phase) -> packed chunk index tap * KC_IN + phase * KC_PH + chunk (PH = 1: t itself)
    auto chunk_of = [](int t) { return ((t % (9 * KC_PH)) / KC_PH) * KC_IN + (t / (9 * KC_PH)) * KC_PH + t % KC_PH; };
    Frag<TC> r1[RING1][2];
#pragma unroll
    for (int s = 0; s < RING1 - 1; ++s)
#pragma unroll
        for (int n = 0; n < 2; ++n) r1[s][n] = W1[((size_t)n * KCT1 + chunk_of(s)) * 64];

    // ---- stage the input halo: PPI pixels x LP K-groups per wave instruction, K-group on the fast lane axis (the LP lanes of a pixel read 16 LP
    //      contiguous bytes); unit u = (pixel group, pass), UNITS over 4 waves
    const int kq = lane & (LP - 1), rp = lane / LP;
    const TIn* xin = reinterpret_cast<const TIn*>(c.x);
    auto issue = [&](int ph, int k0, int nk, Frag<TC>* f, bool* valid) {  // units k0 .. k0 + nk - 1 of this wave
#pragma unroll
        for (int k = 0; k < nk; ++k) {
            const int u = 4 * (k0 + k) + wave;
            const int pg = u / 3, j = u - pg * 3;
            const int p = pg * PPI + rp;
            const int py = p / TINW, px = p - py * TINW;
            const int gy = y0 - 2 + py, gx = x0 - 2 + px;
            valid[k] = u < UNITS && p < TINW * TINH && gy >= 0 && gy < c.H && gx >= 0 && gx < c.W;
            const TIn* src = xin + ((size_t)(b * c.H + (valid[k] ? gy : 0)) * c.W + (valid[k] ? gx : 0)) * CI + (ph * KG_RES + j * LP + kq) * 8;
            f[k] = load_group<TC, TIn>(src);
        }
    };
    auto commit = [&](int k0, int nk, const Frag<TC>* f, const bool* valid) {
#pragma unroll
        for (int k = 0; k < nk; ++k) {
            const int u = 4 * (k0 + k) + wave;
            const int pg = u / 3, j = u - pg * 3;
            const int p = pg * PPI + rp;
            if (u < UNITS && p < IN_ROWS) Ain[(j * LP + kq) * IN_RS + p] = frag_keep_if(valid[k], f[k]);
        }
    };
    {
        constexpr int NP = PH == 1 ? 6 : NU;  // loads in flight
#pragma unroll
        for (int k0 = 0; k0 < NU; k0 += NP) {
            Frag<TC> f[NP];
            bool valid[NP];
            issue(0, k0, NP, f, valid);
            commit(k0, NP, f, valid);
        }
        if (threadIdx.x < 2 * KG_MID) {  // margin cells of the intermediate image (read by discarded edge columns only; keep them finite)
            Frag<TC> z;
            frag_zero(z);
            Amid[(threadIdx.x >> 1) * MID_ROWS + ((threadIdx.x & 1) ? 1 + TIW * TIH : 0)] = z;
        }
    }
    Frag<TC> f1[PH == 1 ? 1 : NU];  // phase 1 of the halo: requested now, written to LDS between the phases
    bool v1[PH == 1 ? 1 : NU];
    if constexpr (PH == 2) issue(1, 0, NU, f1, v1);
    __syncthreads();

    Frag<TC> r2[RING2][3];
    // ---- conv1 + bias + GELU -> intermediate image.  Wave (wm, wn): rows [R1 wm, R1 wm + R1) x channels [32 wn, 32 wn + 32)
    {
        f32x4 acc[R1][2];
#pragma unroll
        for (int m = 0; m < R1; ++m) {
            acc[m][0] = (f32x4)(0.0f);
            acc[m][1] = (f32x4)(0.0f);
        }
        const Frag<TC>* abase0 = Ain + (wm * R1) * TINW + ar + ag * IN_RS;
#pragma unroll
        for (int tt = 0; tt < KCT1; ++tt) {
            const int tp = tt % (9 * KC_PH), tap = tp / KC_PH, kc = tp - tap * KC_PH;
            if (PH == 2 && tt == 9 * KC_PH) {  // phase boundary: every wave is done with the first 96 channels of the halo
                __syncthreads();
                commit(0, NU, f1, v1);
                __syncthreads();
            }
            if (tt + RING1 - 1 < KCT1) {
#pragma unroll
                for (int n = 0; n < 2; ++n) r1[(tt + RING1 - 1) % RING1][n] = W1[((size_t)n * KCT1 + chunk_of(tt + RING1 - 1)) * 64];
            } else if (tt + RING1 - 1 - KCT1 < RING2 - 1) {  // tail of conv1: start conv2's weight stream
                const int s = tt + RING1 - 1 - KCT1;
#pragma unroll
                for (int n = 0; n < 3; ++n) r2[s][n] = W2[((size_t)n * KCT2 + s) * 64];
            }
            const Frag<TC>* arow = abase0 + (tap / 3) * TINW + (tap % 3) + kc * 4 * IN_RS;
#pragma unroll
            for (int m = 0; m < R1; ++m) {
                const Frag<TC> a = arow[m * TINW];
                mma(r1[tt % RING1][0], a, acc[m][0]);
                mma(r1[tt % RING1][1], a, acc[m][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const f32x4 bias0 = load4(c.b1 + (wn * 2) * 16 + ag * 4), bias1 = load4(c.b1 + (wn * 2 + 1) * 16 + ag * 4);
        const int gx = x0 - 1 + ar;
#pragma unroll
        for (int m = 0; m < R1; ++m) {
            const int iy = wm * R1 + m, gy = y0 - 1 + iy;
            const bool inside = gy >= 0 && gy < c.H && gx >= 0 && gx < c.W;  // outside the image the intermediate is conv2's zero padding
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                f32x4 v = acc[m][n] + (n == 0 ? bias0 : bias1);
                const int kg = (wn * 2 + n) * 2 + (ag >> 1);
                if constexpr (X3) {
                    bf16x4 h, l;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float g = inside ? gelu_fast(v[r]) : 0.0f;
                        h[r] = (bf16)g;
                        l[r] = (bf16)(g - (float)h[r]);
                    }
                    char* cell = reinterpret_cast<char*>(Amid + kg * MID_ROWS + 1 + iy * TIW + ar);
                    *reinterpret_cast<bf16x4*>(cell + (ag & 1) * 8) = h;
                    *reinterpret_cast<bf16x4*>(cell + 16 + (ag & 1) * 8) = l;
                    continue;
                }
                if (c.mid_pre && inside && iy >= 1 && iy <= TOH && ar >= 1 && ar <= TOW) {  // the tile's own pixels (every pixel once): the pre-activation for the backward
                    bf16x4 pre;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pre[r] = (bf16)v[r];
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(c.mid_pre) + ((size_t)(b * c.H + gy) * c.W + gx) * CM + (wn * 2 + n) * 16 + ag * 4) = pre;
                }
                bf16x4 o;
                if (c.bwd_pre) {  // backward form: mid = conv(dy) * GELU'(pre); side outputs dmid and GELU(pre) at the tile's own pixels
                    const size_t pix = ((size_t)(b * c.H + (inside ? gy : 0)) * c.W + (inside ? gx : 0)) * CM + (wn * 2 + n) * 16 + ag * 4;
                    const bf16x4 pre4 = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(c.bwd_pre) + pix);
                    bf16x4 g4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float Phi, phi;
                        const float pr = (float)pre4[r];
                        gauss(pr, Phi, phi);
                        o[r] = (bf16)(inside ? v[r] * (Phi + pr * phi) : 0.0f);
                        g4[r] = (bf16)(pr * Phi);
                    }
                    if (inside && iy >= 1 && iy <= TOH && ar >= 1 && ar <= TOW) {
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(c.bwd_dmid) + pix) = o;
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(c.bwd_g) + pix) = g4;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (bf16)(inside ? gelu_fast(v[r]) : 0.0f);
                }
                char* dst = reinterpret_cast<char*>(Amid + kg * MID_ROWS + 1 + iy * TIW + ar) + (ag & 1) * 8;
                *reinterpret_cast<bf16x4*>(dst) = o;
            }
        }
    }
    __syncthreads();

    // ---- conv2 + bias on output rows oy = 1 .. 6 of the intermediate tile; 16 columns per row, the inner 14 are kept.  Wave w: channels [48 w, +48)
    {
        f32x4 acc[TOH][3];
#pragma unroll
        for (int m = 0; m < TOH; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][n] = (f32x4)(0.0f);
        // tap (ky, kx) of output (oy, ox) reads intermediate (oy + ky - 1, ox + kx - 1) = cell 1 + (oy + ky - 1) * 16 + ox + kx - 1, oy = 1 + m
        const Frag<TC>* abase0 = Amid + ar + ag * MID_ROWS;
#pragma unroll
        for (int tt = 0; tt < KCT2; ++tt) {
            const int tap = tt / KC_MID, kc = tt - tap * KC_MID;
            if (tt + RING2 - 1 < KCT2) {
#pragma unroll
                for (int n = 0; n < 3; ++n) r2[(tt + RING2 - 1) % RING2][n] = W2[((size_t)n * KCT2 + tt + RING2 - 1) * 64];
            }
            const Frag<TC>* arow = abase0 + (tap / 3) * TIW + (tap % 3) + kc * 4 * MID_ROWS;
#pragma unroll
            for (int m = 0; m < TOH; ++m) {
                const Frag<TC> a = arow[m * TIW];
#pragma unroll
                for (int n = 0; n < 3; ++n) mma(r2[tt % RING2][n], a, acc[m][n]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 bias[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) bias[n] = load4(c.b2 + (wave * 3 + n) * 16 + ag * 4);
        const int gx = x0 - 1 + ar;
        const bool xin_tile = ar >= 1 && ar <= TOW && gx < c.W;
        f32x4 pool[3] = {(f32x4)(0.0f), (f32x4)(0.0f), (f32x4)(0.0f)};
        TOut* y = reinterpret_cast<TOut*>(c.y);
        // Coalesced store: one row of 16 pixels x this wave's 48 channels at a time through a wave-private fp32 tile in the (dead) input image:
        // accumulator layout in (pixel on the lane axis), 16-byte pieces (8 bf16 / 4 fp32) out with adjacent lanes on adjacent addresses of ONE pixel.
        constexpr int S = 48 * 4 + 16;   // bytes per pixel row of the private tile
        constexpr int NV = 16 / (int)sizeof(TOut);               // channels per 16-byte piece
        constexpr int PP = 48 / NV, NI = (16 * PP + 63) / 64;    // 16-byte pieces per pixel (48 channels), store instructions per row
        char* priv = smem + wave * (16 * S);
#pragma unroll
        for (int m = 0; m < TOH; ++m) {
            const int gy = y0 + m;
            const bool ok = xin_tile && gy < c.H;
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const f32x4 v = acc[m][n] + bias[n];
                if (ok) pool[n] += v;
                *reinterpret_cast<f32x4*>(priv + ar * S + (n * 16 + ag * 4) * 4) = v;
            }
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int idx = k * 64 + lane, px = idx / PP, pc = idx - px * PP;
                if (idx < 16 * PP) {
                    const int gxp = x0 - 1 + px;
                    const float* src = reinterpret_cast<const float*>(priv + px * S) + pc * NV;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
                    if (px >= 1 && px <= TOW && gxp < c.W && gy < c.H) {
                        TOut* dst = y + ((size_t)(b * c.H + gy) * c.W + gxp) * CO + wave * 48 + pc * NV;
                        if constexpr (sizeof(TOut) == 2) {
                            const f32x4 hi = *reinterpret_cast<const f32x4*>(src + 4);
                            bf16x8 o;
#pragma unroll
                            for (int q = 0; q < 4; ++q) o[q] = (bf16)lo[q], o[4 + q] = (bf16)hi[q];
                            *reinterpret_cast<bf16x8*>(dst) = o;
                        } else {
                            store4(dst, lo);
                        }
                    }
                }
            }
        }
        if (c.pool_partial) {  // per-tile channel sums over the valid pixels: deterministic, no atomics
            const int n_slots = tiles_x * tiles_y;
            const int slot = ty * tiles_x + tx;
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                f32x4 p = pool[n];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = p[r];
                    s += __shfl_xor(s, 1, 64);
                    s += __shfl_xor(s, 2, 64);
                    s += __shfl_xor(s, 4, 64);
                    s += __shfl_xor(s, 8, 64);
                    p[r] = s;
                }
                if (ar == 0) store4(c.pool_partial + ((size_t)b * n_slots + slot) * CO + (wave * 3 + n) * 16 + ag * 4, p);
            }
        }
    }
}

static inline int cab_check(const SrCab* p, const char* who) {
    SR_REQUIRE(p && p->x && p->w1p && p->b1 && p->w2p && p->b2 && p->y, "%s: null pointer (CAB operands)", who);
    const SrCab& c = *p;
    SR_REQUIRE(c.B > 0 && c.H > 0 && c.W > 0 && c.Cin_p == CI && c.Cmid_p == CM && c.Cout_p == CO && (c.dtype == SR_BF16 || c.dtype == SR_BF16X3),
               "%s: bad CAB geometry (192 -> 64 -> 192 padded channels; bf16, or SR_BF16X3 on fp32 tensors)", who);
    SR_REQUIRE(c.x != c.y, "%s: y must not alias x (halo reads)", who);
    SR_REQUIRE(!c.bwd_pre || (c.bwd_dmid && c.bwd_g && !c.mid_pre && c.dtype == SR_BF16), "%s: the backward form needs bwd_dmid and bwd_g, bf16, and no mid_pre", who);
    SR_REQUIRE((long)(((c.W + TOW - 1) / TOW) * ((c.H + TOH - 1) / TOH)) * c.B < (1l << 31), "%s: too many tiles", who);
    return SR_OK;
}

SR_CAB_NS_END
