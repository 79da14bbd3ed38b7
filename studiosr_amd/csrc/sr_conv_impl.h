// 3x3 convolution (stride 1, zero pad 1) as an im2col-free implicit GEMM, NHWC.
//
// One workgroup (4 waves) produces a TH x 16 pixel tile of N_T = WN*NW*16 output channels:
//   * the (TH+2) x 18 input halo tile, ALL input channels, is staged once into LDS in the
//     K-group-major image of sr_common.h (zero padding at the image border is written here);
//   * an MFMA row tile is 16 horizontally adjacent output pixels, so its operand for tap
//     (ky, kx) is 16 CONSECUTIVE rows of that image starting at (y+ky)*18 + kx: the nine
//     shifted views are read from the one tile, conflict-free, with no im2col buffer;
//   * weights are streamed per (tap, 32-channel chunk) straight from L2 into registers in
//     fragment order (double buffered), never through LDS; waves split N (and M for narrow N);
//   * no barrier in the 9*Cin/32-step main loop;
//   * epilogue: bias, ReLU/LeakyReLU/GELU, res_scale, residual add and one of three stores:
//     NHWC, NHWC through nn.PixelShuffle (the weight rows were permuted at pack time so that a
//     lane's 4 consecutive accumulators are 4 consecutive channels of ONE shuffled pixel), or the
//     final un-normalise + crop + NCHW fp32 image.  Optionally emits per-tile channel sums for
//     channel attention (deterministic, no atomics).
//
// This header holds the kernel template and its launch / dispatch templates; each (compute type, input type, tile height) variant is
// instantiated in its own translation unit (sr_conv_v*.hip) so that a clean build compiles them in parallel (the one-TU build took 7 min).
#pragma once
#include "sr_common.h"
#include "sr_host.h"

#include <cstdlib>

// weight ring depth of the unrolled K walk: chunk t + SR_CONV_RING - 1 is requested while chunk t runs
#ifndef SR_CONV_RING
#define SR_CONV_RING 3
#endif

namespace sr_conv_impl {

static __device__ unsigned long long sr_dbg_conv[16];  // per translation unit (diagnostic STAMPS builds only)

constexpr int HALO_W = 18;

#define STAMP(i) SR_STAMP(sr_dbg_conv, i)

// operand kinds that take the unrolled / coalesced forms: everything on the bf16 matrix cores (bf16, and since round 5 the split-operand bf3); the exact-fp32
// parity path keeps the generic loop
template <typename TC>
constexpr bool kMatrix16 = !std::is_same<TC, float>::value;

// halo rows of the LDS image: bf16 stages 64 rows per step (two passes of 32), so its images are whole steps
template <int TH, int ESZ = 4>
struct ConvGeo {
    static constexpr int HH = TH + 2;
    static constexpr int ROWS = ESZ == 2 ? ((HH * HALO_W + 63) / 64) * 64 : ((HH * HALO_W + 15) / 16) * 16;
    static constexpr int RS = ROWS + 1;
};

// Coalesced epilogue (bf16-compute kernels): a wave's accumulator tiles have the PIXEL on the lane axis, so storing them directly puts
// adjacent lanes on different pixels = different cache lines (8 or 16 useful bytes per line access).  Instead a wave transposes one
// group of row tiles through a private fp32 LDS tile [pixel][its NW * 16 channels] (row stride S: conflict-free b128 writes) and
// stores 16-byte pieces with adjacent lanes on adjacent addresses of ONE pixel; the residual is read the same way.
template <int NW>
struct EpiGeo {
    static constexpr int S = NW * 64 + 16;       // bytes per pixel row of the private tile
    static constexpr int GMAX = 2;               // row tiles per group (so that 16 * pieces * G is a multiple of 64)
    static constexpr int PRIV = GMAX * 16 * S;   // bytes per wave
};

template <typename TC, typename TIn, int TH, int WM, int WN, int NW, int KCS>
__global__ __launch_bounds__(256) void sr_conv3x3_kernel(SrConv3x3 c, int xcd_order) {
    static_assert(WM * WN == 4 && TH % WM == 0, "wave grid");
    constexpr int ROWS = ConvGeo<TH, kMatrix16<TC> ? 2 : 4>::ROWS;
    // row stride of the K-group-major image in cells: ROWS + 1, so that the staging writes of ONE pixel's 8 K-groups (8 adjacent lanes,
    // see below) fall into different banks; the fragment reads (16 consecutive rows of one K-group) do not care
    constexpr int RS = ConvGeo<TH, kMatrix16<TC> ? 2 : 4>::RS;
    constexpr int HH = ConvGeo<TH>::HH;
    constexpr int MTW = TH / WM;  // row tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<TC>* As = reinterpret_cast<Frag<TC>*>(smem);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_x = (c.W + 15) >> 4;
    const int tiles_y = (c.H + TH - 1) / TH;
    int t = blockIdx.x;
    if (xcd_order) {  // neighbouring tiles (shared halo rows) on one XCD: each residue class mod 8 of the block ids takes a contiguous range of tiles (see sr_conv_big.hip)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = t & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    }
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * 16, y0 = ty * TH;
    const int KG = c.Cin_p >> 3, KC = c.Cin_p >> 5;
    STAMP(0);

    // ---- stage the halo tile: 8 pixels x 8 K-groups per wave instruction with the K-GROUP on the fast lane axis -- the 8 lanes of a pixel
    //      read 128 (bf16) / 256 (fp32) CONTIGUOUS bytes.  (Pixel-fastest lanes made every quad of adjacent lanes touch four cache
    //      lines: the vector-memory path then spends four tag cycles per quad -- profiles/r03_conv_ta_counters.txt.)
    {
        const int kq = lane & 7, r8 = lane >> 3;
        const TIn* xin = reinterpret_cast<const TIn*>(c.x);
        if constexpr (KCS > 0 && (KCS * 4) % 8 == 0) {
            // compile-time channel count (multiple of 64): two row passes per step, all of their loads issued before the first LDS write
            // (out-of-image pixels read a clamped address and are zeroed by a select: no divergent branches)
            constexpr int KI = KCS * 4 / 8;
            static_assert(ROWS % 64 == 0, "two row passes per step");
            for (int pb = wave * 8; pb < ROWS; pb += 64) {
                Frag<TC> f[2][KI];
                bool valid[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int p = pb + u * 32 + r8;
                    const int py = p / HALO_W, px = p - py * HALO_W;
                    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                    valid[u] = p < HH * HALO_W && gy >= 0 && gy < c.H && gx >= 0 && gx < c.W;
                    const TIn* src = xin + ((size_t)(b * c.H + (valid[u] ? gy : 0)) * c.W + (valid[u] ? gx : 0)) * c.Cin_p;
#pragma unroll
                    for (int i = 0; i < KI; ++i) f[u][i] = load_group<TC, TIn>(src + (kq + 8 * i) * 8);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int p = pb + u * 32 + r8;
#pragma unroll
                    for (int i = 0; i < KI; ++i) As[(kq + 8 * i) * RS + p] = frag_keep_if(valid[u], f[u][i]);
                }
            }
        } else {
            for (int pb = wave * 8; pb < ROWS; pb += 32) {
                const int p = pb + r8;
                const int py = p / HALO_W, px = p - py * HALO_W;
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                const bool valid = p < HH * HALO_W && gy >= 0 && gy < c.H && gx >= 0 && gx < c.W;
                const TIn* src = xin + ((size_t)(b * c.H + (valid ? gy : 0)) * c.W + (valid ? gx : 0)) * c.Cin_p;
                for (int kg = kq; kg < KG; kg += 8) {
                    Frag<TC> f;
                    if (valid)
                        f = load_group<TC, TIn>(src + kg * 8);
                    else
                        frag_zero(f);
                    As[kg * RS + p] = f;
                }
            }
        }
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);

    // ---- main loop over (tap, channel chunk)
    const int wm = wave / WN, wn = wave - wm * WN;
    const int ar = lane & 15, ag = lane >> 4;
    const int ntile0 = blockIdx.y * (WN * NW) + wn * NW;
    const int KCT = 9 * KC;
    const Frag<TC>* Bp = reinterpret_cast<const Frag<TC>*>(c.Wp) + (size_t)ntile0 * KCT * 64 + lane;

    // plain residual convs (no activation / scale, NHWC out): the skip tile is the initial accumulator, fetched now
    // coalesced epilogue (see EpiGeo): NHWC, or PixelShuffle when every wave's NW * 16 channels are one sub-pixel's contiguous channels
    bool coal = false;
    if constexpr (kMatrix16<TC> && MTW % 2 == 0)
        coal = c.out_mode == SR_OUT_NHWC || (c.out_mode == SR_OUT_PIXEL_SHUFFLE && c.cps_p % (NW * 16) == 0);
    const bool acc_from_skip = c.skip && c.act == SR_ACT_NONE && c.out_scale == 1.0f && c.out_mode == SR_OUT_NHWC && !c.pool_partial;
    f32x4 acc[MTW][NW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int yy = y0 + wm * MTW + m, xx = x0 + ar;
        const bool inb0 = (yy < c.H) && (xx < c.W);
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            acc[m][n] = (f32x4)(0.0f);
            if (acc_from_skip && inb0) {
                const size_t off = ((size_t)(b * c.H + yy) * c.W + xx) * c.Cout_p + (ntile0 + n) * 16 + ag * 4;
                acc[m][n] = c.skip_dtype == SR_BF16 ? load4(reinterpret_cast<const bf16*>(c.skip) + off) : load4(reinterpret_cast<const float*>(c.skip) + off);
            }
        }
    }

    f32x4 bias_r[NW];  // requested before the MFMA loop: in the epilogue its L2 round trip would be fully exposed
#pragma unroll
    for (int n = 0; n < NW; ++n) bias_r[n] = c.bias ? load4(c.bias + (ntile0 + n) * 16 + ag * 4) : (f32x4)(0.0f);

    if constexpr (KCS > 0) {
        // Fully unrolled (tap, chunk) walk: K-chunk count is a compile-time constant, so the 3-slot weight ring
        // (chunk t in slot t % 3, fetched 2 chunks = 2*MTW*NW MFMAs ahead) and every accumulator stay in fixed registers.
        constexpr int RING = SR_CONV_RING;
        constexpr int KCTS = 9 * KCS;
        // Walk order of the K-chunks.  256 input channels are summed as two phases of 128 (all nine taps of channels 0..127,
        // then of 128..255) -- the order sr_conv_big.hip needs for its two-phase halo tile -- so that a
        // pixel gets the SAME bits whichever of the two kernels the launch size selects; every other channel count is tap-major in both.
        constexpr int PHS = KCS == 8 ? 2 : 1, KCPH = KCS / PHS;
        auto tap_of = [](int t) { return (t % (9 * KCPH)) / KCPH; };
        auto kc_of = [](int t) { return (t / (9 * KCPH)) * KCPH + t % KCPH; };
        auto chunk_of = [&](int t) { return tap_of(t) * KCS + kc_of(t); };  // index into the packed weights (k = tap * Cin + c)
        Frag<TC> br[RING][NW];
#pragma unroll
        for (int t = 0; t < RING - 1; ++t)
#pragma unroll
            for (int n = 0; n < NW; ++n) br[t][n] = Bp[((size_t)n * KCTS + chunk_of(t)) * 64];
        const Frag<TC>* abase0 = As + (wm * MTW) * HALO_W + ar + ag * RS;
        if constexpr (KCS >= 2) {
            // activation fragments are double buffered across chunks: the LDS reads of chunk t+1 are issued before the MFMAs of
            // chunk t, so their latency (the only thing a one-wave-per-SIMD workgroup cannot hide otherwise) is off the critical path
            Frag<TC> af[2][MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) af[0][m] = abase0[m * HALO_W];
#pragma unroll
            for (int t = 0; t < KCTS; ++t) {
                if (t + RING - 1 < KCTS) {
#pragma unroll
                    for (int n = 0; n < NW; ++n) br[(t + RING - 1) % RING][n] = Bp[((size_t)n * KCTS + chunk_of(t + RING - 1)) * 64];
                }
                if (t + 1 < KCTS) {
                    const int tn = t + 1, tap = tap_of(tn), kc = kc_of(tn);
                    const Frag<TC>* arow = abase0 + (tap / 3) * HALO_W + (tap % 3) + kc * 4 * RS;
#pragma unroll
                    for (int m = 0; m < MTW; ++m) af[tn & 1][m] = arow[m * HALO_W];
                }
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int n = 0; n < NW; ++n) mma(br[t % RING][n], af[t & 1][m], acc[m][n]);
                __builtin_amdgcn_sched_barrier(0);  // one fence per chunk: bounds live ranges, keeps the ring order
            }
        } else {
            // K = 9 x 32 only (RGB ingest conv): store-bound, keep the register count low enough for two workgroups per CU
#pragma unroll
            for (int t = 0; t < KCTS; ++t) {
                if (t + RING - 1 < KCTS) {
#pragma unroll
                    for (int n = 0; n < NW; ++n) br[(t + RING - 1) % RING][n] = Bp[((size_t)n * KCTS + t + RING - 1) * 64];
                }
                const Frag<TC>* arow = abase0 + (t / 3) * HALO_W + (t % 3);
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const Frag<TC> a = arow[m * HALO_W];
#pragma unroll
                    for (int n = 0; n < NW; ++n) mma(br[t % RING][n], a, acc[m][n]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        // generic channel count: run-time loop, weights double-buffered one chunk ahead
        Frag<TC> bc[NW], bn[NW];
#pragma unroll
        for (int n = 0; n < NW; ++n) bc[n] = Bp[(size_t)n * KCT * 64];
        int chunk = 0;
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const Frag<TC>* abase = As + (wm * MTW + ky) * HALO_W + kx + ar + ag * RS;
            for (int kc = 0; kc < KC; ++kc, ++chunk) {
                const int cn = chunk + 1 < KCT ? chunk + 1 : chunk;
#pragma unroll
                for (int n = 0; n < NW; ++n) bn[n] = Bp[((size_t)n * KCT + cn) * 64];
                const Frag<TC>* arow = abase + kc * 4 * RS;
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const Frag<TC> a = arow[m * HALO_W];
#pragma unroll
                    for (int n = 0; n < NW; ++n) mma(bc[n], a, acc[m][n]);
                }
#pragma unroll
                for (int n = 0; n < NW; ++n) bc[n] = bn[n];
            }
        }
    }

    STAMP(3);
    // ---- epilogue: lane = pixel (x0 + ar), registers = 4 consecutive output channels
    const int x = x0 + ar;
    f32x4 pool[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) pool[n] = (f32x4)(0.0f);

    // per column tile: channel inside the (shuffled) pixel and, for PixelShuffle, the sub-pixel -- divisions by run-time values,
    // done ONCE per tile column here (hipcc re-did them for every row tile: ~4.5 k of the 14.5 k epilogue cycles of the 64 -> 256 convs)
    int nch[NW], ps_i[NW], ps_j[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int col = (ntile0 + n) * 16 + ag * 4;
        nch[n] = col;
        ps_i[n] = ps_j[n] = 0;
        if (c.out_mode == SR_OUT_PIXEL_SHUFFLE || (c.out_mode == SR_OUT_FINAL_NCHW && c.ps_r > 1)) {
            const int sub = col / c.cps_p;
            nch[n] = col - sub * c.cps_p;
            ps_i[n] = sub / c.ps_r;
            ps_j[n] = sub - ps_i[n] * c.ps_r;
        }
    }
    const bool scaled = c.out_scale != 1.0f;
    const float lrelu_slope = c.act_slope != 0.0f ? c.act_slope : 0.01f;
    if (coal) {
        if constexpr (kMatrix16<TC> && MTW % 2 == 0) {
            __syncthreads();  // every wave has issued its last fragment read of the halo image: the private tiles overlay it
            char* priv = smem + wave * EpiGeo<NW>::PRIV;
            constexpr int S = EpiGeo<NW>::S;
            const bool ps = c.out_mode == SR_OUT_PIXEL_SHUFFLE;
            // PixelShuffle: this wave's channels [ntile0 * 16, +NW * 16) are channels [cb, +NW * 16) of sub-pixel (pi, pj)
            const int sub = ps ? (ntile0 * 16) / c.cps_p : 0, cb = ps ? ntile0 * 16 - sub * c.cps_p : ntile0 * 16;
            const int pi = ps ? sub / c.ps_r : 0, pj = ps ? sub - pi * c.ps_r : 0;
            const int r = ps ? c.ps_r : 1, ldo = ps ? c.cps_p : c.Cout_p;
            const int Wo = c.W * r;
            act_dispatch(c.act, [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
                auto run = [&](auto obf_tag) {
                    constexpr bool OBF = decltype(obf_tag)::value != 0;  // bf16 output: 8 channels per 16-byte piece; fp32: 4
                    constexpr int NV = OBF ? 8 : 4, PP = NW * 16 / NV;
                    constexpr int G = (16 * PP) % 64 == 0 ? 1 : 2, NI = G * 16 * PP / 64;
                    static_assert(MTW % G == 0 && G <= EpiGeo<NW>::GMAX, "row-tile groups");
#pragma unroll
                    for (int mg = 0; mg < MTW / G; ++mg) {
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int n = 0; n < NW; ++n) {
                                f32x4 v = acc[mg * G + g][n] + bias_r[n];
#pragma unroll
                                for (int q = 0; q < 4; ++q) v[q] = act_ct<ACT>(v[q], lrelu_slope);
                                if (c.pool_partial) pool[n] += v;
                                if (scaled) v *= c.out_scale;
                                *reinterpret_cast<f32x4*>(priv + (g * 16 + ar) * S + (n * 16 + ag * 4) * 4) = v;
                            }
#pragma unroll
                        for (int k = 0; k < NI; ++k) {
                            const int idx = k * 64 + lane, px = idx / PP, pc = idx - px * PP;
                            const int y = y0 + wm * MTW + mg * G + (px >> 4), xq = x0 + (px & 15);
                            const float* src = reinterpret_cast<const float*>(priv + px * S) + pc * NV;
                            f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = (f32x4)(0.0f);
                            if constexpr (OBF) hi = *reinterpret_cast<const f32x4*>(src + 4);
                            if (y < c.H && xq < c.W) {
                                const size_t off = ((size_t)((size_t)(b * c.H + y) * r + pi) * Wo + (size_t)xq * r + pj) * ldo + cb + pc * NV;
                                if (c.skip && !acc_from_skip) {
                                    if (c.skip_dtype == SR_BF16) {
                                        if constexpr (OBF) {
                                            const bf16x8 sk = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(c.skip) + off);
#pragma unroll
                                            for (int q = 0; q < 4; ++q) lo[q] += (float)sk[q], hi[q] += (float)sk[4 + q];
                                        } else
                                            lo += load4(reinterpret_cast<const bf16*>(c.skip) + off);
                                    } else {
                                        lo += load4(reinterpret_cast<const float*>(c.skip) + off);
                                        if constexpr (OBF) hi += load4(reinterpret_cast<const float*>(c.skip) + off + 4);
                                    }
                                }
                                if constexpr (OBF) {
                                    bf16x8 o;
#pragma unroll
                                    for (int q = 0; q < 4; ++q) o[q] = (bf16)lo[q], o[4 + q] = (bf16)hi[q];
                                    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(c.out) + off) = o;
                                } else
                                    store4(reinterpret_cast<float*>(c.out) + off, lo);
                            }
                        }
                    }
                };
                if (c.out_dtype == SR_BF16)
                    run(IntC<1>{});
                else
                    run(IntC<0>{});
            });
        }
    } else {
    act_dispatch(c.act, [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int y = y0 + wm * MTW + m;
        const bool inb = (y < c.H) && (x < c.W);
        if (!inb) continue;  // ONE exec-mask branch per row tile (not one per accumulator tile)
        const size_t pix_nhwc = ((size_t)(b * c.H + y) * c.W + x) * c.Cout_p;          // NHWC pixel base
        const size_t ps_row = (size_t)(b * c.H + y) * c.ps_r, ps_col = (size_t)x * c.ps_r;  // PixelShuffle: top-left sub-pixel
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            f32x4 v = acc[m][n] + bias_r[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_ct<ACT>(v[r], lrelu_slope);
            if (c.pool_partial) pool[n] += v;
            if (scaled) v *= c.out_scale;
            if (c.out_mode == SR_OUT_FINAL_NCHW) {
                const int yy = c.ps_r > 1 ? y * c.ps_r + ps_i[n] : y;  // "pixelshuffledirect": shuffle straight into the final image
                const int xx = c.ps_r > 1 ? x * c.ps_r + ps_j[n] : x;
                if (yy < c.fin_h && xx < c.fin_w) {
                    float* o = reinterpret_cast<float*>(c.out);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ch = nch[n] + r;
                        if (ch < c.fin_c) o[((size_t)(b * c.fin_c + ch) * c.fin_h + yy) * c.fin_w + xx] = v[r] * c.fin_scale[ch] + c.fin_bias[ch];
                    }
                }
                continue;
            }
            const size_t off = c.out_mode == SR_OUT_PIXEL_SHUFFLE ? ((ps_row + ps_i[n]) * ((size_t)c.W * c.ps_r) + ps_col + ps_j[n]) * c.cps_p + nch[n]
                                                                 : pix_nhwc + nch[n];
            if (c.skip && !acc_from_skip) {
                if (c.skip_dtype == SR_BF16)
                    v += load4(reinterpret_cast<const bf16*>(c.skip) + off);
                else
                    v += load4(reinterpret_cast<const float*>(c.skip) + off);
            }
            if (c.out_dtype == SR_BF16)
                store4(reinterpret_cast<bf16*>(c.out) + off, v);
            else
                store4(reinterpret_cast<float*>(c.out) + off, v);
        }
    }
    });

    }
    STAMP(4);
    if (c.pool_partial) {
        const int n_tiles = tiles_x * tiles_y * WM;
        const int slot = (ty * tiles_x + tx) * WM + wm;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
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
            if (ar == 0) store4(c.pool_partial + ((size_t)b * n_tiles + slot) * c.Cout_p + (ntile0 + n) * 16 + ag * 4, p);
        }
    }
}

template <typename TC, typename TIn, int TH, int WM, int WN, int NW, int KCS>
int launch_conv_k(const SrConv3x3& c, hipStream_t st) {
    constexpr int RS = ConvGeo<TH, kMatrix16<TC> ? 2 : 4>::RS;
    // halo image, or the four wave-private transpose tiles of the coalesced epilogue if they need more
    const int lds_img = c.Cin_p * RS * (int)sizeof(TC);
    const int lds = lds_img > 4 * EpiGeo<NW>::PRIV ? lds_img : 4 * EpiGeo<NW>::PRIV;
    SR_REQUIRE(lds <= 160 * 1024, "sr_conv3x3: Cin_p=%d needs %d B of LDS", c.Cin_p, lds);
    static SrDeviceOnce attr_once;  // one flag per template instantiation, one bit per device
    {
        const hipError_t e = sr_once_per_device(attr_once, [&] { return sr_allow_lds(sr_conv3x3_kernel<TC, TIn, TH, WM, WN, NW, KCS>, 160 * 1024); });
        SR_REQUIRE(e == hipSuccess, "sr_conv3x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const int tiles = ((c.W + 15) / 16) * ((c.H + TH - 1) / TH) * c.B;
    dim3 grid(tiles, c.Cout_p / (WN * NW * 16));
    const int xcd_env = 1;  // XCD-aware tile order (0 = tiles in block-id order; the A/B switch left in round 5)
    const int xcd_order = (xcd_env && tiles >= 64 && (grid.y == 1 || (tiles & 7) == 0)) ? 1 : 0;  // (the residue class of a block id is that of blockIdx.x)
    hipLaunchKernelGGL((sr_conv3x3_kernel<TC, TIn, TH, WM, WN, NW, KCS>), grid, dim3(256), lds, st, c, xcd_order);
    SR_CHECK_LAUNCH("sr_conv3x3");
    return SR_OK;
}

// unrolled variants for the channel counts of the reference models (32 = RGB ingest, 64, 192 = 180 padded, 256);
// bf16 only -- the exact-fp32 parity path keeps the generic loop
template <typename TC, typename TIn, int TH, int WM, int WN, int NW>
int launch_conv(const SrConv3x3& c, hipStream_t st) {
    if constexpr (kMatrix16<TC>) {  // (round 5: the split-operand convs of RCAN / HAT / the small EDSR launches ran the generic run-time loop)
        if constexpr (NW < 4) {  // the 256-wide tile is only dispatched for Cin_p >= 128
            if (c.Cin_p == 32) return launch_conv_k<TC, TIn, TH, WM, WN, NW, 1>(c, st);
            if (c.Cin_p == 64) return launch_conv_k<TC, TIn, TH, WM, WN, NW, 2>(c, st);
        }
        switch (c.Cin_p) {
            case 192: return launch_conv_k<TC, TIn, TH, WM, WN, NW, 6>(c, st);
            case 256: return launch_conv_k<TC, TIn, TH, WM, WN, NW, 8>(c, st);
            default: break;
        }
    }
    return launch_conv_k<TC, TIn, TH, WM, WN, NW, 0>(c, st);
}

// pick the widest N tile that divides Cout_p
template <typename TC, typename TIn, int TH>
int dispatch_conv(const SrConv3x3& c, hipStream_t st) {
    const int n = c.Cout_p;
    if constexpr (sizeof(TC) == 2) {
        if (n % 256 == 0 && c.Cin_p >= 128) return launch_conv<TC, TIn, TH, 1, 4, 4>(c, st);  // 32 accumulator tiles per wave: one pass over the halo tile
    }
    if (n % 192 == 0) return launch_conv<TC, TIn, TH, 1, 4, 3>(c, st);
    if (n % 128 == 0) return launch_conv<TC, TIn, TH, 1, 4, 2>(c, st);
    if (n % 64 == 0) return launch_conv<TC, TIn, TH, 2, 2, 2>(c, st);
    if (n % 32 == 0) return launch_conv<TC, TIn, TH, 2, 2, 1>(c, st);
    return launch_conv<TC, TIn, TH, 4, 1, 1>(c, st);
}

}  // namespace sr_conv_impl

// one definition per variant TU
#define SR_CONV_VARIANT(NAME, TC, TIN, TH) \
    int NAME(const SrConv3x3& c, hipStream_t st) { return sr_conv_impl::dispatch_conv<TC, TIN, TH>(c, st); }
