// 3x3 convolution, wide-channel variant (Cin_p 192 -> Cout_p 192 k; Cin_p 256 -> Cout_p 256 k; NHWC or PixelShuffle output).
// (64 input channels were tried and lost: with K = 576 the 64-tile epilogue dominates and 8 x 16 tiles overlap better.): same implicit GEMM on an LDS halo
// tile as sr_conv.hip, re-tiled for what the SQ counters showed on the 256 -> 256 conv (profiles/r01_conv256_sq_counters.txt):
// with 8 x 16 pixel tiles every workgroup streams the whole weight set (1.2 MB) for 128 pixels, all 256 CUs pull the same
// fragments at the same time and L2 -> L1 delivery (32 B/clk/CU) plus one-wave-per-SIMD load latency leave the MFMA
// pipe 40 % busy.  Here
//   * one workgroup = 16 (or 12) rows x 16 pixels x ALL output channels: 4 waves split N, each wave holds 16 row tiles x NW column
//     tiles = 48 / 64 accumulator tiles in AGPRs -> 64 MFMAs per 1 KiB-per-column-tile weight fetch (half the L2 traffic
//     per MFMA, 16 LDS fragment reads per 48-64 MFMAs);
//   * the 18 x 18 halo tile does not fit LDS for 256 input channels (166 KB), so K is walked in PH phases of Cin_p / PH
//     channels (192: one phase, 123 KiB; 256: two phases of 128 channels, 82 KiB); the accumulators live across phases;
//   * activation fragments are double buffered by half chunks (8 row tiles), weights run through a register ring;
//   * taps are a run-time loop (9 iterations), the chunks of a tap are unrolled: code stays a few KiB.
#include "sr_common.h"
#include "sr_host.h"

#include <cstdlib>

#ifndef SR_BIG_RING
#define SR_BIG_RING 3
#endif

namespace {

#ifndef SR_CONVBIG_XCD
#define SR_CONVBIG_XCD 1  // EDSR x4 b16 6.31 -> 6.24 ms, SwinIR +-0 (0: tiles in block-id order)
#endif
#ifdef SR_STAMPS
__device__ unsigned long long sr_dbg_convbig[8];
#define BSTAMP(i) SR_STAMP(sr_dbg_convbig, i)
#else
#define BSTAMP(i) do { } while (0)
#endif

constexpr int BT = 16;                  // tile width (pixels) = one MFMA row tile
constexpr int BH = BT + 2;              // halo width

// TH = tile height (rows = MFMA row tiles per wave): 16, or 12 where that turns a half-empty second residency round
// into one full round (72 x 72 images: 8 x 6 x 5 = 240 workgroups of 12 rows instead of 200 of 16)
// PIPE: the LDS image holds ALL input channels and is staged in PH channel phases that are software-pipelined with the MFMAs: the global
// loads of phase p + 1 are in flight (in registers) while phase p's nine taps run, and are converted / written to LDS afterwards -- the
// one-workgroup-per-CU kernel no longer fetches its whole halo before the first MFMA.
// TC = bf16, or bf3 (round 5; compute type SR_BF16X3, precision "fp32x3": every operand a hi + lo bf16 pair, three MFMAs per product, fp32 input only): 32-byte image
// cells, so K is walked in twice as many phases (192 channels: 2 x 96, 256: 4 x 64) and the tile stays within the same LDS footprint.
#ifndef SR_BIG_OCC
#define SR_BIG_OCC 1  // experiment knob (with SR_BIG_TH / SR_BIG_PH): workgroups per CU the register allocation is bounded for
#endif
template <typename TC, typename TIn, int TH, int NW, int KC, int PH, bool PIPE = false, int OCC = SR_BIG_OCC>
__global__ __launch_bounds__(256, OCC) void sr_conv3x3_big_kernel(SrConv3x3 c) {
    static_assert(!PIPE || sizeof(Frag<TC>) == 16, "the pipelined staging exists for bf16 operands only");
    constexpr int HH = TH + 2;                       // halo height
    constexpr int BROWS = ((HH * BH + 7) / 8) * 8;   // halo pixels, padded to a multiple of 8
    constexpr int BRS = BROWS | 1;                   // row stride of the image in cells (odd: the staging writes of one pixel's 8 K-groups hit 8 bank groups)
    constexpr int HALF = TH / 2;                     // row tiles per activation-fragment buffer
    static_assert(TH % 2 == 0, "two half chunks");
    constexpr int KCP = KC / PH;                   // 32-channel chunks per phase
    // ring slots are compile-time: the ring size divides KCP (run-time tap loop), or -- PIPE -- divides 9 * KCP with the tap loop unrolled.
    // PIPE needs the DEEP ring: vector loads return in order, so a weight fragment issued after the next phase's halo loads (HBM latency)
    // cannot be consumed before they land; five chunks of look-ahead (2.9 k MFMA cycles) issued BEFORE them cover that latency.
    constexpr int RING = PIPE ? 6 : ((KCP % SR_BIG_RING == 0) ? SR_BIG_RING : ((KCP % 3 == 0) ? 3 : (KCP % 4 == 0 ? 4 : 2)));
    static_assert(KC % PH == 0 && (PIPE ? (9 * KCP) % RING == 0 : KCP % RING == 0), "phase / ring geometry");
    constexpr int KGP = KCP * 4;                   // 8-channel groups per phase
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<TC>* As = reinterpret_cast<Frag<TC>*>(smem);  // [KGP][BROWS]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_x = (c.W + BT - 1) / BT, tiles_y = (c.H + TH - 1) / TH;
    int t = blockIdx.x;
#if SR_CONVBIG_XCD
    if (gridDim.y == 1 || (gridDim.x & 7) == 0) {  // neighbouring tiles (shared halo rows) on one XCD: each residue class mod 8 takes a contiguous range of tiles
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = t & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    }
#endif
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * BT, y0 = ty * TH;
    const int ar = lane & 15, ag = lane >> 4;
    const int ntile0 = blockIdx.y * (4 * NW) + wave * NW;  // blockIdx.y: slices of 64 * NW output channels (PixelShuffle convs: Cout = r*r*C)
    constexpr int KCT = 9 * KC;
    const Frag<TC>* Bp = reinterpret_cast<const Frag<TC>*>(c.Wp) + (size_t)ntile0 * KCT * 64 + lane;
    const TIn* xin = reinterpret_cast<const TIn*>(c.x);
    BSTAMP(0);

    // plain residual convs: the skip tile is the initial accumulator (fetched now, in the accumulator layout: its latency hides under
    // the staging; in the epilogue a residual read sits on the critical path of every row-tile group: 37 -> 47 us on the RSTB conv)
    const bool acc_from_skip = c.skip && c.act == SR_ACT_NONE && c.out_scale == 1.0f && c.out_mode == SR_OUT_NHWC;
    f32x4 acc[TH][NW];
#pragma unroll
    for (int m = 0; m < TH; ++m) {
        const int yy = y0 + m, xx = x0 + ar;
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

    // PIPE staging: one phase = KGP K-groups of every halo pixel; 8 pixels x 8 K-groups per wave instruction, all BROWS / 32 pixel groups of a
    // wave issued at once and kept in registers (raw input type) until commit() converts and writes them.
    constexpr int NPG = (BROWS + 31) / 32;  // pixel groups per wave
    struct Raw {
        f32x4 lo, hi;  // 8 fp32 channels (fp32 input) or 8 bf16 in `lo` (bf16 input)
    };
    Raw raw[PIPE ? NPG : 1];
    auto halo_pixel = [&](int p, bool& valid) -> size_t {
        const int py = p / BH, px = p - py * BH;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        valid = p < HH * BH && gy >= 0 && gy < c.H && gx >= 0 && gx < c.W;
        return ((size_t)(b * c.H + (valid ? gy : 0)) * c.W + (valid ? gx : 0)) * c.Cin_p;
    };
    auto issue = [&](int ph) {
        static_assert(!PIPE || KGP == 8, "PIPE stages 8 K-groups (64 channels) per phase");
        const int kq = lane & 7, r8 = lane >> 3;
#pragma unroll
        for (int j = 0; j < (PIPE ? NPG : 0); ++j) {
            bool valid;
            const TIn* src = xin + halo_pixel(wave * 8 + j * 32 + r8, valid) + (ph * KGP + kq) * 8;
            if constexpr (sizeof(TIn) == 4) {
                raw[j].lo = *reinterpret_cast<const f32x4*>(src);
                raw[j].hi = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + 4);
            } else {
                raw[j].lo = *reinterpret_cast<const f32x4*>(src);  // 8 bf16 = 16 B
            }
        }
    };
    auto commit = [&](int ph) {
        const int kq = lane & 7, r8 = lane >> 3;
#pragma unroll
        for (int j = 0; j < (PIPE ? NPG : 0); ++j) {
            const int p = wave * 8 + j * 32 + r8;
            bool valid;
            (void)halo_pixel(p, valid);
            Frag<bf16> f;  // (PIPE: bf16 operands only)
            if constexpr (sizeof(TIn) == 4) {
                f.v[0] = (bf16)raw[j].lo[0]; f.v[1] = (bf16)raw[j].lo[1]; f.v[2] = (bf16)raw[j].lo[2]; f.v[3] = (bf16)raw[j].lo[3];
                f.v[4] = (bf16)raw[j].hi[0]; f.v[5] = (bf16)raw[j].hi[1]; f.v[6] = (bf16)raw[j].hi[2]; f.v[7] = (bf16)raw[j].hi[3];
            } else {
                f.v = __builtin_bit_cast(bf16x8, raw[j].lo);
            }
            if constexpr (sizeof(Frag<TC>) == 16) {
                if (p < BROWS) As[(ph * KGP + kq) * BRS + p] = frag_keep_if(valid, f);
            }
        }
    };
    if constexpr (PIPE) {
        issue(0);
        commit(0);
        __syncthreads();
    }

    for (int ph = 0; ph < PH; ++ph) {
        if constexpr (!PIPE) {
        if (ph == 0) BSTAMP(1);
        if (ph > 0) __syncthreads();  // every wave is done reading the previous phase's tile
        // ---- stage channels [ph * KGP * 8, +KGP * 8) of the halo tile: 8 pixels x 8 K-groups per wave instruction
        {
            // K-group on the fast lane axis: the 8 lanes of a pixel read 128 / 256 contiguous bytes (pixel-fastest lanes cost the vector
            // memory path four cache-line accesses per quad of lanes, see sr_conv_impl.h)
            const int kq = lane & 7, r8 = lane >> 3;
            constexpr int KI = KGP / 8;
            constexpr int NPASS = 4;  // row passes whose loads are all in flight before the first LDS write
            for (int pb = wave * 8; pb < BROWS; pb += 32 * NPASS) {
                Frag<TC> f[NPASS][KI];
                bool valid[NPASS];
#pragma unroll
                for (int u = 0; u < NPASS; ++u) {
                    const int p = pb + u * 32 + r8;
                    const int py = p / BH, px = p - py * BH;
                    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                    valid[u] = p < HH * BH && gy >= 0 && gy < c.H && gx >= 0 && gx < c.W;
                    const TIn* src = xin + ((size_t)(b * c.H + (valid[u] ? gy : 0)) * c.W + (valid[u] ? gx : 0)) * c.Cin_p + ph * KGP * 8;
#pragma unroll
                    for (int i = 0; i < KI; ++i) f[u][i] = load_group<TC, TIn>(src + (kq + 8 * i) * 8);
                }
#pragma unroll
                for (int u = 0; u < NPASS; ++u) {
                    const int p = pb + u * 32 + r8;
                    if (p < BROWS) {
#pragma unroll
                        for (int i = 0; i < KI; ++i) As[(kq + 8 * i) * BRS + p] = frag_keep_if(valid[u], f[u][i]);
                    }
                }
            }
        }
        __syncthreads();
        if (ph == 0) BSTAMP(2);
        }

        // ---- 9 taps x KCP chunks of this phase; weight chunk index in the packed order = tap * KC + ph * KCP + kc
        const Frag<TC>* abase0 = As + ar + (ag + (PIPE ? ph * KGP : 0)) * BRS;
        Frag<TC> br[RING][NW];
        auto wload = [&](int slot, int tap, int kc) {  // slot is compile-time at every call site
            int chunk = tap * KC + ph * KCP + kc;
            if (kc >= KCP) chunk += KC - KCP;  // look-ahead ran into the next tap
            if (chunk > KCT - 1) chunk = KCT - 1;  // harmless re-load at the very end
#pragma unroll
            for (int n = 0; n < NW; ++n) br[slot][n] = Bp[((size_t)n * KCT + chunk) * 64];
        };
        auto wload_lin = [&](int slot, int t) {  // PIPE: chunk t = tap * KCP + kc of this phase (t >= 9 * KCP: harmless re-load of the last one)
            const int tt = t < 9 * KCP ? t : 9 * KCP - 1;
            const int chunk = (tt / KCP) * KC + ph * KCP + tt % KCP;
#pragma unroll
            for (int n = 0; n < NW; ++n) br[slot][n] = Bp[((size_t)n * KCT + chunk) * 64];
        };
        if constexpr (PIPE) {
#pragma unroll
            for (int s = 0; s < RING - 1; ++s) wload_lin(s, s);
            if (ph + 1 < PH) issue(ph + 1);  // the next phase's halo channels: in flight (in registers) during this phase's MFMAs
        } else {
#pragma unroll
            for (int s = 0; s < RING - 1; ++s) wload(s, 0, s);
        }
        Frag<TC> af[2][HALF];
#pragma unroll
        for (int m = 0; m < HALF; ++m) af[0][m] = abase0[m * BH];
        if constexpr (PIPE) {
#pragma unroll
            for (int t = 0; t < 9 * KCP; ++t) {
                constexpr int dummy = 0;
                (void)dummy;
                const int tap = t / KCP, kc = t % KCP;
                const Frag<TC>* abase = abase0 + (tap / 3) * BH + (tap % 3);
                const int tn = t + 1 < 9 * KCP ? t + 1 : t;
                const Frag<TC>* nb = abase0 + ((tn / KCP) / 3) * BH + ((tn / KCP) % 3) + (tn % KCP) * 4 * BRS;
                wload_lin((t + RING - 1) % RING, t + RING - 1);
#pragma unroll
                for (int m = 0; m < HALF; ++m) af[1][m] = abase[(HALF + m) * BH + kc * 4 * BRS];
#pragma unroll
                for (int m = 0; m < HALF; ++m)
#pragma unroll
                    for (int n = 0; n < NW; ++n) mma(br[t % RING][n], af[0][m], acc[m][n]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < HALF; ++m) af[0][m] = nb[m * BH];
#pragma unroll
                for (int m = 0; m < HALF; ++m)
#pragma unroll
                    for (int n = 0; n < NW; ++n) mma(br[t % RING][n], af[1][m], acc[HALF + m][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const Frag<TC>* abase = abase0 + ky * BH + kx;
            // first fragment address of the NEXT tap (for the look-ahead at the end of this one)
            const int tn = tap + 1 < 9 ? tap + 1 : tap;
            const Frag<TC>* abase_next = abase0 + (tn / 3) * BH + (tn % 3);
#pragma unroll
            for (int kc = 0; kc < KCP; ++kc) {
                wload((kc + RING - 1) % RING, tap, kc + RING - 1);
                // half chunk 0: the upper rows are in af[0]; fetch the lower rows of this chunk
#pragma unroll
                for (int m = 0; m < HALF; ++m) af[1][m] = abase[(HALF + m) * BH + kc * 4 * BRS];
#pragma unroll
                for (int m = 0; m < HALF; ++m)
#pragma unroll
                    for (int n = 0; n < NW; ++n) mma(br[kc % RING][n], af[0][m], acc[m][n]);
                __builtin_amdgcn_sched_barrier(0);
                // half chunk 1: fetch the upper rows of the next chunk (next tap's first chunk after the last one)
                {
                    const Frag<TC>* nb = (kc + 1 < KCP) ? abase + (kc + 1) * 4 * BRS : abase_next;
#pragma unroll
                    for (int m = 0; m < HALF; ++m) af[0][m] = nb[m * BH];
                }
#pragma unroll
                for (int m = 0; m < HALF; ++m)
#pragma unroll
                    for (int n = 0; n < NW; ++n) mma(br[kc % RING][n], af[1][m], acc[HALF + m][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (PIPE) {
            if (ph + 1 < PH) {
                commit(ph + 1);   // the next phase's channels have been in flight under the MFMAs above
                __syncthreads();  // ... and are now visible to every wave (no WAR hazard: each phase has its own K-group rows)
            }
        }
    }

    BSTAMP(3);
    // ---- epilogue (the coalesced form of sr_conv_impl.h, same op order: (acc + bias) -> activation -> scale -> + skip -> round):
    //      each wave transposes two row tiles at a time through a private fp32 LDS tile [pixel][its NW * 16 channels] and stores / reads
    //      the residual in 16-byte pieces with adjacent lanes on adjacent addresses of one pixel
    f32x4 bias_r[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) bias_r[n] = c.bias ? load4(c.bias + (ntile0 + n) * 16 + ag * 4) : (f32x4)(0.0f);
    const bool scaled = c.out_scale != 1.0f;
    const float lrelu_slope = c.act_slope != 0.0f ? c.act_slope : 0.01f;
    __syncthreads();  // every wave has issued its last fragment read of the halo image: the private tiles overlay it
    constexpr int S = NW * 64 + 16;
    char* priv = smem + wave * (2 * 16 * S);
    const bool ps = c.out_mode == SR_OUT_PIXEL_SHUFFLE;
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
            static_assert(TH % G == 0, "row-tile groups");
#pragma unroll
            for (int mg = 0; mg < TH / G; ++mg) {
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int n = 0; n < NW; ++n) {
                        f32x4 v = acc[mg * G + g][n] + bias_r[n];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = act_ct<ACT>(v[q], lrelu_slope);
                        if (scaled) v *= c.out_scale;
                        *reinterpret_cast<f32x4*>(priv + (g * 16 + ar) * S + (n * 16 + ag * 4) * 4) = v;
                    }
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    const int idx = k * 64 + lane, px = idx / PP, pc = idx - px * PP;
                    const int y = y0 + mg * G + (px >> 4), xq = x0 + (px & 15);
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
    BSTAMP(4);
}

template <typename TC, typename TIn, int TH, int NW, int KC, int PH, bool PIPE = false, int OCC = SR_BIG_OCC>
int launch_big(const SrConv3x3& c, hipStream_t st) {
    constexpr int BROWS = (((TH + 2) * BH + 7) / 8) * 8;
    constexpr int lds_img = (PIPE ? KC : KC / PH) * 4 * (BROWS | 1) * (int)sizeof(Frag<TC>), lds_priv = 4 * 2 * 16 * (NW * 64 + 16);  // halo image; the epilogue's private tiles overlay it
    constexpr int lds = lds_img > lds_priv ? lds_img : lds_priv;
    static_assert(lds <= 160 * 1024, "halo tile must fit LDS");
    static_assert(((KC / PH) * 4) % 8 == 0, "the staging moves 8 K-groups per wave instruction");
    static SrDeviceOnce attr_once;  // one flag per template instantiation, one bit per device
    {
        const hipError_t e = sr_once_per_device(attr_once, [&] { return sr_allow_lds(sr_conv3x3_big_kernel<TC, TIn, TH, NW, KC, PH, PIPE, OCC>, lds); });
        SR_REQUIRE(e == hipSuccess, "sr_conv3x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const int tiles = ((c.W + BT - 1) / BT) * ((c.H + TH - 1) / TH) * c.B;
    hipLaunchKernelGGL((sr_conv3x3_big_kernel<TC, TIn, TH, NW, KC, PH, PIPE, OCC>), dim3(tiles, c.Cout_p / (64 * NW)), dim3(256), lds, st, c);
    SR_CHECK_LAUNCH("sr_conv3x3");
    return SR_OK;
}

int big_nw(const SrConv3x3& c) { return c.Cout_p % 256 == 0 ? 4 : 3; }
int big_tiles(const SrConv3x3& c, int th) { return ((c.W + BT - 1) / BT) * ((c.H + th - 1) / th) * c.B * (c.Cout_p / (64 * big_nw(c))); }

// tile height with the smaller (residency rounds on 256 CUs) x (rows per workgroup); ties go to the taller tile
int big_tile_rows(const SrConv3x3& c) {
    // 256 input channels, bf16 operands (EDSR body and upsampler, edsr.py:34-37): 8-row tiles at TWO workgroups per CU (47 KiB of halo image, 128 accumulator registers) --
    // the staging, MFMA and epilogue phases of the two overlap: EDSR x4 b16 6.16 -> 5.58 ms, b32 13.1 -> 11.1, b8 3.45 -> 3.38 (profiles/r05_conv_big_tiles.txt; 6-row tiles,
    // four phases, three or four workgroups per CU: slower).  Same K order per pixel as every other tile height: same bits.
    if (c.Cin_p == 256 && c.compute_dtype == SR_BF16) return 8;
#ifdef SR_BIG_TH
    if (c.Cin_p == 192) return SR_BIG_TH;  // experiment: short tiles (several workgroups per CU) for the 192-channel conv
#endif
#ifdef SR_BIG_TH256
    if (c.Cin_p == 256) return SR_BIG_TH256;  // ... and for the 256-channel conv (EDSR body)
#endif
    const int cost16 = ((big_tiles(c, 16) + 255) / 256) * 16, cost12 = ((big_tiles(c, 12) + 255) / 256) * 12;
    return cost12 < cost16 ? 12 : 16;
}

template <typename TIn>
int dispatch_big8(const SrConv3x3& c, hipStream_t st) {  // (256 channels only: see big_tile_rows)
#ifdef SR_BIG_PH256
    if (c.Cin_p == 256 && big_nw(c) == 4) return launch_big<bf16, TIn, 8, 4, 8, SR_BIG_PH256, false, 2>(c, st);
#endif
    if (c.Cin_p == 256 && big_nw(c) == 4) return launch_big<bf16, TIn, 8, 4, 8, 2, false, 2>(c, st);
    return SR_EUNSUPPORTED;
}

template <typename TIn, int TH>
int dispatch_big(const SrConv3x3& c, hipStream_t st) {
    // 192 input channels: one phase.  (Three software-pipelined phases of 64 channels -- template parameter PIPE, K walk phase-major -- were
    // measured and lost: 36.7 -> 40.6 us on the RSTB conv even with five chunks of weight look-ahead; vector loads return in order, so the
    // next phase's halo fetch sits in front of every weight fragment issued after it, and all 240 workgroups fetch at the same time anyway.)
#ifdef SR_BIG_PH
    if (c.Cin_p == 192 && big_nw(c) == 3) return launch_big<bf16, TIn, TH, 3, 6, SR_BIG_PH>(c, st);
#endif
    if (c.Cin_p == 192 && big_nw(c) == 3) return launch_big<bf16, TIn, TH, 3, 6, 1>(c, st);
#ifdef SR_BIG_PH256
    if (c.Cin_p == 256 && big_nw(c) == 4) return launch_big<bf16, TIn, TH, 4, 8, SR_BIG_PH256>(c, st);
#endif
    if (c.Cin_p == 256 && big_nw(c) == 4) return launch_big<bf16, TIn, TH, 4, 8, 2>(c, st);
    return SR_EUNSUPPORTED;
}

// split operands (SR_BF16X3, fp32 input): phases of 64 channels (8 K-groups of 32-byte cells: the 16-row halo tile is 84 KiB)
template <int TH>
int dispatch_big_x3(const SrConv3x3& c, hipStream_t st) {
    if (c.Cin_p == 192 && big_nw(c) == 3) return launch_big<bf3, float, TH, 3, 6, 3>(c, st);
    if (c.Cin_p == 256 && big_nw(c) == 4) return launch_big<bf3, float, TH, 4, 8, 4>(c, st);
    return SR_EUNSUPPORTED;
}

}  // namespace

#ifdef SR_STAMPS
extern "C" int sr_debug_convbig_stamps(unsigned long long* host8) {
    return hipMemcpyFromSymbol(host8, HIP_SYMBOL(sr_dbg_convbig), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// true if sr_conv3x3_big covers this conv (bf16 compute, NHWC or PixelShuffle output, no pooling side output)
bool sr_conv3x3_big_supported(const SrConv3x3& c) {
    // Below ~224 workgroups the wide tile leaves too many CUs idle and the 8 x 16 / 4 x 16 kernel wins (192 -> 192: 4 x 64 x 64 33 vs 16.5 us,
    // 8 x 64 x 64 34.7 vs 28.8 us; 8 x 72 x 72 = 240 workgroups 36.8 vs 46.0 us: wide tile).  SR_CONV_BIG_MIN overrides (tools/kbench.py conv).
    static const int min_tiles = getenv("SR_CONV_BIG_MIN") ? atoi(getenv("SR_CONV_BIG_MIN")) : 224;
    const bool x3 = c.compute_dtype == SR_BF16X3 && c.x_dtype == SR_F32;  // round 5: the split-operand instantiation (precision "fp32x3")
    if ((c.compute_dtype != SR_BF16 && !x3) || (c.out_mode != SR_OUT_NHWC && c.out_mode != SR_OUT_PIXEL_SHUFFLE) || c.pool_partial) return false;
    if (!((c.Cin_p == 192 && c.Cout_p % 192 == 0 && c.Cout_p % 256 != 0) || (c.Cin_p == 256 && c.Cout_p % 256 == 0))) return false;
    if (c.out_mode == SR_OUT_PIXEL_SHUFFLE && c.cps_p % (big_nw(c) * 16) != 0) return false;  // the coalesced epilogue needs a wave's channels inside one sub-pixel
    return big_tiles(c, big_tile_rows(c)) >= min_tiles;  // small launches keep the 8 x 16 tiles (more workgroups)
}

int sr_conv3x3_big(const SrConv3x3& c, hipStream_t st) {
#ifdef SR_BIG_TH256
    if (c.compute_dtype == SR_BF16 && c.Cin_p == 256) return c.x_dtype == SR_F32 ? dispatch_big<float, SR_BIG_TH256>(c, st) : dispatch_big<bf16, SR_BIG_TH256>(c, st);
#endif
    if (c.compute_dtype == SR_BF16 && big_tile_rows(c) == 8) return c.x_dtype == SR_F32 ? dispatch_big8<float>(c, st) : dispatch_big8<bf16>(c, st);
#ifdef SR_BIG_TH
    if (c.compute_dtype == SR_BF16 && big_tile_rows(c) == SR_BIG_TH) return c.x_dtype == SR_F32 ? dispatch_big<float, SR_BIG_TH>(c, st) : dispatch_big<bf16, SR_BIG_TH>(c, st);
#endif
    if (c.compute_dtype == SR_BF16X3) return big_tile_rows(c) == 12 ? dispatch_big_x3<12>(c, st) : dispatch_big_x3<16>(c, st);
    if (big_tile_rows(c) == 12) return c.x_dtype == SR_F32 ? dispatch_big<float, 12>(c, st) : dispatch_big<bf16, 12>(c, st);
    return c.x_dtype == SR_F32 ? dispatch_big<float, 16>(c, st) : dispatch_big<bf16, 16>(c, st);
}
