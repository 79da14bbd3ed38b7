// Fused body of a 64-channel residual channel-attention block (rcan.py:11-24): y = conv2(ReLU(conv1(x))) in ONE launch,
// plus the per-tile channel sums the channel-attention gate needs (common.py:156-170).
//
// Why: at RCAN's shapes (64 channels, K = 576) a 3x3 conv is ~1 us of MFMAs wrapped in ~10 us of launch, halo-load and store
// latency; 400 of them per forward made the model launch-latency bound (profiles/r01_model_bench.jsonl).  Here the ReLU'd
// intermediate never leaves the CU:
//   * one workgroup (4 waves) owns a 14 x 14 output tile; it stages the 18 x 18 input halo (all 64 channels, bf16,
//     K-group-major image as in sr_conv.hip), computes the 16 x 16 intermediate tile (conv1 + bias + ReLU; positions
//     outside the image are ZERO: they are conv2's padding) into a second LDS image, then the 14 x 14 outputs from it;
//   * an MFMA row tile is 16 horizontally adjacent pixels in both convs (conv2 evaluates 16 columns per row and keeps the
//     inner 14); conv1: four rows x all channels per wave, conv2: waves split 2 x 2 over (rows, output channels); weights stream
//     L2 -> registers through rings of 5 / 6 slots;
//   * K is walked tap-major exactly like sr_conv3x3, so y has the same bits as the two-launch path.
// 76 KiB LDS -> two workgroups per CU.
// Gated form: the previous block's channel-attention tail (gate * y + skip) is applied while staging the halo, so a chain of RCABs is
// one launch per block; only the last block of a residual group still needs the standalone sr_channel_attention.
#include "sr_common.h"
#include "sr_host.h"
#include "sr_ca.h"

namespace {

constexpr int RC = 64;             // channels (padded)
constexpr int RKG = RC / 8;        // K-groups
constexpr int RKC = RC / 32;       // 32-channel chunks per tap
constexpr int RKCT = 9 * RKC;      // chunks per conv
constexpr int TO = 14;             // output tile edge
constexpr int TI = 16;             // intermediate tile edge (= one MFMA row tile)
constexpr int TIN = 18;            // input halo edge
constexpr int IN_ROWS = 328;       // 18 * 18 = 324 halo pixels, padded to a multiple of 8
constexpr int IN_RS = IN_ROWS + 1;  // row stride of the input image in cells: odd, so that the staging writes of one pixel's 8 K-groups (8 adjacent lanes) hit 8 bank groups
constexpr int MID_ROWS = 264;      // 1 margin cell + 256 intermediate pixels + 1 margin cell, padded to a multiple of 8
#ifndef SR_RCAB_RING
#define SR_RCAB_RING 6  // five chunks of weight look-ahead (16 MFMAs each): RCAN x4 b16 5.92 -> 5.20 ms, b32 8.26 -> 7.69 ms against 3 slots; 8 / 10 slots: 6.4 ms
#endif
constexpr int RRING_BF16 = SR_RCAB_RING;
#ifndef SR_RCAB_RING1
#define SR_RCAB_RING1 5
#endif
constexpr int R1RING_BF16 = SR_RCAB_RING1;  // conv1's ring: 4 fragments per slot

#ifndef SR_RCAB_XCD
#define SR_RCAB_XCD 1  // RCAN x4 b8 3.91 -> 3.65 ms, b16 4.85 -> 4.79, b32 7.38 -> 7.20 (0: tiles in block-id order)
#endif
#ifdef SR_STAMPS
__device__ unsigned long long sr_dbg_rcab[16];
#define RSTAMP(i)                                                                                 \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (blockIdx.x == 7 && threadIdx.x == 0) sr_dbg_rcab[i] = __builtin_amdgcn_s_memtime();   \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#else
#define RSTAMP(i) do { } while (0)
#endif

// 4 fp32 values into the 8-byte half `half` of a split-operand image cell (8 hi | 8 lo)
SR_DEV void st_half_x3(Frag<bf3>* cell, int half, const f32x4& v) {
    bf16x4 h, l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        h[r] = (bf16)v[r];
        l[r] = (bf16)(v[r] - (float)h[r]);
    }
    *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(cell) + half * 8) = h;
    *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(cell) + 16 + half * 8) = l;
}
SR_DEV void st_half_x3(Frag<bf16>*, int, const f32x4&) {}  // (never called: keeps the bf16 instantiation well-formed)

constexpr int GATE_SCRATCH = (8 * RC + RC + 8 + RC) * (int)sizeof(float);  // slice sums | mean | hidden | gate

// TC = bf16: bf16 operands, two workgroups per CU.  TC = bf3 (compute type SR_BF16X3, precision "fp32x3" = what inference() runs; round 5, ABI v11): every operand a hi + lo
// bf16 pair, every product hi*hi + hi*lo + lo*hi; the two images are 32-byte cells (152 KB: one workgroup per CU, 512 registers: shorter weight rings of wider fragments),
// x / y / the skip stream fp32.  Same K walk per output element as the split-operand sr_conv3x3.
template <typename TC, typename TIn, typename TOut, bool GATED>
__global__ __launch_bounds__(256, sizeof(Frag<TC>) == 16 ? 2 : 1) void sr_rcab_kernel(SrRcab c) {  // bf16: two workgroups per CU: <= 256 VGPRs (the gated form sits at the limit)
    constexpr bool X3 = sizeof(Frag<TC>) == 32;
    constexpr int RRING = X3 ? 4 : ::RRING_BF16, R1RING = X3 ? 3 : ::R1RING_BF16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<TC>* Ain = reinterpret_cast<Frag<TC>*>(smem);   // [RKG][IN_ROWS]
    Frag<TC>* Amid = Ain + RKG * IN_RS;                     // [RKG][MID_ROWS], pixel p of the 16 x 16 tile at row 1 + p

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ar = lane & 15, ag = lane >> 4;
    const int tiles_x = (c.W + TO - 1) / TO, tiles_y = (c.H + TO - 1) / TO;
    int t = blockIdx.x;
#if SR_RCAB_XCD
    {   // neighbouring tiles (18 x 18-pixel halos of 14 x 14 tiles) on one XCD: each residue class mod 8 of the block ids takes a contiguous range of tiles
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = t & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    }
#endif
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * TO, y0 = ty * TO;  // output tile origin; intermediate tile origin = (y0 - 1, x0 - 1), halo origin = (y0 - 2, x0 - 2)

    RSTAMP(0);
    const Frag<TC>* W2 = reinterpret_cast<const Frag<TC>*>(c.w2p) + (size_t)(wn * 2) * RKCT * 64 + lane;
    Frag<TC> br[RRING][2];     // conv2: this wave's two channel tiles
    Frag<TC> b1r[R1RING][4];   // conv1: all four channel tiles (see there)
    const Frag<TC>* W1a = reinterpret_cast<const Frag<TC>*>(c.w1p) + lane;
#pragma unroll
    for (int s = 0; s < R1RING - 1; ++s)
#pragma unroll
        for (int n = 0; n < 4; ++n) b1r[s][n] = W1a[((size_t)n * RKCT + s) * 64];

    float gk[8];  // gated input: this lane's 8 channel gates
    // ---- stage the input halo: 8 pixels x 8 K-groups per wave instruction with the K-GROUP on the fast lane axis (the 8 lanes of a pixel read
    //      its 128 / 256 contiguous bytes; pixel-fastest lanes cost the vector-memory path four cache-line accesses per quad of lanes), all loads
    //      of 3 passes in flight
    {
        const int kq = lane & 7, r8 = lane >> 3;
        constexpr int NPASS = 3;
        if constexpr (GATED) {
            // x_eff = x + gate * gate_y (fp32, one fma per element as in sr_channel_attention); the tile interior goes back to HBM as this
            // RCAB's skip.  Two rounds of 6 row passes; the loads of round 0 are in flight while the gate is computed.
            constexpr int GP = 6;
            const float* xin = reinterpret_cast<const float*>(c.x);
            const TOut* yin = reinterpret_cast<const TOut*>(c.gate_y);  // the previous block's y: same dtype as this block's
            float xv[GP][8], yv[GP][8];
            bool valid[GP], interior[GP];
            size_t off[GP];
            auto issue = [&](int pb) {
#pragma unroll
                for (int u = 0; u < GP; ++u) {
                    const int p = pb + u * 32 + r8;
                    const int py = p / TIN, px = p - py * TIN;
                    const int gy = y0 - 2 + py, gx = x0 - 2 + px;
                    valid[u] = p < TIN * TIN && gy >= 0 && gy < c.H && gx >= 0 && gx < c.W;
                    interior[u] = valid[u] && py >= 2 && py < 2 + TO && px >= 2 && px < 2 + TO;
                    off[u] = ((size_t)(b * c.H + (valid[u] ? gy : 0)) * c.W + (valid[u] ? gx : 0)) * RC + kq * 8;
                    load8f(xin + off[u], xv[u]);
                    if constexpr (sizeof(TOut) == 4) {
                        load8f(reinterpret_cast<const float*>(yin) + off[u], yv[u]);
                    } else {
                        const Frag<bf16> fy_ = *reinterpret_cast<const Frag<bf16>*>(yin + off[u]);
#pragma unroll
                        for (int i = 0; i < 8; ++i) yv[u][i] = (float)fy_.v[i];
                    }
                }
            };
            auto commit = [&](int pb) {
#pragma unroll
                for (int u = 0; u < GP; ++u) {
                    const int p = pb + u * 32 + r8;
                    float e[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) e[i] = __builtin_fmaf(yv[u][i], gk[i], xv[u][i]);
                    if (interior[u]) {
                        store4(c.x_out + off[u], f32x4{e[0], e[1], e[2], e[3]});
                        store4(c.x_out + off[u] + 4, f32x4{e[4], e[5], e[6], e[7]});
                    }
                    if (p < IN_ROWS) Ain[kq * IN_RS + p] = frag_keep_if(valid[u], frag_make<TC>(e));
                }
            };
            issue(wave * 8);
            RSTAMP(1);
            {
                float* part = reinterpret_cast<float*>(Amid + RKG * MID_ROWS);  // [8][RC]
                float* mean = part + 8 * RC;
                float* hid = mean + RC;
                float* gate = hid + 8;
                const int tid = threadIdx.x, ch = tid & 63, sl = tid >> 6;
                const int nt = tiles_x * tiles_y * 2, C = c.gate_C, Cr = c.gate_Cr;
                float w1v[2], b1v[2], w2v[8], b2v = 0.f;
        #pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = wave + 4 * u;
                    w1v[u] = (j < Cr && lane < C) ? c.gate_w1[j * C + lane] : 0.f;
                    b1v[u] = j < Cr ? c.gate_b1[j] : 0.f;
                }
        #pragma unroll
                for (int j = 0; j < 8; ++j) w2v[j] = (tid < C && j < Cr) ? c.gate_w2[tid * Cr + j] : 0.f;
                if (tid < C) b2v = c.gate_b2[tid];
                float s0 = 0.f, s1 = 0.f;
                if (ch < C) {
                    const float* pp = c.gate_pool + (size_t)b * nt * RC + ch;
                    constexpr int PF = 4;  // (8: 296 VGPRs = one workgroup per CU: b16 5.2 -> 5.7 ms, b32 7.7 -> 10.3 ms)
                    // 2 x PF partials in flight per thread, added in slot order (the same sums as one dependent load at a time, which was 9.4 k of
                    // this kernel's 41.9 k cycles: profiles/r03_rcab_kernel_stamps.txt)
#ifdef SR_RCAB_SEQPOOL
                    for (int tt = sl; tt < nt; tt += 8) s0 += pp[(size_t)tt * RC];
                    for (int tt = sl + 4; tt < nt; tt += 8) s1 += pp[(size_t)tt * RC];
#else
                    for (int t0 = 0; t0 < nt; t0 += 8 * PF) {
                        float v0[PF], v1[PF];
#pragma unroll
                        for (int k = 0; k < PF; ++k) {
                            v0[k] = pp[(size_t)min(t0 + sl + 8 * k, nt - 1) * RC];
                            v1[k] = pp[(size_t)min(t0 + sl + 4 + 8 * k, nt - 1) * RC];
                        }
#pragma unroll
                        for (int k = 0; k < PF; ++k) {
                            if (t0 + sl + 8 * k < nt) s0 += v0[k];
                            if (t0 + sl + 4 + 8 * k < nt) s1 += v1[k];
                        }
                    }
#endif
                }
                part[sl * RC + ch] = s0;
                part[(sl + 4) * RC + ch] = s1;
                __syncthreads();
                if (tid < RC) {
                    float sum = 0.f;
        #pragma unroll
                    for (int q = 0; q < 8; ++q) sum += part[q * RC + tid];
                    mean[tid] = sum * (1.0f / (float)(c.H * c.W));
                }
                __syncthreads();
        #pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = wave + 4 * u;
                    if (j < Cr) {
                        float sum = lane < C ? w1v[u] * mean[lane] : 0.f;
                        sum = ca_wave_sum(sum);  // (sr_ca.h: the same order of additions as sr_channel_attention, DPP steps instead of a ds_bpermute chain)
                        if (lane == 0) {
                            sum += b1v[u];
                            hid[j] = sum > 0.f ? sum : 0.f;
                        }
                    }
                }
                __syncthreads();
                if (tid < RC) {
                    float sum = 0.f;
                    if (tid < C) {
                        sum = b2v;
        #pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (j < Cr) sum += w2v[j] * hid[j];
                        sum = 1.0f / (1.0f + __expf(-sum));
                    }
                    gate[tid] = sum;
                }
                __syncthreads();
        #pragma unroll
                for (int i = 0; i < 8; ++i) gk[i] = gate[(lane & 7) * 8 + i];
            }
            RSTAMP(2);
            commit(wave * 8);
            issue(wave * 8 + 32 * GP);
            commit(wave * 8 + 32 * GP);
            static_assert(2 * 32 * GP >= IN_ROWS, "two rounds cover the halo");
        } else {
            const TIn* xin = reinterpret_cast<const TIn*>(c.x);
            for (int pb = wave * 8; pb < IN_ROWS; pb += 32 * NPASS) {
                Frag<TC> f[NPASS];
                bool valid[NPASS];
#pragma unroll
                for (int u = 0; u < NPASS; ++u) {
                    const int p = pb + u * 32 + r8;
                    const int py = p / TIN, px = p - py * TIN;
                    const int gy = y0 - 2 + py, gx = x0 - 2 + px;
                    valid[u] = p < TIN * TIN && gy >= 0 && gy < c.H && gx >= 0 && gx < c.W;
                    const TIn* src = xin + ((size_t)(b * c.H + (valid[u] ? gy : 0)) * c.W + (valid[u] ? gx : 0)) * RC + kq * 8;
                    f[u] = load_group<TC, TIn>(src);
                }
#pragma unroll
                for (int u = 0; u < NPASS; ++u) {
                    const int p = pb + u * 32 + r8;
                    if (p < IN_ROWS) Ain[kq * IN_RS + p] = frag_keep_if(valid[u], f[u]);
                }
            }
        }
        if (threadIdx.x < 2 * RKG) {  // margin cells of the intermediate image (read by discarded edge columns only; keep them finite)
            Frag<TC> z;
            frag_zero(z);
            Amid[(threadIdx.x >> 1) * MID_ROWS + ((threadIdx.x & 1) ? 1 + TI * TI : 0)] = z;
        }
    }
    RSTAMP(3);
    __syncthreads();
    RSTAMP(4);

    // ---- conv1 + bias + ReLU -> intermediate image.  Wave w: rows [4 w, 4 w + 4) x ALL 64 channels: 4 activation fragments (LDS) x 4 weight
    //      fragments per step instead of 8 x 2 -- the LDS fragment reads were as long as the MFMAs (8 KiB per wave and step at 128 B/clk per CU);
    //      per output element the K walk is unchanged (same bits)
    {
        f32x4 acc[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4)(0.0f);
        const Frag<TC>* abase0 = Ain + (wave * 4) * TIN + ar + ag * IN_RS;
#pragma unroll
        for (int tt = 0; tt < RKCT; ++tt) {
            const int tap = tt / RKC, kc = tt - tap * RKC;
            if (tt + R1RING - 1 < RKCT) {
#pragma unroll
                for (int n = 0; n < 4; ++n) b1r[(tt + R1RING - 1) % R1RING][n] = W1a[((size_t)n * RKCT + tt + R1RING - 1) * 64];
            } else if (tt + R1RING - 1 - RKCT < RRING - 1) {  // tail of conv1: start conv2's weight stream
#pragma unroll
                for (int n = 0; n < 2; ++n) br[tt + R1RING - 1 - RKCT][n] = W2[((size_t)n * RKCT + tt + R1RING - 1 - RKCT) * 64];
            }
            const Frag<TC>* arow = abase0 + (tap / 3) * TIN + (tap % 3) + kc * 4 * IN_RS;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const Frag<TC> a = arow[m * TIN];
#pragma unroll
                for (int n = 0; n < 4; ++n) mma(b1r[tt % R1RING][n], a, acc[m][n]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s2 = R1RING - 1; s2 < RRING - 1; ++s2)  // the rest of conv2's first slots (its ring is the deeper one)
#pragma unroll
            for (int n = 0; n < 2; ++n) br[s2][n] = W2[((size_t)n * RKCT + s2) * 64];
        RSTAMP(5);
        f32x4 bias[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) bias[n] = load4(c.b1 + n * 16 + ag * 4);
        const int gx = x0 - 1 + ar;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int iy = wave * 4 + m, gy = y0 - 1 + iy;
            const bool inside = gy >= 0 && gy < c.H && gx >= 0 && gx < c.W;  // outside the image the intermediate is conv2's zero padding
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                f32x4 v = acc[m][n] + bias[n];
                const int kg = n * 2 + (ag >> 1);
                if constexpr (X3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = inside ? fmaxf(v[r], 0.0f) : 0.0f;
                    st_half_x3(Amid + kg * MID_ROWS + 1 + iy * TI + ar, ag & 1, v);
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (bf16)(inside ? fmaxf(v[r], 0.0f) : 0.0f);
                    char* dst = reinterpret_cast<char*>(Amid + kg * MID_ROWS + 1 + iy * TI + ar) + (ag & 1) * 8;
                    *reinterpret_cast<bf16x4*>(dst) = o;
                }
            }
        }
    }
    __syncthreads();
    RSTAMP(6);

    // ---- conv2 + bias on output rows oy = 1 + 7 wm .. 7 wm + 7 of the intermediate tile; 16 columns per row, the inner 14 are kept
    {
        f32x4 acc[7][2];
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            acc[m][0] = (f32x4)(0.0f);
            acc[m][1] = (f32x4)(0.0f);
        }
        // tap (ky, kx) of output (oy, ox) reads intermediate (oy + ky - 1, ox + kx - 1) = cell 1 + (oy + ky - 1) * 16 + ox + kx - 1
        const Frag<TC>* abase0 = Amid + (wm * 7) * TI + ar + ag * MID_ROWS;
#pragma unroll
        for (int tt = 0; tt < RKCT; ++tt) {
            const int tap = tt / RKC, kc = tt - tap * RKC;
            if (tt + RRING - 1 < RKCT) {
#pragma unroll
                for (int n = 0; n < 2; ++n) br[(tt + RRING - 1) % RRING][n] = W2[((size_t)n * RKCT + tt + RRING - 1) * 64];
            }
            const Frag<TC>* arow = abase0 + (tap / 3) * TI + (tap % 3) + kc * 4 * MID_ROWS;
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                const Frag<TC> a = arow[m * TI];
                mma(br[tt % RRING][0], a, acc[m][0]);
                mma(br[tt % RRING][1], a, acc[m][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        RSTAMP(7);
        const f32x4 bias0 = load4(c.b2 + (wn * 2) * 16 + ag * 4), bias1 = load4(c.b2 + (wn * 2 + 1) * 16 + ag * 4);
        const int gx = x0 - 1 + ar;
        const bool xin_tile = ar >= 1 && ar <= TO && gx < c.W;
        f32x4 pool[2] = {(f32x4)(0.0f), (f32x4)(0.0f)};
        TOut* y = reinterpret_cast<TOut*>(c.y);
        // Coalesced store: one row of 16 pixels x this wave's 32 channels at a time through a wave-private fp32 tile in the (dead) input image:
        // accumulator layout in (pixel on the lane axis), 16-byte pieces out with adjacent lanes on adjacent addresses of ONE pixel.
        constexpr int S = 2 * 64 + 16;                          // bytes per pixel row of the private tile
        constexpr int NV = 16 / (int)sizeof(TOut);              // channels per 16-byte piece
        constexpr int PP = 32 / NV, NI = 16 * PP / 64;          // pieces per pixel, store instructions per row
        char* priv = smem + wave * (16 * S);
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            const int gy = y0 + wm * 7 + m;
            const bool ok = xin_tile && gy < c.H;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const f32x4 v = acc[m][n] + (n == 0 ? bias0 : bias1);
                if (ok) pool[n] += v;
                *reinterpret_cast<f32x4*>(priv + ar * S + (n * 16 + ag * 4) * 4) = v;
            }
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int idx = k * 64 + lane, px = idx / PP, pc = idx - px * PP;
                const int gxp = x0 - 1 + px;
                const float* src = reinterpret_cast<const float*>(priv + px * S) + pc * NV;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
                f32x4 hi = (f32x4)(0.0f);
                if constexpr (sizeof(TOut) == 2) hi = *reinterpret_cast<const f32x4*>(src + 4);
                if (px >= 1 && px <= TO && gxp < c.W && gy < c.H) {
                    TOut* dst = y + ((size_t)(b * c.H + gy) * c.W + gxp) * RC + wn * 32 + pc * NV;
                    if constexpr (sizeof(TOut) == 2) {
                        bf16x8 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = (bf16)lo[q], o[4 + q] = (bf16)hi[q];
                        *reinterpret_cast<bf16x8*>(dst) = o;
                    } else
                        store4(dst, lo);
                }
            }
        }
        RSTAMP(8);
        if (c.pool_partial) {  // per (tile, row half) channel sums over the valid pixels: deterministic, no atomics
            const int n_slots = tiles_x * tiles_y * 2;
            const int slot = (ty * tiles_x + tx) * 2 + wm;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
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
                if (ar == 0) store4(c.pool_partial + ((size_t)b * n_slots + slot) * RC + (wn * 2 + n) * 16 + ag * 4, p);
            }
        }
    }
    RSTAMP(9);
}

template <typename TC, typename TIn, typename TOut, bool GATED>
int launch_rcab(const SrRcab& c, hipStream_t st) {
    constexpr int lds = (RKG * IN_RS + RKG * MID_ROWS) * (int)sizeof(Frag<TC>) + (GATED ? GATE_SCRATCH : 0);
    static_assert(lds <= 160 * 1024, "LDS");
    static SrDeviceOnce attr_once;  // one flag per template instantiation, one bit per device
    {
        const hipError_t e = sr_once_per_device(attr_once, [&] { return sr_allow_lds(sr_rcab_kernel<TC, TIn, TOut, GATED>, lds); });
        SR_REQUIRE(e == hipSuccess, "sr_rcab_conv_pair: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const int tiles = ((c.W + TO - 1) / TO) * ((c.H + TO - 1) / TO) * c.B;
    hipLaunchKernelGGL((sr_rcab_kernel<TC, TIn, TOut, GATED>), dim3(tiles), dim3(256), lds, st, c);
    SR_CHECK_LAUNCH("sr_rcab_conv_pair");
    return SR_OK;
}

}  // namespace

#ifdef SR_STAMPS
extern "C" int sr_debug_rcab_stamps(unsigned long long* host16) {
    return hipMemcpyFromSymbol(host16, HIP_SYMBOL(sr_dbg_rcab), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int sr_rcab_pool_tiles(int H, int W) { return ((W + TO - 1) / TO) * ((H + TO - 1) / TO) * 2; }

extern "C" int sr_rcab_conv_pair(const SrRcab* p, void* stream) {
    SR_REQUIRE(p && p->x && p->w1p && p->b1 && p->w2p && p->b2 && p->y, "sr_rcab_conv_pair: null pointer");
    const SrRcab& c = *p;
    SR_REQUIRE(c.B > 0 && c.H > 0 && c.W > 0 && c.C_p == RC, "sr_rcab_conv_pair: bad geometry (64 padded channels only)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (c.compute_dtype == SR_BF16X3) {  // split operands (ABI v11): fp32 tensors throughout
        SR_REQUIRE(c.x_dtype == SR_F32 && c.y_dtype == SR_F32, "sr_rcab_conv_pair: SR_BF16X3 needs fp32 x / y");
        if (c.gate_y) {
            SR_REQUIRE(c.gate_pool && c.gate_w1 && c.gate_b1 && c.gate_w2 && c.gate_b2 && c.x_out, "sr_rcab_conv_pair: gated input: null pointer");
            SR_REQUIRE(c.gate_C > 0 && c.gate_C <= RC && c.gate_Cr > 0 && c.gate_Cr <= 8, "sr_rcab_conv_pair: gated input: C <= 64 and Cr <= 8");
            SR_REQUIRE(c.x_out != c.x && c.x_out != c.y && c.gate_y != c.y && c.gate_pool != c.pool_partial, "sr_rcab_conv_pair: gated input: buffers alias");
            return launch_rcab<bf3, float, float, true>(c, st);
        }
        return launch_rcab<bf3, float, float, false>(c, st);
    }
    SR_REQUIRE(c.compute_dtype == 0 || c.compute_dtype == SR_BF16, "sr_rcab_conv_pair: compute_dtype must be SR_BF16 (or 0) or SR_BF16X3");
    if (c.gate_y) {
        SR_REQUIRE(c.x_dtype == SR_F32, "sr_rcab_conv_pair: the gated input needs fp32 x / gate_y");
        SR_REQUIRE(c.gate_pool && c.gate_w1 && c.gate_b1 && c.gate_w2 && c.gate_b2 && c.x_out, "sr_rcab_conv_pair: gated input: null pointer");
        SR_REQUIRE(c.gate_C > 0 && c.gate_C <= RC && c.gate_Cr > 0 && c.gate_Cr <= 8, "sr_rcab_conv_pair: gated input: C <= 64 and Cr <= 8");
        SR_REQUIRE(c.x_out != c.x && c.x_out != c.y && c.gate_y != c.y && c.gate_pool != c.pool_partial, "sr_rcab_conv_pair: gated input: buffers alias");
        return c.y_dtype == SR_F32 ? launch_rcab<bf16, float, float, true>(c, st) : launch_rcab<bf16, float, bf16, true>(c, st);
    }
    if (c.x_dtype == SR_F32) return c.y_dtype == SR_F32 ? launch_rcab<bf16, float, float, false>(c, st) : launch_rcab<bf16, float, bf16, false>(c, st);
    return c.y_dtype == SR_F32 ? launch_rcab<bf16, bf16, float, false>(c, st) : launch_rcab<bf16, bf16, bf16, false>(c, st);
}
