// Fused (shifted-)window attention half of a Swin block, one launch:
//     out = x + proj( softmax(q k^T + bias + mask) v ),   q,k,v = qkv( LayerNorm1(x) )
// (swinir.py:146-171 / hat.py:164-192 with window_partition, torch.roll and window_reverse folded
// into addressing).  The fp32 stream is read ONCE and written ONCE; q, k, v, the logits and the
// attention output never leave the CU.
//
// One workgroup = two 8x8 windows (2 x 64 tokens), 12 waves; wave (w, h) OWNS HEAD h of window w end to end:
//   S0  every wave loads its 64 x 32-column slice of x in accumulator layout (it stays in registers
//       as the residual), LayerNorm statistics are combined across waves through 3 KiB of LDS,
//       normalised bf16 rows go to the K-group-major LDS image                        | 3 barriers
//   S1  QKV projection for head h only (96 of the 576 columns): q, k in "swapped" orientation,
//       v un-swapped, weights streamed through a 3-slot register ring straight from L2
//   S2  the accumulators ARE the next MFMA operands (q/k tiles -> K Q^T fragments, v tiles -> V^T
//       fragments, both with the same permuted d / key order, so no lane movement and no LDS):
//       S^T = K Q^T, + relative-position bias (fragment-ordered table), -100 shift mask computed
//       from window coordinates, softmax in registers (+2 cross-lane steps), O^T = V^T P^T
//   S3  O -> LDS (bf16, natural feature order)                                        | barrier
//       proj for output columns [32h, 32h+32) on top of the residual registers, 16-byte stores
//       scattered back through the window map.
// 2 x 75 KiB LDS and <= 168 VGPRs -> one 12-wave workgroup = 3 waves on every SIMD of a CU.
#include "sr_common.h"
#include "sr_host.h"

namespace {

__device__ unsigned long long sr_dbg_swa[16];
#define STAMP(i) SR_STAMP(sr_dbg_swa, i)

struct SwinAttnDev {
    SrSwinAttn a;
    FastDiv div_nw, div_nwx;  // windows per image, windows per row
};

SR_DEV int region3(int v, int size, int ws, int shift) { return v < size - ws ? 0 : (v < size - shift ? 1 : 2); }

SR_DEV Frag<bf16> pack2(const f32x4& lo, const f32x4& hi) {
    Frag<bf16> f;
    f.v[0] = (bf16)lo[0]; f.v[1] = (bf16)lo[1]; f.v[2] = (bf16)lo[2]; f.v[3] = (bf16)lo[3];
    f.v[4] = (bf16)hi[0]; f.v[5] = (bf16)hi[1]; f.v[6] = (bf16)hi[2]; f.v[7] = (bf16)hi[3];
    return f;
}

// Specialised for Cp = 192, heads = 6, hd_p = 32, ws = 8 (SwinIR / HAT-w8 default geometry).
//
// Register budget (<= 168 VGPRs so that two 6-wave workgroups always fit a CU; round 1 measured 140 B/lane of spills
// = +64 MB of HBM traffic per launch when the residual slice lived in registers): the fp32 residual slice is parked in
// LDS between LayerNorm and the proj epilogue, the attention output reuses the LayerNorm image (one extra barrier),
// q / k / v run as three 32-column passes, the attention core handles one 16-query tile at a time, and the weight
// stream is a 5-slot x 2-fragment ring that runs 4 chunks (~512 cycles) ahead through q, k, v and proj.
template <bool MLP>
__global__ __launch_bounds__(768, 3) void sr_swin_attn_kernel(SwinAttnDev dv) {
    constexpr int NTOK = 64, KC = 6, HEADS = 6, WS = 8, RING = 6;
    const SrSwinAttn& a = dv.a;
    // One workgroup = TWO windows (12 waves = exactly 3 per SIMD): two independent 6-wave workgroups at 3 waves per SIMD
    // only co-reside when the hardware happens to start the second one on the right SIMD (measured: it mostly does not).
    constexpr int LDS_PER_WINDOW = 24 * 64 * 16 + 6 * 8 * 64 * 16 + 64 * 6 * 2 * 4;
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave_all >= HEADS ? 1 : 0;
    char* smem = smem_all + pair * LDS_PER_WINDOW;
    Frag<bf16>* Aimg = reinterpret_cast<Frag<bf16>*>(smem);                       // [24][64] LayerNorm1(x), later the attention output
    f32x4* stash = reinterpret_cast<f32x4*>(Aimg + KC * 4 * NTOK);                // [6 waves][8][64 lanes] residual slice (fp32)
    float* red = reinterpret_cast<float*>(stash + HEADS * 8 * 64);                // [64][6][2] LayerNorm partial sums

    const int lane = threadIdx.x & 63;
    const int h = wave_all - pair * HEADS;  // wave = (window of the pair, head)
    const int ar = lane & 15, ag = lane >> 4;
    STAMP(0);

    // ---- window geometry
    uint32_t bimg, win, wy, wx;
    const int n_windows = a.B * (a.H / WS) * (a.W / WS);
    int widx = blockIdx.x * 2 + pair;
    const bool live = widx < n_windows;  // odd window count: the last workgroup's second half recomputes a window and stores nothing
    if (!live) widx = n_windows - 1;
    dv.div_nw.divmod((uint32_t)widx, bimg, win);
    dv.div_nwx.divmod(win, wy, wx);
    const int shift_y = a.y_mode == SR_Y_ROLL ? a.shift : 0;  // strips arrive already rolled in y (halo exchange)
    auto pixel_row = [&](int m) {  // image-order row of token 16m + ar (roll + partition as one gather); recomputed at
        const int t = m * 16 + ar;  // both ends of the kernel instead of holding 4 addresses in registers throughout
        int y = wy * WS + (t >> 3) + shift_y;
        int x = wx * WS + (t & 7) + a.shift;
        if (y >= a.H) y -= a.H;
        if (x >= a.W) x -= a.W;
        return ((int)bimg * a.H + y) * a.W + x;
    };

    // ---- weight stream: 2 n-tiles x 6 K-chunks for each of q, k, v, proj = positions 0..23; slot = t % RING
    // (uniform fragment base in SGPRs + lane offset in one VGPR: no per-fragment 64-bit address registers)
    const Frag<bf16>* Wq = reinterpret_cast<const Frag<bf16>*>(a.wqkv);
    const Frag<bf16>* Wp = reinterpret_cast<const Frag<bf16>*>(a.wproj);
    Frag<bf16> wr[RING][2];
    auto stream_load = [&](int t) {  // t is a compile-time constant at every call site
        if (t < 3 * KC) {
            const int part = t / KC, c = t - part * KC;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const Frag<bf16>* fb = Wq + ((size_t)(part * 2 * HEADS + 2 * h + n) * KC + c) * 64;  // wave-uniform
                wr[t % RING][n] = fb[lane];
            }
        } else if (t < 4 * KC) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const Frag<bf16>* fb = Wp + ((size_t)(2 * h + n) * KC + (t - 3 * KC)) * 64;
                wr[t % RING][n] = fb[lane];
            }
        } else if (MLP && t < 6 * KC) {  // fc1: wave h owns hidden columns [64h, 64h+64) = n-tiles 4h..4h+3, two per pass
            const int half = (t - 4 * KC) / KC, c = (t - 4 * KC) - half * KC;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const Frag<bf16>* fb = reinterpret_cast<const Frag<bf16>*>(a.w1p) + ((size_t)(4 * h + 2 * half + n) * KC + c) * 64;
                wr[t % RING][n] = fb[lane];
            }
        } else if (MLP && t < 6 * KC + 2 * KC) {  // fc2: K = 384 = 12 chunks, wave h owns output columns [32h, 32h+32)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const Frag<bf16>* fb = reinterpret_cast<const Frag<bf16>*>(a.w2p) + ((size_t)(2 * h + n) * (2 * KC) + (t - 6 * KC)) * 64;
                wr[t % RING][n] = fb[lane];
            }
        }
    };
    // ---- S0: x slice + one-pass LayerNorm1 statistics -> Aimg ; slice parked in LDS
    {
        f32x4 xr[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) xr[m][n] = load4(a.x + (size_t)pixel_row(m) * a.ldx + h * 32 + n * 16 + ag * 4);
        __builtin_amdgcn_sched_barrier(0);
        // the x tile is the only thing LayerNorm1 waits for: its loads go out first, the weight ring (needed ~3k cycles
        // later) queues behind them in the CU's memory pipeline
#pragma unroll
        for (int c = 0; c < RING - 1; ++c) stream_load(c);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(1);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s1 += xr[m][n][r];
                    s2 += xr[m][n][r] * xr[m][n][r];
                }
            s1 = wave_sum_xor(s1, 16);
            s1 = wave_sum_xor(s1, 32);
            s2 = wave_sum_xor(s2, 16);
            s2 = wave_sum_xor(s2, 32);
            if (ag == 0) {
                red[((m * 16 + ar) * HEADS + h) * 2] = s1;
                red[((m * 16 + ar) * HEADS + h) * 2 + 1] = s2;
            }
            stash[(h * 8 + m * 2) * 64 + lane] = xr[m][0];
            stash[(h * 8 + m * 2 + 1) * 64 + lane] = xr[m][1];
        }
        __syncthreads();
        const float inv = 1.0f / (float)a.C;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < HEADS; ++w) {
                s1 += red[((m * 16 + ar) * HEADS + w) * 2];
                s2 += red[((m * 16 + ar) * HEADS + w) * 2 + 1];
            }
            const float mean = s1 * inv;
            const float var = fmaxf(s2 * inv - mean * mean, 0.f);
            const float rstd = rsqrtf(var + a.eps);
            const float nmr = -mean * rstd;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bf16x4 ob;  // gamma / beta are folded into wqkv / bqkv
#pragma unroll
                for (int r = 0; r < 4; ++r) ob[r] = (bf16)__builtin_fmaf(xr[m][n][r], rstd, nmr);
                char* dst = reinterpret_cast<char*>(Aimg + (h * 4 + n * 2 + (ag >> 1)) * NTOK + m * 16 + ar) + (ag & 1) * 8;
                *reinterpret_cast<bf16x4*>(dst) = ob;
            }
        }
    }
    __syncthreads();
    STAMP(2);

    // ---- S1: q, k (swapped: lane = token, registers = 4 features) then v (un-swapped: lane = feature,
    //          registers = 4 tokens) of head h, one 32-column pass each so that only 8 accumulator tiles are live
    Frag<bf16> qf[4], kf[4];
    Frag<bf16> vf[2][2];  // [d tile][32-key step]
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        // Only q carries a bias here: a k bias adds the same q.b_k to every logit of a row and cancels in the softmax,
        // a v bias passes through the row-stochastic P unchanged and is folded into the proj bias at pack time.
        f32x4 b0 = (f32x4)(0.f), b1 = (f32x4)(0.f);
        if (part == 0) {
            b0 = load4(a.bqkv + h * 32 + ag * 4);
            b1 = load4(a.bqkv + h * 32 + 16 + ag * 4);
        }
        f32x4 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m) {  // the bias is the C operand of the first MFMA: no separate add
            acc[m][0] = b0;
            acc[m][1] = b1;
        }
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int t = part * KC + c;
            if (t + RING - 1 < 3 * KC) stream_load(t + RING - 1);  // the proj weights are fetched after the attention core
            const Frag<bf16>* arow = Aimg + (c * 4 + ag) * NTOK + ar;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const Frag<bf16> av = arow[m * 16];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (part < 2)
                        mma(wr[t % RING][n], av, acc[m][n]);
                    else
                        mma(av, wr[t % RING][n], acc[m][n]);
                }
                if (m == 1) __builtin_amdgcn_sched_barrier(0);  // at most two activation fragments in flight (register budget)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(3 + part);
        if (part == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) qf[m] = pack2(acc[m][0], acc[m][1]);
        } else if (part == 1) {
#pragma unroll
            for (int m = 0; m < 4; ++m) kf[m] = pack2(acc[m][0], acc[m][1]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                vf[0][ks] = pack2(acc[2 * ks][0], acc[2 * ks + 1][0]);
                vf[1][ks] = pack2(acc[2 * ks][1], acc[2 * ks + 1][1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // every wave is done reading LayerNorm1(x): the image is reused for the attention output
    STAMP(6);

    // ---- S2: attention for head h, one 16-query tile at a time; the bias fragments of tile qt+1 are in flight
    const bool last_row = a.y_mode != SR_Y_STRIP && (int)wy == a.H / WS - 1, last_col = (int)wx == a.W / WS - 1;
    const bool masked = a.shift > 0 && (last_row || last_col);
    f32x4 colneg = (f32x4)(0.0f);  // -100 where the key's column half differs from the query's (same for every query tile)
    if (masked) {
        const bool qcol = last_col && (ar & 7) >= WS - a.shift;
#pragma unroll
        for (int r = 0; r < 4; ++r) colneg[r] = (last_col && 4 * (ag & 1) + r >= WS - a.shift) != qcol ? -100.0f : 0.0f;
    }
    const f32x4* bias = reinterpret_cast<const f32x4*>(a.bias) + (size_t)h * 16 * 64 + lane;  // [h][qt][kt][lane]
    f32x4 bnext[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) bnext[kt] = bias[kt * 64];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] = bnext[kt];
        if (qt < 3) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) bnext[kt] = bias[((qt + 1) * 4 + kt) * 64];
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) mma(kf[kt], qf[qt], s[kt]);
        if (masked) {
            // label(q) != label(k)  <=>  the row halves differ (last window row only) or the column halves differ (last
            // window column only); key row = 2 kt + (ag >> 1), key column = 4 (ag & 1) + r, query row = 2 qt + (ar >> 3)
            const bool qrow = last_row && 2 * qt + (ar >> 3) >= WS - a.shift;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float rowneg = (last_row && 2 * kt + (ag >> 1) >= WS - a.shift) != qrow ? -100.0f : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] += fminf(rowneg, colneg[r]);
            }
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = wave_max_xor(mx, 16);
        mx = wave_max_xor(mx, 32);
        float sum = 0.f;
        const float nmx = -mx * 1.4426950408889634f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], 1.4426950408889634f, nmx));  // exp(s - max): one fma + v_exp
                s[kt][r] = e;
                sum += e;
            }
        sum = wave_sum_xor(sum, 16);
        sum = wave_sum_xor(sum, 32);
        const float inv_sum = __builtin_amdgcn_rcpf(sum);
        const Frag<bf16> p0 = pack2(s[0], s[1]), p1 = pack2(s[2], s[3]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            f32x4 o = (f32x4)(0.0f);
            mma(vf[dt][0], p0, o);
            mma(vf[dt][1], p1, o);
            o *= inv_sum;
            bf16x4 ob;
            ob[0] = (bf16)o[0]; ob[1] = (bf16)o[1]; ob[2] = (bf16)o[2]; ob[3] = (bf16)o[3];
            char* dst = reinterpret_cast<char*>(Aimg + (h * 4 + dt * 2 + (ag >> 1)) * NTOK + qt * 16 + ar) + (ag & 1) * 8;
            *reinterpret_cast<bf16x4*>(dst) = ob;
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the four query tiles sequential (register pressure)
    }
#pragma unroll
    for (int c = 0; c < RING - 1; ++c) stream_load(3 * KC + c);
    STAMP(7);
    __syncthreads();
    STAMP(8);

    // ---- S3: proj columns [32h, 32h+32) on top of the residual slice (back from LDS) + bias
    {
        const f32x4 bp0 = load4(a.bproj + h * 32 + ag * 4), bp1 = load4(a.bproj + h * 32 + 16 + ag * 4);
        f32x4 xr[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            xr[m][0] = stash[(h * 8 + m * 2) * 64 + lane] + bp0;
            xr[m][1] = stash[(h * 8 + m * 2 + 1) * 64 + lane] + bp1;
        }
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int t = 3 * KC + c;
            stream_load(t + RING - 1);
            const Frag<bf16>* orow = Aimg + (c * 4 + ag) * NTOK + ar;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const Frag<bf16> ov = orow[m * 16];
#pragma unroll
                for (int n = 0; n < 2; ++n) mma(wr[t % RING][n], ov, xr[m][n]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(10);
        if constexpr (MLP) {
            // ================= MLP half of the block on the same 64 tokens: x1 (= xr) never leaves the CU =================
            // LayerNorm2 statistics of x1 across the 6 column slices (through `red`), then the normalised rows replace the
            // attention output in the LDS image; the hidden activations replace the parked residual slices (48 KiB).
            Frag<bf16>* Himg = reinterpret_cast<Frag<bf16>*>(stash);  // [48][64]
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {  // pad channels of x1 are exactly 0 (zero weight rows / bias / residual pad)
                        s1 += xr[m][n][r];
                        s2 += xr[m][n][r] * xr[m][n][r];
                    }
                s1 = wave_sum_xor(s1, 16);
                s1 = wave_sum_xor(s1, 32);
                s2 = wave_sum_xor(s2, 16);
                s2 = wave_sum_xor(s2, 32);
                if (ag == 0) {
                    red[((m * 16 + ar) * HEADS + h) * 2] = s1;
                    red[((m * 16 + ar) * HEADS + h) * 2 + 1] = s2;
                }
            }
            __syncthreads();  // proj reads of the image are done everywhere + LN2 partials are visible
            {
                const float inv = 1.0f / (float)a.C;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int w = 0; w < HEADS; ++w) {
                        s1 += red[((m * 16 + ar) * HEADS + w) * 2];
                        s2 += red[((m * 16 + ar) * HEADS + w) * 2 + 1];
                    }
                    const float mean = s1 * inv;
                    const float rstd = rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + a.eps);
                    const float nmr = -mean * rstd;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        bf16x4 ob;  // gamma2 / beta2 are folded into w1p / b1
#pragma unroll
                        for (int r = 0; r < 4; ++r) ob[r] = (bf16)__builtin_fmaf(xr[m][n][r], rstd, nmr);
                        char* dst = reinterpret_cast<char*>(Aimg + (h * 4 + n * 2 + (ag >> 1)) * NTOK + m * 16 + ar) + (ag & 1) * 8;
                        *reinterpret_cast<bf16x4*>(dst) = ob;
                    }
                }
            }
            __syncthreads();
            STAMP(11);
            // fc1 (two passes of 32 hidden columns) -> GELU -> hidden image
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x4 c0 = load4(a.b1 + h * 64 + half * 32 + ag * 4), c1 = load4(a.b1 + h * 64 + half * 32 + 16 + ag * 4);
                f32x4 acc[4][2];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[m][0] = c0;
                    acc[m][1] = c1;
                }
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const int t = 4 * KC + half * KC + c;
                    stream_load(t + RING - 1);
                    const Frag<bf16>* arow = Aimg + (c * 4 + ag) * NTOK + ar;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const Frag<bf16> av = arow[m * 16];
#pragma unroll
                        for (int n = 0; n < 2; ++n) mma(wr[t % RING][n], av, acc[m][n]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int col = h * 64 + half * 32 + n * 16 + ag * 4;
                    char* hbase = reinterpret_cast<char*>(Himg + (col >> 3) * NTOK + ar) + (ag & 1) * 8;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const f32x4 v = acc[m][n];
                        bf16x4 hb;
#pragma unroll
                        for (int r = 0; r < 4; ++r) hb[r] = (bf16)gelu_bf16(v[r]);
                        *reinterpret_cast<bf16x4*>(hbase + m * 16 * sizeof(Frag<bf16>)) = hb;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                STAMP(12 + half);
            }
            {
                const f32x4 d0 = load4(a.b2 + h * 32 + ag * 4), d1 = load4(a.b2 + h * 32 + 16 + ag * 4);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    xr[m][0] += d0;
                    xr[m][1] += d1;
                }
            }
            __syncthreads();
            STAMP(14);
            // fc2 on top of x1
#pragma unroll
            for (int c = 0; c < 2 * KC; ++c) {
                const int t = 6 * KC + c;
                stream_load(t + RING - 1);
                const Frag<bf16>* hrow = Himg + (c * 4 + ag) * NTOK + ar;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const Frag<bf16> hv = hrow[m * 16];
#pragma unroll
                    for (int n = 0; n < 2; ++n) mma(wr[t % RING][n], hv, xr[m][n]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP(15);
        int arl = ar;
        asm volatile("" : "+v"(arl));  // opaque copy: keeps hipcc from hoisting (and spilling) the four store addresses
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int t = m * 16 + arl;
            int y = wy * WS + (t >> 3) + shift_y;
            int x = wx * WS + (t & 7) + a.shift;
            if (y >= a.H) y -= a.H;
            if (x >= a.W) x -= a.W;
            float* dst = a.out + (size_t)(((int)bimg * a.H + y) * a.W + x) * a.ldx + h * 32 + ag * 4;
            if (live) {
                store4(dst, xr[m][0]);
                store4(dst + 16, xr[m][1]);
            }
        }
    }
    STAMP(9);
}

}  // namespace

extern "C" int sr_debug_swa_stamps(unsigned long long* host16) {
    return hipMemcpyFromSymbol(host16, HIP_SYMBOL(sr_dbg_swa), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}

extern "C" int sr_swin_attn_supported(int Cp, int heads, int hd_p, int ws, int compute_dtype) {
    return (compute_dtype == SR_BF16 && Cp == 192 && heads == 6 && hd_p == 32 && ws == 8) ? 1 : 0;
}

extern "C" int sr_swin_attn_fused(const SrSwinAttn* p, void* stream) {
    SR_REQUIRE(p && p->x && p->out && p->wqkv && p->bqkv && p->wproj && p->bproj && p->bias, "sr_swin_attn_fused: null pointer");
    const SrSwinAttn& a = *p;
    SR_REQUIRE(sr_swin_attn_supported(a.Cp, a.heads, a.hd_p, a.ws, SR_BF16), "sr_swin_attn_fused: unsupported geometry (use sr_gemm + sr_window_attention)");
    SR_REQUIRE(a.B > 0 && a.H % a.ws == 0 && a.W % a.ws == 0 && a.shift >= 0 && a.shift < a.ws && a.C > 0 && a.C <= a.Cp && a.ldx >= a.Cp &&
                   a.y_mode >= SR_Y_ROLL && a.y_mode <= SR_Y_STRIP_LAST,
               "sr_swin_attn_fused: bad geometry");
    if (a.w1p) SR_REQUIRE(a.b1 && a.w2p && a.b2 && a.Hp == 384, "sr_swin_attn_fused: the fused MLP tail needs w1p/b1/w2p/b2 and Hp == 384");
    // the one-window-per-workgroup kernel keeps the softmax denominator in the first pad channel of every head: needs hd = 30 of 32
    if (sr_swin_block_v2_enabled() && a.C == 180) return sr_swin_block_v2(a, reinterpret_cast<hipStream_t>(stream));
    SwinAttnDev dv;
    dv.a = a;
    const int nwx = a.W / a.ws, nwy = a.H / a.ws;
    dv.div_nw = make_fastdiv((uint32_t)(nwx * nwy));
    dv.div_nwx = make_fastdiv((uint32_t)nwx);
    constexpr int lds = 2 * (24 * 64 * 16 + 6 * 8 * 64 * 16 + 64 * 6 * 2 * 4);  // per window: 24 KiB image + 48 KiB residual slices + 3 KiB LN partials
    static SrDeviceOnce attr_once;  // one flag per template instantiation, one bit per device
    {
        const hipError_t e = sr_once_per_device(attr_once, [&] { hipError_t e2 = sr_allow_lds(sr_swin_attn_kernel<false>, lds); return e2 != hipSuccess ? e2 : sr_allow_lds(sr_swin_attn_kernel<true>, lds); });
        SR_REQUIRE(e == hipSuccess, "sr_swin_attn_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const dim3 grid((a.B * nwx * nwy + 1) / 2);
    if (a.w1p) {
        hipLaunchKernelGGL(sr_swin_attn_kernel<true>, grid, dim3(768), lds, reinterpret_cast<hipStream_t>(stream), dv);
    } else {
        hipLaunchKernelGGL(sr_swin_attn_kernel<false>, grid, dim3(768), lds, reinterpret_cast<hipStream_t>(stream), dv);
    }
    SR_CHECK_LAUNCH("sr_swin_attn_fused");
    return SR_OK;
}
