// One launch = one whole SwinTransformerBlock (swinir.py:146-174), round-3 kernel ("stream" form, C ABI v5 sr_swin_block):
//     x1  = x + proj( softmax(q k^T + bias + mask) v ),   q,k,v = qkv( LayerNorm1(x) )
//     out = x1 + fc2( GELU( fc1( LayerNorm2(x1) ) ) )
// with window_partition, torch.roll and window_reverse folded into addressing.
//
// What the round-3 skeleton probe (tools/micro/skel.hip, profiles/r03_block_skeleton.txt) showed: the one-window / 4-wave / three
// workgroups per CU shape of the round-2 kernel runs the block's GEMMs alone in 21-23 us (B = 8) -- two windows per workgroup with
// shared weights (LDS ring or 128-row register tiles) are no faster -- and tolerates ~2,900 plain VALU per wave on top at 29 us.
// The round-2 kernel needed 44-48 us because its QKV stage was 36 short steps (8 + 4 + 4 MFMAs) on one-deep rings, its biases,
// accumulator initialisations and second x read were VALU / TA work, and q was projected twice.  Here:
//   * EVERY GEMM stage is a run of the same step: 4 activation fragments (LDS) x 3 weight fragments (global, fragment order)
//     -> 12 MFMAs, weights two steps ahead in a 3-slot register ring that never drains: the 48 slots of a block
//     (per pass: 6 QKV + 2 proj; then 6 fc1 + 6 fc2 per hidden half) are ONE stream in consumption order, so the ring keeps
//     prefetching through LayerNorm / softmax / GELU phases and every stage starts with its first two slots in registers.
//   * QKV pass p (heads 2p, 2p+1): wave w = (head hh = w >> 1, half = w & 1) projects q, k and v of d-half `half` of its head
//     for all 64 tokens (3 n-tiles x 4 m-tiles); q and k land in K-group-major images, v transposed in the key order the
//     accumulator-as-operand trick needs.  The same wave then runs the attention atom (head hh, queries [32 half, +32)).
//   * no bias anywhere in the kernel: the LayerNorm image carries 1.0 in its pad channels 180 / 181, the weight rows there hold
//     the bias as a hi + lo bf16 pair (q, fc1); v's pad feature 30 is wired to 1.0 the same way, which makes row 30 of O^T the
//     softmax denominator (as in round 2) and, after normalisation, a constant one that carries the proj bias; two hidden pad
//     columns carry the fc2 bias.  Accumulators start as the C = 0 operand of their first MFMA.
//   * LayerNorm1 runs in the accumulator layout like LayerNorm2 (one code path): x is read ONCE, straight into the registers
//     that hold the residual, x1 and finally the output.
//   * softmax on exp2: log2(e) is folded into the q rows and the bias table.
#include "sr_swin_stream.h"

namespace {

#ifdef SR_STAMPS
__device__ unsigned long long sr_dbg_sw3[64];
#ifndef SR_STAMP_WAVE
#define SR_STAMP_WAVE 0
#endif
#define STAMP(i)                                                                                                 \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (blockIdx.x == 7 && threadIdx.x == 64 * SR_STAMP_WAVE) sr_dbg_sw3[i] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

#ifdef SR_WGTRACE
// diagnostic build only: per workgroup [start realtime, end realtime, (XCC id << 32) | HW_ID, start s_memtime, end s_memtime, -]
__device__ unsigned long long sr_dbg_wgtrace[4096 * 6];
#define WGTRACE(k)                                                                                                        \
    do {                                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                                                      \
            sr_dbg_wgtrace[blockIdx.x * 6 + (k)] = __builtin_amdgcn_s_memrealtime();                                      \
            sr_dbg_wgtrace[blockIdx.x * 6 + 3 + (k)] = __builtin_amdgcn_s_memtime();                                      \
            if ((k) == 0)                                                                                                 \
                sr_dbg_wgtrace[blockIdx.x * 6 + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | \
                                                      __builtin_amdgcn_s_getreg(4 | (31 << 11));                         \
        }                                                                                                                 \
    } while (0)
#else
#define WGTRACE(k) do { } while (0)
#endif

struct SwinBlock3Dev {
    SrSwinBlock a;
    FastDiv div_nw, div_nwx;  // windows per image, windows per row
    int nwin;                 // windows of the launch (B * windows per image); the grid may be smaller (SrSwinBlock.max_workgroups): workgroup b then
                              // takes windows b, b + gridDim.x, ... and fetches the next window's rows while its result rows leave
};


// Specialised for C = 180 (Cp = 192), heads = 6, hd = 30 (32), ws = 8, hidden 360 (384): SwinIR / SwinFIR default geometry.
// T = bf16: bf16 operands (three workgroups per CU).  T = bf3 (compute type SR_BF16X3, precision "fp32x3"): every operand is a hi + lo
// bf16 pair and every product hi*hi + hi*lo + lo*hi -- fp32-class accuracy on the bf16 matrix cores; images are 32-B cells (98 KiB:
// one workgroup per CU), GELU is the erf form.
template <typename T>
__global__ __launch_bounds__(256, sizeof(Frag<T>) == 16 ? 3 : 1) void sr_swin_block3_kernel(SwinBlock3Dev dv) {
    const SrSwinBlock& a = dv.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<T>* Aimg = reinterpret_cast<Frag<T>*>(smem);
    Frag<T>* Qimg = Aimg + CELLS_A;
    Frag<T>* Kimg = Qimg + CELLS_Q;
    Frag<T>* Vimg = Kimg + CELLS_K;
    float* red = reinterpret_cast<float*>(smem + Lds<T>::RED_OFF);
    Frag<T>* Himg = Qimg;  // [24][64] hidden half (MLP stage: Q / K / V are dead)

    int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (re-read through an opaque asm at the top of every trip of the window loop, see there)
    const int lane0 = threadIdx.x & 63;
    int lane = lane0, ar = lane & 15, ag = lane >> 4;
    // hipcc would hoist every per-lane LDS / global offset to kernel entry and keep ~40 address registers alive (and spilled);
    // relane() makes the lane id opaque again so that offsets are recomputed where they are used.
    auto relane = [&]() {
        lane = lane0;
        asm volatile("" : "+v"(lane));
        ar = lane & 15;
        ag = lane >> 4;
    };

    // ---- window geometry.  One window per workgroup, or (grid < nwin) a persistent workgroup walking windows blockIdx.x + k gridDim.x
    uint32_t bimg, win, wy, wx;
    const int shift_y = a.y_mode == SR_Y_ROLL ? a.shift : 0;  // strips arrive already rolled in y (halo exchange)
    // Token rows of a window as [row base pointer (one per window row: a wave's 16 tokens are window rows 2 w, 2 w + 1)] + [byte offset of the column,
    // the same eight values for every window row, for x and for out]: roll + partition as one gather with 2 scalar pointers + 8 scalar offsets live
    // instead of 16 pointers (a persistent workgroup addresses two windows at the end of a trip: the next one's rows and its own result rows).
    struct WinRows {
        size_t rbase[2];  // element offset of pixel (y, 0) for the wave's two window rows
        int coff[8];      // byte offset of the window's eight columns inside an image row
    };
    auto rows_of = [&](uint32_t bi, uint32_t gy, uint32_t gx) {
        WinRows r;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int y = gy * WS + 2 * w + j + shift_y;
            if (y >= a.H) y -= a.H;
            r.rbase[j] = (size_t)(((int)bi * a.H + y) * a.W) * a.ldx;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            int x = gx * WS + c + a.shift;
            if (x >= a.W) x -= a.W;
            r.coff[c] = x * a.ldx * 4;
        }
        return r;
    };
    int item = blockIdx.x;
    dv.div_nw.divmod((uint32_t)item, bimg, win);
    dv.div_nwx.divmod(win, wy, wx);
    bimg = __builtin_amdgcn_readfirstlane(bimg);
    wy = __builtin_amdgcn_readfirstlane(wy);
    wx = __builtin_amdgcn_readfirstlane(wx);

    STAMP(0);
    WGTRACE(0);
#ifdef SR_EXP_DEPHASE
    // experiment: the second / third workgroup of a CU (dispatch order: blocks b, b + 256, b + 512 share a CU) starts its x fetch late
    for (int i = 0, k = (blockIdx.x >> 8) % 3; i < k; ++i) __builtin_amdgcn_s_sleep(SR_EXP_DEPHASE);
#endif
    // ---- x: 16 full token rows per wave by LDS-DMA (window gather = one scalar row address per piece), then the first weight slots
    const unsigned tile_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto fetch_rows = [&](uint32_t bi, uint32_t gy, uint32_t gx) {
        const WinRows r = rows_of(bi, gy, gx);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#if !defined(SR_EXP_NOXIO) && !defined(SR_EXP_NOXLOAD)
            dma_row48(a.x + r.rbase[i >> 3], r.coff[i & 7], __builtin_amdgcn_readfirstlane(tile_lds + (16 * w + i) * XS), lane);
#endif
        }
    };
    fetch_rows(bimg, wy, wx);
    __builtin_amdgcn_sched_barrier(0);
    WStream<T> ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wstream), 0, NSLOT * 12 * 64 * (int)sizeof(Frag<T>), 0x00020000);
    ws.wave_frag = w * 3;
#pragma unroll
    for (int s0 = 0; s0 < WStream<T>::DIST; ++s0) ws.load(s0, lane);
    const __amdgpu_buffer_rsrc_t bias_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, 6 * 16 * 64 * 16, 0x00020000);
    __builtin_amdgcn_sched_barrier(0);
    bool first = true;
    for (;;) {  // one window per trip
    // The window's 16 rows per wave have landed; younger operations may still fly: the first weight slots, and behind a previous trip also its 16 row stores
    // (vector-memory operations complete in issue order: rows -> [row stores] -> weight slots).
#ifndef SR_EXP_XNOWAIT
    if (first)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WStream<T>::LOADS_PER_SLOT * WStream<T>::DIST) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WStream<T>::LOADS_PER_SLOT * WStream<T>::DIST + 16) : "memory");
#endif
    BLOCK_SYNC();
    // The wave index is made opaque once per trip: everything derived from it (144 weight-slot offsets, bias-tile offsets, image cells) would otherwise be
    // loop-invariant, get hoisted out of the window loop and spill ~160 SGPRs; this way a trip compiles like the one-window kernel did.
    asm volatile("" : "+s"(w));
    ws.wave_frag = w * 3;
    const int hh = w >> 1, half = w & 1;  // GEMM role: d-half `half` of head 2p + hh; attention atom: queries [32 half, +32) of that head
    f32x4 x1[4][3];  // [m][n]: token 16 m + ar, channels 48 w + 16 n + 4 ag .. +3   (the residual, then x1, then the output)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const char* xm = smem + (m * 16 + ar) * XS + (w * 48 + ag * 4) * 4;
#pragma unroll
        for (int n = 0; n < 3; ++n) x1[m][n] = *reinterpret_cast<const f32x4*>(xm + n * 64);
    }

    // LayerNorm of the 64 tokens held in x1 (each wave owns 48 of the 192 channels) -> bf16 image; gamma / beta live in the weights.
    // Pad channels 180, 181 of the image are the constant one that the bias rows of the stream multiply.
    const bool one_lane = (w == ONE_C / 48) && (ag == (ONE_C % 16) / 4);  // channels 180.. = wave 3, n = 2, ag = 1, r = 0, 1
    static_assert(ONE_C % 48 / 16 == 2 && ONE_C % 4 == 0, "position of the constant-one channels");
    auto layernorm_to_image = [&]() {
        {
            float q1[4], q2[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 t1 = x1[m][0] + x1[m][1] + x1[m][2];  // pad channels of the stream are exactly 0
                f32x4 t2 = x1[m][0] * x1[m][0];
#pragma unroll
                for (int r = 0; r < 4; ++r) t2[r] = __builtin_fmaf(x1[m][2][r], x1[m][2][r], __builtin_fmaf(x1[m][1][r], x1[m][1][r], t2[r]));
                q1[m] = (t1[0] + t1[1]) + (t1[2] + t1[3]);
                q2[m] = (t2[0] + t2[1]) + (t2[2] + t2[3]);
            }
            const float s1 = rows_reduce_scatter4(q1[0], q1[1], q1[2], q1[3]);  // lane row ag: token tile m = ag
            const float s2 = rows_reduce_scatter4(q2[0], q2[1], q2[2], q2[3]);
            *reinterpret_cast<float2*>(red + ((ag * 16 + ar) * 4 + w) * 2) = make_float2(s1, s2);
        }
        BLOCK_SYNC();
        const float inv = 1.0f / (float)a.C;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 pa = *reinterpret_cast<const f32x4*>(red + (m * 16 + ar) * 8), pb = *reinterpret_cast<const f32x4*>(red + (m * 16 + ar) * 8 + 4);
            const float mean = (pa[0] + pa[2] + pb[0] + pb[2]) * inv;
            const float rstd = rsqrtf(fmaxf((pa[1] + pa[3] + pb[1] + pb[3]) * inv - mean * mean, 0.f) + a.eps);
            const float nmr = -mean * rstd;
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                f32x4 nv;
#pragma unroll
                for (int r = 0; r < 4; ++r) nv[r] = __builtin_fmaf(x1[m][n][r], rstd, nmr);
                if (n == 2) {
                    nv[0] = one_lane ? 1.0f : nv[0];
                    nv[1] = one_lane ? 1.0f : nv[1];
                }
                st_half(Aimg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, nv);
            }
        }
    };
    layernorm_to_image();
    STAMP(1);
    BLOCK_SYNC();
    STAMP(2);

    // ---- shift mask terms that do not depend on the pass (common.py:250-274 from window coordinates), in log2 units
    const bool last_row = a.y_mode != SR_Y_STRIP && (int)wy == a.H / WS - 1, last_col = (int)wx == a.W / WS - 1;
    const bool masked = a.shift > 0 && (last_row || last_col);
    constexpr float NEG = -100.0f * 1.4426950408889634f;

    auto loada_img = [&](const Frag<T>* img) {
        return [&, img](int c, int h, Frag<T> (&av)[2]) {
            const Frag<T>* arow = img + (c * 4 + ag) * NTOK + h * 32 + ar;
            av[0] = arow[0];
            av[1] = arow[16];
        };
    };

    // ---- three passes of two heads
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        relane();
        if (p == 0) PHASE_PRIO(0);
        if (p == 2) PHASE_PRIO(1);
        f32x4 s[2][4];   // S^T tiles of the atom
        f32x4 bt[5];     // bias tiles: tile (qt, kt) of a head depends on kt - qt only (tokens are row-major 8 x 8: a 16-token tile = two window rows)
        {
            // -- QKV: q, k, v of (head 2p + hh, d-half) for all 64 tokens
            f32x4 acc[4][3];
            ws.template run<6>(8 * p, lane, loada_img(Aimg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&av)[2]) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (c == 0) {
                        mma0(b[0], av[m], acc[2 * h + m][0]);  // q: lane = token, registers = 4 features
                        mma0(b[1], av[m], acc[2 * h + m][1]);  // k: likewise
                        mma0(av[m], b[2], acc[2 * h + m][2]);  // v: lane = feature, registers = 4 tokens
                    } else {
                        mma(b[0], av[m], acc[2 * h + m][0]);
                        mma(b[1], av[m], acc[2 * h + m][1]);
                        mma(av[m], b[2], acc[2 * h + m][2]);
                    }
                }
            });
            STAMP(3 + 8 * p);
            {
                // bt[j] = tile with kt - qt = j - 2 half - 1; s[q2][kt] (qt = 2 half + q2) takes bt[kt - q2 + 1].  Fetched from the
                // representative tile (qt', kt') = (max(0, -d), max(0, d)) of the [h][qt][kt][lane] table.
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int d = j - 2 * half - 1;
                    const int rep = (d < 0 ? -d * 4 : d);
                    bt[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bias_rsrc, lane * 16, ((2 * p + hh) * 16 + rep) * 1024, 0));
                }
            }
            if (p > 0) BLOCK_SYNC();  // every wave is done with the previous pass's Q / K / V (attention) and O (proj)
            Frag<T>* qb = Qimg + (hh * 4 + 2 * half + (ag >> 1)) * NTOK + ar;  // cell [d-group][token]: d = 16 half + 4 ag + r
            Frag<T>* kb = Kimg + (hh * 4 + 2 * half + (ag >> 1)) * NTOK + ar;
            Frag<T>* vb = Vimg + (hh * 2 * 4 + ag) * 32 + 16 * half + ar;      // cell [step][key group ag][d]: keys {32 s + 4 ag + r} then {32 s + 16 + 4 ag + r}
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                st_half(qb + m * 16, ag & 1, acc[m][0]);
                st_half(kb + m * 16, ag & 1, acc[m][1]);
                st_half(vb + (m >> 1) * 4 * 32, m & 1, acc[m][2]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(4 + 8 * p);
        BLOCK_SYNC();  // Q / K / V of both heads are in LDS
        STAMP(5 + 8 * p);
        relane();

        // -- attention for the atom: S^T = K Q^T (+bias as the C operand), mask, softmax over keys (exp2), O^T = V^T P^T
        {
            const Frag<T>* qrow = Qimg + (hh * 4 + ag) * NTOK + half * 32 + ar;
            const Frag<T>* kb = Kimg + (hh * 4 + ag) * NTOK + ar;
            Frag<T> qf[2], kf[4];
            qf[0] = qrow[0];
            qf[1] = qrow[16];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) kf[kt] = kb[kt * 16];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) s[qt][kt] = mma_c(kf[kt], qf[qt], bt[kt - qt + 1]);
        }
        const Frag<T>* vb = Vimg + (hh * 2 * 4 + ag) * 32 + ar;
        Frag<T> vf[2][2];  // [d tile][key step]
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int st = 0; st < 2; ++st) vf[dt][st] = vb[st * 4 * 32 + dt * 16];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (masked) {
                f32x4 colneg;  // -100 where the key's column half differs from the query's
                const bool qcol = last_col && (ar & 7) >= WS - a.shift;
#pragma unroll
                for (int r = 0; r < 4; ++r) colneg[r] = (last_col && 4 * (ag & 1) + r >= WS - a.shift) != qcol ? NEG : 0.0f;
                // label(q) != label(k)  <=>  the row halves differ (last window row only) or the column halves differ (last window
                // column only); key row = 2 kt + (ag >> 1), key column = 4 (ag & 1) + r, query row = 4 half + 2 qt + (ar >> 3)
                const bool qrow = last_row && 4 * half + 2 * qt + (ar >> 3) >= WS - a.shift;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const float rowneg = (last_row && 2 * kt + (ag >> 1) >= WS - a.shift) != qrow ? NEG : 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[qt][kt][r] += fminf(rowneg, colneg[r]);
                }
            }
#ifndef SR_EXP_NOVALU
            float mx = max3(s[qt][0][0], s[qt][0][1], s[qt][0][2]), my = max3(s[qt][2][0], s[qt][2][1], s[qt][2][2]);
            mx = max3(mx, s[qt][0][3], s[qt][1][0]);
            my = max3(my, s[qt][2][3], s[qt][3][0]);
            mx = max3(mx, s[qt][1][1], s[qt][1][2]);
            my = max3(my, s[qt][3][1], s[qt][3][2]);
            mx = max3(mx, s[qt][1][3], s[qt][3][3]);
            mx = fmaxf(mx, my);
            mx = wave_max_xor(mx, 16);
            mx = wave_max_xor(mx, 32);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#ifdef SR_EXP_NOTRANS
                    s[qt][kt][r] = s[qt][kt][r] - mx;
#else
                    s[qt][kt][r] = __builtin_amdgcn_exp2f(s[qt][kt][r] - mx);  // exp(logit - max): the logits are in log2 units
#endif
                }
#endif
            const Frag<T> p0 = pack2<T>(s[qt][0], s[qt][1]), p1 = pack2<T>(s[qt][2], s[qt][3]);
            f32x4 o0, o1;
            mma0(vf[0][0], p0, o0);
            mma(vf[0][1], p1, o0);
            mma0(vf[1][0], p0, o1);
            mma(vf[1][1], p1, o1);
            // row d = 30 of O^T (lanes 48..63, register 2 of the second d tile) is sum_k P[q][k]: the softmax denominator of query ar
            const float inv_sum = __builtin_amdgcn_rcpf(bcast_row3(o1[PAD_D & 3]));
            o0 *= inv_sum;
            o1 *= inv_sum;  // feature 30 becomes 1: it multiplies the proj bias rows; feature 31 is 0
            Frag<T>* ob = Qimg + (hh * 4 + (ag >> 1)) * NTOK + half * 32 + qt * 16 + ar;  // over this atom's own q cells
            st_half(ob, ag & 1, o0);
            st_half(ob + 2 * NTOK, ag & 1, o1);
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(6 + 8 * p);
        BLOCK_SYNC();  // the O chunk (64 tokens x 64 channels) is complete
        STAMP(7 + 8 * p);

        // -- proj partial: x1 columns [48 w, 48 w + 48) += O_chunk @ Wproj[:, 64 p .. 64 p + 64)
        ws.template run<2>(8 * p + 6, lane, loada_img(Qimg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&ov)[2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) mma(b[n], ov[m], x1[2 * h + m][n]);
        });
        __builtin_amdgcn_sched_barrier(0);
        STAMP(8 + 8 * p);
    }

    // ---- MLP: LayerNorm2, then fc1 / GELU / fc2 in two hidden halves of 192 columns
    relane();
    PHASE_PRIO(2);
    layernorm_to_image();  // (its barrier also orders the last proj reads of O before the hidden image overwrites the region)
    STAMP(27);
    BLOCK_SYNC();
    STAMP(28);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        relane();
        f32x4 acc[4][3];
        ws.template run<6>(24 + 12 * hf, lane, loada_img(Aimg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&av)[2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    if (c == 0)
                        mma0(b[n], av[m], acc[2 * h + m][n]);
                    else
                        mma(b[n], av[m], acc[2 * h + m][n]);
                }
        });
        STAMP(30 + 5 * hf);
        if (hf == 1) BLOCK_SYNC();  // fc2 of the first half has read the hidden image everywhere
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 g;
#pragma unroll
#if defined(SR_EXP_NOVALU)
                for (int r = 0; r < 4; ++r) g[r] = acc[m][n][r];
#elif defined(SR_EXP_NOTRANS)
                for (int r = 0; r < 4; ++r) g[r] = acc[m][n][r] * 0.5f;
#elif defined(SR_EXP_GELUPOLY)
                for (int r = 0; r < 4; ++r) g[r] = gelu_poly(acc[m][n][r]);
#else
                for (int r = 0; r < 4; ++r) g[r] = gelu_op<T>(acc[m][n][r]);
#endif
                if constexpr (sizeof(Frag<T>) == 32) {
                    // split-operand path: hidden pad columns 360, 361 (wave 3, n = 1, ag = 2, r = 0, 1 of the second half) are the constant one that
                    // the fc2 bias rows multiply (the bf16 path wires gelu(1) there through the weights instead)
                    if (hf == 1 && n == 1) {
                        const bool one_h = (w == 3) && (ag == 2);
                        g[0] = one_h ? 1.0f : g[0];
                        g[1] = one_h ? 1.0f : g[1];
                    }
                }
                st_half(Himg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, g);
            }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(31 + 5 * hf);
        BLOCK_SYNC();
        STAMP(32 + 5 * hf);
        // fc2 partial on top of x1: K = the 192 hidden columns of this half
        ws.template run<6>(30 + 12 * hf, lane, loada_img(Himg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&hv)[2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) mma(b[n], hv[m], x1[2 * h + m][n]);
        });
        STAMP(33 + 5 * hf);
    }

    STAMP(40);
    // ---- store (window_reverse + roll back folded into the row address): accumulator layout -> LDS tile -> 16 full rows per wave
    relane();
    BLOCK_SYNC();  // every wave has read its last hidden fragments: the tile region is free
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        char* xm = smem + (m * 16 + ar) * XS + (w * 48 + ag * 4) * 4;
#pragma unroll
        for (int n = 0; n < 3; ++n) *reinterpret_cast<f32x4*>(xm + n * 64) = x1[m][n];
    }
    BLOCK_SYNC();
    {
        f32x4 rowv[16];
        const int l48 = lane < 48 ? lane : 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) rowv[i] = *reinterpret_cast<const f32x4*>(smem + (16 * w + i) * XS + l48 * 16);
        // Persistent form: this workgroup's next window.  Its 16 rows per wave are fetched into the tile rows this wave has just read (each wave reads and
        // refills only its own 16 rows: no barrier), BEFORE the result rows are issued, so that the fetch flies under the stores and the first weight slots.
        const int next = __builtin_amdgcn_readfirstlane(item + (int)gridDim.x);
        const bool more = next < dv.nwin;
        uint32_t nb = 0, nwin_ = 0, ny = 0, nx = 0;
        if (more) {
            dv.div_nw.divmod((uint32_t)next, nb, nwin_);
            dv.div_nwx.divmod(nwin_, ny, nx);
            nb = __builtin_amdgcn_readfirstlane(nb);
            ny = __builtin_amdgcn_readfirstlane(ny);
            nx = __builtin_amdgcn_readfirstlane(nx);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // rowv is in registers: the rows may be overwritten
            fetch_rows(nb, ny, nx);
        }
        {
            const WinRows r = rows_of(bimg, wy, wx);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#if !defined(SR_EXP_NOXIO) && !defined(SR_EXP_NOXSTORE)
                store_row48(a.out + r.rbase[i >> 3], r.coff[i & 7], rowv[i], lane);
#else
                if (rowv[i][0] == 1.2345e-30f) store_row48(a.out + r.rbase[i >> 3], r.coff[i & 7], rowv[i], lane);  // keeps the value live
#endif
            }
        }
        if (!more) break;
        item = next;  // (readfirstlane: the row addresses are scalar operands; the loop-carried values are uniform, which the compiler does not prove)
        bimg = __builtin_amdgcn_readfirstlane(nb);
        wy = __builtin_amdgcn_readfirstlane(ny);
        wx = __builtin_amdgcn_readfirstlane(nx);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s0 = 0; s0 < WStream<T>::DIST; ++s0) ws.load(s0, lane);
        __builtin_amdgcn_sched_barrier(0);
        first = false;
    }
    }  // for (;;)
    STAMP(41);
    WGTRACE(1);
}

}  // namespace

#ifdef SR_STAMPS
extern "C" int sr_debug_sw3_stamps(unsigned long long* host64) {
    return hipMemcpyFromSymbol(host64, HIP_SYMBOL(sr_dbg_sw3), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

#ifdef SR_WGTRACE
extern "C" int sr_debug_sw3_wgtrace(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(sr_dbg_wgtrace), (size_t)n * 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int sr_swin_block_supported(int C, int Cp, int heads, int hd_p, int ws, int Hp, int compute_dtype) {
    return ((compute_dtype == SR_BF16 || compute_dtype == SR_BF16X3) && C == 180 && Cp == 192 && heads == 6 && hd_p == 32 && ws == 8 && Hp == 384) ? 1 : 0;
}

// Workgroups the device holds at once (bf16: three per CU; bf16x3: one): the default grid of a launch with more windows than that
static int resident_workgroups(int per_cu) {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256 * per_cu;
    int n = cus[dev & 63].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev & 63].store(n, std::memory_order_relaxed);
    }
    return n * per_cu;
}

template <typename T>
static int launch_swin_block3(const SwinBlock3Dev& dv, int nblocks, hipStream_t st) {
    static SrDeviceOnce attr_once;  // one flag per instantiation, one bit per device
    {
        const hipError_t e = sr_once_per_device(attr_once, [&] { return sr_allow_lds(sr_swin_block3_kernel<T>, Lds<T>::TOTAL); });
        SR_REQUIRE(e == hipSuccess, "sr_swin_block: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(sr_swin_block3_kernel<T>, dim3(nblocks), dim3(256), Lds<T>::TOTAL, st, dv);
    SR_CHECK_LAUNCH("sr_swin_block");
    return SR_OK;
}

extern "C" int sr_swin_block(const SrSwinBlock* p, void* stream) {
    SR_REQUIRE(p && p->x && p->out && p->wstream && p->bias, "sr_swin_block: null pointer");
    const SrSwinBlock& a = *p;
    const int cdt = a.compute_dtype;  // SR_BF16 or SR_BF16X3
    SR_REQUIRE(sr_swin_block_supported(a.C, a.Cp, a.heads, a.hd_p, a.ws, a.Hp, cdt), "sr_swin_block: unsupported geometry / compute type (use sr_swin_attn_fused / sr_gemm)");
    SR_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.H % a.ws == 0 && a.W % a.ws == 0 && a.shift >= 0 && a.shift < a.ws && a.ldx >= a.Cp &&
                   a.y_mode >= SR_Y_ROLL && a.y_mode <= SR_Y_STRIP_LAST,
               "sr_swin_block: bad geometry");
    SR_REQUIRE((long long)a.B * a.H * a.W < (1ll << 31), "sr_swin_block: more than 2^31 tokens");
    SR_REQUIRE(a.ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0, "sr_swin_block: the stream rows must be 16-byte aligned (ldx a multiple of 4, x 16-byte aligned)");
    SwinBlock3Dev dv;
    dv.a = a;
    const int nwx = a.W / a.ws, nwy = a.H / a.ws;
    dv.div_nw = make_fastdiv((uint32_t)(nwx * nwy));
    dv.div_nwx = make_fastdiv((uint32_t)nwx);
    dv.nwin = a.B * nwx * nwy;
    // grid: one workgroup per window (default: the hardware dispatcher backfills, which measured 2 % FASTER than the persistent form on a 2048 x 2048 image -- 66 k windows --
    // and equal everywhere else); max_workgroups > 0: at most that many persistent workgroups, each walking windows b, b + grid, ... with the next window's rows
    // prefetched; -2: as many as the device holds at once (three per CU; one for split operands)
    int grid = dv.nwin;
    if (a.max_workgroups > 0)
        grid = a.max_workgroups < grid ? a.max_workgroups : grid;
    else if (a.max_workgroups == -2) {
        const int slots = resident_workgroups(cdt == SR_BF16X3 ? 1 : 3);
        grid = slots < grid ? slots : grid;
    }
    if (cdt == SR_BF16X3) return launch_swin_block3<bf3>(dv, grid, reinterpret_cast<hipStream_t>(stream));
    return launch_swin_block3<bf16>(dv, grid, reinterpret_cast<hipStream_t>(stream));
}
