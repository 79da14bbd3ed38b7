// One launch = one whole SwinTransformerBlock (swinir.py:146-174; attention half alone: hat.py:164-192):
//     x1  = x + proj( softmax(q k^T + bias + mask) v ),   q,k,v = qkv( LayerNorm1(x) )
//     out = x1 + fc2( GELU( fc1( LayerNorm2(x1) ) ) )
// with window_partition, torch.roll and window_reverse folded into addressing.
//
// Round-2 structure (replaces the 12-wave / 2-window kernel of sr_swin_attn.hip, which stays as `SR_SWIN_BLOCK=v1`):
//   * ONE window (64 tokens) per workgroup, 4 waves = one wave per SIMD, <= 168 VGPRs and 50 KiB of LDS, so THREE
//     independent workgroups share a CU.  All 648 windows of the bench shape are resident at once (768 slots): no second
//     residency round, and the three waves on a SIMD belong to three different windows in three different stages, so one
//     window's VALU stages (LayerNorm, softmax, GELU) overlap another's MFMA stages instead of all waves of a CU doing the
//     same stage in lock step.
//   * the six heads run as three passes of two heads; a pass = 4 attention "atoms" (head, query half), one per wave:
//       GEMM   q for the wave's own atom (stays in registers: the accumulators are the next MFMA's operand) and ONE
//              head's k (waves 0,1) or v (waves 2,3) for all 64 tokens -> 8 KiB K image / 8 KiB V^T image in LDS, stored
//              in the permuted d / key order that the accumulator-as-operand trick produces on the other side
//       attn   S^T = K Q^T (+bias, +mask), softmax in registers, O^T = V^T P^T -> 8 KiB O chunk image
//       proj   x1[:, 48w..48w+48) += O_chunk @ Wproj[64 input channels of this pass]   (accumulators live across passes)
//   * MLP: LayerNorm2 over the four waves' column slices, fc1 / GELU / fc2 in two hidden halves of 192 columns so the
//     hidden image (24 KiB) reuses the K / V / O region.
//   * x is read twice (token-row layout for LayerNorm1: statistics stay inside one wave, no barrier; accumulator layout as
//     the residual) -- the second read hits L1 / L2 -- and written once.
#include "sr_common.h"
#include "sr_host.h"

namespace {

__device__ unsigned long long sr_dbg_swb[64];
// one lane of wave STAMP_WAVE of one workgroup; the stamp statement carries its own lgkmcnt wait (s_memtime returns out of order)
#ifdef SR_STAMPS
#ifndef SR_STAMP_WAVE
#define SR_STAMP_WAVE 0
#endif
#define STAMP(i)                                                                                             \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (blockIdx.x == 7 && threadIdx.x == 64 * SR_STAMP_WAVE) sr_dbg_swb[i] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

struct SwinBlockDev {
    SrSwinAttn a;
    FastDiv div_nw, div_nwx;  // windows per image, windows per row
};

constexpr int NTOK = 64, KC = 6, WS = 8;
constexpr int PAD_D = 30;  // first pad channel of a head (hd = 30 real + 2 pad): carries the softmax denominator
static_assert(PAD_D >= 16 + 12 && PAD_D < 32, "the denominator must sit in the fourth 16-lane row of the second d tile");
constexpr int LDS_A = 24 * 64 * 16;         // LayerNorm image [24 k-groups][64 tokens] of 16-B cells
constexpr int LDS_K = 2 * 4 * 64 * 16;      // K image   [2 heads][4 d-groups][64 keys]
constexpr int LDS_V = 2 * 2 * 4 * 32 * 16;  // V^T image [2 heads][2 key steps][4 key groups][32 d]
constexpr int LDS_O = 8 * 64 * 16;          // O chunk   [8 k-groups][64 tokens]
constexpr int LDS_RED = 64 * 4 * 2 * 4;     // LayerNorm2 partial sums [64 tokens][4 waves][2]
constexpr int LDS_TOTAL = LDS_A + LDS_K + LDS_V + LDS_O + LDS_RED;
static_assert(LDS_K + LDS_V + LDS_O == LDS_A, "the hidden-half image reuses the K / V / O region");
static_assert(3 * LDS_TOTAL <= 160 * 1024, "three workgroups per CU");

SR_DEV Frag<bf16> pack2(const f32x4& lo, const f32x4& hi) {
    Frag<bf16> f;
    f.v[0] = (bf16)lo[0]; f.v[1] = (bf16)lo[1]; f.v[2] = (bf16)lo[2]; f.v[3] = (bf16)lo[3];
    f.v[4] = (bf16)hi[0]; f.v[5] = (bf16)hi[1]; f.v[6] = (bf16)hi[2]; f.v[7] = (bf16)hi[3];
    return f;
}
SR_DEV bf16x4 cvt4(const f32x4& v) {
    bf16x4 r;
    r[0] = (bf16)v[0]; r[1] = (bf16)v[1]; r[2] = (bf16)v[2]; r[3] = (bf16)v[3];
    return r;
}
// 8-byte half of a 16-B LDS cell
SR_DEV void st_half(Frag<bf16>* cell, int half, const bf16x4& v) { *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(cell) + half * 8) = v; }

// Timing-only experiment builds (tools/exp_variants.sh; never shipped): SR_EXP_W0 = every weight fragment load reads chunk 0 of its
// stream (L1-resident: no L2 weight stream), SR_EXP_NOTRANS = no exp / GELU transcendentals, SR_EXP_NOBAR = no workgroup barriers.
#ifdef SR_EXP_W0
#define WCHUNK(c) 0
#else
#define WCHUNK(c) (c)
#endif
#ifdef SR_EXP_NOBAR
#define BLOCK_SYNC() __builtin_amdgcn_sched_barrier(0)
#else
#define BLOCK_SYNC() __syncthreads()
#endif

// K-loop with the B (weight) fragments streamed L2 -> registers DIST chunks ahead and the A (activation) fragments of the next
// chunk read from LDS while the current chunk's MFMAs run.  loadb(c, dst) fetches the NB weight fragments of K-chunk c,
// loada(c, dst) the NA activation fragments, compute(c, b, a) consumes them.  Everything is unrolled: ring slots are registers.
template <int NB, int NKC, int DIST>
struct WRing {
    Frag<bf16> r[DIST + 1][NB];
    template <typename LoadB>
    SR_DEV void prologue(LoadB&& loadb) {
#pragma unroll
        for (int c = 0; c < DIST && c < NKC; ++c) loadb(c, r[c]);
    }
    // A fragments in two halves per chunk: half 1 of chunk c and half 0 of chunk c + 1 are read from LDS under the MFMAs of the
    // half before them (NA2 fragments live per half: half the registers of a whole-chunk double buffer)
    template <int NA2, typename LoadB, typename LoadA, typename Compute>
    SR_DEV void run(LoadB&& loadb, LoadA&& loada, Compute&& compute) {
        Frag<bf16> a0[NA2], a1[NA2];
        loada(0, 0, a0);
#pragma unroll
        for (int c = 0; c < NKC; ++c) {
            if (c + DIST < NKC) loadb(c + DIST, r[(c + DIST) % (DIST + 1)]);
            loada(c, 1, a1);
            compute(c, 0, r[c % (DIST + 1)], a0);
            if (c + 1 < NKC) loada(c + 1, 0, a0);
            compute(c, 1, r[c % (DIST + 1)], a1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// value of lane (l & 15) + 48 (the fourth 16-lane row) in every lane: two half-exchanges, no LDS crossbar
SR_DEV float bcast_row3(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));  // b = [rows 2,3 | rows 2,3]
    float c = b, d = b;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));  // d = row 3 everywhere
    return d;
}
// v[g] summed over the four 16-lane rows, result for index g delivered to row g (a reduce-scatter: 3 swaps + 3 adds for 4 values)
SR_DEV float rows_reduce_scatter4(float v0, float v1, float v2, float v3) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v0), "+v"(v2));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v1), "+v"(v3));
    float u02 = v0 + v2, u13 = v1 + v3;  // rows 0,1: partials of index 0 (1); rows 2,3: partials of index 2 (3)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u02), "+v"(u13));
    return u02 + u13;
}

// Specialised for Cp = 192, heads = 6, hd_p = 32, ws = 8, Hp = 384 (SwinIR / HAT-w8 default geometry).
template <bool MLP>
__global__ __launch_bounds__(256, 3) void sr_swin_block_kernel(SwinBlockDev dv) {
    const SrSwinAttn& a = dv.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<bf16>* Aimg = reinterpret_cast<Frag<bf16>*>(smem);
    Frag<bf16>* Kimg = Aimg + 24 * 64;
    Frag<bf16>* Vimg = Kimg + 2 * 4 * 64;
    Frag<bf16>* Oimg = Vimg + 2 * 2 * 4 * 32;
    float* red = reinterpret_cast<float*>(Oimg + 8 * 64);
    Frag<bf16>* Himg = Kimg;  // [24][64] hidden half (MLP stage: K / V / O are dead)

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane0 = threadIdx.x & 63;
    int lane = lane0, ar = lane & 15, ag = lane >> 4;
    // Per-lane LDS / global offsets are cheap to recompute; hipcc would otherwise hoist every one of them to kernel entry and
    // keep ~40 address registers alive (and spilled) across the whole kernel.  relane() makes the lane id opaque again.
    auto relane = [&]() {
        lane = lane0;
        asm volatile("" : "+v"(lane));
        ar = lane & 15;
        ag = lane >> 4;
    };

    // ---- window geometry (one window per workgroup)
    uint32_t bimg, win, wy, wx;
    dv.div_nw.divmod((uint32_t)blockIdx.x, bimg, win);
    dv.div_nwx.divmod(win, wy, wx);
    const int shift_y = a.y_mode == SR_Y_ROLL ? a.shift : 0;  // strips arrive already rolled in y (halo exchange)
    auto pixel_row = [&](int t) {  // image-order row of window token t (roll + partition as one gather)
        int y = wy * WS + (t >> 3) + shift_y;
        int x = wx * WS + (t & 7) + a.shift;
        if (y >= a.H) y -= a.H;
        if (x >= a.W) x -= a.W;
        return ((int)bimg * a.H + y) * a.W + x;
    };

    // weight fragments: wave-uniform fragment base (SGPRs) + lane index (one shared VGPR offset), never a per-stream 64-bit VGPR pointer
    const Frag<bf16>* Wq = reinterpret_cast<const Frag<bf16>*>(a.wqkv);
    const Frag<bf16>* Wp = reinterpret_cast<const Frag<bf16>*>(a.wproj);
    const int hh = w >> 1, half = w & 1;  // attention atom of this wave inside a pass: (head 2p + hh, queries [32 half, 32 half + 32))
    const bool krole = w < 2;             // waves 0,1 project k of head 2p + w, waves 2,3 project v of head 2p + w - 2
    const int kvh = w & 1;

    // weight fragments of K-chunk c for pass p: the two 16-column tiles of one head's q, k or v slice
    auto load_q = [&](int p, int c, Frag<bf16> (&b)[2]) {
        const Frag<bf16>* f = Wq + ((size_t)((2 * p + hh) * 2) * KC + WCHUNK(c)) * 64;
        b[0] = f[lane];
        b[1] = f[KC * 64 + lane];
    };
    auto load_kv = [&](int p, int c, Frag<bf16> (&b)[2]) {
        const Frag<bf16>* f = Wq + ((size_t)((krole ? 12 : 24) + (2 * p + kvh) * 2) * KC + WCHUNK(c)) * 64;
        b[0] = f[lane];
        b[1] = f[KC * 64 + lane];
    };

    STAMP(0);
    // ---- S0: x in token-row layout (LayerNorm1) and in accumulator layout (residual), first weight chunks
    f32x4 x1[4][3];  // [m][n]: token 16 m + ar, columns 48 w + 16 n + 4 ag .. +3   (the residual, then x1, then the output)
    WRing<2, KC, 2> ring_kv;
    WRing<2, KC, 1> ring_q;
    {
        f32x4 xr[12];  // token 16 w + ar, channels 16 j + 4 ag .. +3
        const float* xrow = a.x + (size_t)pixel_row(w * 16 + ar) * a.ldx + ag * 4;
#pragma unroll
        for (int j = 0; j < 12; ++j) xr[j] = load4(xrow + j * 16);
        __builtin_amdgcn_sched_barrier(0);
        ring_kv.prologue([&](int c, Frag<bf16> (&b)[2]) { load_kv(0, c, b); });
        __builtin_amdgcn_sched_barrier(0);
        float s1 = 0.f, s2 = 0.f;
        {
            f32x4 p1 = (f32x4)(0.f), p2 = (f32x4)(0.f);  // four independent chains per statistic (a lone wave is latency-bound here)
#pragma unroll
            for (int j = 0; j < 12; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p1[r] += xr[j][r];
                    p2[r] = __builtin_fmaf(xr[j][r], xr[j][r], p2[r]);
                }
            s1 = (p1[0] + p1[1]) + (p1[2] + p1[3]);
            s2 = (p2[0] + p2[1]) + (p2[2] + p2[3]);
        }
        s1 = wave_sum_xor(s1, 16);
        s1 = wave_sum_xor(s1, 32);
        s2 = wave_sum_xor(s2, 16);
        s2 = wave_sum_xor(s2, 32);
        const float inv = 1.0f / (float)a.C;
        const float mean = s1 * inv;
        const float rstd = rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + a.eps);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            f32x4 nv;  // gamma / beta are folded into wqkv / bqkv
#pragma unroll
            for (int r = 0; r < 4; ++r) nv[r] = __builtin_fmaf(xr[j][r], rstd, nmr);
            st_half(Aimg + (2 * j + (ag >> 1)) * NTOK + w * 16 + ar, ag & 1, cvt4(nv));
        }
    }
    STAMP(1);
    BLOCK_SYNC();
    STAMP(2);

    // ---- shift mask terms that do not depend on the pass (common.py:250-274 from window coordinates)
    const bool last_row = a.y_mode != SR_Y_STRIP && (int)wy == a.H / WS - 1, last_col = (int)wx == a.W / WS - 1;
    const bool masked = a.shift > 0 && (last_row || last_col);

    // ---- three passes of two heads
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        relane();
        // -- GEMM 1: k or v of one head for all 64 tokens (4 token tiles x 2 d tiles) -> K / V^T image
        {
            f32x4 kv[4][2];
#pragma unroll
            for (int m = 0; m < 4; ++m) {  // the k bias cancels in the softmax, the v bias is folded into bproj
                kv[m][0] = (f32x4)(0.0f);
                // v: feature 16 + ar of the head; d = 30 is the first pad channel (hd = 30, zero weight rows): a constant 1 there makes
                // row 30 of O^T = V^T P^T the softmax denominator (sum of the bf16 probabilities the MFMA actually multiplies)
                kv[m][1] = (f32x4)((!krole && ar == PAD_D - 16) ? 1.0f : 0.0f);
            }
            auto loadb = [&](int c, Frag<bf16> (&b)[2]) { load_kv(p, c, b); };
            auto loada = [&](int c, int h, Frag<bf16> (&av)[2]) {
                const Frag<bf16>* arow = Aimg + (c * 4 + ag) * NTOK + h * 32 + ar;
                av[0] = arow[0];
                av[1] = arow[16];
            };
            if (krole) {
                ring_kv.run<2>(loadb, loada, [&](int c, int h, Frag<bf16> (&b)[2], Frag<bf16> (&av)[2]) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        mma(b[0], av[m], kv[2 * h + m][0]);  // swapped: lane = token, registers = 4 features
                        mma(b[1], av[m], kv[2 * h + m][1]);
                    }
                    if (c == 4 && h == 1) ring_q.prologue([&](int c2, Frag<bf16> (&b2)[2]) { load_q(p, c2, b2); });
                });
            } else {
                ring_kv.run<2>(loadb, loada, [&](int c, int h, Frag<bf16> (&b)[2], Frag<bf16> (&av)[2]) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        mma(av[m], b[0], kv[2 * h + m][0]);  // un-swapped: lane = feature, registers = 4 tokens
                        mma(av[m], b[1], kv[2 * h + m][1]);
                    }
                    if (c == 4 && h == 1) ring_q.prologue([&](int c2, Frag<bf16> (&b2)[2]) { load_q(p, c2, b2); });
                });
            }
            STAMP(3 + 8 * p);
            if (p > 0) BLOCK_SYNC();  // every wave is done with the previous pass's K / V (attention) and O (proj)
            if (krole) {
                Frag<bf16>* kb = Kimg + (kvh * 4 + ag) * NTOK + ar;  // cell [d-group ag][key]: d = {4 ag + r} then {16 + 4 ag + r}
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) st_half(kb + m * 16, n, cvt4(kv[m][n]));
            } else {
                Frag<bf16>* vb = Vimg + (kvh * 2 * 4 + ag) * 32 + ar;  // cell [step][key group ag][d]: keys {32 s + 4 ag + r} then {32 s + 16 + 4 ag + r}
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) st_half(vb + (m >> 1) * 4 * 32 + n * 16, m & 1, cvt4(kv[m][n]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(4 + 8 * p);
        // -- GEMM 2: q of the wave's own atom (2 query tiles x 2 d tiles); stays in registers as the S^T = K Q^T operand
        Frag<bf16> qf[2];
        {
            f32x4 qa[2][2];
            const f32x4 b0 = load4(a.bqkv + (2 * p + hh) * 32 + ag * 4), b1 = load4(a.bqkv + (2 * p + hh) * 32 + 16 + ag * 4);
#pragma unroll
            for (int m = 0; m < 2; ++m) {  // the bias is the C operand of the first MFMA
                qa[m][0] = b0;
                qa[m][1] = b1;
            }
            ring_q.run<1>([&](int c, Frag<bf16> (&b)[2]) { load_q(p, c, b); },
                          [&](int c, int h, Frag<bf16> (&aq)[1]) { aq[0] = Aimg[(c * 4 + ag) * NTOK + (2 * half + h) * 16 + ar]; },
                          [&](int c, int h, Frag<bf16> (&b)[2], Frag<bf16> (&aq)[1]) {
                              mma(b[0], aq[0], qa[h][0]);
                              mma(b[1], aq[0], qa[h][1]);
                          });
            qf[0] = pack2(qa[0][0], qa[0][1]);
            qf[1] = pack2(qa[1][0], qa[1][1]);
        }
        STAMP(5 + 8 * p);
        f32x4 s[2][4];
        {
            const f32x4* bias = reinterpret_cast<const f32x4*>(a.bias) + ((size_t)(2 * p + hh) * 16 + 2 * half * 4) * 64;  // [h][qt][kt][lane]
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) s[qt][kt] = bias[(qt * 4 + kt) * 64 + lane];
        }
        if (p == 0) {  // the residual in accumulator layout (second read of the x tile: L1 / L2 hits), needed by the first proj partial
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float* xm = a.x + (size_t)pixel_row(m * 16 + ar) * a.ldx + w * 48 + ag * 4;
#pragma unroll
                for (int n = 0; n < 3; ++n) x1[m][n] = load4(xm + n * 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(6 + 8 * p);
        BLOCK_SYNC();  // K / V of both heads are in LDS
        STAMP(7 + 8 * p);
        relane();

        // -- attention for the atom: S^T = K Q^T (+bias as the C operand), mask, softmax over keys, O^T = V^T P^T
        {
            const Frag<bf16>* kb = Kimg + (hh * 4 + ag) * NTOK + ar;
            Frag<bf16> kf[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) kf[kt] = kb[kt * 16];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) mma(kf[kt], qf[qt], s[qt][kt]);
        }
        const Frag<bf16>* vb = Vimg + (hh * 2 * 4 + ag) * 32 + ar;
        Frag<bf16> vf[2][2];  // [d tile][key step]
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int st = 0; st < 2; ++st) vf[dt][st] = vb[st * 4 * 32 + dt * 16];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (masked) {
                f32x4 colneg;  // -100 where the key's column half differs from the query's (recomputed per pass: 4 registers not kept alive)
                const bool qcol = last_col && (ar & 7) >= WS - a.shift;
#pragma unroll
                for (int r = 0; r < 4; ++r) colneg[r] = (last_col && 4 * (ag & 1) + r >= WS - a.shift) != qcol ? -100.0f : 0.0f;
                // label(q) != label(k)  <=>  the row halves differ (last window row only) or the column halves differ (last
                // window column only); key row = 2 kt + (ag >> 1), key column = 4 (ag & 1) + r, query row = 4 half + 2 qt + (ar >> 3)
                const bool qrow = last_row && 4 * half + 2 * qt + (ar >> 3) >= WS - a.shift;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const float rowneg = (last_row && 2 * kt + (ag >> 1) >= WS - a.shift) != qrow ? -100.0f : 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[qt][kt][r] += fminf(rowneg, colneg[r]);
                }
            }
#ifndef SR_EXP_NOVALU
            f32x4 tm;  // per key tile maxima: independent chains
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) tm[kt] = fmaxf(fmaxf(s[qt][kt][0], s[qt][kt][1]), fmaxf(s[qt][kt][2], s[qt][kt][3]));
            float mx = fmaxf(fmaxf(tm[0], tm[1]), fmaxf(tm[2], tm[3]));
            mx = wave_max_xor(mx, 16);
            mx = wave_max_xor(mx, 32);
            const float nmx = -mx * 1.4426950408889634f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#ifdef SR_EXP_NOTRANS
                    s[qt][kt][r] = __builtin_fmaf(s[qt][kt][r], 1.4426950408889634f, nmx);
#else
                    s[qt][kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][kt][r], 1.4426950408889634f, nmx));  // exp(s - max)
#endif
                }
#endif
            const Frag<bf16> p0 = pack2(s[qt][0], s[qt][1]), p1 = pack2(s[qt][2], s[qt][3]);
            f32x4 o0 = (f32x4)(0.0f), o1 = (f32x4)(0.0f);
            mma(vf[0][0], p0, o0);
            mma(vf[0][1], p1, o0);
            mma(vf[1][0], p0, o1);
            mma(vf[1][1], p1, o1);
            // row d = 30 of O^T (lanes 48..63, register 2 of the second d tile) is sum_k P[q][k]: the softmax denominator of query ar
            const float inv_sum = __builtin_amdgcn_rcpf(bcast_row3(o1[PAD_D & 3]));
            o0 *= inv_sum;
            o1 *= inv_sum;  // (the pad channels 30, 31 of O meet zero rows of Wproj)
            Frag<bf16>* ob = Oimg + (hh * 4 + (ag >> 1)) * NTOK + half * 32 + qt * 16 + ar;
            st_half(ob, ag & 1, cvt4(o0));
            st_half(ob + 2 * NTOK, ag & 1, cvt4(o1));
        }
        __builtin_amdgcn_sched_barrier(0);
        // the proj weights of this pass (3 n-tiles x 2 K-chunks) and the next pass's first chunks travel across the barrier
        Frag<bf16> wpj[2][3];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int n = 0; n < 3; ++n) wpj[c][n] = (Wp + ((size_t)(3 * w + n) * KC + WCHUNK(2 * p + c)) * 64)[lane];
        __builtin_amdgcn_sched_barrier(0);
        STAMP(8 + 8 * p);
        BLOCK_SYNC();  // the O chunk (64 tokens x 64 channels) is complete
        STAMP(9 + 8 * p);

        // -- proj partial: x1 columns [48 w, 48 w + 48) += O_chunk @ Wproj[:, 64 p .. 64 p + 64)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c == 1 && p < 2) ring_kv.prologue([&](int c2, Frag<bf16> (&b)[2]) { load_kv(p + 1, c2, b); });  // next pass's first weight chunks
            const Frag<bf16>* orow = Oimg + (c * 4 + ag) * NTOK + ar;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const Frag<bf16> ov = orow[m * 16];
#pragma unroll
                for (int n = 0; n < 3; ++n) mma(wpj[c][n], ov, x1[m][n]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(10 + 8 * p);
    }

    // ---- x1 = x + proj(...) + bproj
    relane();
    {
        const float* bp = a.bproj + w * 48 + ag * 4;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const f32x4 b = load4(bp + n * 16);
#pragma unroll
            for (int m = 0; m < 4; ++m) x1[m][n] += b;
        }
    }

    if constexpr (MLP) {
        const Frag<bf16>* W1 = reinterpret_cast<const Frag<bf16>*>(a.w1p);
        const Frag<bf16>* W2 = reinterpret_cast<const Frag<bf16>*>(a.w2p);
        auto load_fc1 = [&](int hf, int c, Frag<bf16> (&b)[3]) {
#pragma unroll
            for (int n = 0; n < 3; ++n) b[n] = (W1 + ((size_t)(12 * hf + 3 * w + n) * KC + WCHUNK(c)) * 64)[lane];
        };
        auto load_fc2 = [&](int hf, int c, Frag<bf16> (&b)[3]) {
#pragma unroll
            for (int n = 0; n < 3; ++n) b[n] = (W2 + ((size_t)(3 * w + n) * (2 * KC) + WCHUNK(6 * hf + c)) * 64)[lane];
        };
        WRing<3, KC, 2> ring1, ring2;
        ring1.prologue([&](int c, Frag<bf16> (&b)[3]) { load_fc1(0, c, b); });
        // ---- LayerNorm2 statistics of x1 across the four waves' column slices
        {
            float q1[4], q2[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 t1 = x1[m][0] + x1[m][1] + x1[m][2];  // pad channels of x1 are exactly 0 (zero weight rows / bias / residual pad)
                f32x4 t2 = x1[m][0] * x1[m][0];
#pragma unroll
                for (int r = 0; r < 4; ++r) t2[r] = __builtin_fmaf(x1[m][2][r], x1[m][2][r], __builtin_fmaf(x1[m][1][r], x1[m][1][r], t2[r]));
                q1[m] = (t1[0] + t1[1]) + (t1[2] + t1[3]);
                q2[m] = (t2[0] + t2[1]) + (t2[2] + t2[3]);
            }
            // lane row ag ends up with the sums of token tile m = ag: one 8-byte store per lane, no predication
            const float s1 = rows_reduce_scatter4(q1[0], q1[1], q1[2], q1[3]);
            const float s2 = rows_reduce_scatter4(q2[0], q2[1], q2[2], q2[3]);
            *reinterpret_cast<float2*>(red + ((ag * 16 + ar) * 4 + w) * 2) = make_float2(s1, s2);
        }
        STAMP(27);
        BLOCK_SYNC();  // partials visible; every wave has also finished its last proj partial (O image is dead)
        {
            const float inv = 1.0f / (float)a.C;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 pa = *reinterpret_cast<const f32x4*>(red + (m * 16 + ar) * 8), pb = *reinterpret_cast<const f32x4*>(red + (m * 16 + ar) * 8 + 4);
                const float mean = (pa[0] + pa[2] + pb[0] + pb[2]) * inv;
                const float rstd = rsqrtf(fmaxf((pa[1] + pa[3] + pb[1] + pb[3]) * inv - mean * mean, 0.f) + a.eps);
                const float nmr = -mean * rstd;
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    f32x4 nv;  // gamma2 / beta2 are folded into w1p / b1
#pragma unroll
                    for (int r = 0; r < 4; ++r) nv[r] = __builtin_fmaf(x1[m][n][r], rstd, nmr);
                    st_half(Aimg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, cvt4(nv));
                }
            }
        }
        {
            const float* b2 = a.b2 + w * 48 + ag * 4;
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const f32x4 b = load4(b2 + n * 16);
#pragma unroll
                for (int m = 0; m < 4; ++m) x1[m][n] += b;
            }
        }
        STAMP(28);
        BLOCK_SYNC();
        STAMP(29);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            relane();
            // fc1 for hidden columns [192 hf + 48 w, +48) -> GELU -> hidden-half image
            f32x4 acc[4][3];
            {
                const float* b1 = a.b1 + hf * 192 + w * 48 + ag * 4;
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    const f32x4 b = load4(b1 + n * 16);
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m][n] = b;
                }
            }
            ring1.run<2>([&](int c, Frag<bf16> (&b)[3]) { load_fc1(hf, c, b); },
                         [&](int c, int h, Frag<bf16> (&av)[2]) {
                             const Frag<bf16>* arow = Aimg + (c * 4 + ag) * NTOK + h * 32 + ar;
                             av[0] = arow[0];
                             av[1] = arow[16];
                         },
                         [&](int c, int h, Frag<bf16> (&b)[3], Frag<bf16> (&av)[2]) {
#pragma unroll
                             for (int m = 0; m < 2; ++m)
#pragma unroll
                                 for (int n = 0; n < 3; ++n) mma(b[n], av[m], acc[2 * h + m][n]);
                         });
            STAMP(30 + 5 * hf);
            ring2.prologue([&](int c, Frag<bf16> (&b)[3]) { load_fc2(hf, c, b); });
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
#else
                    for (int r = 0; r < 4; ++r) g[r] = gelu_bf16(acc[m][n][r]);
#endif
                    st_half(Himg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, cvt4(g));
                }
            if (hf == 0) ring1.prologue([&](int c, Frag<bf16> (&b)[3]) { load_fc1(1, c, b); });
            __builtin_amdgcn_sched_barrier(0);
            STAMP(31 + 5 * hf);
            BLOCK_SYNC();
            STAMP(32 + 5 * hf);
            // fc2 partial on top of x1: K = the 192 hidden columns of this half
            ring2.run<2>([&](int c, Frag<bf16> (&b)[3]) { load_fc2(hf, c, b); },
                         [&](int c, int h, Frag<bf16> (&hv)[2]) {
                             const Frag<bf16>* hrow = Himg + (c * 4 + ag) * NTOK + h * 32 + ar;
                             hv[0] = hrow[0];
                             hv[1] = hrow[16];
                         },
                         [&](int c, int h, Frag<bf16> (&b)[3], Frag<bf16> (&hv)[2]) {
#pragma unroll
                             for (int m = 0; m < 2; ++m)
#pragma unroll
                                 for (int n = 0; n < 3; ++n) mma(b[n], hv[m], x1[2 * h + m][n]);
                         });
            STAMP(33 + 5 * hf);
        }
    }

    STAMP(40);
    // ---- store (window_reverse + roll back folded into the address)
    relane();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        float* dst = a.out + (size_t)pixel_row(m * 16 + ar) * a.ldx + w * 48 + ag * 4;
#pragma unroll
        for (int n = 0; n < 3; ++n) store4(dst + n * 16, x1[m][n]);
    }
    STAMP(41);
}

}  // namespace

extern "C" int sr_debug_swb_stamps(unsigned long long* host64) {
    return hipMemcpyFromSymbol(host64, HIP_SYMBOL(sr_dbg_swb), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}

bool sr_swin_block_v2_enabled() {  // read per call (a getenv is ~100 ns against a ~30 us kernel): lets one process A/B the two kernels
    const char* e = getenv("SR_SWIN_BLOCK");
    return !(e && e[0] == 'v' && e[1] == '1');
}

int sr_swin_block_v2(const SrSwinAttn& a, hipStream_t st) {
    SwinBlockDev dv;
    dv.a = a;
    const int nwx = a.W / a.ws, nwy = a.H / a.ws;
    dv.div_nw = make_fastdiv((uint32_t)(nwx * nwy));
    dv.div_nwx = make_fastdiv((uint32_t)nwx);
    static SrDeviceOnce attr_once;
    {
        const hipError_t e = sr_once_per_device(attr_once, [&] {
            hipError_t e2 = sr_allow_lds(sr_swin_block_kernel<false>, LDS_TOTAL);
            return e2 != hipSuccess ? e2 : sr_allow_lds(sr_swin_block_kernel<true>, LDS_TOTAL);
        });
        SR_REQUIRE(e == hipSuccess, "sr_swin_attn_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const dim3 grid(a.B * nwx * nwy);
    if (a.w1p)
        hipLaunchKernelGGL(sr_swin_block_kernel<true>, grid, dim3(256), LDS_TOTAL, st, dv);
    else
        hipLaunchKernelGGL(sr_swin_block_kernel<false>, grid, dim3(256), LDS_TOTAL, st, dv);
    SR_CHECK_LAUNCH("sr_swin_attn_fused");
    return SR_OK;
}
