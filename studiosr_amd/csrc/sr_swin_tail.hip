// Everything of a window-attention transformer block that follows the attention kernel, in ONE launch (C ABI v6 sr_swin_tail):
//     x1  = x + proj(O) + b_proj  [ + y * gate[image] ]        (hat.py:172-192: attention output, shortcut, conv_scale * CAB(x))
//     out = x1 + fc2( GELU( fc1( LayerNorm2(x1) ) ) )          (hat.py:194; swinir.py:172-174; common.py:173-195)
// for the geometries whose attention does not fit the one-kernel block (HAT: 16 x 16 windows, overlapping cross attention): O is the
// [tokens, heads * hd_p] output of sr_window_attention / sr_oca in WINDOW order, x / y / out are the image-order stream (window_reverse +
// roll back = the row gather of the block kernel).  Replaces the projection GEMM (sr_gemm, 25 us at HAT's 4 x 64 x 64 tokens) and the
// MLP kernel (16.5 us) with one pass of the stream-form kernel's own stages (sr_swin_stream.h):
//   * one workgroup = 64 consecutive window-order tokens (a quarter of a 16 x 16 window), 4 waves, wave w owns output channels [48 w, +48);
//   * O reaches LDS by coalesced LDS-DMA (no registers) as a token-major image with a 400-B row stride: conflict-free fragment reads
//     (a gathering DMA straight into the K-group-major layout -- lane = token -- took 500 cycles per instruction to issue);
//   * projection = 6 uniform steps, MLP = 2 x (6 fc1 + 6 fc2) steps from ONE 30-slot weight stream two slots ahead in registers;
//     fc1 / fc2 biases ride on the constant-one pad channels exactly as in sr_swin_block (same slots 24..47 of that stream);
//   * x1 never leaves the registers; the result leaves through the LDS row tile as full 768-B rows.
#include <type_traits>
#include <cstdlib>
#include "sr_swin_stream.h"
#include "sr_ca.h"

namespace {

constexpr int TAIL_SLOTS = 30;  // 6 proj, then per hidden half 6 fc1 + 6 fc2
constexpr int QKV_SLOTS = 18;   // optional: the next block's QKV projection (3 passes of two heads x 6 K-chunks)
constexpr int OROW = 384;       // bytes per attention-output row (6 heads x 32 features bf16)
constexpr int OSTRIDE = 400;    // its row stride in LDS
static_assert(NTOK * OSTRIDE == 25 * 1024 && NTOK * OSTRIDE <= CELLS_A * 16 + 1024, "25 one-KiB pieces; the image ends inside the (still unused) hidden-half region");
// HAT's launches are 256..1024 workgroups, i.e. one to four per CU with nothing else resident: weights four slots ahead (two workgroups
// per CU by registers) instead of sr_swin_block's two: -2.7 % on the x4 b4 forward, +-0 at b16 (SR_TAIL_DIST / SR_TAIL_WGS: A/B knobs)
#ifndef SR_TAIL_DIST
#define SR_TAIL_DIST 4
#endif
#ifndef SR_TAIL_WGS
#define SR_TAIL_WGS 2
#endif

#ifdef SR_STAMPS
__device__ unsigned long long sr_dbg_tail[32];
#define TSTAMP(i)                                                                                 \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (QKV && blockIdx.x == 7 && threadIdx.x == 0) sr_dbg_tail[i] = __builtin_amdgcn_s_memtime();   \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

struct SwinTailDev {
    SrSwinTail a;
    FastDiv div_parts_img, div_parts_win, div_nwx;  // 64-token parts per image, per window; windows per row
    int ws_log2, nw;                                // windows per image
    int pool_lds;                                   // the image's pool partials travel to LDS by DMA at kernel entry (see POOL_OFF)
};

// The channel-attention squeeze of the gated second residual (hat.py:25-38) reads the image's pool partials ([n_tiles][Cp] floats, 42 KB at 64 x 64) in every
// workgroup.  As loads inside ca_squeeze they queued behind every other load of the prologue and cost 9.4 k of the kernel's 56 k cycles; now they travel
// to LDS by DMA at kernel entry (no registers, in flight under the x / y loads), behind the O image -- the region up to the 80 KiB that two workgroups
// per CU leave each is free until LayerNorm2 -- with the squeeze scratch behind them.  Launches whose partials do not fit keep the loads.
constexpr int POOL_OFF = NTOK * OSTRIDE;
constexpr int TAIL_LDS_MAX = 80 * 1024;

// 16 B per lane from base + lane_off to LDS at lds_dst + 16 lane
SR_DEV void dma_gather16(const char* base, int lane_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(base), "s"(lds_dst)
                 : "memory");
}

// QKV = true: the kernel goes on with LayerNorm1 + QKV projection of the NEXT block on the same 64 tokens (sr_swin_qkv.hip's passes; weight
// slots 30..47 of the stream) and scatters q / k / v^T into that block's window order (its shift differs): one launch less on the critical
// chain tail -> qkv -> attention of a HAB, and the ring prefetches the QKV weights through the MLP.
// T = bf3 (compute type SR_BF16X3, precision "fp32x3" = what inference() runs): split operands as in sr_swin_block3.hip; O, y and the LayerNorm side output are
// fp32 tensors (the fp32 attention kernel / the split-operand convs write and read those), O passes through LDS as fp32 rows (the x-tile geometry) and is
// split into hi | lo fragments as it is read; 98 KiB of LDS, one workgroup per CU; no fused next-block QKV.
// MT = row tiles of 16 tokens per workgroup: 4 (64 tokens), or 2 (32 tokens; bf16) for launches that would leave the chip with one workgroup per CU or less --
// twice the workgroups of half the rows each: the kernel is a latency chain at one wave per SIMD (57 k cycles for 9 k cycles of MFMAs), and two half-size chains
// per CU overlap.  The LDS images keep their 64-token strides (half of each is unused).
template <typename T, bool QKV, int MT = 4>
__global__ __launch_bounds__(256, sizeof(Frag<T>) == 16 ? SR_TAIL_WGS : 1) void sr_swin_tail_kernel(SwinTailDev dv) {
    constexpr bool X3 = sizeof(Frag<T>) == 32;
    constexpr int NT_ = 16 * MT, RW = NT_ / 4;  // tokens per workgroup; rows per wave in the row-tile passes
    static_assert(!(X3 && QKV), "the fused next-block QKV exists for bf16 operands only");
    static_assert(MT == 4 || (MT == 2 && !X3), "32-token workgroups: bf16 instantiations only");
    const SrSwinTail& a = dv.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<T>* Aimg = reinterpret_cast<Frag<T>*>(smem);  // O image (token-major, 400-B rows) first, then the LayerNorm2 image [24 k-groups][64 tokens]
    Frag<T>* Himg = Aimg + CELLS_A;                    // hidden half [24][64]
    float* red = reinterpret_cast<float*>(smem + Lds<T>::RED_OFF);

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane0 = threadIdx.x & 63;
    int lane = lane0, ar = lane & 15, ag = lane >> 4;
    auto relane = [&]() {  // see sr_swin_block3.hip: keeps hipcc from hoisting (and spilling) every per-lane offset to kernel entry
        lane = lane0;
        asm volatile("" : "+v"(lane));
        ar = lane & 15;
        ag = lane >> 4;
    };

    // ---- geometry: workgroup = part `part` (64 tokens = 64 / ws window rows) of window `win` of image `bimg`
    uint32_t bimg, rem, win, part, wy, wx;
    dv.div_parts_img.divmod((uint32_t)blockIdx.x, bimg, rem);
    dv.div_parts_win.divmod(rem, win, part);
    dv.div_nwx.divmod(win, wy, wx);
    const int shift_y = a.y_mode == SR_Y_ROLL ? a.shift : 0;
    const int wsl = dv.ws_log2, wsm = a.ws - 1;
    auto pixel_row = [&](int t) {  // image-order row of token t of this workgroup (window_reverse + roll back as one gather)
        const int tw = (int)part * NT_ + t;
        int y = ((int)wy << wsl) + (tw >> wsl) + shift_y;
        int x = ((int)wx << wsl) + (tw & wsm) + a.shift;
        if (y >= a.H) y -= a.H;
        if (x >= a.W) x -= a.W;
        return ((int)bimg * a.H + y) * a.W + x;
    };

    TSTAMP(0);
    // ---- loads, all in flight together: O image (LDS-DMA), the first weight slots, x / y in the accumulator layout, gate and bias vectors
    {
        // O: 64 rows of 384 B, contiguous in memory -> token-major LDS image with a 400-B row stride (conflict-free fragment reads): 25
        // coalesced 1-KiB LDS-DMA pieces; cell q = 64 j + lane of piece j is 16-B column q % 25 of row q / 25 (column 24 = padding, masked)
        const unsigned img_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
        const char* orow = reinterpret_cast<const char*>(a.o) + (size_t)blockIdx.x * NT_ * OROW;
        if constexpr (X3) {  // fp32 rows of 768 B -> the x-tile geometry (XS stride), 16 rows per wave
            const float* of = reinterpret_cast<const float*>(a.o) + (size_t)blockIdx.x * NTOK * 192;
#pragma unroll
            for (int i = 0; i < 16; ++i) dma_row48(of + (size_t)(16 * w + i) * 192, __builtin_amdgcn_readfirstlane(img_lds + (16 * w + i) * XS), lane);
        } else
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int j = w + 4 * i;
            if (j * 64 < NT_ * 25) {
                const int q = j * 64 + lane;
                const int row = (q * 1311) >> 15;  // q / 25 for q < 1600
                const int cc = q - row * 25;
                if (cc < 24 && row < NT_) dma_gather16(orow, row * OROW + cc * 16, __builtin_amdgcn_readfirstlane(img_lds + j * 1024));
            }
        }
    }
    if (!X3 && dv.pool_lds) {
        const unsigned pool_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + POOL_OFF;
        const char* pbase = reinterpret_cast<const char*>(a.pool_partial + (size_t)bimg * a.ca_n_tiles * a.Cp);
        const int pbytes = a.ca_n_tiles * a.Cp * 4;
        for (int j = w; j * 1024 < pbytes; j += 4) {
            const int off = j * 1024 + lane * 16;
            if (off < pbytes) dma_gather16(pbase, off, __builtin_amdgcn_readfirstlane(pool_lds + j * 1024));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    TSTAMP(15);
    constexpr int NSL = QKV ? TAIL_SLOTS + QKV_SLOTS : TAIL_SLOTS;
    WStream<T, NSL, SR_TAIL_DIST> ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wstream), 0, NSL * 12 * 64 * (int)sizeof(Frag<T>), 0x00020000);
    ws.wave_frag = w * 3;
#pragma unroll
    for (int s0 = 0; s0 < SR_TAIL_DIST; ++s0) ws.load(s0, lane);
    __builtin_amdgcn_sched_barrier(0);
    TSTAMP(16);
    f32x4 x1[MT][3];  // [m][n]: token 16 m + ar, channels 48 w + 16 n + 4 ag .. +3
    f32x4 gt[3], bp[3];
    typename std::conditional<X3, f32x4, bf16x4>::type yv[MT][3];  // y: fp32 on the split-operand path
    const int ch0 = w * 48 + ag * 4;
    {
        int prow[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) prow[m] = pixel_row(m * 16 + ar);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) x1[m][n] = *reinterpret_cast<const f32x4*>(a.x + (size_t)prow[m] * a.ldx + ch0 + n * 16);
#pragma unroll
        for (int n = 0; n < 3; ++n) bp[n] = *reinterpret_cast<const f32x4*>(a.bproj + ch0 + n * 16);
        if (a.y) {
            if (!a.pool_partial) {
#pragma unroll
                for (int n = 0; n < 3; ++n) gt[n] = *reinterpret_cast<const f32x4*>(a.gate + (size_t)bimg * a.ld_gate + ch0 + n * 16);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    if constexpr (X3)
                        yv[m][n] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.y) + (size_t)prow[m] * a.ldy + ch0 + n * 16);
                    else
                        yv[m][n] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(a.y) + (size_t)prow[m] * a.ldy + ch0 + n * 16);
                }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    TSTAMP(26);
    SrChannelAttn ca;
    CaOps cops = {};
    const bool squeeze = a.y && a.pool_partial;
    const bool squeeze_lds = !X3 && squeeze && dv.pool_lds;
    if (squeeze) {
        // the channel-attention squeeze of this image (hat.py:25-38: mean -> 1x1 -> ReLU -> 1x1 -> sigmoid, times conv_scale), recomputed per
        // workgroup from the pool partials of the CAB convolution: replaces the sr_channel_gate launch at the end of the conv branch.
        ca.pool_partial = a.pool_partial; ca.w1 = a.ca_w1; ca.b1 = a.ca_b1; ca.w2 = a.ca_w2; ca.b2 = a.ca_b2;
        ca.B = a.B; ca.H = a.H; ca.W = a.W; ca.C = a.C; ca.C_p = a.Cp; ca.Cr = a.ca_Cr; ca.n_tiles = a.ca_n_tiles; ca.y_scale = a.y_scale;
        if (squeeze_lds) {
            if (a.ca_Cr == 6)  // its MLP operands fly with everything else; the sums wait for the DMA below
                ca_load_ops<180, 6>(ca, cops);
            else
                ca_load_ops(ca, cops);
        } else {
            // partials from global memory while the loads above fly.  Scratch behind the O image, in the hidden-half region (unused until the MLP).
            const float* gate = ca_squeeze(ca, (int)bimg, reinterpret_cast<float*>(smem + (X3 ? LDS_X : NTOK * OSTRIDE)));
#pragma unroll
            for (int n = 0; n < 3; ++n) gt[n] = *reinterpret_cast<const f32x4*>(gate + ch0 + n * 16);
        }
        TSTAMP(27);
    }
    // everything was issued together (one latency); the O image must be complete before the barrier, x and slot 0 are needed right behind it
    TSTAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BLOCK_SYNC();
    TSTAMP(2);
    if (squeeze_lds) {
        float* scratch = reinterpret_cast<float*>(smem + POOL_OFF + a.ca_n_tiles * a.Cp * 4);
        const float* gate;
        if (a.ca_Cr == 6) {  // HAT's geometry (180 channels in 192, squeeze_factor 30) as compile-time constants: a third of the instructions
            ca_slice_sums<192, 180>(ca, reinterpret_cast<const float*>(smem + POOL_OFF), ca_part(scratch, 192, 6));
            gate = ca_finish<192, 180, 6>(ca, cops, scratch);
        } else {
            ca_slice_sums(ca, reinterpret_cast<const float*>(smem + POOL_OFF), ca_part(scratch, a.Cp, a.ca_Cr));
            gate = ca_finish(ca, cops, scratch);
        }
#pragma unroll
        for (int n = 0; n < 3; ++n) gt[n] = *reinterpret_cast<const f32x4*>(gate + ch0 + n * 16);
        TSTAMP(28);
    }

    auto loada_img = [&](const Frag<T>* img) {
        return [&, img](int c, int h, Frag<T> (&av)[2]) {
            const Frag<T>* arow = img + (c * 4 + ag) * NTOK + h * 32 + ar;
            av[0] = arow[0];
            av[1] = arow[16];
        };
    };

    // ---- projection on top of the shortcut: x1 += O @ Wproj^T (K = 6 heads x 32 features, the pad features are 0 on both sides)
    auto loada_o = [&](int c, int h, Frag<T> (&av)[2]) {
        if constexpr (X3) {  // 8 fp32 features of one token -> hi | lo fragment
            const char* ob = smem + (h * 32 + ar) * XS + (c * 32 + ag * 8) * 4;
            av[0] = pack2<T>(*reinterpret_cast<const f32x4*>(ob), *reinterpret_cast<const f32x4*>(ob + 16));
            av[1] = pack2<T>(*reinterpret_cast<const f32x4*>(ob + 16 * XS), *reinterpret_cast<const f32x4*>(ob + 16 * XS + 16));
        } else {
            const char* ob = smem + (h * 32 + ar) * OSTRIDE + (c * 4 + ag) * 16;
            av[0] = *reinterpret_cast<const Frag<T>*>(ob);
            av[1] = *reinterpret_cast<const Frag<T>*>(ob + 16 * OSTRIDE);
        }
    };
    ws.template run<6, MT / 2>(0, lane, loada_o, [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&ov)[2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) mma(b[n], ov[m], x1[2 * h + m][n]);
    });
    __builtin_amdgcn_sched_barrier(0);
    TSTAMP(3);
    relane();
    {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                x1[m][n] += bp[n];
                if (a.y) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x1[m][n][r] = __builtin_fmaf((float)yv[m][n][r], gt[n][r], x1[m][n][r]);  // as sr_gemm's gated second residual
                }
            }
    }

    // ---- LayerNorm2 of x1 -> bf16 image over the O image (the barrier between its two halves orders the last O reads before the writes)
    const bool one_lane = (w == ONE_C / 48) && (ag == (ONE_C % 16) / 4);
    {
        {
            float q1[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};  // (MT = 2: row tiles 2, 3 do not exist -- their sums are zeros nobody reads)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 t1 = x1[m][0] + x1[m][1] + x1[m][2];
                f32x4 t2 = x1[m][0] * x1[m][0];
#pragma unroll
                for (int r = 0; r < 4; ++r) t2[r] = __builtin_fmaf(x1[m][2][r], x1[m][2][r], __builtin_fmaf(x1[m][1][r], x1[m][1][r], t2[r]));
                q1[m] = (t1[0] + t1[1]) + (t1[2] + t1[3]);
                q2[m] = (t2[0] + t2[1]) + (t2[2] + t2[3]);
            }
            const float s1 = rows_reduce_scatter4(q1[0], q1[1], q1[2], q1[3]);
            const float s2 = rows_reduce_scatter4(q2[0], q2[1], q2[2], q2[3]);
            if (MT == 4 || ag < MT) *reinterpret_cast<float2*>(red + ((ag * 16 + ar) * 4 + w) * 2) = make_float2(s1, s2);
        }
        BLOCK_SYNC();
        const float inv = 1.0f / (float)a.C;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
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
    }
    BLOCK_SYNC();
    TSTAMP(4);

    // the LayerNorm side output leaves from the row tile at the end; its affine is requested here, a whole MLP ahead (full rows, adjacent lanes on adjacent addresses), not from the accumulator layout (a quad of
    // lanes = four pixels = four cache lines per store: 12 such stores were 2 k cycles of vector-memory issue per wave)
    f32x4 n1g = (f32x4)(0.0f), n1b = (f32x4)(0.0f);
    if (a.n1) {
        const int c4 = (lane0 < 48 ? lane0 : 0) * 4;
        n1g = *reinterpret_cast<const f32x4*>(a.n1_gamma + c4);
        n1b = *reinterpret_cast<const f32x4*>(a.n1_beta + c4);
    }
    // ---- MLP in two hidden halves of 192 columns (sr_swin_block3.hip, same slots)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        relane();
        f32x4 acc[MT][3];
        ws.template run<6, MT / 2>(6 + 12 * hf, lane, loada_img(Aimg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&av)[2]) {
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
        TSTAMP(5 + 4 * hf);
        if (hf == 1) BLOCK_SYNC();  // fc2 of the first half has read the hidden image everywhere
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 g;
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = gelu_op<T>(acc[m][n][r]);
                if constexpr (X3) {  // hidden pad columns 360, 361 = the constant one of the fc2 bias rows (sr_swin_block3.hip)
                    if (hf == 1 && n == 1) {
                        const bool one_h = (w == 3) && (ag == 2);
                        g[0] = one_h ? 1.0f : g[0];
                        g[1] = one_h ? 1.0f : g[1];
                    }
                }
                st_half(Himg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, g);
            }
        __builtin_amdgcn_sched_barrier(0);
        TSTAMP(6 + 4 * hf);
        BLOCK_SYNC();
        TSTAMP(7 + 4 * hf);
        ws.template run<6, MT / 2>(12 + 12 * hf, lane, loada_img(Himg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&hv)[2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) mma(b[n], hv[m], x1[2 * h + m][n]);
        });
        TSTAMP(8 + 4 * hf);
    }

    TSTAMP(13);
    // ---- store: accumulator layout -> LDS tile -> 16 full rows per wave; optionally LayerNorm(out) with an affine as a bf16 side output
    //      (HAT: norm1 of the NEXT block, the input of its CAB convolutions, hat.py:165-170 -- saves that block's LayerNorm launch)
    relane();
    const bool want_ln = QKV || a.n1;
    if (want_ln) {
        float q1[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x4 t1 = x1[m][0] + x1[m][1] + x1[m][2];
            f32x4 t2 = x1[m][0] * x1[m][0];
#pragma unroll
            for (int r = 0; r < 4; ++r) t2[r] = __builtin_fmaf(x1[m][2][r], x1[m][2][r], __builtin_fmaf(x1[m][1][r], x1[m][1][r], t2[r]));
            q1[m] = (t1[0] + t1[1]) + (t1[2] + t1[3]);
            q2[m] = (t2[0] + t2[1]) + (t2[2] + t2[3]);
        }
        const float s1 = rows_reduce_scatter4(q1[0], q1[1], q1[2], q1[3]);
        const float s2 = rows_reduce_scatter4(q2[0], q2[1], q2[2], q2[3]);
        if (MT == 4 || ag < MT) *reinterpret_cast<float2*>(red + ((ag * 16 + ar) * 4 + w) * 2) = make_float2(s1, s2);  // (LayerNorm2's partials were consumed many barriers ago)
    }
    BLOCK_SYNC();  // every wave has read its last hidden fragments: the tile region is free
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        char* xm = smem + (m * 16 + ar) * XS + (w * 48 + ag * 4) * 4;
#pragma unroll
        for (int n = 0; n < 3; ++n) *reinterpret_cast<f32x4*>(xm + n * 64) = x1[m][n];
    }
    float mean[MT], rstd[MT];
    if (want_ln) {
        const float inv = 1.0f / (float)a.C;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const f32x4 pa = *reinterpret_cast<const f32x4*>(red + (m * 16 + ar) * 8), pb = *reinterpret_cast<const f32x4*>(red + (m * 16 + ar) * 8 + 4);
            mean[m] = (pa[0] + pa[2] + pb[0] + pb[2]) * inv;
            rstd[m] = rsqrtf(fmaxf((pa[1] + pa[3] + pb[1] + pb[3]) * inv - mean[m] * mean[m], 0.f) + a.eps);
        }
    }
    TSTAMP(17);
    BLOCK_SYNC();
    {
        f32x4 rowv[RW];
        const int l48 = lane < 48 ? lane : 0;
#pragma unroll
        for (int i = 0; i < RW; ++i) rowv[i] = *reinterpret_cast<const f32x4*>(smem + (RW * w + i) * XS + l48 * 16);
        float mr = 0.f, rs = 0.f;  // lane j < 16: the statistics of row 16 w + j (the same expression as mean[] / rstd[] above: the same bits)
        if (a.n1) {
            const float inv = 1.0f / (float)a.C;
            const int rr = RW * w + (lane & (RW - 1));
            const f32x4 pa = *reinterpret_cast<const f32x4*>(red + rr * 8), pb = *reinterpret_cast<const f32x4*>(red + rr * 8 + 4);
            mr = (pa[0] + pa[2] + pb[0] + pb[2]) * inv;
            rs = rsqrtf(fmaxf((pa[1] + pa[3] + pb[1] + pb[3]) * inv - mr * mr, 0.f) + a.eps);
            asm volatile("" : "+v"(n1g), "+v"(n1b));  // the affine has landed HERE, on every path: no vmcnt(0) behind each row's stores below
        }
        if (a.n1) {
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                const int prow = __builtin_amdgcn_readfirstlane(pixel_row(RW * w + i));  // (wave-uniform: the row pointer of store_row48 is a scalar operand)
                store_row48(a.out + (size_t)prow * a.ldx, rowv[i], lane);
                const float mi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mr), i));
                const float ri = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs), i));
                f32x4 nv;
#pragma unroll
                for (int r = 0; r < 4; ++r) nv[r] = __builtin_fmaf((rowv[i][r] - mi) * ri, n1g[r], n1b[r]);  // pad channels: gamma = beta = 0
                if (lane < 48) {
                    if constexpr (X3)
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.n1) + (size_t)prow * a.ldn + l48 * 4) = nv;
                    else
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.n1) + (size_t)prow * a.ldn + l48 * 4) = cvt4(nv);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < RW; ++i) store_row48(a.out + (size_t)__builtin_amdgcn_readfirstlane(pixel_row(RW * w + i)) * a.ldx, rowv[i], lane);
        }
    }
    TSTAMP(18);
    if constexpr (QKV) {
        // ---- the next block's LayerNorm1 + QKV on the same tokens (its LayerNorm affine, attention scale and biases are in the weight slots)
        relane();
        BLOCK_SYNC();  // every wave has read its rows of the tile: the image region is free again
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float nmr = -mean[m] * rstd[m];
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                f32x4 nv;
#pragma unroll
                for (int r = 0; r < 4; ++r) nv[r] = __builtin_fmaf(x1[m][n][r], rstd[m], nmr);
                if (n == 2) {
                    nv[0] = one_lane ? 1.0f : nv[0];
                    nv[1] = one_lane ? 1.0f : nv[1];
                }
                st_half(Aimg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, nv);
            }
        }
        // destinations in the NEXT block's window order (shift2): token t of this workgroup sits at pixel (py, px) of the image
        const int nwx = (int)dv.div_nwx.d, ntok_log2 = 2 * wsl;
        // (oca_pad2 > 0: the stage is the group's overlapping cross attention's -- k / v^T go to the zero-bordered image / plane layouts, addressed by the
        //  bordered pixel index (y + e) (W + 2 e) + x + e, and shift2 is 0)
        const int oe = a.oca_pad2, oWb = a.W + 2 * oe;
        auto dest = [&](int t, int& bw, int& tok) {
            const int tw = (int)part * NT_ + t;
            int y = ((int)wy << wsl) + (tw >> wsl) + a.shift, x = ((int)wx << wsl) + (tw & wsm) + a.shift;  // (y_mode == SR_Y_ROLL only)
            if (y >= a.H) y -= a.H;
            if (x >= a.W) x -= a.W;
            y -= a.shift2;
            x -= a.shift2;
            if (y < 0) y += a.H;
            if (x < 0) x += a.W;
            bw = ((int)bimg * dv.nw + (y >> wsl) * nwx + (x >> wsl)) * a.heads;
            tok = ((y & wsm) << wsl) + (x & wsm);
            return (y + oe) * oWb + x + oe;
        };
        int qbw[MT], qtok[MT], vbw[MT], vtok[MT], kpix[MT], vpix[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            kpix[m] = dest(m * 16 + ar, qbw[m], qtok[m]);       // q, k: lane = token 16 m + ar
            vpix[m] = dest(m * 16 + 4 * ag, vbw[m], vtok[m]);   // v^T: registers = tokens 16 m + 4 ag .. + 3 (they stay adjacent: shifts are multiples of 4)
        }
        const size_t oplane = (size_t)(a.H + 2 * oe) * oWb;
        BLOCK_SYNC();
        TSTAMP(19);
        const int hh = w >> 1, half = w & 1;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            f32x4 acc[MT][3];
            ws.template run<6, MT / 2>(TAIL_SLOTS + 6 * p, lane, loada_img(Aimg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&av)[2]) {
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
            TSTAMP(20 + 2 * p);
            const int head = 2 * p + hh;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                size_t qo, vo;
                if (a.frag_order) {  // fragment order of SrWindowAttn.qkv_frag (see sr_swin_qkv.hip); token = qtok / vtok of the next block's window
                    qo = (((size_t)(qbw[m] + head)) << ntok_log2) * a.hd_p + (size_t)(((qtok[m] >> 4) * 4 + 2 * half + (ag >> 1)) * 16 + (qtok[m] & 15)) * 8 + 4 * (ag & 1);
                    const int t6 = vtok[m] & 63;
                    vo = (((size_t)(vbw[m] + head)) << ntok_log2) * a.hd_p +
                         (size_t)((((vtok[m] >> 6) * 2 + half) * 2 + (t6 >> 5)) * 64 + ((t6 >> 2) & 3) * 16 + ar) * 8 + ((t6 >> 4) & 1) * 4;
                } else {
                    qo = ((((size_t)(qbw[m] + head)) << ntok_log2) + qtok[m]) * a.hd_p + 16 * half + 4 * ag;
                    vo = (((size_t)(vbw[m] + head) * a.hd_p + 16 * half + ar) << ntok_log2) + vtok[m];
                }
                size_t ko = qo;
                if (oe > 0) {  // as sr_swin_qkv.hip's oca_pad branch
                    ko = (((size_t)bimg * oplane + kpix[m]) * a.heads + head) * a.hd_p + 16 * half + 4 * ag;
                    vo = (((size_t)bimg * a.heads + head) * a.hd_p + 16 * half + ar) * oplane + vpix[m];
                }
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.q2) + qo) = cvt4(acc[m][0]);
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.k2) + ko) = cvt4(acc[m][1]);
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.vt2) + vo) = cvt4(acc[m][2]);
            }
            __builtin_amdgcn_sched_barrier(0);
            TSTAMP(21 + 2 * p);
        }
    }
    TSTAMP(14);
}

}  // namespace

#ifdef SR_STAMPS
extern "C" int sr_debug_tail_stamps(unsigned long long* host32) {
    return hipMemcpyFromSymbol(host32, HIP_SYMBOL(sr_dbg_tail), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int sr_swin_tail_supported(int C, int Cp, int heads, int hd_p, int ws, int Hp, int compute_dtype) {
    return ((compute_dtype == SR_BF16 || compute_dtype == SR_BF16X3) && C == 180 && Cp == 192 && heads == 6 && hd_p == 32 && (ws == 8 || ws == 16) && Hp == 384) ? 1 : 0;
}

extern "C" int sr_swin_tail(const SrSwinTail* p, void* stream) {
    SR_REQUIRE(p && p->x && p->out && p->o && p->wstream && p->bproj, "sr_swin_tail: null pointer");
    const SrSwinTail& a = *p;
    SR_REQUIRE(sr_swin_tail_supported(a.C, a.Cp, a.heads, a.hd_p, a.ws, a.Hp, a.compute_dtype), "sr_swin_tail: unsupported geometry / compute type (use sr_gemm + sr_mlp_fused)");
    SR_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.H % a.ws == 0 && a.W % a.ws == 0 && a.shift >= 0 && a.shift < a.ws && a.ldx >= a.Cp &&
                   a.y_mode >= SR_Y_ROLL && a.y_mode <= SR_Y_STRIP_LAST,
               "sr_swin_tail: bad geometry");
    SR_REQUIRE(a.ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0, "sr_swin_tail: the stream rows must be 16-byte aligned (ldx a multiple of 4, x 16-byte aligned)");
    SR_REQUIRE(!a.y || (a.ldy >= a.Cp && a.ldy % 4 == 0), "sr_swin_tail: gated second residual needs ldy");
    SR_REQUIRE(!a.y || a.pool_partial || (a.gate && a.ld_gate >= a.Cp && a.ld_gate % 4 == 0), "sr_swin_tail: gated second residual needs gate / ld_gate or the pool partials");
    SR_REQUIRE(!a.y || !a.pool_partial || (a.ca_w1 && a.ca_b1 && a.ca_w2 && a.ca_b2 && a.ca_Cr > 0 && a.ca_n_tiles > 0 &&
                                           ca_scratch_floats(a.Cp, a.ca_Cr) * 4 <= 22 * 1024),
               "sr_swin_tail: in-kernel gate needs the squeeze weights (scratch: 22 KiB)");
    SR_REQUIRE(!a.n1 || (a.n1_gamma && a.n1_beta && a.ldn >= a.Cp && a.ldn % 4 == 0), "sr_swin_tail: the LayerNorm side output needs n1_gamma, n1_beta, ldn");
    SR_REQUIRE(!a.q2 || (a.k2 && a.vt2 && a.shift2 >= 0 && a.shift2 < a.ws && a.shift2 % 4 == 0 && a.shift % 4 == 0 && a.y_mode == SR_Y_ROLL),
               "sr_swin_tail: the fused next-block QKV needs q2 / k2 / vt2, shifts that are multiples of 4 and y_mode SR_Y_ROLL");
    SR_REQUIRE(!a.frag_order || (a.q2 && a.ws == 16), "sr_swin_tail: frag_order is a layout of q2 / k2 / vt2 for 16 x 16 windows");
    SR_REQUIRE(a.oca_pad2 == 0 || (a.q2 && a.oca_pad2 > 0 && a.oca_pad2 % 4 == 0 && a.shift2 == 0 && !a.frag_order),
               "sr_swin_tail: oca_pad2 needs q2 / k2 / vt2, a border that is a multiple of 4, shift2 == 0 and row-major q2");
    SR_REQUIRE(a.compute_dtype == SR_BF16 || !a.q2, "sr_swin_tail: the fused next-block QKV exists for SR_BF16 only");
    SR_REQUIRE((long long)a.B * a.H * a.W < (1ll << 31), "sr_swin_tail: more than 2^31 tokens");
    SwinTailDev dv;
    dv.a = a;
    // 32-token workgroups (SrSwinTail.wg_tokens; 0: up to 128 workgroups of 64 tokens -- SR_TAIL_MT2_BELOW overrides that bound, read per call so that one process
    // can run both forms)
    SR_REQUIRE(a.wg_tokens == 0 || a.wg_tokens == 64 || (a.wg_tokens == 32 && a.compute_dtype == SR_BF16), "sr_swin_tail: wg_tokens is 0, 64, or 32 with SR_BF16");
    const char* mt2_env = getenv("SR_TAIL_MT2_BELOW");
    const bool mt2 = a.compute_dtype == SR_BF16 && (a.wg_tokens == 32 || (a.wg_tokens == 0 && (long long)a.B * a.H * a.W / 64 < (mt2_env ? atoi(mt2_env) : 129)));
    const int nwx = a.W / a.ws, nwy = a.H / a.ws, parts = a.ws * a.ws / (mt2 ? 32 : 64);
    dv.div_parts_img = make_fastdiv((uint32_t)(nwx * nwy * parts));
    dv.div_parts_win = make_fastdiv((uint32_t)parts);
    dv.div_nwx = make_fastdiv((uint32_t)nwx);
    dv.ws_log2 = a.ws == 8 ? 3 : 4;
    dv.nw = nwx * nwy;
    // pool partials by DMA into LDS (POOL_OFF): bf16 instantiations, squeeze geometries with register-resident MLP operands, partials + scratch within 80 KiB
    int lds16 = Lds<bf16>::TOTAL;
    dv.pool_lds = 0;
    const bool pool_global = false;  // (true keeps the loads
    if (a.compute_dtype == SR_BF16 && a.y && a.pool_partial && a.ca_Cr <= 8 && a.C <= 256 && a.Cp <= 256 && !pool_global) {
        const int need = POOL_OFF + a.ca_n_tiles * a.Cp * 4 + ca_scratch_floats(a.Cp, a.ca_Cr) * 4;
        if (need <= TAIL_LDS_MAX && (reinterpret_cast<uintptr_t>(a.pool_partial) & 15) == 0) {
            dv.pool_lds = 1;
            lds16 = need > lds16 ? (need + 15) & ~15 : lds16;
        }
    }
    const dim3 grid(a.B * nwx * nwy * parts);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (a.compute_dtype == SR_BF16X3) {
        static SrDeviceOnce attr_once;
        const hipError_t e = sr_once_per_device(attr_once, [&] { return sr_allow_lds(sr_swin_tail_kernel<bf3, false>, Lds<bf3>::TOTAL); });
        SR_REQUIRE(e == hipSuccess, "sr_swin_tail: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL((sr_swin_tail_kernel<bf3, false>), grid, dim3(256), Lds<bf3>::TOTAL, st, dv);
    } else if (mt2) {
        static SrDeviceOnce attr_once[2];
        if (a.q2) {
            const hipError_t e = sr_once_per_device(attr_once[1], [&] { return sr_allow_lds(sr_swin_tail_kernel<bf16, true, 2>, TAIL_LDS_MAX); });
            SR_REQUIRE(e == hipSuccess, "sr_swin_tail: hipFuncSetAttribute: %s", hipGetErrorString(e));
            hipLaunchKernelGGL((sr_swin_tail_kernel<bf16, true, 2>), grid, dim3(256), lds16, st, dv);
        } else {
            const hipError_t e = sr_once_per_device(attr_once[0], [&] { return sr_allow_lds(sr_swin_tail_kernel<bf16, false, 2>, TAIL_LDS_MAX); });
            SR_REQUIRE(e == hipSuccess, "sr_swin_tail: hipFuncSetAttribute: %s", hipGetErrorString(e));
            hipLaunchKernelGGL((sr_swin_tail_kernel<bf16, false, 2>), grid, dim3(256), lds16, st, dv);
        }
    } else if (a.q2) {
        static SrDeviceOnce attr_once;
        const hipError_t e = sr_once_per_device(attr_once, [&] { return sr_allow_lds(sr_swin_tail_kernel<bf16, true>, TAIL_LDS_MAX); });
        SR_REQUIRE(e == hipSuccess, "sr_swin_tail: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL((sr_swin_tail_kernel<bf16, true>), grid, dim3(256), lds16, st, dv);
    } else {
        static SrDeviceOnce attr_once;
        const hipError_t e = sr_once_per_device(attr_once, [&] { return sr_allow_lds(sr_swin_tail_kernel<bf16, false>, TAIL_LDS_MAX); });
        SR_REQUIRE(e == hipSuccess, "sr_swin_tail: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL((sr_swin_tail_kernel<bf16, false>), grid, dim3(256), lds16, st, dv);
    }
    SR_CHECK_LAUNCH("sr_swin_tail");
    return SR_OK;
}
