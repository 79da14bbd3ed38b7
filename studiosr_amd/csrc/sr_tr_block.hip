// Fast training path (C ABI v7): the window-attention transformer block of HAT / SwinIR under torch.autocast(bfloat16) (trainer.py:97-109),
// forward AND backward, in the stream form of the inference kernels (sr_swin_stream.h: one workgroup = 64 consecutive window-order tokens,
// K-group-major bf16 LDS images, ONE packed weight stream per kernel, fp32 residual stream / LayerNorm statistics / accumulators):
//   sr_tr_qkv_fwd   n1 = LayerNorm1(x) (affine in-kernel), q, k, v = qkv(n1) in both orientations            (hat.py:164-176; swinir.py:146-160)
//   sr_tr_tail_fwd  x1 = x + s_a (proj(O) + b) + y gate,  out = x1 + s_m (fc2(GELU(fc1(LayerNorm2(x1)))) + b2)  (hat.py:172-194; DropPath s_a, s_m
//                   per image: swinir.py:137,171-172 / hat.py:148,192-193); x1 is kept for the backward
//   sr_tr_tail_bwd  the adjoint of sr_tr_tail_fwd: recomputes LayerNorm2 / fc1 / GELU from x1, produces dx1, dO (both orientations), the CAB
//                   branch's dy and gate partials, and the token-major bf16 operands of the four weight-gradient GEMMs (sr_tr_wgrad)
//   sr_tr_qkv_bwd   the adjoint of sr_tr_qkv_fwd: dn1 = dqkv Wqkv (+ the CAB branch's dn1), LayerNorm1 backward, dx = dx1 + ...
// Differences from the inference kernels: the LayerNorm affine is applied in the kernel (not folded into the weights: its gradient is then
// the textbook one), GELU is the erf form, the hidden pad columns are set to one by the kernel, and biases still ride on the constant-one
// channels 180 / 181 (so every bias gradient is column 180 of a weight-gradient GEMM).
#include "sr_swin_stream.h"
#include "sr_ca.h"

namespace {

#ifndef SR_TR_DIST
#define SR_TR_DIST 4  // weight slots in flight ahead of the MFMAs: 2 / 3 / 4 / 6 = HAT step 15.86 / 15.49 / 15.57 / 15.53 ms, SwinIR 8.36 / 8.12 / 8.05 / 8.10 (one workgroup per CU: 512 registers, every L2 round trip exposed)
#endif
constexpr int TR_DIST = SR_TR_DIST;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct TrGeo {
    FastDiv div_parts_img, div_parts_win, div_nwx;
    int ws_log2, nw;
};

static TrGeo make_geo(int H, int W, int ws) {
    TrGeo g;
    const int nwx = W / ws, nwy = H / ws, parts = ws * ws / 64;
    g.div_parts_img = make_fastdiv((uint32_t)(nwx * nwy * parts));
    g.div_parts_win = make_fastdiv((uint32_t)parts);
    g.div_nwx = make_fastdiv((uint32_t)nwx);
    g.ws_log2 = ws == 8 ? 3 : 4;
    g.nw = nwx * nwy;
    return g;
}

struct Pos {
    int w, lane, ar, ag;
};

// sums of v over the 16 lanes of a row group (tokens), result in every lane of the group
SR_DEV float sum16(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v = wave_sum_xor(v, 8);
    return v;
}

// LayerNorm statistics of the 64 token rows held as x[m][n] (token 16 m + ar, channels 48 w + 16 n + 4 ag ..): mean / rstd per m
SR_DEV void ln_stats(const f32x4 (&x)[4][3], float* red, const Pos& P, int C, float eps, float (&mean)[4], float (&rstd)[4]) {
    float q1[4], q2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 t1 = x[m][0] + x[m][1] + x[m][2];
        f32x4 t2 = x[m][0] * x[m][0];
#pragma unroll
        for (int r = 0; r < 4; ++r) t2[r] = __builtin_fmaf(x[m][2][r], x[m][2][r], __builtin_fmaf(x[m][1][r], x[m][1][r], t2[r]));
        q1[m] = (t1[0] + t1[1]) + (t1[2] + t1[3]);
        q2[m] = (t2[0] + t2[1]) + (t2[2] + t2[3]);
    }
    const float s1 = rows_reduce_scatter4(q1[0], q1[1], q1[2], q1[3]);
    const float s2 = rows_reduce_scatter4(q2[0], q2[1], q2[2], q2[3]);
    *reinterpret_cast<float2*>(red + ((P.ag * 16 + P.ar) * 4 + P.w) * 2) = make_float2(s1, s2);
    BLOCK_SYNC();
    const float inv = 1.0f / (float)C;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4 pa = *reinterpret_cast<const f32x4*>(red + (m * 16 + P.ar) * 8), pb = *reinterpret_cast<const f32x4*>(red + (m * 16 + P.ar) * 8 + 4);
        mean[m] = (pa[0] + pa[2] + pb[0] + pb[2]) * inv;
        rstd[m] = rsqrtf(fmaxf((pa[1] + pa[3] + pb[1] + pb[3]) * inv - mean[m] * mean[m], 0.f) + eps);
    }
    BLOCK_SYNC();  // red may be reused
}

// per-row sums of two quantities over all 192 channels (as ln_stats, without the statistics arithmetic): out1[m], out2[m]
SR_DEV void row_sums2(const float (&q1)[4], const float (&q2)[4], float* red, const Pos& P, float (&o1)[4], float (&o2)[4]) {
    const float s1 = rows_reduce_scatter4(q1[0], q1[1], q1[2], q1[3]);
    const float s2 = rows_reduce_scatter4(q2[0], q2[1], q2[2], q2[3]);
    *reinterpret_cast<float2*>(red + ((P.ag * 16 + P.ar) * 4 + P.w) * 2) = make_float2(s1, s2);
    BLOCK_SYNC();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4 pa = *reinterpret_cast<const f32x4*>(red + (m * 16 + P.ar) * 8), pb = *reinterpret_cast<const f32x4*>(red + (m * 16 + P.ar) * 8 + 4);
        o1[m] = pa[0] + pa[2] + pb[0] + pb[2];
        o2[m] = pa[1] + pa[3] + pb[1] + pb[3];
    }
    BLOCK_SYNC();
}

// LayerNorm backward on the tile: dn = gradient w.r.t. the affine output, xh = normalised input, rstd per row; returns rstd (dxh - mean(dxh) - xh mean(dxh xh))
// in dn (pad channels 0) and writes this workgroup's dgamma / dbeta partials [2][Cp]
SR_DEV void ln_bwd_tile(f32x4 (&dn)[4][3], const f32x4 (&xh)[4][3], const float (&rstd)[4], const f32x4 (&gm)[3], float* red, const Pos& P, int C, float* part, int Cp) {
    // parameter gradients: sum over the 64 tokens
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        f32x4 sg = (f32x4)(0.0f), sb = (f32x4)(0.0f);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            sg += dn[m][n] * xh[m][n];
            sb += dn[m][n];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sg[r] = sum16(sg[r]);
            sb[r] = sum16(sb[r]);
        }
        if (P.ar == 0) {
            *reinterpret_cast<f32x4*>(part + P.w * 48 + n * 16 + P.ag * 4) = sg;
            *reinterpret_cast<f32x4*>(part + Cp + P.w * 48 + n * 16 + P.ag * 4) = sb;
        }
    }
    float q1[4], q2[4], c1[4], c2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 t1 = (f32x4)(0.0f), t2 = (f32x4)(0.0f);
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            dn[m][n] *= gm[n];  // dxh (gamma is 0 in the pad channels)
            t1 += dn[m][n];
            t2 += dn[m][n] * xh[m][n];
        }
        q1[m] = (t1[0] + t1[1]) + (t1[2] + t1[3]);
        q2[m] = (t2[0] + t2[1]) + (t2[2] + t2[3]);
    }
    row_sums2(q1, q2, red, P, c1, c2);
    const float inv = 1.0f / (float)C;
    const bool padw = P.w == 3;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float a1 = c1[m] * inv, a2 = c2[m] * inv;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dn[m][n][r] = rstd[m] * (dn[m][n][r] - a1 - xh[m][n][r] * a2);
            if (n == 2 && padw && P.ag >= 1) dn[m][n] = (f32x4)(0.0f);  // channels 180..191 are padding
        }
    }
}

// ============================================================================================================ qkv forward
struct QkvFwdDev {
    SrTrQkvFwd a;
    TrGeo g;
};

__global__ __launch_bounds__(256, 1) void sr_tr_qkv_fwd_kernel(QkvFwdDev dv) {
    typedef bf16 T;
    const SrTrQkvFwd& a = dv.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<T>* Aimg = reinterpret_cast<Frag<T>*>(smem);
    float* red = reinterpret_cast<float*>(smem + Lds<T>::RED_OFF);
    Pos P;
    P.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    P.lane = threadIdx.x & 63;
    P.ar = P.lane & 15;
    P.ag = P.lane >> 4;
    const int w = P.w, lane = P.lane, ar = P.ar, ag = P.ag;

    uint32_t bimg, rem, win, part, wy, wx;
    dv.g.div_parts_img.divmod((uint32_t)blockIdx.x, bimg, rem);
    dv.g.div_parts_win.divmod(rem, win, part);
    dv.g.div_nwx.divmod(win, wy, wx);
    const int wsl = dv.g.ws_log2, wsm = a.ws - 1;
    auto pixel_row = [&](int t) {
        const int tw = (int)part * NTOK + t;
        int y = ((int)wy << wsl) + (tw >> wsl) + a.shift;
        int x = ((int)wx << wsl) + (tw & wsm) + a.shift;
        if (y >= a.H) y -= a.H;
        if (x >= a.W) x -= a.W;
        return ((int)bimg * a.H + y) * a.W + x;
    };

    WStream<T, 18, TR_DIST> ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wstream), 0, 18 * 12 * 64 * 16, 0x00020000);
    ws.wave_frag = w * 3;
#pragma unroll
    for (int s0 = 0; s0 < TR_DIST; ++s0) ws.load(s0, lane);

    const int ch0 = w * 48 + ag * 4;
    f32x4 x1[4][3], gm[3], bt[3];
    int prow[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) prow[m] = pixel_row(m * 16 + ar);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) x1[m][n] = *reinterpret_cast<const f32x4*>(a.x + (size_t)prow[m] * a.ldx + ch0 + n * 16);
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        gm[n] = *reinterpret_cast<const f32x4*>(a.gamma + ch0 + n * 16);
        bt[n] = *reinterpret_cast<const f32x4*>(a.beta + ch0 + n * 16);
    }
    float mean[4], rstd[4];
    ln_stats(x1, red, P, a.C, a.eps, mean, rstd);
    const bool one_lane = (w == ONE_C / 48) && (ag == (ONE_C % 16) / 4);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            f32x4 nv;
#pragma unroll
            for (int r = 0; r < 4; ++r) nv[r] = __builtin_fmaf((x1[m][n][r] - mean[m]) * rstd[m], gm[n][r], bt[n][r]);
            if (n == 2) {
                nv[0] = one_lane ? 1.0f : nv[0];
                nv[1] = one_lane ? 1.0f : nv[1];
            }
            st_half(Aimg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, nv);
            if (a.n1) *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.n1) + (size_t)prow[m] * a.ldn + ch0 + n * 16) = cvt4(nv);
        }
    }
    BLOCK_SYNC();

    auto loada = [&](int c, int h, Frag<T> (&av)[2]) {
        const Frag<T>* arow = Aimg + (c * 4 + ag) * NTOK + h * 32 + ar;
        av[0] = arow[0];
        av[1] = arow[16];
    };
    const int hh = w >> 1, half = w & 1;
    const int ntok_log2 = 2 * wsl;
    const size_t bwin = (size_t)bimg * dv.g.nw + win;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        f32x4 acc[4][6];
        ws.template run<6>(6 * p, lane, loada, [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&av)[2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if (c == 0) {
                        mma0(b[t], av[m], acc[2 * h + m][t]);      // lane = token, registers = 4 features
                        mma0(av[m], b[t], acc[2 * h + m][3 + t]);  // lane = feature, registers = 4 tokens
                    } else {
                        mma(b[t], av[m], acc[2 * h + m][t]);
                        mma(av[m], b[t], acc[2 * h + m][3 + t]);
                    }
                }
        });
        const int head = 2 * p + hh;
        const size_t bh = bwin * a.heads + head;
        const size_t rm = ((bh << ntok_log2) + part * NTOK + ar) * 32 + 16 * half + 4 * ag;   // [tok][32]
        const size_t tr = ((bh * 32 + 16 * half + ar) << ntok_log2) + part * NTOK + 4 * ag;   // [32][tok]
        bf16* outs_rm[3] = {reinterpret_cast<bf16*>(a.q), reinterpret_cast<bf16*>(a.k), reinterpret_cast<bf16*>(a.v)};
        bf16* outs_tr[3] = {reinterpret_cast<bf16*>(a.qT), reinterpret_cast<bf16*>(a.kT), reinterpret_cast<bf16*>(a.vT)};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                *reinterpret_cast<bf16x4*>(outs_rm[t] + rm + m * 16 * 32) = cvt4(acc[m][t]);
                *reinterpret_cast<bf16x4*>(outs_tr[t] + tr + m * 16) = cvt4(acc[m][3 + t]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ============================================================================================================ tail forward
struct TailFwdDev {
    SrTrTailFwd a;
    TrGeo g;
};

constexpr int OROW = 384, OSTRIDE = 400;

SR_DEV void dma_gather16(const char* base, int lane_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(base), "s"(lds_dst)
                 : "memory");
}

__global__ __launch_bounds__(256, 1) void sr_tr_tail_fwd_kernel(TailFwdDev dv) {
    typedef bf16 T;
    const SrTrTailFwd& a = dv.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<T>* Aimg = reinterpret_cast<Frag<T>*>(smem);
    Frag<T>* Himg = Aimg + CELLS_A;
    float* red = reinterpret_cast<float*>(smem + Lds<T>::RED_OFF);
    Pos P;
    P.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    P.lane = threadIdx.x & 63;
    P.ar = P.lane & 15;
    P.ag = P.lane >> 4;
    const int w = P.w, lane = P.lane, ar = P.ar, ag = P.ag;

    uint32_t bimg, rem, win, part, wy, wx;
    dv.g.div_parts_img.divmod((uint32_t)blockIdx.x, bimg, rem);
    dv.g.div_parts_win.divmod(rem, win, part);
    dv.g.div_nwx.divmod(win, wy, wx);
    const int wsl = dv.g.ws_log2, wsm = a.ws - 1;
    auto pixel_row = [&](int t) {
        const int tw = (int)part * NTOK + t;
        int y = ((int)wy << wsl) + (tw >> wsl) + a.shift;
        int x = ((int)wx << wsl) + (tw & wsm) + a.shift;
        if (y >= a.H) y -= a.H;
        if (x >= a.W) x -= a.W;
        return ((int)bimg * a.H + y) * a.W + x;
    };
    {
        const unsigned img_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
        const char* orow = reinterpret_cast<const char*>(a.o) + (size_t)blockIdx.x * NTOK * OROW;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int j = w + 4 * i;
            if (j < 25) {
                const int q = j * 64 + lane;
                const int row = (q * 1311) >> 15;
                const int cc = q - row * 25;
                if (cc < 24) dma_gather16(orow, row * OROW + cc * 16, __builtin_amdgcn_readfirstlane(img_lds + j * 1024));
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    WStream<T, 30, TR_DIST> ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wstream), 0, 30 * 12 * 64 * 16, 0x00020000);
    ws.wave_frag = w * 3;
#pragma unroll
    for (int s0 = 0; s0 < TR_DIST; ++s0) ws.load(s0, lane);
    __builtin_amdgcn_sched_barrier(0);

    const int ch0 = w * 48 + ag * 4;
    f32x4 xs[4][3], gt[3], bp[3], gm[3], bt[3];
    bf16x4 yv[4][3];
    int prow[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) prow[m] = pixel_row(m * 16 + ar);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) xs[m][n] = *reinterpret_cast<const f32x4*>(a.x + (size_t)prow[m] * a.ldx + ch0 + n * 16);
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        bp[n] = *reinterpret_cast<const f32x4*>(a.bproj + ch0 + n * 16);
        gm[n] = *reinterpret_cast<const f32x4*>(a.gamma + ch0 + n * 16);
        bt[n] = *reinterpret_cast<const f32x4*>(a.beta + ch0 + n * 16);
        gt[n] = (f32x4)(0.0f);
    }
    if (a.y) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) yv[m][n] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(a.y) + (size_t)prow[m] * a.ldy + ch0 + n * 16);
    }
    const float sa = a.s_a ? a.s_a[bimg] : 1.0f, sm = a.s_m ? a.s_m[bimg] : 1.0f;
    __builtin_amdgcn_sched_barrier(0);
    if (a.y) {
        SrChannelAttn ca;
        ca.pool_partial = a.pool_partial; ca.w1 = a.ca_w1; ca.b1 = a.ca_b1; ca.w2 = a.ca_w2; ca.b2 = a.ca_b2;
        ca.B = a.B; ca.H = a.H; ca.W = a.W; ca.C = a.C; ca.C_p = a.Cp; ca.Cr = a.ca_Cr; ca.n_tiles = a.ca_n_tiles; ca.y_scale = a.y_scale;
        const float* gate = ca_squeeze(ca, (int)bimg, reinterpret_cast<float*>(smem + NTOK * OSTRIDE));
#pragma unroll
        for (int n = 0; n < 3; ++n) gt[n] = *reinterpret_cast<const f32x4*>(gate + ch0 + n * 16);
        if (a.gate_out && part == 0 && win == 0 && threadIdx.x < (unsigned)a.Cp) a.gate_out[(size_t)bimg * a.Cp + threadIdx.x] = gate[threadIdx.x];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BLOCK_SYNC();

    auto loada_img = [&](const Frag<T>* img) {
        return [&, img](int c, int h, Frag<T> (&av)[2]) {
            const Frag<T>* arow = img + (c * 4 + ag) * NTOK + h * 32 + ar;
            av[0] = arow[0];
            av[1] = arow[16];
        };
    };
    auto loada_o = [&](int c, int h, Frag<T> (&av)[2]) {
        const char* ob = smem + (h * 32 + ar) * OSTRIDE + (c * 4 + ag) * 16;
        av[0] = *reinterpret_cast<const Frag<T>*>(ob);
        av[1] = *reinterpret_cast<const Frag<T>*>(ob + 16 * OSTRIDE);
    };
    f32x4 x1[4][3];
    ws.template run<6>(0, lane, loada_o, [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&ov)[2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                if (c == 0)
                    mma0(b[n], ov[m], x1[2 * h + m][n]);
                else
                    mma(b[n], ov[m], x1[2 * h + m][n]);
            }
    });
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = __builtin_fmaf(x1[m][n][r] + bp[n][r], sa, xs[m][n][r]);
                if (a.y) v = __builtin_fmaf((float)yv[m][n][r], gt[n][r], v);
                x1[m][n][r] = v;
            }
            *reinterpret_cast<f32x4*>(a.x1 + (size_t)prow[m] * a.ldx + ch0 + n * 16) = x1[m][n];
        }
    BLOCK_SYNC();  // every wave has read its O fragments: the image region may be overwritten
    float mean[4], rstd[4];
    ln_stats(x1, red, P, a.C, a.eps, mean, rstd);
    const bool one_lane = (w == ONE_C / 48) && (ag == (ONE_C % 16) / 4);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            f32x4 nv;
#pragma unroll
            for (int r = 0; r < 4; ++r) nv[r] = __builtin_fmaf((x1[m][n][r] - mean[m]) * rstd[m], gm[n][r], bt[n][r]);
            if (n == 2) {
                nv[0] = one_lane ? 1.0f : nv[0];
                nv[1] = one_lane ? 1.0f : nv[1];
            }
            st_half(Aimg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, nv);
        }
    BLOCK_SYNC();

#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        f32x4 acc[4][3];
        ws.template run<6>(6 + 12 * hf, lane, loada_img(Aimg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&av)[2]) {
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
        if (hf == 1) BLOCK_SYNC();
        const bool ones = hf == 1 && w == 3 && ag == 2;  // hidden columns 360, 361 carry the fc2 bias
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 g;
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = gelu_fast(acc[m][n][r]) * sm;
                if (n == 1 && ones) {
                    g[0] = sm;
                    g[1] = sm;
                }
                st_half(Himg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, g);
            }
        __builtin_amdgcn_sched_barrier(0);
        BLOCK_SYNC();
        ws.template run<6>(12 + 12 * hf, lane, loada_img(Himg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&hv)[2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) mma(b[n], hv[m], x1[2 * h + m][n]);
        });
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) *reinterpret_cast<f32x4*>(a.out + (size_t)prow[m] * a.ldx + ch0 + n * 16) = x1[m][n];
}

// ============================================================================================================ tail backward
struct TailBwdDev {
    SrTrTailBwd a;
    TrGeo g;
};
// weight stream: per hidden half hf: 6 fc1 slots (forward), 6 W2^T slots (rows = hidden columns of the half, K = channels), 6 W1^T slots
// (rows = channels, K = hidden columns of the half); then 6 Wproj^T slots (rows = (head, feature), K = channels)
constexpr int TB_SLOTS = 42;
constexpr int TB_LDS = 3 * CELLS_A * 16 + LDS_RED;

__global__ __launch_bounds__(256, 1) void sr_tr_tail_bwd_kernel(TailBwdDev dv) {
    typedef bf16 T;
    const SrTrTailBwd& a = dv.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<T>* Aimg = reinterpret_cast<Frag<T>*>(smem);
    Frag<T>* Dimg = Aimg + CELLS_A;
    Frag<T>* Himg = Dimg + CELLS_A;
    float* red = reinterpret_cast<float*>(smem + 3 * CELLS_A * 16);
    Pos P;
    P.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    P.lane = threadIdx.x & 63;
    P.ar = P.lane & 15;
    P.ag = P.lane >> 4;
    const int w = P.w, lane = P.lane, ar = P.ar, ag = P.ag;

    uint32_t bimg, rem, win, part, wy, wx;
    dv.g.div_parts_img.divmod((uint32_t)blockIdx.x, bimg, rem);
    dv.g.div_parts_win.divmod(rem, win, part);
    dv.g.div_nwx.divmod(win, wy, wx);
    const int wsl = dv.g.ws_log2, wsm = a.ws - 1;
    auto pixel_row = [&](int t) {
        const int tw = (int)part * NTOK + t;
        int y = ((int)wy << wsl) + (tw >> wsl) + a.shift;
        int x = ((int)wx << wsl) + (tw & wsm) + a.shift;
        if (y >= a.H) y -= a.H;
        if (x >= a.W) x -= a.W;
        return ((int)bimg * a.H + y) * a.W + x;
    };
    WStream<T, TB_SLOTS, TR_DIST> ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wstream), 0, TB_SLOTS * 12 * 64 * 16, 0x00020000);
    ws.wave_frag = w * 3;
#pragma unroll
    for (int s0 = 0; s0 < TR_DIST; ++s0) ws.load(s0, lane);

    const int ch0 = w * 48 + ag * 4;
    const size_t wrow0 = (size_t)blockIdx.x * NTOK;  // first window-order row of this workgroup
    f32x4 d[4][3], xh[4][3], gm[3], bt[3];
    int prow[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) prow[m] = pixel_row(m * 16 + ar);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            xh[m][n] = *reinterpret_cast<const f32x4*>(a.x1 + (size_t)prow[m] * a.ldx + ch0 + n * 16);
            d[m][n] = *reinterpret_cast<const f32x4*>(a.dout + (size_t)prow[m] * a.ldx + ch0 + n * 16);
        }
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        gm[n] = *reinterpret_cast<const f32x4*>(a.gamma + ch0 + n * 16);
        bt[n] = *reinterpret_cast<const f32x4*>(a.beta + ch0 + n * 16);
    }
    const float sa = a.s_a ? a.s_a[bimg] : 1.0f, sm = a.s_m ? a.s_m[bimg] : 1.0f;
    float mean[4], rstd[4];
    ln_stats(xh, red, P, a.C, a.eps, mean, rstd);
    const bool one_lane = (w == ONE_C / 48) && (ag == (ONE_C % 16) / 4);
    const bool padl = w == 3 && ag >= 1;  // n == 2: channels 180..191
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            f32x4 nv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xh[m][n][r] = (xh[m][n][r] - mean[m]) * rstd[m];
                nv[r] = __builtin_fmaf(xh[m][n][r], gm[n][r], bt[n][r]);
            }
            if (n == 2) {
                if (padl) xh[m][n] = (f32x4)(0.0f);
                nv[0] = one_lane ? 1.0f : nv[0];
                nv[1] = one_lane ? 1.0f : nv[1];
            }
            st_half(Aimg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, nv);
            st_half(Dimg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, d[m][n]);
            const size_t ro = (wrow0 + m * 16 + ar) * a.Cp + ch0 + n * 16;
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.n2w) + ro) = cvt4(nv);
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.doutw) + ro) = cvt4(d[m][n]);
        }
    BLOCK_SYNC();

    auto loada_img = [&](const Frag<T>* img) {
        return [&, img](int c, int h, Frag<T> (&av)[2]) {
            const Frag<T>* arow = img + (c * 4 + ag) * NTOK + h * 32 + ar;
            av[0] = arow[0];
            av[1] = arow[16];
        };
    };
    f32x4 dn[4][3];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        f32x4 gp[4][3];
        {
            f32x4 acc[4][3];
            ws.template run<6>(18 * hf, lane, loada_img(Aimg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&av)[2]) {
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
            const bool ones = hf == 1 && w == 3 && ag == 2;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    f32x4 g;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float Phi, phi;
                        const float hv = acc[m][n][r];
                        gauss(hv, Phi, phi);
                        g[r] = hv * Phi * sm;
                        gp[m][n][r] = (Phi + hv * phi) * sm;
                    }
                    if (n == 1 && ones) {
                        g[0] = sm;
                        g[1] = sm;
                    }
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.gw) + (wrow0 + m * 16 + ar) * a.Hp + 192 * hf + ch0 + n * 16) = cvt4(g);
                }
        }
        {
            f32x4 acc[4][3];
            ws.template run<6>(18 * hf + 6, lane, loada_img(Dimg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&av)[2]) {
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
            if (hf == 1) BLOCK_SYNC();  // the W1^T steps of the first half have read the hidden image everywhere
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    const f32x4 dh = acc[m][n] * gp[m][n];
                    st_half(Himg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, dh);
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dhw) + (wrow0 + m * 16 + ar) * a.Hp + 192 * hf + ch0 + n * 16) = cvt4(dh);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        BLOCK_SYNC();
        ws.template run<6>(18 * hf + 12, lane, loada_img(Himg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&hv)[2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    if (c == 0 && hf == 0)
                        mma0(b[n], hv[m], dn[2 * h + m][n]);
                    else
                        mma(b[n], hv[m], dn[2 * h + m][n]);
                }
        });
    }
    // ---- LayerNorm2 backward, + the shortcut
    ln_bwd_tile(dn, xh, rstd, gm, red, P, a.C, a.ln_part + (size_t)blockIdx.x * 2 * a.Cp, a.Cp);
    // (ln_bwd_tile ends with a barrier: every wave is past its last hidden-image read, the LayerNorm image was last read long ago)
    f32x4 gt[3];
#pragma unroll
    for (int n = 0; n < 3; ++n) gt[n] = (f32x4)(0.0f);
    if (a.y) {
#pragma unroll
        for (int n = 0; n < 3; ++n) gt[n] = *reinterpret_cast<const f32x4*>(a.gate + (size_t)bimg * a.Cp + ch0 + n * 16);
    }
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        f32x4 dgs = (f32x4)(0.0f);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            dn[m][n] += d[m][n];  // dx1
            if (n == 2 && padl) dn[m][n] = (f32x4)(0.0f);
            *reinterpret_cast<f32x4*>(a.dx1 + (size_t)prow[m] * a.ldx + ch0 + n * 16) = dn[m][n];
            f32x4 sv = dn[m][n] * sa;
            st_half(Aimg + (6 * w + 2 * n + (ag >> 1)) * NTOK + m * 16 + ar, ag & 1, sv);
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dx1sw) + (wrow0 + m * 16 + ar) * a.Cp + ch0 + n * 16) = cvt4(sv);
            if (a.y) {
                const bf16x4 yv = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(a.y) + (size_t)prow[m] * a.ldy + ch0 + n * 16);
                dgs += dn[m][n] * widen4(yv);
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dyc) + (size_t)prow[m] * a.ldy + ch0 + n * 16) = cvt4(dn[m][n] * gt[n]);
            }
        }
        if (a.y) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dgs[r] = sum16(dgs[r]);
            if (ar == 0) *reinterpret_cast<f32x4*>(a.dgate_part + (size_t)blockIdx.x * a.Cp + ch0 + n * 16) = dgs;
        }
    }
    BLOCK_SYNC();
    // ---- dO = (s_a dx1) Wproj, in both orientations
    {
        f32x4 acc[4][6];
        ws.template run<6>(36, lane, loada_img(Aimg), [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&av)[2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    if (c == 0) {
                        mma0(b[n], av[m], acc[2 * h + m][n]);
                        mma0(av[m], b[n], acc[2 * h + m][3 + n]);
                    } else {
                        mma(b[n], av[m], acc[2 * h + m][n]);
                        mma(av[m], b[n], acc[2 * h + m][3 + n]);
                    }
                }
        });
        const int ntok_log2 = 2 * wsl;
        const size_t bwin = (size_t)bimg * dv.g.nw + win;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int nt = 3 * w + n;  // 16-feature tile: head nt >> 1, feature half nt & 1
            const size_t tr = (((bwin * a.heads + (nt >> 1)) * 32 + 16 * (nt & 1) + ar) << ntok_log2) + part * NTOK + 4 * ag;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dOw) + (wrow0 + m * 16 + ar) * a.Cp + ch0 + n * 16) = cvt4(acc[m][n]);
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dOT) + tr + m * 16) = cvt4(acc[m][3 + n]);
            }
        }
    }
}

// ============================================================================================================ qkv backward
struct QkvBwdDev {
    SrTrQkvBwd a;
    TrGeo g;
};
constexpr int QB_CELLS = 72 * 64;  // dqkv image [72 k-groups][64 tokens]
constexpr int QB_LDS = QB_CELLS * 16 + LDS_RED;

__global__ __launch_bounds__(256, 1) void sr_tr_qkv_bwd_kernel(QkvBwdDev dv) {
    typedef bf16 T;
    const SrTrQkvBwd& a = dv.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<T>* Gimg = reinterpret_cast<Frag<T>*>(smem);
    float* red = reinterpret_cast<float*>(smem + QB_CELLS * 16);
    Pos P;
    P.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    P.lane = threadIdx.x & 63;
    P.ar = P.lane & 15;
    P.ag = P.lane >> 4;
    const int w = P.w, lane = P.lane, ar = P.ar, ag = P.ag;

    uint32_t bimg, rem, win, part, wy, wx;
    dv.g.div_parts_img.divmod((uint32_t)blockIdx.x, bimg, rem);
    dv.g.div_parts_win.divmod(rem, win, part);
    dv.g.div_nwx.divmod(win, wy, wx);
    const int wsl = dv.g.ws_log2, wsm = a.ws - 1;
    auto pixel_row = [&](int t) {
        const int tw = (int)part * NTOK + t;
        int y = ((int)wy << wsl) + (tw >> wsl) + a.shift;
        int x = ((int)wx << wsl) + (tw & wsm) + a.shift;
        if (y >= a.H) y -= a.H;
        if (x >= a.W) x -= a.W;
        return ((int)bimg * a.H + y) * a.W + x;
    };
    WStream<T, 18, TR_DIST> ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wstream), 0, 18 * 12 * 64 * 16, 0x00020000);
    ws.wave_frag = w * 3;
#pragma unroll
    for (int s0 = 0; s0 < TR_DIST; ++s0) ws.load(s0, lane);

    // ---- dq | dk | dv of this workgroup's 64 tokens -> image: k-group (part p, head, g = wave), token = lane
    {
        const int ntok_log2 = 2 * wsl;
        const size_t bwin = (size_t)bimg * dv.g.nw + win;
        const bf16* src_q = reinterpret_cast<const bf16*>(a.dq);
        const bf16* src_k = reinterpret_cast<const bf16*>(a.dk);
        const bf16* src_v = reinterpret_cast<const bf16*>(a.dv);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            Frag<T> tmp[6];
#pragma unroll
            for (int head = 0; head < 6; ++head) {
                const size_t off = ((((bwin * a.heads + head) << ntok_log2) + part * NTOK + lane) * 32) + 8 * w;
                tmp[head] = *reinterpret_cast<const Frag<T>*>((p == 0 ? src_q : p == 1 ? src_k : src_v) + off);
            }
#pragma unroll
            for (int head = 0; head < 6; ++head) {
                Gimg[((p * 6 + head) * 4 + w) * NTOK + lane] = tmp[head];
                // token-major copy [T][3 * heads * 32] (window order): the A operand of the qkv weight gradient
                *reinterpret_cast<Frag<T>*>(reinterpret_cast<bf16*>(a.dqkvw) + ((size_t)blockIdx.x * NTOK + lane) * (3 * 6 * 32) + (p * 6 + head) * 32 + 8 * w) = tmp[head];
            }
        }
    }
    const int ch0 = w * 48 + ag * 4;
    const size_t wrow0 = (size_t)blockIdx.x * NTOK;
    f32x4 xh[4][3], gm[3], bt[3];
    int prow[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) prow[m] = pixel_row(m * 16 + ar);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) xh[m][n] = *reinterpret_cast<const f32x4*>(a.x + (size_t)prow[m] * a.ldx + ch0 + n * 16);
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        gm[n] = *reinterpret_cast<const f32x4*>(a.gamma + ch0 + n * 16);
        bt[n] = *reinterpret_cast<const f32x4*>(a.beta + ch0 + n * 16);
    }
    float mean[4], rstd[4];
    ln_stats(xh, red, P, a.C, a.eps, mean, rstd);  // (its barriers also publish the image)
    const bool one_lane = (w == ONE_C / 48) && (ag == (ONE_C % 16) / 4);
    const bool padl = w == 3 && ag >= 1;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            f32x4 nv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xh[m][n][r] = (xh[m][n][r] - mean[m]) * rstd[m];
                nv[r] = __builtin_fmaf(xh[m][n][r], gm[n][r], bt[n][r]);
            }
            if (n == 2) {
                if (padl) xh[m][n] = (f32x4)(0.0f);
                nv[0] = one_lane ? 1.0f : nv[0];
                nv[1] = one_lane ? 1.0f : nv[1];
            }
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.n1w) + (wrow0 + m * 16 + ar) * a.Cp + ch0 + n * 16) = cvt4(nv);
        }
    auto loada = [&](int c, int h, Frag<T> (&av)[2]) {
        const Frag<T>* arow = Gimg + (c * 4 + ag) * NTOK + h * 32 + ar;
        av[0] = arow[0];
        av[1] = arow[16];
    };
    f32x4 dn[4][3];
    ws.template run<18>(0, lane, loada, [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&av)[2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                if (c == 0)
                    mma0(b[n], av[m], dn[2 * h + m][n]);
                else
                    mma(b[n], av[m], dn[2 * h + m][n]);
            }
    });
    if (a.dn1c) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n)
                dn[m][n] += widen4(*reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(a.dn1c) + (size_t)prow[m] * a.ldn + ch0 + n * 16));
    }
    ln_bwd_tile(dn, xh, rstd, gm, red, P, a.C, a.ln_part + (size_t)blockIdx.x * 2 * a.Cp, a.Cp);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            f32x4 v = dn[m][n] + *reinterpret_cast<const f32x4*>(a.dx1 + (size_t)prow[m] * a.ldx + ch0 + n * 16);
            if (n == 2 && padl) v = (f32x4)(0.0f);
            *reinterpret_cast<f32x4*>(a.dx + (size_t)prow[m] * a.ldx + ch0 + n * 16) = v;
        }
}

// ============================================================================================================ CAB helpers
// Channel-attention backward + its term of the conv-branch gradient (hat.py:25-38 under backward): for image b
//   gate = y_scale sigmoid(z2), z2 = W2 relu(z1) + b2, z1 = W1 mean + b1, mean = avgpool(y);  dgate[c] = sum_px dx1[px][c] y[px][c]
//   -> dW2, db2, dW1, db1 (one partial per image) and dmean; dy[px][c] (already dx1 gate) += dmean[c] / (H W)
__global__ __launch_bounds__(256) void sr_tr_ca_bwd_kernel(SrTrCaBwd a) {
    __shared__ float sm[8 * 256 + 3 * 256 + 64];
    const int b = blockIdx.y, tid = threadIdx.x;
    SrChannelAttn ca;
    ca.pool_partial = a.pool_partial; ca.w1 = a.w1; ca.b1 = a.b1; ca.w2 = a.w2; ca.b2 = a.b2;
    ca.B = a.B; ca.H = a.H; ca.W = a.W; ca.C = a.C; ca.C_p = a.Cp; ca.Cr = a.Cr; ca.n_tiles = a.n_tiles; ca.y_scale = a.y_scale;
    float* scratch = sm;  // ca_squeeze: mean [Cp] | hid [Cr] | gate [Cp] | part [8][Cp]
    const float* gate = ca_squeeze(ca, b, scratch);
    const float* mean = scratch;
    const float* hid = scratch + a.Cp;
    __shared__ float dz2[256], dz1[16], dmean[256];
    {   // dgate[c] = sum over the image's tail-backward workgroups: 4 slices of parts x 48 channel quads per pass, 8 loads in flight, slice order fixed
        float* acc = sm + 8 * 256;  // [4][Cp] (behind ca_squeeze's scratch, which is read-only from here on ... the `part` region is free)
        const int quads = a.Cp >> 2;
        if (tid < 4 * quads) {
            const int sl = tid / quads, q = tid - sl * quads;
            const float* pp = a.dgate_part + (size_t)b * a.parts * a.Cp + 4 * q;
            f32x4 s4 = (f32x4)(0.0f);
            for (int p0 = sl; p0 < a.parts; p0 += 32) {
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(pp + (size_t)min(p0 + 4 * k, a.parts - 1) * a.Cp);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (p0 + 4 * k < a.parts) s4 += v[k];
            }
            *reinterpret_cast<f32x4*>(acc + sl * a.Cp + 4 * q) = s4;
        }
        __syncthreads();
        if (tid < a.Cp) {
            const float dg = (acc[tid] + acc[a.Cp + tid]) + (acc[2 * a.Cp + tid] + acc[3 * a.Cp + tid]);
            const float s = tid < a.C ? gate[tid] / a.y_scale : 0.f;  // sigmoid
            dz2[tid] = tid < a.C ? dg * a.y_scale * s * (1.0f - s) : 0.f;
        }
    }
    __syncthreads();
    {   // dh[j] = sum_c dz2[c] W2[c][j]: one channel per thread (its Cr weights are contiguous), wave sums by butterfly, the four wave sums through LDS
        // (six threads walking 180 dependent L2 loads each was the longest piece of this prologue, which every workgroup of the image repeats)
        __shared__ float wsum[64];  // [4 waves][16]
        float ph[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ph[j] = (tid < a.C && j < a.Cr) ? dz2[tid] * a.w2[tid * a.Cr + j] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = ph[j];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((tid & 63) == 0) wsum[(tid >> 6) * 16 + j] = v;
        }
        __syncthreads();
        if (tid < a.Cr) {
            const float dh = (wsum[tid] + wsum[16 + tid]) + (wsum[32 + tid] + wsum[48 + tid]);
            dz1[tid] = hid[tid] > 0.f ? dh : 0.f;
        }
    }
    __syncthreads();
    if (tid < a.Cp) {
        float dm = 0.f;
        if (tid < a.C)
            for (int j = 0; j < a.Cr; ++j) dm += dz1[j] * a.w1[j * a.C + tid];
        dmean[tid] = dm / (float)(a.H * a.W);
    }
    __syncthreads();
    if (blockIdx.x == 0) {  // parameter-gradient partials of this image: [dw1 Cr*C | db1 Cr | dw2 C*Cr | db2 C]
        float* g = a.dparam_part + (size_t)b * a.dparam_stride;
        for (int i = tid; i < a.Cr * a.C; i += 256) g[i] = dz1[i / a.C] * mean[i % a.C];
        if (tid < a.Cr) g[a.Cr * a.C + tid] = dz1[tid];
        for (int i = tid; i < a.C * a.Cr; i += 256) g[a.Cr * a.C + a.Cr + i] = dz2[i / a.Cr] * hid[i % a.Cr];
        if (tid < a.C) g[a.Cr * a.C + a.Cr + a.C * a.Cr + tid] = dz2[tid];
    }
    // dy += dmean over this workgroup's pixel slab
    const int hw = a.H * a.W;
    const int per = (hw + gridDim.x - 1) / gridDim.x;
    const int p0 = blockIdx.x * per, p1 = min(hw, p0 + per);
    const int quads = a.Cp >> 2;
    bf16* dy = reinterpret_cast<bf16*>(a.dy) + (size_t)b * hw * a.ld;
    // four independent items in flight per thread (the in-place update had been one dependent load -> add -> store at a time: 48 round trips per thread with 16 slabs)
    for (int i0 = p0 * quads + tid; i0 < p1 * quads; i0 += 4 * 256) {
        bf16x4* pt[4];
        bf16x4 raw[4];
        int qq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * 256, p1 * quads - 1);
            const int px = i / quads;
            qq[u] = i - px * quads;
            pt[u] = reinterpret_cast<bf16x4*>(dy + (size_t)px * a.ld + 4 * qq[u]);
            raw[u] = *pt[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * 256 >= p1 * quads) continue;
            f32x4 v = widen4(raw[u]);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += dmean[4 * qq[u] + r];
            *pt[u] = cvt4(v);
        }
    }
}

// GELU pieces of the CAB (hat.py:43: nn.GELU between the two convolutions): g = gelu(x) and dx = dg * gelu'(x), bf16 in / out
__global__ __launch_bounds__(256) void sr_tr_gelu_kernel(const bf16* x, const bf16* dg, bf16* g, bf16* dx, long long n8) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const bf16x8 xv = reinterpret_cast<const bf16x8*>(x)[i];
    bf16x8 dgv = (bf16x8)(0.0f), gv, dxv;
    if (dg) dgv = reinterpret_cast<const bf16x8*>(dg)[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float Phi, phi;
        const float v = (float)xv[j];
        gauss(v, Phi, phi);
        gv[j] = (bf16)(v * Phi);
        dxv[j] = (bf16)((float)dgv[j] * (Phi + v * phi));
    }
    if (g) reinterpret_cast<bf16x8*>(g)[i] = gv;
    if (dx) reinterpret_cast<bf16x8*>(dx)[i] = dxv;
}


// ============================================================================================================ LayerNorm backward alone
// nn.LayerNorm backward on the padded fp32 stream (swinir.py:26,313 / hat.py:460: patch_embed.norm and the final norm): rows in memory order,
// 64 per workgroup; dx = LN'(x)^T dy (+ dskip), dgamma / dbeta partials per workgroup
__global__ __launch_bounds__(256, 1) void sr_tr_ln_bwd_kernel(SrTrLnBwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    Pos P;
    P.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    P.lane = threadIdx.x & 63;
    P.ar = P.lane & 15;
    P.ag = P.lane >> 4;
    const int ch0 = P.w * 48 + P.ag * 4;
    const size_t row0 = (size_t)blockIdx.x * NTOK;
    f32x4 xh[4][3], dn[4][3], gm[3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const size_t o = (row0 + m * 16 + P.ar) * a.ld + ch0 + n * 16;
            xh[m][n] = *reinterpret_cast<const f32x4*>(a.x + o);
            dn[m][n] = a.dy_bf16 ? widen4(*reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(a.dy) + o)) : *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.dy) + o);
        }
#pragma unroll
    for (int n = 0; n < 3; ++n) gm[n] = *reinterpret_cast<const f32x4*>(a.gamma + ch0 + n * 16);
    float mean[4], rstd[4];
    ln_stats(xh, red, P, a.C, a.eps, mean, rstd);
    const bool padl = P.w == 3 && P.ag >= 1;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) xh[m][n][r] = (xh[m][n][r] - mean[m]) * rstd[m];
            if (n == 2 && padl) xh[m][n] = (f32x4)(0.0f);
        }
    ln_bwd_tile(dn, xh, rstd, gm, red, P, a.C, a.ln_part + (size_t)blockIdx.x * 2 * a.Cp, a.Cp);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const size_t o = (row0 + m * 16 + P.ar) * a.ld + ch0 + n * 16;
            f32x4 v = dn[m][n];
            if (a.dskip) v += a.dskip_bf16 ? widen4(*reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(a.dskip) + o)) : *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.dskip) + o);
            if (n == 2 && padl) v = (f32x4)(0.0f);
            *reinterpret_cast<f32x4*>(a.dx + o) = v;
        }
}

}  // namespace

static bool tr_geo_ok(int B, int H, int W, int C, int Cp, int heads, int hd_p, int ws, int shift, int ldx) {
    return B > 0 && H > 0 && W > 0 && C == 180 && Cp == 192 && heads == 6 && hd_p == 32 && (ws == 8 || ws == 16) && H % ws == 0 && W % ws == 0 && shift >= 0 && shift < ws &&
           ldx >= Cp && ldx % 4 == 0 && (long long)B * H * W < (1ll << 31);
}

extern "C" int sr_tr_block_supported(int C, int Cp, int heads, int hd_p, int ws, int Hp) {
    return (C == 180 && Cp == 192 && heads == 6 && hd_p == 32 && (ws == 8 || ws == 16) && Hp == 384) ? 1 : 0;
}

extern "C" int sr_tr_qkv_fwd(const SrTrQkvFwd* p, void* stream) {
    SR_REQUIRE(p && p->x && p->gamma && p->beta && p->wstream && p->q && p->qT && p->k && p->kT && p->v && p->vT, "sr_tr_qkv_fwd: null pointer");
    const SrTrQkvFwd& a = *p;
    SR_REQUIRE(tr_geo_ok(a.B, a.H, a.W, a.C, a.Cp, a.heads, a.hd_p, a.ws, a.shift, a.ldx), "sr_tr_qkv_fwd: unsupported geometry");
    SR_REQUIRE(!a.n1 || (a.ldn >= a.Cp && a.ldn % 4 == 0), "sr_tr_qkv_fwd: ldn");
    QkvFwdDev dv;
    dv.a = a;
    dv.g = make_geo(a.H, a.W, a.ws);
    static SrDeviceOnce once;
    const hipError_t e = sr_once_per_device(once, [&] { return sr_allow_lds(sr_tr_qkv_fwd_kernel, Lds<bf16>::TOTAL); });
    SR_REQUIRE(e == hipSuccess, "sr_tr_qkv_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(sr_tr_qkv_fwd_kernel, dim3(a.B * a.H * a.W / 64), dim3(256), Lds<bf16>::TOTAL, reinterpret_cast<hipStream_t>(stream), dv);
    SR_CHECK_LAUNCH("sr_tr_qkv_fwd");
    return SR_OK;
}

extern "C" int sr_tr_tail_fwd(const SrTrTailFwd* p, void* stream) {
    SR_REQUIRE(p && p->x && p->out && p->x1 && p->o && p->wstream && p->bproj && p->gamma && p->beta, "sr_tr_tail_fwd: null pointer");
    const SrTrTailFwd& a = *p;
    SR_REQUIRE(tr_geo_ok(a.B, a.H, a.W, a.C, a.Cp, a.heads, a.hd_p, a.ws, a.shift, a.ldx) && a.Hp == 384, "sr_tr_tail_fwd: unsupported geometry");
    SR_REQUIRE(!a.y || (a.ldy >= a.Cp && a.ldy % 4 == 0 && a.pool_partial && a.ca_w1 && a.ca_b1 && a.ca_w2 && a.ca_b2 && a.ca_Cr > 0 && a.ca_Cr <= 8 && a.ca_n_tiles > 0),
               "sr_tr_tail_fwd: the gated second residual needs ldy, the pool partials and the squeeze weights");
    TailFwdDev dv;
    dv.a = a;
    dv.g = make_geo(a.H, a.W, a.ws);
    static SrDeviceOnce once;
    const hipError_t e = sr_once_per_device(once, [&] { return sr_allow_lds(sr_tr_tail_fwd_kernel, Lds<bf16>::TOTAL); });
    SR_REQUIRE(e == hipSuccess, "sr_tr_tail_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(sr_tr_tail_fwd_kernel, dim3(a.B * a.H * a.W / 64), dim3(256), Lds<bf16>::TOTAL, reinterpret_cast<hipStream_t>(stream), dv);
    SR_CHECK_LAUNCH("sr_tr_tail_fwd");
    return SR_OK;
}

extern "C" int sr_tr_tail_bwd(const SrTrTailBwd* p, void* stream) {
    SR_REQUIRE(p && p->dout && p->x1 && p->gamma && p->beta && p->wstream && p->dx1 && p->n2w && p->doutw && p->gw && p->dhw && p->dOw && p->dOT && p->dx1sw && p->ln_part,
               "sr_tr_tail_bwd: null pointer");
    const SrTrTailBwd& a = *p;
    SR_REQUIRE(tr_geo_ok(a.B, a.H, a.W, a.C, a.Cp, a.heads, a.hd_p, a.ws, a.shift, a.ldx) && a.Hp == 384, "sr_tr_tail_bwd: unsupported geometry");
    SR_REQUIRE(!a.y || (a.ldy >= a.Cp && a.ldy % 4 == 0 && a.gate && a.dyc && a.dgate_part), "sr_tr_tail_bwd: the conv branch needs y, ldy, gate, dyc, dgate_part");
    TailBwdDev dv;
    dv.a = a;
    dv.g = make_geo(a.H, a.W, a.ws);
    static SrDeviceOnce once;
    const hipError_t e = sr_once_per_device(once, [&] { return sr_allow_lds(sr_tr_tail_bwd_kernel, TB_LDS); });
    SR_REQUIRE(e == hipSuccess, "sr_tr_tail_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(sr_tr_tail_bwd_kernel, dim3(a.B * a.H * a.W / 64), dim3(256), TB_LDS, reinterpret_cast<hipStream_t>(stream), dv);
    SR_CHECK_LAUNCH("sr_tr_tail_bwd");
    return SR_OK;
}

extern "C" int sr_tr_qkv_bwd(const SrTrQkvBwd* p, void* stream) {
    SR_REQUIRE(p && p->dx1 && p->x && p->dq && p->dk && p->dv && p->gamma && p->beta && p->wstream && p->dx && p->n1w && p->dqkvw && p->ln_part, "sr_tr_qkv_bwd: null pointer");
    const SrTrQkvBwd& a = *p;
    SR_REQUIRE(tr_geo_ok(a.B, a.H, a.W, a.C, a.Cp, a.heads, a.hd_p, a.ws, a.shift, a.ldx), "sr_tr_qkv_bwd: unsupported geometry");
    SR_REQUIRE(!a.dn1c || (a.ldn >= a.Cp && a.ldn % 4 == 0), "sr_tr_qkv_bwd: ldn");
    QkvBwdDev dv;
    dv.a = a;
    dv.g = make_geo(a.H, a.W, a.ws);
    static SrDeviceOnce once;
    const hipError_t e = sr_once_per_device(once, [&] { return sr_allow_lds(sr_tr_qkv_bwd_kernel, QB_LDS); });
    SR_REQUIRE(e == hipSuccess, "sr_tr_qkv_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(sr_tr_qkv_bwd_kernel, dim3(a.B * a.H * a.W / 64), dim3(256), QB_LDS, reinterpret_cast<hipStream_t>(stream), dv);
    SR_CHECK_LAUNCH("sr_tr_qkv_bwd");
    return SR_OK;
}

extern "C" int sr_tr_ca_bwd(const SrTrCaBwd* p, void* stream) {
    SR_REQUIRE(p && p->dgate_part && p->pool_partial && p->w1 && p->b1 && p->w2 && p->b2 && p->dy && p->dparam_part, "sr_tr_ca_bwd: null pointer");
    const SrTrCaBwd& a = *p;
    SR_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.C > 0 && a.C <= a.Cp && a.Cp <= 256 && a.Cp % 4 == 0 && a.Cr > 0 && a.Cr <= 8 && a.n_tiles > 0 && a.parts > 0 && a.ld >= a.Cp &&
                   a.ld % 4 == 0 && a.dparam_stride >= 2 * a.Cr * a.C + a.Cr + a.C && a.y_scale != 0.f,
               "sr_tr_ca_bwd: bad geometry");
    const int slabs = 64;  // (16: 36 us per launch at 4 x 64 x 64 -- 48 dependent in-place round trips per thread; every slab repeats the squeeze prologue)
    hipLaunchKernelGGL(sr_tr_ca_bwd_kernel, dim3(slabs, a.B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    SR_CHECK_LAUNCH("sr_tr_ca_bwd");
    return SR_OK;
}

extern "C" int sr_tr_gelu(const void* x, const void* dg, void* g, void* dx, long long n, void* stream) {
    SR_REQUIRE(x && (g || dx) && n > 0 && n % 8 == 0 && (!dx || dg), "sr_tr_gelu: bad arguments");
    const long long n8 = n / 8;
    hipLaunchKernelGGL(sr_tr_gelu_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16*>(x),
                       reinterpret_cast<const bf16*>(dg), reinterpret_cast<bf16*>(g), reinterpret_cast<bf16*>(dx), n8);
    SR_CHECK_LAUNCH("sr_tr_gelu");
    return SR_OK;
}

extern "C" int sr_tr_gelu_args(const SrTrGelu* a, void* stream) {
    SR_REQUIRE(a, "sr_tr_gelu_args: null pointer");
    return sr_tr_gelu(a->x, a->dg, a->g, a->dx, a->n, stream);
}

extern "C" int sr_tr_ln_bwd(const SrTrLnBwd* p, void* stream) {
    SR_REQUIRE(p && p->x && p->dy && p->gamma && p->dx && p->ln_part, "sr_tr_ln_bwd: null pointer");
    const SrTrLnBwd& a = *p;
    SR_REQUIRE(a.M > 0 && a.M % 64 == 0 && a.C == 180 && a.Cp == 192 && a.ld >= a.Cp && a.ld % 4 == 0, "sr_tr_ln_bwd: rows must be a multiple of 64, C 180 / Cp 192");
    hipLaunchKernelGGL(sr_tr_ln_bwd_kernel, dim3((unsigned)(a.M / 64)), dim3(256), LDS_RED, reinterpret_cast<hipStream_t>(stream), a);
    SR_CHECK_LAUNCH("sr_tr_ln_bwd");
    return SR_OK;
}
