// The whole SwinTransformerBlock (swinir.py:146-174) of the reference's LIGHTWEIGHT SwinIR geometry -- SwinIR.from_pretrained(light=True),
// swinir.py:418-427: embed_dim 60, 6 heads of 10, window 8, hidden 120 -- in ONE launch (C ABI v7 sr_swin_light).  The default-geometry kernel
// (sr_swin_block3.hip) is built around 192-channel fragments; at 64 padded channels the whole block is ~460 MFMAs per window and 88 KB of
// weights, so the shape is different: one 64-token window per 4-wave workgroup, every stage's weights fetched as fragments straight from L2
// (they stay L2 / L1 resident: all workgroups read the same 88 KB), heads padded 10 -> 16 features (not 32: the QK^T / PV contractions run
// on half-filled k-groups instead of a 3x padded projection), activations in K-group-major bf16 LDS images, fp32 residual stream /
// LayerNorm statistics / softmax as everywhere else.  LayerNorm affines are folded into the following Linear at pack time.
// T = bf16: bf16 operands, three workgroups per CU.  T = bf3 (compute type SR_BF16X3, precision "fp32x3" = what inference() runs by default): every operand a
// hi + lo bf16 pair, every product hi*hi + hi*lo + lo*hi (fp32-class accuracy), 32-byte image cells (one workgroup per CU), erf GELU.
//   wave w:  QKV n-tiles w, w + 4, ... (one n-tile = one (q|k|v, head));  attention: query tile w of all six heads;  proj / fc2: output
//   channels [16 w, +16);  fc1: hidden tiles w, w + 4.
#include "sr_swin_stream.h"

namespace {

constexpr int L_HEADS = 6;  // padded sizes: 64 channels, head features 10 -> 16, hidden 128
constexpr int L_VT_LD = 144;  // bytes per V^T row (64 keys bf16 + 16 B): 16 rows hit 16 different 4-bank groups
// LDS (bytes): [A image [8][64] cells, later the O image [12][64]] | Q [6][2][64] | K [6][2][64] | V^T [6][16] rows | partial sums; the hidden image
// [16][64] reuses Q | K.  52 KB: three workgroups per CU, i.e. all 648 windows of the 8-tile bench shape resident at once.
template <typename T>
struct LL {
    static constexpr int CELL = (int)sizeof(Frag<T>), PLANES = CELL / 16;  // V^T: one [6][16] row set per bf16 plane (hi, lo)
    static constexpr int OFF_Q = 12 * 64 * CELL, OFF_K = OFF_Q + 12 * 64 * CELL, OFF_V = OFF_K + 12 * 64 * CELL, OFF_O = 0, VPLANE = 6 * 16 * L_VT_LD,
                         OFF_RED = OFF_V + PLANES * VPLANE, TOTAL = OFF_RED + 64 * 4 * 2 * 4;
};
static_assert(3 * LL<bf16>::TOTAL <= 160 * 1024, "bf16: three workgroups per CU");
static_assert(LL<bf3>::TOTAL <= 160 * 1024, "bf16x3: one workgroup per CU");

struct SwinLightDev {
    SrSwinLight a;
    FastDiv div_nw, div_nwx;
};

SR_DEV int region8(int v, int size, int shift) { return v < size - 8 ? 0 : (v < size - shift ? 1 : 2); }

template <typename T>
SR_DEV Frag<T> wfrag(const void* w, int nt, int kc, int c, int lane) {  // fragment (n-tile nt, k-chunk c) of a matrix packed with kc chunks per tile (bf3: per lane 8 hi | 8 lo)
    return *reinterpret_cast<const Frag<T>*>(reinterpret_cast<const char*>(w) + ((size_t)(nt * kc + c) * 64 + lane) * sizeof(Frag<T>));
}
// four consecutive tokens of one V^T row: store / the 8-key fragment read (two 4-key pieces, 16 keys apart)
SR_DEV void vt_store(char* vt, int plane_bytes, const f32x4& v, Frag<bf16>*) { *reinterpret_cast<bf16x4*>(vt) = cvt4(v); }
SR_DEV void vt_store(char* vt, int plane_bytes, const f32x4& v, Frag<bf3>*) {
    const bf16x4 h = cvt4(v);
    *reinterpret_cast<bf16x4*>(vt) = h;
    *reinterpret_cast<bf16x4*>(vt + plane_bytes) = cvt4(v - widen4(h));
}
SR_DEV bf16x8 vt_pair(const char* vp) { return __builtin_shufflevector(*reinterpret_cast<const bf16x4*>(vp), *reinterpret_cast<const bf16x4*>(vp + 32), 0, 1, 2, 3, 4, 5, 6, 7); }
SR_DEV Frag<bf16> vt_load(const char* vp, int plane_bytes, Frag<bf16>*) {
    Frag<bf16> f;
    f.v = vt_pair(vp);
    return f;
}
SR_DEV Frag<bf3> vt_load(const char* vp, int plane_bytes, Frag<bf3>*) {
    Frag<bf3> f;
    f.hi = vt_pair(vp);
    f.lo = vt_pair(vp + plane_bytes);
    return f;
}

// mean / rstd of the 64 token rows held as v[m] (token 16 m + ar, channels 16 w + 4 ag ..) over the C real channels
SR_DEV void light_ln(const f32x4 (&v)[4], float* red, int w, int ar, int ag, int C, float eps, float (&mean)[4], float (&rstd)[4]) {
    float q1[4], q2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        q1[m] = (v[m][0] + v[m][1]) + (v[m][2] + v[m][3]);
        q2[m] = (v[m][0] * v[m][0] + v[m][1] * v[m][1]) + (v[m][2] * v[m][2] + v[m][3] * v[m][3]);
    }
    const float s1 = rows_reduce_scatter4(q1[0], q1[1], q1[2], q1[3]);
    const float s2 = rows_reduce_scatter4(q2[0], q2[1], q2[2], q2[3]);
    *reinterpret_cast<float2*>(red + ((ag * 16 + ar) * 4 + w) * 2) = make_float2(s1, s2);
    __syncthreads();
    const float inv = 1.0f / (float)C;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4 pa = *reinterpret_cast<const f32x4*>(red + (m * 16 + ar) * 8), pb = *reinterpret_cast<const f32x4*>(red + (m * 16 + ar) * 8 + 4);
        mean[m] = (pa[0] + pa[2] + pb[0] + pb[2]) * inv;
        rstd[m] = rsqrtf(fmaxf((pa[1] + pa[3] + pb[1] + pb[3]) * inv - mean[m] * mean[m], 0.f) + eps);
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256, sizeof(Frag<T>) == 16 ? 3 : 1) void sr_swin_light_kernel(SwinLightDev dv) {
    const SrSwinLight& a = dv.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<T>* Aimg = reinterpret_cast<Frag<T>*>(smem);
    Frag<T>* Qimg = reinterpret_cast<Frag<T>*>(smem + LL<T>::OFF_Q);
    Frag<T>* Kimg = reinterpret_cast<Frag<T>*>(smem + LL<T>::OFF_K);
    char* VT = smem + LL<T>::OFF_V;
    Frag<T>* Oimg = reinterpret_cast<Frag<T>*>(smem + LL<T>::OFF_O);
    Frag<T>* Himg = Qimg;
    float* red = reinterpret_cast<float*>(smem + LL<T>::OFF_RED);
    constexpr int VPL = LL<T>::VPLANE;
    Frag<T>* const tag = nullptr;

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, ar = lane & 15, ag = lane >> 4;
    uint32_t bimg, win, wy, wx;
    dv.div_nw.divmod((uint32_t)blockIdx.x, bimg, win);
    dv.div_nwx.divmod(win, wy, wx);
    const int shift_y = a.y_mode == SR_Y_ROLL ? a.shift : 0;
    const int ch0 = 16 * w + 4 * ag;
    size_t prow[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int t = 16 * m + ar;
        int y = wy * 8 + (t >> 3) + shift_y, x = wx * 8 + (t & 7) + a.shift;
        if (y >= a.H) y -= a.H;
        if (x >= a.W) x -= a.W;
        prow[m] = ((size_t)bimg * a.H + y) * a.W + x;
    }
    // ---- x (the shortcut) and LayerNorm1 -> image
    f32x4 x1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) x1[m] = *reinterpret_cast<const f32x4*>(a.x + prow[m] * a.ldx + ch0);
    float mean[4], rstd[4];
    light_ln(x1, red, w, ar, ag, a.C, a.eps, mean, rstd);
    f32x4 cmask;  // 1 for real channels, 0 for the padding (60..63)
#pragma unroll
    for (int r = 0; r < 4; ++r) cmask[r] = ch0 + r < a.C ? 1.0f : 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 nv;
#pragma unroll
        for (int r = 0; r < 4; ++r) nv[r] = (x1[m][r] - mean[m]) * rstd[m] * cmask[r];
        st_half(Aimg + (2 * w + (ag >> 1)) * 64 + 16 * m + ar, ag & 1, nv);
    }
    __syncthreads();
    auto afrag = [&](const Frag<T>* img, int c, int m) { return img[(4 * c + ag) * 64 + 16 * m + ar]; };

    // ---- QKV: n-tile nt = (part, head); q, k as [token][feature] images, v transposed
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int nt = w + 4 * i;
        if (nt >= 18) break;
        const int part = nt / L_HEADS, head = nt - part * L_HEADS;
        const Frag<T> b0 = wfrag<T>(a.wqkv, nt, 2, 0, lane), b1 = wfrag<T>(a.wqkv, nt, 2, 1, lane);
        if (part < 2) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bqkv + nt * 16 + 4 * ag);
            Frag<T>* dst = (part == 0 ? Qimg : Kimg) + (head * 2 + (ag >> 1)) * 64;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 acc = mma_c(b0, afrag(Aimg, 0, m), bias);  // lane = token, registers = 4 features
                mma(b1, afrag(Aimg, 1, m), acc);
                st_half(dst + 16 * m + ar, ag & 1, acc);
            }
        } else {
            const float bias = a.bqkv[nt * 16 + ar];  // lane = feature
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 acc = mma_c(afrag(Aimg, 0, m), b0, (f32x4)(bias));  // lane = feature ar, registers = tokens 16 m + 4 ag ..
                mma(afrag(Aimg, 1, m), b1, acc);
                vt_store(VT + (head * 16 + ar) * L_VT_LD + (16 * m + 4 * ag) * 2, VPL, acc, tag);
            }
        }
    }
    __syncthreads();

    // ---- attention: query tile w of every head; S^T = K Q^T with the bias tile as C, softmax in registers, O^T = V^T P^T
    {
        const int nwx = a.W >> 3, nwy = a.H >> 3;
        const bool ymask = a.y_mode != SR_Y_STRIP;
        const bool masked = a.shift > 0 && ((ymask && (int)wy == nwy - 1) || (int)wx == nwx - 1);
        const int qi = 16 * w + ar;
        const int qlab = (ymask ? 3 * region8(wy * 8 + (qi >> 3), a.H, a.shift) : 0) + region8(wx * 8 + (qi & 7), a.W, a.shift);
        const bool lo = ag < 2;  // head features 16: k-groups 2, 3 of the 32-deep contraction are padding
#pragma unroll
        for (int head = 0; head < L_HEADS; ++head) {
            const Frag<T> qf = frag_keep_if(lo, Qimg[(head * 2 + (ag & 1)) * 64 + 16 * w + ar]);
            f32x4 s[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const Frag<T> kf = frag_keep_if(lo, Kimg[(head * 2 + (ag & 1)) * 64 + 16 * kt + ar]);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(a.bias + (((size_t)(head * 4 + w) * 4 + kt) * 64 + lane) * 4);
                s[kt] = mma_c(kf, qf, bt);  // S^T[key 16 kt + 4 ag + r][query ar]
                if (masked) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ki = 16 * kt + 4 * ag + r;
                        const int klab = (ymask ? 3 * region8(wy * 8 + (ki >> 3), a.H, a.shift) : 0) + region8(wx * 8 + (ki & 7), a.W, a.shift);
                        if (klab != qlab) s[kt][r] += -100.0f;
                    }
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
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][r] = __expf(s[kt][r] - mx);
                    sum += s[kt][r];
                }
            sum = wave_sum_xor(sum, 16);
            sum = wave_sum_xor(sum, 32);
            const float inv = 1.0f / sum;
            f32x4 o = (f32x4)(0.0f);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const Frag<T> pf = pack2<T>(s[2 * ks] * inv, s[2 * ks + 1] * inv);  // row = query ar, k = keys 32 ks + 4 ag + r | + 16
                mma(vt_load(VT + (head * 16 + ar) * L_VT_LD + (32 * ks + 4 * ag) * 2, VPL, tag), pf, o);  // O^T[d = 4 ag + r][query ar]
            }
            st_half(Oimg + (head * 2 + (ag >> 1)) * 64 + 16 * w + ar, ag & 1, o);
        }
    }
    __syncthreads();

    // ---- projection (K = 6 heads x 16 features = 3 chunks) + shortcut
    {
        const f32x4 bp = *reinterpret_cast<const f32x4*>(a.bproj + ch0);
        const Frag<T> b0 = wfrag<T>(a.wproj, w, 3, 0, lane), b1 = wfrag<T>(a.wproj, w, 3, 1, lane), b2 = wfrag<T>(a.wproj, w, 3, 2, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 acc = mma_c(b0, afrag(Oimg, 0, m), x1[m] + bp);
            mma(b1, afrag(Oimg, 1, m), acc);
            mma(b2, afrag(Oimg, 2, m), acc);
            x1[m] = acc * cmask;  // (pad channels stay exactly 0: zero weight rows, zero bias pads)
        }
    }
    // ---- LayerNorm2 -> image, over the O image: the barriers inside light_ln order every wave's projection reads before the writes
    light_ln(x1, red, w, ar, ag, a.C, a.eps, mean, rstd);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 nv;
#pragma unroll
        for (int r = 0; r < 4; ++r) nv[r] = (x1[m][r] - mean[m]) * rstd[m] * cmask[r];
        st_half(Aimg + (2 * w + (ag >> 1)) * 64 + 16 * m + ar, ag & 1, nv);
    }
    __syncthreads();
    // ---- fc1 + GELU -> hidden image (over the dead Q | K images), fc2 + shortcut
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int nt = w + 4 * i;
        const f32x4 b1v = *reinterpret_cast<const f32x4*>(a.b1 + nt * 16 + 4 * ag);
        const Frag<T> b0 = wfrag<T>(a.w1, nt, 2, 0, lane), b1 = wfrag<T>(a.w1, nt, 2, 1, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 acc = mma_c(b0, afrag(Aimg, 0, m), b1v);
            mma(b1, afrag(Aimg, 1, m), acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = gelu_op<T>(acc[r]);
            st_half(Himg + (2 * nt + (ag >> 1)) * 64 + 16 * m + ar, ag & 1, acc);
        }
    }
    __syncthreads();
    {
        const f32x4 b2v = *reinterpret_cast<const f32x4*>(a.b2 + ch0);
        Frag<T> bw[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) bw[c] = wfrag<T>(a.w2, w, 4, c, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 acc = x1[m] + b2v;
#pragma unroll
            for (int c = 0; c < 4; ++c) mma(bw[c], afrag(Himg, c, m), acc);
            *reinterpret_cast<f32x4*>(a.out + prow[m] * a.ldx + ch0) = acc * cmask;
        }
    }
}

}  // namespace

extern "C" int sr_swin_light_supported(int C, int Cp, int heads, int hd, int ws, int hidden, int compute_dtype) {
    return ((compute_dtype == SR_BF16 || compute_dtype == SR_BF16X3) && C <= 64 && C > 48 && Cp == 64 && heads == 6 && hd <= 16 && ws == 8 && hidden <= 128 && hidden > 64) ? 1 : 0;
}

extern "C" int sr_swin_light(const SrSwinLight* p, void* stream) {
    SR_REQUIRE(p && p->x && p->out && p->wqkv && p->bqkv && p->wproj && p->bproj && p->w1 && p->b1 && p->w2 && p->b2 && p->bias, "sr_swin_light: null pointer");
    const SrSwinLight& a = *p;
    SR_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.H % 8 == 0 && a.W % 8 == 0 && a.shift >= 0 && a.shift < 8 && a.C > 48 && a.C <= 64 && a.ldx >= 64 && a.ldx % 4 == 0 &&
                   a.y_mode >= SR_Y_ROLL && a.y_mode <= SR_Y_STRIP_LAST && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0,
               "sr_swin_light: bad geometry");
    SR_REQUIRE((long long)a.B * a.H * a.W < (1ll << 31), "sr_swin_light: more than 2^31 tokens");
    SwinLightDev dv;
    dv.a = a;
    const int nwx = a.W / 8, nwy = a.H / 8;
    dv.div_nw = make_fastdiv((uint32_t)(nwx * nwy));
    dv.div_nwx = make_fastdiv((uint32_t)nwx);
    SR_REQUIRE(a.compute_dtype == SR_BF16 || a.compute_dtype == SR_BF16X3, "sr_swin_light: compute_dtype must be SR_BF16 or SR_BF16X3");
    if (a.compute_dtype == SR_BF16X3) {
        static SrDeviceOnce once3;
        const hipError_t e = sr_once_per_device(once3, [&] { return sr_allow_lds(sr_swin_light_kernel<bf3>, LL<bf3>::TOTAL); });
        SR_REQUIRE(e == hipSuccess, "sr_swin_light: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(sr_swin_light_kernel<bf3>, dim3(a.B * nwx * nwy), dim3(256), LL<bf3>::TOTAL, reinterpret_cast<hipStream_t>(stream), dv);
        SR_CHECK_LAUNCH("sr_swin_light");
        return SR_OK;
    }
    static SrDeviceOnce once;
    const hipError_t e = sr_once_per_device(once, [&] { return sr_allow_lds(sr_swin_light_kernel<bf16>, LL<bf16>::TOTAL); });
    SR_REQUIRE(e == hipSuccess, "sr_swin_light: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(sr_swin_light_kernel<bf16>, dim3(a.B * nwx * nwy), dim3(256), LL<bf16>::TOTAL, reinterpret_cast<hipStream_t>(stream), dv);
    SR_CHECK_LAUNCH("sr_swin_light");
    return SR_OK;
}
