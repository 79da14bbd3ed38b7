// Fused (shifted-)window attention half of a Swin block, one launch:
//     out = x + proj( softmax(q k^T + bias + mask) v ),   q,k,v = qkv( LayerNorm1(x) )
// (swinir.py:146-171 / hat.py:164-192 with window_partition, torch.roll and window_reverse folded
// into addressing).  The fp32 stream is read ONCE and written ONCE; q, k, v, the logits and the
// attention output never leave the CU.
//
// One workgroup = one 8x8 window (64 tokens), 6 waves; wave h OWNS HEAD h end to end:
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
// 48 KiB LDS and <= 168 VGPRs -> two workgroups (12 waves) per CU.
#include "sr_common.cuh"
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
__global__ __launch_bounds__(384, 3) void sr_swin_attn_kernel(SwinAttnDev dv) {
    constexpr int NTOK = 64, KC = 6, HEADS = 6, WS = 8, RING = 3;
    const SrSwinAttn& a = dv.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<bf16>* Aimg = reinterpret_cast<Frag<bf16>*>(smem);   // [24][64] LayerNorm1(x)
    Frag<bf16>* Oimg = Aimg + KC * 4 * NTOK;                   // [24][64] attention output
    float* red = reinterpret_cast<float*>(Oimg + KC * 4 * NTOK);  // [2][64][6] LayerNorm partials

    const int lane = threadIdx.x & 63;
    const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave = head
    const int ar = lane & 15, ag = lane >> 4;

    STAMP(0);
    // ---- window geometry
    uint32_t bimg, win, wy, wx;
    dv.div_nw.divmod(blockIdx.x, bimg, win);
    dv.div_nwx.divmod(win, wy, wx);
    int pix[4];  // image-order row of token 16m + ar (roll + partition as one gather)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int t = m * 16 + ar;
        int y = wy * WS + (t >> 3) + a.shift;
        int x = wx * WS + (t & 7) + a.shift;
        if (y >= a.H) y -= a.H;
        if (x >= a.W) x -= a.W;
        pix[m] = ((int)bimg * a.H + y) * a.W + x;
    }

    // ---- weight stream: 2 n-tiles x 6 K-chunks for each of q, k, v, proj (positions 0..23), 3-slot register ring
    const Frag<bf16>* Wq = reinterpret_cast<const Frag<bf16>*>(a.wqkv) + lane;
    const Frag<bf16>* Wp = reinterpret_cast<const Frag<bf16>*>(a.wproj) + lane;
    Frag<bf16> wr[RING][2];
    auto stream_load = [&](int t, int slot) {
        if (t < 3 * KC) {
            const int part = t / KC, c = t - part * KC;  // compile-time after unrolling
#pragma unroll
            for (int n = 0; n < 2; ++n) wr[slot][n] = Wq[((size_t)(part * 2 * HEADS + 2 * h + n) * KC + c) * 64];
        } else if (t < 4 * KC) {
#pragma unroll
            for (int n = 0; n < 2; ++n) wr[slot][n] = Wp[((size_t)(2 * h + n) * KC + (t - 3 * KC)) * 64];
        }
    };
#pragma unroll
    for (int c = 0; c < RING; ++c) stream_load(c, c);

    // ---- S0: x slice (residual registers) + LayerNorm1 -> Aimg
    f32x4 xr[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) xr[m][n] = load4(a.x + (size_t)pix[m] * a.ldx + h * 32 + n * 16 + ag * 4);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);
    {
        const float inv = 1.0f / (float)a.C;
        float mean[4], rstd[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float s = 0.f;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += xr[m][n][r];
            s = wave_sum_xor(s, 16);
            s = wave_sum_xor(s, 32);
            if (ag == 0) red[(m * 16 + ar) * HEADS + h] = s;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < HEADS; ++w) s += red[(m * 16 + ar) * HEADS + w];
            mean[m] = s * inv;
            float q = 0.f;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = h * 32 + n * 16 + ag * 4 + r;
                    const float d = c < a.C ? xr[m][n][r] - mean[m] : 0.f;
                    q += d * d;
                }
            q = wave_sum_xor(q, 16);
            q = wave_sum_xor(q, 32);
            if (ag == 0) red[NTOK * HEADS + (m * 16 + ar) * HEADS + h] = q;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float q = 0.f;
#pragma unroll
            for (int w = 0; w < HEADS; ++w) q += red[NTOK * HEADS + (m * 16 + ar) * HEADS + w];
            rstd[m] = rsqrtf(q * inv + a.eps);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const f32x4 o = (xr[m][n] - mean[m]) * rstd[m];  // gamma / beta are folded into wqkv / bqkv
                bf16x4 ob;
                ob[0] = (bf16)o[0]; ob[1] = (bf16)o[1]; ob[2] = (bf16)o[2]; ob[3] = (bf16)o[3];
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
        f32x4 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4)(0.0f);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int t = part * KC + c;
            const int slot = t % RING;
            const Frag<bf16>* arow = Aimg + (c * 4 + ag) * NTOK + ar;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const Frag<bf16> av = arow[m * 16];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (part < 2)
                        mma(wr[slot][n], av, acc[m][n]);
                    else
                        mma(av, wr[slot][n], acc[m][n]);
                }
            }
            if (t + RING < 3 * KC) stream_load(t + RING, slot);  // the proj weights are fetched after the attention
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(3 + part);
        if (part < 2) {
            const f32x4 b0 = load4(a.bqkv + part * 192 + h * 32 + ag * 4), b1 = load4(a.bqkv + part * 192 + h * 32 + 16 + ag * 4);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (part == 0)
                    qf[m] = pack2(acc[m][0] + b0, acc[m][1] + b1);
                else
                    kf[m] = pack2(acc[m][0] + b0, acc[m][1] + b1);
            }
        } else {
            const float bv0 = a.bqkv[384 + h * 32 + ar], bv1 = a.bqkv[384 + h * 32 + 16 + ar];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                vf[0][ks] = pack2(acc[2 * ks][0] + bv0, acc[2 * ks + 1][0] + bv0);
                vf[1][ks] = pack2(acc[2 * ks][1] + bv1, acc[2 * ks + 1][1] + bv1);
            }
        }
    }

    STAMP(6);
    // ---- S2: attention for head h, two query halves of 32
    const bool masked = a.shift > 0 && ((int)wy == a.H / WS - 1 || (int)wx == a.W / WS - 1);
    const f32x4* bias = reinterpret_cast<const f32x4*>(a.bias) + (size_t)h * 16 * 64 + lane;  // [h][qt][kt][lane]
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {  // one 16-query tile at a time: 4 accumulator tiles of logits live
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] = bias[(qt * 4 + kt) * 64];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) mma(kf[kt], qf[qt], s[kt]);
        if (masked) {
            const int qi = qt * 16 + ar;
            const int qlab = 3 * region3(wy * WS + (qi >> 3), a.H, WS, a.shift) + region3(wx * WS + (qi & 7), a.W, WS, a.shift);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ki = kt * 16 + ag * 4 + r;
                    const int klab = 3 * region3(wy * WS + (ki >> 3), a.H, WS, a.shift) + region3(wx * WS + (ki & 7), a.W, WS, a.shift);
                    if (klab != qlab) s[kt][r] += -100.0f;
                }
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[kt][r] - mx);
                s[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
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
            char* dst = reinterpret_cast<char*>(Oimg + (h * 4 + dt * 2 + (ag >> 1)) * NTOK + qt * 16 + ar) + (ag & 1) * 8;
            *reinterpret_cast<bf16x4*>(dst) = ob;
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the four query tiles sequential (register pressure)
    }
#pragma unroll
    for (int c = 0; c < RING; ++c) stream_load(3 * KC + c, (3 * KC + c) % RING);
    STAMP(7);
    __syncthreads();
    STAMP(8);

    // ---- S3: proj columns [32h, 32h+32) on top of the residual
    {
        const f32x4 bp0 = load4(a.bproj + h * 32 + ag * 4), bp1 = load4(a.bproj + h * 32 + 16 + ag * 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            xr[m][0] += bp0;
            xr[m][1] += bp1;
        }
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int t = 3 * KC + c;
            const int slot = t % RING;
            const Frag<bf16>* orow = Oimg + (c * 4 + ag) * NTOK + ar;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const Frag<bf16> ov = orow[m * 16];
#pragma unroll
                for (int n = 0; n < 2; ++n) mma(wr[slot][n], ov, xr[m][n]);
            }
            stream_load(t + RING, slot);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) store4(a.out + (size_t)pix[m] * a.ldx + h * 32 + n * 16 + ag * 4, xr[m][n]);
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
    SR_REQUIRE(a.B > 0 && a.H % a.ws == 0 && a.W % a.ws == 0 && a.shift >= 0 && a.shift < a.ws && a.C > 0 && a.C <= a.Cp && a.ldx >= a.Cp,
               "sr_swin_attn_fused: bad geometry");
    SwinAttnDev dv;
    dv.a = a;
    const int nwx = a.W / a.ws, nwy = a.H / a.ws;
    dv.div_nw = make_fastdiv((uint32_t)(nwx * nwy));
    dv.div_nwx = make_fastdiv((uint32_t)nwx);
    constexpr int lds = 2 * 24 * 64 * 16 + 2 * 64 * 6 * 4;  // 48 KiB + 3 KiB
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = sr_allow_lds(sr_swin_attn_kernel, lds);
        SR_REQUIRE(e == hipSuccess, "sr_swin_attn_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL(sr_swin_attn_kernel, dim3(a.B * nwx * nwy), dim3(384), lds, reinterpret_cast<hipStream_t>(stream), dv);
    SR_CHECK_LAUNCH("sr_swin_attn_fused");
    return SR_OK;
}
