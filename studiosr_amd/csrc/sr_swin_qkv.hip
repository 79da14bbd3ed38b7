// LayerNorm1 + QKV projection in front of a separate window-attention kernel, stream form (C ABI v6 sr_swin_qkv):
//     q, k, v = qkv( LayerNorm1(x) )  in the layouts sr_window_attention reads            (hat.py:164-176, 55-83; swinir.py:78-90, 146-160)
// for the geometries the one-kernel block does not cover (HAT: 16 x 16 windows).  Replaces sr_gemm's SR_EPI_QKV launch (three column
// slices x 64-row tiles, each re-reading and re-normalising its x tile: 19-21 us at HAT's 4 x 64 x 64 tokens) by the first stage of
// sr_swin_block3.hip: one workgroup = 64 consecutive window-order tokens, x arrives once as full rows by LDS-DMA (roll + window_partition
// = the row gather), LayerNorm in the accumulator layout into the K-group-major bf16 image, then three passes of six uniform steps
// (pass p: heads 2p, 2p+1; wave w: d-half w & 1 of head 2p + (w >> 1): q, k as [token][feature] tiles, v transposed) from ONE
// 18-slot weight stream.  All three biases ride on the constant-one pad channels 180 / 181 of the image; the attention scale is in the q rows.
// No barrier after the LayerNorm: the passes only read the image and write global memory.
#include <type_traits>
#include "sr_swin_stream.h"

namespace {

constexpr int QKV_SLOTS = 18;
#ifndef SR_QKV_DIST
#define SR_QKV_DIST 4
#endif

struct SwinQkvDev {
    SrSwinQkv a;
    FastDiv div_parts_img, div_parts_win, div_nwx;
    int ws_log2, nw;  // windows per image
};

// T = bf3 (SR_BF16X3, precision "fp32x3"): split operands, q / k / v^T leave as fp32 in the row-major / OCA layouts (what the fp32 attention kernels read).
template <typename T>
__global__ __launch_bounds__(256, sizeof(Frag<T>) == 16 ? 2 : 1) void sr_swin_qkv_kernel(SwinQkvDev dv) {
    constexpr bool X3 = sizeof(Frag<T>) == 32;
    using TO = typename std::conditional<X3, float, bf16>::type;  // element type of q / k / v^T
    auto put4 = [](TO* dst, const f32x4& v) {
#ifdef SR_EXP_NOSTORE
        if (v[0] != 1.2345f) return;
#endif
        if constexpr (X3)
            *reinterpret_cast<f32x4*>(dst) = v;
        else
            *reinterpret_cast<bf16x4*>(dst) = cvt4(v);
    };
    const SrSwinQkv& a = dv.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<T>* Aimg = reinterpret_cast<Frag<T>*>(smem);
    float* red = reinterpret_cast<float*>(smem + Lds<T>::RED_OFF);

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane0 = threadIdx.x & 63;
    int lane = lane0, ar = lane & 15, ag = lane >> 4;
    auto relane = [&]() {
        lane = lane0;
        asm volatile("" : "+v"(lane));
        ar = lane & 15;
        ag = lane >> 4;
    };

    uint32_t bimg, rem, win, part, wy, wx;
    dv.div_parts_img.divmod((uint32_t)blockIdx.x, bimg, rem);
    dv.div_parts_win.divmod(rem, win, part);
    dv.div_nwx.divmod(win, wy, wx);
    const int shift_y = a.y_mode == SR_Y_ROLL ? a.shift : 0;
    const int wsl = dv.ws_log2, wsm = a.ws - 1;
    auto pixel_row = [&](int t) {  // image-order row of token t of this workgroup (roll + window_partition as one gather)
        const int tw = (int)part * NTOK + t;
        int y = ((int)wy << wsl) + (tw >> wsl) + shift_y;
        int x = ((int)wx << wsl) + (tw & wsm) + a.shift;
        if (y >= a.H) y -= a.H;
        if (x >= a.W) x -= a.W;
        return ((int)bimg * a.H + y) * a.W + x;
    };

    // ---- x: 16 full token rows per wave by LDS-DMA, then the first weight slots
    {
        const unsigned tile_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = 16 * w + i;
            dma_row48(a.x + (size_t)pixel_row(t) * a.ldx, __builtin_amdgcn_readfirstlane(tile_lds + t * XS), lane);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    WStream<T, QKV_SLOTS, SR_QKV_DIST> ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wstream), 0, QKV_SLOTS * 12 * 64 * (int)sizeof(Frag<T>), 0x00020000);
    ws.wave_frag = w * 3;
#pragma unroll
    for (int s0 = 0; s0 < SR_QKV_DIST; ++s0) ws.load(s0, lane);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * SR_QKV_DIST) : "memory");  // the 16 rows have landed; the weight slots issued after them may still fly
    BLOCK_SYNC();
    {
        f32x4 x1[4][3];  // [m][n]: token 16 m + ar, channels 48 w + 16 n + 4 ag .. +3
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const char* xm = smem + (m * 16 + ar) * XS + (w * 48 + ag * 4) * 4;
#pragma unroll
            for (int n = 0; n < 3; ++n) x1[m][n] = *reinterpret_cast<const f32x4*>(xm + n * 64);
        }
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
            const float s1 = rows_reduce_scatter4(q1[0], q1[1], q1[2], q1[3]);
            const float s2 = rows_reduce_scatter4(q2[0], q2[1], q2[2], q2[3]);
            *reinterpret_cast<float2*>(red + ((ag * 16 + ar) * 4 + w) * 2) = make_float2(s1, s2);
        }
        BLOCK_SYNC();  // (also: every wave has read its part of the x tile, which the image overwrites)
        const bool one_lane = (w == ONE_C / 48) && (ag == (ONE_C % 16) / 4);
        const float inv = 1.0f / (float)a.C;
        f32x4 gm[3], bt[3];
        if (a.n1) {
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                gm[n] = *reinterpret_cast<const f32x4*>(a.n1_gamma + w * 48 + ag * 4 + n * 16);
                bt[n] = *reinterpret_cast<const f32x4*>(a.n1_beta + w * 48 + ag * 4 + n * 16);
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 pa = *reinterpret_cast<const f32x4*>(red + (m * 16 + ar) * 8), pb = *reinterpret_cast<const f32x4*>(red + (m * 16 + ar) * 8 + 4);
            const float mean = (pa[0] + pa[2] + pb[0] + pb[2]) * inv;
            const float rstd = rsqrtf(fmaxf((pa[1] + pa[3] + pb[1] + pb[3]) * inv - mean * mean, 0.f) + a.eps);
            const float nmr = -mean * rstd;
            if (a.n1) {  // LayerNorm1 with its affine as a side output in image order (the CAB's input: no sr_layernorm launch for a group's first block)
                const size_t noff = (size_t)pixel_row(m * 16 + ar) * a.ldn + w * 48 + ag * 4;
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    f32x4 nv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) nv[r] = __builtin_fmaf((x1[m][n][r] - mean) * rstd, gm[n][r], bt[n][r]);  // pad channels: gamma = beta = 0
                    if constexpr (X3)
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.n1) + noff + n * 16) = nv;
                    else
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.n1) + noff + n * 16) = cvt4(nv);
                }
            }
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

    auto loada = [&](int c, int h, Frag<T> (&av)[2]) {
        const Frag<T>* arow = Aimg + (c * 4 + ag) * NTOK + h * 32 + ar;
        av[0] = arow[0];
        av[1] = arow[16];
    };
    const int hh = w >> 1, half = w & 1;
    const int ntok_log2 = 2 * wsl;
    const size_t bwin = (size_t)bimg * dv.nw + win;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        relane();
        f32x4 acc[4][3];
        ws.template run<6>(6 * p, lane, loada, [&](int c, int h, Frag<T> (&b)[3], Frag<T> (&av)[2]) {
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
        const int head = 2 * p + hh;
        const size_t bh = bwin * a.heads + head;
        if (!X3 && a.frag_order) {
            // fragment order (SrWindowAttn.qkv_frag): q / k cell [tile = token >> 4][g = feature >> 3][i = token & 15][8]; v^T cell
            // [64-key block][d tile][32-key step][g = (key >> 2) & 3][i = d & 15][8] with element (key >> 4 & 1) * 4 + (key & 3)
            const size_t base = (bh << ntok_log2) * a.hd_p;
            bf16* qd = reinterpret_cast<bf16*>(a.q) + base + (size_t)(((part * 4) * 4 + 2 * half + (ag >> 1)) * 16 + ar) * 8 + 4 * (ag & 1);
            bf16* kd = reinterpret_cast<bf16*>(a.k) + base + (size_t)(((part * 4) * 4 + 2 * half + (ag >> 1)) * 16 + ar) * 8 + 4 * (ag & 1);
            bf16* vd = reinterpret_cast<bf16*>(a.vt) + base + (size_t)(((part * 2 + half) * 2) * 64 + ag * 16 + ar) * 8;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                *reinterpret_cast<bf16x4*>(qd + m * 4 * 16 * 8) = cvt4(acc[m][0]);
                *reinterpret_cast<bf16x4*>(kd + m * 4 * 16 * 8) = cvt4(acc[m][1]);
                *reinterpret_cast<bf16x4*>(vd + (m >> 1) * 64 * 8 + (m & 1) * 4) = cvt4(acc[m][2]);
            }
            __builtin_amdgcn_sched_barrier(0);
            continue;
        }
        TO* qd = reinterpret_cast<TO*>(a.q) + (((bh << ntok_log2) + part * NTOK + ar) * a.hd_p + 16 * half + 4 * ag);
#pragma unroll
        for (int m = 0; m < 4; ++m) put4(qd + m * 16 * a.hd_p, acc[m][0]);
        if (a.oca_pad == 0) {
            TO* kd = reinterpret_cast<TO*>(a.k) + (((bh << ntok_log2) + part * NTOK + ar) * a.hd_p + 16 * half + 4 * ag);
            TO* vd = reinterpret_cast<TO*>(a.vt) + (((bh * a.hd_p + 16 * half + ar) << ntok_log2) + part * NTOK + 4 * ag);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                put4(kd + m * 16 * a.hd_p, acc[m][1]);
                put4(vd + m * 16, acc[m][2]);
            }
        } else {
            // overlapping cross attention (hat.py:247-264): k -> zero-bordered image order [B][H+2e][W+2e][heads][hd_p], v -> transposed zero-bordered
            // planes [B][heads][hd_p][(H+2e)(W+2e)] (4 consecutive tokens = 4 consecutive x; e = oca_pad); shift is 0 there
            const int e = a.oca_pad, Wb = a.W + 2 * e;
            const size_t plane = (size_t)(a.H + 2 * e) * Wb;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                {
                    const int tw = (int)part * NTOK + m * 16 + ar;
                    const int y = ((int)wy << wsl) + (tw >> wsl), x = ((int)wx << wsl) + (tw & wsm);
                    const size_t off = ((((size_t)bimg * (a.H + 2 * e) + y + e) * Wb + x + e) * a.heads + head) * a.hd_p + 16 * half + 4 * ag;
                    put4(reinterpret_cast<TO*>(a.k) + off, acc[m][1]);
                }
                {
                    const int tw = (int)part * NTOK + m * 16 + 4 * ag;
                    const int y = ((int)wy << wsl) + (tw >> wsl), x = ((int)wx << wsl) + (tw & wsm);
                    const size_t off = (((size_t)bimg * a.heads + head) * a.hd_p + 16 * half + ar) * plane + (size_t)(y + e) * Wb + x + e;
                    put4(reinterpret_cast<TO*>(a.vt) + off, acc[m][2]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace

extern "C" int sr_swin_qkv_supported(int C, int Cp, int heads, int hd_p, int ws, int compute_dtype) {
    return ((compute_dtype == SR_BF16 || compute_dtype == SR_BF16X3) && C == 180 && Cp == 192 && heads == 6 && hd_p == 32 && (ws == 8 || ws == 16)) ? 1 : 0;
}

extern "C" int sr_swin_qkv(const SrSwinQkv* p, void* stream) {
    SR_REQUIRE(p && p->x && p->q && p->k && p->vt && p->wstream, "sr_swin_qkv: null pointer");
    const SrSwinQkv& a = *p;
    SR_REQUIRE(sr_swin_qkv_supported(a.C, a.Cp, a.heads, a.hd_p, a.ws, a.compute_dtype), "sr_swin_qkv: unsupported geometry / compute type (use sr_gemm with SR_EPI_QKV)");
    SR_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.H % a.ws == 0 && a.W % a.ws == 0 && a.shift >= 0 && a.shift < a.ws && a.ldx >= a.Cp &&
                   a.y_mode >= SR_Y_ROLL && a.y_mode <= SR_Y_STRIP_LAST,
               "sr_swin_qkv: bad geometry");
    SR_REQUIRE(a.ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0, "sr_swin_qkv: the stream rows must be 16-byte aligned (ldx a multiple of 4, x 16-byte aligned)");
    SR_REQUIRE(!a.frag_order || (a.ws == 16 && a.oca_pad == 0 && a.hd_p == 32 && a.compute_dtype == SR_BF16), "sr_swin_qkv: frag_order needs 16 x 16 windows, oca_pad == 0 and SR_BF16");
    SR_REQUIRE(a.oca_pad == 0 || (a.oca_pad > 0 && a.oca_pad % 4 == 0 && a.shift == 0 && a.y_mode == SR_Y_ROLL), "sr_swin_qkv: OCA layouts need shift 0 and a border that is a multiple of 4");
    SR_REQUIRE((long long)a.B * a.H * a.W < (1ll << 31), "sr_swin_qkv: more than 2^31 tokens");
    SR_REQUIRE(!a.n1 || (a.n1_gamma && a.n1_beta && a.ldn >= a.Cp && a.ldn % 4 == 0 && a.oca_pad == 0), "sr_swin_qkv: the LayerNorm side output needs n1_gamma, n1_beta, ldn (and oca_pad == 0)");
    SwinQkvDev dv;
    dv.a = a;
    const int nwx = a.W / a.ws, nwy = a.H / a.ws, parts = a.ws * a.ws / 64;
    dv.div_parts_img = make_fastdiv((uint32_t)(nwx * nwy * parts));
    dv.div_parts_win = make_fastdiv((uint32_t)parts);
    dv.div_nwx = make_fastdiv((uint32_t)nwx);
    dv.ws_log2 = a.ws == 8 ? 3 : 4;
    dv.nw = nwx * nwy;
    if (a.compute_dtype == SR_BF16X3) {
        static SrDeviceOnce once3;
        const hipError_t e = sr_once_per_device(once3, [&] { return sr_allow_lds(sr_swin_qkv_kernel<bf3>, Lds<bf3>::TOTAL); });
        SR_REQUIRE(e == hipSuccess, "sr_swin_qkv: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(sr_swin_qkv_kernel<bf3>, dim3(a.B * nwx * nwy * parts), dim3(256), Lds<bf3>::TOTAL, reinterpret_cast<hipStream_t>(stream), dv);
        SR_CHECK_LAUNCH("sr_swin_qkv");
        return SR_OK;
    }
    static SrDeviceOnce attr_once;
    {
        const hipError_t e = sr_once_per_device(attr_once, [&] { return sr_allow_lds(sr_swin_qkv_kernel<bf16>, Lds<bf16>::TOTAL); });
        SR_REQUIRE(e == hipSuccess, "sr_swin_qkv: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(sr_swin_qkv_kernel<bf16>, dim3(a.B * nwx * nwy * parts), dim3(256), Lds<bf16>::TOTAL, reinterpret_cast<hipStream_t>(stream), dv);
    SR_CHECK_LAUNCH("sr_swin_qkv");
    return SR_OK;
}
