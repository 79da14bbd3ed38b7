// Window attention: softmax(q k^T + bias[head] + shift-mask) v for every (window, head).
//
// One wave owns one (window, head, block of QT*16 queries).  Nothing goes through LDS:
//   S^T = K Q^T   operands are row fragments of K and Q fetched straight from the [.., tok, hd_p]
//                 buffers the QKV projection wrote (16/32-byte loads, one per lane);
//   the S^T accumulator holds, per lane, ONE query (column l & 15) and 4 keys per 16-key tile, so
//   bias add, mask, max, exp and sum are in-register plus two cross-lane steps (xor 16, 32);
//   O^T = V^T P^T the P^T operand is the S^T accumulator itself (converted in place: the K order
//                 inside a 32-key step is permuted identically on both operands), and the V^T
//                 operand comes from the transposed [.., hd_p, tok] buffer with two 8-byte loads;
//   the O^T accumulator gives every lane 4 consecutive features of one query -> 8/16-byte stores.
// The -100 shift mask is recomputed from window coordinates (same labels as the reference's
// calculate_mask) instead of being read from memory.
#include "sr_wattn_lds_body.h"
#include "sr_wattn_qkv_body.h"

namespace {

// KT = key tiles (ntok/16), QT = query tiles per wave, DC = hd_p/32 chunks of the head dim.
template <typename TC, int KT, int QT, int DC>
__global__ __launch_bounds__(256) void sr_window_attn_kernel(SrWindowAttn a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NTOK = KT * 16;
    constexpr int QB = KT / QT;  // query blocks per (window, head)
    const int item = blockIdx.x * 4 + wave;
    const int n_items = a.n_bwin * a.heads * QB;
    if (item >= n_items) return;  // wave-uniform; no barriers in this kernel
    const int qb = item % QB;
    const int bh = item / QB;
    const int head = bh % a.heads;
    const int bwin = bh / a.heads;
    const int hd_p = DC * 32;
    const int lr = lane & 15, lg = lane >> 4;

    const TC* q = reinterpret_cast<const TC*>(a.q) + (size_t)bh * NTOK * hd_p;
    const TC* k = reinterpret_cast<const TC*>(a.k) + (size_t)bh * NTOK * hd_p;
    const TC* vt = reinterpret_cast<const TC*>(a.vt) + (size_t)bh * NTOK * hd_p;

    // ---- S^T[key][query] = sum_d K[key][d] Q[query][d]
    Frag<TC> qf[QT][DC];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int c = 0; c < DC; ++c) qf[t][c] = *reinterpret_cast<const Frag<TC>*>(q + (size_t)((qb * QT + t) * 16 + lr) * hd_p + c * 32 + lg * 8);

    f32x4 s[KT][QT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int t = 0; t < QT; ++t) s[kt][t] = (f32x4)(0.0f);
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            const Frag<TC> kf = *reinterpret_cast<const Frag<TC>*>(k + (size_t)(kt * 16 + lr) * hd_p + c * 32 + lg * 8);
#pragma unroll
            for (int t = 0; t < QT; ++t) mma(kf, qf[t][c], s[kt][t]);
        }
    }

    // ---- + relative position bias, + shift mask
    const float* bias = a.bias + (size_t)head * NTOK * NTOK;
    const int nwx = a.W / a.ws;
    const int nwy = a.H / a.ws;
    const int win = bwin % (nwx * nwy);
    const int wy = win / nwx, wx = win - wy * nwx;
    const bool ymask = a.y_mode != SR_Y_STRIP;  // a middle strip of a larger image has no wrapped window row
    const bool masked = a.shift > 0 && ((ymask && wy == nwy - 1) || wx == nwx - 1);
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = (qb * QT + t) * 16 + lr;
        int qlab = 0;
        if (masked) {
            const int i = qi / a.ws, j = qi - i * a.ws;
            qlab = (ymask ? 3 * region(wy * a.ws + i, a.H, a.ws, a.shift) : 0) + region(wx * a.ws + j, a.W, a.ws, a.shift);
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int k0 = kt * 16 + lg * 4;
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + (size_t)qi * NTOK + k0);
            s[kt][t] += b;
            if (masked) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ki = k0 + r;
                    const int i = ki / a.ws, j = ki - i * a.ws;
                    const int klab = (ymask ? 3 * region(wy * a.ws + i, a.H, a.ws, a.shift) : 0) + region(wx * a.ws + j, a.W, a.ws, a.shift);
                    if (klab != qlab) s[kt][t][r] += -100.0f;
                }
            }
        }
    }

    // ---- softmax over keys (registers + the 4 lane groups sharing a query)
    float inv_sum[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][t][r]);
        mx = wave_max_xor(mx, 16);
        mx = wave_max_xor(mx, 32);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[kt][t][r] - mx);
                s[kt][t][r] = e;
                sum += e;
            }
        sum = wave_sum_xor(sum, 16);
        sum = wave_sum_xor(sum, 32);
        inv_sum[t] = 1.0f / sum;
    }

    // ---- O^T[d][query] = sum_key V^T[d][key] P^T[key][query]
    // K-slot (lane group g, element j) of 32-key step ks <-> key 32*ks + 16*(j>>2) + 4*g + (j&3).
    TC* out = reinterpret_cast<TC*>(a.out);
    const int ldo = a.heads * hd_p;
#pragma unroll
    for (int dt = 0; dt < DC * 2; ++dt) {
        f32x4 o[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) o[t] = (f32x4)(0.0f);
        const TC* vrow = vt + (size_t)(dt * 16 + lr) * NTOK + lg * 4;
#pragma unroll
        for (int ks = 0; ks < KT / 2; ++ks) {
            const Frag<TC> vf = load_vt<TC>(vrow + ks * 32, vrow + ks * 32 + 16);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const Frag<TC> pf = pack_p<TC>(s[2 * ks][t], s[2 * ks + 1][t]);
                mma(vf, pf, o[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int qi = (qb * QT + t) * 16 + lr;
            f32x4 v = o[t] * inv_sum[t];
            store4(out + ((size_t)bwin * NTOK + qi) * ldo + head * hd_p + dt * 16 + lg * 4, v);
        }
    }
}

template <typename TC, int KT, int QT, int DC, bool FR = false>
__global__ __launch_bounds__(256, SR_ATTN_FLASH_WGS) void sr_window_attn_flash_kernel(SrWindowAttn a) {
    wattn_flash_block<TC, KT, QT, DC, FR>(a, blockIdx.x);
}

template <bool FR>
__global__ __launch_bounds__(256, 2) void sr_window_attn_lds_kernel(SrWindowAttn a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wattn_lds_block<FR>(a, blockIdx.x, smem);
}

__global__ __launch_bounds__(256, 2) void sr_window_attn_qkv_kernel(SrWindowAttn a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wattn_qkv_block(a, blockIdx.x, smem);
}

int launch_qkv_lds(const SrWindowAttn& a, hipStream_t st) {
    static SrDeviceOnce once;
    const hipError_t e = sr_once_per_device(once, [&] { return sr_allow_lds(sr_window_attn_qkv_kernel, WQ_LDS); });
    SR_REQUIRE(e == hipSuccess, "sr_window_attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(sr_window_attn_qkv_kernel, dim3(a.n_bwin * a.heads), dim3(256), WQ_LDS, st, a);
    SR_CHECK_LAUNCH("sr_window_attention");
    return SR_OK;
}

template <bool FR>
int launch_lds(const SrWindowAttn& a, hipStream_t st) {
    static SrDeviceOnce once;
    const hipError_t e = sr_once_per_device(once, [&] { return sr_allow_lds(sr_window_attn_lds_kernel<FR>, WL_LDS); });
    SR_REQUIRE(e == hipSuccess, "sr_window_attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(sr_window_attn_lds_kernel<FR>, dim3(a.n_bwin * a.heads), dim3(256), WL_LDS, st, a);
    SR_CHECK_LAUNCH("sr_window_attention");
    return SR_OK;
}

template <typename TC, int KT, int QT, int DC, bool FR = false>
int launch_flash(const SrWindowAttn& a, hipStream_t st) {
    const int items = a.n_bwin * a.heads * (KT / QT);
    hipLaunchKernelGGL((sr_window_attn_flash_kernel<TC, KT, QT, DC, FR>), dim3((items + 3) / 4), dim3(256), 0, st, a);
    SR_CHECK_LAUNCH("sr_window_attention");
    return SR_OK;
}

template <typename TC, int KT, int QT, int DC>
int launch_attn(const SrWindowAttn& a, hipStream_t st) {
    const int items = a.n_bwin * a.heads * (KT / QT);
    hipLaunchKernelGGL((sr_window_attn_kernel<TC, KT, QT, DC>), dim3((items + 3) / 4), dim3(256), 0, st, a);
    SR_CHECK_LAUNCH("sr_window_attention");
    return SR_OK;
}

// SR_BF16X3 (ABI v11): fp32 tensors, split-operand MFMAs -- the flash form only (fragment-ordered bias, head_dim <= 32)
static bool attn_x3_usable(const SrWindowAttn& a) { return a.bias_frag && a.ws % 4 == 0 && a.hd_p == 32 && (a.ntok == 256 || a.ntok == 64) && !a.qkv_frag && !a.bias_tiles && !a.x; }
static int dispatch_attn_x3(const SrWindowAttn& a, hipStream_t st) {
    if (a.ntok == 256) return launch_flash<bf3, 16, 2, 1>(a, st);
    return launch_flash<bf3, 4, 4, 1>(a, st);
}

template <typename TC>
int dispatch_attn(const SrWindowAttn& a, hipStream_t st) {
    if (a.bias_frag && a.ws % 4 == 0) {  // fragment-ordered bias available: flash form
        if (a.ntok == 256 && a.hd_p == 32) {
            if constexpr (sizeof(TC) == 2) {
                if (a.bias_tiles) return a.qkv_frag ? launch_lds<true>(a, st) : launch_lds<false>(a, st);  // K / V^T / distinct bias tiles in LDS
                if (a.qkv_frag) return launch_flash<TC, 16, 2, 1, true>(a, st);
            }
            return launch_flash<TC, 16, 2, 1>(a, st);  // 2 query tiles per wave, two workgroups per CU
        }
        if (a.ntok == 64 && a.hd_p == 32) return launch_flash<TC, 4, 4, 1>(a, st);
        if (a.ntok == 256 && a.hd_p == 64) return launch_flash<TC, 16, 2, 2>(a, st);
        if (a.ntok == 64 && a.hd_p == 64) return launch_flash<TC, 4, 4, 2>(a, st);
    }
    if (a.ntok == 64 && a.hd_p == 32) return launch_attn<TC, 4, 4, 1>(a, st);
    if (a.ntok == 256 && a.hd_p == 32) return launch_attn<TC, 16, 2, 1>(a, st);
    if (a.ntok == 64 && a.hd_p == 64) return launch_attn<TC, 4, 4, 2>(a, st);
    if (a.ntok == 256 && a.hd_p == 64) return launch_attn<TC, 16, 1, 2>(a, st);
    sr_set_error("sr_window_attention: unsupported ntok=%d hd_p=%d", a.ntok, a.hd_p);
    return SR_EUNSUPPORTED;
}

}  // namespace

extern "C" int sr_window_attention(const SrWindowAttn* p, void* stream) {
    SR_REQUIRE(p && ((p->q && p->k && p->vt) || p->x) && p->bias && p->out, "sr_window_attention: null pointer");
    const SrWindowAttn& a = *p;
    SR_REQUIRE(a.ws > 0 && a.ntok == a.ws * a.ws && a.H % a.ws == 0 && a.W % a.ws == 0 && a.shift >= 0 && a.shift < a.ws,
               "sr_window_attention: bad geometry");
    SR_REQUIRE(a.n_bwin > 0 && a.n_bwin % ((a.H / a.ws) * (a.W / a.ws)) == 0, "sr_window_attention: n_bwin");
    SR_REQUIRE(!a.qkv_frag || (a.dtype == SR_BF16 && a.ntok == 256 && a.hd_p == 32 && a.bias_frag && a.ws % 4 == 0), "sr_window_attention: qkv_frag needs bf16, 16 x 16 windows, hd_p 32 and bias_frag");
    SR_REQUIRE(!a.bias_tiles || (a.dtype == SR_BF16 && a.ntok == 256 && a.ws == 16 && a.hd_p == 32 && a.bias_frag), "sr_window_attention: bias_tiles needs bf16, 16 x 16 windows, hd_p 32 (and bias_frag for the fallback contract)");
    SR_REQUIRE(!a.bias_tiles || a.x || ((reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.k) | reinterpret_cast<uintptr_t>(a.vt) | reinterpret_cast<uintptr_t>(a.bias_tiles)) & 15) == 0,
               "sr_window_attention: the LDS form stages 16-byte pieces (q, k, vt, bias_tiles must be 16-byte aligned)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (a.x) {
        SR_REQUIRE(a.wqkv && a.bias_tiles && a.dtype == SR_BF16 && a.ntok == 256 && a.ws == 16 && a.hd_p == 32 && a.heads == 6 && a.C == 180 && a.ldx >= 192 && a.ldx % 4 == 0 &&
                       ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.wqkv)) & 15) == 0,
                   "sr_window_attention: the fused QKV form needs x, wqkv, bias_tiles, bf16, 16 x 16 windows, 6 heads of <= 32, C = 180 in >= 192 padded channels");
        return launch_qkv_lds(a, st);
    }
    if (a.dtype == SR_BF16X3) {
        SR_REQUIRE(attn_x3_usable(a), "sr_window_attention: SR_BF16X3 needs bias_frag, ws %% 4 == 0, hd_p 32, 64 / 256 tokens, row-major q / k / vt");
        return dispatch_attn_x3(a, st);
    }
    return a.dtype == SR_BF16 ? dispatch_attn<bf16>(a, st) : dispatch_attn<float>(a, st);
}
