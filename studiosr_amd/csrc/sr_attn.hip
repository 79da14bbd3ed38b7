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
#include "sr_common.h"
#include "sr_host.h"

namespace {

SR_DEV int region(int v, int size, int ws, int shift) { return v < size - ws ? 0 : (v < size - shift ? 1 : 2); }

template <typename TC>
SR_DEV Frag<TC> load_vt(const TC* p0, const TC* p1);
template <>
SR_DEV Frag<bf16> load_vt<bf16>(const bf16* p0, const bf16* p1) {
    bf16x4 a = *reinterpret_cast<const bf16x4*>(p0);
    bf16x4 b = *reinterpret_cast<const bf16x4*>(p1);
    Frag<bf16> f;
    f.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return f;
}
template <>
SR_DEV Frag<float> load_vt<float>(const float* p0, const float* p1) {
    Frag<float> f;
    f.lo = *reinterpret_cast<const f32x4*>(p0);
    f.hi = *reinterpret_cast<const f32x4*>(p1);
    return f;
}

template <typename TC>
SR_DEV Frag<TC> pack_p(const f32x4& a, const f32x4& b);
template <>
SR_DEV Frag<bf16> pack_p<bf16>(const f32x4& a, const f32x4& b) {
    Frag<bf16> f;
    f.v[0] = (bf16)a[0]; f.v[1] = (bf16)a[1]; f.v[2] = (bf16)a[2]; f.v[3] = (bf16)a[3];
    f.v[4] = (bf16)b[0]; f.v[5] = (bf16)b[1]; f.v[6] = (bf16)b[2]; f.v[7] = (bf16)b[3];
    return f;
}
template <>
SR_DEV Frag<float> pack_p<float>(const f32x4& a, const f32x4& b) {
    Frag<float> f;
    f.lo = a;
    f.hi = b;
    return f;
}

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

// ----------------------------------------------------------------------------- flash form (bias in fragment order)
// Same math, restructured for long windows (ws 16: 256 keys) where the kernel above is latency- and L1-bound (its
// row-major fp32 bias reads touch 16 cache lines per wave instruction and its 32 S^T tiles leave one wave per SIMD):
//   * one wave = (window, head, QT*16 queries); keys are walked in blocks of 64 with an online softmax
//     (running max m, per-lane partial sum l, O rescaled by exp(m_old - m_new)), so only 4 x QT logit tiles are live;
//   * the relative-position bias arrives in ACCUMULATOR-FRAGMENT order ([head][qt][kt][lane][4], packing.bias_fragments):
//     each S^T tile is initialised by one coalesced 1 KiB load and the K Q^T MFMAs accumulate on top of it;
//   * the 4 waves of a workgroup are 4 consecutive windows of the same (head, query block): they read the same bias
//     tiles, which therefore come from the CU's L1.
// 221 VGPRs at two workgroups per CU (HAT x4 b4: 768 workgroups = 1.5 residency rounds); forcing three (168 VGPRs) spills 53 registers:
// HAT b4 3.37 -> 4.49 ms.  The double-buffered bias + K fragments of the next key block are most of the registers.
#ifndef SR_ATTN_FLASH_WGS
#define SR_ATTN_FLASH_WGS 2
#endif
// FR (SrWindowAttn.qkv_frag, hd_p == 32): q, k and v^T arrive in FRAGMENT order -- q / k as [tile of 16 tokens][lane][8], v^T as [64-key block][d tile][32-key
// step][lane][8] (written so by sr_swin_qkv / sr_swin_tail) -- so that every operand fragment is ONE fully coalesced 1-KiB load (lane * 16 B).  In the
// row-major layouts adjacent lanes sit on different rows (64 B / 512 B apart): four cache lines per quad of lanes, ~125 cycles of issue per load, 14 loads
// per key block (profiles/r03_window_attention_ablation.txt).
template <typename TC, int KT, int QT, int DC, bool FR = false>
__global__ __launch_bounds__(256, SR_ATTN_FLASH_WGS) void sr_window_attn_flash_kernel(SrWindowAttn a) {
    static_assert(!FR || DC == 1, "fragment order: head_dim 32");
    static_assert(KT % 4 == 0 && KT % QT == 0, "key blocks of 64");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NTOK = KT * 16;
    constexpr int QB = KT / QT;
    constexpr float LOG2E = 1.4426950408889634f;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.n_bwin * a.heads * QB) return;  // wave-uniform; no barriers in this kernel
    const int bwin = item % a.n_bwin;             // windows fastest: a workgroup shares (head, query block)
    const int hq = item / a.n_bwin;
    const int qb = hq % QB, head = hq / QB;
    const int bh = bwin * a.heads + head;
    constexpr int hd_p = DC * 32;
    const int lr = lane & 15, lg = lane >> 4;

    const TC* q = reinterpret_cast<const TC*>(a.q) + (size_t)bh * NTOK * hd_p;
    const TC* k = reinterpret_cast<const TC*>(a.k) + (size_t)bh * NTOK * hd_p;
    const TC* vt = reinterpret_cast<const TC*>(a.vt) + (size_t)bh * NTOK * hd_p;
    const f32x4* bfrag = reinterpret_cast<const f32x4*>(a.bias_frag) + ((size_t)(head * KT + qb * QT) * KT) * 64 + lane;  // [qt][kt][lane]

    Frag<TC> qf[QT][DC];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            if constexpr (FR)
                qf[t][c] = *reinterpret_cast<const Frag<TC>*>(q + (size_t)((qb * QT + t) * 64 + lane) * 8);
            else
                qf[t][c] = *reinterpret_cast<const Frag<TC>*>(q + (size_t)((qb * QT + t) * 16 + lr) * hd_p + c * 32 + lg * 8);
        }

    // shift mask (common.py:250-274): label(q) != label(k)  <=>  the row halves differ (last window row only) or the column
    // halves differ (last window column only).  ws % 4 == 0, so the 4 keys of a lane group share a window row.
    const int nwx = a.W / a.ws, nwy = a.H / a.ws;
    const int win = bwin % (nwx * nwy);
    const int wy = win / nwx, wx = win - wy * nwx;
    const bool last_row = a.y_mode != SR_Y_STRIP && wy == nwy - 1, last_col = wx == nwx - 1;
    const bool masked = a.shift > 0 && (last_row || last_col);
    const float inv_ws = 1.0f / (float)a.ws;
    bool qrow[QT], qcol[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = (qb * QT + t) * 16 + lr;
        const int i = (int)(((float)qi + 0.5f) * inv_ws), j = qi - i * a.ws;
        qrow[t] = last_row && i >= a.ws - a.shift;
        qcol[t] = last_col && j >= a.ws - a.shift;
    }

    float m_run[QT], l_run[QT];
    f32x4 o[2 * DC][QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_run[t] = -3.0e38f;
        l_run[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 2 * DC; ++dt) o[dt][t] = (f32x4)(0.0f);
    }

    // bf16: the bias tiles and K fragments of key block kb + 1 and the V fragments of block kb are requested before the MFMAs / softmax of
    // block kb (the first version fetched each right before its use: three exposed L2 round trips per key block, 20+ us per wave)
    constexpr bool PF = sizeof(TC) == 2;
    constexpr int NB = PF ? 2 : 1;
    f32x4 bb[NB][4][QT];
    Frag<TC> kk[NB][4][DC];
    auto fetch = [&](int kb, f32x4 (&b)[4][QT], Frag<TC> (&kf)[4][DC]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int t = 0; t < QT; ++t) b[j][t] = bfrag[((size_t)t * KT + kb * 4 + j) * 64];
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                if constexpr (FR)
                    kf[j][c] = *reinterpret_cast<const Frag<TC>*>(k + (size_t)((kb * 4 + j) * 64 + lane) * 8);
                else
                    kf[j][c] = *reinterpret_cast<const Frag<TC>*>(k + (size_t)((kb * 4 + j) * 16 + lr) * hd_p + c * 32 + lg * 8);
            }
        }
    };
    if constexpr (PF) fetch(0, bb[0], kk[0]);
#pragma unroll
    for (int kb = 0; kb < KT / 4; ++kb) {
        const int cur = PF ? (kb & 1) : 0;
        Frag<TC> vf[2 * DC][2];
        if constexpr (PF) {
#pragma unroll
            for (int dt = 0; dt < 2 * DC; ++dt) {
                const TC* vrow = vt + (size_t)(dt * 16 + lr) * NTOK + lg * 4 + kb * 64;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if constexpr (FR)
                        vf[dt][ks] = *reinterpret_cast<const Frag<TC>*>(vt + (size_t)(((kb * 2 + dt) * 2 + ks) * 64 + lane) * 8);
                    else
                        vf[dt][ks] = load_vt<TC>(vrow + ks * 32, vrow + ks * 32 + 16);
                }
            }
            if (kb + 1 < KT / 4) fetch(kb + 1, bb[(kb + 1) & 1], kk[(kb + 1) & 1]);
        } else {
            fetch(kb, bb[0], kk[0]);
        }
        // ---- S^T tiles of this key block: bias tile + K Q^T
        f32x4 s[4][QT];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < QT; ++t) s[j][t] = bb[cur][j][t];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < DC; ++c) {
#pragma unroll
                for (int t = 0; t < QT; ++t) mma(kk[cur][j][c], qf[t][c], s[j][t]);
            }
        if (masked) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int base = (kb * 4 + j) * 16 + lg * 4;
                const int i = (int)(((float)base + 0.5f) * inv_ws), j0 = base - i * a.ws;
                const bool krow = last_row && i >= a.ws - a.shift;
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const float rowneg = krow != qrow[t] ? -100.0f : 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float colneg = (last_col && j0 + r >= a.ws - a.shift) != qcol[t] ? -100.0f : 0.0f;
                        s[j][t][r] += fminf(rowneg, colneg);
                    }
                }
            }
        }
        // ---- online softmax update per query tile
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = s[0][t][0];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[j][t][r]);
            mx = wave_max_xor(mx, 16);
            mx = wave_max_xor(mx, 32);
            const float m_new = fmaxf(m_run[t], mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run[t] - m_new) * LOG2E);
            m_run[t] = m_new;
            const float nm = -m_new * LOG2E;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][t][r], LOG2E, nm));
                    s[j][t][r] = e;
                    sum += e;
                }
            l_run[t] = l_run[t] * alpha + sum;  // per-lane partial (this lane group's keys); combined once at the end
#pragma unroll
            for (int dt = 0; dt < 2 * DC; ++dt) o[dt][t] *= alpha;
        }
        // ---- O^T += V^T P^T  (32-key steps; key order inside a step as in the kernel above)
#pragma unroll
        for (int dt = 0; dt < 2 * DC; ++dt) {
            const TC* vrow = vt + (size_t)(dt * 16 + lr) * NTOK + lg * 4 + kb * 64;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (!PF) vf[dt][ks] = load_vt<TC>(vrow + ks * 32, vrow + ks * 32 + 16);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const Frag<TC> pf = pack_p<TC>(s[2 * ks][t], s[2 * ks + 1][t]);
                    mma(vf[dt][ks], pf, o[dt][t]);
                }
            }
        }
    }

    TC* out = reinterpret_cast<TC*>(a.out);
    const int ldo = a.heads * hd_p;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l = wave_sum_xor(l_run[t], 16);
        l = wave_sum_xor(l, 32);
        const float inv = 1.0f / l;
        const int qi = (qb * QT + t) * 16 + lr;
#pragma unroll
        for (int dt = 0; dt < 2 * DC; ++dt) {
            f32x4 v = o[dt][t] * inv;
            store4(out + ((size_t)bwin * NTOK + qi) * ldo + head * hd_p + dt * 16 + lg * 4, v);
        }
    }
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

template <typename TC>
int dispatch_attn(const SrWindowAttn& a, hipStream_t st) {
    if (a.bias_frag && a.ws % 4 == 0) {  // fragment-ordered bias available: flash form
        if (a.ntok == 256 && a.hd_p == 32) {
            if constexpr (sizeof(TC) == 2) {
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
    SR_REQUIRE(p && p->q && p->k && p->vt && p->bias && p->out, "sr_window_attention: null pointer");
    const SrWindowAttn& a = *p;
    SR_REQUIRE(a.ws > 0 && a.ntok == a.ws * a.ws && a.H % a.ws == 0 && a.W % a.ws == 0 && a.shift >= 0 && a.shift < a.ws,
               "sr_window_attention: bad geometry");
    SR_REQUIRE(a.n_bwin > 0 && a.n_bwin % ((a.H / a.ws) * (a.W / a.ws)) == 0, "sr_window_attention: n_bwin");
    SR_REQUIRE(!a.qkv_frag || (a.dtype == SR_BF16 && a.ntok == 256 && a.hd_p == 32 && a.bias_frag && a.ws % 4 == 0), "sr_window_attention: qkv_frag needs bf16, 16 x 16 windows, hd_p 32 and bias_frag");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return a.dtype == SR_BF16 ? dispatch_attn<bf16>(a, st) : dispatch_attn<float>(a, st);
}
