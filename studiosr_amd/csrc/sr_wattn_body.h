// Flash-form window attention (sr_attn.hip: sr_window_attn_flash_kernel) as a device function, shared with sr_hab_mid.hip (the window attention and the CAB of a
// HAB as ONE launch).
#pragma once
#include "sr_common.h"
#include "sr_host.h"

namespace {

SR_DEV int region(int v, int size, int ws, int shift) { return v < size - ws ? 0 : (v < size - shift ? 1 : 2); }

template <typename TC, typename TS = TC>
SR_DEV Frag<TC> load_vt(const TS* p0, const TS* p1);
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

// Storage type of q / k / v^T / out for a compute fragment type: the split-operand form (bf3, compute type SR_BF16X3 = precision "fp32x3", round 5) works on fp32 tensors --
// every operand fragment is split into hi + lo bf16 as it is loaded, P as it is packed; three bf16 MFMAs per product instead of the eight 16x16x4 fp32 MFMAs of TC = float.
template <typename TC>
struct WaStore {
    typedef TC type;
};
template <>
struct WaStore<bf3> {
    typedef float type;
};
template <>
SR_DEV Frag<bf3> load_vt<bf3, float>(const float* p0, const float* p1) {
    float v[8];
    const f32x4 a = *reinterpret_cast<const f32x4*>(p0), b = *reinterpret_cast<const f32x4*>(p1);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    return frag_make<bf3>(v);
}

template <typename TC>
SR_DEV Frag<TC> pack_p(const f32x4& a, const f32x4& b);
template <>
SR_DEV Frag<bf3> pack_p<bf3>(const f32x4& a, const f32x4& b) {
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return frag_make<bf3>(v);
}
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
SR_DEV void wattn_flash_block(const SrWindowAttn& a, const int block_id) {  // the work of one workgroup: 4 waves = 4 items, no LDS, no barriers
    static_assert(!FR || DC == 1, "fragment order: head_dim 32");
    static_assert(KT % 4 == 0 && KT % QT == 0, "key blocks of 64");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NTOK = KT * 16;
    constexpr int QB = KT / QT;
    constexpr float LOG2E = 1.4426950408889634f;
    const int item = block_id * 4 + wave;
    if (item >= a.n_bwin * a.heads * QB) return;  // wave-uniform; no barriers in this kernel
    const int bwin = item % a.n_bwin;             // windows fastest: a workgroup shares (head, query block)
    const int hq = item / a.n_bwin;
    const int qb = hq % QB, head = hq / QB;
    const int bh = bwin * a.heads + head;
    constexpr int hd_p = DC * 32;
    const int lr = lane & 15, lg = lane >> 4;

    typedef typename WaStore<TC>::type TS;
    static_assert(!FR || sizeof(TS) == 2, "fragment order: bf16 storage");
    const TS* q = reinterpret_cast<const TS*>(a.q) + (size_t)bh * NTOK * hd_p;
    const TS* k = reinterpret_cast<const TS*>(a.k) + (size_t)bh * NTOK * hd_p;
    const TS* vt = reinterpret_cast<const TS*>(a.vt) + (size_t)bh * NTOK * hd_p;
    const f32x4* bfrag = reinterpret_cast<const f32x4*>(a.bias_frag) + ((size_t)(head * KT + qb * QT) * KT) * 64 + lane;  // [qt][kt][lane]

    Frag<TC> qf[QT][DC];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            if constexpr (FR)
                qf[t][c] = load_group<TC, TS>(q + (size_t)((qb * QT + t) * 64 + lane) * 8);
            else
                qf[t][c] = load_group<TC, TS>(q + (size_t)((qb * QT + t) * 16 + lr) * hd_p + c * 32 + lg * 8);
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
                    kf[j][c] = load_group<TC, TS>(k + (size_t)((kb * 4 + j) * 64 + lane) * 8);
                else
                    kf[j][c] = load_group<TC, TS>(k + (size_t)((kb * 4 + j) * 16 + lr) * hd_p + c * 32 + lg * 8);
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
                const TS* vrow = vt + (size_t)(dt * 16 + lr) * NTOK + lg * 4 + kb * 64;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if constexpr (FR)
                        vf[dt][ks] = load_group<TC, TS>(vt + (size_t)(((kb * 2 + dt) * 2 + ks) * 64 + lane) * 8);
                    else
                        vf[dt][ks] = load_vt<TC, TS>(vrow + ks * 32, vrow + ks * 32 + 16);
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
            const TS* vrow = vt + (size_t)(dt * 16 + lr) * NTOK + lg * 4 + kb * 64;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (!PF) vf[dt][ks] = load_vt<TC, TS>(vrow + ks * 32, vrow + ks * 32 + 16);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const Frag<TC> pf = pack_p<TC>(s[2 * ks][t], s[2 * ks + 1][t]);
                    mma(vf[dt][ks], pf, o[dt][t]);
                }
            }
        }
    }

    TS* out = reinterpret_cast<TS*>(a.out);
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


}  // namespace
