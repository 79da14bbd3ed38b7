// Flash-form overlapping cross attention (hat.py:266-283) for HAT's long key sets (ws 16: 576 keys), where the register-only
// kernel of sr_oca.hip is bound by L1 delivery and load latency (every wave pulled its own copy of the fp32
// relative-position bias, 1 KiB per MFMA pair, and 36 logit tiles left one wave per SIMD).  Measured: 428 -> 115 us.
//
//   * one wave = (window, head, QT*16 queries); keys are walked in blocks of 64 with an online softmax, so 4 x QT logit
//     tiles are live instead of (keys/16) x QT;
//   * the 4 waves of a workgroup are 4 CONSECUTIVE WINDOWS of the same (head, query block): they need the same bias tiles.
//     The bias (accumulator-fragment order, packing.bias_fragments; padded key columns hold -1e30) is streamed through a
//     double-buffered LDS slab: each wave fetches a quarter of the next block's tiles while the current block is being
//     processed, one barrier per block;
//   * the V^T fragments of a block are fetched before its S^T MFMAs and the K fragments of the next block right after them
//     (in place), so both land during the softmax; the only exposed global latency is the first block's;
//   * `OcaKeys`: K gathered from the zero-bordered image, V^T from the zero-bordered planes (nn.Unfold is never materialised).
// (Plain window attention keeps the register-only flash kernel of sr_attn.hip: with 256 keys it is as fast without LDS.)
#include "sr_common.h"
#include "sr_host.h"
#include <type_traits>

namespace {

template <typename TC>
SR_DEV Frag<TC> pack_pf(const f32x4& a, const f32x4& b);
template <>
SR_DEV Frag<bf16> pack_pf<bf16>(const f32x4& a, const f32x4& b) {
    Frag<bf16> f;
    f.v[0] = (bf16)a[0]; f.v[1] = (bf16)a[1]; f.v[2] = (bf16)a[2]; f.v[3] = (bf16)a[3];
    f.v[4] = (bf16)b[0]; f.v[5] = (bf16)b[1]; f.v[6] = (bf16)b[2]; f.v[7] = (bf16)b[3];
    return f;
}
template <>
SR_DEV Frag<float> pack_pf<float>(const f32x4& a, const f32x4& b) {
    Frag<float> f;
    f.lo = a;
    f.hi = b;
    return f;
}

// two runs of 4 consecutive keys -> one V^T fragment (K-slot (lane group g, element j) of a 32-key step <-> key 32 ks + 16 (j >> 2) + 4 g + (j & 3))
template <bool ALIGN4>
SR_DEV Frag<bf16> vt_pair(const bf16* p0, const bf16* p1) {
    Frag<bf16> f;
    if constexpr (ALIGN4) {
        const bf16x4 a = *reinterpret_cast<const bf16x4*>(p0), b = *reinterpret_cast<const bf16x4*>(p1);
        f.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    } else {  // 4-byte aligned only (OCA, ws 8: logical pad 2)
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        const bf16x2 a0 = *reinterpret_cast<const bf16x2*>(p0), a1 = *reinterpret_cast<const bf16x2*>(p0 + 2);
        const bf16x2 b0 = *reinterpret_cast<const bf16x2*>(p1), b1 = *reinterpret_cast<const bf16x2*>(p1 + 2);
        f.v[0] = a0[0]; f.v[1] = a0[1]; f.v[2] = a1[0]; f.v[3] = a1[1];
        f.v[4] = b0[0]; f.v[5] = b0[1]; f.v[6] = b1[0]; f.v[7] = b1[1];
    }
    return f;
}
template <bool ALIGN4>
SR_DEV Frag<float> vt_pair(const float* p0, const float* p1) {
    Frag<float> f;
    if constexpr (ALIGN4) {
        f.lo = *reinterpret_cast<const f32x4*>(p0);
        f.hi = *reinterpret_cast<const f32x4*>(p1);
    } else {
        f.lo = f32x4{p0[0], p0[1], p0[2], p0[3]};
        f.hi = f32x4{p1[0], p1[1], p1[2], p1[3]};
    }
    return f;
}

// Storage type of q / k / v^T / out: the split-operand form (bf3, compute type SR_BF16X3 = precision "fp32x3"; round 5, ABI v11) works on fp32 tensors -- operand fragments are
// split into hi + lo bf16 as they are loaded, P as it is packed: three bf16 MFMAs per product instead of the eight 16x16x4 fp32 MFMAs of TC = float.
template <typename TC>
struct FlStore {
    typedef TC type;
};
template <>
struct FlStore<bf3> {
    typedef float type;
};
template <>
SR_DEV Frag<bf3> pack_pf<bf3>(const f32x4& a, const f32x4& b) {
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return frag_make<bf3>(v);
}
template <typename TC>
SR_DEV Frag<TC> fl_convert(const Frag<TC>& f) { return f; }
SR_DEV Frag<bf3> fl_split(const Frag<float>& f) {
    const float v[8] = {f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
    return frag_make<bf3>(v);
}

struct FlashArgs {
    const void *q, *k, *vt;
    const float* bias_frag;  // [heads][qt][ktp][lane][4]
    void* out;
    int n_bwin, heads;
    int ntok;                // queries per window
    int ktp;                 // padded key tiles (multiple of 4)
    int H, W, ws;            // image / window geometry
    int pad, border, nk;     // neighbourhood: logical pad, physical zero border, real key count
};

// keys of the (ws + 2 pad)^2 neighbourhood of a window, read in place from the zero-bordered image / planes
template <typename TC, bool ALIGN4>
struct OcaKeys {
    typedef typename FlStore<TC>::type TS;
    const TS *kimg, *vplane;
    int wse, nk, Wp2, HP, lr, lg;
    size_t plane;
    SR_DEV OcaKeys(const FlashArgs& a, int bwin, int head, int lane) {
        const int nwx = a.W / a.ws, nwy = a.H / a.ws;
        const int win = bwin % (nwx * nwy), b = bwin / (nwx * nwy);
        const int wy = win / nwx, wx = win - wy * nwx;
        wse = a.ws + 2 * a.pad;
        nk = a.nk;
        const int Hp2 = a.H + 2 * a.border;
        Wp2 = a.W + 2 * a.border;
        HP = a.heads * 32;
        lr = lane & 15;
        lg = lane >> 4;
        const int oy = wy * a.ws - a.pad + a.border, ox = wx * a.ws - a.pad + a.border;  // neighbourhood origin, bordered coordinates
        plane = (size_t)Hp2 * Wp2;
        kimg = reinterpret_cast<const TS*>(a.k) + ((size_t)b * plane + (size_t)oy * Wp2 + ox) * HP + head * 32 + lg * 8;
        vplane = reinterpret_cast<const TS*>(a.vt) + (((size_t)b * a.heads + head) * 32 + lr) * plane + (size_t)oy * Wp2 + ox;
    }
    SR_DEV Frag<TC> kfrag(int kt) const {
        int key = kt * 16 + lr;
        if (key >= nk) key = nk - 1;  // padded key: any valid row (its bias column is -1e30)
        const int ky = key / wse, kx = key - ky * wse;
        return load_group<TC, TS>(kimg + ((size_t)ky * Wp2 + kx) * HP);
    }
    SR_DEV Frag<TC> vfrag(int dt, int ks) const {
        int ka = ks * 32 + lg * 4, kb = ka + 16;
        if (ka >= nk) ka = 0;  // padded keys carry p == 0; keep the address in bounds
        if (kb >= nk) kb = 0;
        const int kay = ka / wse, kax = ka - kay * wse, kby = kb / wse, kbx = kb - kby * wse;
        const TS* base = vplane + (size_t)dt * 16 * plane;
        if constexpr (sizeof(Frag<TC>) == 32 && sizeof(TS) == 4 && !std::is_same<TC, float>::value)
            return fl_split(vt_pair<ALIGN4>(base + (size_t)kay * Wp2 + kax, base + (size_t)kby * Wp2 + kbx));
        else
            return vt_pair<ALIGN4>(base + (size_t)kay * Wp2 + kax, base + (size_t)kby * Wp2 + kbx);
    }
};

template <typename TC, typename Src, int QT>
__global__ __launch_bounds__(256) void sr_attn_flash_kernel(FlashArgs a) {
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* bl = reinterpret_cast<f32x4*>(smem);  // [2][QT * 4 tiles][64 lanes]
    constexpr int SLAB = QT * 4 * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int qtt = a.ntok / 16, QB = qtt / QT;
    const int groups = (a.n_bwin + 3) / 4;  // workgroups per (head, query block)
    const int hq = blockIdx.x / groups, grp = blockIdx.x - hq * groups;
    const int qb = hq % QB, head = hq / QB;
    int bwin = grp * 4 + wave;
    const bool live = bwin < a.n_bwin;  // a partial last group recomputes the last window and stores nothing
    if (!live) bwin = a.n_bwin - 1;
    const Src src(a, bwin, head, lane);
    const int KB = a.ktp / 4;

    typedef typename FlStore<TC>::type TS;
    const TS* q = reinterpret_cast<const TS*>(a.q) + ((size_t)bwin * a.heads + head) * a.ntok * 32;
    Frag<TC> qf[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) qf[t] = load_group<TC, TS>(q + (size_t)((qb * QT + t) * 16 + lr) * 32 + lg * 8);

    // bias tile (t, j) of key block kb: [head][qb*QT + t][kb*4 + j]; this wave fetches tiles wave*QT .. wave*QT + QT - 1 of the slab
    const f32x4* bsrc = reinterpret_cast<const f32x4*>(a.bias_frag) + ((size_t)(head * qtt + qb * QT) * a.ktp) * 64 + lane;
    auto bias_tile = [&](int kb, int idx) {  // idx = t * 4 + j
        const int t = idx >> 2, j = idx & 3;
        return bsrc[((size_t)t * a.ktp + kb * 4 + j) * 64];
    };

    float m_run[QT], l_run[QT];
    f32x4 o[2][QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_run[t] = -3.0e38f;
        l_run[t] = 0.f;
        o[0][t] = (f32x4)(0.0f);
        o[1][t] = (f32x4)(0.0f);
    }

    // ---- prologue: block 0 bias -> slab 0, block 0 K fragments -> registers
    Frag<TC> kf[4];
    {
        f32x4 b0[QT];
#pragma unroll
        for (int i = 0; i < QT; ++i) b0[i] = bias_tile(0, wave * QT + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[j] = src.kfrag(j);
#pragma unroll
        for (int i = 0; i < QT; ++i) bl[(wave * QT + i) * 64 + lane] = b0[i];
    }
    __syncthreads();

    for (int kb = 0; kb < KB; ++kb) {
        const f32x4* cur = bl + (kb & 1) * SLAB + lane;
        const bool more = kb + 1 < KB;
        // ---- prefetch: next block's bias quarter + K fragments, this block's V^T fragments
        f32x4 bn[QT];
        Frag<TC> vf[2][2];
        const int kbn = more ? kb + 1 : kb;
#pragma unroll
        for (int i = 0; i < QT; ++i) bn[i] = bias_tile(kbn, wave * QT + i);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) vf[dt][ks] = src.vfrag(dt, kb * 2 + ks);
        __builtin_amdgcn_sched_barrier(0);

        // ---- S^T tiles: bias (LDS) + K Q^T
        f32x4 s[4][QT];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j][t] = cur[(t * 4 + j) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < QT; ++t) mma(kf[j], qf[t], s[j][t]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[j] = src.kfrag(kbn * 4 + j);  // next block's K fragments land during the softmax
        // ---- online softmax
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
            o[0][t] *= alpha;
            o[1][t] *= alpha;
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const Frag<TC> pf = pack_pf<TC>(s[2 * ks][t], s[2 * ks + 1][t]);
                mma(vf[0][ks], pf, o[0][t]);
                mma(vf[1][ks], pf, o[1][t]);
            }
        // ---- publish the next block's bias quarter
        if (more) {
            f32x4* nxt = bl + ((kb + 1) & 1) * SLAB;
#pragma unroll
            for (int i = 0; i < QT; ++i) nxt[(wave * QT + i) * 64 + lane] = bn[i];
        }
        __syncthreads();
    }

    if (live) {
        TS* out = reinterpret_cast<TS*>(a.out);
        const int ldo = a.heads * 32;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float l = wave_sum_xor(l_run[t], 16);
            l = wave_sum_xor(l, 32);
            const float inv = 1.0f / l;
            const int qi = (qb * QT + t) * 16 + lr;
            TS* dst = out + ((size_t)bwin * a.ntok + qi) * ldo + head * 32 + lg * 4;
            store4(dst, o[0][t] * inv);
            store4(dst + 16, o[1][t] * inv);
        }
    }
}

template <typename TC, typename Src, int QT>
int launch(const FlashArgs& a, hipStream_t st, const char* what) {
    const int QB = (a.ntok / 16) / QT;
    const int groups = (a.n_bwin + 3) / 4;
    constexpr int lds = 2 * QT * 4 * 64 * (int)sizeof(f32x4);
    hipLaunchKernelGGL((sr_attn_flash_kernel<TC, Src, QT>), dim3(a.heads * QB * groups), dim3(256), lds, st, a);
    SR_CHECK_LAUNCH(what);
    return SR_OK;
}

}  // namespace

// sr_oca_attention with a fragment-ordered bias whose key dimension is padded to nk_frag (multiple of 64, pad columns -1e30)
bool sr_oca_attention_flash_supported(const SrOcaAttn& o) {
    return o.bias_frag && o.hd_p == 32 && (o.ws * o.ws) % 64 == 0 && o.nk_frag % 64 == 0 && o.nk_frag >= (o.ws + 2 * o.pad) * (o.ws + 2 * o.pad);
}

int sr_oca_attention_flash(const SrOcaAttn& o, hipStream_t st) {
    FlashArgs a{};
    a.q = o.q; a.k = o.k; a.vt = o.vt; a.bias_frag = o.bias_frag; a.out = o.out;
    a.n_bwin = o.B * (o.H / o.ws) * (o.W / o.ws); a.heads = o.heads; a.ntok = o.ws * o.ws; a.ktp = o.nk_frag / 16;
    a.H = o.H; a.W = o.W; a.ws = o.ws; a.pad = o.pad; a.border = o.border; a.nk = (o.ws + 2 * o.pad) * (o.ws + 2 * o.pad);
    const bool align4 = (o.pad % 4 == 0) && ((o.ws + 2 * o.pad) % 4 == 0);  // every 4-key run starts 8-byte aligned in the bordered planes
    if (o.dtype == SR_BF16) {
        if (align4) return launch<bf16, OcaKeys<bf16, true>, 4>(a, st, "sr_oca_attention");
        return launch<bf16, OcaKeys<bf16, false>, 4>(a, st, "sr_oca_attention");
    }
    if (o.dtype == SR_BF16X3) {  // fp32 tensors, split-operand MFMAs (ABI v11)
        if (align4) return launch<bf3, OcaKeys<bf3, true>, 4>(a, st, "sr_oca_attention");
        return launch<bf3, OcaKeys<bf3, false>, 4>(a, st, "sr_oca_attention");
    }
    if (align4) return launch<float, OcaKeys<float, true>, 4>(a, st, "sr_oca_attention");
    return launch<float, OcaKeys<float, false>, 4>(a, st, "sr_oca_attention");
}
