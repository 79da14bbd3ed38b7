// HAT's overlapping cross attention (hat.py:239-283) for 16 x 16 query windows and their 24 x 24 key neighbourhoods with every operand of the key loop in
// LDS (round 4; C ABI v8 SrOcaAttn.bias_rel / SrTrAttnFwd.bias_rel):
//     o = softmax(q k^T + bias[head]) v        one (window, head) per 4-wave workgroup, wave w = queries [64 w, +64) in two passes of 32
// The flash form (sr_attn_flash.hip) streams the bias through an LDS slab but fetches the K / V^T fragments of every 64-key block from L2 per wave (four
// windows per workgroup, 64 queries per wave: 16 waves read each fragment).  Here a workgroup stages, ONCE,
//   * K of its 576 keys as 36 operand fragments and V^T as 36 fragments in the accumulator-as-operand key order (2 x 36 KiB), gathered from the
//     zero-bordered image / planes (inference: nn.Unfold is never materialised) or from the unfolded per-window arrays (training forward), and
//   * the head's relative-position bias as its TABLE: bias[q][k] = table[(ky - qy - 7) * 39 + (kx - qx - 7)] with python-style wrap of negative
//     indices (hat.py:494-517, 276-279), so with j = (ky - qy + 15) * 39 + (kx - qx + 15) in [0, 1521) the bias is rel[j] (the table rotated by 880
//     entries) and a lane's four logits of a tile -- four consecutive keys of one key row -- are four CONSECUTIVE entries (6 KiB per head instead
//     of 590 KiB of gathered bias; packing.oca_bias_rel checks the structure on the gathered bias before offering the operand).
// 78 KiB of LDS: two workgroups per CU.  Online softmax over key blocks of 64 as in the flash form.
#include "sr_common.h"
#include "sr_host.h"
#include <cstdlib>

#ifndef SR_OCA_XCD
#define SR_OCA_XCD 1  // HAT x4 b4 2.41 -> 2.36 ms (0: (window, head) items in block-id order)
#endif
namespace {

constexpr int OL_NK = 576, OL_KT = 36, OL_KB = 9, OL_REL = 1521;
constexpr int OL_OFF_K = 0, OL_OFF_V = OL_KT * 1024, OL_OFF_B = 2 * OL_KT * 1024, OL_LDS = OL_OFF_B + 6144;
static_assert(2 * OL_LDS <= 160 * 1024, "two workgroups per CU");

struct OcaLdsArgs {
    const void *q, *k, *vt;
    const float* rel;   // [heads][1521]
    void* out;          // rows [bwin * 256 + tok][ldo], head at column 32 head
    float* lse;         // optional [bwin][head][256]
    int n_bwin, heads, ldo;
    int H, W, border;   // bordered source only
};

// keys gathered in place from the zero-bordered image [B][H+2e][W+2e][heads][32] / planes [B][heads][32][(H+2e)(W+2e)] (pad == border == 4)
struct OlBordered {
    const bf16 *kimg, *vplane;
    int Wp2, HP;
    size_t plane;
    SR_DEV OlBordered(const OcaLdsArgs& a, int bwin, int head) {
        const int nwx = a.W / 16, nwy = a.H / 16;
        const int win = bwin % (nwx * nwy), b = bwin / (nwx * nwy);
        const int wy = win / nwx, wx = win - wy * nwx;
        Wp2 = a.W + 2 * a.border;
        HP = a.heads * 32;
        plane = (size_t)(a.H + 2 * a.border) * Wp2;
        const int oy = wy * 16, ox = wx * 16;  // neighbourhood origin in bordered coordinates (- pad + border = 0)
        kimg = reinterpret_cast<const bf16*>(a.k) + ((size_t)b * plane + (size_t)oy * Wp2 + ox) * HP + head * 32;
        vplane = reinterpret_cast<const bf16*>(a.vt) + ((size_t)b * a.heads + head) * 32 * plane + (size_t)oy * Wp2 + ox;
    }
    SR_DEV const bf16* k_piece(int key, int g) const {  // features 8 g .. of key `key`
        const int ky = key / 24, kx = key - ky * 24;
        return kimg + ((size_t)ky * Wp2 + kx) * HP + g * 8;
    }
    SR_DEV const bf16* vt_piece(int d, int p8) const {  // keys 8 p8 .. + 7 of feature row d (one key row: 24 = 3 x 8)
        const int ky = p8 / 3, kx = (p8 - ky * 3) * 8;
        return vplane + (size_t)d * plane + (size_t)ky * Wp2 + kx;
    }
};
// unfolded per-window arrays of the training forward: k [bwin][head][576][32], vT [bwin][head][32][576]
struct OlUnfolded {
    const bf16 *k, *vt;
    SR_DEV OlUnfolded(const OcaLdsArgs& a, int bwin, int head) {
        const size_t bh = (size_t)bwin * a.heads + head;
        k = reinterpret_cast<const bf16*>(a.k) + bh * OL_NK * 32;
        vt = reinterpret_cast<const bf16*>(a.vt) + bh * 32 * OL_NK;
    }
    SR_DEV const bf16* k_piece(int key, int g) const { return k + (size_t)key * 32 + g * 8; }
    SR_DEV const bf16* vt_piece(int d, int p8) const { return vt + (size_t)d * OL_NK + p8 * 8; }
};

SR_DEV Frag<bf16> ol_pack(const f32x4& a, const f32x4& b) {
    Frag<bf16> f;
    f.v[0] = (bf16)a[0]; f.v[1] = (bf16)a[1]; f.v[2] = (bf16)a[2]; f.v[3] = (bf16)a[3];
    f.v[4] = (bf16)b[0]; f.v[5] = (bf16)b[1]; f.v[6] = (bf16)b[2]; f.v[7] = (bf16)b[3];
    return f;
}

template <typename Src>
__global__ __launch_bounds__(256, 2) void sr_oca_lds_kernel(OcaLdsArgs a) {
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int QT = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    int block = __builtin_amdgcn_readfirstlane(blockIdx.x);
#if SR_OCA_XCD
    {   // neighbouring windows of one head (their 24 x 24-key neighbourhoods overlap 2.25 x) on one XCD
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = block & 7;
        block = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (block >> 3);
    }
#endif
    const int bwin = block % a.n_bwin, head = block / a.n_bwin;
    const Src src(a, bwin, head);
    const Frag<bf16>* Kl = reinterpret_cast<const Frag<bf16>*>(smem + OL_OFF_K);
    const Frag<bf16>* Vl = reinterpret_cast<const Frag<bf16>*>(smem + OL_OFF_V);
    float* TAB = reinterpret_cast<float*>(smem + OL_OFF_B);

    // ---- stage: 2304 16-byte pieces of K and of V^T, nine of each per thread, through registers (two rounds of 9 loads in flight)
    {
        Frag<bf16> r[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int c = i * 256 + threadIdx.x;  // piece c = (key, g)
            r[i] = *reinterpret_cast<const Frag<bf16>*>(src.k_piece(c >> 2, c & 3));
        }
        for (int i = threadIdx.x; i < OL_REL; i += 256) TAB[i] = a.rel[(size_t)head * OL_REL + i];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int c = i * 256 + threadIdx.x;
            const int key = c >> 2, g = c & 3;
            *reinterpret_cast<Frag<bf16>*>(smem + OL_OFF_K + ((key >> 4) * 64 + g * 16 + (key & 15)) * 16) = r[i];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int c = i * 256 + threadIdx.x;  // piece c = (d, p8): 72 pieces of 8 keys per feature row
            const int d = c / 72, p8 = c - d * 72;
            r[i] = *reinterpret_cast<const Frag<bf16>*>(src.vt_piece(d, p8));
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int c = i * 256 + threadIdx.x;
            const int d = c / 72, k0 = (c - d * 72) * 8;  // keys k0 .. k0 + 7: two groups of 4
            const int kb = k0 >> 6, ks = (k0 >> 5) & 1, e_hi = (k0 >> 4) & 1, g0 = (k0 >> 2) & 3;
            char* cell = smem + OL_OFF_V + ((((kb * 2 + (d >> 4)) * 2 + ks) * 64 + g0 * 16 + (d & 15)) * 16) + e_hi * 8;
            const bf16x8 v8 = r[i].v;
            *reinterpret_cast<bf16x4*>(cell) = __builtin_shufflevector(v8, v8, 0, 1, 2, 3);
            *reinterpret_cast<bf16x4*>(cell + 16 * 16) = __builtin_shufflevector(v8, v8, 4, 5, 6, 7);
        }
    }
    __syncthreads();

    // bias of tile (qt, kt) for this lane: keys 16 kt + 4 lg + r = key row ky, columns kx0 + r (a group of 4 never crosses a key row: 24 = 6 x 4);
    // rel index = (ky - qt + 15) * 39 + (kx0 - lr + 15) + r.  16 kt mod 24 cycles through 0, 16, 8: only tiles with kt % 3 == 1 straddle two key rows.
    const int off_k[3] = {4 * lg, lg < 2 ? 16 + 4 * lg : 39 + 4 * lg - 8, 8 + 4 * lg};
    const bf16* q = reinterpret_cast<const bf16*>(a.q) + ((size_t)bwin * a.heads + head) * 256 * 32;
    bf16* out = reinterpret_cast<bf16*>(a.out);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int qt0 = 4 * w + 2 * half;
        Frag<bf16> qf[QT];
        const float* trow[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            qf[t] = *reinterpret_cast<const Frag<bf16>*>(q + (size_t)((qt0 + t) * 16 + lr) * 32 + lg * 8);
            trow[t] = TAB + (15 - (qt0 + t)) * 39 + 15 - lr;
        }
        float m_run[QT], l_run[QT];
        f32x4 o[2][QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            m_run[t] = -3.0e38f;
            l_run[t] = 0.f;
            o[0][t] = (f32x4)(0.0f);
            o[1][t] = (f32x4)(0.0f);
        }
#pragma unroll
        for (int kb = 0; kb < OL_KB; ++kb) {
            f32x4 s[4][QT];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kt = 4 * kb + j;
                const int kofs = ((16 * kt) / 24) * 39 + off_k[kt % 3];
                const Frag<bf16> kf = Kl[kt * 64 + lane];
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    f32x4 b4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) b4[r] = trow[t][kofs + r];
                    s[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, qf[t].v, b4, 0, 0, 0);  // S^T[key 16 kt + 4 lg + r][query lr]
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
                o[0][t] *= alpha;
                o[1][t] *= alpha;
            }
            // ---- O^T += V^T P^T
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag<bf16> pf[QT];
#pragma unroll
                for (int t = 0; t < QT; ++t) pf[t] = ol_pack(s[2 * ks][t], s[2 * ks + 1][t]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const Frag<bf16> vf = Vl[((kb * 2 + dt) * 2 + ks) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < QT; ++t) mma(vf, pf[t], o[dt][t]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float l = wave_sum_xor(l_run[t], 16);
            l = wave_sum_xor(l, 32);
            const float inv = 1.0f / l;
            const int qi = (qt0 + t) * 16 + lr;
            bf16* dst = out + ((size_t)bwin * 256 + qi) * a.ldo + head * 32 + lg * 4;
            store4(dst, o[0][t] * inv);
            store4(dst + 16, o[1][t] * inv);
            if (a.lse && lg == 0) a.lse[((size_t)bwin * a.heads + head) * 256 + qi] = m_run[t] + __logf(l);
        }
    }
}

template <typename Src>
int ol_launch(const OcaLdsArgs& a, hipStream_t st, const char* what) {
    static SrDeviceOnce once;
    const hipError_t e = sr_once_per_device(once, [&] { return sr_allow_lds(sr_oca_lds_kernel<Src>, OL_LDS); });
    SR_REQUIRE(e == hipSuccess, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
    hipLaunchKernelGGL(sr_oca_lds_kernel<Src>, dim3(a.n_bwin * a.heads), dim3(256), OL_LDS, st, a);
    SR_CHECK_LAUNCH(what);
    return SR_OK;
}

bool ol_enabled() {
    static const bool on = !(getenv("SR_OCA_LDS") && atoi(getenv("SR_OCA_LDS")) == 0);
    return on;
}

}  // namespace

// inference: sr_oca_attention with SrOcaAttn.bias_rel (bf16, 16 x 16 windows, 24 x 24 neighbourhoods read in place from the zero-bordered layouts)
bool sr_oca_attention_lds_supported(const SrOcaAttn& o) {
    return ol_enabled() && o.bias_rel && o.dtype == SR_BF16 && o.hd_p == 32 && o.ws == 16 && o.pad == 4 && o.border == 4 && o.H % 16 == 0 && o.W % 16 == 0 &&
           ((reinterpret_cast<uintptr_t>(o.q) | reinterpret_cast<uintptr_t>(o.k) | reinterpret_cast<uintptr_t>(o.vt)) & 15) == 0;
}

int sr_oca_attention_lds(const SrOcaAttn& o, hipStream_t st) {
    OcaLdsArgs a{};
    a.q = o.q; a.k = o.k; a.vt = o.vt; a.rel = o.bias_rel; a.out = o.out; a.lse = nullptr;
    a.n_bwin = o.B * (o.H / 16) * (o.W / 16); a.heads = o.heads; a.ldo = o.heads * 32; a.H = o.H; a.W = o.W; a.border = o.border;
    return ol_launch<OlBordered>(a, st, "sr_oca_attention");
}

// training forward: sr_tr_attn_fwd with SrTrAttnFwd.bias_rel (unfolded keys / values)
bool sr_tr_attn_fwd_lds_supported(const SrTrAttnFwd& f) {
    return ol_enabled() && f.bias_rel && f.hd_p == 32 && f.Nq == 256 && f.Nk == OL_NK &&
           ((reinterpret_cast<uintptr_t>(f.q) | reinterpret_cast<uintptr_t>(f.k) | reinterpret_cast<uintptr_t>(f.vT)) & 15) == 0;
}

int sr_tr_attn_fwd_lds(const SrTrAttnFwd& f, hipStream_t st) {
    OcaLdsArgs a{};
    a.q = f.q; a.k = f.k; a.vt = f.vT; a.rel = f.bias_rel; a.out = f.out; a.lse = f.lse;
    a.n_bwin = f.n_bwin; a.heads = f.heads; a.ldo = f.ldo;
    return ol_launch<OlUnfolded>(a, st, "sr_tr_attn_fwd");
}
