// Window attention of 16 x 16 windows with every operand of the key loop in LDS (round 4; hat.py:85-110):
//     o = softmax(q k^T + bias[head] + shift mask) v     for one (window, head) per 4-wave workgroup, wave w = queries [64 w, 64 w + 64) in two passes of 32
// Why: the register-only flash form (sr_wattn_body.h) fetches K, V^T and the bias tiles of every key block from L2 per WAVE (8 waves per
// (window, head): 512 KB of L2 -> CU traffic for 32 KB of distinct K / V^T) one key block ahead -- less than an L2 round trip under load at two
// waves per SIMD, so every block stalls (profiles/r03_window_attention_ablation.txt: 3/4 of its time is neither exp, bias loads nor K re-reads).
// Here a workgroup stages, ONCE,
//   * K and V^T of its (window, head) in fragment order (2 x 16 KiB; a fragment = one conflict-free ds_read_b128 per lane), and
//   * the head's relative-position bias as its 31 DISTINCT 16 x 16 tiles (31 KiB): with 16-wide windows a query tile is one window row and a key
//     tile is one window row, so tile (qt, kt) of bias[q][k] = table[(qy - ky + 15) * 31 + (qx - kx + 15)] depends on qt - kt only
//     (SrWindowAttn.bias_tiles, packing.bias_distinct_tiles checks the property on the gathered bias before offering the operand).
// A wave's two query tiles qt0, qt0 + 1 at key tile kt need tiles d = qt0 + 15 + t - kt, t = 0, 1: stepping kt brings in ONE new tile, the other stays in
// registers (ring indexed by (t - kt) & 3, static after unrolling), and the tile is the C operand of the S^T MFMA (no bias add).
// Online softmax over key blocks of 64 as in the flash form; S^T accumulators are the P^T operand of O^T += V^T P^T.
#pragma once
#include "sr_wattn_body.h"  // pack_p

namespace {

constexpr int WL_OFF_K = 0, WL_OFF_V = 16 * 1024, WL_OFF_B = 32 * 1024, WL_NTILE = 31, WL_LDS = WL_OFF_B + WL_NTILE * 1024;

SR_DEV void wl_dma_1k(const void* src, unsigned lds_dst, int lane) {  // 64 lanes x 16 B: global (wave-uniform base) -> LDS at lds_dst + 16 lane, no VGPR staging
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane * 16), "s"(src), "s"(lds_dst)
                 : "memory");
}

template <bool FR>
SR_DEV void wattn_lds_block(const SrWindowAttn& a, const int block_id, char* smem) {
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int NTOK = 256, HD = 32, QT = 2;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int bwin = block_id % a.n_bwin;  // windows fastest: neighbouring workgroups stage the same head's bias tiles (L2)
    const int head = block_id / a.n_bwin;
    const size_t bh = (size_t)bwin * a.heads + head;
    const bf16* q = reinterpret_cast<const bf16*>(a.q) + bh * NTOK * HD;
    const bf16* k = reinterpret_cast<const bf16*>(a.k) + bh * NTOK * HD;
    const bf16* vt = reinterpret_cast<const bf16*>(a.vt) + bh * NTOK * HD;
    const float* btiles = a.bias_tiles + (size_t)head * WL_NTILE * 256;  // [31][lane][4] fp32, accumulator-fragment order
    const Frag<bf16>* Kl = reinterpret_cast<const Frag<bf16>*>(smem + WL_OFF_K);
    const Frag<bf16>* Vl = reinterpret_cast<const Frag<bf16>*>(smem + WL_OFF_V);
    const f32x4* Bl = reinterpret_cast<const f32x4*>(smem + WL_OFF_B);

    // ---- stage K, V^T (this wave: a quarter of each) and the bias tiles
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    if constexpr (FR) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wl_dma_1k(k + (size_t)(4 * w + i) * 512, __builtin_amdgcn_readfirstlane(lds0 + WL_OFF_K + (4 * w + i) * 1024), lane);
            wl_dma_1k(vt + (size_t)(4 * w + i) * 512, __builtin_amdgcn_readfirstlane(lds0 + WL_OFF_V + (4 * w + i) * 1024), lane);
        }
    } else {
        // row-major k [tok][32] -> cell (tile = tok >> 4, g = 16-B piece, i = tok & 15); row-major v^T [d][256] -> cell (kb, dt, ks, g, i = d & 15), element
        // (key >> 4 & 1) * 4 + (key & 3): the fragment order sr_swin_qkv writes (SrWindowAttn.qkv_frag)
        Frag<bf16> kr[4], vr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (4 * w + i) * 64 + lane;  // 16-B piece c of the row-major arrays
            kr[i] = *reinterpret_cast<const Frag<bf16>*>(k + (size_t)c * 8);
            vr[i] = *reinterpret_cast<const Frag<bf16>*>(vt + (size_t)c * 8);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (4 * w + i) * 64 + lane;
            {
                const int tok = c >> 2, g = c & 3;
                *reinterpret_cast<Frag<bf16>*>(smem + WL_OFF_K + (((tok >> 4) * 64 + g * 16 + (tok & 15)) * 16)) = kr[i];
            }
            {
                const int d = c >> 5, k0 = (c & 31) * 8;  // keys k0 .. k0 + 7 of feature row d: two 4-key groups
                const int kb = k0 >> 6, ks = (k0 >> 5) & 1, e_hi = (k0 >> 4) & 1, g0 = (k0 >> 2) & 3;  // g0 is 0 or 2; the second group is g0 + 1
                char* cell = smem + WL_OFF_V + ((((kb * 2 + (d >> 4)) * 2 + ks) * 64 + g0 * 16 + (d & 15)) * 16) + e_hi * 8;
                const bf16x8 v8 = vr[i].v;
                *reinterpret_cast<bf16x4*>(cell) = __builtin_shufflevector(v8, v8, 0, 1, 2, 3);
                *reinterpret_cast<bf16x4*>(cell + 16 * 16) = __builtin_shufflevector(v8, v8, 4, 5, 6, 7);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int d = w + 4 * i;
        if (d < WL_NTILE) wl_dma_1k(btiles + (size_t)d * 256, __builtin_amdgcn_readfirstlane(lds0 + WL_OFF_B + d * 1024), lane);
    }
    // shift mask (common.py:250-274) from window coordinates: ws = 16, so query tile qt is window row qt, key tile kt is window row kt
    const int nwx = a.W / 16, nwy = a.H / 16;
    const int win = bwin % (nwx * nwy);
    const int wy = win / nwx, wx = win - wy * nwx;
    const bool last_row = a.y_mode != SR_Y_STRIP && wy == nwy - 1, last_col = wx == nwx - 1;
    const bool masked = a.shift > 0 && (last_row || last_col);
    const int edge = 16 - a.shift;
    const bool qcol = last_col && lr >= edge;
    f32x4 cm;  // column term of this lane's 4 keys of any key tile (key column 4 lg + r) against its query column lr
#pragma unroll
    for (int r = 0; r < 4; ++r) cm[r] = (last_col && 4 * lg + r >= edge) != qcol ? -100.0f : 0.0f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    bf16* out = reinterpret_cast<bf16*>(a.out);
    const int ldo = a.heads * HD;
    // two passes of 32 queries (QT = 2 query tiles) per wave: 4 x 2 logit tiles live per 64-key block (four tiles at once spill ~140 registers at two workgroups per CU)
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int qt0 = 4 * w + 2 * half;
        Frag<bf16> qf[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            if constexpr (FR)
                qf[t] = *reinterpret_cast<const Frag<bf16>*>(q + (size_t)((qt0 + t) * 64 + lane) * 8);
            else
                qf[t] = *reinterpret_cast<const Frag<bf16>*>(q + (size_t)((qt0 + t) * 16 + lr) * HD + lg * 8);
        }
        f32x4 R[4];  // bias tiles: tile (t, kt) in R[(t - kt) & 3]; live at key tile kt: e = t - kt in {-kt, 1 - kt}, in flight -kt - 1, -kt - 2
        const f32x4* Bw = Bl + (qt0 + 15) * 64 + lane;  // tile e at Bw[e * 64]
#pragma unroll
        for (int e = 1; e >= -1; --e) R[e & 3] = Bw[e * 64];

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
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 s[4][QT];
            // ---- S^T tiles of this key block: bias tile (C operand) + K Q^T
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kt = 4 * kb + j;
                if (kt + 2 <= 15) R[(-kt - 2) & 3] = Bw[(-kt - 2) * 64];  // two key tiles ahead
                const Frag<bf16> kf = Kl[kt * 64 + lane];
#pragma unroll
                for (int t = 0; t < QT; ++t) s[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, qf[t].v, R[(t - kt) & 3], 0, 0, 0);
            }
            if (masked) {  // a real branch (most windows are not on the last window row / column): the asm statement keeps the body from being speculated
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool krow = last_row && (4 * kb + j) >= edge;
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        const bool qrow = last_row && (qt0 + t) >= edge;
                        s[j][t] += krow != qrow ? (f32x4)(-100.0f) : cm;  // min(row term, column term), both 0 or -100
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
                o[0][t] *= alpha;
                o[1][t] *= alpha;
            }
            // ---- O^T += V^T P^T  (32-key steps; the key order inside a step is the accumulator's on both operands)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag<bf16> pf[QT];
#pragma unroll
                for (int t = 0; t < QT; ++t) pf[t] = pack_p<bf16>(s[2 * ks][t], s[2 * ks + 1][t]);
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
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) store4(out + ((size_t)bwin * NTOK + qi) * ldo + head * HD + dt * 16 + lg * 4, o[dt][t] * inv);
        }
    }
}

}  // namespace
