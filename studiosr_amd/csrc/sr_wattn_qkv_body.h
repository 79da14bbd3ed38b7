// LayerNorm1 + the head's QKV projection + window attention of 16 x 16 windows as ONE workgroup role (round 4; hat.py:164-176 + 85-110):
//     q, k, v = qkv_h( LayerNorm1(x) ),   o = softmax(q k^T + bias[head] + shift mask) v          one (window, head) per 4-wave workgroup
// Why: with sr_hab_mid the attention role of a HAB's middle launch finishes in half the time of its CAB role (12.8 vs 19.7 us alone at 4 x 64 x 64
// tokens), while the next block's QKV projection rides on the latency chain of the previous sr_swin_tail (28.5 us with it, 20 without).  Here the
// attention workgroup projects its own head: per 64-token part of the window the four waves normalise 16 token rows each (statistics inside a
// wave: a lane holds 48 of a row's 192 channels, two cross-lane steps), write the K-group-major bf16 image, and run a 64 x 96 x 192 GEMM whose
// weight fragments -- 18 per wave: (q | k | v of one 16-feature half) x 6 K-chunks of the head, read from sr_swin_qkv's stream (LayerNorm affine,
// attention scale and the biases on the constant-one channels folded in: packing.pack_swin_qkv_stream) -- stay in registers for all four parts.  q, k
// and v^T go straight into the LDS fragment arrays the attention loop reads (sr_wattn_lds_body.h); the bias enters as the head's 31 x 31 relative-position
// TABLE (3.8 KiB, recovered from the gathered bias as in sr_tr_attn_lds.hip): a lane's four logits of a tile are four consecutive entries, the C operand
// of the S^T MFMA.  LDS: Q | K | V^T 3 x 16 KiB, image 24 KiB, table 4 KiB = 76 KiB: two workgroups per CU.
// Same arithmetic as sr_swin_qkv followed by the LDS-form attention: q / k / v are rounded to bf16 exactly there, the logits see the same bias values.
#pragma once
#include "sr_wattn_body.h"  // pack_p

namespace {

constexpr int WQ_OFF_K = 0, WQ_OFF_V = 16 * 1024, WQ_OFF_Q = 32 * 1024, WQ_OFF_A = 48 * 1024, WQ_OFF_T = 72 * 1024, WQ_LDS = 76 * 1024;

SR_DEV void wq_st_half(char* cell, int half, const f32x4& v) {
    bf16x4 r;
    r[0] = (bf16)v[0]; r[1] = (bf16)v[1]; r[2] = (bf16)v[2]; r[3] = (bf16)v[3];
    *reinterpret_cast<bf16x4*>(cell + half * 8) = r;
}

SR_DEV void wattn_qkv_block(const SrWindowAttn& a, const int block_id, char* smem) {
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int NTOK = 256, HD = 32, QT = 2, ONE = 180;  // ONE: first constant-one channel of the image (biases ride there, as in sr_swin_stream.h)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int bwin = block_id % a.n_bwin;
    const int head = block_id / a.n_bwin;
    const int nwx = a.W / 16, nwy = a.H / 16, nw = nwx * nwy;
    const int bimg = bwin / nw, win = bwin - bimg * nw;
    const int wy = win / nwx, wx = win - wy * nwx;
    Frag<bf16>* Kl = reinterpret_cast<Frag<bf16>*>(smem + WQ_OFF_K);
    Frag<bf16>* Vl = reinterpret_cast<Frag<bf16>*>(smem + WQ_OFF_V);
    Frag<bf16>* Ql = reinterpret_cast<Frag<bf16>*>(smem + WQ_OFF_Q);
    Frag<bf16>* Aimg = reinterpret_cast<Frag<bf16>*>(smem + WQ_OFF_A);
    float* TAB = reinterpret_cast<float*>(smem + WQ_OFF_T);

    // ---- this wave's weight fragments: (q | k | v) of feature half `hf` of the head, six K-chunks; stream slot (pass p = head / 2, chunk c), fragment 3 (2 hh + hf) + t
    const int hf = w & 1, mp = w >> 1;  // GEMM role: feature half, pair of token tiles
    Frag<bf16> wf[6][3];
    {
        const Frag<bf16>* ws = reinterpret_cast<const Frag<bf16>*>(a.wqkv) + (size_t)((head >> 1) * 6 * 12 + 3 * (2 * (head & 1) + hf)) * 64 + lane;
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int t = 0; t < 3; ++t) wf[c][t] = ws[(size_t)(c * 12 + t) * 64];
    }
    {   // the head's relative-position table from the gathered bias: entry (d, x) = bias[q][k] of any pair with window-row difference d - 15, column difference x - 15
        const float* bias = a.bias + (size_t)head * NTOK * NTOK;
        for (int i = threadIdx.x; i < 961; i += 256) {
            const int d = i / 31, x = i - d * 31;
            const int qy = d >= 15 ? d - 15 : 0, ky = d >= 15 ? 0 : 15 - d, qx = x >= 15 ? x - 15 : 0, kx = x >= 15 ? 0 : 15 - x;
            TAB[i] = bias[(size_t)(qy * 16 + qx) * NTOK + ky * 16 + kx];
        }
    }
    const int shift_y = a.y_mode == SR_Y_ROLL ? a.shift : 0;
    const float inv_c = 1.0f / (float)a.C;

    // ---- four parts of 64 tokens: LayerNorm1 -> image -> q, k, v^T fragments in LDS.  This wave: token rows 16 w .. 16 w + 15 of a part; lane (lr = token, lg):
    //      channels 16 n + 4 lg .. + 3, n = 0..11.  The rows of part p + 1 are requested as soon as part p's are in the image: they fly under its GEMM.
    f32x4 xv[12];
    auto load_rows = [&](int part) {
        const int tw = part * 64 + 16 * w + lr;  // token of the window (row tw >> 4, column tw & 15): roll + window_partition as one gather
        int y = wy * 16 + (tw >> 4) + shift_y, x = wx * 16 + (tw & 15) + a.shift;
        if (y >= a.H) y -= a.H;
        if (x >= a.W) x -= a.W;
        const float* xr = a.x + ((size_t)(bimg * a.H + y) * a.W + x) * a.ldx + 4 * lg;
#pragma unroll
        for (int n = 0; n < 12; ++n) xv[n] = *reinterpret_cast<const f32x4*>(xr + 16 * n);
    };
    load_rows(0);
#pragma unroll 1
    for (int part = 0; part < 4; ++part) {
        {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int n = 0; n < 12; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s1 += xv[n][r];  // pad channels of the stream are exactly 0
                    s2 = __builtin_fmaf(xv[n][r], xv[n][r], s2);
                }
            s1 = wave_sum_xor(s1, 16);
            s1 = wave_sum_xor(s1, 32);
            s2 = wave_sum_xor(s2, 16);
            s2 = wave_sum_xor(s2, 32);
            const float mean = s1 * inv_c;
            const float rstd = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + a.eps);
            const float nmr = -mean * rstd;
            if (part > 0) __syncthreads();  // the previous part's GEMM has read the image everywhere
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                f32x4 nv;
#pragma unroll
                for (int r = 0; r < 4; ++r) nv[r] = __builtin_fmaf(xv[n][r], rstd, nmr);
                if (n == ONE / 16 && lg == (ONE % 16) / 4) {
                    nv[0] = 1.0f;
                    nv[1] = 1.0f;
                }
                wq_st_half(reinterpret_cast<char*>(Aimg + (2 * n + (lg >> 1)) * 64 + 16 * w + lr), lg & 1, nv);
            }
        }
        if (part < 3) load_rows(part + 1);
        __syncthreads();
        // GEMM: token tiles 2 mp, 2 mp + 1 of the part x (q | k | v) of feature half hf
        f32x4 acc[2][3];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const Frag<bf16> av = Aimg[(4 * c + lg) * 64 + (2 * mp + m) * 16 + lr];
                const f32x4 z = (f32x4)(0.0f);
                acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c][0].v, av.v, c == 0 ? z : acc[m][0], 0, 0, 0);  // q: lane = token, registers = 4 features
                acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c][1].v, av.v, c == 0 ? z : acc[m][1], 0, 0, 0);  // k: likewise
                acc[m][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av.v, wf[c][2].v, c == 0 ? z : acc[m][2], 0, 0, 0);  // v: lane = feature, registers = 4 tokens
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int tile = part * 4 + 2 * mp + m;  // query / key tile of the window
            char* qc = reinterpret_cast<char*>(Ql + tile * 64 + (2 * hf + (lg >> 1)) * 16 + lr);
            char* kc = reinterpret_cast<char*>(Kl + tile * 64 + (2 * hf + (lg >> 1)) * 16 + lr);
            wq_st_half(qc, lg & 1, acc[m][0]);
            wq_st_half(kc, lg & 1, acc[m][1]);
            // v^T: feature 16 hf + lr, keys 64 part + 16 (2 mp + m) + 4 lg + r -> cell (kb = part, dt = hf, ks = mp, g = lg, i = lr), element (m * 4 + r)
            wq_st_half(reinterpret_cast<char*>(Vl + ((part * 2 + hf) * 2 + mp) * 64 + lg * 16 + lr), m, acc[m][2]);
        }
    }

    // shift mask (common.py:250-274) from window coordinates: ws = 16, so query tile qt is window row qt, key tile kt is window row kt
    const bool last_row = a.y_mode != SR_Y_STRIP && wy == nwy - 1, last_col = wx == nwx - 1;
    const bool masked = a.shift > 0 && (last_row || last_col);
    const int edge = 16 - a.shift;
    const bool qcol = last_col && lr >= edge;
    f32x4 cm;  // column term of this lane's 4 keys of any key tile (key column 4 lg + r) against its query column lr
#pragma unroll
    for (int r = 0; r < 4; ++r) cm[r] = (last_col && 4 * lg + r >= edge) != qcol ? -100.0f : 0.0f;
    __syncthreads();

    bf16* out = reinterpret_cast<bf16*>(a.out);
    const int ldo = a.heads * HD;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int qt0 = 4 * w + 2 * half;
        Frag<bf16> qf[QT];
        const float* trow[QT];  // logits S^T[key 16 kt + 4 lg + r][query lr]: table entries (qt - kt + 15) * 31 + 15 + lr - 4 lg - r
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            qf[t] = Ql[(qt0 + t) * 64 + lane];
            trow[t] = TAB + (qt0 + t + 15) * 31 + 15 + lr - 4 * lg;
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
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 s[4][QT];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kt = 4 * kb + j;
                const Frag<bf16> kf = Kl[kt * 64 + lane];
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    f32x4 b4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) b4[r] = trow[t][-kt * 31 - r];
                    s[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, qf[t].v, b4, 0, 0, 0);
                }
            }
            if (masked) {  // a real branch (most windows are not on the last window row / column)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool krow = last_row && (4 * kb + j) >= edge;
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        const bool qrow = last_row && (qt0 + t) >= edge;
                        s[j][t] += krow != qrow ? (f32x4)(-100.0f) : cm;
                    }
                }
            }
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
                l_run[t] = l_run[t] * alpha + sum;
                o[0][t] *= alpha;
                o[1][t] *= alpha;
            }
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
