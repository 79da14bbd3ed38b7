// Fast training path: backward of the (shifted-)window attention of 16 x 16 windows with every operand of the inner loops in LDS (round 4;
// swinir.py:83-102 / hat.py:90-107 under loss.backward(), trainer.py:104).  ONE launch, one (window, head) per 4-wave workgroup, instead of
// sr_tr_attn.hip's two register-only passes whose waves fetch K / V / Q / dO fragments and 256-float bias rows from L2 per 16 queries or 32 keys
// (profiles/r04_train_trace_HAT.txt: 85 + 60 us per HAB at 4 x 64 x 64 tokens -- 12 x the forward):
//   phase A (wave w = queries [64 w, +64) in four sub-passes of 16; K, V fragments and K^T in LDS):
//           S^T = K q^T + bias (+ mask) -> softmax (all 256 keys of a query in registers) -> lse, delta -> LDS;  dP^T = V dO^T;
//           dS = P (dP - delta);  dQ = dS K with the dS accumulators as the MFMA operand;  the bias gradient accumulates in 19 register tiles indexed by
//           (query tile - key tile) -- a 16 x 16 logit tile is one (query window row, key window row) pair, so its table rows depend on that difference
//           only (hat.py:480-492) -- and is folded into the workgroup's table-sized LDS partial once (lane rotation + 31 distinct adds per tile, as
//           sr_tr_attn.hip's VAR = 1);
//   phase B (wave w = keys [64 w, +64); Q, dO fragments and Q^T, dO^T in LDS, over phase A's regions):
//           S = q k^T + bias^T (+ mask), P = exp(S - lse), dP = dO v^T, dS = P (dP - delta);  dV = P^T dO, dK = dS^T Q.
// The bias enters both phases as the head's 31 x 31 TABLE in LDS (3.8 KiB, recovered from the gathered bias: entry (d, x) = bias[q][k] of any pair with
// window-row difference d - 15 and column difference x - 15): a lane's four logits of a tile are four consecutive table entries, read as the C operand of
// the S MFMA.  74 KiB of LDS, two workgroups per CU.  Layouts and outputs are sr_tr_attn_bwd's (include/studiosr_hip.h SrTrAttnBwd) with ONE table partial
// per (head, window): dtab_part [heads][n_bwin][Tpad], i.e. the caller's groups * 4 == n_bwin.
#include "sr_common.h"
#include "sr_host.h"
#include <cstdlib>

namespace {

constexpr int AL_R = 16 * 1024;                        // one fragment array: 16 fragments x 1 KiB
constexpr int AL_OFF_TAB = 4 * AL_R;                   // bias table of the head [961] (+ pad)
constexpr int AL_OFF_DTAB = AL_OFF_TAB + 4096;         // its gradient partial [961] (+ pad)
constexpr int AL_OFF_LSE = AL_OFF_DTAB + 4096;         // lse [256], delta [256]
constexpr int AL_LDS = AL_OFF_LSE + 2048;
static_assert(2 * AL_LDS <= 160 * 1024, "two workgroups per CU");

SR_DEV Frag<bf16> al_pack(const f32x4& a, const f32x4& b) {
    Frag<bf16> f;
    f.v[0] = (bf16)a[0]; f.v[1] = (bf16)a[1]; f.v[2] = (bf16)a[2]; f.v[3] = (bf16)a[3];
    f.v[4] = (bf16)b[0]; f.v[5] = (bf16)b[1]; f.v[6] = (bf16)b[2]; f.v[7] = (bf16)b[3];
    return f;
}
SR_DEV f32x4 al_mma(const Frag<bf16>& x, const Frag<bf16>& y, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.v, y.v, c, 0, 0, 0); }

// Row-major [256 rows][32] bf16 (row stride `ld` elements) -> 16 operand fragments [tile = row >> 4][lane = 16 g + (row & 15)][8] (g = 16-B piece of the row);
// this wave's quarter (pieces [256 w, +256)) through registers
SR_DEV void al_stage_rows(const bf16* src, size_t ld, int w, int lane, Frag<bf16> (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (4 * w + i) * 64 + lane;
        r[i] = *reinterpret_cast<const Frag<bf16>*>(src + (size_t)(c >> 2) * ld + (c & 3) * 8);
    }
}
SR_DEV void al_commit_rows(char* dst, int w, int lane, const Frag<bf16> (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (4 * w + i) * 64 + lane;
        const int row = c >> 2, g = c & 3;
        *reinterpret_cast<Frag<bf16>*>(dst + ((row >> 4) * 64 + g * 16 + (row & 15)) * 16) = r[i];
    }
}
// Transposed copy [32 d][256] bf16 -> 16 fragments [dt = d >> 4][step of 32 columns][lane = 16 g + (d & 15)][8], element e = column 32 step + 16 (e >> 2) + 4 g + (e & 3):
// the column order of an accumulator-as-operand (pack of two logit tiles)
SR_DEV void al_stage_t(const bf16* src, int w, int lane, Frag<bf16> (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const Frag<bf16>*>(src + (size_t)((4 * w + i) * 64 + lane) * 8);
}
SR_DEV void al_commit_t(char* dst, int w, int lane, const Frag<bf16> (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (4 * w + i) * 64 + lane;
        const int d = c >> 5, k0 = (c & 31) * 8;  // columns k0 .. k0 + 7 of row d: two groups of 4
        const int st = k0 >> 5, e_hi = (k0 >> 4) & 1, g0 = (k0 >> 2) & 3;
        char* cell = dst + ((((d >> 4) * 8 + st) * 64 + g0 * 16 + (d & 15)) * 16) + e_hi * 8;
        const bf16x8 v8 = r[i].v;
        *reinterpret_cast<bf16x4*>(cell) = __builtin_shufflevector(v8, v8, 0, 1, 2, 3);
        *reinterpret_cast<bf16x4*>(cell + 16 * 16) = __builtin_shufflevector(v8, v8, 4, 5, 6, 7);
    }
}

__global__ __launch_bounds__(256, 2) void sr_tr_attn_bwd_lds_kernel(SrTrAttnBwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int block = __builtin_amdgcn_readfirstlane(blockIdx.x);
    const int bwin = block % a.n_bwin, head = block / a.n_bwin;  // head-major: a head's table partials are consecutive
    const size_t bh = (size_t)bwin * a.heads + head;
    const bf16* q = reinterpret_cast<const bf16*>(a.q) + bh * 256 * 32;
    const bf16* k = reinterpret_cast<const bf16*>(a.k) + bh * 256 * 32;
    const bf16* v = reinterpret_cast<const bf16*>(a.v) + bh * 256 * 32;
    const bf16* qT = reinterpret_cast<const bf16*>(a.qT) + bh * 256 * 32;
    const bf16* kT = reinterpret_cast<const bf16*>(a.kT) + bh * 256 * 32;
    const bf16* dOT = reinterpret_cast<const bf16*>(a.dOT) + bh * 256 * 32;
    const bf16* orow = reinterpret_cast<const bf16*>(a.o) + (size_t)bwin * 256 * a.ldo + head * 32;
    const bf16* dorow = reinterpret_cast<const bf16*>(a.dO) + (size_t)bwin * 256 * a.ldo + head * 32;
    float* TAB = reinterpret_cast<float*>(smem + AL_OFF_TAB);
    float* DTAB = reinterpret_cast<float*>(smem + AL_OFF_DTAB);
    float* LSE = reinterpret_cast<float*>(smem + AL_OFF_LSE);
    float* DELTA = LSE + 256;
    const Frag<bf16>* R0 = reinterpret_cast<const Frag<bf16>*>(smem);
    const Frag<bf16>* R1 = reinterpret_cast<const Frag<bf16>*>(smem + AL_R);
    const Frag<bf16>* R2 = reinterpret_cast<const Frag<bf16>*>(smem + 2 * AL_R);
    const Frag<bf16>* R3 = reinterpret_cast<const Frag<bf16>*>(smem + 3 * AL_R);

    // shift mask (common.py:250-274): ws = 16, so a query / key tile is one window row
    const int nwx = a.W / 16, nwy = a.H / 16;
    const int win = bwin % (nwx * nwy);
    const int wy = win / nwx, wx = win - wy * nwx;
    const bool last_row = wy == nwy - 1, last_col = wx == nwx - 1;
    const bool masked = a.shift > 0 && (last_row || last_col);
    const int edge = 16 - a.shift;

    // ---- stage phase A's operands: K, V fragments (R0, R1), K^T (R2); the head's bias table and a zeroed gradient partial
    {
        Frag<bf16> r0[4], r1[4], r2[4];
        al_stage_rows(k, 32, w, lane, r0);
        al_stage_rows(v, 32, w, lane, r1);
        al_stage_t(kT, w, lane, r2);
        const float* bias = a.bias + (size_t)head * 256 * 256;
        for (int i = threadIdx.x; i < 961; i += 256) {
            const int d = i / 31, x = i - d * 31;
            const int qy = d >= 15 ? d - 15 : 0, ky = d >= 15 ? 0 : 15 - d, qx = x >= 15 ? x - 15 : 0, kx = x >= 15 ? 0 : 15 - x;
            TAB[i] = bias[(size_t)(qy * 16 + qx) * 256 + ky * 16 + kx];
            DTAB[i] = 0.f;
        }
        al_commit_rows(smem, w, lane, r0);
        al_commit_rows(smem + AL_R, w, lane, r1);
        al_commit_t(smem + 2 * AL_R, w, lane, r2);
    }
    __syncthreads();

#if defined(SR_AL_EXP) && SR_AL_EXP == 1  // experiment: staging only
    if (a.n_bwin > 0) return;
#endif
    // ---- phase A: this wave's 64 queries
    f32x4 dbacc[19];  // bias-gradient tiles by e = t - kt + 15 (t = sub-pass)
#pragma unroll
    for (int e = 0; e < 19; ++e) dbacc[e] = (f32x4)(0.0f);
    {
        f32x4 cmA;  // column term of the mask for this lane's 4 keys (key column 4 lg + r) against its query column lr
        const bool qcol = last_col && lr >= edge;
#pragma unroll
        for (int r = 0; r < 4; ++r) cmA[r] = (last_col && 4 * lg + r >= edge) != qcol ? -100.0f : 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            const int qt = 4 * w + t;
            const int qi = qt * 16 + lr;
            const Frag<bf16> qf = *reinterpret_cast<const Frag<bf16>*>(q + (size_t)qi * 32 + lg * 8);
            const Frag<bf16> dof = *reinterpret_cast<const Frag<bf16>*>(dorow + (size_t)qi * a.ldo + lg * 8);
            const Frag<bf16> of = *reinterpret_cast<const Frag<bf16>*>(orow + (size_t)qi * a.ldo + lg * 8);
            // logits: the four table entries of this lane are consecutive, descending with the key column: x = lr - (4 lg + r) + 15
            const float* trow = TAB + (qt + 15) * 31 + 15 + lr - 4 * lg;
            f32x4 s[16];
#pragma unroll
            for (int kt = 0; kt < 16; ++kt) {
                f32x4 b4;
#pragma unroll
                for (int r = 0; r < 4; ++r) b4[r] = trow[-kt * 31 - r];
                s[kt] = al_mma(R0[kt * 64 + lane], qf, b4);  // S^T[key 16 kt + 4 lg + r][query lr]
                if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if (masked) {
                asm volatile("" ::: "memory");  // a real branch
                const bool qrow = last_row && qt >= edge;
#pragma unroll
                for (int kt = 0; kt < 16; ++kt) s[kt] += (last_row && kt >= edge) != qrow ? (f32x4)(-100.0f) : cmA;
            }
            float mx = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 16; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
            mx = wave_max_xor(mx, 16);
            mx = wave_max_xor(mx, 32);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 16; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][r] = __expf(s[kt][r] - mx);
                    sum += s[kt][r];
                }
            sum = wave_sum_xor(sum, 16);
            sum = wave_sum_xor(sum, 32);
            const float inv = 1.0f / sum;
            float dl = 0.f;  // delta = sum_d dO[q][d] O[q][d]
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) dl += (float)dof.v[jj] * (float)of.v[jj];
            dl = wave_sum_xor(dl, 16);
            dl = wave_sum_xor(dl, 32);
            if (lg == 0) {
                LSE[qi] = mx + __logf(sum);
                DELTA[qi] = dl;
            }
            __builtin_amdgcn_sched_barrier(0);
            // dS^T = P o (V dO^T - delta)
#pragma unroll
            for (int k4 = 0; k4 < 16; k4 += 4) {  // four dP tiles in flight before their consumers (an MFMA result read right behind its issue stalls the wave)
                f32x4 dp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) dp[j] = al_mma(R1[(k4 + j) * 64 + lane], dof, (f32x4)(0.0f));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kt = k4 + j;
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[kt][r] = s[kt][r] * inv * (dp[j][r] - dl);
                    dbacc[t - kt + 15] += s[kt];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // dQ[q][d] = sum_key dS[q][key] K[key][d]: the dS accumulators are the operand (keys 32 ks + 4 lg + r | + 16)
            f32x4 dq[2] = {(f32x4)(0.0f), (f32x4)(0.0f)};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const Frag<bf16> pf = al_pack(s[2 * ks], s[2 * ks + 1]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dq[dt] = al_mma(R2[(dt * 8 + ks) * 64 + lane], pf, dq[dt]);  // C[d = 16 dt + 4 lg + r][query lr]
                if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
            bf16* dqp = reinterpret_cast<bf16*>(a.dq) + (bh * 256 + qi) * 32 + lg * 4;
            store4(dqp, dq[0]);
            store4(dqp + 16, dq[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#if defined(SR_AL_EXP) && SR_AL_EXP == 2  // experiment: phase A without the fold
    if (a.n_bwin > 0) {
        f32x4 t_ = dbacc[0];
        for (int e = 1; e < 19; ++e) t_ += dbacc[e];
        if (t_[0] == 1.2345f) DTAB[lane] = t_[1] + t_[2] + t_[3];
        return;
    }
#endif
    // fold the bias-gradient tiles along their diagonals into the table partial: tile e holds (query tile 4 w + t, key tile kt) pairs with t - kt + 15 = e,
    // i.e. table rows (4 w + e) * 31 ..; rotate every key column so that lane = (xq - xk) mod 16, split wrapped / unwrapped, reduce over the lane groups
#pragma unroll
    for (int e = 0; e < 19; ++e) {
        float pos = 0.f, neg = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int xk = 4 * lg + r;
            const float wv = __shfl(dbacc[e][r], (lane & 48) | ((lr + xk) & 15), 64);
            const bool p = lr + xk <= 15;
            pos += p ? wv : 0.f;
            neg += p ? 0.f : wv;
        }
        pos = wave_sum_xor(pos, 16);
        pos = wave_sum_xor(pos, 32);
        neg = wave_sum_xor(neg, 16);
        neg = wave_sum_xor(neg, 32);
        const int row = (4 * w + e) * 31 + 15;  // tile difference (4 w + t) - kt = 4 w + e - 15
        if (lg == 0 && 4 * w + e <= 30) {
            atomicAdd(&DTAB[row + lr], pos);
            if (lr >= 1) atomicAdd(&DTAB[row + lr - 16], neg);
        }
    }

#if defined(SR_AL_EXP) && SR_AL_EXP == 3  // experiment: phase A with the fold
    if (a.n_bwin > 0) return;
#endif
    // ---- phase B: this wave's 64 keys.  Its K / V fragments and the next operands' global loads are issued before the barrier
    __builtin_amdgcn_sched_barrier(0);  // (after the fold: the 19 gradient tiles are dead)
    Frag<bf16> kf[4], vf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int ki = (4 * w + u) * 16 + lr;
        kf[u] = *reinterpret_cast<const Frag<bf16>*>(k + (size_t)ki * 32 + lg * 8);
        vf[u] = *reinterpret_cast<const Frag<bf16>*>(v + (size_t)ki * 32 + lg * 8);
    }
    {
        Frag<bf16> r0[4], r1[4], r2[4], r3[4];
        al_stage_rows(q, 32, w, lane, r0);
        al_stage_rows(dorow, (size_t)a.ldo, w, lane, r1);
        al_stage_t(qT, w, lane, r2);
        al_stage_t(dOT, w, lane, r3);
        __syncthreads();  // every wave is done with K / V / K^T in LDS; lse / delta / the table partial are complete
        al_commit_rows(smem, w, lane, r0);
        al_commit_rows(smem + AL_R, w, lane, r1);
        al_commit_t(smem + 2 * AL_R, w, lane, r2);
        al_commit_t(smem + 3 * AL_R, w, lane, r3);
    }
    {  // the workgroup's table partial leaves while phase B runs
        float* dst = a.dtab_part + (size_t)block * a.Tpad;
        for (int i = threadIdx.x; i < 961; i += 256) dst[i] = DTAB[i];
    }
    __syncthreads();
    {
        f32x4 dk[4][2], dv[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) dk[u][0] = dk[u][1] = dv[u][0] = dv[u][1] = (f32x4)(0.0f);
        const bool kcol = last_col && lr >= edge;  // this lane's key column
        f32x4 cmB;  // column term of the mask: this lane's key column lr against its 4 query columns 4 lg + r
#pragma unroll
        for (int r = 0; r < 4; ++r) cmB[r] = (last_col && 4 * lg + r >= edge) != kcol ? -100.0f : 0.0f;
        // logits S[query 16 qt + 4 lg + r][key lr]: table entries x = (4 lg + r) - lr + 15, ascending with r
        const float* trow = TAB + 15 * 31 + 15 + 4 * lg - lr;
#pragma unroll 2
        for (int qs = 0; qs < 8; ++qs) {
            f32x4 p[4][2], ds[4][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qt = 2 * qs + h;
                const Frag<bf16> qf = R0[qt * 64 + lane], dof = R1[qt * 64 + lane];
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(LSE + qt * 16 + 4 * lg), d4 = *reinterpret_cast<const f32x4*>(DELTA + qt * 16 + 4 * lg);
                const bool qrow = last_row && qt >= edge;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kt = 4 * w + u;
                    f32x4 b4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) b4[r] = trow[(qt - kt) * 31 + r];
                    f32x4 s = al_mma(qf, kf[u], b4);
                    const f32x4 dp = al_mma(dof, vf[u], (f32x4)(0.0f));
                    if (masked) {
                        asm volatile("" ::: "memory");  // a real branch (most windows carry no mask)
                        s += qrow != (last_row && kt >= edge) ? (f32x4)(-100.0f) : cmB;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p[u][h][r] = __expf(s[r] - l4[r]);
                        ds[u][h][r] = p[u][h][r] * (dp[r] - d4[r]);
                    }
                }
            }
            Frag<bf16> dotf[2], qtf[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                qtf[dt] = R2[(dt * 8 + qs) * 64 + lane];
                dotf[dt] = R3[(dt * 8 + qs) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const Frag<bf16> pf = al_pack(p[u][0], p[u][1]), dsf = al_pack(ds[u][0], ds[u][1]);  // row = key lr, k = queries 32 qs + 4 lg + r | + 16
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[u][dt] = al_mma(dotf[dt], pf, dv[u][dt]);  // C[d = 16 dt + 4 lg + r][key lr]
                    dk[u][dt] = al_mma(qtf[dt], dsf, dk[u][dt]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ki = (4 * w + u) * 16 + lr;
            bf16* dkp = reinterpret_cast<bf16*>(a.dk) + (bh * 256 + ki) * 32 + lg * 4;
            bf16* dvp = reinterpret_cast<bf16*>(a.dv) + (bh * 256 + ki) * 32 + lg * 4;
            store4(dkp, dk[u][0]);
            store4(dkp + 16, dk[u][1]);
            store4(dvp, dv[u][0]);
            store4(dvp + 16, dv[u][1]);
        }
    }
}

}  // namespace

bool sr_tr_attn_bwd_lds_usable(const SrTrAttnBwd& a) {
    static const bool on = !(getenv("SR_TR_ATTN_LDS") && atoi(getenv("SR_TR_ATTN_LDS")) == 0);
    return on && a.Nq == 256 && a.Nk == 256 && a.hd_p == 32 && a.toeplitz16 && a.T == 961 && a.groups * 4 == a.n_bwin;
}

int sr_tr_attn_bwd_lds(const SrTrAttnBwd& a, hipStream_t st) {
    static SrDeviceOnce once;
    const hipError_t e = sr_once_per_device(once, [&] { return sr_allow_lds(sr_tr_attn_bwd_lds_kernel, AL_LDS); });
    SR_REQUIRE(e == hipSuccess, "sr_tr_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(sr_tr_attn_bwd_lds_kernel, dim3(a.n_bwin * a.heads), dim3(256), AL_LDS, st, a);
    SR_CHECK_LAUNCH("sr_tr_attn_bwd (lds)");
    return SR_OK;
}
