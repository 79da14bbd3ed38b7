// Fast training path (C ABI v7): backward of (shifted-)window attention and of HAT's overlapping cross attention, flash form
// (swinir.py:83-102, hat.py:90-107, 266-283 under loss.backward(), trainer.py:104).  Nothing of the N x N logits is stored by the forward:
// both passes recompute P = softmax(q k^T + bias + mask) from q, k and the bias table, in registers, exactly like the forward kernel
// sr_window_attn_kernel (sr_attn.hip) -- one wave per item, operands straight from the [.., tok, 32] buffers, no LDS:
//   pass Q  (item = window group x head x 16 queries; the wave walks the windows of its group)
//           S^T = K Q^T + bias + mask -> softmax -> (lse);  dP^T = V dO^T;  delta = rowsum(dO o O);  dS = P (dP - delta);
//           dQ = dS K (the dS accumulator is the MFMA operand, K^T comes from the transposed copy);  the bias gradient
//           sum_windows dS stays in registers across the walk and leaves as ONE partial per group (no atomics, deterministic)
//   pass KV (item = window x head x 16 keys)   S = Q K^T (key on the lane) + bias^T + mask, P = exp(S - lse), dP = dO V^T, dS = P (dP - delta);
//           dV = P^T dO, dK = dS^T Q with the P / dS accumulators as operands and dO^T / Q^T from the transposed copies.
// q is pre-scaled by hd^-0.5 (the scale lives in the packed Wq), so no scale appears here.  Layouts (bf16 unless noted):
//   q, k, v, dq, dk, dv [bwin][head][N][32];  qT, kT, dOT [bwin][head][32][N];  o, dO rows [bwin*Nq + tok][ldo], head at column 32*head;
//   bias [heads][Nq][Nk] fp32, biasT [heads][Nk][Nq] fp32;  lse, delta [bwin][head][Nq] fp32;  dbias partial [groups][heads][Nq][Nk] fp32.
#include "sr_common.h"
#include "sr_host.h"
#include <cstdlib>

namespace {

SR_DEV int region(int v, int size, int ws, int shift) { return v < size - ws ? 0 : (v < size - shift ? 1 : 2); }

SR_DEV Frag<bf16> load_2x4(const bf16* p0, const bf16* p1) {
    bf16x4 a = *reinterpret_cast<const bf16x4*>(p0);
    bf16x4 b = *reinterpret_cast<const bf16x4*>(p1);
    Frag<bf16> f;
    f.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return f;
}
SR_DEV Frag<bf16> pack_p(const f32x4& a, const f32x4& b) {
    Frag<bf16> f;
    f.v[0] = (bf16)a[0]; f.v[1] = (bf16)a[1]; f.v[2] = (bf16)a[2]; f.v[3] = (bf16)a[3];
    f.v[4] = (bf16)b[0]; f.v[5] = (bf16)b[1]; f.v[6] = (bf16)b[2]; f.v[7] = (bf16)b[3];
    return f;
}
SR_DEV f32x4 mma_z(const Frag<bf16>& x, const Frag<bf16>& y) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.v, y.v, (f32x4)(0.0f), 0, 0, 0); }

struct Geo {
    int nwx, nwy, win_per_img;
};

// Bias-table gradient of the overlapping cross attention (hat.py:494-517), folded from the S^T-layout accumulators dbacc[kt] (key 16 kt + 4 lg + r, query 16 qt + lr) of one wave
// into the workgroup's LDS table `tab`:
// Overlapping cross attention (hat.py:494-517): table entry j = (ky - qy + 15) * 39 + (kx - qx + 15) with (ky, kx) the key's position in the 24 x 24 neighbourhood;
// row j - 880, negative rows wrap.  The index-map fold below is 144 ds_add_f32 per wave with up to four lanes per address -- LDS float atomics run lane by lane:
// 40 us of this 145-us launch (a build without them).  Here, as for the 16 x 16 self-attention windows above: rotating the sixteen query columns by the key's
// column inside the tile puts (kx - qx) on the lane column (no wrap: kx0 - lr, wrapped: + 16), the sums over r and the lane groups follow, and 16 lanes add two
// DISTINCT entries per tile.  A tile is 16 consecutive keys of a 24-key row: tiles with kt % 3 == 1 straddle two rows (lane groups 0, 1 | 2, 3).
template <int KT, int KT0 = 0>  // dbacc[i] = key tile KT0 + i
SR_DEV void oca_table_fold(const f32x4 (&dbacc)[KT], float* tab, const int qt, const int lane, const int T) {
    const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int kti = 0; kti < KT; ++kti) {
        const int kt = KT0 + kti;
        float pos = 0.f, neg = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * lg + r;
            const float w = __shfl(dbacc[kti][r], (lane & 48) | ((lr + c) & 15), 64);
            const bool p = lr + c <= 15;
            pos += p ? w : 0.f;
            neg += p ? 0.f : w;
        }
        pos = wave_sum_xor(pos, 16);
        neg = wave_sum_xor(neg, 16);
        const int ky = (16 * kt) / 24;
        int kyr, kx0;
        bool writer;
        if (kt % 3 == 1) {  // keys 16 .. 23 of row ky (lane groups 0, 1) | keys 0 .. 7 of row ky + 1 (lane groups 2, 3)
            kyr = lg < 2 ? ky : ky + 1;
            kx0 = lg < 2 ? 16 : -8;
            writer = (lg & 1) == 0;
        } else {
            pos = wave_sum_xor(pos, 32);
            neg = wave_sum_xor(neg, 32);
            kyr = ky;
            kx0 = kt % 3 == 0 ? 0 : 8;
            writer = lg == 0;
        }
        if (writer) {
            const int dxp = kx0 - lr, dxn = dxp + 16;  // kx - qx of the two sums
            const int row = (kyr - qt + 15) * 39 + 15 - 880;
            if (dxp >= -15 && dxp <= 23) {
                const int t = row + dxp;
                atomicAdd(&tab[t < 0 ? t + T : t], pos);
            }
            if (dxn >= -15 && dxn <= 23) {
                const int t = row + dxn;
                atomicAdd(&tab[t < 0 ? t + T : t], neg);
            }
        }
    }
}

#ifndef SR_OCAQ_ATOMIC_FOLD
#define SR_OCAQ_ATOMIC_FOLD 0  // 1: the overlapping cross attention's table fold as 144 index-map LDS atomics per wave (before round 5: +40 us per launch)
#endif
// ---- pass Q.  KT = key tiles; Nq = 16 * QT_ALL query rows per (window, head)
// LDSK (round 4, the 576-key neighbourhoods of the overlapping cross attention): the four waves of a workgroup work on the same (head, window) at the same
// time, so K and V (as operand fragments) and K^T (in the accumulator-as-operand key order) of the window are staged in LDS once per window walk step
// (3 x KT KiB, two barriers per window) instead of being fetched from L2 by every wave -- at one wave per SIMD (288 registers of logits and gradient tiles)
// those 108 fragment loads per window were exposed L2 round trips: 436 -> see profiles/r04_train_trace_HAT.txt.
// WS = window side of the shift mask (16: HAT; 8: SwinIR, 64 keys = KT 4 -- a 16-key tile is then two window rows).
template <int KT, int VAR, bool LDSK = false, int WS = 16>
__global__ __launch_bounds__(256, KT > 16 ? 1 : 2) void sr_tr_attn_bwd_q_kernel(SrTrAttnBwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem_q[];  // LDSK: K | V | K^T fragments
    const Frag<bf16>* KL = reinterpret_cast<const Frag<bf16>*>(smem_q);
    const Frag<bf16>* VL = KL + KT * 64;
    const Frag<bf16>* KTL = VL + KT * 64;
    float* REL = reinterpret_cast<float*>(smem_q + 3 * KT * 1024);  // LDSK: the head's relative-position table [39 * 39] (SrTrAttnBwd.oca_rel)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NK = KT * 16;
    const int qtiles = a.Nq >> 4;
    const int item = blockIdx.x * 4 + wave;
    const int n_items = a.groups * a.heads * qtiles;
    if (item >= n_items) return;  // never taken: the launcher requires Nq / 16 to be a multiple of 4 (the table reduction below has barriers)
    const int qt = item % qtiles;
    const int gh = item / qtiles;
    const int grp = gh % a.groups, head = gh / a.groups;  // head-major: the 4 waves of a workgroup share (head, group); a head's workgroups are consecutive
    const int lr = lane & 15, lg = lane >> 4;
    const int wpg = (a.n_bwin + a.groups - 1) / a.groups;
    const int qi = qt * 16 + lr;

    const float* bias = a.bias + ((size_t)head * a.Nq + qi) * NK;
    f32x4 dbacc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) dbacc[kt] = (f32x4)(0.0f);

    const int nwx = a.W / a.ws, nwy = a.H / a.ws;
    // LDSK: rel index of this lane's logits of tile kt = (ky - qt + 15) * 39 + (kx0 - lr + 15) + r with (ky, kx0) the key row / column of key 16 kt + 4 lg
    // (a group of 4 keys never crosses a key row: 24 = 6 x 4; 16 kt mod 24 cycles through 0, 16, 8: only tiles with kt % 3 == 1 straddle two rows)
    const int off_k[3] = {4 * lg, lg < 2 ? 16 + 4 * lg : 39 + 4 * lg - 8, 8 + 4 * lg};
    const int rel0 = (15 - qt) * 39 + 15 - lr;
    if constexpr (LDSK) {
        const float* bh_ = a.bias + (size_t)head * a.Nq * NK;
        for (int j = threadIdx.x; j < 39 * 39; j += 256) {  // any (q, k) pair with the row / column differences of entry j
            const int dyp = j / 39, dxp = j - dyp * 39;
            const int qy = dyp >= 15 ? 0 : 15 - dyp, ky = dyp >= 15 ? dyp - 15 : 0, qx = dxp >= 15 ? 0 : 15 - dxp, kx = dxp >= 15 ? dxp - 15 : 0;
            REL[j] = bh_[(size_t)(qy * 16 + qx) * NK + ky * 24 + kx];
        }
    }
    for (int wi = 0; wi < wpg; ++wi) {
        const int bwin = grp * wpg + wi;
        if (bwin >= a.n_bwin) break;
        const size_t bh = (size_t)bwin * a.heads + head;
        const bf16* q = reinterpret_cast<const bf16*>(a.q) + bh * a.Nq * 32;
        const bf16* k = reinterpret_cast<const bf16*>(a.k) + bh * NK * 32;
        const bf16* kT = reinterpret_cast<const bf16*>(a.kT) + bh * NK * 32;
        const bf16* v = reinterpret_cast<const bf16*>(a.v) + bh * NK * 32;
        if constexpr (LDSK) {
            constexpr int PIECES = KT * 64, PER = PIECES / 256;  // 16-byte pieces per array, per thread
            static_assert(PIECES % 256 == 0, "whole rounds of 256 pieces");
            __syncthreads();  // the previous window's fragments have been read by every wave
            Frag<bf16> r[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) r[i] = *reinterpret_cast<const Frag<bf16>*>(k + (size_t)(i * 256 + threadIdx.x) * 8);
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int c = i * 256 + threadIdx.x, key = c >> 2, g = c & 3;
                *reinterpret_cast<Frag<bf16>*>(smem_q + ((key >> 4) * 64 + g * 16 + (key & 15)) * 16) = r[i];
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) r[i] = *reinterpret_cast<const Frag<bf16>*>(v + (size_t)(i * 256 + threadIdx.x) * 8);
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int c = i * 256 + threadIdx.x, key = c >> 2, g = c & 3;
                *reinterpret_cast<Frag<bf16>*>(smem_q + (KT * 64 + (key >> 4) * 64 + g * 16 + (key & 15)) * 16) = r[i];
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) r[i] = *reinterpret_cast<const Frag<bf16>*>(kT + (size_t)(i * 256 + threadIdx.x) * 8);
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int c = i * 256 + threadIdx.x;
                const int d = c / (NK / 8), k0 = (c - d * (NK / 8)) * 8;  // keys k0 .. k0 + 7 of feature row d: two groups of 4
                const int st = k0 >> 5, e_hi = (k0 >> 4) & 1, g0 = (k0 >> 2) & 3;
                char* cell = smem_q + (2 * KT * 64 + ((d >> 4) * (KT / 2) + st) * 64 + g0 * 16 + (d & 15)) * 16 + e_hi * 8;
                const bf16x8 v8 = r[i].v;
                *reinterpret_cast<bf16x4*>(cell) = __builtin_shufflevector(v8, v8, 0, 1, 2, 3);
                *reinterpret_cast<bf16x4*>(cell + 16 * 16) = __builtin_shufflevector(v8, v8, 4, 5, 6, 7);
            }
            __syncthreads();
        }
        const Frag<bf16> qf = *reinterpret_cast<const Frag<bf16>*>(q + (size_t)qi * 32 + lg * 8);
        const size_t orow = ((size_t)bwin * a.Nq + qi) * a.ldo + head * 32 + lg * 8;
        const Frag<bf16> dof = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.dO) + orow);
        const Frag<bf16> of = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.o) + orow);

        f32x4 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if constexpr (LDSK) {  // the four table entries of this lane are consecutive: the C operand of the MFMA
                const float* tr = REL + rel0 + ((16 * kt) / 24) * 39 + off_k[kt % 3];
                f32x4 b4;
#pragma unroll
                for (int r = 0; r < 4; ++r) b4[r] = tr[r];
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(KL[kt * 64 + lane].v, qf.v, b4, 0, 0, 0);
                if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            const Frag<bf16> kf = *reinterpret_cast<const Frag<bf16>*>(k + (size_t)(kt * 16 + lr) * 32 + lg * 8);
            s[kt] = mma_z(kf, qf);  // S^T[key 16 kt + 4 lg + r][query lr]
            if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keeps hipcc from hoisting every operand load of the pass at once (spills)
        }
        // + bias (+ shift mask: key 16 kt + 4 lg + r sits at window row kt, column 4 lg + r of a 16 x 16 window; row 2 kt + (lg >> 1), column 4 (lg & 1) + r of an 8 x 8 one)
        const int win = bwin % (nwx * nwy);
        const int wy = win / nwx, wx = win - wy * nwx;
        const bool masked = a.shift > 0 && (wy == nwy - 1 || wx == nwx - 1);
        const int qrow = region(wy * WS + qi / WS, a.H, WS, a.shift), qcol = region(wx * WS + qi % WS, a.W, WS, a.shift);
        const int kcol0 = WS == 16 ? lg * 4 : (lg & 1) * 4;
        bool cdiff[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) cdiff[r] = region(wx * WS + kcol0 + r, a.W, WS, a.shift) != qcol;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if constexpr (!LDSK) s[kt] += *reinterpret_cast<const f32x4*>(bias + kt * 16 + lg * 4);
            if (masked) {
                const bool rdiff = region(wy * WS + (WS == 16 ? kt : 2 * kt + (lg >> 1)), a.H, WS, a.shift) != qrow;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rdiff || cdiff[r]) s[kt][r] += -100.0f;
            }
            if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = wave_max_xor(mx, 16);
        mx = wave_max_xor(mx, 32);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[kt][r] = __expf(s[kt][r] - mx);
                sum += s[kt][r];
            }
        sum = wave_sum_xor(sum, 16);
        sum = wave_sum_xor(sum, 32);
        const float inv = 1.0f / sum;
        // delta = sum_d dO[q][d] O[q][d]
        float dl = 0.f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) dl += (float)dof.v[jj] * (float)of.v[jj];
        dl = wave_sum_xor(dl, 16);
        dl = wave_sum_xor(dl, 32);
        if (lg == 0) {
            a.lse[bh * a.Nq + qi] = mx + __logf(sum);
            a.delta[bh * a.Nq + qi] = dl;
        }
        // dS^T = P o (V dO^T - delta)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const Frag<bf16> vf = LDSK ? VL[kt * 64 + lane] : *reinterpret_cast<const Frag<bf16>*>(v + (size_t)(kt * 16 + lr) * 32 + lg * 8);
            const f32x4 dp = mma_z(vf, dof);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = s[kt][r] * inv * (dp[r] - dl);
            dbacc[kt] += s[kt];
            if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        // dQ[q][d] = sum_key dS[q][key] K[key][d]: operand = the dS accumulators (keys 32 ks + 4 lg + r | 32 ks + 16 + 4 lg + r per lane group)
        f32x4 dq[2] = {(f32x4)(0.0f), (f32x4)(0.0f)};
#pragma unroll
        for (int ks = 0; ks < KT / 2; ++ks) {
            const Frag<bf16> pf = pack_p(s[2 * ks], s[2 * ks + 1]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16* kp = kT + (size_t)(dt * 16 + lr) * NK + ks * 32 + lg * 4;
                mma(LDSK ? KTL[(dt * (KT / 2) + ks) * 64 + lane] : load_2x4(kp, kp + 16), pf, dq[dt]);  // C[d = 16 dt + 4 lg + r][query lr]
            }
            if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        bf16* dqp = reinterpret_cast<bf16*>(a.dq) + (bh * a.Nq + qi) * 32 + lg * 4;
        store4(dqp, dq[0]);
        store4(dqp + 16, dq[1]);
    }
    // relative_position_bias_table gradient of this workgroup: dS summed over (windows of the group, its 64 queries) and folded through the
    // position index into one table-sized partial (LDS atomics: ~2 k adds per wave, once per launch) -- no [Nq][Nk] gradient ever reaches HBM
    __shared__ float tab[1536];
    for (int i = threadIdx.x; i < 1536; i += 256) tab[i] = 0.f;
    __syncthreads();
    const int* rp = a.rpi + (size_t)qi * NK;
    if (VAR == 1) {
        // 16 x 16 self-attention windows: table row = (yq - yk + 15) * 31 + (xq - xk + 15) (hat.py:480-492) and a 16-key tile is one window row, so the
        // fold is a sum along the diagonals of each 16 x 16 accumulator tile: rotate every key column so that lane = (xq - xk) mod 16 (one
        // ds_bpermute, no conflicts), split wrapped / unwrapped, reduce over the four lane groups, and add 31 DISTINCT table entries per tile
        // (the index-map form below makes up to 4 lanes of an instruction hit one address: 37 us of a 100-us launch)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            float pos = 0.f, neg = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int xk = 4 * lg + r;
                const float w = __shfl(dbacc[kt][r], (lane & 48) | ((lr + xk) & 15), 64);
                const bool p = lr + xk <= 15;
                pos += p ? w : 0.f;
                neg += p ? 0.f : w;
            }
            pos = wave_sum_xor(pos, 16);
            pos = wave_sum_xor(pos, 32);
            neg = wave_sum_xor(neg, 16);
            neg = wave_sum_xor(neg, 32);
            if (lg == 0) {
                const int row = (qt - kt + 15) * 31 + 15;
                atomicAdd(&tab[row + lr], pos);
                if (lr >= 1) atomicAdd(&tab[row + lr - 16], neg);
            }
        }
    } else if constexpr (LDSK && !SR_OCAQ_ATOMIC_FOLD) {
        oca_table_fold<KT>(dbacc, tab, qt, lane, a.T);
    } else {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            int tt[4];
            if constexpr (LDSK) {  // table row of rel entry j: j - 880, negative entries wrap (hat.py:276-279)
                const int j0 = rel0 + ((16 * kt) / 24) * 39 + off_k[kt % 3] - 880;
#pragma unroll
                for (int r = 0; r < 4; ++r) tt[r] = j0 + r;
            } else {
                const int4 t4 = *reinterpret_cast<const int4*>(rp + kt * 16 + lg * 4);
                tt[0] = t4.x; tt[1] = t4.y; tt[2] = t4.z; tt[3] = t4.w;
            }
            if (VAR == 2) {  // one lane group at a time: no two lanes of an instruction on one address
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (lg == g) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) atomicAdd(&tab[tt[r] < 0 ? tt[r] + a.T : tt[r]], dbacc[kt][r]);
                    }
            } else {
#ifndef SR_Q_NOFOLD
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(&tab[tt[r] < 0 ? tt[r] + a.T : tt[r]], dbacc[kt][r]);
#else
                if (tt[0] == -123456) tab[0] = dbacc[kt][0] + dbacc[kt][1] + dbacc[kt][2] + dbacc[kt][3];
#endif
            }
        }
    }
    __syncthreads();
    float* dst = a.dtab_part + (size_t)blockIdx.x * a.Tpad;
    for (int i = threadIdx.x; i < a.T; i += 256) dst[i] = tab[i];
}

// ---- pass KV.  QT = query tiles (Nq / 16, even); every wave owns KPW consecutive key tiles, so each query-side fragment is loaded once per KPW tiles
template <int QT, int KPW, int WS = 16>
__global__ __launch_bounds__(256, 2) void sr_tr_attn_bwd_kv_kernel(SrTrAttnBwd a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NQ = QT * 16;
    const int kgroups = (a.Nk >> 4) / KPW;
    const int item = blockIdx.x * 4 + wave;
    const int n_items = a.n_bwin * a.heads * kgroups;
    if (item >= n_items) return;
    const int kg = item % kgroups;
    const size_t bh = item / kgroups;
    const int head = (int)(bh % a.heads), bwin = (int)(bh / a.heads);
    const int lr = lane & 15, lg = lane >> 4;

    const bf16* q = reinterpret_cast<const bf16*>(a.q) + bh * NQ * 32;
    const bf16* qT = reinterpret_cast<const bf16*>(a.qT) + bh * NQ * 32;
    const bf16* dOT = reinterpret_cast<const bf16*>(a.dOT) + bh * NQ * 32;
    const float* lse = a.lse + bh * NQ;
    const float* delta = a.delta + bh * NQ;
    const int nwx = a.W / a.ws, nwy = a.H / a.ws;
    const int win = bwin % (nwx * nwy);
    const int wy = win / nwx, wx = win - wy * nwx;
    const bool masked = a.shift > 0 && (wy == nwy - 1 || wx == nwx - 1);

    Frag<bf16> kf[KPW], vf[KPW];
    const float* biasT[KPW];
    int krow[KPW];
    bool cdiff[KPW][4];
    f32x4 dk[KPW][2], dv[KPW][2];
#pragma unroll
    for (int u = 0; u < KPW; ++u) {
        const int ki = (kg * KPW + u) * 16 + lr;
        kf[u] = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.k) + (bh * a.Nk + ki) * 32 + lg * 8);
        vf[u] = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.v) + (bh * a.Nk + ki) * 32 + lg * 8);
        biasT[u] = a.biasT + ((size_t)head * a.Nk + ki) * NQ;
        krow[u] = region(wy * WS + ki / WS, a.H, WS, a.shift);
        const int kcol = region(wx * WS + ki % WS, a.W, WS, a.shift);
#pragma unroll
        for (int r = 0; r < 4; ++r) cdiff[u][r] = region(wx * WS + (WS == 16 ? lg * 4 : (lg & 1) * 4) + r, a.W, WS, a.shift) != kcol;  // query 16 qt + 4 lg + r: its column
        dk[u][0] = dk[u][1] = dv[u][0] = dv[u][1] = (f32x4)(0.0f);
    }
#ifndef SR_KV_UNROLL
#define SR_KV_UNROLL 2
#endif
#pragma unroll SR_KV_UNROLL
    for (int qs = 0; qs < QT / 2; ++qs) {
        f32x4 p[KPW][2], ds[KPW][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int qt = 2 * qs + h;
            const Frag<bf16> qf = *reinterpret_cast<const Frag<bf16>*>(q + (size_t)(qt * 16 + lr) * 32 + lg * 8);
            const size_t orow = ((size_t)bwin * NQ + qt * 16 + lr) * a.ldo + head * 32 + lg * 8;
            const Frag<bf16> dof = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.dO) + orow);
            const int q0 = qt * 16 + lg * 4;
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse + q0), d4 = *reinterpret_cast<const f32x4*>(delta + q0);
            const int qrow = region(wy * WS + (WS == 16 ? qt : 2 * qt + (lg >> 1)), a.H, WS, a.shift);
#pragma unroll
            for (int u = 0; u < KPW; ++u) {
                f32x4 s = mma_z(qf, kf[u]);          // S[query 16 qt + 4 lg + r][key lr]
                const f32x4 dp = mma_z(dof, vf[u]);  // dP, same layout
                s += *reinterpret_cast<const f32x4*>(biasT[u] + q0);
                if (masked) {
                    const bool rdiff = qrow != krow[u];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rdiff || cdiff[u][r]) s[r] += -100.0f;
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
            const size_t off = (size_t)(dt * 16 + lr) * NQ + qs * 32 + lg * 4;
            dotf[dt] = load_2x4(dOT + off, dOT + off + 16);
            qtf[dt] = load_2x4(qT + off, qT + off + 16);
        }
#pragma unroll
        for (int u = 0; u < KPW; ++u) {
            const Frag<bf16> pf = pack_p(p[u][0], p[u][1]), dsf = pack_p(ds[u][0], ds[u][1]);  // row = key lr, k = queries 32 qs + 4 lg + r | + 16
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                mma(dotf[dt], pf, dv[u][dt]);  // C[d = 16 dt + 4 lg + r][key lr]
                mma(qtf[dt], dsf, dk[u][dt]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < KPW; ++u) {
        const int ki = (kg * KPW + u) * 16 + lr;
        bf16* dkp = reinterpret_cast<bf16*>(a.dk) + (bh * a.Nk + ki) * 32 + lg * 4;
        bf16* dvp = reinterpret_cast<bf16*>(a.dv) + (bh * a.Nk + ki) * 32 + lg * 4;
        store4(dkp, dk[u][0]);
        store4(dkp + 16, dk[u][1]);
        store4(dvp, dv[u][0]);
        store4(dvp + 16, dv[u][1]);
    }
}


// ---- pass Q of the overlapping cross attention with the FORWARD's log-sum-exp (SrTrAttnBwd.lse_given, round 5, ABI v11).  The pass above recomputes the softmax: all 36 logit
// tiles of a query tile live at once beside the 36 gradient tiles of the table (288 registers: one wave per SIMD) and K / V / K^T of the whole neighbourhood staged per window
// (114 KB: one workgroup per CU) -- 145 us per launch for 11 GFLOP.  With lse known, P = exp(S - lse) tile by tile: a pair of key tiles is live at a time (the dQ MFMA consumes
// them), the neighbourhood is staged in two halves of 18 key tiles (60 KB: TWO workgroups per CU at two waves per SIMD), dQ accumulates in registers across the halves.
// delta = rowsum(dO o O) is still produced here (for pass KV); lse is read, not written.
constexpr int QO_KT = 36, QO_HALF = 18;
constexpr int QO_LDS = 3 * QO_HALF * 1024 + 6144;
__global__ __launch_bounds__(256, 2) void sr_tr_attn_bwd_q_oca_kernel(SrTrAttnBwd a) {
    constexpr int KT = QO_KT, HALF = QO_HALF, NK = KT * 16, NQ = 256;
    extern __shared__ __attribute__((aligned(16))) char smem_qo[];  // K | V | K^T fragments of HALF key tiles, then the head's table
    const Frag<bf16>* KL = reinterpret_cast<const Frag<bf16>*>(smem_qo);
    const Frag<bf16>* VL = KL + HALF * 64;
    const Frag<bf16>* KTL = VL + HALF * 64;
    float* REL = reinterpret_cast<float*>(smem_qo + 3 * HALF * 1024);
    __shared__ float tab[1536];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wave;
    const int qt = item & 15;
    const int gh = item >> 4;
    const int grp = gh % a.groups, head = gh / a.groups;  // head-major: the 4 waves of a workgroup share (head, group)
    const int lr = lane & 15, lg = lane >> 4;
    const int wpg = (a.n_bwin + a.groups - 1) / a.groups;
    const int qi = qt * 16 + lr;
    // rel index of this lane's logits of tile kt = (ky - qt + 15) * 39 + (kx0 - lr + 15) + r, (ky, kx0) = row / column of key 16 kt + 4 lg (see the pass above)
    const int off_k[3] = {4 * lg, lg < 2 ? 16 + 4 * lg : 39 + 4 * lg - 8, 8 + 4 * lg};
    const int rel0 = (15 - qt) * 39 + 15 - lr;
    {
        const float* bh_ = a.bias + (size_t)head * NQ * NK;
        for (int j = threadIdx.x; j < 39 * 39; j += 256) {
            const int dyp = j / 39, dxp = j - dyp * 39;
            const int qy = dyp >= 15 ? 0 : 15 - dyp, ky = dyp >= 15 ? dyp - 15 : 0, qx = dxp >= 15 ? 0 : 15 - dxp, kx = dxp >= 15 ? dxp - 15 : 0;
            REL[j] = bh_[(size_t)(qy * 16 + qx) * NK + ky * 24 + kx];
        }
        for (int i = threadIdx.x; i < 1536; i += 256) tab[i] = 0.f;
    }
    // The two halves of the neighbourhood are the OUTER loop (the table gradient of 18 key tiles = 72 registers lives across the window walk; all 36 would spill at two
    // workgroups per CU): a window's dQ is completed by the second half on top of the first half's (stored as bf16 in between: dq is a bf16 tensor anyway).
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        f32x4 dbacc[HALF];
#pragma unroll
        for (int kt = 0; kt < HALF; ++kt) dbacc[kt] = (f32x4)(0.0f);
        for (int wi = 0; wi < wpg; ++wi) {
            const int bwin = grp * wpg + wi;
            if (bwin >= a.n_bwin) break;  // (uniform over the workgroup)
            const size_t bh = (size_t)bwin * a.heads + head;
            const bf16* k = reinterpret_cast<const bf16*>(a.k) + bh * NK * 32 + (size_t)hf * HALF * 16 * 32;
            const bf16* v = reinterpret_cast<const bf16*>(a.v) + bh * NK * 32 + (size_t)hf * HALF * 16 * 32;
            const bf16* kT = reinterpret_cast<const bf16*>(a.kT) + bh * NK * 32 + hf * HALF * 16;
            const Frag<bf16> qf = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.q) + (bh * NQ + qi) * 32 + lg * 8);
            const size_t orow = ((size_t)bwin * NQ + qi) * a.ldo + head * 32 + lg * 8;
            const Frag<bf16> dof = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.dO) + orow);
            const Frag<bf16> of = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.o) + orow);
            const float lse_q = a.lse[bh * NQ + qi];
            float dl = 0.f;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) dl += (float)dof.v[jj] * (float)of.v[jj];
            dl = wave_sum_xor(dl, 16);
            dl = wave_sum_xor(dl, 32);
            bf16* dqp = reinterpret_cast<bf16*>(a.dq) + (bh * NQ + qi) * 32 + lg * 4;
            f32x4 dq[2] = {(f32x4)(0.0f), (f32x4)(0.0f)};
            if (hf == 0) {
                if (lg == 0) a.delta[bh * NQ + qi] = dl;
            } else {
                const bf16x4 p0 = *reinterpret_cast<const bf16x4*>(dqp), p1 = *reinterpret_cast<const bf16x4*>(dqp + 16);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    dq[0][rr] = (float)p0[rr];
                    dq[1][rr] = (float)p1[rr];
                }
            }
            constexpr int PIECES = HALF * 64, PER = (PIECES + 255) / 256;  // 16-byte pieces per array (1152), per thread (4.5)
            __syncthreads();  // the previous fragments (and, the first time, the table fill) are through for every wave
            __builtin_amdgcn_sched_barrier(0);
            {
                Frag<bf16> r[PER];
#pragma unroll
                for (int arr = 0; arr < 2; ++arr) {
                    const bf16* src = arr == 0 ? k : v;
#pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        const int c = i * 256 + threadIdx.x;
                        if (c < PIECES) r[i] = *reinterpret_cast<const Frag<bf16>*>(src + (size_t)c * 8);
                    }
#pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        const int c = i * 256 + threadIdx.x, key = c >> 2, g = c & 3;
                        if (c < PIECES) *reinterpret_cast<Frag<bf16>*>(smem_qo + (arr * HALF * 64 + (key >> 4) * 64 + g * 16 + (key & 15)) * 16) = r[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int c = i * 256 + threadIdx.x;
                    const int d = c / (HALF * 2), k0 = (c - d * (HALF * 2)) * 8;  // keys k0 .. k0 + 7 (inside this half) of feature row d
                    if (c < PIECES) r[i] = *reinterpret_cast<const Frag<bf16>*>(kT + (size_t)d * NK + k0);
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int c = i * 256 + threadIdx.x;
                    const int d = c / (HALF * 2), k0 = (c - d * (HALF * 2)) * 8;
                    const int stp = k0 >> 5, e_hi = (k0 >> 4) & 1, g0 = (k0 >> 2) & 3;
                    if (c < PIECES) {
                        char* cell = smem_qo + (2 * HALF * 64 + ((d >> 4) * (HALF / 2) + stp) * 64 + g0 * 16 + (d & 15)) * 16 + e_hi * 8;
                        const bf16x8 v8 = r[i].v;
                        *reinterpret_cast<bf16x4*>(cell) = __builtin_shufflevector(v8, v8, 0, 1, 2, 3);
                        *reinterpret_cast<bf16x4*>(cell + 16 * 16) = __builtin_shufflevector(v8, v8, 4, 5, 6, 7);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < HALF / 2; ++ks) {
                f32x4 dsp[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ktl = 2 * ks + e, kt = hf * HALF + ktl;
                    const float* tr = REL + rel0 + ((16 * kt) / 24) * 39 + off_k[kt % 3];
                    f32x4 b4;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) b4[rr] = tr[rr];
                    const f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(KL[ktl * 64 + lane].v, qf.v, b4, 0, 0, 0);  // S^T[key 16 kt + 4 lg + r][query lr] + bias
                    const f32x4 dp = mma_z(VL[ktl * 64 + lane], dof);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) dsp[e][rr] = __expf(sv[rr] - lse_q) * (dp[rr] - dl);
                    dbacc[ktl] += dsp[e];
                }
                const Frag<bf16> pf = pack_p(dsp[0], dsp[1]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) mma(KTL[(dt * (HALF / 2) + ks) * 64 + lane], pf, dq[dt]);  // C[d = 16 dt + 4 lg + r][query lr]
                if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // (keeps hipcc from hoisting every fragment read of the half at once)
            }
            store4(dqp, dq[0]);
            store4(dqp + 16, dq[1]);
        }
        __syncthreads();  // (hf == 0: also orders the table's zero fill before the first adds)
        if (hf == 0)
            oca_table_fold<HALF, 0>(dbacc, tab, qt, lane, a.T);
        else
            oca_table_fold<HALF, HALF>(dbacc, tab, qt, lane, a.T);
    }
    __syncthreads();
    float* dst = a.dtab_part + (size_t)blockIdx.x * a.Tpad;
    for (int i = threadIdx.x; i < a.T; i += 256) dst[i] = tab[i];
}

// ---- pass KV of the overlapping cross attention (256 queries x 576 keys per (window, head), SrTrAttnBwd.oca_rel) with the QUERY side in LDS (round 5).  The generic pass above is
// one wave per (window, head, two key tiles): every wave pulls the window's q / dO / q^T / dO^T fragments (64 KB) and its bias rows from L2 -- 660 MB per launch, 120 us at the
// ~5.5 TB/s the L2 -> CU path gives the other latency-chain kernels too.  Here a workgroup owns a (window, head): the four query-side arrays are staged once in fragment order
// (64 KB), the head's relative-position table (39 x 39, recovered from the gathered bias as in pass Q) and lse / delta sit beside them (72 KB: two workgroups per CU), and each
// wave walks three groups of three key tiles.  Same products and the same summation order per output as the generic pass.
constexpr int KVO_QT = 16, KVO_KT = 36, KVO_KPW = 3;
constexpr int KVO_LDS = 4 * KVO_QT * 1024 + 6144 + 2 * 256 * 4;
__global__ __launch_bounds__(256, 2) void sr_tr_attn_bwd_kv_oca_kernel(SrTrAttnBwd a) {
    constexpr int QT = KVO_QT, NQ = QT * 16, NK = KVO_KT * 16, KPW = KVO_KPW;
    extern __shared__ __attribute__((aligned(16))) char smem_kv[];
    const Frag<bf16>* QL = reinterpret_cast<const Frag<bf16>*>(smem_kv);  // [q tile][lane]: row 16 qt + lr, features 8 lg ..
    const Frag<bf16>* DOL = QL + QT * 64;                                 // dO likewise
    const Frag<bf16>* QTL = DOL + QT * 64;                                // [d tile][32-query step][lane]: feature row 16 dt + lr, queries 32 qs + 4 lg .. | + 16
    const Frag<bf16>* DOTL = QTL + QT * 64;
    float* REL = reinterpret_cast<float*>(smem_kv + 4 * QT * 1024);       // the head's table [39 * 39]
    float* LSE = REL + 1536;
    float* DEL = LSE + NQ;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const size_t bh = blockIdx.x;
    const int head = (int)(bh % a.heads), bwin = (int)(bh / a.heads);
    const bf16* q = reinterpret_cast<const bf16*>(a.q) + bh * NQ * 32;
    const bf16* qT = reinterpret_cast<const bf16*>(a.qT) + bh * NQ * 32;
    const bf16* dOT = reinterpret_cast<const bf16*>(a.dOT) + bh * NQ * 32;
    // ---- staging (once per workgroup)
    {
        constexpr int PER = QT * 64 / 256;  // 16-byte pieces per array and thread
        Frag<bf16> r[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) r[i] = *reinterpret_cast<const Frag<bf16>*>(q + (size_t)(i * 256 + threadIdx.x) * 8);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = i * 256 + threadIdx.x, tok = c >> 2, g = c & 3;
            *reinterpret_cast<Frag<bf16>*>(smem_kv + ((tok >> 4) * 64 + g * 16 + (tok & 15)) * 16) = r[i];
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = i * 256 + threadIdx.x, tok = c >> 2, g = c & 3;
            r[i] = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.dO) + ((size_t)bwin * NQ + tok) * a.ldo + head * 32 + g * 8);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = i * 256 + threadIdx.x, tok = c >> 2, g = c & 3;
            *reinterpret_cast<Frag<bf16>*>(smem_kv + (QT * 64 + (tok >> 4) * 64 + g * 16 + (tok & 15)) * 16) = r[i];
        }
#pragma unroll
        for (int arr = 0; arr < 2; ++arr) {
            const bf16* srcT = arr == 0 ? qT : dOT;
#pragma unroll
            for (int i = 0; i < PER; ++i) r[i] = *reinterpret_cast<const Frag<bf16>*>(srcT + (size_t)(i * 256 + threadIdx.x) * 8);
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int c = i * 256 + threadIdx.x;
                const int d = c / (NQ / 8), k0 = (c - d * (NQ / 8)) * 8;  // queries k0 .. k0 + 7 of feature row d: two groups of 4
                const int stp = k0 >> 5, e_hi = (k0 >> 4) & 1, g0 = (k0 >> 2) & 3;
                char* cell = smem_kv + ((2 + arr) * QT * 64 + ((d >> 4) * (QT / 2) + stp) * 64 + g0 * 16 + (d & 15)) * 16 + e_hi * 8;
                const bf16x8 v8 = r[i].v;
                *reinterpret_cast<bf16x4*>(cell) = __builtin_shufflevector(v8, v8, 0, 1, 2, 3);
                *reinterpret_cast<bf16x4*>(cell + 16 * 16) = __builtin_shufflevector(v8, v8, 4, 5, 6, 7);
            }
        }
        const float* bh_ = a.bias + (size_t)head * NQ * NK;
        for (int j = threadIdx.x; j < 39 * 39; j += 256) {  // any (q, k) pair with the row / column differences of entry j (as pass Q)
            const int dyp = j / 39, dxp = j - dyp * 39;
            const int qy = dyp >= 15 ? 0 : 15 - dyp, ky = dyp >= 15 ? dyp - 15 : 0, qx = dxp >= 15 ? 0 : 15 - dxp, kx = dxp >= 15 ? dxp - 15 : 0;
            REL[j] = bh_[(size_t)(qy * 16 + qx) * NK + ky * 24 + kx];
        }
        LSE[threadIdx.x] = a.lse[bh * NQ + threadIdx.x];
        DEL[threadIdx.x] = a.delta[bh * NQ + threadIdx.x];
    }
    __syncthreads();

    for (int kg = wave; kg < KVO_KT / KPW; kg += 4) {
        Frag<bf16> kf[KPW], vf[KPW];
        int jbase[KPW];
        f32x4 dk[KPW][2], dv[KPW][2];
#pragma unroll
        for (int u = 0; u < KPW; ++u) {
            const int ki = (kg * KPW + u) * 16 + lr;
            kf[u] = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.k) + (bh * NK + ki) * 32 + lg * 8);
            vf[u] = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.v) + (bh * NK + ki) * 32 + lg * 8);
            const int ky = ki / 24, kx = ki - ky * 24;
            jbase[u] = (ky + 15) * 39 + kx - 4 * lg + 15;  // entry of (query row 0, query column 4 lg): query row qt subtracts 39 qt, register r subtracts r
            dk[u][0] = dk[u][1] = dv[u][0] = dv[u][1] = (f32x4)(0.0f);
        }
#pragma unroll 2
        for (int qs = 0; qs < QT / 2; ++qs) {
            f32x4 p[KPW][2], ds[KPW][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qt = 2 * qs + h;
                const Frag<bf16> qf = QL[qt * 64 + lane], dof = DOL[qt * 64 + lane];
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(LSE + qt * 16 + lg * 4), d4 = *reinterpret_cast<const f32x4*>(DEL + qt * 16 + lg * 4);
#pragma unroll
                for (int u = 0; u < KPW; ++u) {
                    const float* tr = REL + jbase[u] - 39 * qt;
                    f32x4 b4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) b4[r] = tr[-r];
                    const f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf.v, kf[u].v, b4, 0, 0, 0);  // S[query 16 qt + 4 lg + r][key lr] + bias
                    const f32x4 dp = mma_z(dof, vf[u]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p[u][h][r] = __expf(sv[r] - l4[r]);
                        ds[u][h][r] = p[u][h][r] * (dp[r] - d4[r]);
                    }
                }
            }
            Frag<bf16> dotf[2], qtf[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dotf[dt] = DOTL[(dt * (QT / 2) + qs) * 64 + lane];
                qtf[dt] = QTL[(dt * (QT / 2) + qs) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < KPW; ++u) {
                const Frag<bf16> pf = pack_p(p[u][0], p[u][1]), dsf = pack_p(ds[u][0], ds[u][1]);  // row = key lr, k = queries 32 qs + 4 lg + r | + 16
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    mma(dotf[dt], pf, dv[u][dt]);  // C[d = 16 dt + 4 lg + r][key lr]
                    mma(qtf[dt], dsf, dk[u][dt]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < KPW; ++u) {
            const int ki = (kg * KPW + u) * 16 + lr;
            bf16* dkp = reinterpret_cast<bf16*>(a.dk) + (bh * NK + ki) * 32 + lg * 4;
            bf16* dvp = reinterpret_cast<bf16*>(a.dv) + (bh * NK + ki) * 32 + lg * 4;
            store4(dkp, dk[u][0]);
            store4(dkp + 16, dk[u][1]);
            store4(dvp, dv[u][0]);
            store4(dvp + 16, dv[u][1]);
        }
    }
}

// ---- 8 x 8 windows (SwinIR, swinir.py:83-102), ONE pass: a wave owns a whole (window, head) -- 64 queries x 64 keys = sixteen logit tiles in 64 registers -- so nothing
// travels between passes (no lse / delta round trip) and every operand is fetched once, in two rounds issued up front (the two register passes above, written for
// 256 / 576 keys, leave a 64-key window as a chain of ~7 dependent L2 round trips: 35 + 17 us per SwinIR block at 4 x 64 x 64).  Orientation 1 (keys on registers,
// queries on lanes, as pass Q): P, delta, dS -> dQ and the bias-table fold; orientation 2 (queries on registers, keys on lanes, as pass KV; P recomputed from the
// wave's own log-sum-exp): dV, dK.  Workgroup = (head, four consecutive windows): one table partial per workgroup, i.e. groups * 4 == n_bwin.
__global__ __launch_bounds__(256, 2) void sr_tr_attn_bwd_w8_kernel(SrTrAttnBwd a) {
    constexpr int N = 64, WS = 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int per_head = a.n_bwin >> 2;
    const int head = blockIdx.x / per_head;
    const int bwin = (blockIdx.x - head * per_head) * 4 + wave;
    const size_t bh = (size_t)bwin * a.heads + head;
    const bf16* q = reinterpret_cast<const bf16*>(a.q) + bh * N * 32;
    const bf16* k = reinterpret_cast<const bf16*>(a.k) + bh * N * 32;
    const bf16* v = reinterpret_cast<const bf16*>(a.v) + bh * N * 32;
    const bf16* qT = reinterpret_cast<const bf16*>(a.qT) + bh * N * 32;
    const bf16* kT = reinterpret_cast<const bf16*>(a.kT) + bh * N * 32;
    const bf16* dOT = reinterpret_cast<const bf16*>(a.dOT) + bh * N * 32;

    // ---- round 1 of operand loads: the row-major fragments (tile t: tokens 16 t + lr, features 8 lg ..) and the bias tiles of orientation 1
    Frag<bf16> qf[4], kf[4], vf[4], dof[4], of[4];
    f32x4 s[4][4];  // [query tile][key tile]: key 16 kt + 4 lg + r, query 16 qt + lr
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const size_t row = (size_t)(t * 16 + lr) * 32 + lg * 8;
        qf[t] = *reinterpret_cast<const Frag<bf16>*>(q + row);
        kf[t] = *reinterpret_cast<const Frag<bf16>*>(k + row);
        vf[t] = *reinterpret_cast<const Frag<bf16>*>(v + row);
        const size_t orow = ((size_t)bwin * N + t * 16 + lr) * a.ldo + head * 32 + lg * 8;
        dof[t] = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.dO) + orow);
        of[t] = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.o) + orow);
    }
    const float* bias = a.bias + (size_t)head * N * N;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[qt][kt] = *reinterpret_cast<const f32x4*>(bias + (size_t)(qt * 16 + lr) * N + kt * 16 + lg * 4);
    __builtin_amdgcn_sched_barrier(0);
    // ---- round 2, in flight under orientation 1: transposed operands (d row 16 dt + lr, tokens 32 ks + 4 lg .. | + 16) and this lane's table rows
    Frag<bf16> ktf[2][2], qtf[2][2], dotf[2][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const size_t off = (size_t)(dt * 16 + lr) * N + ks * 32 + lg * 4;
            ktf[dt][ks] = load_2x4(kT + off, kT + off + 16);
            qtf[dt][ks] = load_2x4(qT + off, qT + off + 16);
            dotf[dt][ks] = load_2x4(dOT + off, dOT + off + 16);
        }
    __builtin_amdgcn_sched_barrier(0);

    __shared__ float tabw[4 * 256];  // one table per wave (= window)

    const int nwx = a.W / WS, nwy = a.H / WS;
    const int win = bwin % (nwx * nwy);
    const int wy = win / nwx, wx = win - wy * nwx;
    const bool masked = a.shift > 0 && (wy == nwy - 1 || wx == nwx - 1);
    // shift-mask regions (common.py:250-274) of the positions this lane meets: as a query / key ON THE LANE (token 16 t + lr: row 2 t + (lr >> 3), column lr & 7)
    // and ON THE REGISTERS (token 16 t + 4 lg + r: row 2 t + (lg >> 1), column 4 (lg & 1) + r)
    const int col_lane = region(wx * WS + (lr & 7), a.W, WS, a.shift);
    int col_reg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) col_reg[r] = region(wx * WS + 4 * (lg & 1) + r, a.W, WS, a.shift);
    auto row_lane = [&](int t) { return region(wy * WS + 2 * t + (lr >> 3), a.H, WS, a.shift); };
    auto row_reg = [&](int t) { return region(wy * WS + 2 * t + (lg >> 1), a.H, WS, a.shift); };

    // ---- orientation 1: S^T = K Q^T + bias (+ mask), softmax over the keys, dS^T = P (V dO^T - delta)
    float lse[4], dl[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt].v, qf[qt].v, s[qt][kt], 0, 0, 0);
        if (masked) {
            const int qrow = row_lane(qt);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const bool rdiff = row_reg(kt) != qrow;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rdiff || col_reg[r] != col_lane) s[qt][kt][r] += -100.0f;
            }
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qt][kt][r]);
        mx = wave_max_xor(mx, 16);
        mx = wave_max_xor(mx, 32);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[qt][kt][r] = __expf(s[qt][kt][r] - mx);
                sum += s[qt][kt][r];
            }
        sum = wave_sum_xor(sum, 16);
        sum = wave_sum_xor(sum, 32);
        const float inv = 1.0f / sum;
        lse[qt] = mx + __logf(sum);
        float d = 0.f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) d += (float)dof[qt].v[jj] * (float)of[qt].v[jj];
        d = wave_sum_xor(d, 16);
        d = wave_sum_xor(d, 32);
        dl[qt] = d;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f32x4 dp = mma_z(vf[kt], dof[qt]);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[qt][kt][r] = s[qt][kt][r] * inv * (dp[r] - d);
        }
        // dQ[q][d] = sum_key dS[q][key] K[key][d]: the dS accumulators are the operand (keys 32 ks + 4 lg + r | + 16 per lane group), K^T from the transposed copy
        f32x4 dq[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            dq[dt] = mma_z(ktf[dt][0], pack_p(s[qt][0], s[qt][1]));
            mma(ktf[dt][1], pack_p(s[qt][2], s[qt][3]), dq[dt]);  // C[d = 16 dt + 4 lg + r][query lr]
        }
        bf16* dqp = reinterpret_cast<bf16*>(a.dq) + (bh * N + qt * 16 + lr) * 32 + lg * 4;
        store4(dqp, dq[0]);
        store4(dqp + 16, dq[1]);
    }
    // ---- relative_position_bias_table gradient of this window: dtab[(dy + 7) * 15 + dx + 7] = sum of dS over the (query, key) pairs with qy - ky = dy, qx - kx = dx
    // (swinir.py:56-67).  No LDS atomics (64 ds_add_f32 per wave with 2-4 lanes per address: 35 us of a 62-us launch): lane = (key row parity kyl = bit 5, key column
    // half kxh = bit 4, query row parity qyl = bit 3, query column qx = bits 0-2).  For register r the key column is kx = 4 kxh + r: rotating the eight query columns by kx
    // (one ds_bpermute) puts dx = j (no wrap) or j - 8 (wrapped) on lane column j, so the x fold is a per-lane sum over r and over the tiles of one diagonal
    // d = qt - kt (dy = 2 d + qyl - kyl); what is left are sums over lanes: kxh (xor 16) and the two (qyl, kyl) pairs of one dy (xor 40).
    float fp[7], fn[7];  // [d + 3]: dx = j | dx = j - 8
#pragma unroll
    for (int i = 0; i < 7; ++i) fp[i] = fn[i] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int kx = 4 * ((lane >> 4) & 1) + r;
        const int src = (lane & 56) | ((lane + kx) & 7);
        const bool nowrap = (lane & 7) + kx <= 7;
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float w = __shfl(s[qt][kt][r], src, 64);
                fp[qt - kt + 3] += nowrap ? w : 0.f;
                fn[qt - kt + 3] += nowrap ? 0.f : w;
            }
    }
    float up[9], un[9];  // the partner lane's sums (other (qyl, kyl) pair of the same row difference), index d + 4 with zeros at both ends
    up[0] = un[0] = up[8] = un[8] = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        fp[i] = wave_sum_xor(fp[i], 16);
        fn[i] = wave_sum_xor(fn[i], 16);
        up[i + 1] = __shfl_xor(fp[i], 40, 64);
        un[i + 1] = __shfl_xor(fn[i], 40, 64);
    }
    {
        // writers (kxh = 0): lane (qyl, kyl) = (0, 0): even rows dy = 2 d = own + partner (1, 1); lane (1, 0): odd rows dy = 2 d + 1 = own d (dy' = +1) + partner (0, 1)'s
        // d + 1 (dy' = -1), d = -4 .. 3.  Every one of the 225 entries is written exactly once per wave.
        float* tw = tabw + wave * 256;
        const int j = lane & 7;
        const bool even = (lane & 56) == 0, odd = (lane & 56) == 8;
        if (even) {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int row = (2 * (i - 3) + 7) * 15 + 7;
                tw[row + j] = fp[i] + up[i + 1];
                if (j >= 1) tw[row + j - 8] = fn[i] + un[i + 1];
            }
        }
        if (odd) {
#pragma unroll
            for (int i = -1; i < 7; ++i) {  // d = i - 3
                const int row = (2 * (i - 3) + 1 + 7) * 15 + 7;
                const float mp = i >= 0 ? fp[i] : 0.f, mn = i >= 0 ? fn[i] : 0.f;
                tw[row + j] = mp + up[i + 2];
                if (j >= 1) tw[row + j - 8] = mn + un[i + 2];
            }
        }
    }
#ifdef SR_W8_NOO2
    if (a.T > 0) { __syncthreads(); if ((int)threadIdx.x < a.T) a.dtab_part[(size_t)blockIdx.x * a.Tpad + threadIdx.x] = tabw[threadIdx.x]; return; }
#endif

    // ---- orientation 2: S = Q K^T + bias^T (+ mask), P = exp(S - lse), dS = P (dO V^T - delta); dV = P^T dO, dK = dS^T Q
    f32x4 l4[4], d4[4];  // lse / delta of query 16 qt + 4 lg + r: held by lane 4 lg + r of orientation 1
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            l4[qt][r] = __shfl(lse[qt], 4 * lg + r, 64);
            d4[qt][r] = __shfl(dl[qt], 4 * lg + r, 64);
        }
    const float* biasT = a.biasT + (size_t)head * N * N;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        f32x4 bt[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) bt[qt] = *reinterpret_cast<const f32x4*>(biasT + (size_t)(kt * 16 + lr) * N + qt * 16 + lg * 4);
        const int krow = row_lane(kt);
        f32x4 dk[2], dv[2];
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            f32x4 p[2], ds[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qt = 2 * qs + h;
                f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qt].v, kf[kt].v, bt[qt], 0, 0, 0);  // S[query 16 qt + 4 lg + r][key 16 kt + lr]
                const f32x4 dp = mma_z(dof[qt], vf[kt]);
                if (masked) {
                    const bool rdiff = row_reg(qt) != krow;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rdiff || col_reg[r] != col_lane) sv[r] += -100.0f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[h][r] = __expf(sv[r] - l4[qt][r]);
                    ds[h][r] = p[h][r] * (dp[r] - d4[qt][r]);
                }
            }
            const Frag<bf16> pf = pack_p(p[0], p[1]), dsf = pack_p(ds[0], ds[1]);  // row = key lr, k = queries 32 qs + 4 lg + r | + 16
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                if (qs == 0) {
                    dv[dt] = mma_z(dotf[dt][0], pf);  // C[d = 16 dt + 4 lg + r][key lr]
                    dk[dt] = mma_z(qtf[dt][0], dsf);
                } else {
                    mma(dotf[dt][1], pf, dv[dt]);
                    mma(qtf[dt][1], dsf, dk[dt]);
                }
            }
        }
        bf16* dkp = reinterpret_cast<bf16*>(a.dk) + (bh * N + kt * 16 + lr) * 32 + lg * 4;
        bf16* dvp = reinterpret_cast<bf16*>(a.dv) + (bh * N + kt * 16 + lr) * 32 + lg * 4;
        store4(dkp, dk[0]);
        store4(dkp + 16, dk[1]);
        store4(dvp, dv[0]);
        store4(dvp + 16, dv[1]);
    }
    __syncthreads();
    if ((int)threadIdx.x < 225)
        a.dtab_part[(size_t)blockIdx.x * a.Tpad + threadIdx.x] = (tabw[threadIdx.x] + tabw[256 + threadIdx.x]) + (tabw[512 + threadIdx.x] + tabw[768 + threadIdx.x]);
}

// ---- overlapping cross attention (hat.py:239-293), training forward: softmax(q k^T + bias) v with Nk = 16 KT keys per window from the unfolded
//      neighbourhood (sr_tr_oca_unfold); one wave = (window, head, 16 queries), everything in registers as in pass Q
template <int KT>
__global__ __launch_bounds__(256, 1) void sr_tr_attn_fwd_kernel(SrTrAttnFwd a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NK = KT * 16;
    const int qtiles = a.Nq >> 4;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.n_bwin * a.heads * qtiles) return;
    const int qt = item % qtiles;
    const size_t bh = item / qtiles;
    const int head = (int)(bh % a.heads), bwin = (int)(bh / a.heads);
    const int lr = lane & 15, lg = lane >> 4;
    const int qi = qt * 16 + lr;
    const bf16* k = reinterpret_cast<const bf16*>(a.k) + bh * NK * 32;
    const bf16* vT = reinterpret_cast<const bf16*>(a.vT) + bh * NK * 32;
    const Frag<bf16> qf = *reinterpret_cast<const Frag<bf16>*>(reinterpret_cast<const bf16*>(a.q) + (bh * a.Nq + qi) * 32 + lg * 8);
    const float* bias = a.bias + ((size_t)head * a.Nq + qi) * NK;
    f32x4 s[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const Frag<bf16> kf = *reinterpret_cast<const Frag<bf16>*>(k + (size_t)(kt * 16 + lr) * 32 + lg * 8);
        s[kt] = mma_z(kf, qf) + *reinterpret_cast<const f32x4*>(bias + kt * 16 + lg * 4);
        if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = wave_max_xor(mx, 16);
    mx = wave_max_xor(mx, 32);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kt][r] = __expf(s[kt][r] - mx);
            sum += s[kt][r];
        }
    sum = wave_sum_xor(sum, 16);
    sum = wave_sum_xor(sum, 32);
    const float inv = 1.0f / sum;
    f32x4 o[2] = {(f32x4)(0.0f), (f32x4)(0.0f)};
#pragma unroll
    for (int ks = 0; ks < KT / 2; ++ks) {
        const Frag<bf16> pf = pack_p(s[2 * ks] * inv, s[2 * ks + 1] * inv);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const bf16* vp = vT + (size_t)(dt * 16 + lr) * NK + ks * 32 + lg * 4;
            mma(load_2x4(vp, vp + 16), pf, o[dt]);  // C[d = 16 dt + 4 lg + r][query lr]
        }
        if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    bf16* op = reinterpret_cast<bf16*>(a.out) + ((size_t)bwin * a.Nq + qi) * a.ldo + head * 32 + lg * 4;
    store4(op, o[0]);
    store4(op + 16, o[1]);
}

// nn.Unfold of OCAB (hat.py:217-221,255-263) on the per-window q / k / v layout: key j of window (wy, wx) is the pixel at offset
// (j / wse - pad, j % wse - pad) of the window's corner (zeros outside the image).  unfold: kwin / vwin [bwin][head][NK][32] and their
// transposes [bwin][head][32][NK] from k / v [bwin][head][256][32]; fold (the adjoint, a gather-sum): dk / dv <- dkwin / dvwin.
__global__ __launch_bounds__(256) void sr_tr_oca_unfold_kernel(SrTrOcaFold a) {
    const int NK = a.wse * a.wse;
    const long long total = (long long)a.B * a.nwy * a.nwx * a.heads * NK * 4;  // 16-byte pieces of a key row
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int piece = (int)(i & 3);
    long long t = i >> 2;
    const int j = (int)(t % NK);
    t /= NK;
    const int head = (int)(t % a.heads);
    const long long bwin = t / a.heads;
    const int win = (int)(bwin % (a.nwy * a.nwx));
    const long long b = bwin / (a.nwy * a.nwx);
    const int wy = win / a.nwx, wx = win - wy * a.nwx;
    const int y = wy * 16 + j / a.wse - a.pad, x = wx * 16 + j % a.wse - a.pad;
    bf16x8 kv = (bf16x8)(0.0f), vv = (bf16x8)(0.0f);
    if (y >= 0 && y < a.nwy * 16 && x >= 0 && x < a.nwx * 16) {
        const long long src = (((b * a.nwy * a.nwx + (y >> 4) * a.nwx + (x >> 4)) * a.heads + head) * 256 + (y & 15) * 16 + (x & 15)) * 32 + piece * 8;
        kv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.k) + src);
        vv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.v) + src);
    }
    const long long bh = bwin * a.heads + head;
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(a.kwin) + (bh * NK + j) * 32 + piece * 8) = kv;
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(a.vwin) + (bh * NK + j) * 32 + piece * 8) = vv;
    bf16* kT = reinterpret_cast<bf16*>(a.kwinT) + bh * 32 * NK + j;
    bf16* vT = reinterpret_cast<bf16*>(a.vwinT) + bh * 32 * NK + j;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        kT[(long long)(piece * 8 + e) * NK] = kv[e];
        vT[(long long)(piece * 8 + e) * NK] = vv[e];
    }
}

__global__ __launch_bounds__(256) void sr_tr_oca_fold_kernel(SrTrOcaFold a) {
    const int NK = a.wse * a.wse;
    const long long total = (long long)a.B * a.nwy * a.nwx * a.heads * 256 * 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int piece = (int)(i & 3);
    long long t = i >> 2;
    const int tok = (int)(t & 255);
    t >>= 8;
    const int head = (int)(t % a.heads);
    const long long bwin = t / a.heads;
    const int win = (int)(bwin % (a.nwy * a.nwx));
    const long long b = bwin / (a.nwy * a.nwx);
    const int wy = win / a.nwx, wx = win - wy * a.nwx;
    const int y = wy * 16 + (tok >> 4), x = wx * 16 + (tok & 15);
    f32x4 k0 = (f32x4)(0.0f), k1 = (f32x4)(0.0f), v0 = (f32x4)(0.0f), v1 = (f32x4)(0.0f);
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int wy2 = wy + dy, wx2 = wx + dx;
            if (wy2 < 0 || wy2 >= a.nwy || wx2 < 0 || wx2 >= a.nwx) continue;
            const int jy = y - wy2 * 16 + a.pad, jx = x - wx2 * 16 + a.pad;
            if (jy < 0 || jy >= a.wse || jx < 0 || jx >= a.wse) continue;
            const long long src = ((((b * a.nwy + wy2) * a.nwx + wx2) * a.heads + head) * NK + jy * a.wse + jx) * 32 + piece * 8;
            const bf16x8 kv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.kwin) + src);
            const bf16x8 vv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.vwin) + src);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                k0[e] += (float)kv[e];
                k1[e] += (float)kv[4 + e];
                v0[e] += (float)vv[e];
                v1[e] += (float)vv[4 + e];
            }
        }
    const long long dst = ((bwin * a.heads + head) * 256 + tok) * 32 + piece * 8;
    store4(reinterpret_cast<bf16*>(a.k) + dst, k0);
    store4(reinterpret_cast<bf16*>(a.k) + dst + 4, k1);
    store4(reinterpret_cast<bf16*>(a.v) + dst, v0);
    store4(reinterpret_cast<bf16*>(a.v) + dst + 4, v1);
}

}  // namespace


extern "C" int sr_tr_attn_fwd(const SrTrAttnFwd* p, void* stream) {
    SR_REQUIRE(p && p->q && p->k && p->vT && p->bias && p->out, "sr_tr_attn_fwd: null pointer");
    const SrTrAttnFwd& a = *p;
    SR_REQUIRE(a.hd_p == 32 && a.Nq == 256 && a.Nk == 576 && a.heads > 0 && a.n_bwin > 0 && a.ldo >= a.heads * 32 && a.ldo % 4 == 0, "sr_tr_attn_fwd: unsupported geometry (Nq 256, Nk 576)");
    if (sr_tr_attn_fwd_lds_supported(a)) return sr_tr_attn_fwd_lds(a, reinterpret_cast<hipStream_t>(stream));
    const int items = a.n_bwin * a.heads * (a.Nq / 16);
    hipLaunchKernelGGL(sr_tr_attn_fwd_kernel<36>, dim3((items + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    SR_CHECK_LAUNCH("sr_tr_attn_fwd");
    return SR_OK;
}

extern "C" int sr_tr_oca_fold(const SrTrOcaFold* p, int unfold, void* stream) {
    SR_REQUIRE(p && p->k && p->v && p->kwin && p->vwin && (!unfold || (p->kwinT && p->vwinT)), "sr_tr_oca_fold: null pointer");
    const SrTrOcaFold& a = *p;
    SR_REQUIRE(a.B > 0 && a.nwy > 0 && a.nwx > 0 && a.heads > 0 && a.wse > 16 && a.pad * 2 + 16 == a.wse, "sr_tr_oca_fold: bad geometry (16 x 16 windows, wse = 16 + 2 pad)");
    const long long total = (long long)a.B * a.nwy * a.nwx * a.heads * (unfold ? a.wse * a.wse : 256) * 4;
    if (unfold)
        hipLaunchKernelGGL(sr_tr_oca_unfold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(sr_tr_oca_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    SR_CHECK_LAUNCH("sr_tr_oca_fold");
    return SR_OK;
}

extern "C" int sr_tr_attn_bwd(const SrTrAttnBwd* p, void* stream) {
    SR_REQUIRE(p && p->q && p->qT && p->k && p->kT && p->v && p->o && p->dO && p->dOT && p->bias && p->biasT && p->dq && p->dk && p->dv && p->lse && p->delta &&
                   p->dtab_part && p->rpi,
               "sr_tr_attn_bwd: null pointer");
    const SrTrAttnBwd& a = *p;
    const bool w8 = a.Nq == 64;  // 8 x 8 windows (SwinIR, swinir.py:83-102): the two register passes with four key / query tiles
    SR_REQUIRE(a.hd_p == 32 && ((a.Nq == 256 && (a.Nk == 256 || a.Nk == 576)) || (w8 && a.Nk == 64 && !a.oca_rel)) && a.heads > 0 && a.n_bwin > 0 && a.groups > 0 &&
                   a.groups <= a.n_bwin && a.ldo >= a.heads * 32 && a.ldo % 8 == 0,
               "sr_tr_attn_bwd: unsupported geometry (hd_p 32; Nq 256 with Nk 256 / 576, or Nq = Nk = 64)");
    SR_REQUIRE(a.shift == 0 || (a.Nk == a.Nq && a.ws * a.ws == a.Nq && a.H % a.ws == 0 && a.W % a.ws == 0 && a.n_bwin % ((a.H / a.ws) * (a.W / a.ws)) == 0),
               "sr_tr_attn_bwd: the shift mask needs the window geometry (ws * ws == Nq)");
    SR_REQUIRE(a.shift == 0 || (a.ws > 0 && a.H > 0 && a.W > 0 && a.shift < a.ws), "sr_tr_attn_bwd: geometry");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SrTrAttnBwd b = a;
    if (b.shift == 0) {  // the kernels divide by the window geometry even when no mask applies
        b.ws = w8 ? 8 : 16;
        b.H = b.W = b.ws;
    }
    SR_REQUIRE(a.T > 0 && a.T <= 1536 && a.Tpad >= a.T, "sr_tr_attn_bwd: the bias table has at most 1536 rows");
    if (w8 && a.groups * 4 == a.n_bwin && a.toeplitz16 && a.T == 225) {  // standard index, one table partial per (head, four windows): the one-pass kernel
        hipLaunchKernelGGL(sr_tr_attn_bwd_w8_kernel, dim3(a.heads * a.groups), dim3(256), 0, st, b);
        SR_CHECK_LAUNCH("sr_tr_attn_bwd (8 x 8 windows)");
        return SR_OK;
    }
    if (w8) {
        const int items = a.groups * a.heads * 4;
        hipLaunchKernelGGL((sr_tr_attn_bwd_q_kernel<4, 0, false, 8>), dim3((items + 3) / 4), dim3(256), 0, st, b);
        SR_CHECK_LAUNCH("sr_tr_attn_bwd (q, 8 x 8 windows)");
        hipLaunchKernelGGL((sr_tr_attn_bwd_kv_kernel<4, 2, 8>), dim3((a.n_bwin * a.heads * 2 + 3) / 4), dim3(256), 0, st, b);
        SR_CHECK_LAUNCH("sr_tr_attn_bwd (kv, 8 x 8 windows)");
        return SR_OK;
    }
    if (sr_tr_attn_bwd_lds_usable(b)) return sr_tr_attn_bwd_lds(b, st);  // 16 x 16 windows, one table partial per (head, window): everything in LDS, one launch
    const int items_q = a.groups * a.heads * (a.Nq / 16);
    const int var = 1;  // (0: the generic index-map fold)
    if (a.Nk == 256) {
        if (var == 1 && a.toeplitz16)
            hipLaunchKernelGGL((sr_tr_attn_bwd_q_kernel<16, 1>), dim3((items_q + 3) / 4), dim3(256), 0, st, b);
        else if (var == 2)
            hipLaunchKernelGGL((sr_tr_attn_bwd_q_kernel<16, 2>), dim3((items_q + 3) / 4), dim3(256), 0, st, b);
        else
            hipLaunchKernelGGL((sr_tr_attn_bwd_q_kernel<16, 0>), dim3((items_q + 3) / 4), dim3(256), 0, st, b);
    } else {
        static const bool ldsk = !(getenv("SR_TR_OCA_LDS") && atoi(getenv("SR_TR_OCA_LDS")) == 0);  // A/B knob
        if (ldsk && a.oca_rel && a.T == 39 * 39 && a.lse_given && a.shift == 0 && a.Tpad >= 1521) {  // the forward's lse: tile-by-tile softmax, two workgroups per CU
            static SrDeviceOnce once_qo;
            const hipError_t e = sr_once_per_device(once_qo, [&] { return sr_allow_lds(sr_tr_attn_bwd_q_oca_kernel, QO_LDS); });
            SR_REQUIRE(e == hipSuccess, "sr_tr_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            hipLaunchKernelGGL(sr_tr_attn_bwd_q_oca_kernel, dim3((items_q + 3) / 4), dim3(256), QO_LDS, st, b);
        } else if (ldsk && a.oca_rel && a.T == 39 * 39 && a.ws == 16) {
            constexpr int lds = 3 * 36 * 1024 + 6144;
            static SrDeviceOnce once;
            const hipError_t e = sr_once_per_device(once, [&] { return sr_allow_lds(sr_tr_attn_bwd_q_kernel<36, 0, true>, lds); });
            SR_REQUIRE(e == hipSuccess, "sr_tr_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            hipLaunchKernelGGL((sr_tr_attn_bwd_q_kernel<36, 0, true>), dim3((items_q + 3) / 4), dim3(256), lds, st, b);
        } else
            hipLaunchKernelGGL((sr_tr_attn_bwd_q_kernel<36, 0>), dim3((items_q + 3) / 4), dim3(256), 0, st, b);
    }
    SR_CHECK_LAUNCH("sr_tr_attn_bwd (q)");
#ifndef SR_KV_KPW
#define SR_KV_KPW 2
#endif
    static const bool kv_lds = !(getenv("SR_TR_OCA_KV_LDS") && atoi(getenv("SR_TR_OCA_KV_LDS")) == 0);  // A/B knob
    if (kv_lds && a.Nk == 576 && a.Nq == 256 && a.oca_rel && a.T == 39 * 39 && a.shift == 0) {
        static SrDeviceOnce once_kv;
        const hipError_t e = sr_once_per_device(once_kv, [&] { return sr_allow_lds(sr_tr_attn_bwd_kv_oca_kernel, KVO_LDS); });
        SR_REQUIRE(e == hipSuccess, "sr_tr_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(sr_tr_attn_bwd_kv_oca_kernel, dim3(a.n_bwin * a.heads), dim3(256), KVO_LDS, st, b);
        SR_CHECK_LAUNCH("sr_tr_attn_bwd (kv, query side in LDS)");
        return SR_OK;
    }
    hipLaunchKernelGGL((sr_tr_attn_bwd_kv_kernel<16, SR_KV_KPW>), dim3((a.n_bwin * a.heads * (a.Nk / 16 / SR_KV_KPW) + 3) / 4), dim3(256), 0, st, b);
    SR_CHECK_LAUNCH("sr_tr_attn_bwd (kv)");
    return SR_OK;
}
