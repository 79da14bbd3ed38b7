// Fast training path (C ABI v7), parameter side: everything between the reference's parameter tensors and the kernels' operand layouts.
//   sr_tr_gather    packed operand arena  <- flat fp32 parameters through host-built index maps (replaces packing.py per optimizer step)
//   sr_tr_wgrad     weight gradients dW[n][k] = sum_t A[t][n] B[t'(t)][k] of nn.Linear / 3x3 nn.Conv2d (trainer.py:104 loss.backward():
//                   the contraction runs over TOKENS) on the bf16 matrix cores from token-major bf16 operands, split over token slices
//   sr_tr_finalize  flat fp32 gradient     <- the slices' partial sums through host-built index maps (the adjoint of sr_tr_gather)
// The token-major operands ([t][n], what every producer kernel writes with coalesced rows) are the WRONG way round for an MFMA whose
// contraction index must be contiguous per lane: the tiles are staged row-major in LDS and read back TRANSPOSED with
// ds_read_b64_tr_b16 (a 4-row x 16-column block per 16 lanes, delivered column-major).  The contraction order inside a 32-token step
// is permuted identically on both operands (k-group g = tokens 4g..4g+3 and 16+4g..16+4g+3): with a 160-byte row stride the eight
// rows a 32-lane half touches per read fall into eight different 8-bank groups, so every transposed read is conflict-free.
#include "sr_common.h"
#include "sr_host.h"
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

// ----------------------------------------------------------------------------- gather pack
__global__ __launch_bounds__(256) void sr_tr_gather_kernel(const float* __restrict__ P, const int* __restrict__ idx, const int* __restrict__ idx2,
                                                          const float* __restrict__ scl, const unsigned char* __restrict__ mode, void* out, int out_dtype,
                                                          long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int a = idx[i];
    float v = scl[i];
    if (a >= 0) v *= P[a];
    if (idx2) {
        const int b = idx2[i];
        if (b >= 0) v *= P[b];
    }
    const int m = mode ? mode[i] : 0;
    if (m == 1) v = (float)(bf16)v;                       // leading bf16 of a value carried as hi + lo on two constant-one channels
    else if (m == 2) v = v - (float)(bf16)v;              // its remainder
    if (out_dtype == SR_BF16) reinterpret_cast<bf16*>(out)[i] = (bf16)v;
    else reinterpret_cast<float*>(out)[i] = v;
}

// ----------------------------------------------------------------------------- finalize (adjoint of the gather)
__global__ __launch_bounds__(256) void sr_tr_finalize_kernel(const float* __restrict__ arena, const long long* __restrict__ src, const int* __restrict__ stride,
                                                            const int* __restrict__ ns, const float* __restrict__ scale, float* __restrict__ grad, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long s0 = src[i];
    if (s0 < 0) {
        if (s0 == -1) grad[i] = 0.0f;  // -2: this element belongs to another map
        return;
    }
    const int st = stride[i], cnt = ns[i];
    const float* p = arena + s0;
    float acc = 0.0f;
    int s = 0;
    for (; s + 4 <= cnt; s += 4) {  // four partials in flight, added in slice order (deterministic)
        const float v0 = p[(long long)s * st], v1 = p[(long long)(s + 1) * st], v2 = p[(long long)(s + 2) * st], v3 = p[(long long)(s + 3) * st];
        acc += v0;
        acc += v1;
        acc += v2;
        acc += v3;
    }
    for (; s < cnt; ++s) acc += p[(long long)s * st];
    grad[i] = acc * scale[i];
}

// The same sums with the work items in ARENA order: item i reads arena[src[i] + s * stride[i]] and writes grad[dst[i]].  The host sorts the items by src, so
// adjacent lanes read adjacent partials also where the parameter order is not the packed order (3x3 conv weights: the nine taps of a (co, ci) pair are adjacent
// parameters but 144 KB apart in the packed gradient -- in parameter order every lane of such a load sat on its own cache line, 16 slices deep); the one
// scattered access per item is the 4-byte result store.  ns[i] = 0 writes a zero.
__global__ __launch_bounds__(256) void sr_tr_finalize_to_kernel(const float* __restrict__ arena, const long long* __restrict__ src, const int* __restrict__ dst,
                                                               const int* __restrict__ stride, const int* __restrict__ ns, const float* __restrict__ scale,
                                                               float* __restrict__ grad, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int st = stride[i], cnt = ns[i];
    const float* p = arena + src[i];
    float acc = 0.0f;
    int s = 0;
    for (; s + 4 <= cnt; s += 4) {  // four partials in flight, added in slice order (deterministic; the same order as sr_tr_finalize)
        const float v0 = p[(long long)s * st], v1 = p[(long long)(s + 1) * st], v2 = p[(long long)(s + 2) * st], v3 = p[(long long)(s + 3) * st];
        acc += v0;
        acc += v1;
        acc += v2;
        acc += v3;
    }
    for (; s < cnt; ++s) acc += p[(long long)s * st];
    grad[dst[i]] = acc * scale[i];
}

// Items with many slices (LayerNorm gamma / beta and bias-table partials: one per workgroup of the producing launch, 64 - 256 of them): eight adjacent lanes share an
// item -- lane j adds slices j, j + 8, ... (four in flight), then a fixed xor tree -- instead of one thread walking 256 dependent strided loads while the rest of
// the launch has long finished (the finalize launches were 95 us of which the four-slices-in-flight loop over 256 partials is ~64 rounds of L2 latency).
__global__ __launch_bounds__(256) void sr_tr_finalize_to8_kernel(const float* __restrict__ arena, const long long* __restrict__ src, const int* __restrict__ dst,
                                                                const int* __restrict__ stride, const int* __restrict__ ns, const float* __restrict__ scale,
                                                                float* __restrict__ grad, long long n) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long i = gid >> 3;
    const int sub = (int)(gid & 7);
    const bool live = i < n;
    const long long ii = live ? i : n - 1;  // (whole waves take part in the shuffles)
    const int st = stride[ii], cnt = live ? ns[ii] : 0;
    const float* p = arena + src[ii];
    float acc = 0.0f;
    int s = sub;
    for (; s + 24 < cnt; s += 32) {
        const float v0 = p[(long long)s * st], v1 = p[(long long)(s + 8) * st], v2 = p[(long long)(s + 16) * st], v3 = p[(long long)(s + 24) * st];
        acc += v0;
        acc += v1;
        acc += v2;
        acc += v3;
    }
    for (; s < cnt; s += 8) acc += p[(long long)s * st];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (live && sub == 0) grad[dst[i]] = acc * scale[i];
}

// ----------------------------------------------------------------------------- wgrad
constexpr int WG_TN = 64, WG_TK = 64, WG_STEP = 32;
constexpr int WG_LD = 160;                       // bytes per LDS tile row (64 bf16 + 32 B): see the header comment
constexpr int WG_SUB = 2;                        // 32-token sub-steps per barrier interval (generic path): the stash -> barrier -> transposed reads -> MFMA chain of a step
                                                 // was ~650 cycles for 4 MFMAs per wave; two sub-steps share one barrier
constexpr int WG_TILE = WG_SUB * WG_STEP * WG_LD;  // one operand tile (all sub-steps)
constexpr int WG_MAXJOBS = 8;

struct WgradJobs {
    SrTrWgradJob j[WG_MAXJOBS];
    int wg0[WG_MAXJOBS + 1];  // first workgroup of each job
    int tiles_n[WG_MAXJOBS], tiles_k[WG_MAXJOBS];
    int nwg[WG_MAXJOBS];      // workgroups of each job (its range of block ids is padded to a multiple of 8 when xcd is set)
    int n;
    int xcd;                  // XCD-aware order of a job's workgroups (see the kernel)
};

// 8 bytes of a transposed fragment: for 16-lane group g, block rows r0 .. r0+3, columns c0 .. c0+15 of a row-major bf16 tile -> lane i gets column c0 + i
template <int LD = WG_LD>
SR_DEV s16x4 tr_read(const char* tile, int r0, int c0, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    auto* ptr = (__attribute__((address_space(3))) s16x4*)(tile + (r0 + q) * LD + (c0 + 4 * p) * 2);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
}
SR_DEV Frag<bf16> frag2(const s16x4& a, const s16x4& b) {
    Frag<bf16> f;
    f.v = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
    return f;
}

// ---- 3x3 weight gradient on 2-D pixel patches: all nine taps from ONE staged halo tile.
// The generic path above treats a tap as a row shift of B and re-reads both operands per tap (43 % of a HAB's weight-gradient traffic).  Here a
// step is a 4 x 8-pixel patch: A (dy, 32 pixels x 32 output channels) and B's 6 x 10-pixel halo (x, 60 pixels x 64 input channels) are staged
// once and every tap's operand is 4 CONSECUTIVE halo rows of the same LDS tile for each k-group (k-group g = pixels (row g >> 1, x 4 (g & 1) ..) and two
// rows below: 8 consecutive halo rows per 32-lane half, conflict-free at the 160-byte stride).  Workgroup tile 32 n x 64 k x 9 taps, wave tile
// 16 x 32 x 9 (72 accumulator registers); 18 MFMAs per wave and step against 11.6 KB of staged operands (generic path: 4 against 8 KB).
constexpr int HC_PH = 4, PW_ = 8, HC_HW = 10, HC_HROWS = 60;
constexpr int HC_LDA = 96;                        // A tile row stride (32 bf16 + 32 B)
constexpr int HC_TA = 32 * HC_LDA, HC_TB = HC_HROWS * WG_LD;
constexpr int HC_BUF = HC_TA + HC_TB;
static_assert(2 * HC_BUF <= 4 * WG_TILE + 8192, "halo path LDS");

SR_DEV void wgrad_conv_halo(const SrTrWgradJob& j, int rem, int tiles_n, int tiles_k, char* smem) {
    const int tk = rem % tiles_k;
    rem /= tiles_k;
    const int tn = rem % tiles_n;
    const int slice = rem / tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lg = lane >> 4;
    const int npx = j.W / PW_, npy = j.H / HC_PH, npi = npx * npy;
    const int total = (j.T / (j.H * j.W)) * npi;
    const int chunk = (total + j.ks - 1) / j.ks;
    const int p_beg = slice * chunk, p_end = min(total, p_beg + chunk);
    const int n0 = tn * 32, k0 = tk * 64;
    const bf16* A = reinterpret_cast<const bf16*>(j.A);
    const bf16* Bm = reinterpret_cast<const bf16*>(j.B);
    const float* Af = reinterpret_cast<const float*>(j.A);
    const float* Bf = reinterpret_cast<const float*>(j.B);
    auto load8 = [](const float* p) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(p), v = *reinterpret_cast<const f32x4*>(p + 4);
        bf16x8 r;
        r[0] = (bf16)u[0]; r[1] = (bf16)u[1]; r[2] = (bf16)u[2]; r[3] = (bf16)u[3];
        r[4] = (bf16)v[0]; r[5] = (bf16)v[1]; r[6] = (bf16)v[2]; r[7] = (bf16)v[3];
        return r;
    };
    // staging roles: A: thread t < 128 -> (pixel t >> 2, 16-byte piece t & 3); B: pieces q = t, t + 256 < 480 -> (halo pixel q >> 3, piece q & 7)
    const int a_px = tid >> 2, a_pc = tid & 3;
    const bool a_on = tid < 128 && n0 + a_pc * 8 < j.Np;
    const int ones_piece = (j.ones_col >= k0 && j.ones_col < k0 + 64) ? (j.ones_col - k0) >> 3 : -1;
    const int ones_elem = (j.ones_col - k0) & 7;
    bf16x8 ra, rb[2];
    auto fetch = [&](int p) {
        ra = (bf16x8)(0.0f);
        rb[0] = rb[1] = (bf16x8)(0.0f);
        if (p >= p_end) return;
        const int b = p / npi, pp = p - b * npi, py = pp / npx, px = pp - py * npx;
        const int y0 = py * HC_PH, x0 = px * PW_;
        const size_t img = (size_t)b * j.H * j.W;
        if (a_on) {
            const size_t row = img + (size_t)(y0 + (a_px >> 3)) * j.W + x0 + (a_px & 7);
            ra = j.a_f32 ? load8(Af + row * j.lda + n0 + a_pc * 8) : *reinterpret_cast<const bf16x8*>(A + row * j.lda + n0 + a_pc * 8);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = tid + 256 * u;
            if (q >= HC_HROWS * 8) continue;
            const int hp = q >> 3, pc = q & 7;
            const int y = y0 + hp / HC_HW - 1, x = x0 + hp % HC_HW - 1;
            if (y < 0 || y >= j.H || x < 0 || x >= j.W) continue;
            const size_t row = img + (size_t)y * j.W + x;
            if (k0 + pc * 8 < j.Kp) rb[u] = j.b_f32 ? load8(Bf + row * j.ldb + k0 + pc * 8) : *reinterpret_cast<const bf16x8*>(Bm + row * j.ldb + k0 + pc * 8);
            if (pc == ones_piece) rb[u][ones_elem] = (bf16)1.0f;
        }
    };
    auto stash = [&](int buf) {
        char* base = smem + buf * HC_BUF;
        if (tid < 128) *reinterpret_cast<bf16x8*>(base + a_px * HC_LDA + a_pc * 16) = ra;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = tid + 256 * u;
            if (q < HC_HROWS * 8) *reinterpret_cast<bf16x8*>(base + HC_TA + (q >> 3) * WG_LD + (q & 7) * 16) = rb[u];
        }
    };
    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t][0] = acc[t][1] = (f32x4)(0.0f);
    const int wn = w & 1, wk = w >> 1;
    // k-group lg: pixels (row lg >> 1, x 4 (lg & 1) .. + 3) and the same two rows below
    const int gy = lg >> 1, gx = 4 * (lg & 1);
    fetch(p_beg);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int p = p_beg; p < p_end; ++p) {
        const bool more = p + 1 < p_end;
        if (more) fetch(p + 1);
        const char* ta = smem + buf * HC_BUF;
        const char* tb = ta + HC_TA;
        const Frag<bf16> xa = frag2(tr_read<HC_LDA>(ta, gy * PW_ + gx, wn * 16, lane), tr_read<HC_LDA>(ta, (gy + 2) * PW_ + gx, wn * 16, lane));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r0 = (gy + t / 3) * HC_HW + gx + t % 3;  // halo row of pixel (gy, gx) shifted by tap t (offset (t / 3 - 1, t % 3 - 1), halo origin (-1, -1))
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const Frag<bf16> yb = frag2(tr_read(tb, r0, wk * 32 + 16 * i, lane), tr_read(tb, r0 + 2 * HC_HW, wk * 32 + 16 * i, lane));
                mma(xa, yb, acc[t][i]);
            }
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* out = j.out + ((size_t)(slice * 9 + t) * j.Np) * j.Kp;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + wk * 32 + 16 * i + (lane & 15);
            if (k >= j.Kp) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 16 + 4 * lg + r;
                if (n < j.Np) out[(size_t)n * j.Kp + k] = acc[t][i][r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void sr_tr_wgrad_kernel(WgradJobs J) {
    __shared__ __attribute__((aligned(16))) char smem[2 * HC_BUF > 4 * WG_TILE ? 2 * HC_BUF : 4 * WG_TILE];  // generic: [buffer][operand]; halo path: [buffer][A | B halo]
    int jb = 0;
#pragma unroll
    for (int i = 1; i < WG_MAXJOBS; ++i)
        if (i < J.n && (int)blockIdx.x >= J.wg0[i]) jb = i;
    const SrTrWgradJob& j = J.j[jb];
    int rem = blockIdx.x - J.wg0[jb];
    if (J.xcd) {
        // Blocks whose ids agree mod 8 share an XCD (= an L2).  A job's work items are ordered tile-fastest, token slice slowest, and all tiles of a slice read
        // the same operand rows: give each of the eight residue classes a CONTIGUOUS eighth of the items (= its own slices), so that a slice's rows enter one L2
        // instead of all eight.  The job's id range is a multiple of 8 long (launcher); ids beyond its items do nothing.
        const int per = (J.wg0[jb + 1] - J.wg0[jb]) >> 3;
        rem = (rem & 7) * per + (rem >> 3);
        if (rem >= J.nwg[jb]) return;
    }
    if (j.halo) {  // (job-uniform branch)
        wgrad_conv_halo(j, rem, J.tiles_n[jb], J.tiles_k[jb], smem);
        return;
    }
    const int tk = rem % J.tiles_k[jb];
    rem /= J.tiles_k[jb];
    const int tn = rem % J.tiles_n[jb];
    rem /= J.tiles_n[jb];
    const int tap = rem % j.taps;
    const int slice = rem / j.taps;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wn = w >> 1, wk = w & 1;  // wave tile: 32 n x 32 k
    const int lg = lane >> 4;
    const int chunk = ((j.T + j.ks - 1) / j.ks + WG_STEP - 1) / WG_STEP * WG_STEP;
    const int t_beg = slice * chunk, t_end = min(j.T, t_beg + chunk);
    const int n0 = tn * WG_TN, k0 = tk * WG_TK;

    // staging: thread -> (row = tid >> 3, 16-byte column piece = tid & 7) of both tiles
    const int srow = tid >> 3, spc = tid & 7;
    const bf16* A = reinterpret_cast<const bf16*>(j.A);
    const bf16* Bm = reinterpret_cast<const bf16*>(j.B);
    const float* Af = reinterpret_cast<const float*>(j.A);
    const float* Bf = reinterpret_cast<const float*>(j.B);
    auto load8 = [](const float* p) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(p), v = *reinterpret_cast<const f32x4*>(p + 4);
        bf16x8 r;
        r[0] = (bf16)u[0]; r[1] = (bf16)u[1]; r[2] = (bf16)u[2]; r[3] = (bf16)u[3];
        r[4] = (bf16)v[0]; r[5] = (bf16)v[1]; r[6] = (bf16)v[2]; r[7] = (bf16)v[3];
        return r;
    };
    const int dy = j.taps == 9 ? tap / 3 - 1 : 0, dx = j.taps == 9 ? tap % 3 - 1 : 0;
    const int hw = j.H * j.W;
    const bool a_col_ok = n0 + spc * 8 < j.Np, b_col_ok = k0 + spc * 8 < j.Kp;
    const int ones_piece = (j.ones_col >= k0 && j.ones_col < k0 + WG_TK) ? (j.ones_col - k0) >> 3 : -1;
    const int ones_elem = (j.ones_col - k0) & 7;

    auto fetch = [&](int t0, bf16x8& ra, bf16x8& rb) {
        const int t = t0 + srow;
        ra = (bf16x8)(0.0f);
        rb = (bf16x8)(0.0f);
        if (t < t_end) {
            if (a_col_ok) ra = j.a_f32 ? load8(Af + (size_t)t * j.lda + n0 + spc * 8) : *reinterpret_cast<const bf16x8*>(A + (size_t)t * j.lda + n0 + spc * 8);
            int ts = t;
            bool ok = true;
            if (j.taps == 9) {
                const int b = t / hw, p = t - b * hw, y = p / j.W, x = p - y * j.W;
                const int ys = y + dy, xs = x + dx;
                ok = ys >= 0 && ys < j.H && xs >= 0 && xs < j.W;
                ts = b * hw + ys * j.W + xs;
            }
            if (ok) {
                if (b_col_ok) rb = j.b_f32 ? load8(Bf + (size_t)ts * j.ldb + k0 + spc * 8) : *reinterpret_cast<const bf16x8*>(Bm + (size_t)ts * j.ldb + k0 + spc * 8);
                if (spc == ones_piece) rb[ones_elem] = (bf16)1.0f;
            }
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4)(0.0f);

    // the launch is a chain of dependent steps per workgroup (global -> registers -> LDS -> MFMA): operands travel TWO steps ahead in registers; a step is
    // WG_SUB sub-steps of 32 tokens (the accumulation order over the tokens is unchanged)
    constexpr int STEP = WG_SUB * WG_STEP;
    bf16x8 ra1[WG_SUB], rb1[WG_SUB], ra2[WG_SUB], rb2[WG_SUB];
    auto fetch_all = [&](int t0, bf16x8 (&ra)[WG_SUB], bf16x8 (&rb)[WG_SUB]) {
#pragma unroll
        for (int u = 0; u < WG_SUB; ++u) fetch(t0 + u * WG_STEP, ra[u], rb[u]);
    };
    auto stash_all = [&](int buf, const bf16x8 (&ra)[WG_SUB], const bf16x8 (&rb)[WG_SUB]) {
#pragma unroll
        for (int u = 0; u < WG_SUB; ++u) {
            *reinterpret_cast<bf16x8*>(smem + (2 * buf) * WG_TILE + (u * WG_STEP + srow) * WG_LD + spc * 16) = ra[u];
            *reinterpret_cast<bf16x8*>(smem + (2 * buf + 1) * WG_TILE + (u * WG_STEP + srow) * WG_LD + spc * 16) = rb[u];
        }
    };
    fetch_all(t_beg, ra1, rb1);
    stash_all(0, ra1, rb1);
    fetch_all(t_beg + STEP, ra1, rb1);
    __syncthreads();
    int buf = 0;
    for (int t0 = t_beg; t0 < t_end; t0 += STEP) {
        const bool more = t0 + STEP < t_end;
        if (t0 + 2 * STEP < t_end) fetch_all(t0 + 2 * STEP, ra2, rb2);
#pragma unroll
        for (int u = 0; u < WG_SUB; ++u) {
            const char* ta = smem + (2 * buf) * WG_TILE + u * WG_STEP * WG_LD;
            const char* tb = ta + WG_TILE;
            Frag<bf16> xa[2], yb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const s16x4 a0 = tr_read(ta, 4 * lg, wn * 32 + 16 * i, lane), a1 = tr_read(ta, 16 + 4 * lg, wn * 32 + 16 * i, lane);
                const s16x4 b0 = tr_read(tb, 4 * lg, wk * 32 + 16 * i, lane), b1 = tr_read(tb, 16 + 4 * lg, wk * 32 + 16 * i, lane);
                xa[i].v = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                yb[i].v = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) mma(xa[a], yb[b], acc[a][b]);
        }
        if (more) stash_all(buf ^ 1, ra1, rb1);
#pragma unroll
        for (int u = 0; u < WG_SUB; ++u) {
            ra1[u] = ra2[u];
            rb1[u] = rb2[u];
        }
        __syncthreads();
        buf ^= 1;
    }
    // acc[a][b]: n = n0 + 32 wn + 16 a + 4 lg + r, k = k0 + 32 wk + 16 b + (lane & 15)
    float* out = j.out + ((size_t)(slice * j.taps + tap) * j.Np) * j.Kp;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int k = k0 + wk * 32 + 16 * b + (lane & 15);
            if (k >= j.Kp) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 32 + 16 * a + 4 * lg + r;
                if (n < j.Np) out[(size_t)n * j.Kp + k] = acc[a][b][r];
            }
        }
}


#ifndef SR_WGW_NF
#define SR_WGW_NF 6
#endif
#ifndef SR_WGW_KF
#define SR_WGW_KF 3
#endif
#ifndef SR_WGW_STAGES
#define SR_WGW_STAGES 3
#endif
// ---- nn.Linear weight gradients on WIDE tiles (round 5; tools/wgrad_time.py: one SwinIR / HAT block's four jobs over 16 k tokens 37.9 -> 28.6 us at ks = 16).
// The 64 x 64 tiles above load 128 operand columns per token for 4,096 MACs: the four GEMMs of a block (qkv 576 x 192, proj 192 x 192, fc1 384 x 192, fc2 192 x 384)
// pull 302 MB through the L2 -> LDS path per launch, and each 64-token step is a dependent chain global -> registers -> LDS -> barrier -> MFMA whose register
// prefetch is one step deep in effect (the copy ra1 = ra2 at the end of a step waits for the load issued at its start).  Here a workgroup owns 32 NF x 32 KF outputs
// (192 x 96: 2 x 2 waves of 96 x 48; 18 MFMAs per wave and 32-token step from 9 transposed fragments) and the operands travel global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no waiting copy) into a ring of three stages (61 KB: two workgroups per CU), two steps ahead of the MFMAs; one
// barrier per step.  A staged row is the tile's columns + 32 bytes of padding (conflict-free transposed reads), i.e. 26 / 14 sixteen-byte slots: the 32 rows of a
// step are 13 + 7 wave-wide DMA instructions whose lanes on padding slots are switched off.  Every slice must consist of whole steps (T a multiple of 32 ks); the
// bias column (ones_col) is patched into the fragment in registers.  Same token order inside a slice as the 64 x 64 path: the partial sums differ by tile shape only.
// What bounds it (variant builds): the launch without its MFMAs AND without the fragment reads 13.6 us (of which ~5 us are the 18.9 MB of fp32 partials, ~3 us the
// launch), + DMA 5 us, + fragment reads and MFMAs 15 us: 36 ds_read_b64_tr_b16 per wave and step = 74 KB per CU and step against 18 MFMAs per SIMD -- the LDS, not
// the matrix cores.  Deeper rings (4 / 7 stages) and a software pipeline of the fragment reads: +-0 / worse; two workgroups per CU (ks = 32: 512 workgroups) 21.8 us
// but twice the partials for sr_tr_finalize_to.
template <int NF, int KF>
__global__ __launch_bounds__(256, 2) void sr_tr_wgrad_wide_kernel(WgradJobs J) {
    constexpr int TN = 32 * NF, TK = 32 * KF;
    constexpr int SA = TN / 8 + 2, SB = TK / 8 + 2;      // 16-byte slots per staged row (data + padding)
    constexpr int LDA = SA * 16, LDB = SB * 16;
    static_assert((LDA / 32) % 2 == 1 && (LDB / 32) % 2 == 1, "conflict-free transposed reads need a row stride that is an odd multiple of 32 bytes");
    constexpr int TILE_A = WG_STEP * LDA, TILE_B = WG_STEP * LDB, STAGE = TILE_A + TILE_B;
    static_assert(TILE_A % 1024 == 0 && TILE_B % 1024 == 0, "whole wave-wide DMA instructions per tile");
    constexpr int NIA = TILE_A / 1024, NI = STAGE / 1024;  // DMA instructions per step: [0, NIA) operand A, [NIA, NI) operand B
    static_assert(NI % 4 == 0, "the same number of DMA instructions per wave (vmcnt bookkeeping)");
    constexpr int U = NI / 4, STAGES = SR_WGW_STAGES, D = STAGES - 1;  // the DMA runs D steps ahead
    static_assert(D * U <= 60, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    int jb = 0;
#pragma unroll
    for (int i = 1; i < WG_MAXJOBS; ++i)
        if (i < J.n && (int)blockIdx.x >= J.wg0[i]) jb = i;
    const SrTrWgradJob& j = J.j[jb];
    int rem = blockIdx.x - J.wg0[jb];
    {   // XCD-aware order, as sr_tr_wgrad_kernel: each of the eight residue classes of the block id takes a contiguous eighth of the items (= its own token slices)
        const int per = (J.wg0[jb + 1] - J.wg0[jb]) >> 3;
        rem = (rem & 7) * per + (rem >> 3);
        if (rem >= J.nwg[jb]) return;
    }
    const int tk = rem % J.tiles_k[jb];
    rem /= J.tiles_k[jb];
    const int tn = rem % J.tiles_n[jb];
    const int slice = rem / J.tiles_n[jb];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w >> 1, wk = w & 1;
    const int lg = lane >> 4;
    const int chunk = j.T / j.ks;  // (a multiple of WG_STEP: launcher)
    const int t_beg = slice * chunk, nsteps = chunk / WG_STEP;
    const int n0 = tn * TN, k0 = tk * TK;
    // this wave's DMA instructions g = w + 4 u: per lane the byte offset of its 16-byte piece relative to the step's first token row (or -1: a padding slot)
    int voff[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int g = w + 4 * u;
        if (g < NIA) {
            const int slot = g * 64 + lane, row = slot / SA, c = slot - row * SA;
            voff[u] = c < TN / 8 ? (row * j.lda + n0 + c * 8) * 2 : -1;
        } else {
            const int slot = (g - NIA) * 64 + lane, row = slot / SB, c = slot - row * SB;
            voff[u] = c < TK / 8 ? (row * j.ldb + k0 + c * 8) * 2 : -1;
        }
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_w;
    const bf16* A = reinterpret_cast<const bf16*>(j.A) + (size_t)t_beg * j.lda;
    const bf16* Bm = reinterpret_cast<const bf16*>(j.B) + (size_t)t_beg * j.ldb;
    auto issue = [&](int step) {  // the WG_STEP token rows of `step` -> ring stage step % STAGES
        const unsigned dst = lds0 + (unsigned)(step % STAGES) * STAGE;
        const bf16* ar = A + (size_t)step * WG_STEP * j.lda;
        const bf16* br = Bm + (size_t)step * WG_STEP * j.ldb;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = w + 4 * u;
            const bf16* base = g < NIA ? ar : br;  // (wave-uniform)
            const unsigned m0v = __builtin_amdgcn_readfirstlane(dst + g * 1024);
            unsigned keep;
            // lanes on padding slots keep their address register harmless and are switched off through the exec mask of the branch
            if (voff[u] >= 0)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff[u]), "s"(base), "s"(m0v) : "memory");
        }
    };
    f32x4 acc[NF][KF];
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int b = 0; b < KF; ++b) acc[a][b] = (f32x4)(0.0f);
    // bias column: B[:, ones_col] reads as one (dW[:, ones_col] = column sums of A): the lane that holds that column sets its fragment to ones
    const int ones_rel = j.ones_col - k0 - wk * 16 * KF;  // relative to this wave's columns
    const int ones_b = (j.ones_col >= 0 && ones_rel >= 0 && ones_rel < 16 * KF) ? ones_rel >> 4 : -1;
    const bool ones_lane = (lane & 15) == (ones_rel & 15);

    auto wait_younger = [&](int c) {  // wait until at most c steps' DMA instructions of this wave are in flight (s_waitcnt takes an immediate)
        switch (c > 0 ? c : 0) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * U) : "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * U <= 63 ? 2 * U : 63) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * U <= 63 ? 3 * U : 63) : "memory"); break;
        }
    };
    static_assert(D <= 4, "wait_younger covers up to three younger steps");
#pragma unroll
    for (int i = 0; i < D; ++i)
        if (i < nsteps) issue(i);
    for (int sidx = 0; sidx < nsteps; ++sidx) {
        wait_younger(min(sidx + D - 1, nsteps - 1) - sidx);  // step sidx has landed for this wave (DMA completes in issue order) ... and, behind the barrier, for all four
        __syncthreads();                                      // also: every wave is through step sidx - 1, whose stage the next issue overwrites
        if (sidx + D < nsteps) issue(sidx + D);
        const char* ta = smem_w + (sidx % STAGES) * STAGE;
        const char* tb = ta + TILE_A;
        Frag<bf16> xa[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i)
            xa[i] = frag2(tr_read<LDA>(ta, 4 * lg, wn * 16 * NF + 16 * i, lane), tr_read<LDA>(ta, 16 + 4 * lg, wn * 16 * NF + 16 * i, lane));
#pragma unroll
        for (int b = 0; b < KF; ++b) {
            Frag<bf16> yb = frag2(tr_read<LDB>(tb, 4 * lg, wk * 16 * KF + 16 * b, lane), tr_read<LDB>(tb, 16 + 4 * lg, wk * 16 * KF + 16 * b, lane));
            if (b == ones_b && ones_lane) {
#pragma unroll
                for (int e = 0; e < 8; ++e) yb.v[e] = (bf16)1.0f;
            }
#pragma unroll
            for (int a = 0; a < NF; ++a) mma(xa[a], yb, acc[a][b]);
        }
    }
    // acc[a][b]: n = n0 + 16 NF wn + 16 a + 4 lg + r, k = k0 + 16 KF wk + 16 b + (lane & 15)
    float* out = j.out + (size_t)slice * j.Np * j.Kp;
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int b = 0; b < KF; ++b) {
            const int k = k0 + wk * 16 * KF + 16 * b + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(n0 + wn * 16 * NF + 16 * a + 4 * lg + r) * j.Kp + k] = acc[a][b][r];
        }
}

// nn.PixelShuffle(r) backward on NHWC bf16 for a conv whose packed output rows are n = (i r + j) cps + c (common.py:124-137): the gradient of
// the conv's packed output [B,H,W,r*r*cps] from the gradient of the shuffled tensor [B,H*r,W*r,cps]: 16-byte pieces
__global__ __launch_bounds__(256) void sr_tr_unshuffle_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int B, int H, int W, int cps, int r) {
    const int pieces = cps >> 3;
    const long long total = (long long)B * H * W * r * r * pieces;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int pc = (int)(i % pieces);
    long long t = i / pieces;
    const int ij = (int)(t % (r * r));
    t /= r * r;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const long long b = t / H;
    const int ii = ij / r, jj = ij - ii * r;
    const long long s = (((b * H * r + y * r + ii) * (long long)W * r) + x * r + jj) * cps + pc * 8;
    const long long d = (((b * H + y) * (long long)W + x) * r * r + ij) * cps + pc * 8;
    *reinterpret_cast<bf16x8*>(dst + d) = *reinterpret_cast<const bf16x8*>(src + s);
}
// LeakyReLU backward from the OUTPUT (same sign as the input): dx = dy * (y > 0 ? 1 : slope)
__global__ __launch_bounds__(256) void sr_tr_lrelu_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ y, bf16* __restrict__ dx, float slope, long long n8) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const bf16x8 g = reinterpret_cast<const bf16x8*>(dy)[i], v = reinterpret_cast<const bf16x8*>(y)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)g[e] * ((float)v[e] > 0.f ? 1.0f : slope));
    reinterpret_cast<bf16x8*>(dx)[i] = o;
}
// torch.optim.Adam's update (trainer.py:133-139: lr, betas, weight_decay as L2 term; no amsgrad) on the flat parameter / gradient / moment buffers
__global__ __launch_bounds__(256) void sr_tr_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n4, float lr,
                                                        float b1, float b2, float eps, float wd, float bc1, float bc2_rsqrt) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i], gv = reinterpret_cast<const f32x4*>(g)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
    const float step = lr / bc1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gg = gv[e] + wd * pv[e];
        mv[e] = mv[e] + (gg - mv[e]) * (1.0f - b1);          // lerp, as torch's _foreach_lerp_
        vv[e] = vv[e] * b2 + gg * gg * (1.0f - b2);
        const float denom = sqrtf(vv[e]) * bc2_rsqrt + eps;
        pv[e] -= step * (mv[e] / denom);
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
}
// out = a + b (fp32; b fp32 or bf16): the residual joins of the backward pass
__global__ __launch_bounds__(256) void sr_tr_add_kernel(const float* __restrict__ a, const void* __restrict__ b, int b_bf16, float* __restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(a)[i];
    if (b_bf16) {
        const bf16x4 w = reinterpret_cast<const bf16x4*>(b)[i];
        v[0] += (float)w[0]; v[1] += (float)w[1]; v[2] += (float)w[2]; v[3] += (float)w[3];
    } else {
        v += reinterpret_cast<const f32x4*>(b)[i];
    }
    reinterpret_cast<f32x4*>(out)[i] = v;
}

}  // namespace

extern "C" int sr_tr_gather(const float* P, const int* idx, const int* idx2, const float* scl, const unsigned char* mode, void* out, int out_dtype, long long n,
                            void* stream) {
    SR_REQUIRE(P && idx && scl && out && n >= 0 && (out_dtype == SR_BF16 || out_dtype == SR_F32), "sr_tr_gather: bad arguments");
    if (n == 0) return SR_OK;
    hipLaunchKernelGGL(sr_tr_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P, idx, idx2, scl, mode, out, out_dtype, n);
    SR_CHECK_LAUNCH("sr_tr_gather");
    return SR_OK;
}

extern "C" int sr_tr_finalize(const float* arena, const long long* src, const int* stride, const int* ns, const float* scale, float* grad, long long n, void* stream) {
    SR_REQUIRE(arena && src && stride && ns && scale && grad && n >= 0, "sr_tr_finalize: bad arguments");
    if (n == 0) return SR_OK;
    hipLaunchKernelGGL(sr_tr_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), arena, src, stride, ns, scale, grad, n);
    SR_CHECK_LAUNCH("sr_tr_finalize");
    return SR_OK;
}

extern "C" int sr_tr_finalize_to(const float* arena, const long long* src, const int* dst, const int* stride, const int* ns, const float* scale, float* grad, long long n,
                                 void* stream) {
    SR_REQUIRE(arena && src && dst && stride && ns && scale && grad && n >= 0, "sr_tr_finalize_to: bad arguments");
    if (n == 0) return SR_OK;
    hipLaunchKernelGGL(sr_tr_finalize_to_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), arena, src, dst, stride, ns, scale, grad, n);
    SR_CHECK_LAUNCH("sr_tr_finalize_to");
    return SR_OK;
}

extern "C" int sr_tr_finalize_to8(const float* arena, const long long* src, const int* dst, const int* stride, const int* ns, const float* scale, float* grad, long long n,
                                  void* stream) {
    SR_REQUIRE(arena && src && dst && stride && ns && scale && grad && n >= 0, "sr_tr_finalize_to8: bad arguments");
    if (n == 0) return SR_OK;
    hipLaunchKernelGGL(sr_tr_finalize_to8_kernel, dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), arena, src, dst, stride, ns, scale, grad, n);
    SR_CHECK_LAUNCH("sr_tr_finalize_to8");
    return SR_OK;
}

extern "C" long long sr_tr_wgrad_out_floats(const SrTrWgradJob* j) { return j ? (long long)j->ks * j->taps * j->Np * j->Kp : 0; }

static bool wgrad_wide_job(const SrTrWgradJob& j) {
    return j.taps == 1 && !j.a_f32 && !j.b_f32 && j.Np % (32 * SR_WGW_NF) == 0 && j.Kp % (32 * SR_WGW_KF) == 0 && j.T % (WG_STEP * j.ks) == 0 &&
           (long long)WG_STEP * (j.lda > j.ldb ? j.lda : j.ldb) * 2 < (1ll << 30);
}

// one launch over a set of jobs: wide = the nn.Linear jobs on sr_tr_wgrad_wide_kernel, otherwise the 64 x 64 / halo kernel
static int wgrad_launch(const SrTrWgradJob* const* jobs, int njobs, bool wide, hipStream_t st) {
    WgradJobs J;
    J.n = njobs;
    J.xcd = 1;  // XCD-aware work order (0 = block ids in item order: 65.2 -> 73.1 us per HAB launch, round 4)
    int wg = 0;
    for (int i = 0; i < njobs; ++i) {
        const SrTrWgradJob& j = *jobs[i];
        J.j[i] = j;
        J.wg0[i] = wg;
        if (wide) {
            J.tiles_n[i] = j.Np / (32 * SR_WGW_NF);
            J.tiles_k[i] = j.Kp / (32 * SR_WGW_KF);
            J.nwg[i] = J.tiles_n[i] * J.tiles_k[i] * j.ks;
        } else if (j.halo) {
            J.tiles_n[i] = (j.Np + 31) / 32;
            J.tiles_k[i] = (j.Kp + 63) / 64;
            J.nwg[i] = J.tiles_n[i] * J.tiles_k[i] * j.ks;
        } else {
            J.tiles_n[i] = (j.Np + WG_TN - 1) / WG_TN;
            J.tiles_k[i] = (j.Kp + WG_TK - 1) / WG_TK;
            J.nwg[i] = J.tiles_n[i] * J.tiles_k[i] * j.taps * j.ks;
        }
        wg += (J.nwg[i] + 7) & ~7;
    }
    J.wg0[njobs] = wg;
    for (int i = njobs; i < WG_MAXJOBS; ++i) {
        J.j[i] = J.j[0];
        J.tiles_n[i] = J.tiles_k[i] = 1;
        J.nwg[i] = 0;
        J.wg0[i + 1 <= WG_MAXJOBS ? i + 1 : i] = wg;
    }
    if (wide) {
        constexpr int lds = SR_WGW_STAGES * WG_STEP * ((32 * SR_WGW_NF) * 2 + 32 + (32 * SR_WGW_KF) * 2 + 32);
        static SrDeviceOnce once;
        const hipError_t e = sr_once_per_device(once, [&] { return sr_allow_lds(sr_tr_wgrad_wide_kernel<SR_WGW_NF, SR_WGW_KF>, lds); });
        SR_REQUIRE(e == hipSuccess, "sr_tr_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL((sr_tr_wgrad_wide_kernel<SR_WGW_NF, SR_WGW_KF>), dim3(wg), dim3(256), lds, st, J);
    } else {
        hipLaunchKernelGGL(sr_tr_wgrad_kernel, dim3(wg), dim3(256), 0, st, J);
    }
    SR_CHECK_LAUNCH("sr_tr_wgrad");
    return SR_OK;
}

extern "C" int sr_tr_wgrad(const SrTrWgradJob* jobs, int njobs, void* stream) {
    SR_REQUIRE(jobs && njobs > 0 && njobs <= WG_MAXJOBS, "sr_tr_wgrad: 1..%d jobs per launch", WG_MAXJOBS);
    static const bool wide_on = !(getenv("SR_WG_WIDE") && atoi(getenv("SR_WG_WIDE")) == 0);  // A/B switch: nn.Linear jobs on the wide-tile kernel
    const SrTrWgradJob* wide[WG_MAXJOBS];
    const SrTrWgradJob* rest[WG_MAXJOBS];
    int nw = 0, nr = 0;
    for (int i = 0; i < njobs; ++i) {
        const SrTrWgradJob& j = jobs[i];
        SR_REQUIRE(j.A && j.B && j.out && j.T > 0 && j.Np > 0 && j.Kp > 0 && j.Np % 8 == 0 && j.Kp % 8 == 0 && j.lda % 8 == 0 && j.ldb % 8 == 0 && j.lda >= j.Np &&
                       j.ldb >= j.Kp && j.ks > 0 && (j.taps == 1 || j.taps == 9) && j.ones_col < j.Kp,
                   "sr_tr_wgrad: bad job %d", i);
        SR_REQUIRE(j.taps == 1 || (j.H > 0 && j.W > 0 && j.T % (j.H * j.W) == 0), "sr_tr_wgrad: a 3x3 job needs H, W with T = B*H*W");
        SR_REQUIRE((((uintptr_t)j.A | (uintptr_t)j.B) & 15) == 0, "sr_tr_wgrad: operands must be 16-byte aligned");
        SR_REQUIRE((j.a_f32 == 0 || j.a_f32 == 1) && (j.b_f32 == 0 || j.b_f32 == 1), "sr_tr_wgrad: a_f32 / b_f32 are flags");
        SR_REQUIRE(!j.halo || (j.taps == 9 && j.H % 4 == 0 && j.W % 8 == 0), "sr_tr_wgrad: the halo form needs 9 taps, H %% 4 == 0, W %% 8 == 0");
        if (wide_on && wgrad_wide_job(j))
            wide[nw++] = &j;
        else
            rest[nr++] = &j;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (nw) {
        const int rc = wgrad_launch(wide, nw, true, st);
        if (rc != SR_OK) return rc;
    }
    if (nr) return wgrad_launch(rest, nr, false, st);
    return SR_OK;
}

extern "C" int sr_tr_unshuffle(const void* src, void* dst, int B, int H, int W, int cps, int r, void* stream) {
    SR_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && cps > 0 && cps % 8 == 0 && r > 0, "sr_tr_unshuffle: bad arguments");
    const long long total = (long long)B * H * W * r * r * (cps / 8);
    hipLaunchKernelGGL(sr_tr_unshuffle_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16*>(src),
                       reinterpret_cast<bf16*>(dst), B, H, W, cps, r);
    SR_CHECK_LAUNCH("sr_tr_unshuffle");
    return SR_OK;
}
extern "C" int sr_tr_lrelu_bwd(const void* dy, const void* y, void* dx, float slope, long long n, void* stream) {
    SR_REQUIRE(dy && y && dx && n > 0 && n % 8 == 0, "sr_tr_lrelu_bwd: bad arguments");
    hipLaunchKernelGGL(sr_tr_lrelu_bwd_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16*>(dy),
                       reinterpret_cast<const bf16*>(y), reinterpret_cast<bf16*>(dx), slope, n / 8);
    SR_CHECK_LAUNCH("sr_tr_lrelu_bwd");
    return SR_OK;
}
extern "C" int sr_tr_add(const float* a, const void* b, int b_dtype, float* out, long long n, void* stream) {
    SR_REQUIRE(a && b && out && n > 0 && n % 4 == 0, "sr_tr_add: bad arguments");
    hipLaunchKernelGGL(sr_tr_add_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, b, b_dtype == SR_BF16 ? 1 : 0, out, n / 4);
    SR_CHECK_LAUNCH("sr_tr_add");
    return SR_OK;
}

extern "C" int sr_tr_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps, float weight_decay, long long step, void* stream) {
    SR_REQUIRE(p && g && m && v && n > 0 && n % 4 == 0 && step > 0, "sr_tr_adam: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(sr_tr_adam_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, n / 4, lr, beta1, beta2, eps, weight_decay,
                       (float)bc1, (float)(1.0 / sqrt(bc2)));
    SR_CHECK_LAUNCH("sr_tr_adam");
    return SR_OK;
}
