// Shared device helpers for the studiosr_amd HIP kernels (gfx950 / CDNA4 only).
//
// Conventions used by every MFMA kernel in this directory
// -------------------------------------------------------
// * One "K-group" = 8 consecutive elements of the contraction axis held by one lane.
//   A K-chunk = 4 K-groups = 32 elements = one v_mfma_f32_16x16x32_bf16 (bf16 mode)
//   or eight v_mfma_f32_16x16x4_f32 (exact-fp32 mode, element j of both operands
//   feeds instruction j, so the two modes share every layout and only differ in
//   the element size).
// * Operand fragment of a 16-row (or 16-col) tile: lane l holds K-group (l >> 4) of
//   row/col (l & 15).  mma(X, Y, C) computes C[i][j] += sum_k X[i][k] * Y[j][k] with
//   C laid out as col j = l & 15, row i = 4 * (l >> 4) + reg.
// * Weights are pre-packed on the host in fragment order [n_tile][k_chunk][lane][8]
//   so that a wave fetches one operand fragment with ONE fully coalesced 1 KiB
//   (bf16) load straight into registers -- weights never pass through LDS.
// * Activation tiles live in LDS "K-group major": image[kgroup][row] of 8-element
//   cells (16 B bf16 / 32 B fp32).  Reading the fragment of 16 CONSECUTIVE rows is
//   then bank-conflict free for ds_read_b128 for any starting row, which is what
//   lets the 3x3 convolution read its nine shifted views from one halo tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// "bf16x3": a value carried as two bf16 numbers hi + lo (lo = bf16(x - hi)): 16 mantissa bits.  A product of two such operands is
// evaluated as hi*hi + hi*lo + lo*hi on the bf16 matrix cores with fp32 accumulation (the lo*lo term is below 2^-16 relative):
// fp32-class accuracy at 3 bf16 MFMAs per K-chunk instead of the 8 quarter... 1/16-rate fp32 MFMAs of the exact path.
struct bf3 {
    bf16 hi, lo;
};

#define SR_DEV __device__ __forceinline__

// In-kernel cycle stamps (diagnostic builds only: `make STAMPS=1`; see tools/stamp_test*.py).  One lane of one
// workgroup writes s_memtime into a __device__ array that only the sr_debug_*_stamps() readers touch.
#ifdef SR_STAMPS
#define SR_STAMP(arr, i) do { if (blockIdx.x == 7 && blockIdx.y == 0 && threadIdx.x == 0) arr[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SR_STAMP(arr, i) do { } while (0)
#endif

// ----------------------------------------------------------------------------- fragments
template <typename T>
struct Frag;
template <>
struct __attribute__((aligned(16))) Frag<bf16> {
    bf16x8 v;
};
template <>
struct __attribute__((aligned(16))) Frag<float> {
    f32x4 lo, hi;
};

template <>
struct __attribute__((aligned(16))) Frag<bf3> {
    bf16x8 hi, lo;
};

SR_DEV void frag_zero(Frag<bf16>& f) { f.v = (bf16x8)(0.0f); }
SR_DEV void frag_zero(Frag<bf3>& f) {
    f.hi = (bf16x8)(0.0f);
    f.lo = (bf16x8)(0.0f);
}
SR_DEV Frag<bf3> frag_keep_if(bool keep, const Frag<bf3>& f) {
    Frag<bf3> r;
    r.hi = keep ? f.hi : (bf16x8)(0.0f);
    r.lo = keep ? f.lo : (bf16x8)(0.0f);
    return r;
}
SR_DEV void frag_zero(Frag<float>& f) {
    f.lo = (f32x4)(0.0f);
    f.hi = (f32x4)(0.0f);
}
SR_DEV Frag<bf16> frag_keep_if(bool keep, const Frag<bf16>& f) {  // keep ? f : 0 as a vector select (no aggregate select: stays in registers)
    Frag<bf16> r;
    r.v = keep ? f.v : (bf16x8)(0.0f);
    return r;
}
SR_DEV Frag<float> frag_keep_if(bool keep, const Frag<float>& f) {
    Frag<float> r;
    r.lo = keep ? f.lo : (f32x4)(0.0f);
    r.hi = keep ? f.hi : (f32x4)(0.0f);
    return r;
}
SR_DEV void frag_set(Frag<bf16>& f, int j, float x) { f.v[j] = (bf16)x; }
SR_DEV void frag_set(Frag<float>& f, int j, float x) {
    if (j < 4)
        f.lo[j] = x;
    else
        f.hi[j - 4] = x;
}
SR_DEV float frag_get(const Frag<bf16>& f, int j) { return (float)f.v[j]; }
SR_DEV float frag_get(const Frag<float>& f, int j) { return j < 4 ? f.lo[j] : f.hi[j - 4]; }

SR_DEV Frag<bf16> frag_from8(const float* x) {
    Frag<bf16> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (bf16)x[j];
    return f;
}
template <typename T>
SR_DEV Frag<T> frag_make(const float* x);
template <>
SR_DEV Frag<bf16> frag_make<bf16>(const float* x) {
    return frag_from8(x);
}
template <>
SR_DEV Frag<float> frag_make<float>(const float* x) {
    Frag<float> f;
    f.lo = f32x4{x[0], x[1], x[2], x[3]};
    f.hi = f32x4{x[4], x[5], x[6], x[7]};
    return f;
}

template <>
SR_DEV Frag<bf3> frag_make<bf3>(const float* x) {
    Frag<bf3> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bf16 h = (bf16)x[j];
        f.hi[j] = h;
        f.lo[j] = (bf16)(x[j] - (float)h);
    }
    return f;
}

// C[i][j] += sum_k X[i][k] Y[j][k]; lane holds C col j = l&15, rows i = 4*(l>>4)+r.
SR_DEV void mma(const Frag<bf16>& x, const Frag<bf16>& y, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.v, y.v, c, 0, 0, 0);
}
SR_DEV void mma(const Frag<bf3>& x, const Frag<bf3>& y, f32x4& c) {  // small terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.lo, y.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.hi, y.lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.hi, y.hi, c, 0, 0, 0);
}
SR_DEV void mma(const Frag<float>& x, const Frag<float>& y, f32x4& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(x.lo[j], y.lo[j], c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(x.hi[j], y.hi[j], c, 0, 0, 0);
}

// ----------------------------------------------------------------------------- 8-element loads
// Load 8 consecutive elements (one K-group) of type TIn and return them as a Frag<TC>.
template <typename TC, typename TIn>
SR_DEV Frag<TC> load_group(const TIn* p);
template <>
SR_DEV Frag<bf16> load_group<bf16, bf16>(const bf16* p) {
    return *reinterpret_cast<const Frag<bf16>*>(p);
}
template <>
SR_DEV Frag<float> load_group<float, float>(const float* p) {
    return *reinterpret_cast<const Frag<float>*>(p);
}
template <>
SR_DEV Frag<bf16> load_group<bf16, float>(const float* p) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    Frag<bf16> f;
    f.v[0] = (bf16)a[0];
    f.v[1] = (bf16)a[1];
    f.v[2] = (bf16)a[2];
    f.v[3] = (bf16)a[3];
    f.v[4] = (bf16)b[0];
    f.v[5] = (bf16)b[1];
    f.v[6] = (bf16)b[2];
    f.v[7] = (bf16)b[3];
    return f;
}

template <>
SR_DEV Frag<bf3> load_group<bf3, float>(const float* p) {
    float v[8];
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    return frag_make<bf3>(v);
}

SR_DEV void load8f(const float* p, float* v) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

// ----------------------------------------------------------------------------- 4-element stores
SR_DEV void store4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
SR_DEV void store4(bf16* p, const f32x4& v) {
    bf16x4 r;
    r[0] = (bf16)v[0]; r[1] = (bf16)v[1]; r[2] = (bf16)v[2]; r[3] = (bf16)v[3];
    *reinterpret_cast<bf16x4*>(p) = r;
}
SR_DEV f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
SR_DEV f32x4 load4(const bf16* p) {
    bf16x4 r = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
}

// ----------------------------------------------------------------------------- activations
enum { SR_ACT_NONE = 0, SR_ACT_RELU = 1, SR_ACT_LRELU = 2, SR_ACT_GELU = 3 };

SR_DEV float apply_act(float x, int act) {
    switch (act) {
        case SR_ACT_RELU: return x > 0.f ? x : 0.f;
        case SR_ACT_LRELU: return x > 0.f ? x : 0.01f * x;  // nn.LeakyReLU default slope
        case SR_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));  // exact erf form
        default: return x;
    }
}

// Compile-time activation (the run-time switch above gets if-converted by hipcc: every element then
// pays for erff even when act == NONE).  Kernels dispatch ONCE per wave with act_dispatch().
template <int ACT>
SR_DEV float act_ct(float x, float slope = 0.01f) {
    if constexpr (ACT == SR_ACT_RELU) return x > 0.f ? x : 0.f;
    if constexpr (ACT == SR_ACT_LRELU) return x > 0.f ? x : slope * x;
    if constexpr (ACT == SR_ACT_GELU) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    return x;
}
template <int V>
struct IntC {
    static constexpr int value = V;
};
// f(IntC<ACT>{}) under a wave-uniform branch on the run-time activation code.
template <typename F>
SR_DEV void act_dispatch(int act, F&& f) {
    if (act == SR_ACT_NONE)
        f(IntC<SR_ACT_NONE>{});
    else if (act == SR_ACT_RELU)
        f(IntC<SR_ACT_RELU>{});
    else if (act == SR_ACT_LRELU)
        f(IntC<SR_ACT_LRELU>{});
    else
        f(IntC<SR_ACT_GELU>{});
}

// GELU with erf from Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7): ~14 VALU ops instead of the
// ~100 of ocml erff.  Used by the bf16 kernels (the exact-fp32 path keeps erff).
SR_DEV float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float e = 1.0f - p * t * __expf(-z * z);  // erf(|x|/sqrt2)
    const float h = 0.5f * x;
    return h + fabsf(h) * e;  // 0.5*x*(1 + sign(x)*erf(|x|/sqrt2))
}

// Phi(x) and phi(x) of the standard normal with erf from Abramowitz-Stegun 7.1.26 (as gelu_fast): gelu = x Phi, gelu' = Phi + x phi
SR_DEV void gauss(float x, float& Phi, float& phi) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float ex = __expf(-z * z);
    const float e = 1.0f - p * t * ex;
    Phi = x >= 0.f ? 0.5f + 0.5f * e : 0.5f - 0.5f * e;
    phi = ex * 0.3989422804014327f;
}

// tanh-form GELU written as x * sigmoid(2u): 5 VALU + v_exp + v_rcp.  |error| <= 4.8e-4 against the exact erf
// form, i.e. ~8x below the bf16 rounding of the value it produces; used ONLY where the result is stored as bf16.
SR_DEV float gelu_bf16(float x) {
    const float s = x * x;
    const float p = fmaf(s, -0.1029432f, -2.3022082f);  // -(2*sqrt(2/pi)*log2(e)) * (1 + 0.044715 x^2)
    const float e = __builtin_amdgcn_exp2f(x * p);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// window-order row -> image-order row (roll(-shift) + window_partition as one gather;
// window_reverse + roll(+shift) is the same map used as a scatter).
struct WinMap {
    int H, W, ws, shift, nwx, ntok, hw;
    int shift_y;  // = shift, or 0 for a row strip that the host already rolled (SR_Y_STRIP*)
    SR_DEV int operator()(int row) const {
        int b = row / hw;
        int rem = row - b * hw;
        int win = rem / ntok;
        int tok = rem - win * ntok;
        int wy = win / nwx, wx = win - wy * nwx;
        int i = tok / ws, j = tok - i * ws;
        int y = wy * ws + i + shift_y;
        int x = wx * ws + j + shift;
        if (y >= H) y -= H;
        if (x >= W) x -= W;
        return (b * H + y) * W + x;
    }
};

// Cross-lane butterfly steps without the LDS crossbar (ds_bpermute costs a full LDS round trip per step and sat on the
// critical path of every LayerNorm / softmax): lane ^ 32 and lane ^ 16 via v_permlane{32,16}_swap (gfx950), lane ^ 8 via
// DPP row_ror:8.  The swap builtins mis-compile when both operands are the same value (ROCm 7.2), hence inline asm; the
// s_nop covers the VALU-write -> permlane-read hazard (2 wait states).
SR_DEV void lane_xor32_pair(float x, float& a, float& b) {
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
SR_DEV void lane_xor16_pair(float x, float& a, float& b) {
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
SR_DEV float wave_sum_xor(float v, int mask) {  // mask is a literal at every call site
    if (mask == 32) {
        float a, b;
        lane_xor32_pair(v, a, b);
        return a + b;
    }
    if (mask == 16) {
        float a, b;
        lane_xor16_pair(v, a, b);
        return a + b;
    }
    if (mask == 8) return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    return v + __shfl_xor(v, mask, 64);
}
SR_DEV float wave_max_xor(float v, int mask) {
    if (mask == 32) {
        float a, b;
        lane_xor32_pair(v, a, b);
        return fmaxf(a, b);
    }
    if (mask == 16) {
        float a, b;
        lane_xor16_pair(v, a, b);
        return fmaxf(a, b);
    }
    return fmaxf(v, __shfl_xor(v, mask, 64));
}

// Exact unsigned division by a run-time constant (Granlund-Montgomery round-up method); the
// multiplier is computed on the host so that no kernel executes an integer divide.
struct FastDiv {
    uint32_t mul, sh1, sh2, d;
    SR_DEV uint32_t div(uint32_t n) const {
        const uint32_t t = __umulhi(mul, n);
        return (t + ((n - t) >> sh1)) >> sh2;
    }
    SR_DEV void divmod(uint32_t n, uint32_t& q, uint32_t& r) const {
        q = div(n);
        r = n - q * d;
    }
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (uint32_t)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}
