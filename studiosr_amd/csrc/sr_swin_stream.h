// Building blocks shared by the stream-form Swin kernels (sr_swin_block3.hip: the whole SwinTransformerBlock; sr_swin_tail.hip: projection +
// MLP half behind a separate attention kernel): LDS image geometry, operand packing, the fp32 tile rows through LDS and the per-wave
// weight stream with its register ring.  Included inside each translation unit (everything is internal linkage).
#pragma once
#include "sr_common.h"
#include "sr_host.h"

namespace {

constexpr int NTOK = 64, WS = 8, NSLOT = 48;
constexpr int PAD_D = 30;  // first pad feature of a head: v[:, 30] = 1 -> row 30 of O^T = softmax denominator
constexpr int ONE_C = 180; // first pad channel of the stream: the LayerNorm images carry 1.0 in channels 180, 181
// LDS image sizes in 16-B (bf16) / 32-B (bf3 = hi | lo split operands, compute type SR_BF16X3) cells
constexpr int CELLS_A = 24 * 64;          // LayerNorm image [24 k-groups][64 tokens]
constexpr int CELLS_Q = 2 * 4 * 64;       // Q image [2 heads][4 d-groups][64 tokens]; the attention output O overwrites it atom by atom
constexpr int CELLS_K = 2 * 4 * 64;       // K image [2 heads][4 d-groups][64 keys]
constexpr int CELLS_V = 2 * 2 * 4 * 32;   // V^T image [2 heads][2 key steps][4 key groups][32 d]
static_assert(CELLS_Q + CELLS_K + CELLS_V == CELLS_A, "the hidden-half image reuses the Q / K / V region");
constexpr int LDS_RED = 64 * 4 * 2 * 4;   // LayerNorm partial sums [64 tokens][4 waves][2]
// The fp32 window tile (x at kernel entry, the result at its end) passes through LDS as 64 token rows of 768 B at a stride of
// XS = 784 B, laid over the image regions (all free at both moments): full rows travel between HBM and LDS with ADJACENT lanes on
// ADJACENT addresses (the accumulator layout has adjacent lanes on different token rows: every such load / store cost the
// vector-memory path four cache-line accesses per quad, 36 % of the kernel's TA busy time in profiles/r03_block_kernel_counters.txt),
// and the 784-B stride makes the accumulator-layout ds_read_b128 / ds_write_b128 side conflict-free.
constexpr int XS = 784;
constexpr int LDS_X = 64 * XS;
template <typename T>
struct Lds {
    static constexpr int IMG = 2 * CELLS_A * (int)sizeof(Frag<T>);
    static constexpr int RED_OFF = IMG > LDS_X ? IMG : LDS_X;  // the LayerNorm partials sit behind the images AND the x tile
    static constexpr int TOTAL = RED_OFF + LDS_RED;
};
static_assert(3 * Lds<bf16>::TOTAL <= 160 * 1024, "bf16: three workgroups per CU");
static_assert(Lds<bf3>::TOTAL <= 160 * 1024, "bf16x3: one workgroup per CU");

SR_DEV bf16x4 cvt4(const f32x4& v) {
    bf16x4 r;
    r[0] = (bf16)v[0]; r[1] = (bf16)v[1]; r[2] = (bf16)v[2]; r[3] = (bf16)v[3];
    return r;
}
SR_DEV f32x4 widen4(const bf16x4& v) { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
// 8 fp32 values (two accumulator quads) as one operand fragment
template <typename T>
SR_DEV Frag<T> pack2(const f32x4& lo, const f32x4& hi);
template <>
SR_DEV Frag<bf16> pack2<bf16>(const f32x4& lo, const f32x4& hi) {
    Frag<bf16> f;
    f.v[0] = (bf16)lo[0]; f.v[1] = (bf16)lo[1]; f.v[2] = (bf16)lo[2]; f.v[3] = (bf16)lo[3];
    f.v[4] = (bf16)hi[0]; f.v[5] = (bf16)hi[1]; f.v[6] = (bf16)hi[2]; f.v[7] = (bf16)hi[3];
    return f;
}
template <>
SR_DEV Frag<bf3> pack2<bf3>(const f32x4& lo, const f32x4& hi) {  // x = h + l with h = bf16(x), l = bf16(x - h): 16 mantissa bits
    Frag<bf3> f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bf16 h0 = (bf16)lo[j], h1 = (bf16)hi[j];
        f.hi[j] = h0;
        f.hi[4 + j] = h1;
        f.lo[j] = (bf16)(lo[j] - (float)h0);
        f.lo[4 + j] = (bf16)(hi[j] - (float)h1);
    }
    return f;
}
// 4 fp32 values into the 8-byte half `half` of an image cell
SR_DEV void st_half(Frag<bf16>* cell, int half, const f32x4& v) { *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(cell) + half * 8) = cvt4(v); }
SR_DEV void st_half(Frag<bf3>* cell, int half, const f32x4& v) {
    const bf16x4 h = cvt4(v);
    const bf16x4 l = cvt4(v - widen4(h));
    *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(cell) + half * 8) = h;
    *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(cell) + 16 + half * 8) = l;
}

// D = X Y^T + C with C a separate register set (the first MFMA of a chain: C = 0 or a bias tile); sr_common.h mma() accumulates in place
SR_DEV f32x4 mma_c(const Frag<bf16>& x, const Frag<bf16>& y, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.v, y.v, c, 0, 0, 0); }
SR_DEV f32x4 mma_c(const Frag<bf3>& x, const Frag<bf3>& y, const f32x4& c) {  // small terms first, as mma(bf3)
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.lo, y.hi, c, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.hi, y.lo, d, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.hi, y.hi, d, 0, 0, 0);
}
// first MFMA of an accumulation chain: C is the inline constant 0 (no v_mov initialisation of the accumulator)
template <typename T>
SR_DEV void mma0(const Frag<T>& x, const Frag<T>& y, f32x4& c) { c = mma_c(x, y, (f32x4)(0.0f)); }

template <typename T>
SR_DEV float gelu_op(float x);
template <>
SR_DEV float gelu_op<bf16>(float x) { return gelu_bf16(x); }  // x * sigmoid form: 4.8e-4 below the bf16 rounding of its own output
template <>
SR_DEV float gelu_op<bf3>(float x) { return gelu_fast(x); }   // erf to 1.5e-7: the fp32-class path

// Wave priority by phase: a workgroup in its attention passes outranks one in its MLP, so the three workgroups of a CU (which start
// together) stay closer in progress and the last one does not finish alone (-2..3 % at B = 8 / 16; SR_EXP_PHPRIO=0 switches it off).
#ifndef SR_EXP_PHPRIO
#define SR_EXP_PHPRIO 1
#endif
#if defined(SR_EXP_PHPRIO) && SR_EXP_PHPRIO == 1
#define PHASE_PRIO(ph) __builtin_amdgcn_s_setprio(2 - (ph))
#elif defined(SR_EXP_PHPRIO) && SR_EXP_PHPRIO == 2
#define PHASE_PRIO(ph) __builtin_amdgcn_s_setprio(ph)
#elif defined(SR_EXP_PHPRIO) && SR_EXP_PHPRIO == 3
#define PHASE_PRIO(ph) __builtin_amdgcn_s_setprio((ph) == 0 ? 3 : 2 - (ph))
#else
#define PHASE_PRIO(ph) do { } while (0)
#endif
#ifdef SR_EXP_W0
#define WSLOT(s) 0
#else
#define WSLOT(s) (s)
#endif
#ifdef SR_EXP_NOBAR
#define BLOCK_SYNC() __builtin_amdgcn_sched_barrier(0)
#elif defined(SR_EXP_RAWBAR)
// experiment: barrier that waits for LDS operations only.  Measured +-0 against __syncthreads(): hipcc already emits a bare s_barrier here
// (its fence only waits for vector-memory operations when an LDS-DMA it knows of is pending; the weight ring survives the barrier)
#define BLOCK_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define BLOCK_SYNC() __syncthreads()
#endif

SR_DEV float bcast_row3(float x) {  // value of lane (l & 15) + 48 in every lane
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    float c = b, d = b;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    return d;
}
// v[g] summed over the four 16-lane rows, the sum of index g delivered to row g
SR_DEV float rows_reduce_scatter4(float v0, float v1, float v2, float v3) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v0), "+v"(v2));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v1), "+v"(v3));
    float u02 = v0 + v2, u13 = v1 + v3;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u02), "+v"(u13));
    return u02 + u13;
}
// experiment (SR_EXP_GELUPOLY): x * clamp(0.5 + x Q(x^2), 0, 1) with a degree-4 Q: 7 plain VALU, |error| <= 1.4e-3 against erf-GELU
SR_DEV float gelu_poly(float x) {
    const float s = x * x;
    float q = __builtin_fmaf(s, 1.30341647e-05f, -4.82968101e-04f);
    q = __builtin_fmaf(q, s, 7.36121539e-03f);
    q = __builtin_fmaf(q, s, -6.23224052e-02f);
    q = __builtin_fmaf(q, s, 3.97474261e-01f);
    const float phi = __builtin_amdgcn_fmed3f(__builtin_fmaf(x, q, 0.5f), 0.0f, 1.0f);
    return x * phi;
}
SR_DEV float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// one token row of the fp32 tile: global (wave-uniform row pointer + 16 B per lane, lanes 0..47) -> LDS at `lds_dst` + 16 B per lane, no VGPR staging
// cache policy of the once-read x rows and the once-written result rows (experiment knobs: 0 default, 1 sc1, 2 nt, 3 sc0 sc1)
#ifndef SR_X_LOAD_POLICY_ID
#define SR_X_LOAD_POLICY_ID 0
#endif
#ifndef SR_X_STORE_POLICY_ID
#define SR_X_STORE_POLICY_ID 1  // sc1 (write-through): the 32 MB result leaves L2 as it is written instead of at the kernel boundary: -6 % per launch at
                                // B = 8; `nt` does the same for one batch in flight but costs 4 % with two; nt on the loads: +2 % (profiles/r03_block_kernel_ablation.txt)
#endif
#define SR_POLICY_STR_0 ""
#define SR_POLICY_STR_1 " sc1"
#define SR_POLICY_STR_2 " nt"
#define SR_POLICY_STR_3 " sc0 sc1"
#define SR_POLICY_CAT(a, b) a##b
#define SR_POLICY_STR(id) SR_POLICY_CAT(SR_POLICY_STR_, id)
#define SR_X_LOAD_POLICY SR_POLICY_STR(SR_X_LOAD_POLICY_ID)
#define SR_X_STORE_POLICY SR_POLICY_STR(SR_X_STORE_POLICY_ID)
SR_DEV void store_row48(float* row, const f32x4& v, int lane) {  // lanes 0..47: 16 B each at row + 16 lane
    if (lane < 48) asm volatile("global_store_dwordx4 %0, %1, %2" SR_X_STORE_POLICY "\n\ts_nop 1" ::"v"(lane * 16), "v"(v), "s"(row) : "memory");
}
SR_DEV void dma_row48(const float* row, unsigned lds_dst, int lane) {
    if (lane < 48) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" SR_X_LOAD_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane * 16), "s"(row), "s"(lds_dst)
                     : "memory");
    }
}
// the same with the row given as [scalar base pointer of the image row] + [scalar byte offset of the pixel]: the offset joins the lane part in a VGPR
// inside the asm block (one v_add per row), so that a window's 16 rows per wave need 2 pointers + 8 offsets in SGPRs instead of 16 pointers
SR_DEV void store_row48(float* base, int byte_off, const f32x4& v, int lane) {
    if (lane < 48) {
        int voff;
        asm volatile("v_add_u32 %0, %1, %2\n\tglobal_store_dwordx4 %0, %3, %4" SR_X_STORE_POLICY "\n\ts_nop 1" : "=&v"(voff) : "s"(byte_off), "v"(lane * 16), "v"(v), "s"(base) : "memory");
    }
}
SR_DEV void dma_row48(const float* base, int byte_off, unsigned lds_dst, int lane) {
    if (lane < 48) {
        unsigned keep;
        int voff;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\tv_add_u32 %1, %3, %2\n\tglobal_load_lds_dwordx4 %1, %4" SR_X_LOAD_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&v"(voff)
                     : "v"(lane * 16), "s"(byte_off), "s"(base), "s"(lds_dst)
                     : "memory");
    }
}

// The weight stream of one wave: slot s = fragments [12 s + 3 w, 12 s + 3 w + 3) of the packed block, ring of RING slots in registers.
// Buffer loads: the fragment's byte offset is a scalar (soffset), the lane part one shared VGPR -- no per-load 64-bit address arithmetic.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#ifndef SR_W_AUX
#define SR_W_AUX 0  // cache policy bits of the weight-fragment loads (experiment knob: 1 sc0, 2 nt, 16 sc1)
#endif
SR_DEV void buf_load_frag(Frag<bf16>& f, __amdgpu_buffer_rsrc_t rsrc, int lane, int frag_index) {
    f.v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, frag_index * 1024, SR_W_AUX));
}
SR_DEV void buf_load_frag(Frag<bf3>& f, __amdgpu_buffer_rsrc_t rsrc, int lane, int frag_index) {  // per lane 8 hi | 8 lo (32 B)
    f.hi = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 32, frag_index * 2048, 0));
    f.lo = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 32, frag_index * 2048 + 16, 0));
}
#ifndef SR_RING_DIST
#define SR_RING_DIST 2
#endif
template <typename T, int NS = NSLOT, int D = SR_RING_DIST>
struct WStream {
    static constexpr int DIST = D, RING = DIST + 1;
    static constexpr int LOADS_PER_SLOT = 3 * (int)sizeof(Frag<T>) / 16;  // buffer_load instructions per slot (vmcnt bookkeeping)
    Frag<T> r[RING][3];
    __amdgpu_buffer_rsrc_t rsrc;
    int wave_frag;  // 3 w (scalar)
    SR_DEV void load(int s, int lane) {  // s is a compile-time constant at every call site (unrolled)
#pragma unroll
        for (int n = 0; n < 3; ++n) buf_load_frag(r[s % RING][n], rsrc, lane, wave_frag + WSLOT(s) * 12 + n);
    }
    // NST uniform steps starting at slot s0: loada(c, h, a) reads the two activation fragments (m-tiles 2h, 2h+1) of the stage's K-chunk c,
    // compute(c, h, b, a) issues their 6 MFMAs.  Half 1 of chunk c and half 0 of chunk c + 1 are read under the MFMAs before them.
    // HV = 1 (32-token workgroups, sr_swin_tail.hip): only half 0 exists; chunk c + 1's fragments are read under chunk c's MFMAs.
    template <int NST, int HV = 2, typename LoadA, typename Compute>
    SR_DEV void run(int s0, int lane, LoadA&& loada, Compute&& compute) {
        Frag<T> a0[2], a1[2];
        loada(0, 0, a0);
        if constexpr (HV == 1) {
#pragma unroll
            for (int c = 0; c < NST; ++c) {
                const int s = s0 + c;
                if (s + DIST < NS) load(s + DIST, lane);
                if (c + 1 < NST) loada(c + 1, 0, (c & 1) ? a0 : a1);
                compute(c, 0, r[s % RING], (c & 1) ? a1 : a0);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#ifdef SR_EXP_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int c = 0; c < NST; ++c) {
            const int s = s0 + c;
            if (s + DIST < NS) load(s + DIST, lane);
            loada(c, 1, a1);
            compute(c, 0, r[s % RING], a0);
            if (c + 1 < NST) loada(c + 1, 0, a0);
            compute(c, 1, r[s % RING], a1);
#ifdef SR_EXP_SGB
            // the step as written: [weight slot loads][a1 reads][MFMAs of half 0][next a0 reads][MFMAs of half 1] -- the machine scheduler otherwise sinks the
            // fragment reads to just before their first use (profiles/r05_block_kernel_ablation.txt)
            {
                constexpr int FR = (int)sizeof(Frag<T>) / 16;  // ds_read_b128 per fragment, MFMAs per product
                constexpr int MM = FR == 1 ? 6 : 18;
                if (s + DIST < NS) __builtin_amdgcn_sched_group_barrier(0x020, LOADS_PER_SLOT, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * FR, 0);
#if SR_EXP_SGB == 2
                for (int q = 0; q < 2; ++q) {  // the next fragments' reads after the first third of a half's MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x008, MM / 3, 0);
                    if (q == 1 && c + 1 < NST) __builtin_amdgcn_sched_group_barrier(0x100, 2 * FR, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MM - MM / 3, 0);
                    if (q == 0 && c + 1 < NST) { }
                }
#else
                __builtin_amdgcn_sched_group_barrier(0x008, MM, 0);
                if (c + 1 < NST) __builtin_amdgcn_sched_group_barrier(0x100, 2 * FR, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MM, 0);
#endif
            }
#endif
#ifndef SR_EXP_NOSB
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
#ifdef SR_EXP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
};

}  // namespace
