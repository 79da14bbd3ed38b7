// Channel-attention squeeze (RCAN / HAT: common.py ChannelAttention, hat.py:25-38), shared by sr_elem.hip (sr_channel_attention,
// sr_channel_gate) and sr_swin_tail.hip (gate recomputed in the tail kernel's prologue).  Included inside an anonymous namespace user.
#pragma once
#include "sr_common.h"
#include <type_traits>

namespace {

// The squeeze half (pool partials -> mean -> 2-layer MLP -> sigmoid) for image b; returns y_scale * gate (C_p floats in the scratch).
// Recomputed by every workgroup that needs it, so it must be short: every step spreads its (independent) loads over all 256 threads
// instead of walking n_tiles / C / Cr dependent loads in a few of them.  Three pieces, so that a caller can request the MLP operands early and
// hand the partial sums over from LDS (sr_swin_tail.hip); ca_squeeze() below is their plain composition.
constexpr int CA_SLICES = 8;
// scratch layout: mean [C_p] | hid [Cr rounded up to 4] | gate [C_p] | part [CA_SLICES][C_p]  (16-byte aligned pieces when C_p is a multiple of 4)
SR_DEV constexpr int ca_hid_pad(int Cr) { return (Cr + 3) & ~3; }
SR_DEV float* ca_part(float* sm, int C_p, int Cr) { return sm + C_p + ca_hid_pad(Cr) + C_p; }
SR_DEV bool ca_small(const SrChannelAttn& a) { return a.Cr <= 8 && a.C <= 256 && a.C_p <= 256; }  // every model here

struct CaOps {  // the MLP operands of one thread (ca_small geometries)
    float w1v[2][4], b1v[2], w2v[8], b2v;
};
// The MLP operands do not depend on the pool sums: they are requested together with the partials instead of one exposed L2 round trip per
// phase (this prologue is the whole of sr_channel_gate: 14 -> 8 us).
// Template arguments CP / CC / CR (here and below): the padded and true channel counts and the squeeze width as compile-time constants (0 = read them from `a`).
// Callers that run the squeeze in every workgroup at one wave per SIMD (sr_swin_tail.hip) instantiate their geometry: the generic form is ~1,000 instructions of
// guards and address arithmetic, and at one wave per SIMD every instruction is ~5 cycles of the chain.  Same arithmetic in the same order either way.
template <int CC = 0, int CR = 0>
SR_DEV void ca_load_ops(const SrChannelAttn& a, CaOps& o) {
    const int C = CC ? CC : a.C, Cr = CR ? CR : a.Cr;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    o.b2v = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = wave + 4 * u;
        o.b1v[u] = j < Cr ? a.b1[j] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int c = lane + 64 * v;
            o.w1v[u][v] = (j < Cr && c < C) ? a.w1[j * C + c] : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o.w2v[j] = (tid < C && j < Cr) ? a.w2[tid * Cr + j] : 0.f;
    if (tid < C) o.b2v = a.b2[tid];
}

// slice sums of one image's partials pool_img[n_tiles][C_p] (global memory, or an LDS copy of the same floats) -> part[CA_SLICES][C_p]: work item =
// (slice, channel quad); 16-byte loads, 8 partials in flight per item (unconditional loads from clamped addresses, added in slot order: the
// same sums as one scalar load at a time, which made this prologue 12 us per 32 slots)
template <int CP = 0, int CC = 0, typename P>
SR_DEV void ca_slice_sums(const SrChannelAttn& a, P pool_img, float* part) {
    const int C_p = CP ? CP : a.C_p, C = CC ? CC : a.C;
    const int tid = threadIdx.x;
    const int quads = C_p >> 2;
    for (int idx = tid; idx < CA_SLICES * quads; idx += 256) {
        const int sl = idx / quads, q = idx - sl * quads;
        f32x4 s = (f32x4)(0.0f);
        for (int t0 = sl; t0 < a.n_tiles; t0 += 8 * CA_SLICES) {
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(pool_img + min(t0 + k * CA_SLICES, a.n_tiles - 1) * C_p + 4 * q);  // (one image: < 2^31 floats)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (t0 + k * CA_SLICES < a.n_tiles) s += v[k];
        }
        f32x4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = (4 * q + r < C) ? s[r] : 0.f;
        *reinterpret_cast<f32x4*>(part + sl * C_p + 4 * q) = o4;
    }
}

// sum over the 64 lanes of a wave without LDS traffic (four DPP steps inside each row of 16 lanes, then the four row sums): every lane returns the same bits
SR_DEV float ca_wave_sum(float v) {
    auto dpp_add = [](float x, auto ctrl) {
        return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v = dpp_add(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1, 0, 3, 2]
    v = dpp_add(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2, 3, 0, 1]
    v = dpp_add(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v = dpp_add(v, std::integral_constant<int, 0x140>{});  // row_mirror
    const int vi = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return (r0 + r1) + (r2 + r3);
}

// part -> mean -> hidden -> gate; `o` is read when ca_small(a).  Starts with the barrier behind the slice sums.  Leaves mean [C_p] | hid [Cr] | gate [C_p]
// in sm (sr_tr_ca_bwd reads all three).  Called with 256 threads, all lanes active.
template <int CP = 0, int CC = 0, int CR = 0>
SR_DEV float* ca_finish(const SrChannelAttn& a_, const CaOps& o, float* sm) {
    struct {  // the geometry as constants where given
        int C_p, C, Cr, H, W;
        const float *w1, *b1, *w2, *b2;
        float y_scale;
    } a = {CP ? CP : a_.C_p, CC ? CC : a_.C, CR ? CR : a_.Cr, a_.H, a_.W, a_.w1, a_.b1, a_.w2, a_.b2, a_.y_scale};
    float* mean = sm;                 // [C_p]
    float* hid = sm + a.C_p;          // [Cr]
    float* gate = hid + ca_hid_pad(a.Cr);  // [C_p]
    float* part = gate + a.C_p;       // [CA_SLICES][C_p] partial channel sums
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const float inv = 1.0f / (float)(a.H * a.W);
    const bool small = (CP && CC && CR) ? true : ca_small(a_);
    __syncthreads();
    if (small) {
        // Three barriers instead of five and no ds_bpermute chains (this runs in the prologue of every sr_swin_tail workgroup): every wave forms the
        // channel means it needs itself (channels lane + 64 v), its two hidden units by DPP sums, and only hid / gate cross waves through LDS.
        float mv[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int c = lane + 64 * v;
            float s = 0.f;
            if (c < a.C_p) {
#pragma unroll
                for (int sl = 0; sl < CA_SLICES; ++sl) s += part[sl * a.C_p + c];
                if (wave == 0) mean[c] = s * inv;
            }
            mv[v] = s * inv;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = wave + 4 * u;
            if (j < a.Cr) {  // one wave per hidden unit: lanes split the channels (the operands of channels >= C are 0)
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v) s += o.w1v[u][v] * mv[v];
                s = ca_wave_sum(s) + o.b1v[u];
                if (lane == 0) hid[j] = s > 0.f ? s : 0.f;
            }
        }
    } else {
        for (int c = tid; c < a.C_p; c += 256) {
            float s = 0.f;
#pragma unroll
            for (int sl = 0; sl < CA_SLICES; ++sl) s += part[sl * a.C_p + c];
            mean[c] = s * inv;
        }
        __syncthreads();
        for (int j = wave; j < a.Cr; j += 4) {
            float s = 0.f;
            for (int c = lane; c < a.C; c += 64) s += a.w1[j * a.C + c] * mean[c];
#pragma unroll
            for (int of = 32; of >= 1; of >>= 1) s += __shfl_xor(s, of, 64);
            if (lane == 0) {
                s += a.b1[j];
                hid[j] = s > 0.f ? s : 0.f;
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < a.C_p; c += 256) {
        float s = 0.f;
        if (c < a.C) {
            if (small) {
                s = o.b2v;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < a.Cr) s += o.w2v[j] * hid[j];
            } else {
                s = a.b2[c];
                for (int j = 0; j < a.Cr; ++j) s += a.w2[c * a.Cr + j] * hid[j];
            }
            s = 1.0f / (1.0f + __expf(-s));
        }
        gate[c] = s * a.y_scale;
    }
    __syncthreads();
    return gate;
}
constexpr int ca_scratch_floats(int C_p, int Cr) { return C_p + ((Cr + 3) & ~3) + C_p + CA_SLICES * C_p; }

SR_DEV float* ca_squeeze(const SrChannelAttn& a, int b, float* sm) {
    CaOps o = {};
    if (ca_small(a)) ca_load_ops(a, o);
    ca_slice_sums(a, a.pool_partial + (size_t)b * a.n_tiles * a.C_p, ca_part(sm, a.C_p, a.Cr));
    return ca_finish(a, o, sm);
}

}  // namespace
