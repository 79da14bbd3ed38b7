// Channel-attention squeeze (RCAN / HAT: common.py ChannelAttention, hat.py:25-38), shared by sr_elem.hip (sr_channel_attention,
// sr_channel_gate) and sr_swin_tail.hip (gate recomputed in the tail kernel's prologue).  Included inside an anonymous namespace user.
#pragma once
#include "sr_common.h"

namespace {

// The squeeze half (pool partials -> mean -> 2-layer MLP -> sigmoid) for image b; leaves y_scale * gate in sm + C_p + Cr (C_p floats).
// Recomputed by every workgroup that needs it, so it must be short: every step spreads its (independent) loads over all 256 threads
// instead of walking n_tiles / C / Cr dependent loads in a few of them.
SR_DEV float* ca_squeeze(const SrChannelAttn& a, int b, float* sm) {
    float* mean = sm;                 // [C_p]
    float* hid = sm + a.C_p;          // [Cr]
    float* gate = hid + a.Cr;         // [C_p]
    float* part = gate + a.C_p;       // [CA_SLICES][C_p] partial channel sums
    constexpr int CA_SLICES = 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const float inv = 1.0f / (float)(a.H * a.W);
    // The MLP operands do not depend on the pool sums: with Cr <= 8 and C <= 256 (every model here) they are requested now, together
    // with the partials, instead of one exposed L2 round trip per phase (this prologue is the whole of sr_channel_gate: 14 -> 8 us).
    const bool small = a.Cr <= 8 && a.C <= 256 && a.C_p <= 256;
    float w1v[2][4], b1v[2], w2v[8], b2v = 0.f;
    if (small) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = wave + 4 * u;
            b1v[u] = j < a.Cr ? a.b1[j] : 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = lane + 64 * v;
                w1v[u][v] = (j < a.Cr && c < a.C) ? a.w1[j * a.C + c] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) w2v[j] = (tid < a.C && j < a.Cr) ? a.w2[tid * a.Cr + j] : 0.f;
        if (tid < a.C) b2v = a.b2[tid];
    }
    // slice sums: work item = (slice, channel quad); 16-byte loads, 8 partials in flight per item (unconditional loads from clamped
    // addresses, added in slot order: the same sums as one scalar load at a time, which made this prologue 12 us per 32 slots)
    {
        const int quads = a.C_p >> 2;
        for (int idx = tid; idx < CA_SLICES * quads; idx += 256) {
            const int sl = idx / quads, q = idx - sl * quads;
            f32x4 s = (f32x4)(0.0f);
            const float* pp = a.pool_partial + (size_t)b * a.n_tiles * a.C_p + 4 * q;
            for (int t0 = sl; t0 < a.n_tiles; t0 += 8 * CA_SLICES) {
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(pp + (size_t)min(t0 + k * CA_SLICES, a.n_tiles - 1) * a.C_p);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (t0 + k * CA_SLICES < a.n_tiles) s += v[k];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) part[sl * a.C_p + 4 * q + r] = (4 * q + r < a.C) ? s[r] : 0.f;
        }
    }
    __syncthreads();
    for (int c = tid; c < a.C_p; c += 256) {
        float s = 0.f;
#pragma unroll
        for (int sl = 0; sl < CA_SLICES; ++sl) s += part[sl * a.C_p + c];
        mean[c] = s * inv;
    }
    __syncthreads();
    if (small) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = wave + 4 * u;
            if (j < a.Cr) {  // one wave per hidden unit: lanes split the channels (same order of additions as the loop form below)
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = lane + 64 * v;
                    if (c < a.C) s += w1v[u][v] * mean[c];
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
                if (lane == 0) {
                    s += b1v[u];
                    hid[j] = s > 0.f ? s : 0.f;
                }
            }
        }
    } else {
        for (int j = wave; j < a.Cr; j += 4) {
            float s = 0.f;
            for (int c = lane; c < a.C; c += 64) s += a.w1[j * a.C + c] * mean[c];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
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
                s = b2v;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < a.Cr) s += w2v[j] * hid[j];
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

}  // namespace
