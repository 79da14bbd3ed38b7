// HAT overlapping cross attention (hat.py:239-283): softmax(q k^T + bias) v where the keys / values of a
// ws x ws query window are its (ws+2p) x (ws+2p) neighbourhood, zero padded at the image border.
//
// Same register-only scheme as sr_attn.hip (one wave = one (window, head, 16-query tile), S^T = K Q^T with the
// S^T accumulator reused as the P^T operand of O^T = V^T P^T); what differs is where operands come from:
//   * K rows are gathered per lane straight from the zero-bordered image-order buffer the QKV projection wrote
//     (nn.Unfold is never materialised; the border supplies the zero padding),
//   * V^T fragments are two 8-byte loads from the transposed zero-bordered planes (4 consecutive keys of one
//     neighbourhood row are 4 consecutive pixels),
//   * the key count (576 / 144) is padded to a multiple of 32 with masked logits.
#include "sr_common.h"
#include "sr_host.h"

namespace {

template <typename TC>
SR_DEV Frag<TC> load_vt2(const TC* p0, const TC* p1);
template <>
SR_DEV Frag<bf16> load_vt2<bf16>(const bf16* p0, const bf16* p1) {
    bf16x4 a = *reinterpret_cast<const bf16x4*>(p0);
    bf16x4 b = *reinterpret_cast<const bf16x4*>(p1);
    Frag<bf16> f;
    f.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return f;
}
template <>
SR_DEV Frag<float> load_vt2<float>(const float* p0, const float* p1) {
    Frag<float> f;
    f.lo = *reinterpret_cast<const f32x4*>(p0);
    f.hi = *reinterpret_cast<const f32x4*>(p1);
    return f;
}
// 4 consecutive keys at a 2-element-aligned address (border logical pad not a multiple of 4: ws = 8, pad = 2)
SR_DEV Frag<bf16> load_vt2_half(const bf16* p0, const bf16* p1) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 a0 = *reinterpret_cast<const bf16x2*>(p0), a1 = *reinterpret_cast<const bf16x2*>(p0 + 2);
    const bf16x2 b0 = *reinterpret_cast<const bf16x2*>(p1), b1 = *reinterpret_cast<const bf16x2*>(p1 + 2);
    Frag<bf16> f;
    f.v[0] = a0[0]; f.v[1] = a0[1]; f.v[2] = a1[0]; f.v[3] = a1[1];
    f.v[4] = b0[0]; f.v[5] = b0[1]; f.v[6] = b1[0]; f.v[7] = b1[1];
    return f;
}
SR_DEV Frag<float> load_vt2_half(const float* p0, const float* p1) {
    Frag<float> f;
    f.lo = f32x4{p0[0], p0[1], p0[2], p0[3]};
    f.hi = f32x4{p1[0], p1[1], p1[2], p1[3]};
    return f;
}

template <typename TC>
SR_DEV Frag<TC> pack_pp(const f32x4& a, const f32x4& b);
template <>
SR_DEV Frag<bf16> pack_pp<bf16>(const f32x4& a, const f32x4& b) {
    Frag<bf16> f;
    f.v[0] = (bf16)a[0]; f.v[1] = (bf16)a[1]; f.v[2] = (bf16)a[2]; f.v[3] = (bf16)a[3];
    f.v[4] = (bf16)b[0]; f.v[5] = (bf16)b[1]; f.v[6] = (bf16)b[2]; f.v[7] = (bf16)b[3];
    return f;
}
template <>
SR_DEV Frag<float> pack_pp<float>(const f32x4& a, const f32x4& b) {
    Frag<float> f;
    f.lo = a;
    f.hi = b;
    return f;
}

// WS: window size; KT: key tiles incl. padding (even); hd_p fixed at 32.
template <typename TC, int WS, int KT, bool ALIGN4>
__global__ __launch_bounds__(256) void sr_oca_kernel(SrOcaAttn a) {
    constexpr int NQ = WS * WS, QT = NQ / 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwx = a.W / WS, nwy = a.H / WS;
    const int n_items = a.B * nwx * nwy * a.heads * QT;
    const int item = blockIdx.x * 4 + wave;
    if (item >= n_items) return;  // wave-uniform, no barriers
    const int qt = item % QT;
    const int bh = item / QT;
    const int head = bh % a.heads;
    const int bwin = bh / a.heads;
    const int win = bwin % (nwx * nwy), b = bwin / (nwx * nwy);
    const int wy = win / nwx, wx = win - wy * nwx;
    const int wse = WS + 2 * a.pad, nk = wse * wse;
    const int Hp2 = a.H + 2 * a.border, Wp2 = a.W + 2 * a.border;
    const int oy = wy * WS - a.pad + a.border, ox = wx * WS - a.pad + a.border;  // neighbourhood origin in bordered coordinates
    const int HP = a.heads * 32;
    const int lr = lane & 15, lg = lane >> 4;

    const TC* q = reinterpret_cast<const TC*>(a.q) + ((size_t)bh * NQ + qt * 16 + lr) * 32 + lg * 8;
    const Frag<TC> qf = *reinterpret_cast<const Frag<TC>*>(q);
    const TC* kimg = reinterpret_cast<const TC*>(a.k) + ((size_t)b * Hp2 * Wp2) * HP + head * 32 + lg * 8;
    const TC* vplane = reinterpret_cast<const TC*>(a.vt) + (((size_t)b * a.heads + head) * 32) * ((size_t)Hp2 * Wp2);

    // ---- S^T = K Q^T  (+ bias, padded keys masked)
    f32x4 s[KT];
    const float* bias = a.bias + ((size_t)head * NQ + qt * 16 + lr) * a.nk_pad + lg * 4;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        int key = kt * 16 + lr;
        if (key >= nk) key = nk - 1;  // padded tile: any in-bounds row, masked below
        const int ky = key / wse, kx = key - ky * wse;
        const Frag<TC> kf = *reinterpret_cast<const Frag<TC>*>(kimg + ((size_t)(oy + ky) * Wp2 + ox + kx) * HP);
        s[kt] = *reinterpret_cast<const f32x4*>(bias + kt * 16);
        mma(kf, qf, s[kt]);
        const int k0 = kt * 16 + lg * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (k0 + r >= nk) s[kt][r] = -1.0e30f;
    }
    // ---- softmax over keys
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
            const float e = __expf(s[kt][r] - mx);
            s[kt][r] = e;
            sum += e;
        }
    sum = wave_sum_xor(sum, 16);
    sum = wave_sum_xor(sum, 32);
    const float inv_sum = 1.0f / sum;

    // ---- O^T = V^T P^T ; K-slot (lane group g, element j) of step ks <-> key 32 ks + 16 (j >> 2) + 4 g + (j & 3)
    TC* out = reinterpret_cast<TC*>(a.out) + ((size_t)bwin * NQ + qt * 16 + lr) * HP + head * 32 + lg * 4;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        f32x4 o = (f32x4)(0.0f);
        const TC* vrow = vplane + (size_t)(dt * 16 + lr) * ((size_t)Hp2 * Wp2);
#pragma unroll
        for (int ks = 0; ks < KT / 2; ++ks) {
            int ka = ks * 32 + lg * 4, kb = ka + 16;
            if (ka >= nk) ka = 0;  // padded keys carry p == 0; keep the address in bounds
            if (kb >= nk) kb = 0;
            const int kay = ka / wse, kax = ka - kay * wse, kby = kb / wse, kbx = kb - kby * wse;
            const TC* pa = vrow + (size_t)(oy + kay) * Wp2 + ox + kax;
            const TC* pb = vrow + (size_t)(oy + kby) * Wp2 + ox + kbx;
            const Frag<TC> vf = ALIGN4 ? load_vt2<TC>(pa, pb) : load_vt2_half(pa, pb);
            mma(vf, pack_pp<TC>(s[2 * ks], s[2 * ks + 1]), o);
        }
        store4(out + dt * 16, o * inv_sum);
    }
}

template <typename TC>
int dispatch_oca(const SrOcaAttn& a, hipStream_t st) {
    const int nk = (a.ws + 2 * a.pad) * (a.ws + 2 * a.pad);
    const int items = a.B * (a.H / a.ws) * (a.W / a.ws) * a.heads * (a.ws * a.ws / 16);
    dim3 grid((items + 3) / 4);
    if (a.ws == 16 && nk == 576 && a.nk_pad == 576) {
        hipLaunchKernelGGL((sr_oca_kernel<TC, 16, 36, true>), grid, dim3(256), 0, st, a);
    } else if (a.ws == 8 && nk == 144 && a.nk_pad == 160) {
        hipLaunchKernelGGL((sr_oca_kernel<TC, 8, 10, false>), grid, dim3(256), 0, st, a);
    } else {
        sr_set_error("sr_oca_attention: unsupported ws=%d pad=%d nk_pad=%d", a.ws, a.pad, a.nk_pad);
        return SR_EUNSUPPORTED;
    }
    SR_CHECK_LAUNCH("sr_oca_attention");
    return SR_OK;
}

}  // namespace

extern "C" int sr_oca_attention(const SrOcaAttn* p, void* stream) {
    SR_REQUIRE(p && p->q && p->k && p->vt && p->bias && p->out, "sr_oca_attention: null pointer");
    const SrOcaAttn& a = *p;
    SR_REQUIRE(a.hd_p == 32 && a.ws > 0 && a.H % a.ws == 0 && a.W % a.ws == 0 && a.pad >= 0 && a.border >= a.pad && a.border % 4 == 0 && (a.ws + 2 * a.pad) % 4 == 0 && (a.pad % 2) == 0,
               "sr_oca_attention: bad geometry");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (a.dtype == SR_BF16X3) {  // ABI v11: fp32 tensors, split-operand MFMAs -- the flash form only
        SR_REQUIRE(sr_oca_attention_flash_supported(a), "sr_oca_attention: SR_BF16X3 needs the flash form (bias_frag, nk_frag)");
        return sr_oca_attention_flash(a, st);
    }
    if (sr_oca_attention_lds_supported(a)) return sr_oca_attention_lds(a, st);  // K / V^T / the bias table in LDS once per (window, head)
    if (sr_oca_attention_flash_supported(a)) return sr_oca_attention_flash(a, st);
    return a.dtype == SR_BF16 ? dispatch_oca<bf16>(a, st) : dispatch_oca<float>(a, st);
}
