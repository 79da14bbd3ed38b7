// Memory-bound helpers around the MFMA kernels: image ingest (pad + normalise + NCHW->NHWC),
// LayerNorm, standalone PixelShuffle and the channel-attention gate.  All HBM-bound: one pass,
// 16-byte accesses, no LDS except the tiny channel-attention MLP.
#include "sr_common.h"
#include "sr_host.h"
#include "sr_ca.h"

namespace {

// ----------------------------------------------------------------------------- ingest
// thread = one (pixel, 8-channel group) of the NHWC output.
template <typename TOut>
__global__ __launch_bounds__(256) void sr_ingest_kernel(const float* __restrict__ x, TOut* __restrict__ out, int B, int C, int H, int W, int Hp, int Wp,
                                                        int Cp, int pad_mode, const float* __restrict__ scale, const float* __restrict__ bias) {
    const int groups = Cp >> 3;
    const long total = (long)B * Hp * Wp * groups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        long p = i / groups;
        const int xp = (int)(p % Wp);
        p /= Wp;
        const int yp = (int)(p % Hp);
        const int b = (int)(p / Hp);
        int ys = yp, xs = xp;
        if (pad_mode == SR_PAD_EVAL_MIRROR) {  // cat([x, flip(x)])[: h + pad]  (edge-inclusive mirror)
            if (ys >= H) ys = 2 * H - 1 - ys;
            if (xs >= W) xs = 2 * W - 1 - xs;
        } else if (pad_mode == SR_PAD_REFLECT) {  // F.pad(..., "reflect")  (edge-exclusive mirror)
            if (ys >= H) ys = 2 * (H - 1) - ys;
            if (xs >= W) xs = 2 * (W - 1) - xs;
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = g * 8 + j;
            v[j] = (c < C) ? x[((size_t)(b * C + c) * H + ys) * W + xs] * scale[c] + bias[c] : 0.f;
        }
        TOut* o = out + (size_t)i * 8;
        store4(o, f32x4{v[0], v[1], v[2], v[3]});
        store4(o + 4, f32x4{v[4], v[5], v[6], v[7]});
    }
}

// ----------------------------------------------------------------------------- uint8 front / back end of Model.inference
// common.py:42-45: x = u8 / scale, HWC -> CHW ... out * scale, round (half to even), clip(0, 255), uint8, CHW -> HWC.
// IEEE fp32 division / multiplication and v_rndne, i.e. bit-identical to the numpy / torch ops of the reference.
__global__ __launch_bounds__(256) void sr_u8_to_nchw_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, long total, int C, int HW, float divisor) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {  // i over [B][C][HW]
        const long p = i % HW, bc = i / HW;
        const long c = bc % C, b = bc / C;
        out[i] = __fdiv_rn((float)in[(b * HW + p) * C + c], divisor);
    }
}
__global__ __launch_bounds__(256) void sr_nchw_to_u8_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, long total, int C, int HW, float mult) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {  // i over [B][HW][C]
        const long c = i % C, bp = i / C;
        const long p = bp % HW, b = bp / HW;
        const float v = rintf(__fmul_rn(in[(b * C + c) * HW + p], mult));
        out[i] = (uint8_t)fminf(fmaxf(v, 0.0f), 255.0f);  // NaN -> 0 like torch's clip + cast on this platform is not relied upon
    }
}

// ----------------------------------------------------------------------------- LayerNorm
// 8 lanes per row (lane kq owns K-groups kq, kq+8, ...), two-pass statistics in registers.
template <typename TOut>
__global__ __launch_bounds__(256) void sr_layernorm_kernel(const float* __restrict__ x, TOut* __restrict__ y, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int M, int C, int Cp, float eps) {
    const int lane = threadIdx.x & 63;
    const int r8 = lane & 7, kq = lane >> 3;
    const int KG = Cp >> 3;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + r8;
    const bool valid = row < M;
    const float* src = x + (size_t)(valid ? row : 0) * Cp;
    float v[6][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int kg = kq + 8 * i;
        if (kg < KG && valid)
            load8f(src + kg * 8, v[i]);
        else
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[i][j];
    }
    s = wave_sum_xor(s, 8);
    s = wave_sum_xor(s, 16);
    s = wave_sum_xor(s, 32);
    const float inv = 1.0f / (float)C;
    const float mean = s * inv;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = ((kq + 8 * i) * 8 + j < C) ? v[i][j] - mean : 0.f;
            q += d * d;
        }
    q = wave_sum_xor(q, 8);
    q = wave_sum_xor(q, 16);
    q = wave_sum_xor(q, 32);
    const float rstd = rsqrtf(q * inv + eps);
    if (!valid) return;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int kg = kq + 8 * i;
        if (kg < KG) {
            float gm[8], bt[8];
            load8f(gamma + kg * 8, gm);
            load8f(beta + kg * 8, bt);
            TOut* o = y + (size_t)row * Cp + kg * 8;
            store4(o, f32x4{(v[i][0] - mean) * rstd * gm[0] + bt[0], (v[i][1] - mean) * rstd * gm[1] + bt[1], (v[i][2] - mean) * rstd * gm[2] + bt[2],
                            (v[i][3] - mean) * rstd * gm[3] + bt[3]});
            store4(o + 4, f32x4{(v[i][4] - mean) * rstd * gm[4] + bt[4], (v[i][5] - mean) * rstd * gm[5] + bt[5], (v[i][6] - mean) * rstd * gm[6] + bt[6],
                                (v[i][7] - mean) * rstd * gm[7] + bt[7]});
        }
    }
}

// ----------------------------------------------------------------------------- PixelShuffle (NCHW, exact copy)
// out[b, c, y*r + i, x*r + j] = in[b, c*r*r + i*r + j, y, x].  Pure HBM traffic (2 x tensor bytes).
// Vector form: one thread owns V = 16 B / sizeof(T) consecutive x of one (b, c, y, i): it loads the r planes j = 0..r-1
// with one 16-B load each, interleaves them in registers and writes r x 16 B of ONE output row contiguously, so both
// sides move whole 16-B (per lane) / 1-KiB (per wave) runs.  Needs W % V == 0; other widths take the scalar form.
template <typename T, int R>
__global__ __launch_bounds__(256) void sr_pixel_shuffle_vec_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int Co, int H, int W) {
    constexpr int V = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const int wv = W / V;
    const long total = (long)B * Co * H * R * wv;
    const size_t plane = (size_t)H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int xv = (int)(idx % wv);
        long p = idx / wv;
        const int i = (int)(p % R);
        p /= R;
        const int y = (int)(p % H);
        const long bc = p / H;  // b * Co + c
        const T* src = in + ((size_t)bc * (R * R) + (size_t)i * R) * plane + (size_t)y * W + (size_t)xv * V;
        vec_t v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = *reinterpret_cast<const vec_t*>(src + (size_t)j * plane);
        T* dst = out + ((size_t)bc * H * R + (size_t)y * R + i) * ((size_t)W * R) + (size_t)xv * V * R;
#pragma unroll
        for (int o = 0; o < R; ++o) {  // output vector o holds elements [o*V, o*V + V) of the interleaved run
            vec_t w;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int k = o * V + e;  // position in the run: x offset k / R, plane k % R
                w[e] = v[k % R][k / R];
            }
            *reinterpret_cast<vec_t*>(dst + o * V) = w;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sr_pixel_shuffle_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int Co, int H, int W, int r) {
    const int Ho = H * r, Wo = W * r;
    const long total = (long)B * Co * Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo);
        long p = i / Wo;
        const int yo = (int)(p % Ho);
        p /= Ho;
        const int c = (int)(p % Co);
        const int b = (int)(p / Co);
        const int ci = c * r * r + (yo % r) * r + (xo % r);
        out[i] = in[((size_t)(b * Co * r * r + ci) * H + yo / r) * W + xo / r];
    }
}

// ----------------------------------------------------------------------------- channel attention gate
// grid = (pixel blocks, B).  Every workgroup recomputes the (tiny) squeeze MLP of its image from the
// per-tile channel sums the producing conv wrote, then streams y -> y*s*y_scale + skip (+ skip2).
template <typename T>
SR_DEV f32x4 ld4(const void* p, size_t off, int dtype) {
    return dtype == SR_BF16 ? load4(reinterpret_cast<const bf16*>(p) + off) : load4(reinterpret_cast<const float*>(p) + off);
}

__global__ __launch_bounds__(256) void sr_channel_gate_kernel(SrChannelAttn a, float* __restrict__ out) {
    extern __shared__ float sm[];
    const float* gate = ca_squeeze(a, blockIdx.x, sm);
    for (int c = threadIdx.x; c < a.C_p; c += 256) out[(size_t)blockIdx.x * a.C_p + c] = gate[c];
}

__global__ __launch_bounds__(256) void sr_channel_attn_kernel(SrChannelAttn a) {
    extern __shared__ float sm[];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const float* gate = ca_squeeze(a, b, sm);
    const int groups = a.C_p >> 2;
    const long per_img = (long)a.H * a.W * groups;
    for (long i = (long)blockIdx.x * 256 + tid; i < per_img; i += (long)gridDim.x * 256) {
        const int g = (int)(i % groups);
        const size_t off = (size_t)b * per_img * 4 + (size_t)i * 4;
        f32x4 v = ld4<float>(a.y, off, a.y_dtype);
        const f32x4 s = *reinterpret_cast<const f32x4*>(gate + g * 4);
        if (a.skip) {  // one fma per element (sr_rcab_conv_pair's gated input repeats exactly this)
            const f32x4 k = ld4<float>(a.skip, off, a.skip_dtype);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], s[r], k[r]);
        } else {
            v *= s;
        }
        if (a.skip2) v += ld4<float>(a.skip2, off, a.skip2_dtype);
        if (a.out_dtype == SR_BF16)
            store4(reinterpret_cast<bf16*>(a.out) + off, v);
        else
            store4(reinterpret_cast<float*>(a.out) + off, v);
    }
}

int vec_grid_for(long nvec) {  // streaming kernels: enough workgroups to fill the chip several times over, grid-stride for the rest
    long g = (nvec + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int sr_ingest_nchw(const float* x, void* out, int out_dtype, int B, int C, int H, int W, int Hp, int Wp, int Cp, int pad_mode, const float* scale,
                              const float* bias, void* stream) {
    SR_REQUIRE(x && out && scale && bias, "sr_ingest_nchw: null pointer");
    SR_REQUIRE(B > 0 && C > 0 && C <= Cp && Cp % 8 == 0 && Hp >= H && Wp >= W, "sr_ingest_nchw: bad geometry");
    if (pad_mode == SR_PAD_EVAL_MIRROR) SR_REQUIRE(Hp <= 2 * H && Wp <= 2 * W, "sr_ingest_nchw: mirror pad larger than the image");
    if (pad_mode == SR_PAD_REFLECT) SR_REQUIRE(Hp - H < H && Wp - W < W, "sr_ingest_nchw: reflect pad must be smaller than the image");
    if (pad_mode == SR_PAD_NONE) SR_REQUIRE(Hp == H && Wp == W, "sr_ingest_nchw: SR_PAD_NONE with Hp != H");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long total = (long)B * Hp * Wp * (Cp / 8);
    if (out_dtype == SR_BF16)
        hipLaunchKernelGGL(sr_ingest_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, st, x, reinterpret_cast<bf16*>(out), B, C, H, W, Hp, Wp, Cp, pad_mode,
                           scale, bias);
    else
        hipLaunchKernelGGL(sr_ingest_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, x, reinterpret_cast<float*>(out), B, C, H, W, Hp, Wp, Cp, pad_mode,
                           scale, bias);
    SR_CHECK_LAUNCH("sr_ingest_nchw");
    return SR_OK;
}

extern "C" int sr_u8_to_nchw(const unsigned char* in, float* out, int B, int C, int H, int W, float divisor, void* stream) {
    SR_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0 && divisor > 0.f, "sr_u8_to_nchw: bad arguments");
    const long total = (long)B * C * H * W;
    hipLaunchKernelGGL(sr_u8_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out, total, C, H * W, divisor);
    SR_CHECK_LAUNCH("sr_u8_to_nchw");
    return SR_OK;
}

extern "C" int sr_nchw_to_u8(const float* in, unsigned char* out, int B, int C, int H, int W, float mult, void* stream) {
    SR_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0, "sr_nchw_to_u8: bad arguments");
    const long total = (long)B * C * H * W;
    hipLaunchKernelGGL(sr_nchw_to_u8_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out, total, C, H * W, mult);
    SR_CHECK_LAUNCH("sr_nchw_to_u8");
    return SR_OK;
}

extern "C" int sr_layernorm_to(const float* x, void* y, int y_dtype, const float* gamma, const float* beta, int M, int C, int Cp, float eps, void* stream) {
    SR_REQUIRE(x && y && gamma && beta, "sr_layernorm: null pointer");
    SR_REQUIRE(M > 0 && C > 0 && C <= Cp && Cp % 8 == 0 && Cp <= 384, "sr_layernorm: bad geometry M=%d C=%d Cp=%d", M, C, Cp);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (y_dtype == SR_BF16)
        hipLaunchKernelGGL(sr_layernorm_kernel<bf16>, dim3((M + 31) / 32), dim3(256), 0, st, x, reinterpret_cast<bf16*>(y), gamma, beta, M, C, Cp, eps);
    else
        hipLaunchKernelGGL(sr_layernorm_kernel<float>, dim3((M + 31) / 32), dim3(256), 0, st, x, reinterpret_cast<float*>(y), gamma, beta, M, C, Cp, eps);
    SR_CHECK_LAUNCH("sr_layernorm");
    return SR_OK;
}

extern "C" int sr_layernorm(const float* x, float* y, const float* gamma, const float* beta, int M, int C, int Cp, float eps, void* stream) {
    return sr_layernorm_to(x, y, SR_F32, gamma, beta, M, C, Cp, eps, stream);
}

extern "C" int sr_pixel_shuffle_nchw(const void* in, void* out, int elem_size, int B, int C_out, int H, int W, int r, void* stream) {
    SR_REQUIRE(in && out && B > 0 && C_out > 0 && H > 0 && W > 0 && r >= 1, "sr_pixel_shuffle_nchw: bad arguments");
    SR_REQUIRE(elem_size == 2 || elem_size == 4, "sr_pixel_shuffle_nchw: elem_size %d", elem_size);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long total = (long)B * C_out * H * r * W * r;
    const int V = 16 / elem_size;
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (W % V == 0 && aligned && r >= 2 && r <= 4) {
        const long nvec = total / V;
        const dim3 grid(vec_grid_for(nvec));
#define SR_PS_VEC(T, R) hipLaunchKernelGGL((sr_pixel_shuffle_vec_kernel<T, R>), grid, dim3(256), 0, st, reinterpret_cast<const T*>(in), reinterpret_cast<T*>(out), B, C_out, H, W)
        if (elem_size == 2) {
            if (r == 2) SR_PS_VEC(uint16_t, 2); else if (r == 3) SR_PS_VEC(uint16_t, 3); else SR_PS_VEC(uint16_t, 4);
        } else {
            if (r == 2) SR_PS_VEC(uint32_t, 2); else if (r == 3) SR_PS_VEC(uint32_t, 3); else SR_PS_VEC(uint32_t, 4);
        }
#undef SR_PS_VEC
    } else if (elem_size == 2)
        hipLaunchKernelGGL(sr_pixel_shuffle_kernel<uint16_t>, dim3(grid_for(total)), dim3(256), 0, st, reinterpret_cast<const uint16_t*>(in),
                           reinterpret_cast<uint16_t*>(out), B, C_out, H, W, r);
    else
        hipLaunchKernelGGL(sr_pixel_shuffle_kernel<uint32_t>, dim3(grid_for(total)), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(in),
                           reinterpret_cast<uint32_t*>(out), B, C_out, H, W, r);
    SR_CHECK_LAUNCH("sr_pixel_shuffle_nchw");
    return SR_OK;
}

extern "C" int sr_channel_gate(const SrChannelAttn* p, float* gate, void* stream) {
    SR_REQUIRE(p && gate && p->pool_partial && p->w1 && p->b1 && p->w2 && p->b2, "sr_channel_gate: null pointer");
    const SrChannelAttn& a = *p;
    SR_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.C > 0 && a.C <= a.C_p && a.C_p % 4 == 0 && a.Cr > 0 && a.n_tiles > 0, "sr_channel_gate: bad geometry");
    const int lds = ca_scratch_floats(a.C_p, a.Cr) * (int)sizeof(float);
    hipLaunchKernelGGL(sr_channel_gate_kernel, dim3(a.B), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a, gate);
    SR_CHECK_LAUNCH("sr_channel_gate");
    return SR_OK;
}

extern "C" int sr_channel_attention(const SrChannelAttn* p, void* stream) {
    SR_REQUIRE(p && p->y && p->pool_partial && p->w1 && p->b1 && p->w2 && p->b2 && p->out, "sr_channel_attention: null pointer");
    const SrChannelAttn& a = *p;
    SR_REQUIRE(a.B > 0 && a.C > 0 && a.C <= a.C_p && a.C_p % 4 == 0 && a.Cr > 0 && a.n_tiles > 0, "sr_channel_attention: bad geometry");
    const int lds = ca_scratch_floats(a.C_p, a.Cr) * (int)sizeof(float);
    const long per_img = (long)a.H * a.W * (a.C_p / 4);
    int gx = (int)((per_img + 256 * 8 - 1) / (256 * 8));  // >= 8 vector groups per thread: the squeeze prologue is paid per workgroup
    if (gx > 512) gx = 512;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(sr_channel_attn_kernel, dim3(gx, a.B), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a);
    SR_CHECK_LAUNCH("sr_channel_attention");
    return SR_OK;
}
