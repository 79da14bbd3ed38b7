// One-time weight layout transforms on the device (`sr_weight_pack_*`): a non-Python host can feed reference
// checkpoints (state_dict layouts of studiosr/models/*.py) to the kernels of this library without re-deriving the MFMA fragment order.
// Pure index shuffling + one optional multiply per element (attention scale on the q rows, swinir.py:83; LayerNorm gamma on the
// columns when the affine is folded into the following Linear) -- bit-identical to studiosr_amd/packing.py (GPU test).
// Fragment order (include/studiosr_hip.h): Wp[n_tile][k_chunk][lane][8], element (n = 16 n_tile + (lane & 15), k = 32 k_chunk + 8 (lane >> 4) + j).
#include "sr_common.h"
#include "sr_host.h"

namespace {

template <typename T>
SR_DEV void put(void* out, long long i, float v) { reinterpret_cast<T*>(out)[i] = (T)v; }

// one thread per packed element
template <typename T>
__global__ void sr_pack_matrix_kernel(const float* __restrict__ w, long long ld, const int* __restrict__ row_idx, const int* __restrict__ col_idx, const float* __restrict__ row_scale,
                                      const float* __restrict__ col_scale, void* __restrict__ out, int N_p, int K_p, int n_rows, int n_cols) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N_p * K_p) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long long frag = i >> 9;  // n_tile * (K_p / 32) + k_chunk
    const int kchunks = K_p / 32;
    const int kc = (int)(frag % kchunks), nt = (int)(frag / kchunks);
    const int n = 16 * nt + (lane & 15), k = 32 * kc + 8 * (lane >> 4) + j;
    const int r = row_idx ? row_idx[n] : (n < n_rows ? n : -1);
    const int c = col_idx ? col_idx[k] : (k < n_cols ? k : -1);
    float v = 0.f;
    if (r >= 0 && c >= 0) {
        v = w[(long long)r * ld + c];
        if (col_scale) v *= col_scale[c];
        if (row_scale) v *= row_scale[n];
    }
    put<T>(out, i, v);
}

// nn.Conv2d weight [Cout, Cin, 3, 3] -> implicit-GEMM matrix rows row_idx[n], k = (ky*3 + kx) * cin_p + c
template <typename T>
__global__ void sr_pack_conv3x3_kernel(const float* __restrict__ w, const int* __restrict__ row_idx, void* __restrict__ out, int N_p, int Cout, int Cin, int cin_p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int K_p = 9 * cin_p;
    if (i >= (long long)N_p * K_p) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long long frag = i >> 9;
    const int kchunks = K_p / 32;
    const int kc = (int)(frag % kchunks), nt = (int)(frag / kchunks);
    const int n = 16 * nt + (lane & 15), k = 32 * kc + 8 * (lane >> 4) + j;
    const int r = row_idx ? row_idx[n] : (n < Cout ? n : -1);
    const int tap = k / cin_p, c = k - tap * cin_p;
    const float v = (r >= 0 && c < Cin) ? w[((long long)r * Cin + c) * 9 + tap] : 0.f;
    put<T>(out, i, v);
}

__global__ void sr_pack_vector_kernel(const float* __restrict__ b, const int* __restrict__ idx, const float* __restrict__ scale, float* __restrict__ out, int n_p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_p) return;
    const int r = idx ? idx[i] : (i < n ? i : -1);
    float v = (b && r >= 0) ? b[r] : 0.f;
    if (scale) v *= scale[i];
    out[i] = v;
}

// relative-position bias in accumulator-fragment order [h][qt][kt][lane][4] = table[rpi[q, k] (python-style wrap)][h],
// q = 16 qt + (lane & 15), k = 16 kt + 4 (lane >> 4) + r   (swinir.py:86-91, hat.py:93-96,276-279)
__global__ void sr_pack_bias_fragments_kernel(const float* __restrict__ table, const long long* __restrict__ rpi, float* __restrict__ out, int T, int heads, int Nq, int Nk) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)heads * Nq * Nk) return;
    const int r = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const long long tile = i >> 8;
    const int nkt = Nk / 16, nqt = Nq / 16;
    const int kt = (int)(tile % nkt), qt = (int)((tile / nkt) % nqt), h = (int)(tile / ((long long)nkt * nqt));
    const int q = 16 * qt + (lane & 15), k = 16 * kt + 4 * (lane >> 4) + r;
    long long t = rpi[(long long)q * Nk + k];
    if (t < 0) t += T;
    out[i] = table[t * heads + h];
}

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int sr_pack_matrix(const float* w, long long ld, const int* row_idx, const int* col_idx, const float* row_scale, const float* col_scale, void* out, int out_dtype,
                              int N_p, int K_p, int n_rows, int n_cols, void* stream) {
    SR_REQUIRE(w && out && N_p > 0 && K_p > 0 && N_p % 16 == 0 && K_p % 32 == 0 && ld > 0, "sr_pack_matrix: N_p %% 16 == 0 and K_p %% 32 == 0 required (N_p=%d K_p=%d)", N_p, K_p);
    SR_REQUIRE(out_dtype == SR_F32 || out_dtype == SR_BF16, "sr_pack_matrix: bad dtype");
    const long long n = (long long)N_p * K_p;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (out_dtype == SR_BF16)
        hipLaunchKernelGGL(sr_pack_matrix_kernel<bf16>, grid, dim3(256), 0, ST, w, ld, row_idx, col_idx, row_scale, col_scale, out, N_p, K_p, n_rows, n_cols);
    else
        hipLaunchKernelGGL(sr_pack_matrix_kernel<float>, grid, dim3(256), 0, ST, w, ld, row_idx, col_idx, row_scale, col_scale, out, N_p, K_p, n_rows, n_cols);
    SR_CHECK_LAUNCH("sr_pack_matrix");
    return SR_OK;
}

extern "C" int sr_pack_conv3x3(const float* w, const int* row_idx, void* out, int out_dtype, int N_p, int Cout, int Cin, int cin_p, void* stream) {
    SR_REQUIRE(w && out && N_p > 0 && N_p % 16 == 0 && cin_p > 0 && (9 * cin_p) % 32 == 0 && Cin <= cin_p && Cout > 0, "sr_pack_conv3x3: bad sizes N_p=%d cin_p=%d", N_p, cin_p);
    SR_REQUIRE(out_dtype == SR_F32 || out_dtype == SR_BF16, "sr_pack_conv3x3: bad dtype");
    const long long n = (long long)N_p * 9 * cin_p;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (out_dtype == SR_BF16)
        hipLaunchKernelGGL(sr_pack_conv3x3_kernel<bf16>, grid, dim3(256), 0, ST, w, row_idx, out, N_p, Cout, Cin, cin_p);
    else
        hipLaunchKernelGGL(sr_pack_conv3x3_kernel<float>, grid, dim3(256), 0, ST, w, row_idx, out, N_p, Cout, Cin, cin_p);
    SR_CHECK_LAUNCH("sr_pack_conv3x3");
    return SR_OK;
}

extern "C" int sr_pack_vector(const float* b, const int* idx, const float* scale, float* out, int n_p, int n, void* stream) {
    SR_REQUIRE(out && n_p > 0, "sr_pack_vector: bad arguments");
    hipLaunchKernelGGL(sr_pack_vector_kernel, dim3((n_p + 255) / 256), dim3(256), 0, ST, b, idx, scale, out, n_p, n);
    SR_CHECK_LAUNCH("sr_pack_vector");
    return SR_OK;
}

extern "C" int sr_pack_bias_fragments(const float* table, const long long* rpi, float* out, int T, int heads, int Nq, int Nk, void* stream) {
    SR_REQUIRE(table && rpi && out && T > 0 && heads > 0 && Nq > 0 && Nk > 0 && Nq % 16 == 0 && Nk % 16 == 0, "sr_pack_bias_fragments: Nq, Nk must be multiples of 16");
    const long long n = (long long)heads * Nq * Nk;
    hipLaunchKernelGGL(sr_pack_bias_fragments_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST, table, rpi, out, T, heads, Nq, Nk);
    SR_CHECK_LAUNCH("sr_pack_bias_fragments");
    return SR_OK;
}
