// 3x3 convolution (stride 1, zero pad 1) as an im2col-free implicit GEMM, NHWC.
//
// One workgroup (4 waves) produces a TH x 16 pixel tile of N_T = WN*NW*16 output channels:
//   * the (TH+2) x 18 input halo tile, ALL input channels, is staged once into LDS in the
//     K-group-major image of sr_common.h (zero padding at the image border is written here);
//   * an MFMA row tile is 16 horizontally adjacent output pixels, so its operand for tap
//     (ky, kx) is 16 CONSECUTIVE rows of that image starting at (y+ky)*18 + kx: the nine
//     shifted views are read from the one tile, conflict-free, with no im2col buffer;
//   * weights are streamed per (tap, 32-channel chunk) straight from L2 into registers in
//     fragment order (double buffered), never through LDS; waves split N (and M for narrow N);
//   * no barrier in the 9*Cin/32-step main loop;
//   * epilogue: bias, ReLU/LeakyReLU/GELU, res_scale, residual add and one of three stores:
//     NHWC, NHWC through nn.PixelShuffle (the weight rows were permuted at pack time so that a
//     lane's 4 consecutive accumulators are 4 consecutive channels of ONE shuffled pixel), or the
//     final un-normalise + crop + NCHW fp32 image.  Optionally emits per-tile channel sums for
//     channel attention (deterministic, no atomics).
//
// Entry point and host-side tile selection; the kernel variants live in sr_conv_impl.h / sr_conv_v_*.hip.
#include "sr_common.h"
#include "sr_host.h"

#include <cstdlib>

int sr_conv_dispatch_bf16_f32_4(const SrConv3x3& c, hipStream_t st);
int sr_conv_dispatch_bf16_bf16_4(const SrConv3x3& c, hipStream_t st);
int sr_conv_dispatch_bf16_f32_8(const SrConv3x3& c, hipStream_t st);
int sr_conv_dispatch_bf16_bf16_8(const SrConv3x3& c, hipStream_t st);
int sr_conv_dispatch_bf3_f32_4(const SrConv3x3& c, hipStream_t st);
int sr_conv_dispatch_f32_f32_4(const SrConv3x3& c, hipStream_t st);

namespace {

// bf16 tile height: 8 rows; 4 rows when 8-row tiles leave the chip under-filled (small batches / single images: every launch is one
// latency chain per workgroup, so twice the workgroups of half the work each finish sooner).  Convs with a pool side output keep 8 unless
// the caller asks for 4 (tile_rows) -- the number of pool slots follows the tile height (sr_conv3x3_pool_tiles_rows).
int conv_tile_rows(const SrConv3x3& c) {
    if (c.tile_rows == 4 || c.tile_rows == 8) return c.tile_rows;
    if (c.pool_partial) return 8;
    const int n = c.Cout_p;
    const int ntile = (n % 256 == 0 && c.Cin_p >= 128) ? 256 : n % 192 == 0 ? 192 : n % 128 == 0 ? 128 : n % 64 == 0 ? 64 : n % 32 == 0 ? 32 : 16;
    const long long wgs8 = (long long)((c.W + 15) / 16) * ((c.H + 7) / 8) * c.B * (n / ntile);
    static const int below = getenv("SR_CONV_TH4_BELOW") ? atoi(getenv("SR_CONV_TH4_BELOW")) : 256;  // A/B knob (tools/kbench.py conv)
    return wgs8 < below ? 4 : 8;
}

int conv_wm(int cout_p) {
    if (cout_p % 192 == 0 || cout_p % 128 == 0) return 1;
    if (cout_p % 32 == 0) return 2;
    return 4;
}

}  // namespace

extern "C" int sr_conv3x3_pool_tiles_rows(int H, int W, int Cout_p, int tile_rows) {
    return ((W + 15) / 16) * ((H + tile_rows - 1) / tile_rows) * conv_wm(Cout_p);
}

extern "C" int sr_conv3x3_pool_tiles(int H, int W, int Cout_p, int compute_dtype) {
    // number of per-image partial-sum slots sr_conv3x3 writes into pool_partial for this geometry
    const int th = compute_dtype == SR_BF16 ? 8 : 4;
    return ((W + 15) / 16) * ((H + th - 1) / th) * conv_wm(Cout_p);
}

extern "C" int sr_conv3x3(const SrConv3x3* p, void* stream) {
    SR_REQUIRE(p && p->x && p->Wp && p->out, "sr_conv3x3: null pointer");
    const SrConv3x3& c = *p;
    SR_REQUIRE(c.B > 0 && c.H > 0 && c.W > 0 && c.Cin_p % 32 == 0 && c.Cout_p % 16 == 0, "sr_conv3x3: bad geometry B=%d H=%d W=%d Cin_p=%d Cout_p=%d", c.B,
               c.H, c.W, c.Cin_p, c.Cout_p);
    if (c.out_mode == SR_OUT_PIXEL_SHUFFLE || (c.out_mode == SR_OUT_FINAL_NCHW && c.ps_r > 1))
        SR_REQUIRE(c.ps_r >= 2 && c.cps_p % 4 == 0 && c.ps_r * c.ps_r * c.cps_p == c.Cout_p, "sr_conv3x3: pixel shuffle r=%d cps_p=%d Cout_p=%d", c.ps_r,
                   c.cps_p, c.Cout_p);
    if (c.out_mode == SR_OUT_FINAL_NCHW)
        SR_REQUIRE(c.fin_scale && c.fin_bias && c.fin_c > 0 && c.fin_c <= c.Cout_p && c.fin_h <= c.H * (c.ps_r > 1 ? c.ps_r : 1) &&
                       c.fin_w <= c.W * (c.ps_r > 1 ? c.ps_r : 1) && !c.skip,
                   "sr_conv3x3: bad FINAL_NCHW arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (sr_conv3x3_narrow_supported(c)) return sr_conv3x3_narrow(c, st);
    if (sr_conv3x3_big_supported(c)) return sr_conv3x3_big(c, st);
    if (c.compute_dtype == SR_BF16) {
        if (conv_tile_rows(c) == 4) {
            if (c.x_dtype == SR_F32) return sr_conv_dispatch_bf16_f32_4(c, st);
            return sr_conv_dispatch_bf16_bf16_4(c, st);
        }
        if (c.x_dtype == SR_F32) return sr_conv_dispatch_bf16_f32_8(c, st);
        return sr_conv_dispatch_bf16_bf16_8(c, st);
    }
    SR_REQUIRE(c.x_dtype == SR_F32, "sr_conv3x3: fp32 / bf16x3 compute needs fp32 input");
    if (c.compute_dtype == SR_BF16X3) return sr_conv_dispatch_bf3_f32_4(c, st);
    return sr_conv_dispatch_f32_f32_4(c, st);
}
