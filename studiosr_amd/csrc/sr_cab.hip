// HAT's channel-attention block body in ONE launch (hat.py:41-49 CAB: conv 180 -> 60, GELU, conv 60 -> 180; the ChannelAttention squeeze
// reads the per-tile channel sums written here):  y = conv2(GELU(conv1(x))),  x = LayerNorm1 output, bf16 NHWC, 192 / 64 / 192 padded channels.
//
// Why: at HAT's shapes (4 x 64 x 64 pixels) each of the two convolutions is ~2 us of MFMAs inside a 17-21 us launch (halo load, weight
// stream start, store), and the pair is the longer of the two branches of a HAB (profiles/r03_hat_timeline.txt).  Same structure as
// sr_rcab.hip (the 64-channel RCAB body), re-tiled for 192 input channels:
//   * one workgroup (4 waves) owns a 14 x 6 output tile: it stages the 18 x 10 input halo (K-group-major bf16 image, 69 KiB), computes the
//     16 x 8 intermediate tile (conv1 + bias + GELU; positions outside the image are ZERO: they are conv2's padding) into a second LDS
//     image, then 16 x 6 outputs from it (the inner 14 columns are kept);
//   * an MFMA row tile is 16 horizontally adjacent pixels in both convs; conv1: waves split 2 x 2 over (rows, channels), conv2: each wave
//     owns 48 of the 192 output channels for all rows (its channel sums are complete: one pool slot per tile, no atomics);
//   * weights stream L2 -> registers through rings (5 / 3 steps ahead); K is walked tap-major like sr_conv3x3 (same packed weights).
#include "sr_cab_body.h"

// The split-operand form (round 5, C ABI v11: SrCab.dtype = SR_BF16X3, precision "fp32x3" = what inference() runs): the same body with 32-byte image cells.  The full-K halo
// image would be 177 KB: the two-phase K walk (96 of the 192 input channels resident at a time) brings the two images to 103 KB -- one workgroup per CU.
#undef SR_CAB_NS_BEGIN
#undef SR_CAB_NS_END
#undef SR_CAB_PH
#define SR_CAB_NS_BEGIN namespace cabx3 {
#define SR_CAB_NS_END }
#define SR_CAB_PH 2
#include "sr_cab_body.h"

namespace {

__global__ __launch_bounds__(256, 1) void sr_cab_x3_kernel(SrCab c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cabx3::cab_block<bf3, float, float>(c, blockIdx.x, smem);
}
static_assert(cabx3::LDS_BYTES_X3 <= 160 * 1024, "split-operand CAB: LDS");

#undef SR_CAB_PH
#define SR_CAB_PH 1
__global__ __launch_bounds__(256, 1) void sr_cab_kernel(SrCab c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cab_block(c, blockIdx.x, smem);
}

}  // namespace

extern "C" int sr_cab_pool_tiles(int H, int W) { return ((W + TOW - 1) / TOW) * ((H + TOH - 1) / TOH); }

extern "C" int sr_cab_supported(int Cin_p, int Cmid_p, int Cout_p, int dtype) { return (Cin_p == CI && Cmid_p == CM && Cout_p == CO && (dtype == SR_BF16 || dtype == SR_BF16X3)) ? 1 : 0; }

extern "C" int sr_cab_fused(const SrCab* p, void* stream) {
    if (const int rc = cab_check(p, "sr_cab_fused")) return rc;
    const SrCab& c = *p;
    SR_REQUIRE(c.tile_rows == 0 || c.tile_rows == TOH, "sr_cab_fused: %d-row tiles only (SrCab.tile_rows = 8 exists in sr_hab_mid)", TOH);
    if (c.dtype == SR_BF16X3) {  // x, y fp32; weights packed hi | lo
        SR_REQUIRE(!c.mid_pre, "sr_cab_fused: mid_pre (the training forward's side output) exists in the bf16 form only");
        static SrDeviceOnce once_x3;
        const hipError_t e = sr_once_per_device(once_x3, [&] { return sr_allow_lds(sr_cab_x3_kernel, cabx3::LDS_BYTES_X3); });
        SR_REQUIRE(e == hipSuccess, "sr_cab_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
        const long tiles_x3 = (long)sr_cab_pool_tiles(c.H, c.W) * c.B;
        hipLaunchKernelGGL(sr_cab_x3_kernel, dim3((unsigned)tiles_x3), dim3(256), cabx3::LDS_BYTES_X3, reinterpret_cast<hipStream_t>(stream), c);
        SR_CHECK_LAUNCH("sr_cab_fused (split operands)");
        return SR_OK;
    }
    static SrDeviceOnce attr_once;
    {
        const hipError_t e = sr_once_per_device(attr_once, [&] { return sr_allow_lds(sr_cab_kernel, LDS_BYTES); });
        SR_REQUIRE(e == hipSuccess, "sr_cab_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const long tiles = (long)sr_cab_pool_tiles(c.H, c.W) * c.B;
    hipLaunchKernelGGL(sr_cab_kernel, dim3((unsigned)tiles), dim3(256), LDS_BYTES, reinterpret_cast<hipStream_t>(stream), c);
    SR_CHECK_LAUNCH("sr_cab_fused");
    return SR_OK;
}
