// The two independent middle stages of a HAB as ONE launch (C ABI v8 sr_hab_mid; hat.py:165-176):
//     y = conv2(GELU(conv1(LayerNorm1(x))))   -- the CAB body, sr_cab.hip                          (hat.py:41-49, 165-170)
//     o = softmax(q k^T + bias + mask) v      -- (S)W-MSA of the 16 x 16 windows, sr_attn.hip      (hat.py:85-110, 172-176)
// Why: both read what the previous block's sr_swin_tail wrote and feed the next sr_swin_tail; as two launches they ran on two HIP streams,
// and in the captured graph every fork / join edge between the two queues costs 5-12 us (profiles/r03_hat_timeline.txt: a HAB period of
// 75.7 us = tail 28.6 + fork 11.7 + attention 24.2 + join 11.2) -- a third of HAT x4 b4.  Here the first n_cab workgroups are CAB tiles (the
// longer latency chain: dispatched first), the rest are attention workgroups (4 waves = 4 (window, head, 32 queries) items, no LDS, no
// barriers); the roles share nothing but the launch, so the chain tail -> mid -> tail stays on ONE queue with back-to-back dispatch.
// The CAB role is the two-K-phase form of sr_cab.hip (52 KiB of LDS: the launch's dynamic LDS applies to attention workgroups too, and two
// workgroups per CU -- the attention role's 221 VGPRs -- must fit); its conv1 therefore sums K phase-major (same products, another fp32 order).
// With SrWindowAttn.bias_tiles the attention role is the LDS form (sr_wattn_lds_body.h: one (window, head) per workgroup, K / V^T / the 31 distinct bias
// tiles staged once; 63 KiB of LDS).
// Two tile heights of the CAB role (SrCab.tile_rows): 14 x 6 outputs per workgroup (small launches: more, shorter latency chains) and 14 x 8 (from 4 x 64 x 64 pixels
// on: a quarter fewer workgroups, 18 % less halo recomputation -- HAT x4 b16 6.52 -> 6.28 ms, 64 tiles 27.7 -> 26.7; a single tile 1.64 -> 1.74 the other way)
#define SR_CAB_PH 2
#define SR_CAB_NS_BEGIN namespace { namespace cab6 {
#define SR_CAB_NS_END } }
#define SR_CAB_TOH 6
#include "sr_cab_body.h"
#undef SR_CAB_NS_BEGIN
#undef SR_CAB_TOH
#define SR_CAB_NS_BEGIN namespace { namespace cab8 {
#define SR_CAB_TOH 8
#include "sr_cab_body.h"
#include "sr_wattn_lds_body.h"
#include "sr_wattn_qkv_body.h"

#ifndef SR_MID_XCD
#define SR_MID_XCD 1  // HAT x4 b4 2.415 -> 2.39 ms, b16 6.69 -> 6.64 (0: CAB tiles in block-id order)
#endif
namespace {

// MODE bit 4: the CAB role's tiles are 8 rows high (cab8) instead of 6
// MODE bit 0: q / k / v^T in fragment order (SrWindowAttn.qkv_frag); bit 1: the LDS form of the attention (SrWindowAttn.bias_tiles: one (window, head) per
// workgroup, sr_wattn_lds_body.h) instead of the register-only flash form (four (window, head, 32 queries) items per workgroup)
template <int MODE>
__global__ __launch_bounds__(256, 2) void sr_hab_mid_kernel(SrWindowAttn a, SrCab c, int n_cab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int block = __builtin_amdgcn_readfirstlane(blockIdx.x);
    if (block < n_cab) {
        int tile = block;
#if SR_MID_XCD
        const int q = n_cab >> 3, r = n_cab & 7, xcd = block & 7;  // neighbouring CAB tiles (18 x 10-pixel halos of 14 x 6 tiles) on one XCD
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (block >> 3);
#endif
        if constexpr ((MODE & 16) != 0)
            cab8::cab_block(c, tile, smem);
        else
            cab6::cab_block(c, tile, smem);
    }
    else if constexpr ((MODE & 4) != 0)
        wattn_qkv_block(a, block - n_cab, smem);  // LayerNorm1 + the head's QKV projection + attention (SrWindowAttn.x)
    else if constexpr ((MODE & 2) != 0)
        wattn_lds_block<(MODE & 1) != 0>(a, block - n_cab, smem);
    else
        wattn_flash_block<bf16, 16, 2, 1, (MODE & 1) != 0>(a, block - n_cab);
}

}  // namespace

extern "C" int sr_hab_mid_supported(int ntok, int hd_p, int ws, int attn_dtype, int Cin_p, int Cmid_p, int Cout_p, int cab_dtype) {
    return (ntok == 256 && hd_p == 32 && ws == 16 && attn_dtype == SR_BF16 && Cin_p == cab6::CI && Cmid_p == cab6::CM && Cout_p == cab6::CO && cab_dtype == SR_BF16) ? 1 : 0;
}

extern "C" int sr_cab_pool_tiles_rows(int H, int W, int tile_rows) {  // pool slots per image of the CAB role for SrCab.tile_rows (0 / 6 / 8)
    const int th = tile_rows == 8 ? cab8::TOH : cab6::TOH;
    return ((W + cab6::TOW - 1) / cab6::TOW) * ((H + th - 1) / th);
}

extern "C" int sr_hab_mid(const SrWindowAttn* pa, const SrCab* pc, void* stream) {
    SR_REQUIRE(pa && ((pa->q && pa->k && pa->vt) || pa->x) && pa->bias_frag && pa->out, "sr_hab_mid: null pointer (attention operands; bias_frag is required)");
    SR_REQUIRE(!pa->x || (pa->wqkv && pa->bias && pa->bias_tiles && pa->heads == 6 && pa->C == 180 && pa->ldx >= 192 && pa->ldx % 4 == 0 &&
                          ((reinterpret_cast<uintptr_t>(pa->x) | reinterpret_cast<uintptr_t>(pa->wqkv)) & 15) == 0),
               "sr_hab_mid: the fused QKV form needs x, wqkv, bias, bias_tiles, 6 heads, C = 180 in >= 192 padded channels, 16-byte aligned rows");
    SR_REQUIRE(pc && (pc->tile_rows == 0 || pc->tile_rows == 6 || pc->tile_rows == 8), "sr_hab_mid: SrCab.tile_rows is 0 (= 6), 6 or 8");
    const bool t8 = pc->tile_rows == 8;
    if (const int rc = t8 ? cab8::cab_check(pc, "sr_hab_mid") : cab6::cab_check(pc, "sr_hab_mid")) return rc;
    const SrWindowAttn& a = *pa;
    const SrCab& c = *pc;
    SR_REQUIRE(sr_hab_mid_supported(a.ntok, a.hd_p, a.ws, a.dtype, c.Cin_p, c.Cmid_p, c.Cout_p, c.dtype), "sr_hab_mid: unsupported geometry (16 x 16 windows, head_dim <= 32, bf16)");
    SR_REQUIRE(a.H % a.ws == 0 && a.W % a.ws == 0 && a.shift >= 0 && a.shift < a.ws && a.heads > 0, "sr_hab_mid: bad attention geometry");
    SR_REQUIRE(a.n_bwin > 0 && a.n_bwin % ((a.H / a.ws) * (a.W / a.ws)) == 0, "sr_hab_mid: n_bwin");
    SR_REQUIRE(!a.bias_tiles || a.x || ((reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.k) | reinterpret_cast<uintptr_t>(a.vt) | reinterpret_cast<uintptr_t>(a.bias_tiles)) & 15) == 0,
               "sr_hab_mid: the LDS form of the attention stages 16-byte pieces (q, k, vt, bias_tiles must be 16-byte aligned)");
    const long n_cab = (long)(((c.W + cab6::TOW - 1) / cab6::TOW) * ((c.H + (t8 ? cab8::TOH : cab6::TOH) - 1) / (t8 ? cab8::TOH : cab6::TOH))) * c.B;
    const bool lds_form = a.bias_tiles != nullptr;
    SR_REQUIRE(!t8 || (lds_form && !a.x), "sr_hab_mid: 8-row CAB tiles exist beside the LDS form of the attention only (bias_tiles set, no x)");
    const long items = (long)a.n_bwin * a.heads * 8;  // flash form: (window, head, block of 32 queries), four per workgroup
    const long blocks = n_cab + (lds_form ? (long)a.n_bwin * a.heads : (items + 3) / 4);
    SR_REQUIRE(blocks < (1l << 31), "sr_hab_mid: too many workgroups");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static SrDeviceOnce once[7];
    auto launch = [&](auto kernel, SrDeviceOnce& o, int lds) -> int {
        const hipError_t e = sr_once_per_device(o, [&] { return sr_allow_lds(kernel, lds); });
        SR_REQUIRE(e == hipSuccess, "sr_hab_mid: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), lds, st, a, c, (int)n_cab);
        SR_CHECK_LAUNCH("sr_hab_mid");
        return SR_OK;
    };
    constexpr int LDS6 = cab6::LDS_BYTES, LDS8 = cab8::LDS_BYTES;
    constexpr int LDS_BOTH = LDS6 > WL_LDS ? LDS6 : WL_LDS;
    constexpr int LDS_BOTH8 = LDS8 > WL_LDS ? LDS8 : WL_LDS;
    constexpr int LDS_QKV = LDS6 > WQ_LDS ? LDS6 : WQ_LDS;
    static_assert(2 * LDS_BOTH <= 160 * 1024 && 2 * LDS_BOTH8 <= 160 * 1024 && 2 * LDS_QKV <= 160 * 1024, "two workgroups per CU");
    if (a.x) return launch(sr_hab_mid_kernel<4>, once[4], LDS_QKV);
    if (t8) return a.qkv_frag ? launch(sr_hab_mid_kernel<19>, once[6], LDS_BOTH8) : launch(sr_hab_mid_kernel<18>, once[5], LDS_BOTH8);
    if (lds_form) return a.qkv_frag ? launch(sr_hab_mid_kernel<3>, once[3], LDS_BOTH) : launch(sr_hab_mid_kernel<2>, once[2], LDS_BOTH);
    return a.qkv_frag ? launch(sr_hab_mid_kernel<1>, once[1], LDS6) : launch(sr_hab_mid_kernel<0>, once[0], LDS6);
}
