// 3x3 convolution onto <= 16 output channels (the RGB tail convs: swinir.py:326 conv_last, edsr.py:44, rcan.py:70, hat.py:467)
// with the final un-normalise + crop + NCHW fp32 store.  These layers are HBM-bound (64 or 256 bf16 channels in, 3 fp32 planes
// out, ~2 FLOP per input byte), and the general kernel of sr_conv.hip was bound by something else: every wave re-streamed the
// whole 9 x Cin x 16 weight block through its CU's L1 for each 32 pixels it produced -- 3x the bytes of the input tile itself.
// Here:
//   * persistent workgroups walk the tile list (grid = a few workgroups per CU); each wave keeps ITS weights in registers for
//     the whole launch (18 fragments = 72 VGPRs), so the vector-memory pipe carries input pixels only;
//       Cin  64: waves split the row tiles of a TH x 16 pixel tile, every wave holds all 9 x 2 K-chunks;
//       Cin 256: waves split K -- wave w owns chunks t = w (mod 4) of the 9 x 8 -- and all row tiles; the four partial tiles are
//                summed through LDS in a fixed order (wave 0 + 1 + 2 + 3), each wave finishing the row tiles m = w (mod 4);
//   * the halo tile of tile i+1 is fetched into registers before the MFMAs of tile i and written to LDS after them: the
//     global-load latency of the next tile is off the critical path even with one workgroup per CU;
//   * halo staging, LDS image and MFMA operand order as in sr_conv.hip (8 pixels x 8 K-groups per wave instruction = full
//     128-byte lines, K-group-major image, 16 horizontally adjacent pixels per row tile).
#include "sr_common.h"
#include "sr_host.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr int NHW = 18;  // halo tile width

template <int KCS, int TH, bool KSPLIT>
struct NarrowGeo {
    static constexpr int HH = TH + 2;
    static constexpr int ROWS = ((HH * NHW + 31) / 32) * 32;  // whole staging steps of 32 pixels
    static constexpr int KG = KCS * 4;
    static constexpr int KCT = 9 * KCS;
    static constexpr int NWF = KSPLIT ? (KCT + 3) / 4 : KCT;  // weight fragments a wave keeps
    static constexpr int MT = KSPLIT ? TH : TH / 4;           // row tiles a wave accumulates
    static constexpr int STEPS = ROWS / 32, KI = KG / 8;
    static constexpr int LDS_A = KG * ROWS * 16;
    static constexpr int LDS_RED = KSPLIT ? 4 * TH * 64 * 16 : 0;
    static_assert(KSPLIT || TH % 4 == 0, "row tiles per wave");
};

template <int KCS, int TH, bool KSPLIT, int DEPTH>
__global__ __launch_bounds__(256, 2) void sr_conv3x3_narrow_kernel(SrConv3x3 c, int n_tiles) {
    using G = NarrowGeo<KCS, TH, KSPLIT>;
    constexpr int ROWS = G::ROWS, KCT = G::KCT, NWF = G::NWF, MT = G::MT, STEPS = G::STEPS, KI = G::KI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Frag<bf16>* As = reinterpret_cast<Frag<bf16>*>(smem);
    f32x4* red = reinterpret_cast<f32x4*>(smem + G::LDS_A);  // [wave][row tile][lane]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ar = lane & 15, ag = lane >> 4;
    const int r8 = lane & 7, kq = lane >> 3;
    const int tiles_x = (c.W + 15) >> 4, tiles_y = (c.H + TH - 1) / TH;

    // ---- this wave's weights, resident for the whole launch
    Frag<bf16> wf[NWF];
    {
        const Frag<bf16>* Bp = reinterpret_cast<const Frag<bf16>*>(c.Wp) + lane;
#pragma unroll
        for (int i = 0; i < NWF; ++i) {
            const int t = KSPLIT ? wave + 4 * i : i;
            wf[i] = Bp[(size_t)(t < KCT ? t : 0) * 64];
        }
    }
    const f32x4 bias_r = c.bias ? load4(c.bias + ag * 4) : (f32x4)(0.0f);
    f32x4 fs = (f32x4)(0.0f), fb = (f32x4)(0.0f);
    if (ag == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < c.fin_c) {
                fs[r] = c.fin_scale[r];
                fb[r] = c.fin_bias[r];
            }
    }

    const bf16* xin = reinterpret_cast<const bf16*>(c.x);
    Frag<bf16> pre[DEPTH][STEPS][KI];  // DEPTH halo tiles in flight per workgroup
    bool pvalid[DEPTH][STEPS];
    auto fetch = [&](auto buf, int tile) {  // halo tile -> registers (out-of-image pixels read a clamped address and are zeroed at the LDS write)
        constexpr int Q = decltype(buf)::value;
        int t = tile;
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y, b = t / tiles_y;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int p = s * 32 + wave * 8 + r8;
            const int py = p / NHW, px = p - py * NHW;
            const int gy = ty * TH - 1 + py, gx = tx * 16 - 1 + px;
            pvalid[Q][s] = p < G::HH * NHW && gy >= 0 && gy < c.H && gx >= 0 && gx < c.W;
            const bf16* src = xin + ((size_t)(b * c.H + (pvalid[Q][s] ? gy : 0)) * c.W + (pvalid[Q][s] ? gx : 0)) * (KCS * 32);
#pragma unroll
            for (int i = 0; i < KI; ++i) pre[Q][s][i] = *reinterpret_cast<const Frag<bf16>*>(src + (kq + 8 * i) * 8);
        }
    };
    auto commit = [&](auto buf) {
        constexpr int Q = decltype(buf)::value;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int p = s * 32 + wave * 8 + r8;
#pragma unroll
            for (int i = 0; i < KI; ++i) As[(kq + 8 * i) * ROWS + p] = frag_keep_if(pvalid[Q][s], pre[Q][s][i]);
        }
    };

    // XCD-aware walk: workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one), each XCD has its own L2, and
    // neighbouring tiles share halo rows / columns: XCD x walks its own contiguous eighth of the tile list, its workgroups side by side.
    int tile, tile_end, stride;
    if ((gridDim.x & 7) == 0) {
        const int per = (n_tiles + 7) >> 3, xcd = blockIdx.x & 7;
        stride = gridDim.x >> 3;
        tile = xcd * per + (blockIdx.x >> 3);
        tile_end = min(n_tiles, (xcd + 1) * per);
    } else {
        tile = blockIdx.x, tile_end = n_tiles, stride = gridDim.x;
    }
    // one tile: halo registers -> LDS, refill that register set with the tile DEPTH strides ahead, MFMAs, (reduction,) store
    auto do_tile = [&](auto buf, int tile) {
        commit(buf);
        __syncthreads();
        const int next = tile + DEPTH * stride;
        if (next < tile_end) fetch(buf, next);  // in flight during the MFMAs of DEPTH tiles

        int t = tile;
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y, b = t / tiles_y;
        const int x0 = tx * 16, y0 = ty * TH;

        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = (f32x4)(0.0f);
        const int row0 = KSPLIT ? 0 : wave * MT;
        const Frag<bf16>* abase = As + row0 * NHW + ar + ag * ROWS;
#pragma unroll
        for (int i = 0; i < NWF; ++i) {
            const int ch = KSPLIT ? wave + 4 * i : i;  // K-chunk index = tap * KCS + kc
            if (KSPLIT && ch >= KCT) break;
            const int tap = ch / KCS, kc = ch - tap * KCS;
            const Frag<bf16>* arow = abase + (tap / 3) * NHW + (tap % 3) + kc * 4 * ROWS;
#pragma unroll
            for (int m = 0; m < MT; ++m) mma(wf[i], arow[m * NHW], acc[m]);
        }

        if constexpr (KSPLIT) {
            // partial tiles -> LDS, then wave w finishes the row tiles m = w (mod 4): fixed order 0 + 1 + 2 + 3
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if ((m & 3) != wave) red[(wave * TH + m) * 64 + lane] = acc[m];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if ((m & 3) == wave) {
                    f32x4 s = wave == 0 ? acc[m] : red[(0 * TH + m) * 64 + lane];
#pragma unroll
                    for (int w = 1; w < 4; ++w) s += (w == wave) ? acc[m] : red[(w * TH + m) * 64 + lane];
                    acc[m] = s;
                }
        }

        // ---- epilogue: lanes 0..15 hold channels 0..3 of pixel x0 + lane
        if (ag == 0) {
            float* o = reinterpret_cast<float*>(c.out);
            const int x = x0 + ar;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (KSPLIT && (m & 3) != wave) continue;
                const int y = y0 + row0 + m;
                if (y < c.fin_h && y < c.H && x < c.fin_w && x < c.W) {
                    const f32x4 v = acc[m] + bias_r;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < c.fin_c) o[((size_t)(b * c.fin_c + r) * c.fin_h + y) * c.fin_w + x] = v[r] * fs[r] + fb[r];
                }
            }
        }
        __syncthreads();  // the halo image (and the reduction buffer) are rewritten by the next tile
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, DEPTH - 1>;
    if (tile < tile_end) fetch(I0{}, tile);
    if (DEPTH > 1 && tile + stride < tile_end) fetch(I1{}, tile + stride);
    for (; tile < tile_end; tile += DEPTH * stride) {
        do_tile(I0{}, tile);
        if (DEPTH > 1 && tile + stride < tile_end) do_tile(I1{}, tile + stride);
    }
}

template <int KCS, int TH, bool KSPLIT, int DEPTH>
int launch_narrow(const SrConv3x3& c, hipStream_t st, int wgs_per_cu) {
    using G = NarrowGeo<KCS, TH, KSPLIT>;
    constexpr int lds = G::LDS_A + G::LDS_RED;
    static SrDeviceOnce attr_once;
    {
        const hipError_t e = sr_once_per_device(attr_once, [&] { return sr_allow_lds(sr_conv3x3_narrow_kernel<KCS, TH, KSPLIT, DEPTH>, lds); });
        SR_REQUIRE(e == hipSuccess, "sr_conv3x3 (narrow): hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const int n_tiles = ((c.W + 15) / 16) * ((c.H + TH - 1) / TH) * c.B;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = n_tiles < cus * wgs_per_cu ? n_tiles : cus * wgs_per_cu;
    if (grid >= 8) grid &= ~7;  // whole rounds of the 8 XCDs
    hipLaunchKernelGGL((sr_conv3x3_narrow_kernel<KCS, TH, KSPLIT, DEPTH>), dim3(grid), dim3(256), lds, st, c, n_tiles);
    SR_CHECK_LAUNCH("sr_conv3x3 (narrow)");
    return SR_OK;
}

}  // namespace

// true if sr_conv3x3_narrow covers this conv: bf16 NHWC input of 64 or 256 channels, <= 4 real output channels, plain final NCHW store
bool sr_conv3x3_narrow_supported(const SrConv3x3& c) {
    const bool off = false;  // (true: the general kernel; the A/B switch left in round 5)
    if (off) return false;
    return c.compute_dtype == SR_BF16 && c.x_dtype == SR_BF16 && c.Cout_p == 16 && (c.Cin_p == 64 || c.Cin_p == 256) && c.out_mode == SR_OUT_FINAL_NCHW &&
           c.ps_r <= 1 && c.fin_c <= 4 && c.act == SR_ACT_NONE && c.out_scale == 1.0f && !c.skip && !c.pool_partial;
}

int sr_conv3x3_narrow(const SrConv3x3& c, hipStream_t st) {
    // Measured on MI355X (tools/kbench.py conv, "final" rows; input bytes / time):
    //   64 ch, 8 x 288 x 288:  TH 8, 3 workgroups / CU, prefetch depth 1: 24.0 us (3.5 TB/s; the general kernel: 36.7 us); depth 2 (2 / CU): 28.2; TH 16: 34.2
    //   256 ch, 16 x 256 x 256: TH 4, 2 workgroups / CU: 168 us (3.2 TB/s; the general kernel: 281 us); TH 8, 1 / CU: 257; TH 4 depth 2 (spills): 261
    // i.e. workgroups per CU beat bytes in flight per workgroup; what remains is the LDS read stream (one 1-KiB fragment per MFMA: N = 16).
    if (c.Cin_p == 64) return launch_narrow<2, 8, false, 1>(c, st, 3);
    return launch_narrow<8, 4, true, 1>(c, st, 2);
}
