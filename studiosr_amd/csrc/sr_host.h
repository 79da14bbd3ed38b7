// Host-side helpers shared by the C-ABI launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../include/studiosr_hip.h"

void sr_set_error(const char* fmt, ...);

// sr_gemm2.hip: specialised hot-shape row GEMM; returns 1 if the shape is not covered.
int sr_gemm_v2_try(const SrGemm& g, hipStream_t st);

#define SR_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            sr_set_error(__VA_ARGS__);        \
            return SR_EINVAL;                 \
        }                                     \
    } while (0)

#define SR_CHECK_LAUNCH(name)                                                     \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            sr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return SR_ELAUNCH;                                                    \
        }                                                                         \
    } while (0)

// Opt a kernel into more than 64 KiB of dynamic LDS (once per instantiation).
template <typename K>
static inline hipError_t sr_allow_lds(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// hipFuncSetAttribute is per device: run `f` once on every device a launcher is used on (bit d of the mask = done on device d).
// Thread-safe; hipFuncSetAttribute itself is idempotent, so a lost race only repeats the call.
#include <atomic>
struct SrDeviceOnce {
    std::atomic<unsigned long long> done{0};
};
template <typename F>
static inline hipError_t sr_once_per_device(SrDeviceOnce& o, F&& f) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (o.done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = f();
    if (e == hipSuccess) o.done.fetch_or(bit, std::memory_order_release);
    return e;
}

// wide-channel 16 x 16-pixel-tile conv (sr_conv_big.hip); sr_conv3x3 routes the shapes it covers there
struct SrConv3x3;
bool sr_conv3x3_big_supported(const SrConv3x3& c);
int sr_conv3x3_big(const SrConv3x3& c, hipStream_t st);
bool sr_conv3x3_narrow_supported(const SrConv3x3& c);  // sr_conv_narrow.hip: RGB tail convs (persistent, register-resident weights)
int sr_conv3x3_narrow(const SrConv3x3& c, hipStream_t st);

// flash-form attention with the bias streamed through LDS (sr_attn_flash.hip)
struct SrOcaAttn;
bool sr_oca_attention_flash_supported(const SrOcaAttn& o);
int sr_oca_attention_flash(const SrOcaAttn& o, hipStream_t st);


// LDS form of the overlapping cross attention (sr_oca_lds.hip): inference (zero-bordered k / v^T) and the training forward (unfolded k / v^T)
struct SrTrAttnFwd;
bool sr_oca_attention_lds_supported(const SrOcaAttn& o);
int sr_oca_attention_lds(const SrOcaAttn& o, hipStream_t st);
bool sr_tr_attn_fwd_lds_supported(const SrTrAttnFwd& f);
int sr_tr_attn_fwd_lds(const SrTrAttnFwd& f, hipStream_t st);

// LDS form of the window-attention backward for 16 x 16 windows (sr_tr_attn_lds.hip): one launch instead of sr_tr_attn.hip's two passes
struct SrTrAttnBwd;
bool sr_tr_attn_bwd_lds_usable(const SrTrAttnBwd& a);
int sr_tr_attn_bwd_lds(const SrTrAttnBwd& a, hipStream_t st);
