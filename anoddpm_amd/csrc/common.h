// Shared host-side helpers for libanoddpm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/anoddpm_hip.h"

namespace anoddpm {

void set_error(const char *fmt, ...);
extern int g_debug[16];      // kernel-variant selectors (anoddpm_internal_variant, executor.hip); all zero in normal operation

#define ANODDPM_REQUIRE(cond, ...)                       \
    do {                                                 \
        if (!(cond)) {                                   \
            ::anoddpm::set_error(__VA_ARGS__);           \
            return ANODDPM_EINVAL;                       \
        }                                                \
    } while (0)

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return ANODDPM_ELAUNCH;
    }
    return ANODDPM_OK;
}

// Kernels that ask for more than 64 KB of dynamic LDS: the limit is a per-DEVICE function attribute.  `done` is the call site's
// own `static bool[ANODDPM_MAX_DEV]`; the flag of a device is set only after the call succeeded there (a repeated call from a
// second thread is harmless: the attribute is idempotent).
constexpr int ANODDPM_MAX_DEV = 64;
inline int allow_big_lds(const void *fn, bool *done, const char *what)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ANODDPM_MAX_DEV) dev = -1;
    if (dev >= 0 && done[dev]) return ANODDPM_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
        set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", what, hipGetErrorString(e));
        return ANODDPM_ELAUNCH;
    }
    if (dev >= 0) done[dev] = true;
    return ANODDPM_OK;
}

int launch_winograd(const anoddpm_igemm_args *a, hipStream_t s);   // winograd.hip (cfg == 2 of anoddpm_igemm)
int launch_winograd43(const anoddpm_igemm_args *a, hipStream_t s); // winograd43.hip (cfg == 3)
int launch_winograd43r(const anoddpm_igemm_args *a, hipStream_t s); // winograd43r.hip (cfg == 3, 128-channel workgroups)
int launch_winograd43w(const anoddpm_igemm_args *a, hipStream_t s, int tsplit); // winograd43w.hip (the same workgroup as four waves, one per SIMD)
int launch_pointwise_stream(const anoddpm_igemm_args *a, hipStream_t s); // pointwise.hip (cfg == 4)
int launch_smallmap(const anoddpm_igemm_args *a, hipStream_t s);         // smallmap.hip (cfg == 5)
int smallmap_tile(int ks, int H, int W, int K, int c0, int N, int B);
int launch_wino23s(const anoddpm_igemm_args *a, hipStream_t s);          // wino23s.hip (cfg == 6)
int launch_winograd43b(const anoddpm_igemm_args *a, hipStream_t s);      // winograd43b.hip (cfg == 7: split-bf16 side configuration)
int wino23s_tile(int H, int W, int K, int c0, int N, int B, int a_mode);
int launch_wgrad43(const anoddpm_wgrad_args *a, hipStream_t s);      // wgrad43.hip (algo == 1 of anoddpm_conv3x3_wgrad)
int wgrad43_groups(int K, int N, int B, int H, int W);
int wgrad43_colsum_items(int K, int N, int B, int H, int W);      // column-sum rows per image of algo 1

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float silu_f(float x)
{
    // x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (~1e-6 relative, well inside the 1e-3 budget); a plain
    // division would expand to the full IEEE div_scale/div_fmas/div_fixup sequence under -fno-fast-math
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

}  // namespace anoddpm
