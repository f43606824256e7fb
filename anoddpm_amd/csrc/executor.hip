// Native executor + library plumbing: error reporting, flat op-list runner (the C++ side of
// UNetModel.forward, UNet.py:390-406 -- one call per forward instead of ~1000 ATen dispatches),
// and HIP-event timing per op class for bench.py's roofline leg.
#include <stdarg.h>
#include <string.h>
#include <vector>
#include "common.h"

namespace anoddpm {

static thread_local char g_err[512] = "";
int g_debug[16] = {0};

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct ProfState {
    bool on = false;
    std::vector<hipEvent_t> pool;       // pairs: start, stop
    std::vector<int> codes;
    std::vector<float> last_ms;         // per recorded op of the last collect, in launch order
    std::vector<int> last_codes;
    size_t used = 0;
    double ms[ANODDPM_OP_MAX] = {0};
    int64_t launches[ANODDPM_OP_MAX] = {0};
};
static ProfState g_prof;

static int dispatch(const anoddpm_op &op, void *stream)
{
    switch (op.code) {
        case ANODDPM_OP_IGEMM: return anoddpm_igemm(static_cast<const anoddpm_igemm_args *>(op.args), stream);
        case ANODDPM_OP_GN_STATS: return anoddpm_gn_stats(static_cast<const anoddpm_gn_args *>(op.args), stream);
        case ANODDPM_OP_SOFTMAX: return anoddpm_softmax_rows(static_cast<const anoddpm_softmax_args *>(op.args), stream);
        case ANODDPM_OP_RESAMPLE: return anoddpm_resample2x(static_cast<const anoddpm_resample_args *>(op.args), stream);
        case ANODDPM_OP_LINEAR: return anoddpm_linear_small(static_cast<const anoddpm_linear_args *>(op.args), stream);
        case ANODDPM_OP_POSEMB: return anoddpm_posemb(static_cast<const anoddpm_posemb_args *>(op.args), stream);
        case ANODDPM_OP_STEM: return anoddpm_conv_stem(static_cast<const anoddpm_stem_args *>(op.args), stream);
        case ANODDPM_OP_LAYOUT: return anoddpm_nhwc_to_nchw(static_cast<const anoddpm_layout_args *>(op.args), stream);
        case ANODDPM_OP_CHAN_STATS: return anoddpm_chan_stats(static_cast<const anoddpm_chan_stats_args *>(op.args), stream);
        case ANODDPM_OP_HEAD: return anoddpm_conv_head(static_cast<const anoddpm_head_args *>(op.args), stream);
        case ANODDPM_OP_GN_FINALIZE: return anoddpm_gn_finalize(static_cast<const anoddpm_gn_finalize_args *>(op.args), stream);
        case ANODDPM_OP_WGRAD3: return anoddpm_conv3x3_wgrad(static_cast<const anoddpm_wgrad_args *>(op.args), stream);
        case ANODDPM_OP_WGRAD1: return anoddpm_wgrad_pointwise(static_cast<const anoddpm_wgrad1_args *>(op.args), stream);
        case ANODDPM_OP_GN_BWD: return anoddpm_gn_silu_backward(static_cast<const anoddpm_gn_bwd_args *>(op.args), stream);
        case ANODDPM_OP_PACK: return anoddpm_pack_weights(static_cast<const anoddpm_pack_args *>(op.args), stream);
        case ANODDPM_OP_SOFTMAX_BWD: return anoddpm_softmax_rows_backward(static_cast<const anoddpm_softmax_bwd_args *>(op.args), stream);
        case ANODDPM_OP_TRANSPOSE: return anoddpm_transpose_square(static_cast<const anoddpm_transpose_args *>(op.args), stream);
        case ANODDPM_OP_LINEAR_BWD: return anoddpm_linear_small_backward(static_cast<const anoddpm_linear_bwd_args *>(op.args), stream);
        case ANODDPM_OP_STEM_BWD: return anoddpm_conv_stem_backward(static_cast<const anoddpm_stem_bwd_args *>(op.args), stream);
        case ANODDPM_OP_HEAD_BWD: return anoddpm_conv_head_backward(static_cast<const anoddpm_head_bwd_args *>(op.args), stream);
        case ANODDPM_OP_COLSUM_FOLD: return anoddpm_colsum_fold(static_cast<const anoddpm_colsum_fold_args *>(op.args), stream);
        case ANODDPM_OP_ATTENTION: return anoddpm_attention(static_cast<const anoddpm_attention_args *>(op.args), stream);
        case ANODDPM_OP_PACK_BATCH: return anoddpm_pack_batch(static_cast<const anoddpm_pack_batch_args *>(op.args), stream);
        case ANODDPM_OP_DROPOUT: return anoddpm_dropout(static_cast<const anoddpm_dropout_args *>(op.args), stream);
        case ANODDPM_OP_LINEAR_BWD_BATCH: return anoddpm_linear_small_backward_batch(static_cast<const anoddpm_linear_bwd_batch_args *>(op.args), stream);
        default: set_error("run_ops: unknown op code %d", op.code); return ANODDPM_EINVAL;
    }
}

}  // namespace anoddpm

using namespace anoddpm;

// Kernel-variant selector for tests and measurements; NOT part of the public ABI (include/anoddpm_hip.h does not declare it).
// In a product build only the keys whose every value still computes the right result are accepted (0: F(2x2) workgroup
// shape, 4: head kernel form, 5: F(4x4) position- vs channel-sliced, 8: weight-gradient fold form); the timing ablations
// that skip work (keys 1, 2, 3, 6, 7) exist only in a -DANODDPM_ABLATE build (ANODDPM_ABLATE=1 python -m anoddpm_amd.build).
extern "C" int anoddpm_internal_variant(int32_t key, int32_t value)
{
    if (key < 0 || key >= 16) return ANODDPM_EINVAL;
#ifndef ANODDPM_ABLATE
    if (!(key == 0 || key == 4 || key == 5 || key == 8 || key == 9)) {
        set_error("internal_variant: key %d selects a timing ablation; this library was built without ANODDPM_ABLATE", key);
        return ANODDPM_EINVAL;
    }
#endif
    g_debug[key] = value;
    return ANODDPM_OK;
}

extern "C" int anoddpm_ablate_build(void)
{
#ifdef ANODDPM_ABLATE
    return 1;
#else
    return 0;
#endif
}

extern "C" int anoddpm_abi_version(void) { return 24; }

extern "C" const char *anoddpm_last_error(void) { return g_err; }

extern "C" int anoddpm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int anoddpm_run_ops(const anoddpm_op *ops, int32_t n, void *stream)
{
    ANODDPM_REQUIRE(ops || n == 0, "run_ops: null op list");
    for (int i = 0; i < n; ++i) {
        ANODDPM_REQUIRE(ops[i].args, "run_ops: op %d has null args", i);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const bool prof = g_prof.on && ops[i].code > 0 && ops[i].code < ANODDPM_OP_MAX;
        if (prof) {
            if (g_prof.used + 2 > g_prof.pool.size()) {
                hipEvent_t a, b;
                if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
                    set_error("run_ops: hipEventCreate failed");
                    return ANODDPM_ELAUNCH;
                }
                g_prof.pool.push_back(a);
                g_prof.pool.push_back(b);
            }
            e0 = g_prof.pool[g_prof.used];
            e1 = g_prof.pool[g_prof.used + 1];
            g_prof.used += 2;
            // Winograd launches of anoddpm_igemm are booked under code 12 (F(2x2,3x3)) / 14 (F(4x4,3x3)) so the dominant kernels can be priced alone
            // ... and the Winograd-domain launches of anoddpm_conv3x3_wgrad (algo 1) under code 15
            const int icfg = ops[i].code == ANODDPM_OP_IGEMM ? static_cast<const anoddpm_igemm_args *>(ops[i].args)->cfg : 0;
            const bool w43 = ops[i].code == ANODDPM_OP_WGRAD3 && static_cast<const anoddpm_wgrad_args *>(ops[i].args)->algo == 1;
            // ... the small-map no-split launches (cfg 5) under code 13
            g_prof.codes.push_back((icfg == 2 || icfg == 6) ? 12 : ((icfg == 3 || icfg == 7) ? 14 : (icfg == 5 ? 13 : (w43 ? 15 : ops[i].code))));
            (void)hipEventRecord(e0, as_stream(stream));
        }
        const int rc = dispatch(ops[i], stream);
        if (prof) (void)hipEventRecord(e1, as_stream(stream));
        if (rc != ANODDPM_OK) {
            char tmp[400];
            strncpy(tmp, g_err, sizeof(tmp) - 1);
            tmp[sizeof(tmp) - 1] = 0;
            set_error("op %d (code %d): %s", i, ops[i].code, tmp);
            return rc;
        }
    }
    return ANODDPM_OK;
}

extern "C" int anoddpm_prof_enable(int32_t enable)
{
    g_prof.on = enable != 0;
    if (enable) {
        g_prof.used = 0;
        g_prof.codes.clear();
        memset(g_prof.ms, 0, sizeof(g_prof.ms));
        memset(g_prof.launches, 0, sizeof(g_prof.launches));
    }
    return ANODDPM_OK;
}

extern "C" int anoddpm_prof_active(void) { return g_prof.on ? 1 : 0; }

extern "C" int anoddpm_prof_collect(double *ms_per_code, int64_t *launches_per_code)
{
    ANODDPM_REQUIRE(ms_per_code && launches_per_code, "prof_collect: null pointer");
    g_prof.last_ms.clear();
    g_prof.last_codes.clear();
    for (size_t i = 0; i < g_prof.codes.size(); ++i) {
        hipEvent_t e0 = g_prof.pool[2 * i], e1 = g_prof.pool[2 * i + 1];
        if (hipEventSynchronize(e1) != hipSuccess) { set_error("prof_collect: event sync failed"); return ANODDPM_ELAUNCH; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { set_error("prof_collect: elapsed failed"); return ANODDPM_ELAUNCH; }
        g_prof.ms[g_prof.codes[i]] += ms;
        g_prof.launches[g_prof.codes[i]] += 1;
        g_prof.last_ms.push_back(ms);
        g_prof.last_codes.push_back(g_prof.codes[i]);
    }
    g_prof.codes.clear();
    g_prof.used = 0;
    for (int c = 0; c < ANODDPM_OP_MAX; ++c) { ms_per_code[c] = g_prof.ms[c]; launches_per_code[c] = g_prof.launches[c]; }
    return ANODDPM_OK;
}

// The ops of the last anoddpm_prof_collect in launch order: their profiler code and HIP-event time.  Returns the count (at most cap
// entries are written); with a plan's op list beside it this is a per-layer profile without an external tracer.
extern "C" int anoddpm_prof_list(int32_t *codes, float *ms, int32_t cap)
{
    const int n = (int)g_prof.last_ms.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (codes) codes[i] = g_prof.last_codes[i];
        if (ms) ms[i] = g_prof.last_ms[i];
    }
    return n;
}

// sizeof() of every ABI struct, so that the Python ctypes mirror can verify its layout at load time.
extern "C" int anoddpm_struct_size(int32_t which)
{
    switch (which) {
        case 0: return (int)sizeof(anoddpm_simplex_args);
        case 1: return (int)sizeof(anoddpm_p_update_args);
        case 2: return (int)sizeof(anoddpm_igemm_args);
        case 3: return (int)sizeof(anoddpm_gn_args);
        case 4: return (int)sizeof(anoddpm_softmax_args);
        case 5: return (int)sizeof(anoddpm_resample_args);
        case 6: return (int)sizeof(anoddpm_linear_args);
        case 7: return (int)sizeof(anoddpm_posemb_args);
        case 8: return (int)sizeof(anoddpm_stem_args);
        case 9: return (int)sizeof(anoddpm_layout_args);
        case 10: return (int)sizeof(anoddpm_op);
        case 11: return (int)sizeof(anoddpm_adamw_args);
        case 12: return (int)sizeof(anoddpm_chan_stats_args);
        case 13: return (int)sizeof(anoddpm_gn_finalize_args);
        case 14: return (int)sizeof(anoddpm_head_args);
        case 15: return (int)sizeof(anoddpm_anomaly_args);
        case 16: return (int)sizeof(anoddpm_vlb_args);
        case 17: return (int)sizeof(anoddpm_wgrad_args);
        case 18: return (int)sizeof(anoddpm_gn_bwd_args);
        case 19: return (int)sizeof(anoddpm_wgrad1_args);
        case 20: return (int)sizeof(anoddpm_pack_args);
        case 21: return (int)sizeof(anoddpm_softmax_bwd_args);
        case 22: return (int)sizeof(anoddpm_transpose_args);
        case 23: return (int)sizeof(anoddpm_linear_bwd_args);
        case 24: return (int)sizeof(anoddpm_stem_bwd_args);
        case 25: return (int)sizeof(anoddpm_head_bwd_args);
        case 26: return (int)sizeof(anoddpm_colsum_fold_args);
        case 27: return (int)sizeof(anoddpm_mri_slice_args);
        case 28: return (int)sizeof(anoddpm_resize_args);
        case 29: return (int)sizeof(anoddpm_attention_args);
        case 30: return (int)sizeof(anoddpm_pack_batch_args);
        case 31: return (int)sizeof(anoddpm_linear_bwd_batch_args);
        case 32: return (int)sizeof(anoddpm_loss_args);
        case 33: return (int)sizeof(anoddpm_dropout_args);
        default: return -1;
    }
}
