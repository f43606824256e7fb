// Fused AdamW + EMA over a flat fp32 parameter buffer, and the sum-of-squares reduction used for
// clip_grad_norm_ -- replaces torch.optim.AdamW.step (diffusion_training.py:75,105), the 536-tensor
// EMA loop (UNet.py:423-427) and clip_grad_norm_ (diffusion_training.py:104).  HBM-bound:
// reads p,g,m,v,ema and writes p,m,v,ema = 36 B per parameter.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adamw_ema_kernel(anoddpm_adamw_args a, float bc1, float bc2)
{
    const float gs = a.grad_scale ? *a.grad_scale : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        const float g = a.g[i] * gs;
        float p = a.p[i];
        p = p * (1.0f - a.lr * a.weight_decay);                 // decoupled weight decay
        const float m = a.beta1 * a.m[i] + (1.0f - a.beta1) * g;
        const float v = a.beta2 * a.v[i] + (1.0f - a.beta2) * g * g;
        const float denom = sqrtf(v) / sqrtf(bc2) + a.eps;
        p = p - (a.lr / bc1) * (m / denom);
        a.p[i] = p;
        a.m[i] = m;
        a.v[i] = v;
        if (a.ema) a.ema[i] = a.ema[i] * a.ema_decay + p * (1.0f - a.ema_decay);
    }
}

constexpr int SUMSQ_BLOCKS = 2048;

// stage 1: one fp64 partial per block; stage 2 (one block): fixed-order fold -> {sum of squares, norm, clip factor}.
// No atomics: data-parallel replicas that hold bit-identical reduced gradients must compute bit-identical clip factors,
// or their parameters drift apart.
__global__ __launch_bounds__(256) void sumsq_kernel(const float *g, int64_t n, double *partial)
{
    __shared__ double part[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = g[i];
        s += (double)v * (double)v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((part[0] + part[1]) + part[2]) + part[3];
}

__global__ __launch_bounds__(256) void sumsq_fold_kernel(const double *partial, int nblocks, float max_norm, float *out)
{
    __shared__ double acc[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
    acc[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 256; ++i) t += acc[i];
        const float ss = (float)t;
        const float norm = sqrtf(ss);
        out[0] = ss;
        out[1] = norm;
        // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
        out[2] = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
    }
}

}  // namespace

extern "C" int anoddpm_adamw_ema(const anoddpm_adamw_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->p && a->m && a->v && a->g && a->n >= 0 && a->step >= 1, "adamw_ema: bad arguments");
    if (a->n == 0) return ANODDPM_OK;
    const float bc1 = 1.0f - powf(a->beta1, (float)a->step);
    const float bc2 = 1.0f - powf(a->beta2, (float)a->step);
    const int64_t blocks = (a->n + 255) / 256;
    hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0,
                       anoddpm::as_stream(stream), *a, bc1, bc2);
    return anoddpm::check_launch("adamw_ema");
}

extern "C" int anoddpm_sumsq(const float *g, int64_t n, float *out, double *workspace, float max_norm, void *stream)
{
    ANODDPM_REQUIRE(g && out && workspace && n >= 0, "sumsq: bad arguments");
    const int64_t blocks = (n + 255) / 256;
    const int nb = (int)(blocks > SUMSQ_BLOCKS ? SUMSQ_BLOCKS : (blocks < 1 ? 1 : blocks));
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, anoddpm::as_stream(stream), g, n, workspace);
    hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, anoddpm::as_stream(stream), workspace, nb, max_norm, out);
    return anoddpm::check_launch("sumsq");
}
