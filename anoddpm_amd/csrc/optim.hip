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

__global__ __launch_bounds__(256) void sumsq_kernel(const float *g, int64_t n, float *out)
{
    __shared__ float part[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += g[i] * g[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
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

extern "C" int anoddpm_sumsq(const float *g, int64_t n, float *out, void *stream)
{
    ANODDPM_REQUIRE(g && out && n >= 0, "sumsq: bad arguments");
    if (n == 0) return ANODDPM_OK;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0,
                       anoddpm::as_stream(stream), g, n, out);
    return anoddpm::check_launch("sumsq");
}
