// Fused AdamW + EMA over a flat fp32 parameter buffer, and the sum-of-squares reduction used for
// clip_grad_norm_ -- replaces torch.optim.AdamW.step (diffusion_training.py:75,105), the 536-tensor
// EMA loop (UNet.py:423-427) and clip_grad_norm_ (diffusion_training.py:104).  HBM-bound:
// reads p,g,m,v,ema and writes p,m,v,ema = 36 B per parameter.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void adamw_one(const anoddpm_adamw_args &a, float gs, float bc1, float bc2, float g, float &p, float &m, float &v, float &e)
{
    g *= gs;
    p = p * (1.0f - a.lr * a.weight_decay);                     // decoupled weight decay
    m = a.beta1 * m + (1.0f - a.beta1) * g;
    v = a.beta2 * v + (1.0f - a.beta2) * g * g;
    const float denom = sqrtf(v) / sqrtf(bc2) + a.eps;
    p = p - (a.lr / bc1) * (m / denom);
    e = e * a.ema_decay + p * (1.0f - a.ema_decay);
}

// Shaped after tools/hbm_patterns.hip: a workgroup owns a contiguous range of `per` 16-byte quads (contiguous ownership keeps the
// write streams at ~6 TB/s at any occupancy), streams are read with the nt policy (nothing here is re-read before the next step,
// 2.6 GB later), ADAM_UNROLL quads per stream in flight.  Per-element arithmetic is unchanged (bit-identical to the scalar form).
constexpr int ADAM_UNROLL = 2;

__global__ __launch_bounds__(256) void adamw_ema_kernel(anoddpm_adamw_args a, float bc1, float bc2, int64_t per)
{
    const float gs = a.grad_scale ? *a.grad_scale : 1.0f;
    const int64_t n4 = a.n >> 2;
    const int64_t q0 = (int64_t)blockIdx.x * per, q1 = (q0 + per < n4) ? q0 + per : n4;
    const f32x4 *G = reinterpret_cast<const f32x4 *>(a.g);
    f32x4 *P = reinterpret_cast<f32x4 *>(a.p), *M = reinterpret_cast<f32x4 *>(a.m), *V = reinterpret_cast<f32x4 *>(a.v);
    f32x4 *E = reinterpret_cast<f32x4 *>(a.ema);
    int64_t i = q0 + threadIdx.x;
    for (; i + (ADAM_UNROLL - 1) * 256 < q1; i += ADAM_UNROLL * 256) {
        f32x4 g[ADAM_UNROLL], p[ADAM_UNROLL], m[ADAM_UNROLL], v[ADAM_UNROLL], e[ADAM_UNROLL];
#pragma unroll
        for (int u = 0; u < ADAM_UNROLL; ++u) {
            g[u] = __builtin_nontemporal_load(G + i + u * 256);
            p[u] = __builtin_nontemporal_load(P + i + u * 256);
            m[u] = __builtin_nontemporal_load(M + i + u * 256);
            v[u] = __builtin_nontemporal_load(V + i + u * 256);
            e[u] = E ? __builtin_nontemporal_load(E + i + u * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < ADAM_UNROLL; ++u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pp = p[u][k], mm = m[u][k], vv = v[u][k], ee = e[u][k];
                adamw_one(a, gs, bc1, bc2, g[u][k], pp, mm, vv, ee);
                p[u][k] = pp; m[u][k] = mm; v[u][k] = vv; e[u][k] = ee;
            }
            P[i + u * 256] = p[u];
            M[i + u * 256] = m[u];
            V[i + u * 256] = v[u];
            if (E) E[i + u * 256] = e[u];
        }
    }
    for (; i < q1; i += 256) {
        f32x4 g = G[i], p = P[i], m = M[i], v = V[i], e = E ? E[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pp = p[k], mm = m[k], vv = v[k], ee = e[k];
            adamw_one(a, gs, bc1, bc2, g[k], pp, mm, vv, ee);
            p[k] = pp; m[k] = mm; v[k] = vv; e[k] = ee;
        }
        P[i] = p; M[i] = m; V[i] = v;
        if (E) E[i] = e;
    }
    // the n % 4 tail elements
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const int64_t t = (n4 << 2) + threadIdx.x;
        float p = a.p[t], m = a.m[t], v = a.v[t], e = a.ema ? a.ema[t] : 0.f;
        adamw_one(a, gs, bc1, bc2, a.g[t], p, m, v, e);
        a.p[t] = p; a.m[t] = m; a.v[t] = v;
        if (a.ema) a.ema[t] = e;
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
    // contiguous range per workgroup, 16-byte nt loads, four in flight; fixed order per (n, grid) -> deterministic
    const int64_t n4 = n >> 2;
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t q0 = (int64_t)blockIdx.x * per, q1 = (q0 + per < n4) ? q0 + per : n4;
    const f32x4 *G = reinterpret_cast<const f32x4 *>(g);
    int64_t i = q0 + threadIdx.x;
    for (; i + 3 * 256 < q1; i += 4 * 256) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(G + i + u * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) s += (double)v[u][k] * (double)v[u][k];
    }
    for (; i < q1; i += 256) {
        const f32x4 v = G[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) s += (double)v[k] * (double)v[k];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
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
    ANODDPM_REQUIRE(((uintptr_t)a->p | (uintptr_t)a->m | (uintptr_t)a->v | (uintptr_t)a->g | (uintptr_t)a->ema) % 16 == 0,
                    "adamw_ema: buffers must be 16-byte aligned");
    // ~16 workgroups per CU, each owning a contiguous range of at least one unrolled trip
    const int64_t n4 = a->n >> 2;
    int64_t per = (n4 + 4095) / 4096;
    const int64_t trip = 256 * ADAM_UNROLL;
    per = per < trip ? trip : ((per + trip - 1) / trip) * trip;
    const int64_t blocks = n4 > 0 ? (n4 + per - 1) / per : 1;
    hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, anoddpm::as_stream(stream), *a, bc1, bc2, per);
    return anoddpm::check_launch("adamw_ema");
}

extern "C" int anoddpm_sumsq(const float *g, int64_t n, float *out, double *workspace, float max_norm, void *stream)
{
    ANODDPM_REQUIRE(g && out && workspace && n >= 0 && (uintptr_t)g % 16 == 0, "sumsq: bad arguments (g must be 16-byte aligned)");
    const int64_t blocks = (n + 255) / 256;
    const int nb = (int)(blocks > SUMSQ_BLOCKS ? SUMSQ_BLOCKS : (blocks < 1 ? 1 : blocks));
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, anoddpm::as_stream(stream), g, n, workspace);
    hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, anoddpm::as_stream(stream), workspace, nb, max_norm, out);
    return anoddpm::check_launch("sumsq");
}
