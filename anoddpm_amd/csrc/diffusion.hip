// Fused forward / reverse diffusion updates -- replaces the ~25 ATen dispatches per step of
// GaussianDiffusion.py:32-36 (extract), :361-382 (sample_q, sample_q_gradual), :228-230
// (predict_x_0_from_eps), :253-267 (posterior mean), :287 (clamp) and :314-317 (sample_p).
// HBM-bound elementwise work: sample_q moves 12 B/pixel, the reverse update 16 B/pixel.
// Compiled with -ffp-contract=off so every product and sum rounds exactly as the reference's
// separate fp32 ATen kernels do (results are bit-identical to its CPU path).
#include "common.h"

namespace {

template <int VEC>
__global__ __launch_bounds__(256) void q_sample_kernel(float *__restrict__ out, const float *__restrict__ x,
                                                       const float *__restrict__ noise,
                                                       const int64_t *__restrict__ t,
                                                       const float *__restrict__ ca,
                                                       const float *__restrict__ cb, int64_t n, int T)
{
    const int b = blockIdx.y;
    long long ti = t[b];
    ti = ti < 0 ? ti + T : ti;                       // python-style negative index
    const float a = ca[ti], c = cb[ti];
    const int64_t base = (int64_t)b * n;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; i < n; i += (int64_t)gridDim.x * 256 * VEC) {
        if (VEC == 4) {
            const float4 xv = *reinterpret_cast<const float4 *>(x + base + i);
            const float4 nv = *reinterpret_cast<const float4 *>(noise + base + i);
            float4 o;
            o.x = a * xv.x + c * nv.x;
            o.y = a * xv.y + c * nv.y;
            o.z = a * xv.z + c * nv.z;
            o.w = a * xv.w + c * nv.w;
            *reinterpret_cast<float4 *>(out + base + i) = o;
        } else {
            out[base + i] = a * x[base + i] + c * noise[base + i];
        }
    }
}

struct StepCoef {
    float recip, recipm1, coef1, coef2, sigma;
};

__device__ __forceinline__ float reverse_one(float xt, float e, float nz, const StepCoef &k, float *px0, float *pmean)
{
    float x0 = k.recip * xt - k.recipm1 * e;
    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    const float mean = k.coef1 * x0 + k.coef2 * xt;
    *px0 = x0;
    *pmean = mean;
    return mean + k.sigma * nz;
}

template <int VEC>
__global__ __launch_bounds__(256) void p_update_kernel(anoddpm_p_update_args a)
{
    const int b = blockIdx.y;
    long long ti = a.t[b];
    const bool nonzero = (ti != 0);
    ti = ti < 0 ? ti + a.T : ti;
    StepCoef k;
    k.recip = a.c_recip[ti];
    k.recipm1 = a.c_recipm1[ti];
    k.coef1 = a.c_coef1[ti];
    k.coef2 = a.c_coef2[ti];
    k.sigma = nonzero ? a.c_sigma[ti] : 0.0f;        // (t != 0).float() * exp(0.5*logvar)
    const int64_t base = (int64_t)b * a.n;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; i < a.n; i += (int64_t)gridDim.x * 256 * VEC) {
        if (VEC == 4) {
            const float4 xv = *reinterpret_cast<const float4 *>(a.x_t + base + i);
            const float4 ev = *reinterpret_cast<const float4 *>(a.eps + base + i);
            float4 nv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.noise) nv = *reinterpret_cast<const float4 *>(a.noise + base + i);
            float4 o, p0, mu;
            o.x = reverse_one(xv.x, ev.x, nv.x, k, &p0.x, &mu.x);
            o.y = reverse_one(xv.y, ev.y, nv.y, k, &p0.y, &mu.y);
            o.z = reverse_one(xv.z, ev.z, nv.z, k, &p0.z, &mu.z);
            o.w = reverse_one(xv.w, ev.w, nv.w, k, &p0.w, &mu.w);
            *reinterpret_cast<float4 *>(a.x_prev + base + i) = o;
            if (a.pred_x0) *reinterpret_cast<float4 *>(a.pred_x0 + base + i) = p0;
            if (a.mean_out) *reinterpret_cast<float4 *>(a.mean_out + base + i) = mu;
        } else {
            float p0, mu;
            const float nz = a.noise ? a.noise[base + i] : 0.0f;
            const float o = reverse_one(a.x_t[base + i], a.eps[base + i], nz, k, &p0, &mu);
            a.x_prev[base + i] = o;
            if (a.pred_x0) a.pred_x0[base + i] = p0;
            if (a.mean_out) a.mean_out[base + i] = mu;
        }
    }
}

__global__ void chain_advance_kernel(int64_t *t, int B, int32_t *step)
{
    const int i = threadIdx.x;
    if (i < B) t[i] -= 1;
    if (i == 0 && step) *step += 1;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int anoddpm_q_sample(float *out, const float *x, const float *noise, const int64_t *t,
                                const float *ca, const float *cb, int32_t B, int64_t n, int32_t T, void *stream)
{
    ANODDPM_REQUIRE(B >= 0 && n >= 0 && T > 0, "q_sample: bad sizes");
    if (B == 0 || n == 0) return ANODDPM_OK;            // empty batch: nothing to do (pointers may be null)
    ANODDPM_REQUIRE(out && x && noise && t && ca && cb, "q_sample: null pointer");
    ANODDPM_REQUIRE(B <= 65535, "q_sample: B > 65535");
    const bool v4 = (n % 4 == 0) && aligned16(out) && aligned16(x) && aligned16(noise);
    const int64_t work = v4 ? n / 4 : n;
    const unsigned gx = (unsigned)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
    if (v4)
        hipLaunchKernelGGL(q_sample_kernel<4>, dim3(gx, B), dim3(256), 0, anoddpm::as_stream(stream), out, x, noise, t, ca, cb, n, T);
    else
        hipLaunchKernelGGL(q_sample_kernel<1>, dim3(gx, B), dim3(256), 0, anoddpm::as_stream(stream), out, x, noise, t, ca, cb, n, T);
    return anoddpm::check_launch("q_sample");
}

extern "C" int anoddpm_p_sample_update(const anoddpm_p_update_args *a, void *stream)
{
    ANODDPM_REQUIRE(a, "p_sample_update: null argument struct");
    ANODDPM_REQUIRE(a->B >= 0 && a->n >= 0 && a->T > 0, "p_sample_update: bad sizes");
    if (a->B == 0 || a->n == 0) return ANODDPM_OK;
    ANODDPM_REQUIRE(a->x_prev && a->x_t && a->eps && a->t, "p_sample_update: null pointer");
    ANODDPM_REQUIRE(a->c_recip && a->c_recipm1 && a->c_coef1 && a->c_coef2 && a->c_sigma, "p_sample_update: null table");
    ANODDPM_REQUIRE(a->B <= 65535, "p_sample_update: B > 65535");
    const bool v4 = (a->n % 4 == 0) && aligned16(a->x_prev) && aligned16(a->x_t) && aligned16(a->eps) &&
                    (!a->noise || aligned16(a->noise)) && (!a->pred_x0 || aligned16(a->pred_x0)) &&
                    (!a->mean_out || aligned16(a->mean_out));
    const int64_t work = v4 ? a->n / 4 : a->n;
    const unsigned gx = (unsigned)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
    if (v4)
        hipLaunchKernelGGL(p_update_kernel<4>, dim3(gx, a->B), dim3(256), 0, anoddpm::as_stream(stream), *a);
    else
        hipLaunchKernelGGL(p_update_kernel<1>, dim3(gx, a->B), dim3(256), 0, anoddpm::as_stream(stream), *a);
    return anoddpm::check_launch("p_sample_update");
}

extern "C" int anoddpm_chain_advance(int64_t *t, int32_t B, int32_t *step, void *stream)
{
    ANODDPM_REQUIRE(t && B >= 0 && B <= 1024, "chain_advance: bad arguments");
    hipLaunchKernelGGL(chain_advance_kernel, dim3(1), dim3(B < 64 ? 64 : ((B + 63) / 64) * 64), 0,
                       anoddpm::as_stream(stream), t, B, step);
    return anoddpm::check_launch("chain_advance");
}
