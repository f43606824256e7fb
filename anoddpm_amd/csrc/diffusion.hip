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
    // the reference's extract() raises IndexError for t outside [-T, T); a kernel cannot raise, so it never reads out
    // of bounds and poisons the sample with NaN instead (the host wrappers validate host-built t, see diffusion.py)
    const bool bad = ti < 0 || ti >= T;
    ti = bad ? 0 : ti;
    const float qnan = __builtin_nanf("");
    const float a = bad ? qnan : ca[ti], c = bad ? qnan : cb[ti];
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
    const bool bad = ti < 0 || ti >= a.T;            // out of range: no out-of-bounds read, NaN result (see q_sample_kernel)
    ti = bad ? 0 : ti;
    StepCoef k;
    k.recip = bad ? __builtin_nanf("") : a.c_recip[ti];
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


// ---------------------------------------------------------------- VLB terms (GaussianDiffusion.py:384-397, 445-478)
// One pass per reverse step of calc_total_vlb: per sample the KL( q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t) ) or, at t = 0, the
// discretised decoder NLL, in bits per dimension; plus mean((pred_x_0 - x_0)^2) and mean((eps' - noise)^2).
// Replaces ~45 ATen dispatches per step.  fp32 element math in the reference's operation order; the per-sample
// means are fp64 partial sums folded in a fixed order.
constexpr int VLB_BLOCKS = 64;

__device__ __forceinline__ float approx_cdf(float x)
{
    // 0.5 * (1 + tanh(sqrt(2/pi) * (x + 0.044715 * x^3)))          GaussianDiffusion.py:56-61
    const float c = 0.7978845608028654f;
    return 0.5f * (1.0f + tanhf(c * (x + 0.044715f * (x * x * x))));
}

__global__ __launch_bounds__(256) void vlb_kernel(anoddpm_vlb_args a, double *__restrict__ partial)
{
    const int b = blockIdx.y;
    long long ti = a.t[b];
    const bool t0 = (ti == 0);
    ti = ti < 0 ? ti + a.T : ti;
    const bool bad = ti < 0 || ti >= a.T;            // out of range: no out-of-bounds read, NaN result (see q_sample_kernel)
    ti = bad ? 0 : ti;
    const float recip = bad ? __builtin_nanf("") : a.c_recip[ti], recipm1 = a.c_recipm1[ti], coef1 = a.c_coef1[ti], coef2 = a.c_coef2[ti];
    const float lv1 = a.c_post_logvar[ti], lv2 = a.c_model_logvar[ti];
    const float kbase = (-1.0f + lv2) - lv1;                    // -1 + logvar2 - logvar1
    const float e1 = expf(lv1 - lv2), e2 = expf(-lv2);
    const float inv_std = expf(-(0.5f * lv2));
    const int64_t base = (int64_t)b * a.n;
    double s_vlb = 0.0, s_x0 = 0.0, s_eps = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        const float xt = a.xt[base + i], x0 = a.x0[base + i], e = a.eps[base + i];
        float pred = recip * xt - recipm1 * e;                   // :228-230
        pred = fminf(fmaxf(pred, -1.0f), 1.0f);                  // :287
        const float mean = coef1 * pred + coef2 * xt;            // :253-267 (model mean)
        const float tmean = coef1 * x0 + coef2 * xt;             // true posterior mean
        float term;
        if (!t0) {
            const float d = tmean - mean;
            term = 0.5f * ((kbase + e1) + (d * d) * e2);         // normal_kl, :43-53
        } else {
            const float cen = x0 - mean;                         // discretised_gaussian_log_likelihood, :64-93
            const float cp = approx_cdf(inv_std * (cen + 1.0f / 255.0f));
            const float cm = approx_cdf(inv_std * (cen - 1.0f / 255.0f));
            const float lcp = logf(fmaxf(cp, 1e-12f));
            const float l1m = logf(fmaxf(1.0f - cm, 1e-12f));
            const float ld = logf(fmaxf(cp - cm, 1e-12f));
            term = -(x0 < -0.999f ? lcp : (x0 > 0.999f ? l1m : ld));
        }
        s_vlb += (double)term;
        const float dx = pred - x0;
        s_x0 += (double)(dx * dx);
        if (a.noise) {
            const float er = (recip * xt - pred) / recipm1;      // predict_eps_from_x_0, :232-235
            const float de = er - a.noise[base + i];
            s_eps += (double)(de * de);
        }
        if (a.pred_x0) a.pred_x0[base + i] = pred;
    }
    __shared__ double red[4][3];
    double v[3] = {s_vlb, s_x0, s_eps};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        partial[((int64_t)b * gridDim.x + blockIdx.x) * 3 + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    }
}

__global__ void vlb_fold_kernel(const double *__restrict__ partial, float *__restrict__ out, int nblocks, int B, double n)
{
    const int b = blockIdx.x, k = threadIdx.x;
    if (k >= 3) return;
    double v = 0.0;
    for (int j = 0; j < nblocks; ++j) v += partial[((int64_t)b * nblocks + j) * 3 + k];
    v /= n;
    if (k == 0) v /= 0.6931471805599453;                        // / np.log(2.0): bits per dimension
    out[(int64_t)k * B + b] = (float)v;
}

// ---------------------------------------------------------------- training loss (GaussianDiffusion.py:399-434) + its gradient
// calc_loss's per-sample terms and p_loss's weighted mean in one pass over [B][n] (+ a tiny fold), and d(total)/d(eps) in one
// more: replaces the ~15 ATen elementwise / reduction dispatches of the loss expression and the ~25 of its autograd backward
// (the "hybrid" loss's VLB term alone is ~45 forward dispatches).  Element math in fp32, sums in fp64 with a fixed order.
constexpr int LOSS_BLOCKS = 64;

struct VlbCoef {
    float recip, recipm1, coef1, coef2, kbase, e1, e2, inv_std;
    bool t0;
};

__device__ __forceinline__ VlbCoef vlb_coef(const anoddpm_loss_args &a, int b)
{
    VlbCoef k;
    long long ti = a.t[b];
    k.t0 = (ti == 0);
    ti = ti < 0 ? ti + a.T : ti;
    const bool bad = ti < 0 || ti >= a.T;
    ti = bad ? 0 : ti;
    k.recip = bad ? __builtin_nanf("") : a.c_recip[ti];
    k.recipm1 = a.c_recipm1[ti];
    k.coef1 = a.c_coef1[ti];
    k.coef2 = a.c_coef2[ti];
    const float lv1 = a.c_post_logvar[ti], lv2 = a.c_model_logvar[ti];
    k.kbase = (-1.0f + lv2) - lv1;
    k.e1 = expf(lv1 - lv2);
    k.e2 = expf(-lv2);
    k.inv_std = expf(-(0.5f * lv2));
    return k;
}

// d/dx of approx_cdf: 0.5 * (1 - tanh(u)^2) * c * (1 + 3 * 0.044715 x^2),  u = c * (x + 0.044715 x^3)
__device__ __forceinline__ float approx_cdf_grad(float x)
{
    const float c = 0.7978845608028654f;
    const float th = tanhf(c * (x + 0.044715f * (x * x * x)));
    return 0.5f * (1.0f - th * th) * (c * (1.0f + 3.0f * 0.044715f * (x * x)));
}

// One element of the VLB term (calc_vlb_xt, :384-397): returns the term (nats, before mean_flat and / ln 2); *dterm_deps
// receives its derivative with respect to the model output eps (through mean = coef1 * clamp(recip x_t - recipm1 eps) + coef2 x_t;
// torch.clamp passes the gradient on [-1, 1] inclusive, clamp(min=1e-12) on [1e-12, inf)).
__device__ __forceinline__ float vlb_element(const VlbCoef &k, float x0, float xt, float e, float *dterm_deps)
{
    const float raw = k.recip * xt - k.recipm1 * e;
    const float pred = fminf(fmaxf(raw, -1.0f), 1.0f);
    const float dpred = (raw >= -1.0f && raw <= 1.0f) ? -k.recipm1 : 0.0f;
    const float mean = k.coef1 * pred + k.coef2 * xt;
    float term, dmean;
    if (!k.t0) {
        const float tmean = k.coef1 * x0 + k.coef2 * xt;
        const float d = tmean - mean;
        term = 0.5f * ((k.kbase + k.e1) + (d * d) * k.e2);
        dmean = -(d * k.e2);
    } else {
        const float cen = x0 - mean;
        const float zp = k.inv_std * (cen + 1.0f / 255.0f), zm = k.inv_std * (cen - 1.0f / 255.0f);
        const float cp = approx_cdf(zp), cm = approx_cdf(zm);
        float lp, dcen;                              // log-probability and its derivative w.r.t. cen
        if (x0 < -0.999f) {
            lp = logf(fmaxf(cp, 1e-12f));
            dcen = cp >= 1e-12f ? approx_cdf_grad(zp) * k.inv_std / cp : 0.0f;
        } else if (x0 > 0.999f) {
            const float om = 1.0f - cm;
            lp = logf(fmaxf(om, 1e-12f));
            dcen = om >= 1e-12f ? -(approx_cdf_grad(zm) * k.inv_std) / om : 0.0f;
        } else {
            const float dl = cp - cm;
            lp = logf(fmaxf(dl, 1e-12f));
            dcen = dl >= 1e-12f ? (approx_cdf_grad(zp) - approx_cdf_grad(zm)) * k.inv_std / dl : 0.0f;
        }
        term = -lp;
        dmean = dcen;                                // term = -lp(cen), cen = x0 - mean  ->  d term / d mean = + d lp / d cen
    }
    *dterm_deps = dmean * k.coef1 * dpred;
    return term;
}

__global__ __launch_bounds__(256) void loss_fwd_kernel(anoddpm_loss_args a)
{
    const int b = blockIdx.y;
    const bool hybrid = a.kind == 2;
    VlbCoef k = {};
    if (hybrid) k = vlb_coef(a, b);
    const int64_t base = (int64_t)b * a.n;
    double s_main = 0.0, s_vlb = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        const float e = a.eps[base + i];
        const float d = e - a.noise[base + i];
        s_main += (double)(a.kind == 0 ? fabsf(d) : d * d);
        if (hybrid) {
            float unused;
            s_vlb += (double)vlb_element(k, a.x0[base + i], a.xt[base + i], e, &unused);
        }
    }
    __shared__ double red[4][2];
    double v[2] = {s_main, s_vlb};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        for (int off = 32; off > 0; off >>= 1) v[j] += __shfl_xor(v[j], off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v[j];
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int j = threadIdx.x;
        a.workspace[((int64_t)b * gridDim.x + blockIdx.x) * 2 + j] = ((red[0][j] + red[1][j]) + red[2][j]) + red[3][j];
    }
}

// one block: per-sample values (thread b, b + 256, ...) then the weighted mean over the batch in index order
__global__ __launch_bounds__(256) void loss_fold_kernel(anoddpm_loss_args a, int nblocks)
{
    __shared__ double acc[256];
    double mine = 0.0;
    for (int b = threadIdx.x; b < a.B; b += 256) {
        double m = 0.0, v = 0.0;
        for (int j = 0; j < nblocks; ++j) {
            m += a.workspace[((int64_t)b * nblocks + j) * 2 + 0];
            v += a.workspace[((int64_t)b * nblocks + j) * 2 + 1];
        }
        const float main_f = (float)(m / (double)a.n);
        float per = main_f;
        if (a.kind == 2) {
            const float vlb_f = (float)(v / (double)a.n / 0.6931471805599453);
            if (a.vlb) a.vlb[b] = vlb_f;
            per = vlb_f + main_f;                    // loss["vlb"] + mean_flat(mse), :414
        }
        a.per_sample[b] = per;
        mine += (double)(a.weights ? per * a.weights[b] : per);
    }
    acc[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        const int nt = a.B < 256 ? a.B : 256;
        for (int i = 0; i < nt; ++i) s += acc[i];
        if (a.total) a.total[0] = (float)(s / (double)a.B);
    }
}

// d_eps[b][i] = c_b * d(main term)/d(eps) / n + v_b * d(vlb term)/d(eps) / (n ln 2) with
//   c_b = g_per[b] + g_total * w_b / B     (loss["loss"][b] and the weighted batch mean both contain sample b's terms)
//   v_b = (hybrid: c_b) + g_vlb[b]
__global__ __launch_bounds__(256) void loss_bwd_kernel(anoddpm_loss_args a)
{
    const int b = blockIdx.y;
    const bool hybrid = a.kind == 2;
    float cb = a.g_per ? a.g_per[b] : 0.0f;
    if (a.g_total) cb += a.g_total[0] * (a.weights ? a.weights[b] : 1.0f) / (float)a.B;
    float vb = hybrid ? cb : 0.0f;
    if (hybrid && a.g_vlb) vb += a.g_vlb[b];
    const float inv_n = 1.0f / (float)a.n;
    const float cm = cb * inv_n, cv = vb * inv_n / 0.6931471805599453f;
    VlbCoef k = {};
    if (hybrid) k = vlb_coef(a, b);
    const int64_t base = (int64_t)b * a.n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        const float e = a.eps[base + i];
        const float d = e - a.noise[base + i];
        float g = a.kind == 0 ? cm * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) : cm * (2.0f * d);
        if (hybrid) {
            float dv;
            (void)vlb_element(k, a.x0[base + i], a.xt[base + i], e, &dv);
            g += cv * dv;
        }
        a.d_eps[base + i] = g;
    }
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

extern "C" int anoddpm_vlb_terms(const anoddpm_vlb_args *a, void *stream)
{
    ANODDPM_REQUIRE(a, "vlb_terms: null argument struct");
    ANODDPM_REQUIRE(a->B >= 0 && a->n >= 0 && a->T > 0, "vlb_terms: bad sizes");
    if (a->B == 0 || a->n == 0) return ANODDPM_OK;
    ANODDPM_REQUIRE(a->x0 && a->xt && a->eps && a->t && a->out && a->workspace, "vlb_terms: null pointer");
    ANODDPM_REQUIRE(a->c_recip && a->c_recipm1 && a->c_coef1 && a->c_coef2 && a->c_post_logvar && a->c_model_logvar, "vlb_terms: null table");
    ANODDPM_REQUIRE(a->B <= 65535, "vlb_terms: B > 65535");
    ANODDPM_REQUIRE(a->workspace_doubles >= (int64_t)a->B * VLB_BLOCKS * 3, "vlb_terms: workspace too small");
    hipStream_t s = anoddpm::as_stream(stream);
    hipLaunchKernelGGL(vlb_kernel, dim3(VLB_BLOCKS, a->B), dim3(256), 0, s, *a, a->workspace);
    hipLaunchKernelGGL(vlb_fold_kernel, dim3(a->B), dim3(64), 0, s, a->workspace, a->out, VLB_BLOCKS, a->B, (double)a->n);
    return anoddpm::check_launch("vlb_terms");
}

static int loss_check(const anoddpm_loss_args *a, const char *who)
{
    ANODDPM_REQUIRE(a, "%s: null argument struct", who);
    ANODDPM_REQUIRE(a->B >= 0 && a->n >= 0 && a->kind >= 0 && a->kind <= 2, "%s: bad sizes / kind", who);
    if (a->B == 0 || a->n == 0) return 1;
    ANODDPM_REQUIRE(a->eps && a->noise, "%s: null eps / noise", who);
    ANODDPM_REQUIRE(a->B <= 65535, "%s: B > 65535", who);
    if (a->kind == 2) {
        ANODDPM_REQUIRE(a->x0 && a->xt && a->t && a->T > 0, "%s: the hybrid loss needs x0, xt, t", who);
        ANODDPM_REQUIRE(a->c_recip && a->c_recipm1 && a->c_coef1 && a->c_coef2 && a->c_post_logvar && a->c_model_logvar, "%s: null table", who);
    }
    return 0;
}

extern "C" int anoddpm_loss_forward(const anoddpm_loss_args *a, void *stream)
{
    const int rc = loss_check(a, "loss_forward");
    if (rc) return rc < 0 ? rc : ANODDPM_OK;
    ANODDPM_REQUIRE(a->per_sample && a->workspace, "loss_forward: null output / workspace");
    ANODDPM_REQUIRE(a->workspace_doubles >= (int64_t)a->B * LOSS_BLOCKS * 2, "loss_forward: workspace too small");
    hipStream_t s = anoddpm::as_stream(stream);
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(LOSS_BLOCKS, a->B), dim3(256), 0, s, *a);
    hipLaunchKernelGGL(loss_fold_kernel, dim3(1), dim3(256), 0, s, *a, LOSS_BLOCKS);
    return anoddpm::check_launch("loss_forward");
}

extern "C" int anoddpm_loss_backward(const anoddpm_loss_args *a, void *stream)
{
    const int rc = loss_check(a, "loss_backward");
    if (rc) return rc < 0 ? rc : ANODDPM_OK;
    ANODDPM_REQUIRE(a->d_eps, "loss_backward: null d_eps");
    const int64_t want = (a->n + 255) / 256;
    const unsigned gx = (unsigned)(want > 1024 ? 1024 : want);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(gx, a->B), dim3(256), 0, anoddpm::as_stream(stream), *a);
    return anoddpm::check_launch("loss_backward");
}
