// Anomaly map + segmentation counts in one pass -- replaces, for one image and its `navg` reconstructions,
//   torch.mean(output, dim=0)                                   GaussianDiffusion.py:517, 572
//   mse = (mean - x_0)^2 * 2 - 1 ; threshold = (mse > 0)*2 - 1   GaussianDiffusion.py:518-520, 581-583; evaluation.py:13-15
//   mse = (image - output)^2 ; (mse > 0.5).float()               detection.py:229-232; evaluation.py:31-32
//   dice / IoU / precision / recall / FPR sums                   evaluation.py:26-76
// HBM-bound: reads (navg + 2) floats per pixel, writes up to 4.  Counts are produced as per-block partials and
// folded in a fixed order by a second tiny kernel (deterministic; counts are integers, exact in fp64).
// Compiled with -ffp-contract=off: the maps are bit-identical to the reference's separate fp32 ATen ops.
#include "common.h"

namespace {

constexpr int NC = ANODDPM_ANOMALY_NCOUNTS;

__global__ __launch_bounds__(256) void anomaly_kernel(anoddpm_anomaly_args a, double *__restrict__ partial)
{
    const int b = blockIdx.y;
    const float *rec = a.recon + (int64_t)b * a.recon_bs;
    const float *real = a.real + (int64_t)b * a.n;
    const float *mask = a.mask ? a.mask + (int64_t)b * a.n : nullptr;
    const float inv = 1.0f / (float)a.navg;
    double c[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) c[i] = 0.0;
    c[10] = -INFINITY;                                  // running max(real): images may be all-negative
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        // torch.mean over dim 0: sequential fp32 sum in index order, then one multiply by 1/navg would differ from
        // ATen's sum/N by an ulp for non-power-of-two N, so divide like ATen does.
        float s = rec[i];
        for (int k = 1; k < a.navg; ++k) s += rec[(int64_t)k * a.recon_as + i];
        const float m = a.navg > 1 ? s / (float)a.navg : s;
        (void)inv;
        const float d = m - real[i];
        const float se = d * d;
        const float img = se * 2.0f - 1.0f;
        const float pred = se > a.threshold ? 1.0f : 0.0f;
        const int64_t o = (int64_t)b * a.n + i;
        if (a.mean) a.mean[o] = m;
        if (a.sqerr) a.sqerr[o] = se;
        if (a.mse_img) a.mse_img[o] = img;
        if (a.thr_img) a.thr_img[o] = img > 0.0f ? 1.0f : -1.0f;
        if (a.pred) a.pred[o] = pred;
        const float mk = mask ? mask[i] : 0.0f;
        c[0] += pred;                                   // sum(mse)                      evaluation.py:34
        c[1] += mk;                                     // sum(real_mask)                evaluation.py:34
        c[2] += pred * mk;                              // sum(mse * real_mask)          evaluation.py:33
        c[3] += (mk == 1.0f && pred == 1.0f) ? 1.0 : 0.0;   // (real==1)&(recon==1)      evaluation.py:59,66
        c[4] += (mk == 1.0f && pred == 0.0f) ? 1.0 : 0.0;   // (real==1)&(recon==0)      evaluation.py:60,72
        c[5] += (mk == 0.0f && pred == 1.0f) ? 1.0 : 0.0;   // (real==0)&(recon==1)      evaluation.py:67
        c[6] += (mk == 0.0f && pred == 0.0f) ? 1.0 : 0.0;   // (real==0)&(recon==0)      evaluation.py:73
        c[7] += (mk != 0.0f && pred != 0.0f) ? 1.0 : 0.0;   // logical_and               evaluation.py:52
        c[8] += (mk != 0.0f || pred != 0.0f) ? 1.0 : 0.0;   // logical_or                evaluation.py:53
        c[9] += (double)se;                             // sum of squared error (PSNR / MSE, evaluation.py:41-42)
        c[10] = fmax(c[10], (double)real[i]);           // max(real)                     evaluation.py:43
    }
    __shared__ double red[4][NC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        double v = c[i];
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(v, off);
            v = (i == 10) ? fmax(v, o) : v + o;
        }
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NC) {
        const int i = threadIdx.x;
        double v = red[0][i];
        for (int w = 1; w < 4; ++w) v = (i == 10) ? fmax(v, red[w][i]) : v + red[w][i];
        partial[((int64_t)b * gridDim.x + blockIdx.x) * NC + i] = v;
    }
}

__global__ void anomaly_fold_kernel(const double *__restrict__ partial, double *__restrict__ counts, int nblocks)
{
    const int b = blockIdx.x, i = threadIdx.x;
    if (i >= NC) return;
    double v = partial[((int64_t)b * nblocks) * NC + i];
    for (int k = 1; k < nblocks; ++k) {
        const double o = partial[((int64_t)b * nblocks + k) * NC + i];
        v = (i == 10) ? fmax(v, o) : v + o;
    }
    counts[(int64_t)b * NC + i] = v;
}

}  // namespace

extern "C" int anoddpm_anomaly_map(const anoddpm_anomaly_args *a, void *stream)
{
    using namespace anoddpm;
    ANODDPM_REQUIRE(a != nullptr, "anomaly_map: null args");
    if (a->B <= 0 || a->n <= 0) return ANODDPM_OK;
    ANODDPM_REQUIRE(a->recon && a->real && a->counts && a->workspace, "anomaly_map: null pointer");
    ANODDPM_REQUIRE(a->navg >= 1, "anomaly_map: navg must be >= 1");
    ANODDPM_REQUIRE(a->B <= 65535, "anomaly_map: batch too large");
    const int nblocks = ANODDPM_ANOMALY_BLOCKS;
    ANODDPM_REQUIRE(a->workspace_doubles >= (int64_t)a->B * nblocks * NC, "anomaly_map: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(anomaly_kernel, dim3(nblocks, a->B), dim3(256), 0, s, *a, a->workspace);
    hipLaunchKernelGGL(anomaly_fold_kernel, dim3(a->B), dim3(64), 0, s, a->workspace, a->counts, nblocks);
    return check_launch("anomaly_map");
}
