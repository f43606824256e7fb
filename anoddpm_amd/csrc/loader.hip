// MRI slice loader on the device -- replaces the per-sample CPU path of MRIDataset (dataset.py:575-643): volume
// normalisation (:585-594), slice cut (:621-625) and the deterministic + random parts of its default transform
// (torchvision RandomAffine(3, translate) -> CenterCrop(235) -> Resize(bilinear) -> ToTensor -> Normalize(0.5, 0.5), :584-593).
// torchvision is a thin wrapper over PIL for this pipeline, so the arithmetic restated here is PIL's (12.x):
//   * Image.transform(AFFINE, NEAREST): 16.16 fixed-point source walk (Geometry.c affine_fixed)
//   * Image.resize(BILINEAR): separable triangle filter widened by the scale factor, horizontal pass then vertical pass,
//     double accumulation in ascending tap order, float32 store after each pass (Resample.c)
// Compiled with -ffp-contract=off: products and sums round separately like the x86 build of PIL, results are bit-identical.
// Whole volumes stay resident in HBM (40 MB each), a batch of slices costs three launches.
#include "common.h"

namespace {

constexpr int NORM_BLOCKS = 256;

// stage 1: per-block partial sums of (x - shift) and (x - shift)^2; shift = 0 on the first pass (mean), = mean on the second
__global__ __launch_bounds__(256) void vol_partial_kernel(const double *__restrict__ v, int64_t n, const double *shift, double *part)
{
    __shared__ double rs[256], rq[256];
    const double m = shift ? *shift : 0.0;
    double s = 0.0, q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double d = v[i] - m;
        s += d;
        q += d * d;
    }
    rs[threadIdx.x] = s;
    rq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rq[threadIdx.x] += rq[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = rs[0]; part[blockIdx.x * 2 + 1] = rq[0]; }
}

// stage 2 (one thread): pass 0 -> st[0] = mean; pass 1 -> st[1] = std, st[2] = lo, st[3] = hi
__global__ void vol_stat_kernel(const double *part, int nblk, int64_t n, double *st, int pass)
{
    double s = 0.0, q = 0.0;
    for (int i = 0; i < nblk; ++i) { s += part[i * 2]; q += part[i * 2 + 1]; }
    if (pass == 0) {
        st[0] = s / (double)n;
    } else {
        const double sd = sqrt(q / (double)n);           // np.std: population standard deviation
        st[1] = sd;
        st[2] = st[0] - 1 * sd;                          // dataset.py:588  img_range = (mean - 1*std, mean + 2*std)
        st[3] = st[0] + 2 * sd;
    }
}

__global__ __launch_bounds__(256) void vol_apply_kernel(const double *__restrict__ v, int64_t n, const double *st, float *__restrict__ out)
{
    const double lo = st[2], hi = st[3], range = hi - lo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double x = v[i];
        x = x < lo ? lo : (x > hi ? hi : x);             // np.clip
        out[i] = (float)(x / range);                     // :590, then astype(float32) (:592)
    }
}

// slice cut + optional affine (nearest, fixed point) + centre crop with zero padding
__global__ __launch_bounds__(256) void slice_prepare_kernel(const anoddpm_mri_slice_args a)
{
    const int b = blockIdx.y;
    const int n = a.crop * a.crop;
    const float *vol = a.vols[b];
    const int Y = a.ydim[b], sl = a.slice_idx[b];
    const int H = a.X, W = a.Z;                          // the slice as an image: rows = first axis, columns = third axis
    const long long *m = a.affine ? a.affine + (long long)b * 6 : nullptr;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int oy = i / a.crop, ox = i % a.crop;
        int y = oy + a.crop_top, x = ox - a.pad_left;    // position in the (affine-transformed) H x W image
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            bool ok = true;
            if (m) {                                     // PIL affine_fixed: xx = a2 + a0*x + a1*y (16.16), source = xx >> 16
                const long long xx = m[2] + m[0] * x + m[1] * y, yy = m[5] + m[3] * x + m[4] * y;
                const long long xi = xx >> 16, yi = yy >> 16;
                ok = xi >= 0 && xi < W && yi >= 0 && yi < H;
                x = (int)xi;
                y = (int)yi;
            }
            if (ok) v = vol[((long long)y * Y + sl) * W + x];
        }
        a.out[(long long)b * n + i] = v;
    }
}

// one pass of PIL's separable resample: out[b][r][o] = (float) sum_k (double)in[b][r][kmin[o] + k] * coef[o][k]
// (vertical == 1: the same along rows).  POST: (v - mean) / std in fp32 (torchvision Normalize) on the final pass.
template <bool VERTICAL, bool POST>
__global__ __launch_bounds__(256) void resample_pass_kernel(const float *__restrict__ in, float *__restrict__ out, const double *__restrict__ coef,
                                                            const int32_t *__restrict__ kmin, const int32_t *__restrict__ kn, int kmax,
                                                            int in_h, int in_w, int out_h, int out_w, float mean, float std)
{
    const int b = blockIdx.y;
    const int n = out_h * out_w;
    const float *src = in + (long long)b * in_h * in_w;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int r = i / out_w, c = i % out_w;
        const int o = VERTICAL ? r : c;
        const int k0 = kmin[o], cnt = kn[o];
        const double *k = coef + (long long)o * kmax;
        double ss = 0.0;
        for (int t = 0; t < cnt; ++t) {
            const float p = VERTICAL ? src[(long long)(k0 + t) * in_w + c] : src[(long long)r * in_w + k0 + t];
            ss += (double)p * k[t];
        }
        float v = (float)ss;
        if (POST) v = (v - mean) / std;
        out[(long long)b * n + i] = v;
    }
}

}  // namespace

using namespace anoddpm;

extern "C" int anoddpm_volume_normalise(const double *vol, int64_t n, float *out, double *workspace, void *stream)
{
    ANODDPM_REQUIRE(vol && out && workspace && n >= 1, "volume_normalise: bad arguments");
    hipStream_t s = as_stream(stream);
    double *part = workspace, *st = workspace + 2 * NORM_BLOCKS;
    hipLaunchKernelGGL(vol_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, s, vol, n, (const double *)nullptr, part);
    hipLaunchKernelGGL(vol_stat_kernel, dim3(1), dim3(1), 0, s, part, NORM_BLOCKS, n, st, 0);
    hipLaunchKernelGGL(vol_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, s, vol, n, (const double *)st, part);
    hipLaunchKernelGGL(vol_stat_kernel, dim3(1), dim3(1), 0, s, part, NORM_BLOCKS, n, st, 1);
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(vol_apply_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, vol, n, (const double *)st, out);
    return check_launch("volume_normalise");
}

extern "C" int anoddpm_mri_slice_prepare(const anoddpm_mri_slice_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->vols && a->ydim && a->slice_idx && a->out, "mri_slice_prepare: null pointer");
    ANODDPM_REQUIRE(a->B >= 1 && a->B <= 65535 && a->X >= 1 && a->Z >= 1 && a->crop >= 1, "mri_slice_prepare: bad sizes");
    const int n = a->crop * a->crop;
    hipLaunchKernelGGL(slice_prepare_kernel, dim3((n + 255) / 256, a->B), dim3(256), 0, as_stream(stream), *a);
    return check_launch("mri_slice_prepare");
}

extern "C" int anoddpm_resize_bilinear_pil(const anoddpm_resize_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->in && a->tmp && a->out && a->kx && a->kx_min && a->kx_n && a->ky && a->ky_min && a->ky_n, "resize_bilinear_pil: null pointer");
    ANODDPM_REQUIRE(a->B >= 1 && a->B <= 65535 && a->in_h >= 1 && a->in_w >= 1 && a->out_h >= 1 && a->out_w >= 1 && a->kmax_x >= 1 && a->kmax_y >= 1,
                    "resize_bilinear_pil: bad sizes");
    ANODDPM_REQUIRE(!a->normalize || a->std != 0.0f, "resize_bilinear_pil: std must be non-zero");
    hipStream_t s = as_stream(stream);
    const int n1 = a->in_h * a->out_w, n2 = a->out_h * a->out_w;
    hipLaunchKernelGGL((resample_pass_kernel<false, false>), dim3((n1 + 255) / 256, a->B), dim3(256), 0, s, a->in, a->tmp, a->kx, a->kx_min, a->kx_n,
                       a->kmax_x, a->in_h, a->in_w, a->in_h, a->out_w, 0.f, 1.f);
    if (a->normalize)
        hipLaunchKernelGGL((resample_pass_kernel<true, true>), dim3((n2 + 255) / 256, a->B), dim3(256), 0, s, a->tmp, a->out, a->ky, a->ky_min, a->ky_n,
                           a->kmax_y, a->in_h, a->out_w, a->out_h, a->out_w, a->mean, a->std);
    else
        hipLaunchKernelGGL((resample_pass_kernel<true, false>), dim3((n2 + 255) / 256, a->B), dim3(256), 0, s, a->tmp, a->out, a->ky, a->ky_min, a->ky_n,
                           a->kmax_y, a->in_h, a->out_w, a->out_h, a->out_w, 0.f, 1.f);
    return check_launch("resize_bilinear_pil");
}
