// Backward of a = SiLU(GroupNorm32(x)) as the forward's fused operand load consumed it (UNet.py:170-171, 190-191, 113,
// 409-411; reference: torch autograd of nn.GroupNorm + nn.SiLU, diffusion_training.py:102).  HBM-bound: the reduction
// pass reads x and da once, the elementwise pass reads them again and writes dx.  Two concatenated sources, the
// nearest-x2 / 2x2-average resampling between a and the conv, and gradient fan-in (acc_dx) are handled in place,
// so neither the activated tensor nor a resampled gradient ever exists in HBM.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// gradient w.r.t. the activated value at SOURCE pixel (sy, sx), channel quad at `c`
__device__ __forceinline__ f32x4 load_da(const anoddpm_gn_bwd_args &a, const float *da, int sy, int sx, int c)
{
    if (a.a_mode == 0) return *reinterpret_cast<const f32x4 *>(da + ((int64_t)sy * a.Ws + sx) * a.da_ld + c);
    if (a.a_mode == 1) {                             // forward: nearest x2 -> sum of the four children
        const int Wd = a.Ws * 2;
        const float *p = da + ((int64_t)(2 * sy) * Wd + 2 * sx) * a.da_ld + c;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(p), v1 = *reinterpret_cast<const f32x4 *>(p + a.da_ld);
        const f32x4 v2 = *reinterpret_cast<const f32x4 *>(p + (int64_t)Wd * a.da_ld);
        const f32x4 v3 = *reinterpret_cast<const f32x4 *>(p + (int64_t)Wd * a.da_ld + a.da_ld);
        return (v0 + v1) + (v2 + v3);
    }
    const int Wd = a.Ws >> 1;                        // forward: 2x2 average -> a quarter of the parent
    const f32x4 v = *reinterpret_cast<const f32x4 *>(da + ((int64_t)(sy >> 1) * Wd + (sx >> 1)) * a.da_ld + c);
    return v * 0.25f;
}

struct ChanParams { f32x4 sc, sh, mu, rs; };

// y = sc*x + sh (sc = gamma*rstd);  dy = da * silu'(y);  xhat = (x - mu)*rs
template <bool ACT>
__device__ __forceinline__ void dy_xhat(const ChanParams &k, f32x4 x, f32x4 g, f32x4 &dy, f32x4 &xh)
{
    xh = (x - k.mu) * k.rs;
    if (ACT) {
        const f32x4 y = x * k.sc + k.sh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-y[e]));
            dy[e] = g[e] * (s * (1.0f + y[e] * (1.0f - s)));
        }
    } else {
        dy = g;
    }
}

__device__ __forceinline__ ChanParams chan_params(const anoddpm_gn_bwd_args &a, int b, int c)
{
    const int C = a.c0 + a.c1, cpg = C / a.groups;
    ChanParams k;
    const f32x4 gm = *reinterpret_cast<const f32x4 *>(a.gamma + c), bt = *reinterpret_cast<const f32x4 *>(a.beta + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int g = (c + e) / cpg;
        k.mu[e] = a.mean[(int64_t)b * a.groups + g];
        k.rs[e] = a.rstd[(int64_t)b * a.groups + g];
    }
    k.sc = gm * k.rs;
    k.sh = bt - k.mu * k.sc;
    return k;
}

constexpr int RED_UNROLL = 4;

// pass 1: grid (nslab, B).  partial[b][slab][c] = { sum_p dy, sum_p dy*xhat } over the slab's pixels.
// The common case -- no resampling between the activation and the conv (A_MODE 0) -- walks its streams with pointer increments
// (the generic form pays a 64-bit address and an integer division per 16-byte load).  Stashing dy = da * silu'(y) over da for
// pass 3 was measured and dropped: the passes are HBM-bound (up to five streams in pass 3), not bound by the derivative's
// v_exp + v_rcp, and the extra 134 MB write per 256^2 layer cost 0.4 ms per step.
template <int A_MODE, bool ACT>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(anoddpm_gn_bwd_args a)
{
    __shared__ float lds_s[256 * 4];
    __shared__ float lds_q[256 * 4];
    const int C = a.c0 + a.c1, C4 = C >> 2, P = a.Hs * a.Ws;
    const int TQ = C4 < 256 ? C4 : 256;
    const int R = 256 / TQ;
    const int npass = (C4 + TQ - 1) / TQ;
    const int tid = threadIdx.x;
    const int tq = tid % TQ, tr = tid / TQ;
    const int b = blockIdx.y, slab = blockIdx.x;
    const int sp = (P + a.nslab - 1) / a.nslab;
    const int p0 = slab * sp, p1 = (p0 + sp < P) ? p0 + sp : P;
    double *out = a.partial + ((int64_t)b * a.nslab + slab) * C * 2;
    const float *da = a.da + (int64_t)b * a.da_bs;
    for (int pass = 0; pass < npass; ++pass) {
        const int quad = pass * TQ + tq;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
        if (tr < R && quad < C4) {
            const int c = quad * 4;
            const ChanParams k = chan_params(a, b, c);
            const bool first = c < a.c0;
            const float *src = first ? a.x0 + (int64_t)b * a.x0_bs + c : a.x1 + (int64_t)b * a.x1_bs + (c - a.c0);
            const int ld = first ? a.x0_ld : a.x1_ld;
            int p = p0 + tr;
            if (A_MODE == 0) {
                const float *xp = src + (int64_t)p * ld;
                const float *gp = da + (int64_t)p * a.da_ld + c;
                const int64_t sx = (int64_t)R * ld, sg = (int64_t)R * a.da_ld;
                for (; p + (RED_UNROLL - 1) * R < p1; p += RED_UNROLL * R, xp += RED_UNROLL * sx, gp += RED_UNROLL * sg) {
                    f32x4 x[RED_UNROLL], g[RED_UNROLL];
#pragma unroll
                    for (int u = 0; u < RED_UNROLL; ++u) {
                        x[u] = *reinterpret_cast<const f32x4 *>(xp + u * sx);
                        g[u] = *reinterpret_cast<const f32x4 *>(gp + u * sg);
                    }
#pragma unroll
                    for (int u = 0; u < RED_UNROLL; ++u) {
                        f32x4 dy, xh;
                        dy_xhat<ACT>(k, x[u], g[u], dy, xh);
                        s += dy;
                        q += dy * xh;
                    }
                }
                for (; p < p1; p += R, xp += sx, gp += sg) {
                    f32x4 dy, xh;
                    dy_xhat<ACT>(k, *reinterpret_cast<const f32x4 *>(xp), *reinterpret_cast<const f32x4 *>(gp), dy, xh);
                    s += dy;
                    q += dy * xh;
                }
            } else {
                for (; p < p1; p += R) {
                    const f32x4 x = *reinterpret_cast<const f32x4 *>(src + (int64_t)p * ld);
                    const f32x4 g = load_da(a, da, p / a.Ws, p % a.Ws, c);
                    f32x4 dy, xh;
                    dy_xhat<ACT>(k, x, g, dy, xh);
                    s += dy;
                    q += dy * xh;
                }
            }
        }
        if (tr < R) {
            const int o = (tr * TQ + tq) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lds_s[o + e] = s[e]; lds_q[o + e] = q[e]; }
        }
        __syncthreads();
        for (int cl = tid; cl < TQ * 4; cl += 256) {
            const int c = pass * TQ * 4 + cl;
            if (c < C) {
                double ss = 0.0, qq = 0.0;
                for (int r = 0; r < R; ++r) { ss += (double)lds_s[r * TQ * 4 + cl]; qq += (double)lds_q[r * TQ * 4 + cl]; }
                out[c * 2] = ss;
                out[c * 2 + 1] = qq;
            }
        }
        __syncthreads();
    }
}

// pass 2: grid (groups), 256 threads.  Folds the slabs per image (thread = (image, channel, slab lane), fixed order); per
// image the group means, per channel dgamma / dbeta (images summed in index order).  Up to 256 / cpg images are folded side
// by side -- the kernel is a latency chain over tiny data, so the images of a batch must not be walked one after the other.
// coef[b][c] = { rstd*gamma, rstd*mean_g(gamma*dy), rstd*mean_g(gamma*dy*xhat), unused }
__global__ __launch_bounds__(256) void gn_bwd_fold_kernel(anoddpm_gn_bwd_args a)
{
    __shared__ double red[256][2];
    __shared__ double chan[256][2];                                  // [image slot * cpg + channel] per-channel sums
    __shared__ double grp[64][2];                                    // [image slot] group sums of gamma * c
    const int C = a.c0 + a.c1, cpg = C / a.groups;                  // cpg <= 64 (launcher)
    const int g = blockIdx.x, tid = threadIdx.x;
    int nbp = 256 / cpg;                                             // images folded side by side
    if (nbp > a.B) nbp = a.B;
    if (nbp > 64) nbp = 64;
    const int per = 256 / nbp;                                       // threads per image
    const int S = per / cpg;                                         // slab lanes per (image, channel)
    const int bl = tid / per, rem = tid - bl * per;
    const int cl = rem % cpg, sl = rem / cpg;
    const double n = (double)a.Hs * a.Ws * cpg;
    double dgam = 0.0, dbet = 0.0;
    for (int b0 = 0; b0 < a.B; b0 += nbp) {
        const int b = b0 + bl;
        const bool live = bl < nbp && b < a.B && sl < S;
        double s1 = 0.0, s2 = 0.0;
        if (live) {
            // eight slab rows requested together, added in slab order: the kernel is a latency chain (16 rows per thread at 256 slabs,
            // one dependent ~0.5 us round trip each when the loop was left rolled: 9 us per launch, 96 launches per training step)
            const double *p0 = a.partial + ((int64_t)b * a.nslab * C + g * cpg + cl) * 2;
            const int64_t rs = (int64_t)C * 2;
            int k = sl;
            for (; k + 7 * S < a.nslab; k += 8 * S) {
                double2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const double2 *>(p0 + (int64_t)(k + j * S) * rs);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1 += v[j].x; s2 += v[j].y; }
            }
            for (; k < a.nslab; k += S) {
                const double2 v = *reinterpret_cast<const double2 *>(p0 + (int64_t)k * rs);
                s1 += v.x;
                s2 += v.y;
            }
        }
        red[tid][0] = s1;
        red[tid][1] = s2;
        __syncthreads();
        if (live && sl == 0) {                                       // per (image, channel): the slab lanes in order
            double c1 = 0.0, c2 = 0.0;
            for (int k = 0; k < S; ++k) { c1 += red[bl * per + k * cpg + cl][0]; c2 += red[bl * per + k * cpg + cl][1]; }
            chan[bl * cpg + cl][0] = c1;
            chan[bl * cpg + cl][1] = c2;
        }
        __syncthreads();
        if (live && sl == 0 && cl == 0) {                            // per image: the group sums, channels in order
            double g1 = 0.0, g2 = 0.0;
            for (int c = 0; c < cpg; ++c) {
                const double gm = (double)a.gamma[g * cpg + c];
                g1 += chan[bl * cpg + c][0] * gm;
                g2 += chan[bl * cpg + c][1] * gm;
            }
            grp[bl][0] = g1;
            grp[bl][1] = g2;
        }
        __syncthreads();
        if (live && sl == 0) {
            const int cc = g * cpg + cl;
            const double gm = (double)a.gamma[cc];
            const double r = (double)a.rstd[(int64_t)b * a.groups + g];
            float *k = a.coef + ((int64_t)b * C + cc) * 4;
            k[0] = (float)(r * gm);
            k[1] = (float)(r * grp[bl][0] / n);
            k[2] = (float)(r * grp[bl][1] / n);
            k[3] = 0.f;
        }
        if (tid < cpg) {                                             // dgamma / dbeta: this chunk's images in index order
            const int lim = (a.B - b0) < nbp ? (a.B - b0) : nbp;
            for (int i = 0; i < lim; ++i) { dbet += chan[i * cpg + tid][0]; dgam += chan[i * cpg + tid][1]; }
        }
        __syncthreads();
    }
    if (tid < cpg) {
        a.dgamma[g * cpg + tid] += (float)dgam;
        a.dbeta[g * cpg + tid] += (float)dbet;
    }
}

// pass 3: dx = coef0*dy - coef1 - xhat*coef2, written (or added) to the source gradients.  A workgroup owns a CONTIGUOUS slab of
// pixels (tools/hbm_patterns.hip: interleaved ownership with many workgroups per CU loses a quarter of the write bandwidth), a
// thread keeps one channel quad -- its 28 per-channel constants are loaded once, not per 16 bytes of payload -- and walks the
// slab with pointer increments, APPLY_UNROLL independent 16-byte loads per stream in flight.  grid (slabs, B).
constexpr int APPLY_UNROLL = 4;

template <int A_MODE, bool ACT>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(anoddpm_gn_bwd_args a, int sp)
{
    const int C = a.c0 + a.c1, C4 = C >> 2, P = a.Hs * a.Ws;
    const int TQ = C4 < 256 ? C4 : 256;
    const int R = 256 / TQ;
    const int npass = (C4 + TQ - 1) / TQ;
    const int tid = threadIdx.x;
    const int tq = tid % TQ, tr = tid / TQ;
    if (tr >= R) return;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * sp, p1 = (p0 + sp < P) ? p0 + sp : P;
    const float *da = a.da + (int64_t)b * a.da_bs;
    for (int pass = 0; pass < npass; ++pass) {
        const int quad = pass * TQ + tq;
        if (quad >= C4) break;
        const int c = quad * 4;
        const ChanParams k = chan_params(a, b, c);
        const bool first = c < a.c0;
        const float *src = first ? a.x0 + (int64_t)b * a.x0_bs + c : a.x1 + (int64_t)b * a.x1_bs + (c - a.c0);
        const int ld = first ? a.x0_ld : a.x1_ld;
        float *dst0 = first ? a.dx0 + (int64_t)b * a.dx0_bs + c : a.dx1 + (int64_t)b * a.dx1_bs + (c - a.c0);
        const int dld = first ? a.dx0_ld : a.dx1_ld;
        const float *res = a.dres ? a.dres + (int64_t)b * a.dres_bs + c : nullptr;
        const bool acc = (a.acc_dx & (first ? 1 : 2)) != 0;
        const float *kc = a.coef + ((int64_t)b * C + c) * 4;
        f32x4 k0, k1, k2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { k0[e] = kc[e * 4 + 0]; k1[e] = kc[e * 4 + 1]; k2[e] = kc[e * 4 + 2]; }
        int p = p0 + tr;
        const float *xp = src + (int64_t)p * ld;
        const float *gp = da + (int64_t)p * a.da_ld + c;                 // A_MODE 0 only
        const float *rp = res ? res + (int64_t)p * a.dres_ld : nullptr;
        float *op = dst0 + (int64_t)p * dld;
        const int64_t sx = (int64_t)R * ld, sg = (int64_t)R * a.da_ld, sr = (int64_t)R * a.dres_ld, so = (int64_t)R * dld;
        for (; p + (APPLY_UNROLL - 1) * R < p1; p += APPLY_UNROLL * R) {
            f32x4 x[APPLY_UNROLL], g[APPLY_UNROLL], r[APPLY_UNROLL], o[APPLY_UNROLL];
#pragma unroll
            for (int u = 0; u < APPLY_UNROLL; ++u) {
                x[u] = *reinterpret_cast<const f32x4 *>(xp + u * sx);
                if (A_MODE == 0) g[u] = *reinterpret_cast<const f32x4 *>(gp + u * sg);
                else { const int pp = p + u * R; g[u] = load_da(a, da, pp / a.Ws, pp % a.Ws, c); }
                if (res) r[u] = *reinterpret_cast<const f32x4 *>(rp + u * sr);
                if (acc) o[u] = *reinterpret_cast<const f32x4 *>(op + u * so);
            }
#pragma unroll
            for (int u = 0; u < APPLY_UNROLL; ++u) {
                f32x4 dy, xh;
                dy_xhat<ACT>(k, x[u], g[u], dy, xh);
                f32x4 dx = k0 * dy - k1 - xh * k2;
                if (res) dx += r[u];
                if (acc) dx += o[u];
                *reinterpret_cast<f32x4 *>(op + u * so) = dx;
            }
            xp += APPLY_UNROLL * sx; gp += APPLY_UNROLL * sg; op += APPLY_UNROLL * so;
            if (res) rp += APPLY_UNROLL * sr;
        }
        for (; p < p1; p += R, xp += sx, gp += sg, op += so) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(xp);
            const f32x4 g = A_MODE == 0 ? *reinterpret_cast<const f32x4 *>(gp) : load_da(a, da, p / a.Ws, p % a.Ws, c);
            f32x4 dy, xh;
            dy_xhat<ACT>(k, x, g, dy, xh);
            f32x4 dx = k0 * dy - k1 - xh * k2;
            if (res) { dx += *reinterpret_cast<const f32x4 *>(rp); rp += sr; }
            if (acc) dx += *reinterpret_cast<const f32x4 *>(op);
            *reinterpret_cast<f32x4 *>(op) = dx;
        }
    }
}

template <int A_MODE, bool ACT>
void launch_reduce_apply(const anoddpm_gn_bwd_args *a, hipStream_t s, int slabs, int sp)
{
    if (!a->partial_ready)                                            // else: written by the epilogue of the data-gradient launch
        hipLaunchKernelGGL((gn_bwd_reduce_kernel<A_MODE, ACT>), dim3(a->nslab, a->B), dim3(256), 0, s, *a);
    hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3(a->groups), dim3(256), 0, s, *a);
    hipLaunchKernelGGL((gn_bwd_apply_kernel<A_MODE, ACT>), dim3((unsigned)slabs, a->B), dim3(256), 0, s, *a, sp);
}

}  // namespace

extern "C" int anoddpm_gn_silu_backward(const anoddpm_gn_bwd_args *a, void *stream)
{
    using namespace anoddpm;
    ANODDPM_REQUIRE(a && a->x0 && a->da && a->gamma && a->beta && a->mean && a->rstd && a->dx0 && a->dgamma && a->dbeta && a->partial && a->coef,
                    "gn_silu_backward: null pointer");
    const int C = a->c0 + a->c1;
    ANODDPM_REQUIRE(a->c0 > 0 && a->c0 % 4 == 0 && a->c1 >= 0 && a->c1 % 4 == 0 && (a->c1 == 0 || (a->x1 && a->dx1)), "gn_silu_backward: bad channel counts");
    ANODDPM_REQUIRE(a->groups > 0 && C % a->groups == 0 && C / a->groups <= 64, "gn_silu_backward: bad group size");
    ANODDPM_REQUIRE(a->B > 0 && a->B <= 65535 && a->Hs > 0 && a->Ws > 0 && a->nslab > 0 && a->nslab <= 65535, "gn_silu_backward: bad sizes");
    ANODDPM_REQUIRE(!a->partial_ready || (a->a_mode == 0 && a->act == 1), "gn_silu_backward: partial_ready needs a_mode 0 and the SiLU form");
    ANODDPM_REQUIRE(a->a_mode >= 0 && a->a_mode <= 2 && (a->a_mode != 2 || (a->Hs % 2 == 0 && a->Ws % 2 == 0)), "gn_silu_backward: bad a_mode");
    ANODDPM_REQUIRE(!a->dres || (a->dres_ld % 4 == 0 && a->dres_bs % 4 == 0), "gn_silu_backward: dres strides must be multiples of 4 floats");
    ANODDPM_REQUIRE(a->x0_ld % 4 == 0 && a->da_ld % 4 == 0 && a->dx0_ld % 4 == 0 && (a->c1 == 0 || (a->x1_ld % 4 == 0 && a->dx1_ld % 4 == 0)),
                    "gn_silu_backward: pixel strides must be multiples of 4 floats");
    hipStream_t s = as_stream(stream);
    // apply: contiguous pixel slabs, ~8 workgroups per CU over the batch, at least one unrolled trip per thread
    const int P = a->Hs * a->Ws, C4 = C / 4;
    const int R = C4 < 256 ? 256 / C4 : 1;
    int sp = (int)(((int64_t)P * a->B + 2047) / 2048);
    const int min_sp = R * APPLY_UNROLL;
    sp = ((sp + min_sp - 1) / min_sp) * min_sp;
    const int slabs = (P + sp - 1) / sp;
    if (a->a_mode == 0) { if (a->act) launch_reduce_apply<0, true>(a, s, slabs, sp); else launch_reduce_apply<0, false>(a, s, slabs, sp); }
    else if (a->a_mode == 1) { if (a->act) launch_reduce_apply<1, true>(a, s, slabs, sp); else launch_reduce_apply<1, false>(a, s, slabs, sp); }
    else { if (a->act) launch_reduce_apply<2, true>(a, s, slabs, sp); else launch_reduce_apply<2, false>(a, s, slabs, sp); }
    return check_launch("gn_silu_backward");
}
