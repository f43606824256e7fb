// Memory-bound UNet helper kernels for gfx950 (everything that is not the MFMA implicit GEMM):
// GroupNorm statistics -> per-channel affine, row softmax, 2x resampling, small-batch linear
// layers, sinusoidal timestep features, the Cin<=4 stem convolution and the NHWC->NCHW edge copy.
// All of them stream NHWC fp32 with 16-byte lanes; roofline = HBM.
#include "common.h"

using anoddpm::silu_f;

namespace {

// ---------------------------------------------------------------- GroupNorm statistics --------
// Stage 1: grid (nslab, B).  Threads are laid out (row, channel-quad); every thread keeps fp64
// sums for its 4 channels over its pixel rows, rows are combined through LDS in a fixed order
// (deterministic, no atomics), groups are reduced by 32 threads.  Stage 2 folds the slabs and
// emits scale/shift.  Two sources = torch.cat([h, skip], 1) read in place (UNet.py:402).
__global__ __launch_bounds__(256) void gn_partial_kernel(anoddpm_gn_args a)
{
    __shared__ double lds_s[256 * 4];
    __shared__ double lds_q[256 * 4];
    const int C = a.c0 + a.c1;
    const int C4 = C >> 2;
    const int TQ = C4 < 256 ? C4 : 256;
    const int R = 256 / TQ;
    const int npass = (C4 + TQ - 1) / TQ;
    const int cpg = C / a.groups;
    const int tid = threadIdx.x;
    const int tq = tid % TQ, tr = tid / TQ;
    const int b = blockIdx.y, slab = blockIdx.x;
    const int sp = (a.P + a.nslab - 1) / a.nslab;
    const int p0 = slab * sp;
    const int p1 = (p0 + sp < a.P) ? p0 + sp : a.P;
    double gs = 0.0, gq = 0.0;                       // group totals (threads 0..groups-1)

    for (int pass = 0; pass < npass; ++pass) {
        const int quad = pass * TQ + tq;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        if (tr < R && quad < C4) {
            const int c = quad * 4;
            const float *src;
            int ld;
            if (c < a.c0) { src = a.a0 + (int64_t)b * a.a0_bs + c; ld = a.a0_ld; }
            else          { src = a.a1 + (int64_t)b * a.a1_bs + (c - a.c0); ld = a.a1_ld; }
            for (int p = p0 + tr; p < p1; p += R) {
                const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)p * ld);
                const double d0 = v.x, d1 = v.y, d2 = v.z, d3 = v.w;
                s0 += d0; s1 += d1; s2 += d2; s3 += d3;
                q0 += d0 * d0; q1 += d1 * d1; q2 += d2 * d2; q3 += d3 * d3;
            }
        }
        if (tr < R) {
            const int o = (tr * TQ + tq) * 4;
            lds_s[o] = s0; lds_s[o + 1] = s1; lds_s[o + 2] = s2; lds_s[o + 3] = s3;
            lds_q[o] = q0; lds_q[o + 1] = q1; lds_q[o + 2] = q2; lds_q[o + 3] = q3;
        }
        __syncthreads();
        if (tid < a.groups) {
            const int clo = pass * TQ * 4, chi = clo + TQ * 4;
            int g0 = tid * cpg, g1 = g0 + cpg;
            g0 = g0 > clo ? g0 : clo;
            g1 = g1 < chi ? g1 : chi;
            for (int c = g0; c < g1; ++c)
                for (int r = 0; r < R; ++r) {
                    gs += lds_s[r * TQ * 4 + (c - clo)];
                    gq += lds_q[r * TQ * 4 + (c - clo)];
                }
        }
        __syncthreads();
    }
    if (tid < a.groups) {
        double *dst = a.partial + ((int64_t)(b * a.nslab + slab) * a.groups + tid) * 2;
        dst[0] = gs;
        dst[1] = gq;
    }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(anoddpm_gn_args a)
{
    __shared__ double red_s[256];
    __shared__ double red_q[256];
    __shared__ double mean_s[64];
    __shared__ double rstd_s[64];
    const int C = a.c0 + a.c1;
    const int cpg = C / a.groups;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    // slab partials are folded by all 256 threads: thread = (slab lane, group), fixed order -> deterministic
    const int lanes = 256 / a.groups;                  // slab lanes per group (groups <= 64)
    const int g = tid % a.groups, sl = tid / a.groups;
    double s = 0.0, q = 0.0;
    if (sl < lanes) {
        for (int k = sl; k < a.nslab; k += lanes) {
            const double *src = a.partial + ((int64_t)(b * a.nslab + k) * a.groups + g) * 2;
            s += src[0];
            q += src[1];
        }
    }
    red_s[tid] = s;
    red_q[tid] = q;
    __syncthreads();
    if (tid < a.groups) {
        double ts = 0.0, tq = 0.0;
        for (int l = 0; l < lanes; ++l) { ts += red_s[l * a.groups + tid]; tq += red_q[l * a.groups + tid]; }
        const double n = (double)a.P * cpg;
        const double mean = ts / n;
        double var = tq / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        mean_s[tid] = mean;
        rstd_s[tid] = 1.0 / sqrt(var + (double)a.eps);
    }
    __syncthreads();
    for (int c = tid; c < C; c += blockDim.x) {
        const int gg = c / cpg;
        const double sc = rstd_s[gg] * (double)a.gamma[c];
        a.scale[(int64_t)b * C + c] = (float)sc;
        a.shift[(int64_t)b * C + c] = (float)((double)a.beta[c] - mean_s[gg] * sc);
    }
}

// Per-channel partial sums (same format as the igemm epilogue's fused statistics).
__global__ __launch_bounds__(256) void chan_stats_kernel(anoddpm_chan_stats_args a)
{
    __shared__ float lds_s[256 * 4];
    __shared__ float lds_q[256 * 4];
    const int C4 = a.C >> 2;
    const int TQ = C4 < 256 ? C4 : 256;
    const int R = 256 / TQ;
    const int npass = (C4 + TQ - 1) / TQ;
    const int tid = threadIdx.x;
    const int tq = tid % TQ, tr = tid / TQ;
    const int b = blockIdx.y, slab = blockIdx.x;
    const int sp = (a.P + a.nslab - 1) / a.nslab;
    const int p0 = slab * sp;
    const int p1 = (p0 + sp < a.P) ? p0 + sp : a.P;
    float *out = a.stats + ((int64_t)b * a.nslab + slab) * a.C * 2;
    for (int pass = 0; pass < npass; ++pass) {
        const int quad = pass * TQ + tq;
        float s0 = 0, s1 = 0, s2 = 0, s3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        if (tr < R && quad < C4) {
            const float *src = a.a + (int64_t)b * a.a_bs + quad * 4;
            for (int p = p0 + tr; p < p1; p += R) {
                const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)p * a.a_ld);
                s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
                q0 += v.x * v.x; q1 += v.y * v.y; q2 += v.z * v.z; q3 += v.w * v.w;
            }
        }
        if (tr < R) {
            const int o = (tr * TQ + tq) * 4;
            lds_s[o] = s0; lds_s[o + 1] = s1; lds_s[o + 2] = s2; lds_s[o + 3] = s3;
            lds_q[o] = q0; lds_q[o + 1] = q1; lds_q[o + 2] = q2; lds_q[o + 3] = q3;
        }
        __syncthreads();
        for (int cl = tid; cl < TQ * 4; cl += 256) {
            const int c = pass * TQ * 4 + cl;
            if (c < a.C) {
                float s = 0.f, q = 0.f;
                for (int r = 0; r < R; ++r) { s += lds_s[r * TQ * 4 + cl]; q += lds_q[r * TQ * 4 + cl]; }
                out[c * 2] = s;
                out[c * 2 + 1] = q;
            }
        }
        __syncthreads();
    }
}

// grid (groups, B): fold per-channel partials of up to two sources into scale/shift for one group.
// A group's channels are contiguous inside a statistics row ({sum, sumsq} pairs), so a thread takes whole rows and
// reads the group's 2*cpg floats with 16-byte loads; fp64 accumulation, fixed order (thread-strided rows, then a
// shuffle tree, then the four wave results in order) -> deterministic.
__global__ __launch_bounds__(256) void gn_finalize2_kernel(anoddpm_gn_finalize_args a)
{
    __shared__ double red_s[4];
    __shared__ double red_q[4];
    const int C = a.c0 + a.c1;
    const int cpg = C / a.groups;
    const int g = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    const int cbeg = g * cpg;
    const bool one_source = (cbeg + cpg <= a.c0) || (cbeg >= a.c0);       // block-uniform
    if (a.fmt0 != 0 || (a.c1 && a.fmt1 != 0)) {
        // at least one source is a single row of fp64 sums (anoddpm_igemm_args.tail_csum): per channel, rows strided over threads
        for (int cc = 0; cc < cpg; ++cc) {
            const int c = cbeg + cc;
            const bool first = c < a.c0;
            const int fmt = first ? a.fmt0 : a.fmt1;
            const int cl = first ? c : c - a.c0, cw = first ? a.c0 : a.c1;
            if (fmt != 0) {
                if (tid == 0) {
                    const double *st = reinterpret_cast<const double *>(first ? a.stats0 : a.stats1) + ((int64_t)b * cw + cl) * 2;
                    s += st[0];
                    q += st[1];
                }
            } else {
                const float *st = first ? a.stats0 : a.stats1;
                const int rows = first ? a.rows0 : a.rows1;
                st += (int64_t)b * rows * cw * 2;
                for (int r = tid; r < rows; r += 256) {
                    const float2 v = *reinterpret_cast<const float2 *>(st + ((int64_t)r * cw + cl) * 2);
                    s += (double)v.x;
                    q += (double)v.y;
                }
            }
        }
    } else if (one_source && (cpg & 1) == 0 && (a.c0 & 1) == 0 && (a.c1 & 1) == 0) {
        // 2*cpg floats = cpg/2 float4 per row, 16-byte aligned
        const float *st;
        int rows, cl, cw;
        if (cbeg < a.c0) { st = a.stats0; rows = a.rows0; cl = cbeg; cw = a.c0; }
        else             { st = a.stats1; rows = a.rows1; cl = cbeg - a.c0; cw = a.c1; }
        st += (int64_t)b * rows * cw * 2 + (int64_t)cl * 2;
        const int nq = cpg >> 1;
#pragma unroll 4
        for (int r = tid; r < rows; r += 256) {                        // independent row loads: keep four in flight
            const float4 *p = reinterpret_cast<const float4 *>(st + (int64_t)r * cw * 2);
            for (int k = 0; k < nq; ++k) {
                const float4 v = p[k];
                s += (double)v.x + (double)v.z;
                q += (double)v.y + (double)v.w;
            }
        }
    } else {
        // a group that straddles the two sources of a virtual concat (e.g. 256 + 128 channels, 12 per group)
        for (int cc = 0; cc < cpg; ++cc) {
            const int c = cbeg + cc;
            const float *st;
            int rows, cl, cw;
            if (c < a.c0) { st = a.stats0; rows = a.rows0; cl = c; cw = a.c0; }
            else          { st = a.stats1; rows = a.rows1; cl = c - a.c0; cw = a.c1; }
            st += (int64_t)b * rows * cw * 2;
            for (int r = tid; r < rows; r += 256) {
                const float2 v = *reinterpret_cast<const float2 *>(st + ((int64_t)r * cw + cl) * 2);
                s += (double)v.x;
                q += (double)v.y;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
    if ((tid & 63) == 0) { red_s[tid >> 6] = s; red_q[tid >> 6] = q; }
    __syncthreads();
    const double ts = ((red_s[0] + red_s[1]) + red_s[2]) + red_s[3];
    const double tq = ((red_q[0] + red_q[1]) + red_q[2]) + red_q[3];
    const double n = (double)a.P * cpg;
    const double mean = ts / n;
    double var = tq / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + (double)a.eps);
    if (tid < cpg) {
        const int c = g * cpg + tid;
        const double sc = rstd * (double)a.gamma[c];
        a.scale[(int64_t)b * C + c] = (float)sc;
        a.shift[(int64_t)b * C + c] = (float)((double)a.beta[c] - mean * sc);
    }
    if (tid == 0 && a.mean_out) {
        a.mean_out[(int64_t)b * a.groups + g] = (float)mean;
        a.rstd_out[(int64_t)b * a.groups + g] = (float)rstd;
    }
}

// ---------------------------------------------------------------- softmax ---------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float *x, int64_t rows, int L)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float *r = x + row * L;
    float m = -INFINITY;
    for (int i = lane; i < L; i += 64) m = fmaxf(m, r[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int i = lane; i < L; i += 64) {
        const float e = __expf(r[i] - m);
        r[i] = e;
        s += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.0f / s;
    for (int i = lane; i < L; i += 64) r[i] *= inv;
}

// ---------------------------------------------------------------- 2x resample -----------------
__global__ __launch_bounds__(256) void resample2x_kernel(anoddpm_resample_args a)
{
    const int C4 = a.C >> 2;
    const bool up = a.mode == 1 || a.mode == 4;
    const int Ho = up ? a.H * 2 : a.H / 2;
    const int Wo = up ? a.W * 2 : a.W / 2;
    const int64_t total = (int64_t)a.B * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % C4);
        int64_t p = i / C4;
        const int xo = (int)(p % Wo);
        p /= Wo;
        const int yo = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const float4 *in = reinterpret_cast<const float4 *>(a.in) + (int64_t)b * a.H * a.W * C4 + q;
        float4 o;
        if (a.mode == 1) {
            o = in[((int64_t)(yo >> 1) * a.W + (xo >> 1)) * C4];
        } else if (a.mode == 4) {                                  // adjoint of the stride-2 pick: values on the even pixels, zeros elsewhere
            o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(yo & 1) && !(xo & 1)) o = in[((int64_t)(yo >> 1) * a.W + (xo >> 1)) * C4];
        } else if (a.mode == 3) {                                  // stride-2 pick: the even pixels of a stride-1 result
            o = in[((int64_t)(2 * yo) * a.W + 2 * xo) * C4];
        } else {
            const float4 v00 = in[((int64_t)(2 * yo) * a.W + 2 * xo) * C4];
            const float4 v01 = in[((int64_t)(2 * yo) * a.W + 2 * xo + 1) * C4];
            const float4 v10 = in[((int64_t)(2 * yo + 1) * a.W + 2 * xo) * C4];
            const float4 v11 = in[((int64_t)(2 * yo + 1) * a.W + 2 * xo + 1) * C4];
            o.x = (((v00.x + v01.x) + v10.x) + v11.x) * 0.25f;
            o.y = (((v00.y + v01.y) + v10.y) + v11.y) * 0.25f;
            o.z = (((v00.z + v01.z) + v10.z) + v11.z) * 0.25f;
            o.w = (((v00.w + v01.w) + v10.w) + v11.w) * 0.25f;
            if (a.out_act) {                                       // the same four pixels, activated first (igemm a_mode 2 order)
                const float4 sc = reinterpret_cast<const float4 *>(a.gn_scale)[(int64_t)b * C4 + q];
                const float4 sh = reinterpret_cast<const float4 *>(a.gn_shift)[(int64_t)b * C4 + q];
                auto act4 = [&](float4 v) {
                    v.x = silu_f(v.x * sc.x + sh.x); v.y = silu_f(v.y * sc.y + sh.y);
                    v.z = silu_f(v.z * sc.z + sh.z); v.w = silu_f(v.w * sc.w + sh.w);
                    return v;
                };
                const float4 a00 = act4(v00), a01 = act4(v01), a10 = act4(v10), a11 = act4(v11);
                float4 oa;
                oa.x = (((a00.x + a01.x) + a10.x) + a11.x) * 0.25f;
                oa.y = (((a00.y + a01.y) + a10.y) + a11.y) * 0.25f;
                oa.z = (((a00.z + a01.z) + a10.z) + a11.z) * 0.25f;
                oa.w = (((a00.w + a01.w) + a10.w) + a11.w) * 0.25f;
                reinterpret_cast<float4 *>(a.out_act)[i] = oa;
            }
        }
        if (a.scale != 0.0f && a.scale != 1.0f) { o.x *= a.scale; o.y *= a.scale; o.z *= a.scale; o.w *= a.scale; }
        if (a.accumulate) {                                    // gradient fan-in (training backward)
            const float4 c = reinterpret_cast<const float4 *>(a.out)[i];
            o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
        }
        reinterpret_cast<float4 *>(a.out)[i] = o;
    }
}

// ---------------------------------------------------------------- small-batch linear ----------
// One wave per output feature; lanes stride over K in float4; up to 16 batch rows in registers.
template <int NB>
__global__ __launch_bounds__(256) void linear_small_kernel(anoddpm_linear_args a)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.N) return;
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    const float4 *w = reinterpret_cast<const float4 *>(a.w + (int64_t)n * a.K);
    const int K4 = a.K >> 2;
    for (int k = lane; k < K4; k += 64) {
        const float4 wv = w[k];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b < a.B) {
                float4 xv = reinterpret_cast<const float4 *>(a.in + (int64_t)b * a.K)[k];
                if (a.act_in) { xv.x = silu_f(xv.x); xv.y = silu_f(xv.y); xv.z = silu_f(xv.z); xv.w = silu_f(xv.w); }
                acc[b] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[b] += __shfl_xor(acc[b], o);
    }
    if (lane == 0) {
        const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b < a.B) {
                float v = acc[b] + bias;
                if (a.act_out) v = silu_f(v);
                a.out[(int64_t)b * a.N + n] = v;
            }
        }
    }
}

// ---------------------------------------------------------------- timestep features -----------
__global__ void posemb_kernel(anoddpm_posemb_args a)
{
    const int half = a.dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // side job (round 6): clear the plan's statistics accumulators (anoddpm_igemm_args.stats_csum), 16 bytes per thread and trip
    for (int64_t z = (int64_t)i * 2; z < a.zero_doubles; z += (int64_t)gridDim.x * blockDim.x * 2) {
        a.zero[z] = 0.0;
        if (z + 1 < a.zero_doubles) a.zero[z + 1] = 0.0;
    }
    if (i >= a.B * half) return;
    const int b = i / half, j = i % half;
    const float arg = ((float)a.t[b] * a.scale) * a.freqs[j];
    a.out[(int64_t)b * a.dim + j] = sinf(arg);
    a.out[(int64_t)b * a.dim + half + j] = cosf(arg);
}

// ---------------------------------------------------------------- stem conv -------------------
// NCHW (Cin <= 4) -> NHWC Cout, 3x3 pad 1.  Write-bound (512 B per pixel at Cout = 128).  Thread = (4 output channels, a strip
// of STEM_STRIP pixels along x): its 9 * Cin weight float4 are loaded ONCE into registers, the three input rows of the strip
// slide through registers, and every store is 16 bytes with the Cout/4 lanes of a pixel contiguous (full 128-byte lines).
// Cin > 2 keeps the generic per-pixel form (the reference's MRI models have Cin = 1).
constexpr int STEM_STRIP = 8;

// A workgroup owns `iters` x (256 / (Cout/4)) consecutive strips = one contiguous range of the NHWC output (32 KB per trip at
// Cout = 128).  With a.stats the kernel also emits the GroupNorm partial sums of its range --
// one row {sum, sum of squares} per channel, the format of the contraction kernels' epilogues -- so that the stem output is not
// read back by a statistics pass (134 MB at 256^2 x 128 x batch 4).
// TWO: a workgroup owns two consecutive trips and requests the input rows of BOTH before it computes the first (round 6: half the
// workgroups, statistics rows and weight loads; the serialised loads of the LOOP form -- 38 us -- are what made more than one
// trip per workgroup lose in round 3)
template <int CIN>
__global__ __launch_bounds__(256) void conv_stem_strip2_kernel(anoddpm_stem_args a)
{
    __shared__ float lds_s[256 * 4];
    __shared__ float lds_q[256 * 4];
    const int QP = a.Cout >> 2;
    const int spb = 256 / QP;
    const int q = threadIdx.x % QP;
    const int sl = threadIdx.x / QP;
    const bool lane_on = sl < spb;
    const int strips_x = a.W / STEM_STRIP;
    const int64_t nstrips = (int64_t)a.B * a.H * strips_x;
    const int64_t strip0 = (int64_t)blockIdx.x * spb * 2;
    const float4 *w = reinterpret_cast<const float4 *>(a.w);
    float4 wr[9 * CIN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) wr[t * CIN + ci] = w[(t * CIN + ci) * QP + (lane_on ? q : 0)];
    const float4 bias = a.bias ? reinterpret_cast<const float4 *>(a.bias)[lane_on ? q : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = make_float4(0.f, 0.f, 0.f, 0.f);
    float in[2][CIN][3][STEM_STRIP + 2];
    float4 *outp[2];
    bool on[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int64_t strip = strip0 + (int64_t)it * spb + sl;
        on[it] = lane_on && strip < nstrips;
        if (!on[it]) strip = 0;                                     // clamped: loaded and discarded
        const int sx = (int)(strip % strips_x);
        const int y = (int)((strip / strips_x) % a.H);
        const int b = (int)(strip / ((int64_t)strips_x * a.H));
        const int x0 = sx * STEM_STRIP;
        const bool has_l = x0 > 0, has_r = x0 + STEM_STRIP < a.W;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const float *plane = a.x + ((int64_t)b * CIN + ci) * a.H * a.W;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = y + dy - 1;
                const bool yin = yy >= 0 && yy < a.H;
                const float *row = plane + (int64_t)(yin ? yy : y) * a.W + x0;
                const float4 m0 = *reinterpret_cast<const float4 *>(row), m1 = *reinterpret_cast<const float4 *>(row + 4);
                const float l = row[has_l ? -1 : 0], r = row[has_r ? STEM_STRIP : STEM_STRIP - 1];
                in[it][ci][dy][0] = (yin && has_l) ? l : 0.f;
                in[it][ci][dy][1] = yin ? m0.x : 0.f; in[it][ci][dy][2] = yin ? m0.y : 0.f;
                in[it][ci][dy][3] = yin ? m0.z : 0.f; in[it][ci][dy][4] = yin ? m0.w : 0.f;
                in[it][ci][dy][5] = yin ? m1.x : 0.f; in[it][ci][dy][6] = yin ? m1.y : 0.f;
                in[it][ci][dy][7] = yin ? m1.z : 0.f; in[it][ci][dy][8] = yin ? m1.w : 0.f;
                in[it][ci][dy][9] = (yin && has_r) ? r : 0.f;
            }
        }
        outp[it] = reinterpret_cast<float4 *>(a.out) + (((int64_t)b * a.H + y) * a.W + x0) * QP + q;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        if (!on[it]) continue;
#pragma unroll
        for (int i = 0; i < STEM_STRIP; ++i) {
            float4 acc = bias;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float v = in[it][ci][dy][i + dx];
                        const float4 wv = wr[(dy * 3 + dx) * CIN + ci];
                        acc.x += v * wv.x; acc.y += v * wv.y; acc.z += v * wv.z; acc.w += v * wv.w;
                    }
            outp[it][(int64_t)i * QP] = acc;
            ssum.x += acc.x; ssum.y += acc.y; ssum.z += acc.z; ssum.w += acc.w;
            ssq.x += acc.x * acc.x; ssq.y += acc.y * acc.y; ssq.z += acc.z * acc.z; ssq.w += acc.w * acc.w;
        }
    }
    if (!a.stats) return;
    if (lane_on) {
        const int o = (sl * QP + q) * 4;
        lds_s[o] = ssum.x; lds_s[o + 1] = ssum.y; lds_s[o + 2] = ssum.z; lds_s[o + 3] = ssum.w;
        lds_q[o] = ssq.x; lds_q[o + 1] = ssq.y; lds_q[o + 2] = ssq.z; lds_q[o + 3] = ssq.w;
    }
    __syncthreads();
    float *row = a.stats + (int64_t)blockIdx.x * a.Cout * 2;
    for (int c = threadIdx.x; c < a.Cout; c += 256) {
        float s = 0.f, qq = 0.f;
        for (int r = 0; r < spb; ++r) { s += lds_s[r * QP * 4 + c]; qq += lds_q[r * QP * 4 + c]; }
        row[c * 2] = s;
        row[c * 2 + 1] = qq;
    }
}

template <int CIN, bool LOOP>
__global__ __launch_bounds__(256) void conv_stem_strip_kernel(anoddpm_stem_args a, int iters_arg)
{
    __shared__ float lds_s[256 * 4];
    __shared__ float lds_q[256 * 4];
    const int QP = a.Cout >> 2;
    const int spb = 256 / QP;                        // strips per block and trip
    const int q = threadIdx.x % QP;
    const int sl = threadIdx.x / QP;
    const bool lane_on = sl < spb;
    const int iters = LOOP ? iters_arg : 1;                        // one trip: no loop-carried addressing state (registers)
    const int strips_x = a.W / STEM_STRIP;
    const int64_t nstrips = (int64_t)a.B * a.H * strips_x;
    const int64_t strip0 = (int64_t)blockIdx.x * spb * iters;
    const float4 *w = reinterpret_cast<const float4 *>(a.w);
    float4 wr[9 * CIN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) wr[t * CIN + ci] = w[(t * CIN + ci) * QP + (lane_on ? q : 0)];
    const float4 bias = a.bias ? reinterpret_cast<const float4 *>(a.bias)[lane_on ? q : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
        const int64_t strip = strip0 + (int64_t)it * spb + sl;
        if (!lane_on || strip >= nstrips) continue;
        const int sx = (int)(strip % strips_x);
        const int y = (int)((strip / strips_x) % a.H);
        const int b = (int)(strip / ((int64_t)strips_x * a.H));
        const int x0 = sx * STEM_STRIP;
        // the three input rows of the strip: every load is unconditional (clamped row / column, zeroed afterwards) -- two
        // 16-byte loads for the strip's own eight columns (W % 8 == 0: aligned, always inside) and one scalar per edge
        float in[CIN][3][STEM_STRIP + 2];
        const bool has_l = x0 > 0, has_r = x0 + STEM_STRIP < a.W;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const float *plane = a.x + ((int64_t)b * CIN + ci) * a.H * a.W;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = y + dy - 1;
                const bool yin = yy >= 0 && yy < a.H;
                const float *row = plane + (int64_t)(yin ? yy : y) * a.W + x0;
                const float4 m0 = *reinterpret_cast<const float4 *>(row), m1 = *reinterpret_cast<const float4 *>(row + 4);
                const float l = row[has_l ? -1 : 0], r = row[has_r ? STEM_STRIP : STEM_STRIP - 1];
                in[ci][dy][0] = (yin && has_l) ? l : 0.f;
                in[ci][dy][1] = yin ? m0.x : 0.f; in[ci][dy][2] = yin ? m0.y : 0.f;
                in[ci][dy][3] = yin ? m0.z : 0.f; in[ci][dy][4] = yin ? m0.w : 0.f;
                in[ci][dy][5] = yin ? m1.x : 0.f; in[ci][dy][6] = yin ? m1.y : 0.f;
                in[ci][dy][7] = yin ? m1.z : 0.f; in[ci][dy][8] = yin ? m1.w : 0.f;
                in[ci][dy][9] = (yin && has_r) ? r : 0.f;
            }
        }
        float4 *out = reinterpret_cast<float4 *>(a.out) + (((int64_t)b * a.H + y) * a.W + x0) * QP + q;
#pragma unroll
        for (int i = 0; i < STEM_STRIP; ++i) {
            float4 acc = bias;
            // same accumulation order as the per-pixel kernel: ci outermost, then dy, dx
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float v = in[ci][dy][i + dx];
                        const float4 wv = wr[(dy * 3 + dx) * CIN + ci];
                        acc.x += v * wv.x; acc.y += v * wv.y; acc.z += v * wv.z; acc.w += v * wv.w;
                    }
            out[(int64_t)i * QP] = acc;
            ssum.x += acc.x; ssum.y += acc.y; ssum.z += acc.z; ssum.w += acc.w;
            ssq.x += acc.x * acc.x; ssq.y += acc.y * acc.y; ssq.z += acc.z * acc.z; ssq.w += acc.w * acc.w;
        }
    }
    if (!a.stats) return;
    // fold the strips of the workgroup per channel (fixed order) and write this range's row
    if (lane_on) {
        const int o = (sl * QP + q) * 4;
        lds_s[o] = ssum.x; lds_s[o + 1] = ssum.y; lds_s[o + 2] = ssum.z; lds_s[o + 3] = ssum.w;
        lds_q[o] = ssq.x; lds_q[o + 1] = ssq.y; lds_q[o + 2] = ssq.z; lds_q[o + 3] = ssq.w;
    }
    __syncthreads();
    float *row = a.stats + (int64_t)blockIdx.x * a.Cout * 2;        // rows are numbered like the workgroups: [B][stats_rows]
    for (int c = threadIdx.x; c < a.Cout; c += 256) {
        float s = 0.f, qq = 0.f;
        for (int r = 0; r < spb; ++r) { s += lds_s[r * QP * 4 + c]; qq += lds_q[r * QP * 4 + c]; }
        row[c * 2] = s;
        row[c * 2 + 1] = qq;
    }
}

// generic form: thread = (pixel, 4 output channels)
__global__ __launch_bounds__(256) void conv_stem_kernel(anoddpm_stem_args a)
{
    const int QP = a.Cout >> 2;
    const int ppb = 256 / QP;                        // pixels per block
    const int q = threadIdx.x % QP;
    const int pl = threadIdx.x / QP;
    if (pl >= ppb) return;
    const int64_t npix = (int64_t)a.B * a.H * a.W;
    const int64_t pix = (int64_t)blockIdx.x * ppb + pl;
    if (pix >= npix) return;
    const int x = (int)(pix % a.W);
    const int y = (int)((pix / a.W) % a.H);
    const int b = (int)(pix / ((int64_t)a.W * a.H));
    float4 acc = a.bias ? reinterpret_cast<const float4 *>(a.bias)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 *w = reinterpret_cast<const float4 *>(a.w);
    for (int ci = 0; ci < a.Cin; ++ci) {
        const float *plane = a.x + ((int64_t)b * a.Cin + ci) * a.H * a.W;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
                const float v = plane[(int64_t)yy * a.W + xx];
                const float4 wv = w[(((dy + 1) * 3 + (dx + 1)) * a.Cin + ci) * QP + q];
                acc.x += v * wv.x; acc.y += v * wv.y; acc.z += v * wv.z; acc.w += v * wv.w;
            }
        }
    }
    reinterpret_cast<float4 *>(a.out)[pix * QP + q] = acc;
}

// ---------------------------------------------------------------- head conv ------------------
// out.2 of the reference (UNet.py:386-387): silu(GroupNorm(x)) -> 3x3 conv to Cout <= 4 channels.
// Memory-bound (reads C floats per pixel, writes Cout): one wave per output pixel row segment.
// A 8x8 pixel tile per block; the transformed (affine + SiLU) halo tile (10x10 pixels x C) is staged in
// LDS once, so the transcendental work is done once per input element rather than once per tap.
template <int COUT>
__global__ __launch_bounds__(256) void conv_head_kernel(anoddpm_head_args a)
{
    // LDS: [100 halo pixels][C + 16] activated values + [9][COUT][C] weights.  The 16-float pad makes the pixel stride
    // = 16 (mod 64) banks, so the 16 lanes of a ds_read_b128 service group (4 pixels x 4 channel parts) hit 16
    // different 4-bank groups.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C = a.C, C4 = C >> 2, CP = C + 16;
    float *wl = lds + 100 * CP;
    const int tid = threadIdx.x;
    const int tiles_x = a.W >> 3;
    const int b = blockIdx.y;
    const int y0 = (blockIdx.x / tiles_x) * 8, x0 = (blockIdx.x % tiles_x) * 8;
    for (int i = tid; i < 9 * C * COUT; i += 256) {                 // global [tap][c][o] -> LDS [tap][o][c]
        const int o = i % COUT, c = (i / COUT) % C, tap = i / (COUT * C);
        wl[(tap * COUT + o) * C + c] = a.w[i];
    }
    const float *sc = a.gn_scale + (int64_t)b * C, *sh = a.gn_shift + (int64_t)b * C;
    for (int i = tid; i < 100 * C4; i += 256) {
        const int p = i / C4, q = i % C4;
        const int gy = y0 + p / 10 - 1, gx = x0 + p % 10 - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            v = *reinterpret_cast<const float4 *>(a.x + (((int64_t)b * a.H + gy) * a.W + gx) * C + q * 4);
            const float4 s4 = *reinterpret_cast<const float4 *>(sc + q * 4);
            const float4 h4 = *reinterpret_cast<const float4 *>(sh + q * 4);
            v.x = silu_f(v.x * s4.x + h4.x); v.y = silu_f(v.y * s4.y + h4.y);
            v.z = silu_f(v.z * s4.z + h4.z); v.w = silu_f(v.w * s4.w + h4.w);
        }
        *reinterpret_cast<float4 *>(lds + p * CP + q * 4) = v;
    }
    __syncthreads();
    // 4 lanes per output pixel; lane `part` sums the channel quads q = part, part + 4, ... (16-byte LDS reads for
    // both the activations and the weights), then a 4-lane shuffle reduce
    const int pix = tid >> 2, part = tid & 3;
    const int py = pix >> 3, px = pix & 7;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const float *src = lds + ((py + tap / 3) * 10 + px + tap % 3) * CP;
        const float *wt = wl + tap * COUT * C;
        for (int q = part; q < C4; q += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(src + q * 4);
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const float4 w4 = *reinterpret_cast<const float4 *>(wt + o * C + q * 4);
                acc[o] += v.x * w4.x + v.y * w4.y + v.z * w4.z + v.w * w4.w;
            }
        }
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
        acc[o] += __shfl_xor(acc[o], 1);
        acc[o] += __shfl_xor(acc[o], 2);
    }
    if (part == 0 && y0 + py < a.H && x0 + px < a.W) {
        // NCHW output: [B][COUT][H][W]
#pragma unroll
        for (int o = 0; o < COUT; ++o)
            a.out[(((int64_t)b * COUT + o) * a.H + y0 + py) * a.W + x0 + px] = acc[o] + (a.bias ? a.bias[o] : 0.f);
    }
}

// Input-centric form of the same layer for C % 32 == 0 (the shipped models: C = base channels): a thread owns ONE input pixel
// of a 16x16 tile (14x14 outputs + halo), reads its C channels once (128 bytes per request, whole cache lines), applies
// GroupNorm + SiLU once and forms the nine tap products  t[tap][o] = sum_c act[c] w[tap][c][o]  in registers -- the weights and
// the GroupNorm affine are wave-uniform, i.e. scalar loads.  The tile's products go through LDS once ([9*COUT][256] floats) and an
// output pixel adds its nine neighbours' entries.  Against the output-centric kernel above (every output re-reads 9 x C activations
// and weights from LDS: LDS-bound, 112 us at 256^2 x 128 x batch 4) this one is bound by the SiLU + tap arithmetic.
constexpr int HT = 16;                                              // input tile edge; HT - 2 outputs per edge

template <int COUT>
__global__ __launch_bounds__(256) void conv_head_taps_kernel(anoddpm_head_args a)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    __shared__ float T[9 * COUT][HT * HT];
    const int C = a.C;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int tiles_x = (a.W + HT - 3) / (HT - 2);
    const int b = blockIdx.y;
    const int oy0 = (blockIdx.x / tiles_x) * (HT - 2), ox0 = (blockIdx.x % tiles_x) * (HT - 2);
    const int gy = oy0 + ty - 1, gx = ox0 + tx - 1;                 // this thread's input pixel
    const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const float *__restrict__ xp = a.x + (((int64_t)b * a.H + (inside ? gy : 0)) * a.W + (inside ? gx : 0)) * C;
    const float *__restrict__ sc = a.gn_scale + (int64_t)b * C, *__restrict__ sh = a.gn_shift + (int64_t)b * C;
    f2 t[9][COUT];                                                  // even / odd channel partial sums (packed FMAs)
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int o = 0; o < COUT; ++o) t[k][o] = f2{0.f, 0.f};
    float4 buf[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[0][i] = reinterpret_cast<const float4 *>(xp)[i];
    const int nchunk = C >> 5;
    for (int ch = 0; ch < nchunk; ch += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cc = ch + half;
            if (cc >= nchunk) break;
            const int nx = cc + 1 < nchunk ? cc + 1 : cc;           // clamped prefetch of the next 32 channels
#pragma unroll
            for (int i = 0; i < 8; ++i) buf[half ^ 1][i] = reinterpret_cast<const float4 *>(xp + nx * 32)[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = cc * 32 + i * 4;                      // wave-uniform: scale / shift / weights are scalar loads
                const float4 v = buf[half][i];
                const f2 a01 = {silu_f(v.x * sc[c] + sh[c]), silu_f(v.y * sc[c + 1] + sh[c + 1])};
                const f2 a23 = {silu_f(v.z * sc[c + 2] + sh[c + 2]), silu_f(v.w * sc[c + 3] + sh[c + 3])};
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int o = 0; o < COUT; ++o) {
                        const float *w = a.w + ((int64_t)k * C + c) * COUT + o;        // [tap][c][o]
                        t[k][o] = __builtin_elementwise_fma(a01, f2{w[0], w[COUT]}, t[k][o]);
                        t[k][o] = __builtin_elementwise_fma(a23, f2{w[2 * COUT], w[3 * COUT]}, t[k][o]);
                    }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int o = 0; o < COUT; ++o) T[k * COUT + o][tid] = inside ? t[k][o][0] + t[k][o][1] : 0.f;   // zero padding of the ACTIVATED map
    __syncthreads();
    const int oy = oy0 + ty - 1, ox = ox0 + tx - 1;                 // thread (ty, tx), 1 <= ty, tx <= HT - 2, also owns output (oy, ox)
    if (ty >= 1 && ty <= HT - 2 && tx >= 1 && tx <= HT - 2 && oy < a.H && ox < a.W) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            float r = a.bias ? a.bias[o] : 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) r += T[k * COUT + o][(ty + k / 3 - 1) * HT + tx + k % 3 - 1];
            a.out[(((int64_t)b * COUT + o) * a.H + oy) * a.W + ox] = r;
        }
    }
}

// The same layer with the tap products on the matrix pipe (round 3): T[pixel][tap * COUT + o] = act[pixel][C] x w[C][9 * COUT] is a
// GEMM with N = 9 * COUT <= 16 * NT, v_mfma_f32_16x16x4_f32 with M = 16 pixels of a tile row.  What it buys over the VALU form
// above is not arithmetic but LOADS: there a lane owns a pixel and every wave-level load touches 64 different cache lines (16
// bytes of each), which costs the texture-address unit as much time as HBM needs for the data (46 us for 134 MB); here the
// four lanes (m, q = 0..3) of a pixel read 64 contiguous bytes per load and a wave instruction covers sixteen consecutive pixels.
// Lane (m = lane & 15, q = lane >> 4) holds channels 16 j + 4 q + s (s = 0..3) of pixel m for channel group j: one float4 load
// feeds four MFMAs; the B operand (weights of those channels for column n = m) sits in NT * NJ * 4 registers for the whole kernel.
// GroupNorm-apply + SiLU run on the loaded float4 (affine of the image in LDS).  The tile's products go through LDS once and an
// output pixel adds its nine neighbours' entries, as above.  NJ = C / 16.
typedef float hf32x4 __attribute__((ext_vector_type(4)));

template <int COUT, int NJ>
__global__ __launch_bounds__(256) void conv_head_mfma_kernel(anoddpm_head_args a)
{
    constexpr int NT = (9 * COUT + 15) / 16;
    __shared__ float T[9 * COUT][HT * HT];
    __shared__ hf32x4 aff[2][NJ * 4];
    const int C = NJ * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int tiles_x = (a.W + HT - 3) / (HT - 2);
    const int b = blockIdx.y;
    const int oy0 = (blockIdx.x / tiles_x) * (HT - 2), ox0 = (blockIdx.x % tiles_x) * (HT - 2);
    for (int i = tid; i < NJ * 4; i += 256) {
        aff[0][i] = reinterpret_cast<const hf32x4 *>(a.gn_scale + (int64_t)b * C)[i];
        aff[1][i] = reinterpret_cast<const hf32x4 *>(a.gn_shift + (int64_t)b * C)[i];
    }
    // B operand: column n = nt * 16 + m is (tap, o) = (n / COUT, n % COUT); rows = this lane's channels
    float bw[NT][NJ][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + m;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                bw[nt][j][s] = n < 9 * COUT ? a.w[((int64_t)(n / COUT) * C + 16 * j + 4 * q + s) * COUT + n % COUT] : 0.f;
    }
    __syncthreads();
    // tile row ty = wave * 4 + mt, pixel tx = m.  Four float4 (64 channels) of loads in flight per lane: latency is covered by the
    // other workgroups of the CU (a double-buffered row needs 192 registers at C = 128 and halves the occupancy).
#pragma unroll 1
    for (int mt = 0; mt < 4; ++mt) {
        const int gy = oy0 + wave * 4 + mt - 1, gx = ox0 + m - 1;
        const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const float *xp = a.x + (((int64_t)b * a.H + (inside ? gy : 0)) * a.W + (inside ? gx : 0)) * C + 4 * q;
        hf32x4 accr[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accr[nt] = hf32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int JB = NJ > 4 ? 4 : NJ;                          // channel groups per batch of loads (64 channels)
#pragma unroll
        for (int j0 = 0; j0 < NJ; j0 += JB) {
            hf32x4 v[JB];
#pragma unroll
            for (int j = 0; j < JB; ++j) v[j] = *reinterpret_cast<const hf32x4 *>(xp + 16 * (j0 + j));
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                const hf32x4 sc = aff[0][4 * (j0 + j) + q], sh = aff[1][4 * (j0 + j) + q];
                const hf32x4 x4 = v[j] * sc + sh;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float av = inside ? silu_f(x4[s]) : 0.f;  // zero padding of the ACTIVATED map
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) accr[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[nt][j0 + j][s], accr[nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                      // keep the batches apart (registers)
        }
        // D[row = 4 q + r][col = m]: this lane holds column n = nt * 16 + m for pixels 4 q .. 4 q + 3 of the row
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + m;
            if (n < 9 * COUT) *reinterpret_cast<hf32x4 *>(&T[n][(wave * 4 + mt) * HT + 4 * q]) = accr[nt];
        }
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    const int oy = oy0 + ty - 1, ox = ox0 + tx - 1;                 // thread (ty, tx), 1 <= ty, tx <= HT - 2, also owns output (oy, ox)
    if (ty >= 1 && ty <= HT - 2 && tx >= 1 && tx <= HT - 2 && oy < a.H && ox < a.W) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            float r = a.bias ? a.bias[o] : 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) r += T[k * COUT + o][(ty + k / 3 - 1) * HT + tx + k % 3 - 1];
            a.out[(((int64_t)b * COUT + o) * a.H + oy) * a.W + ox] = r;
        }
    }
}

// Round 4 form of the kernel above: the same contraction, issue-trimmed.  The inner loop of the round-3 kernel was a dependent
// chain per value wrapped in an exec-mask branch (the `inside` test): saveexec / branch / restore around seven dependent
// instructions, 129 SIMD cycles per 64 values against 32 (MFMA) + ~38 (affine + SiLU) of issue.  Here
//   * the zero padding of the activated map moves to the tile's product rows: out-of-image pixels are zeroed when the wave writes
//     its D rows to LDS (four selects per tile row instead of 32 masked SiLUs); their loads are clamped in-bounds, so what
//     they compute is finite and discarded;
//   * the four values of a float4 run as four independent chains in straight-line code;
//   * the input tile is 16 x 28 (14 x 26 outputs): 1.23x halo instead of 1.31x;
//   * the next tile row's channels are requested before the current row is computed (two rows in registers).
constexpr int HT2R = 28;                                            // input tile rows; HT2R - 2 = 26 output rows: 256 rows = 10 strips, and
                                                                    // 19 x 10 x 4 images = 760 workgroups = 2.97 per CU (30-row strips: 684 = 2.67
                                                                    // per CU, i.e. a third of the CUs carry three workgroups and set the time)

template <int COUT, int NJ>
__global__ __launch_bounds__(256) void conv_head_mfma2_kernel(anoddpm_head_args a)
{
    constexpr int NT = (9 * COUT + 15) / 16;
    constexpr int RPW = HT2R / 4;                                   // tile rows per wave
    __shared__ float T[9 * COUT][HT2R * HT];
    __shared__ hf32x4 aff[2][NJ * 4];
    const int C = NJ * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int tiles_x = (a.W + HT - 3) / (HT - 2);
    const int b = blockIdx.y;
    const int oy0 = (blockIdx.x / tiles_x) * (HT2R - 2), ox0 = (blockIdx.x % tiles_x) * (HT - 2);
    for (int i = tid; i < NJ * 4; i += 256) {
        aff[0][i] = reinterpret_cast<const hf32x4 *>(a.gn_scale + (int64_t)b * C)[i];
        aff[1][i] = reinterpret_cast<const hf32x4 *>(a.gn_shift + (int64_t)b * C)[i];
    }
    float bw[NT][NJ][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + m;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                bw[nt][j][s] = n < 9 * COUT ? a.w[((int64_t)(n / COUT) * C + 16 * j + 4 * q + s) * COUT + n % COUT] : 0.f;
    }
    // this lane's input column (clamped) and, for the D rows it will hold (pixels 4 q .. 4 q + 3 of a tile row), their validity
    const int gx = ox0 + m - 1;
    const int gxc = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
    bool colok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int x = ox0 + 4 * q + r - 1; colok[r] = x >= 0 && x < a.W; }
    auto row_ptr = [&](int mt) {
        const int gy = oy0 + wave * RPW + mt - 1;
        const int gyc = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
        return a.x + (((int64_t)b * a.H + gyc) * a.W + gxc) * C + 4 * q;
    };
    hf32x4 v[2][NJ];
    {
        const float *xp = row_ptr(0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[0][j] = *reinterpret_cast<const hf32x4 *>(xp + 16 * j);
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < RPW; ++mt) {
        const int cur = mt & 1;
        if (mt + 1 < RPW) {
            const float *xp = row_ptr(mt + 1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[cur ^ 1][j] = *reinterpret_cast<const hf32x4 *>(xp + 16 * j);
        }
        hf32x4 accr[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accr[nt] = hf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const hf32x4 x4 = v[cur][j] * aff[0][4 * j + q] + aff[1][4 * j + q];
            hf32x4 e;
#pragma unroll
            for (int s = 0; s < 4; ++s) e[s] = __expf(-x4[s]);
#pragma unroll
            for (int s = 0; s < 4; ++s) e[s] = x4[s] * __builtin_amdgcn_rcpf(1.0f + e[s]);      // silu_f, four chains side by side
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) accr[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(e[s], bw[nt][j][s], accr[nt], 0, 0, 0);
        }
        // D[row = 4 q + r][col = m]: zero padding of the ACTIVATED map = zero product rows of out-of-image pixels
        const int gy = oy0 + wave * RPW + mt - 1;
        const bool rowok = gy >= 0 && gy < a.H;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            hf32x4 d = accr[nt];
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = (rowok && colok[r]) ? d[r] : 0.f;
            const int n = nt * 16 + m;
            if (n < 9 * COUT) *reinterpret_cast<hf32x4 *>(&T[n][(wave * RPW + mt) * HT + 4 * q]) = d;
        }
    }
    __syncthreads();
    for (int p = tid; p < (HT2R - 2) * (HT - 2); p += 256) {
        const int ty = p / (HT - 2) + 1, tx = p % (HT - 2) + 1;       // tile coordinates of the output pixel, 1 .. edge - 2
        const int oy = oy0 + ty - 1, ox = ox0 + tx - 1;
        if (oy < a.H && ox < a.W) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                float r = a.bias ? a.bias[o] : 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k) r += T[k * COUT + o][(ty + k / 3 - 1) * HT + tx + k % 3 - 1];
                a.out[(((int64_t)b * COUT + o) * a.H + oy) * a.W + ox] = r;
            }
        }
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(anoddpm_layout_args a)
{
    const int64_t total = (int64_t)a.B * a.C * a.P;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % a.P);
        const int c = (int)((i / a.P) % a.C);
        const int b = (int)(i / ((int64_t)a.P * a.C));
        a.out[i] = a.in[((int64_t)b * a.P + p) * a.in_ld + c];
    }
}

// ---------------------------------------------------------------- dropout (training) ----------
// keep decision of element `idx`: a 64-bit mix (splitmix64 finaliser) of the seed and the index, compared with p * 2^32
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, unsigned thresh)
{
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 32) >= thresh;
}

__global__ __launch_bounds__(256) void dropout_kernel(anoddpm_dropout_args a, unsigned thresh, float inv_keep)
{
    const int b = blockIdx.y;
    const int64_t base = (int64_t)b * a.n;
    const int64_t n4 = a.n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4 *>(a.x + base)[i];
        if (a.mode == 0) {
            const int c = (int)((i * 4) % a.C);
            const float4 sc = *reinterpret_cast<const float4 *>(a.gn_scale + (int64_t)b * a.C + c);
            const float4 sh = *reinterpret_cast<const float4 *>(a.gn_shift + (int64_t)b * a.C + c);
            v.x = silu_f(v.x * sc.x + sh.x); v.y = silu_f(v.y * sc.y + sh.y);
            v.z = silu_f(v.z * sc.z + sh.z); v.w = silu_f(v.w * sc.w + sh.w);
        }
        const uint64_t e = (uint64_t)(base + i * 4);
        v.x = dropout_keep(a.seed, e, thresh) ? v.x * inv_keep : 0.f;
        v.y = dropout_keep(a.seed, e + 1, thresh) ? v.y * inv_keep : 0.f;
        v.z = dropout_keep(a.seed, e + 2, thresh) ? v.z * inv_keep : 0.f;
        v.w = dropout_keep(a.seed, e + 3, thresh) ? v.w * inv_keep : 0.f;
        reinterpret_cast<float4 *>(a.out + base)[i] = v;
    }
}

inline unsigned cap_grid(int64_t blocks) { return (unsigned)(blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks)); }

}  // namespace

extern "C" int anoddpm_gn_stats(const anoddpm_gn_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->a0 && a->gamma && a->beta && a->scale && a->shift && a->partial, "gn_stats: null pointer");
    const int C = a->c0 + a->c1;
    ANODDPM_REQUIRE(a->c0 > 0 && a->c0 % 4 == 0 && a->c1 % 4 == 0 && (a->c1 == 0 || a->a1), "gn_stats: channel counts must be multiples of 4");
    ANODDPM_REQUIRE(a->groups > 0 && a->groups <= 64 && C % a->groups == 0, "gn_stats: C %% groups != 0");
    ANODDPM_REQUIRE(C / 4 <= 256 * 16, "gn_stats: too many channels");
    ANODDPM_REQUIRE(a->B > 0 && a->B <= 65535 && a->P > 0 && a->nslab > 0 && a->nslab <= 65535, "gn_stats: bad sizes");
    hipStream_t s = anoddpm::as_stream(stream);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(a->nslab, a->B), dim3(256), 0, s, *a);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(a->B), dim3(256), 0, s, *a);
    return anoddpm::check_launch("gn_stats");
}

extern "C" int anoddpm_chan_stats(const anoddpm_chan_stats_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->a && a->stats, "chan_stats: null pointer");
    ANODDPM_REQUIRE(a->C > 0 && a->C % 4 == 0 && a->C / 4 <= 256 * 16 && a->a_ld % 4 == 0, "chan_stats: C must be a multiple of 4");
    ANODDPM_REQUIRE(a->B > 0 && a->B <= 65535 && a->P > 0 && a->nslab > 0 && a->nslab <= 65535, "chan_stats: bad sizes");
    hipLaunchKernelGGL(chan_stats_kernel, dim3(a->nslab, a->B), dim3(256), 0, anoddpm::as_stream(stream), *a);
    return anoddpm::check_launch("chan_stats");
}

extern "C" int anoddpm_gn_finalize(const anoddpm_gn_finalize_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->stats0 && a->gamma && a->beta && a->scale && a->shift, "gn_finalize: null pointer");
    ANODDPM_REQUIRE(a->c0 > 0 && a->c1 >= 0 && (a->c1 == 0 || a->stats1), "gn_finalize: bad channel counts");
    const int C = a->c0 + a->c1;
    ANODDPM_REQUIRE(a->groups > 0 && a->groups <= 65535 && C % a->groups == 0 && C / a->groups <= 256, "gn_finalize: bad group size");
    ANODDPM_REQUIRE(a->B > 0 && a->B <= 65535 && a->P > 0 && (a->fmt0 != 0 || a->rows0 > 0) && (a->c1 == 0 || a->fmt1 != 0 || a->rows1 > 0), "gn_finalize: bad sizes");
    ANODDPM_REQUIRE((a->mean_out == nullptr) == (a->rstd_out == nullptr), "gn_finalize: mean_out and rstd_out go together");
    hipLaunchKernelGGL(gn_finalize2_kernel, dim3(a->groups, a->B), dim3(256), 0, anoddpm::as_stream(stream), *a);
    return anoddpm::check_launch("gn_finalize");
}

extern "C" int anoddpm_softmax_rows(const anoddpm_softmax_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->x && a->rows >= 0 && a->L > 0, "softmax_rows: bad arguments");
    if (a->rows == 0) return ANODDPM_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((a->rows + 3) / 4)), dim3(256), 0,
                       anoddpm::as_stream(stream), a->x, a->rows, a->L);
    return anoddpm::check_launch("softmax_rows");
}

extern "C" int anoddpm_resample2x(const anoddpm_resample_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->in && a->out, "resample2x: null pointer");
    ANODDPM_REQUIRE(a->C % 4 == 0 && (a->mode == 1 || a->mode == 4 || ((a->mode == 2 || a->mode == 3) && a->H % 2 == 0 && a->W % 2 == 0)), "resample2x: bad shape/mode");
    ANODDPM_REQUIRE(!a->out_act || (a->mode == 2 && a->gn_scale && a->gn_shift), "resample2x: the activated output needs mode 2 and a GroupNorm affine");
    const bool up = a->mode == 1 || a->mode == 4;
    const int Ho = up ? a->H * 2 : a->H / 2, Wo = up ? a->W * 2 : a->W / 2;
    const int64_t total = (int64_t)a->B * Ho * Wo * (a->C / 4);
    if (total == 0) return ANODDPM_OK;
    hipLaunchKernelGGL(resample2x_kernel, dim3(cap_grid((total + 255) / 256)), dim3(256), 0, anoddpm::as_stream(stream), *a);
    return anoddpm::check_launch("resample2x");
}

extern "C" int anoddpm_dropout(const anoddpm_dropout_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->x && a->out, "dropout: null pointer");
    ANODDPM_REQUIRE(a->B >= 0 && a->B <= 65535 && a->n >= 0 && a->n % 4 == 0 && a->C > 0 && a->C % 4 == 0 && a->n % a->C == 0, "dropout: bad sizes");
    ANODDPM_REQUIRE(a->mode == 1 || (a->mode == 0 && a->gn_scale && a->gn_shift), "dropout: mode 0 needs the GroupNorm affine");
    ANODDPM_REQUIRE(a->p >= 0.f && a->p < 1.f, "dropout: p must be in [0, 1)");
    if (a->B == 0 || a->n == 0) return ANODDPM_OK;
    const double t = (double)a->p * 4294967296.0;
    const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    hipLaunchKernelGGL(dropout_kernel, dim3(cap_grid((a->n / 4 + 255) / 256), (unsigned)a->B), dim3(256), 0, anoddpm::as_stream(stream),
                       *a, thresh, 1.0f / (1.0f - a->p));
    return anoddpm::check_launch("dropout");
}

extern "C" int anoddpm_linear_small(const anoddpm_linear_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->in && a->w && a->out, "linear_small: null pointer");
    ANODDPM_REQUIRE(a->B >= 1 && a->B <= 4096 && a->K % 4 == 0 && a->N >= 1, "linear_small: need 1<=B<=4096, K%%4==0");
    dim3 grid((a->N + 3) / 4);
    hipStream_t s = anoddpm::as_stream(stream);
    // up to 16 batch rows per launch live in registers; larger batches (round 6: the inference plan stopped at batch 16 here --
    // the detection loop with more than 16 chain slots, a caller's batch of 32) take one launch per 16 rows
    for (int b0 = 0; b0 < a->B; b0 += 16) {
        anoddpm_linear_args c = *a;
        c.in = a->in + (int64_t)b0 * a->K;
        c.out = a->out + (int64_t)b0 * a->N;
        c.B = a->B - b0 < 16 ? a->B - b0 : 16;
        if (c.B <= 4) hipLaunchKernelGGL(linear_small_kernel<4>, grid, dim3(256), 0, s, c);
        else if (c.B <= 8) hipLaunchKernelGGL(linear_small_kernel<8>, grid, dim3(256), 0, s, c);
        else hipLaunchKernelGGL(linear_small_kernel<16>, grid, dim3(256), 0, s, c);
    }
    return anoddpm::check_launch("linear_small");
}

extern "C" int anoddpm_posemb(const anoddpm_posemb_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->t && a->freqs && a->out && a->B >= 1 && a->dim >= 2 && a->dim % 2 == 0, "posemb: bad arguments");
    const int n = a->B * (a->dim / 2);
    ANODDPM_REQUIRE(a->zero_doubles >= 0 && (a->zero_doubles == 0 || (a->zero && (uintptr_t)a->zero % 8 == 0)), "posemb: bad zero range");
    int64_t blocks = (n + 255) / 256;
    const int64_t zblocks = (a->zero_doubles / 2 + 255) / 256;            // 16 bytes per thread and trip; at most one block per CU x 2
    if (zblocks > blocks) blocks = zblocks < 512 ? zblocks : 512;
    hipLaunchKernelGGL(posemb_kernel, dim3((unsigned)blocks), dim3(256), 0, anoddpm::as_stream(stream), *a);
    return anoddpm::check_launch("posemb");
}

// rows per image of the fused stem statistics: one per workgroup range.  One trip per workgroup up to 1024 rows (256^2 x 128: 4096
// workgroups of 32 KB -- measured 27 us against 38 us with four trips per workgroup: the trips of a workgroup serialise on their
// input loads and halve the waves in flight); beyond that (512^2) the trips double until the finalize kernel's row walk fits
extern "C" int anoddpm_stem_stats_rows(int H, int W, int Cin, int Cout)
{
    if (Cin < 1 || Cin > 2 || W <= 0 || H <= 0 || W % STEM_STRIP || Cout <= 0 || Cout % 4 || Cout / 4 > 256) return 0;
    const int ppb = 256 / (Cout / 4);
    const int64_t per_image = (int64_t)H * W / STEM_STRIP;
    if (per_image % ppb) return 0;
    const int64_t trips = per_image / ppb;
    int64_t iters = 1;
    // round 6: two trips per workgroup with BOTH trips' input rows requested up front (conv_stem_strip2_kernel): 31.1 -> 28.4 us at
    // 256^2 x 128 x batch 4 (4.35 -> 4.77 TB/s); Cin = 1 only (the two-channel form would need 244 registers).  ANODDPM_DEBUG9=1
    // keeps one trip per workgroup.
    if (anoddpm::g_debug[9] != 1 && Cin == 1 && trips % 2 == 0 && trips / 2 >= 256) iters = 2;
    while (trips / iters > 1024) iters *= 2;
    while (iters > 1 && trips % iters) iters /= 2;
    return (int)(trips / iters);
}

extern "C" int anoddpm_conv_stem(const anoddpm_stem_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->x && a->w && a->out, "conv_stem: null pointer");
    ANODDPM_REQUIRE(a->Cin >= 1 && a->Cin <= 16 && a->Cout % 4 == 0 && a->Cout / 4 <= 256, "conv_stem: need Cin<=16, Cout%%4==0, Cout<=1024");
    const int ppb = 256 / (a->Cout / 4);
    const int64_t npix = (int64_t)a->B * a->H * a->W;
    if (npix == 0) return ANODDPM_OK;
    if (a->Cin <= 2 && a->W % STEM_STRIP == 0 && (uintptr_t)a->x % 16 == 0) {     // the strip kernel reads its rows with 16-byte loads
        const int64_t nstrips = npix / STEM_STRIP;
        const int64_t per_image = nstrips / a->B;
        int iters;
        if (a->stats) {
            // one statistics row per workgroup, stats_rows workgroups per image: the range of a workgroup must not straddle images
            ANODDPM_REQUIRE(a->stats_rows > 0 && per_image % ((int64_t)a->stats_rows * ppb) == 0, "conv_stem: stats_rows must divide the strips of an image into whole workgroup trips");
            iters = (int)(per_image / ((int64_t)a->stats_rows * ppb));
        } else {
            iters = (a->Cin == 1 && anoddpm::g_debug[9] != 1 && per_image % (2 * (int64_t)ppb) == 0 && per_image / (2 * ppb) >= 256) ? 2 : 1;
        }
        const dim3 grid((unsigned)((nstrips + (int64_t)ppb * iters - 1) / ((int64_t)ppb * iters)));
        hipStream_t st = anoddpm::as_stream(stream);
        if (a->Cin == 1 && iters == 2)      hipLaunchKernelGGL((conv_stem_strip2_kernel<1>), grid, dim3(256), 0, st, *a);
        else if (a->Cin == 1 && iters == 1) hipLaunchKernelGGL((conv_stem_strip_kernel<1, false>), grid, dim3(256), 0, st, *a, 1);
        else if (a->Cin == 1)          hipLaunchKernelGGL((conv_stem_strip_kernel<1, true>), grid, dim3(256), 0, st, *a, iters);
        else if (iters == 1)           hipLaunchKernelGGL((conv_stem_strip_kernel<2, false>), grid, dim3(256), 0, st, *a, 1);
        else                           hipLaunchKernelGGL((conv_stem_strip_kernel<2, true>), grid, dim3(256), 0, st, *a, iters);
        return anoddpm::check_launch("conv_stem");
    }
    ANODDPM_REQUIRE(!a->stats, "conv_stem: fused statistics need Cin <= 2, W %% 8 == 0 and a 16-byte aligned input (use anoddpm_chan_stats)");
    hipLaunchKernelGGL(conv_stem_kernel, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), 0, anoddpm::as_stream(stream), *a);
    return anoddpm::check_launch("conv_stem");
}

extern "C" int anoddpm_conv_head(const anoddpm_head_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->x && a->w && a->out && a->gn_scale && a->gn_shift, "conv_head: null pointer");
    ANODDPM_REQUIRE(a->Cout >= 1 && a->Cout <= 4 && a->C % 4 == 0 && a->W % 8 == 0 && a->H % 8 == 0, "conv_head: need Cout<=4, C%%4==0, H,W%%8==0");
    if (a->C % 32 == 0 && anoddpm::g_debug[4] != 1) {               // ANODDPM_DEBUG4=1: the output-centric kernel
        const int tx = (a->W + HT - 3) / (HT - 2), tyy = (a->H + HT - 3) / (HT - 2);
        dim3 grid(tx * tyy, a->B);
        hipStream_t s = anoddpm::as_stream(stream);
        // matrix-pipe form for the shipped widths (B operand in registers: NT * NJ <= 24); ANODDPM_DEBUG4=2: the VALU tap kernel
        const bool aligned = ((uintptr_t)a->x | (uintptr_t)a->gn_scale | (uintptr_t)a->gn_shift) % 16 == 0;
        if (aligned && anoddpm::g_debug[4] == 0 && a->Cout == 1 && (a->C == 128 || a->C == 64)) {
            // round 4: branch-free 16 x 32 tiles (ANODDPM_DEBUG4=3 keeps the round-3 kernel)
            const int ty2 = (a->H + HT2R - 3) / (HT2R - 2);
            dim3 grid2(tx * ty2, a->B);
            if (a->C == 128) hipLaunchKernelGGL((conv_head_mfma2_kernel<1, 8>), grid2, dim3(256), 0, s, *a);
            else             hipLaunchKernelGGL((conv_head_mfma2_kernel<1, 4>), grid2, dim3(256), 0, s, *a);
            return anoddpm::check_launch("conv_head");
        }
        if (aligned && anoddpm::g_debug[4] != 2) {
#define HEAD_MFMA(CO, NJ_) { hipLaunchKernelGGL((conv_head_mfma_kernel<CO, NJ_>), grid, dim3(256), 0, s, *a); return anoddpm::check_launch("conv_head"); }
            if (a->C == 128) { if (a->Cout == 1) HEAD_MFMA(1, 8) if (a->Cout == 2) HEAD_MFMA(2, 8) if (a->Cout == 3) HEAD_MFMA(3, 8) if (a->Cout == 4) HEAD_MFMA(4, 8) }
            if (a->C == 64)  { if (a->Cout == 1) HEAD_MFMA(1, 4) if (a->Cout == 2) HEAD_MFMA(2, 4) if (a->Cout == 3) HEAD_MFMA(3, 4) if (a->Cout == 4) HEAD_MFMA(4, 4) }
            if (a->C == 256 && a->Cout == 1) HEAD_MFMA(1, 16)
#undef HEAD_MFMA
        }
        switch (a->Cout) {
            case 1: hipLaunchKernelGGL(conv_head_taps_kernel<1>, grid, dim3(256), 0, s, *a); break;
            case 2: hipLaunchKernelGGL(conv_head_taps_kernel<2>, grid, dim3(256), 0, s, *a); break;
            case 3: hipLaunchKernelGGL(conv_head_taps_kernel<3>, grid, dim3(256), 0, s, *a); break;
            default: hipLaunchKernelGGL(conv_head_taps_kernel<4>, grid, dim3(256), 0, s, *a); break;
        }
        return anoddpm::check_launch("conv_head");
    }
    const size_t lds = (size_t)(100 * (a->C + 16) + 9 * a->C * a->Cout) * sizeof(float);
    ANODDPM_REQUIRE(lds <= 64 * 1024, "conv_head: channel count too large for the 64 KiB dynamic LDS tile");
    dim3 grid((a->H / 8) * (a->W / 8), a->B);
    hipStream_t s = anoddpm::as_stream(stream);
    switch (a->Cout) {
        case 1: hipLaunchKernelGGL(conv_head_kernel<1>, grid, dim3(256), lds, s, *a); break;
        case 2: hipLaunchKernelGGL(conv_head_kernel<2>, grid, dim3(256), lds, s, *a); break;
        case 3: hipLaunchKernelGGL(conv_head_kernel<3>, grid, dim3(256), lds, s, *a); break;
        default: hipLaunchKernelGGL(conv_head_kernel<4>, grid, dim3(256), lds, s, *a); break;
    }
    return anoddpm::check_launch("conv_head");
}

extern "C" int anoddpm_nhwc_to_nchw(const anoddpm_layout_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->in && a->out && a->in_ld >= a->C, "nhwc_to_nchw: bad arguments");
    const int64_t total = (int64_t)a->B * a->C * a->P;
    if (total == 0) return ANODDPM_OK;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cap_grid((total + 255) / 256)), dim3(256), 0, anoddpm::as_stream(stream), *a);
    return anoddpm::check_launch("nhwc_to_nchw");
}
