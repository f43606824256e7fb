// GroupNorm finished in the prologue of an F(4x4,3x3) workgroup from fp64 per-channel sums (round 6).
//
// Producer side: the epilogues of wino43r_kernel / wino43_kernel add their per-channel {sum, sum of squares} (fp32 over the
// workgroup's 256 pixels) to anoddpm_igemm_args.stats_csum [B][N][2] with device-scope fp64 atomic adds -- one 2 KB row per image
// instead of one row per workgroup (256 rows per image at 256x256) + a gn_finalize launch that folds them.
// Consumer side (this file): thread t < K/4 reads the 64 bytes of its four channels, the sums go through LDS once, thread t reduces
// its group and turns mean / rstd into the scale / shift float4s of the LDS affine table the staging code reads (UNet.py:409-411:
// nn.GroupNorm(32, C), eps 1e-5, biased variance, computed in fp64 like anoddpm_gn_finalize).  One L2 round trip that overlaps the
// first patch request + one barrier, against a 4.7 us dependent launch between two convolutions.  The atomics are not free: the
// chip retires ~33 of them per ns (profiles/r6_csum_by_layer.txt), so the plan uses this route for launches of <= 100 k adds only.
#pragma once
#include "common.h"

namespace anoddpm {

typedef float gf_f32x4 __attribute__((ext_vector_type(4)));

// Adds (s, q) to csum[(b * N + n) * 2 ..] at device scope.  No return value: the adds drain behind the kernel's stores.
__device__ __forceinline__ void csum_atomic_add(double *csum, int64_t b, int N, int n, float s, float q)
{
    double *d = csum + (b * N + n) * 2;
    __hip_atomic_fetch_add(d, (double)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(d + 1, (double)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The fold in two halves, so that its global loads can be the workgroup's OLDEST requests (vmcnt retires in order: issued behind the
// patch and weight requests they would wait for those to come back first, and the whole fold would sit behind an HBM round trip).
struct FoldLoads {
    double2 v[4];               // {sum, sumsq} of the thread's four channels
    gf_f32x4 gam, bet;
};

// Issues the loads of thread tid < K / 4 (64 bytes of sums, 16 + 16 of gamma / beta).  Requires K % 4 == 0, c0 % 4 == 0.
__device__ __forceinline__ FoldLoads fold_affine_request(const anoddpm_igemm_args &a, int b, int tid)
{
    FoldLoads f;
    const int K4 = (a.c0 + a.c1) >> 2;
    const double2 z = {0.0, 0.0};
    f.v[0] = f.v[1] = f.v[2] = f.v[3] = z;
    f.gam = f.bet = gf_f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < K4) {
        const int c = tid * 4;
        const bool first = c < a.c0;
        const double2 *sd = reinterpret_cast<const double2 *>(reinterpret_cast<const double *>(first ? a.fold_stats0 : a.fold_stats1) +
                                                              ((int64_t)b * (first ? a.c0 : a.c1) + (first ? c : c - a.c0)) * 2);
        f.v[0] = sd[0]; f.v[1] = sd[1]; f.v[2] = sd[2]; f.v[3] = sd[3];
        f.gam = *reinterpret_cast<const gf_f32x4 *>(a.fold_gamma + c);
        f.bet = *reinterpret_cast<const gf_f32x4 *>(a.fold_beta + c);
    }
    return f;
}

// scratch: 2 * K doubles of LDS nobody else touches until the caller's next barrier.  aff: [K/4] scales, then [K/4] shifts.
// P = pixels of the normalised tensor (the SOURCE resolution for a nearest-x2 operand).  ONE barrier inside; the caller publishes
// `aff` with its own.  Every thread reduces the group(s) of its own four channels from the LDS copy of the sums (at most 32 channel
// pairs per group: redundant between the threads of a group, but no second barrier and no 32-thread serial phase -- this sits in
// the prologue of every workgroup).  mean = S / n, var = Q / n - mean^2 (biased, clamped at 0), rstd = (var + eps)^-1/2 in fp64:
// one division per thread (1 / n), the inverse square root from v_rsq_f64 + two Newton steps (full double precision).
// Requires K % groups == 0, K / 4 <= threads of the workgroup.
__device__ __forceinline__ void fold_affine_finish(const anoddpm_igemm_args &a, const FoldLoads &f, int tid, int P, double *scratch, gf_f32x4 *aff)
{
    const int K = a.c0 + a.c1, K4 = K >> 2, groups = a.fold_groups, cpg = K / groups;
    if (tid < K4) {
        double2 *dst = reinterpret_cast<double2 *>(scratch + 8 * tid);
        dst[0] = f.v[0]; dst[1] = f.v[1]; dst[2] = f.v[2]; dst[3] = f.v[3];
    }
    __syncthreads();
    if (tid < K4) {
        const double inv_n = 1.0 / ((double)P * cpg);
        gf_f32x4 sc, sh;
        int gprev = -1;
        double mean = 0.0, rstd = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = (tid * 4 + j) / cpg;
            if (g != gprev) {                                        // cpg >= 4 (K >= 128): once per thread
                double S = 0.0, Q = 0.0;
                for (int c = g * cpg; c < (g + 1) * cpg; ++c) { S += scratch[2 * c]; Q += scratch[2 * c + 1]; }  // channel order
                mean = S * inv_n;
                double var = Q * inv_n - mean * mean;
                var = (var > 0.0 ? var : 0.0) + (double)a.fold_eps;
                double r = __builtin_amdgcn_rsq(var);
                r = r * (1.5 - 0.5 * var * r * r);
                r = r * (1.5 - 0.5 * var * r * r);
                rstd = r;
                gprev = g;
            }
            const double s = rstd * (double)f.gam[j];
            sc[j] = (float)s;
            sh[j] = (float)((double)f.bet[j] - mean * s);
        }
        aff[tid] = sc;
        aff[K4 + tid] = sh;
    }
}

}  // namespace anoddpm
