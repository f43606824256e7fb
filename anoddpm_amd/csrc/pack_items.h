// Per-thread bodies of the device-side weight packing (training re-packs after every optimizer step), shared by the stand-alone
// kernels (wgrad.hip, train_kernels.hip) and by the batched launch that packs every weight of the model at once.
#pragma once
#include "common.h"

namespace anoddpm {

// OIHW 3x3 -> mode 0: direct [9][I/4][O][4]; mode 1: Winograd F(2x2,3x3) U = G g G^T [16][I/4][O][4]; mode 2: F(4x4,3x3)
// [36][I/4][O][4] (fp64 like the host versions, unet.py:_pack_conv / _pack_wino / _pack_wino43).  bwd != 0 packs the data-gradient
// weights W'[o=k][i=n][a][b] = w[n][k][2-a][2-b].  idx = one (o, input-channel quad).
__device__ __forceinline__ void pack_conv3x3_item(const float *__restrict__ w, float *__restrict__ out, int N, int K, int mode, int bwd, int64_t idx)
{
    // thread = (o, input-channel QUAD): every store is one 16-byte slot of the [..][I/4][O][4] layout (consecutive threads ->
    // consecutive slots), and the forward layout reads 4 x 9 contiguous floats
    const int O = bwd ? K : N, I = bwd ? N : K;
    const int I4 = I >> 2;
    if (idx >= (int64_t)O * I4) return;
    const int o = (int)(idx % O), i4 = (int)(idx / O);
    if (mode == 0) {
        // direct layout: a pure permutation -- no round trip through fp64 (the conversions are quarter-rate instructions and this
        // kernel is bound by them and by the fp64 transforms of the other two modes, not by memory: staging the reads through LDS
        // for full coalescing changed nothing, 0.98 vs 1.00 ms per training step)
        float4 *dst = reinterpret_cast<float4 *>(out) + (int64_t)i4 * O + o;
        const int64_t plane = (int64_t)I4 * O;
        const float *s0 = bwd ? w + ((int64_t)(i4 * 4) * K + o) * 9 : w + ((int64_t)o * K + i4 * 4) * 9;
        const int64_t es = bwd ? (int64_t)K * 9 : 9;                 // stride between the quad's four input channels
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ts = bwd ? 8 - t : t;
            dst[t * plane] = make_float4(s0[ts], s0[es + ts], s0[2 * es + ts], s0[3 * es + ts]);
        }
        return;
    }
    double g[4][3][3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = i4 * 4 + e;
        const float *src = bwd ? w + ((int64_t)i * K + o) * 9 : w + ((int64_t)o * K + i) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) g[e][t / 3][t % 3] = (double)src[bwd ? 8 - t : t];
    }
    float4 *dst = reinterpret_cast<float4 *>(out) + (int64_t)i4 * O + o;
    const int64_t plane = (int64_t)I4 * O;                           // float4 slots per position
    if (mode == 2) {
        // Winograd F(4x4,3x3): U = G g G^T with the 6x3 G of interpolation points 0, +-1, +-2, inf; [36 xi = 6u+v][I/4][O][4]
        const double G6[6][3] = {{1.0 / 4, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            double t1[4][3];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b2 = 0; b2 < 3; ++b2) t1[e][b2] = G6[u][0] * g[e][0][b2] + G6[u][1] * g[e][1][b2] + G6[u][2] * g[e][2][b2];
#pragma unroll
            for (int v = 0; v < 6; ++v) {
                float4 o4;
                o4.x = (float)(t1[0][0] * G6[v][0] + t1[0][1] * G6[v][1] + t1[0][2] * G6[v][2]);
                o4.y = (float)(t1[1][0] * G6[v][0] + t1[1][1] * G6[v][1] + t1[1][2] * G6[v][2]);
                o4.z = (float)(t1[2][0] * G6[v][0] + t1[2][1] * G6[v][1] + t1[2][2] * G6[v][2]);
                o4.w = (float)(t1[3][0] * G6[v][0] + t1[3][1] * G6[v][1] + t1[3][2] * G6[v][2]);
                dst[(u * 6 + v) * plane] = o4;
            }
        }
    } else {
        const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
        float U[4][16];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double t1[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int b2 = 0; b2 < 3; ++b2) t1[u][b2] = G[u][0] * g[e][0][b2] + G[u][1] * g[e][1][b2] + G[u][2] * g[e][2][b2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) U[e][u * 4 + v] = (float)(t1[u][0] * G[v][0] + t1[u][1] * G[v][1] + t1[u][2] * G[v][2]);
        }
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) dst[xi * plane] = make_float4(U[0][xi], U[1][xi], U[2][xi], U[3][xi]);
    }
}

// pointwise [N][K] -> [K/4][N][4]; bwd: the data-gradient matrix of a column range, [N/4][kc][4]
__device__ __forceinline__ void pack_pointwise_item(const anoddpm_pack_args &a, int64_t idx)
{
    if (!a.bwd) {
        if (idx >= (int64_t)a.N * a.K) return;
        const int n = (int)(idx % a.N), k = (int)(idx / a.N);
        a.out[((int64_t)(k >> 2) * a.N + n) * 4 + (k & 3)] = a.w[(int64_t)n * a.K + k];
    } else {
        if (idx >= (int64_t)a.N * a.kc) return;
        const int o = (int)(idx % a.kc), n = (int)(idx / a.kc);
        a.out[((int64_t)(n >> 2) * a.kc + o) * 4 + (n & 3)] = a.w[(int64_t)n * a.K + a.k0 + o];
    }
}

// OIHW [N][K][3][3] -> [9][K][N] (stem / head kernels)
__device__ __forceinline__ void pack_small_conv_item(const anoddpm_pack_args &a, int64_t idx)
{
    if (idx >= (int64_t)9 * a.K * a.N) return;
    const int n = (int)(idx % a.N), k = (int)((idx / a.N) % a.K), t = (int)(idx / ((int64_t)a.N * a.K));
    a.out[idx] = a.w[((int64_t)n * a.K + k) * 9 + t];
}

// 256-thread blocks a pack job needs (one item per thread)
inline int64_t pack_job_blocks(const anoddpm_pack_args &a)
{
    int64_t items;
    if (a.kind <= 1 || a.kind == 5) items = (int64_t)a.N * a.K / 4;
    else if (a.kind == 2) items = a.bwd ? (int64_t)a.N * a.kc : (int64_t)a.N * a.K;
    else if (a.kind == 3) items = (int64_t)9 * a.N * a.K;
    else items = (int64_t)a.N * a.K;
    return (items + 255) / 256;
}

}  // namespace anoddpm
