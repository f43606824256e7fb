// Winograd F(2x2,3x3) convolution on the fp32 matrix pipe (cfg = 2 of anoddpm_igemm).
//
// Same contract as the implicit-GEMM path (igemm.hip) for 3x3 / stride 1 / pad 1 layers -- replaces
// nn.Conv2d 3x3 (UNet.py:172,193) with GroupNorm-apply + SiLU (UNet.py:170-171,190-191), nearest-x2 (UNet.py:89),
// torch.cat (UNet.py:402), bias, time-embedding add, residual and the fused GroupNorm statistics -- but the
// contraction runs in the Winograd domain:   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   per 2x2 output tile,
// i.e. 16 independent GEMMs  M_xi[tile][n] = sum_c V_xi[tile][c] * U_xi[c][n]  with 4 instead of 9 multiplies per
// output pixel and input channel: 2.25x fewer MFMAs.  fp32 throughout; the transforms only add/subtract
// (weights are pre-transformed on the host, the 1/2 factors live there), measured error vs the direct form
// is ~1e-6 of the tensor's magnitude.
//
// Mapping to gfx950.  The limit is accumulator capacity: a wave must hold 16 transform positions of its
// tile.  One wave per SIMD (512 registers per lane): wave tile = 32 tiles x 32 output channels x 16 positions
// = 256 accumulator registers.  Workgroup = 4 waves (2x2) = 64 Winograd tiles (a 16x16 output patch) x 64
// output channels; K advances 16 channels per iteration:
//   S1  activated 18x18 input patch (affine + SiLU, zero padding, nearest-x2 / concat on the load) -> LDS
//       and the pre-transformed weight tile U[16][16 ch][64] -> LDS         (both prefetched into registers
//       one iteration ahead, so their HBM/L2 latency sits behind the previous iteration's MFMAs)
//   S2  input transform B^T d B: one thread per (tile, channel quad), 16 ds_read_b128 -> 32 float4 adds ->
//       16 ds_write_b128 into V[16][64 tiles][16 ch] (XOR-swizzled by tile so the MFMA operand reads are
//       bank-conflict free)
//   S3  128 MFMAs per wave (16 positions x K=16), operands by ds_read_b128, K permuted as in igemm.hip
// Epilogue: output transform A^T M A in registers, then bias / time embedding / residual / statistics.
#include "common.h"

using anoddpm::silu_f;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WKC = 16;            // channels per K iteration
constexpr int WBN = 64;            // output channels per workgroup
constexpr int WTILES = 64;         // Winograd tiles per workgroup (8 x 8 tiles = 16 x 16 output pixels)
constexpr int WPATCH = 18 * 18;    // input patch pixels

__device__ __forceinline__ f32x4 wld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

__global__ __launch_bounds__(256, 1) void wino_kernel(const anoddpm_igemm_args a)
{
    // LDS (floats): Dt[324][16] | V[16][64][16] | U[16][4][64][4]
    constexpr int DT_F = WPATCH * WKC, V_F = 16 * WTILES * WKC, U_F = 16 * (WKC / 4) * WBN * 4;
    __shared__ __attribute__((aligned(16))) float lds[DT_F + V_F + U_F];
    f32x4 *ldsD = reinterpret_cast<f32x4 *>(lds);                 // [pixel][4 quads]
    f32x4 *ldsV = reinterpret_cast<f32x4 *>(lds + DT_F);          // [xi][tile][4 quads], quad ^= (tile>>2)&3
    f32x4 *ldsU = reinterpret_cast<f32x4 *>(lds + DT_F + V_F);    // [xi][k4][n]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    const int H = a.H, W = a.W;
    const int K = a.c0 + a.c1, N = a.N, K4 = K >> 2;
    const int bx = blockIdx.x % (W >> 4), by = blockIdx.x / (W >> 4);
    const int y0 = by * 16, x0 = bx * 16;                          // output patch origin
    const int n0 = blockIdx.y * WBN;
    const int b = blockIdx.z;
    const int a_mode = a.a_mode;

    const float *A0 = a.a0 + (int64_t)b * a.a0_bs;
    const float *A1 = a.a1 ? a.a1 + (int64_t)b * a.a1_bs : nullptr;
    const float *gsc = a.gn_scale ? a.gn_scale + (int64_t)b * a.gn_ld : nullptr;
    const float *gsh = a.gn_shift ? a.gn_shift + (int64_t)b * a.gn_ld : nullptr;
    const bool affine = (gsc != nullptr);
    const bool act = a.act != 0;
    const int nchunks = K / WKC;

    // ---- S1 geometry: patch slots of this thread (pixel = idx>>2, quad = idx&3), fixed for the workgroup
    constexpr int PJ = (WPATCH * 4 + 255) / 256;                   // 6 slots
    int spix[PJ];
    const int pq = tid & 3;
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int idx = tid + j * 256;
        const int p = idx >> 2;
        const int py = p / 18, px = p - py * 18;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        int sp = -1;
        if (p < WPATCH && gy >= 0 && gy < H && gx >= 0 && gx < W)
            sp = (a_mode == 0) ? gy * W + gx : (gy >> 1) * (W >> 1) + (gx >> 1);
        spix[j] = sp;
    }
    f32x4 praw[PJ];
    f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
    auto load_patch = [&](int chunk) {                              // unconditional loads, clamped addresses
        const int kbase = chunk * WKC;
        const float *src;
        int ld, koff;
        if (kbase < a.c0) { src = A0; ld = a.a0_ld; koff = kbase; }
        else              { src = A1; ld = a.a1_ld; koff = kbase - a.c0; }
        src += koff + pq * 4;
        if (affine) { asc = wld4(gsc + kbase + pq * 4); ash = wld4(gsh + kbase + pq * 4); }
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int sp = spix[j] >= 0 ? spix[j] : 0;
            praw[j] = wld4(src + (int64_t)sp * ld);
        }
    };
    auto store_patch = [&]() {                                      // transform, zero padding AFTER it
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int idx = tid + j * 256;
            if (idx < WPATCH * 4) {
                f32x4 v = praw[j];
                if (affine) v = v * asc + ash;
                if (act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
                ldsD[idx] = spix[j] >= 0 ? v : zero;
            }
        }
    };

    // ---- U tile: thread owns (k4 = tid>>6, n = tid&63) of every transform position
    f32x4 ureg[16];
    const int un = tid & 63, uk4 = tid >> 6;
    const int unc = n0 + un < N ? n0 + un : N - 1;                  // N tail: clamped, columns discarded later
    auto load_U = [&](int chunk) {
        const float *base = a.bmat + (((int64_t)(chunk * (WKC / 4) + uk4)) * N + unc) * 4;
        const int64_t xi_stride = (int64_t)K4 * N * 4;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) ureg[xi] = wld4(base + xi * xi_stride);
    };
    auto store_U = [&]() {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) ldsU[xi * 256 + tid] = ureg[xi];   // [xi][k4][n] with k4*64+n == tid
    };

    // ---- S2: input transform, one thread per (tile, quad)
    const int ttile = tid >> 2, tquad = tid & 3;
    const int tty = ttile >> 3, ttx = ttile & 7;
    auto input_transform = [&]() {
        // rows: t = B^T d   (B^T x = [x0-x2, x1+x2, x2-x1, x1-x3]); one patch column at a time keeps the
        // live set at 16 + 4 float4
        f32x4 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 d0 = ldsD[((2 * tty + 0) * 18 + 2 * ttx + j) * 4 + tquad];
            const f32x4 d1 = ldsD[((2 * tty + 1) * 18 + 2 * ttx + j) * 4 + tquad];
            const f32x4 d2 = ldsD[((2 * tty + 2) * 18 + 2 * ttx + j) * 4 + tquad];
            const f32x4 d3 = ldsD[((2 * tty + 3) * 18 + 2 * ttx + j) * 4 + tquad];
            t[0][j] = d0 - d2;
            t[1][j] = d1 + d2;
            t[2][j] = d2 - d1;
            t[3][j] = d1 - d3;
        }
        // columns: V = t B
        const int sw = (tquad ^ ((ttile >> 2) & 3));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 v0 = t[u][0] - t[u][2];
            const f32x4 v1 = t[u][1] + t[u][2];
            const f32x4 v2 = t[u][2] - t[u][1];
            const f32x4 v3 = t[u][1] - t[u][3];
            ldsV[((u * 4 + 0) * WTILES + ttile) * 4 + sw] = v0;
            ldsV[((u * 4 + 1) * WTILES + ttile) * 4 + sw] = v1;
            ldsV[((u * 4 + 2) * WTILES + ttile) * 4 + sw] = v2;
            ldsV[((u * 4 + 3) * WTILES + ttile) * 4 + sw] = v3;
        }
    };

    // ---- accumulators: 16 transform positions of a 32 tiles x 32 channels wave tile
    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

    const int atile = wm * 32 + l31;                                // this lane's A row (tile)
    const int asw = (atile >> 2) & 3;
    const int bcol = wn * 32 + l31;                                 // this lane's B column (channel)

    load_patch(0);
    load_U(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        store_patch();
        store_U();
        __syncthreads();
        input_transform();
        if (chunk + 1 < nchunks) { load_patch(chunk + 1); load_U(chunk + 1); }   // in flight behind the MFMAs
        __syncthreads();
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
#pragma unroll
            for (int k8 = 0; k8 < WKC / 8; ++k8) {
                const int q = k8 * 2 + h;
                const f32x4 av = ldsV[(xi * WTILES + atile) * 4 + (q ^ asw)];
                const f32x4 bv = ldsU[(xi * (WKC / 4) + q) * WBN + bcol];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc[xi], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: output transform A^T M A (A^T m = [m0+m1+m2, m1-m2-m3]) + bias / temb / residual / stats
    const int n = n0 + bcol;
    const bool nok = n < N;
    const int nc = nok ? n : 0;
    float *__restrict__ O = a.out + (int64_t)b * a.o_bs;
    const float *__restrict__ R = a.res ? a.res + (int64_t)b * a.r_bs : nullptr;
    float add = 0.f;
    if (a.bias) add += a.bias[nc];
    if (a.temb) add += a.temb[(int64_t)b * a.temb_ld + nc];
    float cs = 0.f, cq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int tile = wm * 32 + row;
        const int oy = y0 + 2 * (tile >> 3), ox = x0 + 2 * (tile & 7);
        float tm[2][4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            tm[0][v] = acc[0 * 4 + v][r] + acc[1 * 4 + v][r] + acc[2 * 4 + v][r];
            tm[1][v] = acc[1 * 4 + v][r] - acc[2 * 4 + v][r] - acc[3 * 4 + v][r];
        }
        float y[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            y[i][0] = tm[i][0] + tm[i][1] + tm[i][2];
            y[i][1] = tm[i][1] - tm[i][2] - tm[i][3];
        }
        float rv[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                rv[i][j] = R ? R[((int64_t)(oy + i) * W + ox + j) * a.res_ld + nc] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float v = a.alpha * y[i][j] + add + rv[i][j];
                if (nok) {
                    O[((int64_t)(oy + i) * W + ox + j) * a.out_ld + nc] = v;
                    cs += v;
                    cq += v * v;
                }
            }
    }
    if (a.stats) {
        float *st = a.stats + ((int64_t)b * (gridDim.x * 2) + blockIdx.x * 2 + wm) * N * 2;
        const float s2 = cs + __shfl_xor(cs, 32);
        const float q2 = cq + __shfl_xor(cq, 32);
        if (h == 0 && nok) { st[n * 2] = s2; st[n * 2 + 1] = q2; }
    }
}

}  // namespace

namespace anoddpm {

// Called by anoddpm_igemm for cfg == 2 (arguments already validated there).
int launch_winograd(const anoddpm_igemm_args *a, hipStream_t s)
{
    ANODDPM_REQUIRE(a->ks == 3 && a->b_mode == 0 && a->heads == 1 && a->ksplit == 1, "winograd: needs a 3x3 conv, packed weights, no split-K");
    ANODDPM_REQUIRE(a->a_mode == 0 || a->a_mode == 1, "winograd: pooled operand loads use the direct kernel");
    ANODDPM_REQUIRE(a->H % 16 == 0 && a->W % 16 == 0, "winograd: H and W must be multiples of 16");
    const int K = a->c0 + a->c1;
    ANODDPM_REQUIRE(K % WKC == 0 && (a->c1 == 0 || a->c0 % WKC == 0), "winograd: channel counts must be multiples of 16");
    dim3 grid((unsigned)((a->H / 16) * (a->W / 16)), (unsigned)((a->N + WBN - 1) / WBN), (unsigned)a->B);
    ANODDPM_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "winograd: grid too large");
    hipLaunchKernelGGL(wino_kernel, grid, dim3(256), 0, s, *a);
    return check_launch("winograd");
}

}  // namespace anoddpm
