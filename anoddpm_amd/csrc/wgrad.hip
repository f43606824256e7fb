// Weight gradient of the 3x3 convolutions (backward twin of anoddpm_igemm / winograd for nn.Conv2d 3x3, UNet.py:172,193):
//   dW[co][ci][ky][kx] = sum_{b,y,x} dY[b][y][x][co] * A[b][y+ky-1][x+kx-1][ci]
// where A is the conv's INPUT as the forward consumed it: GroupNorm-apply + SiLU (UNet.py:170-171,190-191), nearest-x2
// (UNet.py:89) and torch.cat (UNet.py:402) are re-applied on the operand load, so the activated tensor never exists
// in HBM in the backward pass either.  (Reference: torch autograd of F.conv2d, diffusion_training.py:102 loss.backward().)
//
// GEMM view: for each tap, M = ci, N = co, K = pixels -- the contraction index is the PIXEL, so both operands are
// pixel-major in NHWC and a lane's MFMA operand is one dword of an LDS row (conflict-free: 32 consecutive channels).
// One workgroup (4 waves, 2x2) owns a 64 ci x 64 co tile of all NINE taps (9 accumulators x 16 = 144 VGPRs per wave)
// and walks a band of image rows; per row it stages ONE new activated input row (rolling 4-slot window: rows
// y-1, y, y+1 are resident) and one dY row, then issues 9 MFMAs per pixel pair with immediate LDS offsets.
// Split-K over (image, column segment, row band): partial tiles go to a workspace, a second kernel folds them in a
// fixed order into the OIHW gradient (deterministic).  fp32 throughout (v_mfma_f32_32x32x2_f32).
#include "common.h"
#include "pack_items.h"

using anoddpm::silu_f;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CT = 64;              // channel tile (ci and co)
constexpr int AP = CT + 4;          // floats per pixel in the LDS rows (pad: keeps float4 stores 16-byte aligned)

template <int TW>                    // output pixels per row segment (2 ... 32)
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const anoddpm_wgrad_args a, const int nseg, const int nband)
{
    constexpr int AW = TW + 2;                                       // staged input row incl. halo
    constexpr int A_F4 = AW * (CT / 4), D_F4 = TW * (CT / 4);        // float4 per staged row
    constexpr int AJ = (A_F4 + 255) / 256, DJ = (D_F4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float ldsA[4][AW][AP];
    __shared__ __attribute__((aligned(16))) float ldsD[2][TW][AP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int K = a.c0 + a.c1, N = a.N, H = a.H, W = a.W;
    const int ntn = (N + CT - 1) / CT;
    const int ci0 = (blockIdx.x / ntn) * CT, co0 = (blockIdx.x % ntn) * CT;
    // work item = (image, column segment, row band)
    const int item = blockIdx.y;
    const int seg = item % nseg;
    const int bnd = (item / nseg) % nband;
    const int b = item / (nseg * nband);
    const int x0 = seg * TW;
    const int ya = bnd * a.band;
    const int yb = ya + a.band < H ? ya + a.band : H;

    // ---- staging roles (fixed per thread): quad = idx & 15, pixel = idx >> 4
    const int q = tid & 15;
    const int cch = ci0 + q * 4;                                     // this thread's input channels
    const bool cok = cch < K;
    const bool from0 = cch < a.c0;
    const float *asrc = !cok ? a.a0 : (from0 ? a.a0 + (int64_t)b * a.a0_bs + cch : a.a1 + (int64_t)b * a.a1_bs + (cch - a.c0));
    const int ald = from0 ? a.a0_ld : a.a1_ld;
    f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
    if (a.gn_scale && cok) {
        asc = *reinterpret_cast<const f32x4 *>(a.gn_scale + (int64_t)b * a.gn_ld + cch);
        ash = *reinterpret_cast<const f32x4 *>(a.gn_shift + (int64_t)b * a.gn_ld + cch);
    }
    const bool affine = a.gn_scale != nullptr, act = a.act != 0;
    const int dch = co0 + q * 4;                                     // this thread's dY channels
    const bool dok = dch < N;
    const float *dsrc = dok ? a.dy + (int64_t)b * a.dy_bs + dch : a.dy;
    const int Ws = a.a_mode == 1 ? (W >> 1) : W;

    f32x4 areg[AJ], dreg[DJ];
    auto activate = [&](f32x4 v) {
        if (affine) v = v * asc + ash;
        if (act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
        return v;
    };
    auto load_act_row = [&](int gy) {                                // activated input row gy (cols x0-1 .. x0+TW) -> registers
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int idx = tid + j * 256;
            const int gx = x0 - 1 + (idx >> 4);
            const bool ok = cok && idx < A_F4 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            f32x4 v;
            if (a.a_mode == 2) {                                     // forward: 2x2 average of four ACTIVATED source pixels
                const int64_t o = ok ? ((int64_t)(2 * gy) * (2 * W) + 2 * gx) * ald : 0;
                const int64_t down = ok ? (int64_t)(2 * W) * ald : 0, right = ok ? ald : 0;
                v = activate(*reinterpret_cast<const f32x4 *>(asrc + o)) + activate(*reinterpret_cast<const f32x4 *>(asrc + o + right));
                v += activate(*reinterpret_cast<const f32x4 *>(asrc + o + down)) + activate(*reinterpret_cast<const f32x4 *>(asrc + o + down + right));
                v *= 0.25f;
            } else {
                const int sy = a.a_mode == 1 ? (gy >> 1) : gy, sx = a.a_mode == 1 ? (gx >> 1) : gx;
                const int64_t off = ok ? ((int64_t)sy * Ws + sx) * ald : 0;
                v = activate(*reinterpret_cast<const f32x4 *>(asrc + off));
            }
            areg[j] = ok ? v : zero;                                 // zero padding AFTER the transform
        }
    };
    auto store_act_row = [&](int gy) {
        const int slot = (gy + 1) & 3;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int idx = tid + j * 256;
            if (idx < A_F4) *reinterpret_cast<f32x4 *>(&ldsA[slot][idx >> 4][q * 4]) = areg[j];
        }
    };
    auto load_dy_row = [&](int gy) {
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int idx = tid + j * 256;
            const bool ok = dok && idx < D_F4 && gy < H;
            const int64_t off = ok ? ((int64_t)gy * W + x0 + (idx >> 4)) * a.dy_ld : 0;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(dsrc + off);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            dreg[j] = ok ? v : zero;
        }
    };
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};                               // column sums of dY over this item's pixels (bias / temb gradient)
    auto store_dy_row = [&](int gy) {
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int idx = tid + j * 256;
            if (idx < D_F4) {
                *reinterpret_cast<f32x4 *>(&ldsD[gy & 1][idx >> 4][q * 4]) = dreg[j];
                csum += dreg[j];
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // prologue: rows ya-1 and ya resident, row ya+1 and dY row ya in registers
    load_act_row(ya - 1);
    store_act_row(ya - 1);
    load_act_row(ya);
    store_act_row(ya);
    load_act_row(ya + 1);
    load_dy_row(ya);

    const float *Abase = &ldsA[0][h][wm * 32 + l31];                 // + slot row, + (2*kp + dx + 1) pixels
    const float *Dbase = &ldsD[0][h][wn * 32 + l31];
    for (int y = ya; y < yb; ++y) {
        store_act_row(y + 1);                                        // slot of row y-3: every reader passed the last barrier
        store_dy_row(y);
        __syncthreads();
        load_act_row(y + 2);                                         // in flight behind this row's MFMAs
        load_dy_row(y + 1);
        const float *Ar[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) Ar[dy] = Abase + ((y + dy) & 3) * (AW * AP);      // rows y-1, y, y+1: slot (row+1)&3
        const float *Dr = Dbase + (y & 1) * (TW * AP);
#pragma unroll
        for (int kp = 0; kp < TW / 2; ++kp) {
            const float bv = Dr[(2 * kp) * AP];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float av = Ar[dy][(2 * kp + dx) * AP];     // pixel x0 + 2kp + h + (dx - 1): halo index +1
                    acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[dy * 3 + dx], 0, 0, 0);
                }
        }
    }

    // ---- column sums: the ci-tile-0 workgroups publish sum_p dY[p][co] of their item (16 threads share a channel quad)
    if (a.colsum && ci0 == 0) {
        __syncthreads();
        float *red = &ldsA[0][0][0];                                 // 256 x 4 floats of scratch
        *reinterpret_cast<f32x4 *>(red + tid * 4) = csum;
        __syncthreads();
        if (tid < CT) {
            const int qq = tid >> 2, e = tid & 3;
            float s = 0.f;
            for (int k = 0; k < 16; ++k) s += red[(k * 16 + qq) * 4 + e];
            if (co0 + tid < N) a.colsum[(int64_t)item * N + co0 + tid] = s;
        }
    }

    // ---- partial tile of this work item: ws[item][tap][ci][co]
    float *wsp = a.ws + (int64_t)item * 9 * K * N;
    const int co = co0 + wn * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ci < K && co < N) wsp[((int64_t)t * K + ci) * N + co] = acc[t][r];
        }
}

// dW (OIHW) = sum over work items of ws[item][tap][ci][co], fixed order.  Thread = (ci, co) x one tap ROW (blockIdx.y):
// three times the parallelism of a nine-tap thread, reads coalesced over co, sixteen items in flight per tap.
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const anoddpm_wgrad_args a, const int nitems)
{
    const int K = a.c0 + a.c1, N = a.N;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)K * N) return;
    const int co = (int)(idx % N), ci = (int)(idx / N);
    const int t0 = blockIdx.y * 3;
    const int64_t plane = (int64_t)K * N, item = 9 * plane;
    const float *p = a.ws + ((int64_t)t0 * K + ci) * N + co;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    int it = 0;
    // The loop is latency-bound (a 128 x 128 layer has 192 workgroups and 128 items): 48 independent loads in flight per thread,
    // added in item order (the result does not depend on the unroll factor).
    for (; it + 16 <= nitems; it += 16) {
        float v[16][3];
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int t = 0; t < 3; ++t) v[u][t] = p[(int64_t)(it + u) * item + t * plane];
#pragma unroll
        for (int u = 0; u < 16; ++u) { s0 += v[u][0]; s1 += v[u][1]; s2 += v[u][2]; }
    }
    for (; it + 4 <= nitems; it += 4) {
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 3; ++t) v[u][t] = p[(int64_t)(it + u) * item + t * plane];
#pragma unroll
        for (int u = 0; u < 4; ++u) { s0 += v[u][0]; s1 += v[u][1]; s2 += v[u][2]; }
    }
    for (; it < nitems; ++it) {
        s0 += p[(int64_t)it * item];
        s1 += p[(int64_t)it * item + plane];
        s2 += p[(int64_t)it * item + 2 * plane];
    }
    float *o = a.dw + ((int64_t)co * K + ci) * 9 + t0;
    if (a.accumulate) { s0 += o[0]; s1 += o[1]; s2 += o[2]; }
    o[0] = s0; o[1] = s1; o[2] = s2;
}

// The same fold for K % 8 == 0, N % 32 == 0 (every convolution of the shipped models) with the transposition to OIHW done in LDS:
// thread = (ci, co) of an 8 x 32 tile, all nine taps, every load of its items in flight (lanes run along co: 128-byte rows);
// the tile's results leave as 32 runs of 72 contiguous floats (co, ci0..ci0+7, 9 taps).  The row kernel above writes -- and
// with `accumulate` re-reads -- 12 bytes per lane at a stride of 36*K bytes: every 128-byte line of dW is touched by a dozen
// waves on different XCDs, and on the 512-channel layers that costs more than reading the 75 MB of partial slabs.
// Sums run in item order per (ci, co, tap) exactly like the row kernel (bit-identical results).
__global__ __launch_bounds__(256) void wgrad_fold_tile_kernel(const anoddpm_wgrad_args a, const int nitems)
{
    __shared__ float tile[32][8 * 9 + 1];
    const int K = a.c0 + a.c1, N = a.N;
    const int tiles_n = N >> 5;
    const int ci0 = (blockIdx.x / tiles_n) * 8, co0 = (blockIdx.x % tiles_n) * 32;
    const int cl = threadIdx.x >> 5, ol = threadIdx.x & 31;
    const int64_t plane = (int64_t)K * N, item = 9 * plane;
    const float *p = a.ws + (int64_t)(ci0 + cl) * N + co0 + ol;
    float s[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] = 0.f;
    int it = 0;
    for (; it + 4 <= nitems; it += 4) {
        float v[4][9];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 9; ++t) v[u][t] = __builtin_nontemporal_load(p + (int64_t)(it + u) * item + t * plane);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 9; ++t) s[t] += v[u][t];
    }
    for (; it < nitems; ++it)
#pragma unroll
        for (int t = 0; t < 9; ++t) s[t] += p[(int64_t)it * item + t * plane];
#pragma unroll
    for (int t = 0; t < 9; ++t) tile[ol][cl * 9 + t] = s[t];
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 72; i += 256) {
        const int o = i / 72, r = i - o * 72;
        float *dst = a.dw + ((int64_t)(co0 + o) * K + ci0) * 9 + r;
        const float v = tile[o][r];
        *dst = a.accumulate ? *dst + v : v;
    }
}

// Weight packing for the 3x3 kernels on the device (training re-packs after every optimizer step): OIHW ->
//   mode 0: direct layout  [9 taps][I/4][O][4]                    (unet.py:_pack_conv)
//   mode 1: Winograd       [16 xi = 4u+v][I/4][O][4], U = G g G^T  (unet.py:_pack_wino; fp64 like the host version)
// bwd != 0 packs the data-gradient weights W'[o=k][i=n][a][b] = w[n][k][2-a][2-b].  Thread = (o, i).
__global__ __launch_bounds__(256) void pack_conv3x3_kernel(const float *__restrict__ w, float *__restrict__ out,
                                                           int N, int K, int mode, int bwd)
{
    anoddpm::pack_conv3x3_item(w, out, N, K, mode, bwd, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

}  // namespace

extern "C" int anoddpm_conv3x3_wgrad(const anoddpm_wgrad_args *a, void *stream)
{
    using namespace anoddpm;
    ANODDPM_REQUIRE(a && a->a0 && a->dy && a->dw && a->ws, "wgrad: null pointer");
    const int K = a->c0 + a->c1;
    ANODDPM_REQUIRE(a->c0 > 0 && a->c0 % 4 == 0 && a->c1 >= 0 && a->c1 % 4 == 0 && (a->c1 == 0 || a->a1), "wgrad: channel counts must be multiples of 4");
    ANODDPM_REQUIRE(a->N >= 1 && a->N % 4 == 0 && a->B >= 1 && a->H >= 1 && a->W >= 2 && a->band >= 1, "wgrad: bad sizes");
    ANODDPM_REQUIRE(a->a_mode == 0 || a->a_mode == 2 || (a->a_mode == 1 && a->H % 2 == 0 && a->W % 2 == 0), "wgrad: a_mode must be 0, 1 or 2");
    ANODDPM_REQUIRE(a->a0_ld % 4 == 0 && (a->c1 == 0 || a->a1_ld % 4 == 0) && a->dy_ld % 4 == 0 && (a->a0_bs | a->a1_bs | a->dy_bs) % 4 == 0,
                    "wgrad: strides must be multiples of 4 floats");
    ANODDPM_REQUIRE(!a->gn_scale || (a->gn_shift && a->gn_ld % 4 == 0), "wgrad: bad GroupNorm affine");
    ANODDPM_REQUIRE(a->algo == 0 || a->algo == 1, "wgrad: algo must be 0 (direct) or 1 (Winograd F(4x4,3x3) domain)");
    if (a->algo == 1) return launch_wgrad43(a, as_stream(stream));
    const int TW = a->W % 32 == 0 ? 32 : (a->W % 16 == 0 ? 16 : (a->W % 8 == 0 ? 8 : (a->W % 4 == 0 ? 4 : 2)));
    ANODDPM_REQUIRE(a->W % TW == 0, "wgrad: W must be even");
    const int nseg = a->W / TW, nband = (a->H + a->band - 1) / a->band;
    const int64_t nitems = (int64_t)a->B * nseg * nband;
    ANODDPM_REQUIRE(nitems <= 65535, "wgrad: too many work items (raise band)");
    ANODDPM_REQUIRE(a->ws_floats >= nitems * 9 * K * a->N, "wgrad: workspace too small");
    const int tiles = ((K + CT - 1) / CT) * ((a->N + CT - 1) / CT);
    hipStream_t s = as_stream(stream);
    if (TW == 32)      hipLaunchKernelGGL(wgrad_kernel<32>, dim3(tiles, (unsigned)nitems), dim3(256), 0, s, *a, nseg, nband);
    else if (TW == 16) hipLaunchKernelGGL(wgrad_kernel<16>, dim3(tiles, (unsigned)nitems), dim3(256), 0, s, *a, nseg, nband);
    else if (TW == 8)  hipLaunchKernelGGL(wgrad_kernel<8>, dim3(tiles, (unsigned)nitems), dim3(256), 0, s, *a, nseg, nband);
    else if (TW == 4)  hipLaunchKernelGGL(wgrad_kernel<4>, dim3(tiles, (unsigned)nitems), dim3(256), 0, s, *a, nseg, nband);
    else               hipLaunchKernelGGL(wgrad_kernel<2>, dim3(tiles, (unsigned)nitems), dim3(256), 0, s, *a, nseg, nband);
    const int64_t kn = (int64_t)K * a->N;
    // (a 32 x 32 tile with 16-byte loads along co was measured slower: 2.86 vs 2.63 ms per step for the class -- a quarter of the
    // workgroups, and the 512-channel layers have only 8 items to keep in flight)
    if (K % 8 == 0 && a->N % 32 == 0 && g_debug[8] != 1)              // ANODDPM_DEBUG8=1: the row kernel everywhere
        hipLaunchKernelGGL(wgrad_fold_tile_kernel, dim3((unsigned)(kn / 256)), dim3(256), 0, s, *a, (int)nitems);
    else
        hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)((kn + 255) / 256), 3), dim3(256), 0, s, *a, (int)nitems);
    return check_launch("conv3x3_wgrad");
}

extern "C" int anoddpm_wgrad43_groups(int32_t K, int32_t N, int32_t B, int32_t H, int32_t W)
{
    return anoddpm::wgrad43_groups(K, N, B, H, W);
}

extern "C" int anoddpm_wgrad43_colsum_items(int32_t K, int32_t N, int32_t B, int32_t H, int32_t W)
{
    return anoddpm::wgrad43_colsum_items(K, N, B, H, W);
}

extern "C" int anoddpm_pack_conv3x3(const float *w, float *out, int32_t N, int32_t K, int32_t mode, int32_t bwd, void *stream)
{
    using namespace anoddpm;
    ANODDPM_REQUIRE(w && out, "pack_conv3x3: null pointer");
    ANODDPM_REQUIRE(N >= 1 && K >= 1 && mode >= 0 && mode <= 2, "pack_conv3x3: bad arguments");
    ANODDPM_REQUIRE((bwd ? N : K) % 4 == 0, "pack_conv3x3: input channel count must be a multiple of 4");
    const int64_t total = (int64_t)N * K / 4;
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w, out, N, K, mode, bwd);
    return check_launch("pack_conv3x3");
}
