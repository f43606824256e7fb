// Winograd F(4x4,3x3) convolution on the fp32 matrix pipe (cfg = 3 of anoddpm_igemm) -- the large-map twin of winograd.hip.
//
// Same contract as cfg 2 (nn.Conv2d 3x3 / stride 1 / pad 1, UNet.py:172,193, with GroupNorm-apply + SiLU, nearest-x2, two-source
// concat fused into the operand load and bias / time-embedding / residual / GroupNorm statistics in the epilogue), but a
// 4x4 output tile costs 36 multiplies instead of 144: 4x fewer MFMAs than the direct form, 1.78x fewer than F(2x2,3x3).
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A,   B^T 6x6, G 6x3, A^T 4x6  (Lavin & Gray, interpolation points 0, +-1, +-2, inf)
// fp32 throughout; the wider transforms cost accuracy -- measured ~8e-6 of the output's magnitude per layer against 4e-7 for
// F(2x2,3x3) -- so the launcher (unet.choose_conv_cfg) uses this kernel only on the large maps where it pays (>= 128x128),
// inside the 1e-3 activation budget of the north star.
//
// Mapping to gfx950.  36 transform positions x (16 tiles x 128 channels) of accumulators is 288 KB: the whole register file of
// a CU minus operands, so ONE 768-thread workgroup (12 waves, 3 per SIMD, <= 168 VGPRs) owns a 16x16 output patch x 128 output
// channels; wave w accumulates positions 3w .. 3w+2 with v_mfma_f32_16x16x4_f32 (16 tiles x 16 channels x 4 k, same rate as the
// 32x32x2 form).  K advances 16 channels per iteration:
//   * the activated 18x18 halo patch is double-buffered in LDS (fetched two iterations ahead, scalar in-loop addressing);
//   * all twelve waves compute B^T d B for the NEXT chunk (768 items = 16 tiles x 8 channel pairs x 6 transform rows) into a
//     double-buffered V[pos][tile][quad] array, so the MFMA A operands are one ds_read_b128 per position;
//   * the B operands U[pos][k][n] stream from L2 as 16-byte loads (layout [36][K/4][N][4]: one load = one lane's four k);
//   * 96 MFMAs per wave per iteration, one barrier per iteration.
// Epilogue: the 36 x 16 x 128 products go through LDS in three rounds of 48 / 48 / 32 channels (120 KB, aliasing the loop
// buffers); one thread per (tile, channel) forms A^T M A, adds bias / temb / residual, stores 16 pixels through buffer stores with
// scalar pixel offsets and folds the GroupNorm sums.
#include "common.h"
#include "gn_fold.h"

using anoddpm::silu_f;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int F4_NT = 768;                 // threads
constexpr int F4_KC = 16;                  // channels per K iteration
constexpr int F4_PW = 18;                  // patch width / height (16 + 2)
constexpr int F4_PPIX = F4_PW * F4_PW;     // 324 patch pixels
constexpr int F4_PITCH = 5;                // float4 per patch pixel (4 quads + 1 pad)
constexpr int F4_PJ = 2;                   // staging slots per thread (2 * 768 = 1536 >= 324 * 4)
constexpr int F4_SLOTPX = F4_PJ * F4_NT / 4;          // 384 pixel slots per buffer
constexpr int F4_DT = F4_SLOTPX * F4_PITCH;           // float4 per patch buffer
constexpr int F4_V = 36 * 16 * 4;                     // float4 per V buffer: [pos][tile][quad]
constexpr int F4_MS = 52;                             // floats per (pos, tile) row of the exchange buffer (48 channels + pad: 4 * 52 = 16 mod 32)
constexpr int F4_LOOP_FLOATS = (2 * F4_DT + 2 * F4_V) * 4;
constexpr int F4_EX_FLOATS = 36 * 16 * F4_MS;
constexpr int F4_LDS_FLOATS = F4_EX_FLOATS > F4_LOOP_FLOATS ? F4_EX_FLOATS : F4_LOOP_FLOATS;
constexpr int F4_KMAX = 1024;                         // input channels whose GroupNorm affine fits the LDS table
constexpr int F4_AFF_FLOATS = 2 * F4_KMAX;            // [K] scales, then [K] shifts of this workgroup's image, behind both uses of `lds`

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc43(const float *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7ffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 bld4(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned wave_bytes)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)wave_bytes, 0));
}

// DBG (timing ablations only, wrong results; ANODDPM_DEBUG2): 1 no epilogue, 2 no input transform, 3 no patch staging, 4 no B requests
// NTL = 16-channel tiles per workgroup: 8 (128 output channels) or 4 (64 channels: twice the workgroups when the map is too
// small to fill the 256 CUs with 128-channel ones, at the price of repeating the staging / input transform per 64 channels)
#ifndef F43_PAIR_TRANSFORM
#define F43_PAIR_TRANSFORM 0
#endif
template <bool FAST, int DBG = 0, int NTL = 8>
__global__ __launch_bounds__(F4_NT, 1) void wino43_kernel(const anoddpm_igemm_args a)
{
    __shared__ __attribute__((aligned(16))) float lds[F4_LDS_FLOATS + F4_AFF_FLOATS];
    f32x4 *ldsD = reinterpret_cast<f32x4 *>(lds);
    f32x4 *ldsV = ldsD + 2 * F4_DT;
    f32x4 *ldsAff = reinterpret_cast<f32x4 *>(lds + F4_LDS_FLOATS);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    const int K = a.c0 + a.c1, N = a.N, K4 = K >> 2;
    const int tiles_x = W >> 4;
    const int y0 = (blockIdx.x / tiles_x) * 16, x0 = (blockIdx.x % tiles_x) * 16;
    const int n0 = blockIdx.y * (16 * NTL);
    const int b = blockIdx.z;
    const int a_mode = a.a_mode;

    const float *A0 = a.a0 + (int64_t)b * a.a0_bs;
    const float *A1 = a.a1 ? a.a1 + (int64_t)b * a.a1_bs : nullptr;
    const float *gsc = a.gn_scale ? a.gn_scale + (int64_t)b * a.gn_ld : nullptr;
    const float *gsh = a.gn_shift ? a.gn_shift + (int64_t)b * a.gn_ld : nullptr;
    // GroupNorm finished here from fp64 sums (gn_fold.h) -- in the 64-channel variant only: the 128-channel form of THIS kernel is a
    // measurement fallback (ANODDPM_DEBUG5=1; the 128-channel grids run winograd43r.hip) at its 168-register budget
    const bool fold = NTL == 4 && a.fold_gamma != nullptr;
    const bool affine = gsc != nullptr || fold, act = a.act != 0;
    const int nchunks = K / F4_KC;

    // ---- patch staging (pixel = idx >> 2, quad = idx & 3): geometry fixed for the workgroup
    int spix[F4_PJ];
    const int pq = tid & 3;
#pragma unroll
    for (int j = 0; j < F4_PJ; ++j) {
        const int p = (tid + j * F4_NT) >> 2;
        const int py = p / F4_PW, px = p - py * F4_PW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        int sp = -1;
        if (p < F4_PPIX && gy >= 0 && gy < H && gx >= 0 && gx < W)
            sp = (a_mode == 0) ? gy * W + gx : (gy >> 1) * (W >> 1) + (gx >> 1);
        spix[j] = sp;
    }
    f32x4 praw[F4_PJ];
    f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rA0 = rsrc43(A0), rA1 = rsrc43(A1 ? A1 : A0);
    const __amdgpu_buffer_rsrc_t rSc = rsrc43(gsc ? gsc : A0), rSh = rsrc43(gsh ? gsh : A0);
    auto load_patch = [&](int chunk) {                              // unconditional loads, clamped addresses
        const int kbase = chunk * F4_KC;
        const bool first = kbase < a.c0;
        const __amdgpu_buffer_rsrc_t r = first ? rA0 : rA1;
        const unsigned ld = (unsigned)(first ? a.a0_ld : a.a1_ld);
        const unsigned koff = (unsigned)(first ? kbase : kbase - a.c0) * 4u;
#pragma unroll
        for (int j = 0; j < F4_PJ; ++j) {
            const unsigned sp = spix[j] >= 0 ? (unsigned)spix[j] : 0u;
            praw[j] = bld4(r, (sp * ld + (unsigned)(pq * 4)) * 4u, koff);
        }
    };
    auto store_patch = [&](int buf, int chunk) {                    // transform, zero padding AFTER it
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        // the GroupNorm affine of this chunk's channels comes from the LDS table filled in the prologue (round 5).  Until then it
        // was fetched from L2 right here, to keep eight registers free -- but its use two instructions later is an
        // `s_waitcnt vmcnt(0)`, and vmcnt retires in order: every wave drained its B-fragment ring and sat out an L2 round trip in
        // the middle of each chunk's MFMAs
        if (FAST || affine) {
            asc = ldsAff[chunk * 4 + pq];
            ash = ldsAff[K4 + chunk * 4 + pq];
        }
#pragma unroll
        for (int j = 0; j < F4_PJ; ++j) {
            const int idx = tid + j * F4_NT;
            f32x4 v = praw[j];
            if (FAST) {
                v = v * asc + ash;
                v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]);
            } else {
                if (affine) v = v * asc + ash;
                if (act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
            }
            ldsD[buf * F4_DT + (idx >> 2) * F4_PITCH + (idx & 3)] = spix[j] >= 0 ? v : zero;
        }
    };

    // ---- input transform role: ALL twelve waves, 768 items = 16 tiles x 8 channel pairs x 6 transform rows.
    // wave -> (row u = wave % 6, tile-row pair = wave / 6); lane -> (channel pair = lane & 7, tile slot = lane >> 3): the 32 lanes
    // of a ds_read_b64 group then cover one tile row x 8 pairs = 32 distinct 8-byte bank slots, and the 16 lanes of a ds_write_b64
    // group two neighbouring tiles x 8 pairs.  float2 items keep the transform at ~30 live registers.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tu = wave % 6;
    const int tpair = lane & 7;
    const int ttile = ((wave / 6) * 2 + (lane >> 5)) * 4 + ((lane >> 3) & 3);
    const int tbase2 = (((4 * (ttile >> 2)) * F4_PW + 4 * (ttile & 3)) * F4_PITCH) * 2 + tpair;      // float2 index of the tile's patch corner
    // B^T row u as (patch row, coefficient) pairs -- every row of B^T touches at most four patch rows:
    //   u0: 4 d0 - 5 d2 + d4        u1: -4 d1 - 4 d2 + d3 + d4     u2: 4 d1 - 4 d2 - d3 + d4
    //   u3: -2 d1 - d2 + 2 d3 + d4  u4: 2 d1 - d2 - 2 d3 + d4      u5: 4 d1 - 5 d3 + d5
    const int tr0 = (tu == 0) ? 0 : 1, tr1 = (tu == 5) ? 3 : 2, tr2 = (tu == 0) ? 4 : ((tu == 5) ? 5 : 3), tr3 = 4;
    const float tc0 = (tu == 0) ? 4.f : (tu == 1 ? -4.f : (tu == 2 ? 4.f : (tu == 3 ? -2.f : (tu == 4 ? 2.f : 4.f))));
    const float tc1 = (tu == 0 || tu == 5) ? -5.f : ((tu == 1 || tu == 2) ? -4.f : -1.f);
    const float tc2 = (tu == 0 || tu == 5) ? 1.f : (tu == 1 ? 1.f : (tu == 2 ? -1.f : (tu == 3 ? 2.f : -2.f)));
    const float tc3 = (tu == 0 || tu == 5) ? 0.f : 1.f;
    const int to0 = tr0 * F4_PW * F4_PITCH * 2, to1 = tr1 * F4_PW * F4_PITCH * 2, to2 = tr2 * F4_PW * F4_PITCH * 2, to3 = tr3 * F4_PW * F4_PITCH * 2;
    auto transform = [&](int pbuf, int vbuf) {
        const f32x2 *D = reinterpret_cast<const f32x2 *>(ldsD + pbuf * F4_DT) + tbase2;
        f32x2 t[6];
        // stage 1: t[j] = sum_i B^T[u][i] d[i][j]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            t[j] = tc0 * D[to0 + j * F4_PITCH * 2] + tc1 * D[to1 + j * F4_PITCH * 2] + tc2 * D[to2 + j * F4_PITCH * 2] + tc3 * D[to3 + j * F4_PITCH * 2];
            // one column of reads in flight: the register file is full of accumulators (3 waves per SIMD), the LDS latency
            // this exposes is covered by the other waves' MFMAs
            __builtin_amdgcn_sched_barrier(0);
        }
        // stage 2: V[u][v] = sum_j t[j] B^T[v][j];  V layout [pos][tile][quad] float4 = [pos][tile][pair] float2
        const f32x2 p = t[4] - 4.f * t[2], q = t[3] - 4.f * t[1], r = t[4] - t[2], s = t[3] - t[1];
        f32x2 *V = reinterpret_cast<f32x2 *>(ldsV + vbuf * F4_V) + ((tu * 6) * 16 + ttile) * 8 + tpair;
        V[0 * 128] = 4.f * t[0] - 5.f * t[2] + t[4];
        V[1 * 128] = p + q;
        V[2 * 128] = p - q;
        V[3 * 128] = r + 2.f * s;
        V[4 * 128] = r - 2.f * s;
        V[5 * 128] = 4.f * t[1] - 5.f * t[3] + t[5];
    };

    // Round 6, F43_PAIR_TRANSFORM=1 (measurement builds; measured equal to the single-row items, profiles/r6_f43_pair_transform_ab.txt,
    // so NOT the default): row-PAIR items on single channels (the form wgrad43.hip runs; profiles/r6_wgrad43_ab.txt): rows (1,2)
    // and (3,4) of B^T share their partial sums, (0,5) read disjoint patch rows -- 4 operations per column and row pair instead of
    // 4 per row, literal coefficients.  768 items = 16 tiles x 16 channels x 3 row pairs: wave w = (row pair w % 3, tile row w / 3),
    // lane = (tile column lane >> 4, channel lane & 15).
    const int p_up = wave % 3;
    const int p_tx = lane >> 4, p_ch = lane & 15;
    constexpr int PP = F4_PITCH * 4, PROW = F4_PW * PP;             // floats per staged pixel / patch row
    const int p_in = ((4 * (wave / 3)) * F4_PW + 4 * p_tx) * PP + p_ch;
    const int p_out = ((wave / 3) * 4 + p_tx) * 16 + p_ch;          // V[pos][tile][channel]: + pos * 256
    const int p_ua = p_up == 0 ? 0 : (p_up == 1 ? 1 : 3), p_ub = p_up == 0 ? 5 : (p_up == 1 ? 2 : 4);
    auto col_pass = [&](const float (&t)[6], float *V) {
        const float p = t[4] - 4.f * t[2], q = t[3] - 4.f * t[1], r = t[4] - t[2], w = t[3] - t[1];
        V[0 * 256] = 4.f * t[0] - 5.f * t[2] + t[4];
        V[1 * 256] = p + q;
        V[2 * 256] = p - q;
        V[3 * 256] = r + 2.f * w;
        V[4 * 256] = r - 2.f * w;
        V[5 * 256] = 4.f * t[1] - 5.f * t[3] + t[5];
    };
    auto transform_p = [&](int pbuf, int vbuf) {
        const float *D = reinterpret_cast<const float *>(ldsD + pbuf * F4_DT) + p_in;
        float ta[6], tb[6];
        if (p_up == 0) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d0 = D[j * PP], d1 = D[PROW + j * PP], d2 = D[2 * PROW + j * PP], d3 = D[3 * PROW + j * PP], d4 = D[4 * PROW + j * PP], d5 = D[5 * PROW + j * PP];
                ta[j] = 4.f * d0 - 5.f * d2 + d4;
                tb[j] = 4.f * d1 - 5.f * d3 + d5;
                __builtin_amdgcn_sched_barrier(0);                  // one column of reads in flight (see transform())
            }
        } else if (p_up == 1) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d1 = D[PROW + j * PP], d2 = D[2 * PROW + j * PP], d3 = D[3 * PROW + j * PP], d4 = D[4 * PROW + j * PP];
                const float p = d4 - 4.f * d2, q = d3 - 4.f * d1;
                ta[j] = p + q;
                tb[j] = p - q;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d1 = D[PROW + j * PP], d2 = D[2 * PROW + j * PP], d3 = D[3 * PROW + j * PP], d4 = D[4 * PROW + j * PP];
                const float r = d4 - d2, w = d3 - d1;
                ta[j] = r + 2.f * w;
                tb[j] = r - 2.f * w;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float *V = reinterpret_cast<float *>(ldsV + vbuf * F4_V) + p_out;
        col_pass(ta, V + p_ua * 6 * 256);
        col_pass(tb, V + p_ub * 6 * 256);
    };
    auto transform_sel = [&](int pbuf, int vbuf) {
        if (F43_PAIR_TRANSFORM) transform_p(pbuf, vbuf);
        else                    transform(pbuf, vbuf);
    };

    // ---- accumulators: positions 3*wave + {0,1,2} x 8 channel tiles of 16
    f32x4 acc[3][NTL];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) acc[p][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rU = rsrc43(a.bmat);
    const unsigned xi_bytes = (unsigned)K4 * (unsigned)N * 16u;                       // bytes per position of U
    const unsigned ulane = ((unsigned)kq * (unsigned)N + (unsigned)(n0 + l15)) * 16u;  // + nt*256 + chunk*4*N*16 + pos*xi_bytes
    const int vread = (wave * 3) * 64 + l15 * 4 + kq;                                 // + p*64 float4

    // B operands: 24 (position, channel tile) groups of four MFMAs per iteration, streamed through a four-deep register ring
    // (group g + 4 is requested when group g is consumed; the last four requests belong to the next iteration)
    f32x4 ring[4];
    constexpr int NG = 3 * NTL;                                     // (position, channel tile) groups per iteration
    auto load_group = [&](int chunk, int g, int slot) {             // g = p * NTL + nt (compile-time), slot = g % 4
        const unsigned w = (unsigned)(wave * 3 + g / NTL) * xi_bytes + (unsigned)(chunk * 4) * (unsigned)N * 16u;
        ring[slot] = bld4(rU, ulane + (unsigned)(g % NTL) * 256u, w);
    };

    // prologue: patch(0) -> LDS -> V(0); patch(1) -> LDS; patch(2) requested
    const int last = nchunks - 1;
    const int c1 = last >= 1 ? 1 : 0, c2 = last >= 2 ? 2 : last;
    // GroupNorm affine of the image -> LDS, requested first (threads 0 .. K/4-1: one float4 of scales and one of shifts each)
    f32x4 aff_sc = {0.f, 0.f, 0.f, 0.f}, aff_sh = {0.f, 0.f, 0.f, 0.f};
    const bool aff_slot = (FAST || affine) && tid < K4 && !fold;
    anoddpm::FoldLoads fl;
    if (aff_slot) {
        aff_sc = bld4(rSc, (unsigned)(tid * 16), 0u);
        aff_sh = bld4(rSh, (unsigned)(tid * 16), 0u);
    } else if (fold) {
        fl = anoddpm::fold_affine_request(a, b, tid);                 // oldest requests of the workgroup, like the table loads above
    }
    load_patch(0);
    f32x4 praw0[F4_PJ];
#pragma unroll
    for (int j = 0; j < F4_PJ; ++j) praw0[j] = praw[j];
    load_patch(c1);                                                 // both requests in flight: one HBM latency, not two
#pragma unroll
    for (int g = 0; g < 4; ++g) load_group(0, g, g);
    if (FAST || affine) {
        if (fold) {
            // scratch = the V buffers (first written by transform() below, behind two more barriers)
            anoddpm::fold_affine_finish(a, fl, tid, a_mode == 1 ? (H >> 1) * (W >> 1) : H * W, reinterpret_cast<double *>(ldsV), ldsAff);
        } else if (aff_slot) {                                         // oldest requests: no wait for the patches behind them
            ldsAff[tid] = aff_sc;
            ldsAff[K4 + tid] = aff_sh;
        }
        __syncthreads();
    }
    {
        f32x4 keep[F4_PJ];
#pragma unroll
        for (int j = 0; j < F4_PJ; ++j) { keep[j] = praw[j]; praw[j] = praw0[j]; }
        store_patch(0, 0);
#pragma unroll
        for (int j = 0; j < F4_PJ; ++j) praw[j] = keep[j];
    }
    __syncthreads();
    transform_sel(0, 0);
    store_patch(1, c1);
    load_patch(c2);
    __syncthreads();

    // ONE basic block per iteration: no `if (more)` -- the tail iterations transform / re-stage already consumed buffers with
    // clamped chunk indices (valid memory, results unused); branches here would turn the register-carried ring and the 96
    // accumulators into phi copies.  Schedule of iteration c (vmcnt retires in order, so a B request issued after a patch
    // request inherits its HBM latency: patch requests go out late, their stores early in the NEXT iteration):
    //   T  V(c+1) <- patch(c+1)      groups 0..5      S  patch(c+2) -> LDS (requested in iteration c-1)
    //   groups 6..17                 L  request patch(c+3)                groups 18..23      barrier
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (DBG != 2) transform_sel((chunk + 1) & 1, (chunk + 1) & 1);
        const f32x4 *V = ldsV + (chunk & 1) * F4_V + vread;
        const int nxt = chunk < last ? chunk + 1 : last;            // clamped: the tail re-loads valid memory, unused
        const int s2 = chunk + 2 <= last ? chunk + 2 : last, l3 = chunk + 3 <= last ? chunk + 3 : last;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const f32x4 av = V[p * 64];
            __builtin_amdgcn_sched_barrier(0);                      // one position's A fragment live at a time
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                const int g = p * NTL + nt;
                if (g == NG / 4 && DBG != 3) store_patch(chunk & 1, s2); // patch(chunk+2) replaces patch(chunk): its readers passed the last barrier
                if (g == (3 * NG) / 4 && DBG != 3) load_patch(l3);
                const f32x4 bv = ring[g % 4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    acc[p][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv[kk], acc[p][nt], 0, 0, 0);
                if (DBG != 4) {
                    if (g + 4 < NG) load_group(chunk, g + 4, g % 4);
                    else            load_group(nxt, g + 4 - NG, g % 4);
                }
                __builtin_amdgcn_sched_barrier(0);                  // keep the ring at four requests: no hoisting of later loads
            }
        }
        __syncthreads();                                            // publishes V(chunk+1) and patch(chunk+2); retires V(chunk)
    }

    // ---- epilogue: three rounds of <= 48 output channels through LDS
    if (DBG == 1) {
        float sum = 0.f;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) sum += (acc[p][nt][0] + acc[p][nt][1]) + (acc[p][nt][2] + acc[p][nt][3]);
        if (sum == 12345.678f) a.out[0] = sum;
        return;
    }
    // Three rounds over the channel tiles (3 + 3 + 2 of the eight 16-channel tiles): a 48-channel round is 16 tiles x 48 = 768
    // items, exactly one per thread.  Pixel addresses are one per-lane byte offset (tile origin, channel) plus a wave-uniform
    // scalar offset per pixel of the 4x4 tile, so the 16 stores / residual loads of an item carry no vector address arithmetic.
    float *M = lds;
    const float *TE = a.temb ? a.temb + (int64_t)b * a.temb_ld : nullptr;
    const __amdgpu_buffer_rsrc_t rO = rsrc43(a.out + (int64_t)b * a.o_bs);
    const __amdgpu_buffer_rsrc_t rR = rsrc43(a.res ? a.res + (int64_t)b * a.r_bs : a.out);
    const bool has_res = a.res != nullptr;
    const unsigned uW = (unsigned)W, o_ld = (unsigned)a.out_ld, r_ld = (unsigned)a.res_ld;
    constexpr int ROUNDS = NTL == 8 ? 3 : 2;                        // 8 tiles: 3 + 3 + 2; 4 tiles: 2 + 2
    constexpr int RT = NTL == 8 ? 3 : 2;
    // Item geometry of a round: tile = tid / nch, channel = tid % nch (nch = 48, 48, 32 resp. 32, 32).
    const bool res_up = a.res_mode == 1;                            // residual at half resolution, nearest x2 on the read
    const unsigned hW = uW >> 1;
    struct Item { int tile, ch, n; bool active; unsigned vo, vr; };
    auto item_of = [&](int round) {
        const int nch = ((NTL == 8 && round == 2) ? 2 : RT) * 16;
        Item it;
        it.tile = tid / nch;
        it.ch = tid - it.tile * nch;
        it.active = it.tile < 16;                                   // third round: 512 items, waves 8..11 idle (wave-uniform)
        it.n = n0 + round * RT * 16 + it.ch;
        const unsigned pix0 = (unsigned)(y0 + (it.tile >> 2) * 4) * uW + (unsigned)(x0 + (it.tile & 3) * 4);
        it.vo = (pix0 * o_ld + (unsigned)it.n) * 4u;
        it.vr = res_up ? (((((unsigned)(y0 + (it.tile >> 2) * 4) >> 1) * hW + ((unsigned)(x0 + (it.tile & 3) * 4) >> 1)) * r_ld + (unsigned)it.n) * 4u)
                       : ((pix0 * r_ld + (unsigned)it.n) * 4u);
        return it;
    };
    // The residual pixels and the per-channel addend of a round are requested one round AHEAD (round 0: before the accumulators
    // go through LDS): the output stores may alias them as far as the compiler knows, so left next to their use each of the 16
    // loads would wait out its own HBM latency (measured: 256^2 128->128 with residual 310 -> 270 us).
    auto prefetch = [&](const Item &it, float (&rv)[16], float &add) {
        add = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) rv[i] = 0.f;
        if (it.active) {
            if (has_res && res_up) {
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rR, (int)it.vr, (int)((((unsigned)i2 * hW + (unsigned)j2) * 4u) * r_ld), 0));
                        rv[(2 * i2) * 4 + 2 * j2] = v; rv[(2 * i2) * 4 + 2 * j2 + 1] = v;
                        rv[(2 * i2 + 1) * 4 + 2 * j2] = v; rv[(2 * i2 + 1) * 4 + 2 * j2 + 1] = v;
                    }
            } else if (has_res) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        rv[i * 4 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rR, (int)it.vr, (int)((((unsigned)i * uW + (unsigned)j) * 4u) * r_ld), 0));
            }
            if (a.bias) add += a.bias[it.n];
            if (TE) add += TE[it.n];
        }
    };
    float rv[ROUNDS][16], add[ROUNDS], cs[ROUNDS], cq[ROUNDS];
    prefetch(item_of(0), rv[0], add[0]);
#pragma unroll
    for (int round = 0; round < ROUNDS; ++round) {
        const int nt0 = round * RT;
        const int ntn = (NTL == 8 && round == 2) ? 2 : RT;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
                if (nt < ntn) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        M[((wave * 3 + p) * 16 + kq * 4 + r) * F4_MS + nt * 16 + l15] = acc[p][nt0 + nt][r];
                }
        __syncthreads();
        if (round + 1 < ROUNDS) prefetch(item_of(round + 1), rv[round + 1], add[round + 1]);
        const Item it = item_of(round);
        cs[round] = 0.f;
        cq[round] = 0.f;
        if (it.active) {
            const float *m = M + (size_t)it.tile * F4_MS + it.ch;
            // columns first: y[i][v] = sum_u A^T[i][u] m[u][v]
            float y[4][6];
#pragma unroll
            for (int v = 0; v < 6; ++v) {
                float mu[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) mu[u] = m[(size_t)((u * 6 + v) * 16) * F4_MS];
                const float s12 = mu[1] + mu[2], d12 = mu[1] - mu[2], s34 = mu[3] + mu[4], d34 = mu[3] - mu[4];
                y[0][v] = mu[0] + s12 + s34;
                y[1][v] = d12 + 2.f * d34;
                y[2][v] = s12 + 4.f * s34;
                y[3][v] = d12 + 8.f * d34 + mu[5];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s12 = y[i][1] + y[i][2], d12 = y[i][1] - y[i][2], s34 = y[i][3] + y[i][4], d34 = y[i][3] - y[i][4];
                float o4[4];
                o4[0] = y[i][0] + s12 + s34;
                o4[1] = d12 + 2.f * d34;
                o4[2] = s12 + 4.f * s34;
                o4[3] = d12 + 8.f * d34 + y[i][5];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned so = ((unsigned)i * uW + (unsigned)j) * 4u;          // wave-uniform pixel offset (x ld below)
                    const float v = a.alpha * o4[j] + add[round] + rv[round][i * 4 + j];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rO, (int)it.vo, (int)(so * o_ld), 0);
                    cs[round] += v;
                    cq[round] += v * v;
                }
            }
        }
        __syncthreads();                                            // all reads of M done
    }
    if (a.stats || a.stats_csum) {
        // per-channel sums over the workgroup's 256 pixels: ONE statistics row per workgroup, reduced once for all rounds
        float *rs = M, *rq = M + ROUNDS * 16 * 48;
#pragma unroll
        for (int round = 0; round < ROUNDS; ++round) {
            const Item it = item_of(round);
            if (it.active) { rs[(round * 16 + it.tile) * 48 + it.ch] = cs[round]; rq[(round * 16 + it.tile) * 48 + it.ch] = cq[round]; }
        }
        __syncthreads();
        if (tid < NTL * 16) {
            const int round = tid / (RT * 16), chl = tid - round * (RT * 16);
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) { s += rs[(round * 16 + t) * 48 + chl]; q += rq[(round * 16 + t) * 48 + chl]; }
            if (a.stats_csum) {
                anoddpm::csum_atomic_add(a.stats_csum, b, N, n0 + tid, s, q);   // one fp64 pair per workgroup and channel (gn_fold.h)
            } else {
                float *st = a.stats + (((int64_t)b * gridDim.x + blockIdx.x) * N + n0 + tid) * 2;
                st[0] = s;
                st[1] = q;
            }
        }
    }
}

}  // namespace

extern "C" int anoddpm_f43_channel_sliced(int32_t H, int32_t W, int32_t N, int32_t B);

namespace anoddpm {

// Called by anoddpm_igemm for cfg == 3 (common arguments already validated there).
int launch_winograd43(const anoddpm_igemm_args *a, hipStream_t s)
{
    ANODDPM_REQUIRE(a->ks == 3 && a->b_mode == 0 && a->heads == 1, "winograd43: needs a 3x3 conv with packed weights");
    ANODDPM_REQUIRE(a->ksplit == 1 || (a->N % 128 == 0 && a->ws && (int64_t)a->B * a->ksplit <= 65535), "winograd43: split-K needs N %% 128 == 0 and a workspace");
    ANODDPM_REQUIRE(a->a_mode == 0 || a->a_mode == 1, "winograd43: pooled operand loads use the direct kernel");
    ANODDPM_REQUIRE(a->H % 16 == 0 && a->W % 16 == 0 && a->N % 64 == 0, "winograd43: H, W must be multiples of 16 and N of 64");
    const int K = a->c0 + a->c1;
    ANODDPM_REQUIRE(K % F4_KC == 0 && (a->c1 == 0 || a->c0 % F4_KC == 0), "winograd43: channel counts must be multiples of 16");
    ANODDPM_REQUIRE((int64_t)36 * K * a->N * 4 < ((int64_t)1 << 31), "winograd43: transformed weights exceed 32-bit buffer offsets");
    ANODDPM_REQUIRE(!(a->gn_scale || a->fold_gamma) || K <= F4_KMAX, "winograd43: the GroupNorm affine table holds %d input channels", F4_KMAX);
    if (a->fold_gamma) {
        ANODDPM_REQUIRE(a->fold_beta && a->fold_stats0 && (a->c1 == 0 || a->fold_stats1) && a->fold_fmt0 == 1 && (a->c1 == 0 || a->fold_fmt1 == 1),
                        "winograd43: the GroupNorm fold takes fp64 sums (fold_fmt 1: stats_csum / tail_csum sources) only");
        ANODDPM_REQUIRE(a->fold_groups >= 1 && a->fold_groups <= 64 && K % a->fold_groups == 0 && a->ksplit == 1,
                        "winograd43: GroupNorm fold: bad group count, or split-K");
        ANODDPM_REQUIRE(((uintptr_t)a->fold_stats0 | (uintptr_t)a->fold_stats1 | (uintptr_t)a->fold_gamma | (uintptr_t)a->fold_beta) % 16 == 0,
                        "winograd43: GroupNorm fold: sums, gamma and beta must be 16-byte aligned");
    }
    if (a->stats_csum)
        ANODDPM_REQUIRE(!a->stats && a->ksplit == 1 && a->heads == 1, "winograd43: stats_csum excludes stats rows, split-K and heads");
    ANODDPM_REQUIRE((int64_t)a->H * a->W * (a->a0_ld > a->a1_ld ? a->a0_ld : a->a1_ld) * 4 < ((int64_t)1 << 31),
                    "winograd43: operand slice exceeds 32-bit buffer offsets");
    // 64-channel workgroups when 128-channel ones would leave CUs idle (or N is not a multiple of 128): rounds over the 256 CUs x
    // the cost of a workgroup -- a 64-channel one takes 0.71 of a 128-channel one (it transforms the same patch for half the
    // outputs: 68 against 96 us on the 64x64 256 -> 256 layer).  128 workgroups (batch 4): one round of 256 halves, 0.71; 160
    // (batch 5, the detection loop's five chains): one round of 128-channel workgroups, 1.0, against two rounds of halves, 1.42
    const int64_t wg128 = (int64_t)(a->H / 16) * (a->W / 16) * (a->N / 128) * a->B;
    const bool half = (a->N % 128 != 0) || ((2 * wg128 + 255) / 256) * 71 < ((wg128 + 255) / 256) * 100;
    ANODDPM_REQUIRE(!a->gnb_partial || (a->ksplit == 1 && anoddpm_f43_channel_sliced(a->H, a->W, a->N, a->B) == 1),
                    "winograd43: gnb_partial needs the channel-sliced 128-channel kernel (anoddpm_f43_channel_sliced) and ksplit 1");
    const int nblk = half ? 64 : 128;
    dim3 grid((unsigned)((a->H / 16) * (a->W / 16)), (unsigned)(a->N / nblk), (unsigned)a->B);
    ANODDPM_REQUIRE(a->B <= 65535, "winograd43: batch too large");
    const bool fast = (a->gn_scale || a->fold_gamma) && a->act;
    ANODDPM_REQUIRE(a->res_mode == 0 || (a->res_mode == 1 && a->res && a->ksplit == 1), "winograd43: res_mode 1 needs a residual and ksplit 1");
#ifdef ANODDPM_ABLATE
    const int dbg = anoddpm::g_debug[2];
#else
    const int dbg = 0;
#endif
    // 128-channel grid: the channel-sliced kernel (output transform in registers, winograd43r.hip); ANODDPM_DEBUG5=1 keeps this
    // file's position-sliced kernel
    // (ANODDPM_DEBUG5=3: also where this file's 64-channel variant would be chosen -- the op tests reach the kernel on small shapes)
    if (a->ksplit > 1) {
        const int K16 = K / F4_KC, cps = (K16 + a->ksplit - 1) / a->ksplit;
        ANODDPM_REQUIRE((a->ksplit - 1) * cps < K16, "winograd43: ksplit leaves a slice without channels");
        return launch_winograd43r(a, s);
    }
    // (round 5: a persistent form of the channel-sliced kernel -- one workgroup per CU, the step pipeline running across tile
    // boundaries -- measured SLOWER, 307 vs 285 us on the 256x256 128->128 layer: vmcnt retires in order, so the next tile's B
    // fragments wait behind the epilogue's 64 stores per lane, while a fresh workgroup starts beside its predecessor's
    // draining stores for free; profiles/r5_f43_persistent_ab.txt)
    // ANODDPM_DEBUG5 = 4 / 5: the one-wave-per-SIMD form of the 128-channel workgroup (winograd43w.hip; 5 = transform interleaved
    // with the MFMA stream), 6 / 7: the same wherever N % 128 == 0 (op tests on small shapes)
    // 8 / 9: as 4 / 5 with four instead of six positions of B fragments in flight
    {
        const int v5 = anoddpm::g_debug[5];
        if (dbg == 0 && a->N % 128 == 0 && ((!half && (v5 == 4 || v5 == 5 || v5 == 8 || v5 == 9)) || v5 == 6 || v5 == 7))
            return launch_winograd43w(a, s, (v5 & 1) | (v5 >= 8 ? 2 : 0));
    }
    if (dbg == 0 && ((!half && anoddpm::g_debug[5] != 1) || (a->N % 128 == 0 && anoddpm::g_debug[5] == 3))) return launch_winograd43r(a, s);
#ifdef ANODDPM_ABLATE           // timing ablations (wrong results): measurement builds only
    if (fast && dbg == 1) hipLaunchKernelGGL((wino43_kernel<true, 1>), grid, dim3(F4_NT), 0, s, *a);
    else if (fast && dbg == 2) hipLaunchKernelGGL((wino43_kernel<true, 2>), grid, dim3(F4_NT), 0, s, *a);
    else if (fast && dbg == 3) hipLaunchKernelGGL((wino43_kernel<true, 3>), grid, dim3(F4_NT), 0, s, *a);
    else if (fast && dbg == 4) hipLaunchKernelGGL((wino43_kernel<true, 4>), grid, dim3(F4_NT), 0, s, *a);
    else
#endif
    ANODDPM_REQUIRE(half || !a->fold_gamma, "winograd43: the position-sliced 128-channel kernel (ANODDPM_DEBUG5=1) does not fold the GroupNorm");
    if (fast && half) hipLaunchKernelGGL((wino43_kernel<true, 0, 4>), grid, dim3(F4_NT), 0, s, *a);
    else if (half) hipLaunchKernelGGL((wino43_kernel<false, 0, 4>), grid, dim3(F4_NT), 0, s, *a);
    else if (fast) hipLaunchKernelGGL((wino43_kernel<true>), grid, dim3(F4_NT), 0, s, *a);
    else      hipLaunchKernelGGL((wino43_kernel<false>), grid, dim3(F4_NT), 0, s, *a);
    return check_launch("winograd43");
}

}  // namespace anoddpm

extern "C" int anoddpm_f43_channel_sliced(int32_t H, int32_t W, int32_t N, int32_t B)
{
    if (H <= 0 || W <= 0 || N <= 0 || B <= 0 || H % 16 || W % 16 || N % 128) return 0;
    const int64_t wg128 = (int64_t)(H / 16) * (W / 16) * (N / 128) * B;      // the rule of launch_winograd43 (ksplit 1)
    const bool half = ((2 * wg128 + 255) / 256) * 71 < ((wg128 + 255) / 256) * 100;
    const int v5 = anoddpm::g_debug[5];                                        // ANODDPM_DEBUG5: 0 in normal operation
#ifdef ANODDPM_ABLATE
    if (anoddpm::g_debug[2] != 0) return 0;
#endif
    return ((!half && v5 == 0) || v5 == 3) ? 1 : 0;                            // 3: the op tests' selector (small shapes)
}
