// Winograd F(4x4,3x3) convolution, channel-sliced variant of winograd43.hip for 128-channel workgroups (cfg = 3 of anoddpm_igemm).
//
// Same contract and the same staging / input transform as wino43_kernel (nn.Conv2d 3x3, UNet.py:172,193, with GroupNorm-apply +
// SiLU, nearest-x2 and the two-source concat fused into the operand load; bias / time-embedding / residual / GroupNorm statistics
// in the epilogue).  What differs is who owns which accumulator:
//
//   wino43_kernel   wave w = transform positions 3w..3w+2 of ALL 128 channels.  The output transform Y = A^T M A needs the 36
//                   positions of a (tile, channel) together, so the 288 KB of accumulators go through LDS in three rounds
//                   (108 ds_write + 108 ds_read per thread, six barriers): ~13 % of a workgroup's life at K = 128.
//   this kernel     wave w = ALL 36 positions of channels 16w..16w+15 (8 waves, 144 accumulator registers each, two waves per
//                   SIMD).  A lane then holds the complete 6x6 transform-domain block of four tiles of one channel: the output
//                   transform, bias / embedding / residual add, the 16 stores per tile and the GroupNorm sums all happen in
//                   registers -- no exchange buffer, no barrier after the K loop, and the statistics row needs no cross-wave fold.
//                   Price: every wave reads all of V (36 ds_read_b128 per 16-channel chunk instead of 3; LDS has the bandwidth:
//                   ~2.3 k of the chunk's 9.2 k MFMA cycles), and the input transform's 768 items run on 512 threads.
//
// K advances 16 channels per iteration; per iteration a wave issues 144 v_mfma_f32_16x16x4_f32 (16 tiles x 16 channels x 4 k),
// 36 A-fragment reads (LDS) and 36 B-fragment loads (U[pos][k/4][n][4] from L2, six-deep register ring).
#include <type_traits>

#include "common.h"
#include "gn_fold.h"

using anoddpm::silu_f;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int R4_NT = 512;                 // threads: 8 waves
constexpr int R4_KC = 16;                  // channels per K iteration
constexpr int R4_PW = 18;                  // patch width / height (16 + 2)
constexpr int R4_PPIX = R4_PW * R4_PW;     // 324 patch pixels
constexpr int R4_PITCH = 5;                // float4 per patch pixel (4 quads + 1 pad)
constexpr int R4_PJ = 3;                   // staging slots per thread (3 * 512 = 1536 >= 324 * 4)
constexpr int R4_SLOTPX = R4_PJ * R4_NT / 4;          // 384 pixel slots per buffer
constexpr int R4_DT = R4_SLOTPX * R4_PITCH;           // float4 per patch buffer
constexpr int R4_V = 36 * 16 * 4;                     // float4 per V buffer: [pos][tile][quad]
constexpr int R4_KMAX = 1024;                         // input channels whose GroupNorm affine fits the LDS table (launcher: larger K is refused)
constexpr int R4_AFF = 2 * R4_KMAX / 4;               // float4: [K/4] scales, then [K/4] shifts of this workgroup's image (K / 4 <= 256 <= threads)
constexpr int R4_LDS_FLOATS = (2 * R4_DT + 2 * R4_V + R4_AFF) * 4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7ffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 bld4(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned wave_bytes)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)wave_bytes, 0));
}
template <int AUX>
__device__ __forceinline__ f32x4 bld4x(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned wave_bytes)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)wave_bytes, AUX));
}

// A^T of F(4x4,3x3) applied to six values: rows (1 1 1 1 1 0), (0 1 -1 2 -2 0), (0 1 1 4 4 0), (0 1 -1 8 -8 1)
__device__ __forceinline__ void at6(const float (&m)[6], float (&o)[4])
{
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    o[0] = m[0] + s12 + s34;
    o[1] = d12 + 2.f * d34;
    o[2] = s12 + 4.f * s34;
    o[3] = d12 + 8.f * d34 + m[5];
}

// DBG (timing ablations only, wrong results; ANODDPM_DEBUG6): 1 no epilogue, 2 no input transform, 3 no patch staging, 4 no B requests
// DBG 7: every patch request reads the tile's first pixel (same instruction stream, no HBM latency in the in-order vmcnt queue)
// DBG 11: patches requested but not activated / staged (the VALU + LDS half of DBG 3)
// DBG 24 / 25: epilogue without its residual requests / with one store per lane instead of 64
// DBG 23: A fragments of positions >= 2 not read from LDS (34 of 36 ds_read_b128 per chunk and wave gone)
// DBG 22: real patch requests, but every staged pixel gets the same values (separates the memory effect of DBG 7 from its data effect)
// DBG 5 / 6 (tools/f43_phases.py; results stay correct): wave 0 records s_memtime at the phase boundaries + its CU into
// a.ws[block][8] (int64): entry, prologue done, K loop done, epilogue issued, stores acknowledged (5: waited for; 6: not waited for)
// R4_RING = B-fragment requests in flight per wave
// NPOS < 36 (timing only, wrong results): only the first NPOS transform positions are multiplied -- frees accumulator registers for
// ring-depth experiments (the epilogue is skipped)
// VAR (correct results; measurement switches of round 6): bit 0 = the two waves of a SIMD transform half a chunk apart (waves
// 4..7 run T at position 18 instead of before position 0, so that a SIMD's MFMA-free transform block of one wave sits beside the
// other wave's MFMAs); bit 1 = input transform in scalar f32 (v_fma_f32 with SGPR coefficients) instead of packed f32
// F43_PAIR_TRANSFORM=1 (measurement builds, ANODDPM_EXTRA_FLAGS): the product launches use the row-pair input transform (VAR bit 5)
#ifndef F43_PAIR_TRANSFORM
#define F43_PAIR_TRANSFORM 0
#endif
constexpr int R4_VAR = F43_PAIR_TRANSFORM ? 32 : 0;
// GNB (round 6, training): the launch is a data gradient and its epilogue also performs the reduction pass of the GroupNorm + SiLU
// backward (anoddpm_igemm_args.gnb_*): the GroupNorm's input x rides in the residual slot of the epilogue (same requests, same
// registers: a data gradient has no residual), and per stored value the lane forms dy = da * silu'(y), xhat and their two sums.
template <bool FAST, int DBG = 0, int R4_RING = 6, int NPOS = 36, int VAR = 0, bool GNB = false>
__global__ __launch_bounds__(R4_NT, 1) void wino43r_kernel(const anoddpm_igemm_args a)
{
    constexpr bool V_DEPHASE = (VAR & 1) != 0, V_SCALAR_T = (VAR & 2) != 0;
    // bit 5 = row-pair items on single channels for the input transform (round 6, transform_p below): half the transform's VALU
    // work, 241 instead of 253 VGPRs, all F43 op tests green -- and the class, the step and the per-layer times do not move
    // (profiles/r6_f43_pair_transform_ab.txt): the kernel does not wait for the transform's issue slots
    constexpr bool V_PAIRS = (VAR & 32) != 0;
    // bit 2 = the MFMAs of two positions interleaved (consecutive MFMAs on different accumulators: no dependent back-to-back chain
    // when the SIMD's other wave is not issuing MFMAs); bit 3 = static s_setprio 1 for the younger half (waves 4..7)
    constexpr bool V_PAIR = (VAR & 4) != 0, V_PRIO = (VAR & 8) != 0;
    // cache policy experiments (results unchanged): DBG 18 patch requests nt; 19 patch + residual requests and stores nt;
    // 20 stores sc1 (written through, not kept in L2); 21 patch + residual nt, stores sc1
    constexpr int PATCH_AUX = (DBG == 18 || DBG == 19 || DBG == 21) ? 2 : 0;
    constexpr int RES_AUX = (DBG == 19 || DBG == 21) ? 2 : 0;
    constexpr int STORE_AUX = DBG == 19 ? 2 : ((DBG == 20 || DBG == 21) ? 16 : 0);
    unsigned long long tstamp[5];
    if (DBG == 5 || DBG == 6) tstamp[0] = __builtin_amdgcn_s_memtime();
    __shared__ __attribute__((aligned(16))) float lds[R4_LDS_FLOATS];
    f32x4 *ldsD = reinterpret_cast<f32x4 *>(lds);
    f32x4 *ldsV = ldsD + 2 * R4_DT;
    f32x4 *ldsAff = ldsV + 2 * R4_V;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    const int K = a.c0 + a.c1, N = a.N, K4 = K >> 2;
    const int tiles_x = W >> 4;
    // VAR bit 4: XCD-aware tile walk.  Workgroup b runs on XCD b % 8 (observed dispatch rule); handing XCD k the k-th contiguous
    // eighth of the tile list keeps neighbouring tiles -- which share two halo rows / columns -- behind ONE L2 instead of eight
    int tile = blockIdx.x;
    if ((VAR & 16) && (gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
    const int n0 = blockIdx.y * 128;
    // split-K (grids that would leave CUs idle, e.g. the 64x64 maps of a batch of four): blockIdx.z = image * ksplit + slice; a
    // slice accumulates a contiguous range of 16-channel chunks and leaves its output-transformed partial tile in the workspace
    // (the output transform is linear), anoddpm_igemm's split-K tail adds bias / embedding / residual and the statistics
    const int ksplit = a.ksplit;
    const int ksi = blockIdx.z % ksplit;
    const int b = blockIdx.z / ksplit;
    const int a_mode = a.a_mode;

    const float *A0 = a.a0 + (int64_t)b * a.a0_bs;
    const float *A1 = a.a1 ? a.a1 + (int64_t)b * a.a1_bs : nullptr;
    const float *gsc = a.gn_scale ? a.gn_scale + (int64_t)b * a.gn_ld : nullptr;
    const float *gsh = a.gn_shift ? a.gn_shift + (int64_t)b * a.gn_ld : nullptr;
    const bool fold = a.fold_gamma != nullptr;                      // GroupNorm finished here from fp64 sums (gn_fold.h)
    const bool affine = gsc != nullptr || fold, act = a.act != 0;
    const int cps = (K / R4_KC + ksplit - 1) / ksplit;
    const int cb = ksi * cps;                                       // first chunk of this slice
    const int nchunks = (cb + cps <= K / R4_KC ? cps : K / R4_KC - cb);

    // ---- patch staging (pixel = idx >> 2, quad = idx & 3): geometry fixed for the workgroup
    int spix[R4_PJ];
    const int pq = tid & 3;
#pragma unroll
    for (int j = 0; j < R4_PJ; ++j) {
        const int p = (tid + j * R4_NT) >> 2;
        const int py = p / R4_PW, px = p - py * R4_PW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        int sp = -1;
        if (p < R4_PPIX && gy >= 0 && gy < H && gx >= 0 && gx < W)
            sp = (a_mode == 0) ? gy * W + gx : (gy >> 1) * (W >> 1) + (gx >> 1);
        spix[j] = sp;
    }
    f32x4 praw[R4_PJ];
    f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rA0 = rsrc(A0), rA1 = rsrc(A1 ? A1 : A0);
    const __amdgpu_buffer_rsrc_t rSc = rsrc(gsc ? gsc : A0), rSh = rsrc(gsh ? gsh : A0);
    auto load_patch = [&](int chunk) {                              // unconditional loads, clamped addresses
        const int kbase = (cb + chunk) * R4_KC;
        const bool first = kbase < a.c0;
        const __amdgpu_buffer_rsrc_t r = first ? rA0 : rA1;
        const unsigned ld = (unsigned)(first ? a.a0_ld : a.a1_ld);
        const unsigned koff = (unsigned)(first ? kbase : kbase - a.c0) * 4u;
#pragma unroll
        for (int j = 0; j < R4_PJ; ++j) {
            const unsigned sp = (DBG == 7) ? (unsigned)(y0 * W + x0) : (spix[j] >= 0 ? (unsigned)spix[j] : 0u);
            praw[j] = bld4x<PATCH_AUX>(r, (sp * ld + (unsigned)(pq * 4)) * 4u, koff);
        }
    };
    auto store_patch = [&](int buf, int chunk) {                    // GroupNorm-apply + SiLU, zero padding AFTER it
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (FAST || affine) {
            // from the LDS table filled in the prologue: a global load here was followed by `s_waitcnt vmcnt(0)` -- vmcnt retires
            // in order, so every wave drained its B-fragment ring and sat out an L2 round trip in the middle of each chunk's MFMAs
            asc = ldsAff[(cb + chunk) * 4 + pq];
            ash = ldsAff[K4 + (cb + chunk) * 4 + pq];
        }
#pragma unroll
        for (int j = 0; j < R4_PJ; ++j) {
            const int idx = tid + j * R4_NT;
            f32x4 v = praw[j];
            if (DBG == 22) {                                        // real requests, regular DATA: every pixel of the patch gets the same values
                const f32x4 same = {0.25f, -0.5f, 0.75f, 1.0f};
                v = (praw[j][0] == 12345.678f) ? praw[j] : same;
            }
            if (FAST) {
                v = v * asc + ash;
                v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]);
            } else {
                if (affine) v = v * asc + ash;
                if (act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
            }
            ldsD[buf * R4_DT + (idx >> 2) * R4_PITCH + (idx & 3)] = spix[j] >= 0 ? v : zero;
        }
    };

    // ---- input transform: 768 items = 16 tiles x 8 channel pairs x 6 transform rows, as twelve "virtual waves" of 64 items
    // (virtual wave: row u, tile-row pair; lane: channel pair = lane & 7, tile slot = lane >> 3 -- the LDS service groups of
    // wino43_kernel).  Physical wave w runs (u = w % 6, pair = w / 6); waves 2..5 also run (u = w, pair 1) -- the SAME row
    // coefficients at a constant address offset -- so every SIMD hosts three passes per chunk (balanced issue time) and the
    // coefficients stay loop-invariant scalars.
    const int tu = wave % 6;
    const int tpair = lane & 7;
    const int ttile = ((wave / 6) * 2 + (lane >> 5)) * 4 + ((lane >> 3) & 3);
    const int tbase2 = (((4 * (ttile >> 2)) * R4_PW + 4 * (ttile & 3)) * R4_PITCH) * 2 + tpair;      // float2 index of the tile's patch corner
    const bool two_pass = wave >= 2 && wave <= 5;                   // wave-uniform
    constexpr int PASS_D = 8 * R4_PW * R4_PITCH * 2;                // tile row + 2 = patch row + 8 (float2 units)
    constexpr int PASS_V = 8 * 8;                                   // tile + 8 in V[pos][tile][pair]
    // B^T row u as (patch row, coefficient) pairs -- every row of B^T touches at most four patch rows:
    //   u0: 4 d0 - 5 d2 + d4        u1: -4 d1 - 4 d2 + d3 + d4     u2: 4 d1 - 4 d2 - d3 + d4
    //   u3: -2 d1 - d2 + 2 d3 + d4  u4: 2 d1 - d2 - 2 d3 + d4      u5: 4 d1 - 5 d3 + d5
    const int tr0 = (tu == 0) ? 0 : 1, tr1 = (tu == 5) ? 3 : 2, tr2 = (tu == 0) ? 4 : ((tu == 5) ? 5 : 3), tr3 = 4;
    const float tc0 = (tu == 0) ? 4.f : (tu == 1 ? -4.f : (tu == 2 ? 4.f : (tu == 3 ? -2.f : (tu == 4 ? 2.f : 4.f))));
    const float tc1 = (tu == 0 || tu == 5) ? -5.f : ((tu == 1 || tu == 2) ? -4.f : -1.f);
    const float tc2 = (tu == 0 || tu == 5) ? 1.f : (tu == 1 ? 1.f : (tu == 2 ? -1.f : (tu == 3 ? 2.f : -2.f)));
    const float tc3 = (tu == 0 || tu == 5) ? 0.f : 1.f;
    const int to0 = tr0 * R4_PW * R4_PITCH * 2, to1 = tr1 * R4_PW * R4_PITCH * 2, to2 = tr2 * R4_PW * R4_PITCH * 2, to3 = tr3 * R4_PW * R4_PITCH * 2;
    auto transform = [&](int pbuf, int vbuf, int dofs, int vofs) {
        const f32x2 *D = reinterpret_cast<const f32x2 *>(ldsD + pbuf * R4_DT) + tbase2 + dofs;
        f32x2 t[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            t[j] = tc0 * D[to0 + j * R4_PITCH * 2] + tc1 * D[to1 + j * R4_PITCH * 2] + tc2 * D[to2 + j * R4_PITCH * 2] + tc3 * D[to3 + j * R4_PITCH * 2];
        const f32x2 p = t[4] - 4.f * t[2], q = t[3] - 4.f * t[1], r = t[4] - t[2], s = t[3] - t[1];
        f32x2 *V = reinterpret_cast<f32x2 *>(ldsV + vbuf * R4_V) + ((tu * 6) * 16 + ttile) * 8 + tpair + vofs;
        V[0 * 128] = 4.f * t[0] - 5.f * t[2] + t[4];
        V[1 * 128] = p + q;
        V[2 * 128] = p - q;
        V[3 * 128] = r + 2.f * s;
        V[4 * 128] = r - 2.f * s;
        V[5 * 128] = 4.f * t[1] - 5.f * t[3] + t[5];
    };
    // the same transform on separate floats: v_fma_f32 takes the wave-uniform row coefficients from SGPRs (the packed form keeps
    // them in eight VGPRs) and dependent packed operations need an s_nop between them
    auto transform_s = [&](int pbuf, int vbuf, int dofs, int vofs) {
        const f32x2 *D = reinterpret_cast<const f32x2 *>(ldsD + pbuf * R4_DT) + tbase2 + dofs;
        float tx[6], ty[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const f32x2 d0 = D[to0 + j * R4_PITCH * 2], d1 = D[to1 + j * R4_PITCH * 2], d2 = D[to2 + j * R4_PITCH * 2], d3 = D[to3 + j * R4_PITCH * 2];
            tx[j] = tc0 * d0[0] + tc1 * d1[0] + tc2 * d2[0] + tc3 * d3[0];
            ty[j] = tc0 * d0[1] + tc1 * d1[1] + tc2 * d2[1] + tc3 * d3[1];
        }
        f32x2 *V = reinterpret_cast<f32x2 *>(ldsV + vbuf * R4_V) + ((tu * 6) * 16 + ttile) * 8 + tpair + vofs;
        const float px = tx[4] - 4.f * tx[2], qx = tx[3] - 4.f * tx[1], rx = tx[4] - tx[2], sx = tx[3] - tx[1];
        const float py = ty[4] - 4.f * ty[2], qy = ty[3] - 4.f * ty[1], ry = ty[4] - ty[2], sy = ty[3] - ty[1];
        V[0 * 128] = f32x2{4.f * tx[0] - 5.f * tx[2] + tx[4], 4.f * ty[0] - 5.f * ty[2] + ty[4]};
        V[1 * 128] = f32x2{px + qx, py + qy};
        V[2 * 128] = f32x2{px - qx, py - qy};
        V[3 * 128] = f32x2{rx + 2.f * sx, ry + 2.f * sy};
        V[4 * 128] = f32x2{rx - 2.f * sx, ry - 2.f * sy};
        V[5 * 128] = f32x2{4.f * tx[1] - 5.f * tx[3] + tx[5], 4.f * ty[1] - 5.f * ty[3] + ty[5]};
    };
    // Round 6 (VAR bit 5, measured equal: not the default): items that form TWO rows of B^T d from one set of reads, on SINGLE channels (the form wgrad43.hip runs:
    // profiles/r6_wgrad43_ab.txt).  Rows (1,2) and (3,4) share their partial sums (u1, u2 = p +- q with p = d4 - 4 d2, q = d3 - 4 d1;
    // u3, u4 = r +- 2 s with r = d4 - d2, s = d3 - d1) and (0,5) read disjoint patch rows: 4 operations per column and row PAIR where
    // the single-row items above spend 4 per row, literal coefficients instead of eight coefficient registers, 24-36 4-byte LDS
    // reads per pair instead of 2 x 24 8-byte ones.  768 items = 16 tiles x 16 channels x 3 row pairs = twelve virtual waves
    // (row pair v % 3, tile row v / 3); lane = (tile column lane >> 4, channel lane & 15).  Physical wave w runs virtual wave w;
    // waves 2..5 also run virtual wave w + 6 -- the same row pair two tile rows further down -- so every SIMD hosts three passes.
    const int p_up = wave % 3;                                      // 0: rows (0,5), 1: (1,2), 2: (3,4)
    const int p_tx = lane >> 4, p_ch = lane & 15;
    constexpr int PP = R4_PITCH * 4, PROW = R4_PW * PP;             // floats per staged pixel / patch row
    const int p_in = ((4 * (wave / 3)) * R4_PW + 4 * p_tx) * PP + p_ch;
    const int p_out = ((wave / 3) * 4 + p_tx) * 16 + p_ch;          // V[pos][tile][channel]: + pos * 256
    const int p_ua = p_up == 0 ? 0 : (p_up == 1 ? 1 : 3), p_ub = p_up == 0 ? 5 : (p_up == 1 ? 2 : 4);
    constexpr int PASS_DP = 8 * PROW, PASS_VP = 2 * 4 * 16;         // second pass: tile row + 2
    auto col_pass = [&](const float (&t)[6], float *V) {
        const float p = t[4] - 4.f * t[2], q = t[3] - 4.f * t[1], r = t[4] - t[2], w = t[3] - t[1];
        V[0 * 256] = 4.f * t[0] - 5.f * t[2] + t[4];
        V[1 * 256] = p + q;
        V[2 * 256] = p - q;
        V[3 * 256] = r + 2.f * w;
        V[4 * 256] = r - 2.f * w;
        V[5 * 256] = 4.f * t[1] - 5.f * t[3] + t[5];
    };
    auto transform_p = [&](int pbuf, int vbuf, int dofs, int vofs) {
        const float *D = reinterpret_cast<const float *>(ldsD + pbuf * R4_DT) + p_in + dofs;
        float ta[6], tb[6];
        if (p_up == 0) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d0 = D[j * PP], d1 = D[PROW + j * PP], d2 = D[2 * PROW + j * PP], d3 = D[3 * PROW + j * PP], d4 = D[4 * PROW + j * PP], d5 = D[5 * PROW + j * PP];
                ta[j] = 4.f * d0 - 5.f * d2 + d4;
                tb[j] = 4.f * d1 - 5.f * d3 + d5;
            }
        } else if (p_up == 1) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d1 = D[PROW + j * PP], d2 = D[2 * PROW + j * PP], d3 = D[3 * PROW + j * PP], d4 = D[4 * PROW + j * PP];
                const float p = d4 - 4.f * d2, q = d3 - 4.f * d1;
                ta[j] = p + q;
                tb[j] = p - q;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d1 = D[PROW + j * PP], d2 = D[2 * PROW + j * PP], d3 = D[3 * PROW + j * PP], d4 = D[4 * PROW + j * PP];
                const float r = d4 - d2, w = d3 - d1;
                ta[j] = r + 2.f * w;
                tb[j] = r - 2.f * w;
            }
        }
        float *V = reinterpret_cast<float *>(ldsV + vbuf * R4_V) + p_out + vofs;
        col_pass(ta, V + p_ua * 6 * 256);
        col_pass(tb, V + p_ub * 6 * 256);
    };
    auto transform_all = [&](int pbuf, int vbuf) {
        if (V_PAIRS) {
            transform_p(pbuf, vbuf, 0, 0);
            if (two_pass) transform_p(pbuf, vbuf, PASS_DP, PASS_VP);
        } else if (V_SCALAR_T) {
            transform_s(pbuf, vbuf, 0, 0);
            if (two_pass) transform_s(pbuf, vbuf, PASS_D, PASS_V);
        } else {
            transform(pbuf, vbuf, 0, 0);
            if (two_pass) transform(pbuf, vbuf, PASS_D, PASS_V);    // wave-uniform branch
        }
    };

    // ---- accumulators: all 36 positions x this wave's 16 channels x 16 tiles
    f32x4 acc[NPOS];
#pragma unroll
    for (int p = 0; p < NPOS; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15, kq = lane >> 4;
    const int nw = n0 + wave * 16 + l15;                            // this lane's output channel
    const __amdgpu_buffer_rsrc_t rU = rsrc(a.bmat);
    const unsigned xi_bytes = (unsigned)K4 * (unsigned)N * 16u;      // bytes per position of U
    const unsigned ulane = ((unsigned)kq * (unsigned)N + (unsigned)nw) * 16u;   // + chunk*4*N*16 + pos*xi_bytes
    const int vread = l15 * 4 + kq;                                  // + pos*64 float4

    f32x4 ring[R4_RING];
    auto load_b = [&](int chunk, int pos, int slot) {
        ring[slot] = bld4(rU, ulane, (unsigned)pos * xi_bytes + (unsigned)((cb + chunk) * 4) * (unsigned)N * 16u);
    };

    // prologue: patch(0) -> LDS -> V(0); patch(1) -> LDS; patch(2) requested
    const int last = nchunks - 1;
    const int c1 = last >= 1 ? 1 : 0, c2 = last >= 2 ? 2 : last;
    // GroupNorm affine of the image -> LDS, requested first (one float4 per thread: K / 4 scales, K / 4 shifts, K <= 1024)
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
    f32x4 praw0[R4_PJ];
#pragma unroll
    for (int j = 0; j < R4_PJ; ++j) praw0[j] = praw[j];
    load_patch(c1);
#pragma unroll
    for (int g = 0; g < R4_RING; ++g) load_b(0, g, g);
    if (FAST || affine) {
        if (fold) {
            // scratch = the V buffers (first written by transform_all below, behind two more barriers)
            anoddpm::fold_affine_finish(a, fl, tid, a_mode == 1 ? (H >> 1) * (W >> 1) : H * W, reinterpret_cast<double *>(ldsV), ldsAff);
        } else if (aff_slot) {                                         // oldest requests: no wait for the patches behind them
            ldsAff[tid] = aff_sc;
            ldsAff[K4 + tid] = aff_sh;
        }
        __syncthreads();
    }
    {
        f32x4 keep[R4_PJ];
#pragma unroll
        for (int j = 0; j < R4_PJ; ++j) { keep[j] = praw[j]; praw[j] = praw0[j]; }
        store_patch(0, 0);
#pragma unroll
        for (int j = 0; j < R4_PJ; ++j) praw[j] = keep[j];
    }
    __syncthreads();
    transform_all(0, 0);
    store_patch(1, c1);
    load_patch(c2);
    __syncthreads();
    if (DBG == 5 || DBG == 6) tstamp[1] = __builtin_amdgcn_s_memtime();

    // One step per 16-channel chunk c; what a step does besides its 144 MFMAs is fixed at compile time, so that the last three
    // chunks of the tile run without the work nobody would consume (round 5: the clamped-index form re-staged and re-transformed
    // already consumed patches there -- two SiLU passes and one transform per tile, ~2 us of issue time -- to keep ONE loop body):
    //   T  V(c+1) <- patch(c+1)    positions 0..8    S  patch(c+2) -> LDS    positions 9..26    L  request patch(c+3)
    //   positions 27..35           barrier (not after the last chunk)
    //   R  (last chunk only) from position 30 on no B fragment is requested any more: the first tile's 16 residual pixels are,
    //      so that the epilogue starts with them in flight instead of waiting out an HBM round trip first
    auto step = [&](const int chunk, auto doT, auto doS, auto doL, auto doR, auto &&res_prefetch) {
        if (decltype(doT)::value && DBG != 2 && !(V_DEPHASE && wave >= 4)) transform_all((chunk + 1) & 1, (chunk + 1) & 1);
        const f32x4 *V = ldsV + (chunk & 1) * R4_V + vread;
        constexpr int PS = NPOS / 4, PL = 3 * NPOS / 4, PR = NPOS - 6;
        if (V_PAIR) {
            f32x4 av4[4];                                           // A fragments of two positions, the next two in flight
            av4[0] = V[0];
            av4[1] = V[64];
#pragma unroll
            for (int p = 0; p < NPOS; p += 2) {
                if (V_DEPHASE && p == NPOS / 2 && decltype(doT)::value && wave >= 4) transform_all((chunk + 1) & 1, (chunk + 1) & 1);
                if ((p == PS || p == PS + 1) && decltype(doS)::value) store_patch(chunk & 1, chunk + 2);
                if ((p == PL || p == PL + 1) && decltype(doL)::value) load_patch(chunk + 3);
                if ((p == PR || p == PR + 1) && decltype(doR)::value) res_prefetch();
                const f32x4 a0 = av4[p % 4], a1 = av4[(p + 1) % 4];
                if (p + 2 < NPOS) { av4[(p + 2) % 4] = V[(p + 2) * 64]; av4[(p + 3) % 4] = V[(p + 3) * 64]; }
                const f32x4 b0 = ring[p % R4_RING], b1 = ring[(p + 1) % R4_RING];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kk], b0[kk], acc[p], 0, 0, 0);
                    acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kk], b1[kk], acc[p + 1], 0, 0, 0);
                }
#pragma unroll
                for (int q = p; q < p + 2; ++q) {
                    if (q + R4_RING < NPOS)           load_b(chunk, q + R4_RING, q % R4_RING);
                    else if (!decltype(doR)::value)   load_b(chunk + 1, q + R4_RING - NPOS, q % R4_RING);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!decltype(doR)::value) __syncthreads();
            return;
        }
        f32x4 av[3];                                                // A fragments: two positions ahead of the MFMAs
        av[0] = V[0];
        av[1] = V[64];
#pragma unroll
        for (int p = 0; p < NPOS; ++p) {
            // (round 5, measured and dropped: the two waves of a SIMD staging half a chunk apart -- S / L at positions 0 / 9 for waves
            // 4..7 -- so that they would not wait for their patch requests together: 8.99 vs 9.01 ms per step, no difference;
            // nor does a deeper B ring or a non-temporal policy on the streamed tensors help: profiles/r5_f43_phases_ablations.txt)
            if (V_DEPHASE && p == NPOS / 2 && decltype(doT)::value && wave >= 4) transform_all((chunk + 1) & 1, (chunk + 1) & 1);
            if (p == PS && decltype(doS)::value && DBG != 3 && DBG != 11) store_patch(chunk & 1, chunk + 2);   // patch(c+2) replaces patch(c): its readers passed the last barrier
            if (p == PL && decltype(doL)::value && DBG != 3) load_patch(chunk + 3);
            if (p == PR && decltype(doR)::value) res_prefetch();
            const f32x4 a_cur = av[p % 3];
            if (p + 2 < NPOS && DBG != 23) av[(p + 2) % 3] = V[(p + 2) * 64];   // DBG 23: only the first two A fragments of a chunk are read
            const f32x4 bv = ring[p % R4_RING];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[kk], bv[kk], acc[p], 0, 0, 0);
            if (DBG != 4) {
                if (p + R4_RING < NPOS)           load_b(chunk, p + R4_RING, p % R4_RING);
                else if (!decltype(doR)::value)   load_b(chunk + 1, p + R4_RING - NPOS, p % R4_RING);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!decltype(doR)::value) __syncthreads();                 // publishes V(c+1) and patch(c+2); retires V(c)
    };
    if (V_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};
    auto nothing = []() {};
    int chunk = 0;
    for (; chunk + 3 <= last; ++chunk) step(chunk, YES, YES, YES, NO, nothing);

    // ---- epilogue, in registers: lane = (channel nw, tiles kq*4 .. kq*4+3); tile r of the lane sits in component r of every acc.
    // (The transposed form -- D = channels x tiles, one 16-byte store per pixel and lane -- measured slower: 292 vs 282 us on the
    // 256x256 128->128 layer; the stores of a round are HBM-burst-bound, not issue-bound.)
    const bool part = ksplit > 1;
    const float *TE = (a.temb && !part) ? a.temb + (int64_t)b * a.temb_ld : nullptr;
    const __amdgpu_buffer_rsrc_t rO = part ? rsrc(a.ws + ((int64_t)ksi * a.B + b) * ((int64_t)H * W * N))
                                           : rsrc(a.out + (int64_t)b * a.o_bs);
    // GNB: this wave's 16 channels of x come from the first or the second concatenated source (gnb_c0 % 16 == 0: wave-uniform)
    const bool gx_first = !GNB || n0 + wave * 16 < a.gnb_c0;
    const float *gx = !GNB ? nullptr : (gx_first ? a.gnb_x0 + (int64_t)b * a.gnb_x0_bs : a.gnb_x1 + (int64_t)b * a.gnb_x1_bs);
    const __amdgpu_buffer_rsrc_t rR = GNB ? rsrc(gx) : rsrc(a.res ? a.res + (int64_t)b * a.r_bs : a.out);
    const bool has_res = GNB || (a.res != nullptr && !part);
    const unsigned uW = (unsigned)W, o_ld = part ? (unsigned)N : (unsigned)a.out_ld;
    const unsigned r_ld = GNB ? (unsigned)(gx_first ? a.gnb_x0_ld : a.gnb_x1_ld) : (unsigned)a.res_ld;
    const float alpha = part ? 1.0f : a.alpha;
    float add = 0.f;
    if (a.bias && !part) add += a.bias[nw];
    if (TE) add += TE[nw];
    // per-lane byte offset of tile (kq, 0)'s first pixel; tile r and pixel (i, j) add the wave-uniform (r*4 + i*W + j) pixels
    const unsigned pix0 = (unsigned)(y0 + kq * 4) * uW + (unsigned)x0;
    const unsigned nwr = GNB ? (unsigned)(gx_first ? nw : nw - a.gnb_c0) : (unsigned)nw;      // channel within the residual / x source
    const unsigned vo = (pix0 * o_ld + (unsigned)nw) * 4u, vr = (pix0 * r_ld + nwr) * 4u;
    // res_mode 1: the residual lives at half resolution (nearest x2 on the read): a 4 x 4 output tile reads its 2 x 2 source pixels
    const bool res_up = !GNB && a.res_mode == 1;
    const unsigned hW = uW >> 1;
    const unsigned vrh = ((((unsigned)(y0 + kq * 4) >> 1) * hW + ((unsigned)x0 >> 1)) * r_ld + nwr) * 4u;
    auto load_res = [&](int r, float (&rv)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) rv[i] = 0.f;
        if (DBG == 24) return;                                      // epilogue ablation: no residual requests
        if (has_res && res_up) {
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rR, (int)vrh, (int)((((unsigned)(r * 2) + (unsigned)i2 * hW + (unsigned)j2) * 4u) * r_ld), RES_AUX));
                    rv[(2 * i2) * 4 + 2 * j2] = v; rv[(2 * i2) * 4 + 2 * j2 + 1] = v;
                    rv[(2 * i2 + 1) * 4 + 2 * j2] = v; rv[(2 * i2 + 1) * 4 + 2 * j2 + 1] = v;
                }
        } else if (has_res) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rv[i * 4 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rR, (int)vr, (int)((((unsigned)(r * 4) + (unsigned)i * uW + (unsigned)j) * 4u) * r_ld), RES_AUX));
        }
    };
    float rv[2][16];
    // the tile's last three chunks (fewer when the slice has fewer)
    if (last >= 2) step(last - 2, YES, YES, NO, NO, nothing);
    if (last >= 1) step(last - 1, YES, NO, NO, NO, nothing);
    step(last, NO, NO, NO, YES, [&]() { load_res(0, rv[0]); });

    if (DBG == 5 || DBG == 6) tstamp[2] = __builtin_amdgcn_s_memtime();
    if (DBG == 1 || NPOS < 36) {
        float sum = 0.f;
#pragma unroll
        for (int p = 0; p < NPOS; ++p) sum += (acc[p][0] + acc[p][1]) + (acc[p][2] + acc[p][3]);
        if (sum == 12345.678f) a.out[0] = sum;
        return;
    }
    float cs = 0.f, cq = 0.f;
    // GNB: y = g_sc * x + g_sh (g_sc = gamma * rstd), xhat = (x - g_mu) * g_rs, as anoddpm_gn_silu_backward's chan_params
    float g_sc = 0.f, g_sh = 0.f, g_mu = 0.f, g_rs = 0.f;
    if (GNB) {
        const int grp = nw / (N / a.gnb_groups);
        g_mu = a.gnb_mean[(int64_t)b * a.gnb_groups + grp];
        g_rs = a.gnb_rstd[(int64_t)b * a.gnb_groups + grp];
        g_sc = a.gnb_gamma[nw] * g_rs;
        g_sh = a.gnb_beta[nw] - g_mu * g_sc;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r + 1 < 4) load_res(r + 1, rv[(r + 1) & 1]);           // the next tile's residual pixels ride behind this tile's arithmetic
        // columns first: y[i][v] = sum_u A^T[i][u] m[u][v]
        float y[4][6];
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            float mu[6], o[4];
#pragma unroll
            for (int u = 0; u < 6; ++u) mu[u] = acc[u * 6 + v][r];
            at6(mu, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i][v] = o[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float o4[4];
            at6(y[i], o4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned so = ((unsigned)(r * 4) + (unsigned)i * uW + (unsigned)j) * 4u;      // wave-uniform pixel offset (x ld below)
                const float v = GNB ? alpha * o4[j] + add : alpha * o4[j] + add + rv[r & 1][i * 4 + j];
                if (DBG != 25 || (r == 0 && i == 0 && j == 0))      // DBG 25: one store per lane instead of 64
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rO, (int)vo, (int)(so * o_ld), STORE_AUX);
                if (GNB) {                                          // v = da at this pixel, rv = x: the two sums of the reduction pass
                    const float x = rv[r & 1][i * 4 + j];
                    const float y = x * g_sc + g_sh;
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                    const float dyv = v * (sg * (1.0f + y * (1.0f - sg)));
                    cs += dyv;
                    cq += dyv * ((x - g_mu) * g_rs);
                } else {
                    cs += v;
                    cq += v * v;
                }
            }
        }
    }
    if (GNB) {
        cs += __shfl_xor(cs, 16);
        cq += __shfl_xor(cq, 16);
        cs += __shfl_xor(cs, 32);
        cq += __shfl_xor(cq, 32);
        if (kq == 0) {                                              // partial[b][tile][channel] = {sum dy, sum dy * xhat}
            double *pp = a.gnb_partial + (((int64_t)b * gridDim.x + tile) * N + nw) * 2;
            pp[0] = (double)cs;
            pp[1] = (double)cq;
        }
    } else if ((a.stats || a.stats_csum) && !part) {
        // the lane's 64 outputs of channel nw; the four kq lane groups hold the other tiles of the same channel
        cs += __shfl_xor(cs, 16);
        cq += __shfl_xor(cq, 16);
        cs += __shfl_xor(cs, 32);
        cq += __shfl_xor(cq, 32);
        if (kq == 0) {
            if (a.stats_csum) {
                anoddpm::csum_atomic_add(a.stats_csum, b, N, nw, cs, cq);     // one fp64 pair per workgroup and channel (gn_fold.h)
            } else {
                float *st = a.stats + (((int64_t)b * gridDim.x + blockIdx.x) * N + nw) * 2;
                st[0] = cs;
                st[1] = cq;
            }
        }
    }
    if (DBG == 5 || DBG == 6) {
        tstamp[3] = __builtin_amdgcn_s_memtime();
        if (DBG == 5) __builtin_amdgcn_s_waitcnt(0);                  // vmcnt(0) (and the other counters): the stores are acknowledged
        tstamp[4] = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned long long *d = reinterpret_cast<unsigned long long *>(a.ws) +
                                    (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
#pragma unroll
            for (int i = 0; i < 5; ++i) d[i] = tstamp[i];
            d[5] = (unsigned long long)__builtin_amdgcn_s_getreg(63492);            // HW_REG_HW_ID (wave, SIMD, CU, SH, SE)
            d[6] = (unsigned long long)__builtin_amdgcn_s_getreg(63508);            // HW_REG_XCC_ID
            d[7] = 0;
        }
    }
}

}  // namespace

namespace anoddpm {

// Called by launch_winograd43 for the 128-channel grid (arguments validated there).
int launch_winograd43r(const anoddpm_igemm_args *a, hipStream_t s)
{
    dim3 grid((unsigned)((a->H / 16) * (a->W / 16)), (unsigned)(a->N / 128), (unsigned)(a->B * a->ksplit));
    const bool fast = (a->gn_scale || a->fold_gamma) && a->act;
    ANODDPM_REQUIRE(!(a->gn_scale || a->fold_gamma) || a->c0 + a->c1 <= R4_KMAX, "winograd43r: GroupNorm affine table holds %d input channels", R4_KMAX);
#ifdef ANODDPM_ABLATE           // timing ablations (wrong results) and ring-depth variants: measurement builds only
    const int dbg = g_debug[6];
    if (fast && dbg == 1) hipLaunchKernelGGL((wino43r_kernel<true, 1>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 2) hipLaunchKernelGGL((wino43r_kernel<true, 2>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 3) hipLaunchKernelGGL((wino43r_kernel<true, 3>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 4) hipLaunchKernelGGL((wino43r_kernel<true, 4>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 5) hipLaunchKernelGGL((wino43r_kernel<true, 5>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 6) hipLaunchKernelGGL((wino43r_kernel<true, 6>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 7) hipLaunchKernelGGL((wino43r_kernel<true, 7, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 11) hipLaunchKernelGGL((wino43r_kernel<true, 11, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 12) hipLaunchKernelGGL((wino43r_kernel<true, 3, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 13) hipLaunchKernelGGL((wino43r_kernel<true, 4, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 24) hipLaunchKernelGGL((wino43r_kernel<true, 24, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 25) hipLaunchKernelGGL((wino43r_kernel<true, 25, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 23) hipLaunchKernelGGL((wino43r_kernel<true, 23, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 22) hipLaunchKernelGGL((wino43r_kernel<true, 22, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 18) hipLaunchKernelGGL((wino43r_kernel<true, 18, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 19) hipLaunchKernelGGL((wino43r_kernel<true, 19, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 20) hipLaunchKernelGGL((wino43r_kernel<true, 20, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 21) hipLaunchKernelGGL((wino43r_kernel<true, 21, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 14) hipLaunchKernelGGL((wino43r_kernel<true, 0, 9, 18>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 15) hipLaunchKernelGGL((wino43r_kernel<true, 0, 18, 18>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 16) hipLaunchKernelGGL((wino43r_kernel<true, 7, 9, 18>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 17) hipLaunchKernelGGL((wino43r_kernel<true, 0, 6, 18>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 8) hipLaunchKernelGGL((wino43r_kernel<true, 0, 8>), grid, dim3(R4_NT), 0, s, *a);   // (8 does not divide 36: timing only)
    else if (fast && dbg == 10) hipLaunchKernelGGL((wino43r_kernel<true, 0, 6>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 9) hipLaunchKernelGGL((wino43r_kernel<true, 0, 4>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 30) hipLaunchKernelGGL((wino43r_kernel<true, 0, 9, 36, 1>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 31) hipLaunchKernelGGL((wino43r_kernel<true, 0, 9, 36, 2>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 32) hipLaunchKernelGGL((wino43r_kernel<true, 0, 9, 36, 3>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 33) hipLaunchKernelGGL((wino43r_kernel<true, 0, 6, 36, 1>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 34) hipLaunchKernelGGL((wino43r_kernel<true, 0, 6, 36, 3>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 39) hipLaunchKernelGGL((wino43r_kernel<true, 0, 9, 36, 16>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 35) hipLaunchKernelGGL((wino43r_kernel<true, 0, 6, 36, 4>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 36) hipLaunchKernelGGL((wino43r_kernel<true, 0, 6, 36, 5>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 37) hipLaunchKernelGGL((wino43r_kernel<true, 0, 9, 36, 8>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 38) hipLaunchKernelGGL((wino43r_kernel<true, 0, 9, 36, 9>), grid, dim3(R4_NT), 0, s, *a);
    else if (fast && dbg == 40) hipLaunchKernelGGL((wino43r_kernel<true, 0, 9, 36, 32>), grid, dim3(R4_NT), 0, s, *a);   // row-pair input transform
    else if (fast && dbg == 41) hipLaunchKernelGGL((wino43r_kernel<true, 0, 12, 36, 32>), grid, dim3(R4_NT), 0, s, *a);   // ... with a twelve-deep B ring (the registers it frees)
    else
#endif
    // nine B fragments in flight (250 VGPRs) for the GroupNorm + SiLU form: 9.05 -> 9.01 ms per config-2 step over six (round 5,
    // once the per-chunk vmcnt(0) drain was gone; the depth must divide 36); the plain form sits at 247 VGPRs with six
    if (a->gnb_partial) {
        ANODDPM_REQUIRE(!fast && !a->gn_scale && !a->fold_gamma && !a->act && a->ksplit == 1 && !a->res && !a->bias && !a->temb && !a->stats && !a->stats_csum,
                        "winograd43r: gnb_partial needs a plain data-gradient launch (no gn / act / bias / temb / res / stats, ksplit 1)");
        ANODDPM_REQUIRE(a->gnb_x0 && a->gnb_gamma && a->gnb_beta && a->gnb_mean && a->gnb_rstd && a->gnb_groups > 0 && a->N % a->gnb_groups == 0 &&
                        a->gnb_c0 > 0 && a->gnb_c0 % 16 == 0 && a->gnb_c0 <= a->N && (a->gnb_c0 == a->N || a->gnb_x1),
                        "winograd43r: gnb_partial: bad GroupNorm arguments");
        ANODDPM_REQUIRE((int64_t)a->H * a->W * (a->gnb_x0_ld > a->gnb_x1_ld ? a->gnb_x0_ld : a->gnb_x1_ld) * 4 < ((int64_t)1 << 31),
                        "winograd43r: gnb_partial: source slice exceeds 32-bit buffer offsets");
        hipLaunchKernelGGL((wino43r_kernel<false, 0, 6, 36, R4_VAR, true>), grid, dim3(R4_NT), 0, s, *a);
        return check_launch("winograd43r");
    }
    if (fast) hipLaunchKernelGGL((wino43r_kernel<true, 0, 9, 36, R4_VAR>), grid, dim3(R4_NT), 0, s, *a);
    else      hipLaunchKernelGGL((wino43r_kernel<false, 0, 6, 36, R4_VAR>), grid, dim3(R4_NT), 0, s, *a);
    return check_launch("winograd43r");
}

}  // namespace anoddpm
