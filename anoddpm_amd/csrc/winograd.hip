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
// Mapping to gfx950 (measured history in DESIGN.md 5b).  The limit is accumulator capacity: 16 transform positions of a
// 32 tiles x 32 channels wave tile are 256 registers, so they are split over TWO waves (xh = 0/1: 8 positions = 128
// accumulator VGPRs each) that exchange partial output transforms through LDS in the epilogue.  A workgroup covers an
// 8x16 output patch (32 tiles) x 32*WNW channels; K advances 16 channels per iteration, each iteration ONE basic block:
//   * the activated halo patch (GN-apply + SiLU, zero padding, nearest-x2 / concat resolved on the load) is the only
//     staged operand: double-buffered in LDS, fetched two iterations ahead through buffer loads whose in-loop
//     address parts are all scalar;
//   * the B operands U[xi][k][n] stream from L2 straight into a two-deep register ring (every lane needs different
//     words, LDS would only add traffic);
//   * the A operands B^T d B are either transformed by each lane for exactly the values it multiplies (WNW = 2) or
//     computed once per workgroup one iteration ahead and shared through LDS (VSH, WNW = 4);
//   * 64 MFMAs per wave per iteration in four bursts of 16, operand reads for the next burst issued before each.
// The fp32 MFMA does not hide VALU issue on this chip (tools/mfma_ubench.hip), so the design minimises VALU
// instructions per MFMA: no 64-bit address arithmetic, no predicated loads, no repeated transforms.
#include "common.h"

using anoddpm::silu_f;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WKC = 16;            // channels per K iteration

// Buffer loads: descriptor (SGPRs) + per-lane 32-bit byte offset (VGPR) + wave-uniform 32-bit byte offset (SGPR).
// All address arithmetic that changes inside the K loop is then scalar -- VALU instructions are NOT hidden by the
// fp32 MFMA on gfx950, 64-bit VALU pointer adds would come straight out of the matrix issue time.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wrsrc(const float *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7ffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 wbld4(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned wave_bytes)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)wave_bytes, 0));
}

constexpr int DPITCH = 5;          // float4 per patch pixel in LDS: 4 quads + 1 pad (spreads the stride-2-pixel reads over banks)

// FAST = every operand element gets GroupNorm-apply + SiLU (all 3x3 layers of the UNet): the staging code is then
// branch-free, which keeps the whole K iteration one basic block for the scheduler.
// WMW = wave rows: a workgroup is 4*WMW waves on an (8*WMW) x 16 output patch.  WMW = 1 (256 threads) lets TWO independent
// workgroups share a CU, so one's prologue / epilogue (patch latency, LDS exchange, stores: ~20 % of a K = 128
// workgroup's life) overlaps the other's MFMA stream; WMW = 2 is the original 512-thread shape (smaller halo).
// WNW = wave columns: 32 * WNW output channels per workgroup.  WNW = 4 (all 128 channels of the common layers in one
// 512-thread workgroup) activates each patch element once instead of once per 64-channel workgroup.
// VSH (needs WNW = 4): the input transform is computed ONCE per workgroup (512 quarter-items = one per thread) one
// K iteration ahead and shared through a double-buffered LDS array V[pos][quad][tile]; the MFMA operands are then
// plain ds_read_b128s.  Without it each of the four wn waves repeats the transform of the tiles it multiplies
// (VALU issue is not hidden by the fp32 MFMA, so that repetition costs ~9 % of the loop).
template <bool FAST, int WMW, int WNW, bool PROBE = false, bool VSH = false>
__global__ __launch_bounds__(128 * WMW * WNW, 2) void wino_kernel(const anoddpm_igemm_args a)
{
    // PROBE (tools/wino_phases.py only): wave 0 records s_memtime at the phase boundaries into a.ws[block][8]
    unsigned long long tstamp[5];
    if (PROBE) tstamp[0] = __builtin_amdgcn_s_memtime();
    constexpr int NT = 128 * WMW * WNW;                            // threads: waves (xh 2) x (wm WMW) x (wn WNW)
    constexpr int WBN = 32 * WNW;                                  // output channels per workgroup
    constexpr int PROWS = 8 * WMW + 2;                             // patch rows (18 columns)
    constexpr int WPATCH = PROWS * 18;                             // input patch pixels
    constexpr int PJ = (WPATCH * 4 + NT - 1) / NT;                 // staging slots per thread (3, or 2 for 512 threads on 8x16)
    constexpr int SLOTPX = PJ * NT / 4;                            // pixel slots per buffer (>= WPATCH)
    // LDS: only the activated input patch, double buffered: Dt[2][SLOTPX pixel slots][5 float4]  (30 KB * WMW; WPATCH
    // pixels are real, the rest absorbs the unconditional stores of the last staging slot).
    // Neither V (transformed input) nor U (transformed weights) ever touch LDS:
    //   * a lane needs V only for ITS tile and channel quad, so it transforms patch -> A-operand registers;
    //   * the B operand U[xi][k][n] is read by each lane straight from L2 (layout [xi][K/4][N][4] makes it
    //     one coalesced 16-byte load per operand) through a two-deep register ring.
    // One barrier per 16-channel iteration; the exchange buffer of the epilogue re-uses the same LDS.
    constexpr int DT_F4 = SLOTPX * DPITCH;                         // float4 per buffer
    constexpr int EX_FLOATS = (WMW * WNW) * 2 * 8 * 4 * 64;        // exchange buffer of the epilogue: 16 KB per wave pair
    constexpr int V_F4 = 16 * 4 * 32;                              // VSH: V[16 positions][4 quads][32 tiles] float4 = 32 KB
    constexpr int LDS_FLOATS = VSH ? (2 * DT_F4 + 2 * V_F4) * 4 : EX_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    static_assert(VSH ? (EX_FLOATS <= LDS_FLOATS && WNW == 4 && WMW == 1) : (2 * DT_F4 * 4 <= EX_FLOATS), "LDS layout");
    static_assert(PJ * NT >= WPATCH * 4 && PJ * NT <= SLOTPX * 4, "staging slots");
    f32x4 *ldsD = reinterpret_cast<f32x4 *>(lds);

    // 4*WMW waves: wave = (xh, wm, wn).  (wm, wn) picks the 32 tiles x 32 channels wave tile; xh picks which half of
    // the transform rows (u = 2*xh, 2*xh+1 -> 8 of the 16 positions) the wave accumulates: 128 accumulators.
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int xh = __builtin_amdgcn_readfirstlane(wave / (WMW * WNW));
    const int wm = (wave / WNW) & (WMW - 1), wn = wave & (WNW - 1);
    const int h = lane >> 5, l31 = lane & 31;

    const int H = a.H, W = a.W;
    const int K = a.c0 + a.c1, N = a.N, K4 = K >> 2;
    const int bx = blockIdx.x % (W >> 4), by = blockIdx.x / (W >> 4);
    const int y0 = by * (8 * WMW), x0 = bx * 16;                   // output patch origin
    const int n0 = blockIdx.y * WBN;
    const int ksplit = a.ksplit;                                   // split-K over 16-channel chunks (small maps)
    const int ksi = blockIdx.z % ksplit;
    const int b = blockIdx.z / ksplit;
    const int a_mode = a.a_mode;

    const float *A0 = a.a0 + (int64_t)b * a.a0_bs;
    const float *A1 = a.a1 ? a.a1 + (int64_t)b * a.a1_bs : nullptr;
    const float *gsc = a.gn_scale ? a.gn_scale + (int64_t)b * a.gn_ld : nullptr;
    const float *gsh = a.gn_shift ? a.gn_shift + (int64_t)b * a.gn_ld : nullptr;
    const bool affine = (gsc != nullptr);
    const bool act = a.act != 0;
    const int nchunks_all = K / WKC;
    const int cps = (nchunks_all + ksplit - 1) / ksplit;
    const int cbeg = ksi * cps;                                     // launcher guarantees cbeg < nchunks_all
    const int nchunks = cbeg + cps < nchunks_all ? cbeg + cps : nchunks_all;   // end of this block's chunk range

    // ---- patch staging: slots of this thread (pixel = idx>>2, quad = idx&3), geometry fixed for the workgroup
    int spix[PJ];
    const int pq = tid & 3;
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int idx = tid + j * NT;
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
    const __amdgpu_buffer_rsrc_t rA0 = wrsrc(A0), rA1 = wrsrc(A1 ? A1 : A0);
    const __amdgpu_buffer_rsrc_t rSc = wrsrc(gsc ? gsc : A0), rSh = wrsrc(gsh ? gsh : A0);
    auto load_patch = [&](int chunk) {                              // unconditional loads, clamped addresses
        const int kbase = chunk * WKC;
        const bool first = kbase < a.c0;                            // wave-uniform source choice
        const __amdgpu_buffer_rsrc_t r = first ? rA0 : rA1;
        const unsigned ld = (unsigned)(first ? a.a0_ld : a.a1_ld);
        const unsigned koff = (unsigned)(first ? kbase : kbase - a.c0) * 4u;
        if (FAST || affine) {
            asc = wbld4(rSc, (unsigned)(pq * 16), (unsigned)kbase * 4u);
            ash = wbld4(rSh, (unsigned)(pq * 16), (unsigned)kbase * 4u);
        }
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const unsigned sp = spix[j] >= 0 ? (unsigned)spix[j] : 0u;
            praw[j] = wbld4(r, (sp * ld + (unsigned)(pq * 4)) * 4u, koff);
        }
    };
    auto store_patch = [&](int buf) {                               // transform, zero padding AFTER it
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int idx = tid + j * NT;                           // < 3 * NT <= SLOTPX * 4: always in the buffer
            f32x4 v = praw[j];
            if (FAST) {
                v = v * asc + ash;
                v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]);
            } else {
                if (affine) v = v * asc + ash;
                if (act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
            }
            ldsD[buf * DT_F4 + (idx >> 2) * DPITCH + (idx & 3)] = spix[j] >= 0 ? v : zero;
        }
    };

    // ---- accumulators: 8 transform positions (u = 2*xh + uu, v = 0..3; index uu*4 + v) of a 32 x 32 wave tile
    f32x16 acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;

    const int atile = wm * 32 + l31;                                // this lane's A row (tile)
    const int tty = atile >> 3, ttx = atile & 7;
    const int dbase = ((2 * tty) * 18 + 2 * ttx) * DPITCH + h;      // float4 index: tile's patch corner, quad h
    const int bcol = wn * 32 + l31;                                 // this lane's B column (channel)
    const int nbc = n0 + bcol < N ? n0 + bcol : N - 1;              // N tail: clamped, column discarded later
    const int64_t xi_stride = (int64_t)K4 * N * 4;

    // Row u = 2*xh + uu of t = B^T d is  X - Y  or  X + Y  of two patch rows (wave-uniform choice):
    //   xh=0: uu=0: d0 - d2   uu=1: d1 + d2        xh=1: uu=0: d2 - d1   uu=1: d1 - d3
    const int rowX0 = xh ? 2 : 0, rowY0 = xh ? 1 : 2;
    const int rowX1 = 1,          rowY1 = xh ? 3 : 2;
    const float sg1 = xh ? -1.f : 1.f;                              // sign of Y for uu = 1 (uu = 0 is always -1)
    const int offX[2] = {rowX0 * 18 * DPITCH, rowX1 * 18 * DPITCH};
    const int offY[2] = {rowY0 * 18 * DPITCH, rowY1 * 18 * DPITCH};

    // A group = 16 MFMAs: (chunk, kg, uu) -> 4 positions v x K = 4 (quad q = 2*kg + h of the 16-channel chunk).
    // B operands of a group: U[xi = 8xh + 4uu + v][k4 = 4*chunk + 2kg + h][n], v = 0..3
    const __amdgpu_buffer_rsrc_t rU = wrsrc(a.bmat);
    const unsigned xi_bytes = (unsigned)xi_stride * 4u;             // launcher guarantees 16 * xi_bytes < 2^31
    const unsigned ubase = (unsigned)(xh * 8) * xi_bytes;           // wave-uniform
    const unsigned ulane = (unsigned)(nbc * 4 + h * N * 4) * 4u;    // per-lane byte offset
    f32x4 bvr[2][4];
    auto load_b = [&](int chunk, int kg, int uu, int set) {
        const unsigned w = ubase + (unsigned)(chunk * 4 + 2 * kg) * (unsigned)N * 16u + (unsigned)(uu * 4) * xi_bytes;
#pragma unroll
        for (int v = 0; v < 4; ++v) bvr[set][v] = wbld4(rU, ulane, w + (unsigned)v * xi_bytes);
    };
    f32x4 rawX[4], rawY[4], av[4];
    auto issue_reads = [&](int buf, int kg, int uu, int which) {    // 4 ds_read_b128: one patch row of the tile
        const f32x4 *D = ldsD + buf * DT_F4 + dbase + 2 * kg;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (which == 0) rawX[j] = D[offX[uu] + j * DPITCH];
            else            rawY[j] = D[offY[uu] + j * DPITCH];
        }
    };
    auto make_av = [&](int uu) {                                    // row transform, then the column transform
        const float sg = uu ? sg1 : -1.f;
        f32x4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = rawX[j] + sg * rawY[j];
        av[0] = t[0] - t[2];
        av[1] = t[1] + t[2];
        av[2] = t[2] - t[1];
        av[3] = t[1] - t[3];
    };

    const int last = nchunks - 1;
    if constexpr (VSH) {
        f32x4 *ldsV = ldsD + 2 * DT_F4;
        // transform role of this thread: quarter-item (tile, quad, row u) of B^T d B -> 4 positions (u, v = 0..3)
        const int ttile = tid & 31, tquad = (tid >> 5) & 3;
        const int urow = __builtin_amdgcn_readfirstlane(tid >> 7);              // waves 2u, 2u+1
        const int trX = urow == 0 ? 0 : (urow == 2 ? 2 : 1);                    // u: 0: d0-d2  1: d1+d2  2: d2-d1  3: d1-d3
        const int trY = urow == 0 ? 2 : (urow == 1 ? 2 : (urow == 2 ? 1 : 3));
        const float tsg = urow == 1 ? 1.f : -1.f;
        const int tbase = ((2 * (ttile >> 3)) * 18 + 2 * (ttile & 7)) * DPITCH + tquad;
        const int tX = tbase + trX * 18 * DPITCH, tY = tbase + trY * 18 * DPITCH;
        const int vw = ((urow * 4) * 4 + tquad) * 32 + ttile;                   // + v * 128 float4 per position
        f32x4 tX4[4], tY4[4];
        auto t_reads = [&](int buf) {
            const f32x4 *D = ldsD + buf * DT_F4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { tX4[j] = D[tX + j * DPITCH]; tY4[j] = D[tY + j * DPITCH]; }
        };
        auto t_write = [&](int vbuf) {
            f32x4 t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = tX4[j] + tsg * tY4[j];
            f32x4 *V = ldsV + vbuf * V_F4 + vw;
            V[0 * 128] = t[0] - t[2];
            V[1 * 128] = t[1] + t[2];
            V[2 * 128] = t[2] - t[1];
            V[3 * 128] = t[1] - t[3];
        };
        // MFMA operands: A fragment of group (kg, uu): V[pos = (2xh+uu)*4 + v][quad = 2kg + h][tile = l31]
        const int vr = ((xh * 8) * 4 + h) * 32 + l31;
        f32x4 af[2][4];
        auto a_reads = [&](int vbuf, int kg, int uu, int set) {
            const f32x4 *V = ldsV + vbuf * V_F4 + vr + ((uu * 4) * 4 + 2 * kg) * 32;
#pragma unroll
            for (int v = 0; v < 4; ++v) af[set][v] = V[v * 128];
        };
        // prologue: patch(cbeg) -> LDS, V(cbeg), patch(cbeg+1) -> LDS, patch(cbeg+2) in flight
        const int c1 = cbeg < last ? cbeg + 1 : last, c2 = cbeg + 2 < nchunks ? cbeg + 2 : last;
        load_patch(cbeg);
        store_patch(cbeg & 1);
        load_patch(c1);
        load_b(cbeg, 0, 0, 0);
        load_b(cbeg, 0, 1, 1);
        __syncthreads();
        t_reads(cbeg & 1);
        store_patch((cbeg + 1) & 1);
        load_patch(c2);
        t_write(cbeg & 1);
        __syncthreads();
        a_reads(cbeg & 1, 0, 0, 0);
        if (PROBE) tstamp[1] = __builtin_amdgcn_s_memtime();
        for (int chunk = cbeg; chunk < nchunks; ++chunk) {
            const int nxt = chunk < last ? chunk + 1 : last;        // clamped: the tail re-loads valid memory, unused
            const int nxt3 = chunk + 3 < nchunks ? chunk + 3 : last;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int uu = g4 & 1, set = g4 & 1;
                // One barrier per iteration, ahead of the last group: it publishes V(chunk+1) and patch(chunk+2), both
                // written earlier in this iteration, and retires every read of the buffers iteration chunk+1 overwrites.
                if (g4 == 3) __syncthreads();
                if (g4 < 3) a_reads(chunk & 1, (g4 + 1) >> 1, (g4 + 1) & 1, set ^ 1);
                else        a_reads((chunk + 1) & 1, 0, 0, set ^ 1);
                if (g4 == 1) t_reads((chunk + 1) & 1);              // patch(chunk+1): stored last iteration
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        acc[uu * 4 + v] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][v][kk], bvr[uu][v][kk], acc[uu * 4 + v], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (g4 < 2) load_b(chunk, 1, uu, uu);
                else        load_b(nxt, 0, uu, uu);
                // register budget: the staged patch (praw, asc, ash) lives across the bursts of groups 3 and 0, the raw
                // transform rows across the burst of group 1 -- never both
                if (g4 == 0) store_patch(chunk & 1);                // patch(chunk+2) replaces patch(chunk), consumed an iteration ago
                if (g4 == 1) t_write((chunk + 1) & 1);              // V(chunk+1)
                if (g4 == 2) load_patch(nxt3);
            }
        }
    } else {
        load_patch(cbeg);
        store_patch(cbeg & 1);
        load_patch(cbeg < last ? cbeg + 1 : last);
        load_b(cbeg, 0, 0, 0);
        load_b(cbeg, 0, 1, 1);
        __syncthreads();
        issue_reads(cbeg & 1, 0, 0, 0);
        issue_reads(cbeg & 1, 0, 0, 1);
        make_av(0);
        if (PROBE) tstamp[1] = __builtin_amdgcn_s_memtime();
        for (int chunk = cbeg; chunk < nchunks; ++chunk) {
            const int nxt = chunk < last ? chunk + 1 : last;            // clamped: the tail re-loads valid memory, unused
            const int nxt2 = chunk + 2 < nchunks ? chunk + 2 : last;
    #pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int uu = g4 & 1;
                // The next group's two patch rows are requested ahead of / in the middle of this group's MFMAs (the second
                // row lands in registers the first half of the burst has released).  Group 3 reads the NEXT chunk's
                // buffer: its stores were made during this iteration; the barrier publishes them and also retires
                // every read of the buffer that iteration chunk+1 overwrites.
                if (g4 == 3) __syncthreads();
                const int rbuf = g4 < 3 ? (chunk & 1) : ((chunk + 1) & 1);
                const int rkg = g4 < 3 ? (g4 + 1) >> 1 : 0, ruu = (g4 + 1) & 1;
                issue_reads(rbuf, rkg, ruu, 0);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int v = 0; v < 2; ++v)
    #pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        acc[uu * 4 + v] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[v][kk], bvr[uu][v][kk], acc[uu * 4 + v], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                issue_reads(rbuf, rkg, ruu, 1);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int v = 2; v < 4; ++v)
    #pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        acc[uu * 4 + v] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[v][kk], bvr[uu][v][kk], acc[uu * 4 + v], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // ring: the set just consumed is refilled with the operands of the group after next
                if (g4 < 2) load_b(chunk, 1, uu, uu);
                else        load_b(nxt, 0, uu, uu);
                if (g4 == 0) {
                    // the next iteration's patch goes to the other buffer while this iteration's MFMAs are in flight
                    store_patch((chunk + 1) & 1);
                    load_patch(nxt2);
                }
                make_av(ruu);
            }
        }

    }
    if (PROBE) tstamp[2] = __builtin_amdgcn_s_memtime();
    // ---- epilogue.  The output transform Y = A^T M A is linear in M, so each wave forms the PARTIAL 2x2 outputs
    // of its 8 positions (A^T = [[1,1,1,0],[0,1,-1,-1]]: rows u=0,1 give tm0 = M0+M1, tm1 = M1; rows u=2,3 give
    // tm0 = M2, tm1 = -M2-M3), the two halves swap partials through LDS (each finalises 8 of the 16 tile rows).
    auto partial = [&](int r, float (&out4)[4]) {                  // partial 2x2 output of tile row r from this wave's 8 positions
        float tm[2][4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float m0 = acc[v][r], m1 = acc[4 + v][r];        // positions (2xh, v) and (2xh+1, v)
            if (xh == 0) { tm[0][v] = m0 + m1; tm[1][v] = m1; }
            else         { tm[0][v] = m0;      tm[1][v] = -m0 - m1; }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            out4[i * 2 + 0] = tm[i][0] + tm[i][1] + tm[i][2];
            out4[i * 2 + 1] = tm[i][1] - tm[i][2] - tm[i][3];
        }
    };
    const int n = n0 + bcol;
    const bool nok = n < N;
    const int nc = nok ? n : 0;
    // split-K: raw partial sums go to this block's slab ws[ksi][b][pixel][N]; the tail launch of anoddpm_igemm
    // folds the slabs and applies alpha / bias / temb / residual / statistics
    const bool part = ksplit > 1;
    float *__restrict__ O = part ? a.ws + ((int64_t)ksi * a.B + b) * ((int64_t)H * W) * N : a.out + (int64_t)b * a.o_bs;
    const int o_ld = part ? N : a.out_ld;
    const float alpha = part ? 1.f : a.alpha;
    const float *__restrict__ R = (a.res && !part) ? a.res + (int64_t)b * a.r_bs : nullptr;
    float add = 0.f;
    if (a.bias && !part) add += a.bias[nc];
    if (a.temb && !part) add += a.temb[(int64_t)b * a.temb_ld + nc];
    float cs = 0.f, cq = 0.f;
    float rv[8][4];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {                                // all residual loads first
        const int r = xh * 8 + rr;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int tile = wm * 32 + row;
        const int oy = y0 + 2 * (tile >> 3), ox = x0 + 2 * (tile & 7);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            rv[rr][e] = R ? R[((int64_t)(oy + (e >> 1)) * W + ox + (e & 1)) * a.res_ld + nc] : 0.f;
    }
    // exchange buffer (re-uses the V region; the loop's last barrier has passed): ex[pair][writer xh][8 rows][4][64 lanes]
    float *ex = lds;
    const int pair = wave & (WMW * WNW - 1);
    if (xh == 0) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {                            // rows 8..15 are finalised by the xh = 1 wave
            float p4[4];
            partial(8 + rr, p4);
#pragma unroll
            for (int e = 0; e < 4; ++e) ex[(((pair * 2 + 0) * 8 + rr) * 4 + e) * 64 + lane] = p4[e];
        }
    } else {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {                            // rows 0..7 are finalised by the xh = 0 wave
            float p4[4];
            partial(rr, p4);
#pragma unroll
            for (int e = 0; e < 4; ++e) ex[(((pair * 2 + 1) * 8 + rr) * 4 + e) * 64 + lane] = p4[e];
        }
    }
    __syncthreads();
    if (PROBE) tstamp[3] = __builtin_amdgcn_s_memtime();

    auto finalize = [&](int rbase) {                               // rbase is a literal (0 or 8): static accumulator indices
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = rbase + rr;
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int tile = wm * 32 + row;
            const int oy = y0 + 2 * (tile >> 3), ox = x0 + 2 * (tile & 7);
            float p4[4];
            partial(r, p4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float other = ex[(((pair * 2 + (1 - xh)) * 8 + rr) * 4 + e) * 64 + lane];
                const float v = alpha * (p4[e] + other) + add + rv[rr][e];
                if (nok) {
                    O[((int64_t)(oy + (e >> 1)) * W + ox + (e & 1)) * o_ld + nc] = v;
                    cs += v;
                    cq += v * v;
                }
            }
        }
    };
    if (xh == 0) finalize(0); else finalize(8);
    if (a.stats && !part) {
        float *st = a.stats + ((int64_t)b * (gridDim.x * 2 * WMW) + blockIdx.x * (2 * WMW) + wm * 2 + xh) * N * 2;
        const float s2 = cs + __shfl_xor(cs, 32);
        const float q2 = cq + __shfl_xor(cq, 32);
        if (h == 0 && nok) { st[n * 2] = s2; st[n * 2 + 1] = q2; }
    }
    if (PROBE) {
        __builtin_amdgcn_s_waitcnt(0);                               // stores retired: what the wave waits for before it can end
        tstamp[4] = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(a.ws) +
                                      (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8;
            for (int i = 0; i < 5; ++i) dbg[i] = tstamp[i];
        }
    }
}

}  // namespace

namespace anoddpm {

// Called by anoddpm_igemm for cfg == 2 (arguments already validated there).
int launch_winograd(const anoddpm_igemm_args *a, hipStream_t s)
{
    ANODDPM_REQUIRE(a->ks == 3 && a->b_mode == 0 && a->heads == 1, "winograd: needs a 3x3 conv with packed weights");
    ANODDPM_REQUIRE(a->a_mode == 0 || a->a_mode == 1, "winograd: pooled operand loads use the direct kernel");
    ANODDPM_REQUIRE(a->H % 16 == 0 && a->W % 16 == 0, "winograd: H and W must be multiples of 16");
    const int K = a->c0 + a->c1;
    ANODDPM_REQUIRE(K % WKC == 0 && (a->c1 == 0 || a->c0 % WKC == 0), "winograd: channel counts must be multiples of 16");
    const int cps = (K / WKC + a->ksplit - 1) / a->ksplit;
    ANODDPM_REQUIRE((a->ksplit - 1) * cps < K / WKC, "winograd: ksplit leaves a block without channels");
    ANODDPM_REQUIRE(a->ksplit == 1 || !a->stats || a->stats_rows >= 1, "winograd: split-K statistics need stats_rows");
    // Workgroup shape: 8x16 output patch; all 128 channels in one 512-thread workgroup when N is a multiple of 128
    // (each patch element is activated once), else 64 channels in a 256-thread workgroup (two per CU).
    // ANODDPM_DEBUG0: 1 = the 16x16-patch 512-thread shape, 2 = force the 64-channel shape, 3 = per-wave transforms.
    const int dbg = anoddpm::g_debug[0];
    const int wmw = dbg == 1 ? 2 : 1;
    const bool fast = a->gn_scale && a->act;
#ifdef ANODDPM_ABLATE
    const bool probe = anoddpm::g_debug[1] == 1 && fast && a->ksplit == 1 && a->ws && wmw == 1;   // tools/wino_phases.py (writes ticks into ws)
#else
    const bool probe = false;
#endif
    const int wnw = (wmw == 1 && dbg != 2 && !probe && a->N % 128 == 0) ? 4 : 2;
    const int wbn = 32 * wnw;
    dim3 grid((unsigned)((a->H / (8 * wmw)) * (a->W / 16)), (unsigned)((a->N + wbn - 1) / wbn), (unsigned)(a->B * a->ksplit));
    ANODDPM_REQUIRE(grid.y <= 65535 && (int64_t)a->B * a->ksplit <= 65535, "winograd: grid too large");
    ANODDPM_REQUIRE((int64_t)16 * K * a->N * 4 < ((int64_t)1 << 31), "winograd: transformed weights exceed 32-bit buffer offsets");
    ANODDPM_REQUIRE((int64_t)a->H * a->W * (a->a0_ld > a->a1_ld ? a->a0_ld : a->a1_ld) * 4 < ((int64_t)1 << 31),
                    "winograd: operand slice exceeds 32-bit buffer offsets");
    if (wmw == 2) {
        if (fast) hipLaunchKernelGGL((wino_kernel<true, 2, 2>), grid, dim3(512), 0, s, *a);
        else      hipLaunchKernelGGL((wino_kernel<false, 2, 2>), grid, dim3(512), 0, s, *a);
#ifdef ANODDPM_ABLATE
    } else if (probe) {
        hipLaunchKernelGGL((wino_kernel<true, 1, 2, true>), grid, dim3(256), 0, s, *a);
#endif
    } else if (wnw == 4 && dbg != 3) {                              // shared input transform (ANODDPM_DEBUG0=3 disables)
        if (fast) hipLaunchKernelGGL((wino_kernel<true, 1, 4, false, true>), grid, dim3(512), 0, s, *a);
        else      hipLaunchKernelGGL((wino_kernel<false, 1, 4, false, true>), grid, dim3(512), 0, s, *a);
    } else if (wnw == 4) {
        if (fast) hipLaunchKernelGGL((wino_kernel<true, 1, 4>), grid, dim3(512), 0, s, *a);
        else      hipLaunchKernelGGL((wino_kernel<false, 1, 4>), grid, dim3(512), 0, s, *a);
    } else {
        if (fast) hipLaunchKernelGGL((wino_kernel<true, 1, 2>), grid, dim3(256), 0, s, *a);
        else      hipLaunchKernelGGL((wino_kernel<false, 1, 2>), grid, dim3(256), 0, s, *a);
    }
    return check_launch("winograd");
}

}  // namespace anoddpm
