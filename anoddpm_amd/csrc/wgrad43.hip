// Weight gradient of the 3x3 convolutions in the Winograd F(4x4,3x3) domain (algo == 1 of anoddpm_conv3x3_wgrad), gfx950.
//
// The forward kernel (winograd43.hip) computes  Y = A^T [ sum_ci U(co,ci) (.) V(ci) ] A  per 4x4 output tile, with
// U = G g G^T and V = B^T d B (d = the activated 6x6 input tile).  Its exact adjoint w.r.t. the weights is
//     dU[pos][ci][co] = sum over tiles of  V[pos][tile][ci] * Z[pos][tile][co],     Z = A dY A^T  (6x6 from the 4x4 dY tile),
//     dg (3x3)        = G^T dU G,
// i.e. 36 multiplies per (ci, co) and tile instead of the 144 of the direct form (wgrad.hip): 4x fewer MFMAs, and the gradient
// of exactly the function the forward kernel evaluates.  Replaces the autograd backward of nn.Conv2d w.r.t. its weight
// (UNet.py:172,193; diffusion_training.py:102) for the layers the forward runs on the F(4x4,3x3) kernel.
//
// Mapping.  36 positions x (32 input x 64 output channels) of accumulators = 288 KB: one 768-thread workgroup per CU (12 waves,
// wave w owns positions 3w..3w+2: 3 x 2 x 4 accumulator tiles of 16 x 16, v_mfma_f32_16x16x4_f32 with the reduction over tiles).
// A workgroup = one (32 ci, 64 co) block x a strided set of 16x8-pixel output patches (8 tiles), which it walks one by one:
//   1. the activated 18x10 input patch (2 chunks of 16 channels) and the dY patch (4 chunks of 16 channels, two at a time) are
//      staged in LDS from registers loaded one phase ahead (GroupNorm-apply + SiLU, nearest-x2 / concat resolved on the load);
//   2. all twelve waves transform: V = B^T d B and Z = A dY A^T (two rounds of 32 output channels), one wave-item per wave and
//      transform -- an item = (tile, channel, ROW PAIR of B^T / A): the pairs (1,2) and (3,4) share their partial sums, so a pair
//      costs what one row cost (round 6; until round 5: 768 + 2 x 768 single-row items on packed channel pairs); the (1,1,1,1) row
//      of the Z transform also emits the column sums of dY per tile row of the patch (bias / embedding gradients) for the
//      workgroups of the first input-channel block;
//   3. 48 MFMAs per wave: operands are 4-byte LDS reads of V / Z [pos][tile][channel] (16 channels x 4 tiles per wave read).
// VALU (transforms) and MFMA phases alternate -- on this hardware they share the issue port anyway -- four barriers per patch.
// The per-workgroup gradient goes to a workspace [PG][9][K][N] with dg = G^T dU G applied in the epilogue (two exchanges through LDS);
// one small kernel sums the PG slabs into dw.
#include "common.h"

using anoddpm::silu_f;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_g(const float *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7ffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 bld4g(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned wave_bytes)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)wave_bytes, 0));
}

constexpr int G4_NT = 768;
constexpr int G4_PW = 18;                          // input patch: 18 x 10 pixels (16 x 8 outputs + halo)
constexpr int G4_PPIX = 180;
constexpr int G4_PITCH = 5;                        // float4 per staged pixel (16 channels + pad)
constexpr int G4_TILES = 8;                        // 4 x 2 tiles of 4 x 4 outputs
constexpr int G4_V = 36 * G4_TILES * 4;            // float4 per transformed 16-channel chunk: [pos][tile][quad]
constexpr int G4_KB = 32, G4_NB = 64;
constexpr int G4_PIN = G4_PPIX * G4_PITCH;         // float4 per input staging buffer
constexpr int G4_PDY = 128 * G4_PITCH;             // float4 per dY staging buffer
constexpr int G4_AFF = 240;                        // float4: GroupNorm affine of the workgroup's 32 channels for up to 15 images
constexpr int G4_LDS4 = 6 * G4_V + 2 * G4_PIN + 2 * G4_PDY + G4_AFF;      // 10,232 float4 = 163,712 B of the CU's 163,840
static_assert(G4_LDS4 * 16 <= 160 * 1024, "wgrad43: LDS budget");

// PROBE (measurement builds, ANODDPM_DEBUG12=1; tools/wgrad_phases.py): wave 0 sums s_memtime over the five phases of every patch and
// leaves { stage, V + Z0 transforms, dY round 1, Z1 transform + requests, MFMAs, patches } behind the workspace slabs.
template <bool PROBE>
__global__ __launch_bounds__(G4_NT, 1) void wgrad43_kernel(const anoddpm_wgrad_args a, const int PG, const int tiles_x, const int tiles_y, const float inv_tx)
{
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tprev = 0;
    auto mark = [&](const int k) {
        if (PROBE) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            ph[k] += t - tprev;
            tprev = t;
        }
    };
    __shared__ __attribute__((aligned(16))) f32x4 lds4[G4_LDS4];
    f32x4 *ldsV = lds4, *ldsZ = lds4 + 2 * G4_V, *ldsPin = ldsZ + 4 * G4_V, *ldsPdy = ldsPin + 2 * G4_PIN, *ldsAff = ldsPdy + 2 * G4_PDY;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, N = a.N, K = a.c0 + a.c1;
    const int nkb = K / G4_KB;
    const int kb = blockIdx.y % nkb, nb = blockIdx.y / nkb;
    const int k0 = kb * G4_KB, n0 = nb * G4_NB;
    const int ppi = tiles_x * tiles_y, patches = a.B * ppi;
    const int pg = blockIdx.x;
    const int a_mode = a.a_mode;

    // GroupNorm affine of this block's 32 input channels for every image: [b][chunk][quad] {scale}, then {shift}
    for (int i = tid; i < a.B * 8; i += G4_NT) {
        const int b = i >> 3, q = i & 7;
        ldsAff[i] = *reinterpret_cast<const f32x4 *>(a.gn_scale + (int64_t)b * a.gn_ld + k0 + q * 4);
        ldsAff[a.B * 8 + i] = *reinterpret_cast<const f32x4 *>(a.gn_shift + (int64_t)b * a.gn_ld + k0 + q * 4);
    }

    // ---- staging geometry (fixed per thread) ----
    const int sq = tid & 3, sp_pix = tid >> 2;                      // input slot: pixel 0..179 (tid < 720), quad
    const bool in_slot = tid < G4_PPIX * 4;
    const int spy = sp_pix / G4_PW, spx = sp_pix - spy * G4_PW;
    // dY slots: s = tid + j*768 (j = 0, 1), 1024 per round: chunk-in-round = s >> 9, pixel = (s & 511) >> 2, quad = s & 3
    f32x4 in_raw[2], dy_raw[2];
    int in_valid = 0;
    // patch coordinates are wave-uniform and advance incrementally (no integer division in the loop): image b, patch r of the image
    struct Geo { int b, r, y0, x0; };
    auto locate = [&](Geo &g) {
        const int ty = (int)(((float)g.r + 0.5f) * inv_tx);           // exact for r < 2^23
        g.y0 = ty * 8;
        g.x0 = (g.r - ty * tiles_x) * 16;
    };
    auto advance = [&](Geo g) {
        g.r += PG;
        while (g.r >= ppi) { g.r -= ppi; ++g.b; }
        if (g.b >= a.B) { g.b = a.B - 1; g.r = ppi - 1; }            // past the end: a valid patch, loaded and never used
        locate(g);
        return g;
    };
    // 32-bit buffer addressing (tensors < 2 GB, checked by the launcher): per-lane byte offset + wave-uniform byte offset
    const __amdgpu_buffer_rsrc_t rA0 = rsrc_g(a.a0), rA1 = rsrc_g(a.a1 ? a.a1 : a.a0), rDY = rsrc_g(a.dy);
    auto load_in = [&](const Geo &g) {
        const int gy = g.y0 + spy - 1, gx = g.x0 + spx - 1;
        const bool ok = in_slot && gy >= 0 && gy < H && gx >= 0 && gx < W;
        in_valid = ok ? 1 : 0;
        const unsigned sp = ok ? (unsigned)((a_mode == 0) ? gy * W + gx : (gy >> 1) * (W >> 1) + (gx >> 1)) : 0u;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int kc = k0 + c * 16;
            const bool first = kc < a.c0;                           // wave-uniform
            const unsigned ld = (unsigned)(first ? a.a0_ld : a.a1_ld);
            const unsigned wave_off = (unsigned)g.b * (unsigned)(first ? a.a0_bs : a.a1_bs) * 4u + (unsigned)(first ? kc : kc - a.c0) * 4u;
            in_raw[c] = bld4g(first ? rA0 : rA1, (sp * ld + (unsigned)sq * 4u) * 4u, wave_off);
        }
    };
    auto store_in = [&](const Geo &g) {
        const int b = g.b;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x4 sc = ldsAff[b * 8 + c * 4 + sq], sh = ldsAff[a.B * 8 + b * 8 + c * 4 + sq];
            f32x4 v = in_raw[c] * sc + sh;
            v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]);
            if (!in_valid) v = f32x4{0.f, 0.f, 0.f, 0.f};            // zero padding of the ACTIVATED map
            if (in_slot) ldsPin[c * G4_PIN + sp_pix * G4_PITCH + sq] = v;
        }
    };
    // dY slot geometry is fixed per thread: chunk-in-round, pixel, quad of slots tid and tid + 768 (the second clamped to 1023)
    unsigned dy_lane[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int s0 = tid + j * G4_NT;
        const int sc = s0 < 1024 ? s0 : 1023;
        const int p = (sc & 511) >> 2, q = sc & 3;
        dy_lane[j] = ((unsigned)((p >> 4) * W + (p & 15)) * (unsigned)a.dy_ld + (unsigned)((sc >> 9) * 16 + q * 4)) * 4u;
    }
    auto load_dy = [&](const Geo &g, int round) {
        const unsigned wave_off = ((unsigned)g.b * (unsigned)a.dy_bs + (unsigned)(g.y0 * W + g.x0) * (unsigned)a.dy_ld +
                                   (unsigned)(n0 + round * 32)) * 4u;
#pragma unroll
        for (int j = 0; j < 2; ++j) dy_raw[j] = bld4g(rDY, dy_lane[j], wave_off);
    };
    auto store_dy = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int s = tid + j * G4_NT;
            if (s < 1024) ldsPdy[(s >> 9) * G4_PDY + ((s & 511) >> 2) * G4_PITCH + (s & 3)] = dy_raw[j];
        }
    };

    const __amdgpu_buffer_rsrc_t rCS = rsrc_g(a.colsum ? a.colsum : a.dy);
    // Column sums of dY (bias / embedding gradients): the waves of row pair 1 of the first input-channel block sum their tile row
    // over the workgroup's patches of an image and store ONE row per image: colsum[b][pg * 2 + tile row][n] (2 PG rows per image
    // instead of one per tile row of every patch: 64 instead of 1 024 at 256^2); images the workgroup has no patch in get zeros.
    const bool cs_role = a.colsum != nullptr && kb == 0 && (wave % 3) == 1;       // wave-uniform
    float cs_acc0 = 0.f, cs_acc1 = 0.f;
    // ---- round 6: transform items that form TWO rows of B^T d / A dY from one set of reads, on SINGLE channels.  The row pairs (1,2)
    // and (3,4) share their partial sums (u1, u2 = p +- q; u3, u4 = r +- 2 s), (0,5) read disjoint rows: 4 instead of 8 operations and
    // half the LDS reads per column and row pair, with coefficients that are literals instead of per-wave registers.
    // lane = (tile column lane >> 4, channel lane & 15), wave = (row pair w % 3, chunk slot (w / 3) & 1, tile row w / 6): V, Z of
    // round 0 and Z of round 1 are twelve wave-items each -- one per wave and phase, and every SIMD (waves w, w + 4, w + 8) hosts one
    // wave of each row pair.  150 VGPRs, nothing spilled: the packed-pair form of the same idea (f32x2 items, six wave-items per
    // phase) kept 24 more registers live and reloaded spills inside the loop -- vmcnt retires in order, so a reload behind the
    // dY requests waited out their HBM latency in every patch and the step gained nothing (profiles/r6_wgrad43_ab.txt).
    const int s_tx = lane >> 4, s_ch = lane & 15;
    const int s_up = wave % 3, s_slot = (wave / 3) & 1, s_trow = wave / 6;
    constexpr int SP = G4_PITCH * 4, SVROW = G4_PW * SP, SZROW = 16 * SP;           // floats per staged pixel / input row / dY row
    const int s_vin = ((4 * s_trow) * G4_PW + 4 * s_tx) * SP + s_ch;
    const int s_zin = ((4 * s_trow) * 16 + 4 * s_tx) * SP + s_ch;
    const int s_out = (s_trow * 4 + s_tx) * 16 + s_ch;                               // [pos][tile][channel]: + pos * 128
    const int s_ua = s_up == 0 ? 0 : (s_up == 1 ? 1 : 3), s_ub = s_up == 0 ? 5 : (s_up == 1 ? 2 : 4);
    auto col_pass_v1 = [&](const float (&t)[6], float *V) {
        const float p = t[4] - 4.f * t[2], q = t[3] - 4.f * t[1], r = t[4] - t[2], w = t[3] - t[1];
        V[0 * 128] = 4.f * t[0] - 5.f * t[2] + t[4];
        V[1 * 128] = p + q;
        V[2 * 128] = p - q;
        V[3 * 128] = r + 2.f * w;
        V[4 * 128] = r - 2.f * w;
        V[5 * 128] = 4.f * t[1] - 5.f * t[3] + t[5];
    };
    auto transform_v_s = [&]() {
        const float *D = reinterpret_cast<const float *>(ldsPin + s_slot * G4_PIN) + s_vin;
        float ta[6], tb[6];
        if (s_up == 0) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d0 = D[j * SP], d1 = D[SVROW + j * SP], d2 = D[2 * SVROW + j * SP], d3 = D[3 * SVROW + j * SP], d4 = D[4 * SVROW + j * SP], d5 = D[5 * SVROW + j * SP];
                ta[j] = 4.f * d0 - 5.f * d2 + d4;
                tb[j] = 4.f * d1 - 5.f * d3 + d5;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (s_up == 1) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d1 = D[SVROW + j * SP], d2 = D[2 * SVROW + j * SP], d3 = D[3 * SVROW + j * SP], d4 = D[4 * SVROW + j * SP];
                const float p = d4 - 4.f * d2, q = d3 - 4.f * d1;
                ta[j] = p + q;
                tb[j] = p - q;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d1 = D[SVROW + j * SP], d2 = D[2 * SVROW + j * SP], d3 = D[3 * SVROW + j * SP], d4 = D[4 * SVROW + j * SP];
                const float r = d4 - d2, w = d3 - d1;
                ta[j] = r + 2.f * w;
                tb[j] = r - 2.f * w;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float *V = reinterpret_cast<float *>(ldsV + s_slot * G4_V) + s_out;
        col_pass_v1(ta, V + s_ua * 6 * 128);
        col_pass_v1(tb, V + s_ub * 6 * 128);
    };
    auto col_pass_z1 = [&](const float (&t)[4], float *Z) {
        const float e02 = t[0] + t[2], o13 = t[1] + t[3], e024 = t[0] + 4.f * t[2], o138 = 2.f * t[1] + 8.f * t[3];
        Z[0 * 128] = t[0];
        Z[1 * 128] = e02 + o13;
        Z[2 * 128] = e02 - o13;
        Z[3 * 128] = e024 + o138;
        Z[4 * 128] = e024 - o138;
        Z[5 * 128] = t[3];
    };
    auto transform_z_s = [&](const int round, const Geo &g) {
        const float *D = reinterpret_cast<const float *>(ldsPdy + s_slot * G4_PDY) + s_zin;
        float ta[4], tb[4];
        if (s_up == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { ta[j] = D[j * SP]; tb[j] = D[3 * SZROW + j * SP]; }
        } else if (s_up == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = D[j * SP] + D[2 * SZROW + j * SP], o = D[SZROW + j * SP] + D[3 * SZROW + j * SP];
                ta[j] = e + o;
                tb[j] = e - o;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = D[j * SP] + 4.f * D[2 * SZROW + j * SP], o = 2.f * D[SZROW + j * SP] + 8.f * D[3 * SZROW + j * SP];
                ta[j] = e + o;
                tb[j] = e - o;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int zc = 2 * round + s_slot;
        float *Z = reinterpret_cast<float *>(ldsZ + zc * G4_V) + s_out;
        col_pass_z1(ta, Z + s_ua * 6 * 128);
        col_pass_z1(tb, Z + s_ub * 6 * 128);
        if (cs_role) {                                              // row (1,1,1,1): ta[j] are the column sums of the lane's tile
            float cs = (ta[0] + ta[2]) + (ta[1] + ta[3]);
            cs += __shfl_xor(cs, 16);                                // the wave's four tiles (one tile row of the patch)
            cs += __shfl_xor(cs, 32);
            if (round == 0) cs_acc0 += cs; else cs_acc1 += cs;      // summed over the workgroup's patches of an image (cs_flush)
        }
    };

    // ---- accumulators: positions 3*wave + p, 2 input-channel tiles x 4 output-channel tiles of 16 x 16
    f32x4 acc[3][2][4];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[p][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, kq = lane >> 4;
    const float *Vf = reinterpret_cast<const float *>(ldsV), *Zf = reinterpret_cast<const float *>(ldsZ);

    auto cs_store = [&](const int b, const float v0, const float v1) {
        if (s_tx == 0) {
            const unsigned row = ((unsigned)b * 2u * (unsigned)PG + (unsigned)pg * 2u + (unsigned)s_trow) * (unsigned)N + (unsigned)n0;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rCS, (int)(s_ch * 4), (int)((row + (unsigned)(s_slot * 16)) * 4u), 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rCS, (int)(s_ch * 4), (int)((row + (unsigned)((2 + s_slot) * 16)) * 4u), 0);
        }
    };
    Geo cur;
    cur.b = pg / ppi;
    cur.r = pg - cur.b * ppi;
    locate(cur);
    const int iters = (patches - pg + PG - 1) / PG;                  // pg < patches (launcher)
    load_in(cur);
    load_dy(cur, 0);
    __syncthreads();                                                // affine table
    int cs_b = cur.b;
    if (cs_role)
        for (int bb = 0; bb < cs_b; ++bb) cs_store(bb, 0.f, 0.f);
    if (PROBE) tprev = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (cs_role && cur.b != cs_b) {                             // the walk reached another image: its predecessor's sums are final
            cs_store(cs_b, cs_acc0, cs_acc1);
            for (int bb = cs_b + 1; bb < cur.b; ++bb) cs_store(bb, 0.f, 0.f);
            cs_b = cur.b;
            cs_acc0 = cs_acc1 = 0.f;
        }
        store_dy();
        load_dy(cur, 1);                                            // requested BEFORE the input staging: its GroupNorm + SiLU pass covers the latency
        store_in(cur);
        __syncthreads();
        mark(0);
        transform_v_s();
        transform_z_s(0, cur);
        __syncthreads();
        mark(1);
        store_dy();
        const Geo nxt = advance(cur);                               // prefetch of the next patch
        __syncthreads();
        mark(2);
        transform_z_s(1, cur);
        load_in(nxt);                                               // requested behind the last transform (no request registers live across
        load_dy(nxt, 0);                                            // it); the MFMA phase covers the latency: -2.3 % against requesting before it
        cur = nxt;
        __syncthreads();
        mark(3);
        // (round 6, measured and dropped: the twelve operand reads of position p + 1 issued before the sixteen MFMAs of position p --
        // 164 VGPRs, no spill -- 10.10-10.12 against 10.00 ms of weight-gradient ops per step; the compiler's own order stays)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int pos = wave * 3 + p;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int off = ((pos * G4_TILES + 4 * e + kq) * 16 + l15);
                float va[2], zb[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) va[i] = Vf[i * G4_V * 4 + off];
#pragma unroll
                for (int j = 0; j < 4; ++j) zb[j] = Zf[j * G4_V * 4 + off];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[i], zb[j], acc[p][i][j], 0, 0, 0);
            }
        }
        mark(4);
        // no barrier here: the next iteration only writes the staging buffers before its first barrier
    }
    if (PROBE && tid == 0) {
        unsigned long long *d = reinterpret_cast<unsigned long long *>(a.ws + (int64_t)PG * 9 * K * N) + ((int64_t)blockIdx.y * PG + pg) * 8;
#pragma unroll
        for (int k = 0; k < 5; ++k) d[k] = ph[k];
        d[5] = (unsigned long long)iters;
    }

    if (cs_role) {
        cs_store(cs_b, cs_acc0, cs_acc1);
        for (int bb = cs_b + 1; bb < a.B; ++bb) cs_store(bb, 0.f, 0.f);
    }
    // ---- partial gradient of this workgroup with dg = G^T dU G applied (round 6; until then the 36 planes of dU went to the workspace).
    // First half, r[u][b] = sum_v dU[u][v] G[v][b]: wave w holds half a row of dU -- u = w >> 1, v = 3 (w & 1) + {0, 1, 2} -- so it forms
    // its three partial sums in registers and the odd wave of a pair hands them to the even one through LDS (lane to lane: both map
    // lanes to (k, n) alike).
    __syncthreads();                                                // every wave is past its last MFMA operand reads: LDS is scratch now
    {
        // G (6 x 3): rows (1/4, 0, 0), (-1/6, -1/6, -1/6), (-1/6, 1/6, -1/6), (1/24, 1/12, 1/6), (1/24, -1/12, 1/6), (0, 0, 1)
        const bool hi = (wave & 1) != 0;                            // wave-uniform
        const int u = wave >> 1;
        float *xch = reinterpret_cast<float *>(lds4) + (u * 3) * 32 * 64 + lane;      // [u][b][element e = (i, j, r)][lane]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d0 = acc[0][i][j][r], d1 = acc[1][i][j][r], d2 = acc[2][i][j][r];
                    float r0, r1, r2;
                    if (!hi) {                                      // v = 0, 1, 2
                        const float s12 = d1 + d2, m12 = d2 - d1;
                        r0 = 0.25f * d0 - (1.f / 6) * s12;
                        r1 = (1.f / 6) * m12;
                        r2 = -(1.f / 6) * s12;
                    } else {                                        // v = 3, 4, 5
                        const float s01 = d0 + d1, m01 = d0 - d1;
                        r0 = (1.f / 24) * s01;
                        r1 = (1.f / 12) * m01;
                        r2 = (1.f / 6) * s01 + d2;
                    }
                    acc[0][i][j][r] = r0; acc[1][i][j][r] = r1; acc[2][i][j][r] = r2;
                }
        if (hi) {
#pragma unroll
            for (int bb = 0; bb < 3; ++bb)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) xch[(bb * 32 + (i * 4 + j) * 4 + r) * 64] = acc[bb][i][j][r];
        }
        __syncthreads();
        if (!hi) {                                                  // r[u][b] complete, back into the exchange buffer (same lane, same address)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb)
#pragma unroll
                for (int e = 0; e < 32; ++e) {
                    float *q = xch + (bb * 32 + e) * 64;
                    *q = acc[bb][e >> 4][(e >> 2) & 3][e & 3] + *q;
                }
        }
        __syncthreads();
        // second half: dg[a][b] = sum_u G[u][a] r[u][b].  All twelve waves: wave = (column b = w % 3, eight of the 32 elements e = (i, j, r)
        // per lane: group w / 3); six LDS reads -> three stores.  ws[pg][a * 3 + b][k][n]: NINE planes per (k, n), a quarter of the
        // round-5 slab traffic of this kernel and of the fold (which only sums the PG slabs now).
        {
            const int b2 = wave % 3, eg = wave / 3;
            const float *R = reinterpret_cast<const float *>(lds4) + lane;
#pragma unroll
            for (int ee = 0; ee < 8; ++ee) {
                const int e = eg * 8 + ee;                          // = (i * 4 + j) * 4 + r
                const int i = e >> 4, j = (e >> 2) & 3, r = e & 3;
                float ru[6];
#pragma unroll
                for (int uu = 0; uu < 6; ++uu) ru[uu] = R[((uu * 3 + b2) * 32 + e) * 64];
                const float s12 = ru[1] + ru[2], m12 = ru[2] - ru[1], s34 = ru[3] + ru[4], m34 = ru[3] - ru[4];
                const float g0 = 0.25f * ru[0] - (1.f / 6) * s12 + (1.f / 24) * s34;
                const float g1 = (1.f / 6) * m12 + (1.f / 12) * m34;
                const float g2 = -(1.f / 6) * s12 + (1.f / 6) * s34 + ru[5];
                float *dst = a.ws + (((int64_t)pg * 9 + b2) * K + k0 + 4 * kq + i * 16 + r) * N + n0 + l15 + j * 16;
                const int64_t plane3 = (int64_t)3 * K * N;          // a -> a + 1: three planes further
                dst[0] = g0;
                dst[plane3] = g1;
                dst[2 * plane3] = g2;
            }
        }
    }
}

// ws[0][ab][k][n] <- sum over the PG slabs, fixed order (thread = one (ab, k, n); 16 slabs in flight)
__global__ __launch_bounds__(256) void wgrad43_sum_kernel(float *ws, const int64_t slab, const int PG)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= slab) return;
    const float *p = ws + idx;
    float s = 0.f;
    int g = 0;
    for (; g + 16 <= PG; g += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(int64_t)(g + u) * slab];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; g < PG; ++g) s += p[(int64_t)g * slab];
    ws[idx] = s;
}

// dw[n][k][a][b] (+)= ws[0][a * 3 + b][k][n]   (thread = one (k, n))
__global__ __launch_bounds__(256) void wgrad43_out_kernel(const anoddpm_wgrad_args a)
{
    const int K = a.c0 + a.c1, N = a.N;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)K * N) return;
    const int n = (int)(idx % N), k = (int)(idx / N);
    float *o = a.dw + ((int64_t)n * K + k) * 9;
#pragma unroll
    for (int ab = 0; ab < 9; ++ab) {
        const float s = a.ws[((int64_t)ab * K + k) * N + n];
        o[ab] = a.accumulate ? o[ab] + s : s;
    }
}

// Both steps in one launch: workgroup = two input channels k x 64 output channels, 288 threads.  Thread (k of the pair, plane ab,
// lane quad q) folds four output channels of its plane over the PG slabs (slab order; sixteen 16-byte loads in flight), the
// 2 x 9 x 64 sums meet in LDS and leave as dw[n][k][3][3] (36-byte runs per (n, k)).  Same order of every sum as wgrad43_sum_kernel +
// wgrad43_out_kernel (bit-identical), one launch and one pass over the slabs.  blk0: block offset (the column-sum blocks alone).
__global__ __launch_bounds__(288) void wgrad43_fold_kernel(const anoddpm_wgrad_args a, const int PG, const int blk0)
{
    const int bid = (int)blockIdx.x + blk0;
    __shared__ __attribute__((aligned(16))) float rr[2][9][64];
    const int K = a.c0 + a.c1, N = a.N;
    const int tiles_n = N >> 6;
    if (bid >= (K >> 1) * tiles_n) {
        // the N / 32 blocks behind the weight blocks fold the kernel's column sums (2 PG rows per image): dimg[b][n] = the image's
        // sum of dy (embedding gradient), dbias[n] += their sum over the images in index order -- what anoddpm_colsum_fold does in
        // a launch of its own.  32 channels x 9 row lanes.
        float (*red)[32] = reinterpret_cast<float (*)[32]>(&rr[0][0][0]);
        const int l = threadIdx.x & 31, il = threadIdx.x >> 5;
        const int n = (bid - (K >> 1) * tiles_n) * 32 + l;
        const int rows = 2 * PG;
        float tot = 0.f;
        for (int b = 0; b < a.B; ++b) {
            float sacc = 0.f;
            const float *p = a.colsum + ((int64_t)b * rows) * N + n;
            for (int i = il; i < rows; i += 9) sacc += p[(int64_t)i * N];
            red[il][l] = sacc;
            __syncthreads();
            if (il == 0) {
                float t = 0.f;
#pragma unroll
                for (int kk = 0; kk < 9; ++kk) t += red[kk][l];
                a.dimg[(int64_t)b * N + n] = t;
                tot += t;
            }
            __syncthreads();
        }
        if (il == 0 && a.dbias) a.dbias[n] += tot;
        return;
    }
    const int k0 = (bid / tiles_n) * 2, n0 = (bid % tiles_n) * 64;
    const int kk = threadIdx.x / 144, rem = threadIdx.x - kk * 144;
    const int plane = rem >> 4, q = rem & 15;
    const int64_t slab = (int64_t)9 * K * N;
    {
        const float *p = a.ws + ((int64_t)plane * K + k0 + kk) * N + n0 + 4 * q;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int g = 0;
        for (; g + 16 <= PG; g += 16) {
            f32x4 x[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p + (int64_t)(g + i) * slab));
#pragma unroll
            for (int i = 0; i < 16; ++i) s += x[i];
        }
        for (; g < PG; ++g) s += *reinterpret_cast<const f32x4 *>(p + (int64_t)g * slab);
        *reinterpret_cast<f32x4 *>(&rr[kk][plane][4 * q]) = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 9 * 64; i += 288) {            // (n, k of the pair, ab): 18 consecutive floats of dw per n
        const int n = i / 18, r18 = i - n * 18, k2 = r18 / 9, ab = r18 - k2 * 9;
        const float v = rr[k2][ab][n];
        float *o = a.dw + ((int64_t)(n0 + n) * K + k0 + k2) * 9 + ab;
        *o = a.accumulate ? *o + v : v;
    }
}

}  // namespace

namespace anoddpm {

// Patch groups of the Winograd-domain weight gradient: one workgroup per CU in total (shared with the host side through
// anoddpm_wgrad43_groups so that the caller can size the workspace: PG * 36 * K * N floats).
int wgrad43_groups(int K, int N, int B, int H, int W)
{
    const int blocks = (K / G4_KB) * (N / G4_NB);
    const int patches = B * (H / 8) * (W / 16);
    int pg = 256 / (blocks > 0 ? blocks : 1);
    if (pg < 1) pg = 1;
    if (pg > patches) pg = patches;
    return pg;
}

// column-sum rows per image the kernel writes: one per patch group and tile row (exported: callers size colsum with it)
int wgrad43_colsum_items(int K, int N, int B, int H, int W) { return 2 * wgrad43_groups(K, N, B, H, W); }

int launch_wgrad43(const anoddpm_wgrad_args *a, hipStream_t s)
{
    const int K = a->c0 + a->c1;
    ANODDPM_REQUIRE(a->gn_scale && a->gn_shift && a->act == 1, "wgrad (Winograd): needs the fused GroupNorm + SiLU operand");
    ANODDPM_REQUIRE(a->a_mode == 0 || a->a_mode == 1, "wgrad (Winograd): a_mode must be 0 or 1");
    ANODDPM_REQUIRE(a->H % 8 == 0 && a->W % 16 == 0 && K % G4_KB == 0 && a->N % G4_NB == 0 && (a->c1 == 0 || a->c0 % 16 == 0),
                    "wgrad (Winograd): H %% 8, W %% 16, K %% 32, N %% 64, c0 %% 16 must be 0");
    ANODDPM_REQUIRE(a->B <= 15, "wgrad (Winograd): batch > 15");
    ANODDPM_REQUIRE((int64_t)a->B * a->H * a->W * (K > a->N ? K : a->N) * 4 < ((int64_t)1 << 31) &&
                    (int64_t)a->B * a->a0_bs * 4 < ((int64_t)1 << 31) && (int64_t)a->B * a->dy_bs * 4 < ((int64_t)1 << 31),
                    "wgrad (Winograd): tensors must stay below 2 GB (32-bit buffer offsets)");
    const int pg = wgrad43_groups(K, a->N, a->B, a->H, a->W);
    ANODDPM_REQUIRE(!a->dimg || a->colsum, "wgrad (Winograd): dimg / dbias are folded from colsum");
    ANODDPM_REQUIRE(!a->dbias || a->dimg, "wgrad (Winograd): dbias needs dimg (the per-image sums it adds up)");
    ANODDPM_REQUIRE((int64_t)a->B * 2 * pg * a->N * 4 < ((int64_t)1 << 31), "wgrad (Winograd): colsum exceeds 32-bit buffer offsets");
    const int64_t slab = (int64_t)9 * K * a->N;                      // the kernel stores dg = G^T dU G of its patches: 9 planes per (k, n)
    ANODDPM_REQUIRE(a->ws_floats >= (int64_t)pg * slab, "wgrad (Winograd): workspace too small");
    const dim3 grid((unsigned)pg, (unsigned)((K / G4_KB) * (a->N / G4_NB)));
#ifdef ANODDPM_ABLATE
    if (g_debug[12] == 1) {                                          // phase probe: the caller sized ws with 8 x 8 bytes per workgroup behind the slabs
        ANODDPM_REQUIRE(a->ws_floats >= (int64_t)pg * slab + (int64_t)grid.x * grid.y * 16, "wgrad (Winograd): probe needs 16 floats per workgroup behind the slabs");
        hipLaunchKernelGGL(wgrad43_kernel<true>, grid, dim3(G4_NT), 0, s, *a, pg, a->W / 16, a->H / 8, 1.0f / (float)(a->W / 16));
    } else
#endif
    hipLaunchKernelGGL(wgrad43_kernel<false>, grid, dim3(G4_NT), 0, s, *a, pg, a->W / 16, a->H / 8, 1.0f / (float)(a->W / 16));
    if (g_debug[8] != 1) {                                           // ANODDPM_DEBUG8=1: the two-launch fold
        hipLaunchKernelGGL(wgrad43_fold_kernel, dim3((unsigned)((int64_t)(K / 2) * (a->N / 64) + (a->dimg ? a->N / 32 : 0))), dim3(288), 0, s, *a, pg, 0);
        return check_launch("conv3x3_wgrad (Winograd)");
    }
    if (pg > 1) hipLaunchKernelGGL(wgrad43_sum_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, s, a->ws, slab, pg);
    hipLaunchKernelGGL(wgrad43_out_kernel, dim3((unsigned)(((int64_t)K * a->N + 255) / 256)), dim3(256), 0, s, *a);
    if (a->dimg) hipLaunchKernelGGL(wgrad43_fold_kernel, dim3((unsigned)(a->N / 32)), dim3(288), 0, s, *a, pg, (K / 2) * (a->N / 64));   // column sums only
    return check_launch("conv3x3_wgrad (Winograd)");
}

}  // namespace anoddpm
