// Winograd F(4x4,3x3) with SPLIT-bf16 products (cfg 7 of anoddpm_igemm) -- an OPT-IN side configuration, never the default.
//
// Same layer, same fusions and the same transforms as winograd43r.hip (nn.Conv2d 3x3 of UNet.py:172,193 with GroupNorm-apply +
// SiLU, nearest-x2 and the two-source concat on the operand load; bias / time-embedding / residual / GroupNorm statistics in the
// epilogue; transforms B^T d B, G g G^T, A^T M A in fp32 / fp64 exactly as there).  What differs is the arithmetic of the
// position-wise products M[pos] = V[pos] U[pos]:
//
//   winograd43r.hip   fp32 x fp32 on v_mfma_f32_16x16x4_f32 (the reference's own arithmetic class; 64 flop / clk / SIMD)
//   this kernel       every fp32 operand is split into THREE bf16 pieces  x = hi + mid + lo  (hi = bf16(x), mid = bf16(x - hi),
//                     lo = bf16(x - hi - mid): 24 significant bits, the split is exact to 2^-24 |x|), and the six products of
//                     total order <= 2 --  hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi  -- are accumulated in fp32 on
//                     v_mfma_f32_16x16x32_bf16 (1024 flop / clk / SIMD): 6/16 of the fp32 matrix time.  The dropped products
//                     (mid*lo, lo*mid, lo*lo) are below 2^-24 of |x||y|.
//
// This is NOT the reference's arithmetic (products are formed from rounded pieces), which is why it is a side line: bench.py
// --arith bf16split3 / ANODDPM_ARITH=bf16split3 select it, the error table against fp64 is profiles/r4_bf16split3_errors.csv.
//
// Structure (first version, measured as such): K advances 32 channels per iteration (the K of one bf16 MFMA); the patch and the
// three-piece V live in LDS single-buffered (46 + 108 KB), so an iteration is  stage | barrier | transform + split | barrier |
// 36 x 6 MFMAs per wave | barrier; the raw patch of the next iteration is requested before the MFMA phase.  Weights: three bf16
// planes [piece][pos][K/8][N][8] (pack kind 6), one 16-byte load per piece and position per lane through a register ring.
#include "common.h"

using anoddpm::silu_f;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int B4_NT = 512;                 // 8 waves: wave w = all 36 positions x channels 16 w .. 16 w + 15 x 16 tiles
constexpr int B4_KC = 32;                  // channels per K iteration
constexpr int B4_PW = 18;
constexpr int B4_PPIX = B4_PW * B4_PW;     // 324 patch pixels
constexpr int B4_PITCH = 9;                // float4 per patch pixel: 8 quads + 1 pad
constexpr int B4_PJ = 6;                   // staging slots per thread: 6 * 512 = 3072 >= 324 * 8
constexpr int B4_DT = B4_PPIX * B4_PITCH;              // float4 of the patch buffer (exactly the patch: LDS is full)
constexpr int B4_VPIECE = 36 * 16 * 64;                // bytes of one piece of V: [pos][tile][32 bf16]
constexpr int B4_LDS_BYTES = B4_DT * 16 + 3 * B4_VPIECE;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t b4_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7ffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 b4_bld4(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned wave_bytes)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)wave_bytes, 0));
}

// x -> {hi, mid, lo} bf16 pieces (round to nearest even each), two values at a time: returns the packed pairs
__device__ __forceinline__ void split3(f32x2 x, unsigned &hi, unsigned &mid, unsigned &lo)
{
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 h = {(__bf16)x[0], (__bf16)x[1]};
    const f32x2 r1 = {x[0] - (float)h[0], x[1] - (float)h[1]};
    const bf16x2 m = {(__bf16)r1[0], (__bf16)r1[1]};
    const f32x2 r2 = {r1[0] - (float)m[0], r1[1] - (float)m[1]};
    const bf16x2 l = {(__bf16)r2[0], (__bf16)r2[1]};
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
    lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ void b4_at6(const float (&m)[6], float (&o)[4])
{
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    o[0] = m[0] + s12 + s34;
    o[1] = d12 + 2.f * d34;
    o[2] = s12 + 4.f * s34;
    o[3] = d12 + 8.f * d34 + m[5];
}

__global__ __launch_bounds__(B4_NT, 1) void wino43b_kernel(const anoddpm_igemm_args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    f32x4 *ldsD = reinterpret_cast<f32x4 *>(ldsb);
    unsigned char *ldsV = ldsb + B4_DT * 16;                        // [3 pieces][36 pos][16 tiles][32 bf16]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    const int K = a.c0 + a.c1, N = a.N, K8 = K >> 3;
    const int tiles_x = W >> 4;
    const int y0 = (blockIdx.x / tiles_x) * 16, x0 = (blockIdx.x % tiles_x) * 16;
    const int n0 = blockIdx.y * 128;
    const int b = blockIdx.z;
    const int a_mode = a.a_mode;
    const float *A0 = a.a0 + (int64_t)b * a.a0_bs;
    const float *A1 = a.a1 ? a.a1 + (int64_t)b * a.a1_bs : nullptr;
    const float *gsc = a.gn_scale ? a.gn_scale + (int64_t)b * a.gn_ld : nullptr;
    const float *gsh = a.gn_shift ? a.gn_shift + (int64_t)b * a.gn_ld : nullptr;
    const bool affine = gsc != nullptr, act = a.act != 0;
    const int nchunks = K / B4_KC;

    // ---- patch staging (pixel = idx >> 3, quad = idx & 7)
    int spix[B4_PJ];
    const int pq = tid & 7;
#pragma unroll
    for (int j = 0; j < B4_PJ; ++j) {
        const int p = (tid + j * B4_NT) >> 3;
        const int py = p / B4_PW, px = p - py * B4_PW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        int sp = -1;
        if (p < B4_PPIX && gy >= 0 && gy < H && gx >= 0 && gx < W)
            sp = (a_mode == 0) ? gy * W + gx : (gy >> 1) * (W >> 1) + (gx >> 1);
        spix[j] = sp;
    }
    f32x4 praw[B4_PJ];
    const __amdgpu_buffer_rsrc_t rA0 = b4_rsrc(A0), rA1 = b4_rsrc(A1 ? A1 : A0);
    const __amdgpu_buffer_rsrc_t rSc = b4_rsrc(gsc ? gsc : A0), rSh = b4_rsrc(gsh ? gsh : A0);
    auto load_patch = [&](int chunk) {                              // unconditional loads, clamped addresses
        if (chunk >= nchunks) chunk = nchunks - 1;
        const int kbase = chunk * B4_KC;
        const bool first = kbase < a.c0;                             // c0 % 32 == 0: a chunk does not straddle the two sources
        const __amdgpu_buffer_rsrc_t r = first ? rA0 : rA1;
        const unsigned ld = (unsigned)(first ? a.a0_ld : a.a1_ld);
        const unsigned koff = (unsigned)(first ? kbase : kbase - a.c0) * 4u;
#pragma unroll
        for (int j = 0; j < B4_PJ; ++j) {
            const unsigned sp = spix[j] >= 0 ? (unsigned)spix[j] : 0u;
            praw[j] = b4_bld4(r, (sp * ld + (unsigned)(pq * 4)) * 4u, koff);
        }
    };
    auto store_patch = [&](int chunk) {                             // GroupNorm-apply + SiLU, zero padding AFTER it
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
        if (affine) {
            asc = b4_bld4(rSc, (unsigned)(pq * 16), (unsigned)(chunk * B4_KC) * 4u);
            ash = b4_bld4(rSh, (unsigned)(pq * 16), (unsigned)(chunk * B4_KC) * 4u);
        }
#pragma unroll
        for (int j = 0; j < B4_PJ; ++j) {
            const int idx = tid + j * B4_NT;
            f32x4 v = praw[j];
            if (affine) v = v * asc + ash;
            if (act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
            if ((idx >> 3) < B4_PPIX) ldsD[(idx >> 3) * B4_PITCH + (idx & 7)] = spix[j] >= 0 ? v : zero;
        }
    };

    // ---- input transform + split: 1536 items = 16 tiles x 16 channel pairs x 6 transform rows = 24 virtual waves of 64 items,
    // three per physical wave.  Virtual wave v: channel half v / 12, row u = (v % 12) % 6, tile half (v % 12) / 6;
    // lane: channel pair = lane & 7 (of the half), tile = (tile half) * 8 + (lane >> 3).
    auto transform = [&](int vw) {                                  // vw is wave-uniform
        const int half = vw / 12, w12 = vw - half * 12;
        const int tu = w12 % 6;
        const int tpair = (lane & 7) + 8 * half;                    // channel pair 0..15 of the chunk
        const int ttile = (w12 / 6) * 8 + (lane >> 3);
        // B^T row u as (patch row, coefficient) pairs:
        //   u0: 4 d0 - 5 d2 + d4        u1: -4 d1 - 4 d2 + d3 + d4     u2: 4 d1 - 4 d2 - d3 + d4
        //   u3: -2 d1 - d2 + 2 d3 + d4  u4: 2 d1 - d2 - 2 d3 + d4      u5: 4 d1 - 5 d3 + d5
        const int tr0 = (tu == 0) ? 0 : 1, tr1 = (tu == 5) ? 3 : 2, tr2 = (tu == 0) ? 4 : ((tu == 5) ? 5 : 3), tr3 = 4;
        const float tc0 = (tu == 0) ? 4.f : (tu == 1 ? -4.f : (tu == 2 ? 4.f : (tu == 3 ? -2.f : (tu == 4 ? 2.f : 4.f))));
        const float tc1 = (tu == 0 || tu == 5) ? -5.f : ((tu == 1 || tu == 2) ? -4.f : -1.f);
        const float tc2 = (tu == 0 || tu == 5) ? 1.f : (tu == 1 ? 1.f : (tu == 2 ? -1.f : (tu == 3 ? 2.f : -2.f)));
        const float tc3 = (tu == 0 || tu == 5) ? 0.f : 1.f;
        const int rp = B4_PW * B4_PITCH * 2;                         // float2 per patch row
        const f32x2 *D = reinterpret_cast<const f32x2 *>(ldsD) + ((4 * (ttile >> 2)) * B4_PW + 4 * (ttile & 3)) * B4_PITCH * 2 + tpair;
        f32x2 t[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            t[j] = tc0 * D[tr0 * rp + j * B4_PITCH * 2] + tc1 * D[tr1 * rp + j * B4_PITCH * 2] + tc2 * D[tr2 * rp + j * B4_PITCH * 2] +
                   tc3 * D[tr3 * rp + j * B4_PITCH * 2];
        const f32x2 p = t[4] - 4.f * t[2], q = t[3] - 4.f * t[1], r = t[4] - t[2], s = t[3] - t[1];
        f32x2 v[6];
        v[0] = 4.f * t[0] - 5.f * t[2] + t[4];
        v[1] = p + q;
        v[2] = p - q;
        v[3] = r + 2.f * s;
        v[4] = r - 2.f * s;
        v[5] = 4.f * t[1] - 5.f * t[3] + t[5];
        // V[piece][pos = 6 u + v][tile][32 bf16]: this item's two channels are one 4-byte slot
        unsigned char *dst = ldsV + ((tu * 6) * 16 + ttile) * 64 + tpair * 4;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            unsigned hi, mid, lo;
            split3(v[c], hi, mid, lo);
            *reinterpret_cast<unsigned *>(dst + c * 16 * 64) = hi;
            *reinterpret_cast<unsigned *>(dst + c * 16 * 64 + B4_VPIECE) = mid;
            *reinterpret_cast<unsigned *>(dst + c * 16 * 64 + 2 * B4_VPIECE) = lo;
        }
    };

    // ---- accumulators: all 36 positions x this wave's 16 channels x 16 tiles
    f32x4 acc[36];
#pragma unroll
    for (int p = 0; p < 36; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, kq = lane >> 4;
    const int nw = n0 + wave * 16 + l15;                            // this lane's output channel
    const __amdgpu_buffer_rsrc_t rU = b4_rsrc(a.bmat);
    const unsigned piece_bytes = 36u * (unsigned)K8 * (unsigned)N * 16u;
    const unsigned pos_bytes = (unsigned)K8 * (unsigned)N * 16u;
    const unsigned ulane = ((unsigned)kq * (unsigned)N + (unsigned)nw) * 16u;        // + (chunk * 4) * N * 16 + pos * pos_bytes + piece * piece_bytes
    constexpr int RING = 3;
    f32x4 ring[RING][3];
    auto load_b = [&](int chunk, int pos, int slot) {
        if (chunk >= nchunks) chunk = nchunks - 1;
        const unsigned o = (unsigned)pos * pos_bytes + (unsigned)(chunk * 4) * (unsigned)N * 16u;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) ring[slot][pc] = b4_bld4(rU, ulane, o + (unsigned)pc * piece_bytes);
    };
    const unsigned char *vA = ldsV + l15 * 64 + kq * 16;            // + pos * 1024 + piece * B4_VPIECE

    load_patch(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        store_patch(chunk);
        __syncthreads();                                            // patch(chunk) complete; V(chunk - 1) retired by the barrier below
        transform(wave);
        __builtin_amdgcn_sched_barrier(0);                          // one item at a time: 24 patch values + 6 outputs in registers beside
        transform(wave + 8);                                        // the 144 accumulators
        __builtin_amdgcn_sched_barrier(0);
        transform(wave + 16);
        __builtin_amdgcn_sched_barrier(0);
        load_patch(chunk + 1);                                      // lands during the MFMA phase
#pragma unroll
        for (int g = 0; g < RING; ++g) load_b(chunk, g, g);         // the ring is primed per iteration: it does not live through the
        __syncthreads();                                            // staging / transform phases (register budget).  V(chunk) complete
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(vA + p * 1024);
            const bf16x8 am = *reinterpret_cast<const bf16x8 *>(vA + p * 1024 + B4_VPIECE);
            const bf16x8 al = *reinterpret_cast<const bf16x8 *>(vA + p * 1024 + 2 * B4_VPIECE);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, ring[p % RING][0]);
            const bf16x8 bm = __builtin_bit_cast(bf16x8, ring[p % RING][1]);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, ring[p % RING][2]);
            f32x4 c = acc[p];                                       // smallest products first
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
            acc[p] = c;
            if (p + RING < 36) load_b(chunk, p + RING, p % RING);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                            // every wave is done with V(chunk) and the patch buffer
    }

    // ---- epilogue, in registers (as winograd43r.hip): lane = (channel nw, tiles kq*4 .. kq*4+3); tile r sits in component r
    const float *TE = a.temb ? a.temb + (int64_t)b * a.temb_ld : nullptr;
    const __amdgpu_buffer_rsrc_t rO = b4_rsrc(a.out + (int64_t)b * a.o_bs);
    const __amdgpu_buffer_rsrc_t rR = b4_rsrc(a.res ? a.res + (int64_t)b * a.r_bs : a.out);
    const bool has_res = a.res != nullptr;
    const unsigned uW = (unsigned)W, o_ld = (unsigned)a.out_ld, r_ld = (unsigned)a.res_ld;
    float add = 0.f;
    if (a.bias) add += a.bias[nw];
    if (TE) add += TE[nw];
    const unsigned pix0 = (unsigned)(y0 + kq * 4) * uW + (unsigned)x0;
    const unsigned vo = (pix0 * o_ld + (unsigned)nw) * 4u, vr = (pix0 * r_ld + (unsigned)nw) * 4u;
    float cs = 0.f, cq = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float rv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) rv[i] = 0.f;
        if (has_res) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rv[i * 4 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rR, (int)vr, (int)((((unsigned)(r * 4) + (unsigned)i * uW + (unsigned)j) * 4u) * r_ld), 0));
        }
        float y[4][6];
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            float mu[6], o[4];
#pragma unroll
            for (int u = 0; u < 6; ++u) mu[u] = acc[u * 6 + v][r];
            b4_at6(mu, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i][v] = o[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float o4[4];
            b4_at6(y[i], o4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned so = ((unsigned)(r * 4) + (unsigned)i * uW + (unsigned)j) * 4u;
                const float v = a.alpha * o4[j] + add + rv[i * 4 + j];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rO, (int)vo, (int)(so * o_ld), 0);
                cs += v;
                cq += v * v;
            }
        }
    }
    if (a.stats) {
        cs += __shfl_xor(cs, 16);
        cq += __shfl_xor(cq, 16);
        cs += __shfl_xor(cs, 32);
        cq += __shfl_xor(cq, 32);
        if (kq == 0) {
            float *st = a.stats + (((int64_t)b * gridDim.x + blockIdx.x) * N + nw) * 2;
            st[0] = cs;
            st[1] = cq;
        }
    }
}

// OIHW 3x3 -> the three bf16 planes of U = G g G^T (fp64 transform, rounded once to fp32, then split): [piece][36][K/8][N][8].
// idx = one (o, input-channel quad): the quad is half of an 8-channel slot.
__global__ __launch_bounds__(256) void pack_wino43b_kernel(const float *__restrict__ w, unsigned short *__restrict__ out, int N, int K)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int K4 = K >> 2;
    if (idx >= (int64_t)N * K4) return;
    const int o = (int)(idx % N), i4 = (int)(idx / N);
    double g[4][3][3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float *src = w + ((int64_t)o * K + i4 * 4 + e) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) g[e][t / 3][t % 3] = (double)src[t];
    }
    const double G6[6][3] = {{1.0 / 4, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const int64_t plane = (int64_t)36 * (K >> 3) * N * 8;           // bf16 elements per piece
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        double t1[4][3];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int b2 = 0; b2 < 3; ++b2) t1[e][b2] = G6[u][0] * g[e][0][b2] + G6[u][1] * g[e][1][b2] + G6[u][2] * g[e][2][b2];
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) f[e] = (float)(t1[e][0] * G6[v][0] + t1[e][1] * G6[v][1] + t1[e][2] * G6[v][2]);
            unsigned h01, m01, l01, h23, m23, l23;
            split3(f32x2{f[0], f[1]}, h01, m01, l01);
            split3(f32x2{f[2], f[3]}, h23, m23, l23);
            const int64_t at = ((((int64_t)(u * 6 + v) * (K >> 3) + (i4 >> 1)) * N + o) * 8 + (i4 & 1) * 4);
            *reinterpret_cast<uint2 *>(out + at) = make_uint2(h01, h23);
            *reinterpret_cast<uint2 *>(out + plane + at) = make_uint2(m01, m23);
            *reinterpret_cast<uint2 *>(out + 2 * plane + at) = make_uint2(l01, l23);
        }
    }
}

}  // namespace

namespace anoddpm {

bool wino43b_ok(const anoddpm_igemm_args *a)
{
    const int K = a->c0 + a->c1;
    return a->ks == 3 && a->b_mode == 0 && a->heads == 1 && a->ksplit == 1 && (a->a_mode == 0 || a->a_mode == 1) && a->H % 16 == 0 &&
           a->W % 16 == 0 && K % B4_KC == 0 && (a->c1 == 0 || a->c0 % B4_KC == 0) && a->N % 128 == 0;
}

int launch_winograd43b(const anoddpm_igemm_args *a, hipStream_t s)
{
    ANODDPM_REQUIRE(wino43b_ok(a), "winograd43b: needs ks 3, a_mode 0 / 1, H, W %% 16 == 0, K %% 32 == 0 (each source), N %% 128 == 0, ksplit 1");
    const int K = a->c0 + a->c1;
    ANODDPM_REQUIRE((int64_t)3 * 36 * K * a->N * 2 < ((int64_t)1 << 31), "winograd43b: weight planes exceed 32-bit buffer offsets");
    ANODDPM_REQUIRE((int64_t)a->H * a->W * (a->a0_ld > a->a1_ld ? a->a0_ld : a->a1_ld) * 4 < ((int64_t)1 << 31), "winograd43b: operand slice exceeds 32-bit buffer offsets");
    ANODDPM_REQUIRE(!a->tail_csum && !a->fold_gamma, "winograd43b: no split-K tail, no GroupNorm fold");
    ANODDPM_REQUIRE(a->B <= 65535, "winograd43b: batch too large");
    static bool attr_done[ANODDPM_MAX_DEV];
    if (int rc = allow_big_lds(reinterpret_cast<const void *>(&wino43b_kernel), attr_done, "igemm(winograd43b)")) return rc;
    dim3 grid((unsigned)((a->H / 16) * (a->W / 16)), (unsigned)(a->N / 128), (unsigned)a->B);
    hipLaunchKernelGGL(wino43b_kernel, grid, dim3(B4_NT), B4_LDS_BYTES, s, *a);
    return check_launch("igemm(winograd43b)");
}

}  // namespace anoddpm

extern "C" int anoddpm_pack_wino43_bf16x3(const float *w, void *out, int32_t N, int32_t K, void *stream)
{
    ANODDPM_REQUIRE(w && out && N >= 1 && K >= 8 && K % 8 == 0, "pack_wino43_bf16x3: bad arguments (K %% 8 == 0)");
    const int64_t items = (int64_t)N * (K / 4);
    hipLaunchKernelGGL(pack_wino43b_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, anoddpm::as_stream(stream), w,
                       reinterpret_cast<unsigned short *>(out), N, K);
    return anoddpm::check_launch("pack_wino43_bf16x3");
}
