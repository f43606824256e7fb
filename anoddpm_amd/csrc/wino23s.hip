// Winograd F(2x2,3x3) on the 16x16 / 32x32 maps WITHOUT split-K (cfg 6 of anoddpm_igemm), gfx950.
//
// Replaces, on those maps, what winograd.hip does in two or three launches: the 3x3 convolutions of the ResBlocks
// (UNet.py:172,193) with GroupNorm32-apply + SiLU (UNet.py:170-171,190-191), the nearest-x2 of an up block (UNet.py:89,206),
// the virtual torch.cat([h, skip]) (UNet.py:402), bias / timestep-embedding / residual adds, the GroupNorm partial sums of the
// output -- and, as smallmap.hip, the GroupNorm FINALIZE of the operand in the prologue (`fold_*`, UNet.py:409-411).
//
// Why.  winograd.hip's workgroup is 32 tiles x 128 channels: a 16x16 map of a batch of four is 8 such patches, so it splits K
// over eight workgroups per patch to fill 256 CUs and needs a tail launch to fold the slabs (plus a GroupNorm finalize for its
// consumer on the 32x32 maps): 26 layers x (28-80 us + 7-11 us + 5 us) per configuration-2 step.  Here a workgroup owns
// 16 tiles (8x8 output pixels of one image) x 32 or 64 channels over ALL of K: 256 workgroups without splitting anything.
// Its eight waves split the 16 transform POSITIONS (two each): a wave's accumulators are complete sums, the output transform
// Y = A^T M A gathers the positions through LDS once, after the K loop.
//
// Arithmetic: as winograd.hip -- U = G g G^T is packed in fp64 and rounded once ([16][K/4][N][4], pack kind 1), V = B^T d B and
// Y = A^T M A only add and subtract, products and sums on v_mfma_f32_16x16x4_f32 (fp32 in / fp32 accumulate).
// Per 32-channel K chunk (one barrier):  transform patch(c+1) -> V(c+1)   |   32 or 64 MFMAs per wave on V(c)   |   activate and
// store the raw patch (c+2) (GroupNorm-apply + SiLU, nearest-x2 / concat resolved, zero padding after the transform)   |   request
// patch(c+3).  The B operand streams from L2 through a register ring (buffer loads, scalar offsets), two steps ahead.
#include "common.h"

using anoddpm::silu_f;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WS_NT = 512;
constexpr int WS_KCH = 32;                 // channels per K chunk
constexpr int WS_PITCH = WS_KCH + 4;       // floats between pixels (patch) / tiles (V) in LDS: 9 16-byte slots, odd
constexpr int WS_PPX = 100;                // 10 x 10 input pixels: the 8 x 8 outputs of a workgroup + halo
constexpr int WS_VBUF = 16 * 16 * WS_PITCH;    // floats per V buffer [pos][tile][PITCH]
constexpr int WS_PBUF = 104 * WS_PITCH;        // floats per patch buffer (100 pixels + slack for the unconditional last slot)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(const float *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7ffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 ws_bld4(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned wave_bytes)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)wave_bytes, 0));
}

template <int CT>                           // 16-channel column tiles per workgroup: 2 (32 channels) or 4 (64)
__global__ __launch_bounds__(WS_NT) void wino23s_kernel(const anoddpm_igemm_args a)
{
    constexpr int TN = 16 * CT;
    constexpr int MP = TN + 4;                                  // channel pitch of the epilogue's M[pos][tile][.] rows
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, q = lane >> 4;
    const int K = a.c0 + a.c1, K4 = K >> 2, N = a.N;
    const int H = a.H, W = a.W, P = H * W;
    const int nchunks = K / WS_KCH;
    float *aff_sc = lds, *aff_sh = lds + K;
    float *Vb = lds + 2 * K;                                    // [2][16][16][PITCH]
    float *Pb = Vb + 2 * WS_VBUF;                               // [2][104][PITCH]

    const int n0 = blockIdx.x * TN;
    const int bxn = W >> 3, byn = H >> 3;
    const int blk = blockIdx.y;
    const int b = blk / (bxn * byn), bin = blk - b * (bxn * byn);
    const int oy0 = (bin / bxn) * 8, ox0 = (bin % bxn) * 8;
    const bool up = a.a_mode == 1;                              // nearest x2 fused: the sources are (H/2) x (W/2)
    const int Ws = up ? W >> 1 : W;
    const float *A0 = a.a0 + (int64_t)b * a.a0_bs;
    const float *A1 = a.a1 ? a.a1 + (int64_t)b * a.a1_bs : A0;

    // ---- B stream: positions 2 wave, 2 wave + 1; step (chunk, i): 16 k; per-lane byte offset + scalar offset ----
    const __amdgpu_buffer_rsrc_t rU = ws_rsrc(a.bmat);
    const unsigned ulane = ((unsigned)q * (unsigned)N + (unsigned)(n0 + l16)) * 16u;
    const unsigned pos_bytes = (unsigned)K4 * (unsigned)N * 16u;
    auto b_off = [&](int chunk, int i, int pp) -> unsigned {     // i, pp compile-time
        if (chunk >= nchunks) chunk = nchunks - 1;              // past the end: re-read the last chunk (unused)
        return (unsigned)(2 * wave + pp) * pos_bytes + (unsigned)((chunk * WS_KCH + 16 * i) >> 2) * (unsigned)N * 16u;
    };
    f32x4 ring[2][2][CT];                                       // [step i][position pp][column tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const unsigned o = b_off(0, i, pp);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) ring[i][pp][ct] = ws_bld4(rU, ulane, o + ct * 256u);
        }

    // ---- patch staging: item j of this thread = (halo pixel hp, channel quad kq); 100 x 8 items over 512 threads ----
    const int kq = tid & 7;
    int soff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int hp = (tid >> 3) + 64 * j;
        int sp = -1;
        if (hp < WS_PPX) {
            const int hy = hp / 10, hx = hp - hy * 10;
            const int gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) sp = up ? (gy >> 1) * Ws + (gx >> 1) : gy * W + gx;
        }
        soff[j] = sp;
    }
    f32x4 praw[2];
    unsigned pvalid = 0;
    auto load_patch = [&](int chunk) {
        if (chunk >= nchunks) chunk = nchunks - 1;
        const int k = chunk * WS_KCH + 4 * kq;
        const bool first = k < a.c0;
        const float *src = first ? A0 + k : A1 + (k - a.c0);
        const int ld = first ? a.a0_ld : a.a1_ld;
        pvalid = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            praw[j] = *reinterpret_cast<const f32x4 *>(src + (int64_t)(soff[j] >= 0 ? soff[j] : 0) * ld);   // unconditional, clamped
            pvalid |= (soff[j] >= 0 ? 1u : 0u) << j;
        }
    };
    const bool affine = a.gn_scale != nullptr || a.fold_gamma != nullptr;
    const bool act = a.act != 0;
    auto store_patch = [&](int chunk) {
        if (chunk >= nchunks) chunk = nchunks - 1;
        float *dst = Pb + (chunk & 1) * WS_PBUF;
        const int k = chunk * WS_KCH + 4 * kq;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (affine) { sc = *reinterpret_cast<const f32x4 *>(aff_sc + k); sh = *reinterpret_cast<const f32x4 *>(aff_sh + k); }
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int hp = (tid >> 3) + 64 * j;                 // < 128; slots 100..103 exist, beyond them nothing is stored
            f32x4 v = praw[j];
            if (affine) v = v * sc + sh;
            if (act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
            if (hp < 104) *reinterpret_cast<f32x4 *>(dst + hp * WS_PITCH + 4 * kq) = ((pvalid >> j) & 1) ? v : zero;   // zero padding AFTER the transform
        }
    };
    // ---- input transform: thread = (tile, channel quad, row u) -> positions (u, 0..3) of V = B^T d B ----
    const int t_tile = tid >> 5, t_q = (tid >> 2) & 7, t_u = tid & 3;
    const int t_py = 2 * (t_tile >> 2), t_px = 2 * (t_tile & 3);
    // row u of B^T d: rows (r0, r1) of the 4x4 patch, combined as r0 - r1 (u = 0, 3) / r0 + r1 (u = 1) / r1 - r0 (u = 2)
    const int t_r0 = t_u == 0 ? 0 : 1, t_r1 = t_u == 3 ? 3 : 2;
    const float t_s0 = t_u == 2 ? -1.f : 1.f, t_s1 = t_u == 1 ? 1.f : (t_u == 2 ? 1.f : -1.f);
    auto transform = [&](int buf) {
        const float *src = Pb + buf * WS_PBUF + 4 * t_q;
        const float *p0 = src + ((t_py + t_r0) * 10 + t_px) * WS_PITCH, *p1 = src + ((t_py + t_r1) * 10 + t_px) * WS_PITCH;
        f32x4 c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c[j] = *reinterpret_cast<const f32x4 *>(p0 + j * WS_PITCH) * t_s0 + *reinterpret_cast<const f32x4 *>(p1 + j * WS_PITCH) * t_s1;
        float *dst = Vb + buf * WS_VBUF + ((t_u * 4) * 16 + t_tile) * WS_PITCH + 4 * t_q;
        *reinterpret_cast<f32x4 *>(dst) = c[0] - c[2];
        *reinterpret_cast<f32x4 *>(dst + 16 * WS_PITCH) = c[1] + c[2];
        *reinterpret_cast<f32x4 *>(dst + 32 * WS_PITCH) = c[2] - c[1];
        *reinterpret_cast<f32x4 *>(dst + 48 * WS_PITCH) = c[1] - c[3];
    };

    load_patch(0);
    // ---- GroupNorm affine of the operand: given, or finished here from the producers' statistics (as smallmap.hip) ----
    if (a.fold_gamma) {
        double *csum = reinterpret_cast<double *>(Vb);           // scratch in the still unused V buffers: K <= 1024 -> 16 KB + 1 KB
        double *gst = csum + 2 * K;
        const int groups = a.fold_groups, cpg = K / groups;
        for (int c = tid; c < K; c += WS_NT) {
            const bool first = c < a.c0;
            const int cl = first ? c : c - a.c0, cw = first ? a.c0 : a.c1;
            const int fmt = first ? a.fold_fmt0 : a.fold_fmt1, rows = first ? a.fold_rows0 : a.fold_rows1;
            const float *st = first ? a.fold_stats0 : a.fold_stats1;
            double s = 0.0, qq = 0.0;
            if (fmt) {
                const double *sd = reinterpret_cast<const double *>(st) + ((int64_t)b * cw + cl) * 2;
                s = sd[0]; qq = sd[1];
            } else {
                const float *sr = st + ((int64_t)b * rows * cw + cl) * 2;
                for (int r0 = 0; r0 < rows; r0 += 4) {
                    float2 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = r0 + u < rows ? r0 + u : rows - 1;
                        v[u] = *reinterpret_cast<const float2 *>(sr + (int64_t)r * cw * 2);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (r0 + u < rows) { s += (double)v[u].x; qq += (double)v[u].y; }
                }
            }
            csum[2 * c] = s; csum[2 * c + 1] = qq;
        }
        __syncthreads();
        if (tid < groups) {
            double S = 0.0, Q = 0.0;
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { S += csum[2 * c]; Q += csum[2 * c + 1]; }
            const double n = (double)(up ? P / 4 : P) * cpg;    // the statistics are those of the (possibly half-resolution) source
            const double mean = S / n;
            double var = Q / n - mean * mean;
            var = var > 0.0 ? var : 0.0;
            gst[2 * tid] = mean;
            gst[2 * tid + 1] = 1.0 / sqrt(var + (double)a.fold_eps);
        }
        __syncthreads();
        for (int c = tid; c < K; c += WS_NT) {
            const int g = c / cpg;
            const double sc = gst[2 * g + 1] * (double)a.fold_gamma[c];
            aff_sc[c] = (float)sc;
            aff_sh[c] = (float)((double)a.fold_beta[c] - gst[2 * g] * sc);
        }
        __syncthreads();
    } else if (a.gn_scale) {
        for (int c = tid; c < K; c += WS_NT) {
            aff_sc[c] = a.gn_scale[(int64_t)b * a.gn_ld + c];
            aff_sh[c] = a.gn_shift[(int64_t)b * a.gn_ld + c];
        }
        __syncthreads();
    }
    // prologue: patch(0) -> LDS -> V(0); patch(1) -> LDS; patch(2) requested
    store_patch(0);
    load_patch(1);
    __syncthreads();
    transform(0);
    store_patch(1);
    load_patch(2);
    __syncthreads();

    f32x4 acc[2][CT];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[pp][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int vlane = l16 * WS_PITCH + 4 * q;                    // this lane's tile row and k quad inside a position's V rows

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk + 1 < nchunks) transform((chunk + 1) & 1);
        const float *V = Vb + (chunk & 1) * WS_VBUF + (2 * wave) * 16 * WS_PITCH + vlane;
        f32x4 av[2][2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) av[0][pp] = *reinterpret_cast<const f32x4 *>(V + pp * 16 * WS_PITCH);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 0) {
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) av[1][pp] = *reinterpret_cast<const f32x4 *>(V + pp * 16 * WS_PITCH + 16);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[pp][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][pp][e], ring[i][pp][ct][e], acc[pp][ct], 0, 0, 0);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const unsigned o = b_off(chunk + 1, i, pp);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) ring[i][pp][ct] = ws_bld4(rU, ulane, o + ct * 256u);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (chunk + 2 < nchunks) store_patch(chunk + 2);         // replaces patch(chunk): its transform ran an iteration ago
        if (chunk + 3 < nchunks) load_patch(chunk + 3);
        __syncthreads();
    }

    // ---- epilogue: M[pos][tile][n] through LDS, Y = A^T M A per (tile, channel quad), adds, stores, statistics ----
    float *M = Vb;                                              // 16 x 16 x MP floats <= the two V buffers
    // D layout of v_mfma_f32_16x16x4_f32: this lane holds tile rows 4 q + e (e = 0..3) of channel column l16
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) M[((2 * wave + pp) * 16 + 4 * q + e) * MP + ct * 16 + l16] = acc[pp][ct][e];
    __syncthreads();
    float *O = a.out + (int64_t)b * a.o_bs;
    const float *R = a.res ? a.res + (int64_t)b * a.r_bs : nullptr;
    const float *TE = a.temb ? a.temb + (int64_t)b * a.temb_ld : nullptr;
    float *SS = Pb;                                             // per-tile channel sums for the statistics: [16 tiles][TN][2]
    constexpr int ITEMS = 16 * (TN / 4);
    if (tid < ITEMS) {
        const int tile = tid / (TN / 4), c4 = tid - tile * (TN / 4);
        const float *m0 = M + tile * MP + c4 * 4;
        f32x4 mm[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) mm[p] = *reinterpret_cast<const f32x4 *>(m0 + p * 16 * MP);
        // t[a][v] = sum_u A^T[a][u] M[u][v],  A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]
        f32x4 t0[4], t1[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            t0[v] = (mm[v] + mm[4 + v]) + mm[8 + v];
            t1[v] = (mm[4 + v] - mm[8 + v]) - mm[12 + v];
        }
        f32x4 y[2][2];
        y[0][0] = (t0[0] + t0[1]) + t0[2];
        y[0][1] = (t0[1] - t0[2]) - t0[3];
        y[1][0] = (t1[0] + t1[1]) + t1[2];
        y[1][1] = (t1[1] - t1[2]) - t1[3];
        const int n = n0 + c4 * 4;
        f32x4 add = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) add += *reinterpret_cast<const f32x4 *>(a.bias + n);
        if (TE) add += *reinterpret_cast<const f32x4 *>(TE + n);
        const int py = oy0 + 2 * (tile >> 2), px = ox0 + 2 * (tile & 3);
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        f32x4 rv[2][2];
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
                rv[aa][bb] = R ? *reinterpret_cast<const f32x4 *>(R + (int64_t)((py + aa) * W + px + bb) * a.res_ld + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const f32x4 v = y[aa][bb] * a.alpha + add + rv[aa][bb];
                *reinterpret_cast<f32x4 *>(O + (int64_t)((py + aa) * W + px + bb) * a.out_ld + n) = v;
                s += v;
                s2 += v * v;
            }
        if (a.stats) {
            *reinterpret_cast<f32x4 *>(SS + (tile * TN + c4 * 4) * 2) = f32x4{s[0], s2[0], s[1], s2[1]};
            *reinterpret_cast<f32x4 *>(SS + (tile * TN + c4 * 4) * 2 + 4) = f32x4{s[2], s2[2], s[3], s2[3]};
        }
    }
    if (a.stats) {
        __syncthreads();
        if (tid < TN) {
            float s = 0.f, qq = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) { s += SS[(t * TN + tid) * 2]; qq += SS[(t * TN + tid) * 2 + 1]; }   // fixed order
            float *st = a.stats + (((int64_t)b * (bxn * byn) + bin) * N + n0 + tid) * 2;
            st[0] = s; st[1] = qq;
        }
    }
}

}  // namespace

namespace anoddpm {

// Column tiles (2 or 4) cfg 6 uses for this layer, or 0 when it does not take it.
int wino23s_tile(int H, int W, int K, int c0, int N, int B, int a_mode)
{
    if (H != W || (H != 16 && H != 32) || (a_mode != 0 && a_mode != 1)) return 0;
    if (K % WS_KCH || c0 % 4 || K > 1024 || K < 64 || N % 32) return 0;
    const int blocks = B * (H / 8) * (W / 8);
    // 64-channel workgroups (each activates and transforms its patch for twice the outputs) when they still fill the chip
    if (N % 64 == 0 && blocks * (N / 64) >= 200) return 4;
    if (blocks * (N / 32) >= 128) return 2;
    return 0;
}

int launch_wino23s(const anoddpm_igemm_args *a, hipStream_t s)
{
    const int K = a->c0 + a->c1;
    ANODDPM_REQUIRE(a->ks == 3 && a->b_mode == 0 && a->heads == 1 && a->ksplit == 1, "wino23s: needs ks 3, packed weights, heads 1, ksplit 1");
    const int ct = wino23s_tile(a->H, a->W, K, a->c0, a->N, a->B, a->a_mode);
    ANODDPM_REQUIRE(ct != 0, "wino23s: shape not supported (16x16 or 32x32 map, a_mode 0 / 1, K %% 32 == 0, 64 <= K <= 1024, N %% 32 == 0, >= 128 workgroups)");
    ANODDPM_REQUIRE(a->c1 == 0 || a->a1, "wino23s: dual source needs a1");
    ANODDPM_REQUIRE(a->out_ld % 4 == 0 && (!a->res || a->res_ld % 4 == 0) && a->a0_ld % 4 == 0 && (a->c1 == 0 || a->a1_ld % 4 == 0),
                    "wino23s: pixel strides must be multiples of 4 floats");
    ANODDPM_REQUIRE(!a->tail_csum, "wino23s: no split-K tail");
    ANODDPM_REQUIRE(!a->gn_scale || a->gn_shift, "wino23s: gn_scale without gn_shift");
    ANODDPM_REQUIRE((int64_t)16 * K * a->N * 4 < ((int64_t)1 << 31), "wino23s: transformed weights exceed 32-bit buffer offsets");
    if (a->fold_gamma) {
        ANODDPM_REQUIRE(a->fold_beta && a->fold_stats0 && (a->c1 == 0 || a->fold_stats1), "wino23s: GroupNorm fold: null pointer");
        ANODDPM_REQUIRE(a->fold_groups >= 1 && a->fold_groups <= 64 && K % a->fold_groups == 0, "wino23s: GroupNorm fold: bad group count");
        ANODDPM_REQUIRE((a->fold_fmt0 != 0 || a->fold_rows0 >= 1) && (a->c1 == 0 || a->fold_fmt1 != 0 || a->fold_rows1 >= 1),
                        "wino23s: GroupNorm fold: statistics rows missing");
    }
    const size_t lds = (size_t)(2 * K + 2 * WS_VBUF + 2 * WS_PBUF) * sizeof(float);
    ANODDPM_REQUIRE(lds <= 160 * 1024, "wino23s: LDS budget exceeded");
    dim3 grid((unsigned)(a->N / (16 * ct)), (unsigned)(a->B * (a->H / 8) * (a->W / 8)));
    static bool attr_done2[ANODDPM_MAX_DEV], attr_done4[ANODDPM_MAX_DEV];
    if (int rc = allow_big_lds(reinterpret_cast<const void *>(&wino23s_kernel<2>), attr_done2, "igemm(wino23s)")) return rc;
    if (int rc = allow_big_lds(reinterpret_cast<const void *>(&wino23s_kernel<4>), attr_done4, "igemm(wino23s)")) return rc;
    if (ct == 2) hipLaunchKernelGGL((wino23s_kernel<2>), grid, dim3(WS_NT), lds, s, *a);
    else         hipLaunchKernelGGL((wino23s_kernel<4>), grid, dim3(WS_NT), lds, s, *a);
    return check_launch("igemm(wino23s)");
}

}  // namespace anoddpm

extern "C" int anoddpm_wino23s_tile(int32_t H, int32_t W, int32_t K, int32_t c0, int32_t N, int32_t B, int32_t a_mode)
{
    return anoddpm::wino23s_tile(H, W, K, c0, N, B, a_mode);
}
