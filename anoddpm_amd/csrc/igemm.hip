// Implicit-GEMM convolution / GEMM for gfx950 on the fp32 matrix pipe.
//
// Replaces (reference file:line): nn.Conv2d 3x3 / 1x1 (UNet.py:172,193,200,387), nn.Conv1d k=1
// (UNet.py:115,117), the QK^T and AV einsums of QKVAttention (UNet.py:148-152) and, fused into
// the operand load, GroupNorm32-apply + SiLU (UNet.py:170-171,190-191,386), AvgPool2d /
// nearest-x2 on the h path (UNet.py:70,89,206) and torch.cat([h, skip], 1) (UNet.py:402).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- fp32 inputs, fp32 accumulate, bitwise an fmaf chain, so the
// result is plain fp32 (no reduced-precision inputs; the 1e-3 parity budget is not spent here).
// Roofline: compute (157.3 TFLOP/s fp32 matrix peak); AI of the 3x3 layers is ~280 flop/B.
//
// Tiling.  A workgroup (4 waves, 2x2) owns BM output pixels x BN output channels of ONE image;
// the pixels form a TH x TW patch so that all nine taps of a 3x3 filter read one LDS-resident
// halo tile ((TH+2) x (TW+2) pixels x 32 channels, staged once per 32-channel slice of K and
// transformed -- affine, SiLU, resample -- exactly once on the way in).  The weight tile for
// (tap, K-slice) is [8][BN][4] floats so that each lane's four consecutive k values are one
// ds_read_b128.  Pipeline: weight tiles are double-buffered in LDS and loaded two K-steps ahead
// (registers for one step, LDS for the next) so a K-step costs ONE barrier; the raw halo tile of the
// NEXT K-slice sits in registers for the nine taps of the current slice, so HBM/L2 latency of the
// activation stream is never on the critical path.  K order inside a 32-slice is permuted (lane half h takes k = 8j+4h..8j+4h+3) -- legal
// because A and B use the same permutation.
//
// LDS layout.  A: [pixel][8 quads] with the quad index XOR-ed by (pixel>>1)&7, which spreads the 16
// lanes of every ds_read_b128 service group (non-contiguous lane sets, MI355X_MICROARCH "LDS") over
// all 16 16-byte slots of the 256-byte bank row.  B: [k/4][n][4], lanes read consecutive n ->
// conflict-free by construction.
#include "common.h"

using anoddpm::silu_f;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 32;  // K slice held in LDS

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

template <int BM, int BN, bool CONV>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const anoddpm_igemm_args a, const int log2TW, const int TH, const int tiles_x)
{
    constexpr int MT = BM / 64, NT = BN / 64;          // 32x32 MFMA tiles per wave (2x2 waves)
    constexpr int APIX = (BM == 128) ? 204 : 136;      // largest halo tile in pixels
    constexpr int AJ = (APIX + 31) / 32;               // A float4 slots per thread per K-slice
    constexpr int BJ = 8 * BN / 256;                   // B float4 slots per thread per step
    constexpr int BTILE = 8 * BN * 4;                  // floats per weight tile
    __shared__ __attribute__((aligned(16))) float lds[APIX * KC + 2 * BTILE];
    f32x4 *ldsA = reinterpret_cast<f32x4 *>(lds);
    float *ldsBbase = lds + APIX * KC;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    const int KS = a.ks, pad = KS >> 1, taps = KS * KS;
    const int TW = 1 << log2TW;
    const int H = a.H, W = a.W;
    const int HP = TW + 2 * pad;
    const int npix = (TH + 2 * pad) * HP;
    const int K = a.c0 + a.c1;
    const int N = a.N;
    const int K4 = K >> 2;

    const int y0 = (blockIdx.x / tiles_x) * TH;
    const int x0 = (blockIdx.x % tiles_x) << log2TW;
    const int n0 = blockIdx.y * BN;
    const int ksplit = a.ksplit;
    const int ksi = blockIdx.z % ksplit;
    const int z = blockIdx.z / ksplit;
    const int b = z / a.heads, hd = z % a.heads;

    const float *A0 = a.a0 + (int64_t)b * a.a0_bs + (int64_t)hd * a.a0_hs;
    const float *A1 = a.a1 ? a.a1 + (int64_t)b * a.a1_bs + (int64_t)hd * a.a1_hs : nullptr;
    const float *Bm = a.b_mode ? a.bmat + (int64_t)b * a.b_bs + (int64_t)hd * a.b_hs : a.bmat;
    const float *gsc = a.gn_scale ? a.gn_scale + (int64_t)b * a.gn_ld : nullptr;
    const float *gsh = a.gn_shift ? a.gn_shift + (int64_t)b * a.gn_ld : nullptr;
    const bool affine = (gsc != nullptr);
    const bool act = a.act != 0;
    const int a_mode = a.a_mode;

    const int nchunks = (K + KC - 1) / KC;
    const int cps = (nchunks + ksplit - 1) / ksplit;
    const int c_begin = ksi * cps;
    const int c_end = (c_begin + cps < nchunks) ? c_begin + cps : nchunks;
    const int nsteps = (c_end > c_begin) ? (c_end - c_begin) * taps : 0;

    // lane's pixel rows inside the halo tile (tap 0,0)
    int pixbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = (wm * MT + mt) * 32 + l31;
        pixbase[mt] = (m >> log2TW) * HP + (m & (TW - 1));
    }
    int ncol[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ncol[nt] = (wn * NT + nt) * 32 + l31;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // ---- B tile: global -> registers -> LDS (double buffered) -------------------------------------
    // All prefetch loads are UNCONDITIONAL (indices clamped in-bounds, result selected to zero
    // afterwards): a predicated load makes hipcc branch around it and wait vmcnt(0) per element.
    // CONV (packed weights, K % 32 == 0): the per-thread element offset is fixed, only a uniform
    // (tap, K-slice) base moves -> one 32-bit add per load in the loop.
    f32x4 breg[BJ];
    int boff[BJ];            // CONV: element offset inside a [8][N][4] slab; else unused
    unsigned bok = 0;
    if (CONV) {
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int idx = tid + j * 256;
            const int r = idx / BN, n = idx % BN;
            const int nc = n0 + n < N ? n0 + n : N - 1;
            boff[j] = (r * N + nc) * 4;
            bok |= (n0 + n < N ? 1u : 0u) << j;
        }
    }
    auto load_B = [&](int chunk, int tap) {
        const int kbase = chunk * KC;
        if (CONV) {
            const float *base = Bm + ((int64_t)tap * K4 + (kbase >> 2)) * N * 4;     // uniform
#pragma unroll
            for (int j = 0; j < BJ; ++j) breg[j] = ld4(base + boff[j]);   // NO use of the value here: a select
            return;                                                       // would force vmcnt(0) right away
        }
        bok = 0;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int idx = tid + j * 256;
            f32x4 v;
            bool ok;
            if (a.b_mode == 0) {
                const int r = idx / BN, n = idx % BN;
                const int k4 = (kbase >> 2) + r;
                ok = (k4 < K4) && (n0 + n < N);
                const int k4c = k4 < K4 ? k4 : K4 - 1;
                const int nc = n0 + n < N ? n0 + n : N - 1;
                v = ld4(Bm + (((int64_t)tap * K4 + k4c) * N + nc) * 4);
            } else if (a.b_mode == 1) {            // rows [N][K]: B[k][n] = src[n][k]
                const int n = idx >> 3, r = idx & 7;
                const int k = kbase + r * 4;
                ok = (k < K) && (n0 + n < N);
                const int kc = k < K ? k : 0;
                const int nc = n0 + n < N ? n0 + n : N - 1;
                v = ld4(Bm + (int64_t)nc * a.ldb + kc);
            } else {                               // rows [K][N]
                const int k = idx / (BN / 4), n4 = idx % (BN / 4);
                ok = (kbase + k < K) && (n0 + n4 * 4 < N);
                const int kc = kbase + k < K ? kbase + k : 0;
                const int nc = n0 + n4 * 4 < N ? n0 + n4 * 4 : 0;
                v = ld4(Bm + (int64_t)kc * a.ldb + nc);
            }
            breg[j] = v;
            bok |= (ok ? 1u : 0u) << j;                  // applied when the tile is written to LDS
        }
    };
    auto store_B = [&](int buf) {
        float *ldsBf = ldsBbase + buf * BTILE;
        f32x4 *ldsB = reinterpret_cast<f32x4 *>(ldsBf);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int idx = tid + j * 256;
            const f32x4 bj = ((bok >> j) & 1) ? breg[j] : zero;     // out-of-range k / n -> 0
            if (CONV || a.b_mode == 0) {
                ldsB[idx] = bj;                    // idx = r*BN + n already
            } else if (a.b_mode == 1) {
                const int n = idx >> 3, r = idx & 7;
                ldsB[r * BN + n] = bj;
            } else {
                const int k = idx / (BN / 4), n4 = idx % (BN / 4);
                float *dst = ldsBf + (((k >> 2) * BN + n4 * 4) * 4 + (k & 3));
                dst[0] = bj[0];
                dst[4] = bj[1];
                dst[8] = bj[2];
                dst[12] = bj[3];
            }
        }
    };

    // ---- A halo tile: global -> registers (one K-slice ahead) -> (affine, SiLU, resample) -> LDS ----
    const int c4 = tid & 7;                 // this thread's channel quad inside the 32-channel slice
    f32x4 areg[AJ];
    f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
    unsigned avalid = 0;
    auto xform = [&](f32x4 v) {
        if (affine) v = v * asc + ash;
        if (act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
        return v;
    };
    // halo geometry is fixed for the block: source pixel index per slot (-1 = zero padding / unused)
    int spix[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int p = (tid >> 3) + j * 32;
        const int hy = p / HP, hx = p - hy * HP;
        const int gy = y0 + hy - pad, gx = x0 + hx - pad;
        int sp = -1;
        if (p < npix && gy >= 0 && gy < H && gx >= 0 && gx < W) {
            if (a_mode == 0)      sp = gy * W + gx;
            else if (a_mode == 1) sp = (gy >> 1) * (W >> 1) + (gx >> 1);
            else                  sp = (2 * gy) * (2 * W) + 2 * gx;
        }
        spix[j] = sp;
    }
    auto slice_src = [&](int chunk, const float *&src, int &ld, bool &kvalid) {
        const int kbase = chunk * KC;
        int koff;
        if (kbase < a.c0) { src = A0; ld = a.a0_ld; koff = kbase; }
        else              { src = A1; ld = a.a1_ld; koff = kbase - a.c0; }
        const int k = kbase + c4 * 4;
        kvalid = k < K;
        src += kvalid ? koff + c4 * 4 : 0;                   // K tail: stay in-bounds, value is discarded
        if (affine) { const int kc = kvalid ? k : 0; asc = ld4(gsc + kc); ash = ld4(gsh + kc); }
    };
    auto load_A = [&](int chunk) {          // a_mode 0 / 1: raw values into registers (unconditional loads)
        const float *src; int ld; bool kvalid;
        slice_src(chunk, src, ld, kvalid);
        avalid = 0;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const bool ok = kvalid && spix[j] >= 0;
            const int sp = spix[j] >= 0 ? spix[j] : 0;
            areg[j] = ld4(src + (int64_t)sp * ld);           // padding / tail slots read pixel 0 and are discarded
            avalid |= (ok ? 1u : 0u) << j;
        }
    };
    auto store_A = [&]() {                  // registers -> transform -> LDS (zero padding AFTER the transform)
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int p = (tid >> 3) + j * 32;
            if (p < npix) ldsA[p * 8 + (c4 ^ ((p >> 1) & 7))] = ((avalid >> j) & 1) ? xform(areg[j]) : zero;
        }
    };
    auto stage_A_pool = [&](int chunk) {    // a_mode 2: source is (2H, 2W); mean of the 4 transformed values
        const float *src; int ld; bool kvalid;
        slice_src(chunk, src, ld, kvalid);
        const int64_t w2 = (int64_t)W * 2;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int p = (tid >> 3) + j * 32;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kvalid && spix[j] >= 0) {
                const float *q = src + (int64_t)spix[j] * ld;
                const f32x4 r00 = ld4(q), r01 = ld4(q + ld), r10 = ld4(q + w2 * ld), r11 = ld4(q + (w2 + 1) * ld);
                v = (((xform(r00) + xform(r01)) + xform(r10)) + xform(r11)) * 0.25f;
            }
            if (p < npix) ldsA[p * 8 + (c4 ^ ((p >> 1) & 7))] = v;
        }
    };

    // ---- main loop: one barrier per K-step, loads one step (B) / one K-slice (A) ahead ---------------
    // (tap, slice) of the step being computed and of the B tile being fetched are carried as counters:
    // no runtime division in the loop.
    int ld_tap = 0, ld_chunk = c_begin;          // next B tile to fetch
    auto advance_ld = [&]() { if (++ld_tap == taps) { ld_tap = 0; ++ld_chunk; } };
    if (nsteps > 0) {
        load_B(ld_chunk, ld_tap); advance_ld();
        if (a_mode == 2) stage_A_pool(c_begin);
        else { load_A(c_begin); store_A(); }
        store_B(0);
        if (nsteps > 1) { load_B(ld_chunk, ld_tap); advance_ld(); }
        if (a_mode != 2 && c_begin + 1 < c_end) load_A(c_begin + 1);
        __syncthreads();
    }
    int tap = 0, chunk = c_begin, tapoff = 0, tx = 0;
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        const f32x4 *ldsB = reinterpret_cast<const f32x4 *>(ldsBbase + cur * BTILE);
        int pA[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) pA[mt] = pixbase[mt] + tapoff;
        // fragments are software-pipelined over the four k-groups of the slice: group g+1 is read from
        // LDS (into a second register set) before group g's sixteen MFMAs are issued
        f32x4 av[2][MT], bv[2][NT];
        auto read_frags = [&](int k8, int set) {
            const int q = k8 * 2 + h;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[set][mt] = ldsA[pA[mt] * 8 + (q ^ ((pA[mt] >> 1) & 7))];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[set][nt] = ldsB[q * BN + ncol[nt]];
        };
        read_frags(0, 0);
#pragma unroll
        for (int k8 = 0; k8 < KC / 8; ++k8) {
            const int set = k8 & 1;
            if (k8 + 1 < KC / 8) read_frags(k8 + 1, set ^ 1);
            // pin the reads ABOVE this group's MFMAs: left alone, the scheduler sinks them next to their
            // first use and every group then stalls for the LDS latency (~25 % of the step)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][mt][kk], bv[set][nt][kk], acc[mt][nt], 0, 0, 0);
            if (k8 == 0 && step + 1 < nsteps) {
                // staging of the following steps rides behind the first MFMA group (other LDS buffer)
                store_B(cur ^ 1);                        // tile for step+1 (its loads had a full step to land)
                if (step + 2 < nsteps) { load_B(ld_chunk, ld_tap); advance_ld(); }
            }
        }
        // advance (tap, slice); tapoff walks the 3x3 window: +1 along x, then down one halo row
        ++tap; ++tx; ++tapoff;
        if (tx == KS) { tx = 0; tapoff += HP - KS; }
        if (tap == taps) {
            tap = 0; tx = 0; tapoff = 0; ++chunk;
            if (step + 1 < nsteps) {                     // K-slice boundary: replace the halo tile
                __syncthreads();                         // every wave is done reading ldsA
                if (a_mode == 2) stage_A_pool(chunk);
                else {
                    store_A();
                    if (chunk + 1 < c_end) load_A(chunk + 1);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------------
    // Two phases per 32x32 tile: ALL residual loads are issued first, then the adds and stores.  (A
    // load->add->store chain per element is serialised by the possible aliasing of `res` and `out`:
    // 64 dependent L2 round trips per thread.)
    const int P = H * W;
    const int Z = a.B * a.heads;
    float *__restrict__ O = a.out + (int64_t)b * a.o_bs + (int64_t)hd * a.o_hs;
    const float *__restrict__ R = a.res ? a.res + (int64_t)b * a.r_bs + (int64_t)hd * a.r_hs : nullptr;
    const float *TE = a.temb ? a.temb + (int64_t)b * a.temb_ld : nullptr;
    float *__restrict__ WS = ksplit > 1 ? a.ws + ((int64_t)ksi * Z + z) * P * N : nullptr;
    // fused GroupNorm statistics: per-column (= output channel) sums over this wave's rows
    float cs[NT], cq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { cs[nt] = 0.f; cq[nt] = 0.f; }
    // per-lane row geometry: pixel index of row r of tile mt (or -1 outside the image)
    int pixr[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = (wm * MT + mt) * 32 + row;
            const int oy = y0 + (m >> log2TW), ox = x0 + (m & (TW - 1));
            pixr[mt][r] = (oy < H && ox < W) ? oy * W + ox : -1;
        }
    if (ksplit > 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + ncol[nt];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (n < N && pixr[mt][r] >= 0) WS[(int64_t)pixr[mt][r] * N + n] = acc[mt][nt][r];
            }
        return;
    }
    // phase 1: every residual load of the whole 64x64 wave tile is in flight before the first add
    float rv[MT][NT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + ncol[nt];
            const int nc = n < N ? n : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pc = pixr[mt][r] >= 0 ? pixr[mt][r] : 0;
                rv[mt][nt][r] = R ? R[(int64_t)pc * a.res_ld + nc] : 0.f;      // unconditional, clamped
            }
        }
    // phase 2: add, store, accumulate the fused statistics
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + ncol[nt];
            const bool nok = n < N;
            const int nc = nok ? n : 0;
            float add = 0.f;
            if (a.bias) add += a.bias[nc];
            if (TE) add += TE[nc];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = a.alpha * acc[mt][nt][r] + add + rv[mt][nt][r];
                if (nok && pixr[mt][r] >= 0) {
                    O[(int64_t)pixr[mt][r] * a.out_ld + nc] = v;
                    cs[nt] += v;
                    cq[nt] += v * v;
                }
            }
        }
    if (a.stats && ksplit == 1) {
        // rows of this wave = MT*32 pixels; lane halves hold different rows of the same column
        float *st = a.stats + ((int64_t)b * (gridDim.x * 2) + blockIdx.x * 2 + wm) * N * 2;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float s2 = cs[nt] + __shfl_xor(cs[nt], 32);
            const float q2 = cq[nt] + __shfl_xor(cq[nt], 32);
            const int n = n0 + ncol[nt];
            if (h == 0 && n < N) { st[n * 2] = s2; st[n * 2 + 1] = q2; }
        }
    }
}

// Split-K tail: out = alpha * sum_ks ws[ks] + bias + temb + res   (fixed summation order)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const anoddpm_igemm_args a)
{
    const int P = a.H * a.W;
    const int N4 = a.N >> 2;
    const int Z = a.B * a.heads;
    const int64_t total = (int64_t)Z * P * N4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n4 = (int)(i % N4);
        const int64_t zp = i / N4;
        const int64_t pix = zp % P;
        const int z = (int)(zp / P);
        const int b = z / a.heads, hd = z % a.heads;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int ks = 0; ks < a.ksplit; ++ks)
            s += ld4(a.ws + ((((int64_t)ks * Z + z) * P + pix) * a.N) + n4 * 4);
        s *= a.alpha;
        if (a.bias) s += ld4(a.bias + n4 * 4);
        if (a.temb) s += ld4(a.temb + (int64_t)b * a.temb_ld + n4 * 4);
        if (a.res) s += ld4(a.res + (int64_t)b * a.r_bs + (int64_t)hd * a.r_hs + pix * a.res_ld + n4 * 4);
        *reinterpret_cast<f32x4 *>(a.out + (int64_t)b * a.o_bs + (int64_t)hd * a.o_hs + pix * a.out_ld + n4 * 4) = s;
    }
}

// Split-K tail + fused GroupNorm statistics: grid (slabs, Z).  Thread = (pixel row, channel quad); a block
// owns a slab of pixels and ALL channels, so it can emit one row of per-channel {sum, sumsq} of the values it
// writes (same format as the igemm epilogue's statistics) -- no separate pass over the tensor.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const anoddpm_igemm_args a, const int nslab)
{
    __shared__ float lds_s[256 * 4];
    __shared__ float lds_q[256 * 4];
    const int P = a.H * a.W, N = a.N, N4 = N >> 2;
    const int Z = a.B * a.heads;                  // heads == 1 here
    const int TQ = N4 < 256 ? N4 : 256;
    const int R = 256 / TQ;
    const int npass = (N4 + TQ - 1) / TQ;
    const int tid = threadIdx.x;
    const int tq = tid % TQ, tr = tid / TQ;
    const int z = blockIdx.y, slab = blockIdx.x;
    const int sp = (P + nslab - 1) / nslab;
    const int p0 = slab * sp;
    const int p1 = (p0 + sp < P) ? p0 + sp : P;
    float *st = a.stats + ((int64_t)z * nslab + slab) * N * 2;
    for (int pass = 0; pass < npass; ++pass) {
        const int n4 = pass * TQ + tq;
        f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cq = {0.f, 0.f, 0.f, 0.f};
        if (tr < R && n4 < N4) {
            f32x4 add = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) add += ld4(a.bias + n4 * 4);
            if (a.temb) add += ld4(a.temb + (int64_t)z * a.temb_ld + n4 * 4);
            for (int pix = p0 + tr; pix < p1; pix += R) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                for (int ks = 0; ks < a.ksplit; ++ks)
                    v += ld4(a.ws + ((((int64_t)ks * Z + z) * P + pix) * N) + n4 * 4);
                v = v * a.alpha + add;
                if (a.res) v += ld4(a.res + (int64_t)z * a.r_bs + (int64_t)pix * a.res_ld + n4 * 4);
                *reinterpret_cast<f32x4 *>(a.out + (int64_t)z * a.o_bs + (int64_t)pix * a.out_ld + n4 * 4) = v;
                cs += v;
                cq += v * v;
            }
        }
        if (tr < R) {
            const int o = (tr * TQ + tq) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lds_s[o + e] = cs[e]; lds_q[o + e] = cq[e]; }
        }
        __syncthreads();
        for (int cl = tid; cl < TQ * 4; cl += 256) {
            const int c = pass * TQ * 4 + cl;
            if (c < N) {
                float s = 0.f, q = 0.f;
                for (int r = 0; r < R; ++r) { s += lds_s[r * TQ * 4 + cl]; q += lds_q[r * TQ * 4 + cl]; }
                st[c * 2] = s;
                st[c * 2 + 1] = q;
            }
        }
        __syncthreads();
    }
}

// Split-K tail partitioned by (GroupNorm group, image): grid (tail_groups, Z).  A block folds the K slabs of ITS channels over
// all pixels of its image -- so the per-channel sums it ends up with are complete and GroupNorm needs no second launch: it writes
// them (fp64) to tail_csum and, with a consumer GroupNorm attached (tail_gamma), the group's scale / shift.  The group may reach
// into the second source of a virtual concat: those channels' sums come from that tensor's tail_csum.  Every reduction runs in a
// fixed order (deterministic).  Thread = (pixel row tr, channel quad tq) of the block's <= 64 own channels.
constexpr int GT_NT = 512;             // threads of the group-partitioned tail
constexpr int GT_PARTS = 8;            // row-sum partials per channel (second reduction stage)

__global__ __launch_bounds__(GT_NT) void splitk_reduce_gn_kernel(const anoddpm_igemm_args a)
{
    __shared__ double lds_s[GT_NT * 4];
    __shared__ double lds_q[GT_NT * 4];
    __shared__ double part_s[GT_PARTS * 64];
    __shared__ double part_q[GT_PARTS * 64];
    __shared__ double grp[2];
    const int P = a.H * a.W, N = a.N;
    const int Z = a.B;                                          // heads == 1
    const int z = blockIdx.y, g = blockIdx.x;
    const int C = N + a.tail_c1;
    const int cpg = C / a.tail_groups;
    const int c_lo = g * cpg, c_hi = c_lo + cpg;                // the group's channels in the concatenation
    const int lo = c_lo < N ? c_lo : N, hi = c_hi < N ? c_hi : N;   // ... that belong to this output
    const int nq = (hi - lo) >> 2;
    const int tid = threadIdx.x;
    double cs[4] = {0.0, 0.0, 0.0, 0.0}, cq[4] = {0.0, 0.0, 0.0, 0.0};
    int rows = 0;
    if (nq > 0) {
        rows = GT_NT / nq;
        const int tq = tid % nq, tr = tid / nq;
        if (tr < rows) {
            const int n0 = lo + tq * 4;
            f32x4 add = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) add += ld4(a.bias + n0);
            if (a.temb) add += ld4(a.temb + (int64_t)z * a.temb_ld + n0);
            const int64_t slab = (int64_t)Z * P * N;
            for (int pix = tr; pix < P; pix += rows) {
                const float *src = a.ws + ((int64_t)z * P + pix) * N + n0;
                f32x4 rv = {0.f, 0.f, 0.f, 0.f};
                if (a.res) rv = ld4(a.res + (int64_t)z * a.r_bs + (int64_t)pix * a.res_ld + n0);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
                for (int ks = 0; ks < a.ksplit; ++ks) v += ld4(src + ks * slab);      // slab order: as the plain tail
                v = v * a.alpha + add;
                if (a.res) v += rv;
                *reinterpret_cast<f32x4 *>(a.out + (int64_t)z * a.o_bs + (int64_t)pix * a.out_ld + n0) = v;
#pragma unroll
                for (int e = 0; e < 4; ++e) { cs[e] += (double)v[e]; cq[e] += (double)v[e] * (double)v[e]; }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { lds_s[tid * 4 + e] = cs[e]; lds_q[tid * 4 + e] = cq[e]; }
    __syncthreads();
    // per-channel totals over the pixel rows in two fixed-order stages: GT_PARTS partial sums per channel, then their sum
    const int nown = hi - lo;                                   // <= 64
    {
        const int c = tid % 64, part = tid / 64;                // GT_NT / 64 == GT_PARTS
        if (c < nown) {
            const int cq4 = c >> 2, ce = c & 3;
            const int r0 = (rows * part) / GT_PARTS, r1 = (rows * (part + 1)) / GT_PARTS;
            double s = 0.0, q = 0.0;
            for (int r = r0; r < r1; ++r) { s += lds_s[(r * nq + cq4) * 4 + ce]; q += lds_q[(r * nq + cq4) * 4 + ce]; }
            part_s[part * 64 + c] = s;
            part_q[part * 64 + c] = q;
        }
    }
    __syncthreads();
    double s = 0.0, q = 0.0;
    if (tid < nown) {
#pragma unroll
        for (int p = 0; p < GT_PARTS; ++p) { s += part_s[p * 64 + tid]; q += part_q[p * 64 + tid]; }
        double *dst = a.tail_csum + ((int64_t)z * N + lo + tid) * 2;
        dst[0] = s;
        dst[1] = q;
    }
    if (!a.tail_gamma) return;
    // group statistics: own channels + the channels of the second source, channel order
    __syncthreads();
    if (tid < nown) { lds_s[tid] = s; lds_q[tid] = q; }
    __syncthreads();
    if (tid == 0) {
        double S = 0.0, Q = 0.0;
        for (int c = 0; c < nown; ++c) { S += lds_s[c]; Q += lds_q[c]; }
        const int o_lo = (c_lo > N ? c_lo : N) - N, o_hi = (c_hi > N ? c_hi : N) - N;
        for (int c = o_lo; c < o_hi; ++c) {
            const double *src = a.tail_other + ((int64_t)z * a.tail_c1 + c) * 2;
            S += src[0];
            Q += src[1];
        }
        const double n = (double)P * cpg;
        const double mean = S / n;
        double var = Q / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double rstd = 1.0 / sqrt(var + (double)a.tail_eps);
        grp[0] = mean;
        grp[1] = rstd;
        if (a.tail_mean) {
            a.tail_mean[(int64_t)z * a.tail_groups + g] = (float)mean;
            a.tail_rstd[(int64_t)z * a.tail_groups + g] = (float)rstd;
        }
    }
    __syncthreads();
    if (tid < cpg) {
        const int c = c_lo + tid;
        const double sc = grp[1] * (double)a.tail_gamma[c];
        a.tail_scale[(int64_t)z * C + c] = (float)sc;
        a.tail_shift[(int64_t)z * C + c] = (float)((double)a.tail_beta[c] - grp[0] * sc);
    }
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Second launch of a split-K contraction (direct or Winograd): fold the slabs, add bias / temb / residual, and
// (optionally) emit the GroupNorm statistics rows.
void launch_splitk_tail(const anoddpm_igemm_args *a, int64_t Z, int64_t P, hipStream_t s)
{
    if (a->tail_csum) {
        hipLaunchKernelGGL(splitk_reduce_gn_kernel, dim3((unsigned)a->tail_groups, (unsigned)Z), dim3(GT_NT), 0, s, *a);
    } else if (a->stats) {
        hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3((unsigned)a->stats_rows, (unsigned)Z), dim3(256), 0, s, *a, a->stats_rows);
    } else {
        const int64_t total = Z * P * (a->N / 4);
        const int64_t blocks = (total + 255) / 256;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, *a);
    }
}

}  // namespace

extern "C" int anoddpm_igemm(const anoddpm_igemm_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->a0 && a->bmat && a->out, "igemm: null pointer");
    ANODDPM_REQUIRE(a->ks == 1 || a->ks == 3, "igemm: ks must be 1 or 3");
    ANODDPM_REQUIRE(a->cfg >= 0 && a->cfg <= 7, "igemm: cfg must be 0 (128x128), 1 (64x64), 2 (Winograd F(2x2,3x3)), 3 (Winograd F(4x4,3x3)), 4 (streaming 1x1), 5 (small maps, no split-K), 6 (F(2x2,3x3) on 16x16 / 32x32 maps, no split-K) or 7 (F(4x4,3x3) with split-bf16 products)");
    ANODDPM_REQUIRE(a->cfg == 5 || a->cfg == 6 || a->cfg == 3 || !a->fold_gamma, "igemm: the GroupNorm fold of the operand (fold_*) is a cfg 3 / 5 / 6 feature");
    ANODDPM_REQUIRE(a->cfg == 3 || !a->stats_csum, "igemm: stats_csum (atomically accumulated fp64 sums) is a cfg 3 feature");
    ANODDPM_REQUIRE(a->cfg == 3 || !a->gnb_partial, "igemm: gnb_partial (GroupNorm-backward sums from the data-gradient epilogue) is a cfg 3 feature");
    ANODDPM_REQUIRE(a->res_mode == 0 || (a->res_mode == 1 && a->cfg == 3 && a->res && a->H % 2 == 0 && a->W % 2 == 0),
                    "igemm: res_mode 1 (half-resolution residual, nearest x2 on the read) is a cfg 3 feature");
    const int BM = a->cfg == 0 ? 128 : 64, BN = BM;
    ANODDPM_REQUIRE(a->c0 > 0 && a->c0 % 4 == 0 && a->c1 >= 0 && a->c1 % 4 == 0, "igemm: channel counts must be multiples of 4");
    ANODDPM_REQUIRE(a->c1 == 0 || (a->a1 && a->c0 % KC == 0), "igemm: dual source needs a1 and c0 %% 32 == 0");
    ANODDPM_REQUIRE(a->N >= 1 && a->B >= 1 && a->heads >= 1 && a->ksplit >= 1, "igemm: bad sizes");
    ANODDPM_REQUIRE(a->a_mode >= 0 && a->a_mode <= 2 && a->b_mode >= 0 && a->b_mode <= 2, "igemm: bad mode");
    ANODDPM_REQUIRE(a->a_mode == 0 || a->ks == 3, "igemm: resampling is only fused into 3x3 loads");
    ANODDPM_REQUIRE(a->b_mode == 0 || a->ks == 1, "igemm: activation B operands need ks == 1");
    ANODDPM_REQUIRE(a->b_mode != 2 || a->N % 4 == 0, "igemm: b_mode 2 needs N %% 4 == 0");
    ANODDPM_REQUIRE(!a->stats || a->heads == 1, "igemm: fused statistics need heads == 1");
    ANODDPM_REQUIRE(!a->stats || a->ksplit == 1 || a->stats_rows >= 1, "igemm: split-K statistics need stats_rows");
    if (a->tail_csum) {
        ANODDPM_REQUIRE(a->ksplit > 1 && a->heads == 1 && !a->stats && a->N % 4 == 0, "igemm: the GroupNorm tail needs split-K, heads == 1 and no statistics rows");
        ANODDPM_REQUIRE(a->tail_groups >= 1 && a->tail_groups <= 65535 && a->tail_c1 >= 0 && (a->N + a->tail_c1) % a->tail_groups == 0 &&
                        ((a->N + a->tail_c1) / a->tail_groups) % 4 == 0 && (a->N + a->tail_c1) / a->tail_groups <= 64,
                        "igemm: GroupNorm tail: (N + tail_c1) / tail_groups must be a multiple of 4 and <= 64");
        ANODDPM_REQUIRE(a->tail_c1 == 0 || !a->tail_gamma || a->tail_other, "igemm: GroupNorm tail over a concatenation needs tail_other");
        ANODDPM_REQUIRE(!a->tail_gamma || (a->tail_beta && a->tail_scale && a->tail_shift), "igemm: GroupNorm tail: null affine / output");
        ANODDPM_REQUIRE((a->tail_mean == nullptr) == (a->tail_rstd == nullptr), "igemm: tail_mean and tail_rstd go together");
        ANODDPM_REQUIRE(a->tail_gamma || a->tail_c1 == 0, "igemm: tail_c1 without a consumer GroupNorm");
    }
    ANODDPM_REQUIRE(a->b_mode == 0 || a->ldb % 4 == 0, "igemm: ldb must be a multiple of 4");
    ANODDPM_REQUIRE(a->a0_ld % 4 == 0 && (a->c1 == 0 || a->a1_ld % 4 == 0), "igemm: pixel strides must be multiples of 4 floats");
    ANODDPM_REQUIRE(al16(a->a0) && al16(a->bmat) && (!a->a1 || al16(a->a1)), "igemm: operands must be 16-byte aligned");
    ANODDPM_REQUIRE((a->a0_bs | a->a0_hs | a->a1_bs | a->a1_hs | a->b_bs | a->b_hs) % 4 == 0, "igemm: strides must be multiples of 4 floats");
    ANODDPM_REQUIRE(!a->gn_scale || (a->gn_shift && a->gn_ld % 4 == 0 && al16(a->gn_scale) && al16(a->gn_shift)), "igemm: bad GroupNorm affine");
    ANODDPM_REQUIRE(a->ksplit == 1 || (a->ws && a->N % 4 == 0 && al16(a->ws) && a->out_ld % 4 == 0 && al16(a->out) &&
                                       (!a->res || (a->res_ld % 4 == 0 && al16(a->res))) && (!a->bias || al16(a->bias)) &&
                                       (!a->temb || (a->temb_ld % 4 == 0 && al16(a->temb)))),
                    "igemm: split-K needs a workspace and 16-byte aligned epilogue operands");
    int H = a->H, W = a->W, log2TW, TH;
    ANODDPM_REQUIRE(H >= 1 && W >= 1, "igemm: bad image size");
    const int64_t P = (int64_t)H * W;
    if (a->cfg == 3) {
        const int rc = anoddpm::launch_winograd43(a, anoddpm::as_stream(stream));
        if (rc != ANODDPM_OK) return rc;
        if (a->ksplit > 1) launch_splitk_tail(a, (int64_t)a->B, P, anoddpm::as_stream(stream));
        return anoddpm::check_launch("igemm(winograd43 split-K tail)");
    }
    if (a->cfg == 4) return anoddpm::launch_pointwise_stream(a, anoddpm::as_stream(stream));
    if (a->cfg == 5) return anoddpm::launch_smallmap(a, anoddpm::as_stream(stream));
    if (a->cfg == 6) return anoddpm::launch_wino23s(a, anoddpm::as_stream(stream));
    if (a->cfg == 7) return anoddpm::launch_winograd43b(a, anoddpm::as_stream(stream));
    if (a->cfg == 2) {
        const int rc = anoddpm::launch_winograd(a, anoddpm::as_stream(stream));
        if (rc != ANODDPM_OK) return rc;
        if (a->ksplit > 1) launch_splitk_tail(a, (int64_t)a->B, P, anoddpm::as_stream(stream));
        return anoddpm::check_launch("igemm(winograd split-K tail)");
    }
    anoddpm_igemm_args k = *a;
    int tiles_x, tiles_y;
    if (a->ks == 1) {                 // treat the image as one row of P pixels
        k.H = 1; k.W = (int)P; H = 1; W = (int)P;
        TH = 1;
        log2TW = (BM == 128) ? 7 : 6;
        tiles_x = (int)((P + BM - 1) / BM);
        tiles_y = 1;
    } else {
        ANODDPM_REQUIRE((W & (W - 1)) == 0 && W >= 2, "igemm: 3x3 path needs power-of-two width >= 2");
        const int TW = W < 32 ? W : 32;
        log2TW = 0;
        while ((1 << log2TW) < TW) ++log2TW;
        TH = BM / TW;
        tiles_x = W / TW;
        tiles_y = (H + TH - 1) / TH;
        ANODDPM_REQUIRE((TH + 2) * (TW + 2) <= (BM == 128 ? 204 : 136), "igemm: halo tile exceeds LDS budget for this shape");
    }
    const int64_t Z = (int64_t)a->B * a->heads;
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)((a->N + BN - 1) / BN), (unsigned)(Z * a->ksplit));
    ANODDPM_REQUIRE(grid.y <= 65535 && Z * a->ksplit <= 65535, "igemm: grid too large");
    hipStream_t s = anoddpm::as_stream(stream);
    const bool conv = (a->b_mode == 0) && ((a->c0 + a->c1) % KC == 0);
    if (BM == 128) {
        if (conv) hipLaunchKernelGGL((igemm_kernel<128, 128, true>), grid, dim3(256), 0, s, k, log2TW, TH, tiles_x);
        else      hipLaunchKernelGGL((igemm_kernel<128, 128, false>), grid, dim3(256), 0, s, k, log2TW, TH, tiles_x);
    } else {
        if (conv) hipLaunchKernelGGL((igemm_kernel<64, 64, true>), grid, dim3(256), 0, s, k, log2TW, TH, tiles_x);
        else      hipLaunchKernelGGL((igemm_kernel<64, 64, false>), grid, dim3(256), 0, s, k, log2TW, TH, tiles_x);
    }
    if (a->ksplit > 1) launch_splitk_tail(a, Z, P, s);
    return anoddpm::check_launch("igemm");
}
