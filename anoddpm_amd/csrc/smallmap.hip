// Small-map contractions without split-K (cfg 5 of anoddpm_igemm), gfx950.
//
// Replaces, on maps of <= 256 pixels (the 8x8 / 16x16 levels of the denoiser): nn.Conv2d 3x3 of the ResBlocks (UNet.py:172,193),
// the 1x1 skip convolutions (UNet.py:200) and the Conv1d k=1 of AttentionBlock (to_qkv / proj_out, UNet.py:115,117), with the same
// fusions as the other contraction kernels -- GroupNorm32-apply + SiLU on the operand (UNet.py:170-171,190-191,113), the virtual
// torch.cat([h, skip]) (UNet.py:402), bias / timestep-embedding / residual adds, the GroupNorm partial sums of the output -- and one
// more: the GroupNorm FINALIZE of the operand (UNet.py:409-411) runs in this kernel's prologue from the producer's statistics
// rows (`fold_*`), so that no anoddpm_gn_finalize launch sits between producer and consumer.
//
// Why another kernel.  On these maps the whole batch is M = B * P <= 1024 output rows against K' = taps * K up to 9216: the
// 64 x 64-tile kernel (igemm.hip) has to split K over 8-16 workgroups per tile to fill 256 CUs and then needs a second launch
// to fold the slabs.  A dependent launch costs 4.5 us here whatever it does (profiles/r4a_tl_*), a 8x8 layer is two of them
// plus a 20 us main kernel whose workgroups live for nine K-steps.  Here the batch is folded into M, a workgroup owns a
// 16- or 32-row x 32- or 96-channel output tile over ALL of K', and its eight waves split K' between them: the cross-wave
// fold goes through LDS inside the launch, in a fixed order.  256 workgroups for every shape of the configuration-2 model.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (fp32 in / fp32 accumulate, bitwise an fmaf chain per wave), fp32 cross-wave fold.
// Operands.  A: the activated halo tile of a 256-channel K chunk is staged in LDS once per workgroup (double buffered; raw
// values of the next chunk wait in registers during the current chunk's MFMAs); a lane reads one ds_read_b128 (four consecutive
// k of its row) per four MFMAs.  B: the packed weights [tap][K/4][N][4] stream from L2 straight into a register ring, one
// 16-byte load per lane per four MFMAs, PD steps ahead; the N tile is the fast grid dimension, so the workgroups of one XCD
// (block id mod 8) share weight columns in their L2.
#include "common.h"

using anoddpm::silu_f;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SM_NT = 512;                 // 8 waves: they split K'

__host__ __device__ constexpr int sm_kch(int ks) { return ks == 3 ? 256 : 512; }     // channels per K chunk: 32 / 64 per wave
__host__ __device__ constexpr int sm_hpix(int ks, int tm, int w) { return ks == 1 ? tm : (tm / w + 2) * (w + 2); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sm_rsrc(const float *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7ffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 sm_bld4(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned wave_bytes)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)wave_bytes, 0));
}

template <int KS, int RT, int CT, int LOG2W>
__global__ __launch_bounds__(SM_NT) void smallmap_kernel(const anoddpm_igemm_args a)
{
    constexpr int TM = 16 * RT, TN = 16 * CT, TAPS = KS * KS;
    constexpr int KCH = sm_kch(KS), PITCH = KCH + 4;            // floats between halo pixels in LDS (16-byte slots advance by an odd count)
    constexpr int CPW = KCH / 8;                                // channels of a chunk per wave
    constexpr int SPT = CPW / 16;                               // 16-k steps per tap per wave
    constexpr int SPC = TAPS * SPT;                             // steps per chunk per wave: 18 (3x3) / 4 (1x1)
    constexpr int PD = KS == 3 ? 6 : 4;                         // B prefetch distance in steps (divides SPC: static ring slots)
    constexpr int W = 1 << LOG2W;
    constexpr int HPIX = sm_hpix(KS, TM, W);
    constexpr int HP = W + 2;                                   // halo row pitch in pixels (KS == 3)
    constexpr int QPP = KCH / 4;                                // channel quads per pixel of a chunk
    constexpr int PPI = SM_NT / QPP;                            // pixels staged per pass of the workgroup
    constexpr int NI = (HPIX + PPI - 1) / PPI;                  // staged float4 per thread per chunk
    constexpr int RP = TN + 4;                                  // row pitch of the cross-wave fold buffers
    static_assert(SPC % PD == 0, "ring slots must be static");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, q = lane >> 4;
    const int K = a.c0 + a.c1, K4 = K >> 2, N = a.N;
    const int P = a.H * a.W;
    const int nchunks = (K + KCH - 1) / KCH;
    const int Kpad = nchunks * KCH;
    const int nbuf = nchunks > 1 ? 2 : 1;
    float *aff_sc = lds, *aff_sh = lds + Kpad;
    float *Abuf = lds + 2 * Kpad;                               // [nbuf][HPIX][PITCH]
    float *red = Abuf;                                          // epilogue: [5][TM][RP] (the staging buffers are dead by then)

    const int n0 = blockIdx.x * TN;
    const int p0g = blockIdx.y * TM;                            // first output row of the tile in [B * P]
    const int b = p0g / P, p0 = p0g - b * P;
    const int y0 = p0 >> LOG2W;
    const float *A0 = a.a0 + (int64_t)b * a.a0_bs;
    const float *A1 = a.a1 ? a.a1 + (int64_t)b * a.a1_bs : A0;

    // ---- B stream: buffer loads, per-lane byte offset + a scalar offset per (chunk, step): no vector address arithmetic ----
    const __amdgpu_buffer_rsrc_t rW = sm_rsrc(a.bmat);
    const unsigned ulane = ((unsigned)q * (unsigned)N + (unsigned)(n0 + l16)) * 16u;
    const int total_steps = nchunks * SPC;
    auto b_off = [&](int chunk, int s) -> unsigned {            // s is a compile-time constant at every call site
        if (chunk * SPC + s >= total_steps) { chunk = nchunks - 1; s = SPC - 1; }       // past the end: re-read the last step (unused)
        const int tap = s / SPT, i = s % SPT;
        int k4 = (chunk * KCH + wave * CPW + 16 * i) >> 2;
        if (k4 + 4 > K4) k4 = 0;                                // a step beyond K (K % 16 == 0: all of it; the A operand is zero there)
        return ((unsigned)tap * (unsigned)K4 + (unsigned)k4) * (unsigned)N * 16u;
    };
    f32x4 ring[PD][CT];
#pragma unroll
    for (int s = 0; s < PD; ++s) {
        const unsigned o = b_off(0, s);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) ring[s][ct] = sm_bld4(rW, ulane, o + ct * 256u);
    }

    // ---- A staging geometry: item j of this thread = halo pixel hp_j, channel quad kq ----
    const int kq = tid % QPP;
    int soff[NI];                                               // source pixel index inside the image, -1 = zero padding / no item
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int hp = tid / QPP + PPI * j;
        int sp = -1;
        if (hp < HPIX) {
            if (KS == 1) sp = p0 + hp;
            else {
                const int hy = hp / HP, hx = hp - hy * HP;
                const int gy = y0 - 1 + hy, gx = hx - 1;
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < W) sp = gy * W + gx;
            }
        }
        soff[j] = sp;
    }
    f32x4 araw[NI];
    unsigned avalid = 0;
    auto load_A = [&](int chunk) {
        const int k = chunk * KCH + 4 * kq;
        const bool kv = k < K;
        const bool first = k < a.c0;
        const float *src = first ? A0 + (kv ? k : 0) : A1 + (kv ? k - a.c0 : 0);
        const int ld = first ? a.a0_ld : a.a1_ld;
        avalid = 0;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const bool ok = kv && soff[j] >= 0;
            araw[j] = *reinterpret_cast<const f32x4 *>(src + (int64_t)(soff[j] >= 0 ? soff[j] : 0) * ld);   // unconditional, clamped
            avalid |= (ok ? 1u : 0u) << j;
        }
    };
    const bool affine = a.gn_scale != nullptr || a.fold_gamma != nullptr;
    const bool act = a.act != 0;
    auto store_A = [&](int chunk) {
        float *dst = Abuf + (chunk & (nbuf - 1)) * (HPIX * PITCH);
        const int k = chunk * KCH + 4 * kq;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (affine) { sc = *reinterpret_cast<const f32x4 *>(aff_sc + k); sh = *reinterpret_cast<const f32x4 *>(aff_sh + k); }
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int hp = tid / QPP + PPI * j;
            f32x4 v = araw[j];
            if (affine) v = v * sc + sh;
            if (act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
            if (hp < HPIX) *reinterpret_cast<f32x4 *>(dst + hp * PITCH + 4 * kq) = ((avalid >> j) & 1) ? v : zero;   // zero padding AFTER the transform
        }
    };
    load_A(0);

    // ---- GroupNorm affine of the operand: given, or finished here from the producer's statistics ----
    if (a.fold_gamma) {
        // scratch in the (still unused) staging buffers: per-channel fp64 {sum, sumsq}, then per-group {mean, rstd}
        double *csum = reinterpret_cast<double *>(Abuf);
        double *gst = csum + 2 * Kpad;
        const int groups = a.fold_groups, cpg = K / groups;
        for (int c = tid; c < K; c += SM_NT) {
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
                for (int r0 = 0; r0 < rows; r0 += 4) {           // four independent row loads in flight; summed in row order
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
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { S += csum[2 * c]; Q += csum[2 * c + 1]; }   // channel order
            const double n = (double)P * cpg;
            const double mean = S / n;
            double var = Q / n - mean * mean;
            var = var > 0.0 ? var : 0.0;
            gst[2 * tid] = mean;
            gst[2 * tid + 1] = 1.0 / sqrt(var + (double)a.fold_eps);
        }
        __syncthreads();
        for (int c = tid; c < Kpad; c += SM_NT) {
            float scv = 0.f, shv = 0.f;
            if (c < K) {
                const int g = c / cpg;
                const double sc = gst[2 * g + 1] * (double)a.fold_gamma[c];
                scv = (float)sc;
                shv = (float)((double)a.fold_beta[c] - gst[2 * g] * sc);
            }
            aff_sc[c] = scv; aff_sh[c] = shv;
        }
        __syncthreads();
    } else if (a.gn_scale) {
        for (int c = tid; c < Kpad; c += SM_NT) {
            aff_sc[c] = c < K ? a.gn_scale[(int64_t)b * a.gn_ld + c] : 0.f;
            aff_sh[c] = c < K ? a.gn_shift[(int64_t)b * a.gn_ld + c] : 0.f;
        }
        __syncthreads();
    }
    store_A(0);
    __syncthreads();

    // ---- main loop: one basic block per chunk; A fragments one step ahead, B ring PD steps ahead ----
    int hp0[RT];                                                // halo pixel of this lane's row of row tile rt, tap (0, 0)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int m = rt * 16 + l16;
        hp0[rt] = (KS == 1 ? m : (m >> LOG2W) * HP + (m & (W - 1))) * PITCH + 4 * q;
    }
    f32x4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const float *Ab = Abuf + (chunk & (nbuf - 1)) * (HPIX * PITCH) + wave * CPW;
        const bool more = chunk + 1 < nchunks;
        if (more) load_A(chunk + 1);
        f32x4 av[2][RT];
        auto read_frags = [&](int s, int set) {                  // s compile-time
            const int tap = s / SPT, i = s % SPT;
            const int toff = (KS == 1 ? 0 : (tap / 3) * HP + (tap % 3)) * PITCH + 16 * i;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) av[set][rt] = *reinterpret_cast<const f32x4 *>(Ab + hp0[rt] + toff);
        };
        read_frags(0, 0);
#pragma unroll
        for (int s = 0; s < SPC; ++s) {
            const int cur = s & 1, slot = s % PD;
            if (s + 1 < SPC) read_frags(s + 1, cur ^ 1);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][rt][e], ring[slot][ct][e], acc[rt][ct], 0, 0, 0);
            // refill the slot with step s + PD (possibly of the next chunk)
            const unsigned o = (s + PD < SPC) ? b_off(chunk, s + PD) : b_off(chunk + 1, s + PD - SPC);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) ring[slot][ct] = sm_bld4(rW, ulane, o + ct * 256u);
            __builtin_amdgcn_sched_barrier(0);                   // keep the requests where they are: left alone, hipcc sinks them next to their use
        }
        if (more) {
            if (nbuf == 1) __syncthreads();
            store_A(chunk + 1);                                  // the other buffer: its readers passed the previous barrier
        }
        __syncthreads();
    }

    // ---- cross-wave fold (fixed order), epilogue, statistics ----
    // D layout of v_mfma_f32_16x16x4_f32: this lane holds rows 4 q + e (e = 0..3) of column l16
    auto red_at = [&](int slot, int rt, int ct, int e) { return red + ((slot * TM + rt * 16 + 4 * q + e) * RP + ct * 16 + l16); };
    if (wave >= 4) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) *red_at(wave - 4, rt, ct, e) = acc[rt][ct][e];
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[rt][ct][e] += *red_at(wave, rt, ct, e);
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) *red_at(wave, rt, ct, e) = acc[rt][ct][e];
    }
    __syncthreads();
    float *O = a.out + (int64_t)b * a.o_bs;
    const float *R = a.res ? a.res + (int64_t)b * a.r_bs : nullptr;
    const float *TE = a.temb ? a.temb + (int64_t)b * a.temb_ld : nullptr;
    constexpr int ITEMS = TM * (TN / 4);
    for (int it = tid; it < ITEMS; it += SM_NT) {
        const int row = it / (TN / 4), c4 = it - row * (TN / 4);
        const float *r0 = red + row * RP + c4 * 4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(r0);
        v += *reinterpret_cast<const f32x4 *>(r0 + TM * RP);
        v += *reinterpret_cast<const f32x4 *>(r0 + 2 * TM * RP);
        v += *reinterpret_cast<const f32x4 *>(r0 + 3 * TM * RP);
        const int n = n0 + c4 * 4;
        v = v * a.alpha;
        if (a.bias) v += *reinterpret_cast<const f32x4 *>(a.bias + n);
        if (TE) v += *reinterpret_cast<const f32x4 *>(TE + n);
        if (R) v += *reinterpret_cast<const f32x4 *>(R + (int64_t)(p0 + row) * a.res_ld + n);
        *reinterpret_cast<f32x4 *>(O + (int64_t)(p0 + row) * a.out_ld + n) = v;
        if (a.stats) *reinterpret_cast<f32x4 *>(red + (4 * TM + row) * RP + c4 * 4) = v;       // a fifth slot: the final values
    }
    if (a.stats) {
        __syncthreads();
        if (tid < TN) {
            const float *col = red + 4 * TM * RP + tid;
            float s = 0.f, qq = 0.f;
#pragma unroll 4
            for (int r = 0; r < TM; ++r) { const float v = col[r * RP]; s += v; qq += v * v; }
            float *st = a.stats + (((int64_t)b * (P / TM) + p0 / TM) * N + n0 + tid) * 2;
            st[0] = s; st[1] = qq;
        }
    }
}

}  // namespace

namespace anoddpm {

// Tile shape for a small-map contraction, or 0 when cfg 5 does not take it.  Returns RT * 16 + CT; the same rule is used by the
// launcher below and (through anoddpm_smallmap_tile) by the planner, which needs TM for the statistics row count.
int smallmap_tile(int ks, int H, int W, int K, int c0, int N, int B)
{
    const int P = H * W;
    if (!(ks == 1 || ks == 3) || P > 256 || P < 16 || (W != 4 && W != 8 && W != 16) || H * W != P) return 0;
    if (K % 16 || c0 % 4 || K > 1024 || N % 32 || K < 16) return 0;
    const int M = B * P;
    // wide column tiles first: every workgroup of a row tile stages (and normalises) the same operand rows, so fewer, wider
    // column tiles mean less redundant prologue work (to_qkv at 8x8: 16 x 96 tiles = 256 workgroups instead of 32 x 32 = 384)
    static const int cand[4][2] = {{2, 6}, {1, 6}, {2, 2}, {1, 2}};
    int best = 0;
    for (int c = 0; c < 4; ++c) {
        const int rt = cand[c][0], ct = cand[c][1];
        const int tm = 16 * rt, tn = 16 * ct;
        if (P % tm || tm % W || N % tn) continue;
        if (ks == 3 && ct == 6) continue;                       // instantiated: 3x3 with 32-channel tiles, 1x1 with 32 / 96
        const int wgs = (M / tm) * (N / tn);
        if (wgs >= 200) return rt * 16 + ct;                    // largest tile that still fills the chip
        best = rt * 16 + ct;                                    // else the smallest fitting one (candidates are ordered by size)
    }
    return best;
}

int launch_smallmap(const anoddpm_igemm_args *a, hipStream_t s)
{
    const int K = a->c0 + a->c1;
    ANODDPM_REQUIRE(a->b_mode == 0 && a->heads == 1 && a->a_mode == 0 && a->ksplit == 1, "smallmap: needs packed weights, heads 1, a_mode 0, ksplit 1");
    const int tile = smallmap_tile(a->ks, a->H, a->W, K, a->c0, a->N, a->B);
    ANODDPM_REQUIRE(tile != 0, "smallmap: shape not supported (H*W <= 256, W in {4, 8, 16}, K %% 16 == 0, 16 <= K <= 1024, N %% 32 == 0)");
    ANODDPM_REQUIRE(a->c1 == 0 || a->a1, "smallmap: dual source needs a1");
    ANODDPM_REQUIRE(a->out_ld % 4 == 0 && (!a->res || a->res_ld % 4 == 0) && a->a0_ld % 4 == 0 && (a->c1 == 0 || a->a1_ld % 4 == 0),
                    "smallmap: pixel strides must be multiples of 4 floats");
    ANODDPM_REQUIRE(!a->tail_csum, "smallmap: no split-K tail");
    ANODDPM_REQUIRE(!a->gn_scale || a->gn_shift, "smallmap: gn_scale without gn_shift");
    if (a->fold_gamma) {
        ANODDPM_REQUIRE(a->fold_beta && a->fold_stats0 && (a->c1 == 0 || a->fold_stats1), "smallmap: GroupNorm fold: null pointer");
        ANODDPM_REQUIRE(a->fold_groups >= 1 && a->fold_groups <= 64 && K % a->fold_groups == 0, "smallmap: GroupNorm fold: bad group count");
        ANODDPM_REQUIRE((a->fold_fmt0 != 0 || a->fold_rows0 >= 1) && (a->c1 == 0 || a->fold_fmt1 != 0 || a->fold_rows1 >= 1),
                        "smallmap: GroupNorm fold: statistics rows missing");
    }
    const int rt = tile >> 4, ct = tile & 15;
    const int TM = 16 * rt, TN = 16 * ct;
    const int P = a->H * a->W;
    const int kch = sm_kch(a->ks);
    const int nchunks = (K + kch - 1) / kch, Kpad = nchunks * kch;
    const int hpix = sm_hpix(a->ks, TM, a->W);
    size_t stage = (size_t)(nchunks > 1 ? 2 : 1) * hpix * (kch + 4) * sizeof(float);
    const size_t fold = (size_t)(2 * Kpad + 128) * sizeof(double);
    const size_t redb = (size_t)5 * TM * (TN + 4) * sizeof(float);
    if (stage < fold) stage = fold;
    if (stage < redb) stage = redb;
    const size_t lds = (size_t)2 * Kpad * sizeof(float) + stage;
    ANODDPM_REQUIRE(lds <= 160 * 1024, "smallmap: LDS budget exceeded");
    dim3 grid((unsigned)(a->N / TN), (unsigned)((int64_t)a->B * P / TM));
#define SM_LAUNCH(KS_, RT_, CT_, LW_)                                                                                     \
    do {                                                                                                                  \
        static bool attr_done[ANODDPM_MAX_DEV];                                                                           \
        if (int rc_ = allow_big_lds(reinterpret_cast<const void *>(&smallmap_kernel<KS_, RT_, CT_, LW_>), attr_done,      \
                                    "igemm(smallmap)"))                                                                   \
            return rc_;                                                                                                   \
        hipLaunchKernelGGL((smallmap_kernel<KS_, RT_, CT_, LW_>), grid, dim3(SM_NT), lds, s, *a);                         \
        return check_launch("igemm(smallmap)");                                                                           \
    } while (0)
    const int lw = a->W == 4 ? 2 : (a->W == 8 ? 3 : 4);
    if (a->ks == 3) {
        if (rt == 1 && ct == 2) { if (lw == 2) SM_LAUNCH(3, 1, 2, 2); if (lw == 3) SM_LAUNCH(3, 1, 2, 3); SM_LAUNCH(3, 1, 2, 4); }
        if (rt == 2 && ct == 2) { if (lw == 2) SM_LAUNCH(3, 2, 2, 2); if (lw == 3) SM_LAUNCH(3, 2, 2, 3); SM_LAUNCH(3, 2, 2, 4); }
    } else {
        // LOG2W does not enter the 1x1 kernel (the tile is TM consecutive pixels)
        if (rt == 1 && ct == 2) SM_LAUNCH(1, 1, 2, 3);
        if (rt == 1 && ct == 6) SM_LAUNCH(1, 1, 6, 3);
        if (rt == 2 && ct == 2) SM_LAUNCH(1, 2, 2, 3);
        if (rt == 2 && ct == 6) SM_LAUNCH(1, 2, 6, 3);
    }
#undef SM_LAUNCH
    set_error("smallmap: no instantiation for tile %d x %d", TM, TN);
    return ANODDPM_EINVAL;
}

}  // namespace anoddpm

extern "C" int anoddpm_smallmap_tile(int32_t ks, int32_t H, int32_t W, int32_t K, int32_t c0, int32_t N, int32_t B)
{
    return anoddpm::smallmap_tile(ks, H, W, K, c0, N, B);
}
