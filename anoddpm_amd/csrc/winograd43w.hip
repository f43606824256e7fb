// Winograd F(4x4,3x3) convolution, ONE WAVE PER SIMD: the 128-channel workgroup of winograd43r.hip as four 64-lane waves with 512
// registers each (round 6).  Same contract, same LDS layouts, same staging and input transform as wino43r_kernel
// (nn.Conv2d 3x3, UNet.py:172,193, with GroupNorm-apply + SiLU, nearest-x2 and the two-source concat fused into the operand load;
// bias / time-embedding / residual / GroupNorm statistics in the epilogue).  What differs:
//
//   wino43r_kernel   8 waves, wave w = all 36 positions x channels 16w..16w+15 x 16 tiles: 144 accumulator registers, two waves
//                    per SIMD.  Every wave reads all of V (36 ds_read_b128 per chunk) and its own 36 B fragments.
//   this kernel      4 waves, wave w = all 36 positions x channels 32w..32w+31: 288 accumulator registers (256 of them AGPRs), one
//                    wave per SIMD.  An A fragment feeds TWO MFMAs (half the LDS fragment reads per MFMA), the B ring is 4 or 6
//                    positions x 2 fragments deep, and -- with TSPLIT -- the input transform does not run as an MFMA-free block at the
//                    top of the chunk: each of the wave's three transform passes issues its 12 ds_read2_b64 at one position of the
//                    MFMA stream and consumes them three positions later, operands resident.
//
// The MFMA and VALU instruction totals per SIMD are those of wino43r_kernel (the 768 transform items and 1 536 staging slots of a
// chunk do not shrink with fewer waves): on this pipe, where the fp32 MFMA and the VALU share the issue port, the kernel can only
// win what the two-wave structure loses to idle time.  MEASURED (profiles/r6_f43_onewave_ab.txt): 13-19 % slower than
// wino43r_kernel on every layer shape of config 2 -- the accumulators overflow the 256 AGPRs (68-116 v_accvgpr moves per chunk) and a
// lone wave has no partner to cover its waits.  Kept as a measurement selector (ANODDPM_DEBUG5 = 4 / 5 / 8 / 9; 6 / 7 in the op
// tests); no plan selects it.
#include <type_traits>

#include "common.h"
#include "gn_fold.h"

using anoddpm::silu_f;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int W4_NT = 256;                 // threads: 4 waves, one per SIMD
constexpr int W4_KC = 16;                  // channels per K iteration
constexpr int W4_PW = 18;                  // patch width / height (16 + 2)
constexpr int W4_PPIX = W4_PW * W4_PW;     // 324 patch pixels
constexpr int W4_PITCH = 5;                // float4 per patch pixel (4 quads + 1 pad)
constexpr int W4_PJ = 6;                   // staging slots per thread (6 * 256 = 1536 >= 324 * 4)
constexpr int W4_SLOTPX = W4_PJ * W4_NT / 4;          // 384 pixel slots per buffer
constexpr int W4_DT = W4_SLOTPX * W4_PITCH;           // float4 per patch buffer
constexpr int W4_V = 36 * 16 * 4;                     // float4 per V buffer: [pos][tile][quad]
constexpr int W4_KMAX = 1024;
constexpr int W4_AFF = 2 * W4_KMAX / 4;
constexpr int W4_LDS_FLOATS = (2 * W4_DT + 2 * W4_V + W4_AFF) * 4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrcw(const float *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7ffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 bldw(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned wave_bytes)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)wave_bytes, 0));
}

// A^T of F(4x4,3x3) applied to six values: rows (1 1 1 1 1 0), (0 1 -1 2 -2 0), (0 1 1 4 4 0), (0 1 -1 8 -8 1)
__device__ __forceinline__ void at6w(const float (&m)[6], float (&o)[4])
{
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    o[0] = m[0] + s12 + s34;
    o[1] = d12 + 2.f * d34;
    o[2] = s12 + 4.f * s34;
    o[3] = d12 + 8.f * d34 + m[5];
}

// W4_RING = positions of B fragments in flight (x 2 fragments each).  288 accumulator registers exceed the 256 AGPRs: hipcc keeps
// every MFMA destination in an AGPR and moves the overflow through VGPRs (68-116 v_accvgpr moves per chunk); with 9 positions in
// flight the hot loop also spills (34 scratch loads per chunk), with 6 a dozen, with 4 none.
template <bool FAST, bool TSPLIT, int W4_RING>
__global__ __launch_bounds__(W4_NT, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino43w_kernel(const anoddpm_igemm_args a)
{
    __shared__ __attribute__((aligned(16))) float lds[W4_LDS_FLOATS];
    f32x4 *ldsD = reinterpret_cast<f32x4 *>(lds);
    f32x4 *ldsV = ldsD + 2 * W4_DT;
    f32x4 *ldsAff = ldsV + 2 * W4_V;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    const int K = a.c0 + a.c1, N = a.N, K4 = K >> 2;
    const int tiles_x = W >> 4;
    const int y0 = (blockIdx.x / tiles_x) * 16, x0 = (blockIdx.x % tiles_x) * 16;
    const int n0 = blockIdx.y * 128;
    const int b = blockIdx.z;
    const int a_mode = a.a_mode;

    const float *A0 = a.a0 + (int64_t)b * a.a0_bs;
    const float *A1 = a.a1 ? a.a1 + (int64_t)b * a.a1_bs : nullptr;
    const float *gsc = a.gn_scale ? a.gn_scale + (int64_t)b * a.gn_ld : nullptr;
    const float *gsh = a.gn_shift ? a.gn_shift + (int64_t)b * a.gn_ld : nullptr;
    const bool fold = a.fold_gamma != nullptr;
    const bool affine = gsc != nullptr || fold, act = a.act != 0;
    const int nchunks = K / W4_KC;

    // ---- patch staging (pixel = idx >> 2, quad = idx & 3): geometry fixed for the workgroup
    int spix[W4_PJ];
    const int pq = tid & 3;
#pragma unroll
    for (int j = 0; j < W4_PJ; ++j) {
        const int p = (tid + j * W4_NT) >> 2;
        const int py = p / W4_PW, px = p - py * W4_PW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        int sp = -1;
        if (p < W4_PPIX && gy >= 0 && gy < H && gx >= 0 && gx < W)
            sp = (a_mode == 0) ? gy * W + gx : (gy >> 1) * (W >> 1) + (gx >> 1);
        spix[j] = sp;
    }
    f32x4 praw[W4_PJ];
    f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rA0 = rsrcw(A0), rA1 = rsrcw(A1 ? A1 : A0);
    const __amdgpu_buffer_rsrc_t rSc = rsrcw(gsc ? gsc : A0), rSh = rsrcw(gsh ? gsh : A0);
    auto load_patch = [&](int chunk) {                              // unconditional loads, clamped addresses
        const int kbase = chunk * W4_KC;
        const bool first = kbase < a.c0;
        const __amdgpu_buffer_rsrc_t r = first ? rA0 : rA1;
        const unsigned ld = (unsigned)(first ? a.a0_ld : a.a1_ld);
        const unsigned koff = (unsigned)(first ? kbase : kbase - a.c0) * 4u;
#pragma unroll
        for (int j = 0; j < W4_PJ; ++j) {
            const unsigned sp = spix[j] >= 0 ? (unsigned)spix[j] : 0u;
            praw[j] = bldw(r, (sp * ld + (unsigned)(pq * 4)) * 4u, koff);
        }
    };
    // GroupNorm-apply + SiLU of slots j0 .. j0 + 2, zero padding AFTER it (two halves: three slots each)
    auto store_patch = [&](int buf, int chunk, int j0) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (FAST || affine) {
            asc = ldsAff[chunk * 4 + pq];
            ash = ldsAff[K4 + chunk * 4 + pq];
        }
#pragma unroll
        for (int j = j0; j < j0 + 3; ++j) {
            const int idx = tid + j * W4_NT;
            f32x4 v = praw[j];
            if (FAST) {
                v = v * asc + ash;
                v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]);
            } else {
                if (affine) v = v * asc + ash;
                if (act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
            }
            ldsD[buf * W4_DT + (idx >> 2) * W4_PITCH + (idx & 3)] = spix[j] >= 0 ? v : zero;
        }
    };

    // ---- input transform: 768 items = 16 tiles x 8 channel pairs x 6 transform rows = twelve "virtual waves" (row u, tile-row
    // pair); this wave runs virtual waves wave, wave + 4, wave + 8 -- three passes with wave-uniform (scalar) row coefficients each
    const int tpair = lane & 7;
    const int ttile = (lane >> 5) * 4 + ((lane >> 3) & 3);
    const int tbase2 = (((4 * (ttile >> 2)) * W4_PW + 4 * (ttile & 3)) * W4_PITCH) * 2 + tpair;
    constexpr int PASS_D = 8 * W4_PW * W4_PITCH * 2;                // tile row + 2 = patch row + 8 (float2 units)
    constexpr int PASS_V = 8 * 8;                                   // tile + 8 in V[pos][tile][pair]
    struct PassGeo { int to0, to1, to2, to3, dofs, vofs; float tc0, tc1, tc2, tc3; };
    PassGeo pg[3];
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
        const int v = wave + 4 * ps;
        const int tu = v % 6, pr = v / 6;
        const int tr0 = (tu == 0) ? 0 : 1, tr1 = (tu == 5) ? 3 : 2, tr2 = (tu == 0) ? 4 : ((tu == 5) ? 5 : 3), tr3 = 4;
        pg[ps].tc0 = (tu == 0) ? 4.f : (tu == 1 ? -4.f : (tu == 2 ? 4.f : (tu == 3 ? -2.f : (tu == 4 ? 2.f : 4.f))));
        pg[ps].tc1 = (tu == 0 || tu == 5) ? -5.f : ((tu == 1 || tu == 2) ? -4.f : -1.f);
        pg[ps].tc2 = (tu == 0 || tu == 5) ? 1.f : (tu == 1 ? 1.f : (tu == 2 ? -1.f : (tu == 3 ? 2.f : -2.f)));
        pg[ps].tc3 = (tu == 0 || tu == 5) ? 0.f : 1.f;
        pg[ps].to0 = tr0 * W4_PW * W4_PITCH * 2; pg[ps].to1 = tr1 * W4_PW * W4_PITCH * 2;
        pg[ps].to2 = tr2 * W4_PW * W4_PITCH * 2; pg[ps].to3 = tr3 * W4_PW * W4_PITCH * 2;
        pg[ps].dofs = pr * PASS_D;
        pg[ps].vofs = ((tu * 6) * 16 + ttile) * 8 + tpair + pr * PASS_V;
    }
    f32x2 tld[24];                                                  // the 24 operands of the pass in flight (TSPLIT)
    auto t_issue = [&](int pbuf, int ps) {
        const f32x2 *D = reinterpret_cast<const f32x2 *>(ldsD + pbuf * W4_DT) + tbase2 + pg[ps].dofs;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            tld[4 * j + 0] = D[pg[ps].to0 + j * W4_PITCH * 2];
            tld[4 * j + 1] = D[pg[ps].to1 + j * W4_PITCH * 2];
            tld[4 * j + 2] = D[pg[ps].to2 + j * W4_PITCH * 2];
            tld[4 * j + 3] = D[pg[ps].to3 + j * W4_PITCH * 2];
        }
    };
    auto t_finish = [&](int vbuf, int ps) {
        f32x2 t[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            t[j] = pg[ps].tc0 * tld[4 * j] + pg[ps].tc1 * tld[4 * j + 1] + pg[ps].tc2 * tld[4 * j + 2] + pg[ps].tc3 * tld[4 * j + 3];
        const f32x2 p = t[4] - 4.f * t[2], q = t[3] - 4.f * t[1], r = t[4] - t[2], s = t[3] - t[1];
        f32x2 *V = reinterpret_cast<f32x2 *>(ldsV + vbuf * W4_V) + pg[ps].vofs;
        V[0 * 128] = 4.f * t[0] - 5.f * t[2] + t[4];
        V[1 * 128] = p + q;
        V[2 * 128] = p - q;
        V[3 * 128] = r + 2.f * s;
        V[4 * 128] = r - 2.f * s;
        V[5 * 128] = 4.f * t[1] - 5.f * t[3] + t[5];
    };
    auto transform_all = [&](int pbuf, int vbuf) {
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) { t_issue(pbuf, ps); t_finish(vbuf, ps); }
    };

    // ---- accumulators: all 36 positions x this wave's 2 x 16 channels x 16 tiles
    f32x4 acc0[36], acc1[36];
#pragma unroll
    for (int p = 0; p < 36; ++p) { acc0[p] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[p] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int l15 = lane & 15, kq = lane >> 4;
    const int nw = n0 + wave * 32 + l15;                            // this lane's first output channel; the second is nw + 16
    const __amdgpu_buffer_rsrc_t rU = rsrcw(a.bmat);
    const unsigned xi_bytes = (unsigned)K4 * (unsigned)N * 16u;      // bytes per position of U
    const unsigned ulane = ((unsigned)kq * (unsigned)N + (unsigned)nw) * 16u;
    const int vread = l15 * 4 + kq;

    f32x4 ring0[W4_RING], ring1[W4_RING];
    auto load_b = [&](int chunk, int pos, int slot) {
        const unsigned w = (unsigned)pos * xi_bytes + (unsigned)(chunk * 4) * (unsigned)N * 16u;
        ring0[slot] = bldw(rU, ulane, w);
        ring1[slot] = bldw(rU, ulane + 256u, w);
    };

    // prologue: patch(0) -> LDS -> V(0); patch(1) -> LDS; patch(2) requested
    const int last = nchunks - 1;
    const int c1 = last >= 1 ? 1 : 0, c2 = last >= 2 ? 2 : last;
    f32x4 aff_sc = {0.f, 0.f, 0.f, 0.f}, aff_sh = {0.f, 0.f, 0.f, 0.f};
    const bool aff_slot = (FAST || affine) && tid < K4 && !fold;
    anoddpm::FoldLoads fl;
    if (aff_slot) {
        aff_sc = bldw(rSc, (unsigned)(tid * 16), 0u);
        aff_sh = bldw(rSh, (unsigned)(tid * 16), 0u);
    } else if (fold) {
        fl = anoddpm::fold_affine_request(a, b, tid);
    }
    load_patch(0);
    f32x4 praw0[W4_PJ];
#pragma unroll
    for (int j = 0; j < W4_PJ; ++j) praw0[j] = praw[j];
    load_patch(c1);
#pragma unroll
    for (int g = 0; g < W4_RING; ++g) load_b(0, g, g);
    if (FAST || affine) {
        if (fold) {
            anoddpm::fold_affine_finish(a, fl, tid, a_mode == 1 ? (H >> 1) * (W >> 1) : H * W, reinterpret_cast<double *>(ldsV), ldsAff);
        } else if (aff_slot) {
            ldsAff[tid] = aff_sc;
            ldsAff[K4 + tid] = aff_sh;
        }
        __syncthreads();
    }
    {
        f32x4 keep[W4_PJ];
#pragma unroll
        for (int j = 0; j < W4_PJ; ++j) { keep[j] = praw[j]; praw[j] = praw0[j]; }
        store_patch(0, 0, 0);
        store_patch(0, 0, 3);
#pragma unroll
        for (int j = 0; j < W4_PJ; ++j) praw[j] = keep[j];
    }
    __syncthreads();
    transform_all(0, 0);
    store_patch(1, c1, 0);
    store_patch(1, c1, 3);
    load_patch(c2);
    __syncthreads();

    // One step per 16-channel chunk c (compile-time T / S / L / R flags as in wino43r_kernel):
    //   T  V(c+1) <- patch(c+1): as a block before position 0, or (TSPLIT) pass k issued at position 3k, consumed at 3k + 2
    //   S  patch(c+2) -> LDS: slots 0..2 at position 12, slots 3..5 at position 16     L  request patch(c+3) at position 27
    //   R  (last chunk) the first residual tile requested from position 30 on
    auto step = [&](const int chunk, auto doT, auto doS, auto doL, auto doR, auto &&res_prefetch) {
        constexpr bool T = decltype(doT)::value;
        if (T && !TSPLIT) transform_all((chunk + 1) & 1, (chunk + 1) & 1);
        const f32x4 *V = ldsV + (chunk & 1) * W4_V + vread;
        f32x4 av[3];                                                // A fragments: two positions ahead of the MFMAs
        av[0] = V[0];
        av[1] = V[64];
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            if (T && TSPLIT) {
                if (p == 0 || p == 3 || p == 6) t_issue((chunk + 1) & 1, p / 3);
                if (p == 2 || p == 5 || p == 8) t_finish((chunk + 1) & 1, p / 3);
            }
            if (p == 12 && decltype(doS)::value) store_patch(chunk & 1, chunk + 2, 0);
            if (p == 16 && decltype(doS)::value) store_patch(chunk & 1, chunk + 2, 3);
            if (p == 27 && decltype(doL)::value) load_patch(chunk + 3);
            if (p == 30 && decltype(doR)::value) res_prefetch();
            const f32x4 a_cur = av[p % 3];
            if (p + 2 < 36) av[(p + 2) % 3] = V[(p + 2) * 64];
            const f32x4 b0 = ring0[p % W4_RING], b1 = ring1[p % W4_RING];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc0[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[kk], b0[kk], acc0[p], 0, 0, 0);
                acc1[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[kk], b1[kk], acc1[p], 0, 0, 0);
            }
            if (p + W4_RING < 36)               load_b(chunk, p + W4_RING, p % W4_RING);
            else if (!decltype(doR)::value)     load_b(chunk + 1, p + W4_RING - 36, p % W4_RING);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!decltype(doR)::value) __syncthreads();                 // publishes V(c+1) and patch(c+2); retires V(c)
    };
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};
    auto nothing = []() {};
    int chunk = 0;
    for (; chunk + 3 <= last; ++chunk) step(chunk, YES, YES, YES, NO, nothing);

    // ---- epilogue, in registers: lane = (channels nw and nw + 16, tiles kq*4 .. kq*4+3); tile r of the lane sits in component r
    const float *TE = a.temb ? a.temb + (int64_t)b * a.temb_ld : nullptr;
    const __amdgpu_buffer_rsrc_t rO = rsrcw(a.out + (int64_t)b * a.o_bs);
    const __amdgpu_buffer_rsrc_t rR = rsrcw(a.res ? a.res + (int64_t)b * a.r_bs : a.out);
    const bool has_res = a.res != nullptr;
    const unsigned uW = (unsigned)W, o_ld = (unsigned)a.out_ld, r_ld = (unsigned)a.res_ld;
    const float alpha = a.alpha;
    float add[2] = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (a.bias) add[g] += a.bias[nw + 16 * g];
        if (TE) add[g] += TE[nw + 16 * g];
    }
    const unsigned pix0 = (unsigned)(y0 + kq * 4) * uW + (unsigned)x0;
    const unsigned vo = (pix0 * o_ld + (unsigned)nw) * 4u, vr = (pix0 * r_ld + (unsigned)nw) * 4u;
    const bool res_up = a.res_mode == 1;
    const unsigned hW = uW >> 1;
    const unsigned vrh = ((((unsigned)(y0 + kq * 4) >> 1) * hW + ((unsigned)x0 >> 1)) * r_ld + (unsigned)nw) * 4u;
    auto load_res = [&](int it, float (&rv)[16]) {                  // it = g * 4 + r
        const int g = it >> 2, r = it & 3;
#pragma unroll
        for (int i = 0; i < 16; ++i) rv[i] = 0.f;
        if (has_res && res_up) {
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rR, (int)(vrh + 64u * g), (int)((((unsigned)(r * 2) + (unsigned)i2 * hW + (unsigned)j2) * 4u) * r_ld), 0));
                    rv[(2 * i2) * 4 + 2 * j2] = v; rv[(2 * i2) * 4 + 2 * j2 + 1] = v;
                    rv[(2 * i2 + 1) * 4 + 2 * j2] = v; rv[(2 * i2 + 1) * 4 + 2 * j2 + 1] = v;
                }
        } else if (has_res) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rv[i * 4 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rR, (int)(vr + 64u * g), (int)((((unsigned)(r * 4) + (unsigned)i * uW + (unsigned)j) * 4u) * r_ld), 0));
        }
    };
    float rv[2][16];
    if (last >= 2) step(last - 2, YES, YES, NO, NO, nothing);
    if (last >= 1) step(last - 1, YES, NO, NO, NO, nothing);
    step(last, NO, NO, NO, YES, [&]() { load_res(0, rv[0]); });

#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float cs = 0.f, cq = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int it = g * 4 + r;
            if (it + 1 < 8) load_res(it + 1, rv[(it + 1) & 1]);
            float y[4][6];
#pragma unroll
            for (int v = 0; v < 6; ++v) {
                float mu[6], o[4];
#pragma unroll
                for (int u = 0; u < 6; ++u) mu[u] = g == 0 ? acc0[u * 6 + v][r] : acc1[u * 6 + v][r];
                at6w(mu, o);
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i][v] = o[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float o4[4];
                at6w(y[i], o4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned so = ((unsigned)(r * 4) + (unsigned)i * uW + (unsigned)j) * 4u;
                    const float v = alpha * o4[j] + add[g] + rv[it & 1][i * 4 + j];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rO, (int)(vo + 64u * g), (int)(so * o_ld), 0);
                    cs += v;
                    cq += v * v;
                }
            }
        }
        if (a.stats || a.stats_csum) {
            cs += __shfl_xor(cs, 16);
            cq += __shfl_xor(cq, 16);
            cs += __shfl_xor(cs, 32);
            cq += __shfl_xor(cq, 32);
            if (kq == 0) {
                if (a.stats_csum) {
                    anoddpm::csum_atomic_add(a.stats_csum, b, N, nw + 16 * g, cs, cq);
                } else {
                    float *st = a.stats + (((int64_t)b * gridDim.x + blockIdx.x) * N + nw + 16 * g) * 2;
                    st[0] = cs;
                    st[1] = cq;
                }
            }
        }
    }
}

}  // namespace

namespace anoddpm {

// Called by launch_winograd43 instead of launch_winograd43r when ANODDPM_DEBUG5 = 4 / 5 (block / interleaved transform) selects
// the one-wave-per-SIMD kernel; arguments validated there.  No split-K form.
int launch_winograd43w(const anoddpm_igemm_args *a, hipStream_t s, int tsplit)
{
    dim3 grid((unsigned)((a->H / 16) * (a->W / 16)), (unsigned)(a->N / 128), (unsigned)a->B);
    const bool fast = (a->gn_scale || a->fold_gamma) && a->act;
    ANODDPM_REQUIRE(a->ksplit == 1, "winograd43w: no split-K form");
    ANODDPM_REQUIRE(!(a->gn_scale || a->fold_gamma) || a->c0 + a->c1 <= W4_KMAX, "winograd43w: GroupNorm affine table holds %d input channels", W4_KMAX);
    const bool r4 = (tsplit & 2) != 0;                              // bit 1: four positions in flight instead of six
    const bool ts = (tsplit & 1) != 0;
#define W4_LAUNCH(F, T, R) hipLaunchKernelGGL((wino43w_kernel<F, T, R>), grid, dim3(W4_NT), 0, s, *a)
    if (fast) { if (ts) { if (r4) W4_LAUNCH(true, true, 4); else W4_LAUNCH(true, true, 6); } else { if (r4) W4_LAUNCH(true, false, 4); else W4_LAUNCH(true, false, 6); } }
    else      { if (ts) { if (r4) W4_LAUNCH(false, true, 4); else W4_LAUNCH(false, true, 6); } else { if (r4) W4_LAUNCH(false, false, 4); else W4_LAUNCH(false, false, 6); } }
#undef W4_LAUNCH
    return check_launch("winograd43w");
}

}  // namespace anoddpm
