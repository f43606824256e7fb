// Fused QKVAttention core (UNet.py:137-153, QKVAttentionLegacy channel order), gfx950:
//     weight = softmax((q * s)^T (k * s)),  a = weight v,   s = ch^-1/4   (alpha = ch^-1/2 on the product)
// in ONE launch per attention block instead of QK^T (split-K + reduce) -> softmax_rows -> AV (split-K + reduce) with the
// [B*heads, L, L] score matrix going through HBM twice.
//
// Workgroup = 16 query rows of one (image, head); 8 waves.
//   1. the 16 x ch query block is staged in LDS once;
//   2. wave w forms the 16 x 16 score tiles of key tiles w, w+8, ... with v_mfma_f32_16x16x4_f32 (A = q from LDS, one
//      ds_read_b128 per four MFMAs; B = k straight from L2, 16 bytes per lane, double-buffered in registers; the k index is
//      permuted identically on both operands) and writes them, scaled, to the LDS score block S[16][L];
//   3. softmax over the rows of S in LDS (32 threads per row, exact: the whole row is resident, L <= 1024);
//   4. wave w forms the output channel tiles w*ch/128 ... of  P v  (A = P from LDS, B = v from L2, double-buffered) and stores
//      them to a[b][row][head*ch + c].
// Optionally P is also written to HBM (the training plan's backward reads it).  fp32 throughout, fixed summation order.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AT_LMAX = 1024;
constexpr int AT_WAVES = 8;                                         // 16 query rows x (image, head) is all the parallelism there is:
constexpr int AT_NT = AT_WAVES * 64;                                // 8 waves split the key tiles / output channel tiles of a block

template <int CHQ>                                                  // ch / 16
__global__ __launch_bounds__(AT_NT) void attention_kernel(const anoddpm_attention_args a)
{
    constexpr int CH = CHQ * 16;
    constexpr int QP = CH + 4;                                      // query row pitch (floats)
    constexpr int TPW = CHQ >= AT_WAVES ? CHQ / AT_WAVES : 1;       // output channel tiles per wave
    __shared__ __attribute__((aligned(16))) float lds[16 * (AT_LMAX + 4) + 16 * QP];
    const int L = a.L, SP = L + 4;
    float *S = lds, *Qs = lds + 16 * (AT_LMAX + 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * 16, hd = blockIdx.y, b = blockIdx.z;
    const int C3 = 3 * a.heads * CH;
    const float *__restrict__ qkv = a.qkv + (int64_t)b * L * C3 + hd * 3 * CH;      // q of this head; k at +CH, v at +2CH
    const int nkt = L >> 4;

    // ---- 1. query block -> LDS ----
    for (int idx = tid; idx < 16 * (CH / 4); idx += AT_NT) {
        const int r = idx / (CH / 4), c4 = idx - r * (CH / 4);
        *reinterpret_cast<f32x4 *>(Qs + r * QP + c4 * 4) = *reinterpret_cast<const f32x4 *>(qkv + (int64_t)(row0 + r) * C3 + c4 * 4);
    }
    // ---- 2. scores ----
    constexpr bool DB = CHQ < 32;                                   // ch = 512: two key buffers would not fit the 256-VGPR budget
    f32x4 kb[DB ? 2 : 1][CHQ];
    auto load_k = [&](f32x4 (&dst)[CHQ], int jt) {
        const float *kp = qkv + (int64_t)(jt * 16 + l16) * C3 + CH + kq * 4;
#pragma unroll
        for (int g = 0; g < CHQ; ++g) dst[g] = *reinterpret_cast<const f32x4 *>(kp + g * 16);
    };
    auto score_tile = [&](const f32x4 (&kk)[CHQ], int jt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < CHQ; ++g) {
            const f32x4 qa = *reinterpret_cast<const f32x4 *>(Qs + l16 * QP + g * 16 + kq * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[e], kk[g][e], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(kq * 4 + r) * SP + jt * 16 + l16] = acc[r] * a.scale;
    };
    if (wave < nkt) load_k(kb[0], wave);
    __syncthreads();                                                // Q staged
    if constexpr (DB) {
        for (int jt = wave; jt < nkt; jt += 2 * AT_WAVES) {
            if (jt + AT_WAVES < nkt) load_k(kb[1], jt + AT_WAVES);
            score_tile(kb[0], jt);
            if (jt + AT_WAVES < nkt) {
                if (jt + 2 * AT_WAVES < nkt) load_k(kb[0], jt + 2 * AT_WAVES);
                score_tile(kb[1], jt + AT_WAVES);
            }
        }
    } else {
        for (int jt = wave; jt < nkt; jt += AT_WAVES) {
            if (jt != wave) load_k(kb[0], jt);
            score_tile(kb[0], jt);
        }
    }
    // first v operands go out before the softmax so that their latency hides behind it
    const int ct0 = wave * TPW;
    const bool pv_active = ct0 < CHQ;
    float vb[2][TPW][4];
    auto load_v = [&](float (&dst)[TPW][4], int kg) {
        const float *vp = qkv + (int64_t)(kg * 16 + kq * 4) * C3 + 2 * CH + ct0 * 16 + l16;
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[t][e] = vp[(int64_t)e * C3 + t * 16];
    };
    if (pv_active) load_v(vb[0], 0);
    __syncthreads();                                                // S complete

    // ---- 3. softmax over the 16 rows: 32 threads per row ----
    {
        constexpr int TPR = AT_NT / 16;                             // threads per row (32)
        const int r = tid / TPR, sub = tid % TPR;
        float *srow = S + r * SP;
        float m = -INFINITY;
        for (int j = sub; j < L; j += TPR) m = fmaxf(m, srow[j]);
#pragma unroll
        for (int o = TPR / 2; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, TPR));
        float sum = 0.f;
        for (int j = sub; j < L; j += TPR) {
            const float e = __expf(srow[j] - m);
            srow[j] = e;
            sum += e;
        }
#pragma unroll
        for (int o = TPR / 2; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, TPR);
        const float inv = 1.0f / sum;
        float *prow = a.probs ? a.probs + (((int64_t)b * a.heads + hd) * L + row0 + r) * L : nullptr;
        for (int j = sub; j < L; j += TPR) {
            const float p = srow[j] * inv;
            srow[j] = p;
            if (prow) prow[j] = p;
        }
    }
    __syncthreads();

    // ---- 4. a = P v ----
    if (!pv_active) return;
    f32x4 oacc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) oacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto pv_group = [&](const float (&vv)[TPW][4], int kg) {
        const f32x4 pa = *reinterpret_cast<const f32x4 *>(S + l16 * SP + kg * 16 + kq * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < TPW; ++t) oacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[e], vv[t][e], oacc[t], 0, 0, 0);
    };
    for (int kg = 0; kg < nkt; kg += 2) {
        if (kg + 1 < nkt) load_v(vb[1], kg + 1);
        pv_group(vb[0], kg);
        if (kg + 1 < nkt) {
            if (kg + 2 < nkt) load_v(vb[0], kg + 2);
            pv_group(vb[1], kg + 1);
        }
    }
    float *__restrict__ O = a.out + ((int64_t)b * L + row0 + kq * 4) * (a.heads * CH) + hd * CH + ct0 * 16 + l16;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) O[(int64_t)r * (a.heads * CH) + t * 16] = oacc[t][r];
}

}  // namespace

extern "C" int anoddpm_attention(const anoddpm_attention_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->qkv && a->out, "attention: null pointer");
    ANODDPM_REQUIRE(a->B >= 1 && a->heads >= 1 && a->B <= 65535 && a->heads <= 65535, "attention: bad batch / head count");
    ANODDPM_REQUIRE(a->L >= 16 && a->L % 16 == 0 && a->L <= AT_LMAX, "attention: L must be a multiple of 16 in [16, 1024]");
    ANODDPM_REQUIRE(a->ch >= 16 && a->ch <= 512 && (a->ch & (a->ch - 1)) == 0, "attention: head width must be a power of two in [16, 512]");
    ANODDPM_REQUIRE(((uintptr_t)a->qkv & 15) == 0, "attention: qkv must be 16-byte aligned");
    const dim3 grid((unsigned)(a->L / 16), (unsigned)a->heads, (unsigned)a->B);
    hipStream_t s = anoddpm::as_stream(stream);
    switch (a->ch / 16) {
        case 1: hipLaunchKernelGGL((attention_kernel<1>), grid, dim3(AT_NT), 0, s, *a); break;
        case 2: hipLaunchKernelGGL((attention_kernel<2>), grid, dim3(AT_NT), 0, s, *a); break;
        case 4: hipLaunchKernelGGL((attention_kernel<4>), grid, dim3(AT_NT), 0, s, *a); break;
        case 8: hipLaunchKernelGGL((attention_kernel<8>), grid, dim3(AT_NT), 0, s, *a); break;
        case 16: hipLaunchKernelGGL((attention_kernel<16>), grid, dim3(AT_NT), 0, s, *a); break;
        default: hipLaunchKernelGGL((attention_kernel<32>), grid, dim3(AT_NT), 0, s, *a); break;
    }
    return anoddpm::check_launch("attention");
}
