// Streaming 1x1 convolution for large maps (cfg == 4 of anoddpm_igemm), gfx950.
//
// Replaces: nn.Conv2d(in, out, 1) of ResBlock.skip_connection (UNet.py:200, applied at UNet.py:208) on the 256^2 / 128^2 maps,
// where the operand is the raw (un-normalised) block input -- possibly the torch.cat of UNet.py:402 -- and the result is the
// residual of the block's second 3x3 convolution; in training also its data gradient (the same contraction on W^T).
//
// Shape: out[M, N] = x[M, K] W[K, N] with M = B*H*W in the hundreds of thousands, K <= 512, N = 64..256: the weight matrix is
// tiny (128 KB at 256 -> 128) and the activation is read exactly once.  The generic direct kernel (igemm.hip) stages both
// operands through LDS per 128-pixel tile and pays a prologue + epilogue per tile: 58 % MFMA-busy on this shape.  Here
//   * the weights of the workgroup's output-channel block live in LDS for the whole launch ([K/4][NB][4] floats, the packed
//     layout of _pack_conv with the row pitch padded by two float4 so the two k-halves of a wave hit different banks);
//   * every wave owns 32-pixel tiles end to end (no barrier after the weight load): its A operand goes global -> registers
//     (lane = pixel x k-half, four 16-byte loads per 32-channel chunk, a four-deep ring that runs on across tile boundaries),
//     its B operand is one ds_read_b128 per four MFMAs;
//   * v_mfma_f32_32x32x2_f32 with the k index permuted identically on both operands (channel 4q+e of half h is k-step (q, e)).
// Work split: 256 workgroups (one per CU, 8 waves) x output-channel blocks; wave w of workgroup g takes tiles g*8+w, +stride, ...
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PW_WAVES = 8;
constexpr int PW_NT = PW_WAVES * 64;

// DBG (timing ablations only, wrong results; ANODDPM_DEBUG3): 1 no output stores, 2 no A requests after the first three, 3 both, 4 no phase shift between the two waves of a SIMD
template <int NB, int KMAX, int DBG = 0>
__global__ __launch_bounds__(PW_NT) void pointwise_stream_kernel(const anoddpm_igemm_args a, const int tiles, const int tiles_per_image)
{
    constexpr int NBP = NB + 2;                                     // row pitch in float4
    constexpr int NTN = NB / 32;
    __shared__ __attribute__((aligned(16))) f32x4 wl[(KMAX / 4) * NBP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int K = a.c0 + a.c1, N = a.N;
    const int n0 = blockIdx.y * NB;
    const int nchunks = K >> 5;

    {   // weights of this channel block: [K/4][N] float4 in global -> [K/4][NBP] in LDS
        const f32x4 *__restrict__ wg = reinterpret_cast<const f32x4 *>(a.bmat);
        const int total = (K >> 2) * NB;
        for (int idx = tid; idx < total; idx += PW_NT) {
            const int q = idx / NB, n = idx - q * NB;
            wl[q * NBP + n] = wg[(int64_t)q * N + n0 + n];
        }
    }
    __syncthreads();

    const int gw = blockIdx.x * PW_WAVES + wave, stride = gridDim.x * PW_WAVES;
    if (gw >= tiles) return;
    const int my_tiles = (tiles - gw + stride - 1) / stride;

    // ---- A operand ring: chunk = 32 channels of the wave's 32 pixels; lane (p, h) holds channels h*16 .. h*16+15 ----
    f32x4 ring[4][4];
    int ld_tile = 0, ld_chunk = 0;                                  // position of the load stream (tile index local to this wave)
    const float *ld_base0 = nullptr, *ld_base1 = nullptr;           // lane's pixel row in source 0 / 1 for the load stream's tile
    auto set_tile = [&](int t) {
        const int tile = gw + (t < my_tiles ? t : my_tiles - 1) * stride;   // past the end: re-read the last tile (harmless)
        const int b = tile / tiles_per_image, pl = (tile - b * tiles_per_image) * 32 + p;
        ld_base0 = a.a0 + (int64_t)b * a.a0_bs + (int64_t)pl * a.a0_ld + h * 16;
        ld_base1 = a.c1 ? a.a1 + (int64_t)b * a.a1_bs + (int64_t)pl * a.a1_ld + h * 16 : ld_base0;
    };
    auto issue = [&](f32x4 (&dst)[4]) {
        if ((DBG & 2) && ld_tile + ld_chunk > 2) return;
        const int c = ld_chunk * 32;                                // wave-uniform
        const float *src = c < a.c0 ? ld_base0 + c : ld_base1 + (c - a.c0);
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const f32x4 *>(src + 4 * i);
        if (++ld_chunk == nchunks) { ld_chunk = 0; set_tile(++ld_tile); }
    };
    set_tile(0);
    issue(ring[0]);
    issue(ring[1]);
    issue(ring[2]);
    // The two waves of a SIMD would otherwise run in lockstep and sit out their epilogues' store latency together (vmcnt
    // retires in order: the first chunks of the next tile wait behind the stores).  Half a tile of phase shift lets one wave's
    // MFMA stream cover the other's drain; a single unstalled wave saturates the pipe, so the shift itself costs nothing.
    if (wave >= PW_WAVES / 2 && my_tiles >= 2 && !(DBG & 4)) {
        for (int i = 0; i < nchunks / 2; ++i) __builtin_amdgcn_s_sleep(127);     // 8128 cycles each; a tile is ~4100 cycles per chunk
    }

    const f32x4 *wrow = wl + h * 4 * NBP + p;                       // + (chunk*8 + i) * NBP + nt*32
    // B operand: one ds_read_b128 per channel-block tile and k-quad, requested one quad (16 MFMAs) ahead -- across chunk and
    // tile boundaries too (the weights do not depend on the tile), so no LDS latency is exposed after the very first quad.
    f32x4 bq[2][NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) bq[0][nt] = wrow[nt * 32];
    for (int t = 0; t < my_tiles; ++t) {
        f32x16 acc[NTN];
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        for (int j0 = 0; j0 < nchunks; j0 += 4) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                issue(ring[(s + 3) & 3]);                           // three chunks ahead
                const int j = j0 + s;
                const f32x4 *wq = wrow + j * 8 * NBP;
                const f32x4 *wn = wrow + (j + 1 == nchunks ? 0 : j + 1) * 8 * NBP;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 *nx = i < 3 ? wq + (i + 1) * NBP : wn;
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt) bq[(i + 1) & 1][nt] = nx[nt * 32];
                    __builtin_amdgcn_sched_barrier(0);             // keep the requests ahead of the MFMAs they overlap
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int nt = 0; nt < NTN; ++nt)
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[i & 1][nt][e], ring[s][i][e], acc[nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- epilogue of the tile: alpha * acc + bias + temb + residual ----
        // The weights are the MFMA's A operand, so D = out^T: lane (p, h) holds pixel p and, per 32-channel tile, the channel
        // quads 8g + 4h .. + 3 (g = 0..3) in acc[4g .. 4g+3] -- 16-byte stores, 16 per tile instead of 64 scalar ones.
        const int tile = gw + t * stride;
        const int b = tile / tiles_per_image, pl = (tile - b * tiles_per_image) * 32 + p;
        float *__restrict__ O = a.out + (int64_t)b * a.o_bs + (int64_t)pl * a.out_ld + n0 + 4 * h;
        const float *R = a.res ? a.res + (int64_t)b * a.r_bs + (int64_t)pl * a.res_ld + n0 + 4 * h : nullptr;
        const float *TE = a.temb ? a.temb + (int64_t)b * a.temb_ld + n0 + 4 * h : nullptr;
        const float *BI = a.bias ? a.bias + n0 + 4 * h : nullptr;
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            f32x4 rv[4], add[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {                           // all residual loads of the sub-tile before its stores
                const int c = nt * 32 + g * 8;
                rv[g] = R ? *reinterpret_cast<const f32x4 *>(R + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                add[g] = BI ? *reinterpret_cast<const f32x4 *>(BI + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (TE) add[g] += *reinterpret_cast<const f32x4 *>(TE + c);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = a.alpha * acc[nt][g * 4 + e] + add[g][e] + rv[g][e];
                if (!(DBG & 1) || v[0] == 12345.678f) *reinterpret_cast<f32x4 *>(O + nt * 32 + g * 8) = v;
            }
        }
    }
}

}  // namespace

namespace anoddpm {

// Called by anoddpm_igemm for cfg == 4 (common arguments already validated there).
int launch_pointwise_stream(const anoddpm_igemm_args *a, hipStream_t s)
{
    const int K = a->c0 + a->c1;
    const int64_t P = (int64_t)a->H * a->W;
    ANODDPM_REQUIRE(a->ks == 1 && a->a_mode == 0 && a->b_mode == 0 && a->heads == 1 && a->ksplit == 1,
                    "pointwise stream: needs an unsplit 1x1 convolution with packed weights");
    ANODDPM_REQUIRE(!a->gn_scale && a->act == 0 && !a->stats, "pointwise stream: no fused GroupNorm / activation / statistics");
    ANODDPM_REQUIRE(K % 128 == 0 && a->c0 % 32 == 0 && K <= 512, "pointwise stream: K must be a multiple of 128, <= 512, sources split at a multiple of 32");
    ANODDPM_REQUIRE(P % 32 == 0 && a->N % 64 == 0, "pointwise stream: H*W must be a multiple of 32 and N of 64");
    const int64_t tiles = a->B * (P / 32);
    ANODDPM_REQUIRE(tiles < ((int64_t)1 << 31), "pointwise stream: too many pixels");
    const bool wide = (a->N % 128 == 0) && K <= 256;
    const int NB = wide ? 128 : 64;
    const int ny = a->N / NB;
    int gx = 256 / ny;                                               // one workgroup per CU in total
    const int64_t need = (tiles + PW_WAVES - 1) / PW_WAVES;
    if (gx > need) gx = (int)need;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)ny);
#ifdef ANODDPM_ABLATE
    const int dbg = anoddpm::g_debug[3];
#endif
    const int tpi = (int)(P / 32);
#define PW_LAUNCH(NB_, KMAX_, DBG_) \
    hipLaunchKernelGGL((pointwise_stream_kernel<NB_, KMAX_, DBG_>), grid, dim3(PW_NT), 0, s, *a, (int)tiles, tpi)
#ifdef ANODDPM_ABLATE           // timing ablations (wrong results): measurement builds only
    if (wide && dbg == 1) PW_LAUNCH(128, 256, 1);
    else if (wide && dbg == 2) PW_LAUNCH(128, 256, 2);
    else if (wide && dbg == 3) PW_LAUNCH(128, 256, 3);
    else if (wide && dbg == 4) PW_LAUNCH(128, 256, 4);
    else
#endif
    if (wide) PW_LAUNCH(128, 256, 0);
    else PW_LAUNCH(64, 512, 0);
#undef PW_LAUNCH
    return check_launch("igemm(pointwise stream)");
}

}  // namespace anoddpm
