// Backward twins of the small / pointwise UNet operators and the per-step weight packing -- everything the training
// step (diffusion_training.py:99-107, loss.backward() of UNet.py:390-406) needs beyond the 3x3 kernels:
//   * pointwise-convolution weight gradient (skip_connection UNet.py:200, to_qkv / proj_out UNet.py:115-117) on the fp32 MFMA
//   * row-softmax backward and square transposes for QKVAttention (UNet.py:137-153)
//   * small-batch linear backward (time MLP UNet.py:271-276, embedding projections UNet.py:185-188)
//   * stem / head convolution backward (UNet.py:280, 384-388)
//   * column-sum folds (bias / embedding gradients) and device-side weight packing
// Reference semantics: torch autograd of the same expressions.  All reductions are two-stage with a fixed fold order
// (deterministic); fp32 arithmetic, fp32 accumulation.
#include "common.h"
#include "pack_items.h"

using anoddpm::silu_f;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float silu_grad(float x)
{
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-x));
    return s * (1.0f + x * (1.0f - s));
}

// ------------------------------------------------------------------------------------------------ pointwise wgrad
// dW[n][k] = sum_p dY[p][n] * A[p][k].  GEMM view: M = k (input channels), N = n, contraction = pixels, so both operands
// are pixel-major in NHWC and a lane's MFMA operand is one dword of an LDS row (32 consecutive channels: conflict-free).
// Workgroup = 4 waves (2 x 2), tile 128 k x 128 n, pixels in chunks of 32 staged through registers -> LDS (double buffer).
constexpr int W1T = 128, W1P = 32, W1L = W1T + 4;

__global__ __launch_bounds__(256, 2) void wgrad1_kernel(const anoddpm_wgrad1_args a, const int nspan)
{
    __shared__ __attribute__((aligned(16))) float lds[2][2][W1P][W1L];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int K = a.c0 + a.c1, N = a.N;
    const int ntn = (N + W1T - 1) / W1T;
    const int ci0 = (blockIdx.x / ntn) * W1T, co0 = (blockIdx.x % ntn) * W1T;
    const int item = blockIdx.y;
    const int b = item / nspan, sp = item % nspan;
    const int p0 = sp * a.span;
    const int p1 = p0 + a.span < a.P ? p0 + a.span : a.P;

    // staging role: channel quad pq (0..31), pixel rows pr, pr + 8, pr + 16, pr + 24 of the chunk
    const int pq = tid & 31, pr = tid >> 5;
    const int cch = ci0 + pq * 4;
    const bool cok = cch < K;
    const bool from0 = cch < a.c0;
    const float *asrc = !cok ? a.a0 : (from0 ? a.a0 + (int64_t)b * a.a0_bs + cch : a.a1 + (int64_t)b * a.a1_bs + (cch - a.c0));
    const int ald = from0 ? a.a0_ld : a.a1_ld;
    f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
    const bool affine = a.gn_scale != nullptr, act = a.act != 0;
    if (affine && cok) {
        asc = *reinterpret_cast<const f32x4 *>(a.gn_scale + (int64_t)b * a.gn_ld + cch);
        ash = *reinterpret_cast<const f32x4 *>(a.gn_shift + (int64_t)b * a.gn_ld + cch);
    }
    const int dch = co0 + pq * 4;
    const bool dok = dch < N;
    const float *dsrc = dok ? a.dy + (int64_t)b * a.dy_bs + dch : a.dy;

    f32x4 areg[4], dreg[4];
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    auto load_chunk = [&](int pc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = pc + pr + 8 * j;
            const bool ok = p < p1;
            const int pp = ok ? p : p0;
            f32x4 v = *reinterpret_cast<const f32x4 *>(asrc + (cok ? (int64_t)pp * ald : 0));
            if (affine) v = v * asc + ash;
            if (act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
            areg[j] = (ok && cok) ? v : zero;
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dsrc + (dok ? (int64_t)pp * a.dy_ld : 0));
            dreg[j] = (ok && dok) ? d : zero;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<f32x4 *>(&lds[buf][0][pr + 8 * j][pq * 4]) = areg[j];
            *reinterpret_cast<f32x4 *>(&lds[buf][1][pr + 8 * j][pq * 4]) = dreg[j];
            csum += dreg[j];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_chunk(p0);
    int buf = 0;
    for (int pc = p0; pc < p1; pc += W1P) {
        store_chunk(buf);
        __syncthreads();
        if (pc + W1P < p1) load_chunk(pc + W1P);                       // in flight behind this chunk's MFMAs
        const float *Ab = &lds[buf][0][h][wm * 64 + l31];
        const float *Db = &lds[buf][1][h][wn * 64 + l31];
#pragma unroll
        for (int kp = 0; kp < W1P / 2; ++kp) {
            const float a0v = Ab[(2 * kp) * W1L], a1v = Ab[(2 * kp) * W1L + 32];
            const float d0v = Db[(2 * kp) * W1L], d1v = Db[(2 * kp) * W1L + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v, d0v, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v, d1v, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, d0v, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, d1v, acc[1][1], 0, 0, 0);
        }
        buf ^= 1;                                                      // the other buffer's readers passed the barrier above
    }

    // partial tile of this work item: ws[item][k][n]; column sums behind them: ws[nitems*K*N + item*N + n]
    float *wsp = a.ws + (int64_t)item * K * N;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int co = co0 + wn * 64 + nt * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (ci < K && co < N) wsp[(int64_t)ci * N + co] = acc[mt][nt][r];
            }
        }
    if (a.dbias && ci0 == 0) {
        __syncthreads();
        float *red = &lds[0][0][0][0];                                 // 256 x 4 floats of scratch
        *reinterpret_cast<f32x4 *>(red + tid * 4) = csum;
        __syncthreads();
        if (tid < W1T) {
            const int qq = tid >> 2, e = tid & 3;
            float s = 0.f;
            for (int k = 0; k < 8; ++k) s += red[(k * 32 + qq) * 4 + e];
            const int64_t nitems = (int64_t)gridDim.y;
            if (co0 + tid < N) a.ws[nitems * K * N + (int64_t)item * N + co0 + tid] = s;
        }
    }
}

__global__ __launch_bounds__(256) void wgrad1_fold_kernel(const anoddpm_wgrad1_args a, const int nitems)
{
    const int K = a.c0 + a.c1, N = a.N;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < (int64_t)K * N) {
        const int co = (int)(idx % N), ci = (int)(idx / N);
        float s = 0.f;
        const float *p = a.ws + (int64_t)ci * N + co;
        const int64_t item = (int64_t)K * N;
        int it = 0;
        for (; it + 16 <= nitems; it += 16) {                       // latency-bound: 16 independent loads in flight, item order kept
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[(it + u) * item];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; it < nitems; ++it) s += p[it * item];
        float *o = a.dw + (int64_t)co * K + ci;
        *o = a.accumulate ? *o + s : s;
    }
    if (a.dbias && idx < N) {
        float s = 0.f;
        const float *cs = a.ws + (int64_t)nitems * K * N;
        for (int it = 0; it < nitems; ++it) s += cs[(int64_t)it * N + idx];
        a.dbias[idx] += s;
    }
}

// The same fold for layers with many work items (the 1x1 skip convolutions on the large maps: 256 items of a 128 KB slab) and
// N % 32 == 0: workgroup = 8 item lanes x 32 output channels of ONE input channel, so the launch has 8x the threads -- and
// loads in flight -- of the one-thread-per-weight kernel, which is parallelism-bound there (128 workgroups x 16 loads per thread
// = 2 MB in flight: 0.5 TB/s).  Order of the sum per weight: each item lane in item order, then the lanes in order -- fixed.
__global__ __launch_bounds__(256) void wgrad1_fold_lanes_kernel(const anoddpm_wgrad1_args a, const int nitems)
{
    __shared__ float part[8][32];
    const int K = a.c0 + a.c1, N = a.N;
    const int tiles_n = N >> 5;
    // row K (one more row of workgroups, launched only with dbias) folds the column sums stored behind the items' tiles the same
    // way: the serial walk `for k < nitems: b += cs[k][co]` by 32 threads of the first row was the kernel's critical path (256
    // dependent-latency loads: 66 us for a 17 us fold)
    const int ci = blockIdx.x / tiles_n, co0 = (blockIdx.x % tiles_n) * 32;
    const bool bias_row = ci == K;
    const int il = threadIdx.x >> 5, ol = threadIdx.x & 31;
    const int64_t item = bias_row ? (int64_t)N : (int64_t)K * N;
    const float *p = (bias_row ? a.ws + (int64_t)nitems * K * N : a.ws + (int64_t)ci * N) + co0 + ol;
    float s = 0.f;
    int it = il;
    for (; it + 8 * 15 < nitems; it += 8 * 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = __builtin_nontemporal_load(p + (int64_t)(it + 8 * u) * item);
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; it < nitems; it += 8) s += p[(int64_t)it * item];
    part[il][ol] = s;
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 8; ++l) t += part[l][threadIdx.x];
        if (bias_row) {
            a.dbias[co0 + threadIdx.x] += t;
        } else {
            float *dst = a.dw + (int64_t)(co0 + threadIdx.x) * K + ci;
            *dst = a.accumulate ? *dst + t : t;
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight packing
__global__ __launch_bounds__(256) void pack_pointwise_kernel(const anoddpm_pack_args a)
{
    anoddpm::pack_pointwise_item(a, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

__global__ __launch_bounds__(256) void pack_small_conv_kernel(const anoddpm_pack_args a)
{
    anoddpm::pack_small_conv_item(a, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// Every pack job of a model in ONE launch (the training forward re-packs ~270 weights per step: launch-latency bound as separate
// kernels).  jobs / block0 live in device memory; a block finds its job by bisection over the prefix sums of the jobs' block counts.
__global__ __launch_bounds__(256) void pack_batch_kernel(const anoddpm_pack_batch_args b)
{
    int lo = 0, hi = b.njobs;                                       // block0[lo] <= blockIdx.x < block0[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (b.block0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const anoddpm_pack_args a = b.jobs[lo];
    const int64_t idx = (int64_t)((int)blockIdx.x - b.block0[lo]) * 256 + threadIdx.x;
    if (a.kind <= 1) anoddpm::pack_conv3x3_item(a.w, a.out, a.N, a.K, a.kind, a.bwd, idx);
    else if (a.kind == 5) anoddpm::pack_conv3x3_item(a.w, a.out, a.N, a.K, 2, a.bwd, idx);
    else if (a.kind == 2) anoddpm::pack_pointwise_item(a, idx);
    else if (a.kind == 3) anoddpm::pack_small_conv_item(a, idx);
    else if (idx < (int64_t)a.N * a.K) a.out[idx] = a.w[idx];
}

__global__ __launch_bounds__(256) void copy_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = in[i];
}

// ------------------------------------------------------------------------------------------------ attention pieces
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float *__restrict__ p, float *dp, int64_t rows, int L)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *pr = p + row * L;
    float *dr = dp + row * L;
    float dot = 0.f;
    for (int i = lane; i < L; i += 64) dot += pr[i] * dr[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    for (int i = lane; i < L; i += 64) dr[i] = pr[i] * (dr[i] - dot);
}

__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int L)
{
    __shared__ float tile[32][33];
    const int z = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
    const float *src = in + (int64_t)z * L * L;
    float *dst = out + (int64_t)z * L * L;
    const int x = blockIdx.x * 32 + tx;
    for (int j = ty; j < 32; j += 8) {
        const int y = blockIdx.y * 32 + j;
        if (x < L && y < L) tile[j][tx] = src[(int64_t)y * L + x];
    }
    __syncthreads();
    const int ox = blockIdx.y * 32 + tx;
    for (int j = ty; j < 32; j += 8) {
        const int oy = blockIdx.x * 32 + j;
        if (ox < L && oy < L) dst[(int64_t)oy * L + ox] = tile[tx][j];
    }
}

// ------------------------------------------------------------------------------------------------ small linear
// dw: thread = (n, k quad); dx: thread = (b, k); db: thread = n.  B <= 16.
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const anoddpm_linear_bwd_args a)
{
    const int K4 = a.K >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.N * K4) return;
    const int k4 = (int)(idx % K4), n = (int)(idx / K4);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    float sb = 0.f;
    for (int b = 0; b < a.B; ++b) {
        f32x4 x = *reinterpret_cast<const f32x4 *>(a.x + (int64_t)b * a.K + k4 * 4);
        if (a.act_in) { x[0] = silu_f(x[0]); x[1] = silu_f(x[1]); x[2] = silu_f(x[2]); x[3] = silu_f(x[3]); }
        const float d = a.dy[(int64_t)b * a.N + n];
        s += x * d;
        sb += d;
    }
    f32x4 *o = reinterpret_cast<f32x4 *>(a.dw + (int64_t)n * a.K + k4 * 4);
    *o = a.acc_w ? *o + s : s;
    if (k4 == 0 && a.db) a.db[n] = a.acc_w ? a.db[n] + sb : sb;
}

// dx: grid (K/64) blocks of 1024 threads: lane = k inside a 64-wide chunk, wave w takes n = w, w + 16, ... for ALL batch rows
// (W is read once, coalesced); the 16 partial sums per (b, k) are folded through LDS in a fixed order.
template <int NB>
__global__ __launch_bounds__(1024) void linear_bwd_x_kernel(const anoddpm_linear_bwd_args a)
{
    __shared__ float red[16][NB][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    if (k < a.K)
#pragma unroll 8
        for (int n = wave; n < a.N; n += 16) {                         // 8 independent weight loads in flight
            const float w = a.w[(int64_t)n * a.K + k];
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (b < a.B) acc[b] += a.dy[(int64_t)b * a.N + n] * w;
        }
#pragma unroll
    for (int b = 0; b < NB; ++b) red[wave][b][lane] = acc[b];
    __syncthreads();
    for (int i = threadIdx.x; i < a.B * 64; i += 1024) {
        const int b = i >> 6, l = i & 63;
        const int kk = blockIdx.x * 64 + l;
        if (kk >= a.K) continue;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += red[w][b][l];
        const int64_t idx = (int64_t)b * a.K + kk;
        if (a.act_in) s *= silu_grad(a.x[idx]);
        a.dx[idx] = a.acc_x ? a.dx[idx] + s : s;
    }
}

// Batched form for the 42 per-block embedding projections (UNet.py:185-188): they share the input silu(temb) and were one launch in
// the forward; their backward as separate launches was 86 launches of 8-workgroup kernels (1.5 ms per config-3 step).
// jobs: DEVICE array of anoddpm_linear_bwd_args (w, dy, dw, db, N per job; x / B / K / act_in / acc_w taken from the batch header).
__global__ __launch_bounds__(256) void linear_bwd_w_batch_kernel(const anoddpm_linear_bwd_batch_args h)
{
    anoddpm_linear_bwd_args a = h.jobs[blockIdx.y];
    a.x = h.x; a.B = h.B; a.K = h.K; a.act_in = h.act_in; a.acc_w = h.acc_w;
    const int K4 = a.K >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.N * K4) return;
    const int k4 = (int)(idx % K4), n = (int)(idx / K4);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    float sb = 0.f;
    for (int b = 0; b < a.B; ++b) {
        f32x4 x = *reinterpret_cast<const f32x4 *>(a.x + (int64_t)b * a.K + k4 * 4);
        if (a.act_in) { x[0] = silu_f(x[0]); x[1] = silu_f(x[1]); x[2] = silu_f(x[2]); x[3] = silu_f(x[3]); }
        const float d = a.dy[(int64_t)b * a.N + n];
        s += x * d;
        sb += d;
    }
    f32x4 *o = reinterpret_cast<f32x4 *>(a.dw + (int64_t)n * a.K + k4 * 4);
    *o = a.acc_w ? *o + s : s;
    if (k4 == 0 && a.db) a.db[n] = a.acc_w ? a.db[n] + sb : sb;
}

// partial dx of one job: ws[job][b][k] = sum_n dy[b][n] w[n][k]   (grid (K/64, njobs), 1024 threads; as linear_bwd_x_kernel)
template <int NB>
__global__ __launch_bounds__(1024) void linear_bwd_x_batch_kernel(const anoddpm_linear_bwd_batch_args h)
{
    __shared__ float red[16][NB][64];
    const anoddpm_linear_bwd_args a = h.jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    if (k < h.K)
#pragma unroll 8
        for (int n = wave; n < a.N; n += 16) {
            const float w = a.w[(int64_t)n * h.K + k];
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (b < h.B) acc[b] += a.dy[(int64_t)b * a.N + n] * w;
        }
#pragma unroll
    for (int b = 0; b < NB; ++b) red[wave][b][lane] = acc[b];
    __syncthreads();
    for (int i = threadIdx.x; i < h.B * 64; i += 1024) {
        const int b = i >> 6, l = i & 63;
        const int kk = blockIdx.x * 64 + l;
        if (kk >= h.K) continue;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += red[w][b][l];
        h.ws[((int64_t)blockIdx.y * h.B + b) * h.K + kk] = s;
    }
}

// dx[b][k] (+)= act_in'(x) * sum over the jobs (fixed order) of their partials
__global__ __launch_bounds__(256) void linear_bwd_x_fold_kernel(const anoddpm_linear_bwd_batch_args h)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t bk = (int64_t)h.B * h.K;
    if (idx >= bk) return;
    float s = 0.f;
    for (int j = 0; j < h.njobs; ++j) s += h.ws[(int64_t)j * bk + idx];
    if (h.act_in) s *= silu_grad(h.x[idx]);
    h.dx[idx] = h.acc_x ? h.dx[idx] + s : s;
}

// ------------------------------------------------------------------------------------------------ column-sum fold
// grid (ceil(N/64)): 64 channels x 16 item lanes, eight rows in flight per thread (the Winograd-domain weight gradient emits one
// row per 16x8 patch: 512 per image at 256^2); the images are walked in order, so the block also owns the batch sum = bias gradient.
__global__ __launch_bounds__(1024) void colsum_fold_kernel(const anoddpm_colsum_fold_args a)
{
    __shared__ float red[16][64];
    const int l = threadIdx.x & 63, il = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + l;
    float tot = 0.f;
    for (int b = 0; b < a.B; ++b) {
        float s = 0.f;
        if (n < a.N) {
            const float *p = a.colsum + ((int64_t)b * a.ipb) * a.N + n;
            int i = il;
            for (; i + 7 * 16 < a.ipb; i += 8 * 16) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(i + u * 16) * a.N];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; i < a.ipb; i += 16) s += p[(int64_t)i * a.N];
        }
        red[il][l] = s;
        __syncthreads();
        if (il == 0 && n < a.N) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][l];
            a.dimg[(int64_t)b * a.N + n] = t;
            tot += t;
        }
        __syncthreads();
    }
    if (il == 0 && n < a.N && a.dbias) a.dbias[n] += tot;
}

// ------------------------------------------------------------------------------------------------ stem backward
// grid (nblk, Cin): block = 1024 pixels of one image and one input channel; thread = (channel quad, pixel lane).
constexpr int STEM_PIX = 1024;

__global__ __launch_bounds__(256) void stem_bwd_w_kernel(const anoddpm_stem_bwd_args a, const int bpi)
{
    __shared__ float red[256 * 4];
    const int Q = a.Cout >> 2;                       // channel quads (<= 64)
    const int PL = 256 / Q;                          // pixel lanes
    const int q = threadIdx.x % Q, pl = threadIdx.x / Q;
    const int ci = blockIdx.y;
    const int b = blockIdx.x / bpi, chunk = blockIdx.x % bpi;
    const int P = a.H * a.W;
    const int p0 = chunk * STEM_PIX, p1 = (p0 + STEM_PIX < P) ? p0 + STEM_PIX : P;
    const float *plane = a.x + ((int64_t)b * a.Cin + ci) * P;
    f32x4 acc[9], sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = sb;
    if (pl < PL)
        for (int p = p0 + pl; p < p1; p += PL) {
            const f32x4 d = *reinterpret_cast<const f32x4 *>(a.dy + ((int64_t)b * P + p) * a.Cout + q * 4);
            const int y = p / a.W, x = p % a.W;
            sb += d;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                const float v = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? plane[(int64_t)yy * a.W + xx] : 0.f;
                acc[t] += d * v;
            }
        }
    // reduce over the pixel lanes (fixed order), one tap at a time; slot 9 = bias sums
    float *out = a.ws + ((int64_t)blockIdx.x * (a.Cin * 9 + 1)) * a.Cout;
    for (int t = 0; t < 10; ++t) {
        __syncthreads();
        const f32x4 v = t < 9 ? acc[t] : sb;
        if (pl < PL) *reinterpret_cast<f32x4 *>(red + threadIdx.x * 4) = v;
        __syncthreads();
        if (threadIdx.x < a.Cout && (t < 9 || ci == 0)) {
            const int c = threadIdx.x;
            float s = 0.f;
            for (int k = 0; k < PL; ++k) s += red[(k * Q + (c >> 2)) * 4 + (c & 3)];
            out[(t < 9 ? (int64_t)(ci * 9 + t) : (int64_t)a.Cin * 9) * a.Cout + c] = s;
        }
    }
}

__device__ __forceinline__ float fold_rows8(const float *p, int64_t stride, int nblk, float (*part)[32]);

__global__ __launch_bounds__(256) void stem_bwd_fold_kernel(const anoddpm_stem_bwd_args a, const int nblk)
{
    __shared__ float part[8][32];
    const int rows = a.Cin * 9 + 1;
    const int idx = blockIdx.x * 32 + (threadIdx.x & 31);
    const int idc = idx < rows * a.Cout ? idx : rows * a.Cout - 1;
    const int c = idc % a.Cout, r = idc / a.Cout;
    const float s = fold_rows8(a.ws + (int64_t)r * a.Cout + c, (int64_t)rows * a.Cout, nblk, part);
    if (threadIdx.x >= 32 || idx >= rows * a.Cout) return;
    if (r < a.Cin * 9) a.dw[((int64_t)c * a.Cin + r / 9) * 9 + r % 9] += s;      // OIHW
    else a.db[c] += s;
}

__global__ __launch_bounds__(256) void stem_bwd_x_kernel(const anoddpm_stem_bwd_args a)
{
    // dx[b][ci][p] = sum_{tap,co} w[co][ci][tap] * dy[b][p - off(tap)][co]; test-only path (the network input rarely needs a gradient)
    const int P = a.H * a.W;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.B * a.Cin * P) return;
    const int p = (int)(idx % P), ci = (int)((idx / P) % a.Cin), b = (int)(idx / ((int64_t)P * a.Cin));
    const int y = p / a.W, x = p % a.W;
    float s = 0.f;
    for (int t = 0; t < 9; ++t) {
        const int yy = y - (t / 3 - 1), xx = x - (t % 3 - 1);
        if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
        const float *d = a.dy + ((int64_t)b * P + (int64_t)yy * a.W + xx) * a.Cout;
        for (int co = 0; co < a.Cout; ++co) s += a.w[((int64_t)co * a.Cin + ci) * 9 + t] * d[co];
    }
    a.dx[idx] = s;
}

// ------------------------------------------------------------------------------------------------ head backward
// da: thread = (pixel, channel quad), weights in LDS as [o][tap][C].
template <int COUT>
__global__ __launch_bounds__(256) void head_bwd_da_kernel(const anoddpm_head_bwd_args a)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];        // [COUT][9][C]
    const int C = a.C, C4 = C >> 2, P = a.H * a.W;
    for (int i = threadIdx.x; i < COUT * 9 * C; i += 256) {
        const int c = i % C, t = (i / C) % 9, o = i / (9 * C);
        wl[i] = a.w[((int64_t)o * C + c) * 9 + t];
    }
    __syncthreads();
    const int64_t total = (int64_t)a.B * P * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % C4);
        const int64_t bp = i / C4;
        const int p = (int)(bp % P), b = (int)(bp / P);
        const int y = p / a.W, x = p % a.W;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y - (t / 3 - 1), xx = x - (t % 3 - 1);     // output pixel that read this input through tap t
            if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const float d = a.dy[((int64_t)b * COUT + o) * P + (int64_t)yy * a.W + xx];
                s += *reinterpret_cast<const f32x4 *>(wl + (o * 9 + t) * C + q * 4) * d;
            }
        }
        *reinterpret_cast<f32x4 *>(a.da + bp * C + q * 4) = s;
    }
}

// dw: block = 512 input pixels of one image; thread = (channel quad, pixel lane); acc[tap][o] float4.
constexpr int HEAD_PIX = 512;

template <int COUT>
__global__ __launch_bounds__(256) void head_bwd_w_kernel(const anoddpm_head_bwd_args a, const int bpi)
{
    __shared__ float red[256 * 4];
    const int C = a.C, Q = C >> 2;                   // Q <= 64
    const int PL = 256 / Q;
    const int q = threadIdx.x % Q, pl = threadIdx.x / Q;
    const int b = blockIdx.x / bpi, chunk = blockIdx.x % bpi;
    const int P = a.H * a.W;
    const int p0 = chunk * HEAD_PIX, p1 = (p0 + HEAD_PIX < P) ? p0 + HEAD_PIX : P;
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.gn_scale + (int64_t)b * C + (q < Q ? q : 0) * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.gn_shift + (int64_t)b * C + (q < Q ? q : 0) * 4);
    f32x4 acc[9][COUT];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[t][o] = zero;
    float sb[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) sb[o] = 0.f;
    if (pl < PL)
        for (int p = p0 + pl; p < p1; p += PL) {                      // p = INPUT pixel
            f32x4 v = *reinterpret_cast<const f32x4 *>(a.x + ((int64_t)b * P + p) * C + q * 4) * sc + sh;
            v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]);
            const int y = p / a.W, x = p % a.W;
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const float *dyo = a.dy + ((int64_t)b * COUT + o) * P;
                if (q == 0) sb[o] += dyo[p];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y - (t / 3 - 1), xx = x - (t % 3 - 1);
                    const float d = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? dyo[(int64_t)yy * a.W + xx] : 0.f;
                    acc[t][o] += v * d;
                }
            }
        }
    // ws[block][row][C]: rows = o*9 + t, then COUT bias rows (only column 0 used)
    float *out = a.ws + (int64_t)blockIdx.x * (9 * COUT + COUT) * C;
    for (int r = 0; r < 9 * COUT + COUT; ++r) {
        __syncthreads();
        f32x4 v = zero;
        if (r < 9 * COUT) v = acc[r % 9][r / 9];
        else if (q == 0) v[0] = sb[r - 9 * COUT];
        if (pl < PL) *reinterpret_cast<f32x4 *>(red + threadIdx.x * 4) = v;
        __syncthreads();
        if (threadIdx.x < C) {
            const int c = threadIdx.x;
            float s = 0.f;
            for (int k = 0; k < PL; ++k) s += red[(k * Q + (c >> 2)) * 4 + (c & 3)];
            out[(int64_t)r * C + c] = s;
        }
    }
}

// sum over the nblk partial rows of one (row, channel): 32 entries x 8 row lanes per workgroup, sixteen loads in flight per thread
// (one thread per entry walking all rows one after the other took 122 us for the 1 024 rows of a 256^2 batch of four)
__device__ __forceinline__ float fold_rows8(const float *p, int64_t stride, int nblk, float (*part)[32])
{
    const int il = threadIdx.x >> 5, ol = threadIdx.x & 31;
    float s = 0.f;
    int k = il;
    for (; k + 8 * 15 < nblk; k += 8 * 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(int64_t)(k + 8 * u) * stride];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; k < nblk; k += 8) s += p[(int64_t)k * stride];
    part[il][ol] = s;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x < 32) {
#pragma unroll
        for (int l = 0; l < 8; ++l) t += part[l][threadIdx.x];
    }
    return t;
}

__global__ __launch_bounds__(256) void head_bwd_fold_kernel(const anoddpm_head_bwd_args a, const int nblk)
{
    __shared__ float part[8][32];
    const int C = a.C, rows = 9 * a.Cout + a.Cout;
    const int idx = blockIdx.x * 32 + (threadIdx.x & 31);
    const int idc = idx < rows * C ? idx : rows * C - 1;              // clamped: every thread takes part in the barrier
    const int c = idc % C, r = idc / C;
    const float s = fold_rows8(a.ws + (int64_t)r * C + c, (int64_t)rows * C, nblk, part);
    if (threadIdx.x >= 32 || idx >= rows * C) return;
    if (r < 9 * a.Cout) a.dw[((int64_t)(r / 9) * C + c) * 9 + r % 9] += s;       // OIHW [Cout][C][3][3]
    else if (c == 0) a.db[r - 9 * a.Cout] += s;
}

inline unsigned cap_grid(int64_t blocks) { return (unsigned)(blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks)); }

}  // namespace

using namespace anoddpm;

extern "C" int anoddpm_wgrad_pointwise(const anoddpm_wgrad1_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->a0 && a->dy && a->dw && a->ws, "wgrad_pointwise: null pointer");
    const int K = a->c0 + a->c1;
    ANODDPM_REQUIRE(a->c0 > 0 && a->c0 % 4 == 0 && a->c1 >= 0 && a->c1 % 4 == 0 && (a->c1 == 0 || a->a1), "wgrad_pointwise: channel counts must be multiples of 4");
    ANODDPM_REQUIRE(a->N >= 4 && a->N % 4 == 0 && a->B >= 1 && a->P >= 1, "wgrad_pointwise: bad sizes");
    ANODDPM_REQUIRE(a->span >= W1P && a->span % W1P == 0, "wgrad_pointwise: span must be a positive multiple of 32");
    ANODDPM_REQUIRE(a->a0_ld % 4 == 0 && (a->c1 == 0 || a->a1_ld % 4 == 0) && a->dy_ld % 4 == 0 && (a->a0_bs | a->a1_bs | a->dy_bs) % 4 == 0,
                    "wgrad_pointwise: strides must be multiples of 4 floats");
    ANODDPM_REQUIRE(!a->gn_scale || (a->gn_shift && a->gn_ld % 4 == 0), "wgrad_pointwise: bad GroupNorm affine");
    const int nspan = (a->P + a->span - 1) / a->span;
    const int64_t nitems = (int64_t)a->B * nspan;
    ANODDPM_REQUIRE(nitems <= 65535, "wgrad_pointwise: too many work items (raise span)");
    ANODDPM_REQUIRE(a->ws_floats >= nitems * ((int64_t)K * a->N + a->N), "wgrad_pointwise: workspace too small");
    const int tiles = ((K + W1T - 1) / W1T) * ((a->N + W1T - 1) / W1T);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(wgrad1_kernel, dim3(tiles, (unsigned)nitems), dim3(256), 0, s, *a, nspan);
    const int64_t kn = (int64_t)K * a->N;
    if (nitems >= 32 && a->N % 32 == 0 && g_debug[8] != 1)             // ANODDPM_DEBUG8=1: the one-thread-per-weight kernel everywhere
        hipLaunchKernelGGL(wgrad1_fold_lanes_kernel, dim3((unsigned)(kn / 32 + (a->dbias ? a->N / 32 : 0))), dim3(256), 0, s, *a, (int)nitems);
    else
        hipLaunchKernelGGL(wgrad1_fold_kernel, dim3((unsigned)((kn + 255) / 256)), dim3(256), 0, s, *a, (int)nitems);
    return check_launch("wgrad_pointwise");
}

extern "C" int anoddpm_pack_batch(const anoddpm_pack_batch_args *b, void *stream)
{
    ANODDPM_REQUIRE(b && b->jobs && b->block0 && b->njobs >= 1 && b->nblocks >= 1, "pack_batch: bad arguments");
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)b->nblocks), dim3(256), 0, as_stream(stream), *b);
    return check_launch("pack_batch");
}

extern "C" int64_t anoddpm_pack_job_blocks(const anoddpm_pack_args *a)
{
    return a ? anoddpm::pack_job_blocks(*a) : -1;
}

extern "C" int anoddpm_pack_weights(const anoddpm_pack_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->w && a->out, "pack_weights: null pointer");
    ANODDPM_REQUIRE(a->N >= 1 && a->K >= 1 && a->kind >= 0 && a->kind <= 5, "pack_weights: bad arguments");
    hipStream_t s = as_stream(stream);
    if (a->kind <= 1) return anoddpm_pack_conv3x3(a->w, a->out, a->N, a->K, a->kind, a->bwd, stream);
    if (a->kind == 5) return anoddpm_pack_conv3x3(a->w, a->out, a->N, a->K, 2, a->bwd, stream);
    if (a->kind == 2) {
        if (a->bwd) ANODDPM_REQUIRE(a->N % 4 == 0 && a->k0 >= 0 && a->kc >= 1 && a->k0 + a->kc <= a->K, "pack_weights: bad column range");
        else ANODDPM_REQUIRE(a->K % 4 == 0, "pack_weights: K must be a multiple of 4");
        const int64_t total = a->bwd ? (int64_t)a->N * a->kc : (int64_t)a->N * a->K;
        hipLaunchKernelGGL(pack_pointwise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, *a);
    } else if (a->kind == 3) {
        const int64_t total = (int64_t)9 * a->N * a->K;
        hipLaunchKernelGGL(pack_small_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, *a);
    } else {
        const int64_t total = (int64_t)a->N * a->K;
        hipLaunchKernelGGL(copy_kernel, dim3(cap_grid((total + 255) / 256)), dim3(256), 0, s, a->w, a->out, total);
    }
    return check_launch("pack_weights");
}

extern "C" int anoddpm_softmax_rows_backward(const anoddpm_softmax_bwd_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->p && a->dp && a->rows >= 0 && a->L > 0, "softmax_rows_backward: bad arguments");
    if (a->rows == 0) return ANODDPM_OK;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((a->rows + 3) / 4)), dim3(256), 0, as_stream(stream), a->p, a->dp, a->rows, a->L);
    return check_launch("softmax_rows_backward");
}

extern "C" int anoddpm_transpose_square(const anoddpm_transpose_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->in && a->out && a->in != a->out && a->Z >= 0 && a->L >= 1, "transpose_square: bad arguments");
    if (a->Z == 0) return ANODDPM_OK;
    ANODDPM_REQUIRE(a->Z <= 65535, "transpose_square: too many matrices");
    const unsigned t = (unsigned)((a->L + 31) / 32);
    hipLaunchKernelGGL(transpose_kernel, dim3(t, t, (unsigned)a->Z), dim3(256), 0, as_stream(stream), a->in, a->out, a->L);
    return check_launch("transpose_square");
}

extern "C" int anoddpm_linear_small_backward(const anoddpm_linear_bwd_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->x && a->w && a->dy && a->dw, "linear_small_backward: null pointer");
    ANODDPM_REQUIRE(a->B >= 1 && a->B <= 16 && a->K % 4 == 0 && a->K >= 4 && a->N >= 1, "linear_small_backward: need 1<=B<=16, K%%4==0");
    hipStream_t s = as_stream(stream);
    const int64_t tw = (int64_t)a->N * (a->K / 4);
    hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((unsigned)((tw + 255) / 256)), dim3(256), 0, s, *a);
    if (a->dx) {
        const dim3 g((unsigned)((a->K + 63) / 64));
        if (a->B <= 4) hipLaunchKernelGGL(linear_bwd_x_kernel<4>, g, dim3(1024), 0, s, *a);
        else if (a->B <= 8) hipLaunchKernelGGL(linear_bwd_x_kernel<8>, g, dim3(1024), 0, s, *a);
        else hipLaunchKernelGGL(linear_bwd_x_kernel<16>, g, dim3(1024), 0, s, *a);
    }
    return check_launch("linear_small_backward");
}

extern "C" int anoddpm_linear_small_backward_batch(const anoddpm_linear_bwd_batch_args *h, void *stream)
{
    ANODDPM_REQUIRE(h && h->jobs && h->x && h->njobs >= 1 && h->njobs <= 65535 && h->max_n >= 1, "linear_small_backward_batch: bad arguments");
    ANODDPM_REQUIRE(h->B >= 1 && h->B <= 16 && h->K % 4 == 0 && h->K >= 4, "linear_small_backward_batch: need 1<=B<=16, K%%4==0");
    ANODDPM_REQUIRE(!h->dx || h->ws, "linear_small_backward_batch: dx needs the partial-sum workspace [njobs][B][K]");
    hipStream_t s = as_stream(stream);
    const int64_t tw = (int64_t)h->max_n * (h->K / 4);
    hipLaunchKernelGGL(linear_bwd_w_batch_kernel, dim3((unsigned)((tw + 255) / 256), (unsigned)h->njobs), dim3(256), 0, s, *h);
    if (h->dx) {
        const dim3 g((unsigned)((h->K + 63) / 64), (unsigned)h->njobs);
        if (h->B <= 4) hipLaunchKernelGGL(linear_bwd_x_batch_kernel<4>, g, dim3(1024), 0, s, *h);
        else if (h->B <= 8) hipLaunchKernelGGL(linear_bwd_x_batch_kernel<8>, g, dim3(1024), 0, s, *h);
        else hipLaunchKernelGGL(linear_bwd_x_batch_kernel<16>, g, dim3(1024), 0, s, *h);
        hipLaunchKernelGGL(linear_bwd_x_fold_kernel, dim3((unsigned)(((int64_t)h->B * h->K + 255) / 256)), dim3(256), 0, s, *h);
    }
    return check_launch("linear_small_backward_batch");
}

extern "C" int anoddpm_colsum_fold(const anoddpm_colsum_fold_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->colsum && a->dimg && a->B >= 1 && a->B <= 65535 && a->ipb >= 1 && a->N >= 1, "colsum_fold: bad arguments");
    hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((a->N + 63) / 64)), dim3(1024), 0, as_stream(stream), *a);
    return check_launch("colsum_fold");
}

extern "C" int anoddpm_conv_stem_backward(const anoddpm_stem_bwd_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->x && a->w && a->dy && a->dw && a->db && a->ws, "conv_stem_backward: null pointer");
    ANODDPM_REQUIRE(a->Cin >= 1 && a->Cin <= 16 && a->Cout % 4 == 0 && a->Cout >= 4 && a->Cout <= 256, "conv_stem_backward: need Cin<=16, Cout%%4==0, Cout<=256");
    ANODDPM_REQUIRE(a->B >= 1 && a->H >= 1 && a->W >= 1, "conv_stem_backward: bad sizes");
    const int P = a->H * a->W;
    const int bpi = (P + STEM_PIX - 1) / STEM_PIX;
    const int nblk = a->B * bpi;
    ANODDPM_REQUIRE(a->ws_floats >= (int64_t)nblk * (a->Cin * 9 + 1) * a->Cout, "conv_stem_backward: workspace too small");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(stem_bwd_w_kernel, dim3(nblk, a->Cin), dim3(256), 0, s, *a, bpi);
    const int tot = (a->Cin * 9 + 1) * a->Cout;
    hipLaunchKernelGGL(stem_bwd_fold_kernel, dim3((tot + 31) / 32), dim3(256), 0, s, *a, nblk);
    if (a->dx) {
        const int64_t n = (int64_t)a->B * a->Cin * P;
        hipLaunchKernelGGL(stem_bwd_x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *a);
    }
    return check_launch("conv_stem_backward");
}

extern "C" int anoddpm_conv_head_backward(const anoddpm_head_bwd_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->x && a->gn_scale && a->gn_shift && a->w && a->dy && a->da && a->dw && a->db && a->ws, "conv_head_backward: null pointer");
    ANODDPM_REQUIRE(a->Cout >= 1 && a->Cout <= 4 && a->C % 4 == 0 && a->C >= 4 && a->C <= 256, "conv_head_backward: need Cout<=4, C%%4==0, C<=256");
    ANODDPM_REQUIRE(a->B >= 1 && a->H >= 1 && a->W >= 1, "conv_head_backward: bad sizes");
    const int P = a->H * a->W;
    const int bpi = (P + HEAD_PIX - 1) / HEAD_PIX;
    const int nblk = a->B * bpi;
    ANODDPM_REQUIRE(a->ws_floats >= (int64_t)nblk * 10 * a->Cout * a->C, "conv_head_backward: workspace too small");
    hipStream_t s = as_stream(stream);
    const size_t lds = (size_t)a->Cout * 9 * a->C * sizeof(float);
    const int64_t work = (int64_t)a->B * P * (a->C / 4);
    const dim3 gda(cap_grid((work + 255) / 256));
    switch (a->Cout) {
        case 1: hipLaunchKernelGGL(head_bwd_da_kernel<1>, gda, dim3(256), lds, s, *a); hipLaunchKernelGGL(head_bwd_w_kernel<1>, dim3(nblk), dim3(256), 0, s, *a, bpi); break;
        case 2: hipLaunchKernelGGL(head_bwd_da_kernel<2>, gda, dim3(256), lds, s, *a); hipLaunchKernelGGL(head_bwd_w_kernel<2>, dim3(nblk), dim3(256), 0, s, *a, bpi); break;
        case 3: hipLaunchKernelGGL(head_bwd_da_kernel<3>, gda, dim3(256), lds, s, *a); hipLaunchKernelGGL(head_bwd_w_kernel<3>, dim3(nblk), dim3(256), 0, s, *a, bpi); break;
        default: hipLaunchKernelGGL(head_bwd_da_kernel<4>, gda, dim3(256), lds, s, *a); hipLaunchKernelGGL(head_bwd_w_kernel<4>, dim3(nblk), dim3(256), 0, s, *a, bpi); break;
    }
    const int tot = 10 * a->Cout * a->C;
    hipLaunchKernelGGL(head_bwd_fold_kernel, dim3((tot + 31) / 32), dim3(256), 0, s, *a, nblk);
    return check_launch("conv_head_backward");
}
