// How should an HBM-bound helper kernel be shaped on MI355X?  Sweep of streaming patterns (write / read / copy / 2-read-1-write)
// over workgroups per CU, 16-byte accesses in flight per lane, interleaved vs chunked ownership and the nt policy, on a buffer
// that exceeds the 256 MiB Infinity Cache (1 GiB) and on one that fits (128 MiB, the size of a 256x256x128 fp32 batch of four).
// Prints GB/s per variant (HIP events, best of 5).  The product's helper kernels (stem, head, resample, chan_stats, GroupNorm
// backward, optimiser) follow the best rows -- DESIGN.md section 8c.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/hbm_patterns.bin tools/hbm_patterns.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ void st16(f32x4 *p, f32x4 v)
{
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <bool NT> __device__ __forceinline__ f32x4 ld16(const f32x4 *p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}

// MODE 0: interleaved -- the whole grid sweeps the buffer front to back, UNROLL x grid-size 16-byte accesses per iteration
// MODE 1: chunked -- a workgroup owns one contiguous range
template <int OP, int UNROLL, bool NT, int MODE>
__global__ __launch_bounds__(256) void stream_kernel(f32x4 *__restrict__ dst, const f32x4 *__restrict__ a, const f32x4 *__restrict__ b,
                                                     size_t n16, float *sink)
{
    const size_t G = (size_t)gridDim.x * 256;
    size_t i, step, end;
    if (MODE == 0) {
        i = (size_t)blockIdx.x * 256 + threadIdx.x;
        step = G;
        end = n16;
    } else {
        const size_t per = n16 / gridDim.x;
        i = (size_t)blockIdx.x * per + threadIdx.x;
        step = 256;
        end = (size_t)(blockIdx.x + 1) * per;
    }
    f32x4 acc = {0, 0, 0, 0};
    const f32x4 cst = {1.f, 2.f, 3.f, 4.f};
    for (; i + (UNROLL - 1) * step < end; i += UNROLL * step) {
        f32x4 va[UNROLL], vb[UNROLL];
        if (OP != 0) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) va[u] = ld16<NT>(a + i + u * step);
        }
        if (OP == 3) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) vb[u] = ld16<NT>(b + i + u * step);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (OP == 0) st16<NT>(dst + i + u * step, cst);
            else if (OP == 1) acc += va[u];
            else if (OP == 2) st16<NT>(dst + i + u * step, va[u]);
            else st16<NT>(dst + i + u * step, va[u] * vb[u]);
        }
    }
    if (OP == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

struct Bufs { f32x4 *d, *a, *b; float *sink; };

template <int OP, int UNROLL, bool NT, int MODE>
static double run(const Bufs &B, size_t bytes, int wg_per_cu)
{
    const size_t n16 = bytes / 16;
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        stream_kernel<OP, UNROLL, NT, MODE><<<grid, 256>>>(B.d, B.a, B.b, n16, B.sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, ms);
    }
    const double moved = (double)bytes * (OP == 0 || OP == 1 ? 1 : OP == 2 ? 2 : 3);
    return moved / (best * 1e-3) / 1e9;
}

template <int OP, int UNROLL>
static void sweep(const Bufs &B, size_t bytes, const char *name)
{
    for (int wg : {1, 2, 4, 8, 16}) {
        printf("%s,%zu,%d,%d,%.0f,%.0f,%.0f,%.0f\n", name, bytes >> 20, UNROLL, wg,
               run<OP, UNROLL, false, 0>(B, bytes, wg), run<OP, UNROLL, true, 0>(B, bytes, wg),
               run<OP, UNROLL, false, 1>(B, bytes, wg), run<OP, UNROLL, true, 1>(B, bytes, wg));
        fflush(stdout);
    }
}

int main()
{
    Bufs B;
    const size_t big = (size_t)1 << 30;
    CK(hipMalloc((void **)&B.d, big));
    CK(hipMalloc((void **)&B.a, big));
    CK(hipMalloc((void **)&B.b, big));
    CK(hipMalloc((void **)&B.sink, 64));
    CK(hipMemset(B.d, 0, big));
    CK(hipMemset(B.a, 0, big));
    CK(hipMemset(B.b, 0, big));
    printf("op,MiB_per_tensor,unroll,wg_per_cu,GBps_interleaved,GBps_interleaved_nt,GBps_chunked,GBps_chunked_nt\n");
    for (size_t bytes : {big, (size_t)128 << 20}) {
        sweep<0, 1>(B, bytes, "write");
        sweep<0, 4>(B, bytes, "write");
        sweep<0, 8>(B, bytes, "write");
        sweep<1, 1>(B, bytes, "read");
        sweep<1, 4>(B, bytes, "read");
        sweep<1, 8>(B, bytes, "read");
        sweep<2, 1>(B, bytes, "copy");
        sweep<2, 4>(B, bytes, "copy");
        sweep<2, 8>(B, bytes, "copy");
        sweep<3, 1>(B, bytes, "mul2to1");
        sweep<3, 4>(B, bytes, "mul2to1");
    }
    // reference point: the runtime's own fill
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        CK(hipMemsetAsync(B.d, 0, big, 0));
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemsetAsync,1024,0,0,%.0f,,,\n", (double)big / (ms * 1e-3) / 1e9);
    }
    return 0;
}
