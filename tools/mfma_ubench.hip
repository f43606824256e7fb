// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 in the register/LDS context of the igemm kernel.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_ubench.hip -o /tmp/mfma_ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed * (i & 15);
    __syncthreads();
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const f32x4 *l4 = reinterpret_cast<const f32x4 *>(lds);
    f32x4 av[2] = {l4[threadIdx.x], l4[threadIdx.x + 256]}, bv[2] = {l4[threadIdx.x + 512], l4[threadIdx.x + 768]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            if (MODE >= 1) {       // fragment reads from LDS each group (as the igemm does)
                const int o = ((it * 4 + k8) * 64 + threadIdx.x) & 1023;
                av[0] = l4[o]; av[1] = l4[o + 1024 - 512]; bv[0] = l4[(o + 256) & 1023]; bv[1] = l4[(o + 512) & 1023];
                if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][kk], bv[n][kk], acc[m][n], 0, 0, 0);
        }
        if (MODE == 3) __syncthreads();
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int blocks, int iters, size_t dynlds)
{
    float *out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), dynlds, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), dynlds, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 64;
    const double flops = mfmas * 4096.0;
    // cycles per MFMA per SIMD assuming 2.4 GHz and blocks spread over 256 CUs x 4 SIMDs
    const double waves_per_simd = blocks / 256.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 64.0 * waves_per_simd);
    printf("%-34s blocks=%4d  %8.3f ms  %7.1f TFLOP/s  %6.1f cycles/MFMA/SIMD\n", name, blocks, ms, flops / ms / 1e9, cyc);
    hipFree(out);
}

int main()
{
    const int it = 4000;
    run<0>("mfma only, 1 wave/SIMD", 256, it, 100 * 1024);
    run<0>("mfma only, 2 waves/SIMD", 512, it, 0);
    run<1>("mfma + LDS frags, 1 wave/SIMD", 256, it, 100 * 1024);
    run<1>("mfma + LDS frags, 2 waves/SIMD", 512, it, 0);
    run<2>("mfma + pinned frags, 1 wave/SIMD", 256, it, 100 * 1024);
    run<2>("mfma + pinned frags, 2 waves/SIMD", 512, it, 0);
    run<3>("mfma + barrier/64, 2 waves/SIMD", 512, it, 0);
    run<0>("mfma only, 4 rounds (2048 blk)", 2048, it / 4, 0);
    return 0;
}
