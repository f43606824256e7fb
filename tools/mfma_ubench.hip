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


// ---- does VALU work hide in the shadow of the fp32 MFMA?  Per iteration: 16 MFMAs (4 accumulators x 4) and
// NV independent v_fma_f32 (12 chains).  VM 0: no VALU.  1: interleaved by sched_group_barrier (1 MFMA : NV/16 VALU).
// 2: clustered (all VALU, then the 16 MFMAs).  3: clustered + one workgroup barrier per iteration.  4: VALU only.
template <int VM, int NV>
__global__ __launch_bounds__(256, 2) void kv(float *out, int iters, float seed)
{
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float x[12];
    for (int i = 0; i < 12; ++i) x[i] = seed * (threadIdx.x + i);
    const float c1 = 1.0f + seed * 1e-7f, c2 = seed * 1e-3f;
    float av = seed * (threadIdx.x & 7), bv = seed;
    for (int it = 0; it < iters; ++it) {
        if (VM != 0) {
#pragma unroll
            for (int j = 0; j < NV; ++j) x[j % 12] = __builtin_fmaf(x[j % 12], c1, c2);
        }
        if (VM == 2 || VM == 3) __builtin_amdgcn_sched_barrier(0);
        if (VM != 4) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
        }
        if (VM == 1) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NV / 16, 0);
            }
        }
        if (VM == 2 || VM == 3) __builtin_amdgcn_sched_barrier(0);
        if (VM == 3) __syncthreads();
    }
    float s = 0.f;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    for (int i = 0; i < 12; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VM, int NV>
void runv(const char *name, int blocks, int iters, size_t dynlds)
{
    float *out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kv<VM, NV>), dim3(blocks), dim3(256), dynlds, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kv<VM, NV>), dim3(blocks), dim3(256), dynlds, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = blocks / 256.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * waves_per_simd);     // cycles per iteration per wave slot
    printf("%-44s NV=%3d blocks=%4d  %8.3f ms  %8.1f cycles/iter/wave-slot (16 MFMA = 1024 + VALU 4/instr = %d)\n",
           name, NV, blocks, ms, cyc, NV * 4);
    hipFree(out);
}

// ---- which instruction classes cost issue time next to the fp32 MFMA?  16 MFMAs + 48 instructions of class CL,
// interleaved 1:3, 2 waves/SIMD.  CL 0 v_fma_f32, 1 v_add_u32, 2 v_exp_f32, 3 v_pk_fma_f32, 4 ds_read_b128,
// 5 s_add_u32, 6 v_mov_b32, 7 v_add_f32, 8 v_pk_add_f32, 9 global_load_dwordx4 (L2-resident)
template <int CL, bool WITH_MFMA>
__global__ __launch_bounds__(256, 2) void kc(float *out, const float *gsrc, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * (i & 15);
    __syncthreads();
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float x[12];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 x2[12];
    f32x4 x4[12];
    int xi[12];
    for (int i = 0; i < 12; ++i) { x[i] = seed * (threadIdx.x + i); x2[i] = f32x2{x[i], x[i] + 1.f}; xi[i] = threadIdx.x + i; x4[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float c1 = 1.0f + seed * 1e-7f, c2 = seed * 1e-3f;
    const f32x2 c12 = {c1, c1}, c22 = {c2, c2};
    int sacc = iters;
    const uint32_t laddr = (threadIdx.x * 16) & 16383;
    const float *gp = gsrc + threadIdx.x * 4;
    float av = seed * (threadIdx.x & 7), bv = seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (WITH_MFMA) {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(av), "v"(bv));
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int i = (g * 3 + j) % 12;
                if (CL == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
                if (CL == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(xi[i]) : "v"(sacc));
                if (CL == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                if (CL == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x2[i]) : "v"(c12), "v"(c22));
                if (CL == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(x4[i]) : "v"(laddr));
                if (CL == 5) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                if (CL == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(c1));
                if (CL == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c2));
                if (CL == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x2[i]) : "v"(c22));
                if (CL == 9) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x4[i]) : "v"(gp));
            }
        }
        if (CL == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (CL == 9) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float s = (float)sacc;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    for (int i = 0; i < 12; ++i) s += x[i] + x2[i][0] + x2[i][1] + (float)xi[i] + x4[i][0] + x4[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CL>
void runc(const char *name)
{
    const int blocks = 512, iters = 8000;
    float *out, *src; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int w = 0; w < 2; ++w) {
        if (w) hipLaunchKernelGGL((kc<CL, true>), dim3(blocks), dim3(256), 0, 0, out, src, 10, 1.0f);
        else   hipLaunchKernelGGL((kc<CL, false>), dim3(blocks), dim3(256), 0, 0, out, src, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (w) hipLaunchKernelGGL((kc<CL, true>), dim3(blocks), dim3(256), 0, 0, out, src, iters, 1.0f);
        else   hipLaunchKernelGGL((kc<CL, false>), dim3(blocks), dim3(256), 0, 0, out, src, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[w], e0, e1);
    }
    const double alone = ms[0] * 1e-3 * 2.4e9 / (iters * 2.0), both = ms[1] * 1e-3 * 2.4e9 / (iters * 2.0);
    printf("class %-22s 48 instr alone %7.1f cyc/iter/wave-slot | with 16 MFMA %7.1f (MFMA alone 1033) -> exposed %6.1f = %4.1f cyc/instr\n",
           name, alone, both, both - 1033.0, (both - 1033.0) / 48.0);
    hipFree(out); hipFree(src);
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
    const int iv = 16000;
    runv<0, 48>("no VALU, 1 wave/SIMD", 256, iv, 100 * 1024);
    runv<0, 48>("no VALU, 2 waves/SIMD", 512, iv, 0);
    runv<4, 48>("VALU only, 1 wave/SIMD", 256, iv, 100 * 1024);
    runv<4, 48>("VALU only, 2 waves/SIMD", 512, iv, 0);
    runv<1, 48>("interleaved, 1 wave/SIMD", 256, iv, 100 * 1024);
    runv<1, 48>("interleaved, 2 waves/SIMD", 512, iv, 0);
    runv<2, 48>("clustered, 1 wave/SIMD", 256, iv, 100 * 1024);
    runv<2, 48>("clustered, 2 waves/SIMD", 512, iv, 0);
    runv<3, 48>("clustered + barrier, 2 waves/SIMD", 512, iv, 0);
    runv<1, 96>("interleaved, 1 wave/SIMD", 256, iv, 100 * 1024);
    runv<1, 96>("interleaved, 2 waves/SIMD", 512, iv, 0);
    runv<2, 96>("clustered, 2 waves/SIMD", 512, iv, 0);
    runv<3, 96>("clustered + barrier, 2 waves/SIMD", 512, iv, 0);
    runv<1, 192>("interleaved, 2 waves/SIMD", 512, iv, 0);
    runv<3, 192>("clustered + barrier, 2 waves/SIMD", 512, iv, 0);
    runc<0>("v_fma_f32"); runc<7>("v_add_f32"); runc<3>("v_pk_fma_f32"); runc<8>("v_pk_add_f32"); runc<1>("v_add_u32");
    runc<6>("v_mov_b32"); runc<2>("v_exp_f32"); runc<5>("s_add_u32"); runc<4>("ds_read_b128"); runc<9>("global_load_dwordx4");
    return 0;
}
