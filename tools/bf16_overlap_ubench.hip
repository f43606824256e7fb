// Micro-benchmark behind DESIGN §5i: before building a wave-specialised split-bf16 F(4x4) kernel, measure what it rests on.
//   (1) does v_mfma_f32_16x16x32_bf16 issued by ONE wave of a SIMD overlap the VALU work of the OTHER wave of that SIMD
//       (the fp32 MFMA does not: tools/mfma_ubench.hip)?  Workgroup of 8 waves; "consumer" waves run MFMAs on 8 independent
//       accumulators, "producer" waves run 12 independent chains of one VALU class.  Three launches per class: consumers only,
//       producers only, both.  Two wave->role maps: consumers = waves 0-3 (w and w + 4 share a SIMD if waves are dealt round-robin)
//       and consumers = even waves.
//   (2) the same inside ONE wave (1 MFMA : 3 VALU, program order).
//   (3) the rate of an L2-resident stream of 16-byte buffer loads per CU (the weight planes of §5i: 884 KB per chunk).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bf16_overlap_ubench.bin tools/bf16_overlap_ubench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NT = 512;
constexpr int MF = 48;     // MFMAs per iteration (consumer)
constexpr int NV = 96;     // VALU instructions per iteration (producer)

// CL: 0 v_fma_f32  1 v_pk_fma_f32  2 v_exp_f32  3 v_cvt_pk_bf16_f32  4 v_dot2_f32_bf16  5 v_and_b32  6 v_perm_b32  7 v_pk_add_f32
//     8 v_rcp_f32  9 ds_write_b32  10 ds_read_b64
template <int CL>
__device__ __forceinline__ void valu_block(float (&x)[12], f32x2 (&x2)[12], unsigned (&xi)[12], float c1, float c2, unsigned lds_addr)
{
    const f32x2 c12 = {c1, c1}, c22 = {c2, c2};
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = j % 12;
        if (CL == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
        if (CL == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x2[i]) : "v"(c12), "v"(c22));
        if (CL == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        if (CL == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(xi[i]) : "v"(x[i]), "v"(c1));
        if (CL == 4) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(xi[i]), "v"(xi[(i + 1) % 12]));
        if (CL == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(xi[i]) : "v"(xi[(i + 1) % 12]));
        if (CL == 6) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(xi[i]) : "v"(xi[(i + 1) % 12]), "v"(xi[(i + 2) % 12]));
        if (CL == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x2[i]) : "v"(c22));
        if (CL == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
        if (CL == 9) asm volatile("ds_write_b32 %0, %1" :: "v"(lds_addr), "v"(xi[i]) : "memory");
        if (CL == 10) asm volatile("ds_read_b64 %0, %1" : "=v"(x2[i]) : "v"(lds_addr) : "memory");
    }
    if (CL == 9 || CL == 10) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// ROLE: 0 consumers only, 1 producers only, 2 both.  MAP: 0 consumers = waves 0-3, 1 consumers = even waves, 2 every wave does both
// (1 MFMA : 2 VALU in program order)
template <int CL, int ROLE, int MAP, int PRIO = 0>
__global__ __launch_bounds__(NT, 1) void overlap_kernel(float *out, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[NT * 4];          // producers touch 8 bytes per lane: conflict-free b64
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool consumer = MAP == 0 ? wave < 4 : (MAP == 1 ? (wave & 1) == 0 : true);
    float x[12];
    f32x2 x2[12];
    unsigned xi[12];
    for (int i = 0; i < 12; ++i) { x[i] = seed * (threadIdx.x + i); x2[i] = f32x2{x[i], x[i] + 1.f}; xi[i] = threadIdx.x * 2654435761u + i; }
    const float c1 = 1.0f + seed * 1e-7f, c2 = seed * 1e-3f;
    const unsigned lds_addr = (unsigned)(size_t)(&lds[0]) + (threadIdx.x & 255) * 8;
    float s = 0.f;
    if (MAP == 2) {
        f32x4 acc[8];
        for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (threadIdx.x & 7)); b[i] = (__bf16)seed; }
        const f32x2 c12 = {c1, c1}, c22 = {c2, c2};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < MF; ++g) {
                if (ROLE != 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[g & 7]) : "v"(a), "v"(b));
                if (ROLE != 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int i = (g * 2 + j) % 12;
                        if (CL == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
                        if (CL == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x2[i]) : "v"(c12), "v"(c22));
                        if (CL == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                    }
                }
            }
        }
        for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][3];
    } else if (consumer) {
        if (ROLE != 1) {
            f32x4 acc[8];
            for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (threadIdx.x & 7)); b[i] = (__bf16)seed; }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < MF; ++g) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[g & 7]) : "v"(a), "v"(b));
            }
            for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][3];
        }
    } else {
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);                   // producers ahead of the MFMA wave in the SIMD's arbiter
        if (ROLE != 0)
            for (int it = 0; it < iters; ++it) valu_block<CL>(x, x2, xi, c1, c2, lds_addr);
    }
    for (int i = 0; i < 12; ++i) s += x[i] + x2[i][0] + x2[i][1] + (float)xi[i];
    out[blockIdx.x * NT + threadIdx.x] = s + lds[threadIdx.x];
}

template <int CL, int ROLE, int MAP, int PRIO = 0>
static float time_overlap(float *out, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((overlap_kernel<CL, ROLE, MAP, PRIO>), dim3(256), dim3(NT), 0, 0, out, 10, 1.0f);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((overlap_kernel<CL, ROLE, MAP, PRIO>), dim3(256), dim3(NT), 0, 0, out, iters, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    return best;
}

template <int CL>
static void run_class(const char *name, float *out)
{
    const int iters = 4000;
    const double ghz = 2.4e9;
    for (int map = 0; map < 2; ++map) {
        float m, v, b;
        if (map == 0) { m = time_overlap<CL, 0, 0>(out, iters); v = time_overlap<CL, 1, 0>(out, iters); b = time_overlap<CL, 2, 0>(out, iters); }
        else          { m = time_overlap<CL, 0, 1>(out, iters); v = time_overlap<CL, 1, 1>(out, iters); b = time_overlap<CL, 2, 1>(out, iters); }
        const double cm = m * 1e-3 * ghz / iters, cv = v * 1e-3 * ghz / iters, cb = b * 1e-3 * ghz / iters;
        printf("%-20s %-22s  %d MFMA alone %7.1f cyc (%4.1f / MFMA) | %d VALU alone %7.1f (%4.1f / instr) | both %7.1f  -> sum %7.1f, max %7.1f\n",
               name, map ? "consumers = even waves" : "consumers = waves 0-3", MF, cm, cm / MF, NV, cv, cv / NV, cb, cm + cv, cm > cv ? cm : cv);
    }
}

template <int CL>
static void run_prio(const char *name, float *out)
{
    const int iters = 4000;
    const double ghz = 2.4e9;
    const float m = time_overlap<CL, 0, 0>(out, iters), v = time_overlap<CL, 1, 0>(out, iters);
    const float b0 = time_overlap<CL, 2, 0, 0>(out, iters), b1 = time_overlap<CL, 2, 0, 1>(out, iters), b3 = time_overlap<CL, 2, 0, 3>(out, iters);
    const double k = 1e-3 * ghz / iters;
    printf("%-20s producers = waves 4-7 with s_setprio: MFMA alone %7.1f | VALU alone %7.1f | both: prio 0 %7.1f, prio 1 %7.1f, prio 3 %7.1f\n", name, m * k,
           v * k, b0 * k, b1 * k, b3 * k);
}

template <int CL>
static void run_same_wave(const char *name, float *out)
{
    const int iters = 4000;
    const double ghz = 2.4e9;
    const float m = time_overlap<CL, 0, 2>(out, iters), v = time_overlap<CL, 1, 2>(out, iters), b = time_overlap<CL, 2, 2>(out, iters);
    const double cm = m * 1e-3 * ghz / iters, cv = v * 1e-3 * ghz / iters, cb = b * 1e-3 * ghz / iters;
    printf("%-20s same wave, 1 MFMA : 2 VALU (2 waves / SIMD)  MFMA alone %7.1f | VALU alone %7.1f | both %7.1f -> sum %7.1f, max %7.1f\n", name, cm, cv,
           cb, cm + cv, cm > cv ? cm : cv);
}

// ---- (3) L2-resident 16-byte buffer-load stream: every wave walks a window of `bytes` again and again
__global__ __launch_bounds__(NT, 1) void stream_kernel(const f32x4 *src, float *out, int iters, int n16, int waves_active)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (wave < waves_active) {
        const int per_wave = n16 / waves_active;                      // contiguous range per wave
        const f32x4 *p = src + (size_t)wave * per_wave + (threadIdx.x & 63);
        for (int it = 0; it < iters; ++it) {
            for (int i = 0; i < per_wave; i += 64 * 8) {
                f32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = p[i + j * 64];
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[j];
            }
        }
    }
    out[blockIdx.x * NT + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

static void run_stream(float *out, int kb, int waves_active)
{
    const int n16 = kb * 1024 / 16;
    f32x4 *src;
    CK(hipMalloc(&src, (size_t)n16 * 16));
    CK(hipMemset(src, 0, (size_t)n16 * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 40;
    hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(NT), 0, 0, src, out, 2, n16, waves_active);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(NT), 0, 0, src, out, iters, n16, waves_active);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes_per_cu = (double)n16 * 16 * iters;
    printf("L2-resident stream, window %5d KB (all CUs read the same), %d waves / CU: %6.1f B / clk / CU at 2.4 GHz  (%.2f TB/s chip)\n", kb, waves_active,
           bytes_per_cu / (ms * 1e-3 * 2.4e9), bytes_per_cu * 256 / (ms * 1e-3) / 1e12);
    CK(hipFree(src));
}


// ---- (4) what kind of non-overlap is it?  Cross-wave test again (consumers = waves 0-3: MFMAs only; producers = waves 4-7: v_fma_f32
// only) with:  GRID workgroups (64: a quarter of the chip -- a power cap would not bind),  NACC independent accumulators (2: the MFMA
// stream is latency-bound and leaves the pipe idle half of the time),  NOP: s_nop between the consumer's MFMAs
template <int NACC, int NOP, int ROLE>
__global__ __launch_bounds__(NT, 1) void cross_kernel(float *out, int iters, float seed)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float x[12];
    for (int i = 0; i < 12; ++i) x[i] = seed * (threadIdx.x + i);
    const float c1 = 1.0f + seed * 1e-7f, c2 = seed * 1e-3f;
    float s = 0.f;
    if (wave < 4) {
        if (ROLE != 1) {
            f32x4 acc[8];
            for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (threadIdx.x & 7)); b[i] = (__bf16)seed; }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < MF; ++g) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[g % NACC]) : "v"(a), "v"(b));
                    if (NOP == 1) asm volatile("s_nop 7");
                    if (NOP == 2) asm volatile("s_nop 7\n s_nop 3");
                }
            }
            for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][3];
        }
    } else if (ROLE != 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j % 12]) : "v"(c1), "v"(c2));
        }
    }
    for (int i = 0; i < 12; ++i) s += x[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
}

static int g_dynlds = 0;                                              // dynamic LDS bytes of the cross / pinned launches (allocated, never touched)
template <int NACC, int NOP, int ROLE>
static float time_cross(float *out, int grid, int iters)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&cross_kernel<NACC, NOP, ROLE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((cross_kernel<NACC, NOP, ROLE>), dim3(grid), dim3(NT), g_dynlds, 0, out, 10, 1.0f);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((cross_kernel<NACC, NOP, ROLE>), dim3(grid), dim3(NT), g_dynlds, 0, out, iters, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    return best;
}

template <int NACC, int NOP>
static void run_cross(float *out, int grid, int iters)
{
    const double k = 1e-3 * 2.4e9 / iters;
    const float m = time_cross<NACC, NOP, 0>(out, grid, iters), v = time_cross<NACC, NOP, 1>(out, grid, iters), b = time_cross<NACC, NOP, 2>(out, grid, iters);
    printf("cross-wave v_fma_f32 (dynamic LDS %6d B): grid %3d, %d accumulators, nop %d, %5d iterations:  MFMA alone %7.1f | VALU alone %7.1f | both %7.1f  (sum %7.1f)\n", g_dynlds, grid, NACC,
           NOP, iters, m * k, v * k, b * k, (m + v) * k);
}

// ---- (5) the same cross-wave test with the producer's registers PINNED: chains in v[CH0 .. CH0 + 11], the two constants in
// v[KA], v[KB].  (Found by accident: two compilations of the same instruction stream that differed only in the producer's
// register numbers ran 874 and 1296 cycles per iteration beside the same MFMA wave.)
#define STR2(x) #x
#define STR(x) STR2(x)
template <int CH0, int KA, int KB, int ROLE>
__global__ __launch_bounds__(NT, 1) void pinned_kernel(float *out, int iters, float seed)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float c1 = 1.0f + seed * 1e-7f, c2 = seed * 1e-3f;
    float s = 0.f;
    if (wave < 4) {
        if (ROLE != 1) {
            f32x4 acc[8];
            for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (threadIdx.x & 7)); b[i] = (__bf16)seed; }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < MF; ++g) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[g & 7]) : "v"(a), "v"(b));
            }
            for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][3];
        }
    } else if (ROLE != 0) {
        float r = 0.f;
        asm volatile(
            "v_mov_b32 v[%c3], %1\n v_mov_b32 v[%c4], %2\n"
            "v_mov_b32 v[%c5+0], %1\n v_mov_b32 v[%c5+1], %1\n v_mov_b32 v[%c5+2], %1\n v_mov_b32 v[%c5+3], %1\n v_mov_b32 v[%c5+4], %1\n v_mov_b32 v[%c5+5], %1\n"
            "v_mov_b32 v[%c5+6], %1\n v_mov_b32 v[%c5+7], %1\n v_mov_b32 v[%c5+8], %1\n v_mov_b32 v[%c5+9], %1\n v_mov_b32 v[%c5+10], %1\n v_mov_b32 v[%c5+11], %1\n"
            "s_mov_b32 s40, %6\n"
            "1:\n"
            ".rept 8\n"
            "v_fma_f32 v[%c5+0], v[%c5+0], v[%c3], v[%c4]\n v_fma_f32 v[%c5+1], v[%c5+1], v[%c3], v[%c4]\n v_fma_f32 v[%c5+2], v[%c5+2], v[%c3], v[%c4]\n"
            "v_fma_f32 v[%c5+3], v[%c5+3], v[%c3], v[%c4]\n v_fma_f32 v[%c5+4], v[%c5+4], v[%c3], v[%c4]\n v_fma_f32 v[%c5+5], v[%c5+5], v[%c3], v[%c4]\n"
            "v_fma_f32 v[%c5+6], v[%c5+6], v[%c3], v[%c4]\n v_fma_f32 v[%c5+7], v[%c5+7], v[%c3], v[%c4]\n v_fma_f32 v[%c5+8], v[%c5+8], v[%c3], v[%c4]\n"
            "v_fma_f32 v[%c5+9], v[%c5+9], v[%c3], v[%c4]\n v_fma_f32 v[%c5+10], v[%c5+10], v[%c3], v[%c4]\n v_fma_f32 v[%c5+11], v[%c5+11], v[%c3], v[%c4]\n"
            ".endr\n"
            "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"
            "v_mov_b32 %0, v[%c5+0]\n"
            : "=v"(r)
            : "v"(c1), "v"(c2), "n"(KA), "n"(KB), "n"(CH0), "s"(iters)
            : "s40", "scc", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21",
              "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43",
              "v44", "v45", "v46", "v47", "v48", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
              "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
        s += r;
    }
    out[blockIdx.x * NT + threadIdx.x] = s;
}

template <int CH0, int KA, int KB>
static void run_pinned(float *out)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float t[3];
    for (int role = 0; role < 3; ++role) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (role == 0) hipLaunchKernelGGL((pinned_kernel<CH0, KA, KB, 0>), dim3(256), dim3(NT), 0, 0, out, iters, 1.0f);
            if (role == 1) hipLaunchKernelGGL((pinned_kernel<CH0, KA, KB, 1>), dim3(256), dim3(NT), 0, 0, out, iters, 1.0f);
            if (role == 2) hipLaunchKernelGGL((pinned_kernel<CH0, KA, KB, 2>), dim3(256), dim3(NT), 0, 0, out, iters, 1.0f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best = ms < best ? ms : best;
        }
        t[role] = best;
    }
    const double k = 1e-3 * 2.4e9 / iters;
    printf("pinned producer: chains v[%d..%d], constants v%d v%d (banks %d %d):  MFMA alone %7.1f | 96 v_fma alone %7.1f | both %7.1f (sum %7.1f)\n", CH0, CH0 + 11, KA,
           KB, KA & 3, KB & 3, t[0] * k, t[1] * k, t[2] * k, (t[0] + t[1]) * k);
}

int main()
{
    float *out;
    CK(hipMalloc(&out, 256 * NT * 4));
    run_cross<8, 0>(out, 256, 4000);
    run_pinned<100, 120, 121>(out);
    run_class<0>("v_fma_f32", out);
    run_class<1>("v_pk_fma_f32", out);
    run_class<7>("v_pk_add_f32", out);
    run_class<2>("v_exp_f32", out);
    run_class<8>("v_rcp_f32", out);
    run_class<3>("v_cvt_pk_bf16_f32", out);
    run_class<4>("v_dot2_f32_bf16", out);
    run_class<5>("v_and_b32", out);
    run_class<6>("v_perm_b32", out);
    run_class<9>("ds_write_b32", out);
    run_class<10>("ds_read_b64", out);
    run_prio<0>("v_fma_f32", out);
    run_prio<1>("v_pk_fma_f32", out);
    run_prio<2>("v_exp_f32", out);
    run_prio<3>("v_cvt_pk_bf16_f32", out);
    run_prio<10>("ds_read_b64", out);
    run_prio<9>("ds_write_b32", out);
    run_same_wave<0>("v_fma_f32", out);
    run_same_wave<1>("v_pk_fma_f32", out);
    run_same_wave<2>("v_exp_f32", out);
    run_cross<8, 0>(out, 256, 4000);
    run_cross<8, 0>(out, 64, 4000);
    run_cross<8, 0>(out, 8, 4000);
    run_cross<8, 0>(out, 256, 100000);
    run_cross<8, 0>(out, 8, 100000);
    run_cross<2, 0>(out, 256, 4000);
    run_cross<1, 0>(out, 256, 4000);
    run_cross<8, 1>(out, 256, 4000);
    run_cross<8, 2>(out, 256, 4000);
    for (int l : {4096, 8192, 65536, 150 * 1024}) {
        g_dynlds = l;
        run_cross<8, 0>(out, 256, 4000);
    }
    g_dynlds = 0;
    run_pinned<1, 13, 14>(out);
    run_pinned<5, 2, 4>(out);
    run_pinned<3, 1, 2>(out);
    run_pinned<20, 2, 4>(out);
    run_pinned<20, 34, 38>(out);
    run_pinned<34, 2, 4>(out);
    run_pinned<34, 46, 47>(out);
    run_pinned<100, 2, 4>(out);
    run_pinned<100, 120, 121>(out);
    run_pinned<100, 120, 122>(out);
    run_pinned<100, 122, 120>(out);
    run_pinned<100, 121, 122>(out);
    run_pinned<100, 121, 123>(out);
    run_pinned<100, 120, 124>(out);
    run_pinned<101, 122, 120>(out);
    run_pinned<102, 122, 120>(out);
    run_pinned<97, 114, 116>(out);
    run_pinned<97, 113, 114>(out);
    run_class<0>("v_fma_f32 (again, late)", out);
    run_cross<8, 0>(out, 256, 4000);
    for (int kb : {512, 3584}) {
        run_stream(out, kb, 4);
        run_stream(out, kb, 8);
    }
    return 0;
}
