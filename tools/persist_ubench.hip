// VERDICT r3 item 1(c), measured instead of priced: is a PERSISTENT section (one launch, grid barriers between dependent phases)
// cheaper than a chain of dependent launches for the geometry of the <= 16x16 section of the denoiser?
//
// Model of a phase (one small-map layer): 256 workgroups of 512 threads, one per CU; workgroup i reads the slab workgroup
// (i + 97) % 256 wrote in the previous phase (a cross-CU, cross-XCD dependency, like a split-K tail or the next layer's operand),
// does WORK fused multiply-adds per element on it and writes its own slab of SLAB bytes.  N phases, result checked.
//   L  N launches captured in one hipGraph, replayed                           (what the plan does today)
//   P  ONE launch, N phases separated by a grid barrier: flat counter barrier with agent-scope release before the arrive and
//      acquire after the wait (MI355X_MICROARCH.md "barrier-counter"), or the XCD-hierarchical form ("barrier-xcd": per-XCD
//      counters, the XCD's last arriver goes to the top counter and releases its XCD through a generation word)
// Prints microseconds per phase.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/persist_ubench.bin tools/persist_ubench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NWG = 256, NT = 512;

template <int WORK>
__device__ __forceinline__ void phase_body(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, int wg, int slab16)
{
    const f32x4 *src = in + (size_t)((wg + 97) % NWG) * slab16;
    f32x4 *dst = out + (size_t)wg * slab16;
    for (int i = threadIdx.x; i < slab16; i += NT) {
        f32x4 v = src[i];
#pragma unroll
        for (int w = 0; w < WORK; ++w) v = v * 1.0000001f + 1e-9f;
        v += 1.0f;
        dst[i] = v;
    }
}

template <int WORK>
__global__ __launch_bounds__(NT) void phase_kernel(const f32x4 *in, f32x4 *out, int slab16)
{
    phase_body<WORK>(in, out, blockIdx.x, slab16);
}

__device__ __forceinline__ unsigned ld_relaxed(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// flat counter barrier: every workgroup arrives on one monotonic counter
__device__ __forceinline__ void barrier_flat(unsigned *counter, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (unsigned spins = 0; ld_relaxed(counter) < target && spins < (1u << 20); ++spins) __builtin_amdgcn_s_sleep(2);   // bounded
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// XCD-hierarchical: 8 per-XCD counters (32 workgroups each: block b runs on XCD b % 8 -- used for speed only, correctness holds
// for any placement because every workgroup releases before it arrives and acquires after it leaves), one top counter, one
// generation word per XCD
__device__ __forceinline__ void barrier_xcd(unsigned *xcnt, unsigned *top, unsigned *gen, unsigned phase)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const int x = blockIdx.x & 7;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(xcnt + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == (phase + 1) * 32 - 1) {                        // last arriver of this XCD group
            __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned spins = 0; ld_relaxed(top) < (phase + 1) * 8 && spins < (1u << 20); ++spins) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(gen + x * 32, phase + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (unsigned spins = 0; ld_relaxed(gen + x * 32) < phase + 1 && spins < (1u << 20); ++spins) __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int WORK, int BAR>
__global__ __launch_bounds__(NT) void persistent_kernel(f32x4 *a, f32x4 *b, int slab16, int nphases, unsigned *sync)
{
    f32x4 *in = a, *out = b;
    for (int p = 0; p < nphases; ++p) {
        phase_body<WORK>(in, out, blockIdx.x, slab16);
        if (BAR == 0) barrier_flat(sync, (unsigned)(p + 1) * NWG);
        else          barrier_xcd(sync + 64, sync + 32, sync + 512, (unsigned)p);
        f32x4 *t = in; in = out; out = t;
    }
}

template <int WORK>
static void run(int slab_bytes, int nphases)
{
    const int slab16 = slab_bytes / 16;
    const size_t n16 = (size_t)NWG * slab16;
    f32x4 *a, *b;
    unsigned *sync;
    CK(hipMalloc(&a, n16 * 16));
    CK(hipMalloc(&b, n16 * 16));
    CK(hipMalloc(&sync, 4096 * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> host(4);
    auto check = [&](const char *what) {
        const f32x4 *res = (nphases & 1) ? b : a;
        CK(hipMemcpy(host.data(), res, 16, hipMemcpyDeviceToHost));
        if (!(host[0] > nphases - 0.5f && host[0] < nphases + 0.5f)) printf("   !! %s: result %f, expected ~%d\n", what, host[0], nphases);
    };
    // L: graph of N dependent launches
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int p = 0; p < nphases; ++p)
        hipLaunchKernelGGL(phase_kernel<WORK>, dim3(NWG), dim3(NT), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, slab16);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best_l = 1e9f, best_p[2] = {1e9f, 1e9f};
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemsetAsync(a, 0, n16 * 16, s));
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best_l = ms < best_l ? ms : best_l;
    }
    check("launch chain");
    for (int bar = 0; bar < 2; ++bar)
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(a, 0, n16 * 16, s));
            CK(hipMemsetAsync(sync, 0, 4096 * 4, s));
            CK(hipEventRecord(e0, s));
            if (bar == 0) hipLaunchKernelGGL((persistent_kernel<WORK, 0>), dim3(NWG), dim3(NT), 0, s, a, b, slab16, nphases, sync);
            else          hipLaunchKernelGGL((persistent_kernel<WORK, 1>), dim3(NWG), dim3(NT), 0, s, a, b, slab16, nphases, sync);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best_p[bar] = ms < best_p[bar] ? ms : best_p[bar];
            if (rep == 5) check(bar ? "persistent, xcd barrier" : "persistent, flat barrier");
        }
    printf("slab %6d B/WG  work %3d fma/elem  phases %3d:  launches %6.2f us/phase   persistent flat barrier %6.2f us/phase   xcd barrier %6.2f us/phase\n",
           slab_bytes, WORK, nphases, best_l * 1000 / nphases, best_p[0] * 1000 / nphases, best_p[1] * 1000 / nphases);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipFree(a));
    CK(hipFree(b));
    CK(hipFree(sync));
}

int main()
{
    int dev_cus = 0;
    CK(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs %d; %d workgroups x %d threads, one per CU\n", dev_cus, NWG, NT);
    if (dev_cus < NWG) { printf("needs %d CUs resident at once\n", NWG); return 0; }
    for (int slab : {2048, 8192, 65536}) {
        run<0>(slab, 64);
        run<64>(slab, 64);
        run<512>(slab, 64);
    }
    return 0;
}
