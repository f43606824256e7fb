// HBM counter calibration for rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section: "calibrate on a
// known byte count in your own access pattern").  Every kernel touches each byte of a 1 GiB buffer (4x the 256 MiB Infinity
// Cache) exactly once, in one of the access patterns the UNet kernels use:
//   read_stream16   16 B per lane, lanes contiguous (1 KiB per wave instruction)            -- the guide's calibrated pattern
//   read_patch16    16 B per lane, 4 lanes = one pixel's 64-byte channel chunk, pixel stride 512 B (C = 128 floats); the other
//                   64-byte half of every 128-byte line is read by a LATER loop iteration    -- wino43 / wino patch staging
//   read_patch16_c256  the same with pixel stride 1024 B (C = 256)
//   write_stream16  16 B per lane contiguous stores                                          -- igemm / pointwise epilogues
//   write_pixel4    4 B per lane, 64 lanes = 256 contiguous bytes of one pixel, 16 pixels per thread with a W*C stride
//                                                                                            -- wino43 epilogue (buffer_store_dword)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/hbm_calib.bin tools/hbm_calib.hip ; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/hbm_calib.bin     and   --pmc WRITE_SIZE   (separate passes)
// tools/traffic_from_pmc.py turns the two counter files into profiles/r3_hbm_calibration.json.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr size_t BYTES = (size_t)1 << 30;

__global__ __launch_bounds__(256) void read_stream16(const f32x4 *p, size_t n16, float *sink)
{
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

// pixels x C floats; a block owns 64 consecutive pixels and walks the channel chunks of 16 floats (64 B) like the conv K loop
template <int C>
__global__ __launch_bounds__(256) void read_patch16(const float *p, size_t npix, float *sink)
{
    f32x4 acc = {0, 0, 0, 0};
    const int quad = threadIdx.x & 3, pl = threadIdx.x >> 2;
    for (size_t p0 = (size_t)blockIdx.x * 64; p0 < npix; p0 += (size_t)gridDim.x * 64) {
        const float *row = p + (p0 + pl) * C + quad * 4;
#pragma unroll 1
        for (int chunk = 0; chunk < C / 16; ++chunk) {
            acc += *reinterpret_cast<const f32x4 *>(row + chunk * 16);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

__global__ __launch_bounds__(256) void write_stream16(f32x4 *p, size_t n16)
{
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = v;
}

// image of W x W pixels x 128 channels; a thread owns (4x4 pixel tile, channel) and writes its 16 pixels one dword at a time
__global__ __launch_bounds__(256) void write_pixel4(float *p, int W, size_t nimg)
{
    const int C = 128;
    const int ch = threadIdx.x & 127, th = threadIdx.x >> 7;                 // two tiles per block
    const int tiles = (W / 4) * (W / 4);
    for (size_t t = (size_t)blockIdx.x * 2 + th; t < nimg * tiles; t += (size_t)gridDim.x * 2) {
        const size_t img = t / tiles;
        const int tt = (int)(t % tiles);
        const int y0 = (tt / (W / 4)) * 4, x0 = (tt % (W / 4)) * 4;
        float *base = p + ((img * W + y0) * W + x0) * C + ch;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) base[((size_t)i * W + j) * C] = (float)(i + j);
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main()
{
    void *buf, *sink;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, BYTES));
    const int grid = 256 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        read_stream16<<<grid, 256>>>((const f32x4 *)buf, BYTES / 16, (float *)sink);
        read_patch16<128><<<grid, 256>>>((const float *)buf, BYTES / (128 * 4), (float *)sink);
        read_patch16<256><<<grid, 256>>>((const float *)buf, BYTES / (256 * 4), (float *)sink);
        write_stream16<<<grid, 256>>>((f32x4 *)buf, BYTES / 16);
        write_pixel4<<<grid, 256>>>((float *)buf, 256, BYTES / ((size_t)256 * 256 * 128 * 4));
        CK(hipDeviceSynchronize());
    }
    printf("hbm_calib: bytes per kernel = %zu\n", BYTES);
    return 0;
}
