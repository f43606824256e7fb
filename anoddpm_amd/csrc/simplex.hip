// OpenSimplex-v1 3-D multi-octave noise on gfx950 -- replaces simplex.py:166-192 (_init, host),
// :202-208 (_extrapolate3), :321-830 (_noise3), :833-840 (_noise3a) and the octave loops of
// :37-54 / :75-93.  Bit-exact with the reference's fp64 arithmetic: this translation unit is
// compiled with -ffp-contract=off (no FMA contraction), divisions are IEEE, and every lattice
// vertex is displaced in the reference's operation order.
//
// Kernel shape: one thread per output pixel, 256-thread blocks covering 64x4 pixel tiles; the
// permutation / gradient-index tables (512 B) and the 24 gradient vectors live in LDS; the octave
// loop runs in registers and the field is written once (8 B or 4 B per pixel).  The work is fp64
// ALU + LDS bound (~200 fp64 ops per octave-evaluation), not HBM bound.
//
// A vertex is an integer lattice offset (i,j,k); its displacement component along an axis is
//     ((d0 - A) - n*SQUISH) - C          n = i+j+k
// with (A,C) = (offset, 0) normally.  Two reference branches build one component as
// "(d0 - 1 - 3*SQUISH) - 1" (simplex.py:503,506) or "(d0 - 2*SQUISH) - 2" (:737-743); they map
// to (A,C) = (offset-1, 1) and (0, 2).  Subtracting +0.0 is exact, so one formula serves all.
#include "common.h"

namespace {

constexpr double STRETCH3 = -1.0 / 6;
constexpr double SQUISH3 = 1.0 / 3;
constexpr double NORM3 = 103.0;

__constant__ signed char kGrad3[72] = {
    -11, 4, 4,  -4, 11, 4,  -4, 4, 11,   11, 4, 4,   4, 11, 4,   4, 4, 11,
    -11,-4, 4,  -4,-11, 4,  -4,-4, 11,   11,-4, 4,   4,-11, 4,   4,-4, 11,
    -11, 4,-4,  -4, 11,-4,  -4, 4,-11,   11, 4,-4,   4, 11,-4,   4, 4,-11,
    -11,-4,-4,  -4,-11,-4,  -4,-4,-11,   11,-4,-4,   4,-11,-4,   4,-4,-11,
};

// LDS tables.  The 24 gradient vectors have components +-4 / +-11: exact in the high dword of a double (low dword 0), so a
// vertex fetches its whole gradient with ONE 16-byte LDS read instead of three 8-byte ones.
struct Tables {
    unsigned char perm[256];
    unsigned char g24[256];          // gradient index perm % 24 (the reference's pgi3 / 3)
    int4 gradhi[24];                 // {hi(gx), hi(gy), hi(gz), 0}
};

__device__ __forceinline__ void load_tables(Tables &T, const int16_t *src)
{
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        T.perm[i] = (unsigned char)src[i];
        T.g24[i] = (unsigned char)(src[256 + i] / 3);
    }
    if (threadIdx.x < 24) {
        const int g = threadIdx.x;
        T.gradhi[g] = make_int4(__double2hiint((double)kGrad3[3 * g]), __double2hiint((double)kGrad3[3 * g + 1]),
                                __double2hiint((double)kGrad3[3 * g + 2]), 0);
    }
    __syncthreads();
}

// x / 103 with IEEE rounding from the correctly rounded reciprocal and two FMAs (Markstein): q = x*r, q' = q + (x - 103 q) r.
// Exact for every normal-range quotient (checked against hardware division on 4e8 samples, incl. random bit patterns); values
// near the under / overflow thresholds take the plain division.
__device__ __forceinline__ double div_norm3(double v)
{
    const double av = fabs(v);
    if (!(av > 1e-280 && av < 1e280)) return v / NORM3;
    const double r = 1.0 / NORM3;
    const double q = v * r;
    return fma(fma(-NORM3, q, v), r, q);
}

struct Vtx {
    int i, j, k;        // lattice offset
    int lx, ly, lz;     // 0 std, 1 "late -1", 2 "late -2" per component
};

__device__ __forceinline__ Vtx mk(int i, int j, int k) { return Vtx{i, j, k, 0, 0, 0}; }

__device__ __forceinline__ double component(double d0, int off, double sq, int late)
{
    const double A = (late == 2) ? 0.0 : (double)(off - (late == 1 ? 1 : 0));
    const double C = (late == 2) ? 2.0 : (late == 1 ? 1.0 : 0.0);
    return ((d0 - A) - sq) - C;
}

__device__ __forceinline__ double vertex_term(const Tables &T, int xsb, int ysb, int zsb,
                                              double dx0, double dy0, double dz0, Vtx v, bool on)
{
    const int n = v.i + v.j + v.k;
    const double sq = (double)n * SQUISH3;          // n*SQUISH formed as one rounded constant
    const double dx = component(dx0, v.i, sq, v.lx);
    const double dy = component(dy0, v.j, sq, v.ly);
    const double dz = component(dz0, v.k, sq, v.lz);
    double attn = 2 - dx * dx - dy * dy - dz * dz;
    // branch-free: the hash chain and the gradient read always run (indices are masked, so always valid); a vertex outside
    // the kernel radius or an unused slot contributes +0.0 exactly as the reference's skipped term does
    const int h0 = T.perm[(xsb + v.i) & 0xFF];
    const int h1 = T.perm[(h0 + ysb + v.j) & 0xFF];
    const int g = T.g24[(h1 + zsb + v.k) & 0xFF];
    const int4 gh = T.gradhi[g];
    const double gx = __hiloint2double(gh.x, 0), gy = __hiloint2double(gh.y, 0), gz = __hiloint2double(gh.z, 0);
    attn = on ? fmax(attn, 0.0) : 0.0;              // out of radius / unused: +0.0, a signed-zero term leaves the sum unchanged
    attn *= attn;
    return attn * attn * (gx * dx + gy * dy + gz * dz);
}

__device__ __forceinline__ double noise3(const Tables &T, double x, double y, double z)
{
    const double stretch = (x + y + z) * STRETCH3;
    const double xs = x + stretch, ys = y + stretch, zs = z + stretch;
    const double fx = floor(xs), fy = floor(ys), fz = floor(zs);
    // lattice base: only (xsb + i) & 0xFF reaches the hash, and (double)(xsb + ysb + zsb) == fx + fy + fz exactly while the
    // floors stay below 2^50; beyond 2^31 (never in practice) the low bits come from the 64-bit conversion
    int xsb, ysb, zsb;
    if (fabs(fx) < 2147483000.0 && fabs(fy) < 2147483000.0 && fabs(fz) < 2147483000.0) {
        xsb = (int)fx; ysb = (int)fy; zsb = (int)fz;
    } else {
        xsb = (int)((long long)fx & 0xFF); ysb = (int)((long long)fy & 0xFF); zsb = (int)((long long)fz & 0xFF);
    }
    const double squish = ((fx + fy) + fz) * SQUISH3;
    const double xb = fx + squish, yb = fy + squish, zb = fz + squish;
    const double xins = xs - fx, yins = ys - fy, zins = zs - fz;
    const double in_sum = xins + yins + zins;
    const double dx0 = x - xb, dy0 = y - yb, dz0 = z - zb;

    Vtx e0 = mk(0, 0, 0), e1 = mk(0, 0, 0);
    int body;       // bit c set: cube corner c (bit0 = +x, bit1 = +y, bit2 = +z) is on this region's vertex list

    if (in_sum <= 1) {                       // tetrahedron at (0,0,0): simplex.py:354-468
        int ap = 1, bp = 2;
        double as = xins, bs = yins;
        if (as >= bs && zins > bs) { bs = zins; bp = 4; }
        else if (as < bs && zins > as) { as = zins; ap = 4; }
        const double wins = 1 - in_sum;
        if (wins > as || wins > bs) {
            const int c = (bs > as) ? bp : ap;
            if (c & 1) { e0.i = 1; e1.i = 1; } else { e0.i = -1; e1.i = 0; }
            if (c & 2) { e0.j = 1; e1.j = 1; } else if (c & 1) { e0.j = -1; } else { e1.j = -1; }
            if (c & 4) { e0.k = 1; e1.k = 1; } else { e1.k = -1; }
        } else {
            const int c = ap | bp;
            e0 = mk(c & 1, (c >> 1) & 1, (c >> 2) & 1);
            e1 = mk((c & 1) ? 1 : -1, (c & 2) ? 1 : -1, (c & 4) ? 1 : -1);
        }
        body = 0x17;                         // corners 0, 1, 2, 4
    } else if (in_sum >= 2) {                // tetrahedron at (1,1,1): simplex.py:469-586
        int ap = 6, bp = 5;
        double as = xins, bs = yins;
        if (as <= bs && zins < bs) { bs = zins; bp = 3; }
        else if (as > bs && zins < as) { as = zins; ap = 3; }
        const double wins = 3 - in_sum;
        if (wins < as || wins < bs) {
            const int c = (bs < as) ? bp : ap;
            if (c & 1) { e0.i = 2; e1.i = 1; }
            if (c & 2) {
                e0.j = 1; e1.j = 1;
                if (c & 1) { e1.j = 2; e1.ly = 1; } else { e0.j = 2; e0.ly = 1; }
            }
            if (c & 4) { e0.k = 1; e1.k = 2; }
        } else {
            const int c = ap & bp;
            e0 = mk(c & 1, (c >> 1) & 1, (c >> 2) & 1);
            e1 = mk(2 * (c & 1), 2 * ((c >> 1) & 1), 2 * ((c >> 2) & 1));
        }
        body = 0xE8;                         // corners 3, 5, 6, 7
    } else {                                 // octahedron: simplex.py:587-798
        double as, bs;
        int ap, bp;
        bool af, bf;
        const double p1 = xins + yins, p2 = xins + zins, p3 = yins + zins;
        if (p1 > 1) { as = p1 - 1; ap = 3; af = true; } else { as = 1 - p1; ap = 4; af = false; }
        if (p2 > 1) { bs = p2 - 1; bp = 5; bf = true; } else { bs = 1 - p2; bp = 2; bf = false; }
        if (p3 > 1) {
            const double sc = p3 - 1;
            if (as <= bs && as < sc) { ap = 6; af = true; }
            else if (as > bs && bs < sc) { bp = 6; bf = true; }
        } else {
            const double sc = 1 - p3;
            if (as <= bs && as < sc) { ap = 1; af = false; }
            else if (as > bs && bs < sc) { bp = 1; bf = false; }
        }
        if (af == bf) {
            if (af) {
                const int c = ap & bp;
                e0 = mk(1, 1, 1);
                e1 = (c & 1) ? mk(2, 0, 0) : (c & 2) ? mk(0, 2, 0) : mk(0, 0, 2);
            } else {
                const int c = ap | bp;
                e1 = !(c & 1) ? mk(-1, 1, 1) : !(c & 2) ? mk(1, -1, 1) : mk(1, 1, -1);
            }
        } else {
            const int c1 = af ? ap : bp;
            const int c2 = af ? bp : ap;
            e0 = !(c1 & 1) ? mk(-1, 1, 1) : !(c1 & 2) ? mk(1, -1, 1) : mk(1, 1, -1);
            if (c2 & 1)      { e1 = mk(2, 0, 0); e1.lx = 2; }
            else if (c2 & 2) { e1 = mk(0, 2, 0); e1.ly = 2; }
            else             { e1 = mk(0, 0, 2); e1.lz = 2; }
        }
        body = 0x7E;                         // corners 1, 2, 4, 3, 5, 6
    }

    // ---- the unit cube's corners.  Every region's vertex list is a subset of the eight corners taken in the order
    // 0,1,2,4,3,5,6,7 (tetra0: 0 1 2 4; octahedron: 1 2 4 3 5 6; tetra1: 3 5 6 7), so all eight are evaluated with COMPILE-TIME
    // offsets -- displacement (d0 - i) - n*SQUISH, hash chain shared per (i) and (i,j) -- and a corner outside the region's
    // list contributes +0.0, exactly like an out-of-radius vertex.  This replaces per-slot decoding of runtime vertex codes.
    const double X[2] = {dx0, dx0 - 1.0}, Y[2] = {dy0, dy0 - 1.0}, Z[2] = {dz0, dz0 - 1.0};
    const double SQ[4] = {0.0, 1.0 * SQUISH3, 2.0 * SQUISH3, 3.0 * SQUISH3};
    int h0[2], h1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) h0[i] = T.perm[(xsb + i) & 0xFF];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) h1[i][j] = T.perm[(h0[i] + ysb + j) & 0xFF];
    double value = 0.0;
    constexpr int ORDER[8] = {0, 1, 2, 4, 3, 5, 6, 7};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int code = ORDER[s];
        const bool listed = (body >> code) & 1;
        const int i = code & 1, j = (code >> 1) & 1, k = (code >> 2) & 1, n = i + j + k;
        const double dx = n ? X[i] - SQ[n] : X[i];
        const double dy = n ? Y[j] - SQ[n] : Y[j];
        const double dz = n ? Z[k] - SQ[n] : Z[k];
        double attn = 2 - dx * dx - dy * dy - dz * dz;
        const int g = T.g24[(h1[i][j] + zsb + k) & 0xFF];
        const int4 gh = T.gradhi[g];
        const double gx = __hiloint2double(gh.x, 0), gy = __hiloint2double(gh.y, 0), gz = __hiloint2double(gh.z, 0);
        // attn <= 0 or an unlisted corner -> attn := +0.0, and 0^4 * (g . d) is a signed zero that leaves the sum unchanged
        attn = listed ? fmax(attn, 0.0) : 0.0;
        attn *= attn;
        value += attn * attn * (gx * dx + gy * dy + gz * dz);
    }
    value += vertex_term(T, xsb, ysb, zsb, dx0, dy0, dz0, e0, true);
    value += vertex_term(T, xsb, ysb, zsb, dx0, dy0, dz0, e1, true);
    return div_norm3(value);
}

template <typename OutT>
__global__ __launch_bounds__(256) void simplex3_octaves_kernel(anoddpm_simplex_args a)
{
    __shared__ Tables T;
    const int s = blockIdx.z;
    long long tab = (a.table_sel ? (long long)(*a.table_sel) * a.table_sel_scale : 0) +
                    (long long)s * a.table_slice_stride;
    load_tables(T, a.tables + tab * 512);

    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const long long zi = a.zvals ? a.zvals[s] : a.z0 + s;

    double acc = 0.0, amp = 1.0, f = a.frequency;
    // x / f: when f is a power of two (every frequency the reference uses: 64, 2^i, and their halvings) the quotient is an
    // exact scaling and equals x * (1/f) bit for bit; any other f takes the IEEE division
    int fe;
    const bool pow2 = (frexp(f, &fe) == 0.5) && fe > -900 && fe - a.octaves > -900 && fe < 900;
    for (int o = 0; o < a.octaves; ++o) {
        double cx, cy, cz;
        if (pow2) { const double rf = 1.0 / f; cx = (double)x * rf; cy = (double)y * rf; cz = (double)zi * rf; }
        else      { cx = (double)x / f; cy = (double)y / f; cz = (double)zi / f; }
        const double n = noise3(T, cx, cy, cz);
        acc = acc + amp * n;            // noise += amplitude * field, octave 0 first (simplex.py:90)
        f = f / 2;
        amp = amp * a.persistence;
    }
    OutT *out = reinterpret_cast<OutT *>(a.out) + (long long)s * a.out_slice_stride + (long long)y * a.W + x;
    *out = (OutT)acc;
}

// simplex.py:833-840 (_noise3a): arbitrary coordinate vectors, out[z][y][x] = noise3(X[x], Y[y], Z[z]).
__global__ __launch_bounds__(256) void simplex3_grid_kernel(double *out, const double *X, int nx, const double *Y, int ny,
                                                            const double *Z, int nz, const int16_t *tables)
{
    __shared__ Tables T;
    load_tables(T, tables);
    const long long total = (long long)nx * ny * nz;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ix = (int)(i % nx);
        const int iy = (int)((i / nx) % ny);
        const int iz = (int)(i / ((long long)nx * ny));
        out[i] = noise3(T, X[ix], Y[iy], Z[iz]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// 2-D OpenSimplex (simplex.py:194-199, 211-318; rand_2d_octaves :56-73).  Vertex = lattice offset (i, j), displacement
// (d0 - i) - n*SQUISH2 with n = i + j (0 for the (1,-1) / (-1,1) extras); accumulation order (1,0), (0,1), base,
// extra -- bit-identical to the reference's fp64 arithmetic (this file is compiled with -ffp-contract=off).
constexpr double STRETCH2 = -0.211324865405187;
constexpr double SQUISH2 = 0.366025403784439;
constexpr double NORM2 = 47.0;
__constant__ signed char kGrad2[16] = {5, 2, 2, 5, -5, 2, -2, 5, 5, -2, 2, -5, -5, -2, -2, -5};

struct Tables2 {
    unsigned char perm[256];
    double grad[16];
};

__device__ __forceinline__ double term2(const Tables2 &T, long long xsb, long long ysb, double dx0, double dy0, int i, int j)
{
    const double sq = (double)(i + j) * SQUISH2;
    const double dx = (dx0 - (double)i) - sq;
    const double dy = (dy0 - (double)j) - sq;
    double attn = 2 - dx * dx - dy * dy;
    double r = 0.0;
    if (attn > 0) {
        const int h0 = T.perm[(int)((xsb + i) & 0xFF)];
        const int idx = T.perm[(int)((h0 + ysb + j) & 0xFF)] & 0x0E;
        attn *= attn;
        r = attn * attn * (T.grad[idx] * dx + T.grad[idx + 1] * dy);
    }
    return r;
}

__device__ double noise2(const Tables2 &T, double x, double y)
{
    const double stretch = (x + y) * STRETCH2;
    const double xs = x + stretch, ys = y + stretch;
    const double fx = floor(xs), fy = floor(ys);
    const long long xsb = (long long)fx, ysb = (long long)fy;
    const double squish = (double)(xsb + ysb) * SQUISH2;
    const double xins = xs - fx, yins = ys - fy;
    const double in_sum = xins + yins;
    const double dx0 = x - (fx + squish), dy0 = y - (fy + squish);
    double value = 0.0;
    value += term2(T, xsb, ysb, dx0, dy0, 1, 0);
    value += term2(T, xsb, ysb, dx0, dy0, 0, 1);
    int bi, bj, ei, ej;
    if (in_sum <= 1) {
        const double zins = 1 - in_sum;
        bi = 0; bj = 0;
        if (zins > xins || zins > yins) {
            if (xins > yins) { ei = 1; ej = -1; } else { ei = -1; ej = 1; }
        } else { ei = 1; ej = 1; }
    } else {
        const double zins = 2 - in_sum;
        bi = 1; bj = 1;
        if (zins < xins || zins < yins) {
            if (xins > yins) { ei = 2; ej = 0; } else { ei = 0; ej = 2; }
        } else { ei = 0; ej = 0; }
    }
    value += term2(T, xsb, ysb, dx0, dy0, bi, bj);
    value += term2(T, xsb, ysb, dx0, dy0, ei, ej);
    return value / NORM2;
}

__device__ __forceinline__ void load_tables2(Tables2 &T, const int16_t *tables)
{
    for (int i = threadIdx.x; i < 256; i += blockDim.x) T.perm[i] = (unsigned char)tables[i];
    if (threadIdx.x < 16) T.grad[threadIdx.x] = (double)kGrad2[threadIdx.x];
    __syncthreads();
}

// X == nullptr: octave mode, out[i][j] = sum_o p^o * noise2(j / f_o, i / f_o)   (rand_2d_octaves on an n x n field)
// X != nullptr: grid mode,   out[i][j] = noise2(X[j], Y[i])                      (_noise2a on a square grid)
__global__ __launch_bounds__(256) void simplex2_kernel(double *out, const double *X, const double *Y, int n,
                                                       const int16_t *tables, int octaves, double persistence, double frequency)
{
    __shared__ Tables2 T;
    load_tables2(T, tables);
    const long long total = (long long)n * n;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long long)gridDim.x * 256) {
        const int i = (int)(k / n), j = (int)(k % n);
        if (X) {
            out[k] = noise2(T, X[j], Y[i]);
        } else {
            double acc = 0.0, amp = 1.0, f = frequency;
            for (int o = 0; o < octaves; ++o) {
                acc = acc + amp * noise2(T, (double)j / f, (double)i / f);
                f = f / 2;
                amp = amp * persistence;
            }
            out[k] = acc;
        }
    }
}

}  // namespace

extern "C" int anoddpm_simplex3_grid_f64(double *out, const double *X, int32_t nx, const double *Y, int32_t ny,
                                         const double *Z, int32_t nz, const int16_t *tables, void *stream)
{
    ANODDPM_REQUIRE(nx >= 0 && ny >= 0 && nz >= 0, "simplex3_grid: negative size");
    const long long total = (long long)nx * ny * nz;
    if (total == 0) return ANODDPM_OK;
    ANODDPM_REQUIRE(out && X && Y && Z && tables, "simplex3_grid: null pointer");
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(simplex3_grid_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0,
                       anoddpm::as_stream(stream), out, X, nx, Y, ny, Z, nz, tables);
    return anoddpm::check_launch("simplex3_grid");
}

extern "C" int anoddpm_simplex_perm_init(int64_t seed, int16_t *perm, int16_t *pgi3)
{
    ANODDPM_REQUIRE(perm && pgi3, "simplex_perm_init: null table pointer");
    int16_t source[256];
    for (int i = 0; i < 256; ++i) source[i] = (int16_t)i;
    uint64_t s = (uint64_t)seed;
    const uint64_t MUL = 6364136223846793005ULL, INC = 1442695040888963407ULL;
    for (int w = 0; w < 3; ++w) s = s * MUL + INC;            // three warm-up rounds (simplex.py:181-183)
    for (int i = 255; i >= 0; --i) {
        s = s * MUL + INC;                                    // int64 wrap-around
        __int128 wide = (__int128)(int64_t)s + 31;            // "+31" does NOT wrap (Python int)
        __int128 m = wide % (i + 1);
        if (m < 0) m += i + 1;                                // floor-mod
        const int r = (int)m;
        perm[i] = source[r];
        pgi3[i] = (int16_t)((perm[i] % 24) * 3);
        source[r] = source[i];
    }
    return ANODDPM_OK;
}

template <typename OutT>
static int launch_simplex(const anoddpm_simplex_args *a, void *stream)
{
    ANODDPM_REQUIRE(a && a->out && a->tables, "simplex3_octaves: null pointer");
    ANODDPM_REQUIRE(a->nslices >= 0 && a->H >= 0 && a->W >= 0 && a->octaves >= 0, "simplex3_octaves: negative size");
    ANODDPM_REQUIRE(a->nslices <= 65535, "simplex3_octaves: nslices > 65535 (split the launch)");
    ANODDPM_REQUIRE(a->out_slice_stride >= (int64_t)a->H * a->W, "simplex3_octaves: slice stride < H*W");
    if (a->nslices == 0 || a->H == 0 || a->W == 0) return ANODDPM_OK;
    dim3 grid((a->W + 63) / 64, (a->H + 3) / 4, a->nslices);
    ANODDPM_REQUIRE(grid.y <= 65535, "simplex3_octaves: H too large");
    hipLaunchKernelGGL(simplex3_octaves_kernel<OutT>, grid, dim3(256), 0, anoddpm::as_stream(stream), *a);
    return anoddpm::check_launch("simplex3_octaves");
}

extern "C" int anoddpm_simplex3_octaves_f64(const anoddpm_simplex_args *a, void *stream)
{
    return launch_simplex<double>(a, stream);
}

extern "C" int anoddpm_simplex3_octaves_f32(const anoddpm_simplex_args *a, void *stream)
{
    return launch_simplex<float>(a, stream);
}

static int launch_simplex2(double *out, const double *X, const double *Y, int32_t n, const int16_t *tables, int32_t octaves,
                           double persistence, double frequency, void *stream, const char *what)
{
    ANODDPM_REQUIRE(n >= 0 && octaves >= 0, "simplex2: negative size");
    if (n == 0) return ANODDPM_OK;
    ANODDPM_REQUIRE(out && tables, "simplex2: null pointer");
    const long long blocks = ((long long)n * n + 255) / 256;
    hipLaunchKernelGGL(simplex2_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0,
                       anoddpm::as_stream(stream), out, X, Y, n, tables, octaves, persistence, frequency);
    return anoddpm::check_launch(what);
}

extern "C" int anoddpm_simplex2_octaves_f64(double *out, int32_t n, const int16_t *tables, int32_t octaves,
                                            double persistence, double frequency, void *stream)
{
    return launch_simplex2(out, nullptr, nullptr, n, tables, octaves, persistence, frequency, stream, "simplex2_octaves");
}

extern "C" int anoddpm_simplex2_grid_f64(double *out, const double *X, const double *Y, int32_t n, const int16_t *tables,
                                         void *stream)
{
    ANODDPM_REQUIRE(n == 0 || (X && Y), "simplex2_grid: null coordinate vector");
    return launch_simplex2(out, X, Y, n, tables, 0, 0.0, 1.0, stream, "simplex2_grid");
}
