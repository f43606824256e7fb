/*
 * oracle/simplex_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, IEEE-754 double, no FMA contraction) of the OpenSimplex-v1 3-D
 * noise path of the AnoDDPM reference.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the shipped path (anoddpm_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_simplex.py checks every function here bit-for-bit
 * (uint64 view of the doubles) against tests/golden/simplex_*.npz, which were produced by
 * importing the reference itself (tests/golden/make_golden.py).
 *
 * What is restated (reference file:line):
 *   oracle_simplex_init            simplex.py:166-192  (overflow, _init)
 *   lattice_gradient_dot           simplex.py:202-208  (_extrapolate3) + :116-127 (GRADIENTS3)
 *   oracle_noise3                  simplex.py:321-830  (_noise3)
 *   oracle_noise3_grid             simplex.py:833-840  (_noise3a)
 *   oracle_rand_3d_octaves         simplex.py:37-54
 *   oracle_rand_3d_fixed_T_octaves simplex.py:75-93
 *
 * Formulation.  The reference spells out every lattice vertex of the three regions of the
 * simplectic honeycomb by hand.  Here a vertex is just its integer lattice offset (i,j,k):
 * its displacement from the sample point is  (d0 - i) - (i+j+k)*SQUISH  evaluated in exactly
 * that order, which is the order the reference's expressions `d0 - i - n*SQUISH` use.  Two of
 * the reference's branches build a component in a different order (a `-= 1` / `-= 2` applied
 * after the squish term, simplex.py:503,506 and :737,740,743); those are flagged per vertex
 * (ORDER_LATE1 / ORDER_LATE2) so the rounding is reproduced bit-for-bit.  Contributions are
 * accumulated in the reference's order (region vertices first, then the two extra vertices);
 * a vertex outside the kernel radius adds +0.0, which leaves the running sum unchanged.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define STRETCH3 (-1.0 / 6)   /* simplex.py:156 */
#define SQUISH3  (1.0 / 3)    /* simplex.py:157 */
#define NORM3    103.0        /* simplex.py:162 */

/* simplex.py:116-127: 24 gradient directions, stored flat (index = 3*g). */
static const double GRAD3[72] = {
    -11, 4, 4,  -4, 11, 4,  -4, 4, 11,
     11, 4, 4,   4, 11, 4,   4, 4, 11,
    -11,-4, 4,  -4,-11, 4,  -4,-4, 11,
     11,-4, 4,   4,-11, 4,   4,-4, 11,
    -11, 4,-4,  -4, 11,-4,  -4, 4,-11,
     11, 4,-4,   4, 11,-4,   4, 4,-11,
    -11,-4,-4,  -4,-11,-4,  -4,-4,-11,
     11,-4,-4,   4,-11,-4,   4,-4,-11,
};

/* simplex.py:174-192.  64-bit LCG with wrap-around; the `+31` and the modulo are taken in
 * unbounded integer arithmetic (Python ints), with floor-mod semantics. */
void oracle_simplex_init(int64_t seed, int64_t *perm, int64_t *pgi3)
{
    int64_t source[256];
    uint64_t s = (uint64_t)seed;
    const uint64_t MUL = 6364136223846793005ULL, INC = 1442695040888963407ULL;
    for (int i = 0; i < 256; ++i) source[i] = i;
    s = s * MUL + INC;
    s = s * MUL + INC;
    s = s * MUL + INC;
    for (int i = 255; i >= 0; --i) {
        s = s * MUL + INC;
        __int128 wide = (__int128)(int64_t)s + 31;          /* no wrap here (simplex.py:186) */
        __int128 m = wide % (i + 1);
        if (m < 0) m += i + 1;                               /* floor-mod */
        int r = (int)m;
        perm[i] = source[r];
        pgi3[i] = (perm[i] % 24) * 3;                        /* simplex.py:190 */
        source[r] = source[i];
    }
}

enum { ORDER_STD = 0, ORDER_LATE1 = 1, ORDER_LATE2 = 2 };

typedef struct {
    int i, j, k;      /* lattice offset from the cell origin                */
    int ox, oy, oz;   /* operation order per component (ORDER_*)            */
} vertex_t;

static inline double displace(double d0, int off, int nsq, int order)
{
    /* n*SQUISH is formed first as one constant, exactly as `n * SQUISH_CONSTANT3`. */
    const double sq = (nsq == 0) ? 0.0 : (double)nsq * SQUISH3;
    if (order == ORDER_LATE1) return ((d0 - (double)(off - 1)) - sq) - 1.0;   /* simplex.py:503,506 */
    if (order == ORDER_LATE2) return (d0 - sq) - 2.0;                         /* simplex.py:730-743 */
    return (d0 - (double)off) - sq;
}

/* simplex.py:202-208 */
static inline double lattice_gradient_dot(const int64_t *perm, const int64_t *pgi3,
                                          int64_t xs, int64_t ys, int64_t zs,
                                          double dx, double dy, double dz)
{
    int64_t idx = pgi3[(perm[(perm[xs & 0xFF] + ys) & 0xFF] + zs) & 0xFF];
    return GRAD3[idx] * dx + GRAD3[idx + 1] * dy + GRAD3[idx + 2] * dz;
}

static inline double vertex_term(const int64_t *perm, const int64_t *pgi3,
                                 int64_t xsb, int64_t ysb, int64_t zsb,
                                 double dx0, double dy0, double dz0, vertex_t v)
{
    const int nsq = v.i + v.j + v.k;
    const double dx = displace(dx0, v.i, nsq, v.ox);
    const double dy = displace(dy0, v.j, nsq, v.oy);
    const double dz = displace(dz0, v.k, nsq, v.oz);
    double attn = 2 - dx * dx - dy * dy - dz * dz;
    if (attn > 0) {
        attn *= attn;
        return attn * attn * lattice_gradient_dot(perm, pgi3, xsb + v.i, ysb + v.j, zsb + v.k, dx, dy, dz);
    }
    return 0.0;
}

static inline vertex_t V(int i, int j, int k) { vertex_t v = {i, j, k, 0, 0, 0}; return v; }
static inline vertex_t bits(int c) { return V(c & 1, (c >> 1) & 1, (c >> 2) & 1); }

/* simplex.py:321-830 */
double oracle_noise3(double x, double y, double z, const int64_t *perm, const int64_t *pgi3)
{
    const double stretch = (x + y + z) * STRETCH3;
    const double xs = x + stretch, ys = y + stretch, zs = z + stretch;
    const int64_t xsb = (int64_t)floor(xs), ysb = (int64_t)floor(ys), zsb = (int64_t)floor(zs);
    const double squish = (double)(xsb + ysb + zsb) * SQUISH3;
    const double xb = (double)xsb + squish, yb = (double)ysb + squish, zb = (double)zsb + squish;
    const double xins = xs - (double)xsb, yins = ys - (double)ysb, zins = zs - (double)zsb;
    const double in_sum = xins + yins + zins;
    const double dx0 = x - xb, dy0 = y - yb, dz0 = z - zb;

    vertex_t body[6];
    int nbody;
    vertex_t e0, e1;

    if (in_sum <= 1) {                                   /* tetrahedron at (0,0,0): :354-468 */
        int a_point = 1, b_point = 2;
        double a_score = xins, b_score = yins;
        if (a_score >= b_score && zins > b_score) { b_score = zins; b_point = 4; }
        else if (a_score < b_score && zins > a_score) { a_score = zins; a_point = 4; }
        const double wins = 1 - in_sum;
        if (wins > a_score || wins > b_score) {          /* origin among the two closest */
            const int c = (b_score > a_score) ? b_point : a_point;
            /* the axis of c gets +1 on both; the other two axes get one -1 each, split over e0/e1 */
            e0 = V(0, 0, 0); e1 = V(0, 0, 0);
            if (c & 1) { e0.i = 1; e1.i = 1; } else { e0.i = -1; e1.i = 0; }
            if (c & 2) { e0.j = 1; e1.j = 1; } else if (c & 1) { e0.j = -1; } else { e1.j = -1; }
            if (c & 4) { e0.k = 1; e1.k = 1; } else { e0.k = 0; e1.k = -1; }
        } else {
            const int c = a_point | b_point;             /* two bits */
            e0 = bits(c);                                /* the far vertex, 2*SQUISH */
            e1 = V((c & 1) ? 1 : -1, (c & 2) ? 1 : -1, (c & 4) ? 1 : -1);   /* 1*SQUISH */
        }
        body[0] = V(0, 0, 0); body[1] = V(1, 0, 0); body[2] = V(0, 1, 0); body[3] = V(0, 0, 1);
        nbody = 4;
    } else if (in_sum >= 2) {                            /* tetrahedron at (1,1,1): :469-586 */
        int a_point = 6, b_point = 5;
        double a_score = xins, b_score = yins;
        if (a_score <= b_score && zins < b_score) { b_score = zins; b_point = 3; }
        else if (a_score > b_score && zins < a_score) { a_score = zins; a_point = 3; }
        const double wins = 3 - in_sum;
        if (wins < a_score || wins < b_score) {          /* (1,1,1) among the two closest */
            const int c = (b_score < a_score) ? b_point : a_point;
            e0 = V(0, 0, 0); e1 = V(0, 0, 0);
            if (c & 1) { e0.i = 2; e1.i = 1; }
            if (c & 2) {
                e0.j = 1; e1.j = 1;
                if (c & 1) { e1.j = 2; e1.oy = ORDER_LATE1; } else { e0.j = 2; e0.oy = ORDER_LATE1; }
            }
            if (c & 4) { e0.k = 1; e1.k = 2; }
        } else {
            const int c = a_point & b_point;             /* one bit */
            e0 = bits(c);
            e1 = V(2 * (c & 1), 2 * ((c >> 1) & 1), 2 * ((c >> 2) & 1));
        }
        body[0] = V(1, 1, 0); body[1] = V(1, 0, 1); body[2] = V(0, 1, 1); body[3] = V(1, 1, 1);
        nbody = 4;
    } else {                                             /* octahedron: :587-798 */
        double a_score, b_score, score;
        int a_point, b_point, a_far, b_far;
        const double p1 = xins + yins, p2 = xins + zins, p3 = yins + zins;
        if (p1 > 1) { a_score = p1 - 1; a_point = 3; a_far = 1; } else { a_score = 1 - p1; a_point = 4; a_far = 0; }
        if (p2 > 1) { b_score = p2 - 1; b_point = 5; b_far = 1; } else { b_score = 1 - p2; b_point = 2; b_far = 0; }
        if (p3 > 1) {
            score = p3 - 1;
            if (a_score <= b_score && a_score < score) { a_point = 6; a_far = 1; }
            else if (a_score > b_score && b_score < score) { b_point = 6; b_far = 1; }
        } else {
            score = 1 - p3;
            if (a_score <= b_score && a_score < score) { a_point = 1; a_far = 0; }
            else if (a_score > b_score && b_score < score) { b_point = 1; b_far = 0; }
        }
        if (a_far == b_far) {
            if (a_far) {
                const int c = a_point & b_point;
                e0 = V(1, 1, 1);
                e1 = (c & 1) ? V(2, 0, 0) : (c & 2) ? V(0, 2, 0) : V(0, 0, 2);
            } else {
                const int c = a_point | b_point;
                e0 = V(0, 0, 0);
                e1 = !(c & 1) ? V(-1, 1, 1) : !(c & 2) ? V(1, -1, 1) : V(1, 1, -1);
            }
        } else {
            const int c1 = a_far ? a_point : b_point;
            const int c2 = a_far ? b_point : a_point;
            e0 = !(c1 & 1) ? V(-1, 1, 1) : !(c1 & 2) ? V(1, -1, 1) : V(1, 1, -1);
            if (c2 & 1)      { e1 = V(2, 0, 0); e1.ox = ORDER_LATE2; }
            else if (c2 & 2) { e1 = V(0, 2, 0); e1.oy = ORDER_LATE2; }
            else             { e1 = V(0, 0, 2); e1.oz = ORDER_LATE2; }
        }
        body[0] = V(1, 0, 0); body[1] = V(0, 1, 0); body[2] = V(0, 0, 1);
        body[3] = V(1, 1, 0); body[4] = V(1, 0, 1); body[5] = V(0, 1, 1);
        nbody = 6;
    }

    double value = 0.0;
    for (int n = 0; n < nbody; ++n)
        value += vertex_term(perm, pgi3, xsb, ysb, zsb, dx0, dy0, dz0, body[n]);
    value += vertex_term(perm, pgi3, xsb, ysb, zsb, dx0, dy0, dz0, e0);
    value += vertex_term(perm, pgi3, xsb, ysb, zsb, dx0, dy0, dz0, e1);
    return value / NORM3;
}

/* simplex.py:833-840: out[z][y][x] = noise3(X[x], Y[y], Z[z]) */
void oracle_noise3_grid(const double *X, int64_t nx, const double *Y, int64_t ny,
                        const double *Z, int64_t nz, const int64_t *perm, const int64_t *pgi3,
                        double *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t iz = 0; iz < nz; ++iz)
        for (int64_t iy = 0; iy < ny; ++iy)
            for (int64_t ix = 0; ix < nx; ++ix)
                out[(iz * ny + iy) * nx + ix] = oracle_noise3(X[ix], Y[iy], Z[iz], perm, pgi3);
}

/* Shared octave loop of simplex.py:37-54 and :75-93.  zvals are the integer z indices
 * (arange(depth) for rand_3d_octaves, the timestep array T for rand_3d_fixed_T_octaves);
 * coordinates are index / frequency in double, frequency halves and amplitude scales by
 * `persistence` after every octave, octave 0 is accumulated first into a zero field. */
void oracle_octaves(const int64_t *zvals, int64_t nz, int64_t height, int64_t width,
                    int octaves, double persistence, double frequency,
                    const int64_t *perm, const int64_t *pgi3, double *out)
{
    const int64_t total = nz * height * width;
    for (int64_t n = 0; n < total; ++n) out[n] = 0.0;
    double amplitude = 1.0;
    for (int o = 0; o < octaves; ++o) {
        const double f = frequency, a = amplitude;
#pragma omp parallel for collapse(2) schedule(static)
        for (int64_t iz = 0; iz < nz; ++iz)
            for (int64_t iy = 0; iy < height; ++iy) {
                const double zc = (double)zvals[iz] / f, yc = (double)iy / f;
                double *row = out + (iz * height + iy) * width;
                for (int64_t ix = 0; ix < width; ++ix)
                    row[ix] = row[ix] + a * oracle_noise3((double)ix / f, yc, zc, perm, pgi3);
            }
        frequency /= 2;
        amplitude *= persistence;
    }
}

/* ------------------------------------------------------------------------------------------
 * 2-D OpenSimplex (simplex.py:194-199 _extrapolate2, :211-308 _noise2, :311-318 _noise2a,
 * :56-73 rand_2d_octaves).  Same formulation as the 3-D restatement: a contributing vertex is
 * its lattice offset (i, j); its displacement is (d0 - i) - n*SQUISH2 with n the number of
 * squish units the reference subtracts for that vertex (i + j; 0 for the (1,-1) / (-1,1)
 * extras, simplex.py:262-271).  Order of accumulation: (1,0), (0,1), base vertex, extra vertex.
 */
#define STRETCH2 (-0.211324865405187)   /* simplex.py:154 */
#define SQUISH2  (0.366025403784439)    /* simplex.py:155 */
#define NORM2    47.0                   /* simplex.py:161 */

static const double GRAD2[16] = {5, 2, 2, 5, -5, 2, -2, 5, 5, -2, 2, -5, -5, -2, -2, -5};   /* :103-111 */

static inline double term2(const int64_t *perm, int64_t xsb, int64_t ysb, double dx0, double dy0, int i, int j)
{
    const double sq = (double)(i + j) * SQUISH2;
    const double dx = (dx0 - (double)i) - sq;
    const double dy = (dy0 - (double)j) - sq;
    double attn = 2 - dx * dx - dy * dy;
    if (!(attn > 0)) return 0.0;
    const int64_t idx = perm[(perm[(xsb + i) & 0xFF] + (ysb + j)) & 0xFF] & 0x0E;
    attn *= attn;
    return attn * attn * (GRAD2[idx] * dx + GRAD2[idx + 1] * dy);
}

double oracle_noise2(double x, double y, const int64_t *perm)
{
    const double stretch = (x + y) * STRETCH2;
    const double xs = x + stretch, ys = y + stretch;
    const double fx = floor(xs), fy = floor(ys);
    const int64_t xsb = (int64_t)fx, ysb = (int64_t)fy;
    const double squish = (double)(xsb + ysb) * SQUISH2;
    const double xins = xs - fx, yins = ys - fy;
    const double in_sum = xins + yins;
    const double dx0 = x - (fx + squish), dy0 = y - (fy + squish);

    double value = 0.0;
    value += term2(perm, xsb, ysb, dx0, dy0, 1, 0);
    value += term2(perm, xsb, ysb, dx0, dy0, 0, 1);
    int bi, bj, ei, ej;                       /* base and extra vertex */
    if (in_sum <= 1) {                        /* triangle at (0,0), simplex.py:260-277 */
        const double zins = 1 - in_sum;
        bi = 0; bj = 0;
        if (zins > xins || zins > yins) {
            if (xins > yins) { ei = 1; ej = -1; } else { ei = -1; ej = 1; }
        } else { ei = 1; ej = 1; }
    } else {                                  /* triangle at (1,1), simplex.py:278-297 */
        const double zins = 2 - in_sum;
        bi = 1; bj = 1;
        if (zins < xins || zins < yins) {
            if (xins > yins) { ei = 2; ej = 0; } else { ei = 0; ej = 2; }
        } else { ei = 0; ej = 0; }
    }
    value += term2(perm, xsb, ysb, dx0, dy0, bi, bj);
    value += term2(perm, xsb, ysb, dx0, dy0, ei, ej);
    return value / NORM2;
}

/* simplex.py:311-318 for SQUARE grids (n x n): out[i][j] = noise2(X[j], Y[i]). */
void oracle_noise2_grid(const double *X, const double *Y, int64_t n, const int64_t *perm, double *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j)
            out[i * n + j] = oracle_noise2(X[j], Y[i], perm);
}

/* simplex.py:56-73 for square shapes: noise[i][j] += amplitude * noise2(j / f, i / f). */
void oracle_octaves2(int64_t n, int octaves, double persistence, double frequency, const int64_t *perm, double *out)
{
    for (int64_t k = 0; k < n * n; ++k) out[k] = 0.0;
    double amplitude = 1.0;
    for (int o = 0; o < octaves; ++o) {
        const double f = frequency, a = amplitude;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i)
            for (int64_t j = 0; j < n; ++j)
                out[i * n + j] = out[i * n + j] + a * oracle_noise2((double)j / f, (double)i / f, perm);
        frequency /= 2;
        amplitude *= persistence;
    }
}
