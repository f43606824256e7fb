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
#include "simplex_tables.h"

namespace {

constexpr double STRETCH3 = -1.0 / 6;
constexpr double SQUISH3 = 1.0 / 3;
constexpr double NORM3 = 103.0;


// LDS tables.  Everything the hash chain touches is stored as BYTE ADDRESSES inside this struct (the kernel's only LDS object, at LDS
// address 0) so that an LDS address is one integer add and every table is reached through an instruction's immediate offset.
// The kernel is LDS-bound as much as VALU-bound (round 6 counters: SQ_LDS_IDX_ACTIVE 92 % of the launch, 45 % of it conflict cycles), so the
// tables are shaped for the LDS-array cycles of each read (guide: ds_read_b64 2 cycles over 64 banks, ds_read_b128 4, ds_read2_b32 4 over
// 32 banks, ds_read2_b64 8):
//   grad[g] = {gx, gy, gz} as doubles (components +-4 / +-11 of GRADIENTS3, simplex.py:116-127) in 48-byte rows at offset 0: a gradient
//           byte offset 48 g IS its address; rows are 16-byte aligned: one ds_read_b128 (gx, gy) + one ds_read_b64 (gz) = 6 cycles
//           (the former ds_read2_b64 + ds_read_b64: 10)
//   H[m]  = {h(m), h(m + 1)},  h(m) = H_ENTRY0 + 8 * perm[m & 255]          m = -1 .. 514: a hash value plus a masked lattice coordinate
//   G[m]  = {48 * pgi(m), 48 * pgi(m + 1)},  pgi = perm-gradient index      plus a lattice offset of -1 .. 2 never wraps.
//           The rows of a lattice-corner PAIR (coordinate c and c + 1 of one chain level) are entries m and m + 1: each table holds
//           the pair in ONE 8-byte entry, so a pair is one ds_read_b64 (2 cycles, 64 banks) instead of a ds_read2_b32 (4 cycles, 32
//           banks).  Entry 0 of H sits at byte H_ENTRY0 = 2048 and entry 0 of G at H_ENTRY0 + G_DELTA, both multiples of 2048: the x
//           coordinate's masked offset takes H's base and the z coordinate's takes G_DELTA with the same bit-field insert that masks
//           them, and every hash VALUE carries H's base -- the chain's next address is hash + masked coordinate.
//   lut[idx] = byte addresses of the two extra vertices' rows (lo / hi half) for the region index the comparison bits form
//   vtx[id] = displacement recipe / hash offsets of a vertex: {ax, ay, az, sq, cx, cy, cz, (i8 | j8 << 32), (k8), -}
// (lut / vtx: from the generated tables of simplex_tables.h.)
constexpr unsigned H_ENTRY0 = 2048, G_DELTA = 6144;
struct Tables {
    double grad[24][6];              // 1152 B at offset 0
    unsigned char pad0[H_ENTRY0 - 8 - 24 * 48];
    uint2 H[516];                    // entry m (m = -1 .. 514) at H[m + 1]: the extra vertices' offsets -1 .. 2 need no wrap-around mask
    unsigned lut[REGION_LUT_SIZE];
    unsigned char pad1[G_DELTA - 516 * 8 - REGION_LUT_SIZE * 4];
    uint2 G[516];
    alignas(16) double vtx[128][10];   // rows are read with ds_read_b128: 16-byte aligned (an 8-byte-aligned base halves the kernel's speed)
};
static_assert(offsetof(Tables, vtx) % 16 == 0, "vertex rows 16-byte aligned");
static_assert(offsetof(Tables, vtx) == VTX_LDS_BASE, "kRegionLutAddr holds addresses relative to this base");
static_assert(offsetof(Tables, lut) % 8 == 0 && REGION_LUT_SIZE % 2 == 0, "lut copied as 8-byte words");
static_assert(offsetof(Tables, H) == H_ENTRY0 - 8, "H entry 0 at byte 2048");
static_assert(offsetof(Tables, G) == H_ENTRY0 - 8 + G_DELTA, "G entry 0 at byte 2048 + 6144");
static_assert((H_ENTRY0 & 0x7F8u) == 0 && (G_DELTA & 0x7F8u) == 0, "bases outside the coordinate mask");
static_assert(offsetof(Tables, vtx) + 128 * 80 < 65536, "vertex row addresses fit 16 bits");

__device__ __forceinline__ void load_tables(Tables &T, const int16_t *src)
{
    for (int i = threadIdx.x; i < 516; i += blockDim.x) {
        const int m = (i - 1) & 255, m1 = i & 255;
        T.H[i] = make_uint2(H_ENTRY0 + 8u * (unsigned)(src[m] & 0xFF), H_ENTRY0 + 8u * (unsigned)(src[m1] & 0xFF));
        T.G[i] = make_uint2(16u * (unsigned)src[256 + m], 16u * (unsigned)src[256 + m1]);     // 48 * (pgi3 / 3): pgi3 is a multiple of 3
    }
    // the seed-independent tables come from simplex_tables.h in exactly this layout: plain copies
    for (int i = threadIdx.x; i < 24 * 48 / 16; i += blockDim.x)
        reinterpret_cast<uint4 *>(T.grad)[i] = reinterpret_cast<const uint4 *>(kGradRows)[i];
    for (int i = threadIdx.x; i < REGION_LUT_SIZE / 2; i += blockDim.x)
        reinterpret_cast<uint2 *>(T.lut)[i] = reinterpret_cast<const uint2 *>(kRegionLutAddr)[i];
    for (int i = threadIdx.x; i < 128 * 80 / 16; i += blockDim.x)
        reinterpret_cast<uint4 *>(T.vtx)[i] = reinterpret_cast<const uint4 *>(kVertexRows)[i];
    __syncthreads();
}

// x / 103 with IEEE rounding from the correctly rounded reciprocal and two FMAs (Markstein): q = x*r, q' = q + (x - 103 q) r.
// Exact for every normal-range quotient (checked against hardware division on 4e8 samples, incl. random bit patterns) and for
// +-0 (q = q' = the zero).  A noise value is 0 or a sum of terms attn^4 (g . d) with attn >= 2^-52 and |sum| < 1e4, far inside
// the range where the residual x - 103 q is exact; NaN / inf / subnormal inputs (coordinates beyond the double range) take the
// plain division -- one v_cmp_class_f64 instead of two range comparisons.
__device__ __forceinline__ double div_norm3(double v)
{
    // classes: signaling / quiet NaN (0, 1), -inf (2), -subnormal (4), +subnormal (7), +inf (9)
    if (__builtin_amdgcn_class(v, 0x1 | 0x2 | 0x4 | 0x10 | 0x80 | 0x200)) return v / NORM3;
    const double r = 1.0 / NORM3;
    const double q = v * r;
    return fma(fma(-NORM3, q, v), r, q);
}

// LDS accesses by byte address inside the Tables struct (LDS address 0: the compiler folds `&T`)
__device__ __forceinline__ uint2 lds_pair(const Tables &T, unsigned addr)      // {entry m, entry m + 1} of H or G: one ds_read_b64
{
    return *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(&T) + addr);
}
__device__ __forceinline__ unsigned lds_one(const Tables &T, unsigned addr)      // entry m alone (the extra vertices)
{
    return *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(&T) + addr);
}

// attn^4 * (g . d) of one vertex whose displacement and gradient byte offset are known (simplex.py:202-208 + the kernel term)
__device__ __forceinline__ double kernel_term(const Tables &T, double two, double dx, double dy, double dz, unsigned goff)
{
    const char *g = reinterpret_cast<const char *>(&T) + goff;     // grad rows start at address 0; 48-byte rows: 16-byte aligned
    const double2 gxy = *reinterpret_cast<const double2 *>(g);
    const double gx = gxy.x, gy = gxy.y, gz = *reinterpret_cast<const double *>(g + 16);
    double attn = two - dx * dx - dy * dy - dz * dz;                 // `two` = 2.0, or -inf for a corner outside the region's list
    attn = fmax(attn, 0.0);                                         // out of radius / unlisted: +0.0, the term is a signed zero
    attn *= attn;
    return attn * attn * (gx * dx + gy * dy + gz * dz);
}

// ABL (timing ablations only, wrong results; ANODDPM_DEBUG7): 1 no extra vertices, 2 no region logic, 3 four corners instead of eight
// SAFE: the caller guarantees |floor(coordinate)| < 2^31 (the octave kernels check their largest coordinate once per thread)
template <int ABL = 0, bool SAFE = false>
__device__ __forceinline__ double noise3(const Tables &T, double x, double y, double z)
{
    const double stretch = (x + y + z) * STRETCH3;
    const double xs = x + stretch, ys = y + stretch, zs = z + stretch;
    const double fx = floor(xs), fy = floor(ys), fz = floor(zs);
    // lattice base: only (xsb + i) & 0xFF reaches the hash, as the byte offset 8 * ((xsb + i) & 0xFF) of a table entry.  SAFE (floors below
    // 2^31): floor + 1.5 * 2^49 has an ulp of 1/8, so the low word of the sum IS 8 * xsb in two's complement -- one exact fp64 add
    // replaces convert + shift (round 6).  Beyond 2^31 (never in practice) the low bits come from the 64-bit conversion.  The
    // x offset takes H's base address and the z offset the distance from H to G with the mask (v_bfi_b32); the y offset is added to hash values that carry H's base.
    unsigned xb2, yb2, zb2;
    if (SAFE) {
        constexpr double MAGIC = 844424930131968.0;                // 1.5 * 2^49
        // (lo & 0x7F8) | 0x800 as ONE bit-field insert (hipcc turns the plain expression into v_and + v_add)
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(xb2) : "s"(0x7F8u), "v"(__double2loint(fx + MAGIC)), "v"(H_ENTRY0));
        // (as asm: hipcc's SLP pass packs the two 16-bit masks into one register -- v_perm + two ands + a shift for two ands)
        asm("v_and_b32 %0, %1, %2" : "=v"(yb2) : "s"(0x7F8u), "v"(__double2loint(fy + MAGIC)));
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(zb2) : "s"(0x7F8u), "v"(__double2loint(fz + MAGIC)), "v"(G_DELTA));
    } else {
        int xsb, ysb, zsb;
        if (fabs(fx) < 2147483000.0 && fabs(fy) < 2147483000.0 && fabs(fz) < 2147483000.0) {
            xsb = (int)fx; ysb = (int)fy; zsb = (int)fz;
        } else {
            xsb = (int)((long long)fx & 0xFF); ysb = (int)((long long)fy & 0xFF); zsb = (int)((long long)fz & 0xFF);
        }
        xb2 = (((unsigned)xsb & 0xFFu) * 8u) | H_ENTRY0; yb2 = ((unsigned)ysb & 0xFFu) * 8u; zb2 = (((unsigned)zsb & 0xFFu) * 8u) | G_DELTA;
    }
    const double squish = ((fx + fy) + fz) * SQUISH3;
    const double xb = fx + squish, yb = fy + squish, zb = fz + squish;
    const double xins = xs - fx, yins = ys - fy, zins = zs - fz;
    const double in_sum = xins + yins + zins;
    const double dx0 = x - xb, dy0 = y - yb, dz0 = z - zb;

    // ---- region decisions of simplex.py:354-798, branch-free.  The reference's if / elif chains depend on nineteen comparisons of the
    // inside coordinates only; their results are gathered into a region index and ONE table lookup (kRegionLut, generated from the
    // chains themselves by tools/gen_simplex_tables.py) yields the two extra vertices -- no score is materialised, no decision code is
    // built from selects.  Round 6, second form: a comparison a < b IS the sign bit of the fp64 difference a - b (the difference of two
    // distinct doubles is never zero, a - a is +0.0), and one v_alignbit_b32 shifts that bit into the index -- two instructions per
    // comparison instead of compare + select + or through an SGPR pair (106 -> 65 -> 40 VALU instructions for this block).
    //   * The tetrahedron at (1,1,1) asks the MIRRORED questions of the tetrahedron at (0,0,0) (x <= y for x >= y, w < x for w > x, ...):
    //     with every operand negated -- an exact sign flip -- they are the same seven differences, so both share one chain; its seed is the region bit.
    //   * Octahedron: the scores |p - 1| come from r = 1 - p (the exact negative of p - 1; p > 1 is r's sign); "<=" is the complement of
    //     ">" on the same pair and needs no bit of its own.
    const bool regA = in_sum <= 1, regB = in_sum >= 2;
    unsigned idx;
    if (ABL == 2) {
        idx = 256;
    } else {
        auto push = [](unsigned acc, double d) { return __builtin_amdgcn_alignbit(acc, (unsigned)__double2hiint(d), 31u); };
        // mirror = times -1.0 (exact; one v_mul_f64 -- a sign flip through the high word costs an extra move per 64-bit operand)
        const double sgn = __hiloint2double(regB ? (int)0xBFF00000 : 0x3FF00000, 0);
        const double xm = xins * sgn, ym = yins * sgn, zm = zins * sgn;
        // w = 1 - in_sum (tetrahedron at the origin) / the negative of w = 3 - in_sum.  in_sum * sgn is exact, so the explicit fma
        // rounds once, exactly like the subtraction it stands for: -3 + in_sum == -(3 - in_sum)
        const double wm = fma(in_sum, -sgn, __hiloint2double(regB ? (int)0xC0080000 : 0x3FF00000, 0));
        unsigned iT = regB ? 1u : 0u;
        iT = push(iT, xm - ym);                                     // x < y   (origin: the complement of x >= y; far corner: of x <= y)
        iT = push(iT, ym - zm);                                     // z > y   (far corner: z < y)
        iT = push(iT, xm - zm);                                     // z > x   (z < x)
        iT = push(iT, zm - ym);                                     // z < y   (z > y)
        iT = push(iT, xm - wm);                                     // w > x   (w < x)
        iT = push(iT, ym - wm);                                     // w > y   (w < y)
        iT = push(iT, zm - wm);                                     // w > z   (w < z)
        const double r1 = 1 - (xins + yins), r2 = 1 - (xins + zins), r3 = 1 - (yins + zins);
        unsigned iO = 4u;                                           // 4 << 6 = 256: the octahedron's entries follow the two tetrahedra's
        iO = push(iO, r1);                                          // p1 > 1
        iO = push(iO, r2);                                          // p2 > 1
        iO = push(iO, r3);                                          // p3 > 1
        iO = push(iO, fabs(r2) - fabs(r1));                         // |p1 - 1| >  |p2 - 1|   (complement: <=)
        iO = push(iO, fabs(r1) - fabs(r3));                         // |p1 - 1| <  |p3 - 1|
        iO = push(iO, fabs(r2) - fabs(r3));                         // |p2 - 1| <  |p3 - 1|
        idx = (regA || regB) ? iT : iO;
    }
    const unsigned pair = T.lut[idx];                               // byte addresses of the two extra vertices' rows

    // ---- the unit cube's corners.  Every region's vertex list is a subset of the eight corners taken in the order
    // 0,1,2,4,3,5,6,7 (tetra0: 0 1 2 4; octahedron: 1 2 4 3 5 6; tetra1: 3 5 6 7), so all eight are evaluated with COMPILE-TIME
    // offsets -- displacement (d0 - i) - n*SQUISH, hash chain shared per (i) and (i,j) -- and a corner outside the region's
    // list starts its attenuation from -inf instead of 2, i.e. contributes +0.0 exactly like an out-of-radius vertex.
    // Corner 0 is on the list of tetra0 only and corner 7 on that of tetra1 only: they share ONE slot (the lane's region picks
    // the corner; in the octahedron the slot starts from -inf and is a signed zero).  The slot's term is evaluated once and ADDED
    // where its region's vertex order has it: first for tetra0, last (of the cube terms) otherwise.  `value` starts at +0.0 and a
    // sum of terms can never become -0.0 from there, so the zero terms the other positions used to add are exact no-ops.
    const int NINF = (int)0xFFF00000, TWO = 0x40000000;
    const double two1 = __hiloint2double(regB ? NINF : TWO, 0);                  // corners 1, 2, 4
    const double two2 = __hiloint2double(regA ? NINF : TWO, 0);                  // corners 3, 5, 6
    const double two07 = __hiloint2double((regA || regB) ? TWO : NINF, 0);       // the shared slot
    const double X[2] = {dx0, dx0 - 1.0}, Y[2] = {dy0, dy0 - 1.0}, Z[2] = {dz0, dz0 - 1.0};
    const double SQ[4] = {0.0, 1.0 * SQUISH3, 2.0 * SQUISH3, 3.0 * SQUISH3};
    // The hash rows of a lattice-corner PAIR (coordinate c and c + 1 of the same chain level) sit in one 8-byte entry (Tables): 1 + 2 + 4
    // ds_read_b64 for the cube's 2 + 4 + 8 entries; h(m + 1) never wraps (entries -1 .. 514: 255 (hash) + 256 (coordinate + 1) and the
    // extra vertices' offsets).
    unsigned h1[2][2], gz[2][2][2];
    const uint2 h0 = lds_pair(T, xb2);
    {
        const uint2 a = lds_pair(T, h0.x + yb2), b = lds_pair(T, h0.y + yb2);
        h1[0][0] = a.x; h1[0][1] = a.y; h1[1][0] = b.x; h1[1][1] = b.y;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint2 g = lds_pair(T, h1[i][j] + zb2);            // zb2 carries the distance from H to G
            gz[i][j][0] = g.x; gz[i][j][1] = g.y;
        }
    // shared slot: corner 0 (region A) or corner 7 (otherwise)
    // corner 0: d0; corner 7: (d0 - 1) - 3 SQUISH, and 3 SQUISH rounds to exactly 1.0 -- so both are (d0 - w) - w with w = 0.0 or 1.0 (subtracting
    // +0.0 is exact and keeps the sign of a zero): one select of w's high word instead of six on the components
    static_assert(3.0 * SQUISH3 == 1.0, "corner 7's squish term is exactly 1.0");
    const double w07 = __hiloint2double(regA ? 0 : 0x3FF00000, 0);
    const double sx = (dx0 - w07) - w07, sy = (dy0 - w07) - w07, sz = (dz0 - w07) - w07;
    const double slot = kernel_term(T, two07, sx, sy, sz, regA ? gz[0][0][0] : gz[1][1][1]);
    // tetra0: corner 0 is the first term (0.0 + t: a -0.0 term gives +0.0); otherwise the slot is the last cube term.  Both positions as
    // ONE explicit fma each with a 1.0 / 0.0 factor: t * 1 and t * 0 = +-0 are exact, so the fma rounds once exactly like the add it
    // replaces, and adding +-0 to a sum that started at +0.0 changes nothing (round 6: two instead of six instructions)
    const double mA = __hiloint2double(regA ? 0x3FF00000 : 0, 0), mB = __hiloint2double(regA ? 0 : 0x3FF00000, 0);
    double value = fma(slot, mA, 0.0);
    constexpr int ORDER[6] = {1, 2, 4, 3, 5, 6};
#pragma unroll
    for (int s = 0; s < (ABL == 3 ? 3 : 6); ++s) {
        const int code = ORDER[s];
        const int i = code & 1, j = (code >> 1) & 1, k = (code >> 2) & 1, n = i + j + k;
        const double dx = X[i] - SQ[n], dy = Y[j] - SQ[n], dz = Z[k] - SQ[n];
        value += kernel_term(T, n == 1 ? two1 : two2, dx, dy, dz, gz[i][j][k]);
    }
    value = fma(slot, mB, value);                                   // tetra1: corner 7 is the last cube term; octahedron: +-0.0
    // ---- the two extra vertices: displacement recipe and hash offsets from the vertex table (simplex_tables.h)
    if (ABL != 1) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const char *row = reinterpret_cast<const char *>(&T) + (e == 0 ? (pair & 0xFFFFu) : (pair >> 16));
            const double2 a01 = *reinterpret_cast<const double2 *>(row), a23 = *reinterpret_cast<const double2 *>(row + 16);
            const double2 c01 = *reinterpret_cast<const double2 *>(row + 32);
            const double2 czij = *reinterpret_cast<const double2 *>(row + 48);      // {cz, (i8 | j8 << 32)}: one ds_read_b128
            const double cz = czij.x;
            const int2 ij = make_int2(__double2loint(czij.y), __double2hiint(czij.y));
            const int k2 = *reinterpret_cast<const int *>(row + 64);
            const double dx = ((dx0 - a01.x) - a23.y) - c01.x;
            const double dy = ((dy0 - a01.y) - a23.y) - c01.y;
            const double dz = ((dz0 - a23.x) - a23.y) - cz;
            // masked coordinate (0 .. 255) + lattice offset (-1 .. 2) [+ hash value (0 .. 255)]: entries -1 .. 512 of the extended table,
            // whose entry m holds the permutation row of m & 255 -- the sums need no wrap-around mask (round 6: six v_and less)
            const unsigned e0h = lds_one(T, xb2 + (unsigned)ij.x);
            const unsigned e1h = lds_one(T, e0h + yb2 + (unsigned)ij.y);
            const unsigned goff = lds_one(T, e1h + zb2 + (unsigned)k2);
            value += kernel_term(T, 2.0, dx, dy, dz, goff);
        }
    }
    return div_norm3(value);
}

// sum_o persistence^o * noise3(x / f_o, y / f_o, z / f_o), f_o = f0 / 2^o, octave 0 first (simplex.py:89-92, :50-52)
template <int ABL, bool SAFE>
__device__ __forceinline__ double octave_sum(const Tables &T, double xd, double yd, double zd, double f0, bool pow2, int octaves,
                                             double persistence)
{
    double acc = 0.0, amp = 1.0;
    if (pow2) {                                     // x / f == x * (1 / f) bit for bit, and 1 / f doubles exactly per octave
        double rf = 1.0 / f0;
        for (int o = 0; o < octaves; ++o) {
            const double n = noise3<ABL, SAFE>(T, xd * rf, yd * rf, zd * rf);
            acc = acc + amp * n;        // noise += amplitude * field, octave 0 first (simplex.py:90)
            rf = rf * 2.0;
            amp = amp * persistence;
        }
    } else {
        double f = f0;
        for (int o = 0; o < octaves; ++o) {
            const double n = noise3<ABL, SAFE>(T, xd / f, yd / f, zd / f);
            acc = acc + amp * n;
            f = f / 2;
            amp = amp * persistence;
        }
    }
    return acc;
}

// six waves per SIMD (80 registers): the kernel is VALU-issue bound with the LDS close behind (round 6: five waves 4.25 ms, six / seven 4.20,
// eight 4.27); the compiler's own choice drifts
// to 97-104 registers = four waves.  -DSIMPLEX_WAVES=n for measurement builds.
#ifndef SIMPLEX_WAVES
#define SIMPLEX_WAVES 6
#endif
template <typename OutT, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SIMPLEX_WAVES))) void simplex3_octaves_kernel(anoddpm_simplex_args a, int rows_per_block)
{
    __shared__ Tables T;
    const int s = blockIdx.z;
    long long tab = (a.table_sel ? (long long)(*a.table_sel) * a.table_sel_scale : 0) +
                    (long long)s * a.table_slice_stride;
    load_tables(T, a.tables + tab * 512);

    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    if (x >= a.W) return;
    const long long zi = a.zvals ? a.zvals[s] : a.z0 + s;

    // x / f: when f is a power of two (every frequency the reference uses: 64, 2^i, and their halvings) the quotient is an
    // exact scaling and equals x * (1/f) bit for bit -- and 1/f doubles exactly from octave to octave; any other f takes the
    // IEEE division
    const double f0 = a.frequency;
    int fe;
    const bool pow2 = (frexp(f0, &fe) == 0.5) && fe > -900 && fe - a.octaves > -900 && fe < 900;
    const double xd = (double)x, zd = (double)zi;
    // largest |coordinate| of any octave (the last one: f0 / 2^(octaves-1)), with room for the stretch term: below 2^31 the lattice
    // base converts with a plain 32-bit conversion
    const double fmin = ldexp(f0, -(a.octaves > 0 ? a.octaves - 1 : 0));
    // a workgroup walks rows_per_block four-row tiles with one copy of the tables (launch_simplex: only when the grid stays many
    // rounds deep): loading and building them is ~150 VALU instructions per thread against ~2 400 per pixel
    for (int ty = 0; ty < rows_per_block; ++ty) {
        const int y = (blockIdx.y * rows_per_block + ty) * 4 + (threadIdx.x >> 6);
        if (y >= a.H) break;
        const double yd = (double)y;
        const bool safe = fmin > 0.0 && fmax(fmax(fabs(xd), fabs(yd)), fabs(zd)) * 2.0 < 2147483000.0 * fmin;
        double acc = 0.0;
        if (safe) acc = octave_sum<ABL, true>(T, xd, yd, zd, f0, pow2, a.octaves, a.persistence);
        else      acc = octave_sum<ABL, false>(T, xd, yd, zd, f0, pow2, a.octaves, a.persistence);
        OutT *out = reinterpret_cast<OutT *>(a.out) + (long long)s * a.out_slice_stride + (long long)y * a.W + x;
        *out = (OutT)acc;
    }
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
        out[i] = noise3<0>(T, X[ix], Y[iy], Z[iz]);
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
    // four-row tiles per workgroup: 4 / 2 while the grid stays at least eight rounds of the chip's resident workgroups deep
    // (6 per CU x 256 CUs), else 1 -- the per-step fields of a reverse chain (a few slices) keep one tile per workgroup
    const long long tiles = (long long)((a->W + 63) / 64) * ((a->H + 3) / 4) * a->nslices;
    const int rpb = tiles >= 4ll * 8 * 1536 ? 4 : (tiles >= 2ll * 8 * 1536 ? 2 : 1);
    dim3 grid((a->W + 63) / 64, ((a->H + 3) / 4 + rpb - 1) / rpb, a->nslices);
    ANODDPM_REQUIRE(grid.y <= 65535, "simplex3_octaves: H too large");
#ifdef ANODDPM_ABLATE           // timing ablations (wrong results): measurement builds only
    const int abl = anoddpm::g_debug[7];
    if (abl == 1) hipLaunchKernelGGL((simplex3_octaves_kernel<OutT, 1>), grid, dim3(256), 0, anoddpm::as_stream(stream), *a, rpb);
    else if (abl == 2) hipLaunchKernelGGL((simplex3_octaves_kernel<OutT, 2>), grid, dim3(256), 0, anoddpm::as_stream(stream), *a, rpb);
    else if (abl == 3) hipLaunchKernelGGL((simplex3_octaves_kernel<OutT, 3>), grid, dim3(256), 0, anoddpm::as_stream(stream), *a, rpb);
    else
#endif
    hipLaunchKernelGGL((simplex3_octaves_kernel<OutT, 0>), grid, dim3(256), 0, anoddpm::as_stream(stream), *a, rpb);
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
