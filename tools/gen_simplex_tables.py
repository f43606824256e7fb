#!/usr/bin/env python3
"""Generate anoddpm_amd/csrc/simplex_tables.h: the two lookup tables of the OpenSimplex-3D kernel (csrc/simplex.hip).

  kRegionLut[320]  the two "extra" lattice vertices of an evaluation, u16 = id(e0) | id(e1) << 8, indexed DIRECTLY BY THE COMPARISON
                   BITS of _noise3's region decisions (simplex.py:354-798; round 6 -- until then the kernel reduced the comparisons to
                   a decision code with ~45 selects first).  The kernel's index is built from sign bits (layout in main() below); it
                   is a permutation of the comparison-bit table `region` that main() derives first:
                       tetrahedron at (0,0,0):  [0, 128):   x>=y | z>y << 1 | z>x << 2 | z<y << 3 | w>x << 4 | w>y << 5 | w>z << 6      (w = 1 - in_sum)
                       tetrahedron at (1,1,1):  [128, 256): x<=y | z<y << 1 | z<x << 2 | z>y << 3 | w<x << 4 | w<y << 5 | w<z << 6      (w = 3 - in_sum)
                       octahedron:              [256, 384): p1>1 | p2>1 << 1 | p3>1 << 2 | |p1-1|<=|p2-1| << 3 | |p1-1|<|p3-1| << 4 |
                                                            |p1-1|>|p2-1| << 5 | |p2-1|<|p3-1| << 6
                   Every entry is the old decision code's entry (lut[] below: (single << 3) | c, 16 + ..., 128 + (a_point | b_point << 3 |
                   a_further << 6 | b_further << 7)) for the code the reference's if / elif chains reach from those comparison results.
  kVertexRows[128] per vertex id = (i+1) | (j+1) << 2 | (k+1) << 4 | late << 6 the displacement recipe
                       d = ((d0 - A) - n*SQUISH) - C            per axis (n = i + j + k)
                   as doubles {Ax, Ay, Az, n*SQUISH, Cx, Cy, Cz} + the hash offsets {2i, 2j, 2k}.  `late` marks the five vertices the
                   reference displaces in a different operation order ((d0 - 1 - 3*SQUISH) - 1 at simplex.py:503,506; (d0 - 2*SQUISH) - 2
                   at :737-743): (1,2,0) and (0,2,1) with A_y = 1, C_y = 1; (2,0,0) / (0,2,0) / (0,0,2) with A = 0, C = 2 on the long axis.

The decision logic below restates the kernel's own (bit-exact, KAT-tested) region code so that the kernel can replace its divergent
branches and per-slot offset decoding by one table lookup; tests/test_gpu_simplex.py pins the result against the reference's values."""
import os

SQUISH3 = 1.0 / 3


def vid(i, j, k, late=0):
    return (i + 1) | (j + 1) << 2 | (k + 1) << 4 | late << 6


def bits(c):
    return (c & 1, (c >> 1) & 1, (c >> 2) & 1)


def extras_tetra0(single, c):
    if single:                                   # c in {1, 2, 4}
        e0, e1 = [0, 0, 0], [0, 0, 0]
        if c & 1:
            e0[0], e1[0] = 1, 1
        else:
            e0[0], e1[0] = -1, 0
        if c & 2:
            e0[1], e1[1] = 1, 1
        elif c & 1:
            e0[1] = -1
        else:
            e1[1] = -1
        if c & 4:
            e0[2], e1[2] = 1, 1
        else:
            e1[2] = -1
        return vid(*e0), vid(*e1)
    b = bits(c)                                  # c = a_point | b_point
    return vid(*b), vid(*[1 if x else -1 for x in b])


def extras_tetra1(single, c):
    if single:                                   # c in {3, 5, 6}
        e0, e1, l0, l1 = [0, 0, 0], [0, 0, 0], 0, 0
        if c & 1:
            e0[0], e1[0] = 2, 1
        if c & 2:
            e0[1], e1[1] = 1, 1
            if c & 1:
                e1[1], l1 = 2, 1
            else:
                e0[1], l0 = 2, 1
        if c & 4:
            e0[2], e1[2] = 1, 2
        return vid(*e0, late=l0), vid(*e1, late=l1)
    b = bits(c)                                  # c = a_point & b_point
    return vid(*b), vid(*[2 * x for x in b])


def extras_octa(ap, bp, af, bf):
    if af == bf:
        if af:
            c = ap & bp
            e0 = vid(1, 1, 1)
            e1 = vid(2, 0, 0) if c & 1 else (vid(0, 2, 0) if c & 2 else vid(0, 0, 2))
        else:
            c = ap | bp
            e0 = vid(0, 0, 0)
            e1 = vid(-1, 1, 1) if not c & 1 else (vid(1, -1, 1) if not c & 2 else vid(1, 1, -1))
        return e0, e1
    c1, c2 = (ap, bp) if af else (bp, ap)
    e0 = vid(-1, 1, 1) if not c1 & 1 else (vid(1, -1, 1) if not c1 & 2 else vid(1, 1, -1))
    e1 = vid(2, 0, 0, late=1) if c2 & 1 else (vid(0, 2, 0, late=1) if c2 & 2 else vid(0, 0, 2, late=1))
    return e0, e1


def vertex(idx):
    i, j, k, late = (idx & 3) - 1, ((idx >> 2) & 3) - 1, ((idx >> 4) & 3) - 1, idx >> 6
    off = [i, j, k]
    lt = [0, 0, 0]
    if late:
        if (i, j, k) in ((1, 2, 0), (0, 2, 1)):
            lt[1] = 1
        elif (i, j, k) == (2, 0, 0):
            lt[0] = 2
        elif (i, j, k) == (0, 2, 0):
            lt[1] = 2
        elif (i, j, k) == (0, 0, 2):
            lt[2] = 2
    A = [0.0 if lt[a] == 2 else float(off[a] - (1 if lt[a] == 1 else 0)) for a in range(3)]
    C = [2.0 if lt[a] == 2 else (1.0 if lt[a] == 1 else 0.0) for a in range(3)]
    sq = float(i + j + k) * SQUISH3
    return A, sq, C, [2 * i, 2 * j, 2 * k]


def main():
    lut = [0] * 384
    for single in (0, 1):
        for c in range(8):
            e0, e1 = extras_tetra0(single, c)
            lut[(single << 3) | c] = e0 | e1 << 8
            e0, e1 = extras_tetra1(single, c)
            lut[16 + ((single << 3) | c)] = e0 | e1 << 8
    for ap in range(8):
        for bp in range(8):
            for af in (0, 1):
                for bf in (0, 1):
                    e0, e1 = extras_octa(ap, bp, af, bf)
                    lut[128 + (ap | bp << 3 | af << 6 | bf << 7)] = e0 | e1 << 8
    region = [0] * 384
    for m in range(128):
        bit = [(m >> k) & 1 for k in range(7)]
        # tetrahedron at (0,0,0) -- the kernel's former select chain, verbatim
        x_ge_y, z_gt_y, z_gt_x, z_lt_y, wx, wy, wz = bit
        a1 = x_ge_y and z_gt_y
        a2 = (not a1) and ((not x_ge_y) and z_gt_x)
        s = (wz if a2 else wx) or (wz if a1 else wy)
        b_gt_a = z_gt_x if a1 else (z_lt_y if a2 else (not x_ge_y))
        ap, bp = (4 if a2 else 1), (4 if a1 else 2)
        c = (bp if b_gt_a else ap) if s else (ap | bp)
        region[m] = lut[(8 if s else 0) | c]
        # tetrahedron at (1,1,1)
        x_le_y, z_lt_y, z_lt_x, z_gt_y, wx, wy, wz = bit
        b1 = x_le_y and z_lt_y
        b2 = (not b1) and ((not x_le_y) and z_lt_x)
        s = (wz if b2 else wx) or (wz if b1 else wy)
        b_lt_a = z_lt_x if b1 else (z_gt_y if b2 else (not x_le_y))
        ap, bp = (3 if b2 else 6), (3 if b1 else 5)
        c = (bp if b_lt_a else ap) if s else (ap & bp)
        region[128 + m] = lut[16 + ((8 if s else 0) | c)]
        # octahedron
        f1, f2, f3, le_ab, lt_ac, gt_ab, lt_bc = bit
        t1 = le_ab and lt_ac
        t2 = (not t1) and (gt_ab and lt_bc)
        p3c = 6 if f3 else 1
        ap = p3c if t1 else (3 if f1 else 4)
        bp = p3c if t2 else (5 if f2 else 2)
        af = f3 if t1 else f1
        bf = f3 if t2 else f2
        region[256 + m] = lut[128 + (ap | bp << 3 | (64 if af else 0) | (128 if bf else 0))]
    # Round 6, second form: the kernel shifts the SIGN BITS of fp64 differences into the index (first pushed = highest bit):
    #   tetrahedra [0, 256): regB << 7 | (x<y) << 6 | (z>y) << 5 | (z>x) << 4 | (z<y) << 3 | (w>x) << 2 | (w>y) << 1 | (w>z)
    #                        -- for the tetrahedron at (1,1,1) on NEGATED operands, i.e. (y<x), (z<y), (z<x), (z>y), (w<x), (w<y), (w<z)
    #   octahedron [256, 320): (p1>1) << 5 | (p2>1) << 4 | (p3>1) << 3 | (|p1-1|>|p2-1|) << 2 | (|p1-1|<|p3-1|) << 1 | (|p2-1|<|p3-1|)
    # mapped onto the comparison-bit table above (`region`, the kernel's former index).
    lut = [0] * 320
    for n in range(256):
        reg_b = n >> 7
        lt_xy, z_gt_y, z_gt_x, z_lt_y, wx, wy, wz = [(n >> k) & 1 for k in (6, 5, 4, 3, 2, 1, 0)]
        # x >= y is the complement of x < y (far corner, on negated operands: x <= y the complement of y < x)
        m = (1 - lt_xy) | z_gt_y << 1 | z_gt_x << 2 | z_lt_y << 3 | wx << 4 | wy << 5 | wz << 6
        lut[n] = region[128 * reg_b + m]
    for n in range(64):
        f1, f2, f3, gt_ab, lt_ac, lt_bc = [(n >> k) & 1 for k in (5, 4, 3, 2, 1, 0)]
        m = f1 | f2 << 1 | f3 << 2 | (1 - gt_ab) << 3 | lt_ac << 4 | gt_ab << 5 | lt_bc << 6
        lut[256 + n] = region[256 + m]
    import struct

    def u64(v):
        return struct.unpack("<Q", struct.pack("<d", v))[0]

    # Everything below is emitted IN THE KERNEL'S LDS LAYOUT (struct Tables of csrc/simplex.hip), so that a workgroup loads the
    # seed-independent tables with plain 16-byte copies (round 6: building them per workgroup cost ~130 VALU instructions per thread,
    # 5 % of the kernel):
    #   kRegionLutAddr[n] = LDS byte addresses of the two extra vertices' rows, lo / hi half (row id at VTX_LDS_BASE + 80 id)
    #   kVertexRows[id]   = 80-byte rows {ax, ay, az, n*SQUISH, cx, cy, cz} as doubles, (8i | 8j << 32), (8k), 0
    #   kGradRows[g]      = 48-byte rows {gx, gy, gz, 0, 0, 0} as doubles (GRADIENTS3, simplex.py:116-127)
    VTX_LDS_BASE = 12320
    grads = [(-11, 4, 4), (-4, 11, 4), (-4, 4, 11), (11, 4, 4), (4, 11, 4), (4, 4, 11),
             (-11, -4, 4), (-4, -11, 4), (-4, -4, 11), (11, -4, 4), (4, -11, 4), (4, -4, 11),
             (-11, 4, -4), (-4, 11, -4), (-4, 4, -11), (11, 4, -4), (4, 11, -4), (4, 4, -11),
             (-11, -4, -4), (-4, -11, -4), (-4, -4, -11), (11, -4, -4), (4, -11, -4), (4, -4, -11)]
    out = ["// GENERATED by tools/gen_simplex_tables.py -- do not edit.  Lookup tables of csrc/simplex.hip in the kernel's LDS layout (see the",
           "// generator's docstring).", "#pragma once", "#include <stdint.h>", "",
           f"constexpr int REGION_LUT_SIZE = {len(lut)};",
           f"constexpr unsigned VTX_LDS_BASE = {VTX_LDS_BASE};      // offsetof(Tables, vtx), asserted in simplex.hip",
           "__device__ alignas(16) const uint32_t kRegionLutAddr[REGION_LUT_SIZE] = {"]
    for r in range(0, len(lut), 8):
        out.append("    " + ", ".join(f"0x{(VTX_LDS_BASE + 80 * (v & 0x7f)) | ((VTX_LDS_BASE + 80 * (v >> 8)) << 16):08x}" for v in lut[r:r + 8]) + ",")
    out += ["};", "", "__device__ alignas(16) const uint64_t kVertexRows[128 * 10] = {"]
    for idx in range(128):
        A, sq, C, h = vertex(idx)
        w = [u64(v) for v in (A[0], A[1], A[2], sq, C[0], C[1], C[2])]
        w.append(((4 * h[0]) & 0xFFFFFFFF) | (((4 * h[1]) & 0xFFFFFFFF) << 32))
        w.append((4 * h[2]) & 0xFFFFFFFF)
        w.append(0)
        out.append("    " + ", ".join(f"0x{v:016x}ull" for v in w) + ",")
    out += ["};", "", "__device__ alignas(16) const uint64_t kGradRows[24 * 6] = {"]
    for g in grads:
        out.append("    " + ", ".join(f"0x{u64(float(v)):016x}ull" for v in g) + ", 0, 0, 0,")
    out += ["};", ""]
    import sys
    # optional argument: write somewhere else (tests/test_simplex_tables.py compares a fresh copy with the committed header)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                               "anoddpm_amd", "csrc", "simplex_tables.h")
    with open(path, "w") as fh:
        fh.write("\n".join(out))
    print(path, len(lut), "lut entries, 128 vertices")


if __name__ == "__main__":
    main()
