"""CPU check of the GENERATED lookup tables of csrc/simplex.hip (anoddpm_amd/csrc/simplex_tables.h, tools/gen_simplex_tables.py).

The OpenSimplex kernel replaces the if / elif chains of simplex.py:354-798 by a region index built from the sign bits of fp64
differences, one lookup in kRegionLutAddr, and per-vertex displacement recipes (kVertexRows, kGradRows).  This file restates THAT
table-driven algorithm in numpy -- same index bits, same mirrored chain for the two tetrahedra, same operation order, tables parsed
from the generated header -- and pins it bit for bit to the C oracle (itself pinned to the reference's values by
tests/test_oracle_simplex.py) on random points, lattice points and ties.  A wrong permutation of the region table, a wrong vertex
row or a stale header fails here without a GPU; the kernel itself is checked against the same oracle by tests/test_gpu_simplex.py.
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "anoddpm_amd", "csrc", "simplex_tables.h")


def _array(text, name):
    m = re.search(r"\b%s\[[^\]]*\]\s*=\s*\{(.*?)\};" % name, text, re.S)
    assert m, name
    return [int(t.rstrip("ul"), 0) for t in re.findall(r"0x[0-9a-fA-F]+(?:ull)?|\b\d+\b", m.group(1))]


@pytest.fixture(scope="module")
def tables():
    text = open(HEADER).read()
    base = int(re.search(r"VTX_LDS_BASE\s*=\s*(\d+)", text).group(1))
    lut = np.array(_array(text, "kRegionLutAddr"), dtype=np.uint32)
    vtx = np.array(_array(text, "kVertexRows"), dtype=np.uint64).reshape(128, 10)
    grad = np.array(_array(text, "kGradRows"), dtype=np.uint64).reshape(24, 6).view(np.float64)[:, :3]
    return base, lut, vtx, grad


def test_header_is_what_the_generator_writes(tmp_path):
    """The committed header equals a fresh run of tools/gen_simplex_tables.py (the kernel must never run on a hand-edited table)."""
    fresh = str(tmp_path / "simplex_tables.h")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_simplex_tables.py"), fresh], check=True, capture_output=True)
    assert open(fresh).read() == open(HEADER).read()


def test_layout_constants(tables):
    base, lut, vtx, grad = tables
    assert lut.size == 320 and base % 16 == 0
    rows0, rows1 = (lut & 0xFFFF).astype(np.int64) - base, (lut >> 16).astype(np.int64) - base
    assert (rows0 % 80 == 0).all() and (rows1 % 80 == 0).all() and rows0.max() < 128 * 80 and rows1.max() < 128 * 80
    # GRADIENTS3 (simplex.py:116-127): permutations of (+-11, +-4, +-4)
    assert sorted(set(np.abs(grad).sum(axis=1))) == [19.0] and len({tuple(g) for g in grad}) == 24


def _noise3_tables(tab, perm, pgi3, x, y, z):
    """csrc/simplex.hip noise3(), vectorised: numpy's float64 arithmetic is IEEE without contraction, like the kernel's."""
    base, lut, vtx, grad = tab
    S, Q = -1.0 / 6, 1.0 / 3
    stretch = (x + y + z) * S
    xs, ys, zs = x + stretch, y + stretch, z + stretch
    fx, fy, fz = np.floor(xs), np.floor(ys), np.floor(zs)
    squish = ((fx + fy) + fz) * Q
    xins, yins, zins = xs - fx, ys - fy, zs - fz
    in_sum = xins + yins + zins
    dx0, dy0, dz0 = x - (fx + squish), y - (fy + squish), z - (fz + squish)
    regA, regB = in_sum <= 1, in_sum >= 2
    # region index: sign bits, first pushed = highest bit; both tetrahedra on one chain of MIRRORED operands
    sgn = np.where(regB, -1.0, 1.0)
    xm, ym, zm = xins * sgn, yins * sgn, zins * sgn
    wm = np.where(regB, -3.0, 1.0) - in_sum * sgn
    iT = regB.astype(np.int64)
    for d in (xm - ym, ym - zm, xm - zm, zm - ym, xm - wm, ym - wm, zm - wm):
        iT = (iT << 1) | np.signbit(d)
    r1, r2, r3 = 1 - (xins + yins), 1 - (xins + zins), 1 - (yins + zins)
    iO = np.full(x.shape, 4, dtype=np.int64)
    for d in (r1, r2, r3, np.abs(r2) - np.abs(r1), np.abs(r1) - np.abs(r3), np.abs(r2) - np.abs(r3)):
        iO = (iO << 1) | np.signbit(d)
    idx = np.where(regA | regB, iT, iO)
    pair = lut[idx]
    xsb, ysb, zsb = fx.astype(np.int64) & 255, fy.astype(np.int64) & 255, fz.astype(np.int64) & 255

    def gidx(i, j, k):                                   # the hash chain of _extrapolate3 (simplex.py:202-206) on masked coordinates
        h = perm[(xsb + i) & 255]
        h = perm[(h + ysb + j) & 255]
        return pgi3[(h + zsb + k) & 255] // 3

    def term(two, dx, dy, dz, g):
        gv = grad[g]
        with np.errstate(invalid="ignore"):
            attn = np.maximum(two - dx * dx - dy * dy - dz * dz, 0.0)
        attn = attn * attn
        return attn * attn * (gv[:, 0] * dx + gv[:, 1] * dy + gv[:, 2] * dz)

    ninf = -np.inf
    two1, two2 = np.where(regB, ninf, 2.0), np.where(regA, ninf, 2.0)
    two07 = np.where(regA | regB, 2.0, ninf)
    w07 = np.where(regA, 0.0, 1.0)
    slot = term(two07, (dx0 - w07) - w07, (dy0 - w07) - w07, (dz0 - w07) - w07, np.where(regA, gidx(0, 0, 0), gidx(1, 1, 1)))
    value = slot * np.where(regA, 1.0, 0.0) + 0.0
    SQ = [0.0, 1.0 * Q, 2.0 * Q, 3.0 * Q]
    for code in (1, 2, 4, 3, 5, 6):
        i, j, k = code & 1, (code >> 1) & 1, (code >> 2) & 1
        n = i + j + k
        value = value + term(two1 if n == 1 else two2, (dx0 - i) - SQ[n] if i else dx0 - SQ[n], (dy0 - j) - SQ[n] if j else dy0 - SQ[n],
                             (dz0 - k) - SQ[n] if k else dz0 - SQ[n], gidx(i, j, k))
    value = slot * np.where(regA, 0.0, 1.0) + value
    for half in (pair & 0xFFFF, pair >> 16):
        row = vtx[(half.astype(np.int64) - base) // 80]
        f = row[:, :7].copy().view(np.float64)
        ij = row[:, 7]
        i8 = (ij & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32).astype(np.int64)
        j8 = (ij >> np.uint64(32)).astype(np.uint32).view(np.int32).astype(np.int64)
        k8 = (row[:, 8] & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32).astype(np.int64)
        assert (i8 % 8 == 0).all() and (j8 % 8 == 0).all() and (k8 % 8 == 0).all()
        dx = ((dx0 - f[:, 0]) - f[:, 3]) - f[:, 4]
        dy = ((dy0 - f[:, 1]) - f[:, 3]) - f[:, 5]
        dz = ((dz0 - f[:, 2]) - f[:, 3]) - f[:, 6]
        h = perm[(xsb + i8 // 8) & 255]
        h = perm[(h + ysb + j8 // 8) & 255]
        value = value + term(2.0, dx, dy, dz, pgi3[(h + zsb + k8 // 8) & 255] // 3)
    return value / 103.0


@pytest.mark.parametrize("seed", [12345, -987654321])
def test_table_driven_noise3_equals_the_oracle_bit_for_bit(tables, seed):
    from oracle.simplex_oracle import OracleSimplex, init
    perm, pgi3 = init(seed)
    o = OracleSimplex(seed)
    rng = np.random.RandomState(abs(seed) % 1000)
    pts = [rng.uniform(-40, 40, size=(6000, 3)),                              # all three honeycomb regions
           rng.randint(-20, 20, size=(1500, 3)) / 2.0,                        # f = 0.5 coordinates: ties between the inside coordinates
           rng.randint(-30, 30, size=(1500, 3)) / 3.0,
           rng.randint(-12, 12, size=(800, 3)).astype(np.float64),            # lattice-aligned inputs
           rng.randint(0, 256, size=(1200, 3)) / 64.0]                        # config 4's first octave (x / 64)
    p = np.concatenate(pts)
    got = _noise3_tables(tables, perm, pgi3, p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy())
    ref = np.array([o.noise3(*q) for q in p])
    bad = np.nonzero(got.view(np.uint64) != ref.view(np.uint64))[0]
    # +0.0 / -0.0: the kernel's sum starts at +0.0 and can never become -0.0; the oracle agrees (compared as bit patterns)
    assert bad.size == 0, (bad[:5], p[bad[:5]], got[bad[:5]], ref[bad[:5]])
