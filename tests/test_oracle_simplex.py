"""Pins oracle/simplex_oracle.c bit-for-bit against fixtures generated from the reference
(simplex.py:166-192, 202-208, 321-840, 37-93).  CPU only."""
import os

import numpy as np
import pytest

from oracle.simplex_oracle import OracleSimplex, init

from conftest import GOLDEN


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(GOLDEN, "simplex_kat.npz"))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_init_tables(kat):
    for s, p, g in zip(kat["init_seeds"], kat["init_perm"], kat["init_pgi3"]):
        perm, pgi3 = init(int(s))
        assert (perm == p).all() and (pgi3 == g).all()
        assert sorted(perm.tolist()) == list(range(256))


def test_known_answers():
    # SURVEY.md 8c probe values (seed 3 / 12345)
    s = OracleSimplex(3)
    assert s._perm[:6].tolist() == [164, 187, 231, 144, 104, 73]
    assert s.noise3(0.1, 0.2, 0.3) == 0.42740996556634286
    assert s.noise3(0, 0, 0) == 5.2240987598031274e-67
    assert s.noise3(1.5, 2.25, 0.15625) == -0.0996353869486468
    assert s.noise3(510, 510, 1998) == -4.701688883822813e-66
    s.newSeed(12345)
    assert s._perm[:8].tolist() == [33, 182, 149, 37, 26, 75, 1, 19]
    assert s._perm_grad_index3[:8].tolist() == [27, 42, 15, 39, 6, 9, 3, 57]
    assert s.noise3(0.1, 0.2, 0.3) == -0.06351836649838188


def test_points_bit_exact(kat):
    assert (kat["points_region_hist"] > 1000).all()      # all three honeycomb regions covered
    pts = kat["points"]
    for seed in (3, 12345):
        s = OracleSimplex(seed)
        got = np.array([s.noise3(*p) for p in pts])
        assert (bits(got) == bits(kat[f"points_val_seed{seed}"])).all()


def test_fixed_T_octaves(kat):
    s = OracleSimplex(int(kat["fixedT_seed"]))
    for i, t in enumerate(kat["fixedT_t"]):
        a = s.rand_3d_fixed_T_octaves((64, 64), np.array([t]), 6, 0.8, 64)
        assert a.shape == (1, 64, 64)
        assert (bits(a[0]) == bits(kat["fixedT_64x64_o6"][i])).all()
        b = s.rand_3d_fixed_T_octaves((40, 24), np.array([t]), 8, 0.7, 32)
        assert (bits(b[0]) == bits(kat["fixedT_40x24_o8_f32"][i])).all()


def test_volume_and_c4_crops(kat):
    s = OracleSimplex(int(kat["c4_seed"]))
    v = s.rand_3d_octaves((5, 12, 20), 3, 0.5, 8)
    assert (bits(v) == bits(kat["vol_5x12x20_o3"])).all()
    # crops of the config-4 volume: evaluate the octave sum on the crop's coordinates
    zs = kat["c4_z"]
    for ci, (y0, x0) in enumerate(kat["c4_crop_origin_yx"]):
        acc = np.zeros((len(zs), 32, 32))
        f, a = 64, 1
        for _ in range(8):
            acc += a * s.noise3array(np.arange(x0, x0 + 32) / f, np.arange(y0, y0 + 32) / f, zs / f)
            f /= 2
            a *= 0.8
        assert (bits(acc) == bits(kat["c4_crops"][ci])).all()


def test_empty_and_ragged():
    s = OracleSimplex(3)
    assert s.rand_3d_fixed_T_octaves((0, 7), np.array([3]), 2, 0.5, 8).shape == (1, 0, 7)
    assert s.rand_3d_octaves((2, 3, 5), 0, 0.5, 8).tolist() == np.zeros((2, 3, 5)).tolist()
    r = s.rand_3d_fixed_T_octaves((3, 5), np.array([1, 7, 7]), 2, 0.5, 8)
    assert r.shape == (3, 3, 5) and (r[1] == r[2]).all()


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree absent")
def test_live_reference_crosscheck():
    import sys
    sys.path.insert(0, GOLDEN)
    import _refimport
    rs, _, _ = _refimport.load()
    R = rs.Simplex_CLASS()
    R.newSeed(424242)
    O = OracleSimplex(424242)
    assert (R._perm == O._perm).all()
    rng = np.random.RandomState(1)
    for p in rng.uniform(-9, 9, (500, 3)):
        assert bits(R.noise3(*p)) == bits(O.noise3(*p))


# ---------------------------------------------------------------------------- 2-D (simplex.py:211-318, 56-73)
@pytest.fixture(scope="module")
def kat2():
    return np.load(os.path.join(GOLDEN, "simplex2_kat.npz"))


@pytest.mark.parametrize("seed", [3, 12345, -9999999999])
def test_noise2_bit_exact(kat2, seed):
    s = OracleSimplex(seed)
    got = np.array([s.noise2(x, y) for x, y in kat2["points"]])
    assert (bits(got) == bits(kat2[f"s{seed}_values"])).all()
    assert (bits(s.noise2array(kat2["grid_x"], kat2["grid_y"])) == bits(kat2[f"s{seed}_grid"])).all()
    assert (bits(s.rand_2d_octaves((32, 32), 4, 0.7, 16)) == bits(kat2[f"s{seed}_oct_32_4_07_16"])).all()
    assert (bits(s.rand_2d_octaves((64, 64), 6, 0.8, 64)) == bits(kat2[f"s{seed}_oct_64_6_08_64"])).all()


def test_noise2_square_only():
    s = OracleSimplex(3)
    with pytest.raises(ValueError):
        s.rand_2d_octaves((4, 6))
    with pytest.raises(ValueError):
        s.noise2array(np.arange(3.0), np.arange(4.0))
