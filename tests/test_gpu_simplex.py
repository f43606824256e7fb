"""-m gpu: the HIP OpenSimplex kernel against the reference-generated fixtures and the C oracle.
Bar: bit-exact (compared as uint64 / uint32 patterns)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(GOLDEN, "simplex_kat.npz"))


def test_point_kats_bit_exact(kat):
    from simplex import Simplex_CLASS
    pts = kat["points"]
    s = Simplex_CLASS()
    for seed in (3, 12345):
        s.newSeed(seed)
        # noise3array evaluates the outer product grid; take its diagonal in chunks
        got = np.empty(len(pts))
        for i0 in range(0, len(pts), 64):
            p = pts[i0:i0 + 64]
            g = s.noise3array(p[:, 0], p[:, 1], p[:, 2])
            idx = np.arange(len(p))
            got[i0:i0 + 64] = g[idx, idx, idx]
        bad = np.nonzero(bits(got) != bits(kat[f"points_val_seed{seed}"]))[0]
        assert bad.size == 0, (seed, bad[:5], pts[bad[:5]], got[bad[:5]], kat[f"points_val_seed{seed}"][bad[:5]])
    s.newSeed(3)
    assert s.noise3(0.1, 0.2, 0.3) == 0.42740996556634286
    assert s.noise3(510, 510, 1998) == -4.701688883822813e-66


def test_fixed_T_and_volume_bit_exact(kat):
    from simplex import Simplex_CLASS
    s = Simplex_CLASS()
    s.newSeed(int(kat["fixedT_seed"]))
    for i, t in enumerate(kat["fixedT_t"]):
        a = s.rand_3d_fixed_T_octaves((64, 64), np.array([t]), 6, 0.8, 64)
        assert a.shape == (1, 64, 64) and a.dtype == np.float64
        assert (bits(a[0]) == bits(kat["fixedT_64x64_o6"][i])).all()
        b = s.rand_3d_fixed_T_octaves((40, 24), np.array([t]), 8, 0.7, 32)
        assert (bits(b[0]) == bits(kat["fixedT_40x24_o8_f32"][i])).all()
    # batched extension: slice b is the field at T[b]
    allt = s.rand_3d_fixed_T_octaves((64, 64), kat["fixedT_t"], 6, 0.8, 64)
    assert (bits(allt) == bits(kat["fixedT_64x64_o6"])).all()
    s.newSeed(int(kat["c4_seed"]))
    v = s.rand_3d_octaves((5, 12, 20), 3, 0.5, 8)
    assert (bits(v) == bits(kat["vol_5x12x20_o3"])).all()


def test_config4_volume_crops_and_oracle_slices(kat):
    """BASELINE config 4: rand_3d_octaves((1000,256,256), 8, 0.8, 64), seed 12345 -- full size on the GPU,
    checked against reference crops and against full oracle slices."""
    from simplex import Simplex_CLASS
    from oracle.simplex_oracle import OracleSimplex
    s = Simplex_CLASS()
    s.newSeed(12345)
    vol = s.rand_3d_octaves((1000, 256, 256), 8, 0.8, 64)
    assert vol.shape == (1000, 256, 256)
    for ci, (y0, x0) in enumerate(kat["c4_crop_origin_yx"]):
        for zi, z in enumerate(kat["c4_z"]):
            assert (bits(vol[z, y0:y0 + 32, x0:x0 + 32]) == bits(kat["c4_crops"][ci][zi])).all(), (ci, z)
    o = OracleSimplex(12345)
    for z in (0, 333, 999):
        ref = o._octaves(np.array([z]), 256, 256, 8, 0.8, 64)[0]
        assert (bits(vol[z]) == bits(ref)).all()
    assert np.isfinite(vol).all() and abs(vol.mean()) < 0.05 and 0.2 < vol.std() < 1.5


@pytest.mark.parametrize("nz,freq", [(1300, 64.0), (2600, 24.0)])
def test_deep_ragged_volumes_walk_several_row_tiles_per_workgroup(nz, freq):
    """Volumes deep enough that a workgroup of the octave kernel walks 2 (nz 1300: 26 000 tiles) / 4 (nz 2600: 52 000) four-row tiles with one copy of
    its tables (csrc/simplex.hip launch_simplex), on a shape whose height is neither a multiple of 4 nor of the tile walk: every
    slice of a spread, incl. the last rows and columns, against the C oracle; frequency 24 takes the IEEE-division octave loop."""
    from simplex import Simplex_CLASS
    from oracle.simplex_oracle import OracleSimplex
    s = Simplex_CLASS()
    s.newSeed(-987654321)
    vol = s.rand_3d_octaves((nz, 38, 70), 4, 0.8, freq)
    assert vol.shape == (nz, 38, 70)
    o = OracleSimplex(-987654321)
    zs = np.array([0, 1, nz // 3, nz // 2 + 1, nz - 2, nz - 1])
    ref = o._octaves(zs, 38, 70, 4, 0.8, freq)
    for i, z in enumerate(zs):
        assert (bits(vol[z]) == bits(ref[i])).all(), z
    assert np.isfinite(vol).all()


def test_random_seeds_vs_oracle_and_f32_fill():
    from simplex import Simplex_CLASS
    from oracle.simplex_oracle import OracleSimplex
    rng = np.random.RandomState(7)
    s = Simplex_CLASS()
    for _ in range(4):
        seed = int(rng.randint(-10 ** 10, 10 ** 10))
        s.newSeed(seed)
        o = OracleSimplex(seed)
        t = rng.randint(0, 1000, size=3)
        oc, pers, fr = int(rng.randint(1, 9)), float(rng.choice([0.5, 0.8, 0.85])), float(rng.choice([2, 16, 64, 128]))
        a = s.rand_3d_fixed_T_octaves((37, 53), t, oc, pers, fr)        # ragged (non-tile-multiple) shape
        b = o.rand_3d_fixed_T_octaves((37, 53), t, oc, pers, fr)
        assert (bits(a) == bits(b)).all()
        # the fused fp32 fill used by generate_simplex_noise: fp64 field rounded once to fp32
        out = torch.full((3, 2, 37, 53), float("nan"), device="cuda:0")
        s.fill_fixed_T_octaves_(out, torch.from_numpy(t).to("cuda:0"), oc, pers, fr, channel=1)
        got = out[:, 1].cpu().numpy()
        assert (got.view(np.uint32) == b.astype(np.float32).view(np.uint32)).all()
        assert torch.isnan(out[:, 0]).all()
    assert s.rand_3d_fixed_T_octaves((0, 5), np.array([1]), 2, 0.5, 8).shape == (1, 0, 5)
    assert (s.rand_3d_octaves((2, 3, 4), 0, 0.5, 8) == 0).all()


# ---------------------------------------------------------------------------- 2-D (simplex.py:211-318, 56-73)
def test_noise2_bit_exact_vs_reference_and_oracle():
    from simplex import Simplex_CLASS
    from oracle.simplex_oracle import OracleSimplex
    k2 = np.load(os.path.join(GOLDEN, "simplex2_kat.npz"))
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    s = Simplex_CLASS()
    for seed in (3, 12345, -9999999999):
        s.newSeed(seed)
        assert (bits(s.noise2array(k2["grid_x"], k2["grid_y"])) == bits(k2[f"s{seed}_grid"])).all()
        assert (bits(s.rand_2d_octaves((32, 32), 4, 0.7, 16)) == bits(k2[f"s{seed}_oct_32_4_07_16"])).all()
        assert (bits(s.rand_2d_octaves((64, 64), 6, 0.8, 64)) == bits(k2[f"s{seed}_oct_64_6_08_64"])).all()
        pts = k2["points"][:64]
        got = np.array([s.noise2(x, y) for x, y in pts])
        assert (bits(got) == bits(k2[f"s{seed}_values"][:64])).all()
    # all point KATs at once through the grid kernel's diagonal, and a full-size field against the oracle
    s.newSeed(12345)
    o = OracleSimplex(12345)
    n = 512
    assert (bits(s.rand_2d_octaves((n, n), 8, 0.8, 64)) == bits(o.rand_2d_octaves((n, n), 8, 0.8, 64))).all()
    P = k2["points"]
    g = s.noise2array(P[:, 0], P[:, 1])
    assert (bits(np.diagonal(g)) == bits(k2["s12345_values"])).all()
    assert s.rand_2d_octaves((0, 0)).shape == (0, 0)
    with pytest.raises(ValueError):
        s.rand_2d_octaves((8, 16))
