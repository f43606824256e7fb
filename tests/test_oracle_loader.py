"""-m "not gpu": the numpy loader oracle (oracle/loader_oracle.py) against tests/golden/mri_loader.npz -- the reference's
MRIDataset normalisation / slicing and Pillow's own crop / resize / affine outputs."""
import os

import numpy as np

from conftest import GOLDEN


def test_loader_oracle_matches_reference_and_pillow():
    from oracle import loader_oracle as lo
    g = np.load(os.path.join(GOLDEN, "mri_loader.npz"))
    vol = lo.synthetic_volume()
    assert np.array_equal(vol.flatten()[::9973], g["volume_probe"])
    npy = lo.normalise_volume(vol)
    assert tuple(npy.shape) == tuple(g["npy_shape"]) and npy.dtype == np.float32
    assert np.array_equal(npy.flatten()[::97], g["npy_probe"])                       # the reference's .npy cache, bit for bit
    s80 = lo.take_slice(npy, 80)
    assert np.array_equal(s80, g["slice80"])
    for i, si in enumerate(g["random_slices_idx"]):
        assert np.array_equal(lo.take_slice(npy, int(si)), g["random_slices"][i])
    crop = lo.center_crop(s80, 235)
    assert np.array_equal(crop, g["crop235"])
    for (oh, ow) in ((64, 64), (256, 256), (32, 48)):
        r = lo.resize_bilinear(crop, oh, ow)
        assert np.array_equal(r.view(np.uint32), g[f"resized_{oh}x{ow}"].view(np.uint32))
        assert np.array_equal(lo.default_transform(s80, (oh, ow)), g[f"final_{oh}x{ow}"])
    for i in range(3):
        ang, tr = float(g["affine_angle"][i]), tuple(int(v) for v in g["affine_translate"][i])
        m = lo.inverse_affine_matrix((192 * 0.5, 256 * 0.5), ang, tr)
        assert np.array_equal(lo.affine_nearest(s80, m), g[f"affine{i}"])
        assert np.array_equal(lo.default_transform(s80, (64, 64), affine=(ang, tr)), g[f"affine{i}_final_64x64"])


def test_center_crop_geometry_matches_torchvision_rule():
    from oracle import loader_oracle as lo
    assert lo.center_crop_geometry(256, 192, 235) == (21, 0, 10, 0)                  # pad 21 | 22 columns, crop rows 10..244
    assert lo.center_crop_geometry(300, 300, 235) == (0, 0, 32, 32)
