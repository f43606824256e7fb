"""-m gpu: the device MRI slice loader (anoddpm_amd/dataset.py, csrc/loader.hip) against tests/golden/mri_loader.npz: the
reference's own normalised .npy / slices and Pillow's crop / resize / affine outputs -- bit for bit."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

from conftest import GOLDEN


@pytest.fixture(scope="module")
def root(tmp_path_factory):
    from oracle import loader_oracle as lo
    d = tmp_path_factory.mktemp("mri")
    vol = lo.synthetic_volume()
    for name in ("vol0", "vol1"):
        os.makedirs(d / name)
    return str(d), vol


def test_volume_normalisation_matches_reference_cache(root):
    from anoddpm_amd.dataset import normalise_volume
    g = np.load(os.path.join(GOLDEN, "mri_loader.npz"))
    npy = normalise_volume(root[1], DEV).cpu().numpy()
    ref = g["npy_probe"]
    got = npy.flatten()[::97]
    # mean / std are folded in a different order than numpy's pairwise sums: 1 ulp of float32 at most
    assert np.abs(got - ref).max() <= 1.2e-7 and (got != ref).mean() < 0.01
    np.save(os.path.join(root[0], "vol0", "vol0.npy"), npy)
    np.save(os.path.join(root[0], "vol1", "vol1.npy"), npy[:, ::-1, :].copy())


def test_dataset_default_pipeline_bit_exact(root):
    from anoddpm_amd.dataset import MRIDataset
    g = np.load(os.path.join(GOLDEN, "mri_loader.npz"))
    from oracle import loader_oracle as lo
    np.save(os.path.join(root[0], "vol0", "vol0.npy"), lo.normalise_volume(root[1]))          # the reference's exact cache
    np.save(os.path.join(root[0], "vol1", "vol1.npy"), lo.normalise_volume(root[1]))
    for size in ((64, 64), (256, 256), (32, 48)):
        ds = MRIDataset(root[0], img_size=size, random_slice=False, device=DEV, augment=False)
        i0 = ds.filenames.index("vol0")
        s = ds[i0]
        assert s["filenames"] == "vol0" and s["image"].is_cuda and tuple(s["image"].shape) == (1,) + size
        ref = g[f"final_{size[0]}x{size[1]}"]
        assert np.array_equal(s["image"].cpu().numpy().view(np.uint32), ref.view(np.uint32)), size
    # a user transform receives the numpy slice, like upstream
    ds = MRIDataset(root[0], transform=lambda a: a, img_size=(64, 64), device=DEV)
    assert np.array_equal(ds[ds.filenames.index("vol0")]["image"], g["slice80"])
    random.seed(5)
    ds = MRIDataset(root[0], transform=lambda a: a, img_size=(64, 64), random_slice=True, device=DEV)
    for i in range(3):
        assert np.array_equal(ds[ds.filenames.index("vol0")]["image"], g["random_slices"][i])


def test_random_affine_matches_pillow_and_batches(root, monkeypatch):
    from anoddpm_amd import dataset as D
    g = np.load(os.path.join(GOLDEN, "mri_loader.npz"))
    ds = D.MRIDataset(root[0], img_size=(64, 64), device=DEV, augment=True)
    i0 = ds.filenames.index("vol0")
    # inject the three RandomAffine draws (angle, tx, ty)
    for i in range(3):
        ang, (tx, ty) = float(g["affine_angle"][i]), (int(v) for v in g["affine_translate"][i])
        draws = iter([ang, float(tx), float(ty)])
        monkeypatch.setattr(D, "_uniform", lambda lo, hi: next(draws))
        img = ds[i0]["image"]
        monkeypatch.undo()
        assert np.array_equal(img.cpu().numpy().view(np.uint32), g[f"affine{i}_final_64x64"].view(np.uint32)), i
    # batches: one call == the per-item calls under the same RNG state, and the endless loader yields device batches
    torch.manual_seed(3)
    a = torch.stack([ds[0]["image"], ds[1]["image"], ds[0]["image"]])
    torch.manual_seed(3)
    b, names = ds.get_batch([0, 1, 0])
    assert torch.equal(a, b) and names == [ds.filenames[0], ds.filenames[1], ds.filenames[0]]
    assert b.min() >= -1.0 - 1e-6 and b.max() <= 1.0 + 1e-6
    it = D.init_dataset_loader(ds, {"Batch_Size": 2})
    x = next(it)["image"]
    assert x.shape == (2, 1, 64, 64) and x.is_cuda
