"""oracle/loader_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the MRI slice loader path (reference dataset.py:575-643, `MRIDataset`):
  normalise_volume      dataset.py:585-592   clip to [mean - std, mean + 2 std], divide by the range, float32
  take_slice            dataset.py:621       image[:, s:s+1, :].reshape(256, 192)
  default transform     dataset.py:584-593   torchvision RandomAffine(3, translate=(0.02, 0.09)) -> CenterCrop(235) ->
                                             Resize(img_size, BILINEAR) -> ToTensor -> Normalize(0.5, 0.5)
torchvision is NOT installed in the build image (and is a third-party dependency of the reference, not part of it).  For
this pipeline it is a thin layer over Pillow, which IS installed, so the algorithms restated here are Pillow's
(12.x: Geometry.c affine_fixed for AFFINE/NEAREST, Resample.c precompute_coeffs + the two 32-bit passes for BILINEAR) plus
torchvision's parameter glue (transforms/functional.py: center_crop padding rule, _get_inverse_affine_matrix,
RandomAffine.get_params; v0.15-0.20 are identical for these).

Parity status: PINNED for the normalisation / slice (tests/golden/mri_loader.npz is produced by the reference's own
`MRIDataset.__getitem__` with a stubbed nibabel) and for crop / resize / affine (the same fixture holds Pillow's own
outputs); the torchvision parameter glue (which random numbers become which matrix) is restated from its published
source and is UNPINNED -- the fixture generator uses this module for it.
"""
import math

import numpy as np


def synthetic_volume(seed=31, shape=(256, 104, 192)):
    """Brain-shaped synthetic T1-like volume (float64, as nibabel's get_fdata() returns): reproducible anywhere."""
    rng = np.random.RandomState(seed)
    X, Y, Z = shape
    xx, yy, zz = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
    blob = np.exp(-(((xx - X / 2) / 80.0) ** 2 + ((yy - Y * 0.6) / 45.0) ** 2 + ((zz - Z / 2) / 60.0) ** 2) * 2.0)
    vol = blob * 900.0 + rng.gamma(2.0, 20.0, size=shape) * (blob > 0.05) + rng.rand(*shape) * 5.0
    return vol.astype(np.float32).astype(np.float64)


def normalise_volume(image):
    image = np.asarray(image)
    mean, std = np.mean(image), np.std(image)
    lo, hi = mean - 1 * std, mean + 2 * std
    return (np.clip(image, lo, hi) / (hi - lo)).astype(np.float32)


def take_slice(volume, slice_idx):
    return volume[:, slice_idx:slice_idx + 1, :].reshape(volume.shape[0], volume.shape[2]).astype(np.float32)


def center_crop_geometry(h, w, crop):
    """torchvision.transforms.functional.center_crop: zero padding when the image is smaller, then the centred window.
    Returns (pad_left, pad_top, crop_top, crop_left) such that out[y][x] = img[y + crop_top - pad_top][x + crop_left - pad_left]."""
    pad_left = (crop - w) // 2 if crop > w else 0
    pad_top = (crop - h) // 2 if crop > h else 0
    pad_right = (crop - w + 1) // 2 if crop > w else 0
    pad_bottom = (crop - h + 1) // 2 if crop > h else 0
    H, W = h + pad_top + pad_bottom, w + pad_left + pad_right
    crop_top = int(round((H - crop) / 2.0))
    crop_left = int(round((W - crop) / 2.0))
    return pad_left, pad_top, crop_top, crop_left


def center_crop(img, crop):
    h, w = img.shape
    pl, pt, ct, cl = center_crop_geometry(h, w, crop)
    out = np.zeros((crop, crop), dtype=img.dtype)
    for y in range(crop):
        sy = y + ct - pt
        if 0 <= sy < h:
            x0, x1 = max(0, pl - cl), min(crop, w + pl - cl)
            out[y, x0:x1] = img[sy, x0 + cl - pl:x1 + cl - pl]
    return out


def resize_coeffs(in_size, out_size):
    """Pillow Resample.c precompute_coeffs for the bilinear (triangle) filter over the whole input."""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 1.0 * fs
    kmax = int(math.ceil(support)) * 2 + 1
    k = np.zeros((out_size, kmax), dtype=np.float64)
    kmin = np.zeros(out_size, dtype=np.int32)
    kn = np.zeros(out_size, dtype=np.int32)
    ss = 1.0 / fs
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            w = 1.0 - abs(v) if abs(v) < 1.0 else 0.0
            k[xx, x] = w
            ww += w
        if ww != 0.0:
            k[xx, :xmax] /= ww
        kmin[xx], kn[xx] = xmin, xmax
    return k, kmin, kn


def resize_bilinear(img, out_h, out_w):
    """Pillow Image.resize((out_w, out_h), BILINEAR) on a float32 image: horizontal pass, then vertical pass."""
    h, w = img.shape
    kx, xmin, xn = resize_coeffs(w, out_w)
    ky, ymin, yn = resize_coeffs(h, out_h)
    tmp = np.zeros((h, out_w), dtype=np.float32)
    for xo in range(out_w):
        acc = np.zeros(h, dtype=np.float64)
        for t in range(xn[xo]):
            acc = acc + img[:, xmin[xo] + t].astype(np.float64) * kx[xo, t]
        tmp[:, xo] = acc.astype(np.float32)
    out = np.zeros((out_h, out_w), dtype=np.float32)
    for yo in range(out_h):
        acc = np.zeros(out_w, dtype=np.float64)
        for t in range(yn[yo]):
            acc = acc + tmp[ymin[yo] + t, :].astype(np.float64) * ky[yo, t]
        out[yo, :] = acc.astype(np.float32)
    return out


def inverse_affine_matrix(center, angle, translate):
    """torchvision _get_inverse_affine_matrix with scale 1, shear 0 (what RandomAffine(3, translate=...) produces)."""
    rot = math.radians(angle)
    cx, cy = center
    tx, ty = translate
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def affine_fixed_coeffs(m):
    """Pillow affine_fixed: the six 16.16 fixed-point coefficients (a0, a1, a2, a3, a4, a5) with the half-pixel folded in."""
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def affine_nearest(img, m):
    """Pillow Image.transform(size, AFFINE, m, NEAREST, fill 0) on a 2-D array."""
    h, w = img.shape
    a0, a1, a2, a3, a4, a5 = affine_fixed_coeffs(m)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
    xi = (a2 + a0 * xs + a1 * ys) >> 16
    yi = (a5 + a3 * xs + a4 * ys) >> 16
    ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
    out = np.zeros_like(img)
    out[ok] = img[yi[ok], xi[ok]]
    return out


def random_affine_params(w, h, degrees=3.0, translate=(0.02, 0.09)):
    """torchvision RandomAffine.get_params for degrees=(-d, d), translate only: three draws from torch's CPU generator."""
    import torch
    angle = float(torch.empty(1).uniform_(-float(degrees), float(degrees)).item())
    max_dx, max_dy = float(translate[0] * w), float(translate[1] * h)
    tx = int(round(torch.empty(1).uniform_(-max_dx, max_dx).item()))
    ty = int(round(torch.empty(1).uniform_(-max_dy, max_dy).item()))
    return angle, (tx, ty)


def default_transform(slice2d, img_size, affine=None, crop=235):
    """The reference's default transform with the random affine parameters given (None = identity)."""
    img = slice2d
    if affine is not None:
        h, w = img.shape
        img = affine_nearest(img, inverse_affine_matrix((w * 0.5, h * 0.5), affine[0], affine[1]))
    img = center_crop(img, crop)
    img = resize_bilinear(img, img_size[0], img_size[1])
    return ((img - np.float32(0.5)) / np.float32(0.5))[None]
