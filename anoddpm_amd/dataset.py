"""`MRIDataset` -- the reference's healthy-MRI slice loader (dataset.py:575-643) with the per-sample work on the MI355X.

Kept from the reference: constructor signature, `__len__`, `__getitem__(idx) -> {"image": [1,H,W] float tensor in [-1,1],
"filenames": name}`, the `.npy` cache next to each volume, `random_slice` drawing `random.randint(40, 100)`, and the default
transform RandomAffine(3, translate=(0.02, 0.09)) -> CenterCrop(235) -> Resize(img_size, BILINEAR) -> ToTensor ->
Normalize(0.5, 0.5) -- which upstream builds from torchvision (not installed here): its arithmetic is Pillow's and is
restated in csrc/loader.hip (bit-identical to Pillow 12), the parameter glue follows torchvision's functional.py.

MI355X-first: every volume is uploaded ONCE and stays resident in HBM (a normalised fp32 volume is 40 MB; the whole NFBS
set is ~5 GB of the 288 GB), a slice never touches the host again, and a whole batch costs three launches (`get_batch`).
A user-supplied `transform` callable is honoured the reference's way (it receives the numpy slice on the host).
NIfTI input needs nibabel exactly like upstream; without it only the `.npy` cache path works.
"""
import ctypes
import os
import random

import numpy as np
import torch

from . import _lib
from ._lib import MriSliceArgs, ResizeArgs, check, current_stream, lib

__all__ = ["MRIDataset", "normalise_volume", "resize_coeffs", "center_crop_geometry", "affine_fixed_coeffs", "cycle",
           "init_dataset_loader"]

CROP = 235


def resize_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs (bilinear) as (k [out][kmax] fp64, kmin, kn): host side, once per size pair."""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = fs
    kmax = int(np.ceil(support)) * 2 + 1
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


def center_crop_geometry(h, w, crop):
    """torchvision center_crop: (pad_left, pad_top, crop_top, crop_left)."""
    pad_left = (crop - w) // 2 if crop > w else 0
    pad_top = (crop - h) // 2 if crop > h else 0
    pad_right = (crop - w + 1) // 2 if crop > w else 0
    pad_bottom = (crop - h + 1) // 2 if crop > h else 0
    H, W = h + pad_top + pad_bottom, w + pad_left + pad_right
    return pad_left, pad_top, int(round((H - crop) / 2.0)), int(round((W - crop) / 2.0))


def affine_fixed_coeffs(w, h, angle, translate):
    """torchvision's inverse affine matrix (centre = image centre, scale 1, no shear) as Pillow's six 16.16 coefficients."""
    import math
    rot = math.radians(angle)
    cx, cy = w * 0.5, h * 0.5
    tx, ty = translate
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def _uniform(lo, hi):
    """One draw from torch's CPU generator, the way torchvision's RandomAffine.get_params draws."""
    return float(torch.empty(1).uniform_(float(lo), float(hi)).item())


def normalise_volume(volume, device):
    """dataset.py:585-592 on the device: returns the normalised fp32 volume (device tensor, same shape)."""
    v = torch.as_tensor(np.ascontiguousarray(volume, dtype=np.float64)).to(device)
    _lib.require_cuda(v, "normalise_volume")
    out = torch.empty(v.shape, dtype=torch.float32, device=v.device)
    ws = torch.empty(2 * 256 + 4, dtype=torch.float64, device=v.device)
    check(lib().anoddpm_volume_normalise(v.data_ptr(), v.numel(), out.data_ptr(), ws.data_ptr(), current_stream()), "volume_normalise")
    return out


class MRIDataset(torch.utils.data.Dataset):
    """Healthy MRI dataset (dataset.py:575-643)."""

    def __init__(self, ROOT_DIR, transform=None, img_size=(32, 32), random_slice=False, device=None, augment=True):
        self.transform = transform
        self.img_size = tuple(int(v) for v in img_size)
        self.filenames = os.listdir(ROOT_DIR)
        if ".DS_Store" in self.filenames:
            self.filenames.remove(".DS_Store")
        self.ROOT_DIR = ROOT_DIR
        self.random_slice = random_slice
        self.augment = augment
        # default: the CALLING process's current device (one process per GPU sets it with torch.cuda.set_device(local_rank));
        # a hard-wired cuda:0 would make ranks > 0 upload their volumes to, and launch on, somebody else's GPU
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cuda:0")
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        self._vols = {}
        self._coef = {}

    def __len__(self):
        return len(self.filenames)

    # ---- volumes: one upload each, resident afterwards
    def _volume(self, name):
        v = self._vols.get(name)
        if v is not None:
            return v
        npy = os.path.join(self.ROOT_DIR, name, f"{name}.npy")
        if os.path.exists(npy):
            v = torch.from_numpy(np.ascontiguousarray(np.load(npy), dtype=np.float32)).to(self.device)
        else:
            nii = os.path.join(self.ROOT_DIR, name, f"sub-{name}_ses-NFB3_T1w.nii.gz")
            try:
                import nibabel as nib
            except ImportError as e:
                raise _lib.AnoddpmError(f"{npy} is missing and reading {nii} needs nibabel (as upstream does)") from e
            v = normalise_volume(nib.load(nii).get_fdata(), self.device)
            np.save(npy, v.cpu().numpy())                                    # dataset.py:591-595
        if v.dim() != 3:
            raise ValueError(f"{name}: expected a 3-D volume, got shape {tuple(v.shape)}")
        self._vols[name] = v
        return v

    def _tables(self, in_h, in_w):
        key = (in_h, in_w, self.img_size)
        t = self._coef.get(key)
        if t is None:
            kx, xmin, xn = resize_coeffs(in_w, self.img_size[1])
            ky, ymin, yn = resize_coeffs(in_h, self.img_size[0])
            up = lambda a: torch.from_numpy(a).to(self.device)
            t = self._coef[key] = (up(kx), up(xmin), up(xn), kx.shape[1], up(ky), up(ymin), up(yn), ky.shape[1])
        return t

    def _draw(self, idx):
        """The host-side random draws of one sample, in the reference's order: slice index, then RandomAffine's three."""
        if torch.is_tensor(idx):
            idx = idx.tolist()
        name = self.filenames[idx]
        vol = self._volume(name)
        slice_idx = random.randint(40, 100) if self.random_slice else 80
        aff = None
        if self.augment and self.transform is None:
            X, Z = vol.shape[0], vol.shape[2]
            angle = _uniform(-3.0, 3.0)
            max_dx, max_dy = float(0.02 * Z), float(0.09 * X)
            tx = int(round(_uniform(-max_dx, max_dx)))
            ty = int(round(_uniform(-max_dy, max_dy)))
            aff = affine_fixed_coeffs(Z, X, angle, (tx, ty))
        return name, vol, slice_idx, aff

    def get_batch(self, indices):
        """[B,1,H,W] device tensor + the file names: the default pipeline for a whole batch in three launches."""
        if self.transform is not None:
            items = [self[i] for i in indices]
            return torch.stack([it["image"] for it in items]), [it["filenames"] for it in items]
        with torch.cuda.device(self.device):                                 # launches go to THIS dataset's device and its stream
            return self._get_batch_on_device(indices)

    def _get_batch_on_device(self, indices):
        draws = [self._draw(i) for i in indices]
        B = len(draws)
        X, Z = draws[0][1].shape[0], draws[0][1].shape[2]
        if any(d[1].shape[0] != X or d[1].shape[2] != Z for d in draws):
            raise ValueError("volumes of one batch must share their first and last dimensions")
        dev = self.device
        vols = torch.tensor([d[1].data_ptr() for d in draws], dtype=torch.int64).to(dev)
        ydim = torch.tensor([d[1].shape[1] for d in draws], dtype=torch.int32).to(dev)
        sl = torch.tensor([d[2] for d in draws], dtype=torch.int32).to(dev)
        for d in draws:
            if not 0 <= d[2] < d[1].shape[1]:
                raise IndexError(f"slice {d[2]} is outside volume {d[0]} with {d[1].shape[1]} slices")
        aff = None
        if draws[0][3] is not None:
            aff = torch.tensor([d[3] for d in draws], dtype=torch.int64).to(dev)
        pl, pt, ct, cl = center_crop_geometry(X, Z, CROP)
        crop = torch.empty((B, CROP, CROP), dtype=torch.float32, device=dev)
        a = MriSliceArgs()
        a.vols, a.ydim, a.slice_idx, a.affine, a.out = vols.data_ptr(), ydim.data_ptr(), sl.data_ptr(), (aff.data_ptr() if aff is not None else None), crop.data_ptr()
        a.B, a.X, a.Z, a.crop, a.pad_left, a.crop_top = B, X, Z, CROP, pl - cl, ct - pt
        check(lib().anoddpm_mri_slice_prepare(ctypes.byref(a), current_stream()), "mri_slice_prepare")
        kx, xmin, xn, kmx, ky, ymin, yn, kmy = self._tables(CROP, CROP)
        H, W = self.img_size
        tmp = torch.empty((B, CROP, W), dtype=torch.float32, device=dev)
        out = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
        r = ResizeArgs()
        r.inp, r.tmp, r.out = crop.data_ptr(), tmp.data_ptr(), out.data_ptr()
        r.kx, r.kx_min, r.kx_n, r.ky, r.ky_min, r.ky_n = kx.data_ptr(), xmin.data_ptr(), xn.data_ptr(), ky.data_ptr(), ymin.data_ptr(), yn.data_ptr()
        r.B, r.in_h, r.in_w, r.out_h, r.out_w, r.kmax_x, r.kmax_y = B, CROP, CROP, H, W, kmx, kmy
        r.mean, r.std, r.normalize = 0.5, 0.5, 1
        check(lib().anoddpm_resize_bilinear_pil(ctypes.byref(r), current_stream()), "resize_bilinear_pil")
        return out, [d[0] for d in draws]

    def __getitem__(self, idx):
        if self.transform is not None:                                       # the reference's contract: numpy slice in, anything out
            with torch.cuda.device(self.device):
                name, vol, slice_idx, _ = self._draw(idx)
            image = vol[:, slice_idx:slice_idx + 1, :].reshape(vol.shape[0], vol.shape[2]).cpu().numpy()
            return {"image": self.transform(image), "filenames": name}
        out, names = self.get_batch([idx])
        return {"image": out[0], "filenames": names[0]}


def cycle(iterable):
    """dataset.py:13-16"""
    while True:
        for x in iterable:
            yield x


def init_dataset_loader(mri_dataset, args, shuffle=True):
    """dataset.py:361-370: an endless DataLoader; samples are device tensors already, so no workers / pinning."""
    return cycle(torch.utils.data.DataLoader(mri_dataset, batch_size=args["Batch_Size"], shuffle=shuffle, num_workers=0, drop_last=True))
