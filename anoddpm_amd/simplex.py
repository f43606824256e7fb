"""`Simplex_CLASS` -- the reference's OpenSimplex noise generator (simplex.py:14-93) on MI355X.

Same constructor / method surface and RNG consumption (`newSeed()` draws from the global numpy
stream exactly like simplex.py:19-22); the permutation tables are built by the library's host
routine and every field is evaluated by the HIP kernel in anoddpm_amd/csrc/simplex.hip, whose
fp64 results are bit-identical to the reference's numba/CPython arithmetic.

The numpy-returning methods mirror the reference API (fresh float64 arrays).  The hot path uses
`fill_fixed_T_octaves_()` which writes fp32 noise straight into a device tensor with no host
round trip (the reference does D2H of `t`, CPU evaluation and an H2D copy every step,
GaussianDiffusion.py:131-136).

Batched semantics (an extension: the reference raises for len(T) > 1, simplex.py:86-90):
slice b of the result is the field at z = T[b], all slices share the instance's permutation.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import SimplexArgs, check, current_stream, lib, ptr

__all__ = ["Simplex_CLASS", "perm_tables"]


def _wrap64(seed):
    """Python int -> the int64 the LCG sees after its first wrap (simplex.py:166-171,181)."""
    return ((int(seed) + (1 << 63)) % (1 << 64)) - (1 << 63)


def perm_tables(seed):
    """simplex.py:174-192 -> int16[512] = perm[256] ++ perm_grad_index3[256] (host)."""
    tab = np.zeros(512, dtype=np.int16)
    p = tab.ctypes.data_as(ctypes.POINTER(ctypes.c_int16))
    check(lib().anoddpm_simplex_perm_init(_wrap64(seed), p, ctypes.cast(ctypes.addressof(p.contents) + 512, ctypes.POINTER(ctypes.c_int16))),
          "simplex_perm_init")
    return tab


def _device():
    if not torch.cuda.is_available():
        raise _lib.AnoddpmError("Simplex_CLASS: no HIP device visible; anoddpm_amd has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class Simplex_CLASS:

    def __init__(self):
        self._dev_tables = None
        self.newSeed()

    def newSeed(self, seed=None):
        if not seed:
            seed = np.random.randint(-10000000000, 10000000000)
        self._tables = perm_tables(seed)
        self._perm = self._tables[:256].astype(np.int64)
        self._perm_grad_index3 = self._tables[256:].astype(np.int64)
        self._dev_tables = None

    # ------------------------------------------------------------------ device plumbing
    def device_tables(self, device=None):
        device = device or _device()
        if self._dev_tables is None or self._dev_tables.device != device:
            self._dev_tables = torch.from_numpy(self._tables).to(device)
        return self._dev_tables

    def fill_fixed_T_octaves_(self, out, t, octaves=1, persistence=0.5, frequency=32, channel=0,
                              tables=None, table_sel=None, table_sel_scale=1):
        """out[:, channel] <- rand_3d_fixed_T_octaves(out.shape[-2:], t, ...) as fp32, on device.
        out: [B, C, H, W] float32 cuda tensor (contiguous); t: int64 cuda tensor [B]."""
        _lib.require_cuda(out, "Simplex_CLASS.fill_fixed_T_octaves_")
        assert out.dtype == torch.float32 and out.is_contiguous() and out.dim() == 4
        B, C, H, W = out.shape
        if t.dtype != torch.int64 or t.device != out.device:
            t = t.to(device=out.device, dtype=torch.int64)
        t = t.contiguous()
        assert t.numel() == B
        tab = tables if tables is not None else self.device_tables(out.device)
        a = SimplexArgs()
        a.out = out.data_ptr() + 4 * channel * H * W
        a.zvals = t.data_ptr()
        a.tables = tab.data_ptr()
        a.table_sel = table_sel.data_ptr() if table_sel is not None else None
        a.z0 = 0
        a.out_slice_stride = C * H * W
        a.nslices, a.H, a.W = B, H, W
        a.table_slice_stride = 0
        a.table_sel_scale = table_sel_scale
        a.octaves = int(octaves)
        a.persistence = float(persistence)
        a.frequency = float(frequency)
        check(lib().anoddpm_simplex3_octaves_f32(ctypes.byref(a), current_stream()), "simplex3_octaves_f32")
        return out

    def _octaves_f64(self, zvals, z0, nslices, H, W, octaves, persistence, frequency):
        dev = _device()
        out = torch.empty((nslices, H, W), dtype=torch.float64, device=dev)
        if out.numel() == 0 or octaves <= 0:
            return out.zero_().cpu().numpy()
        a = SimplexArgs()
        a.out = out.data_ptr()
        zt = None
        if zvals is not None:
            zt = torch.from_numpy(np.ascontiguousarray(zvals, dtype=np.int64)).to(dev)
            a.zvals = zt.data_ptr()
        a.tables = self.device_tables(dev).data_ptr()
        a.table_sel = None
        a.z0 = int(z0)
        a.out_slice_stride = H * W
        a.H, a.W = H, W
        a.table_slice_stride = 0
        a.table_sel_scale = 1
        a.octaves = int(octaves)
        a.persistence = float(persistence)
        a.frequency = float(frequency)
        stream = current_stream()
        for s0 in range(0, nslices, 32768):         # grid.z limit
            n = min(32768, nslices - s0)
            a.out = out.data_ptr() + 8 * s0 * H * W
            a.nslices = n
            if zt is not None:
                a.zvals = zt.data_ptr() + 8 * s0
            a.z0 = int(z0) + s0
            check(lib().anoddpm_simplex3_octaves_f64(ctypes.byref(a), stream), "simplex3_octaves_f64")
        return out.cpu().numpy()

    # ------------------------------------------------------------------ reference API (3-D path)
    def noise3(self, x, y, z):
        return self.noise3array(np.array([x], dtype=np.float64), np.array([y], dtype=np.float64),
                                np.array([z], dtype=np.float64))[0, 0, 0]

    def noise3array(self, x, y, z):
        dev = _device()
        X = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64).reshape(-1)).to(dev)
        Y = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64).reshape(-1)).to(dev)
        Z = torch.from_numpy(np.ascontiguousarray(z, dtype=np.float64).reshape(-1)).to(dev)
        out = torch.empty((Z.numel(), Y.numel(), X.numel()), dtype=torch.float64, device=dev)
        check(lib().anoddpm_simplex3_grid_f64(ptr(out), ptr(X), X.numel(), ptr(Y), Y.numel(), ptr(Z), Z.numel(),
                                              ptr(self.device_tables(dev)), current_stream()), "simplex3_grid_f64")
        return out.cpu().numpy()

    def rand_3d_octaves(self, shape, octaves=1, persistence=0.5, frequency=32):
        """Layered fractal noise over a (Z, Y, X) index grid -> float64[Z, Y, X] (simplex.py:37-54)."""
        assert len(shape) == 3
        return self._octaves_f64(None, 0, int(shape[0]), int(shape[1]), int(shape[2]), octaves, persistence, frequency)

    def rand_3d_fixed_T_octaves(self, shape, T, octaves=1, persistence=0.5, frequency=32):
        """Layered fractal noise of an (H, W) image at z = T -> float64[len(T), H, W] (simplex.py:75-93)."""
        assert len(shape) == 2
        T = np.atleast_1d(np.asarray(T))
        return self._octaves_f64(T, 0, T.size, int(shape[0]), int(shape[1]), octaves, persistence, frequency)

    # ------------------------------------------------------------------ reference API (2-D path)
    # Dead code on the reference's hot path (its call sites are commented out, GaussianDiffusion.py:115-118,127-130)
    # but part of Simplex_CLASS's surface (simplex.py:25-29, 56-73).  Upstream's _noise2a / rand_2d_octaves are
    # only well defined for SQUARE grids (simplex.py:315-318 indexes noise[i * y.size + j]; :69 adds a (W,H) array to
    # a (H,W) field, which numpy rejects), so non-square requests raise ValueError here too.
    def noise2(self, x, y):
        return self.noise2array(np.array([x], dtype=np.float64), np.array([y], dtype=np.float64))[0, 0]

    def noise2array(self, x, y):
        dev = _device()
        X = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64).reshape(-1)).to(dev)
        Y = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64).reshape(-1)).to(dev)
        if X.numel() != Y.numel():
            raise ValueError("noise2array: upstream's flat indexing (simplex.py:315-318) is only defined for square grids")
        n = X.numel()
        out = torch.empty((n, n), dtype=torch.float64, device=dev)
        check(lib().anoddpm_simplex2_grid_f64(ptr(out), ptr(X), ptr(Y), n, ptr(self.device_tables(dev)), current_stream()),
              "simplex2_grid_f64")
        return out.cpu().numpy()

    def rand_2d_octaves(self, shape, octaves=1, persistence=0.5, frequency=32):
        """Layered fractal noise over a (Y, X) index grid -> float64[Y, X] (simplex.py:56-73)."""
        assert len(shape) == 2
        if int(shape[0]) != int(shape[1]):
            raise ValueError("operands could not be broadcast together: upstream adds a "
                             f"({shape[1]},{shape[0]}) array to a ({shape[0]},{shape[1]}) field (simplex.py:69)")
        dev = _device()
        n = int(shape[0])
        out = torch.empty((n, n), dtype=torch.float64, device=dev)
        check(lib().anoddpm_simplex2_octaves_f64(ptr(out), n, ptr(self.device_tables(dev)), int(octaves),
                                                 float(persistence), float(frequency), current_stream()), "simplex2_octaves_f64")
        return out.cpu().numpy()
