"""ctypes binding of oracle/simplex_oracle.c (TEST INFRASTRUCTURE, see that file's header).

Mirrors the reference `Simplex_CLASS` surface (simplex.py:14-93) closely enough that parity
tests read like calls into the reference.  Parity status: pinned bit-for-bit by
tests/golden/simplex_*.npz (generated from the imported reference).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_simplex.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "simplex_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_simplex.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i64p = ctypes.POINTER(ctypes.c_int64)
        f64p = ctypes.POINTER(ctypes.c_double)
        L.oracle_simplex_init.argtypes = [ctypes.c_int64, i64p, i64p]
        L.oracle_simplex_init.restype = None
        L.oracle_noise3.argtypes = [ctypes.c_double] * 3 + [i64p, i64p]
        L.oracle_noise3.restype = ctypes.c_double
        L.oracle_noise3_grid.argtypes = [f64p, ctypes.c_int64, f64p, ctypes.c_int64, f64p,
                                         ctypes.c_int64, i64p, i64p, f64p]
        L.oracle_noise3_grid.restype = None
        L.oracle_octaves.argtypes = [i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                     ctypes.c_int, ctypes.c_double, ctypes.c_double, i64p, i64p, f64p]
        L.oracle_octaves.restype = None
        L.oracle_noise2.argtypes = [ctypes.c_double, ctypes.c_double, i64p]
        L.oracle_noise2.restype = ctypes.c_double
        L.oracle_noise2_grid.argtypes = [f64p, f64p, ctypes.c_int64, i64p, f64p]
        L.oracle_noise2_grid.restype = None
        L.oracle_octaves2.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_double, i64p, f64p]
        L.oracle_octaves2.restype = None
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def wrap_seed(seed: int) -> int:
    """Python int -> the int64 the LCG's first wrap would see (simplex.py:166-171,181)."""
    return ((int(seed) + (1 << 63)) % (1 << 64)) - (1 << 63)


def init(seed):
    perm = np.zeros(256, dtype=np.int64)
    pgi3 = np.zeros(256, dtype=np.int64)
    lib().oracle_simplex_init(wrap_seed(seed), _p(perm, ctypes.c_int64), _p(pgi3, ctypes.c_int64))
    return perm, pgi3


class OracleSimplex:
    def __init__(self, seed=3):
        self.newSeed(seed)

    def newSeed(self, seed=None):
        if not seed:
            seed = np.random.randint(-10000000000, 10000000000)
        self._perm, self._perm_grad_index3 = init(seed)

    def noise3(self, x, y, z):
        return lib().oracle_noise3(float(x), float(y), float(z), _p(self._perm, ctypes.c_int64),
                                   _p(self._perm_grad_index3, ctypes.c_int64))

    def noise3array(self, x, y, z):
        X = np.ascontiguousarray(x, dtype=np.float64)
        Y = np.ascontiguousarray(y, dtype=np.float64)
        Z = np.ascontiguousarray(z, dtype=np.float64)
        out = np.empty((Z.size, Y.size, X.size), dtype=np.float64)
        lib().oracle_noise3_grid(_p(X, ctypes.c_double), X.size, _p(Y, ctypes.c_double), Y.size,
                                 _p(Z, ctypes.c_double), Z.size, _p(self._perm, ctypes.c_int64),
                                 _p(self._perm_grad_index3, ctypes.c_int64), _p(out, ctypes.c_double))
        return out

    def _octaves(self, zvals, height, width, octaves, persistence, frequency):
        zvals = np.ascontiguousarray(zvals, dtype=np.int64)
        out = np.empty((zvals.size, height, width), dtype=np.float64)
        lib().oracle_octaves(_p(zvals, ctypes.c_int64), zvals.size, height, width, int(octaves),
                             float(persistence), float(frequency), _p(self._perm, ctypes.c_int64),
                             _p(self._perm_grad_index3, ctypes.c_int64), _p(out, ctypes.c_double))
        return out

    def rand_3d_octaves(self, shape, octaves=1, persistence=0.5, frequency=32):
        assert len(shape) == 3
        return self._octaves(np.arange(shape[0]), shape[1], shape[2], octaves, persistence, frequency)

    def rand_3d_fixed_T_octaves(self, shape, T, octaves=1, persistence=0.5, frequency=32):
        assert len(shape) == 2
        return self._octaves(np.atleast_1d(T), shape[0], shape[1], octaves, persistence, frequency)

    # ---- 2-D (simplex.py:25-29, 56-73, 211-318)
    def noise2(self, x, y):
        return lib().oracle_noise2(float(x), float(y), _p(self._perm, ctypes.c_int64))

    def noise2array(self, x, y):
        X = np.ascontiguousarray(x, dtype=np.float64)
        Y = np.ascontiguousarray(y, dtype=np.float64)
        if X.size != Y.size:
            raise ValueError("upstream's _noise2a indexes noise[i * y.size + j] and reshapes to (x.size, y.size): "
                             "only square grids are well defined")
        out = np.empty((Y.size, X.size), dtype=np.float64)
        lib().oracle_noise2_grid(_p(X, ctypes.c_double), _p(Y, ctypes.c_double), X.size,
                                 _p(self._perm, ctypes.c_int64), _p(out, ctypes.c_double))
        return out

    def rand_2d_octaves(self, shape, octaves=1, persistence=0.5, frequency=32):
        assert len(shape) == 2
        if shape[0] != shape[1]:
            raise ValueError("upstream adds a (W,H) array to a (H,W) field: only square shapes work")
        out = np.empty(tuple(shape), dtype=np.float64)
        lib().oracle_octaves2(shape[0], int(octaves), float(persistence), float(frequency),
                              _p(self._perm, ctypes.c_int64), _p(out, ctypes.c_double))
        return out
