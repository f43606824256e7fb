"""Import the upstream AnoDDPM reference (read-only, /root/reference) with stubbed
third-party modules.  Used ONLY by tests/golden/make_golden.py (fixture generator) and by
the optional `-m "not gpu"` cross-checks that skip when /root/reference is absent (it does
not exist on the GPU box).  Recipe: SURVEY.md section 8c.
"""
import importlib
import os
import sys
import types

REF = os.environ.get("ANODDPM_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF, "simplex.py"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load():
    """Returns (simplex, UNet, GaussianDiffusion) reference modules under private names."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    import matplotlib
    matplotlib.use("Agg")

    saved = {k: sys.modules.get(k) for k in
             ("simplex", "UNet", "GaussianDiffusion", "helpers", "evaluation", "numba",
              "torchvision", "torchvision.utils", "skimage", "skimage.metrics", "cv2", "nibabel")}

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    _stub("numba", njit=njit, prange=range)
    tv = _stub("torchvision")
    tv.utils = _stub("torchvision.utils", make_grid=None)
    tv.datasets = _stub("torchvision.datasets")
    tv.transforms = _stub("torchvision.transforms")
    sk = _stub("skimage")
    sk.metrics = _stub("skimage.metrics", structural_similarity=lambda *a, **k: 0.0)
    _stub("cv2")
    _stub("nibabel")
    for k in ("simplex", "UNet", "GaussianDiffusion", "helpers", "evaluation"):
        sys.modules.pop(k, None)

    sys.path.insert(0, REF)
    try:
        ref_simplex = importlib.import_module("simplex")
        ref_unet = importlib.import_module("UNet")
        ref_gd = importlib.import_module("GaussianDiffusion")
    finally:
        sys.path.remove(REF)
        # leave the reference modules reachable only through the returned handles
        for k in ("simplex", "UNet", "GaussianDiffusion", "helpers", "evaluation", "numba",
                  "torchvision", "torchvision.utils", "torchvision.datasets",
                  "torchvision.transforms", "skimage", "skimage.metrics", "cv2", "nibabel"):
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
    return ref_simplex, ref_unet, ref_gd
