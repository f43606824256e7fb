"""Config / checkpoint helpers with the reference's surface (helpers.py:1-101).

`from helpers import *` in the reference also leaks `torch, os, json, defaultdict` into its
importers (GaussianDiffusion.py:8 relies on that); the same names are exported here.
`gridify_output` is implemented without torchvision (not installed in this image).

Not here on purpose: `load_checkpoint` / `load_parameters` (helpers.py:26-93).  They resolve argv and the ./model
checkpoint directory for the CLI drivers -- control plane, outside the hot path; a caller
that wants them keeps the reference's own helpers.py, which needs nothing from this package.
"""
import json
import os
from collections import defaultdict

import torch

__all__ = ["json", "os", "defaultdict", "torch", "gridify_output", "defaultdict_from_json"]


def _make_grid(img, nrow, padding=2, pad_value=0):
    """Minimal torchvision.utils.make_grid for [N,C,H,W] tensors."""
    n, c, h, w = img.shape
    if c == 1:
        img = img.expand(n, 3, h, w)
        c = 3
    xmaps = min(nrow, n) if nrow > 0 else n
    ymaps = (n + xmaps - 1) // xmaps
    hh, ww = h + padding, w + padding
    grid = img.new_full((c, hh * ymaps + padding, ww * xmaps + padding), pad_value)
    k = 0
    for yy in range(ymaps):
        for xx in range(xmaps):
            if k >= n:
                break
            grid[:, yy * hh + padding:yy * hh + padding + h, xx * ww + padding:xx * ww + padding + w] = img[k]
            k += 1
    return grid


def gridify_output(img, row_size=-1):
    """helpers.py:9-16: [-1,1] images -> uint8 grid, HWC."""
    scaled = ((img + 1) * 127.5).clamp(0, 255).to(torch.uint8)
    return _make_grid(scaled, row_size, pad_value=-1).cpu().data.permute(0, 2, 1).contiguous().permute(2, 1, 0)


def defaultdict_from_json(jsonDict):
    """helpers.py:19-23: missing keys read as ''."""
    dd = defaultdict(str)
    dd.update(jsonDict)
    return dd
