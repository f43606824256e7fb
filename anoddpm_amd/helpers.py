"""Config / checkpoint helpers with the reference's surface (helpers.py:1-101).

`from helpers import *` in the reference also leaks `torch, os, json, defaultdict` into its
importers (GaussianDiffusion.py:8 relies on that); the same names are exported here.
`gridify_output` is implemented without torchvision (not installed in this image).

`load_checkpoint` / `load_parameters` (helpers.py:26-93) are plain host Python: they resolve argv and the ./model
checkpoint directory for the CLI drivers and return the checkpoint dict of SURVEY 8b ("Data formats") untouched.
"""
import json
import os
from collections import defaultdict

import torch

__all__ = ["json", "os", "defaultdict", "torch", "gridify_output", "defaultdict_from_json", "load_checkpoint",
           "load_parameters"]


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


def load_checkpoint(param, use_checkpoint, device):
    """helpers.py:26-45: `./model/diff-params-ARGS={param}/params-final.pt`, or the newest checkpoint under
    `.../checkpoint/` that `torch.load` can read (files that raise RuntimeError -- truncated writes -- are skipped).
    Returns the dict `{'n_epoch', 'model_state_dict', 'optimizer_state_dict', 'ema', 'args'}` as saved
    (diffusion_training.py:169-189).  Like upstream: an empty / all-corrupt checkpoint directory ends in
    UnboundLocalError, a missing directory in FileNotFoundError.  `weights_only=False`: the saved `args` entry is a
    `defaultdict(str)` (diffusion_training.py:303), which torch >= 2.6's default safe unpickler refuses -- upstream was written
    against the old default, and the files are the user's own checkpoints."""
    root = f'./model/diff-params-ARGS={param}'
    if not use_checkpoint:
        return torch.load(f'{root}/params-final.pt', map_location=device, weights_only=False)
    names = os.listdir(f'{root}/checkpoint')
    names.sort(reverse=True)
    for name in names:
        try:
            loaded_model = torch.load(f"{root}/checkpoint/{name}", map_location=device, weights_only=False)
            break
        except RuntimeError:
            continue
    return loaded_model


def load_parameters(device):
    """helpers.py:48-93: argv (`28`, `args28`, `args28.json`, optional leading `CHECKPOINT`) or, without arguments, the
    entries of ./model, resolved to `(args, checkpoint)`.  Only the FIRST parameter is loaded (upstream returns inside its
    loop); a checkpoint without an `args` entry falls back to `./test_args/args{param[17:]}.json`; `noise_fn` defaults
    to "gauss"; anything else raises ValueError."""
    import sys

    params = sys.argv[1:] if len(sys.argv[1:]) > 0 else os.listdir("./model")
    if ".DS_Store" in params:
        params.remove(".DS_Store")
    use_checkpoint = params[0] == "CHECKPOINT"
    if use_checkpoint:
        params = params[1:]
    print(params)
    for param in params:
        if param.isnumeric():
            output = load_checkpoint(param, use_checkpoint, device)
        elif param[:4] == "args" and param[-5:] == ".json":
            output = load_checkpoint(param[4:-5], use_checkpoint, device)
        elif param[:4] == "args":
            output = load_checkpoint(param[4:], use_checkpoint, device)
        else:
            raise ValueError(f"Unsupported input {param}")
        if "args" in output:
            args = output["args"]
        else:
            try:
                with open(f'./test_args/args{param[17:]}.json', 'r') as f:
                    args = json.load(f)
                args['arg_num'] = param[17:]
                args = defaultdict_from_json(args)
            except FileNotFoundError:
                raise ValueError(f"args{param[17:]} doesn't exist for {param}")
        if "noise_fn" not in args:
            args["noise_fn"] = "gauss"
        return args, output
