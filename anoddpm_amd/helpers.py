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


def _model_dir(arg_num):
    return os.path.join(".", "model", f"diff-params-ARGS={arg_num}")


def load_checkpoint(param, use_checkpoint, device):
    """helpers.py:26-45.  `use_checkpoint` False: `./model/diff-params-ARGS={param}/params-final.pt`; True: the newest file under
    `.../checkpoint/` (names sorted descending) that `torch.load` can read -- files that raise RuntimeError (truncated writes)
    are skipped.  Returns the dict `{'n_epoch', 'model_state_dict', 'optimizer_state_dict', 'ema', 'args'}` as saved
    (diffusion_training.py:169-189).  Error behaviour as upstream: a missing directory is FileNotFoundError, a checkpoint
    directory without one readable file UnboundLocalError.  Files are read through a restricted unpickler (`_load_saved_dict`)."""
    root = _model_dir(param)
    if not use_checkpoint:
        return _load_saved_dict(os.path.join(root, "params-final.pt"), device)
    ckpt_dir = os.path.join(root, "checkpoint")
    for name in sorted(os.listdir(ckpt_dir), reverse=True):
        try:
            newest_readable = _load_saved_dict(os.path.join(ckpt_dir, name), device)
        except RuntimeError:
            continue
        break
    return newest_readable                                           # unbound when nothing could be read: UnboundLocalError, as upstream


def _load_saved_dict(path, device):
    """torch.load of a training checkpoint through a RESTRICTED unpickler (round-5 advisor finding).  torch's own weights-only
    loader cannot rebuild the `args` entry -- a `collections.defaultdict(str)` (diffusion_training.py:303; its SETITEMS is refused
    even when the class is allow-listed) -- so the file is read with an unpickler whose `find_class` admits exactly torch's
    weights-only allow-list plus `collections.defaultdict` and `builtins.str`.  A pickle that names any other global is refused
    (pickle.UnpicklingError) unless ANODDPM_UNSAFE_CHECKPOINTS=1 says the files are the user's own and the full unpickler may run."""
    import pickle
    import types
    from torch._weights_only_unpickler import _get_allowed_globals

    allowed = dict(_get_allowed_globals())
    allowed.update({"collections.defaultdict": defaultdict, "builtins.str": str, "__builtin__.unicode": str, "__builtin__.str": str})   # (torch.save writes protocol 2)

    class _Restricted(pickle.Unpickler):
        def find_class(self, module, name):
            key = f"{module}.{name}"
            if key in allowed:
                return allowed[key]
            raise pickle.UnpicklingError(f"global {key} is not on the checkpoint allow-list")

    shim = types.ModuleType("anoddpm_restricted_pickle")
    shim.Unpickler = _Restricted
    shim.load = lambda f, **kw: _Restricted(f, **kw).load()
    shim.__dict__.update({k: getattr(pickle, k) for k in ("UnpicklingError", "PickleError", "HIGHEST_PROTOCOL", "dumps", "dump", "Pickler")})
    try:
        return torch.load(path, map_location=device, weights_only=False, pickle_module=shim)
    except pickle.UnpicklingError as e:
        if os.environ.get("ANODDPM_UNSAFE_CHECKPOINTS", "0") != "1":
            raise pickle.UnpicklingError(f"{path}: refused by the restricted checkpoint loader ({e}); set ANODDPM_UNSAFE_CHECKPOINTS=1 "
                                         "to unpickle it fully (arbitrary code execution: only for files you wrote yourself)") from e
        print(f"load_checkpoint: {path} needs the full unpickler (ANODDPM_UNSAFE_CHECKPOINTS=1)")
        return torch.load(path, map_location=device, weights_only=False)


def _arg_number(spec):
    """`28`, `args28` or `args28.json` -> "28" (helpers.py:71-78); anything else is a ValueError."""
    if spec.isnumeric():
        return spec
    if spec.startswith("args"):
        return spec[4:-5] if spec.endswith(".json") else spec[4:]
    raise ValueError(f"Unsupported input {spec}")


def load_parameters(device):
    """helpers.py:48-93: which trained model the detection / evaluation drivers work on.  The specs come from argv (an optional
    leading `CHECKPOINT` selects the newest checkpoint instead of the final parameters) or, without arguments, from the entries
    of ./model; only the FIRST spec is loaded (upstream returns from inside its loop).  Returns `(args, checkpoint)`: `args` is
    the checkpoint's own entry or, for checkpoints saved without one, `./test_args/args{spec[17:]}.json` (the tail of a
    `diff-params-ARGS=N` directory name) as a defaultdict with `arg_num` set; `noise_fn` defaults to "gauss"."""
    import sys

    specs = list(sys.argv[1:]) or os.listdir("./model")
    specs = [s_ for s_ in specs if s_ != ".DS_Store"]
    newest = specs[0] == "CHECKPOINT"                                # IndexError on an empty list, as upstream
    if newest:
        specs = specs[1:]
    print(specs)
    for spec in specs:
        checkpoint = load_checkpoint(_arg_number(spec), newest, device)
        args = checkpoint.get("args") if "args" in checkpoint else None
        if args is None:
            tail = spec[17:]
            path = os.path.join(".", "test_args", f"args{tail}.json")
            if not os.path.exists(path):
                raise ValueError(f"args{tail} doesn't exist for {spec}")
            with open(path, "r") as f:
                args = defaultdict_from_json(dict(json.load(f), arg_num=tail))
        args.setdefault("noise_fn", "gauss")
        return args, checkpoint
