"""-m "not gpu": the CPU training oracle (oracle/train_oracle.py) against the reference-generated training
fixtures (tests/golden/train_*.npz: two optimiser steps of the reference loop body, diffusion_training.py:99-107)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

CASES = {
    "i64_b64_h2": dict(img_size=64, base_channels=64, n_heads=2, attention_resolutions="16,8"),
    "i128_b32_hc32": dict(img_size=128, base_channels=32, n_head_channels=32, attention_resolutions="16,8"),
}

# reference-run fixtures that are only replayed on the device (the CPU oracle would need minutes per step at this width):
# base 128 at 128^2, batch 4 -- the smallest shape whose 3x3 layers run on the Winograd F(4x4,3x3) forward / data-gradient /
# weight-gradient kernels (tests/test_gpu_training.py)
GPU_ONLY_CASES = {
    "i128_b128_f43": dict(img_size=128, base_channels=128, n_heads=2, attention_resolutions="16,8"),
}


def probe(v, n=256):
    f = v.detach().flatten()
    stride = max(1, f.numel() // n)
    return np.resize(f[::stride][:n].cpu().numpy(), n)


def check_against_fixture(g, step, keys, loss, grads, norm, params, ema, lr, tol=1e-3):
    """Shared by the CPU-oracle test and the -m gpu test (tests/test_gpu_training.py)."""
    p = f"s{step}/"
    assert abs(float(loss) - float(g[p + "loss"])) < 1e-4 * abs(float(g[p + "loss"])), (float(loss), float(g[p + "loss"]))
    assert abs(float(norm) - float(g[p + "grad_norm"])) < tol * float(g[p + "grad_norm"])
    gn_ref = g[p + "gnorm"]
    gmax = gn_ref.max()
    gabs = np.abs(g[p + "gprobe"]).max()
    worst = (0.0, "")
    for i, k in enumerate(keys):
        ref = g[p + "gprobe"][i]
        got = probe(grads[k])
        n = float(grads[k].double().norm())
        # gradients that are mathematically zero (a per-channel constant ahead of a one-channel-per-group GroupNorm)
        # are rounding noise in every implementation: errors are measured against max(|ref|, 1e-4 of the largest gradient element)
        den = max(np.abs(ref).max(), 1e-4 * gabs)
        e = np.abs(got - ref).max() / den
        en = abs(n - gn_ref[i]) / max(gn_ref[i], 1e-4 * gmax)    # noise-level tensors (norm ~1e-8 of the largest): absolute
        worst = max(worst, (e, k), (en, k + " (norm)"))
    assert worst[0] < tol, worst
    # parameters / EMA after the step: Adam normalises the update to O(lr) per element, so errors are measured
    # against the step size; elements whose gradient is noise-level may legitimately differ by up to 2*lr per step
    nstep = step + 1
    bad = tot = 0
    for i, k in enumerate(keys):
        d = np.abs(probe(params[k]) - g[p + "pprobe"][i])
        assert d.max() <= 2.2 * lr * nstep, (k, d.max())
        if gn_ref[i] < 1e-4 * gmax:
            continue          # noise-level gradient: Adam turns rounding noise into +-lr steps, only the bound above applies
        bad += int((d > 0.02 * lr).sum())
        tot += d.size
        de = np.abs(probe(ema[k]) - g[p + "eprobe"][i])
        assert de.max() <= 2.2 * lr * nstep * 1e-4 + 5e-7, (k, de.max())          # + a few fp32 ulps of O(1) values
    assert bad <= 0.002 * tot, (bad, tot)


@pytest.mark.parametrize("name", list(CASES))
def test_training_oracle_matches_reference_steps(name):
    from oracle import unet_oracle as uo
    from oracle.train_oracle import TrainState
    g = np.load(os.path.join(GOLDEN, f"train_{name}.npz"))
    kw = CASES[name]
    torch.set_num_threads(8)
    shapes = uo.param_shapes(kw["img_size"], kw["base_channels"], "", 2, kw["attention_resolutions"], 1)
    keys = [str(k) for k in g["keys"]]
    assert keys == list(shapes)
    sd = uo.perturb(uo.fill_deterministic(shapes))
    st = TrainState(sd, kw, lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    for step in range(2):
        x0, noise, t = (torch.from_numpy(g[f"s{step}/{n}"]) for n in ("x0", "noise", "t"))
        loss, grads, norm, x_t, eps = st.step(x0, t, noise)
        assert np.abs(probe(x_t, 1024) - g[f"s{step}/x_t"]).max() == 0
        assert np.abs(probe(eps, 1024) - g[f"s{step}/eps"]).max() < 1e-4
        check_against_fixture(g, step, keys, loss, grads, norm, st.params, st.ema, float(g["lr"]))
