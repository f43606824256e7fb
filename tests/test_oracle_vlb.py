"""oracle/diffusion_oracle.vlb_terms against values produced by the reference's calc_vlb_xt / predict_eps_from_x_0
(tests/golden/vlb_kat.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import diffusion_oracle as do

G = np.load(os.path.join(GOLDEN, "vlb_kat.npz"))


@pytest.mark.parametrize("name", ["linear", "cosine"])
@pytest.mark.parametrize("tag", ["mixed", "zero"])
def test_vlb_terms_match_reference(name, tag):
    tb = do.tables(do.beta_schedule(1000, name))
    x0, eps, noise = (torch.from_numpy(G[k]) for k in ("x0", "eps", "noise"))
    t, x_t = torch.from_numpy(G[f"{name}_{tag}_t"]), torch.from_numpy(G[f"{name}_{tag}_x_t"])
    vlb, x0_mse, mse, pred = do.vlb_terms(tb, x0, x_t, t, eps, noise)
    assert np.array_equal(pred.numpy(), G[f"{name}_{tag}_pred_x_0"])
    np.testing.assert_allclose(vlb.numpy(), G[f"{name}_{tag}_vlb"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(x0_mse.numpy(), G[f"{name}_{tag}_x_0_mse"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(mse.numpy(), G[f"{name}_{tag}_mse"], rtol=1e-6, atol=0)
    if tag == "mixed":
        assert (G[f"{name}_{tag}_t"] == 0).sum() == 1            # both branches are exercised


@pytest.mark.parametrize("name", ["linear", "cosine"])
def test_total_vlb_loop_matches_reference(name):
    """calc_total_vlb itself (GaussianDiffusion.py:445-478), T = 100: the oracle's per-step terms over the reference's injected
    draws reproduce the reference's vb / x_0_mse / mse curves, and vb.sum + prior its total."""
    GT = np.load(os.path.join(GOLDEN, "vlb_total_kat.npz"))
    T = int(GT["T"])
    tb = do.tables(do.beta_schedule(T, name))
    x0, draws = torch.from_numpy(GT["x0"]), torch.from_numpy(GT["draws"])
    B = x0.shape[0]
    model = lambda x, t: 0.3 * x - 0.05 * t.view(-1, 1, 1, 1).float() / T
    cols = []
    for col, t in enumerate(reversed(range(T))):
        tt = torch.full((B,), t, dtype=torch.int64)
        x_t = do.q_sample(tb, x0, tt, draws[col])
        cols.append(do.vlb_terms(tb, x0, x_t, tt, model(x_t, tt), draws[col])[:3])
    for k, key in enumerate(("vb", "x_0_mse", "mse")):
        got = torch.stack([c[k] for c in cols], dim=1).numpy()
        np.testing.assert_allclose(got, GT[f"{name}_{key}"], rtol=2e-6, atol=1e-9)
    vb = torch.stack([c[0] for c in cols], dim=1)
    np.testing.assert_allclose((vb.sum(dim=1)).numpy() + GT[f"{name}_prior_vlb"], GT[f"{name}_total_vlb"], rtol=1e-5)
