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
