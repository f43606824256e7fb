"""Pins oracle/diffusion_oracle.py against reference-generated fixtures
(GaussianDiffusion.py:12-29, 184-217, 228-318, 361-382).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as do

from conftest import GOLDEN


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(GOLDEN, "diffusion_kat.npz"))


@pytest.mark.parametrize("name", ["linear", "cosine"])
def test_tables_bit_exact(kat, name):
    betas = do.beta_schedule(1000, name)
    assert (betas.view(np.uint64) == kat[f"{name}_betas"].view(np.uint64)).all()
    tb = do.tables(betas)
    for k in ("sqrt_alphas", "sqrt_betas", "alphas_cumprod", "alphas_cumprod_prev",
              "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
              "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
              "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
        assert (tb[k].view(np.uint64) == kat[f"{name}_{k}"].view(np.uint64)).all(), k


def test_schedule_other_T_and_errors(kat):
    assert (do.beta_schedule(250, "linear") == kat["linear_T250_betas"]).all()
    with pytest.raises(NotImplementedError):
        do.beta_schedule(10, "quadratic")


@pytest.mark.parametrize("name", ["linear", "cosine"])
def test_sampling_bit_exact(kat, name):
    tb = do.tables(do.beta_schedule(1000, name))
    x, eps, noise = (torch.from_numpy(kat[k]) for k in ("x", "eps", "noise"))
    t = torch.from_numpy(kat["t"])
    assert torch.equal(do.q_sample(tb, x, t, noise), torch.from_numpy(kat[f"{name}_sample_q"]))
    assert torch.equal(do.q_sample_gradual(tb, x, t, noise), torch.from_numpy(kat[f"{name}_sample_q_gradual"]))
    pmv = do.p_mean_variance_eps(tb, x, t, eps)
    for k in ("mean", "variance", "log_variance", "pred_x_0"):
        assert torch.equal(pmv[k].contiguous(), torch.from_numpy(kat[f"{name}_pmv_{k}"])), k
    s, p0 = do.p_sample_update(tb, x, t, eps, noise)
    assert torch.equal(s, torch.from_numpy(kat[f"{name}_sample_p_sample"]))
    assert torch.equal(p0, torch.from_numpy(kat[f"{name}_sample_p_pred_x_0"]))
    assert torch.equal(s[0], pmv["mean"][0])          # t == 0 adds no noise
