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


# ---------------------------------------------------------------------------------------------------------------------------------
# round 6: the serial detection loops (GaussianDiffusion.py:480-594) -- the oracle's restatement against the reference's own run
# (tests/golden/detection_loops_kat.npz: keyed torch.randn_like draws, seeded numpy stream for the simplex forward noise)

def _loops_setup():
    import sys
    sys.path.insert(0, GOLDEN)
    import keyed
    from oracle import unet_oracle as uo
    g = np.load(os.path.join(GOLDEN, "detection_loops_kat.npz"))
    kw = dict(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8")
    sd = uo.fill_deterministic(uo.param_shapes(32, 32, "", 2, "16,8", 1, True))
    model = lambda x, t: uo.forward(sd, x, t, 32, 32, attention_resolutions="16,8", n_heads=2)
    T = int(g["T"])
    tb = do.tables(do.beta_schedule(T, "linear"))
    return g, keyed, model, tb, T, torch.from_numpy(g["x0"]), torch.from_numpy(g["mask"])


def _simplex_forward(np_seed):
    """generate_simplex_noise (GaussianDiffusion.py:96-137) through the C oracle: a fresh seed from the global numpy stream per
    call, the (octave 6, persistence 0.8) field at z = t, fp64 -> fp32."""
    from oracle import simplex_oracle as so
    np.random.seed(np_seed)
    sx = so.OracleSimplex(3)

    def draw(shape, t, frequency):
        sx.newSeed()
        f = sx.rand_3d_fixed_T_octaves(shape[-2:], np.array([t]), 6, 0.8, frequency)
        return torch.from_numpy(f.astype(np.float32)).reshape(1, 1, *shape[-2:])
    return draw


def test_detection_B_gauss_loop_matches_reference():
    g, keyed, model, tb, T, x0, mask = _loops_setup()
    navg = int(g["B_gauss_total_avg"])
    settings = [(None, t) for t in range(50, int(T * 0.8), 50)][:2]          # the first two settings (6 chains, 450 steps) keep it short
    grids = do.detection_loop(tb, model, x0, mask, settings, navg,
                              lambda key, c: keyed.keyed_normal(c, keyed.FORWARD, x0.shape),
                              lambda c, t: keyed.keyed_normal(c, t, x0.shape))
    for j, gr in enumerate(grids):
        ref = torch.from_numpy(g["B_gauss"][j])
        assert gr.shape == ref.shape
        assert float((gr[:5] - ref[:5]).abs().max()) < 2e-4, (j, float((gr[:5] - ref[:5]).abs().max()))   # x0, chains, mean
        assert float((gr[5] - ref[5]).abs().max()) < 1e-3                                                  # mse image
        away = ref[5].abs() > 2e-3
        assert torch.equal(gr[6][away], ref[6][away]) and torch.equal(gr[7], ref[7])


def test_detection_B_octave_and_detection_A_forward_noise_order():
    """The simplex modes: the forward noise consumes the global numpy stream once per chain in upstream's loop order (detection_A:
    per frequency, per t_distance, per avg).  First setting of each routine."""
    g, keyed, model, tb, T, x0, mask = _loops_setup()
    draw = _simplex_forward(int(g["B_octave_np_seed"]))
    grids = do.detection_loop(tb, model, x0, mask, [(64, 50)], int(g["B_octave_total_avg"]),
                              lambda f, c: draw(x0.shape, 50, f), lambda c, t: keyed.keyed_normal(c, t, x0.shape))
    ref = torch.from_numpy(g["B_octave"][0])
    assert float((grids[0][:5] - ref[:5]).abs().max()) < 2e-4
    draw = _simplex_forward(int(g["A_np_seed"]))
    grids = do.detection_loop(tb, model, x0, mask, [(2 ** 7, 50)], int(g["A_total_avg"]),
                              lambda f, c: draw(x0.shape, 50, f), lambda c, t: keyed.keyed_normal(c, t, x0.shape))
    ref = torch.from_numpy(g["A"][0])                     # cat[x0, output[:3] (= 2 chains), mean, mse, thr, mask]
    assert float((grids[0][:4] - ref[:4]).abs().max()) < 2e-4
