"""-m gpu: fused variational-bound terms (anoddpm_vlb_terms through the C ABI) against the reference-generated
golden values and the oracle; calc_vlb_xt / calc_total_vlb semantics on top of it."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN

DEV = "cuda:0"
G = np.load(os.path.join(GOLDEN, "vlb_kat.npz"))
# device tanhf / logf / expf differ from ATen's CPU vector math by a few ulp per element; the means agree to:
RTOL = 2e-5


def dv(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("name", ["linear", "cosine"])
@pytest.mark.parametrize("tag", ["mixed", "zero"])
def test_fused_terms_match_reference(name, tag):
    import GaussianDiffusion as GD
    d = GD.GaussianDiffusionModel([16, 16], GD.get_beta_schedule(1000, name), noise="gauss")
    x0, eps, noise = dv(G["x0"]), dv(G["eps"]), dv(G["noise"])
    t, x_t = dv(G[f"{name}_{tag}_t"]), dv(G[f"{name}_{tag}_x_t"])
    vlb, x0_mse, mse, pred = d.vlb_terms(x0, x_t, t, eps, noise=noise)
    assert np.array_equal(pred.cpu().numpy(), G[f"{name}_{tag}_pred_x_0"])          # same fp32 expression: bit-exact
    np.testing.assert_allclose(vlb.cpu().numpy(), G[f"{name}_{tag}_vlb"], rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(x0_mse.cpu().numpy(), G[f"{name}_{tag}_x_0_mse"], rtol=RTOL, atol=0)
    np.testing.assert_allclose(mse.cpu().numpy(), G[f"{name}_{tag}_mse"], rtol=RTOL, atol=0)
    # the reference-shaped entry point (no autograd -> fused path) returns the same
    with torch.no_grad():
        r = d.calc_vlb_xt(None, x0, x_t, t, estimate_noise=eps)
    assert torch.equal(r["output"], vlb) and torch.equal(r["pred_x_0"], pred)
    # with autograd recording it stays differentiable and agrees with the fused value
    e2 = eps.clone().requires_grad_(True)
    r2 = d.calc_vlb_xt(None, x0, x_t, t, estimate_noise=e2)
    assert r2["output"].requires_grad
    np.testing.assert_allclose(r2["output"].detach().cpu().numpy(), vlb.cpu().numpy(), rtol=RTOL, atol=1e-7)


def test_full_size_and_ragged_vs_oracle():
    from oracle import diffusion_oracle as do
    import GaussianDiffusion as GD
    tb = do.tables(do.beta_schedule(1000, "linear"))
    d = GD.GaussianDiffusionModel([256, 256], GD.get_beta_schedule(1000, "linear"), noise="gauss")
    for shape in ((4, 1, 256, 256), (3, 3, 7, 5), (0, 1, 8, 8)):
        g = torch.Generator().manual_seed(3)
        x0 = (torch.rand(shape, generator=g) * 2 - 1).round(decimals=1)           # many exact +-1 values
        x_t, eps, noise = (torch.randn(shape, generator=g) for _ in range(3))
        t = torch.randint(0, 1000, (shape[0],), generator=g)
        if shape[0]:
            t[0] = 0
        vlb, m0, me, pred = d.vlb_terms(x0.to(DEV), x_t.to(DEV), t.to(DEV), eps.to(DEV), noise=noise.to(DEV))
        if shape[0] == 0:
            assert vlb.shape == (0,)
            continue
        ov, om0, ome, opred = do.vlb_terms(tb, x0, x_t, t, eps, noise)
        assert torch.equal(pred.cpu(), opred)
        np.testing.assert_allclose(vlb.cpu().numpy(), ov.numpy(), rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(m0.cpu().numpy(), om0.numpy(), rtol=RTOL)
        np.testing.assert_allclose(me.cpu().numpy(), ome.numpy(), rtol=RTOL)


def test_calc_total_vlb_structure_and_consistency():
    """T = 100 chain: shapes / keys of GaussianDiffusion.py:470-478, and every column equals the oracle evaluated on
    the very noise the device drew (recovered from x_t)."""
    from oracle import diffusion_oracle as do
    import GaussianDiffusion as GD
    T = 100
    betas = GD.get_beta_schedule(T, "linear")
    d = GD.GaussianDiffusionModel([16, 16], betas, noise="gauss")
    tb = do.tables(betas)
    model = lambda x, t: 0.3 * x - 0.05 * t.view(-1, 1, 1, 1).float() / T
    x0 = torch.rand(2, 1, 16, 16, device=DEV) * 2 - 1
    torch.manual_seed(5)
    out = d.calc_total_vlb(x0, model, {"Batch_Size": 2})
    assert set(out) == {"total_vlb", "prior_vlb", "vb", "x_0_mse", "mse"}
    assert out["vb"].shape == (2, T) and out["x_0_mse"].shape == (2, T) and out["mse"].shape == (2, T)
    assert out["total_vlb"].shape == (2,) and torch.isfinite(out["total_vlb"]).all()
    torch.manual_seed(5)
    for col, t in enumerate(reversed(range(T))):
        noise = torch.randn_like(x0)
        tb_t = torch.full((2,), t, dtype=torch.int64)
        x_t = do.q_sample(tb, x0.cpu(), tb_t, noise.cpu())
        ov, om0, ome, _ = do.vlb_terms(tb, x0.cpu(), x_t, tb_t, model(x_t, tb_t), noise.cpu())
        np.testing.assert_allclose(out["vb"][:, col].cpu().numpy(), ov.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(out["x_0_mse"][:, col].cpu().numpy(), om0.numpy(), rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(out["mse"][:, col].cpu().numpy(), ome.numpy(), rtol=1e-4, atol=1e-7)


def test_evaluation_testing_runs_the_reference_loop_on_device(capsys):
    """evaluation.testing (evaluation.py:90-186; called by diffusion_training.py:153): draw counts from the loader, the
    chains / VLB / PSNR passes in upstream's order, the six printed lines -- video dump skipped.  Real `UNetModel`s."""
    import GaussianDiffusion as GD
    import UNet as UN
    import evaluation as EV
    T, B = 200, 2
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(T, "linear"), noise="gauss")
    torch.manual_seed(11)
    model = UN.UNetModel(32, 32, attention_resolutions="16").to(DEV)
    with torch.no_grad():
        for n, p in model.named_parameters():                     # zero-init output conv would make every eps 0
            if n.startswith("out.2"):
                p.normal_(0, 0.05)
    ema = UN.UNetModel(32, 32, attention_resolutions="16").to(DEV)
    ema.load_state_dict(model.state_dict())
    drawn = []

    def loader():
        g = torch.Generator().manual_seed(1)
        while True:
            drawn.append(1)
            yield {"image": (torch.rand(B, 1, 32, 32, generator=g) * 2 - 1)}

    args = {"arg_num": "t", "sample_distance": 250, "dataset": "mri", "Batch_Size": B, "T": T}
    res = EV.testing(loader(), d, args, ema, model, test_iters=0)
    rounds = 0 // B + 5
    assert len(drawn) == 2 + 2 * rounds                            # i in {100, 200}, then VLB and PSNR batches
    assert res["sequence_lengths"] == [102, 202]                  # "half": t_distance + 2 (GaussianDiffusion.py:320-359)
    out = capsys.readouterr().out
    for key in ("total VLB", "prior VLB", "vb @ t=200", "x_0_mse @ t=200", "mse @ t=200", "PSNR"):
        assert f"Test set {key}:" in out
    for k in ("total_vlb", "prior_vlb", "vb@200", "x_0_mse@200", "mse@200", "PSNR"):
        assert np.isfinite(res[k]).all(), k
    assert not model.training and not ema.training
    # keyword form of diffusion_training.py:153
    res2 = EV.testing(loader(), d, ema=ema, args=args, model=model, test_iters=0, sequences=False)
    assert res2["sequence_lengths"] == []


@pytest.mark.parametrize("name", ["linear", "cosine"])
def test_calc_total_vlb_matches_reference_fixture(name, monkeypatch):
    """calc_total_vlb against the REFERENCE's own loop (tests/golden/vlb_total_kat.npz: T = 100, analytic eps-model, its
    `torch.randn_like` draws injected and replayed here in the same order): all five returned curves."""
    import GaussianDiffusion as GD
    GT = np.load(os.path.join(GOLDEN, "vlb_total_kat.npz"))
    T = int(GT["T"])
    d = GD.GaussianDiffusionModel([16, 16], GD.get_beta_schedule(T, name), noise="gauss")
    x0, draws = dv(GT["x0"]), dv(GT["draws"])
    B = x0.shape[0]
    model = lambda x, t: 0.3 * x - 0.05 * t.view(-1, 1, 1, 1).float() / T
    it = iter(draws)
    monkeypatch.setattr(torch, "randn_like", lambda x, *a, **k: next(it).clone())
    out = d.calc_total_vlb(x0, model, {"Batch_Size": B})
    monkeypatch.undo()
    assert next(it, None) is None                                    # exactly T draws, in the reference's order
    for k in ("vb", "x_0_mse", "mse"):
        np.testing.assert_allclose(out[k].cpu().numpy(), GT[f"{name}_{k}"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out["prior_vlb"].cpu().numpy(), GT[f"{name}_prior_vlb"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(out["total_vlb"].cpu().numpy(), GT[f"{name}_total_vlb"], rtol=1e-4)
