"""-m gpu: the fused training loss (anoddpm_loss_forward / anoddpm_loss_backward behind GaussianDiffusionModel.p_loss /
calc_loss / calc_vlb_xt) against the reference's own p_loss + autograd (tests/golden/loss_kat.npz) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(GOLDEN, "loss_kat.npz"))


def dv(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _grad_close(got, ref, tail=5e-3):
    """Every element to rounding, except the few ill-conditioned decoder-NLL tail elements (see tests/test_oracle_loss.py)."""
    scale = np.abs(ref).max()
    err = np.abs(got - ref) / scale
    assert err.max() <= 2e-2 and (err > 2e-5).mean() < tail, (err.max(), (err > 2e-5).mean())


@pytest.mark.parametrize("kind", ["l1", "l2", "hybrid"])
@pytest.mark.parametrize("lw", ["none", "prop-t"])
def test_p_loss_matches_reference_values_and_gradient(kind, lw, monkeypatch):
    import GaussianDiffusion as GD
    tag = f"{kind}_{lw}"
    d = GD.GaussianDiffusionModel([16, 16], GD.get_beta_schedule(1000, "linear"), loss_type=kind, loss_weight=lw, noise="gauss")
    x0, noise = dv(G["x0"]), dv(G["noise"])
    d.noise_fn = lambda a, b: noise
    eps = dv(G["eps"]).clone().requires_grad_(True)
    t_inj = dv(G[f"{tag}_t"])
    monkeypatch.setattr(torch, "randint", lambda *a, **k: t_inj.clone())
    np.random.seed(5)                                            # the prop-t draw of the fixture (sample_t_with_weights)
    total, (ld, x_t, e_out) = d.p_loss(lambda a, b: eps, x0, {"train_start": True, "sample_distance": 800})
    monkeypatch.undo()
    assert e_out is eps and torch.equal(x_t.cpu(), torch.from_numpy(G[f"{tag}_x_t"]))
    np.testing.assert_allclose(ld["loss"].detach().cpu().numpy(), G[f"{tag}_loss"], rtol=3e-5)
    if kind == "hybrid":
        np.testing.assert_allclose(ld["vlb"].detach().cpu().numpy(), G[f"{tag}_vlb"], rtol=3e-5)
        assert list(ld) == ["vlb", "loss"]
    else:
        assert list(ld) == ["loss"]
    np.testing.assert_allclose(total.item(), G[f"{tag}_total"], rtol=3e-5)
    assert total.dim() == 0 and total.requires_grad
    total.backward()
    _grad_close(eps.grad.cpu().numpy(), G[f"{tag}_d_eps"])
    if kind == "l1":
        assert (eps.grad[:, :, 3, :2] == 0).all()                # sign(0) = 0


def test_per_sample_and_vlb_outputs_are_differentiable_too():
    """calc_loss's dict entries carry gradients like upstream's (loss['loss'].mean().backward(), a weighted sum of loss['vlb']),
    and calc_vlb_xt under autograd is the fused term with a native backward."""
    import GaussianDiffusion as GD
    from oracle import diffusion_oracle as do
    tb = do.tables(do.beta_schedule(1000, "linear"))
    d = GD.GaussianDiffusionModel([16, 16], GD.get_beta_schedule(1000, "linear"), loss_type="hybrid", noise="gauss")
    x0c, noisec, epsc = (torch.from_numpy(G[k]) for k in ("x0", "noise", "eps"))
    t = torch.tensor([3, 1, 2, 500, 998, 40])
    coef = torch.tensor([0.5, -1.0, 2.0, 0.25, 1.0, 3.0])
    d.noise_fn = lambda a, b: noisec.to(DEV)
    eps = epsc.to(DEV).requires_grad_(True)
    ld, x_t, _ = d.calc_loss(lambda a, b: eps, x0c.to(DEV), t.to(DEV))
    (ld["loss"].mean() + (ld["vlb"] * coef.to(DEV)).sum()).backward()
    e2 = epsc.clone().requires_grad_(True)
    per, vlb, _ = do.loss_terms(tb, x0c, t, e2, noisec, None, "hybrid")
    (per.mean() + (vlb * coef).sum()).backward()
    np.testing.assert_allclose(ld["loss"].detach().cpu().numpy(), per.detach().numpy(), rtol=3e-5)
    _grad_close(eps.grad.cpu().numpy(), e2.grad.numpy())
    # calc_vlb_xt with autograd recording
    eps3 = epsc.to(DEV).requires_grad_(True)
    r = d.calc_vlb_xt(None, x0c.to(DEV), x_t, t.to(DEV), estimate_noise=eps3)
    assert r["output"].requires_grad and not r["pred_x_0"].requires_grad
    (r["output"] * coef.to(DEV)).sum().backward()
    e4 = epsc.clone().requires_grad_(True)
    v4 = do.vlb_terms(tb, x0c, x_t.cpu(), t, e4)[0]
    (v4 * coef).sum().backward()
    np.testing.assert_allclose(r["output"].detach().cpu().numpy(), v4.detach().numpy(), rtol=3e-5, atol=1e-7)
    _grad_close(eps3.grad.cpu().numpy(), e4.grad.numpy())


@pytest.mark.parametrize("shape", [(4, 1, 256, 256), (3, 3, 7, 5), (0, 1, 8, 8)])
def test_full_size_ragged_and_empty_vs_oracle(shape):
    import GaussianDiffusion as GD
    from oracle import diffusion_oracle as do
    tb = do.tables(do.beta_schedule(1000, "linear"))
    g = torch.Generator().manual_seed(8)
    x0 = (torch.rand(shape, generator=g) * 2 - 1).round(decimals=1)
    noise, eps0 = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    B = shape[0]
    t = torch.randint(1, 1000, (B,), generator=g)
    w = torch.rand(B, generator=g) + 0.5
    for kind in ("l1", "l2", "hybrid"):
        d = GD.GaussianDiffusionModel(list(shape[2:]), GD.get_beta_schedule(1000, "linear"), loss_type=kind, noise="gauss")
        d.noise_fn = lambda a, b: noise.to(DEV)
        eps = eps0.to(DEV).requires_grad_(True)
        terms, x_t, _, total = d._loss_terms(lambda a, b: eps, x0.to(DEV), t.to(DEV), w.to(DEV))
        if B == 0:
            assert terms["loss"].shape == (0,)
            continue
        total.backward()
        e2 = eps0.clone().requires_grad_(True)
        per, vlb, tot = do.loss_terms(tb, x0, t, e2, noise, w, kind)
        tot.backward()
        np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), per.detach().numpy(), rtol=3e-5)
        np.testing.assert_allclose(total.item(), tot.item(), rtol=3e-5)
        _grad_close(eps.grad.cpu().numpy(), e2.grad.numpy(), tail=1e-3)
