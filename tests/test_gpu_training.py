"""-m gpu: the training-side kernels and the train step (diffusion_training.py:99-107) on the device."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_fused_adamw_ema_matches_torch_reference():
    from anoddpm_amd.training import FlatBuffers, FusedAdamWEMA
    from UNet import update_ema_params
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 64), torch.nn.SiLU(), torch.nn.Linear(64, 5)).to(DEV)
    ref = copy.deepcopy(net)
    ema, ema_ref = copy.deepcopy(net), copy.deepcopy(net)
    flat, flat_ema = FlatBuffers(net), FlatBuffers(ema)
    opt = FusedAdamWEMA(flat, flat_ema, lr=1e-3, weight_decay=0.01, max_norm=1.0)
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=0.01, betas=(0.9, 0.999))
    for step in range(3):
        x = torch.randn(16, 37, device=DEV) * 3
        flat.zero_grad()
        net(x).square().mean().backward()
        norm = opt.step()
        opt_ref.zero_grad()
        ref(x).square().mean().backward()
        norm_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), 1)
        opt_ref.step()
        update_ema_params(ema_ref, ref)
        assert torch.allclose(norm.squeeze(), norm_ref, rtol=1e-4)
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    for a, b in zip(ema.parameters(), ema_ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_train_step_decreases_loss_and_refreshes_hip_plan():
    import GaussianDiffusion as GD
    from UNet import UNetModel
    from anoddpm_amd.training import FlatBuffers, FusedAdamWEMA, train_step
    torch.manual_seed(1)
    np.random.seed(1)
    model = UNetModel(32, 32, n_heads=2, attention_resolutions="16,8").to(DEV)
    ema = copy.deepcopy(model)
    flat, flat_ema = FlatBuffers(model), FlatBuffers(ema)
    opt = FusedAdamWEMA(flat, flat_ema, lr=2e-3, notify=(model, ema))
    diff = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(1000, "linear"), noise="simplex")
    args = {"train_start": True, "sample_distance": 800}
    x = torch.rand(4, 1, 32, 32, device=DEV) * 2 - 1
    t = torch.tensor([5, 300, 600, 799], device=DEV)
    with torch.no_grad():
        y0 = ema(x, t).clone()
    losses = []
    for _ in range(6):
        loss, (ld, x_t, eps) = train_step(model, diff, x, args, flat, None, opt)
        losses.append(loss.item())
        assert x_t.shape == x.shape and eps.shape == x.shape
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    with torch.no_grad():
        y1 = ema(x, t)                    # EMA weights moved -> the HIP plan must have been re-packed
        yt = model(x, t)
    assert not torch.equal(y0, y1)
    yg = model(x, t)                      # differentiable path on the same weights
    assert ((yg.detach() - yt).abs().max() / yt.abs().max().clamp_min(1e-6)) < 1e-3


@pytest.mark.parametrize("kw", [dict(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8"),
                                dict(img_size=64, base_channels=32, n_heads=1)])
def test_native_backward_matches_torch_autograd(kw, monkeypatch):
    """The training forward/backward with the hand-written fused 3x3 blocks (train_ops.FusedGNSiLUConv3x3: Winograd /
    direct forward, wgrad, dgrad, GroupNorm+SiLU backward kernels) against the all-torch differentiable path:
    same output, same gradient for every parameter and for the input."""
    from UNet import UNetModel
    from oracle import unet_oracle as uo
    torch.manual_seed(3)
    m = UNetModel(**kw)
    sd = uo.fill_deterministic({k: tuple(v.shape) for k, v in m.state_dict().items()})
    # the reference zero-initialises the last conv of every block; perturb so that every gradient is exercised
    g = torch.Generator().manual_seed(9)
    sd = {k: v + 0.02 * torch.randn(v.shape, generator=g) for k, v in sd.items()}
    m.load_state_dict(sd)
    m.to(DEV).train()
    S = kw["img_size"]
    x = (torch.rand(2, 1, S, S, device=DEV) * 2 - 1).requires_grad_(True)
    t = torch.tensor([17, 640], device=DEV)
    tgt = torch.randn(2, 1, S, S, device=DEV)

    def run(torch_only):
        monkeypatch.setenv("ANODDPM_TORCH_BACKWARD", "1" if torch_only else "0")
        m.zero_grad(set_to_none=True)
        if x.grad is not None:
            x.grad = None
        y = m(x, t)
        loss = ((y - tgt) ** 2).mean()
        loss.backward()
        return y.detach().clone(), loss.item(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, x.grad.detach().clone()

    y_ref, l_ref, g_ref, gx_ref = run(True)
    y_nat, l_nat, g_nat, gx_nat = run(False)
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
    assert rel(y_nat, y_ref) < 1e-4 and abs(l_nat - l_ref) < 1e-4 * abs(l_ref)
    assert rel(gx_nat, gx_ref) < 5e-4, rel(gx_nat, gx_ref)
    # Gradients that are mathematically zero (a per-channel constant in front of a GroupNorm with one channel per
    # group: conv / embedding biases of the 32-channel blocks) are rounding noise of size 1e-9 in both paths, so errors
    # are measured against max(|reference|, 1e-4 * the largest gradient in the model).
    gmax = max(v.abs().max().item() for v in g_ref.values())
    relg = lambda a, b: ((a - b).abs().max() / max(b.abs().max().item(), 1e-4 * gmax)).item()
    worst = max((relg(g_nat[k], g_ref[k]), k) for k in g_ref)
    assert worst[0] < 1e-3, worst
