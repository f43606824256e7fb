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


def test_optimizer_state_dict_is_torch_adamw_layout():
    """FusedAdamWEMA.state_dict() loads into torch.optim.AdamW and vice versa (the reference checkpoints store
    optimiser.state_dict(), diffusion_training.py:173,184, and resume from it, :77): after the exchange both
    optimisers take the same next step."""
    from anoddpm_amd.training import FlatBuffers, FusedAdamWEMA
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(19, 32), torch.nn.SiLU(), torch.nn.Linear(32, 3)).to(DEV)
    flat = FlatBuffers(net)
    opt = FusedAdamWEMA(flat, None, lr=1e-3, weight_decay=0.01, max_norm=None)
    xs = [torch.randn(8, 19, device=DEV) for _ in range(4)]
    for x in xs[:2]:
        flat.zero_grad()
        net(x).square().mean().backward()
        opt.step()
    sd = opt.state_dict()
    ref = copy.deepcopy(net)
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=5.0)            # hyper-parameters come from the loaded state
    opt_ref.load_state_dict(sd)
    assert opt_ref.param_groups[0]["lr"] == 1e-3 and opt_ref.param_groups[0]["weight_decay"] == 0.01
    flat.zero_grad()
    net(xs[2]).square().mean().backward()
    opt.step()
    opt_ref.zero_grad()
    ref(xs[2]).square().mean().backward()
    opt_ref.step()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    # and back: torch's state into a fresh fused optimiser
    net2 = copy.deepcopy(ref)
    flat2 = FlatBuffers(net2)
    opt2 = FusedAdamWEMA(flat2, None, lr=7.0, max_norm=None)
    opt2.load_state_dict(opt_ref.state_dict())
    assert opt2.step_count == 3 and opt2.lr == 1e-3
    flat2.zero_grad()
    net2(xs[3]).square().mean().backward()
    opt2.step()
    opt_ref.zero_grad()
    ref(xs[3]).square().mean().backward()
    opt_ref.step()
    for a, b in zip(net2.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_fused_optimizer_refreshes_owners_without_notify_and_checks_layout():
    import GaussianDiffusion as GD
    from UNet import UNetModel
    from anoddpm_amd.training import FlatBuffers, FusedAdamWEMA, train_step
    torch.manual_seed(2)
    model = UNetModel(32, 32, n_heads=2, attention_resolutions="16,8").to(DEV)
    ema = copy.deepcopy(model)
    flat, flat_ema = FlatBuffers(model), FlatBuffers(ema)
    opt = FusedAdamWEMA(flat, flat_ema, lr=5e-3, ema_decay=0.5)             # no notify=: the owners are found from the flats
    diff = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(1000, "linear"), noise="gauss")
    x = torch.rand(2, 1, 32, 32, device=DEV) * 2 - 1
    t = torch.tensor([5, 300], device=DEV)
    with torch.no_grad():
        y0, e0 = model(x, t).clone(), ema(x, t).clone()                      # plans + packed weights now cached
    train_step(model, diff, x, {"train_start": False}, flat, None, opt)
    with torch.no_grad():
        y1, e1 = model(x, t), ema(x, t)
    assert not torch.equal(y0, y1) and not torch.equal(e0, e1), "stale packed weights after the raw optimiser kernel"
    yg = model(x, t)                                                         # differentiable path reads the parameters directly
    assert ((yg.detach() - y1).abs().max() / y1.abs().max().clamp_min(1e-6)) < 1e-3
    frozen = copy.deepcopy(model)
    next(frozen.parameters()).requires_grad_(False)
    with pytest.raises(ValueError):
        FusedAdamWEMA(flat, FlatBuffers(frozen))


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


@pytest.mark.parametrize("name", ["i64_b64_h2", "i128_b32_hc32", "i128_b128_f43"])
def test_train_step_matches_reference_fixture(name, monkeypatch):
    """Two optimiser steps of the reference's own loop body (diffusion_training.py:99-107, run by the reference's classes
    on CPU: tests/golden/train_*.npz) against `p_loss` -> backward (hand-written kernels) -> FusedAdamWEMA on the device,
    with the same injected draws: loss, every parameter's gradient, the clip norm, parameters and EMA after each step."""
    import os
    import GaussianDiffusion as GD
    from UNet import UNetModel
    from anoddpm_amd.training import FlatBuffers, FusedAdamWEMA
    from conftest import GOLDEN
    from oracle import unet_oracle as uo
    from test_oracle_training import CASES, GPU_ONLY_CASES, check_against_fixture
    g = np.load(os.path.join(GOLDEN, f"train_{name}.npz"))
    kw = {**CASES, **GPU_ONLY_CASES}[name]
    S = kw["img_size"]
    model = UNetModel(**kw)
    keys = [str(k) for k in g["keys"]]
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert keys == list(shapes)
    model.load_state_dict(uo.perturb(uo.fill_deterministic(shapes)))
    model.to(DEV).train()
    ema = copy.deepcopy(model)
    flat, flat_ema = FlatBuffers(model), FlatBuffers(ema)
    lr = float(g["lr"])
    opt = FusedAdamWEMA(flat, flat_ema, lr=lr, weight_decay=float(g["weight_decay"]), notify=(model, ema))
    diff = GD.GaussianDiffusionModel([S, S], GD.get_beta_schedule(1000, "linear"), loss_type="l2", noise="gauss")
    args = {"train_start": True, "sample_distance": 800, "Batch_Size": int(g["s0/x0"].shape[0])}
    real_randint = torch.randint
    nsteps = sum(f"s{i}/x0" in g.files for i in range(4))
    for step in range(nsteps):
        x0, noise, t = (torch.from_numpy(g[f"s{step}/{n}"]).to(DEV) for n in ("x0", "noise", "t"))
        diff.noise_fn = lambda a, b, _n=noise: _n
        monkeypatch.setattr(torch, "randint", lambda *a, **k: t.clone())
        loss, (ld, x_t, eps) = diff.p_loss(model, x0, args)
        monkeypatch.setattr(torch, "randint", real_randint)
        flat.zero_grad()
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        norm = opt.step()
        from test_oracle_training import probe
        assert np.abs(probe(x_t, 1024) - g[f"s{step}/x_t"]).max() == 0           # sample_q is bit-exact
        assert np.abs(probe(eps, 1024) - g[f"s{step}/eps"]).max() < 1e-3 * np.abs(g[f"s{step}/eps"]).max()
        check_against_fixture(g, step, keys, loss.item(), grads, norm.item(), dict(model.named_parameters()),
                              dict(ema.named_parameters()), lr)
    if name == "i128_b128_f43":
        # this fixture exists to pin the Winograd F(4x4,3x3) kernels to a REFERENCE-run gradient: make sure the plan used them
        # (forward + data gradient on both kernel variants, the Winograd-domain weight gradient)
        from anoddpm_amd import _lib
        plan = next(iter(model._tplans.values()))
        f43 = [st for code, st in plan.ops + plan.bops if code == _lib.OP_IGEMM and st.cfg == 3]
        assert any(st.H == 128 and st.N == 128 for st in f43) and any(st.H == 64 for st in f43)
        assert sum(1 for code, st in plan.bops if code == _lib.OP_IGEMM and st.cfg == 3) >= 4
        assert any(st.algo == 1 for code, st in plan.bops if code == _lib.OP_WGRAD3)


@pytest.mark.parametrize("kw,B", [(dict(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8"), 2),
                                  (dict(img_size=64, base_channels=32, n_heads=1), 2),
                                  # BASELINE config 3's per-GPU share: 256^2, base 128, attention at 16/8, batch 4
                                  (dict(img_size=256, base_channels=128, n_heads=2, attention_resolutions="16,8"), 4)])
def test_native_backward_matches_torch_autograd(kw, B):
    """The native training plan (train_plan.py: Winograd / direct forward, wgrad, dgrad, GroupNorm+SiLU backward kernels)
    against the differentiable PyTorch-ROCm restatement of the same forward (oracle/unet_oracle.forward_autograd, test
    infrastructure, run on the device): same output, same gradient for every parameter and for the input."""
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
    x = (torch.rand(B, 1, S, S, device=DEV) * 2 - 1).requires_grad_(True)
    t = torch.tensor([17, 640, 3, 999][:B], device=DEV)
    tgt = torch.randn(B, 1, S, S, device=DEV)

    # reference: stock differentiable ops on the same weights
    leaves = {k: v.to(DEV).clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.detach().clone().requires_grad_(True)
    y_ref = uo.forward_autograd(leaves, xr, t, **kw)
    l_ref = ((y_ref - tgt) ** 2).mean()
    l_ref.backward()
    g_ref, gx_ref = {k: v.grad.detach() for k, v in leaves.items()}, xr.grad.detach()
    # native
    y = m(x, t)
    assert len(m._tplans) == 1
    loss = ((y - tgt) ** 2).mean()
    loss.backward()
    y_nat, l_nat, g_nat, gx_nat = y.detach(), loss.item(), {k: p.grad.detach() for k, p in m.named_parameters()}, x.grad.detach()
    y_ref, l_ref = y_ref.detach(), l_ref.item()
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


def test_dropout_trains_natively_with_the_masks_it_reports():
    """dropout > 0 in training mode (UNet.py:192) runs on the native plan: the hash-generated masks are read back from the plan's
    dropped-activation buffers and injected into the differentiable restatement -- same output, same gradients; a second
    forward draws different masks; eval mode drops nothing."""
    from UNet import UNetModel
    from oracle import unet_oracle as uo
    kw = dict(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8")
    p_drop = 0.3
    torch.manual_seed(4)
    m = UNetModel(dropout=p_drop, **kw)
    sd = uo.perturb(uo.fill_deterministic({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    m.load_state_dict(sd)
    m.to(DEV).train()
    B = 2
    x = torch.rand(B, 1, 32, 32, device=DEV) * 2 - 1
    t = torch.tensor([40, 911], device=DEV)
    tgt = torch.randn(B, 1, 32, 32, device=DEV)
    y = m(x, t)
    plan = next(iter(m._tplans.values()))
    res_prefixes = [b[0] for grp in (m._blocks[0], [m._blocks[1]], m._blocks[2]) for blk in grp for b in blk if b[1] == "res"]
    assert plan.p_drop == p_drop and len(plan._drop_ops) == len(res_prefixes)  # one per ResBlock, forward order
    # keep masks of this forward, per block prefix: the dropped activation is exactly zero where dropped (NHWC buffers)
    masks, kept = {}, []
    for prefix, (fwd, _, _) in zip(res_prefixes, plan._drop_ops):
        C = fwd.C
        n = fwd.n
        buf = next(tn for tn in plan.keep if torch.is_tensor(tn) and tn.data_ptr() == fwd.out)
        a2 = buf.reshape(B, n // C, C)
        Hh = int(round((n // C) ** 0.5))
        masks[prefix] = (a2 != 0).reshape(B, Hh, Hh, C).permute(0, 3, 1, 2).float()
        kept.append(masks[prefix].mean().item())
    assert abs(np.mean(kept) - (1 - p_drop)) < 0.02, np.mean(kept)
    loss = ((y - tgt) ** 2).mean()
    loss.backward()
    leaves = {k: v.to(DEV).clone().requires_grad_(True) for k, v in sd.items()}
    y_ref = uo.forward_autograd(leaves, x, t, dropout=lambda p, h: h * masks[p] / (1 - p_drop), **kw)
    ((y_ref - tgt) ** 2).mean().backward()
    assert ((y.detach() - y_ref.detach()).abs().max() / y_ref.detach().abs().max()).item() < 1e-4
    gmax = max(v.grad.abs().max().item() for v in leaves.values())
    worst = max((((p.grad - leaves[k].grad).abs().max() / max(leaves[k].grad.abs().max().item(), 1e-4 * gmax)).item(), k)
                for k, p in m.named_parameters())
    assert worst[0] < 1e-3, worst
    y2 = m(x, t)                                                             # another forward: other masks
    assert not torch.equal(y2.detach(), y.detach())
    m.eval()
    with torch.no_grad():
        y_eval = m(x, t)
    y_plain = uo.forward(leaves, x, t, **kw)
    assert ((y_eval - y_plain).abs().max() / y_plain.abs().max()).item() < 1e-4


def test_frozen_parameters_get_no_gradient():
    """Parameters with requires_grad=False (fine-tuning a sub-network) take the native path too: no .grad appears on them."""
    from UNet import UNetModel
    m = UNetModel(32, 32, n_heads=2, attention_resolutions="16,8").to(DEV).train()
    frozen = [p for k, p in m.named_parameters() if k.startswith("down.")]
    for p in frozen:
        p.requires_grad_(False)
    x = torch.rand(2, 1, 32, 32, device=DEV)
    m(x, torch.tensor([5, 6], device=DEV)).square().mean().backward()
    assert all(p.grad is None for p in frozen)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)


@pytest.mark.parametrize("conv", [True, False])
def test_conv_resample_variant_gradients_match_cpu_oracle(conv):
    """biggan_updown=False (Downsample / Upsample layers, UNet.py:60-92) on the native training plan: the stride-2 convolution's
    backward goes through the zero-stuffed gradient, the nearest-x2 + convolution through the fused operand load.  Output, loss
    and every gradient against the CPU oracle's autograd on the same weights."""
    from UNet import UNetModel
    from oracle import unet_oracle as uo
    from anoddpm_amd import train_plan
    kw = dict(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8", biggan_updown=False, conv_resample=conv)
    m = UNetModel(**kw)
    assert train_plan.eligible(m, 2, 32)
    assert any(".downsample." in k for k in m.state_dict()) == conv and any(k.endswith(".conv.weight") for k in m.state_dict()) == conv
    sd = uo.perturb(uo.fill_deterministic({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    m.load_state_dict(sd)
    m.to(DEV).train()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 1, 32, 32, generator=g) * 2 - 1
    tgt = torch.randn(2, 1, 32, 32, generator=g)
    t = torch.tensor([40, 911])
    y = m(x.to(DEV), t.to(DEV))
    assert len(m._tplans) == 1                                               # the native plan, not a PyTorch expression
    loss = ((y - tgt.to(DEV)) ** 2).mean()
    loss.backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y_ref = uo.forward_autograd(leaves, x, t, **kw)
    l_ref = ((y_ref - tgt) ** 2).mean()
    l_ref.backward()
    assert ((y.detach().cpu() - y_ref.detach()).abs().max() / y_ref.detach().abs().max()).item() < 1e-4
    assert abs(loss.item() - l_ref.item()) < 1e-4 * abs(l_ref.item())
    gmax = max(v.grad.abs().max().item() for v in leaves.values())
    worst = max((((p.grad.cpu() - leaves[k].grad).abs().max() / max(leaves[k].grad.abs().max().item(), 1e-4 * gmax)).item(), k)
                for k, p in m.named_parameters())
    assert worst[0] < 1e-3, worst


def test_rccl_world_size_1_reducer_with_native_backward():
    """The data-parallel machinery on the real backend: `nccl` (= RCCL) process group of one rank, GradAllReducer on a
    UNetModel whose backward runs the hand-written kernels (custom autograd Functions + post-accumulate hooks + flat
    gradient views).  Every bucket must be launched from a hook and the step must equal the reducer-free step."""
    import os
    import torch.distributed as dist
    import GaussianDiffusion as GD
    from UNet import UNetModel
    from anoddpm_amd.training import FlatBuffers, FusedAdamWEMA, GradAllReducer, train_step
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        torch.manual_seed(5)
        base = UNetModel(32, 32, n_heads=2, attention_resolutions="16,8").to(DEV)
        diff = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(1000, "linear"), noise="gauss")
        x = torch.rand(2, 1, 32, 32, device=DEV) * 2 - 1
        noise = torch.randn(2, 1, 32, 32, device=DEV)
        diff.noise_fn = lambda a, b: noise
        results = []
        for use_reducer in (False, True):
            model = copy.deepcopy(base)
            flat = FlatBuffers(model)
            opt = FusedAdamWEMA(flat, None, lr=1e-3)
            red = GradAllReducer(flat, bucket_bytes=1 << 20, force=True) if use_reducer else None
            if red is not None:
                assert len(red.buckets) > 3 and len(red.hooks) == len(flat.params)
            torch.manual_seed(77)                                  # same t draw inside p_loss
            loss, _ = train_step(model, diff, x, {"train_start": False}, flat, red, opt)
            if red is not None:
                assert red.launched == len(red.buckets) and all(p == len(m) for p, (_, _, m) in zip(red.pending, red.buckets))
                # the native backward is cut at the bucket boundaries: buckets go out while later ops are still to be
                # enqueued (overlap), in backward order, and none is left for finish() to launch
                log = red.last_launch_log
                assert sorted(b for b, _ in log) == list(range(len(red.buckets)))
                done = [d for _, d in log]
                assert all(d is not None for d in done) and done == sorted(done)
                plan = next(iter(model._tplans.values()))
                assert done[0] < len(plan.bops) // 2 and len(set(done)) > 3
            results.append((loss.item(), flat.flat_param.clone(), flat.flat_grad.clone()))
        assert results[0][0] == results[1][0]
        assert torch.equal(results[0][2], results[1][2]) and torch.equal(results[0][1], results[1][1])
    finally:
        dist.destroy_process_group()


def test_frozen_parameters_under_an_active_reducer():
    """data-parallel fine-tuning with requires_grad=False parameters.  Frozen parameters are not in the flat buffers and
    never get a .grad; with an active GradAllReducer (RCCL process group of one rank, force=True) the step must run -- every bucket
    launched from the cut backward -- and equal the reducer-free step; frozen parameters stay untouched."""
    import os
    import torch.distributed as dist
    import GaussianDiffusion as GD
    from UNet import UNetModel
    from anoddpm_amd.training import FlatBuffers, FusedAdamWEMA, GradAllReducer, train_step
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29537")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        torch.manual_seed(5)
        base = UNetModel(32, 32, n_heads=2, attention_resolutions="16,8").to(DEV)
        frozen = [k for k, _ in base.named_parameters() if k.startswith("down.1.") or k.startswith("middle.1.")]
        assert len(frozen) > 8
        diff = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(1000, "linear"), noise="gauss")
        x = torch.rand(2, 1, 32, 32, device=DEV) * 2 - 1
        noise = torch.randn(2, 1, 32, 32, device=DEV)
        diff.noise_fn = lambda a, b: noise
        results = []
        for use_reducer in (False, True):
            model = copy.deepcopy(base)
            named = dict(model.named_parameters())
            for k in frozen:
                named[k].requires_grad_(False)
            before = {k: named[k].detach().clone() for k in frozen}
            flat = FlatBuffers(model)
            assert not any(k in flat.names for k in frozen)
            opt = FusedAdamWEMA(flat, None, lr=1e-3)
            red = GradAllReducer(flat, bucket_bytes=1 << 20, force=True) if use_reducer else None
            for _ in range(2):                                     # twice: the second step reuses the plan and its cut schedule
                torch.manual_seed(77)
                loss, _ = train_step(model, diff, x, {"train_start": False}, flat, red, opt)
            if red is not None:
                assert red.launched == len(red.buckets) and all(d is not None for _, d in red.last_launch_log)
            assert all(named[k].grad is None and torch.equal(named[k], before[k]) for k in frozen)
            results.append((loss.item(), flat.flat_param.clone(), flat.flat_grad.clone()))
        assert results[0][0] == results[1][0]
        assert torch.equal(results[0][2], results[1][2]) and torch.equal(results[0][1], results[1][1])
    finally:
        dist.destroy_process_group()


def test_train_mode_dropout_is_active_under_no_grad_and_absent_in_eval():
    """nn.Dropout acts in train() mode whatever the grad mode (UNet.py:192) -- the sampling previews of the training
    script run the train-mode model under no_grad.  Two no-grad forwards in train mode differ (fresh masks from torch's generator,
    reproducible under manual_seed), eval mode is deterministic and equals the inference plan."""
    from UNet import UNetModel
    import GaussianDiffusion as GD
    torch.manual_seed(3)
    m = UNetModel(32, 32, n_heads=2, attention_resolutions="16,8", dropout=0.25).to(DEV)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith("out_layers.3.weight") or k.startswith("out.2."):
                p.normal_(0, 0.05)                                  # the reference zero-initialises these: make dropout visible
    x = torch.rand(2, 1, 32, 32, device=DEV) * 2 - 1
    t = torch.tensor([10, 500], device=DEV)
    m.train()
    with torch.no_grad():
        torch.manual_seed(1)
        a = m(x, t)
        b = m(x, t)
        torch.manual_seed(1)
        a2 = m(x, t)
    assert not torch.equal(a, b) and torch.equal(a, a2)
    m.eval()
    with torch.no_grad():
        e1, e2 = m(x, t), m(x, t)
    assert torch.equal(e1, e2) and not torch.equal(e1, a)
    # the reverse chain takes the eager path for such a model (no captured graph with frozen masks)
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(100, "linear"), noise="gauss")
    m.train()
    ch = d.reverse_chain(m, x, 3, "gauss")
    assert not ch.hip_model and not ch.use_graph
    ch.step()
    m.eval()
    assert d.reverse_chain(m, x, 3, "gauss").hip_model
