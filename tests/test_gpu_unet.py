"""-m gpu: full UNetModel forward on the HIP plan against the reference-generated outputs
(tests/golden/unet_*.npz) and, layer by layer, against the stock-PyTorch CPU oracle.
North-star tolerance: 1e-3 relative for fp32 activations (relative to the tensor's max magnitude)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN

DEV = "cuda:0"
REL = 1e-3

CASES = {
    "i32_b32_h1": dict(img_size=32, base_channels=32),
    "i32_b32_h2_a16_8": dict(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8"),
    "i64_b32_hc32": dict(img_size=64, base_channels=32, n_head_channels=32, attention_resolutions="16,8"),
    "i64_b64_c3": dict(img_size=64, base_channels=64, n_heads=2, in_channels=3),
    "i128_b32_h2": dict(img_size=128, base_channels=32, n_heads=2, attention_resolutions="16,8"),
    "c5like_i128_b32": dict(img_size=128, base_channels=32, n_heads=2, channel_mults=(1, 1, 2, 2, 4, 4),
                            attention_resolutions="32,16,8"),
    "c2_256_b128": dict(img_size=256, base_channels=128, n_heads=2, attention_resolutions="16,8"),
    # biggan_updown=False: Downsample / Upsample layers (UNet.py:60-92), with and without their convolutions
    "convrs_i64_b32": dict(img_size=64, base_channels=32, n_heads=2, attention_resolutions="16,8", biggan_updown=False,
                           conv_resample=True),
    "poolrs_i32_b32": dict(img_size=32, base_channels=32, biggan_updown=False, conv_resample=False),
    # BASELINE config 5: 512^2, explicit mults (1,1,2,2,4,4), attention at 32/16/8 (sequence lengths 256 / 1024 / 4096)
    "c5_512_b128": dict(img_size=512, base_channels=128, n_heads=2, channel_mults=(1, 1, 2, 2, 4, 4),
                        attention_resolutions="32,16,8"),
}


def build(name):
    from UNet import UNetModel
    from oracle import unet_oracle as uo
    kw = CASES[name]
    m = UNetModel(**kw)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = uo.fill_deterministic(shapes)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd, kw


@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_reference_output(name):
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    m, sd, kw = build(name)
    x, t = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    with torch.no_grad():
        y = m(x, t)
    assert y.shape == x.shape and y.dtype == x.dtype and y.is_cuda
    ref = torch.from_numpy(g["y"])
    err = ((y.cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < REL, f"{name}: rel err {err:.3e}"
    assert err < 5e-5, f"{name}: fp32 path should be far inside the budget, got {err:.3e}"
    # a second call reuses the plan and returns a fresh tensor
    with torch.no_grad():
        y2 = m(x, t)
    assert torch.equal(y, y2) and y2.data_ptr() != y.data_ptr()


def test_config2_at_its_benchmarked_batch_matches_reference_output():
    """BASELINE config 2 exactly as `bench.py` times it -- 256^2, base 128, attention 16/8, BATCH 4 (the plan picks other kernels
    than at batch 1: 128-channel F(4x4) grids, other split-K factors), four different timesteps -- against the reference model's
    output for the same inputs (tests/golden/unet_c2_256_b128_batch4.npz, make_golden.py:gen_unet_c2_batch4)."""
    g = np.load(os.path.join(GOLDEN, "unet_c2_256_b128_batch4.npz"))
    m, sd, kw = build("c2_256_b128")
    x, t = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    assert x.shape[0] == 4 and len(set(g["t"].tolist())) == 4
    with torch.no_grad():
        y = m(x, t)
    ref = torch.from_numpy(g["y"])
    for b in range(4):                                                  # per image: none may hide behind another's magnitude
        err = ((y[b].cpu() - ref[b]).abs().max() / ref[b].abs().max()).item()
        assert err < 5e-5, f"image {b} (t = {int(g['t'][b])}): rel err {err:.3e}"


@pytest.mark.parametrize("name", ["i64_b32_hc32", "c5like_i128_b32", "c2_256_b128", "c5_512_b128", "convrs_i64_b32", "poolrs_i32_b32"])
def test_layerwise_against_reference_probes(name):
    """Every per-block activation the fixture recorded from the REFERENCE model (forward hooks on each module of
    down / middle / up, tests/golden/make_golden.py:run_unet_case) against the HIP plan's NHWC buffer of that block."""
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    m, sd, kw = build(name)
    x, t = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    with torch.no_grad():
        m(x, t)
    plan = next(iter(m._plans.values()))
    B = x.shape[0]
    probes = [k[len("probe/"):] for k in g.files if k.startswith("probe/")]
    blocks = [k for k in probes if k != "time_embed"]
    assert len(blocks) == len(plan.block_out) and set(blocks) == set(plan.block_out), "every block of the plan is probed"
    te = plan.temb.cpu().flatten()
    ref = g["probe/time_embed"]
    assert np.abs(te[::max(1, te.numel() // 256)][:256].numpy() - ref).max() < 1e-5 * np.abs(ref).max()
    worst = (0.0, "")
    for k in blocks:
        buf, C, Hc = plan.block_out[k]
        shape = tuple(int(v) for v in g["shape/" + k])
        assert shape == (B, C, Hc, Hc), (k, shape, (B, C, Hc, Hc))
        got = buf.view(B, Hc, Hc, C).permute(0, 3, 1, 2).contiguous().cpu().flatten()
        stride = max(1, got.numel() // 256)
        ref = g["probe/" + k]
        mean, absmean, std = g["stat/" + k]
        scale = max(np.abs(ref).max(), absmean)
        err = np.abs(got[::stride][:256].numpy() - ref).max() / scale
        worst = max(worst, (float(err), k))
        # whole-tensor statistics as well: a block that is wrong away from the 256 probe points still moves these
        assert abs(got.double().mean().item() - mean) < 1e-3 * max(absmean, 1e-6), k
        assert abs(got.double().abs().mean().item() - absmean) < 1e-3 * absmean, k
        assert abs(got.double().std().item() - std) < 1e-3 * std, k
    assert worst[0] < 1e-3, worst


def test_time_embedding_matches_oracle():
    from oracle import unet_oracle as uo
    name = "i64_b32_hc32"
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    m, sd, kw = build(name)
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    rec = {}
    uo.forward(sd, x, t, record=rec, **kw)
    with torch.no_grad():
        m(x.to(DEV), t.to(DEV))
    plan = next(iter(m._plans.values()))
    te = plan.temb.cpu()
    assert ((te - rec["time_embed"]).abs().max() / rec["time_embed"].abs().max()) < 1e-5


def test_weight_update_invalidates_packed_weights():
    name = "i32_b32_h1"
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    m, sd, kw = build(name)
    x, t = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    with torch.no_grad():
        y0 = m(x, t)
        m.out["2"].weight.mul_(2.0)
        m.out["2"].bias.mul_(2.0)
        y1 = m(x, t)
    assert torch.allclose(y1, 2 * y0, rtol=1e-5, atol=1e-6)


def test_batch_invariance_and_training_path_agree():
    """Each image is independent (the property multi-GPU sharding relies on), and the differentiable
    training forward computes the same function as the HIP plan."""
    name = "i64_b32_hc32"
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    m, sd, kw = build(name)
    x, t = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    with torch.no_grad():
        y = m(x, t)
        y0 = m(x[:1], t[:1])
        y1 = m(x[1:], t[1:])
    # not bitwise: the batch size may change a layer's kernel (split-K factors; since round 4 the 3x3 layers of <= 256 rows over
    # the batch run the direct no-split-K kernel where larger batches run Winograd F(2x2,3x3)) -- same bound as the config-2 test
    assert ((torch.cat([y0, y1]) - y).abs().max() / y.abs().max()).item() < 2e-5
    yt = m(x, t)                                       # autograd recording -> differentiable path
    assert yt.requires_grad
    assert ((yt.detach() - y).abs().max() / y.abs().max()) < REL
    yt.square().mean().backward()
    assert m.out["2"].weight.grad is not None and torch.isfinite(m.out["2"].weight.grad).all()


@pytest.mark.parametrize("B", [4, 5, 8, 12, 16])
def test_config2_batch_equals_single_image_forwards(B):
    """BASELINE config 2 runs at batch 4 per GPU; the detection loops step the same model at batch 16 / 12 / 8 (one image per chain
    slot, `GaussianDiffusionModel._run_chains`; batch 5 was round 4's five averaged chains); the reference fixture pins batch 1 (and
    batch 4, `test_config2_at_its_benchmarked_batch`).  Images are independent, so the batched forward -- different kernels per
    layer than batch 1: split-K factors, the 64- / 128-channel F(4x4,3x3) workgroups (batch >= 5 takes the 128-channel ones on the
    64x64 maps), the no-split-K small-map kernels (row tiles over the batch) -- must reproduce B batch-1 forwards, the first of
    which is the reference-pinned input.  Timesteps are per slot and spread over [0, 999] as in the slot-batched loop."""
    name = "c2_256_b128"
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    m, sd, kw = build(name)
    gen = torch.Generator().manual_seed(5)
    x = torch.cat([torch.from_numpy(g["x"]), torch.rand(B - 1, 1, 256, 256, generator=gen) * 2 - 1]).to(DEV)
    spread = [int(g["t"][0]), 0, 500, 999, 250] + [int(v) for v in np.linspace(1, 998, 11)]
    t = torch.tensor(spread[:B], device=DEV)
    with torch.no_grad():
        yb = m(x, t)
        y1 = torch.cat([m(x[i:i + 1], t[i:i + 1]) for i in range(B)])
    ref = torch.from_numpy(g["y"])
    assert ((yb[:1].cpu() - ref).abs().max() / ref.abs().max()).item() < 5e-5
    assert ((yb - y1).abs().max() / y1.abs().max()).item() < 2e-5


def test_batches_beyond_sixteen():
    """Round 6: the inference plan stopped at batch 16 (the time-embedding MLP keeps 16 batch rows in registers); the reference has
    no such limit (UNet.py:390-406).  Batch 24 and 40 on a small model == the same images in batches of at most 16."""
    name = "i64_b32_hc32"
    m, sd, kw = build(name)
    gen = torch.Generator().manual_seed(11)
    for B in (24, 40):
        x = (torch.rand(B, 1, 64, 64, generator=gen) * 2 - 1).to(DEV)
        t = torch.randint(0, 1000, (B,), generator=gen).to(DEV)
        with torch.no_grad():
            y = m(x, t)
            parts = torch.cat([m(x[i:i + 16], t[i:i + 16]) for i in range(0, B, 16)])
        assert y.shape == x.shape and torch.isfinite(y).all()
        assert ((y - parts).abs().max() / parts.abs().max()).item() < 2e-5


def test_config2_with_atomic_groupnorm_sums_matches_reference(monkeypatch):
    """ANODDPM_CSUM=1 (round 6, opt-in: measured +-0 on the step): the 128x128 / 64x64 F(4x4,3x3) layers accumulate
    their GroupNorm sums with fp64 atomics and their F(4x4,3x3) consumers finish the GroupNorm in the prologue -- 24 gn_finalize
    launches fewer per forward.  Same reference fixture, same bound as the default plan; and the accumulators are cleared by every
    forward (a second call gives the same output)."""
    monkeypatch.setenv("ANODDPM_CSUM", "1")
    from anoddpm_amd import _lib
    name = "c2_256_b128"
    g = np.load(os.path.join(GOLDEN, f"unet_{name}_batch4.npz"))
    m, sd, kw = build(name)
    x, t = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    with torch.no_grad():
        y = m(x, t)
        y2 = m(x, t)
    plan = next(iter(m._plans.values()))
    assert plan.csum_mode and plan._csum_used > 0
    nfin = sum(1 for code, _ in plan.ops if code == _lib.OP_GN_FINALIZE)
    nfold = sum(1 for code, st in plan.ops if code == _lib.OP_IGEMM and st.cfg == 3 and st.fold_gamma)
    assert nfold >= 20 and nfin <= 35, (nfold, nfin)
    ref = torch.from_numpy(g["y"])
    assert ((y.cpu() - ref).abs().max() / ref.abs().max()).item() < 5e-5
    assert ((y2 - y).abs().max() / y.abs().max()).item() < 1e-6        # (fp64 atomics: equal up to the order of the adds)


def test_packed_weights_are_not_repacked_every_forward():
    """Regression: packed buffers once aliased fp32 parameters, so every forward bumped the parameters' version
    counters and re-packed all weights."""
    name = "i32_b32_h1"
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    m, sd, kw = build(name)
    x, t = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    with torch.no_grad():
        m(x, t)
        plan = next(iter(m._plans.values()))
        tok = plan.token
        versions = [p._version for p in m.parameters()]
        m(x, t)
        m(x, t)
    assert plan.token == tok and versions == [p._version for p in m.parameters()] and plan.pack_count == 1
    ptrs = {p.data_ptr() for p in m.parameters()}
    assert all(st.out not in ptrs for _, st in plan._pack_jobs)
    # a parameter update (optimizer step, load_state_dict) is picked up: exactly one more batched packing launch
    with torch.no_grad():
        y0 = m(x, t).clone()
        for p in m.parameters():
            p.mul_(1.01)
        y1 = m(x, t)
    assert plan.pack_count == 2 and not torch.equal(y0, y1)


def test_input_view_at_an_odd_storage_offset():
    """A contiguous input that starts 4 bytes into its storage (not 16-byte aligned): the stem kernel's 16-byte row loads must
    not see it -- the forward copies such a view; result identical to the aligned tensor's, with and without autograd."""
    name = "i32_b32_h1"
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    m, sd, kw = build(name)
    x, t = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    flat = torch.zeros(x.numel() + 1, device=DEV)
    flat[1:].copy_(x.reshape(-1))
    xo = flat[1:].view_as(x)
    assert xo.is_contiguous() and xo.data_ptr() % 16 == 4
    with torch.no_grad():
        assert torch.equal(m(xo, t), m(x, t))
    m.train()
    ya, yb = m(xo, t), m(x, t)
    assert ya.requires_grad and torch.equal(ya.detach(), yb.detach())
