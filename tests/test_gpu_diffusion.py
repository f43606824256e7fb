"""-m gpu: fused diffusion kernels + GaussianDiffusionModel semantics against reference-generated
fixtures (bit-exact fp32) and the numpy/torch oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN

DEV = "cuda:0"


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(GOLDEN, "diffusion_kat.npz"))


def T(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


@pytest.mark.parametrize("name", ["linear", "cosine"])
def test_sample_q_and_sample_p_bit_exact(kat, name):
    import GaussianDiffusion as GD
    d = GD.GaussianDiffusionModel([16, 16], GD.get_beta_schedule(1000, name))
    x, eps, noise, t = T(kat["x"]), T(kat["eps"]), T(kat["noise"]), T(kat["t"])
    assert torch.equal(d.sample_q(x, t, noise).cpu(), torch.from_numpy(kat[f"{name}_sample_q"]))
    assert torch.equal(d.q_sample(x, t, noise).cpu(), torch.from_numpy(kat[f"{name}_sample_q"]))
    assert torch.equal(d.sample_q_gradual(x, t, noise).cpu(), torch.from_numpy(kat[f"{name}_sample_q_gradual"]))
    pmv = d.p_mean_variance(None, x, t, estimate_noise=eps)
    for k in ("mean", "variance", "log_variance", "pred_x_0"):
        assert pmv[k].shape == x.shape
        assert torch.equal(pmv[k].contiguous().cpu(), torch.from_numpy(kat[f"{name}_pmv_{k}"])), k
    sp = d.sample_p(lambda a, b: eps, x, t, denoise_fn=lambda a, b: noise)
    assert torch.equal(sp["sample"].cpu(), torch.from_numpy(kat[f"{name}_sample_p_sample"]))
    assert torch.equal(sp["pred_x_0"].cpu(), torch.from_numpy(kat[f"{name}_sample_p_pred_x_0"]))
    pe = d.predict_eps_from_x_0(x, t, pmv["pred_x_0"])
    np.testing.assert_allclose(pe.cpu().numpy(), kat[f"{name}_predict_eps_from_x_0"], rtol=1e-5, atol=1e-5)


def test_ragged_and_empty_shapes_vs_oracle():
    import GaussianDiffusion as GD
    from oracle import diffusion_oracle as do
    betas = GD.get_beta_schedule(1000, "linear")
    d = GD.GaussianDiffusionModel([7, 5], betas)
    tb = do.tables(betas)
    g = torch.Generator().manual_seed(3)
    for shape in ((3, 1, 7, 5), (2, 3, 9, 9), (1, 1, 256, 256)):      # n % 4 != 0 -> scalar path
        x = torch.rand(shape, generator=g) * 2 - 1
        e, n = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
        t = torch.randint(0, 1000, (shape[0],), generator=g)
        assert torch.equal(d.sample_q(x.to(DEV), t.to(DEV), n.to(DEV)).cpu(), do.q_sample(tb, x, t, n))
        s, p0 = do.p_sample_update(tb, x, t, e, n)
        sp = d.sample_p(lambda a, b: e.to(DEV), x.to(DEV), t.to(DEV), denoise_fn=lambda a, b: n.to(DEV))
        assert torch.equal(sp["sample"].cpu(), s) and torch.equal(sp["pred_x_0"].cpu(), p0)
    empty = torch.zeros(0, 1, 4, 4, device=DEV)
    assert d.sample_q(empty, torch.zeros(0, dtype=torch.int64, device=DEV), empty).shape == (0, 1, 4, 4)


def test_forward_backward_sequences(kat):
    import GaussianDiffusion as GD
    d = GD.GaussianDiffusionModel([16, 16], GD.get_beta_schedule(1000, "linear"))
    x1 = T(kat["x"][:1])
    fixed = T(kat["noise"][:1])
    d.noise_fn = lambda a, b: fixed
    model = lambda a, b: 0.3 * a - 0.1
    half = d.forward_backward(model, x1, "half", 5, denoise_fn=lambda a, b: 0.5 * fixed)
    whole = d.forward_backward(model, x1, "whole", 4, denoise_fn=lambda a, b: 0.5 * fixed)
    final = d.forward_backward(model, x1, None, 5, denoise_fn=lambda a, b: 0.5 * fixed)
    assert len(half) == int(kat["fb_half_len"]) == 7 and len(whole) == int(kat["fb_whole_len"]) == 9
    assert all(not s.is_cuda for s in half)                       # sequences are CPU tensors upstream
    np.testing.assert_allclose(torch.stack(half).numpy(), kat["fb_half_seq"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(torch.stack(whole).numpy(), kat["fb_whole_seq"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(final.cpu().numpy(), kat["fb_final"], rtol=0, atol=2e-6)
    assert final.is_cuda and torch.equal(d.forward_backward(model, x1, None, 0), x1)
    assert torch.equal(d.p_sample_loop(model, x1, None, 0), x1)


def test_losses_with_injected_noise(kat):
    import GaussianDiffusion as GD
    x, noise, t = T(kat["x"]), T(kat["noise"]), T(kat["t"])
    model = lambda a, b: 0.3 * a - 0.1
    for lt in ("l1", "l2", "hybrid"):
        d = GD.GaussianDiffusionModel([16, 16], GD.get_beta_schedule(1000, "linear"), loss_type=lt)
        d.noise_fn = lambda a, b: noise
        loss, x_t, est = d.calc_loss(model, x, t)
        np.testing.assert_allclose(loss["loss"].cpu().numpy(), kat[f"loss_{lt}"], rtol=2e-5, atol=1e-6)
        if lt == "hybrid":
            np.testing.assert_allclose(loss["vlb"].cpu().numpy(), kat["loss_hybrid_vlb"], rtol=2e-5, atol=1e-6)
    args = {"train_start": True, "sample_distance": 800}
    total, (ld, x_t, eps_t) = d.p_loss(model, x, args)
    assert total.dim() == 0 and x_t.shape == x.shape and eps_t.shape == x.shape


def test_simplex_noise_path_matches_oracle_and_rng_order():
    import GaussianDiffusion as GD
    from oracle.simplex_oracle import OracleSimplex
    d = GD.GaussianDiffusionModel([48, 40], GD.get_beta_schedule(1000, "linear"), noise="simplex", img_channels=2)
    x = torch.zeros(1, 2, 48, 40, device=DEV)
    t = torch.tensor([249], device=DEV)
    np.random.seed(99)
    got = d.noise_fn(x, t)
    o = OracleSimplex(3)                                          # explicit seed: the constructor draws nothing
    np.random.seed(99)
    for c in range(2):
        o.newSeed()                                               # one fresh seed per channel (:101-102)
        ref = o.rand_3d_fixed_T_octaves((48, 40), np.array([249]), 6, 0.8, 64)[0].astype(np.float32)
        assert (got[0, c].cpu().numpy().view(np.uint32) == ref.view(np.uint32)).all()
    # sample_p with a simplex string: default 6 octaves / 0.8 / 64 (GaussianDiffusion.py:97,310)
    np.random.seed(5)
    eps = torch.zeros_like(x)
    out = d.sample_p(lambda a, b: eps, x, t, denoise_fn="simplex")
    o2 = OracleSimplex(3)
    np.random.seed(5)
    o2.newSeed()
    n0 = o2.rand_3d_fixed_T_octaves((48, 40), np.array([249]), 6, 0.8, 64)[0].astype(np.float32)
    from oracle import diffusion_oracle as do
    tb = do.tables(GD.get_beta_schedule(1000, "linear"))
    o2.newSeed()
    n1 = o2.rand_3d_fixed_T_octaves((48, 40), np.array([249]), 6, 0.8, 64)[0].astype(np.float32)
    nz = torch.from_numpy(np.stack([n0, n1])[None])
    ref, _ = do.p_sample_update(tb, x.cpu(), t.cpu(), eps.cpu(), nz)
    assert torch.equal(out["sample"].cpu(), ref)


def test_graph_replay_equals_eager_chain():
    """The HIP-graph replayed reverse chain must produce exactly what eager launches produce."""
    import GaussianDiffusion as GD
    from UNet import UNetModel
    from oracle import unet_oracle as uo
    kw = dict(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8")
    m = UNetModel(**kw)
    m.load_state_dict(uo.fill_deterministic({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    m.to(DEV).eval()
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(1000, "linear"), noise="simplex")
    x = torch.rand(2, 1, 32, 32, device=DEV) * 2 - 1
    outs = []
    for use_graph in (False, True):
        np.random.seed(11)
        ch = GD.ReverseChain(d, m, x, 9, GD.SimplexNoiseFn(d.simplex, octave=4), use_graph=use_graph)
        for _ in range(9):
            ch.step()
        ch.finish()
        outs.append(ch.x.clone())
        assert (ch.t == -1).all() and int(ch.step_idx) == 9
    assert torch.equal(outs[0], outs[1])
    np.random.seed(11)
    full = d.forward_backward(m, x, None, 9, denoise_fn=GD.SimplexNoiseFn(d.simplex, octave=4))
    assert full.shape == x.shape and torch.isfinite(full).all()


def test_kept_chain_is_not_replayed_for_a_train_mode_dropout_model():
    """a chain captured under model.eval() replays the dropout-free inference graph; once the SAME model is in
    train() mode with dropout > 0 (UNet.py:192 draws a fresh mask per forward) forward_backward must not reuse it, and an explicit
    use_graph=True is refused instead of replaying one baked-in mask."""
    import GaussianDiffusion as GD
    from UNet import UNetModel
    from oracle import unet_oracle as uo
    m = UNetModel(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8", dropout=0.5)
    m.load_state_dict(uo.fill_deterministic({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    m.to(DEV).eval()
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(100, "linear"), noise="gauss")
    x = torch.rand(2, 1, 32, 32, device=DEV) * 2 - 1
    torch.manual_seed(1)
    a = d.forward_backward(m, x, None, 4)
    assert len(d._chains) == 1
    kept = next(iter(d._chains.values()))
    assert kept.use_graph and kept.hip_model
    torch.manual_seed(1)
    assert torch.equal(d.forward_backward(m, x, None, 4), a)          # eval: the kept chain is restarted, same draws -> same result
    m.train()
    torch.manual_seed(1)
    b = d.forward_backward(m, x, None, 4)                             # train + dropout: a fresh eager chain through model.forward
    assert torch.isfinite(b).all() and not torch.equal(a, b)          # masks were drawn
    assert len(d._chains) == 1 and next(iter(d._chains.values())) is kept      # the eager chain is not kept
    with pytest.raises(ValueError, match="dropout-free"):
        GD.ReverseChain(d, m, x, 4, "gauss", use_graph=True)
    m.eval()
    torch.manual_seed(1)
    assert torch.equal(d.forward_backward(m, x, None, 4), a)          # back in eval: the kept chain again


def test_no_garbage_collection_inside_graph_capture(monkeypatch):
    """Round 6 regression: torch.cuda.graph() no longer collects garbage before a capture (torch.compiler.config.force_cudagraph_gc
    is off by default), so the cyclic collector could run INSIDE ReverseChain's stream capture and destroy an older, unreachable
    chain's captured graph / plan there -- HIP calls that are illegal during a global-mode capture (`Fatal Python error: Aborted`
    with the interpreter "Garbage-collecting" under _step_body: two of ten full GPU test runs).  ReverseChain.step now collects
    before the capture and keeps the collector off until it has ended: with the collector at its most eager (threshold 1) and
    unreachable chains lying around, no collection may START between capture begin and end."""
    import gc
    import GaussianDiffusion as GD
    from UNet import UNetModel
    from oracle import unet_oracle as uo

    def model():
        m = UNetModel(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8")
        m.load_state_dict(uo.fill_deterministic({k: tuple(v.shape) for k, v in m.state_dict().items()}))
        return m.to(DEV).eval()

    state = {"in_capture": False, "captures": 0, "hits": []}
    enter, leave = torch.cuda.graph.__enter__, torch.cuda.graph.__exit__

    def _enter(self):
        r = enter(self)
        state["in_capture"] = True
        state["captures"] += 1
        return r

    def _exit(self, *a):
        state["in_capture"] = False
        return leave(self, *a)
    monkeypatch.setattr(torch.cuda.graph, "__enter__", _enter)
    monkeypatch.setattr(torch.cuda.graph, "__exit__", _exit)

    def on_gc(phase, info):
        if phase == "start" and state["in_capture"]:
            state["hits"].append(info["generation"])
    x = torch.rand(2, 1, 32, 32, device=DEV) * 2 - 1
    old = gc.get_threshold()
    gc.callbacks.append(on_gc)
    try:
        for _ in range(2):                                     # chains with captured graphs, left unreachable in reference cycles
            m = model()
            d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(100, "linear"), noise="gauss")
            d.forward_backward(m, x, None, 4)
            d._self, m._d = d, d
        del m, d
        m2 = model()
        d2 = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(100, "linear"), noise="gauss")
        gc.set_threshold(1, 1, 1)                              # from here on every allocation may start a collection
        torch.manual_seed(3)
        a = d2.forward_backward(m2, x, None, 5)                # eager step, capture, three replays
        gc.set_threshold(*old)
        torch.manual_seed(3)
        b = d2.forward_backward(m2, x, None, 5)
        assert state["captures"] >= 3 and state["hits"] == [], state
        assert torch.isfinite(a).all() and torch.equal(a, b)
    finally:
        gc.callbacks.remove(on_gc)
        gc.set_threshold(*old)
        gc.enable()
        gc.collect()
