"""-m gpu: the detection compute loops (GaussianDiffusion.py:480-594) with their `total_avg` chains batched."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def tiny():
    import GaussianDiffusion as GD
    from UNet import UNetModel
    from oracle import unet_oracle as uo
    m = UNetModel(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8")
    m.load_state_dict(uo.fill_deterministic({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    m.to(DEV).eval()
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(100, "linear"), noise="gauss")
    return GD, m, d


def test_batched_chains_equal_serial_chains():
    """Stacking the `avg` chains as a batch gives each chain exactly what running it alone gives, when both see
    the same noise (the only difference left is the kernels' batch-dependent tiling: 1e-4)."""
    GD, m, d = tiny()
    torch.manual_seed(3)
    x_0 = torch.rand(1, 1, 32, 32, device=DEV) * 2 - 1
    navg, tdist = 3, 7
    fwd = [torch.randn(1, 1, 32, 32, device=DEV) for _ in range(navg)]
    it = iter(fwd)
    d.noise_fn = lambda x, t: next(it)
    torch.manual_seed(17)
    batched = d._avg_chains(m, x_0, tdist, navg)
    assert batched.shape == (navg, 1, 32, 32) and torch.isfinite(batched).all()
    # serial: one chain at a time, fed the slices of the same per-step normal draws
    torch.manual_seed(17)
    steps = [torch.randn(navg, 1, 32, 32, device=DEV) for _ in range(tdist)]
    for b in range(navg):
        t_tensor = torch.full((1,), tdist, device=DEV, dtype=torch.int64)
        x = d.sample_q(x_0, t_tensor, fwd[b])
        for i, t in enumerate(range(tdist - 1, -1, -1)):
            tb = torch.full((1,), t, device=DEV, dtype=torch.int64)
            with torch.no_grad():
                x = d.sample_p(m, x, tb, denoise_fn=lambda xx, tt, i=i, b=b: steps[i][b:b + 1])["sample"]
        assert torch.allclose(batched[b:b + 1], x, atol=1e-4, rtol=0), float((batched[b:b + 1] - x).abs().max())


def test_mixed_length_chains_on_slots_equal_serial_chains():
    """All (t_distance, avg) chains of one image in ONE batched loop: chains of different lengths share `slots`
    chain slots, a finished slot starts the next pending chain.  Every chain equals the same chain run alone with the draws its
    slot saw (global step k of the batched loop, row `slot`)."""
    GD, m, d = tiny()
    torch.manual_seed(3)
    x_0 = torch.rand(1, 1, 32, 32, device=DEV) * 2 - 1
    dists = [9, 9, 6, 6, 3, 3, 1, 0, 12]                       # incl. a chain without reverse steps and one longer than the rest
    n, G = len(dists), 4
    fwd = torch.randn(n, 1, 32, 32, device=DEV)
    torch.manual_seed(17)
    out = d._run_chains(m, x_0, dists, fwd, slots=G)
    sched = d.last_chain_schedule
    assert out.shape == (n, 1, 32, 32) and torch.isfinite(out).all()
    assert sched["slots"] == G and sched["chain_steps"] == sum(dists) and sched["steps"] >= -(-sum(dists) // G)
    torch.manual_seed(17)
    draws = [torch.randn(G, 1, 32, 32, device=DEV) for _ in range(sched["steps"])]
    for c, dist in enumerate(dists):
        x = d.sample_q(x_0, torch.full((1,), dist, device=DEV, dtype=torch.int64), fwd[c:c + 1])
        if dist == 0:
            assert torch.equal(out[c:c + 1], x)
            continue
        slot, start = sched["place"][c]
        for i, t in enumerate(range(dist - 1, -1, -1)):
            tb = torch.full((1,), t, device=DEV, dtype=torch.int64)
            with torch.no_grad():
                x = d.sample_p(m, x, tb, denoise_fn=lambda xx, tt, k=start + i, s=slot: draws[k][s:s + 1])["sample"]
        assert torch.allclose(out[c:c + 1], x, atol=1e-4, rtol=0), (c, float((out[c:c + 1] - x).abs().max()))
    # a second sweep on the kept chain (graph replay from the first step on), other slot count chosen by the cost rule
    out2 = d._run_chains(m, x_0, [5, 4, 3], fwd[:3])
    assert out2.shape == (3, 1, 32, 32) and torch.isfinite(out2).all() and d.last_chain_schedule["slots"] == 3
    with pytest.raises(IndexError):
        d._run_chains(m, x_0, [100], fwd[:1])                     # sample_q at t = T: upstream's extract() raises too
    assert d._run_chains(m, x_0, [], None).shape == (0, 1, 32, 32)


def test_reverse_chain_is_reused_across_settings_and_matches_a_fresh_chain():
    """The detection loops run many chains of one batch shape: after the first, a chain is restarted on the same device buffers
    and the same captured graph (ReverseChain.reset).  Restarted chains give bit-identical results to freshly built ones -- other
    input, other length, re-drawn simplex seeds, and weights that changed in between."""
    GD, m, _ = tiny()
    mk = lambda: GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(100, "linear"), noise="simplex")
    d = mk()
    fn = GD.SimplexNoiseFn(d.simplex, octave=4, persistence=0.6, frequency=16)
    torch.manual_seed(5)
    xa, xb = torch.rand(2, 1, 32, 32, device=DEV) * 2 - 1, torch.rand(2, 1, 32, 32, device=DEV) * 2 - 1

    def fresh(x, tdist, seed):
        d2 = mk()                                                    # no cached chain
        np.random.seed(seed)
        with torch.no_grad():
            return d2._reverse_chain(m, x, tdist, GD.SimplexNoiseFn(d2.simplex, octave=4, persistence=0.6, frequency=16), None).clone()

    np.random.seed(3)
    with torch.no_grad():
        a = d._reverse_chain(m, xa, 7, fn, None).clone()
    assert len(d._chains) == 1
    chain = next(iter(d._chains.values()))
    assert chain.use_graph and chain.graph is not None
    np.random.seed(4)
    with torch.no_grad():
        b = d._reverse_chain(m, xb, 5, fn, None).clone()             # restarted: other input, shorter
    assert len(d._chains) == 1 and next(iter(d._chains.values())) is chain
    assert torch.equal(a, fresh(xa, 7, 3)) and torch.equal(b, fresh(xb, 5, 4))
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.01)
    m.mark_weights_changed()
    np.random.seed(6)
    with torch.no_grad():
        c = d._reverse_chain(m, xa, 6, fn, None).clone()             # restarted after the weights moved
    assert torch.equal(c, fresh(xa, 6, 6)) and not torch.equal(c[:, :, :4], a[:, :, :4])
    # another noise source or batch shape gets its own chain
    with torch.no_grad():
        d._reverse_chain(m, xa, 3, "gauss", None)
        d._reverse_chain(m, xa[:1], 3, fn, None)
    assert len(d._chains) == 3


def test_detection_B_records_and_return(tmp_path, monkeypatch):
    from oracle import metrics_oracle as mo
    GD, m, d = tiny()
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(1)
    x_0 = torch.rand(1, 1, 32, 32, device=DEV) * 2 - 1
    mask = (torch.rand(1, 1, 32, 32, device=DEV) > 0.7).float()
    args = {"arg_num": 9, "T": 100, "img_size": [32, 32]}
    out = d.detection_B(m, x_0, args, ("vol", "slice"), mask, denoise_fn="gauss", total_avg=2)
    assert out == [None]                                     # range(50, 80, 50): upstream appends heatmap()'s None
    assert d.last_chain_schedule["chain_steps"] == 100 and d.last_chain_schedule["slots"] == 2
    assert not os.path.exists(tmp_path / "diffusion-videos")
    rec = d.last_detection[0]
    assert rec["t_distance"] == 50 and rec["output"].shape == (2, 1, 32, 32)
    omaps, oc = mo.anomaly_maps(x_0.cpu().numpy(), rec["output"].cpu().numpy()[:, None], mask.cpu().numpy())
    assert np.array_equal(rec["mean"].cpu().numpy(), omaps["mean"])
    assert np.array_equal(rec["mse"].cpu().numpy(), omaps["mse_img"])
    assert np.array_equal(rec["threshold"].cpu().numpy(), omaps["thr_img"])
    assert np.array_equal(rec["counts"].cpu().numpy()[:, :9], oc[:, :9])
    # octave variant re-assigns noise_fn (stateful, like upstream) and shortens the range; like upstream it needs
    # a model constructed with a simplex noise type (self.simplex only exists then, GaussianDiffusion.py:164-165)
    with pytest.raises(AttributeError):
        d.detection_B(m, x_0, args, ("vol", "slice"), mask, denoise_fn="octave", total_avg=2)
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(100, "linear"), noise="simplex")
    np.random.seed(4)
    out = d.detection_B(m, x_0, args, ("vol", "slice"), mask, denoise_fn="octave", total_avg=2)
    assert out == [None] and d.last_detection[0]["output"].shape == (2, 1, 32, 32)
    n = d.noise_fn(x_0, torch.zeros(1, dtype=torch.int64, device=DEV))
    assert n.shape == x_0.shape and n.is_cuda


def test_detection_B_returns_upstream_list_and_keeps_device_results(tmp_path, monkeypatch):
    GD, m, d = tiny()
    monkeypatch.chdir(tmp_path)
    x_0 = torch.rand(1, 1, 32, 32, device=DEV) * 2 - 1
    mask = torch.zeros(1, 1, 32, 32, device=DEV)
    args = {"arg_num": 9, "T": 100, "img_size": [32, 32]}
    out = d.detection_B(m, x_0, args, ("vol", "slice"), mask, denoise_fn="gauss", total_avg=3)
    assert out == [None]                                             # upstream appends evaluation.heatmap()'s return value
    rec = d.last_detection[0]
    assert rec["t_distance"] == 50 and rec["output"].shape == (3, 1, 32, 32) and rec["mean"].is_cuda
    assert not os.listdir(tmp_path)                                  # no figure / directory output from the product


def test_detection_A_frequency_sweep(tmp_path, monkeypatch):
    GD, m, _ = tiny()
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(100, "linear"), noise="simplex")
    monkeypatch.chdir(tmp_path)
    np.random.seed(2)
    x_0 = torch.rand(1, 1, 32, 32, device=DEV) * 2 - 1
    mask = torch.zeros(1, 1, 32, 32, device=DEV)
    args = {"arg_num": 9, "T": 100, "img_size": [32, 32]}
    assert d.detection_A(m, x_0, args, ("vol", "slice"), mask, total_avg=2) is None
    recs = d.last_detection
    assert [r["freq"] for r in recs] == [7, 6, 5, 4, 3, 2, 1] and all(r["t_distance"] == 50 for r in recs)
    assert all(torch.isfinite(r["output"]).all() and r["output"].abs().max() <= 1.0 + 1e-6 for r in recs)
    # every setting's chains are the caller's own tensor: the kept ReverseChain's buffer is reused, the records must not alias it
    ptrs = [r["output"].data_ptr() for r in recs]
    assert len(set(ptrs)) == len(ptrs)
    assert all(not torch.equal(recs[0]["output"], r["output"]) for r in recs[1:])
    with pytest.raises(ValueError):
        d._avg_chains(m, x_0.repeat(2, 1, 1, 1), 5, 2)


def test_forward_backward_results_are_independent_tensors_and_chains_can_be_released():
    """the kept chain's buffer is overwritten by the next call -- a returned tensor must not change afterwards
    (the reference returns a fresh tensor from every forward_backward, GaussianDiffusion.py:351-359)."""
    import copy
    import pickle
    GD, m, d = tiny()
    torch.manual_seed(11)
    xa = torch.rand(2, 1, 32, 32, device=DEV) * 2 - 1
    xb = torch.rand(2, 1, 32, 32, device=DEV) * 2 - 1
    a = d.forward_backward(m, xa, see_whole_sequence=None, t_distance=6)
    keep = a.clone()
    b = d.forward_backward(m, xb, see_whole_sequence=None, t_distance=6)       # same (model, shape, noise): the chain is restarted
    assert len(d._chains) == 1 and a.data_ptr() != b.data_ptr()
    assert torch.equal(a, keep) and not torch.equal(a, b)
    chain = next(iter(d._chains.values()))
    assert a.data_ptr() != chain.x.data_ptr() and b.data_ptr() != chain.x.data_ptr()
    # device-side caches do not travel: deepcopy works after sampling (CUDAGraph objects cannot be copied), and the copy samples
    d2 = copy.deepcopy(d)
    assert "_chains" not in d2.__dict__ and d2._dev == {} and len(d._chains) == 1
    c = d2.forward_backward(m, xa, see_whole_sequence=None, t_distance=3)
    assert torch.isfinite(c).all()
    st = d.__getstate__()
    assert "_chains" not in st and st["_dev"] == {}
    pickle.dumps({k: v for k, v in st.items() if not callable(v)})            # everything but the (lambda) noise functions pickles
    d.release_chains()
    assert "_chains" not in d.__dict__
    e = d.forward_backward(m, xa, see_whole_sequence=None, t_distance=6)        # builds and captures a fresh chain
    assert len(d._chains) == 1 and torch.isfinite(e).all()


def test_detection_A_fixedT_matches_reference_fixture():
    """detection_A_fixedT (GaussianDiffusion.py:596-623) against the reference's own output on CPU
    (tests/golden/detection_fixedT.npz): two frequencies x (forward simplex noise + a 250-step simplex reverse chain),
    every draw from the seeded numpy stream.  250 steps amplify the fp32 differences between the HIP UNet and ATen's CPU
    kernels, so images are compared to 2e-3 and the thresholded map only away from its discontinuity."""
    from conftest import GOLDEN
    import GaussianDiffusion as GD
    g = np.load(os.path.join(GOLDEN, "detection_fixedT.npz"))
    _, m, _ = tiny()
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(1000, "linear"), noise="simplex")
    x0, mask = torch.from_numpy(g["x0"]).to(DEV), torch.from_numpy(g["mask"]).to(DEV)
    np.random.seed(int(g["np_seed"]))
    out = d.detection_A_fixedT(m, x0, {"img_size": [32, 32]}, mask, end_freq=int(g["end_freq"]))
    assert np.random.randint(-10000000000, 10000000000) == int(g["next_randint"])      # same numpy-stream consumption
    ref = torch.from_numpy(g["output"])
    out = out.cpu()
    assert out.shape == ref.shape
    for f in range(int(g["end_freq"])):
        o, r = out[6 * f:6 * f + 6], ref[6 * f:6 * f + 6]
        assert torch.equal(o[0], r[0]) and torch.equal(o[5], r[5])                      # x_0, mask
        assert torch.equal(o[1], r[1]), "x_noised: sample_q of bit-exact simplex noise is bit-exact"
        assert (o[2] - r[2]).abs().max() < 2e-3, float((o[2] - r[2]).abs().max())       # reconstruction after 250 steps
        assert (o[3] - r[3]).abs().max() < 1e-2                                         # mse image (2*sq - 1)
        away = r[3].abs() > 2e-2
        assert torch.equal(o[4][away], r[4][away])                                      # threshold image


def test_graph_capture_only_for_capture_safe_noise():
    """Only device-side / pre-drawn noise sources may be replayed from a captured graph; host-RNG callables run eagerly
    (a captured newSeed() + table upload would repeat the first step's seed on every replay)."""
    GD, m, _ = tiny()
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(1000, "linear"), noise="simplex")
    x = torch.rand(1, 1, 32, 32, device=DEV) * 2 - 1
    lam = lambda xx, tt: GD.generate_simplex_noise(d.simplex, xx, tt, False, frequency=8).float()
    assert GD.ReverseChain(d, m, x, 4, lam).use_graph is False
    assert GD.ReverseChain(d, m, x, 4, "gauss").use_graph is True
    assert GD.ReverseChain(d, m, x, 4, "simplex").use_graph is True
    assert GD.ReverseChain(d, m, x, 4, "noise_fn").use_graph is True               # default simplex noise_fn -> SimplexNoiseFn
    d.noise_fn = lam                                                                # user-replaced noise_fn: host RNG
    assert GD.ReverseChain(d, m, x, 4, "noise_fn").use_graph is False
    with pytest.raises(ValueError):
        GD.ReverseChain(d, m, x, 4, lam, use_graph=True)
    d2 = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(1000, "linear"), noise="simplex_randParam")
    assert GD.ReverseChain(d2, m, x, 4, "noise_fn").use_graph is False
    # "noise_fn" on a simplex diffusion: graph replay == eager launches == the per-step callable, draw for draw
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(1000, "linear"), noise="simplex")
    outs = []
    for mode in ("graph", "eager", "callable"):
        np.random.seed(21)
        fn = "noise_fn" if mode != "callable" else (lambda xx, tt: GD.generate_simplex_noise(d.simplex, xx, tt, False).float())
        ch = GD.ReverseChain(d, m, x, 6, fn, use_graph=(True if mode == "graph" else False))
        for _ in range(6):
            ch.step()
        ch.finish()
        outs.append((ch.x.clone(), np.random.randint(0, 1 << 30)))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[1][0], outs[2][0])
    assert outs[0][1] == outs[1][1] == outs[2][1]


def test_out_of_range_t_distance_raises_like_extract():
    GD, m, d = tiny()                                                # 100-step schedule
    x = torch.rand(1, 1, 32, 32, device=DEV)
    with pytest.raises(IndexError):
        GD.ReverseChain(d, m, x, 101, "gauss")
    # the kernels themselves never read past the tables: out-of-range t poisons the sample instead
    bad = d.sample_q(x, torch.tensor([100], device=DEV), torch.zeros_like(x))
    assert torch.isnan(bad).all()
    ok = d.sample_q(x, torch.tensor([-1], device=DEV), torch.zeros_like(x))       # python-style negative index, as numpy
    assert torch.equal(ok, d.sample_q(x, torch.tensor([99], device=DEV), torch.zeros_like(x)))


# ---------------------------------------------------------------------------------------------------------------------------------
# The slot-batched loops against the REFERENCE's serial loops (tests/golden/detection_loops_kat.npz, make_golden.py:
# gen_detection_loops): the reference ran detection_B / detection_A on CPU with torch.randn_like replaced by a keyed stream
# (tests/golden/keyed.py: the value depends on (chain in upstream's loop order, t)); here the same keyed values are handed to
# whichever slot holds that chain at that timestep.  Pins what the scheduler could get wrong: the forward-noise draw order,
# which chain lands in output[avg], and the mean -> mse -> threshold post-processing (GaussianDiffusion.py:514-520, 569-576).

class _KeyedDraws:
    """torch.randn_like replacement for the build: forward noise while the chains are being set up, per-slot step noise inside
    _run_chains (slot -> (chain, t) from the schedule the loop itself published in last_chain_schedule)."""

    def __init__(self, d, lens):
        self.d, self.lens, self.fwd, self.k, self.in_loop = d, lens, 0, 0, False
        real = d._run_chains

        def run_chains(*a, **kw):
            self.in_loop, self.k = True, 0
            try:
                return real(*a, **kw)
            finally:
                self.in_loop = False
        d._run_chains = run_chains

    def __call__(self, x, *a, **kw):
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
        import keyed
        if not self.in_loop:                                            # detection_B "gauss": one forward draw per chain, upstream order
            c, self.fwd = self.fwd, self.fwd + 1
            return keyed.keyed_normal(c, keyed.FORWARD, x.shape).to(x.device)
        sched = self.d.last_chain_schedule
        out = torch.zeros(x.shape, dtype=torch.float32)
        for c, (slot, start) in sched["place"].items():
            if start <= self.k < start + self.lens[c]:
                out[slot] = keyed.keyed_normal(c, self.lens[c] - 1 - (self.k - start), x.shape[1:])
        self.k += 1
        return out.to(x.device)


def _loops_fixture():
    from conftest import GOLDEN
    import GaussianDiffusion as GD
    g = np.load(os.path.join(GOLDEN, "detection_loops_kat.npz"))
    _, m, _ = tiny()
    T = int(g["T"])
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(T, "linear"), noise="simplex")
    args = {"arg_num": 0, "T": T, "img_size": [32, 32]}
    return g, m, d, args, torch.from_numpy(g["x0"]).to(DEV), torch.from_numpy(g["mask"]).to(DEV)


def _check_settings(recs, ref, x0, mask, navg):
    """ref[j] = what upstream hands to its figure for setting j: cat[x_0, output[:3], mean, mse, threshold, mask]."""
    assert len(recs) == ref.shape[0]
    nshow = min(3, navg)
    for j, rec in enumerate(recs):
        r = torch.from_numpy(ref[j])
        assert torch.equal(r[0], x0[0].cpu()) and torch.equal(r[-1], mask[0].cpu())
        out = rec["output"].cpu()
        assert out.shape[0] == navg
        for a in range(nshow):                                          # the chain upstream stored in output[a]
            err = float((out[a] - r[1 + a]).abs().max())
            assert err < 2e-3, (j, a, err)
        mean, mse, thr = r[1 + nshow], r[2 + nshow], r[3 + nshow]
        assert float((rec["mean"].cpu()[0] - mean).abs().max()) < 2e-3
        assert float((rec["mse"].cpu()[0] - mse).abs().max()) < 1e-2
        away = mse.abs() > 2e-2
        assert torch.equal(rec["threshold"].cpu()[0][away], thr[away])


@pytest.mark.parametrize("slots", [None, 3])
@pytest.mark.parametrize("mode", ["gauss", "octave"])
def test_detection_B_matches_reference_loop(mode, slots, monkeypatch, tmp_path):
    g, m, d, args, x0, mask = _loops_fixture()
    navg = int(g[f"B_{mode}_total_avg"])
    end = int(args["T"] * (0.6 if mode == "octave" else 0.8))
    lens = [t for t in range(50, end, 50) for _ in range(navg)]
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("ANODDPM_NO_GRAPH", "1")                           # the keyed draws are host values: eager steps
    if slots is not None:
        monkeypatch.setenv("ANODDPM_DET_SLOTS", str(slots))
    monkeypatch.setattr(torch, "randn_like", _KeyedDraws(d, lens))
    np.random.seed(int(g[f"B_{mode}_np_seed"]))
    ret = d.detection_B(m, x0, args, ("f", "0"), mask, denoise_fn=mode, total_avg=navg)
    assert np.random.randint(-10000000000, 10000000000) == int(g[f"B_{mode}_next_randint"])
    assert ret == [None] * len(range(50, end, 50))
    assert d.last_chain_schedule["slots"] == min(slots or 8, len(lens)) and d.last_chain_schedule["chain_steps"] == sum(lens)
    assert [r["t_distance"] for r in d.last_detection] == list(range(50, end, 50))
    _check_settings(d.last_detection, g[f"B_{mode}"], x0, mask, navg)


def test_detection_A_matches_reference_loop(monkeypatch, tmp_path):
    g, m, d, args, x0, mask = _loops_fixture()
    navg = int(g["A_total_avg"])
    tds = list(range(50, int(args["T"] * 0.6), 50))
    lens = [t for _ in range(7) for t in tds for _ in range(navg)]
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("ANODDPM_NO_GRAPH", "1")
    monkeypatch.setattr(torch, "randn_like", _KeyedDraws(d, lens))
    np.random.seed(int(g["A_np_seed"]))
    assert d.detection_A(m, x0, args, ("f", "0"), mask, total_avg=navg) is None
    assert np.random.randint(-10000000000, 10000000000) == int(g["A_next_randint"])
    assert d.last_chain_schedule["slots"] == 16 and d.last_chain_schedule["chain_steps"] == sum(lens)
    assert [(r["freq"], r["t_distance"]) for r in d.last_detection] == [(i, t) for i in range(7, 0, -1) for t in tds]
    _check_settings(d.last_detection, g["A"], x0, mask, navg)


@pytest.mark.parametrize("slots", [None, 16])
def test_config2_chains_on_slots_equal_serial_chains(slots):
    """The slot-batched loop at the size the product runs it: the 256^2 / base-128 model of BASELINE config 2 stepped at batch
    16 / 12 / 8 (per-layer kernel choice differs from batch 1 and 4).  Every chain equals the same chain run alone through
    sample_p (batch 1, itself pinned to the reference) with the draws its slot saw."""
    import GaussianDiffusion as GD
    from test_gpu_unet import build
    m, _, _ = build("c2_256_b128")
    d = GD.GaussianDiffusionModel([256, 256], GD.get_beta_schedule(1000, "linear"), noise="gauss")
    torch.manual_seed(9)
    x_0 = torch.rand(1, 1, 256, 256, device=DEV) * 2 - 1
    dists = [3, 3, 3, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 3, 2]      # 19 chains, 32 chain-steps
    n = len(dists)
    fwd = torch.randn(n, 1, 256, 256, device=DEV)
    torch.manual_seed(23)
    out = d._run_chains(m, x_0, dists, fwd, slots=slots)
    sched = d.last_chain_schedule
    G = sched["slots"]
    assert G in (16, 12, 8) and (slots is None or G == 16) and sched["chain_steps"] == sum(dists)
    torch.manual_seed(23)
    draws = [torch.randn(G, 1, 256, 256, device=DEV) for _ in range(sched["steps"])]
    worst = 0.0
    for c, dist in enumerate(dists):
        x = d.sample_q(x_0, torch.full((1,), dist, device=DEV, dtype=torch.int64), fwd[c:c + 1])
        slot, start = sched["place"][c]
        for i, t in enumerate(range(dist - 1, -1, -1)):
            tb = torch.full((1,), t, device=DEV, dtype=torch.int64)
            with torch.no_grad():
                x = d.sample_p(m, x, tb, denoise_fn=lambda xx, tt, k=start + i, s=slot: draws[k][s:s + 1])["sample"]
        worst = max(worst, float((out[c:c + 1] - x).abs().max()))
    assert worst < 1e-4, worst
