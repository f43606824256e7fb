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


def test_detection_B_records_and_return(tmp_path, monkeypatch):
    from oracle import metrics_oracle as mo
    GD, m, d = tiny()
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(1)
    x_0 = torch.rand(1, 1, 32, 32, device=DEV) * 2 - 1
    mask = (torch.rand(1, 1, 32, 32, device=DEV) > 0.7).float()
    args = {"arg_num": 9, "T": 100, "img_size": [32, 32]}
    out = d.detection_B(m, x_0, args, ("vol", "slice"), mask, denoise_fn="gauss", total_avg=2, save=False)
    assert out == [None]                                     # range(50, 80, 50): upstream appends heatmap()'s None
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
        d.detection_B(m, x_0, args, ("vol", "slice"), mask, denoise_fn="octave", total_avg=2, save=False)
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(100, "linear"), noise="simplex")
    np.random.seed(4)
    out = d.detection_B(m, x_0, args, ("vol", "slice"), mask, denoise_fn="octave", total_avg=2, save=False)
    assert out == [None] and d.last_detection[0]["output"].shape == (2, 1, 32, 32)
    n = d.noise_fn(x_0, torch.zeros(1, dtype=torch.int64, device=DEV))
    assert n.shape == x_0.shape and n.is_cuda


def test_detection_B_writes_figures(tmp_path, monkeypatch):
    pytest.importorskip("matplotlib")
    GD, m, d = tiny()
    monkeypatch.chdir(tmp_path)
    x_0 = torch.rand(1, 1, 32, 32, device=DEV) * 2 - 1
    mask = torch.zeros(1, 1, 32, 32, device=DEV)
    args = {"arg_num": 9, "T": 100, "img_size": [32, 32]}
    d.detection_B(m, x_0, args, ("vol", "slice"), mask, denoise_fn="gauss", total_avg=3)
    files = sorted(os.listdir(tmp_path / "diffusion-videos/ARGS=9/Anomalous/vol/slice/gauss"))
    assert len(files) == 2 and files[0].startswith("heatmap-t=50-") and files[1].startswith("t=50-")


def test_detection_A_frequency_sweep(tmp_path, monkeypatch):
    GD, m, _ = tiny()
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(100, "linear"), noise="simplex")
    monkeypatch.chdir(tmp_path)
    np.random.seed(2)
    x_0 = torch.rand(1, 1, 32, 32, device=DEV) * 2 - 1
    mask = torch.zeros(1, 1, 32, 32, device=DEV)
    args = {"arg_num": 9, "T": 100, "img_size": [32, 32]}
    assert d.detection_A(m, x_0, args, ("vol", "slice"), mask, total_avg=2, save=False) is None
    recs = d.last_detection
    assert [r["freq"] for r in recs] == [7, 6, 5, 4, 3, 2, 1] and all(r["t_distance"] == 50 for r in recs)
    assert all(torch.isfinite(r["output"]).all() and r["output"].abs().max() <= 1.0 + 1e-6 for r in recs)
    with pytest.raises(ValueError):
        d._avg_chains(m, x_0.repeat(2, 1, 1, 1), 5, 2)
