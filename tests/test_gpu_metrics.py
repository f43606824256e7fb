"""-m gpu: the fused anomaly-map pass (anoddpm_anomaly_map through the C ABI) and the evaluation.py-shaped
functions on top of it, against the oracle and the reference-generated golden values."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN

DEV = "cuda:0"
G = np.load(os.path.join(GOLDEN, "metrics_kat.npz"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("case", ["one", "batch"])
def test_maps_and_counts_match_oracle_and_golden(case):
    import evaluation as ev
    from oracle import metrics_oracle as mo
    real, recon, mask = G[f"{case}_real"], G[f"{case}_recon"], G[f"{case}_mask"]
    maps, counts = ev.anomaly_maps(dev(real), dev(recon), dev(mask))
    omaps, oc = mo.anomaly_maps(real, recon, mask)
    for k in ("mean", "sqerr", "mse_img", "thr_img", "pred"):
        got = maps[k].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), omaps[k].view(np.uint32)), k          # bit-exact (fp-contract off)
        assert np.array_equal(got.view(np.uint32), G[f"{case}_{k}"].view(np.uint32)), k
    c = counts.cpu().numpy()
    assert np.array_equal(c[:, :9], oc[:, :9])                                           # integer counts: exact
    assert np.allclose(c[:, 9], oc[:, 9], rtol=1e-12) and np.array_equal(c[:, 10], oc[:, 10])


@pytest.mark.parametrize("case", ["one", "batch"])
def test_evaluation_functions_match_reference(case):
    import evaluation as ev
    real, mask = dev(G[f"{case}_real"]), dev(G[f"{case}_mask"])
    mean, pred = dev(G[f"{case}_mean"]), dev(G[f"{case}_pred"])
    assert np.isclose(ev.dice_coeff(real, mean, mask).item(), G[f"{case}_dice"], rtol=1e-6)
    assert np.isclose(ev.dice_coeff(real, mean, mask, mse=pred).item(), G[f"{case}_dice_mse"], rtol=1e-6)
    assert np.isclose(ev.precision(mask, pred).item(), G[f"{case}_precision"], rtol=1e-6)
    assert np.isclose(ev.recall(mask, pred).item(), G[f"{case}_recall"], rtol=1e-6)
    assert np.isclose(ev.FPR(mask, pred).item(), G[f"{case}_FPR"], rtol=1e-6)
    assert np.isclose(ev.IoU(mask, pred), G[f"{case}_IoU"], rtol=1e-6)
    assert np.isclose(ev.PSNR(mean, real), G[f"{case}_PSNR"], rtol=1e-5)
    r = ev.anomaly_metrics(real, dev(G[f"{case}_recon"]), mask)
    for k in ("dice", "precision", "recall", "FPR", "IoU", "PSNR"):
        assert np.isclose(r[k], G[f"{case}_{k}"], rtol=1e-5), k


def test_full_size_properties():
    """256x256, batch 4, 5 chains: the counts partition the image, the mean of identical chains is the chain,
    and a reconstruction equal to the input gives an empty prediction."""
    import evaluation as ev
    torch.manual_seed(5)
    real = torch.rand(4, 1, 256, 256, device=DEV) * 2 - 1
    recon = (real[None] + 0.7 * torch.randn(5, 4, 1, 256, 256, device=DEV)).clamp(-1, 1)
    mask = (torch.rand(4, 1, 256, 256, device=DEV) > 0.8).float()
    maps, counts = ev.anomaly_maps(real, recon, mask)
    c = counts.cpu().numpy()
    assert np.all(c[:, 3] + c[:, 4] + c[:, 5] + c[:, 6] == 256 * 256)
    assert np.array_equal(c[:, 0], maps["pred"].sum(dim=(1, 2, 3)).cpu().numpy().astype(np.float64))
    assert torch.allclose(maps["mean"], recon.mean(dim=0), atol=1e-6)
    same, c2 = ev.anomaly_maps(real, real.clone()[None].repeat(4, 1, 1, 1, 1), mask)
    assert torch.equal(same["mean"], real) and float(c2[:, 0].sum()) == 0.0
    assert torch.equal(same["thr_img"], torch.full_like(real, -1.0))


def test_rejects_host_tensors():
    import evaluation as ev
    from anoddpm_amd._lib import AnoddpmError
    with pytest.raises(AnoddpmError):
        ev.anomaly_maps(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
