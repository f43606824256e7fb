"""Anomaly-map metrics on the device -- the build's counterpart of the reference's `evaluation.py`
(same function names, argument meaning and return types) on top of ONE fused HIP pass
(`anoddpm_anomaly_map`, csrc/metrics.hip) instead of ~20 ATen dispatches and several D2H copies per image.

Reference call sites: detection.py:229-250 (per test image: squared error -> threshold 0.5 -> dice / precision /
recall / IoU / FPR), GaussianDiffusion.py:517-520, 572-583 (mean of the averaged chains, `mse` / threshold images).
ROC / AUC (sklearn) and SSIM (skimage) stay on the host as upstream (evaluation.py:46-47, 78-87).

`anomaly_metrics` is the native entry: everything the metric loop needs from one launch and one 96-byte D2H copy.
The individual functions accept the reference's arguments; they use the fused pass when handed device tensors of
the shapes the reference passes and raise `AnoddpmError` otherwise (no CPU path)."""
import ctypes

import torch

from . import _lib
from ._lib import AnomalyArgs, check, current_stream, lib

__all__ = ["anomaly_maps", "anomaly_metrics", "heatmap", "dice_coeff", "PSNR", "SSIM", "IoU", "precision", "recall",
           "FPR", "ROC_AUC", "AUC_score", "testing"]

NC = _lib.ANOMALY_NCOUNTS


def _f32c(x, name):
    _lib.require_cuda(x, name)
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous()


def anomaly_maps(real, recon, mask=None, threshold=0.5, want=("mean", "sqerr", "mse_img", "thr_img", "pred")):
    """One fused pass.  real: [B,C,H,W]; recon: [B,C,H,W] (one reconstruction per image) or [navg,B,C,H,W] /
    ([navg,C,H,W] with B == 1: the `output` tensor of detection_A/B); mask like real or None.
    Returns (maps dict of [B,C,H,W] tensors, counts [B,12] float64 on the device)."""
    real = _f32c(real, "anomaly_maps(real)")
    recon = _f32c(recon, "anomaly_maps(recon)")
    B = real.shape[0]
    n = real[0].numel()
    if recon.dim() == real.dim() + 1:
        navg = recon.shape[0]
    elif recon.shape == real.shape:
        navg = 1
    elif B == 1 and recon.dim() == real.dim() and recon.shape[1:] == real.shape[1:]:
        navg = recon.shape[0]                      # [navg,C,H,W] for a single image
    else:
        raise ValueError(f"recon shape {tuple(recon.shape)} does not match real {tuple(real.shape)}")
    if recon.numel() != navg * B * n:
        raise ValueError("recon size mismatch")
    if mask is not None:
        mask = _f32c(mask, "anomaly_maps(mask)")
        if mask.numel() != B * n:
            raise ValueError("mask size mismatch")
    dev = real.device
    maps = {k: torch.empty_like(real) for k in want}
    counts = torch.empty((B, NC), dtype=torch.float64, device=dev)
    ws = torch.empty((_lib.ANOMALY_BLOCKS * B * NC,), dtype=torch.float64, device=dev)
    a = AnomalyArgs()
    a.recon, a.real, a.mask = recon.data_ptr(), real.data_ptr(), (mask.data_ptr() if mask is not None else None)
    for k in ("mean", "sqerr", "mse_img", "thr_img", "pred"):
        setattr(a, k, maps[k].data_ptr() if k in maps else None)
    a.counts, a.workspace, a.workspace_doubles = counts.data_ptr(), ws.data_ptr(), ws.numel()
    a.n, a.recon_as, a.recon_bs = n, B * n, n
    a.navg, a.B, a.threshold = navg, B, float(threshold)
    check(lib().anoddpm_anomaly_map(ctypes.byref(a), current_stream()), "anomaly_map")
    return maps, counts


def _ratios(c, smooth=0.000001):
    """The reference's formulas (evaluation.py:33-36, 50-76) on the summed counts; c: [B,12] float64 (host)."""
    out = {}
    out["dice_per_image"] = (2.0 * c[:, 2] + smooth) / (c[:, 0] + c[:, 1] + smooth)
    out["dice"] = out["dice_per_image"].mean()
    tp, fp_ref, fn_ref, tn = c[:, 3].sum(), c[:, 4].sum(), c[:, 5].sum(), c[:, 6].sum()
    out["precision"] = tp / (tp + fp_ref + 1e-6)            # evaluation.py:58-61 (its "FP" is mask==1 & recon==0)
    out["recall"] = tp / (tp + fn_ref + 1e-6)               # evaluation.py:65-68
    out["FPR"] = fp_ref / (fp_ref + tn + 1e-6)              # evaluation.py:71-74
    out["IoU"] = c[:, 7].sum() / (c[:, 8].sum() + 1e-8)     # evaluation.py:50-55
    return out


def anomaly_metrics(real, recon, mask, threshold=0.5):
    """dice / IoU / precision / recall / FPR / mse / PSNR of detection.py:229-250 from one launch.
    Returns a dict of Python floats plus the maps (device tensors)."""
    maps, counts = anomaly_maps(real, recon, mask, threshold)
    c = counts.cpu()
    r = {k: float(v) for k, v in _ratios(c).items() if k != "dice_per_image"}
    n_total = real.numel()
    mse = float(c[:, 9].sum()) / n_total
    r["mse"] = mse
    r["PSNR"] = float(20.0 * torch.log10(torch.tensor(float(c[:, 10].max())) / torch.sqrt(torch.tensor(mse)))) if mse > 0 else float("inf")
    r["maps"] = maps
    return r


# ---------------------------------------------------------------------------------- evaluation.py surface
def heatmap(real, recon, mask, filename, save=True):
    """evaluation.py:12-22 computes the squared-error / threshold images, plots them and returns None.  The images come from the
    fused pass (anomaly_maps); writing the figure is plot I/O, out of scope: `filename` / `save` are accepted and ignored."""
    anomaly_maps(real, recon, None, want=("mse_img", "thr_img"))
    return None


def dice_coeff(real, recon, real_mask, smooth=0.000001, mse=None):
    """evaluation.py:26-36.  `mse`, when given, is the already thresholded map (detection.py:232)."""
    if mse is None:
        _, counts = anomaly_maps(real, recon, real_mask, threshold=0.5, want=())
    else:
        # thresholded map supplied: pred = (mse > 0.5) reproduces it for a {0,1} map; run the same pass on it
        zeros = torch.zeros_like(_f32c(mse, "dice_coeff(mse)"))
        sq = _f32c(mse, "dice_coeff(mse)").sqrt()                    # (sqrt(m) - 0)^2 = m for m in {0,1}
        _, counts = anomaly_maps(zeros, sq, real_mask, threshold=0.5, want=())
    d = (2.0 * counts[:, 2] + smooth) / (counts[:, 0] + counts[:, 1] + smooth)
    return d.mean(dim=0).float()


def PSNR(recon, real):
    """evaluation.py:39-44 (returns a numpy scalar like upstream)."""
    _, counts = anomaly_maps(real, recon, None, want=())
    c = counts.cpu()
    mse = c[:, 9].sum() / real.numel()
    return (20 * torch.log10(c[:, 10].max() / torch.sqrt(mse))).float().numpy()


def SSIM(real, recon):
    """evaluation.py:46-47 -- host-side (skimage) as upstream."""
    from skimage.metrics import structural_similarity as ssim       # raises ImportError when skimage is absent
    return ssim(real.detach().cpu().numpy(), recon.detach().cpu().numpy(), channel_axis=2)


def _mask_counts(real_mask, recon_mask):
    zeros = torch.zeros_like(_f32c(recon_mask, "metrics(recon_mask)"))
    m = _f32c(recon_mask, "metrics(recon_mask)")
    if not bool(((m == 0) | (m == 1)).all()):
        raise ValueError("recon_mask must be a {0,1} map (detection.py:232)")
    _, counts = anomaly_maps(zeros.reshape(1, -1), m.reshape(1, -1), real_mask.reshape(1, -1), threshold=0.5, want=())
    return counts[0]


def IoU(real, recon):
    """evaluation.py:50-55."""
    c = _mask_counts(real, recon).cpu()
    return float(c[7] / (c[8] + 1e-8))


def precision(real_mask, recon_mask):
    """evaluation.py:58-61."""
    c = _mask_counts(real_mask, recon_mask)
    return (c[3] / (c[3] + c[4] + 1e-6)).float()


def recall(real_mask, recon_mask):
    """evaluation.py:65-68."""
    c = _mask_counts(real_mask, recon_mask)
    return (c[3] / (c[3] + c[5] + 1e-6)).float()


def FPR(real_mask, recon_mask):
    """evaluation.py:71-74."""
    c = _mask_counts(real_mask, recon_mask)
    return (c[4] / (c[4] + c[6] + 1e-6)).float()


def ROC_AUC(real_mask, square_error):
    """evaluation.py:78-82 -- host-side (sklearn) as upstream."""
    from sklearn.metrics import roc_curve
    if isinstance(real_mask, torch.Tensor):
        return roc_curve(real_mask.detach().cpu().numpy().flatten(), square_error.detach().cpu().numpy().flatten())
    return roc_curve(real_mask.flatten(), square_error.flatten())


def AUC_score(fpr, tpr):
    """evaluation.py:85-86."""
    from sklearn.metrics import auc
    return auc(fpr, tpr)


def testing(testing_dataset_loader, diffusion, args, ema, model, test_iters=40, sequences=True):
    """evaluation.py:90-186: the test-set pass `diffusion_training.train` ends with (diffusion_training.py:153).

    Compute kept, in upstream's order and with its draw counts: (1) for `i in range(100, sample_distance, 100)` one
    `forward_backward(ema, x, "half", t_distance=i)` (upstream turns the returned sequence into an mp4 -- the video
    dump is plot I/O and is skipped, the chain itself runs so the loader and the RNG streams advance as upstream;
    `sequences=False`, an extension, skips these chains); (2) `test_iters // Batch_Size + 5` batches of
    `calc_total_vlb(x, model, args)`; (3) as many batches of `PSNR(forward_backward(ema, x, None, T // 2), x)`.
    Prints upstream's six summary lines and (extension) returns them as a dict of (mean, std) pairs.

    Upstream reads the module globals `device`, `np` and `animation`, which only exist when evaluation.py runs as
    `__main__` (evaluation.py:222-232), so its call from diffusion_training.py:153 ends in NameError; here the
    device is the model's."""
    import numpy as np

    device = next(model.parameters()).device
    ema.eval()
    model.eval()

    def batch(pairs=("cifar",)):
        # upstream's own rule per loop: the sequence loop reads data[0] for "cifar" AND "carpet" (evaluation.py:118-120), the VLB
        # and PSNR loops only for "cifar" (:144-149, :155-160) -- the carpet loader yields dicts, and with sequences=False or
        # sample_distance <= 100 upstream runs on it
        data = next(testing_dataset_loader)
        if args["dataset"] in pairs:
            return data[0].to(device)                      # [data, class] pairs
        return data["image"].to(device)

    seq_lens = []
    if sequences:
        for i in range(100, args['sample_distance'], 100):
            out = diffusion.forward_backward(ema, batch(("cifar", "carpet")), see_whole_sequence="half", t_distance=i)
            seq_lens.append(len(out))
    rounds = test_iters // args["Batch_Size"] + 5
    vlb = [diffusion.calc_total_vlb(batch(), model, args) for _ in range(rounds)]
    psnr = []
    for _ in range(rounds):
        x = batch()
        out = diffusion.forward_backward(ema, x, see_whole_sequence=None, t_distance=args["T"] // 2)
        psnr.append(PSNR(out, x))

    def ms(vals):
        return float(np.mean(vals)), float(np.std(vals))

    k = min(199, diffusion.num_timesteps - 1)              # upstream indexes [0][199] (T >= 200 in every config)
    res = {
        "total_vlb": ms([v['total_vlb'].mean(dim=-1).cpu().item() for v in vlb]),
        "prior_vlb": ms([v['prior_vlb'].mean(dim=-1).cpu().item() for v in vlb]),
        "vb@200": ms([v['vb'][0][k].cpu().item() for v in vlb]),
        "x_0_mse@200": ms([v['x_0_mse'][0][k].cpu().item() for v in vlb]),
        "mse@200": ms([v['mse'][0][k].cpu().item() for v in vlb]),
        "PSNR": ms(psnr),
        "sequence_lengths": seq_lens,
    }
    print(f"Test set total VLB: {res['total_vlb'][0]} +- {res['total_vlb'][1]}")
    print(f"Test set prior VLB: {res['prior_vlb'][0]} +- {res['prior_vlb'][1]}")
    print(f"Test set vb @ t=200: {res['vb@200'][0]} +- {res['vb@200'][1]}")
    print(f"Test set x_0_mse @ t=200: {res['x_0_mse@200'][0]} +- {res['x_0_mse@200'][1]}")
    print(f"Test set mse @ t=200: {res['mse@200'][0]} +- {res['mse@200'][1]}")
    print(f"Test set PSNR: {res['PSNR'][0]} +- {res['PSNR'][1]}")
    return res
