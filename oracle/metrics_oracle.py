"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's anomaly-map arithmetic.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Follows:
  GaussianDiffusion.py:517-520, 572, 581-583   mean over the averaged chains, `mse` image, threshold image
  detection.py:229-232                         squared error, (mse > 0.5).float()
  evaluation.py:26-36  dice_coeff   :39-44 PSNR   :50-55 IoU   :58-61 precision   :65-68 recall   :71-74 FPR
Pinned against the reference's own functions by tests/golden/metrics_kat.npz (tests/golden/make_golden.py metrics).
"""
import numpy as np

NCOUNTS = 12


def anomaly_maps(real, recon, mask=None, threshold=0.5):
    """real [B,...] fp32; recon [navg,B,...] fp32; mask like real or None.
    Returns dict(mean, sqerr, mse_img, thr_img, pred) and counts [B,12] float64 (layout of include/anoddpm_hip.h)."""
    real = np.asarray(real, np.float32)
    recon = np.asarray(recon, np.float32)
    navg = recon.shape[0]
    s = recon[0].copy()
    for k in range(1, navg):                      # torch.mean(dim=0): fp32 running sum in index order, then / N
        s = (s + recon[k]).astype(np.float32)
    mean = (s / np.float32(navg)).astype(np.float32) if navg > 1 else s
    d = (mean - real).astype(np.float32)
    se = (d * d).astype(np.float32)
    img = (se * np.float32(2.0) - np.float32(1.0)).astype(np.float32)
    thr = np.where(img > 0, np.float32(1.0), np.float32(-1.0)).astype(np.float32)
    pred = (se > np.float32(threshold)).astype(np.float32)
    B = real.shape[0]
    mk = np.zeros_like(real) if mask is None else np.asarray(mask, np.float32)
    c = np.zeros((B, NCOUNTS), np.float64)
    for b in range(B):
        p, m = pred[b].ravel().astype(np.float64), mk[b].ravel().astype(np.float64)
        c[b, 0], c[b, 1], c[b, 2] = p.sum(), m.sum(), (p * m).sum()
        c[b, 3] = np.sum((m == 1) & (p == 1))
        c[b, 4] = np.sum((m == 1) & (p == 0))
        c[b, 5] = np.sum((m == 0) & (p == 1))
        c[b, 6] = np.sum((m == 0) & (p == 0))
        c[b, 7] = np.sum((m != 0) & (p != 0))
        c[b, 8] = np.sum((m != 0) | (p != 0))
        c[b, 9] = se[b].astype(np.float64).sum()
        c[b, 10] = real[b].max()
    return dict(mean=mean, sqerr=se, mse_img=img, thr_img=thr, pred=pred), c


def ratios(c, smooth=0.000001):
    c = np.asarray(c, np.float64)
    dice = np.mean((2.0 * c[:, 2] + smooth) / (c[:, 0] + c[:, 1] + smooth))
    tp, fp, fn, tn = c[:, 3].sum(), c[:, 4].sum(), c[:, 5].sum(), c[:, 6].sum()
    return dict(dice=dice, precision=tp / (tp + fp + 1e-6), recall=tp / (tp + fn + 1e-6), FPR=fp / (fp + tn + 1e-6),
                IoU=c[:, 7].sum() / (c[:, 8].sum() + 1e-8))


def psnr(c, numel):
    mse = c[:, 9].sum() / numel
    return 20.0 * np.log10(c[:, 10].max() / np.sqrt(mse))
