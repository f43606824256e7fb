"""Drop-in module name of the reference (`import evaluation`, GaussianDiffusion.py:7, detection.py:9)."""
from anoddpm_amd.metrics import *  # noqa: F401,F403
from anoddpm_amd.metrics import (AUC_score, FPR, IoU, PSNR, ROC_AUC, SSIM, anomaly_maps, anomaly_metrics,  # noqa: F401
                                 dice_coeff, heatmap, precision, recall, testing)
