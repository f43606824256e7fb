"""anoddpm_amd -- MI355X-native (gfx950) implementation of the AnoDDPM hot path.

Only what the path needs: csrc/ (HIP kernels + C ABI, include/anoddpm_hip.h), the ctypes binding
(_lib), and the host-side mirrors of the reference's interface: simplex.Simplex_CLASS,
diffusion.GaussianDiffusionModel / get_beta_schedule, unet.UNetModel / update_ema_params, helpers.
The repo-root modules GaussianDiffusion.py / UNet.py / simplex.py / helpers.py re-export these
under the reference's module names so detection.py / diffusion_training.py import them unchanged.
"""
__version__ = "0.1.0"
