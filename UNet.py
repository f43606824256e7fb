"""Drop-in module name of the reference (`from UNet import UNetModel, update_ema_params`,
detection.py:14, diffusion_training.py:16).  Implementation: anoddpm_amd/unet.py."""
from anoddpm_amd.unet import GroupNorm32, UNetModel, update_ema_params, zero_module  # noqa: F401
