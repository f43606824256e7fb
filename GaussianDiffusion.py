"""Drop-in module name of the reference (`from GaussianDiffusion import GaussianDiffusionModel,
get_beta_schedule`, detection.py:10, diffusion_training.py:12).  Implementation: anoddpm_amd/diffusion.py."""
from anoddpm_amd.helpers import *  # noqa: F401,F403  (the reference leaks torch/os/json/defaultdict this way)
from anoddpm_amd.diffusion import *  # noqa: F401,F403
from anoddpm_amd.diffusion import GaussianDiffusionModel, get_beta_schedule  # noqa: F401
from anoddpm_amd.simplex import Simplex_CLASS  # noqa: F401
