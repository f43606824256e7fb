"""Drop-in module name of the reference (`from simplex import Simplex_CLASS`, GaussianDiffusion.py:9).
Implementation: anoddpm_amd/simplex.py (HIP kernel: anoddpm_amd/csrc/simplex.hip)."""
from anoddpm_amd.simplex import Simplex_CLASS, perm_tables  # noqa: F401
