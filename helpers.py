"""Drop-in module name of the reference (`from helpers import *`, detection.py:13).
Implementation: anoddpm_amd/helpers.py."""
from anoddpm_amd.helpers import *  # noqa: F401,F403
from anoddpm_amd.helpers import __all__  # noqa: F401
