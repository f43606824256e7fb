"""Keyed normal draws shared by the fixture generator (make_golden.py: gen_detection_loops) and the tests that replay them.

The reference's detection_A / detection_B (GaussianDiffusion.py:480-594) draw `torch.randn_like` once per reverse step (sample_p,
:304-306) and, in detection_B's "gauss" mode, once per chain for the forward noise (:552).  A draw is identified by what it is FOR
-- (chain index in upstream's loop order, timestep) -- not by its position in the global stream, so that a scheduler that runs
the chains in another order (the slot-batched loop of this repo) can be fed exactly the values the serial reference saw.  Data
only: no reference code here."""
import torch

FORWARD = -1          # "timestep" of a chain's forward-noise draw


def keyed_normal(chain: int, t: int, shape):
    """N(0, 1) field for (chain, t): a CPU generator seeded with a function of the key."""
    g = torch.Generator().manual_seed(1_000_003 * (int(chain) + 1) + int(t) + 2)
    return torch.randn(tuple(shape), generator=g)
