"""The Winograd F(4x4,3x3) algebra of csrc/winograd43.hip / wgrad43.hip (oracle/winograd_oracle.py, fp64) against the direct 3x3
convolution and its weight gradient as stock PyTorch computes them -- i.e. what the reference's nn.Conv2d does (UNet.py:172,193)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import winograd_oracle as wo


def test_transform_matrices_are_a_minimal_filtering_algorithm():
    # 1-D identity behind the 2-D nesting: A^T [(G g) (.) (B^T d)] = valid correlation of d (6) with g (3) -> 4 outputs
    rs = np.random.RandomState(0)
    d, g = rs.standard_normal(6), rs.standard_normal(3)
    direct = np.array([d[i:i + 3] @ g for i in range(4)])
    assert np.allclose(wo.AT @ ((wo.G @ g) * (wo.BT @ d)), direct, atol=1e-12)


def test_forward_and_adjoint_match_direct_convolution():
    rs = np.random.RandomState(1)
    x = rs.standard_normal((5, 16, 24))
    w = rs.standard_normal((7, 5, 3, 3))
    dy = rs.standard_normal((7, 16, 24))
    xt = torch.from_numpy(x)[None]
    wt = torch.from_numpy(w).requires_grad_(True)
    y = F.conv2d(xt, wt, padding=1)
    assert np.allclose(wo.conv3x3_f43(x, w), y[0].detach().numpy(), atol=1e-10)
    y.backward(torch.from_numpy(dy)[None])
    assert np.allclose(wo.wgrad3x3_f43(x, dy), wt.grad.numpy(), atol=1e-9)
    # adjoint identity <conv(x, w), dy> == <w, wgrad(x, dy)>
    assert abs((wo.conv3x3_f43(x, w) * dy).sum() - (w * wo.wgrad3x3_f43(x, dy)).sum()) < 1e-8
