"""oracle/winograd_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

fp64 numpy restatement of the Winograd F(4x4,3x3) algebra the HIP kernels evaluate (interpolation points 0, +-1, +-2, inf;
Lavin & Gray), forward and weight-gradient (adjoint) forms:

    forward   (csrc/winograd43.hip):  Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          per 4x4 output tile, d = 6x6 input tile
    wgrad     (csrc/wgrad43.hip):     dU = sum_tiles (B^T d B) (.) (A dY A^T),   dg = G^T dU G

They replace nn.Conv2d(3x3, stride 1, padding 1) of the reference (UNet.py:172,193) and its autograd weight gradient
(diffusion_training.py:102); this file has no reference counterpart to be pinned against -- it is pinned against the DIRECT
convolution / correlation in tests/test_oracle_winograd.py, which is what the reference computes.
"""
import numpy as np

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
               [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
              [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def _tiles(x):
    """x: [C][H+2][W+2] zero-padded input -> [C][H/4][W/4][6][6] overlapping tiles."""
    C, Hp, Wp = x.shape
    th, tw = (Hp - 2) // 4, (Wp - 2) // 4
    out = np.empty((C, th, tw, 6, 6))
    for i in range(th):
        for j in range(tw):
            out[:, i, j] = x[:, 4 * i:4 * i + 6, 4 * j:4 * j + 6]
    return out


def conv3x3_f43(x, w):
    """x [Cin][H][W], w [Cout][Cin][3][3] -> y [Cout][H][W] (cross-correlation, zero padding 1), H, W % 4 == 0."""
    Cin, H, W = x.shape
    d = _tiles(np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1))))
    V = np.einsum("ua,cijab,vb->cijuv", BT, d, BT)
    U = np.einsum("ua,ocab,vb->ocuv", G, w.astype(np.float64), G)
    M = np.einsum("ocuv,cijuv->oijuv", U, V)
    Y = np.einsum("pu,oijuv,qv->oijpq", AT, M, AT)
    return Y.transpose(0, 1, 3, 2, 4).reshape(w.shape[0], H, W)


def wgrad3x3_f43(x, dy):
    """x [Cin][H][W], dy [Cout][H][W] -> dw [Cout][Cin][3][3]: the adjoint of conv3x3_f43 w.r.t. w."""
    Cin, H, W = x.shape
    d = _tiles(np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1))))
    V = np.einsum("ua,cijab,vb->cijuv", BT, d, BT)
    t = dy.astype(np.float64).reshape(dy.shape[0], H // 4, 4, W // 4, 4).transpose(0, 1, 3, 2, 4)
    Z = np.einsum("pu,oijpq,qv->oijuv", AT, t, AT)                       # A dY A^T with A = AT^T
    dU = np.einsum("cijuv,oijuv->ocuv", V, Z)
    return np.einsum("ua,ocuv,vb->ocab", G, dU, G)
