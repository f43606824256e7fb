"""Debug: Winograd-domain weight gradient workspace of anoddpm_conv3x3_wgrad algo 1 (dg[a][b] = sum_uv G[u][a] dU[u][v] G[v][b] of the workgroup's patches: the kernel
stores 9 planes per (k, n) since round 6) vs an fp64 torch evaluation."""
import ctypes, sys, os
import torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import hipops
from anoddpm_amd._lib import WgradArgs, lib, check, current_stream
torch.manual_seed(0)
dev = torch.device("cuda:0")
B, c0, N, H = 1, 32, 64, 16
x = torch.randn(B, c0, H, H, dtype=torch.float64)
dy = torch.randn(B, N, H, H, dtype=torch.float64)
gamma, beta = torch.ones(c0, dtype=torch.float64), torch.zeros(c0, dtype=torch.float64)
xs = hipops.nhwc(x.float().to(dev)).contiguous()
gn = hipops.gn_affine([xs], gamma.float().to(dev), beta.float().to(dev))
dyn = hipops.nhwc(dy.float().to(dev)).contiguous()
sc, sh = gn[0].double().cpu(), gn[1].double().cpu()          # [B][C]
act = torch.nn.functional.silu(x * sc[:, :, None, None] + sh[:, :, None, None])
ap = torch.nn.functional.pad(act, (1, 1, 1, 1))
Bt = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
A = torch.tensor([[1, 0, 0, 0], [1, 1, 1, 1], [1, -1, 1, -1], [1, 2, 4, 8], [1, -2, 4, -8], [0, 0, 0, 1]], dtype=torch.float64)
dU = torch.zeros(36, c0, N, dtype=torch.float64)
per_tile = {}
Vs, Zs = {}, {}
for b in range(B):
    for ty in range(H // 4):
        for tx in range(H // 4):
            d = ap[b, :, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6]                  # [C,6,6]
            V = torch.einsum("ur,crs,vs->cuv", Bt, d, Bt)
            g = dy[b, :, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4]
            Z = torch.einsum("ur,nrs,vs->nuv", A, g, A)
            Vs[(ty, tx)], Zs[(ty, tx)] = V, Z
            per_tile[(ty, tx)] = torch.einsum("cuv,nuv->uvcn", V, Z).reshape(36, c0, N)
            dU += per_tile[(ty, tx)]
K = c0
pg = lib().anoddpm_wgrad43_groups(K, N, B, H, H)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
def half_g(t):                                             # [36, ...] -> [9, ...]: dg[a][b] = sum_uv G[u][a] t[u][v] G[v][b]
    return torch.einsum("uv...,ua,vb->ab...", t.reshape(6, 6, *t.shape[1:]), G, G).reshape(9, *t.shape[1:])
dU = half_g(dU)
ws = torch.zeros(pg * 9 * K * N, device=dev)
dw = torch.zeros(N, K, 3, 3, device=dev)
st = WgradArgs()
st.a0, st.a1 = xs.data_ptr(), None
st.gn_scale, st.gn_shift = gn[0].data_ptr(), gn[1].data_ptr()
st.dy, st.dw, st.ws, st.ws_floats = dyn.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel()
st.a0_bs, st.a1_bs, st.dy_bs = H * H * c0, 0, H * H * N
st.c0, st.c1, st.a0_ld, st.a1_ld, st.dy_ld = c0, 0, c0, 4, N
st.H, st.W, st.N, st.B = H, H, N, B
st.a_mode, st.act, st.gn_ld, st.band, st.accumulate, st.algo = 0, 1, K, 4, 0, 1
check(lib().anoddpm_conv3x3_wgrad(ctypes.byref(st), current_stream()), "wgrad")
torch.cuda.synchronize()
got = ws.view(pg, 9, K, N).double().sum(0).cpu()
print("pg", pg, "colsum rows per image", lib().anoddpm_wgrad43_colsum_items(K, N, B, H, H))
err = (got - dU).abs().amax(dim=(1, 2)) / dU.abs().amax()
print("per-plane relative error (3 x 3 taps):")
print(np.array2string(err.view(3, 3).numpy(), precision=3, suppress_small=True))
e_k = (got - dU).abs().amax(dim=(0, 2)) / dU.abs().amax()
print("per input channel:", np.array2string(e_k.numpy(), precision=2))
e_n = (got - dU).abs().amax(dim=(0, 1)) / dU.abs().amax()
print("per output channel:", np.array2string(e_n.numpy(), precision=2))
print("ratio got/ref where ref large:", (got / dU)[dU.abs() > dU.abs().amax() * 0.3][:10])

# which (V tile, Z tile) products does each workgroup's slab contain?  least squares over all 16 x 16 tile pairs
slabs = ws.view(pg, 9, K, N).double().cpu()
keys = sorted(Vs)
basis = torch.stack([half_g(torch.einsum("cuv,nuv->uvcn", Vs[a], Zs[b]).reshape(36, c0, N)).reshape(-1) for a in keys for b in keys], 1)    # [9*K*N, 256]
for g in range(pg):
    sol = torch.linalg.lstsq(basis, slabs[g].reshape(-1, 1)).solution.view(len(keys), len(keys))
    big = [(keys[i], keys[j], round(float(sol[i, j]), 3)) for i in range(len(keys)) for j in range(len(keys)) if abs(sol[i, j]) > 0.05]
    res = (basis @ sol.reshape(-1, 1) - slabs[g].reshape(-1, 1)).abs().max() / slabs[g].abs().max()
    print("slab", g, "residual", float(res), "terms (V tile, Z tile, coef):", big)
