#!/usr/bin/env python3
"""Phase timing of one Winograd workgroup (probe variant of wino_kernel, ANODDPM_DEBUG1=1): s_memtime at kernel
entry, after the prologue, after the K loop, after the epilogue's exchange barrier and after the last store retired.
Run on the GPU box:  ANODDPM_DEBUG1=1 python tools/wino_phases.py [H] [K] [N] [B]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ANODDPM_DEBUG1", "1")


def main():
    import hipops
    from anoddpm_amd._lib import IgemmArgs, check, current_stream, lib
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(B, H, H, K, device=dev)
    w = torch.randn(N, K, 3, 3, device=dev) / (3 * K ** 0.5)
    gamma, beta = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    gn = hipops.gn_affine([x], gamma, beta)
    res = torch.randn(B, H, H, N, device=dev)
    nblocks = (H // 8) * (H // 16) * ((N + 63) // 64) * B
    dbg = torch.zeros(nblocks * 8, dtype=torch.int64, device=dev)
    wp = hipops._pack_wino(w)
    out = torch.empty(B, H, H, N, device=dev)
    st = IgemmArgs()
    st.a0, st.a1, st.gn_scale, st.gn_shift = x.data_ptr(), None, gn[0].data_ptr(), gn[1].data_ptr()
    st.bmat, st.bias, st.temb, st.res, st.out, st.ws = wp.data_ptr(), None, None, res.data_ptr(), out.data_ptr(), dbg.data_ptr()
    P = H * H
    st.a0_bs, st.o_bs, st.r_bs = P * K, P * N, P * N
    st.c0, st.c1, st.a0_ld, st.a1_ld = K, 0, K, 4
    st.H, st.W, st.ks, st.a_mode, st.act = H, H, 3, 0, 1
    st.b_mode, st.ldb, st.N, st.temb_ld, st.out_ld, st.res_ld = 0, 0, N, 0, N, N
    st.B, st.heads, st.ksplit, st.cfg, st.alpha, st.gn_ld = B, 1, 1, 2, 1.0, K
    for _ in range(3):
        check(lib().anoddpm_igemm(ctypes.byref(st), current_stream()), "igemm")
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(nblocks, 8)[:, :5].astype(np.float64)
    t0 = t[:, 0].min()
    d = np.diff(t, axis=1)                         # prologue, loop, exchange, finalize+stores
    # s_memtime ticks are shader-clock cycles (about 2.1 GHz under this load); counters of different XCDs are not
    # synchronised, so only per-workgroup differences are meaningful
    names = ["prologue", "K loop", "partials+exchange", "finalize+stores"]
    life = (t[:, 4] - t[:, 0])
    print(f"layer {H}x{H} {K}->{N} batch {B}: {nblocks} workgroups of 256 threads, mean life {life.mean():.0f} ticks")
    for i, nme in enumerate(names):
        print(f"  {nme:20s} mean {d[:, i].mean():9.0f} ticks = {100 * d[:, i].mean() / life.mean():5.1f} %   "
              f"p10 {np.percentile(d[:, i], 10):9.0f}   p90 {np.percentile(d[:, i], 90):9.0f}")
    print(f"  K loop per 16-channel chunk: {d[:, 1].mean() / (K // 16):.0f} ticks (two workgroups share each SIMD)")


if __name__ == "__main__":
    main()
