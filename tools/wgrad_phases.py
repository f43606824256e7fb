#!/usr/bin/env python3
"""Phase timing of wgrad43_kernel (probe instance of an ANODDPM_ABLATE build, ANODDPM_DEBUG12): per workgroup the s_memtime sums of
the five phases of a patch -- stage | V + Z0 transforms | dY round 1 | Z1 transform + next requests | MFMAs -- as fractions of the
loop, and the launch time with WARM (one operand set re-read) and COLD operands (six sets cycled: 1.6 GB, nothing resident).
    ANODDPM_ABLATE=1 ANODDPM_BUILD_TAG=abl python -m anoddpm_amd.build;  ANODDPM_LIB_TAG=abl python tools/wgrad_phases.py [H] [K] [N] [B]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops  # noqa: E402
from anoddpm_amd._lib import WgradArgs, check, current_stream, lib  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 128
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
NSET = 6
xs = [torch.randn(B, H, H, K, device=dev) for _ in range(NSET)]
dys = [torch.randn(B, H, H, N, device=dev) for _ in range(NSET)]
gn = hipops.gn_affine([xs[0]], torch.ones(K, device=dev), torch.zeros(K, device=dev))
pg = lib().anoddpm_wgrad43_groups(K, N, B, H, H)
blocks = (K // 32) * (N // 64)
nws = pg * 9 * K * N + pg * blocks * 16
ws = torch.zeros(nws, device=dev)
dw = torch.zeros(N, K, 3, 3, device=dev)


def args(i):
    st = WgradArgs()
    st.a0, st.a1 = xs[i].data_ptr(), None
    st.gn_scale, st.gn_shift = gn[0].data_ptr(), gn[1].data_ptr()
    st.dy, st.dw, st.ws, st.ws_floats = dys[i].data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel()
    st.a0_bs, st.a1_bs, st.dy_bs = H * H * K, 0, H * H * N
    st.c0, st.c1, st.a0_ld, st.a1_ld, st.dy_ld = K, 0, K, 4, N
    st.H, st.W, st.N, st.B, st.a_mode, st.act, st.gn_ld, st.band, st.accumulate, st.algo = H, H, N, B, 0, 1, K, 4, 0, 1
    return st


sts = [args(i) for i in range(NSET)]


def run(seq, n):
    for i in seq[:2]:
        check(lib().anoddpm_conv3x3_wgrad(ctypes.byref(sts[i]), current_stream()), "wgrad")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n):
        lib().anoddpm_conv3x3_wgrad(ctypes.byref(sts[seq[k % len(seq)]]), current_stream())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


names = ["stage (SiLU, dY round 0 -> LDS)", "transforms V + Z0", "dY round 1 -> LDS", "transform Z1 + requests", "MFMAs"]
for label, seq in (("warm", [0]), ("cold", list(range(NSET)))):
    lib().anoddpm_internal_variant(12, 0)
    us = run(seq, 12)
    assert lib().anoddpm_internal_variant(12, 1) == 0, "the probe exists only in a measurement build (ANODDPM_ABLATE=1)"
    us_p = run(seq, 12)
    lib().anoddpm_internal_variant(12, 0)
    t = ws[pg * 9 * K * N:].view(torch.int64).cpu().numpy().reshape(pg * blocks, 8).astype(np.float64)
    tot = t[:, :5].sum(axis=1)
    frac = (t[:, :5] / tot[:, None]).mean(axis=0)
    per_patch = (tot / t[:, 5]).mean()
    print(f"{label}: {H}x{H} {K}->{N} batch {B}: kernel + fold {us:.1f} us ({us_p:.1f} with the probe), {int(t[0, 5])} patches per workgroup, "
          f"{per_patch:.0f} ticks per patch")
    for n_, f in zip(names, frac):
        print(f"    {n_:34s} {100 * f:5.1f} %   = {f * (us - 25) / t[0, 5]:.2f} us per patch (of the launch minus ~25 us of prologue / epilogue / fold)")
