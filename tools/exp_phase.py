"""Experiment: are the F(4x4,3x3) launches limited by all workgroups hitting HBM in phase?  One B=4 launch vs two B=2 launches
of the same layer on two streams (different hardware queues -> the two grids run out of phase on disjoint CUs)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops
from anoddpm_amd._lib import lib
from ctypes import c_void_p
dev = torch.device("cuda:0")


def prep(B, c0, c1, N, H):
    C = c0 + c1
    x = torch.randn(B, H, H, C, device=dev)
    srcs = [x[..., :c0].contiguous()] + ([x[..., c0:].contiguous()] if c1 else [])
    w = torch.randn(N, C, 3, 3, device=dev) * 0.02
    b = torch.zeros(N, device=dev)
    gn = hipops.gn_affine(srcs, torch.ones(C, device=dev), torch.zeros(C, device=dev))
    hipops.conv_igemm(srcs, w, b, Hout=H, ks=3, gn=gn, act=1, cfg=3, res=torch.randn(B, H, H, N, device=dev), temb=torch.randn(B, N, device=dev), stats_out=[])
    return hipops.LAST_IGEMM, hipops._LAST_KEEP


def run(st, stream, n):
    for _ in range(n):
        lib().anoddpm_igemm(ctypes.byref(st), c_void_p(stream.cuda_stream))


for (c0, c1, N, H) in ((128, 0, 128, 256), (128, 128, 128, 256), (128, 0, 128, 128)):
    s4, k4 = prep(4, c0, c1, N, H)
    sa, ka = prep(2, c0, c1, N, H)
    sb, kb = prep(2, c0, c1, N, H)
    main = torch.cuda.Stream()
    cands = [torch.cuda.Stream() for _ in range(6)]
    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(main):
            e0.record(main)
        fn()
        with torch.cuda.stream(main):
            e1.record(main)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000 / 10
    t4 = timed(lambda: run(s4, main, 10))
    t2s = timed(lambda: (run(sa, main, 10), run(sb, main, 10)))
    best = 1e9
    for s in cands:
        def two():
            s.wait_stream(main)
            run(sa, main, 10)
            run(sb, s, 10)
            main.wait_stream(s)
        best = min(best, timed(two))
    print(f"{c0}+{c1}->{N} @{H}: B=4 one launch {t4:7.1f} us | two B=2 serial {t2s:7.1f} us | two B=2 on two queues {best:7.1f} us per pair", flush=True)
