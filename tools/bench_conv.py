#!/usr/bin/env python3
"""Time single 3x3 layers through the C ABI: python tools/bench_conv.py  (cfg 2 = F(2x2,3x3), cfg 3 = F(4x4,3x3), 0 = direct)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops  # noqa: E402
import ctypes  # noqa: E402
from anoddpm_amd._lib import lib, current_stream  # noqa: E402

SHAPES = [  # B, (c0, c1), N, H
    (4, (128, 0), 128, 256), (4, (128, 128), 128, 256), (4, (128, 0), 128, 128), (4, (256, 0), 256, 128),
    (4, (256, 128), 128, 128), (4, (128, 0), 256, 128), (4, (256, 0), 256, 64), (4, (512, 0), 256, 64), (1, (128, 0), 128, 512),
]
dev = torch.device("cuda:0")
for (B, (c0, c1), N, H) in ([] if (os.environ.get('PW') or os.environ.get('WG')) else SHAPES):
    C = c0 + c1
    x = torch.randn(B, H, H, C, device=dev)
    srcs = [x[..., :c0].contiguous()] + ([x[..., c0:].contiguous()] if c1 else [])
    w = torch.randn(N, C, 3, 3, device=dev) * 0.02
    b = torch.zeros(N, device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    gn = hipops.gn_affine(srcs, gamma, beta)
    line = f"B{B} {c0}+{c1}->{N} @{H}:"
    full = bool(os.environ.get("FULL"))             # FULL=1: + residual, time-embedding addend and GroupNorm statistics
    extra = dict(res=torch.randn(B, H, H, N, device=dev), temb=torch.randn(B, N, device=dev), stats_out=[]) if full else {}
    for cfg in ((3,) if os.environ.get('ONLY3') else (2, 3)):
        for _ in range(2):
            hipops.conv_igemm(srcs, w, b, Hout=H, ks=3, gn=gn, act=1, cfg=cfg, **extra)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # re-launch the prepared struct without the per-call packing of hipops: time 10 launches of the kernel alone
        st = hipops.LAST_IGEMM
        e0.record()
        for _ in range(10):
            lib().anoddpm_igemm(ctypes.byref(st), current_stream())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        gf = 2.0 * C * N * 9 * H * H * B / 1e9
        line += f"  cfg{cfg} {us:7.1f} us ({gf / us * 1e3:6.1f} alg TFLOP/s)"
    print(line, flush=True)

if os.environ.get("PW"):                                 # PW=1: the 1x1 skip shapes, direct (cfg 0 / 1) vs streaming (cfg 4)
    for (B, (c0, c1), N, H) in [(4, (128, 128), 128, 256), (4, (256, 0), 128, 128), (4, (256, 128), 128, 128), (4, (512, 0), 256, 64),
                                (4, (256, 128), 256, 64), (4, (128, 0), 256, 256)]:
        C = c0 + c1
        x = torch.randn(B, H, H, C, device=dev)
        srcs = [x[..., :c0].contiguous()] + ([x[..., c0:].contiguous()] if c1 else [])
        w = torch.randn(N, C, 1, 1, device=dev) * 0.05
        b = torch.zeros(N, device=dev)
        line = f"1x1 B{B} {c0}+{c1}->{N} @{H}:"
        for cfg in (0, 4):
            for _ in range(2):
                hipops.conv_igemm(srcs, w, b, Hout=H, ks=1, cfg=cfg)
            st = hipops.LAST_IGEMM
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib().anoddpm_igemm(ctypes.byref(st), current_stream())
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            gf = 2.0 * C * N * H * H * B / 1e9
            line += f"  cfg{cfg} {us:7.1f} us ({gf / us * 1e3:6.1f} TFLOP/s)"
        print(line, flush=True)

if os.environ.get("WG"):                                 # WG=1: 3x3 weight gradient, direct (algo 0) vs Winograd domain (algo 1)
    from anoddpm_amd._lib import WgradArgs, check
    for (B, (c0, c1), N, H) in [(4, (128, 0), 128, 256), (4, (128, 128), 128, 256), (4, (128, 0), 128, 128), (4, (256, 0), 256, 128),
                                (4, (256, 128), 128, 128), (4, (256, 0), 256, 64), (4, (512, 0), 256, 64)]:
        K = c0 + c1
        x = torch.randn(B, H, H, K, device=dev)
        srcs = [x[..., :c0].contiguous()] + ([x[..., c0:].contiguous()] if c1 else [])
        dy = torch.randn(B, H, H, N, device=dev)
        gn = hipops.gn_affine(srcs, torch.ones(K, device=dev), torch.zeros(K, device=dev))
        line = f"wgrad B{B} {c0}+{c1}->{N} @{H}:"
        for algo in (0, 1):
            TW = 32
            tiles = -(-K // 64) * -(-N // 64)
            per_band = tiles * B * (H // TW)
            nband = max(1, min(H, round(512 / per_band)))
            band = -(-H // nband)
            nitems = B * (H // TW) * -(-H // band)
            nws = lib().anoddpm_wgrad43_groups(K, N, B, H, H) * 9 * K * N if algo else nitems * 9 * K * N
            ws = torch.empty(nws, device=dev)
            dw = torch.zeros(N, K, 3, 3, device=dev)
            st = WgradArgs()
            st.a0, st.a1 = srcs[0].data_ptr(), srcs[1].data_ptr() if c1 else None
            st.gn_scale, st.gn_shift = gn[0].data_ptr(), gn[1].data_ptr()
            st.dy, st.dw, st.ws, st.ws_floats = dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel()
            st.a0_bs, st.a1_bs, st.dy_bs = H * H * c0, H * H * c1, H * H * N
            st.c0, st.c1, st.a0_ld, st.a1_ld, st.dy_ld = c0, c1, c0, max(c1, 4), N
            st.H, st.W, st.N, st.B, st.a_mode, st.act, st.gn_ld, st.band, st.accumulate, st.algo = H, H, N, B, 0, 1, K, band, 0, algo
            for _ in range(2):
                check(lib().anoddpm_conv3x3_wgrad(ctypes.byref(st), current_stream()), "wgrad")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                lib().anoddpm_conv3x3_wgrad(ctypes.byref(st), current_stream())
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 200
            gf = 2.0 * K * N * 9 * H * H * B / 1e9
            line += f"  algo{algo} {us:7.1f} us ({gf / us * 1e3:6.1f} alg TFLOP/s)"
        print(line, flush=True)

if os.environ.get("SM"):                                 # SM=1: the small-map no-split kernel (cfg 5), where its time goes
    def timeit(st, n=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            lib().anoddpm_igemm(ctypes.byref(st), current_stream())
        e0.record()
        for _ in range(n):
            lib().anoddpm_igemm(ctypes.byref(st), current_stream())
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000 / n
    for (B, (c0, c1), N, H, ks) in [(4, (512, 0), 512, 8, 3), (4, (512, 512), 512, 8, 3), (4, (512, 0), 1536, 8, 1), (4, (512, 0), 512, 8, 1),
                                    (4, (512, 0), 1536, 16, 1), (4, (512, 0), 512, 16, 1), (4, (512, 0), 512, 16, 3)]:
        C = c0 + c1
        x = torch.randn(B, H, H, C, device=dev)
        srcs = [x[..., :c0].contiguous()] + ([x[..., c0:].contiguous()] if c1 else [])
        w = torch.randn(N, C, ks, ks, device=dev) * 0.02
        b = torch.zeros(N, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        gn = hipops.gn_affine(srcs, gamma, beta)
        stats = [(hipops.chan_stats(s_, nslab=4), 0) for s_ in srcs]
        line = f"{ks}x{ks} B{B} {c0}+{c1}->{N} @{H}:"
        for name, kw in (("plain", dict()), ("given+silu", dict(gn=gn, act=1)), ("fold+silu", dict(fold=dict(stats=stats, gamma=gamma, beta=beta), act=1)),
                         ("fold+silu+stats", dict(fold=dict(stats=stats, gamma=gamma, beta=beta), act=1, stats_out=[]))):
            hipops.conv_igemm(srcs, w, b, Hout=H, ks=ks, cfg=5, **kw)
            line += f"  {name} {timeit(hipops.LAST_IGEMM):6.1f}"
        if ks == 3:
            cfg, kspl = (2, 8) if H >= 16 else (1, 16)
            hipops.conv_igemm(srcs, w, b, Hout=H, ks=ks, cfg=cfg, ksplit=kspl, gn=gn, act=1, stats_out=[])
            line += f"  | cfg{cfg} ksplit{kspl} main+tail {timeit(hipops.LAST_IGEMM):6.1f}"
        print(line + " us", flush=True)
