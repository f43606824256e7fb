#!/usr/bin/env python3
"""Phase timing of the channel-sliced F(4x4,3x3) kernel (probe variants of wino43r_kernel in an ANODDPM_ABLATE build):
s_memtime at kernel entry, after the prologue, after the K loop, after the epilogue's last store was ISSUED and -- variant 5 --
after the stores were acknowledged, plus the CU the workgroup ran on.  From those: the time a workgroup spends in each phase and
the GAP between one workgroup's end and its successor's entry on the same CU (teardown + dispatch + whatever the hardware waits for).
Run on the GPU box:  ANODDPM_ABLATE=1 python -m anoddpm_amd.build --force; python tools/f43_phases.py [H] [K] [N] [B]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import hipops
    from anoddpm_amd._lib import check, current_stream, lib
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(B, H, H, K, device=dev)
    w = torch.randn(N, K, 3, 3, device=dev) / (3 * K ** 0.5)
    gn = hipops.gn_affine([x], torch.ones(K, device=dev), torch.zeros(K, device=dev))
    res = torch.randn(B, H, H, N, device=dev)
    temb = torch.randn(B, N, device=dev)
    nblocks = (H // 16) * (H // 16) * (N // 128) * B
    for variant in (5, 6):
        assert lib().anoddpm_internal_variant(6, 0) == 0
        stats = []
        hipops.conv_igemm([x], w, torch.zeros(N, device=dev), Hout=H, ks=3, gn=gn, act=1, cfg=3, res=res, temb=temb, stats_out=stats)
        st = hipops.LAST_IGEMM
        dbg = torch.zeros(nblocks * 8, dtype=torch.int64, device=dev)
        st.ws = dbg.data_ptr()
        assert lib().anoddpm_internal_variant(6, variant) == 0, "probe variants exist only in a measurement build: ANODDPM_ABLATE=1 python -m anoddpm_amd.build --force"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            check(lib().anoddpm_igemm(ctypes.byref(st), current_stream()), "igemm")
        e0.record()
        check(lib().anoddpm_igemm(ctypes.byref(st), current_stream()), "igemm")
        e1.record()
        torch.cuda.synchronize()
        lib().anoddpm_internal_variant(6, 0)
        us = e0.elapsed_time(e1) * 1000
        t = dbg.cpu().numpy().reshape(nblocks, 8)
        ts = t[:, :5].astype(np.float64)
        hw, xcc = t[:, 5], t[:, 6] & 0xF
        cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
        d = np.diff(ts, axis=1)
        life = ts[:, 4] - ts[:, 0]
        # ticks per microsecond from the kernel's own span on one XCD (counters of different XCDs are not synchronised)
        span = max(ts[xcc == x, 4].max() - ts[xcc == x, 0].min() for x in np.unique(xcc))
        tick_us = span / us
        print(f"variant {variant} ({'stores waited for' if variant == 5 else 'stores not waited for'}): layer {H}x{H} {K}->{N} batch {B}, "
              f"{nblocks} workgroups on {len(np.unique(cu))} CUs, launch {us:.1f} us, ~{tick_us:.0f} ticks/us")
        for i, nme in enumerate(["prologue", "K loop", "epilogue (issue)", "store drain"]):
            print(f"  {nme:18s} mean {d[:, i].mean() / tick_us:7.2f} us = {100 * d[:, i].mean() / life.mean():5.1f} %   "
                  f"p10 {np.percentile(d[:, i], 10) / tick_us:7.2f}   p90 {np.percentile(d[:, i], 90) / tick_us:7.2f}")
        print(f"  K loop per 16-channel chunk: {d[:, 1].mean() / (K // 16) / tick_us:.2f} us;  workgroup life {life.mean() / tick_us:.2f} us")
        gaps, first = [], []
        for c in np.unique(cu):
            rows = ts[cu == c]
            rows = rows[np.argsort(rows[:, 0])]
            first.append(rows[0, 0])
            gaps += list(rows[1:, 0] - rows[:-1, 4])
        gaps = np.array(gaps) if gaps else np.zeros(1)               # one workgroup per CU: no successions
        print(f"  gap end -> next entry on the same CU: mean {gaps.mean() / tick_us:6.2f} us  p10 {np.percentile(gaps, 10) / tick_us:6.2f}  "
              f"p50 {np.percentile(gaps, 50) / tick_us:6.2f}  p90 {np.percentile(gaps, 90) / tick_us:6.2f}   ({len(gaps)} successions)")
        per_cu = np.array([np.sum(cu == c) for c in np.unique(cu)])
        print(f"  workgroups per CU: min {per_cu.min()} max {per_cu.max()};  sum of phases + gaps per CU ~ "
              f"{(life.mean() * per_cu.mean() + gaps.mean() * (per_cu.mean() - 1)) / tick_us:.1f} us")


if __name__ == "__main__":
    main()
