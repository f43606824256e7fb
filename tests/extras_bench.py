#!/usr/bin/env python3
"""Side measurements that are parity-test configurations rather than the bench line (BASELINE configs 3 & 4):
  * config 4: rand_3d_octaves((1000,256,256), 8, 0.8, 64) -- HIP kernel time vs the C/OpenMP oracle on host cores
  * config 3 (per-GPU share): one training step (p_loss -> backward -> clip -> fused AdamW+EMA), batch 4 at 256^2
  * detection: the `total_avg` chains of one detection_B setting, batched (this build) vs one at a time (upstream's loop)
Lives under tests/ because the config-4 leg times the CPU oracle.
Run on the GPU box:  python tests/extras_bench.py [c4] [c3] [detect]  (prints one JSON object per measurement)."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def simplex_c4():
    from anoddpm_amd import _lib
    from anoddpm_amd._lib import SimplexArgs, check, current_stream, lib
    from simplex import Simplex_CLASS
    from oracle.simplex_oracle import OracleSimplex
    s = Simplex_CLASS()
    s.newSeed(12345)
    dev = torch.device("cuda:0")
    out = torch.empty((1000, 256, 256), dtype=torch.float64, device=dev)
    a = SimplexArgs()
    a.out, a.zvals, a.tables, a.table_sel = out.data_ptr(), None, s.device_tables(dev).data_ptr(), None
    a.z0, a.out_slice_stride, a.nslices, a.H, a.W = 0, 256 * 256, 1000, 256, 256
    a.table_slice_stride, a.table_sel_scale, a.octaves, a.persistence, a.frequency = 0, 1, 8, 0.8, 64.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    check(lib().anoddpm_simplex3_octaves_f64(ctypes.byref(a), current_stream()))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        check(lib().anoddpm_simplex3_octaves_f64(ctypes.byref(a), current_stream()))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    evals = 1000 * 256 * 256 * 8
    o = OracleSimplex(12345)
    t0 = time.perf_counter()
    ref = o._octaves(np.arange(0, 1000, 25), 256, 256, 8, 0.8, 64)          # 40 of the 1000 slices
    cpu_s = (time.perf_counter() - t0) * 25
    ok = bool((out[::25].cpu().numpy().view(np.uint64) == ref.view(np.uint64)).all())
    return {"what": "config4 simplex volume 1000x256x256 x 8 octaves (fp64 out)", "gpu_ms": ms,
            "gpu_Gevals_per_s": evals / ms / 1e6, "gpu_out_GBps": 1000 * 256 * 256 * 8 / ms / 1e6,
            "cpu_oracle_s_scaled_from_40_slices": cpu_s, "cpu_threads": os.cpu_count(), "speedup": cpu_s * 1e3 / ms,
            "bit_exact_vs_oracle_on_sample": ok}


def train_step_c3(batch=4, steps=3):
    import copy
    import GaussianDiffusion as GD
    from UNet import UNetModel
    from anoddpm_amd.training import FlatBuffers, FusedAdamWEMA, train_step
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
    model = UNetModel(256, 128, n_heads=2, attention_resolutions="16,8").to(dev)
    ema = copy.deepcopy(model)
    flat, flat_ema = FlatBuffers(model), FlatBuffers(ema)
    opt = FusedAdamWEMA(flat, flat_ema, lr=1e-4, notify=(model, ema))
    diff = GD.GaussianDiffusionModel([256, 256], GD.get_beta_schedule(1000, "linear"), noise="simplex")
    args = {"train_start": True, "sample_distance": 800}
    x = torch.rand(batch, 1, 256, 256, device=dev) * 2 - 1
    train_step(model, diff, x, args, flat, None, opt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = train_step(model, diff, x, args, flat, None, opt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    mode = "native training plan fwd+bwd"
    return {"what": f"config3 per-GPU share: train step batch {batch} @256^2 base128 ({mode} + fused AdamW+EMA)",
            "sec_per_step": dt, "images_per_s": batch / dt, "loss": float(loss), "params": flat.numel,
            "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}


def detection_chains(total_avg=5, t_distance=50):
    import GaussianDiffusion as GD
    from UNet import UNetModel
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = UNetModel(256, 128, n_heads=2, attention_resolutions="16,8").to(dev).eval()
    diff = GD.GaussianDiffusionModel([256, 256], GD.get_beta_schedule(1000, "linear"), noise="gauss")
    x_0 = torch.rand(1, 1, 256, 256, device=dev) * 2 - 1
    out = {}
    for name, fn in (("batched", lambda: diff._avg_chains(model, x_0, t_distance, total_avg)),
                     ("serial", lambda: [diff._avg_chains(model, x_0, t_distance, 1) for _ in range(total_avg)])):
        fn()                                             # plan build + graph capture for this batch size
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out[name] = time.perf_counter() - t0
    return {"what": f"detection setting: {total_avg} chains x {t_distance} reverse steps @256^2 base128 (GaussianDiffusion.py:554-569)",
            "batched_s": out["batched"], "serial_s": out["serial"], "speedup": out["serial"] / out["batched"],
            "chain_steps_per_s_batched": total_avg * t_distance / out["batched"]}


def conv_backward_kernels():
    """Weight- and data-gradient kernels of the big 3x3 layers of config 2 (batch 4), timed with device events."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import hipops
    dev = torch.device("cuda:0")
    out = []
    for (H, K, N) in ((256, 128, 128), (256, 256, 128), (128, 128, 128), (64, 256, 256), (32, 256, 256), (16, 512, 512)):
        B = 4
        x = torch.randn(B, H, H, K, device=dev)
        dy = torch.randn(B, H, H, N, device=dev)
        gn = hipops.gn_affine([x], torch.ones(K, device=dev), torch.zeros(K, device=dev))
        w = torch.randn(N, K, 3, 3, device=dev) / (3 * K ** 0.5)
        wt = w.flip(2, 3).transpose(0, 1).contiguous()
        tiles = (-(-K // 64)) * (-(-N // 64))
        TW = 32 if H % 32 == 0 else 16
        want_items = max(1, -(-768 // tiles))
        band = max(1, min(H, (B * (H // TW) * H) // want_items))
        flop = 2.0 * B * H * H * K * N * 9
        res = {"layer": f"{H}x{H} {K}->{N} batch {B}", "wgrad_band": band}
        for name, fn in (("wgrad", lambda: hipops.conv_wgrad([x], dy, gn=gn, act=1, band=band)),
                         ("dgrad_winograd", lambda: hipops.conv_igemm([dy], wt, None, Hout=H, ks=3, cfg=2, ksplit=(1 if H >= 64 else 4)))):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3          # includes hipops' own allocation + sync per call: upper bound
            res[name + "_ms"] = ms
            res[name + "_TFLOPs"] = flop / ms / 1e9
        out.append(res)
    return {"what": "3x3 conv backward kernels (weight gradient: anoddpm_conv3x3_wgrad; data gradient: Winograd forward kernel on flipped weights)",
            "layers": out}


if __name__ == "__main__":
    which = sys.argv[1:] or ["c4", "c3", "detect", "bwd"]
    if "c4" in which:
        print(json.dumps(simplex_c4()), flush=True)
    if "c3" in which:
        print(json.dumps(train_step_c3()), flush=True)
    if "detect" in which:
        print(json.dumps(detection_chains()), flush=True)
    if "bwd" in which:
        print(json.dumps(conv_backward_kernels()), flush=True)
