#!/usr/bin/env python3
"""bench.py -- reverse-diffusion throughput of the AnoDDPM hot path on MI355X.

Metric (BASELINE.json): reverse-diffusion images/sec @256x256, T=1000, simplex noise.
Workload at N=1 (BASELINE config 2): 256x256 MRI-shaped synthetic batch of 4, UNet base 128,
channel mults (1,1,2,2,4,4), heads 2, attention at 16/8, simplex denoise noise with 8 octaves.

A "step" is one reverse-diffusion step (sample_p) over the per-GPU batch: UNet forward + on-device
simplex field + fused update.  Every step of the T=1000 chain costs the same, so K steps are timed and
images/s = global_batch / (T * seconds_per_step)  (SURVEY.md 8d allows N << T with N stated).

Multi-GPU: one process per GPU (torchrun), each rank denoises its own shard of the batch; inference has
no data-path collective, scaling is weak (per-GPU batch fixed).  RCCL is used only for the
timing barrier and the max-over-ranks reduction.

Extra objects on the JSON line: `roofline` for the dominant kernel (the MFMA implicit-GEMM conv,
bound = fp32 matrix peak) measured with HIP events on the launch stream, and `cpu_baseline` = the
stock-PyTorch CPU restatement (oracle/, "port") timed on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    "c2": dict(img=256, base=128, mults="", attn="16,8", heads=2, batch=4, octaves=8,
               name="256x256 simplex(8 oct) T=1000 base128 attn16,8 batch4/GPU (BASELINE config 2)"),
    "c5": dict(img=512, base=128, mults=(1, 1, 2, 2, 4, 4), attn="32,16,8", heads=2, batch=1, octaves=8,
               name="512x512 simplex(8 oct) T=1000 base128 attn32,16,8 batch1/GPU (BASELINE config 5)"),
    "c1": dict(img=64, base=64, mults="", attn="32,16,8", heads=1, batch=1, octaves=6,
               name="64x64 base64 batch1 (BASELINE config 1 shape, on GPU)"),
}
T_STEPS = 1000
MEASURED_TRAFFIC_BYTES_PER_LAUNCH = 1.674e8      # wino_kernel, profiles/r1f_pmc_hbm_by_kernel.csv, config c2 batch 4
PEAK_FP32_MATRIX_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def mri_like(batch, size, device, seed=1234):
    """SURVEY 8d synthetic input: background -1, centred ellipse filled with smoothed noise, in [-1,1]."""
    g = torch.Generator().manual_seed(seed)
    n = torch.randn(batch, 1, size, size, generator=g) * 0.2
    n = torch.nn.functional.avg_pool2d(n, 5, 1, 2)
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    inside = (((xx - size / 2) / (0.38 * size)) ** 2 + ((yy - size / 2) / (0.45 * size)) ** 2) <= 1
    x = torch.where(inside, n.clamp(-1, 1), torch.full_like(n, -1.0))
    return x.to(device)


def fill_weights(model, seed=1234):
    """Random-init weights of the architecture (no checkpoints offline): N(0, 0.02) conv/linear weights
    (incl. the reference's zero-initialised ones so no work is skipped), zero bias, GN affine = 1/0."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in model.named_parameters():
            is_norm = (".in_layers.0." in k or ".out_layers.0." in k or ".norm." in k or k.startswith("out.0."))
            if is_norm:
                p.copy_(torch.ones_like(p) if k.endswith("weight") else torch.zeros_like(p))
            elif k.endswith("bias"):
                p.zero_()
            else:
                fan_in = int(np.prod(p.shape[1:]))
                p.copy_(torch.randn(p.shape, generator=g) * max(0.02, 1.0 / np.sqrt(fan_in)))


def cpu_baseline(cfg, steps=2):
    """The reference's CPU path restated (oracle/unet_oracle.py + simplex oracle + diffusion oracle), one
    image, timed on the host cores: warm-up 1, then `steps` full reverse steps."""
    from oracle import unet_oracle as uo, diffusion_oracle as do
    from oracle.simplex_oracle import OracleSimplex
    # host threads actually used: all cores the process may run on, capped at 64 (the stock ATen CPU
    # kernels stop scaling -- and with SMT siblings regress badly -- beyond that on the 256-thread hosts)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 64))
    torch.set_num_threads(cores)
    kw = dict(img_size=cfg["img"], base_channels=cfg["base"], channel_mults=cfg["mults"],
              attention_resolutions=cfg["attn"], n_heads=cfg["heads"])
    shapes = uo.param_shapes(cfg["img"], cfg["base"], cfg["mults"], 2, cfg["attn"], 1)
    sd = uo.fill_deterministic(shapes)
    tb = do.tables(do.beta_schedule(T_STEPS, "linear"))
    sx = OracleSimplex(12345)
    x = mri_like(1, cfg["img"], "cpu")
    t = torch.tensor([T_STEPS - 1])

    def one_step(x, t):
        eps = uo.forward(sd, x, t, **kw)
        sx.newSeed()
        nz = torch.from_numpy(sx.rand_3d_fixed_T_octaves((cfg["img"], cfg["img"]), t.numpy(), cfg["octaves"], 0.8, 64)
                              .astype(np.float32))[None]
        return do.p_sample_update(tb, x, t, eps, nz)[0]
    x = one_step(x, t)
    t0 = time.perf_counter()
    for i in range(steps):
        x = one_step(x, t - 1 - i)
    dt = (time.perf_counter() - t0) / steps
    return {"value": 1.0 / (T_STEPS * dt), "unit": "images/s", "cores": cores, "kind": "port",
            "sec_per_step": dt,
            "sample": f"{steps} reverse steps (UNet fwd + simplex + update) of 1 image at {cfg['img']}x{cfg['img']}, "
                      f"stock PyTorch CPU fp32 + C/OpenMP simplex, scaled to T={T_STEPS}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=list(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the HIP-event instrumented pass")
    ap.add_argument("--dump-plan", default="", help="write the igemm launch list of the compiled plan (JSON) here")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP) device; there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import GaussianDiffusion as GD
    from UNet import UNetModel
    from anoddpm_amd import _lib

    cfg = dict(CONFIGS[args.config])
    B = args.batch or cfg["batch"]
    torch.manual_seed(1234)
    np.random.seed(1234 + rank)
    model = UNetModel(cfg["img"], cfg["base"], channel_mults=cfg["mults"], n_heads=cfg["heads"],
                      attention_resolutions=cfg["attn"])
    fill_weights(model)
    model.to(dev).eval()
    diff = GD.GaussianDiffusionModel([cfg["img"]] * 2, GD.get_beta_schedule(T_STEPS, "linear"), noise="simplex")
    noise_fn = GD.SimplexNoiseFn(diff.simplex, octave=cfg["octaves"], persistence=0.8, frequency=64)
    diff.noise_fn = noise_fn
    x0 = mri_like(B, cfg["img"], dev, seed=1234 + rank)           # each rank owns a different shard
    t_T = torch.full((B,), T_STEPS - 1, device=dev, dtype=torch.int64)
    x_T = diff.sample_q(x0, t_T, diff.noise_fn(x0, t_T).float())

    total = args.warmup + args.steps
    chain = diff.reverse_chain(model, x_T, T_STEPS, noise_fn)
    assert total * 2 + 2 <= T_STEPS

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        chain.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        chain.step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = el.item()
    if args.dump_plan and rank == 0:
        with open(args.dump_plan, "w") as f:
            json.dump(next(iter(model._plans.values())).igemm_log, f)
    ms_per_step = 1000.0 * elapsed / args.steps
    value = (B * world) / (T_STEPS * ms_per_step / 1000.0)
    finite = bool(torch.isfinite(chain.x).all().item())

    # ---- roofline leg: the same steps again with one HIP-event pair per op on the launch stream
    roofline = None
    prof_ms_per_step = None
    if not args.no_prof:
        import ctypes
        L = _lib.lib()
        plan = next(iter(model._plans.values()))
        L.anoddpm_prof_enable(1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            chain.step()
        torch.cuda.synchronize()
        prof_ms_per_step = 1000.0 * (time.perf_counter() - t1) / args.steps
        ms = (ctypes.c_double * 16)()
        cnt = (ctypes.c_int64 * 16)()
        _lib.check(L.anoddpm_prof_collect(ms, cnt), "prof_collect")
        L.anoddpm_prof_enable(0)
        WINO = 12                                              # profiler slot of the Winograd launches
        w_ms, w_n = ms[WINO], cnt[WINO]
        d_ms, d_n = ms[_lib.OP_IGEMM], cnt[_lib.OP_IGEMM]      # direct implicit-GEMM launches (1x1, small maps, attention)
        ig_ms, ig_n = w_ms + d_ms, w_n + d_n
        w_flops = sum(e["gflop"] for e in plan.igemm_log if e["wino"]) * 1e9      # algorithmic (direct-convolution) FLOPs
        d_flops = sum(e["gflop"] for e in plan.igemm_log if not e["wino"]) * 1e9
        flops_per_step = plan.igemm_flops

        def tf(flops, msec):
            return flops * args.steps / (msec / 1000.0) / 1e12 if msec > 0 else 0.0
        # dominant kernel = wino_kernel when the plan uses it, else the direct kernel
        dom_wino = w_ms >= d_ms
        achieved = tf(w_flops, w_ms) if dom_wino else tf(d_flops, d_ms)
        # FLOPs the matrix pipe actually executes: Winograd F(2x2,3x3) issues 4/9 of the direct count
        executed = tf(w_flops * 4.0 / 9.0, w_ms) if dom_wino else achieved
        roofline = {"bound": "mfma",
                    "kernel": ("wino_kernel (Winograd F(2x2,3x3) 3x3 convolutions, v_mfma_f32_32x32x2_f32)" if dom_wino else
                               "igemm_kernel (direct implicit-GEMM convolution, v_mfma_f32_32x32x2_f32)"),
                    "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_MATRIX_TFLOPS,
                    # HBM-side bytes per launch from the committed PMC passes (not collectable from inside this
                    # process): (2*FETCH_SIZE + WRITE_SIZE) KB averaged over the kernel's 63 launches per step, with the
                    # guide's gfx950 FETCH_SIZE x2 correction.  Only quoted for the workload it was measured on.
                    "traffic": MEASURED_TRAFFIC_BYTES_PER_LAUNCH if (args.config == "c2" and B == 4 and dom_wino) else None,
                    "traffic_source": "profiles/r1f_pmc_hbm_by_kernel.csv",
                    "achieved_is": "ALGORITHMIC direct-convolution FLOPs of the kernel's launches / their HIP-event time (exceeds the executed rate: Winograd does 2.25x fewer multiplies)",
                    "executed_tflops": executed, "executed_frac": executed / PEAK_FP32_MATRIX_TFLOPS,
                    "launches_per_step": (w_n if dom_wino else d_n) / args.steps,
                    "avg_launch_ms": (w_ms / max(w_n, 1)) if dom_wino else (d_ms / max(d_n, 1)),
                    "algorithmic_gflop_per_launch": ((w_flops / 1e9) / max(w_n / args.steps, 1)) if dom_wino else ((d_flops / 1e9) / max(d_n / args.steps, 1)),
                    "share_of_model_flops": (w_flops if dom_wino else d_flops) / max(flops_per_step, 1.0),
                    "other_contraction_kernel": {"kernel": "igemm_kernel (direct: 1x1, pool-fused and 8x8 3x3, qkv/proj, attention)" if dom_wino else "wino_kernel",
                                                 "achieved": tf(d_flops, d_ms) if dom_wino else tf(w_flops, w_ms),
                                                 "launches_per_step": (d_n if dom_wino else w_n) / args.steps,
                                                 "ms_per_step": (d_ms if dom_wino else w_ms) / args.steps},
                    "all_contractions": {"achieved": tf(flops_per_step, ig_ms), "executed_tflops": tf(w_flops * 4.0 / 9.0 + d_flops, ig_ms),
                                         "ms_per_step": ig_ms / args.steps, "algorithmic_gflop_per_step": flops_per_step / 1e9},
                    "class_ms_per_step": {name: ms[code] / args.steps for name, code in
                                          (("winograd", 12), ("igemm_direct", 1), ("gn_stats", 2), ("softmax", 3), ("resample", 4), ("linear", 5),
                                           ("posemb", 6), ("stem", 7), ("layout", 8), ("chan_stats", 9),
                                           ("gn_finalize", 10), ("head", 11))},
                    "instrumented_ms_per_step": prof_ms_per_step}

    out = {
        "metric": "reverse-diffusion images/sec @256x256 T=1000 simplex" if cfg["img"] == 256 else
                  f"reverse-diffusion images/sec @{cfg['img']}x{cfg['img']} T=1000 simplex",
        "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["name"], "per_gpu_batch": B, "global_batch": B * world, "T": T_STEPS,
                   "timed_steps_scaled_to_T": True, "parallelism": f"batch-sharded x{world} (no data-path collective)",
                   "output_finite": finite},
    }
    if roofline:
        out["roofline"] = roofline
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg)
        except Exception as e:                                   # the baseline must never sink the GPU number
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
