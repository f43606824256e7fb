#!/usr/bin/env python3
"""bench.py -- throughput of the AnoDDPM hot path on MI355X, one JSON line per run.

Default (`--config c2`, BASELINE.json's metric): reverse-diffusion images/sec @256x256, T=1000, simplex noise.
Workload at N=1 (BASELINE config 2): 256x256 MRI-shaped synthetic batch of 4, UNet base 128, channel mults
(1,1,2,2,4,4), heads 2, attention at 16/8, simplex denoise noise with 8 octaves.  A "step" is one reverse-diffusion
step (sample_p) over the per-GPU batch: UNet forward + on-device simplex field + fused update.  Every step of the
T=1000 chain costs the same, so K steps are timed and images/s = global_batch / (T * seconds_per_step)
(SURVEY.md 8d allows N << T with N stated).

Other BASELINE configurations, same JSON contract (`--config`):
  c5  512x512, mults (1,1,2,2,4,4), attention 32/16/8, batch 1 per GPU -- reverse step (as c2)
  c3  the training step of diffusion_training.py:99-107 at 256x256, batch 4 per GPU: p_loss (q_sample + UNet forward)
      -> backward -> [RCCL bucketed all-reduce when N > 1] -> clip + AdamW + EMA; value = trained images/s
  c4  simplex microbench: rand_3d_octaves((1000,256,256), 8 octaves) volumes on the device; value = noise voxels/s
  c1  config 1's 64x64 model on the GPU (used by the contract test; runs in seconds)
  det one image's whole detection_B sweep end to end (every (t_distance, avg) chain of range(50, 600, 50) x 5 in one slot-batched
      reverse loop + on-device anomaly maps per setting): the loop the reference runs around the hot path; a step
      is one image (~32 s; default --steps 2 --warmup 1); value = reverse chain-steps/s; a side line, no roofline object

Multi-GPU: one process per GPU (torchrun), each rank works on its own shard; inference and the simplex microbench
have no data-path collective -- RCCL is used only for the timing barrier and the max-over-ranks
reduction; the training step all-reduces the gradient.  Scaling is weak (per-GPU batch fixed).

Extra objects on the JSON line: `roofline` for the dominant kernel, measured with HIP events on the launch stream in an
instrumented repeat of the same K steps right after the timed region; and `cpu_baseline` = the CPU restatement under
oracle/ ("port") timed on the host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import ctypes
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
    "c2": dict(kind="reverse", img=256, base=128, mults="", attn="16,8", heads=2, batch=4, octaves=8,
               name="256x256 simplex(8 oct) T=1000 base128 attn16,8 batch4/GPU (BASELINE config 2)"),
    "c5": dict(kind="reverse", img=512, base=128, mults=(1, 1, 2, 2, 4, 4), attn="32,16,8", heads=2, batch=1, octaves=8,
               name="512x512 simplex(8 oct) T=1000 base128 mults 1,1,2,2,4,4 attn32,16,8 batch1/GPU (BASELINE config 5)"),
    "c1": dict(kind="reverse", img=64, base=64, mults="", attn="32,16,8", heads=1, batch=1, octaves=6,
               name="64x64 base64 batch1 (BASELINE config 1 shape, on GPU)"),
    "c3": dict(kind="train", img=256, base=128, mults="", attn="16,8", heads=2, batch=4, octaves=6,
               name="training step @256x256 simplex base128 attn16,8 batch4/GPU: q_sample + UNet fwd/bwd + clip + AdamW + EMA "
                    "(BASELINE config 3)"),
    "c4": dict(kind="simplex", img=256, slices=1000, octaves=8, batch=1,
               name="simplex rand_3d_octaves volume 1000x256x256, 8 octaves, persistence 0.8, frequency 64, fp64 "
                    "(BASELINE config 4)"),
    "det": dict(kind="detect", img=256, base=128, mults="", attn="16,8", heads=2, batch=5,
                name="detection_B sweep @256x256 (GaussianDiffusion.py:531-594): every (t_distance, avg) chain of range(50, 600, 50) x 5 "
                     "from octave-simplex-noised x_0, slot-batched, + anomaly maps per setting, base128 attn16,8, one image per GPU "
                     "(SURVEY 8f row 1)"),
}
T_STEPS = 1000
PEAK_FP32_MATRIX_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_FP64_VECTOR_TFLOPS = 78.6        # MI355X_MICROARCH.md: fp64 vector (FMA = 2 flop)
BOOST_CLOCK_GHZ = 2.4                 # the clock the peak figures are quoted at
SUSTAINED_CLOCK_GHZ = 2.28            # GRBM_GUI_ACTIVE / duration of the F(4x4) kernels in the committed counter pass
SIMPLEX_FP64_FLOP_PER_EVAL = 200.0    # fp64 operations of one 3-D OpenSimplex evaluation


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


def host_threads():
    """Host threads the CPU baselines use: every core the process may run on, capped at 64 (the stock ATen CPU kernels
    stop scaling -- and with SMT siblings regress -- beyond that on the 256-thread hosts)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 64))


def host_description():
    """What the CPU baseline ran on: os.cpu_count() and the CPU model string, beside the thread count used."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return {"host_cpu_count": os.cpu_count() or 1, "host_cpus_available": avail, "cpu_model": model}


def tree_hash():
    """Short hash of the sources that decide which kernels a step launches and how (csrc/*, the plans): the committed PMC
    traffic files carry the hash of the tree they were measured on, so that a line can say when its `traffic` is stale."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "anoddpm_amd", "csrc", "*.h*"))) + \
        [os.path.join(ROOT, "anoddpm_amd", f) for f in ("unet.py", "train_plan.py")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


# The sources that decide the byte traffic of ONE kernel class (its kernels, the headers they include, the plan that picks
# shapes and configurations).  A committed traffic file is stale for a class only when THESE changed: round 5's driver line carried
# `stale: true` because measurement-only edits of other kernels had moved the whole-tree hash.
CLASS_SOURCES = {
    "f43": ["csrc/winograd43.hip", "csrc/winograd43r.hip"],
    "f23": ["csrc/winograd.hip", "csrc/wino23s.hip"],
    "igemm": ["csrc/igemm.hip", "csrc/pointwise.hip"],
    "smallmap": ["csrc/smallmap.hip"],
    "attention": ["csrc/attention.hip"],
}
CLASS_COMMON = ["csrc/common.h", "csrc/pack_items.h", "unet.py"]


def class_hash(cls):
    import hashlib
    h = hashlib.sha1()
    for rel in CLASS_SOURCES[cls] + CLASS_COMMON:
        h.update(rel.encode())
        with open(os.path.join(ROOT, "anoddpm_amd", rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


# ------------------------------------------------------------------------------------------ CPU baselines (oracle/)
def cpu_baseline_reverse(cfg, steps=5):
    """The reference's CPU path restated (oracle/unet_oracle.py + simplex oracle + diffusion oracle), one image:
    warm-up 1, then `steps` full reverse steps."""
    from oracle import unet_oracle as uo, diffusion_oracle as do
    from oracle.simplex_oracle import OracleSimplex
    cores = host_threads()
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
            "sample": f"{steps} reverse steps (UNet fwd + simplex + update) of 1 image at {cfg['img']}x{cfg['img']} after 1 warm-up, "
                      f"stock PyTorch CPU fp32 + C/OpenMP simplex, scaled to T={T_STEPS}"}


def cpu_baseline_train(cfg, steps=3):
    """diffusion_training.py:99-107 restated on CPU (oracle/train_oracle.py), one image per step."""
    from oracle import unet_oracle as uo
    from oracle.train_oracle import TrainState
    cores = host_threads()
    torch.set_num_threads(cores)
    kw = dict(img_size=cfg["img"], base_channels=cfg["base"], channel_mults=cfg["mults"],
              attention_resolutions=cfg["attn"], n_heads=cfg["heads"])
    sd = uo.fill_deterministic(uo.param_shapes(cfg["img"], cfg["base"], cfg["mults"], 2, cfg["attn"], 1))
    st = TrainState(sd, kw)
    x = mri_like(1, cfg["img"], "cpu")
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(x.shape, generator=g)
    st.step(x, torch.tensor([500]), noise)
    t0 = time.perf_counter()
    for i in range(steps):
        st.step(x, torch.tensor([100 + i]), noise)
    dt = (time.perf_counter() - t0) / steps
    return {"value": 1.0 / dt, "unit": "images/s", "cores": cores, "kind": "port", "sec_per_step": dt,
            "sample": f"{steps} optimiser steps (q_sample + UNet fwd/bwd + clip + AdamW + EMA) on 1 image at "
                      f"{cfg['img']}x{cfg['img']} after 1 warm-up, stock PyTorch CPU fp32 autograd"}


def cpu_baseline_simplex(cfg, nslices=40):
    """rand_3d_octaves restated in C/OpenMP (oracle/simplex_oracle.c) on every 25th slice of the volume."""
    from oracle.simplex_oracle import OracleSimplex
    o = OracleSimplex(12345)
    S, Z = cfg["img"], cfg["slices"]
    zs = np.arange(0, Z, max(1, Z // nslices))[:nslices]
    o._octaves(zs[:2], S, S, cfg["octaves"], 0.8, 64)
    t0 = time.perf_counter()
    o._octaves(zs, S, S, cfg["octaves"], 0.8, 64)
    dt = time.perf_counter() - t0
    return {"value": len(zs) * S * S / dt, "unit": "voxels/s", "cores": os.cpu_count() or 1, "kind": "port",
            "sec_per_volume_scaled": dt * Z / len(zs),
            "sample": f"{len(zs)} of the {Z} z-slices ({S}x{S}, {cfg['octaves']} octaves), C + OpenMP restatement of simplex.py"}


# ------------------------------------------------------------------------------------------ workloads
class Ctx:
    pass


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: re-exec under `torch.distributed.run` with one rank per
    GPU (the same command line the driver uses for its scaling runs).  Fewer visible devices than ranks is an error, never a
    silent single-rank run -- except with ANODDPM_BENCH_SHARE_GPU=1 (tests on a 1-GPU box: the ranks share the devices
    round-robin and talk over gloo, because RCCL cannot place two ranks on one device)."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < args.gpus and os.environ.get("ANODDPM_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} HIP device(s) visible; refusing to run fewer ranks than asked for")
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X (HIP) device; there is no CPU fallback for the product path")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    raise SystemExit(subprocess.call(cmd, env=env))


def setup_dist(args):
    c = Ctx()
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    c.rank = int(os.environ.get("RANK", "0"))
    c.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    c.dist = None
    if c.world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={c.world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP) device; there is no CPU fallback for the product path")
    ndev = torch.cuda.device_count()
    c.shared = False
    if c.local_rank >= ndev:
        if os.environ.get("ANODDPM_BENCH_SHARE_GPU") != "1":
            raise SystemExit(f"bench.py: rank {c.rank} has no device of its own ({ndev} visible for {c.world} ranks)")
        c.shared = True
    c.shared = c.shared or (c.world > ndev)
    dev_index = c.local_rank % ndev
    if c.world > 1 or os.environ.get("ANODDPM_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if c.shared:
            dist.init_process_group("gloo", rank=c.rank, world_size=c.world)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), rank=c.rank, world_size=c.world)
        c.dist = dist
    torch.cuda.set_device(dev_index)
    c.dev = torch.device("cuda", dev_index)
    return c


def timed(c, args, step_fn):
    """W warm-up steps, then exactly K steps between barrier + synchronize; max over ranks."""
    def barrier():
        torch.cuda.synchronize()
        if c.dist is not None:
            c.dist.barrier()
    for _ in range(args.warmup):
        step_fn()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_fn()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if c.dist is not None:
        c.dist.barrier()
        el = torch.tensor([elapsed], device=("cpu" if c.shared else c.dev), dtype=torch.float64)
        c.dist.all_reduce(el, op=c.dist.ReduceOp.MAX)
        elapsed = el.item()
    return elapsed


def committed_traffic(config_name, batch, kernel_prefixes, cls=None):
    """HBM-side bytes per launch of a kernel class from the committed PMC passes (profiles/r*_traffic_<config>.json, written by
    tools/traffic_from_pmc.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same bench command, corrected with
    the calibration of tools/hbm_calib.hip).  PMC counters cannot be read from inside the process, so the bench line carries the
    committed measurement of the same workload -- or null when none matches (other config / batch)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_{config_name}.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("per_gpu_batch") == batch:
            best = (path, d)
    if best is None:
        return None
    path, d = best
    n = f = w = 0.0
    for k, v in d["kernels"].items():
        if any(k.startswith(p) for p in kernel_prefixes):
            n += v["launches"]
            f += v["fetch_bytes_per_launch"] * v["launches"]
            w += v["write_bytes_per_launch"] * v["launches"]
    if n == 0:
        return None
    # stale = measured on other sources than the ones running now (kernels / plans changed since the PMC passes): the byte counts
    # are of that tree, the timings beside them of this one.  Compared per kernel class when the file carries class hashes
    # (round 6 on), over the whole tree otherwise.
    if cls is not None and cls in d.get("class_hashes", {}):
        was, now = d["class_hashes"][cls], class_hash(cls)
    else:
        was, now = d.get("sources_hash"), tree_hash()
    return {"bytes_per_launch": (f + w) / n, "fetch_bytes_per_launch": f / n, "write_bytes_per_launch": w / n,
            "source": os.path.relpath(path, ROOT), "fetch_factor": d["calibration"]["fetch_factor"],
            "write_factor": d["calibration"]["write_factor"], "tree": d.get("tree"),
            "measured_on_sources": was, "running_sources": now, "stale": was != now}


def per_op_profile(L, plan, steps):
    """Average HIP-event microseconds of every op of the plan (launch order) over the instrumented steps, or None when the
    recorded list does not tile into steps (other run_ops calls were recorded too)."""
    n = L.anoddpm_prof_list(None, None, 0)
    nops = len(plan.ops)
    if n != nops * steps:
        return None
    codes = (ctypes.c_int32 * n)()
    msv = (ctypes.c_float * n)()
    L.anoddpm_prof_list(codes, msv, n)
    return [1000.0 * sum(msv[s * nops + i] for s in range(steps)) / steps for i in range(nops)]


def dump_layers(path, plan, per_op_us):
    """Per-layer profile of the contraction launches (what tools/by_layer.py builds from a rocprofv3 trace), from the executor's own
    HIP events: kind, shape, configuration, launches per step, average us, algorithmic / executed TFLOP/s."""
    from collections import defaultdict
    from anoddpm_amd import _lib
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    it = iter(plan.igemm_log)
    for (code, st), us in zip(plan.ops, per_op_us):
        if code != _lib.OP_IGEMM:
            continue
        e = next(it)
        key = (e["kind"], e["H"], e["K"], e["N"], e["ks"], e["a_mode"], e["cfg"], e["ksplit"])
        a = agg[key]
        a[0] += 1
        a[1] += us
        a[2] = e["gflop"]
    with open(path, "w") as f:
        f.write("kind,H,K,N,ks,a_mode,cfg,ksplit,launches_per_step,avg_us,algorithmic_TFLOPs,executed_TFLOPs,total_us_per_step\n")
        for k, (n, us, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            avg = us / n
            ex = {2: 4.0 / 9.0, 6: 4.0 / 9.0, 3: 0.25}.get(k[6], 1.0)
            f.write(",".join(str(v) for v in k) + f",{n},{avg:.1f},{gf / avg * 1e3:.1f},{gf * ex / avg * 1e3:.1f},{us:.0f}\n")


PEAK_HBM_GBPS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (about 6.3 TB/s achievable)
ACHIEVABLE_HBM_GBPS = 6300.0          # MI355X_MICROARCH.md: 6.29 TB/s measured with a float4 copy (79 % of peak)


def hbm_kernel_rows(plan, B, ms, cnt, steps, extra=()):
    """The memory-bound kernels of the step against the HBM roofline: algorithmic bytes (what the op must move: its inputs once,
    its outputs once) / HIP-event time of its launches.  This is the one place the north star's '>= 70 % of the memory-bandwidth
    roofline' is meaningful."""
    from anoddpm_amd import _lib
    S = plan.S
    by = {"stem": 0.0, "head": 0.0, "resample": 0.0, "chan_stats": 0.0}
    for code, st in plan.ops:
        if code == _lib.OP_STEM:
            by["stem"] += 4.0 * st.B * st.H * st.W * (st.Cin + st.Cout)
        elif code == _lib.OP_HEAD:
            by["head"] += 4.0 * st.B * st.H * st.W * (st.C + st.Cout)
        elif code == _lib.OP_RESAMPLE:
            pin = st.B * st.H * st.W * st.C
            pout = pin * 4 if st.mode == 1 else pin // 4
            by["resample"] += 4.0 * (pin + pout + (pout if st.out_act else 0))
        elif code == _lib.OP_CHAN_STATS:
            by["chan_stats"] += 4.0 * st.B * st.P * st.C
    slots = {"stem": (_lib.OP_STEM, "conv_stem_strip (3x3, Cin=1 -> base channels, NCHW in / NHWC out)"),
             "head": (_lib.OP_HEAD, "conv_head_mfma2 (GroupNorm-apply + SiLU + 3x3 -> 1 channel)"),
             "resample": (_lib.OP_RESAMPLE, "resample2x (avg-pool / nearest x2 of the skip path, + pooled activated operand)"),
             "chan_stats": (_lib.OP_CHAN_STATS, "chan_stats (GroupNorm partial sums of the stem output)")}
    rows = []
    for key, (slot, name) in slots.items():
        t = ms[slot] / steps
        if t > 0 and by[key] > 0:
            gbps = by[key] / (t / 1000.0) / 1e9
            rows.append({"kernel": name, "launches_per_step": cnt[slot] / steps, "ms_per_step": t, "algorithmic_MB_per_step": by[key] / 1e6,
                         "GBps": gbps, "frac_of_8TBps": gbps / PEAK_HBM_GBPS})
    rows.extend(extra)
    return rows


def contraction_roofline(plan, ms, cnt, steps, prof_ms_per_step):
    """`roofline` object of a reverse step from the executor's HIP-event totals (`ms`, `cnt` per profiler slot over `steps`
    instrumented steps of `plan`): the dominant contraction class against the fp32 matrix peak, the other classes, the per-class
    time table.  Returns (roofline, classes, dominant class name)."""
    from anoddpm_amd import _lib

    # contraction classes: profiler slot, launches of the plan, fraction of the direct-convolution FLOPs the matrix pipe executes
    classes = {
        "wino43_kernel (Winograd F(4x4,3x3) 3x3 convolutions on maps >= 64x64, v_mfma_f32_16x16x4_f32)":
            (14, [e for e in plan.igemm_log if e.get("f43")], 36.0 / (16 * 9)),
        "wino_kernel (Winograd F(2x2,3x3) 3x3 convolutions, v_mfma_f32_32x32x2_f32)":
            (12, [e for e in plan.igemm_log if e["wino"] and not e.get("f43")], 4.0 / 9.0),
        "igemm_kernel / pointwise_stream_kernel (direct implicit GEMM with 64x64 / 128x128 tiles, streaming 1x1; v_mfma_f32_32x32x2_f32)":
            (_lib.OP_IGEMM, [e for e in plan.igemm_log if not e["wino"] and e["cfg"] != 5], 1.0),
        "smallmap_kernel (maps <= 16x16 without split-K: 8x8 3x3, 1x1, qkv / proj, GroupNorm finalize in the prologue; v_mfma_f32_16x16x4_f32)":
            (13, [e for e in plan.igemm_log if e["cfg"] == 5], 1.0),
        "attention_kernel (fused QK^T - softmax - AV per attention block, v_mfma_f32_16x16x4_f32)":
            (_lib.OP_ATTENTION, getattr(plan, "attention_log", []), 1.0),
    }
    flops_per_step = plan.igemm_flops

    def tf(flops, msec):
        return flops * steps / (msec / 1000.0) / 1e12 if msec > 0 else 0.0
    rows = {}
    for name, (slot, entries, frac_exec) in classes.items():
        fl = sum(e["gflop"] for e in entries) * 1e9
        rows[name] = dict(ms=ms[slot], n=cnt[slot], alg=fl, exe=fl * frac_exec)
    # dominant kernel = the class with the most time.  `achieved` / `frac` price the FLOPs the matrix pipe EXECUTES
    # (Winograd issues 4/9 resp. 1/4 of the direct-convolution count), so frac <= 1 is a pipe utilisation; the
    # algorithmic (direct-convolution) rate of the same launches is a side field.
    dom = max(rows, key=lambda k: rows[k]["ms"])
    d = rows[dom]
    nl = d["n"] / steps
    ig_ms = sum(r["ms"] for r in rows.values())
    roofline = {"bound": "mfma", "kernel": dom,
                "achieved": tf(d["exe"], d["ms"]), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                "frac": tf(d["exe"], d["ms"]) / PEAK_FP32_MATRIX_TFLOPS,
                "achieved_is": "FLOPs the matrix pipe executes for the kernel's launches (Winograd F(4x4,3x3): 1/4, F(2x2,3x3): 4/9 of the "
                               "direct-convolution count) / their HIP-event time",
                "algorithmic_tflops": tf(d["alg"], d["ms"]),
                # HBM-side bytes per launch of the dominant kernel: PMC counters cannot be read from inside the process, so this
                # is the committed measurement of the same command (profiles/, corrected by the calibrated FETCH_SIZE factor)
                "traffic": None,
                "launches_per_step": nl,
                "avg_launch_ms": d["ms"] / max(d["n"], 1),
                "ms_per_step": d["ms"] / steps,
                "algorithmic_gflop_per_launch": d["alg"] / 1e9 / max(nl, 1),
                "executed_gflop_per_launch": d["exe"] / 1e9 / max(nl, 1),
                "share_of_model_flops": d["alg"] / max(flops_per_step, 1.0),
                "other_contraction_kernels": [{"kernel": k, "achieved": tf(r["exe"], r["ms"]), "algorithmic_tflops": tf(r["alg"], r["ms"]),
                                               "launches_per_step": r["n"] / steps, "ms_per_step": r["ms"] / steps}
                                              for k, r in rows.items() if k != dom and r["n"]],
                "all_contractions": {"executed_tflops": tf(sum(r["exe"] for r in rows.values()), ig_ms),
                                     "algorithmic_tflops": tf(flops_per_step, ig_ms),
                                     "ms_per_step": ig_ms / steps, "algorithmic_gflop_per_step": flops_per_step / 1e9},
                "class_ms_per_step": {name: ms[code] / steps for name, code in
                                      (("winograd_f43", 14), ("winograd_f23", 12), ("igemm_direct", 1), ("smallmap", 13), ("gn_stats", 2), ("softmax", 3), ("resample", 4),
                                       ("linear", 5), ("posemb", 6), ("stem", 7), ("layout", 8), ("chan_stats", 9),
                                       ("gn_finalize", 10), ("head", 11), ("attention", 26))},
                "instrumented_ms_per_step": prof_ms_per_step}
    # the contract figure prices the pipe at its 2.4 GHz boost clock; under this kernel mix the chip holds the clock the committed
    # GRBM_GUI_ACTIVE pass shows (profiles/r5_c2_sq_by_kernel.csv: 2.28 GHz for the F(4x4) class; tools/mfma_ubench.hip probes 2.1 GHz
    # at full matrix load) -- the same achieved rate against that ceiling, for reference
    roofline["peak_at_sustained_clock"] = {"clock_GHz": SUSTAINED_CLOCK_GHZ, "peak": PEAK_FP32_MATRIX_TFLOPS * SUSTAINED_CLOCK_GHZ / BOOST_CLOCK_GHZ,
                                           "frac": roofline["achieved"] / (PEAK_FP32_MATRIX_TFLOPS * SUSTAINED_CLOCK_GHZ / BOOST_CLOCK_GHZ),
                                           "source": "GRBM_GUI_ACTIVE / kernel duration of the committed SQ counter pass (profiles/)"}
    return roofline, classes, dom


def run_reverse(c, args, cfg):
    import GaussianDiffusion as GD
    from UNet import UNetModel
    from anoddpm_amd import _lib
    B = args.batch or cfg["batch"]
    torch.manual_seed(1234)
    np.random.seed(1234 + c.rank)
    model = UNetModel(cfg["img"], cfg["base"], channel_mults=cfg["mults"], n_heads=cfg["heads"],
                      attention_resolutions=cfg["attn"])
    fill_weights(model)
    model.to(c.dev).eval()
    diff = GD.GaussianDiffusionModel([cfg["img"]] * 2, GD.get_beta_schedule(T_STEPS, "linear"), noise="simplex")
    noise_fn = GD.SimplexNoiseFn(diff.simplex, octave=cfg["octaves"], persistence=0.8, frequency=64)
    diff.noise_fn = noise_fn
    x0 = mri_like(B, cfg["img"], c.dev, seed=1234 + c.rank)           # each rank owns a different shard
    t_T = torch.full((B,), T_STEPS - 1, device=c.dev, dtype=torch.int64)
    x_T = diff.sample_q(x0, t_T, diff.noise_fn(x0, t_T).float())
    chain = diff.reverse_chain(model, x_T, T_STEPS, noise_fn)
    table_setup_ms = getattr(chain, "table_setup_ms", 0.0)             # all T steps' newSeed() tables, drawn before the timed region
    assert (args.warmup + args.steps) * 2 + 2 <= T_STEPS
    elapsed = timed(c, args, chain.step)
    if args.dump_plan and c.rank == 0:
        with open(args.dump_plan, "w") as f:
            json.dump(next(iter(model._plans.values())).igemm_log, f)
    ms_per_step = 1000.0 * elapsed / args.steps
    value = (B * c.world) / (T_STEPS * ms_per_step / 1000.0)
    finite = bool(torch.isfinite(chain.x).all().item())

    # ---- roofline leg: the same steps again with one HIP-event pair per op on the launch stream
    roofline = None
    if not args.no_prof:
        L = _lib.lib()
        plan = next(iter(model._plans.values()))
        # the update and the noise kernel are launched by the chain, not by the plan's executor: one HIP-event pair around each
        # of their launches inside the same instrumented steps (in the stream of a running step, not as lone launches)
        pu_t = _TimedEntry(L, "anoddpm_p_sample_update", lambda a: 0.0)
        sx_t = _TimedEntry(L, "anoddpm_simplex3_octaves_f32", lambda a: 0.0)
        L.anoddpm_prof_enable(1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            chain.step()
        torch.cuda.synchronize()
        prof_ms_per_step = 1000.0 * (time.perf_counter() - t1) / args.steps
        ms = (ctypes.c_double * _lib.OP_MAX)()
        cnt = (ctypes.c_int64 * _lib.OP_MAX)()
        _lib.check(L.anoddpm_prof_collect(ms, cnt), "prof_collect")
        L.anoddpm_prof_enable(0)
        per_op_us = per_op_profile(L, plan, args.steps)
        pu_ms_tot, pu_n, _ = pu_t.finish()
        sx_ms_tot, sx_n, _ = sx_t.finish()
        roofline, classes, dom = contraction_roofline(plan, ms, cnt, args.steps, prof_ms_per_step)
        prefix = {14: ["wino43_kernel", "wino43r_kernel"], 12: ["wino_kernel"], _lib.OP_IGEMM: ["igemm_kernel", "pointwise_stream_kernel"], 13: ["smallmap_kernel"],
                  _lib.OP_ATTENTION: ["attention_kernel"]}[classes[dom][0]]
        cls = {14: "f43", 12: "f23", _lib.OP_IGEMM: "igemm", 13: "smallmap", _lib.OP_ATTENTION: "attention"}[classes[dom][0]]
        tr = committed_traffic(args.config, B, prefix, cls)
        if tr is not None:
            # algorithmic bytes of the same launches: operand read once + the fused residual read once (full or quarter
            # resolution; part of the layer the kernel computes) + output written once + the layer's packed
            # weights once
            ent = classes[dom][1]
            wmul = {14: 36.0, 12: 16.0}.get(classes[dom][0], None)
            alg = sum(4.0 * B * e["H"] * e["W"] * (e["K"] * (0.25 if e.get("a_mode") == 1 else 1.0) + e["N"] * (1.0 + e.get("res", 0.0)))
                      + 4.0 * e["K"] * e["N"] * (wmul if wmul else e["ks"] * e["ks"]) for e in ent) / max(len(ent), 1)
            roofline["traffic"] = tr["bytes_per_launch"]
            roofline["traffic_detail"] = dict(tr, algorithmic_bytes_per_launch=alg, ratio_to_algorithmic=tr["bytes_per_launch"] / alg,
                                              GBps_at_avg_launch=tr["bytes_per_launch"] / (roofline["avg_launch_ms"] / 1000.0) / 1e9)
        # the memory-bound kernels against the HBM roofline (HIP events of the same instrumented pass; p_update timed here)
        extra = []
        if pu_n:
            pu_ms = pu_ms_tot / pu_n
            pu_bytes = 16.0 * chain.x.numel()
            extra.append({"kernel": "p_update (fused reverse update: read x_t, eps, noise; write x_{t-1})", "launches_per_step": pu_n / args.steps,
                          "ms_per_step": pu_ms_tot / args.steps, "algorithmic_MB_per_step": pu_bytes / 1e6, "GBps": pu_bytes / (pu_ms / 1000.0) / 1e9,
                          "frac_of_8TBps": pu_bytes / (pu_ms / 1000.0) / 1e9 / PEAK_HBM_GBPS,
                          "note": "4 MB per launch at batch 4: a few microseconds of data behind a launch; HIP-event pair around the launch "
                                  "inside the instrumented step (stream busy before and after)"})
        if per_op_us is not None:
            # the large resample launches one by one (the class row above averages them with five launch-latency-bound ones of < 10 MB)
            for (code, st), us in zip(plan.ops, per_op_us):
                if code == _lib.OP_RESAMPLE:
                    pin = st.B * st.H * st.W * st.C
                    pout = pin * 4 if st.mode == 1 else pin // 4
                    nbytes = 4.0 * (pin + pout + (pout if st.out_act else 0))
                    if nbytes >= 32e6:
                        gbps = nbytes / (us * 1e-6) / 1e9
                        extra.append({"kernel": f"resample2x {st.H}x{st.W}x{st.C} mode {st.mode}{' + activated operand' if st.out_act else ''} (one launch)",
                                      "launches_per_step": 1.0, "ms_per_step": us / 1000.0, "algorithmic_MB_per_step": nbytes / 1e6,
                                      "GBps": gbps, "frac_of_8TBps": gbps / PEAK_HBM_GBPS})
            if args.dump_layers and c.rank == 0:
                dump_layers(args.dump_layers, plan, per_op_us)
        roofline["hbm_kernels"] = hbm_kernel_rows(plan, B, ms, cnt, args.steps, extra)
        for r in roofline["hbm_kernels"]:
            r["frac_of_6.3TBps_achievable"] = r["GBps"] / ACHIEVABLE_HBM_GBPS
        if sx_n:
            roofline["class_ms_per_step"]["simplex"] = sx_ms_tot / args.steps
        if pu_n:
            roofline["class_ms_per_step"]["p_update"] = pu_ms_tot / args.steps
        # kernel launches of one step: the plan's ops, a tail launch per split-K contraction, noise + update + chain_advance
        roofline["kernel_launches_per_step"] = (len(plan.ops) + sum(1 for code, st in plan.ops if code == _lib.OP_IGEMM and st.ksplit > 1)
                                                + (pu_n + sx_n) / args.steps + 1)
    metric = ("reverse-diffusion images/sec @256x256 T=1000 simplex" if cfg["img"] == 256 else
              f"reverse-diffusion images/sec @{cfg['img']}x{cfg['img']} T=1000 simplex")
    out = {"metric": metric, "value": value, "unit": "images/s", "ms_per_step": ms_per_step, "scaling": "weak", "dtype": "f32",
           "config": {"workload": cfg["name"], "per_gpu_batch": B, "global_batch": B * c.world, "T": T_STEPS,
                      "timed_steps_scaled_to_T": True, "parallelism": f"batch-sharded x{c.world} (no data-path collective)",
                      "output_finite": finite,
                      # work outside the timed region: the permutation tables of ALL T steps are drawn (numpy stream order of the
                      # per-step newSeed() calls) and uploaded when the chain is built; per step that is table_setup_ms / T
                      "table_setup_ms": table_setup_ms, "table_setup_us_per_step": 1000.0 * table_setup_ms / T_STEPS}}
    return out, roofline, (lambda: cpu_baseline_reverse(cfg))


class _TimedEntry:
    """Bench-only: wraps one C-ABI entry point of the loaded library with a HIP-event pair on torch's current stream (the
    stream the training operators launch on) and books the FLOPs of each call."""

    def __init__(self, L, name, flops_of):
        self.L, self.name, self.fn, self.flops_of = L, name, getattr(L, name), flops_of
        self.events = []
        setattr(L, name, self)

    def __call__(self, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = self.fn(*a)
        e1.record()
        self.events.append((e0, e1, self.flops_of(a[0]._obj)))
        return rc

    def finish(self):
        setattr(self.L, self.name, self.fn)
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in self.events)
        return ms, len(self.events), sum(f for _, _, f in self.events)


def run_train(c, args, cfg):
    import copy
    import GaussianDiffusion as GD
    from UNet import UNetModel
    from anoddpm_amd import _lib
    from anoddpm_amd.training import FlatBuffers, FusedAdamWEMA, GradAllReducer, train_step
    B = args.batch or cfg["batch"]
    torch.manual_seed(1234)
    np.random.seed(1234 + c.rank)
    model = UNetModel(cfg["img"], cfg["base"], channel_mults=cfg["mults"], n_heads=cfg["heads"],
                      attention_resolutions=cfg["attn"])
    fill_weights(model)
    model.to(c.dev).train()
    ema = copy.deepcopy(model)
    flat, flat_ema = FlatBuffers(model), FlatBuffers(ema)
    reducer = GradAllReducer(flat, force=True) if c.dist is not None else None
    opt = FusedAdamWEMA(flat, flat_ema, lr=1e-4, weight_decay=0.0)
    diff = GD.GaussianDiffusionModel([cfg["img"]] * 2, GD.get_beta_schedule(T_STEPS, "linear"), noise="simplex")
    targs = {"train_start": True, "sample_distance": 800, "Batch_Size": B}
    x = mri_like(B, cfg["img"], c.dev, seed=1234 + c.rank)
    last = {}

    def step():
        last["loss"], _ = train_step(model, diff, x, targs, flat, reducer, opt)
    elapsed = timed(c, args, step)
    ms_per_step = 1000.0 * elapsed / args.steps
    value = B * c.world / (ms_per_step / 1000.0)
    finite = bool(torch.isfinite(last["loss"]).item())
    roofline = None
    if not args.no_prof:
        L = _lib.lib()
        plan = next(iter(model._tplans.values()), None)
        if plan is not None:
            # native training plan: the C++ executor books every op class with HIP events on the launch stream
            L.anoddpm_prof_enable(1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            prof_ms = 1000.0 * (time.perf_counter() - t1) / args.steps
            ms = (ctypes.c_double * _lib.OP_MAX)()
            cnt = (ctypes.c_int64 * _lib.OP_MAX)()
            _lib.check(L.anoddpm_prof_collect(ms, cnt), "prof_collect")
            L.anoddpm_prof_enable(0)
            def conv_fl(st):
                return 2.0 * (st.c0 + st.c1) * st.N * 9 * st.H * st.W * st.B
            names = {1: "igemm_direct", 12: "winograd_f23", 13: "smallmap", 14: "winograd_f43", 3: "softmax", 4: "resample", 5: "linear", 6: "posemb", 7: "stem", 9: "chan_stats",
                     10: "gn_finalize", 11: "head", 15: "wgrad3x3_winograd", 16: "wgrad3x3_direct", 17: "wgrad_pointwise", 18: "gn_silu_backward",
                     19: "pack_weights", 27: "pack_weights_batched", 28: "linear_backward_batched", 20: "softmax_backward", 21: "transpose", 22: "linear_backward", 23: "stem_backward", 24: "head_backward",
                     25: "colsum_fold", 26: "attention"}
            igs = [st for code, st in plan.ops + plan.bops if code == _lib.OP_IGEMM]
            wgs = [st for code, st in plan.bops if code == _lib.OP_WGRAD3]

            def ig_fl(sts):
                return sum(2.0 * (st.c0 + st.c1) * st.N * st.ks * st.ks * st.H * st.W * st.B * st.heads for st in sts)
            # contraction classes of the step: profiler slot, algorithmic FLOPs per step, fraction of them the matrix pipe executes
            classes = {
                "wino43_kernel (forward + data-gradient 3x3 convolutions in Winograd F(4x4,3x3), v_mfma_f32_16x16x4_f32)":
                    (14, ig_fl([st for st in igs if st.cfg == 3]), 0.25),
                "wgrad43_kernel (3x3 weight gradient in the Winograd F(4x4,3x3) domain, v_mfma_f32_16x16x4_f32)":
                    (15, sum(conv_fl(st) for st in wgs if st.algo == 1), 0.25),
                "wgrad_kernel (3x3 weight gradient, nine-tap MFMA tiles: small maps and pool-fused operands, v_mfma_f32_32x32x2_f32)":
                    (16, sum(conv_fl(st) for st in wgs if st.algo != 1), 1.0),
                "wino_kernel / wino23s_kernel (forward + data-gradient 3x3 convolutions in Winograd F(2x2,3x3), v_mfma_f32_32x32x2_f32 / 16x16x4)":
                    (12, ig_fl([st for st in igs if st.cfg in (2, 6)]), 4.0 / 9.0),
                "smallmap_kernel (maps <= 16x16 without split-K: 8x8 3x3, 1x1, qkv / proj, forward + data gradient; v_mfma_f32_16x16x4_f32)":
                    (13, ig_fl([st for st in igs if st.cfg == 5]), 1.0),
                "igemm_kernel / pointwise_stream_kernel (1x1 on large maps, attention backward GEMMs; v_mfma_f32_32x32x2_f32)":
                    (1, ig_fl([st for st in igs if st.cfg not in (2, 3, 5, 6)]), 1.0),
            }
            rows = {k: dict(ms=ms[slot] / args.steps, n=cnt[slot] / args.steps, alg=fl, exe=fl * fr) for k, (slot, fl, fr) in classes.items()}

            def tf(fl, msec):
                return fl / (msec / 1000.0) / 1e12 if msec > 0 else 0.0
            dom = max(rows, key=lambda k: rows[k]["ms"])
            d = rows[dom]
            roofline = {"bound": "mfma", "kernel": dom,
                        "achieved": tf(d["exe"], d["ms"]), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                        "frac": tf(d["exe"], d["ms"]) / PEAK_FP32_MATRIX_TFLOPS,
                        "achieved_is": "FLOPs the matrix pipe executes for the kernel's launches (Winograd F(4x4,3x3) domain: 1/4, F(2x2,3x3): 4/9 of "
                                       "the direct-convolution count) / their HIP-event time",
                        "algorithmic_tflops": tf(d["alg"], d["ms"]),
                        "traffic": None, "launches_per_step": d["n"], "avg_launch_ms": d["ms"] / max(d["n"], 1),
                        "algorithmic_gflop_per_launch": d["alg"] / 1e9 / max(d["n"], 1), "ms_per_step": d["ms"],
                        "other_contraction_kernels": [{"kernel": k, "achieved": tf(r["exe"], r["ms"]), "algorithmic_tflops": tf(r["alg"], r["ms"]),
                                                       "launches_per_step": r["n"], "ms_per_step": r["ms"]}
                                                      for k, r in rows.items() if k != dom and r["n"]],
                        "all_contractions": {"executed_tflops": tf(sum(r["exe"] for r in rows.values()), sum(r["ms"] for r in rows.values())),
                                             "algorithmic_tflops": tf(sum(r["alg"] for r in rows.values()), sum(r["ms"] for r in rows.values())),
                                             "ms_per_step": sum(r["ms"] for r in rows.values())},
                        "class_ms_per_step": {names[c]: ms[c] / args.steps for c in names},
                        "class_launches_per_step": {names[c]: cnt[c] / args.steps for c in names},
                        "unet_ms_per_step": sum(ms[c] for c in names) / args.steps,
                        "instrumented_ms_per_step": prof_ms}
        else:
            wg = _TimedEntry(L, "anoddpm_conv3x3_wgrad", lambda a: 2.0 * (a.c0 + a.c1) * a.N * 9 * a.H * a.W * a.B)
            ig = _TimedEntry(L, "anoddpm_igemm", lambda a: 2.0 * (a.c0 + a.c1) * a.N * a.ks * a.ks * a.H * a.W * a.B * a.heads)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            prof_ms = 1000.0 * (time.perf_counter() - t1) / args.steps
            w_ms, w_n, w_fl = wg.finish()
            i_ms, i_n, i_fl = ig.finish()
            ach = w_fl / (w_ms / 1000.0) / 1e12 if w_ms > 0 else 0.0
            roofline = {"bound": "mfma", "kernel": "wgrad_kernel (3x3 weight gradient, nine-tap MFMA tiles, v_mfma_f32_32x32x2_f32)",
                        "achieved": ach, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MATRIX_TFLOPS,
                        "traffic": None, "launches_per_step": w_n / args.steps, "avg_launch_ms": w_ms / max(w_n, 1),
                        "gflop_per_launch": w_fl / max(w_n, 1) / 1e9, "ms_per_step": w_ms / args.steps,
                        "other_contraction_kernel": {"kernel": "anoddpm_igemm launches (forward convs + data gradients; Winograd where eligible)",
                                                     "algorithmic_tflops": i_fl / (i_ms / 1000.0) / 1e12 if i_ms > 0 else 0.0,
                                                     "launches_per_step": i_n / args.steps, "ms_per_step": i_ms / args.steps},
                        "instrumented_ms_per_step": prof_ms}
    out = {"metric": f"training images/sec @{cfg['img']}x{cfg['img']} (q_sample + UNet fwd/bwd + AdamW + EMA)", "value": value,
           "unit": "images/s", "ms_per_step": ms_per_step, "scaling": "weak", "dtype": "f32",
           "config": {"workload": cfg["name"], "per_gpu_batch": B, "global_batch": B * c.world,
                      "parallelism": (f"data-parallel x{c.world}, bucketed RCCL all-reduce of the flat gradient" if c.dist is not None
                                      else "single GPU (no collective)"),
                      "loss_finite": finite, "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}}
    if reducer is not None and reducer.active:
        # one more step with the stream synchronised inside finish(): where the buckets' all-reduces were enqueued in the
        # backward (host clock, ms after the first) and how long the step then waited for them -- the EXPOSED part of the
        # collective.  No 8-GPU node was available to the build: ANODDPM_BUCKET_MB (default 64) and the cut positions are untuned,
        # this is the line to tune them from.
        reducer.timing_sync = True
        step()
        torch.cuda.synchronize()
        reducer.timing_sync = False
        t = reducer.last_timing
        out["config"]["allreduce"] = {"buckets": t["buckets"], "bucket_MB": [round(v, 1) for v in t["bucket_MB"]],
                                      "enqueue_offset_ms": [round(v, 3) for v in t["enqueue_offset_ms"]],
                                      "exposed_wait_ms": t["exposed_wait_ms"], "world": c.world,
                                      "bucket_MB_setting": reducer.bucket_bytes / (1 << 20),
                                      "cut_ops": [ops for _, ops in reducer.last_launch_log],
                                      "note": "host-clock offsets of the enqueues; exposed_wait_ms = finish(): work.wait() + stream sync"}
    return out, roofline, (lambda: cpu_baseline_train(cfg))


def run_simplex(c, args, cfg):
    from anoddpm_amd._lib import SimplexArgs, check, current_stream, lib
    from simplex import Simplex_CLASS
    s = Simplex_CLASS()
    s.newSeed(12345 + c.rank)
    S, Z = cfg["img"], cfg["slices"]
    out_t = torch.empty((Z, S, S), dtype=torch.float64, device=c.dev)
    a = SimplexArgs()
    a.out, a.zvals, a.tables, a.table_sel = out_t.data_ptr(), None, s.device_tables(c.dev).data_ptr(), None
    a.z0, a.out_slice_stride, a.nslices, a.H, a.W = 0, S * S, Z, S, S
    a.table_slice_stride, a.table_sel_scale, a.octaves, a.persistence, a.frequency = 0, 1, cfg["octaves"], 0.8, 64.0

    def step():
        check(lib().anoddpm_simplex3_octaves_f64(ctypes.byref(a), current_stream()), "simplex3_octaves_f64")
    elapsed = timed(c, args, step)
    ms_per_step = 1000.0 * elapsed / args.steps
    voxels = Z * S * S
    value = voxels * c.world / (ms_per_step / 1000.0)
    roofline = None
    if not args.no_prof:
        evs = []
        for _ in range(args.steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        k_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)
        evals = voxels * cfg["octaves"]
        ach = evals * SIMPLEX_FP64_FLOP_PER_EVAL / (k_ms / 1000.0) / 1e12
        roofline = {"bound": "fp64-alu", "kernel": "simplex3_octaves_kernel<double> (lattice hash in LDS, octaves accumulated in registers)",
                    "achieved": ach, "peak": PEAK_FP64_VECTOR_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP64_VECTOR_TFLOPS,
                    # bit-exactness with the reference forbids contraction (-ffp-contract=off: every multiply and add is its own
                    # instruction), so the ceiling this kernel can reach is the fp64 vector rate WITHOUT fma, half the quoted peak
                    "peak_no_fma": PEAK_FP64_VECTOR_TFLOPS / 2, "frac_of_peak_no_fma": ach / (PEAK_FP64_VECTOR_TFLOPS / 2),
                    "traffic": None, "avg_launch_ms": k_ms, "launches_per_step": 1,
                    "evaluations_per_launch": evals, "fp64_flop_per_evaluation": SIMPLEX_FP64_FLOP_PER_EVAL,
                    "Gevals_per_s": evals / k_ms / 1e6,
                    "hbm_write_GBps": voxels * 8 / (k_ms / 1000.0) / 1e9,
                    "note": "HBM floor (8 B written per voxel at 8 TB/s) is 66 us per volume: the kernel is fp64-ALU bound"}
    out = {"metric": "simplex rand_3d_octaves noise voxels/sec (256x256xT=1000 volume, 8 octaves)", "value": value, "unit": "voxels/s",
           "ms_per_step": ms_per_step, "scaling": "weak", "dtype": "f64",
           "config": {"workload": cfg["name"], "volumes_per_step_per_gpu": 1,
                      "parallelism": f"replicas x{c.world} (z-slabs / volumes are independent; no collective)",
                      "output_finite": bool(torch.isfinite(out_t[::97]).all().item())}}
    return out, roofline, (lambda: cpu_baseline_simplex(cfg))


def run_detect(c, args, cfg):
    """The product loop around the hot path: the WHOLE detection_B sweep of one image per GPU -- every
    (t_distance, avg) chain of `range(50, end, 50)` x `total_avg` (GaussianDiffusion.py:531-594, "octave" variant: 6-octave simplex
    forward noise, end = 0.6 T) in one slot-batched reverse loop (graph replay, per-slot timesteps, longest chain first), then the
    mean / mse / threshold maps and segmentation counts of every setting on the device.  A step is one whole image; value =
    reverse chain-steps per second.  `--det-end` shortens the sweep (tests); `--batch` is total_avg."""
    import GaussianDiffusion as GD
    from UNet import UNetModel
    navg = args.batch or cfg["batch"]
    end = args.det_end or int(T_STEPS * 0.6)
    torch.manual_seed(1234)
    np.random.seed(1234 + c.rank)
    model = UNetModel(cfg["img"], cfg["base"], channel_mults=cfg["mults"], n_heads=cfg["heads"], attention_resolutions=cfg["attn"])
    fill_weights(model)
    model.to(c.dev).eval()
    diff = GD.GaussianDiffusionModel([cfg["img"]] * 2, GD.get_beta_schedule(T_STEPS, "linear"), noise="simplex")
    x0 = mri_like(1, cfg["img"], c.dev, seed=1234 + c.rank)
    mask = (mri_like(1, cfg["img"], c.dev, seed=99 + c.rank) > 0.2).float()
    dargs = {"arg_num": "bench", "T": int(round(end / 0.6)), "img_size": [cfg["img"]] * 2}      # detection_B: end = int(T * 0.6)

    def step():
        diff.detection_B(model, x0, dargs, ("bench", "image"), mask, denoise_fn="octave", total_avg=navg)
    elapsed = timed(c, args, step)
    ms_per_step = 1000.0 * elapsed / args.steps
    sched = diff.last_chain_schedule
    settings = [r["t_distance"] for r in diff.last_detection]
    chain_steps = sched["chain_steps"]
    value = chain_steps * c.world / (ms_per_step / 1000.0)
    out = {"metric": f"detection_B reverse chain-steps/sec @{cfg['img']}x{cfg['img']} (whole sweep of one image: t_distance "
                     f"{settings[0]}..{settings[-1]} step 50 x {navg} averaged chains + anomaly maps per setting)",
           "value": value, "unit": "chain-steps/s", "ms_per_step": ms_per_step, "scaling": "weak", "dtype": "f32",
           "config": {"workload": cfg["name"], "chains_per_setting": navg, "settings": settings, "chains": len(settings) * navg,
                      "chain_steps_per_image": chain_steps, "slots": sched["slots"], "batched_steps_per_image": sched["steps"],
                      "slot_utilisation": chain_steps / (sched["slots"] * sched["steps"]),
                      "ms_per_image": ms_per_step, "ms_per_chain_step": ms_per_step / chain_steps,
                      "ms_per_batched_step": ms_per_step / sched["steps"],
                      # metric_version 2 (round 5 on): a step is one image's WHOLE sweep and value = chain-steps / s; version 1
                      # (rounds 3-4) timed one setting's batch-of-total_avg chain -- the two are not comparable line to line
                      "metric_version": 2,
                      "parallelism": f"images x{c.world} (one image's sweep per rank, no collective)",
                      "output_finite": bool(all(torch.isfinite(r["mse"]).all().item() for r in diff.last_detection))}}
    # ---- roofline leg: what a batched step of the sweep is made of -- `slots` images through the same plan the sweep ran
    # (per-slot timesteps spread like a running sweep's), one HIP-event pair per op, five instrumented steps
    roofline = None
    if not args.no_prof:
        from anoddpm_amd import _lib
        L = _lib.lib()
        G = sched["slots"]
        xs = mri_like(G, cfg["img"], c.dev, seed=77 + c.rank)
        chain = diff.reverse_chain(model, xs, T_STEPS, "gauss")
        chain.t.copy_(torch.linspace(T_STEPS - 1, 60, G, device=c.dev).long())
        for _ in range(2):
            chain.step()
        nsteps = 5
        L.anoddpm_prof_enable(1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nsteps):
            chain.step()
        torch.cuda.synchronize()
        prof_ms = 1000.0 * (time.perf_counter() - t1) / nsteps
        ms = (ctypes.c_double * _lib.OP_MAX)()
        cnt = (ctypes.c_int64 * _lib.OP_MAX)()
        _lib.check(L.anoddpm_prof_collect(ms, cnt), "prof_collect")
        L.anoddpm_prof_enable(0)
        plan = model._plan_for(G, cfg["img"], c.dev)
        roofline, _, _ = contraction_roofline(plan, ms, cnt, nsteps, prof_ms)
        roofline["batch"] = G
        roofline["note"] = (f"one batched step of the sweep = {G} chain slots through the UNet plan + update; HIP-event pairs over {nsteps} "
                            "instrumented steps after the timed sweep")
        roofline["ms_per_image_step_instrumented"] = sum(roofline["class_ms_per_step"].values()) / G
    return out, roofline, (lambda: {"value": None, "note": "see config c2: the per-step CPU baseline is the same reverse step"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; config det, whose step is a whole image's sweep: 2)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 3; config det: 1)")
    ap.add_argument("--config", default="c2", choices=list(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--det-end", type=int, default=0, help="config det: end of the t_distance sweep range(50, end, 50) (default 0.6 T = 600)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the HIP-event instrumented pass")
    ap.add_argument("--dump-plan", default="", help="write the igemm launch list of the compiled plan (JSON) here")
    ap.add_argument("--dump-layers", default="", help="write the per-layer profile of the instrumented pass (CSV, split-K tails included in their layer) here")
    ap.add_argument("--arith", default="fp32", choices=["fp32", "bf16split3"],
                    help="bf16split3: OPT-IN side line -- the large F(4x4,3x3) layers on split-bf16 products (csrc/winograd43b.hip); "
                         "not the reference's arithmetic class, never the default, reported with its own dtype / arith fields")
    ap.add_argument("--no-extra", action="store_true", help="c2 only: skip the config-3 training step measured after the timed region")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.steps is None:
        args.steps = 2 if args.config == "det" else 20
    if args.warmup is None:
        args.warmup = 1 if args.config == "det" else 3
    if args.arith != "fp32":
        os.environ["ANODDPM_ARITH"] = args.arith              # read by the inference plan when it is built
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    c = setup_dist(args)
    cfg = dict(CONFIGS[args.config])
    run = {"reverse": run_reverse, "train": run_train, "simplex": run_simplex, "detect": run_detect}[cfg["kind"]]
    out, roofline, cpu_fn = run(c, args, cfg)
    line = {"metric": out.pop("metric"), "value": out.pop("value"), "unit": out.pop("unit"), "n_gpus": c.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": out.pop("ms_per_step"), "higher_is_better": True,
            "scaling": out.pop("scaling"), "vs_baseline": None, "dtype": out.pop("dtype"), "data": "synthetic",
            "config": out.pop("config")}
    if args.arith != "fp32":
        line["dtype"] = "f32 with bf16x3-split products on the 128-channel F(4x4,3x3) layers (fp32 accumulate)"
        line["arith"] = args.arith
        line["config"]["side_line"] = ("opt-in arithmetic, NOT the headline: products formed from three bf16 pieces per operand; "
                                       "error table profiles/r4_bf16split3_errors.csv")
    if c.shared:
        # test-only layout (ANODDPM_BENCH_SHARE_GPU=1): the ranks do NOT each own a GPU, so this is not a scaling point
        line["config"]["ranks_share_devices"] = True
    if roofline:
        line["roofline"] = roofline
    if c.rank == 0 and c.world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_fn()
            line["cpu_baseline"].update(host_description())
        except Exception as e:                                   # the baseline must never sink the GPU number
            line["cpu_baseline"] = {"value": None, "error": repr(e)}
    if args.config == "c2" and args.arith == "fp32" and not args.no_extra and os.environ.get("ANODDPM_BENCH_NO_EXTRA", "0") != "1":
        line["extra"] = extra_c3(c, args, line)
    if c.rank == 0:
        print(json.dumps(line), flush=True)
    if c.dist is not None:
        c.dist.destroy_process_group()


def extra_c3(c, args, line):
    """After the timed region of the headline (c2) run: BASELINE config 3's training step on the same ranks (batch 4 per GPU;
    N > 1: the bucketed RCCL all-reduce of the 521 MB gradient, the one data-path collective of the repo), 2 warm-up + 3 timed
    steps, as `extra.c3_ms_per_step` -- so that the driver's N = 1, 2, 4, 8 runs also carry the training step and the first
    real multi-rank RCCL numbers.  It can never cost the headline: any exception is reported in the object, and a watchdog
    thread prints the line without the extra and exits if the leg has not finished within its budget (a hung collective cannot
    be interrupted from Python)."""
    import threading
    budget = float(os.environ.get("ANODDPM_BENCH_EXTRA_TIMEOUT", "300"))
    done = threading.Event()

    def watchdog():
        if not done.wait(budget):
            if c.rank == 0:
                line["extra"] = {"c3_ms_per_step": None, "error": f"config-3 leg did not finish within {budget:.0f} s (abandoned)"}
                print(json.dumps(line), flush=True)
            os._exit(0)
    th = threading.Thread(target=watchdog, daemon=True)
    th.start()
    try:
        a3 = argparse.Namespace(**vars(args))
        a3.steps, a3.warmup, a3.no_prof, a3.batch, a3.config = 3, 2, True, 0, "c3"
        out, _, _ = run_train(c, a3, dict(CONFIGS["c3"]))
        res = {"c3_ms_per_step": out["ms_per_step"], "c3_images_per_s": out["value"], "c3_steps": a3.steps, "c3_warmup": a3.warmup,
               "c3_config": out["config"], "note": "measured after the timed region of this line's metric; not part of `value`"}
    except Exception as e:                                         # noqa: BLE001 -- the extra must never sink the headline
        res = {"c3_ms_per_step": None, "error": repr(e)}
    done.set()
    return res


if __name__ == "__main__":
    main()
