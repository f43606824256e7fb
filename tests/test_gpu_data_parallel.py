"""-m gpu: the N > 1 training path on ONE MI355X -- two processes share cuda:0 and talk over gloo (RCCL cannot place two
ranks on one device; training.GradAllReducer stages gloo buckets through host memory).  Everything else is the real thing:
UNetModel on the native training plan, the backward cut at the gradient-bucket boundaries, GradAllReducer, FusedAdamWEMA.
SURVEY 8e: shards of the injected (x_0, t, noise) must reproduce the single-process large-batch step."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KW = dict(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev):
    from UNet import UNetModel
    from oracle import unet_oracle as uo
    m = UNetModel(**KW)
    sd = uo.perturb(uo.fill_deterministic({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    m.load_state_dict(sd)
    return m.to(dev).train()


def _step(model, flat, red, opt, x0, t, noise):
    import GaussianDiffusion as GD
    diff = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(1000, "linear"), loss_type="l2", noise="gauss")
    diff.noise_fn = lambda a, b: noise
    loss_d, x_t, eps = diff.calc_loss(model, x0, t)
    loss = loss_d["loss"].mean()
    flat.zero_grad()
    loss.backward()
    if red is not None:
        red.finish()
    grad = flat.flat_grad.clone()
    norm = opt.step()
    return loss.item(), grad, norm.item()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from anoddpm_amd.training import FlatBuffers, FusedAdamWEMA, GradAllReducer, reducer_of, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    data = torch.load(os.path.join(out_dir, "data.pt"))
    model = _build(DEV)
    flat = FlatBuffers(model)
    red = GradAllReducer(flat, bucket_bytes=1 << 18)
    assert red.active and red.staged and reducer_of(model) is red and len(red.buckets) > 4
    opt = FusedAdamWEMA(flat, None, lr=1e-3)
    lo, hi = shard_range(data["x0"].shape[0], rank, world)
    x0, t, noise = (data[k][lo:hi].to(DEV) for k in ("x0", "t", "noise"))
    outs = []
    for step in range(2):                                   # twice: the second step re-uses the plan and the cached schedule
        if step == 1:
            for p in model.parameters():                    # a caller's optimiser.zero_grad(set_to_none=True): re-bound before the forward
                p.grad = None
        loss, grad, norm = _step(model, flat, red, opt, x0, t, noise)
        plan = next(iter(model._tplans.values()))
        outs.append(dict(loss=loss, grad=grad.cpu(), norm=norm, param=flat.flat_param.clone().cpu(),
                         log=list(red.last_launch_log), nbops=len(plan.bops), nbuckets=len(red.buckets)))
    assert len(model._tplans) == 1
    torch.save(outs, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_native_cut_backward_match_single_process_large_batch(tmp_path):
    from anoddpm_amd.training import FlatBuffers, FusedAdamWEMA
    g = torch.Generator().manual_seed(11)
    data = {"x0": torch.rand(4, 1, 32, 32, generator=g) * 2 - 1, "t": torch.tensor([3, 250, 640, 999]),
            "noise": torch.randn(4, 1, 32, 32, generator=g)}
    torch.save(data, tmp_path / "data.pt")
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    # single process, the whole batch, no reducer
    model = _build(DEV)
    flat = FlatBuffers(model)
    opt = FusedAdamWEMA(flat, None, lr=1e-3)
    x0, t, noise = (data[k].to(DEV) for k in ("x0", "t", "noise"))
    for step in range(2):
        loss, grad, norm = _step(model, flat, None, opt, x0, t, noise)
        grad, param = grad.cpu(), flat.flat_param.clone().cpu()
        gmax = grad.abs().max().item()
        for o in (outs[0][step], outs[1][step]):
            assert (o["grad"] - grad).abs().max().item() < 2e-4 * gmax, (step, (o["grad"] - grad).abs().max().item(), gmax)
            assert abs(o["norm"] - norm) < 2e-4 * norm
            # the first AdamW steps move every parameter by ~lr * sign(g) = 1e-3: elements whose gradient is rounding noise may
            # differ by a fraction of that, everything else agrees closely
            dp = (o["param"] - param).abs()
            assert dp.max().item() < 5e-4 and dp.mean().item() < 2e-6, (dp.max().item(), dp.mean().item())
            # every bucket was launched from inside the cut backward, in bucket order, the first well before the end
            done = [d for _, d in o["log"]]
            assert [b for b, _ in o["log"]] == list(range(o["nbuckets"])) and all(d is not None for d in done) and done == sorted(done)
            assert done[0] < o["nbops"] // 2 and done[-1] <= o["nbops"]
        assert torch.equal(outs[0][step]["grad"], outs[1][step]["grad"])   # identical reduced gradient -> identical clip + update
        assert torch.equal(outs[0][step]["param"], outs[1][step]["param"])
        assert abs(0.5 * (outs[0][step]["loss"] + outs[1][step]["loss"]) - loss) < 1e-5 * max(abs(loss), 1.0)


def _bench(*extra, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", *extra], cwd=ROOT,
                          capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))


def test_bench_gpus_flag_spawns_ranks_or_fails_loudly():
    """`python bench.py --gpus N` launches N ranks itself; on a box with fewer devices it errors instead of printing n_gpus: 1."""
    ndev = torch.cuda.device_count()
    out = _bench("--gpus", str(ndev + 1), "--config", "c1", "--no-prof")
    assert out.returncode != 0 and "only" in (out.stderr + out.stdout) and not any(ln.startswith("{") for ln in out.stdout.splitlines())
    # two ranks sharing the visible device(s) over gloo (test layout, flagged in the line): reverse chain shards + training step
    for cfg, extra in (("c1", []), ("c3", ["--batch", "1"])):
        out = _bench("--gpus", "2", "--config", cfg, "--no-prof", *extra, env={"ANODDPM_BENCH_SHARE_GPU": "1"} if ndev < 2 else None)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak" and "cpu_baseline" not in d
        assert d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"]
        assert d["config"].get("ranks_share_devices", False) == (ndev < 2)


def test_bench_eight_ranks_on_shared_devices():
    """the driver's first 8-GPU run must not fail on plumbing.  `bench.py --gpus 8` here: eight ranks over the
    visible device(s) (rank r uses device r % ndev), gloo -- reverse-chain shards (c1) and the config-3 training step with its
    bucketed gradient all-reduce at batch 1 per rank.  One line, n_gpus 8, global batch = 8 x per-GPU batch."""
    ndev = torch.cuda.device_count()
    env = {"ANODDPM_BENCH_SHARE_GPU": "1", "ANODDPM_BUCKET_MB": "128"} if ndev < 8 else {"ANODDPM_BUCKET_MB": "128"}
    for cfg, extra in (("c1", []), ("c3", ["--batch", "1"])):
        out = _bench("--gpus", "8", "--config", cfg, "--no-prof", *extra, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 8 and d["value"] > 0 and d["scaling"] == "weak" and "cpu_baseline" not in d
        assert d["config"]["global_batch"] == 8 * d["config"]["per_gpu_batch"]
        assert d["config"].get("ranks_share_devices", False) == (ndev < 8)
        if cfg == "c3":
            assert d["config"]["loss_finite"] is True and "data-parallel x8" in d["config"]["parallelism"]
