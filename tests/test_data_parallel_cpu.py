"""N>1 path on CPU: two gloo ranks.  The sharded gradient all-reduce must reproduce the single-process
large-batch gradient, shards must tile the batch, and the bench's max-over-ranks timing
reduction must work.  CPU only (the reducer is model-agnostic; HIP kernels are not involved)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from anoddpm_amd.training import FlatBuffers, GradAllReducer, shard_range

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(1, 8, 3, padding=1), torch.nn.SiLU(), torch.nn.Conv2d(8, 8, 3, padding=1),
                               torch.nn.GroupNorm(4, 8), torch.nn.Conv2d(8, 1, 3, padding=1))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(123)
    x = torch.randn(8, 1, 12, 12)
    tgt = torch.randn(8, 1, 12, 12)
    model = _toy()
    flat = FlatBuffers(model)
    red = GradAllReducer(flat, bucket_bytes=256)            # tiny buckets -> several async all-reduces
    assert len(red.buckets) > 2
    lo, hi = shard_range(8, rank, world)
    for _ in range(2):                                      # twice: hooks/buckets must reset correctly
        flat.zero_grad()
        loss = (model(x[lo:hi]) - tgt[lo:hi]).square().mean()
        loss.backward()
        red.finish()
    # timing reduction used by bench.py: max over ranks
    el = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    torch.save({"grad": flat.flat_grad.clone(), "max": el.item(), "range": (lo, hi)}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_large_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    torch.manual_seed(123)
    x = torch.randn(8, 1, 12, 12)
    tgt = torch.randn(8, 1, 12, 12)
    model = _toy()
    flat = FlatBuffers(model)
    (model(x) - tgt).square().mean().backward()
    ref = flat.flat_grad
    for o in outs:
        assert torch.allclose(o["grad"], ref, rtol=1e-5, atol=1e-7)
        assert o["max"] == 2.0
    assert torch.equal(outs[0]["grad"], outs[1]["grad"])            # identical on every rank -> one clip norm
    assert [o["range"] for o in outs] == [(0, 4), (4, 8)]


def _worker_cut_backward(rank, world, port, out_dir):
    """The protocol of the native training plan (train_plan.TrainPlan.run_backward): gradients are written straight into the
    flat buffer (no autograd hooks fire) and every bucket is handed to the reducer as soon as its last gradient is final."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(123)
    x = torch.randn(8, 1, 12, 12)
    tgt = torch.randn(8, 1, 12, 12)
    model = _toy()
    flat = FlatBuffers(model)
    red = GradAllReducer(flat, bucket_bytes=256)
    from anoddpm_amd.training import reducer_of
    assert reducer_of(model) is red                          # what the plan looks up
    lo, hi = shard_range(8, rank, world)
    shadow = _toy()                                          # same weights, plain autograd: stands in for the HIP backward
    (shadow(x[lo:hi]) - tgt[lo:hi]).square().mean().backward()
    grads = [p.grad for p in shadow.parameters()]
    logs = []
    for step in range(2):
        flat.zero_grad()
        order = list(range(len(red.buckets)))                # buckets are in backward order already
        held_back = order[-1] if step == 0 else None         # first step: leave one bucket for finish()
        for i, b in enumerate(order):
            for pi in red.buckets[b][2]:
                flat.params[pi].grad.copy_(grads[pi])        # "kernels" write in place: no post-accumulate hook
            if b != held_back:
                red.launch_bucket(b, ops_done=10 * (i + 1))
                red.launch_bucket(b, ops_done=-1)            # a second call for the same bucket is a no-op
        red.finish()
        logs.append(list(red.last_launch_log))
        # telemetry of the finished step (bench.py --config c3 --gpus N prints it): one enqueue offset per bucket, first one 0,
        # non-decreasing in bucket order; the wait inside finish()
        tm = red.last_timing
        assert tm["buckets"] == len(red.buckets) and len(tm["enqueue_offset_ms"]) == len(red.buckets) and tm["exposed_wait_ms"] >= 0
        assert min(tm["enqueue_offset_ms"]) == 0.0 and len(tm["bucket_MB"]) == len(red.buckets)
    torch.save({"grad": flat.flat_grad.clone(), "logs": logs, "nb": len(red.buckets)}, os.path.join(out_dir, f"c{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_cut_backward_protocol_matches_large_batch(tmp_path):
    world = 2
    mp.spawn(_worker_cut_backward, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"c{r}.pt") for r in range(world)]
    torch.manual_seed(123)
    x = torch.randn(8, 1, 12, 12)
    tgt = torch.randn(8, 1, 12, 12)
    model = _toy()
    flat = FlatBuffers(model)
    (model(x) - tgt).square().mean().backward()
    for o in outs:
        assert torch.allclose(o["grad"], flat.flat_grad, rtol=1e-5, atol=1e-7)
        nb = o["nb"]
        first, second = o["logs"]
        # step 0: every bucket once, the held-back one launched by finish() (ops_done None); step 1: all from the cut backward, in order
        assert sorted(b for b, _ in first) == list(range(nb)) and first[-1] == (nb - 1, None)
        assert [b for b, _ in second] == list(range(nb)) and [d for _, d in second] == [10 * (i + 1) for i in range(nb)]
    assert torch.equal(outs[0]["grad"], outs[1]["grad"])


def test_shard_range_tiles_any_batch():
    for n in (0, 1, 7, 8, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flat_buffers_keep_module_semantics():
    m = _toy()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = FlatBuffers(m)
    assert all(torch.equal(before[k], v) for k, v in m.state_dict().items())
    assert all(p.data_ptr() == flat.flat_param.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets))
    (m(torch.ones(1, 1, 6, 6))).sum().backward()
    assert flat.flat_grad.abs().sum() > 0
    torch.optim.SGD(m.parameters(), lr=0.1).zero_grad(set_to_none=True)
    flat.zero_grad()
    assert all(p.grad is not None for p in flat.params) and flat.flat_grad.abs().sum() == 0
    m.load_state_dict(before)
    assert torch.equal(flat.flat_param[:flat.params[0].numel()].view_as(flat.params[0]), before["0.weight"])


class _Blk(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(4, 4, 3)
        self.embed_layers = torch.nn.ModuleDict({"1": torch.nn.Linear(8, 4)})


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.time_embedding = torch.nn.ModuleDict({"1": torch.nn.Linear(4, 8)})
        self.down = torch.nn.ModuleList([_Blk(), _Blk()])
        self.out = torch.nn.Conv2d(4, 1, 3)


def test_flat_layout_puts_embedding_projections_in_the_last_bucket_and_module_pickles():
    """FlatBuffers stores the timestep MLP / embedding projections at the bottom of the buffer (their gradients are final
    last in the native backward), the reducer cuts buckets from the top down, and attaching a reducer leaves the module
    picklable (the module -> reducer map lives outside the module)."""
    import io
    import pickle
    net = _Net()
    flat = FlatBuffers(net)
    late = [i for i, k in enumerate(flat.names) if "embed_layers." in k or k.startswith("time_embedding.")]
    early = [i for i in range(len(flat.names)) if i not in late]
    assert sorted(flat.layout[:len(late)]) == late and flat.layout[len(late):] == early and len(late) == 6
    # inside the late group the blocks' projection weights are ONE matrix in memory and their biases one vector (round 6: the
    # training plan computes every block's projection of a step with one linear launch)
    ew = [i for i in late if flat.names[i].endswith("embed_layers.1.weight")]
    eb = [i for i in late if flat.names[i].endswith("embed_layers.1.bias")]
    assert len(ew) == 2 and len(eb) == 2
    assert flat.offsets[ew[1]] == flat.offsets[ew[0]] + flat.params[ew[0]].numel()
    assert flat.offsets[eb[0]] == flat.offsets[ew[1]] + flat.params[ew[1]].numel()
    assert flat.offsets[eb[1]] == flat.offsets[eb[0]] + flat.params[eb[0]].numel()
    assert max(flat.offsets[i] for i in late) < min(flat.offsets[i] for i in early)
    assert flat.names == [k for k, _ in net.named_parameters()]              # optimiser state-dict order is the module's
    red = GradAllReducer(flat, bucket_bytes=64)
    assert not red.active and red.hooks == []                                # no process group: inert
    assert red.buckets[0][1] == flat.numel and red.buckets[-1][0] == 0
    assert set(late) <= set(m for b in red.buckets if b[0] < flat.offsets[early[0]] for m in b[2])
    his = [b[1] for b in red.buckets]
    assert his == sorted(his, reverse=True) and all(a[0] == b[1] for a, b in zip(red.buckets, red.buckets[1:]))
    pickle.loads(pickle.dumps(net))
    torch.save(net, io.BytesIO())
    from anoddpm_amd.training import reducer_of
    assert reducer_of(net) is None                                           # inert reducers are not handed to the plan


def _worker8(rank, world, port, out_dir):
    """Eight ranks, batch 32 (config 3's global batch): bucket size from ANODDPM_BUCKET_MB, buckets must tile the flat buffer
    identically on every rank, shards must tile the batch, the reduced gradient must be the large-batch one."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["ANODDPM_BUCKET_MB"] = str(192 / (1 << 20))           # 192 bytes: several buckets on the toy model
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(123)
    x = torch.randn(32, 1, 12, 12)
    tgt = torch.randn(32, 1, 12, 12)
    model = _toy()
    flat = FlatBuffers(model)
    red = GradAllReducer(flat)
    assert red.bucket_bytes == 192 and red.active and red.world == 8
    lo, hi = shard_range(32, rank, world)
    for _ in range(2):
        flat.zero_grad()
        # mean over the shard; the reducer's mean over ranks then equals the mean over the global batch (equal shards)
        (model(x[lo:hi]) - tgt[lo:hi]).square().mean().backward()
        red.finish()
    torch.save({"grad": flat.flat_grad.clone(), "range": (lo, hi), "bounds": red.bounds, "launched": red.launched},
               os.path.join(out_dir, f"e{rank}.pt"))
    dist.destroy_process_group()


def test_eight_rank_gradient_allreduce_bucket_tiling_and_shards(tmp_path):
    """the first real 8-GPU run must not fail on plumbing -- 8 gloo ranks, B = 32."""
    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"e{r}.pt") for r in range(world)]
    torch.manual_seed(123)
    x = torch.randn(32, 1, 12, 12)
    tgt = torch.randn(32, 1, 12, 12)
    model = _toy()
    flat = FlatBuffers(model)
    (model(x) - tgt).square().mean().backward()
    assert [o["range"] for o in outs] == [(4 * r, 4 * r + 4) for r in range(world)]
    bounds = outs[0]["bounds"]
    assert len(bounds) > 2 and bounds[0][1] == flat.numel and bounds[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(bounds, bounds[1:]))               # buckets tile the flat buffer, top down
    for o in outs:
        assert o["bounds"] == bounds and o["launched"] == len(bounds)
        assert torch.allclose(o["grad"], flat.flat_grad, rtol=1e-5, atol=1e-7)
        assert torch.equal(o["grad"], outs[0]["grad"])                          # identical everywhere -> one clip factor


def test_bucket_size_env_and_validation(monkeypatch):
    net = _Net()
    flat = FlatBuffers(net)
    monkeypatch.setenv("ANODDPM_BUCKET_MB", "0.5")
    assert GradAllReducer(flat).bucket_bytes == 512 * 1024
    monkeypatch.delenv("ANODDPM_BUCKET_MB")
    assert GradAllReducer(flat).bucket_bytes == 64 << 20
    with pytest.raises(ValueError):
        GradAllReducer(flat, bucket_bytes=0)
