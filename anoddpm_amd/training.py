"""Data-parallel training step for the AnoDDPM denoiser -- the reference's loop body
(diffusion_training.py:99-107: p_loss -> zero_grad -> backward -> clip_grad_norm_(1) -> AdamW -> EMA)
re-designed for one process per MI355X with a single collective per step.

The reference has no distributed code at all; what is built here:

* `FlatBuffers`   parameters, gradients, Adam moments and the EMA copy each live in ONE contiguous fp32
                  buffer (parameters / grads are views into it), so the optimizer is one kernel launch
                  and the gradient all-reduce needs no packing copies.
* `GradAllReducer` bucketed all-reduce (sum, then 1/world) of the flat gradient over RCCL
                  (`torch.distributed`, backend "nccl" on ROCm; "gloo" in the CPU tests).  Buckets are cut
                  in reverse parameter order -- the up path's gradients are ready first -- and each bucket
                  is launched from an autograd post-accumulate hook as soon as it is complete, so the
                  reduction overlaps the rest of backward.  On an 8-GPU xGMI node every pair is directly
                  linked; bucket size defaults to 64 MiB (8 buckets for the 521 MB gradient): large enough
                  that the per-link ring time dominates launch latency, small enough to overlap.
* `FusedAdamWEMA`  clip-by-global-norm factor computed on the device from `anoddpm_sumsq`, then ONE
                  `anoddpm_adamw_ema` launch: decoupled weight decay, bias-corrected moments, EMA.
                  Clipping sees the REDUCED gradient, i.e. the single-process semantics of :103-105.
* `train_step`     the six calls of the reference loop body in order.

Inference needs none of this: it shards the batch with `shard_range` and never communicates.
"""
import ctypes

import torch

from . import _lib
from ._lib import AdamwArgs, check, current_stream, lib, ptr

__all__ = ["shard_range", "FlatBuffers", "GradAllReducer", "FusedAdamWEMA", "train_step", "reducer_of"]

# module -> weak reference to the GradAllReducer working on its flat gradient buffer.  Kept OUTSIDE the module (a weakref stored as
# a module attribute would break torch.save(module) / pickle for every nn.Module that is not UNetModel); the native training plan
# looks its reducer up here to cut the backward at the bucket boundaries.
import weakref

_REDUCERS = weakref.WeakKeyDictionary()


def reducer_of(module):
    """The live, ACTIVE GradAllReducer attached to `module`'s flat buffers, or None."""
    ref = _REDUCERS.get(module)
    red = ref() if ref is not None else None
    return red if (red is not None and red.active) else None


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of a batch of n_items for `rank` (remainder spread over low ranks)."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatBuffers:
    """Re-homes a module's parameters (and their .grad) into contiguous fp32 buffers.

    Parameter i occupies flat[offset_i : offset_i + numel_i]; `param.data` and `param.grad` become views,
    so state_dict / load_state_dict / checkpoints keep working unchanged.

    `names` / `params` keep the module's parameter order (the order torch.optim.AdamW's state_dict indexes by); the STORAGE
    order (`layout`, ascending offsets) puts the parameters whose gradients become final last in the backward first: the
    timestep MLP and every block's embedding projection (`late`, by default a name test).  The gradient buckets of the
    data-parallel reducer are cut from the top of the buffer down, so those parameters share the LAST bucket(s) and the native
    backward can finish all embedding projections in one batched launch at its very end without holding any other bucket back."""

    @staticmethod
    def _late(name):
        return "embed_layers." in name or name.startswith("time_embedding.")

    def __init__(self, module, late=None):
        self.module = module          # owner: told when a raw kernel rewrites the flat buffer (mark_weights_changed)
        self.names = [k for k, p in module.named_parameters() if p.requires_grad]
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        assert dt == torch.float32 and all(p.device == dev and p.dtype == dt for p in self.params)
        late = late or self._late
        idx = range(len(self.params))
        # inside the late group: the timestep MLP, then every block's embedding-projection WEIGHT, then every projection's BIAS (module
        # order each) -- the concatenated weights / biases are then one [sum cout][ted] matrix and one vector in memory, and the
        # training plan computes all projections of a step with ONE linear launch (train_plan: `emb_all`), as the inference plan does
        # on its packed copy
        def rank(i):
            k = self.names[i]
            return (1 if k.endswith("embed_layers.1.weight") else 2 if k.endswith("embed_layers.1.bias") else 0) if late(k) else 3
        self.layout = sorted(idx, key=lambda i: (rank(i), i))
        self.offsets = [0] * len(self.params)
        n = 0
        for i in self.layout:
            self.offsets[i] = n
            n += (self.params[i].numel() + 3) // 4 * 4   # keep every view 16-byte aligned
        self.numel = n
        self.flat_param = torch.zeros(n, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(n, device=dev, dtype=dt)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                self.flat_param[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat_param[o:o + p.numel()].view(p.shape)
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)

    def bind_grads(self):
        """Re-attach any `.grad` that is not the flat view (optimiser.zero_grad(set_to_none=True) drops them); a re-attached
        slice is zeroed.  Returns the number of re-attached gradients."""
        n = 0
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                v = self.flat_grad[o:o + p.numel()].view(p.shape)
                v.zero_()
                p.grad = v
                n += 1
        return n

    def zero_grad(self):
        self.flat_grad.zero_()
        self.bind_grads()


class GradAllReducer:
    """Bucketed, backward-overlapped all-reduce of `FlatBuffers.flat_grad` (mean over ranks).

    Active (hooks registered, the native backward cut at the bucket boundaries) when a process group with more than one
    rank exists; at world size 1 every all-reduce would be a no-op that still costs its launches, so the reducer stays
    inert unless `force=True` (tests / `ANODDPM_BENCH_FORCE_DIST=1` exercise the RCCL path on one GPU that way).
    Backend "gloo" with device tensors (two test ranks sharing one GPU -- RCCL cannot do that) stages each bucket through
    host memory: correct, slow, test-only."""

    def __init__(self, flat, process_group=None, bucket_bytes=None, force=False):
        import os
        import torch.distributed as dist
        if bucket_bytes is None:
            # 64 MiB unless ANODDPM_BUCKET_MB says otherwise (the first real 8-GPU runs tune it without a code change)
            bucket_bytes = int(float(os.environ.get("ANODDPM_BUCKET_MB", "64")) * (1 << 20))
        if bucket_bytes < 4:
            raise ValueError(f"GradAllReducer: bucket_bytes must be >= 4, got {bucket_bytes}")
        self.bucket_bytes = int(bucket_bytes)
        self.dist = dist
        self.flat = flat
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.staged = bool(dist.is_initialized() and dist.get_backend(process_group) == "gloo" and flat.flat_grad.is_cuda)
        # buckets from the TOP of the flat buffer down: storage order is roughly forward order (FlatBuffers.layout), so the
        # gradients of the highest offsets are final first in the backward
        self.buckets = []            # (lo, hi, [param indices])
        cap = max(1, bucket_bytes // 4)
        hi, members = None, []
        order = list(flat.layout)
        for pos in reversed(range(len(order))):
            i = order[pos]
            o = flat.offsets[i]
            end = flat.offsets[order[pos + 1]] if pos + 1 < len(order) else flat.numel
            if hi is None:
                hi = end
            members.append(i)
            if hi - o >= cap or pos == 0:
                self.buckets.append((o, hi, members))
                hi, members = None, []
        self.bounds = tuple((lo, hi) for lo, hi, _ in self.buckets)
        self.bucket_of = {}
        for b, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self.bucket_of[i] = b
        self.pending = [0] * len(self.buckets)
        self.works = []
        self.hooks = []
        self.launched = 0            # buckets reduced by the last finish() (tests)
        self.launch_log = []         # (bucket, backward ops already enqueued or None) of the step in flight
        self.last_launch_log = []    # ... of the last finished step (tests)
        # host-clock telemetry of the last finished step (the first real multi-GPU run should be able to tune
        # ANODDPM_BUCKET_MB from one bench line): when each bucket's all-reduce was ENQUEUED relative to the first one, and how
        # long finish() then WAITED for the collectives -- the exposed (not hidden behind the backward) part of the all-reduce
        self._t_first = None
        self._enqueue_ms = []
        self.last_timing = {"buckets": len(self.buckets), "bucket_MB": [4.0 * (hi - lo) / (1 << 20) for lo, hi, _ in self.buckets],
                            "enqueue_offset_ms": [], "exposed_wait_ms": None, "host_timed": True}
        self.timing_sync = False     # bench.py sets it: finish() then synchronises the stream so that exposed_wait_ms is device time
        self.active = bool(dist.is_initialized() and (self.world > 1 or force))
        if getattr(flat, "module", None) is not None:
            _REDUCERS[flat.module] = weakref.ref(self)        # the native training plan cuts its backward at our buckets
        if self.active:
            for i, p in enumerate(flat.params):
                self.hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def reset(self):
        self.pending = [len(m) for (_, _, m) in self.buckets]
        self.works = []

    def _make_hook(self, i):
        def hook(param):
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def launch_bucket(self, b, ops_done=None):
        """Called by the native training plan (train_plan.TrainPlan.run_backward) once every gradient of bucket `b` is final."""
        if self.pending[b] > 0:
            self.pending[b] = 0
            self.launch_log.append((b, ops_done))
            self._launch(b)

    def _launch(self, b):
        import time
        now = time.perf_counter()
        if self._t_first is None:
            self._t_first = now
        self._enqueue_ms.append((b, 1000.0 * (now - self._t_first)))
        lo, hi, _ = self.buckets[b]
        view = self.flat.flat_grad[lo:hi]
        if self.staged:
            host = view.cpu()                                  # stream-ordered copy + host sync: the bucket's kernels are done
            self.works.append((self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True), view, host))
        else:
            self.works.append((self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True), view, None))

    def finish(self):
        """Wait for every bucket (launching any the hooks did not see) and turn sums into means."""
        if not self.active:
            return
        for b, left in enumerate(self.pending):
            if left > 0:                       # parameter unused this step: its grad is zero, still reduce
                self.pending[b] = 0
                self.launch_log.append((b, None))
                self._launch(b)
        import time
        t0 = time.perf_counter()
        for work, view, host in self.works:
            work.wait()
            if host is not None:
                view.copy_(host)
        if self.flat.flat_grad.is_cuda and not self.staged:
            # work.wait() only orders the current stream behind the collective: the host-side figure needs the device to get there
            torch.cuda.current_stream(self.flat.flat_grad.device).synchronize() if self.timing_sync else None
        self.last_timing = dict(self.last_timing, enqueue_offset_ms=[ms for _, ms in sorted(self._enqueue_ms)],
                                exposed_wait_ms=1000.0 * (time.perf_counter() - t0), synced=bool(self.timing_sync))
        self._t_first, self._enqueue_ms = None, []
        self.launched = len(self.works)
        self.last_launch_log, self.launch_log = self.launch_log, []
        if self.world > 1:
            self.flat.flat_grad.mul_(1.0 / self.world)
        self.reset()


class FusedAdamWEMA:
    """AdamW(lr, betas, eps, weight_decay) + EMA(decay) + clip_grad_norm_(max_norm) in two launches."""

    def __init__(self, flat, ema_flat=None, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 ema_decay=0.9999, max_norm=1.0, notify=()):
        _lib.require_cuda(flat.flat_param, "FusedAdamWEMA")
        self.flat, self.ema_flat = flat, ema_flat
        if ema_flat is not None and (ema_flat.numel != flat.numel or ema_flat.offsets != flat.offsets):
            # the kernel walks both flat buffers with ONE index: a differently laid-out EMA copy (e.g. frozen
            # parameters on one side) would be updated at the wrong offsets or out of bounds
            raise ValueError("FusedAdamWEMA: the EMA module's flat layout differs from the trained module's")
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.ema_decay, self.max_norm = ema_decay, max_norm
        # modules whose packed-weight caches must be refreshed after the raw kernel wrote their parameters behind
        # autograd's back: the owners of both flat buffers, plus anything the caller adds
        self.notify = []
        for mod in [flat.module, ema_flat.module if ema_flat is not None else None, *notify]:
            if mod is not None and hasattr(mod, "mark_weights_changed") and all(mod is not m for m in self.notify):
                self.notify.append(mod)
        self.m = torch.zeros_like(flat.flat_param)
        self.v = torch.zeros_like(flat.flat_param)
        self.step_count = 0
        self.norm_out = torch.zeros(3, device=flat.flat_param.device)       # {sum of squares, norm, clip factor}
        self.norm_ws = torch.zeros(2048, device=flat.flat_param.device, dtype=torch.float64)
        self.last_norm = None

    def step(self):
        f = self.flat
        stream = current_stream()
        scale_ptr = None
        if self.max_norm is not None:
            # norm + clip_grad_norm_'s factor on the device (deterministic two-stage reduction: identical on every replica)
            check(lib().anoddpm_sumsq(ptr(f.flat_grad), f.numel, ptr(self.norm_out), ptr(self.norm_ws), float(self.max_norm), stream),
                  "sumsq")
            self.last_norm = self.norm_out[1:2]
            scale_ptr = self.norm_out.data_ptr() + 8
        self.step_count += 1
        a = AdamwArgs()
        a.p, a.m, a.v, a.g = f.flat_param.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), f.flat_grad.data_ptr()
        a.ema = self.ema_flat.flat_param.data_ptr() if self.ema_flat is not None else None
        a.grad_scale = scale_ptr
        a.n = f.numel
        a.lr, a.beta1, a.beta2, a.eps = self.lr, self.betas[0], self.betas[1], self.eps
        a.weight_decay, a.ema_decay, a.step = self.wd, self.ema_decay, self.step_count
        check(lib().anoddpm_adamw_ema(ctypes.byref(a), stream), "adamw_ema")
        for mod in self.notify:                 # the raw kernel bypasses autograd's version counters
            mod.mark_weights_changed()
        return self.last_norm

    # -- checkpoint format of the reference: torch.optim.AdamW's state_dict (diffusion_training.py:77,173,184) --------
    def state_dict(self):
        """The layout `torch.optim.AdamW.state_dict()` produces for the same parameter list: per-parameter `step`,
        `exp_avg`, `exp_avg_sq` sliced out of the flat moment buffers, one param group."""
        f = self.flat
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(f.params, f.offsets)):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.m[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.v[o:o + n].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": True, "params": list(range(len(f.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts a `torch.optim.AdamW` state_dict (e.g. `resume["optimizer_state_dict"]` of a reference checkpoint)."""
        f = self.flat
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(f.params):
            raise ValueError("FusedAdamWEMA.load_state_dict: expected one param group covering every parameter")
        g = groups[0]
        if g.get("amsgrad") or g.get("maximize"):
            raise ValueError("FusedAdamWEMA.load_state_dict: amsgrad / maximize are not supported")
        self.lr, self.betas, self.eps, self.wd = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        self.m.zero_()
        self.v.zero_()
        steps = set()
        for idx, pid in enumerate(g["params"]):
            st = sd["state"].get(pid)
            if st is None:
                continue
            p, o = f.params[idx], f.offsets[idx]
            n = p.numel()
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"FusedAdamWEMA.load_state_dict: moment shape mismatch for parameter {idx}")
            self.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FusedAdamWEMA.load_state_dict: parameters carry different step counts")
        self.step_count = steps.pop() if steps else 0


def train_step(model, diffusion, x, args, flat, reducer, optim):
    """One optimiser step = the body of diffusion_training.py:99-107 on this rank's shard `x`."""
    loss, estimates = diffusion.p_loss(model, x, args)
    flat.zero_grad()
    loss.backward()
    if reducer is not None:
        reducer.finish()                        # gradients are now the global mean on every rank
    optim.step()                                # clip (global norm of the reduced gradient) + AdamW + EMA
    return loss.detach(), estimates
