"""`GaussianDiffusionModel` & friends -- the reference diffusion process (GaussianDiffusion.py)
driven from PyTorch-ROCm host code with the arithmetic in fused HIP kernels.

Kept from the reference (names, signatures, return structures, RNG consumption order):
  get_beta_schedule, extract, mean_flat, normal_kl, approx_standard_normal_cdf,
  discretised_gaussian_log_likelihood, generate_simplex_noise, random_noise,
  GaussianDiffusionModel.{sample_t_with_weights, predict_x_0_from_eps, predict_eps_from_x_0,
  q_mean_variance, q_posterior_mean_variance, p_mean_variance, sample_p, forward_backward,
  sample_q, sample_q_gradual, calc_vlb_xt, calc_loss, p_loss, prior_vlb, calc_total_vlb,
  detection_A_fixedT} plus the north-star aliases q_sample / p_sample_loop.

What changed underneath (MI355X-first):
  * the 14 fp64 schedule tables are built exactly as GaussianDiffusion.py:184-217 and uploaded
    ONCE per device as fp32 (the reference uploads an 8 KB table 8x per step, :32-36);
  * sample_q is one kernel (12 B/pixel), the whole reverse update of sample_p -- predict x0, clamp,
    posterior mean, sigma*noise -- is one kernel (16 B/pixel) instead of ~25 dispatches;
  * simplex noise is generated on the GPU straight into the noise tensor (no D2H of t, no numba,
    no H2D), seeded from the global numpy stream exactly like Simplex_CLASS.newSeed();
  * forward_backward keeps `t` on the device and, for the built-in UNetModel, replays one captured
    HIP graph per reverse step.

All tensors must live on a HIP device: there is no CPU fallback (CPU tensors raise AnoddpmError).
"""
import ctypes
import random

import numpy as np
import torch

from . import _lib
from ._lib import LossArgs, PUpdateArgs, VlbArgs, check, current_stream, lib, ptr
from .simplex import Simplex_CLASS, perm_tables

__all__ = ["SimplexNoiseFn", "ReverseChain", "get_beta_schedule", "extract", "mean_flat", "normal_kl", "approx_standard_normal_cdf",
           "discretised_gaussian_log_likelihood", "generate_simplex_noise", "random_noise",
           "GaussianDiffusionModel"]

_RANDOM_PARAMS = [(2, 0.6, 16), (6, 0.6, 32), (7, 0.7, 32), (10, 0.8, 64), (5, 0.8, 16), (4, 0.6, 16),
                  (1, 0.6, 64), (7, 0.8, 128), (6, 0.9, 64), (2, 0.85, 128), (2, 0.85, 64), (2, 0.85, 32),
                  (2, 0.85, 16), (2, 0.85, 8), (2, 0.85, 4), (2, 0.85, 2), (1, 0.85, 128), (1, 0.85, 64),
                  (1, 0.85, 32), (1, 0.85, 16), (1, 0.85, 8), (1, 0.85, 4), (1, 0.85, 2)]


def get_beta_schedule(num_diffusion_steps, name="cosine"):
    """GaussianDiffusion.py:12-29 (host, fp64)."""
    if name == "cosine":
        def abar(u):
            return np.cos((u + 0.008) / 1.008 * np.pi / 2) ** 2
        n = num_diffusion_steps
        return np.array([min(1 - abar((i + 1) / n) / abar(i / n), 0.999) for i in range(n)])
    if name == "linear":
        k = 1000 / num_diffusion_steps
        return np.linspace(k * 0.0001, k * 0.02, num_diffusion_steps, dtype=np.float64)
    raise NotImplementedError(f"unknown beta schedule: {name}")


def extract(arr, timesteps, broadcast_shape, device):
    """GaussianDiffusion.py:32-36: gather in fp64, then cast to fp32, broadcast to `broadcast_shape`."""
    res = torch.from_numpy(np.asarray(arr)).to(device=timesteps.device)[timesteps].float()
    return res.reshape(res.shape + (1,) * (len(broadcast_shape) - res.dim())).expand(broadcast_shape).to(device)


def mean_flat(tensor):
    """GaussianDiffusion.py:39-40: mean over everything but the batch dimension."""
    if tensor.dim() < 2:
        return torch.mean(tensor, dim=[])                        # what the reference's empty dim list does
    return tensor.reshape(tensor.shape[0], -1).mean(dim=1)


def normal_kl(mean1, logvar1, mean2, logvar2):
    """KL(N(mean1, e^logvar1) || N(mean2, e^logvar2)) in nats, elementwise (GaussianDiffusion.py:43-53).  API surface only: the
    training / logging paths evaluate this inside anoddpm_vlb_terms / anoddpm_loss_forward."""
    dl = logvar2 - logvar1
    return 0.5 * (dl - 1.0 + torch.exp(-dl) + torch.exp(-logvar2) * (mean1 - mean2) ** 2)


_SQRT_2_OVER_PI = float(np.sqrt(2.0 / np.pi))


def approx_standard_normal_cdf(x):
    """tanh approximation of the standard normal CDF (GaussianDiffusion.py:56-61)."""
    return 0.5 * (1.0 + torch.tanh(_SQRT_2_OVER_PI * (x + 0.044715 * x * x * x)))


def discretised_gaussian_log_likelihood(x, means, log_scales):
    """log-likelihood of a Gaussian discretised to 8-bit bins of width 2/255 on [-1, 1], open-ended at +-1
    (GaussianDiffusion.py:64-93).  API surface only (see normal_kl)."""
    if not (x.shape == means.shape == log_scales.shape):
        raise AssertionError("discretised_gaussian_log_likelihood: shape mismatch")
    z = torch.exp(-log_scales)
    upper = approx_standard_normal_cdf(z * (x - means + 1.0 / 255.0))
    lower = approx_standard_normal_cdf(z * (x - means - 1.0 / 255.0))
    floor = 1e-12
    inner = torch.log((upper - lower).clamp(min=floor))
    left = torch.log(upper.clamp(min=floor))                     # x at the lower edge: everything below counts
    right = torch.log((1.0 - lower).clamp(min=floor))            # x at the upper edge: everything above counts
    return torch.where(x < -0.999, left, torch.where(x > 0.999, right, inner))


def generate_simplex_noise(Simplex_instance, x, t, random_param=False, octave=6, persistence=0.8, frequency=64,
                           in_channels=1):
    """GaussianDiffusion.py:96-137 on the device.

    Per channel: a fresh seed from the global numpy stream (newSeed), then the multi-octave field at
    z = t written as fp32 into noise[:, i].  `random_param=True` draws `random.choice` like the
    reference does; the reference then overwrites that field with the default-parameter one
    (:125-136 has no else), so only the RNG draw is observable and only it is reproduced.
    Batch > 1: sample b gets the field at z = t[b] (all equal inside forward_backward, which is the
    reference's `.repeat(B,1,1,1)` behaviour); the reference itself only runs at batch 1.
    """
    _lib.require_cuda(x, "generate_simplex_noise")
    noise = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    for i in range(in_channels):
        Simplex_instance.newSeed()
        if random_param:
            random.choice(_RANDOM_PARAMS)
        Simplex_instance.fill_fixed_T_octaves_(noise, t, octave, persistence, frequency, channel=i)
    return noise


def random_noise(Simplex_instance, x, t):
    """GaussianDiffusion.py:140-147."""
    if random.choice(["gauss", "simplex"]) == "gauss":
        return torch.randn_like(x)
    return generate_simplex_noise(Simplex_instance, x, t)


class SimplexNoiseFn:
    """A `denoise_fn` / `noise_fn` callable equivalent to
    `lambda x, t: generate_simplex_noise(simplex, x, t, False, octave, persistence, frequency, in_channels)`
    that the reverse chain can recognise: its seeds are then drawn up front (same numpy-stream order,
    one per step per channel) and the permutation tables of all steps are uploaded once."""

    def __init__(self, simplex, octave=6, persistence=0.8, frequency=64, in_channels=1):
        self.simplex, self.octave, self.persistence, self.frequency, self.in_channels = \
            simplex, octave, persistence, frequency, in_channels

    def __call__(self, x, t):
        return generate_simplex_noise(self.simplex, x, t, False, self.octave, self.persistence, self.frequency,
                                      self.in_channels)


def plan_chain_slots(lengths, slots):
    """Longest-first list schedule of reverse chains on `slots` chain slots (host logic of `_run_chains`).

    lengths[i] > 0: steps of chain i.  Returns (makespan, [(slot, first_step), ...] per chain): a chain occupies one slot for
    `lengths[i]` consecutive global steps; a freed slot takes the longest pending chain.  makespan <= ceil(sum / slots) + max - 1
    (Graham's bound for list scheduling), and with the detection sweeps' lengths the slots end within one short chain of each other."""
    import heapq
    if slots < 1:
        raise ValueError("plan_chain_slots: slots must be >= 1")
    if any(int(l) <= 0 for l in lengths):
        raise ValueError("plan_chain_slots: chain lengths must be positive")
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    free = [(0, s) for s in range(slots)]
    heapq.heapify(free)
    place = [None] * len(lengths)
    makespan = 0
    for i in order:
        t0, slot = heapq.heappop(free)
        place[i] = (slot, t0)
        t1 = t0 + int(lengths[i])
        makespan = max(makespan, t1)
        heapq.heappush(free, (t1, slot))
    return makespan, place


class ReverseChain:
    """Device-resident state of the reverse loop (GaussianDiffusion.py:351-357): x, t and a step counter
    live in HBM; one step = model forward + noise + ONE fused update launch + t -= 1."""

    def __init__(self, owner, model, x, t_distance, denoise_fn, use_graph=None):
        _lib.require_cuda(x, "ReverseChain")
        if not 0 <= int(t_distance) <= owner.num_timesteps:
            # extract() of the reference indexes the T-entry tables with t_distance - 1 and raises for anything else
            raise IndexError(f"t_distance {t_distance} is out of range for a {owner.num_timesteps}-step schedule")
        self.owner, self.model, self.denoise_fn = owner, model, denoise_fn
        import os
        self.graph = None
        self._graph_state = 0          # 0: next step eager (warm-up), 1: capture, 2: replay
        self.B = x.shape[0]
        self.x = owner._f32(x.detach()).clone()
        self.t = torch.full((self.B,), t_distance - 1, device=x.device, dtype=torch.int64)
        self.step_idx = torch.zeros(1, device=x.device, dtype=torch.int32)
        self.remaining = int(t_distance)
        # a train()-mode model with dropout draws a fresh mask per forward from the host generator (UNet.py:192): it goes
        # through model.forward (eager), not through the captured inference plan
        self.hip_model = hasattr(model, "forward_hip") and not self._draws_dropout(model)
        self._plan = None              # the inference plan a captured graph points into (set at capture)
        self.noise = None
        self.tables = None
        # Which noise sources may run inside a captured HIP graph: device-side philox draws ("gauss" / "random":
        # torch.randn_like registers its generator state with the capture) and SimplexNoiseFn, whose per-step seeds are
        # drawn up front and whose tables are selected by the device-resident step counter.  Anything else -- a plain
        # callable, a user-replaced `noise_fn`, the randParam / random mixtures -- draws from numpy / `random` on the
        # host every step (newSeed(), GaussianDiffusion.py:102) and uploads a table: captured once, every replay
        # would reuse the first step's seed.  Those run eagerly.
        fn, capture_safe = self._resolve_noise(owner, denoise_fn)
        self._last_seed = None
        self.reuse_key = None
        if isinstance(fn, SimplexNoiseFn):
            self.simplex_fn = fn
            self.noise = torch.empty_like(self.x)
            # room for the permutation tables of a full-length chain, so that reset() can start another chain on the same buffers
            self.tables = torch.empty((max(self.remaining, owner.num_timesteps) * fn.in_channels, 512), dtype=torch.int16, device=x.device)
            self._draw_tables(self.remaining)
        self.reuse_key = self._reuse_key_of(owner, denoise_fn)
        self.capture_safe = capture_safe
        # HIP-graph replay of the step: on by default for the built-in UNetModel (every launch of a step is
        # stream-ordered, allocation-free C-ABI work) with a capture-safe noise source; ANODDPM_NO_GRAPH=1 forces
        # eager launches.  An explicit use_graph=True with an unsafe noise source is refused, not silently wrong.
        if use_graph is None:
            use_graph = self.hip_model and capture_safe and os.environ.get("ANODDPM_NO_GRAPH", "0") != "1"
        elif use_graph and not capture_safe:
            raise ValueError("ReverseChain(use_graph=True): this denoise_fn draws host-side random numbers every step "
                             "and cannot be replayed from a captured graph; pass a SimplexNoiseFn or 'gauss'")
        elif use_graph and not self.hip_model:
            # a train()-mode dropout model seeds its masks on the host per forward: captured once, every replay would repeat
            # the first step's mask; a foreign callable's launches are not known to be capture-safe at all
            raise ValueError("ReverseChain(use_graph=True): the model is not the built-in UNetModel in a dropout-free "
                             "mode (model.eval(), or dropout == 0); its forward cannot be replayed from a captured graph")
        self.use_graph = bool(use_graph)

    @staticmethod
    def _draws_dropout(model):
        return bool(getattr(model, "training", False) and getattr(model, "dropout", 0) > 0)

    @staticmethod
    def _resolve_noise(owner, denoise_fn):
        """-> (SimplexNoiseFn or the original denoise_fn, capture_safe): how the reverse loop's noise request is served."""
        fn = denoise_fn
        capture_safe = False
        if type(fn) == str:
            if fn in ("gauss", "random"):
                capture_safe = True
            elif fn == "noise_fn":
                nf = owner.noise_fn
                if isinstance(nf, SimplexNoiseFn):
                    fn = nf
                elif nf is getattr(owner, "_default_noise_fn", None) and owner.noise_kind == "gauss":
                    capture_safe = True                                      # torch.randn_like
                elif nf is getattr(owner, "_default_noise_fn", None) and owner.noise_kind not in ("simplex_randParam", "random"):
                    fn = SimplexNoiseFn(owner.simplex, in_channels=owner.img_channels)      # :179-181 defaults
            else:
                fn = SimplexNoiseFn(owner.simplex, in_channels=owner.img_channels)          # :310 defaults
        if isinstance(fn, SimplexNoiseFn):
            capture_safe = True
        return fn, capture_safe

    @staticmethod
    def _reuse_key_of(owner, denoise_fn):
        """Key under which a graph-replaying chain for this noise request may be restarted with reset() (None: never)."""
        fn, capture_safe = ReverseChain._resolve_noise(owner, denoise_fn)
        if isinstance(fn, SimplexNoiseFn):
            return ("simplex", id(fn.simplex), fn.octave, fn.persistence, fn.frequency, fn.in_channels)
        return ("gauss",) if capture_safe else None

    def _draw_tables(self, nsteps):
        """Every seed of the chain, drawn now in the order the per-step newSeed() calls would (numpy global stream), and the
        permutation tables uploaded in one copy."""
        import time
        t0 = time.perf_counter()
        fn = self.simplex_fn
        n = nsteps * fn.in_channels
        tabs = np.empty((n, 512), dtype=np.int16)
        self._last_seed = None
        for i in range(n):
            seed = np.random.randint(-10000000000, 10000000000)
            tabs[i] = perm_tables(seed)
            self._last_seed = seed
        if n:
            self.tables[:n].copy_(torch.from_numpy(tabs))
        # host time of the whole chain's newSeed() work, paid before the first step (upstream pays one newSeed() per step):
        # bench.py reports it beside the per-step time
        self.table_setup_ms = 1000.0 * (time.perf_counter() - t0)
        self.table_setup_steps = nsteps

    def reset(self, x, t_distance):
        """Start another chain of the same batch shape on this chain's device buffers: the captured HIP graph (and the plan behind
        it) is reused instead of being built again -- the detection loops run hundreds of chains of one shape."""
        if tuple(x.shape) != tuple(self.x.shape) or x.device != self.x.device:
            raise ValueError("ReverseChain.reset: shape / device differ from the chain's buffers")
        if not 0 <= int(t_distance) <= self.owner.num_timesteps:
            raise IndexError(f"t_distance {t_distance} is out of range for a {self.owner.num_timesteps}-step schedule")
        self.x.copy_(self.owner._f32(x.detach()))
        self.t.fill_(int(t_distance) - 1)
        self.step_idx.zero_()
        self.remaining = int(t_distance)
        if self.tables is not None:
            self._draw_tables(self.remaining)
        if self.hip_model and self._graph_state == 2:
            # the replayed graph reads the packed weights of the plan it was captured with (NOT whatever _plan_for would pick
            # now -- ANODDPM_ARITH may have changed since): let THAT plan re-pack them if the parameters moved since
            self._plan.refresh_weights()
        return self

    def step(self):
        if self.use_graph and lib().anoddpm_prof_active() == 0:
            if self._graph_state == 2:
                self.graph.replay()
                self.remaining -= 1
                return self.x
            if self._graph_state == 1:
                # everything a step touches is static (x, t, step counter, noise, tables, plan buffers)
                torch.cuda.synchronize()
                if self.hip_model:
                    self._plan = self.model._plan_for(self.B, self.x.shape[2], self.x.device)
                g = torch.cuda.CUDAGraph()
                # No cyclic garbage collection INSIDE the capture (round 6: a sporadic `Fatal Python error: Aborted` with the
                # interpreter "Garbage-collecting" under _step_body, about one GPU test run in five): the collector may pick that
                # moment to destroy an older chain's captured graph / plan buffers left unreachable by an earlier caller, and HIP
                # API calls made by those destructors are illegal while a stream capture is open in global mode.  Collect first
                # (twice: C++ owners release further Python cycles), then keep the collector off until the capture has ended.
                import gc
                gc.collect()
                gc.collect()
                gc_was_enabled = gc.isenabled()
                gc.disable()
                try:
                    with torch.cuda.graph(g):
                        self._step_body()
                finally:
                    if gc_was_enabled:
                        gc.enable()
                self.graph = g
                self._graph_state = 2
                g.replay()                      # capture does not execute: run the captured step once
                self.remaining -= 1
                return self.x
            self._graph_state = 1               # first step runs eagerly (builds the plan, packs weights)
        self._step_body()
        self.remaining -= 1
        return self.x

    def _step_body(self):
        o = self.owner

        with torch.no_grad():
            # (forking the step's simplex field -- it depends on t and the step counter only -- onto a second stream beside the
            # denoiser's latency-bound timestep MLP was measured: the captured graph gains a second branch and replays 0.14 ms
            # SLOWER, 9.52 vs 9.38 ms per step)
            eps = self.model.forward_hip(self.x, self.t, borrow=True) if self.hip_model else self.model(self.x, self.t)
            if self.tables is not None:
                fn = self.simplex_fn
                for c in range(fn.in_channels):
                    fn.simplex.fill_fixed_T_octaves_(self.noise, self.t, fn.octave, fn.persistence, fn.frequency,
                                                     channel=c, tables=self.tables[c:], table_sel=self.step_idx,
                                                     table_sel_scale=fn.in_channels)
                noise = self.noise
            else:
                noise = o._denoise_noise(self.x, self.t, self.denoise_fn)
            o._reverse_update(self.x, self.t, eps, noise, want_pred=False, out=self.x)     # in place
            check(lib().anoddpm_chain_advance(ptr(self.t), self.B, ptr(self.step_idx), current_stream()), "chain_advance")

    def finish(self):
        if self.tables is not None and self._last_seed is not None:
            self.simplex_fn.simplex.newSeed(self._last_seed)       # leave the generator where upstream would


class _DeviceTables:
    """fp32 device copies of the schedule tables (one upload per device, not eight per step)."""

    NAMES = ("sqrt_alphas", "sqrt_betas", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
             "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
             "posterior_mean_coef2", "model_variance", "model_log_variance", "sigma")

    def __init__(self, owner, device):
        def up(a):
            # extract() gathers in fp64 and casts the gathered value: casting the table is the same thing
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).float().to(device)
        for n in self.NAMES[:8]:
            setattr(self, n, up(getattr(owner, n)))
        model_var = np.append(owner.posterior_variance[1], owner.betas[1:])       # :282-283
        model_logvar = np.log(model_var)
        self.model_variance = up(model_var)
        self.model_log_variance = up(model_logvar)
        self.posterior_log_variance_clipped = up(owner.posterior_log_variance_clipped)
        # exp(0.5*log_variance) evaluated by the same fp32 torch expression the reference uses (:317)
        self.sigma = torch.exp(0.5 * torch.from_numpy(model_logvar).float()).to(device)


class _FusedLoss(torch.autograd.Function):
    """(per-sample loss [B], vlb [B], weighted batch mean []) of calc_loss / p_loss from the model output, and in backward the
    gradient with respect to the model output -- anoddpm_loss_forward / anoddpm_loss_backward.  kind: 0 l1, 1 l2, 2 hybrid,
    3 = the VLB term alone (calc_vlb_xt).  Only `eps` is differentiable."""

    @staticmethod
    def _args(owner, eps, noise, x0, xt, t, weights, kind):
        tb = owner._tables(eps.device)
        a = LossArgs()
        a.eps, a.noise = ptr(eps), ptr(noise if noise is not None else eps)
        a.weights = ptr(weights) if weights is not None else None
        if kind >= 2:
            a.x0, a.xt, a.t = ptr(x0), ptr(xt), ptr(t)
            a.c_recip, a.c_recipm1 = ptr(tb.sqrt_recip_alphas_cumprod), ptr(tb.sqrt_recipm1_alphas_cumprod)
            a.c_coef1, a.c_coef2 = ptr(tb.posterior_mean_coef1), ptr(tb.posterior_mean_coef2)
            a.c_post_logvar, a.c_model_logvar = ptr(tb.posterior_log_variance_clipped), ptr(tb.model_log_variance)
        B = eps.shape[0]
        a.n, a.B, a.T, a.kind = (eps[0].numel() if B else 0), B, owner.num_timesteps, min(kind, 2)
        return a

    @staticmethod
    def forward(ctx, eps, noise, x0, xt, t, weights, owner, kind):
        _lib.require_cuda(eps, "GaussianDiffusionModel loss")
        f32 = owner._f32
        e = f32(eps.detach())
        nz = f32(noise.detach()) if noise is not None else None
        x0c, xtc, tt = (f32(x0.detach()), f32(xt.detach()), owner._t64(t, e.device)) if kind >= 2 else (None, None, None)
        w = f32(weights.detach().to(e.device)) if weights is not None else None
        B = e.shape[0]
        out = torch.empty((2 * B + 1,), dtype=torch.float32, device=e.device)
        ws = torch.empty((max(B, 1) * 64 * 2,), dtype=torch.float64, device=e.device)
        a = _FusedLoss._args(owner, e, nz, x0c, xtc, tt, w, kind)
        a.per_sample, a.vlb, a.total = ptr(out[:B]), ptr(out[B:2 * B]), ptr(out[2 * B:])
        a.workspace, a.workspace_doubles = ptr(ws), ws.numel()
        if B:
            check(lib().anoddpm_loss_forward(ctypes.byref(a), current_stream()), "loss_forward")
        ctx.owner, ctx.kind, ctx.saved = owner, kind, (e, nz, x0c, xtc, tt, w)
        ctx.set_materialize_grads(False)
        per, vlb, total = out[:B], out[B:2 * B], out[2 * B]
        if kind == 3:
            # the VLB term alone: the fused kernel's per-sample value is vlb + mse(eps, eps) = vlb
            return per, vlb, total
        return per, vlb, total

    @staticmethod
    def backward(ctx, g_per, g_vlb, g_total):
        e, nz, x0c, xtc, tt, w = ctx.saved
        kind = ctx.kind
        a = _FusedLoss._args(ctx.owner, e, nz, x0c, xtc, tt, w, kind)
        keep = []

        def dev(g):
            if g is None:
                return None
            g = g.detach().float().contiguous()
            keep.append(g)
            return ptr(g)
        if kind == 3:
            # per == vlb here (no main term): fold both upstream gradients into g_vlb, and leave the main-term coefficient at zero
            gv = g_vlb if g_per is None else (g_per if g_vlb is None else g_per + g_vlb)
            a.g_per, a.g_vlb, a.g_total = None, dev(gv), None
            if g_total is not None:
                raise _lib.AnoddpmError("calc_vlb_xt: only the per-sample output is differentiable")
        else:
            a.g_per, a.g_vlb, a.g_total = dev(g_per), dev(g_vlb), dev(g_total)
        d = torch.empty_like(e)
        a.d_eps = ptr(d)
        if e.shape[0]:
            check(lib().anoddpm_loss_backward(ctypes.byref(a), current_stream()), "loss_backward")
        return d, None, None, None, None, None, None, None


class GaussianDiffusionModel:
    def __init__(self, img_size, betas, img_channels=1, loss_type="l2", loss_weight='none', noise="gauss"):
        super().__init__()
        if noise == "gauss":
            self.noise_fn = lambda x, t: torch.randn_like(x)
        else:
            self.simplex = Simplex_CLASS()
            if noise == "simplex_randParam":
                self.noise_fn = lambda x, t: generate_simplex_noise(self.simplex, x, t, True, in_channels=img_channels)
            elif noise == "random":
                self.noise_fn = lambda x, t: random_noise(self.simplex, x, t)
            else:
                self.noise_fn = lambda x, t: generate_simplex_noise(self.simplex, x, t, False, in_channels=img_channels)
        self.noise_kind = noise
        self._default_noise_fn = self.noise_fn      # lets the reverse chain tell a user-replaced noise_fn from this one

        self.img_size = img_size
        self.img_channels = img_channels
        self.loss_type = loss_type
        self.num_timesteps = len(betas)

        if loss_weight == 'prop-t':
            self.weights = np.arange(self.num_timesteps, 0, -1)
        elif loss_weight == "uniform":
            self.weights = np.ones(self.num_timesteps)
        self.loss_weight = loss_weight

        # fp64 host tables, GaussianDiffusion.py:184-217
        betas = np.asarray(betas, dtype=np.float64)
        alphas = 1 - betas
        self.betas = betas
        self.sqrt_alphas = np.sqrt(alphas)
        self.sqrt_betas = np.sqrt(betas)
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self._dev = {}

    # ------------------------------------------------------------------ device plumbing
    def _tables(self, device):
        tb = self._dev.get(device)
        if tb is None:
            tb = self._dev[device] = _DeviceTables(self, device)
        return tb

    @staticmethod
    def _t64(t, device):
        if t.dtype != torch.int64 or t.device != device or not t.is_contiguous():
            t = t.to(device=device, dtype=torch.int64).contiguous()
        return t

    @staticmethod
    def _f32(x):
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        return x

    def _axpby(self, ca, cb, x, t, noise):
        _lib.require_cuda(x, "GaussianDiffusionModel.sample_q")
        x = self._f32(x.detach() if not x.requires_grad else x)
        noise = self._f32(noise).to(x.device)
        t = self._t64(t, x.device)
        out = torch.empty_like(x)
        B = x.shape[0]
        check(lib().anoddpm_q_sample(ptr(out), ptr(x), ptr(noise), ptr(t), ptr(ca), ptr(cb), B,
                                     x.numel() // max(B, 1), self.num_timesteps, current_stream()), "q_sample")
        return out

    def _reverse_update(self, x_t, t, eps, noise, want_pred=True, want_mean=False, out=None):
        """One fused launch for GaussianDiffusion.py:287-288 + :314-317."""
        _lib.require_cuda(x_t, "GaussianDiffusionModel.sample_p")
        tb = self._tables(x_t.device)
        x_t, eps = self._f32(x_t.detach()), self._f32(eps.detach())
        if noise is not None:
            noise = self._f32(noise.detach())
        t = self._t64(t, x_t.device)
        a = PUpdateArgs()
        x_prev = out if out is not None else torch.empty_like(x_t)
        pred = torch.empty_like(x_t) if want_pred else None
        mean = torch.empty_like(x_t) if want_mean else None
        a.x_prev, a.pred_x0, a.mean_out = x_prev.data_ptr(), pred.data_ptr() if want_pred else None, mean.data_ptr() if want_mean else None
        a.x_t, a.eps, a.noise, a.t = x_t.data_ptr(), eps.data_ptr(), noise.data_ptr() if noise is not None else None, t.data_ptr()
        a.c_recip, a.c_recipm1 = tb.sqrt_recip_alphas_cumprod.data_ptr(), tb.sqrt_recipm1_alphas_cumprod.data_ptr()
        a.c_coef1, a.c_coef2, a.c_sigma = tb.posterior_mean_coef1.data_ptr(), tb.posterior_mean_coef2.data_ptr(), tb.sigma.data_ptr()
        a.B, a.T = x_t.shape[0], self.num_timesteps
        a.n = x_t.numel() // max(x_t.shape[0], 1)
        check(lib().anoddpm_p_sample_update(ctypes.byref(a), current_stream()), "p_sample_update")
        return x_prev, pred, mean

    # ------------------------------------------------------------------ reference API
    def sample_t_with_weights(self, b_size, device):
        """GaussianDiffusion.py:220-226: timesteps drawn with probability proportional to `self.weights` from the global numpy
        stream, plus the importance weights p[t] / T (fp64 product, then fp32)."""
        prob = self.weights / np.sum(self.weights)
        drawn = np.random.choice(prob.size, size=b_size, p=prob)
        return (torch.from_numpy(drawn).long().to(device),
                torch.from_numpy(1 / prob.size * prob[drawn]).float().to(device))

    def _at(self, table, t, like):
        """extract() on a host table: gather in fp64, cast, shape [B,1,...] broadcastable against `like`."""
        return extract(table, t, like.shape, like.device)

    def predict_x_0_from_eps(self, x_t, t, eps):
        """GaussianDiffusion.py:228-230 (differentiable API form; the sampling path fuses it into anoddpm_p_sample_update)."""
        a, b = self._at(self.sqrt_recip_alphas_cumprod, t, x_t), self._at(self.sqrt_recipm1_alphas_cumprod, t, x_t)
        return a * x_t - b * eps

    def predict_eps_from_x_0(self, x_t, t, pred_x_0):
        """GaussianDiffusion.py:232-235."""
        a, b = self._at(self.sqrt_recip_alphas_cumprod, t, x_t), self._at(self.sqrt_recipm1_alphas_cumprod, t, x_t)
        return (a * x_t - pred_x_0) / b

    def q_mean_variance(self, x_0, t):
        """q(x_t | x_0): mean, variance, log-variance (GaussianDiffusion.py:237-251)."""
        return (self._at(self.sqrt_alphas_cumprod, t, x_0) * x_0, self._at(1.0 - self.alphas_cumprod, t, x_0),
                self._at(self.log_one_minus_alphas_cumprod, t, x_0))

    def q_posterior_mean_variance(self, x_0, x_t, t):
        """q(x_{t-1} | x_t, x_0): mean, variance, clipped log-variance (GaussianDiffusion.py:253-267)."""
        c1, c2 = self._at(self.posterior_mean_coef1, t, x_t), self._at(self.posterior_mean_coef2, t, x_t)
        return (c1 * x_0 + c2 * x_t, self._at(self.posterior_variance, t, x_t),
                self._at(self.posterior_log_variance_clipped, t, x_t))

    def p_mean_variance(self, model, x_t, t, estimate_noise=None):
        """GaussianDiffusion.py:269-296; mean / pred_x_0 come from the fused update kernel."""
        if estimate_noise is None:
            estimate_noise = model(x_t, t)
        if torch.is_grad_enabled() and (x_t.requires_grad or estimate_noise.requires_grad):
            # differentiable API form (the training losses do not come through here: anoddpm_loss_backward carries the VLB
            # term's gradient); fixed-large variance of :282-283
            fixed_var = np.append(self.posterior_variance[1], self.betas[1:])
            pred_x_0 = torch.clamp(self.predict_x_0_from_eps(x_t, t, estimate_noise), -1, 1)
            return {"mean": self.q_posterior_mean_variance(pred_x_0, x_t, t)[0], "variance": self._at(fixed_var, t, x_t),
                    "log_variance": self._at(np.log(fixed_var), t, x_t), "pred_x_0": pred_x_0}
        tb = self._tables(x_t.device)
        tt = self._t64(t, x_t.device)
        _, pred, mean = self._reverse_update(x_t, tt, estimate_noise, None, want_pred=True, want_mean=True)
        shape = x_t.shape
        view = (-1,) + (1,) * (len(shape) - 1)
        return {"mean": mean, "variance": tb.model_variance[tt].view(view).expand(shape),
                "log_variance": tb.model_log_variance[tt].view(view).expand(shape), "pred_x_0": pred}

    def _denoise_noise(self, x_t, t, denoise_fn):
        """Noise selection of sample_p (GaussianDiffusion.py:301-312)."""
        if type(denoise_fn) == str:
            if denoise_fn == "gauss":
                return torch.randn_like(x_t)
            if denoise_fn == "noise_fn":
                return self.noise_fn(x_t, t).float()
            if denoise_fn == "random":
                return torch.randn_like(x_t)
            return generate_simplex_noise(self.simplex, x_t, t, False, in_channels=self.img_channels).float()
        return denoise_fn(x_t, t)

    def sample_p(self, model, x_t, t, denoise_fn="gauss"):
        """One reverse step (GaussianDiffusion.py:298-318): model -> noise -> ONE fused update launch."""
        eps = model(x_t, t)
        noise = self._denoise_noise(x_t, t, denoise_fn)
        sample, pred, _ = self._reverse_update(x_t, t, eps, noise)
        return {"sample": sample, "pred_x_0": pred}

    def forward_backward(self, model, x, see_whole_sequence="half", t_distance=None, denoise_fn="gauss"):
        """GaussianDiffusion.py:320-359.  `t` lives on the device for the whole chain."""
        assert see_whole_sequence == "whole" or see_whole_sequence == "half" or see_whole_sequence == None

        if t_distance == 0:
            return x.detach()
        if t_distance is None:
            t_distance = self.num_timesteps
        _lib.require_cuda(x, "GaussianDiffusionModel.forward_backward")
        seq = [x.cpu().detach()]
        B = x.shape[0]
        if see_whole_sequence == "whole":
            for t in range(int(t_distance)):
                t_batch = torch.full((B,), t, device=x.device, dtype=torch.int64)
                noise = self.noise_fn(x, t_batch).float()
                with torch.no_grad():
                    x = self.sample_q_gradual(x, t_batch, noise)
                seq.append(x.cpu().detach())
        else:
            t_tensor = torch.full((B,), t_distance - 1, device=x.device, dtype=torch.int64)
            x = self.sample_q(x, t_tensor, self.noise_fn(x, t_tensor).float())
            if see_whole_sequence == "half":
                seq.append(x.cpu().detach())

        with torch.no_grad():
            x = self._reverse_chain(model, x, int(t_distance), denoise_fn, seq if see_whole_sequence else None)
        return x.detach() if not see_whole_sequence else seq

    p_sample_loop = forward_backward          # north-star alias

    def _chain_for(self, model, x, t_distance, denoise_fn):
        """A ReverseChain for (model, batch shape, noise source): chains that replay a captured graph are kept -- at most eight,
        oldest dropped first; they hold their model -- and restarted with reset() (same device buffers, same graph); anything
        else is built fresh."""
        cache = self.__dict__.setdefault("_chains", {})
        want = ReverseChain._reuse_key_of(self, denoise_fn)
        if want is not None:
            key = (id(model), tuple(x.shape), str(x.device), want)
            chain = cache.get(key)
            # a kept chain replays the dropout-free inference graph: not for a model that has since been put in train() mode
            # with dropout > 0 (it gets a fresh eager chain below, which is not kept)
            if chain is not None and chain.model is model and not ReverseChain._draws_dropout(model):
                return chain.reset(x, t_distance)
        chain = ReverseChain(self, model, x, t_distance, denoise_fn)
        if chain.use_graph and chain.reuse_key is not None:
            if len(cache) >= 8:
                if x.is_cuda:
                    torch.cuda.synchronize(x.device)              # its captured graph may still be replaying (see _run_chains)
                cache.pop(next(iter(cache)))
            cache[(id(model), tuple(x.shape), str(x.device), chain.reuse_key)] = chain
        return chain

    def _reverse_chain(self, model, x, t_distance, denoise_fn, seq):
        """t = t_distance-1 ... 0 of sample_p with a device-resident timestep (no per-step H2D)."""
        chain = self._chain_for(model, x, t_distance, denoise_fn)
        for _ in range(t_distance):
            chain.step()
            if seq is not None:
                seq.append(chain.x.cpu().detach())
        chain.finish()
        # a kept chain's buffer is overwritten by the next reset(): hand the caller its own tensor (the reference returns a fresh
        # one from every call, GaussianDiffusion.py:351-359); an un-kept chain's buffer dies with the chain, no copy needed
        return chain.x.clone() if chain in self.__dict__.get("_chains", {}).values() else chain.x

    def release_chains(self):
        """Drop every kept ReverseChain (captured HIP graph + the device buffers and plan it holds: about 3.4 GB per chain at
        config 2); the next forward_backward / detection call builds and captures a fresh one."""
        if self.__dict__.get("_chains") and torch.cuda.is_available():
            torch.cuda.synchronize()                               # no captured graph may be destroyed while a replay is in flight
        self.__dict__.pop("_chains", None)

    def __getstate__(self):
        """Pickle / deepcopy without the device-side caches: kept chains hold CUDAGraph objects, `_dev` holds device tensors."""
        d = dict(self.__dict__)
        d.pop("_chains", None)
        d["_dev"] = {}
        return d

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def reverse_chain(self, model, x_t, t_distance, denoise_fn="gauss"):
        """Stepping form of the reverse loop of forward_backward (:351-357): returns a ReverseChain whose
        .step() performs one sample_p on the device-resident state (bench.py times K of these)."""
        return ReverseChain(self, model, x_t, int(t_distance), denoise_fn)

    def sample_q(self, x_0, t, noise):
        """q(x_t | x_0), GaussianDiffusion.py:361-371 -- one fused launch."""
        if torch.is_grad_enabled() and (x_0.requires_grad or noise.requires_grad):
            return self._at(self.sqrt_alphas_cumprod, t, x_0) * x_0 + self._at(self.sqrt_one_minus_alphas_cumprod, t, x_0) * noise
        _lib.require_cuda(x_0, "GaussianDiffusionModel.sample_q")
        tb = self._tables(x_0.device)
        return self._axpby(tb.sqrt_alphas_cumprod, tb.sqrt_one_minus_alphas_cumprod, x_0, t, noise)

    q_sample = sample_q                       # north-star alias

    def sample_q_gradual(self, x_t, t, noise):
        """q(x_t | x_{t-1}), GaussianDiffusion.py:373-382."""
        _lib.require_cuda(x_t, "GaussianDiffusionModel.sample_q_gradual")
        tb = self._tables(x_t.device)
        return self._axpby(tb.sqrt_alphas, tb.sqrt_betas, x_t, t, noise)

    def vlb_terms(self, x_0, x_t, t, eps, noise=None, want_pred=True):
        """Fused variational-bound terms of one step (anoddpm_vlb_terms): returns (vlb [B] in bits/dim,
        mean((pred_x_0-x_0)^2) [B], mean((eps'-noise)^2) [B], pred_x_0 or None).  No autograd."""
        _lib.require_cuda(x_t, "GaussianDiffusionModel.vlb_terms")
        tb = self._tables(x_t.device)
        B = x_t.shape[0]
        x_0, x_t, eps = self._f32(x_0), self._f32(x_t), self._f32(eps)
        noise = self._f32(noise) if noise is not None else None
        tt = self._t64(t, x_t.device)
        out = torch.empty((3, B), dtype=torch.float32, device=x_t.device)
        ws = torch.empty((64 * B * 3,), dtype=torch.float64, device=x_t.device)
        pred = torch.empty_like(x_t) if want_pred else None
        a = VlbArgs()
        a.x0, a.xt, a.eps, a.noise, a.t = ptr(x_0), ptr(x_t), ptr(eps), (ptr(noise) if noise is not None else None), ptr(tt)
        a.c_recip, a.c_recipm1 = ptr(tb.sqrt_recip_alphas_cumprod), ptr(tb.sqrt_recipm1_alphas_cumprod)
        a.c_coef1, a.c_coef2 = ptr(tb.posterior_mean_coef1), ptr(tb.posterior_mean_coef2)
        a.c_post_logvar, a.c_model_logvar = ptr(tb.posterior_log_variance_clipped), ptr(tb.model_log_variance)
        a.pred_x0 = ptr(pred) if pred is not None else None
        a.out, a.workspace, a.workspace_doubles = ptr(out), ptr(ws), ws.numel()
        a.n, a.B, a.T = (x_t[0].numel() if B else 0), B, self.num_timesteps
        check(lib().anoddpm_vlb_terms(ctypes.byref(a), current_stream()), "vlb_terms")
        return out[0], out[1], out[2], pred

    def calc_vlb_xt(self, model, x_0, x_t, t, estimate_noise=None):
        """GaussianDiffusion.py:384-397: KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) or, at t = 0, the discretised decoder NLL, in
        bits per dimension -- one fused launch after the model call (anoddpm_vlb_terms).  With autograd recording the term is
        differentiable with respect to the model output (anoddpm_loss_backward); `pred_x_0` is returned detached."""
        eps = model(x_t, t) if estimate_noise is None else estimate_noise
        if torch.is_grad_enabled() and eps.requires_grad:
            _, vlb, _ = _FusedLoss.apply(eps, None, x_0, x_t, t, None, self, 3)
            with torch.no_grad():
                pred = self._reverse_update(x_t, t, eps, None, want_pred=True)[1]
            return {"output": vlb, "pred_x_0": pred}
        vlb, _, _, pred = self.vlb_terms(x_0, x_t, t, eps)
        return {"output": vlb, "pred_x_0": pred}

    _LOSS_KINDS = {"l1": 0, "l2": 1, "hybrid": 2}

    def _loss_terms(self, model, x_0, t, weights=None):
        """calc_loss + the weighted batch mean of p_loss: noise draw, q-sample, model call, then ONE fused launch (+ fold) for the
        per-sample terms and the scalar; its autograd backward is one more launch producing d(loss)/d(eps)."""
        noise = self.noise_fn(x_0, t).float()
        x_t = self.sample_q(x_0, t, noise)
        eps = model(x_t, t)
        kind = self._LOSS_KINDS.get(self.loss_type, 1)           # unknown strings mean l2, as upstream's final else (:415-416)
        per, vlb, total = _FusedLoss.apply(eps, noise, x_0, x_t, t, weights, self, kind)
        terms = {"loss": per}
        if kind == 2:
            terms = {"vlb": vlb, "loss": per}
        return terms, x_t, eps, total

    def calc_loss(self, model, x_0, t):
        """GaussianDiffusion.py:399-417 -> (loss dict, x_t, estimate_noise)."""
        terms, x_t, eps, _ = self._loss_terms(model, x_0, t)
        return terms, x_t, eps

    def p_loss(self, model, x_0, args):
        """GaussianDiffusion.py:419-434: draws t (torch.randint, or the weighted numpy draw), returns
        (mean over the batch of loss * weights, (loss dict, x_t, estimate_noise))."""
        B = x_0.shape[0]
        if self.loss_weight == "none":
            hi = min(args["sample_distance"], self.num_timesteps) if args["train_start"] else self.num_timesteps
            t, weights = torch.randint(0, hi, (B,), device=x_0.device), None
        else:
            t, weights = self.sample_t_with_weights(B, x_0.device)
        terms, x_t, eps, total = self._loss_terms(model, x_0, t, weights)
        return total, (terms, x_t, eps)

    def prior_vlb(self, x_0, args):
        """GaussianDiffusion.py:436-443: KL(q(x_T | x_0) || N(0, I)) in bits per dimension."""
        t = torch.full((args["Batch_Size"],), self.num_timesteps - 1, device=x_0.device, dtype=torch.int64)
        mean, _, logvar = self.q_mean_variance(x_0, t)
        zero = torch.zeros((), device=x_0.device)
        return mean_flat(normal_kl(mean, logvar, zero, zero)) / np.log(2.0)

    def calc_total_vlb(self, x_0, model, args):
        """GaussianDiffusion.py:445-478: T model calls; everything after each call (KL / decoder NLL, the two MSE
        curves) is one fused launch, and `t` is the only host-built tensor per step."""
        vb, x_0_mse, mse = [], [], []
        for t in reversed(list(range(self.num_timesteps))):
            t_batch = torch.tensor([t] * args["Batch_Size"], device=x_0.device)
            noise = torch.randn_like(x_0)
            x_t = self.sample_q(x_0=x_0, t=t_batch, noise=noise)
            with torch.no_grad():
                eps = model(x_t, t_batch)
                v, m0, me, _ = self.vlb_terms(x_0, x_t, t_batch, eps, noise=noise, want_pred=False)
            vb.append(v)
            x_0_mse.append(m0)
            mse.append(me)
        vb = torch.stack(vb, dim=1)
        x_0_mse = torch.stack(x_0_mse, dim=1)
        mse = torch.stack(mse, dim=1)
        prior_vlb = self.prior_vlb(x_0, args)
        total_vlb = vb.sum(dim=1) + prior_vlb
        return {"total_vlb": total_vlb, "prior_vlb": prior_vlb, "vb": vb, "x_0_mse": x_0_mse, "mse": mse}

    # ------------------------------------------------------------------ detection compute loops
    def detection_A_fixedT(self, model, x_0, args, mask, end_freq=6):
        """GaussianDiffusion.py:596-623 (pure compute; no file output upstream either)."""
        t_distance = 250
        output = torch.empty((6 * end_freq, 1, *args["img_size"]), device=x_0.device)
        for i in range(1, end_freq + 1):
            freq = 2 ** i
            # == lambda x, t: generate_simplex_noise(self.simplex, x, t, False, frequency=freq).float()   (:602), as an
            # object the reverse chain recognises (per-step seeds pre-drawn in the same numpy-stream order)
            noise_fn = SimplexNoiseFn(self.simplex, frequency=freq)
            t_tensor = torch.tensor([t_distance - 1], device=x_0.device).repeat(x_0.shape[0])
            x = self.sample_q(x_0, t_tensor, noise_fn(x_0, t_tensor).float())
            x_noised = x.clone().detach()
            with torch.no_grad():
                x = self._reverse_chain(model, x, t_distance, noise_fn, None)
            mse = ((x_0 - x).square() * 2) - 1
            mse_threshold = mse > 0
            mse_threshold = (mse_threshold.float() * 2) - 1
            output[(i - 1) * 6:i * 6, ...] = torch.cat((x_0, x_noised, x, mse, mse_threshold, mask))
        return output

    # The (t_distance, avg) loops of detection_A / detection_B.  Upstream runs every chain of one image one after the other
    # (:499-514, :554-569).  The chains are independent of each other -- every (t_distance, avg) pair, and in detection_A every
    # frequency too -- so ALL chains of a call share one batched reverse loop: `slots` device-resident chain slots
    # with a per-slot timestep (the UNet and the fused update take per-sample `t`), stepped together by ONE captured graph; a slot
    # whose chain reaches t = 0 hands its image over and starts the next pending chain (longest first, so the slots drain together).
    # The batch size is therefore constant (the quantisation-free sizes of DESIGN 10-6: 16 by default) whatever `total_avg` and the
    # t_distance sweep are.  The mean / mse / threshold images and the segmentation counts of a setting come from one fused pass
    # (metrics.anomaly_maps).  RNG: the same generators are consumed (np.random for simplex seeds, torch's for randn), the forward
    # noise in upstream's order, but the reverse-step draws of different chains interleave differently from the serial loop, so
    # outputs are equal in distribution, not sample-for-sample.
    def _avg_chains(self, model, x_0, t_distance, total_avg):
        """One setting: `total_avg` chains of the same length as one batch (all slots start and end together)."""
        _lib.require_cuda(x_0, "GaussianDiffusionModel.detection")
        if x_0.shape[0] != 1:
            raise ValueError("detection loops take one image (upstream stores each chain into output[avg], :514)")
        t_tensor = torch.full((1,), int(t_distance), device=x_0.device, dtype=torch.int64)
        noise = torch.cat([self.noise_fn(x_0, t_tensor).float() for _ in range(total_avg)])
        x = self.sample_q(x_0.repeat(total_avg, 1, 1, 1), t_tensor.repeat(total_avg), noise)
        with torch.no_grad():
            return self._reverse_chain(model, x, int(t_distance), "gauss", None)      # sample_p default noise, :508

    def _forward_noise(self, x_0, t_distance, n):
        """`n` draws of the forward noise of one setting, in upstream's order (:501-505, :556-560)."""
        t_tensor = torch.full((1,), int(t_distance), device=x_0.device, dtype=torch.int64)
        return [self.noise_fn(x_0, t_tensor).float() for _ in range(n)]

    def _run_chains(self, model, x_0, t_distances, noise, slots=None):
        """Reverse chains of ONE image with individual lengths, batched over `slots` chain slots.

        t_distances[c] / noise[c]: chain c is `sample_q(x_0, t_distances[c], noise[c])` followed by t_distances[c] steps of
        `sample_p(model, x, t)` with the default gaussian step noise (GaussianDiffusion.py:501-512, 556-567).  Returns the final
        images, [len(t_distances), C, H, W].  The schedule (which slot, which global step) is kept in `self.last_chain_schedule`."""
        import os
        _lib.require_cuda(x_0, "GaussianDiffusionModel.detection")
        if x_0.shape[0] != 1:
            raise ValueError("detection loops take one image (upstream stores each chain into output[avg], :514)")
        n = len(t_distances)
        lens = [int(d) for d in t_distances]
        out = torch.empty((n,) + tuple(x_0.shape[1:]), device=x_0.device, dtype=torch.float32)
        if n == 0:
            return out
        if min(lens) < 0 or max(lens) > self.num_timesteps - 1:
            # sample_q at t = t_distance reads the T-entry tables: upstream's extract() raises for t_distance >= T
            raise IndexError(f"t_distance {max(lens)} is out of range for a {self.num_timesteps}-step schedule")
        if slots is None and os.environ.get("ANODDPM_DET_SLOTS"):
            try:
                slots = int(os.environ["ANODDPM_DET_SLOTS"])
            except ValueError:
                slots = 0
            if slots < 1:
                raise ValueError(f"ANODDPM_DET_SLOTS={os.environ['ANODDPM_DET_SLOTS']!r}: expected an integer >= 1")
        if slots is not None and int(slots) < 1:
            raise ValueError("_run_chains: slots must be >= 1")
        auto = slots is None
        if auto:
            # a batched step costs about (2 + G) image-units (a batch-independent floor worth two images): take the
            # slot count among the quantisation-free sizes whose longest-first schedule is cheapest -- and whose plan fits: a
            # slot costs the activation buffers of one image (about 0.8 GB at 256^2 / base 128, scaled by pixels x base width;
            # 512^2 models at batch 16 are 50 GB), so sizes that would take more than half of the free device memory are skipped
            pos = [l for l in lens if l > 0] or [1]
            per_slot = 24.0 * 4.0 * float(getattr(model, "model_channels", 128)) * x_0.shape[-1] * x_0.shape[-2]
            free = torch.cuda.mem_get_info(x_0.device)[0] if x_0.is_cuda else float("inf")
            cands = [g for g in (16, 12, 8) if g <= max(n, 8) and (g == 8 or g * per_slot <= 0.5 * free)]
            while cands[-1] > 1 and cands[-1] * per_slot > 0.5 * free:
                cands.append(cands[-1] // 2)                             # even eight images do not fit: 4, 2, 1 slots
            cands = [g for g in cands if g * per_slot <= 0.5 * free] or [1]
            slots = min(cands, key=lambda g: plan_chain_slots(pos, min(g, len(pos)))[0] * (2 + min(g, len(pos))))
        G = max(1, min(int(slots), n))
        t_all = torch.tensor(lens, device=x_0.device, dtype=torch.int64)
        x_start = self.sample_q(x_0.repeat(n, 1, 1, 1), t_all, noise)
        live = [c for c in range(n) if lens[c] > 0]
        for c in range(n):
            if lens[c] == 0:
                out[c].copy_(x_start[c])                               # no reverse step: the noised image itself
        makespan, place = plan_chain_slots([lens[c] for c in live], G)
        self.last_chain_schedule = {"slots": G, "steps": makespan, "chain_steps": sum(lens),
                                    "place": {live[i]: place[i] for i in range(len(live))}}
        if makespan == 0:
            return out
        refill, harvest = {}, {}
        last_busy = [0] * G                                            # first global step at which a slot has nothing left to do
        for i, (slot, start) in enumerate(place):
            c = live[i]
            refill.setdefault(start, []).append((slot, c))
            harvest.setdefault(start + lens[c] - 1, []).append((slot, c))
            last_busy[slot] = max(last_busy[slot], start + lens[c])
        with torch.no_grad():
            # kept slot chains of this model at OTHER slot counts are superseded (each holds a plan + captured graph: 3.4 GB per
            # four images at config 2): drop them before the new plan is allocated (round-5 advisor finding)
            cache = self.__dict__.get("_chains", {})
            stale = [k for k, ch in cache.items() if getattr(ch, "slot_chain", False) and k[0] == id(model)
                     and k[1][1:] == tuple(x_start.shape[1:]) and k[1][0] != G]
            if stale:
                # the dropped chains' captured graphs may still have replays (and the copies that harvested their last images) in
                # flight on this stream: let the device finish before their hipGraphExec objects are destroyed
                torch.cuda.synchronize(x_start.device)
                for key in stale:
                    cache.pop(key)
            while True:
                try:
                    chain = self._chain_for(model, x_start[:1].expand(G, -1, -1, -1).contiguous(), 1, "gauss")
                    break
                except torch.cuda.OutOfMemoryError:
                    if not auto or G == 1:
                        raise
                    G = max(1, G // 2)                                   # the estimate was too optimistic: fewer slots, new schedule
                    torch.cuda.empty_cache()
                    makespan, place = plan_chain_slots([lens[c] for c in live], G)
                    self.last_chain_schedule.update(slots=G, steps=makespan, place={live[i]: place[i] for i in range(len(live))})
                    refill, harvest, last_busy = {}, {}, [0] * G
                    for i, (slot, start) in enumerate(place):
                        c = live[i]
                        refill.setdefault(start, []).append((slot, c))
                        harvest.setdefault(start + lens[c] - 1, []).append((slot, c))
                        last_busy[slot] = max(last_busy[slot], start + lens[c])
            chain.slot_chain = True
            chain.remaining = makespan
            for slot in range(G):
                if last_busy[slot] == 0:                                # fewer chains than slots cannot happen (G <= n); kept for safety
                    chain.t[slot:slot + 1].fill_(makespan - 1)
            for k in range(makespan):
                for slot, c in refill.get(k, ()):
                    chain.x[slot].copy_(x_start[c])
                    chain.t[slot:slot + 1].fill_(lens[c] - 1)
                chain.step()
                for slot, c in harvest.get(k, ()):
                    out[c].copy_(chain.x[slot])
                    if last_busy[slot] == k + 1 and k + 1 < makespan:
                        # nothing left for this slot: it idles on its last image with a timestep that stays >= 0 to the end
                        chain.t[slot:slot + 1].fill_(makespan - k - 2)
            chain.finish()
        return out

    def _detection_record(self, x_0, output, mask, extra):
        from . import metrics
        maps, counts = metrics.anomaly_maps(x_0, output, mask, threshold=0.5)
        rec = dict(extra)
        rec.update(output=output, mean=maps["mean"], mse=maps["mse_img"], threshold=maps["thr_img"],
                   counts=counts)
        return rec, maps

    def detection_A(self, model, x_0, args, file, mask, total_avg=2):
        """GaussianDiffusion.py:480-529: simplex frequencies 2^7..2^1 x t_distance 50..0.6T step 50, `total_avg` chains each -- all
        of them one batched reverse loop (`_run_chains`).  Returns None as upstream; the per-setting results upstream only plots
        (the figure files are file / plot I/O, out of scope) are kept in `self.last_detection`, in upstream's loop order: mean /
        mse / threshold images and the segmentation counts, on the device."""
        settings, dists, noise = [], [], []
        for i in range(7, 0, -1):
            freq = 2 ** i
            self.noise_fn = SimplexNoiseFn(self.simplex, frequency=freq, in_channels=self.img_channels)     # :491-494
            for t_distance in range(50, int(args["T"] * 0.6), 50):
                settings.append({"freq": i, "t_distance": t_distance})
                dists += [t_distance] * total_avg
                noise += self._forward_noise(x_0, t_distance, total_avg)
        outputs = self._run_chains(model, x_0, dists, torch.cat(noise) if noise else None)
        self.last_detection = []
        for j, extra in enumerate(settings):
            rec, _ = self._detection_record(x_0, outputs[j * total_avg:(j + 1) * total_avg].clone(), mask, extra)
            self.last_detection.append(rec)

    def detection_B(self, model, x_0, args, file, mask, denoise_fn="gauss", total_avg=5):
        """GaussianDiffusion.py:531-594: t_distance 50..end step 50 with gaussian or 6-octave simplex forward noise,
        `total_avg` chains each -- all (t_distance, avg) chains one batched reverse loop (`_run_chains`).  Returns the list upstream
        returns -- the values of `evaluation.heatmap(...)`, which are None -- and keeps the device-side results in
        `self.last_detection` (the figures upstream writes are out of scope)."""
        assert type(file) == tuple
        if denoise_fn == "octave":
            end = int(args["T"] * 0.6)
            self.noise_fn = SimplexNoiseFn(self.simplex, octave=6, persistence=0.8, frequency=64)           # :547-550
        else:
            end = int(args["T"] * 0.8)
            self.noise_fn = lambda x, t: torch.randn_like(x)
        settings = list(range(50, end, 50))
        dists, noise = [], []
        for t_distance in settings:
            dists += [t_distance] * total_avg
            noise += self._forward_noise(x_0, t_distance, total_avg)
        outputs = self._run_chains(model, x_0, dists, torch.cat(noise) if noise else None)
        dice_coeff = []
        self.last_detection = []
        for j, t_distance in enumerate(settings):
            rec, _ = self._detection_record(x_0, outputs[j * total_avg:(j + 1) * total_avg].clone(), mask, {"t_distance": t_distance})
            self.last_detection.append(rec)
            dice_coeff.append(None)                                 # evaluation.heatmap() returns None (evaluation.py:12-22)
        return dice_coeff
