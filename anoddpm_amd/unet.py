"""`UNetModel` -- the reference denoiser (UNet.py:220-406) as an MI355X-native module.

Boundary kept from the reference: constructor signature, `forward(x, time) -> Tensor like x`,
`nn.Module` behaviour (state_dict keys / shapes / init order identical, so checkpoints and
`copy.deepcopy` / `optim.AdamW(model.parameters())` work unchanged), `update_ema_params`.

Execution: the module owns only parameters.  `forward` under `torch.no_grad()` (sampling, the
hot path) compiles a *plan* once per (batch, device): packed weights, NHWC activation buffers and
a flat list of C-ABI ops (include/anoddpm_hip.h) that the native executor `anoddpm_run_ops`
launches in one call -- GroupNorm statistics, MFMA implicit-GEMM convolutions with the
GroupNorm-apply/SiLU/resample/concat fused into their operand load, attention as two MFMA GEMMs
around a softmax, and the batched timestep-embedding projections.  There is no eager-PyTorch or
CPU fallback for this path: CPU tensors raise.

When autograd is recording (training, diffusion_training.py:99-105) the forward and the backward are the two
op lists of the native training plan (train_plan.py) behind one autograd Function.
"""
import ctypes
import math
import os

import numpy as np
import torch
from torch import nn

from . import _lib
from ._lib import (AttentionArgs, ChanStatsArgs, GnFinalizeArgs, HeadArgs, IgemmArgs, LayoutArgs, LinearArgs, Op, PackArgs, PackBatchArgs,
                   PosembArgs, ResampleArgs, SoftmaxArgs, StemArgs, check, lib)

__all__ = ["UNetModel", "update_ema_params", "zero_module", "GroupNorm32"]

_DEFAULT_MULTS = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4),
                  64: (1, 2, 3, 4), 32: (1, 2, 3, 4)}


# ----------------------------------------------------------------------------- parameters
class _Affine(nn.Module):
    """weight/bias holder for a GroupNorm(32, C) (UNet.py:409-411)."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _Weights(nn.Module):
    """weight/bias holder with nn.Conv*/nn.Linear default initialisation (same RNG draws)."""

    def __init__(self, shape, zero=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape))
        self.bias = nn.Parameter(torch.empty(shape[0]))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = int(np.prod(shape[1:]))
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        nn.init.uniform_(self.bias, -bound, bound)
        if zero:                                   # zero_module (UNet.py:414-420)
            with torch.no_grad():
                self.weight.zero_()
                self.bias.zero_()


def _indexed(**mods):
    d = nn.ModuleDict()
    for k, v in mods.items():
        d[k.lstrip("_")] = v
    return d


class _ResBlockParams(nn.Module):
    def __init__(self, cin, ted, cout):
        super().__init__()
        self.in_layers = _indexed(_0=_Affine(cin), _2=_Weights((cout, cin, 3, 3)))
        self.embed_layers = _indexed(_1=_Weights((cout, ted)))
        self.out_layers = _indexed(_0=_Affine(cout), _3=_Weights((cout, cout, 3, 3), zero=True))
        if cin != cout:
            self.skip_connection = _Weights((cout, cin, 1, 1))


class _AttentionParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = _Affine(c)
        self.to_qkv = _Weights((3 * c, c, 1))
        self.proj_out = _Weights((c, c, 1), zero=True)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


def update_ema_params(target, source, decay_rate=0.9999):
    """ema = decay*ema + (1-decay)*src per named parameter (UNet.py:423-427).  Two multi-tensor launches instead of two
    launches per parameter (536 tensors at config 2); training.FusedAdamWEMA folds the same update into the optimizer
    kernel and is what the training step uses."""
    tp = dict(target.named_parameters())
    sp = dict(source.named_parameters())
    keys = list(tp)
    with torch.no_grad():
        dst = [tp[k].data for k in keys]
        src = [sp[k].data for k in keys]
        torch._foreach_mul_(dst, decay_rate)
        torch._foreach_add_(dst, src, alpha=1 - decay_rate)
    if hasattr(target, "mark_weights_changed"):
        target.mark_weights_changed()


# ----------------------------------------------------------------------------- topology
def _topology(img_size, base, mults, num_res_blocks, attention_resolutions, in_channels, biggan_updown=True, conv_resample=True):
    """Block list of UNet.py:278-388: entries (prefix, kind, cin, cout, resample).  With biggan_updown=False the level
    transitions are Downsample / Upsample layers (UNet.py:60-92; kind "downsample" / "upsample", resample = "conv" when
    conv_resample) instead of resampling ResBlocks."""
    attn_ds = [img_size // int(r) for r in attention_resolutions.split(",")]
    ch = int(mults[0] * base)
    down = [[("down.0.0", "stem", in_channels, base, None)]]
    skip_ch = [ch]
    ds = 1
    for level, mult in enumerate(mults):
        for _ in range(num_res_blocks):
            n = len(down)
            cout = int(base * mult)
            blk = [(f"down.{n}.0", "res", ch, cout, None)]
            ch = cout
            if ds in attn_ds:
                blk.append((f"down.{n}.1", "attn", ch, ch, None))
            down.append(blk)
            skip_ch.append(ch)
        if level != len(mults) - 1:
            if biggan_updown:
                down.append([(f"down.{len(down)}.0", "res", ch, ch, "down")])
            else:
                down.append([(f"down.{len(down)}.0", "downsample", ch, ch, "conv" if conv_resample else None)])
            ds *= 2
            skip_ch.append(ch)
    middle = [("middle.0", "res", ch, ch, None), ("middle.1", "attn", ch, ch, None),
              ("middle.2", "res", ch, ch, None)]
    up = []
    for level, mult in reversed(list(enumerate(mults))):
        for j in range(num_res_blocks + 1):
            n = len(up)
            cin = ch + skip_ch.pop()
            cout = int(base * mult)
            blk = [(f"up.{n}.0", "res", cin, cout, None)]
            ch = cout
            m = 1
            if ds in attn_ds:
                blk.append((f"up.{n}.{m}", "attn", ch, ch, None))
                m += 1
            if level and j == num_res_blocks:
                if biggan_updown:
                    blk.append((f"up.{n}.{m}", "res", ch, ch, "up"))
                else:
                    blk.append((f"up.{n}.{m}", "upsample", ch, ch, "conv" if conv_resample else None))
                ds //= 2
            up.append(blk)
    return down, middle, up, ch


class UNetModel(nn.Module):
    def __init__(self, img_size, base_channels, conv_resample=True, n_heads=1, n_head_channels=-1,
                 channel_mults="", num_res_blocks=2, dropout=0, attention_resolutions="32,16,8",
                 biggan_updown=True, in_channels=1):
        super().__init__()
        self.dtype = torch.float32
        if channel_mults == "":
            if img_size not in _DEFAULT_MULTS:
                raise ValueError(f"unsupported image size: {img_size}")
            channel_mults = _DEFAULT_MULTS[img_size]
        self.biggan_updown = bool(biggan_updown)
        self.image_size = img_size
        self.in_channels = in_channels
        self.model_channels = base_channels
        self.out_channels = in_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mults
        self.conv_resample = conv_resample
        self.num_heads = n_heads
        self.num_head_channels = n_head_channels

        ted = base_channels * 4
        self._ted = ted
        down, middle, up, out_ch = _topology(img_size, base_channels, channel_mults, num_res_blocks,
                                             attention_resolutions, in_channels, self.biggan_updown, conv_resample)
        self._blocks = (down, middle, up)
        self._final_cin = int(base_channels * channel_mults[0])
        assert out_ch == self._final_cin or True

        # parameters are created in the reference constructor's order so that the same torch seed
        # gives the same initial weights (UNet.py:271-388)
        self.time_embedding = _indexed(_1=_Weights((ted, base_channels)), _3=_Weights((ted, ted)))

        def make(blk):
            prefix, kind, cin, cout, _ = blk
            if kind == "stem":
                return _Weights((cout, cin, 3, 3))
            if kind == "res":
                return _ResBlockParams(cin, ted, cout)
            if kind == "downsample":                   # Downsample(ch, conv_resample): keys `.downsample.weight/.bias` or none
                return _indexed(downsample=_Weights((cout, cin, 3, 3))) if blk[4] == "conv" else nn.Module()
            if kind == "upsample":                     # Upsample(ch, conv_resample): keys `.conv.weight/.bias` or none
                return _indexed(conv=_Weights((cout, cin, 3, 3))) if blk[4] == "conv" else nn.Module()
            heads = n_heads if n_head_channels == -1 else None
            if heads is None:
                assert cin % n_head_channels == 0, \
                    f"q,k,v channels {cin} is not divisible by num_head_channels {n_head_channels}"
            return _AttentionParams(cin)

        def seq(blks):
            d = nn.ModuleDict()
            for b in blks:
                d[b[0].split(".")[-1]] = make(b)
            return d

        self.down = nn.ModuleList([seq(b) for b in down])
        self.middle = seq(middle)
        self.up = nn.ModuleList([seq(b) for b in up])
        self.out = _indexed(_0=_Affine(out_ch), _2=_Weights((in_channels, self._final_cin, 3, 3), zero=True))
        self._plans = {}
        self._tplans = {}
        self._weights_epoch = 0

    # ------------------------------------------------------------------ helpers
    def mark_weights_changed(self):
        """Tell the compiled plans that parameters were modified behind autograd's back (e.g. by the
        fused optimizer kernel, which writes the flat parameter buffer directly)."""
        self._weights_epoch += 1

    def _heads_for(self, c):
        return self.num_heads if self.num_head_channels == -1 else c // self.num_head_channels

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_plans", "_tplans"):
                new.__dict__[k] = {}
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_plans"] = {}
        d["_tplans"] = {}
        return d

    # ------------------------------------------------------------------ forward
    def forward(self, x, time):
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            return self._forward_autograd(x, time)
        if self.training and self.dropout > 0:
            return self._forward_train_mode_no_grad(x, time)
        return self.forward_hip(x, time)

    def _forward_train_mode_no_grad(self, x, time):
        """train() mode with dropout > 0 called under no_grad (the sampling previews of the training script run the train-mode
        model through forward_backward): nn.Dropout is active in train mode whatever the grad mode (UNet.py:192), so this runs the
        training plan's forward list (the only one with the dropout launches) and drops its saved activations."""
        _lib.require_cuda(x, "UNetModel.forward (train mode)")
        if not (x.dim() == 4 and x.shape[2] == x.shape[3] and x.shape[1] == self.in_channels):
            raise ValueError(f"expected [B,{self.in_channels},S,S] input, got {tuple(x.shape)}")
        from . import train_plan
        B, S = x.shape[0], x.shape[2]
        if not train_plan.eligible(self, B, S):
            raise NotImplementedError(f"UNetModel in train() mode with dropout at batch {B}, size {S}: outside the native training plan; "
                                      "call .eval() for sampling")
        plan = self._train_plan_for(B, S, x.device, False, float(self.dropout))
        xin = x.detach()
        if xin.dtype != torch.float32 or not xin.is_contiguous():
            xin = xin.float().contiguous()
        if xin.data_ptr() % 16:
            xin = xin.clone()
        t = time.detach()
        if t.dtype != torch.int64 or t.device != x.device or not t.is_contiguous():
            t = t.to(device=x.device, dtype=torch.int64).contiguous()
        y = plan.run_forward(xin, t)
        plan.fwd_epoch = getattr(plan, "fwd_epoch", 0) + 1          # a pending backward of an earlier forward is now stale
        return y.clone().to(x.dtype)

    def forward_hip(self, x, time, out=None, borrow=False):
        """Inference forward on the HIP plan (no autograd).  Returns a fresh tensor like x; `borrow=True` returns the plan's own
        output buffer instead (valid until the plan runs again: the reverse chain consumes it in its next launch)."""
        _lib.require_cuda(x, "UNetModel.forward")
        B, C, H, W = x.shape
        if C != self.in_channels or H != W:
            raise ValueError(f"expected [B,{self.in_channels},S,S] input, got {tuple(x.shape)}")
        plan = self._plan_for(B, H, x.device)
        xin = x.detach()
        if xin.dtype != torch.float32 or not xin.is_contiguous():
            xin = xin.float().contiguous()
        if xin.data_ptr() % 16:
            xin = xin.clone()                        # a view at an odd storage offset: the stem kernel reads 16-byte rows
        t = time.detach()
        if t.dtype != torch.int64 or t.device != x.device or not t.is_contiguous():
            t = t.to(device=x.device, dtype=torch.int64).contiguous()
        y = plan.run(xin, t)
        if out is not None:
            out.copy_(y.view_as(out))
            return out
        if borrow and x.dtype == torch.float32:
            return y.view_as(x)
        return y.clone().to(x.dtype)

    def _plan_for(self, B, S, device):
        key = (B, S, device, os.environ.get("ANODDPM_ARITH", "fp32"))
        plan = self._plans.get(key)
        if plan is None:
            plan = _Plan(self, B, S, device)
            self._plans[key] = plan
        plan.refresh_weights()
        return plan

    # ------------------------------------------------------------------ differentiable forward (training)
    def _train_plan_for(self, B, S, device, want_dx, p_drop=0.0):
        from .train_plan import TrainPlan
        key = (B, S, device, bool(want_dx), float(p_drop))
        plan = self._tplans.get(key)
        if plan is None or not plan.params_match():
            plan = self._tplans[key] = TrainPlan(self, B, S, device, want_dx=want_dx, p_drop=p_drop)
        return plan

    def _forward_autograd(self, x, time):
        """Forward with autograd recording (training, diffusion_training.py:99-102): the native training plan (train_plan.py:
        forward + backward as flat C-ABI op lists, no ATen / MIOpen compute) -- the ONE training path: every constructor option
        (dropout, the Downsample / Upsample topology), frozen parameters and input gradients included.  The differentiable
        PyTorch restatement the parity tests compare against lives in oracle/unet_oracle.py (test infrastructure)."""
        _lib.require_cuda(x, "UNetModel.forward (training)")
        if not (x.dim() == 4 and x.shape[2] == x.shape[3] and x.shape[1] == self.in_channels):
            raise ValueError(f"expected [B,{self.in_channels},S,S] input, got {tuple(x.shape)}")
        from . import train_plan
        from .training import reducer_of
        B, S = x.shape[0], x.shape[2]
        if not train_plan.eligible(self, B, S):
            raise NotImplementedError(f"UNetModel training at batch {B}, size {S}, base {self.model_channels}: outside the native "
                                      "training plan (1 <= batch <= 16 per GPU, base_channels % 4 == 0 and <= 256, <= 4 image channels)")
        red = reducer_of(self)
        if red is not None:
            red.flat.bind_grads()              # data parallel: gradients live in the flat views on EVERY rank -> same cut backward
        p_drop = float(self.dropout) if (self.training and self.dropout > 0) else 0.0
        plan = self._train_plan_for(B, S, x.device, x.requires_grad, p_drop)
        anchor = next((p for p in self.parameters() if p.requires_grad), x)       # any tensor that requires grad: makes autograd call backward()
        return train_plan.TrainPlanFunction.apply(x, time, anchor, plan)


def _posemb_freqs(half):
    """exp(arange(half) * -(ln 1e4 / half)) in fp32 exactly as UNet.py:53-54 computes it (host)."""
    step = np.log(10000) / half
    return torch.exp(torch.arange(half) * -step)


# ----------------------------------------------------------------------------- plan
class _GnFold:
    """A GroupNorm that is finished inside its consumer (anoddpm_igemm_args.fold_*): statistics sources (pointer, rows, format)
    of the one or two concatenated tensors + the affine parameters."""


def _use_winograd():
    import os
    return os.environ.get("ANODDPM_NO_WINOGRAD", "0") != "1"


def fused_attention_ok(L, ch):
    """Shapes anoddpm_attention takes: the score rows of a 16-query block live in LDS (L <= 1024), power-of-two head width."""
    return (L % 16 == 0 and 16 <= L <= 1024 and 16 <= ch <= 512 and (ch & (ch - 1)) == 0
            and os.environ.get("ANODDPM_NO_FUSED_ATTENTION", "0") != "1")


def smallmap_ok(H, W, K, N, B, *, ks, a_mode=0, b_mode=0, heads=1, c0=None):
    """cfg 5 (csrc/smallmap.hip: no split-K, batch folded into M, GroupNorm of the operand finished in the prologue) takes this
    launch: every 1x1 / conv1d on maps of <= 256 pixels and the 3x3 convolutions on maps of <= ANODDPM_SMALLMAP_CONV_MAXP pixels
    (default 64) or of <= ANODDPM_SMALLMAP_CONV_MAXM rows over the whole batch (default 256: the 16x16 maps of a batch of ONE --
    config 5 -- are the same 256-row problem as the 8x8 maps of a batch of four: 34.0 -> 21 us per 512 -> 512 layer against the
    direct kernel with split-K 16; with more rows the Winograd kernel's 2.25x fewer multiplies win).  ANODDPM_NO_SMALLMAP=1
    disables it."""
    if os.environ.get("ANODDPM_NO_SMALLMAP", "0") == "1" or a_mode != 0 or b_mode != 0 or heads != 1:
        return False
    P = H * W
    if ks == 3 and P > int(os.environ.get("ANODDPM_SMALLMAP_CONV_MAXP", 64)) and P * B > int(os.environ.get("ANODDPM_SMALLMAP_CONV_MAXM", 256)):
        return False
    if ks == 1 and P > int(os.environ.get("ANODDPM_SMALLMAP_GEMM_MAXP", 256)):
        return False
    return lib().anoddpm_smallmap_tile(ks, H, W, K, K if c0 is None else c0, N, B) != 0


def wino23s_ok(H, W, K, N, B, *, ks, a_mode=0, b_mode=0, heads=1, c0=None):
    """cfg 6 (csrc/wino23s.hip: Winograd F(2x2,3x3) on the 16x16 / 32x32 maps without split-K, GroupNorm of the operand finished
    in the prologue) takes this launch.  ANODDPM_NO_WINO23S=1 disables it."""
    if os.environ.get("ANODDPM_NO_WINO23S", "0") == "1" or ks != 3 or b_mode != 0 or heads != 1 or not _use_winograd():
        return False
    # Measured (config 2, gpurun_out/r4e): every 32- or 64-channel workgroup re-stages and re-transforms the input for its
    # own channels, so the kernel wins where K is short -- 256 channels: 32x32 38.3 + 5 (finalize) -> 38.8 us, 16x16 35.6 -> 25.9 --
    # and loses to split-K + tail on the deep layers (16x16 512 -> 512: 42.2 -> 45.6 us, 1024 -> 512: 59.9 -> 82.6)
    if K > int(os.environ.get("ANODDPM_WINO23S_MAXK", 256)):
        return False
    return lib().anoddpm_wino23s_tile(H, W, K, K if c0 is None else c0, N, B, a_mode) != 0


F43_MAX_GN_K = 1024          # R4_KMAX / W43_KMAX of the F(4x4,3x3) kernels: input channels whose GroupNorm affine fits their LDS table


def choose_conv_cfg(H, W, K, N, Z, *, ks=3, a_mode=0, b_mode=0, heads=1, c0=None, c1=0, wino=True, f43=False, plain=False, small=False):
    """Tile configuration (0: 128x128 direct, 1: 64x64 direct, 4: streaming 1x1 for large maps, 2: Winograd F(2x2,3x3),
    3: Winograd F(4x4,3x3) -- only when the
    caller can supply its weights, `f43`, and only on maps >= 64x64 with enough workgroups, where its 1.78x fewer MFMAs outweigh
    the looser fp32 rounding: ~8e-6 per layer instead of 4e-7) and split-K of one
    anoddpm_igemm launch; shared by the inference plan and the training plan.  Policy: fill the 256
    CUs -- >= 512 workgroups for the direct kernels when K allows it, one full round of >= 4-chunk workgroups for
    Winograd on small maps."""
    c0 = K if c0 is None else c0
    P = H * W
    if small and smallmap_ok(H, W, K, N, Z, ks=ks, a_mode=a_mode, b_mode=b_mode, heads=heads, c0=c0):
        return 5, 1
    if small and wino and wino23s_ok(H, W, K, N, Z, ks=ks, a_mode=a_mode, b_mode=b_mode, heads=heads, c0=c0):
        return 6, 1
    if (plain and ks == 1 and a_mode == 0 and b_mode == 0 and heads == 1 and K % 128 == 0 and K <= 512 and c0 % 32 == 0
            and P % 32 == 0 and N % 64 == 0 and os.environ.get("ANODDPM_NO_STREAM1X1", "0") != "1"):
        # cfg 4: streaming 1x1 (weights resident in LDS, one 32-pixel tile per wave pass) -- `plain` = no fused GroupNorm /
        # activation / statistics; needs enough 32-pixel tiles x channel blocks for the 2048 waves of the chip -- half of them
        # still beats the direct kernel + split-K tail (config 2, 64x64 128 -> 256: step 9.381 -> 9.362 ms; a quarter: 9.394)
        nb = 128 if (N % 128 == 0 and K <= 256) else 64
        if (Z * P // 32) * (N // nb) >= int(os.environ.get("ANODDPM_STREAM1X1_MIN_TILES", 1024)):
            return 4, 1

    def ok128():
        if P % 128:
            return False
        if ks == 1:
            return True
        tw = min(W, 32)
        th = 128 // tw
        return th <= H and H % th == 0
    blocks128 = (P // 128) * ((N + 127) // 128) * Z if ok128() else 0
    cfg = 0 if (blocks128 >= 256 and N >= 96) else 1
    wino_blocks = (H // 16) * (W // 16) * ((N + 63) // 64) * Z
    wino_ksplit = 1
    wino_ok = (wino and ks == 3 and b_mode == 0 and heads == 1 and a_mode in (0, 1) and H % 16 == 0 and W % 16 == 0
               and K % 16 == 0 and (c1 == 0 or c0 % 16 == 0) and N >= 32 and _use_winograd())
    if wino_ok and wino_blocks < 200:
        # small maps: split K over 16-channel chunks until one round of workgroups fills the 256 CUs, keeping
        # >= 4 chunks per workgroup (the prologue / epilogue of a workgroup cost about two chunks)
        wch = K // 16
        wino_ksplit = int(min(max(1, wch // 4), -(-256 // wino_blocks)))
        if N % 4 or wino_blocks * wino_ksplit < 128 or os.environ.get("ANODDPM_NO_WINOGRAD_SPLITK"):
            wino_ok = False
        else:
            cps = -(-wch // wino_ksplit)
            wino_ksplit = -(-wch // cps)                   # no empty trailing block
    # the F(4x4) kernels keep the image's GroupNorm affine in an LDS table of F43_MAX_GN_K input channels (csrc/winograd43.hip,
    # winograd43r.hip refuse more): wider GroupNorm-fused layers -- none in the shipped configurations, max 1024 -- fall through to
    # F(2x2) / the direct kernels (round-5 advisor finding); `plain` launches carry no affine and are not limited
    f43 = f43 and (plain or K <= F43_MAX_GN_K)
    # (the same on the 16x16 maps -- one tile per image, 8-16 K slices -- measured 8.974 / 8.972 / 8.985 against 8.960 / 8.950 / 8.961 ms:
    # not taken, profiles/r6_f43_32_ab.txt)
    if (wino_ok and f43 and N % 128 == 0 and H * W == 32 * 32 and K >= int(os.environ.get("ANODDPM_F43_32_MINK", 512))
            and os.environ.get("ANODDPM_F43_32", "1") == "1"):
        # the deep 32x32 layers on the channel-sliced F(4x4) kernel with split-K -- 16 x 16-pixel tiles x 128 channels x K slices of
        # >= 4 chunks -- instead of F(2x2) + split-K.  Round 5 measured it inside the box-to-box noise and left it off; round 6
        # re-measured it as three alternating pairs of 40-step runs in one session: 8.916 / 8.935 / 8.957 against 8.993 / 9.010 / 9.011 ms
        # per config-2 step (-0.7 %; F(2x2) class -0.37 ms, F(4x4) class +0.30: profiles/r6_f43_32_ab.txt).  ON by default;
        # ANODDPM_F43_32=0 restores F(2x2) there.
        wg = (H // 16) * (W // 16) * (N // 128) * Z
        ks43 = int(min(max(1, (K // 16) // 4), -(-256 // wg)))
        cps = -(-(K // 16) // ks43)
        ks43 = -(-(K // 16) // cps)
        if wg * ks43 >= 200:                  # (batch 1, config 5: 128 workgroups of four chunks -- measured slower than F(2x2): 9.60 vs 9.52 ms)
            return 3, ks43
    if wino_ok and f43 and N % 64 == 0 and H * W >= int(os.environ.get("ANODDPM_F43_MIN_PIXELS", 64 * 64)) \
            and (H // 16) * (W // 16) * (N // 64) * Z >= 128 and os.environ.get("ANODDPM_NO_F43", "0") != "1":
        # the kernel picks 64- or 128-channel workgroups itself.  ANODDPM_F43_SPLITK=1 (measured slower, off): a 128-channel grid
        # of 100..199 workgroups (the 64x64 maps of a batch of four) split in two K slices on the channel-sliced kernel + the
        # split-K tail, instead of the 64-channel variant: config-2 step 9.93 vs 9.84 ms
        wg128 = (H // 16) * (W // 16) * (N // 128) * Z if N % 128 == 0 else 0
        if 100 <= wg128 < 200 and K // 16 >= 8 and os.environ.get("ANODDPM_F43_SPLITK", "0") == "1":
            return 3, 2
        if 0 < wg128 <= 64 and os.environ.get("ANODDPM_F43_DEEP_SPLITK", "1") == "1":
            # a quarter of the chip or less in 128-channel workgroups and a deep K (batch 1: the 64x64 512 -> 512 layers of config 5,
            # 64 workgroups of 32 chunks): K slices of >= 4 chunks on the channel-sliced kernel + the split-K tail, as on the deep
            # 32x32 layers above, when that fills the chip (round 6, alternating runs: config-5 step 9.61 / 9.59 -> 9.51 / 9.49 ms;
            # configs at batch >= 4 have no such grid).  ANODDPM_F43_DEEP_SPLITK=0 restores the 64-channel workgroups there
            ks43 = int(min(max(1, (K // 16) // 4), -(-256 // wg128)))
            cps = -(-(K // 16) // ks43)
            ks43 = -(-(K // 16) // cps)
            if ks43 > 1 and wg128 * ks43 >= 200:
                return 3, ks43
        return 3, 1
    if wino_ok:
        return 2, wino_ksplit
    bm = 128 if cfg == 0 else 64
    blocks = -(-P // bm) * ((N + bm - 1) // bm) * Z
    nchunks = (K + 31) // 32
    ksplit = 1
    target = int(os.environ.get("ANODDPM_SPLITK_TARGET", 512))
    if blocks < target and nchunks > 1 and N % 4 == 0:
        ksplit = int(min(nchunks, 16, max(1, -(-target // blocks))))
    return cfg, ksplit


class _Plan:
    """Compiled forward for one (batch, size, device): buffers + packed weights + flat op list."""

    def __init__(self, model, B, S, device):
        self.model = model
        self.B, self.S, self.device = B, S, device
        self.keep = []          # tensors / ctypes structs that must outlive the op list
        self.ops = []           # (code, struct)
        self._pack_jobs = []    # (parameter name, PackArgs): one device-side packing job per packed buffer
        self._bf16_jobs = []    # (parameter name, destination): F(4x4,3x3) weights as three bf16 planes (side configuration)
        self._packed = {}
        # ANODDPM_ARITH=bf16split3: the OPT-IN side configuration of the large 3x3 layers (csrc/winograd43b.hip: split-bf16 products,
        # not the reference's arithmetic class); inference plan only, never the default
        self.arith = os.environ.get("ANODDPM_ARITH", "fp32") if type(self) is _Plan else "fp32"
        if self.arith not in ("fp32", "bf16split3"):
            raise ValueError(f"ANODDPM_ARITH={self.arith}: expected fp32 or bf16split3")
        self.token = None
        self.flops = {"conv3": 0.0, "conv1": 0.0, "attn": 0.0, "qkvproj": 0.0}
        self.igemm_flops = 0.0
        self._ws_need = 0
        self.stats_of = {}
        self.igemm_log = []
        self.block_out = {}      # block prefix -> (NHWC buffer [B][H*W][C], C, H): layer-wise parity tests read these
        # Round 6: the F(4x4,3x3) layers of the INFERENCE plan accumulate their output's GroupNorm sums with fp64 atomics into
        # [B][N][2] (anoddpm_igemm_args.stats_csum) and an F(4x4,3x3) consumer finishes the GroupNorm in its prologue (fold_*):
        # no gn_finalize launch between two such layers.  One arena for all accumulators, cleared by the plan's first op (posemb).
        # OPT-IN (ANODDPM_CSUM=1): measured on the config-2 step it removes 24-26 launches (227 -> 201-203) and moves the step by
        # 0.2-0.6 % (9.05-9.09 -> 9.05-9.06 ms, 9.02-9.06 -> 8.98-9.00 on another box; profiles/r6_csum_ab.txt) -- the atomics and
        # the fold put 2-3 us of the 4.7 us launch they remove back into the F(4x4) launches.  Off by default: statistics rows +
        # finalize launches everywhere (what the training plan always does: its backward needs mean / rstd as tensors).
        self.csum_mode = type(self) is _Plan and os.environ.get("ANODDPM_CSUM", "0") == "1"
        self._csum_arena, self._csum_used = None, 0
        if self.csum_mode:
            cap = 2 * B * sum(p.shape[0] for p in model.parameters() if p.dim() == 4 and p.shape[-1] == 3)
            self._csum_arena = torch.zeros(max(cap, 2), dtype=torch.float64, device=device)
            self.keep.append(self._csum_arena)
        with torch.no_grad():
            self._build()
        if self.csum_mode and getattr(self, "posemb", None) is not None:
            self.posemb.zero = self._csum_arena.data_ptr() if self._csum_used else None
            self.posemb.zero_doubles = self._csum_used
        n = len(self.ops)
        self.op_array = (Op * n)()
        for i, (code, st) in enumerate(self.ops):
            self.op_array[i].code = code
            self.op_array[i].flags = 0
            self.op_array[i].args = ctypes.addressof(st)

    # -- small helpers ------------------------------------------------------------------
    def buf(self, *shape, dtype=torch.float32):
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        self.keep.append(t)
        return t

    _PACK_KINDS = {"conv": None, "wino": 1, "wino43": 5, "copy": 4, "small": 3}

    def packed(self, key, kind, out=None):
        """Plan-owned device buffer holding parameter `key` in the layout `kind` names ("conv": [taps][I/4][O][4] direct 3x3 or
        pointwise, "wino" / "wino43": Winograd-domain 3x3, "small": [9][I][O] for the stem / head kernels, "copy": as is).  Filled by
        ONE anoddpm_pack_batch launch over all of the plan's weights (refresh_weights) -- no ATen / rocBLAS kernel."""
        ck = (key, kind, None if out is None else out.data_ptr())
        hit = self._packed.get(ck)
        if hit is not None:
            return hit
        p = dict(self.model.named_parameters())[key]
        shape = tuple(p.shape)
        if kind == "wino43b":
            N, K = shape[0], shape[1]
            dst = torch.empty(54 * N * K, dtype=torch.float32, device=self.device)       # 3 planes x 36 x N x K bf16
            self.keep.append(dst)
            self._bf16_jobs.append((key, dst, N, K))
            self._packed[ck] = dst
            return dst
        k = self._PACK_KINDS[kind]
        if kind == "conv":
            k = 0 if (len(shape) == 4 and shape[2] == 3) else 2
        if k == 4:
            N, K, n_out = p.numel(), 1, p.numel()
        else:
            N, K = shape[0], shape[1]
            n_out = {0: 9, 1: 16, 2: 1, 3: 9, 5: 36}[k] * N * K
        if k in (0, 1, 2, 5) and K % 4:
            raise NotImplementedError(f"{key}: input channel count {K} must be a multiple of 4")
        dst = out if out is not None else torch.empty(n_out, dtype=torch.float32, device=self.device)
        self.keep.append(dst)
        st = PackArgs()
        st.w, st.out, st.N, st.K, st.kind, st.bwd, st.k0, st.kc = None, dst.data_ptr(), N, K, k, 0, 0, 0
        self._pack_jobs.append((key, st))
        self._packed[ck] = dst
        return dst

    def add(self, code, st):
        self.keep.append(st)
        self.ops.append((code, st))
        return st

    def refresh_weights(self):
        """(Re)pack every weight when a parameter changed (version counter / storage / mark_weights_changed epoch): the job table
        gets the parameters' current addresses and one batched launch fills all packed buffers."""
        params = list(self.model.parameters())
        token = (self.model._weights_epoch,) + tuple((p.data_ptr(), p._version) for p in params)
        if token == self.token:
            return
        named = dict(self.model.named_parameters())
        jobs = (PackArgs * len(self._pack_jobs))()
        block0 = [0]
        for i, (key, st) in enumerate(self._pack_jobs):
            src = named[key]
            if src.device != self.device:
                raise _lib.AnoddpmError(f"parameter {key} is on {src.device}, plan is on {self.device}")
            if src.dtype != torch.float32 or not src.is_contiguous():
                raise _lib.AnoddpmError(f"parameter {key} must be contiguous fp32")
            st.w = src.data_ptr()
            ctypes.memmove(ctypes.addressof(jobs[i]), ctypes.addressof(st), ctypes.sizeof(PackArgs))
            block0.append(block0[-1] + int(lib().anoddpm_pack_job_blocks(ctypes.byref(st))))
        with torch.no_grad():
            raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(self.device)
            b0 = torch.tensor(block0, dtype=torch.int32).to(self.device)
        pb = PackBatchArgs()
        pb.jobs, pb.block0, pb.njobs, pb.nblocks = raw.data_ptr(), b0.data_ptr(), len(self._pack_jobs), block0[-1]
        check(lib().anoddpm_pack_batch(ctypes.byref(pb), _lib.current_stream()), "weight packing")
        for key, dst, N, K in self._bf16_jobs:
            check(lib().anoddpm_pack_wino43_bf16x3(named[key].data_ptr(), dst.data_ptr(), N, K, _lib.current_stream()), "bf16 weight planes")
        self._pack_table = (raw, b0, pb)             # alive until the next refresh (the launch is asynchronous)
        self.token = token
        self.pack_count = getattr(self, "pack_count", 0) + 1

    # -- op emitters ----------------------------------------------------------------------
    def chan_stats(self, buf, C, P):
        """Stand-alone per-channel partial sums for tensors whose producer could not emit them
        (stem output, split-K outputs)."""
        B = self.B
        nslab = max(1, min(128, (P * (C // 4)) // 4096))
        stats = self.buf(B, nslab, C, 2)
        st = ChanStatsArgs()
        st.a, st.stats, st.a_bs = buf.data_ptr(), stats.data_ptr(), P * C
        st.C, st.a_ld, st.P, st.B, st.nslab = C, C, P, B, nslab
        self.add(_lib.OP_CHAN_STATS, st)
        self.stats_of[buf.data_ptr()] = ("rows", stats, nslab)

    fuse_gn_tail = True          # split-K tails fold the consumer's GroupNorm in (anoddpm_igemm_args.tail_*); see gn()

    def stats_source(self, st, which, src, c):
        """Point the `which`-th statistics source of a GnFinalizeArgs at the partial sums of tensor `src`."""
        kind, buf, extra = self.stats_of[src.data_ptr()]
        setattr(st, f"stats{which}", buf.data_ptr())
        setattr(st, f"rows{which}", extra if kind == "rows" else 1)
        setattr(st, f"fmt{which}", 0 if kind == "rows" else 1)

    def gn_tail_job(self, srcs, gamma, beta, want_mean=False):
        """GroupNorm over `srcs` finished inside the split-K tail that produces srcs[0] (no launch of its own): possible when that
        tensor comes from a split-K contraction with the group-partitioned tail (its folded sums are complete there), no other
        GroupNorm has claimed the tail yet, and the second source -- the skip tensor of a virtual concat -- has folded sums too.
        Returns (scale, shift[, mean, rstd]) or None."""
        B = self.B
        c0 = srcs[0][1]
        c1 = srcs[1][1] if len(srcs) > 1 else 0
        C = c0 + c1
        ent = self.stats_of.get(srcs[0][0].data_ptr())
        if ent is None or ent[0] != "csum" or ent[2].tail_gamma or C % 32 or (C // 32) % 4 or C // 32 > 64:
            return None
        other = None
        if c1:
            e1 = self.stats_of.get(srcs[1][0].data_ptr())
            if e1 is None or e1[0] not in ("csum", "asum"):         # both hold [B][c1][2] fp64 sums, complete before this launch
                return None
            other = e1[1]
        st = ent[2]
        scale, shift = self.buf(B, C), self.buf(B, C)
        st.tail_gamma, st.tail_beta = gamma, beta
        st.tail_scale, st.tail_shift = scale.data_ptr(), shift.data_ptr()
        st.tail_other, st.tail_c1 = (other.data_ptr() if other is not None else None), c1
        st.tail_groups, st.tail_eps = 32, 1e-5
        if want_mean:
            mean, rstd = self.buf(B, 32), self.buf(B, 32)
            st.tail_mean, st.tail_rstd = mean.data_ptr(), rstd.data_ptr()
            return scale, shift, mean, rstd
        return scale, shift

    def gn(self, srcs, P, gamma_key, beta_key, fold=False):
        """GroupNorm(32) affine of one or two concatenated sources from their per-channel partial sums
        (emitted by the producing igemm's epilogue).  Returns (scale, shift) [B][Ctot] -- or, with `fold` (the only consumer is a
        cfg 5 contraction, which finishes the GroupNorm in its own prologue), a _GnFold descriptor and NO launch."""
        B = self.B
        c0 = srcs[0][1]
        c1 = srcs[1][1] if len(srcs) > 1 else 0
        C = c0 + c1
        for buf, c in [(s[0], s[1]) for s in srcs]:
            if buf.data_ptr() not in self.stats_of:
                self.chan_stats(buf, c, P)
        gamma, beta = self.packed(gamma_key, "copy").data_ptr(), self.packed(beta_key, "copy").data_ptr()
        job = self.gn_tail_job(srcs, gamma, beta)
        if job is not None:
            return job
        if fold == "f43" and not all(self.stats_of[s_[0].data_ptr()][0] in ("csum", "asum") for s_ in srcs):
            fold = False                      # an F(4x4,3x3) consumer folds fp64 sums only: a rows source keeps the finalize launch
        if fold and C % 32 == 0:
            d = _GnFold()
            d.gamma, d.beta, d.groups, d.eps = gamma, beta, 32, 1e-5
            d.src = []
            for s in srcs:
                kind, buf, extra = self.stats_of[s[0].data_ptr()]
                d.src.append((buf.data_ptr(), extra if kind == "rows" else 1, 0 if kind == "rows" else 1))
            return d
        st = GnFinalizeArgs()
        self.stats_source(st, 0, srcs[0][0], c0)
        if c1:
            self.stats_source(st, 1, srcs[1][0], c1)
        else:
            st.stats1, st.rows1, st.fmt1 = None, 0, 0
        st.gamma, st.beta = gamma, beta
        scale, shift = self.buf(B, C), self.buf(B, C)
        st.scale, st.shift = scale.data_ptr(), shift.data_ptr()
        st.c0, st.c1, st.P, st.B, st.groups, st.eps = c0, c1, P, B, 32, 1e-5
        self.add(_lib.OP_GN_FINALIZE, st)
        return scale, shift

    def csum_take(self, n):
        """`n` doubles of the plan's accumulator arena (cleared at the start of every forward), as a tensor view."""
        if self._csum_used + n > self._csum_arena.numel():
            raise _lib.AnoddpmError("plan: statistics accumulator arena exhausted")
        v = self._csum_arena[self._csum_used:self._csum_used + n]
        self._csum_used += n
        return v

    def f43_fold(self, H, W, K, N, *, a_mode=0, c0=None, c1=0):
        """"f43" when a 3x3 contraction with these parameters will run on the F(4x4,3x3) kernels without split-K -- they finish a
        GroupNorm whose sources are all fp64 sums in their prologue (gn(fold="f43")) -- else False."""
        if not self.csum_mode or self.arith != "fp32" or a_mode not in (0, 1) or os.environ.get("ANODDPM_CSUM_NOFOLD", "0") == "1":
            return False
        cfg, ksplit = choose_conv_cfg(H, W, K, N, self.B, ks=3, a_mode=a_mode, c0=(K if c0 is None else c0), c1=c1,
                                      wino=True, f43=True, plain=False, small=True)
        return "f43" if (cfg == 3 and ksplit == 1 and K <= F43_MAX_GN_K) else False

    def small(self, H, W, K, N, *, ks, a_mode=0, c0=None):
        """True when the contraction with these parameters will run on cfg 5 / 6 (so its GroupNorm can be folded into it)."""
        return (smallmap_ok(H, W, K, N, self.B, ks=ks, a_mode=a_mode, c0=c0) or
                wino23s_ok(H, W, K, N, self.B, ks=ks, a_mode=a_mode, c0=c0))

    def attention(self, qkv, att, L, heads, ch, probs=None):
        """One fused launch for softmax(q^T k / sqrt(ch)) v (csrc/attention.hip) when the shape allows it; False otherwise (the
        caller then emits the three-launch form).  ANODDPM_NO_FUSED_ATTENTION=1 disables it."""
        if not fused_attention_ok(L, ch):
            return False
        st = AttentionArgs()
        st.qkv, st.out = qkv.data_ptr(), att.data_ptr()
        st.probs = probs.data_ptr() if probs is not None else None
        st.B, st.L, st.heads, st.ch, st.scale = self.B, L, heads, ch, 1.0 / math.sqrt(ch)
        self.add(_lib.OP_ATTENTION, st)
        fl = 4.0 * L * L * ch * heads * self.B
        if not hasattr(self, "attention_log"):
            self.attention_log = []
        self.attention_log.append(dict(kind="attn_fused", L=L, ch=ch, heads=heads, gflop=fl / 1e9))
        self.flops["attn"] = self.flops.get("attn", 0.0) + fl
        self.igemm_flops += fl
        return True

    def igemm(self, *, srcs, H, W, ks, N, bmat, out, out_ld=None, gn=None, act=0, a_mode=0, bias=None,
              temb=None, temb_ld=0, res=None, res_ld=0, b_mode=0, ldb=0, heads=1, alpha=1.0,
              a_strides=None, b_strides=(0, 0), o_strides=None, r_strides=None, kind="conv3", want_stats=False,
              wino=None, wino43=None, res_up=None):
        B = self.B
        P = H * W
        c0 = srcs[0][1]
        c1 = srcs[1][1] if len(srcs) > 1 else 0
        K = c0 + c1
        st = IgemmArgs()
        st.a0 = srcs[0][0] if isinstance(srcs[0][0], int) else srcs[0][0].data_ptr()
        st.a1 = (srcs[1][0].data_ptr() if c1 else None)
        a0_ld = srcs[0][2] if len(srcs[0]) > 2 else c0
        a1_ld = (srcs[1][2] if len(srcs[1]) > 2 else c1) if c1 else 4
        st.a0_ld, st.a1_ld = a0_ld, a1_ld
        if a_strides is None:
            Pin = P if a_mode == 0 else (P // 4 if a_mode == 1 else P * 4)
            st.a0_bs, st.a0_hs, st.a1_bs, st.a1_hs = Pin * a0_ld, 0, Pin * a1_ld if c1 else 0, 0
        else:
            st.a0_bs, st.a0_hs = a_strides
            st.a1_bs = st.a1_hs = 0
        fold = gn if isinstance(gn, _GnFold) else None
        st.gn_scale = gn[0].data_ptr() if (gn and fold is None) else None
        st.gn_shift = gn[1].data_ptr() if (gn and fold is None) else None
        st.gn_ld = K
        st.fold_gamma = st.fold_beta = st.fold_stats0 = st.fold_stats1 = None
        if fold is not None:
            st.fold_gamma, st.fold_beta, st.fold_groups, st.fold_eps = fold.gamma, fold.beta, fold.groups, fold.eps
            st.fold_stats0, st.fold_rows0, st.fold_fmt0 = fold.src[0]
            if c1:
                st.fold_stats1, st.fold_rows1, st.fold_fmt1 = fold.src[1]
        self._pending_bmat = (st, bmat, wino)
        st.b_bs, st.b_hs = b_strides
        st.bias = (bias if isinstance(bias, int) else bias.data_ptr()) if bias is not None else None
        st.temb = temb if temb else None
        st.temb_ld = temb_ld
        st.res = (res if isinstance(res, int) else res.data_ptr()) if res is not None else None
        out_ld = out_ld or N
        st.out = out if isinstance(out, int) else out.data_ptr()
        st.out_ld, st.res_ld = out_ld, (res_ld or N)
        if o_strides is None:
            st.o_bs, st.o_hs = P * out_ld, 0
        else:
            st.o_bs, st.o_hs = o_strides
        if r_strides is None:
            st.r_bs, st.r_hs = P * (res_ld or N), 0
        else:
            st.r_bs, st.r_hs = r_strides
        st.c0, st.c1 = c0, c1
        st.H, st.W, st.ks, st.a_mode, st.act = H, W, ks, a_mode, act
        st.b_mode, st.ldb, st.N = b_mode, ldb, N
        st.B, st.heads, st.alpha = B, heads, alpha
        Z = B * heads
        cfg, ksplit = choose_conv_cfg(H, W, K, N, Z, ks=ks, a_mode=a_mode, b_mode=b_mode, heads=heads, c0=c0, c1=c1,
                                      wino=bool(wino), f43=bool(wino43),
                                      plain=(gn is None and act == 0 and not want_stats), small=True)
        if fold is not None and not (cfg in (5, 6) or (cfg == 3 and ksplit == 1 and st.fold_fmt0 == 1 and (not c1 or st.fold_fmt1 == 1))):
            raise _lib.AnoddpmError("plan: a folded GroupNorm needs a cfg 5 / 6 consumer, or a cfg 3 one with fp64-sum sources")
        if (cfg == 3 and self.arith == "bf16split3" and ksplit == 1 and N % 128 == 0 and K % 32 == 0 and (c1 == 0 or c0 % 32 == 0)
                and (H // 16) * (W // 16) * (N // 128) * Z >= 200):
            cfg = 7                                             # the 128-channel grids (what winograd43r.hip runs) on split-bf16 products
        bm = 128 if cfg == 0 else 64
        st.cfg, st.ksplit = cfg, ksplit
        st.res_mode = 0
        if res_up is not None:
            # residual = nearest x2 of the half-resolution tensor `res_up` (the x_upd path of an up-sampling ResBlock, UNet.py:196-198):
            # the F(4x4,3x3) kernels repeat the source pixels on the read (res_mode 1: no resample launch, a quarter of the residual
            # bytes); every other kernel gets the materialised tensor
            if cfg == 3 and ksplit == 1 and os.environ.get("ANODDPM_NO_RES_UP", "0") != "1":
                st.res, st.res_mode = res_up.data_ptr(), 1
                st.res_ld, st.r_bs, st.r_hs = N, (P // 4) * N, 0
            else:
                sk = self.buf(B, P, N)
                rs = ResampleArgs()
                rs.inp, rs.out = res_up.data_ptr(), sk.data_ptr()
                rs.B, rs.H, rs.W, rs.C, rs.mode = B, H // 2, W // 2, N, 1
                self.add(_lib.OP_RESAMPLE, rs)
                st.res = sk.data_ptr()
        _st, _bmat, _wino = self._pending_bmat
        if cfg == 7:
            _bmat = wino43("wino43b")                          # F(4x4,3x3) weights as three bf16 planes
        elif cfg == 3:
            _bmat = wino43()                                   # F(4x4,3x3) weights
        elif cfg in (2, 6):
            _bmat = _wino()                                    # Winograd-domain weights for this layer
        elif callable(_bmat):
            _bmat = _bmat()                                    # packed lazily: only the layout this launch uses
        st.bmat = _bmat if isinstance(_bmat, int) else _bmat.data_ptr()
        st.ws = None                      # patched after the build (one shared workspace)
        if ksplit > 1:
            self._ws_need = max(self._ws_need, ksplit * Z * P * N)
        st.stats = None
        if want_stats and cfg in (5, 6):
            # one row per workgroup tile: TM pixels (cfg 5) / 8x8 pixels (cfg 6)
            rows = P // 64 if cfg == 6 else P // (16 * (lib().anoddpm_smallmap_tile(ks, H, W, K, c0, N, B) >> 4))
            stats = self.buf(B, rows, N, 2)
            st.stats = stats.data_ptr()
            self.stats_of[out.data_ptr()] = ("rows", stats, rows)
        elif (want_stats and ksplit == 1 and heads == 1 and cfg == 3 and self.csum_mode
              and 2 * N * (H // 16) * (W // 16) * B <= int(os.environ.get("ANODDPM_CSUM_MAX_ATOMICS", 100000))):
            # (measured, profiles/r6_csum_by_layer.txt: the chip retires about 33 of these atomics per ns, so a launch pays
            # 2 N tiles B / 33 ns for them -- 8-15 us on the 256x256 maps of a batch of four, more than the 4.7 us finalize launch
            # it replaces, 0.5-2 us on the 128x128 / 64x64 maps: only launches below the threshold accumulate atomically)
            # F(4x4,3x3): per-channel sums accumulated atomically into ONE fp64 row per image (no rows to fold)
            csum = self.csum_take(B * N * 2)
            st.stats_csum = csum.data_ptr()
            self.stats_of[out.data_ptr()] = ("asum", csum, None)
        elif want_stats and ksplit == 1 and heads == 1:
            # the epilogue emits per-channel {sum, sumsq} per wave-row of every pixel tile
            if cfg in (2, 3, 7):
                tiles = (H // 16) * (W // 16)
            elif ks == 1:
                tiles = -(-P // bm)
            else:
                tw = min(W, 32)
                tiles = (W // tw) * -(-H // (bm // tw))
            rows = tiles * {2: 4, 3: 1, 7: 1}.get(cfg, 2)      # partial rows per pixel tile (F(4x4,3x3): one per workgroup)
            stats = self.buf(B, rows, N, 2)
            st.stats = stats.data_ptr()
            self.stats_of[out.data_ptr()] = ("rows", stats, rows)
        st.stats_rows = 0
        st.tail_csum = None
        if (want_stats and ksplit > 1 and heads == 1 and N % 128 == 0 and self.fuse_gn_tail
                and P <= int(os.environ.get("ANODDPM_GN_TAIL_MAX_P", 256)) and os.environ.get("ANODDPM_NO_GN_TAIL", "0") != "1"):
            # group-partitioned tail: folded per-channel sums now, the consumer's GroupNorm attached later by gn()
            csum = self.buf(B, N, 2, dtype=torch.float64)
            st.tail_csum, st.tail_groups, st.tail_c1, st.tail_eps = csum.data_ptr(), 32, 0, 1e-5
            st.tail_gamma = st.tail_beta = st.tail_scale = st.tail_shift = st.tail_other = st.tail_mean = st.tail_rstd = None
            self.stats_of[out.data_ptr()] = ("csum", csum, st)
        elif want_stats and ksplit > 1 and heads == 1:
            # the split-K reduction emits the statistics: one row per pixel slab
            rows_per_block = 256 // min(N // 4, 256)          # pixel rows a 256-thread block covers per pass
            nslab = max(1, -(-P // (4 * rows_per_block)))     # <= 4 pixels per thread: the k-slab loads are serial
            stats = self.buf(B, nslab, N, 2)
            st.stats, st.stats_rows = stats.data_ptr(), nslab
            self.stats_of[out.data_ptr()] = ("rows", stats, nslab)
        self.add(_lib.OP_IGEMM, st)
        self.igemm_log.append(dict(wino=(cfg in (2, 3, 6, 7)), f43=(cfg in (3, 7)), kind=kind, H=H, W=W, K=K, N=N, ks=ks, a_mode=a_mode, b_mode=b_mode, heads=heads,
                                   cfg=cfg, ksplit=ksplit, dual=bool(c1), gflop=2.0 * K * N * ks * ks * P * Z / 1e9,
                                   # the fused residual the epilogue reads: 0 none, 1 full resolution, 0.25 nearest-x2 source (res_mode 1)
                                   res=(0.0 if not st.res else (0.25 if st.res_mode == 1 else 1.0))))
        if want_stats and st.stats is None and not st.tail_csum and not st.stats_csum:
            self.chan_stats(out, N, P)
        fl = 2.0 * K * N * ks * ks * P * Z
        self.flops[kind] = self.flops.get(kind, 0.0) + fl
        self.igemm_flops += fl
        return st

    # -- network ------------------------------------------------------------------------------
    def _build(self):
        m = self.model
        B, S, dev = self.B, self.S, self.device
        base, ted = m.model_channels, m._ted
        down, middle, up = m._blocks
        # --- timestep path: features -> MLP -> all per-block projections in one launch (UNet.py:271-276,185-188)
        half = base // 2
        freqs = _posemb_freqs(half).to(dev)
        self.keep.append(freqs)
        pe = self.buf(B, base)
        self.posemb = PosembArgs()
        self.posemb.t, self.posemb.freqs, self.posemb.out = None, freqs.data_ptr(), pe.data_ptr()
        self.posemb.B, self.posemb.dim, self.posemb.scale = B, base, 1.0
        self.add(_lib.OP_POSEMB, self.posemb)

        def linear(inp, wkey, bkey, K, N, act_in, act_out, w=None, b=None):
            st = LinearArgs()
            st.inp = inp.data_ptr()
            wt = w if w is not None else self.packed(wkey, "copy")
            bt = b if b is not None else self.packed(bkey, "copy")
            st.w, st.bias = wt.data_ptr(), bt.data_ptr()
            o = self.buf(B, N)
            st.out = o.data_ptr()
            st.B, st.K, st.N, st.act_in, st.act_out = B, K, N, act_in, act_out
            self.add(_lib.OP_LINEAR, st)
            return o
        h1 = linear(pe, "time_embedding.1.weight", "time_embedding.1.bias", base, ted, 0, 1)
        temb = linear(h1, "time_embedding.3.weight", "time_embedding.3.bias", ted, ted, 0, 0)
        self.temb = temb

        res_blocks = [b for grp in (down, [middle], up) for blk in grp for b in blk if b[1] == "res"]
        offs, tot = {}, 0
        for b in res_blocks:
            offs[b[0]] = tot
            tot += b[3]
        w_all = self.buf(tot, ted)
        b_all = self.buf(tot)
        for blk_ in res_blocks:                                  # every block's projection copied into the concatenated matrix
            o = offs[blk_[0]]
            self.packed(blk_[0] + ".embed_layers.1.weight", "copy", out=w_all[o:o + blk_[3]])
            self.packed(blk_[0] + ".embed_layers.1.bias", "copy", out=b_all[o:o + blk_[3]])
        emb_all = linear(temb, None, None, ted, tot, 1, 0, w=w_all, b=b_all)
        self.emb_tot = tot

        # --- blocks ---------------------------------------------------------------------------
        def res_block(prefix, srcs, Hin, cout, resample):
            cin = sum(s[1] for s in srcs)
            Hout = Hin * 2 if resample == "up" else (Hin // 2 if resample == "down" else Hin)
            Pin, Pout = Hin * Hin, Hout * Hout
            g1 = self.gn(srcs, Pin, prefix + ".in_layers.0.weight", prefix + ".in_layers.0.bias",
                         fold=(resample in (None, "up") and (self.small(Hout, Hout, cin, cout, ks=3, c0=srcs[0][1],
                                                                        a_mode=(1 if resample == "up" else 0)) or
                                                             self.f43_fold(Hout, Hout, cin, cout, a_mode=(1 if resample == "up" else 0),
                                                                           c0=srcs[0][1], c1=cin - srcs[0][1]))))
            h1 = self.buf(B, Pout, cout)
            pooled = None
            if resample == "down" and len(srcs) == 1 and os.environ.get("ANODDPM_NO_POOL_ACT", "0") != "1":
                # down block: ONE pass over x gives both the pooled skip input and the pooled ACTIVATED operand of the first
                # convolution, which then runs on the Winograd kernels instead of the pool-fused direct one
                pooled = self.buf(B, Pout, cin)
                sk_pool = self.buf(B, Pout, cin)
                st = ResampleArgs()
                st.inp, st.out = srcs[0][0].data_ptr(), sk_pool.data_ptr()
                st.B, st.H, st.W, st.C, st.mode, st.scale, st.accumulate = B, Hin, Hin, cin, 2, 1.0, 0
                st.gn_scale, st.gn_shift, st.out_act = g1[0].data_ptr(), g1[1].data_ptr(), pooled.data_ptr()
                self.add(_lib.OP_RESAMPLE, st)
            self.igemm(srcs=([(pooled, cin)] if pooled is not None else srcs), H=Hout, W=Hout, ks=3, N=cout,
                       gn=(None if pooled is not None else g1), act=(0 if pooled is not None else 1),
                       a_mode=(0 if pooled is not None else {None: 0, "up": 1, "down": 2}[resample]),
                       bmat=lambda p=prefix: self.packed(p + ".in_layers.2.weight", "conv"),
                       wino=lambda p=prefix: self.packed(p + ".in_layers.2.weight", "wino"),
                       wino43=lambda kind="wino43", p=prefix: self.packed(p + ".in_layers.2.weight", kind),
                       bias=self.packed(prefix + ".in_layers.2.bias", "copy"),
                       temb=emb_all.data_ptr() + 4 * offs[prefix], temb_ld=tot, out=h1, want_stats=True)
            g2 = self.gn([(h1, cout)], Pout, prefix + ".out_layers.0.weight", prefix + ".out_layers.0.bias",
                         fold=(self.small(Hout, Hout, cout, cout, ks=3) or self.f43_fold(Hout, Hout, cout, cout)))
            if cin != cout:
                sk = self.buf(B, Pout, cout)
                assert resample is None
                self.igemm(srcs=srcs, H=Hout, W=Hout, ks=1, N=cout, kind="conv1",
                           bmat=self.packed(prefix + ".skip_connection.weight", "conv"),
                           bias=self.packed(prefix + ".skip_connection.bias", "copy"), out=sk)
            elif resample is not None and pooled is not None:
                sk = sk_pool
            elif resample == "up":
                assert len(srcs) == 1
                sk = None                                          # igemm(res_up=...) below: fused into the F(4x4) epilogue, else materialised there
            elif resample is not None:
                assert len(srcs) == 1
                sk = self.buf(B, Pout, cout)
                st = ResampleArgs()
                st.inp, st.out = srcs[0][0].data_ptr(), sk.data_ptr()
                st.B, st.H, st.W, st.C, st.mode = B, Hin, Hin, cin, 2
                self.add(_lib.OP_RESAMPLE, st)
            else:
                if len(srcs) != 1:
                    raise NotImplementedError("identity skip over a concatenated input (cin == cout) is not built")
                sk = srcs[0][0]
            h2 = self.buf(B, Pout, cout)
            self.igemm(srcs=[(h1, cout)], H=Hout, W=Hout, ks=3, N=cout, gn=g2, act=1,
                       bmat=lambda p=prefix: self.packed(p + ".out_layers.3.weight", "conv"),
                       wino=lambda p=prefix: self.packed(p + ".out_layers.3.weight", "wino"),
                       wino43=lambda kind="wino43", p=prefix: self.packed(p + ".out_layers.3.weight", kind),
                       bias=self.packed(prefix + ".out_layers.3.bias", "copy"),
                       res=sk, res_up=(srcs[0][0] if (resample == "up" and sk is None) else None), out=h2, want_stats=True)
            return h2, Hout

        def attn_block(prefix, x, Hc, C):
            L = Hc * Hc
            heads = m._heads_for(C)
            ch = C // heads
            if ch % 4:
                raise NotImplementedError(f"attention head width {ch} must be a multiple of 4")
            g = self.gn([(x, C)], L, prefix + ".norm.weight", prefix + ".norm.bias", fold=self.small(Hc, Hc, C, 3 * C, ks=1))
            qkv = self.buf(B, L, 3 * C)
            self.igemm(srcs=[(x, C)], H=Hc, W=Hc, ks=1, N=3 * C, gn=g, act=0, kind="qkvproj",
                       bmat=self.packed(prefix + ".to_qkv.weight", "conv"),
                       bias=self.packed(prefix + ".to_qkv.bias", "copy"), out=qkv)
            att = self.buf(B, L, C)
            if not self.attention(qkv, att, L, heads, ch):
                S_ = self.buf(B * heads, L, L)
                qp = qkv.data_ptr()
                # scores = (q*s)^T (k*s), s = ch^-1/4  ->  alpha = ch^-1/2 on the product (UNet.py:147-150)
                self.igemm(srcs=[(qp, ch, 3 * C)], H=1, W=L, ks=1, N=L, b_mode=1, ldb=3 * C, heads=heads,
                           bmat=qp + 4 * ch, alpha=1.0 / math.sqrt(ch), kind="attn",
                           a_strides=(L * 3 * C, 3 * ch), b_strides=(L * 3 * C, 3 * ch),
                           out=S_, out_ld=L, o_strides=(heads * L * L, L * L))
                sm = SoftmaxArgs()
                sm.x, sm.rows, sm.L = S_.data_ptr(), B * heads * L, L
                self.add(_lib.OP_SOFTMAX, sm)
                self.igemm(srcs=[(S_.data_ptr(), L, L)], H=1, W=L, ks=1, N=ch, b_mode=2, ldb=3 * C, heads=heads,
                           bmat=qp + 4 * 2 * ch, kind="attn",
                           a_strides=(heads * L * L, L * L), b_strides=(L * 3 * C, 3 * ch),
                           out=att, out_ld=C, o_strides=(L * C, ch))
            y = self.buf(B, L, C)
            self.igemm(srcs=[(att, C)], H=Hc, W=Hc, ks=1, N=C, kind="qkvproj",
                       bmat=self.packed(prefix + ".proj_out.weight", "conv"),
                       bias=self.packed(prefix + ".proj_out.bias", "copy"),
                       res=x, out=y, want_stats=True)
            return y

        def resample_layer(prefix, kind, x, Hc, C, conv):
            """Downsample / Upsample of the biggan_updown=False topology (UNet.py:60-92) on raw activations (no norm, no
            activation).  The stride-2 convolution runs as the stride-1 one on the existing kernels followed by the even-pixel
            pick (its outputs are exactly those of the stride-1 result at (2i, 2j)): 4x the necessary work on a layer no
            shipped configuration uses, zero new contraction code."""
            def rs(inp, H, mode, out):
                st = ResampleArgs()
                st.inp, st.out = inp.data_ptr(), out.data_ptr()
                st.B, st.H, st.W, st.C, st.mode, st.scale, st.accumulate = B, H, H, C, mode, 1.0, 0
                self.add(_lib.OP_RESAMPLE, st)
            if kind == "downsample":
                Ho = Hc // 2
                out = self.buf(B, Ho * Ho, C)
                if not conv:
                    rs(x, Hc, 2, out)                                   # nn.AvgPool2d(2, 2)
                    return out, Ho
                full = self.buf(B, Hc * Hc, C)
                self.igemm(srcs=[(x, C)], H=Hc, W=Hc, ks=3, N=C, act=0,
                           bmat=lambda p=prefix: self.packed(p + ".downsample.weight", "conv"),
                           wino=lambda p=prefix: self.packed(p + ".downsample.weight", "wino"),
                           wino43=lambda kind="wino43", p=prefix: self.packed(p + ".downsample.weight", kind),
                           bias=self.packed(prefix + ".downsample.bias", "copy"), out=full)
                rs(full, Hc, 3, out)
                return out, Ho
            Ho = Hc * 2
            out = self.buf(B, Ho * Ho, C)
            if not conv:
                rs(x, Hc, 1, out)                                       # F.interpolate(scale_factor=2, mode="nearest")
                return out, Ho
            self.igemm(srcs=[(x, C)], H=Ho, W=Ho, ks=3, N=C, act=0, a_mode=1,      # nearest x2 fused into the operand load
                       bmat=lambda p=prefix: self.packed(p + ".conv.weight", "conv"),
                       wino=lambda p=prefix: self.packed(p + ".conv.weight", "wino"),
                       wino43=lambda kind="wino43", p=prefix: self.packed(p + ".conv.weight", kind),
                       bias=self.packed(prefix + ".conv.bias", "copy"), out=out, want_stats=True)
            return out, Ho

        def run(blks, srcs, Hc):
            for (prefix, kind, cin, cout, resample) in blks:
                if kind == "stem":
                    h0 = self.buf(B, S * S, cout)
                    self.stem = StemArgs()
                    self.stem.x = None
                    self.stem.w = self.packed(prefix + ".weight", "small").data_ptr()
                    self.stem.bias = self.packed(prefix + ".bias", "copy").data_ptr()
                    self.stem.out = h0.data_ptr()
                    self.stem.B, self.stem.H, self.stem.W, self.stem.Cin, self.stem.Cout = B, S, S, cin, cout
                    rows = lib().anoddpm_stem_stats_rows(S, S, cin, cout)
                    if rows > 0 and os.environ.get("ANODDPM_NO_STEM_STATS", "0") != "1":
                        # GroupNorm partial sums of the stem output from the stem kernel itself (no chan_stats pass over it)
                        sstats = self.buf(B, rows, cout, 2)
                        self.stem.stats, self.stem.stats_rows = sstats.data_ptr(), rows
                        self.stats_of[h0.data_ptr()] = ("rows", sstats, rows)
                    self.add(_lib.OP_STEM, self.stem)
                    self.flops["conv3"] += 2.0 * cin * cout * 9 * S * S * B
                    srcs = [(h0, cout)]
                elif kind == "res":
                    h, Hc = res_block(prefix, srcs, Hc, cout, resample)
                    srcs = [(h, cout)]
                elif kind in ("downsample", "upsample"):
                    h, Hc = resample_layer(prefix, kind, srcs[0][0], Hc, cin, resample == "conv")
                    srcs = [(h, cin)]
                else:
                    h = attn_block(prefix, srcs[0][0], Hc, cin)
                    srcs = [(h, cin)]
                self.block_out[prefix] = (srcs[0][0], srcs[0][1], Hc)
            return srcs, Hc

        Hc = S
        srcs = None
        skips = []
        for blk in down:
            srcs, Hc = run(blk, srcs, Hc)
            skips.append(srcs[0])
        srcs, Hc = run(middle, srcs, Hc)
        for blk in up:
            srcs, Hc = run(blk, [srcs[0], skips.pop()], Hc)

        # --- head: GN + SiLU + 3x3 conv to in_channels (UNet.py:384-388,405)
        hfin, cfin = srcs[0]
        g = self.gn([(hfin, cfin)], S * S, "out.0.weight", "out.0.bias")
        nout = m.in_channels
        if nout <= 4 and S % 8 == 0 and (100 * (cfin + 16) + 9 * cfin * nout) * 4 <= 64 * 1024:
            # dedicated HBM-bound head kernel, writes the caller's NCHW layout directly
            self.y = self.buf(B, nout, S, S)
            st = HeadArgs()
            st.x = hfin.data_ptr()
            st.w = self.packed("out.2.weight", "small").data_ptr()
            st.bias = self.packed("out.2.bias", "copy").data_ptr()
            st.gn_scale, st.gn_shift, st.out = g[0].data_ptr(), g[1].data_ptr(), self.y.data_ptr()
            st.B, st.H, st.W, st.C, st.Cout = B, S, S, cfin, nout
            self.add(_lib.OP_HEAD, st)
            self.flops["conv3"] += 2.0 * cfin * nout * 9 * S * S * B
        else:
            y_nhwc = self.buf(B, S * S, nout)
            self.igemm(srcs=[(hfin, cfin)], H=S, W=S, ks=3, N=nout, gn=g, act=1,
                       bmat=self.packed("out.2.weight", "conv"),
                       bias=self.packed("out.2.bias", "copy"), out=y_nhwc)
            if nout == 1:
                self.y = y_nhwc.view(B, 1, S, S)
            else:
                self.y = self.buf(B, nout, S, S)
                st = LayoutArgs()
                st.inp, st.out, st.B, st.P, st.C, st.in_ld = y_nhwc.data_ptr(), self.y.data_ptr(), B, S * S, nout, nout
                self.add(_lib.OP_LAYOUT, st)
        # one split-K workspace shared by every op (ops run in stream order)
        if self._ws_need:
            ws = self.buf(self._ws_need)
            for code, st in self.ops:
                if code == _lib.OP_IGEMM and st.ksplit > 1:
                    st.ws = ws.data_ptr()

    def run(self, x, t):
        """x: contiguous fp32 [B,C,S,S] on self.device; t: int64 [B].  Returns the plan-owned output."""
        self.stem.x = x.data_ptr()
        self.posemb.t = t.data_ptr()
        check(lib().anoddpm_run_ops(self.op_array, len(self.ops), _lib.current_stream()), "UNet forward")
        return self.y
