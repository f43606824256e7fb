"""Training-time operators with hand-written forward AND backward: the fused 3x3 block of the UNet
   y = conv3x3( resample( SiLU( GroupNorm32(x) ) ) ) + bias + temb[:, :, None, None] + res
(UNet.py:170-172 / 190-193 with :89 nearest-x2, :70 2x2 average, :200-216 the embedding add and the residual),
i.e. 95 % of the model's FLOPs in both directions.  `FusedGNSiLUConv3x3` is a torch.autograd.Function over
channels_last (= NHWC in memory) tensors, so it drops into UNetModel's differentiable forward without layout
copies; everything it launches goes through the C ABI:

  forward   anoddpm_chan_stats + anoddpm_gn_finalize (statistics, mean/rstd kept) -> anoddpm_igemm (Winograd or direct)
  backward  d_bias / d_temb   anoddpm_chan_stats on dy (per-image column sums)
            dW                anoddpm_conv3x3_wgrad (GN-apply / SiLU / resample re-applied on the operand load)
            da                anoddpm_igemm on dy with the flipped, transposed weights (Winograd when eligible)
            dx, dgamma, dbeta anoddpm_gn_silu_backward

Reference semantics: torch autograd of the same expression (diffusion_training.py:102 loss.backward()).
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import ChanStatsArgs, GnBwdArgs, GnFinalizeArgs, IgemmArgs, WgradArgs, check, current_stream, lib

__all__ = ["FusedGNSiLUConv3x3", "fused_gn_silu_conv3x3"]

_CL = torch.channels_last


def _cl(t):
    """NCHW-shaped tensor whose memory is NHWC (no copy when it already is)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous(memory_format=_CL)


class _Cache:
    """Packed weights per parameter OBJECT (weak reference: a freed parameter's address and version can be reused
    by another model) and kind, valid for one (autograd version, owner epoch) -- the fused optimizer kernel writes
    parameters without touching autograd's version counter and bumps the owner's epoch instead
    (UNetModel.mark_weights_changed) -- and grow-only scratch buffers per device."""
    packs = {}
    scratch = {}

    @classmethod
    def packed(cls, w, kind, epoch):
        import weakref
        key = id(w)
        ent = cls.packs.get(key)
        if ent is None or ent["ref"]() is not w:
            ent = {"ref": weakref.ref(w, lambda _r, k=key: cls.packs.pop(k, None)), "kinds": {}}
            cls.packs[key] = ent
        stamp = (w._version, epoch, w.data_ptr())
        hit = ent["kinds"].get(kind)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        N, K = w.shape[0], w.shape[1]
        wino, bwd = kind.endswith("wino"), kind.startswith("bwd")
        src = w.detach()
        if src.dtype != torch.float32 or not src.is_contiguous():
            src = src.float().contiguous()
        buf = hit[1] if hit is not None else torch.empty((16 if wino else 9) * N * K, device=w.device, dtype=torch.float32)
        check(lib().anoddpm_pack_conv3x3(src.data_ptr(), buf.data_ptr(), N, K, 1 if wino else 0, 1 if bwd else 0,
                                         current_stream()), "pack_conv3x3")
        ent["kinds"][kind] = (stamp, buf)
        return buf

    @classmethod
    def buf(cls, name, numel, dtype, device):
        key = (name, dtype, device)
        b = cls.scratch.get(key)
        if b is None or b.numel() < numel:
            b = torch.empty(max(numel, 1), dtype=dtype, device=device)
            cls.scratch[key] = b
        return b


def _conv_cfg(H, W, K, N, B, a_mode):
    """Tile configuration and split-K of one 3x3 launch: the inference plan's policy (unet.choose_conv_cfg)."""
    from .unet import choose_conv_cfg
    return choose_conv_cfg(H, W, K, N, B, ks=3, a_mode=a_mode)


def _launch_conv(x, K, Hs, Ws, w, kind, epoch, *, H, W, N, a_mode, gn, act, bias, temb, res):
    """anoddpm_igemm for one 3x3 layer on channels_last tensors.  x: [B,K,Hs,Ws]; returns [B,N,H,W] channels_last."""
    B = x.shape[0]
    dev = x.device
    cfg, ksplit = _conv_cfg(H, W, K, N, B, a_mode)
    wp = _Cache.packed(w, kind + ("_wino" if cfg == 2 else "_direct"), epoch)
    out = torch.empty((B, N, H, W), device=dev, dtype=torch.float32, memory_format=_CL)
    st = IgemmArgs()
    st.a0, st.a1 = x.data_ptr(), None
    st.a0_ld, st.a1_ld, st.c0, st.c1 = K, 4, K, 0
    st.a0_bs, st.a1_bs = Hs * Ws * K, 0
    st.gn_scale = gn[0].data_ptr() if gn else None
    st.gn_shift = gn[1].data_ptr() if gn else None
    st.gn_ld = K
    st.bmat = wp.data_ptr()
    st.bias = bias.data_ptr() if bias is not None else None
    st.temb = temb.data_ptr() if temb is not None else None
    st.temb_ld = N
    st.res = res.data_ptr() if res is not None else None
    st.out, st.out_ld, st.res_ld = out.data_ptr(), N, N
    st.o_bs = st.r_bs = H * W * N
    st.H, st.W, st.ks, st.a_mode, st.act = H, W, 3, a_mode, act
    st.b_mode, st.ldb, st.N, st.B, st.heads, st.alpha = 0, 0, N, B, 1, 1.0
    st.cfg, st.ksplit = cfg, ksplit
    if ksplit > 1:
        st.ws = _Cache.buf("splitk", ksplit * B * H * W * N, torch.float32, dev).data_ptr()
    check(lib().anoddpm_igemm(ctypes.byref(st), current_stream()), "igemm")
    return out


def _column_sums(t, C, P):
    """Per-image per-channel sums of a channels_last tensor via anoddpm_chan_stats: [B][C] (fp64 fold of the slabs)."""
    B = t.shape[0]
    nslab = max(1, min(64, P // 256))
    stats = torch.empty((B, nslab, C, 2), device=t.device, dtype=torch.float32)
    st = ChanStatsArgs()
    st.a, st.stats, st.a_bs, st.C, st.a_ld, st.P, st.B, st.nslab = t.data_ptr(), stats.data_ptr(), P * C, C, C, P, B, nslab
    check(lib().anoddpm_chan_stats(ctypes.byref(st), current_stream()), "chan_stats")
    return stats, nslab


class FusedGNSiLUConv3x3(torch.autograd.Function):
    """y = conv3x3(resample(silu(group_norm(x, 32)))) + bias + temb[:, :, None, None] + res.
    a_mode: 0 same resolution, 1 nearest x2 between activation and conv, 2 2x2 average between them."""

    @staticmethod
    def forward(ctx, x, gamma, beta, weight, bias, temb, res, a_mode, epoch):
        _lib.require_cuda(x, "FusedGNSiLUConv3x3")
        x = _cl(x.detach())
        B, K, Hs, Ws = x.shape
        N = weight.shape[0]
        H, W = (Hs, Ws) if a_mode == 0 else ((Hs * 2, Ws * 2) if a_mode == 1 else (Hs // 2, Ws // 2))
        dev = x.device
        # GroupNorm statistics -> per-image per-channel affine (+ mean / rstd for the backward)
        stats, nslab = _column_sums(x, K, Hs * Ws)
        scale = torch.empty((B, K), device=dev)
        shift = torch.empty((B, K), device=dev)
        mean = torch.empty((B, 32), device=dev)
        rstd = torch.empty((B, 32), device=dev)
        g = gamma.detach().float().contiguous()
        bt = beta.detach().float().contiguous()
        fa = GnFinalizeArgs()
        fa.stats0, fa.rows0, fa.stats1, fa.rows1 = stats.data_ptr(), nslab, None, 0
        fa.gamma, fa.beta, fa.scale, fa.shift = g.data_ptr(), bt.data_ptr(), scale.data_ptr(), shift.data_ptr()
        fa.c0, fa.c1, fa.P, fa.B, fa.groups, fa.eps = K, 0, Hs * Ws, B, 32, 1e-5
        fa.mean_out, fa.rstd_out = mean.data_ptr(), rstd.data_ptr()
        check(lib().anoddpm_gn_finalize(ctypes.byref(fa), current_stream()), "gn_finalize")
        tb = temb.detach().float().contiguous() if temb is not None else None
        rs = _cl(res.detach()) if res is not None else None
        bs = bias.detach().float().contiguous() if bias is not None else None
        y = _launch_conv(x, K, Hs, Ws, weight, "fwd", epoch, H=H, W=W, N=N, a_mode=a_mode, gn=(scale, shift), act=1,
                         bias=bs, temb=tb, res=rs)
        ctx.save_for_backward(x, g, bt, weight, scale, shift, mean, rstd)
        ctx.a_mode, ctx.epoch, ctx.dims = a_mode, epoch, (B, K, Hs, Ws, N, H, W)
        ctx.has = (bias is not None, temb is not None, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, bt, weight, scale, shift, mean, rstd = ctx.saved_tensors
        B, K, Hs, Ws, N, H, W = ctx.dims
        a_mode = ctx.a_mode
        dev = x.device
        dy = _cl(dy.detach())
        need = ctx.needs_input_grad
        has_bias, has_temb, has_res = ctx.has
        d_bias = d_temb = None
        want_sums = (has_bias and need[4]) or (has_temb and need[5])
        if want_sums and not need[3]:
            stats, _ = _column_sums(dy, N, H * W)
            d_temb = stats[..., 0].double().sum(dim=1).float()               # [B][N]
            d_bias = d_temb.sum(dim=0)
        dW = None
        if need[3]:
            # split-K granularity: whole rounds of the 512 resident workgroups (2 per CU), as few as fill the chip
            tiles = -(-K // 64) * -(-N // 64)
            TW = next(t for t in (32, 16, 8, 4, 2) if W % t == 0)
            per_band = tiles * B * (W // TW)
            nband = max(1, min(H, round(512 / per_band)))
            band = -(-H // nband)
            nitems = B * (W // TW) * -(-H // band)
            ws = _Cache.buf("wgrad", nitems * 9 * K * N, torch.float32, dev)
            dW = torch.empty((N, K, 3, 3), device=dev, dtype=torch.float32)
            wa = WgradArgs()
            wa.a0, wa.a1, wa.gn_scale, wa.gn_shift = x.data_ptr(), None, scale.data_ptr(), shift.data_ptr()
            wa.dy, wa.dw, wa.ws, wa.ws_floats = dy.data_ptr(), dW.data_ptr(), ws.data_ptr(), ws.numel()
            wa.a0_bs, wa.a1_bs, wa.dy_bs = Hs * Ws * K, 0, H * W * N
            wa.c0, wa.c1, wa.a0_ld, wa.a1_ld, wa.dy_ld = K, 0, K, 4, N
            wa.H, wa.W, wa.N, wa.B = H, W, N, B
            wa.a_mode, wa.act, wa.gn_ld, wa.band, wa.accumulate = a_mode, 1, K, band, 0
            colsum = torch.empty((B, nitems // B, N), device=dev, dtype=torch.float32) if want_sums else None
            wa.colsum = colsum.data_ptr() if want_sums else None
            check(lib().anoddpm_conv3x3_wgrad(ctypes.byref(wa), current_stream()), "conv3x3_wgrad")
            if want_sums:                                                    # the kernel summed dy over each item's pixels
                d_temb = colsum.double().sum(dim=1).float()                  # [B][N]
                d_bias = d_temb.sum(dim=0)
        dx = dgamma = dbeta = None
        if need[0] or need[1] or need[2]:
            # data gradient w.r.t. the tensor the conv read (conv-output resolution): forward kernels on flipped weights
            da = _launch_conv(dy, N, H, W, weight, "bwd", ctx.epoch, H=H, W=W, N=K, a_mode=0, gn=None, act=0, bias=None, temb=None, res=None)
            dx = torch.empty((B, K, Hs, Ws), device=dev, dtype=torch.float32, memory_format=_CL)
            dgamma = torch.zeros(K, device=dev)
            dbeta = torch.zeros(K, device=dev)
            P = Hs * Ws
            nslab = max(1, min(64, P // 64))
            part = _Cache.buf("gnbwd_part", B * nslab * K * 2, torch.float64, dev)
            coef = _Cache.buf("gnbwd_coef", B * K * 4, torch.float32, dev)
            ga = GnBwdArgs()
            ga.x0, ga.x1, ga.da = x.data_ptr(), None, da.data_ptr()
            ga.gamma, ga.beta, ga.mean, ga.rstd = g.data_ptr(), bt.data_ptr(), mean.data_ptr(), rstd.data_ptr()
            ga.dx0, ga.dx1, ga.dgamma, ga.dbeta = dx.data_ptr(), None, dgamma.data_ptr(), dbeta.data_ptr()
            ga.partial, ga.coef = part.data_ptr(), coef.data_ptr()
            ga.x0_bs, ga.x1_bs, ga.da_bs, ga.dx0_bs, ga.dx1_bs = P * K, 0, H * W * K, P * K, 0
            ga.c0, ga.c1, ga.x0_ld, ga.x1_ld, ga.da_ld, ga.dx0_ld, ga.dx1_ld = K, 0, K, 4, K, K, 4
            ga.Hs, ga.Ws, ga.B, ga.groups, ga.nslab = Hs, Ws, B, 32, nslab
            ga.act, ga.a_mode, ga.acc_dx = 1, a_mode, 0
            check(lib().anoddpm_gn_silu_backward(ctypes.byref(ga), current_stream()), "gn_silu_backward")
        return (dx if need[0] else None, dgamma if need[1] else None, dbeta if need[2] else None, dW,
                d_bias if (has_bias and need[4]) else None, d_temb if (has_temb and need[5]) else None,
                dy if (has_res and need[6]) else None, None, None)


def fused_gn_silu_conv3x3(x, gamma, beta, weight, bias=None, temb=None, res=None, a_mode=0, epoch=0):
    """`epoch`: the owner's weight epoch (UNetModel._weights_epoch) -- invalidates packed weights after raw-kernel updates."""
    return FusedGNSiLUConv3x3.apply(x, gamma, beta, weight, bias, temb, res, a_mode, epoch)
