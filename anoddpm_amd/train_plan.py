"""Training-time execution of `UNetModel`: forward AND backward as two flat op lists over static NHWC buffers, run by
the native executor (`anoddpm_run_ops`) -- the reference's `loss.backward()` (diffusion_training.py:102) through
UNet.py:390-406 without a single ATen / MIOpen / rocBLAS compute kernel.

Design (MI355X-first, mirrors the inference plan of unet.py):
  * every activation of the forward keeps its buffer (the backward needs it anyway); every activation has a gradient
    buffer of the same shape; 288 GB of HBM makes recycling pointless and keeps all pointers static;
  * the forward op list starts with the device-side weight packing (3x3 direct / Winograd-domain, pointwise, small
    convs; forward and data-gradient layouts) of the CURRENT parameters, so an optimizer step needs no notification;
  * GroupNorm-apply + SiLU + resample + concat stay fused into the operand loads in both directions: forward convs,
    weight gradients (`anoddpm_conv3x3_wgrad`, `anoddpm_wgrad_pointwise`), and the GroupNorm backward
    (`anoddpm_gn_silu_backward`) read the raw block inputs; the data gradient of a conv is the forward kernel on the
    flipped / transposed weights;
  * QKVAttention backward = five MFMA GEMMs (anoddpm_igemm with activation B operands) around a softmax-backward and two
    square transposes;
  * gradient fan-in (skip connections, residuals, the shared time embedding) is resolved at BUILD time: the first writer
    of a gradient buffer overwrites, later writers accumulate -- no zero-fill and no add kernels;
  * parameter gradients are accumulated straight into `param.grad` storage (the flat gradient buffer of
    training.FlatBuffers when present), so neither autograd nor the optimizer copies them.

`UNetModel.forward` under autograd wraps the two lists in one autograd Function (`TrainPlanFunction`); everything else
of the reference loop body (p_loss, the loss expression) stays the reference's torch code on [B,1,S,S] tensors.
"""
import ctypes
import math
import os

import torch

from . import _lib
from ._lib import (ChanStatsArgs, ColsumFoldArgs, DropoutArgs, GnBwdArgs, GnFinalizeArgs, HeadArgs, HeadBwdArgs, LinearArgs, LinearBwdArgs,
                   LinearBwdBatchArgs, Op, PackArgs, PackBatchArgs, PosembArgs, ResampleArgs, SoftmaxArgs, SoftmaxBwdArgs, StemArgs, StemBwdArgs, TransposeArgs,
                   Wgrad1Args, WgradArgs, check, lib)
from .unet import _Plan, _posemb_freqs

__all__ = ["TrainPlan", "TrainPlanFunction", "eligible"]


def eligible(model, B, S):
    """Shapes the native training plan covers: everything the reference's args files use and every constructor option
    (dropout, the Downsample / Upsample topology); what is left out are extreme sizes (batch > 16 per GPU, base_channels > 256,
    more than four image channels), which raise."""
    cfin, nout = model._final_cin, model.in_channels
    head_ok = nout <= 4 and S % 8 == 0 and (100 * (cfin + 16) + 9 * cfin * nout) * 4 <= 64 * 1024 and cfin <= 256
    return head_ok and 1 <= B <= 16 and model.model_channels % 4 == 0 and model.model_channels <= 256


def _op_array(ops):
    arr = (Op * len(ops))()
    for i, (code, st) in enumerate(ops):
        arr[i].code = code
        arr[i].flags = 0
        arr[i].args = ctypes.addressof(st)
    return arr


class TrainPlan(_Plan):
    def __init__(self, model, B, S, device, want_dx=False, p_drop=0.0):
        self.want_dx = want_dx
        self.p_drop = float(p_drop)  # nn.Dropout of ResBlock.out_layers (UNet.py:192), training mode only
        self._drop_ops = []          # (forward struct, backward struct, layer index): the seeds are patched per forward
        self.bops = []               # backward op list
        self.pack_ops = []           # weight packing, runs ahead of the forward
        self._packs = {}
        self._bw = []                # closures emitting the backward of each forward stage (run in reverse)
        self._grads = {}             # data_ptr of an activation -> its gradient buffer
        self.gwritten = set()
        self._touched = set()
        self.bwd_marks = []          # (backward op count after a stage, parameter names whose gradient that stage wrote)
        self._ws_patch = []          # (struct, field, floats) sharing one training workspace
        self._tws_need = 0
        super().__init__(model, B, S, device)
        self.fwd_list = self._batched_packs() + self.ops
        self.fwd_array = _op_array(self.fwd_list)
        self.bwd_array = _op_array(self.bops)

    def _batched_packs(self):
        """The ~270 per-weight pack launches of a forward as ONE launch (anoddpm_pack_batch: device-resident job table, a block finds
        its job by bisection); ANODDPM_PACK_BATCH=0 keeps one launch per weight."""
        if os.environ.get("ANODDPM_PACK_BATCH", "1") == "0" or not self.pack_ops:
            return list(self.pack_ops)
        jobs = (PackArgs * len(self.pack_ops))()
        block0 = [0]
        for i, (_, st) in enumerate(self.pack_ops):
            ctypes.memmove(ctypes.addressof(jobs[i]), ctypes.addressof(st), ctypes.sizeof(PackArgs))
            block0.append(block0[-1] + int(lib().anoddpm_pack_job_blocks(ctypes.byref(st))))
        raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(self.device)
        b0 = torch.tensor(block0, dtype=torch.int32, device=self.device)
        pb = PackBatchArgs()
        pb.jobs, pb.block0, pb.njobs, pb.nblocks = raw.data_ptr(), b0.data_ptr(), len(self.pack_ops), block0[-1]
        self.keep += [raw, b0, pb]
        return [(_lib.OP_PACK_BATCH, pb)]

    # ------------------------------------------------------------------ parameters
    def _bind_params(self):
        """Destinations of the parameter gradients: the parameter's existing `.grad` storage (e.g. the views of
        training.FlatBuffers) or a slice of a plan-owned arena."""
        self.named = dict(self.model.named_parameters())
        need = sum((p.numel() + 3) // 4 * 4 for p in self.named.values() if p.grad is None)
        self.arena = torch.zeros(max(need, 4), device=self.device)
        self.pptr, self.gptr, self.gview = {}, {}, {}
        off = 0
        for k, p in self.named.items():
            if p.device != self.device or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.AnoddpmError(f"parameter {k}: the training plan needs contiguous fp32 parameters on {self.device}")
            self.pptr[k] = p.data_ptr()
            if p.grad is not None:
                if p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                    raise _lib.AnoddpmError(f"parameter {k}: .grad must be contiguous fp32")
                self.gptr[k] = p.grad.data_ptr()
                self.gview[k] = None
            else:
                v = self.arena[off:off + p.numel()].view(p.shape)
                off += (p.numel() + 3) // 4 * 4
                self.gptr[k] = v.data_ptr()
                self.gview[k] = v

    def params_match(self):
        """True while every parameter (and its gradient destination, if it has one) still lives where the op lists point."""
        for k, p in self.named.items():
            if p.data_ptr() != self.pptr[k]:
                return False
            if p.grad is not None and p.grad.data_ptr() != self.gptr[k]:
                return False
            if p.grad is None and self.gview[k] is None:
                return False
        return True

    def W(self, key):
        return self.pptr[key]

    def dW(self, key):
        self._touched.add(key)                # backward stage that (last) writes this parameter's gradient: see bwd_marks
        return self.gptr[key]

    def pack(self, key, kind, bwd=0, k0=0, kc=0):
        """Packed copy of a weight, refreshed by an OP_PACK at the head of every forward."""
        ck = (key, kind, bwd, k0, kc)
        hit = self._packs.get(ck)
        if hit is not None:
            return hit
        w = self.named[key]
        N, K = w.shape[0], w.shape[1]
        n = {0: 9 * N * K, 1: 16 * N * K, 2: (N * kc if bwd else N * K), 3: 9 * N * K, 4: N * K, 5: 36 * N * K}[kind]
        out = self.buf(n)
        st = PackArgs()
        st.w, st.out, st.N, st.K, st.kind, st.bwd, st.k0, st.kc = self.W(key), out.data_ptr(), N, K, kind, bwd, k0, kc
        self.keep.append(st)
        self.pack_ops.append((_lib.OP_PACK, st))
        self._packs[ck] = out
        return out

    # ------------------------------------------------------------------ gradient buffers
    def G(self, t):
        g = self._grads.get(t.data_ptr())
        if g is None:
            g = self._grads[t.data_ptr()] = self.buf(*t.shape)
        return g

    def gacc(self, t):
        """0 for the first writer of t's gradient in backward order (overwrite), 1 afterwards (accumulate)."""
        k = t.data_ptr()
        if k in self.gwritten:
            return 1
        self.gwritten.add(k)
        return 0

    def badd(self, code, st):
        self.keep.append(st)
        self.bops.append((code, st))
        return st

    def tws(self, st, field, floats):
        """The op's workspace is a slice of one shared training workspace (ops run in stream order)."""
        self._tws_need = max(self._tws_need, int(floats))
        self._ws_patch.append((st, field))

    # ------------------------------------------------------------------ emitters
    def gn_t(self, srcs, P, prefix):
        """GroupNorm statistics -> (scale, shift, mean, rstd); parameters are read in place."""
        B = self.B
        c0 = srcs[0][1]
        c1 = srcs[1][1] if len(srcs) > 1 else 0
        C = c0 + c1
        for buf, c in [(s[0], s[1]) for s in srcs]:
            if buf.data_ptr() not in self.stats_of:
                self.chan_stats(buf, c, P)
        job = self.gn_tail_job(srcs, self.W(prefix + ".weight"), self.W(prefix + ".bias"), want_mean=True)
        if job is not None:
            return job
        st = GnFinalizeArgs()
        self.stats_source(st, 0, srcs[0][0], c0)
        if c1:
            self.stats_source(st, 1, srcs[1][0], c1)
        else:
            st.stats1, st.rows1, st.fmt1 = None, 0, 0
        st.gamma, st.beta = self.W(prefix + ".weight"), self.W(prefix + ".bias")
        scale, shift, mean, rstd = self.buf(B, C), self.buf(B, C), self.buf(B, 32), self.buf(B, 32)
        st.scale, st.shift, st.mean_out, st.rstd_out = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        st.c0, st.c1, st.P, st.B, st.groups, st.eps = c0, c1, P, B, 32, 1e-5
        self.add(_lib.OP_GN_FINALIZE, st)
        return scale, shift, mean, rstd

    def linear_t(self, inp, wkey, bkey, K, N, act_in):
        st = LinearArgs()
        st.inp, st.w, st.bias = inp.data_ptr(), self.W(wkey), self.W(bkey)
        o = self.buf(self.B, N)
        st.out, st.B, st.K, st.N, st.act_in, st.act_out = o.data_ptr(), self.B, K, N, act_in, 0
        self.add(_lib.OP_LINEAR, st)
        return o

    def linear_bwd(self, x, wkey, bkey, dy, K, N, act_in, dx):
        st = LinearBwdArgs()
        st.x, st.w, st.dy = x.data_ptr(), self.W(wkey), dy.data_ptr()
        st.dw, st.db = self.dW(wkey), self.dW(bkey)
        st.dx = dx.data_ptr() if dx is not None else None
        st.B, st.K, st.N, st.act_in, st.acc_w = self.B, K, N, act_in, 1
        st.acc_x = self.gacc(dx) if dx is not None else 0
        self.badd(_lib.OP_LINEAR_BWD, st)

    def linear_bwd_batch(self, x, jobs, K, dx):
        """Backward of several linear layers that share the input x (anoddpm_linear_small_backward_batch)."""
        if not jobs:
            return
        arr = (LinearBwdArgs * len(jobs))()
        for i, (wkey, bkey, dy, N) in enumerate(jobs):
            arr[i].w, arr[i].dy, arr[i].dw, arr[i].db, arr[i].N = self.W(wkey), dy.data_ptr(), self.dW(wkey), self.dW(bkey), N
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        ws = self.buf(len(jobs) * self.B * K)
        st = LinearBwdBatchArgs()
        st.jobs, st.x, st.dx, st.ws = raw.data_ptr(), x.data_ptr(), dx.data_ptr(), ws.data_ptr()
        st.njobs, st.max_n = len(jobs), max(j[3] for j in jobs)
        st.B, st.K, st.act_in, st.acc_w, st.acc_x = self.B, K, 1, 1, self.gacc(dx)
        self.keep.append(raw)
        self.badd(_lib.OP_LINEAR_BWD_BATCH, st)

    def in_backward(self):
        """Context: igemm / resample ops emitted inside go to the backward list."""
        plan = self

        class _Ctx:
            def __enter__(self):
                self.saved = plan.ops
                plan.ops = plan.bops

            def __exit__(self, *a):
                plan.ops = self.saved
        return _Ctx()

    def wgrad3(self, srcs, Hs, H, gn, a_mode, dy, N, wkey, bkey, d_emb=None):
        """3x3 weight gradient (+ the column sums of dy: bias gradient and, optionally, the per-image embedding gradient)."""
        B = self.B
        c0 = srcs[0][1]
        c1 = srcs[1][1] if len(srcs) > 1 else 0
        K = c0 + c1
        W = H
        tiles = -(-K // 64) * -(-N // 64)
        TW = next(t for t in (32, 16, 8, 4, 2) if W % t == 0)
        per_band = tiles * B * (W // TW)
        nband = max(1, min(H, round(512 / per_band)))
        band = -(-H // nband)
        nband = -(-H // band)
        ipb = (W // TW) * nband
        nitems = B * ipb
        # Winograd-domain weight gradient (csrc/wgrad43.hip: the adjoint of the F(4x4,3x3) forward kernel, 4x fewer MFMAs) on the
        # maps where the forward uses that kernel; ANODDPM_NO_WGRAD43=1 keeps the direct nine-tap kernel everywhere
        # (a plain operand -- dropout output, Downsample / Upsample inputs -- takes the direct kernel: gn is None)
        algo = int(gn is not None and a_mode in (0, 1) and H % 8 == 0 and W % 16 == 0 and K % 32 == 0 and N % 64 == 0 and (c1 == 0 or c0 % 16 == 0)
                   and B <= 15 and H * W >= int(os.environ.get("ANODDPM_WGRAD43_MIN_PIXELS", 16 * 16))   # round 6: the 16x16 maps too (40.55 -> 40.32 ms per config-3 step, profiles/r6_knob_sweep_c3.txt; 32x32 was the limit before)
                   and os.environ.get("ANODDPM_NO_WGRAD43", "0") != "1")
        wa = WgradArgs()
        wa.a0 = srcs[0][0].data_ptr()
        wa.a1 = srcs[1][0].data_ptr() if c1 else None
        wa.gn_scale, wa.gn_shift = (gn[0].data_ptr(), gn[1].data_ptr()) if gn is not None else (None, None)
        wa.dy, wa.dw = dy.data_ptr(), self.dW(wkey)
        wa.algo = algo
        if algo:
            wa.ws_floats = lib().anoddpm_wgrad43_groups(K, N, B, H, W) * 9 * K * N
            ipb = lib().anoddpm_wgrad43_colsum_items(K, N, B, H, W)   # column-sum rows per image: one per workgroup set and tile row
        else:
            wa.ws_floats = nitems * 9 * K * N
        self.tws(wa, "ws", wa.ws_floats)
        Ps = Hs * Hs
        wa.a0_bs, wa.a1_bs, wa.dy_bs = Ps * c0, Ps * c1, H * W * N
        wa.c0, wa.c1, wa.a0_ld, wa.a1_ld, wa.dy_ld = c0, c1, c0, (c1 if c1 else 4), N
        wa.H, wa.W, wa.N, wa.B = H, W, N, B
        wa.a_mode, wa.act, wa.gn_ld, wa.band, wa.accumulate = a_mode, (1 if gn is not None else 0), K, band, 1
        colsum = self.buf(B, ipb, N)
        wa.colsum = colsum.data_ptr()
        if d_emb is None:
            d_emb = self.buf(B, N)                               # per-image sums: scratch when only the bias gradient is wanted
        wa.dimg = wa.dbias = None
        if algo:
            # the fold launch of the Winograd-domain kernel also folds the column sums (round 6: 64 colsum_fold launches less per step)
            wa.dimg, wa.dbias = d_emb.data_ptr(), self.dW(bkey)
            self.badd(_lib.OP_WGRAD3, wa)
            return
        self.badd(_lib.OP_WGRAD3, wa)
        cf = ColsumFoldArgs()
        cf.colsum, cf.dimg, cf.dbias = colsum.data_ptr(), d_emb.data_ptr(), self.dW(bkey)
        cf.B, cf.ipb, cf.N = B, ipb, N
        self.badd(_lib.OP_COLSUM_FOLD, cf)

    def wgrad1(self, srcs, P, gn, act, dy, dy_ld, N, wkey, bkey):
        B = self.B
        c0 = srcs[0][1]
        c1 = srcs[1][1] if len(srcs) > 1 else 0
        K = c0 + c1
        tiles = -(-K // 128) * -(-N // 128)
        want_items = max(1, 512 // tiles)
        span = max(32, (B * P // want_items + 31) // 32 * 32)
        span = min(span, (P + 31) // 32 * 32)
        nitems = B * -(-P // span)
        st = Wgrad1Args()
        st.a0 = srcs[0][0].data_ptr()
        st.a1 = srcs[1][0].data_ptr() if c1 else None
        st.gn_scale = gn[0].data_ptr() if gn else None
        st.gn_shift = gn[1].data_ptr() if gn else None
        st.dy, st.dw, st.dbias = dy if isinstance(dy, int) else dy.data_ptr(), self.dW(wkey), self.dW(bkey)
        st.ws_floats = nitems * (K * N + N)
        self.tws(st, "ws", st.ws_floats)
        st.a0_bs, st.a1_bs, st.dy_bs = P * c0, P * c1, P * dy_ld
        st.c0, st.c1, st.a0_ld, st.a1_ld, st.dy_ld = c0, c1, c0, (c1 if c1 else 4), dy_ld
        st.P, st.N, st.B, st.act, st.gn_ld, st.span, st.accumulate = P, N, B, act, K, span, 1
        self.badd(_lib.OP_WGRAD1, st)

    def gn_bwd(self, srcs, Hs, da, da_P, gnp, prefix, act, a_mode, dres=None, fused=None):
        """Backward of the fused GroupNorm(+SiLU)(+resample) operand load into the gradients of its sources.
        fused = (partial buffer, rows per image) when the data-gradient launch that produced `da` already wrote the
        reduction pass's partial sums (dgrad3(..., gnb=...)): fold + elementwise launches only."""
        B = self.B
        c0 = srcs[0][1]
        c1 = srcs[1][1] if len(srcs) > 1 else 0
        C = c0 + c1
        P = Hs * Hs
        nslab = max(1, min(int(os.environ.get("ANODDPM_GNBWD_SLABS", 256)), P // 16))     # >= 4 workgroups per CU on the large maps
        if fused is not None:
            assert a_mode == 0 and act == 1
            nslab = fused[1]
        ga = GnBwdArgs()
        ga.x0 = srcs[0][0].data_ptr()
        ga.x1 = srcs[1][0].data_ptr() if c1 else None
        ga.da = da.data_ptr()
        ga.gamma, ga.beta = self.W(prefix + ".weight"), self.W(prefix + ".bias")
        ga.mean, ga.rstd = gnp[2].data_ptr(), gnp[3].data_ptr()
        g0 = self.G(srcs[0][0])
        ga.dx0 = g0.data_ptr()
        acc = self.gacc(srcs[0][0])
        if c1:
            g1 = self.G(srcs[1][0])
            ga.dx1 = g1.data_ptr()
            acc |= self.gacc(srcs[1][0]) << 1
        else:
            ga.dx1 = None
        ga.dgamma, ga.dbeta = self.dW(prefix + ".weight"), self.dW(prefix + ".bias")
        part = fused[0] if fused is not None else self.buf(B * nslab * C * 2, dtype=torch.float64)
        coef = self.buf(B * C * 4)
        ga.partial, ga.coef = part.data_ptr(), coef.data_ptr()
        ga.partial_ready = 1 if fused is not None else 0
        ga.x0_bs, ga.x1_bs, ga.da_bs, ga.dx0_bs, ga.dx1_bs = P * c0, P * c1, da_P * C, P * c0, P * c1
        ga.c0, ga.c1, ga.x0_ld, ga.x1_ld, ga.da_ld, ga.dx0_ld, ga.dx1_ld = c0, c1, c0, (c1 if c1 else 4), C, c0, (c1 if c1 else 4)
        ga.Hs, ga.Ws, ga.B, ga.groups, ga.nslab = Hs, Hs, B, 32, nslab
        ga.act, ga.a_mode, ga.acc_dx = act, a_mode, acc
        if dres is not None:
            ga.dres, ga.dres_bs, ga.dres_ld = dres.data_ptr(), P * C, C
        else:
            ga.dres, ga.dres_bs, ga.dres_ld = None, 0, 0
        self.badd(_lib.OP_GN_BWD, ga)

    # ------------------------------------------------------------------ network
    def _build(self):
        m = self.model
        B, S, dev = self.B, self.S, self.device
        base, ted = m.model_channels, m._ted
        down, middle, up = m._blocks
        self.post_pack = []
        self._bind_params()
        bias = self.W

        # --- timestep path (UNet.py:271-276): pre-activations are kept, SiLU rides on the next layer's input
        half = base // 2
        freqs = _posemb_freqs(half).to(dev)
        self.keep.append(freqs)
        pe = self.buf(B, base)
        self.posemb = PosembArgs()
        self.posemb.t, self.posemb.freqs, self.posemb.out = None, freqs.data_ptr(), pe.data_ptr()
        self.posemb.B, self.posemb.dim, self.posemb.scale = B, base, 1.0
        self.add(_lib.OP_POSEMB, self.posemb)
        z1 = self.linear_t(pe, "time_embedding.1.weight", "time_embedding.1.bias", base, ted, 0)
        temb = self.linear_t(z1, "time_embedding.3.weight", "time_embedding.3.bias", ted, ted, 1)
        self.temb = temb
        g_temb, g_z1 = self.G(temb), self.G(z1)

        # Forward: every ResBlock's embedding projection in ONE launch (round 6: 22 launches of ~8 us at config 3) when the projections'
        # weights and biases are consecutive in memory in one order -- training.FlatBuffers stores them that way; a model with
        # separately allocated parameters keeps one launch per block.  linear_small computes one output feature per wave, so the
        # values are the per-block launches' bit for bit.
        self._emb_all, self._emb_off, self._emb_tot = None, {}, 0
        ekeys = [k[:-len(".weight")] for k in self.named if k.endswith(".embed_layers.1.weight")]
        if len(ekeys) > 1 and os.environ.get("ANODDPM_BATCH_EMB_FWD", "1") != "0":
            wp, bp, ok = self.W(ekeys[0] + ".weight"), self.W(ekeys[0] + ".bias"), True
            for k in ekeys:
                w = self.named[k + ".weight"]
                ok = ok and self.W(k + ".weight") == wp and self.W(k + ".bias") == bp and w.shape[1] == ted
                self._emb_off[k[:-len(".embed_layers.1")]] = self._emb_tot
                wp += 4 * w.numel()
                bp += 4 * w.shape[0]
                self._emb_tot += w.shape[0]
            if ok:
                st = LinearArgs()
                st.inp, st.w, st.bias = temb.data_ptr(), self.W(ekeys[0] + ".weight"), self.W(ekeys[0] + ".bias")
                self._emb_all = self.buf(B, self._emb_tot)
                st.out, st.B, st.K, st.N, st.act_in, st.act_out = self._emb_all.data_ptr(), B, ted, self._emb_tot, 1, 0
                self.add(_lib.OP_LINEAR, st)

        def emb_of(prefix, cout):
            """(pointer, row pitch) of a block's embedding projection: a column range of the batched launch, or its own launch."""
            if self._emb_all is not None:
                return self._emb_all.data_ptr() + 4 * self._emb_off[prefix], self._emb_tot
            return self.linear_t(temb, prefix + ".embed_layers.1.weight", prefix + ".embed_layers.1.bias", ted, cout, 1).data_ptr(), cout

        self._emb_jobs = []          # (weight key, bias key, d_emb buffer, cout) of every ResBlock's embedding projection
        # One batched launch for all of them at the END of the backward.  With a data-parallel reducer attached that is still
        # right: training.FlatBuffers stores the embedding projections (and the timestep MLP) at the bottom of the flat buffer,
        # so they share the last gradient bucket(s) and no other bucket's all-reduce waits for them.
        self._batch_emb = os.environ.get("ANODDPM_BATCH_EMB_BWD", "1") != "0"

        def time_bwd():
            self.linear_bwd_batch(temb, self._emb_jobs, ted, g_temb)        # all embedding projections at once
            self.linear_bwd(z1, "time_embedding.3.weight", "time_embedding.3.bias", g_temb, ted, ted, 1, g_z1)
            self.linear_bwd(pe, "time_embedding.1.weight", "time_embedding.1.bias", g_z1, base, ted, 0, None)
        self._bw.append(time_bwd)

        def conv3(srcs, Hout, N, gnp, a_mode, wkey, bkey, out, temb_ptr=None, temb_ld=0, res=None, res_up=None):
            self.igemm(srcs=srcs, H=Hout, W=Hout, ks=3, N=N, gn=(gnp[0], gnp[1]), act=1, a_mode=a_mode,
                       bmat=lambda: self.pack(wkey, 0), wino=lambda: self.pack(wkey, 1), wino43=lambda: self.pack(wkey, 5),
                       bias=bias(bkey),
                       temb=temb_ptr, temb_ld=temb_ld, res=res, res_up=res_up, out=out, want_stats=True)

        fuse_gnb = os.environ.get("ANODDPM_NO_GNB_FUSE", "0") != "1"

        def dgrad3(dy, Hout, Kc, N, wkey, gnb=None):
            """da [B][Hout^2][Kc] = conv3x3(dy, flipped / transposed weights): the forward kernels on the packed twin.
            gnb = (sources, gn_t tuple, GroupNorm prefix) of the operand a = SiLU(GroupNorm(x)) this gradient belongs to: where the
            launch runs on the channel-sliced F(4x4,3x3) kernel its epilogue also writes the partial sums of the GroupNorm
            backward's reduction pass (round 6: one pass over x and da less per layer); returns (da, fused) for gn_bwd."""
            da = self.buf(B, Hout * Hout, Kc)
            st = self.igemm(srcs=[(dy, N)], H=Hout, W=Hout, ks=3, N=Kc, bmat=lambda: self.pack(wkey, 0, bwd=1),
                            wino=lambda: self.pack(wkey, 1, bwd=1), wino43=lambda: self.pack(wkey, 5, bwd=1), out=da)
            st.gnb_partial = None
            if gnb is None:
                return da
            fused = None
            xs, gnp, gprefix = gnb
            xc0 = xs[0][1]
            if (fuse_gnb and st.cfg == 3 and st.ksplit == 1 and xc0 % 16 == 0 and sum(x[1] for x in xs) == Kc
                    and lib().anoddpm_f43_channel_sliced(Hout, Hout, Kc, B) == 1):
                rows = (Hout // 16) * (Hout // 16)
                part = self.buf(B * rows * Kc * 2, dtype=torch.float64)
                st.gnb_partial = part.data_ptr()
                st.gnb_x0 = xs[0][0].data_ptr()
                st.gnb_x1 = xs[1][0].data_ptr() if len(xs) > 1 else None
                st.gnb_gamma, st.gnb_beta = self.W(gprefix + ".weight"), self.W(gprefix + ".bias")
                st.gnb_mean, st.gnb_rstd = gnp[2].data_ptr(), gnp[3].data_ptr()
                P = Hout * Hout
                xc1 = xs[1][1] if len(xs) > 1 else 0
                st.gnb_x0_bs, st.gnb_x1_bs = P * xc0, P * xc1
                st.gnb_c0, st.gnb_x0_ld, st.gnb_x1_ld, st.gnb_groups = xc0, xc0, (xc1 if xc1 else 4), 32
                fused = (part, rows)
            return da, fused

        def res_block(prefix, srcs, Hin, cout, resample):
            cin = sum(s[1] for s in srcs)
            Hout = Hin * 2 if resample == "up" else (Hin // 2 if resample == "down" else Hin)
            Pin, Pout = Hin * Hin, Hout * Hout
            am = {None: 0, "up": 1, "down": 2}[resample]
            g1 = self.gn_t(srcs, Pin, prefix + ".in_layers.0")
            emb_ptr, emb_ld = emb_of(prefix, cout)
            h1 = self.buf(B, Pout, cout)
            sk_pool = None
            if resample == "down":
                # one pass over x: pooled skip input + pooled ACTIVATED conv operand (so the conv runs on the Winograd kernels);
                # the backward is unchanged -- weight gradient and GroupNorm backward re-apply the pooling on the raw input
                assert len(srcs) == 1
                pooled, sk_pool = self.buf(B, Pout, cin), self.buf(B, Pout, cin)
                st = ResampleArgs()
                st.inp, st.out = srcs[0][0].data_ptr(), sk_pool.data_ptr()
                st.B, st.H, st.W, st.C, st.mode, st.scale, st.accumulate = B, Hin, Hin, cin, 2, 1.0, 0
                st.gn_scale, st.gn_shift, st.out_act = g1[0].data_ptr(), g1[1].data_ptr(), pooled.data_ptr()
                self.add(_lib.OP_RESAMPLE, st)
                wk, bk = prefix + ".in_layers.2.weight", prefix + ".in_layers.2.bias"
                self.igemm(srcs=[(pooled, cin)], H=Hout, W=Hout, ks=3, N=cout, bmat=lambda: self.pack(wk, 0),
                           wino=lambda: self.pack(wk, 1), wino43=lambda: self.pack(wk, 5), bias=bias(bk),
                           temb=emb_ptr, temb_ld=emb_ld, out=h1, want_stats=True)
            else:
                conv3(srcs, Hout, cout, g1, am, prefix + ".in_layers.2.weight", prefix + ".in_layers.2.bias", h1,
                      temb_ptr=emb_ptr, temb_ld=emb_ld)
            g2 = self.gn_t([(h1, cout)], Pout, prefix + ".out_layers.0")
            skip_kind = "identity"
            if cin != cout:
                assert resample is None
                skip_kind = "conv"
                sk = self.buf(B, Pout, cout)
                self.igemm(srcs=srcs, H=Hout, W=Hout, ks=1, N=cout, kind="conv1",
                           bmat=lambda: self.pack(prefix + ".skip_connection.weight", 2),
                           bias=bias(prefix + ".skip_connection.bias"), out=sk)
            elif resample is not None and sk_pool is not None:
                skip_kind = "resample"
                sk = sk_pool
            elif resample == "up" and self.p_drop == 0:
                # nearest x2 of the block input as the residual: read at half resolution by the F(4x4) epilogue (igemm(res_up=...)
                # materialises it for every other kernel); the backward of this path is the pooling of gh2 below, unchanged
                assert len(srcs) == 1
                skip_kind = "resample"
                sk = None
            elif resample is not None:
                assert len(srcs) == 1
                skip_kind = "resample"
                sk = self.buf(B, Pout, cout)
                st = ResampleArgs()
                st.inp, st.out = srcs[0][0].data_ptr(), sk.data_ptr()
                st.B, st.H, st.W, st.C, st.mode, st.scale, st.accumulate = B, Hin, Hin, cin, (1 if resample == "up" else 2), 1.0, 0
                self.add(_lib.OP_RESAMPLE, st)
            else:
                if len(srcs) != 1:
                    raise NotImplementedError("identity skip over a concatenated input (cin == cout) is not built")
                sk = srcs[0][0]
            h2 = self.buf(B, Pout, cout)
            a2 = None
            if self.p_drop > 0:
                # Dropout between the activation and the convolution: the dropped activation is materialised once (one
                # elementwise launch) and the convolution, its weight gradient and its data gradient see a plain operand
                a2 = self.buf(B, Pout, cout)
                dr = DropoutArgs()
                dr.x, dr.out, dr.gn_scale, dr.gn_shift = h1.data_ptr(), a2.data_ptr(), g2[0].data_ptr(), g2[1].data_ptr()
                dr.n, dr.B, dr.C, dr.mode, dr.p, dr.seed = Pout * cout, B, cout, 0, self.p_drop, 0
                self.add(_lib.OP_DROPOUT, dr)
                drop_entry = [dr, None, len(self._drop_ops)]         # forward order; the backward twin is filled in below
                self._drop_ops.append(drop_entry)
                wk, bk = prefix + ".out_layers.3.weight", prefix + ".out_layers.3.bias"
                self.igemm(srcs=[(a2, cout)], H=Hout, W=Hout, ks=3, N=cout, bmat=lambda: self.pack(wk, 0),
                           wino=lambda: self.pack(wk, 1), wino43=lambda: self.pack(wk, 5), bias=bias(bk), res=sk, out=h2, want_stats=True)
            else:
                conv3([(h1, cout)], Hout, cout, g2, 0, prefix + ".out_layers.3.weight", prefix + ".out_layers.3.bias", h2, res=sk,
                      res_up=(srcs[0][0] if (resample == "up" and sk is None) else None))

            def bwd():
                gh2 = self.G(h2)
                with self.in_backward():
                    # 1. skip path
                    if skip_kind == "conv":
                        self.wgrad1(srcs, Pin, None, 0, gh2, cout, cout, prefix + ".skip_connection.weight",
                                    prefix + ".skip_connection.bias")
                        k0 = 0
                        for (src, c) in [(s[0], s[1]) for s in srcs]:
                            gs = self.G(src)
                            acc = self.gacc(src)
                            self.igemm(srcs=[(gh2, cout)], H=Hout, W=Hout, ks=1, N=c, kind="conv1",
                                       bmat=self.pack(prefix + ".skip_connection.weight", 2, bwd=1, k0=k0, kc=c),
                                       res=(gs if acc else None), out=gs)
                            k0 += c
                    elif skip_kind == "resample":
                        src = srcs[0][0]
                        st = ResampleArgs()
                        st.inp, st.out = gh2.data_ptr(), self.G(src).data_ptr()
                        st.B, st.H, st.W, st.C = B, Hout, Hout, cout
                        # forward nearest-up -> backward sums the four children (avg pool * 4); forward avg pool -> nearest-up / 4
                        st.mode, st.scale = (2, 4.0) if resample == "up" else (1, 0.25)
                        st.accumulate = self.gacc(src)
                        self.add(_lib.OP_RESAMPLE, st)
                    # 2-4. out_layers: weight gradient, data gradient, GroupNorm + SiLU backward into g(h1)
                    if a2 is not None:
                        self.wgrad3([(a2, cout)], Hout, Hout, None, 0, gh2, cout, prefix + ".out_layers.3.weight", prefix + ".out_layers.3.bias")
                        da2, fused2 = dgrad3(gh2, Hout, cout, cout, prefix + ".out_layers.3.weight"), None
                        db = DropoutArgs()                           # d(activation) = mask / (1 - p) * d(dropped), in place
                        db.x, db.out, db.n, db.B, db.C, db.mode, db.p, db.seed = da2.data_ptr(), da2.data_ptr(), Pout * cout, B, cout, 1, self.p_drop, 0
                        self.add(_lib.OP_DROPOUT, db)
                        drop_entry[1] = db
                    else:
                        self.wgrad3([(h1, cout)], Hout, Hout, g2, 0, gh2, cout, prefix + ".out_layers.3.weight", prefix + ".out_layers.3.bias")
                        da2, fused2 = dgrad3(gh2, Hout, cout, cout, prefix + ".out_layers.3.weight",
                                             gnb=([(h1, cout)], g2, prefix + ".out_layers.0"))
                    self.gn_bwd([(h1, cout)], Hout, da2, Pout, g2, prefix + ".out_layers.0", 1, 0, fused=fused2)
                    gh1 = self.G(h1)
                    # 5. in_layers weight gradient; its dy column sums are the conv bias and the embedding gradients
                    d_emb = self.buf(B, cout)
                    self.wgrad3(srcs, Hin, Hout, g1, am, gh1, cout, prefix + ".in_layers.2.weight", prefix + ".in_layers.2.bias", d_emb=d_emb)
                    if self._batch_emb:
                        self._emb_jobs.append((prefix + ".embed_layers.1.weight", prefix + ".embed_layers.1.bias", d_emb, cout))
                    else:
                        self.linear_bwd(temb, prefix + ".embed_layers.1.weight", prefix + ".embed_layers.1.bias", d_emb, ted, cout, 1, g_temb)
                    # 6-7. data gradient and the fused operand load's backward into the block inputs
                    if am == 0:
                        da1, fused1 = dgrad3(gh1, Hout, cin, cout, prefix + ".in_layers.2.weight", gnb=(srcs, g1, prefix + ".in_layers.0"))
                    else:                                            # a resampling sits between the activation and the convolution
                        da1, fused1 = dgrad3(gh1, Hout, cin, cout, prefix + ".in_layers.2.weight"), None
                    self.gn_bwd(srcs, Hin, da1, Pout, g1, prefix + ".in_layers.0", 1, am,
                                dres=(gh2 if skip_kind == "identity" else None), fused=fused1)
            self._bw.append(bwd)
            return h2, Hout

        def attn_block(prefix, x, Hc, C):
            L = Hc * Hc
            heads = m._heads_for(C)
            ch = C // heads
            if ch % 4:
                raise NotImplementedError(f"attention head width {ch} must be a multiple of 4")
            Z = B * heads
            alpha = 1.0 / math.sqrt(ch)
            g = self.gn_t([(x, C)], L, prefix + ".norm")
            qkv = self.buf(B, L, 3 * C)
            self.igemm(srcs=[(x, C)], H=Hc, W=Hc, ks=1, N=3 * C, gn=(g[0], g[1]), act=0, kind="qkvproj",
                       bmat=lambda: self.pack(prefix + ".to_qkv.weight", 2), bias=bias(prefix + ".to_qkv.bias"), out=qkv)
            Pm = self.buf(Z, L, L)                               # softmax output, kept for the backward
            qp = qkv.data_ptr()
            att = self.buf(B, L, C)
            if not self.attention(qkv, att, L, heads, ch, probs=Pm):
                self.igemm(srcs=[(qp, ch, 3 * C)], H=1, W=L, ks=1, N=L, b_mode=1, ldb=3 * C, heads=heads,
                           bmat=qp + 4 * ch, alpha=alpha, kind="attn",
                           a_strides=(L * 3 * C, 3 * ch), b_strides=(L * 3 * C, 3 * ch),
                           out=Pm, out_ld=L, o_strides=(heads * L * L, L * L))
                sm = SoftmaxArgs()
                sm.x, sm.rows, sm.L = Pm.data_ptr(), Z * L, L
                self.add(_lib.OP_SOFTMAX, sm)
                self.igemm(srcs=[(Pm.data_ptr(), L, L)], H=1, W=L, ks=1, N=ch, b_mode=2, ldb=3 * C, heads=heads,
                           bmat=qp + 4 * 2 * ch, kind="attn",
                           a_strides=(heads * L * L, L * L), b_strides=(L * 3 * C, 3 * ch),
                           out=att, out_ld=C, o_strides=(L * C, ch))
            y = self.buf(B, L, C)
            self.igemm(srcs=[(att, C)], H=Hc, W=Hc, ks=1, N=C, kind="qkvproj",
                       bmat=lambda: self.pack(prefix + ".proj_out.weight", 2), bias=bias(prefix + ".proj_out.bias"),
                       res=x, out=y, want_stats=True)

            def bwd():
                gy = self.G(y)
                with self.in_backward():
                    # proj_out: dW, d(att)
                    self.wgrad1([(att, C)], L, None, 0, gy, C, C, prefix + ".proj_out.weight", prefix + ".proj_out.bias")
                    datt = self.buf(B, L, C)
                    self.igemm(srcs=[(gy, C)], H=Hc, W=Hc, ks=1, N=C, kind="qkvproj",
                               bmat=self.pack(prefix + ".proj_out.weight", 2, bwd=1, k0=0, kc=C), out=datt)
                    dqkv = self.buf(B, L, 3 * C)
                    dq = dqkv.data_ptr()
                    dP = self.buf(Z, L, L)
                    T1 = self.buf(Z, L, L)
                    # dP = d(att)_h v_h^T
                    self.igemm(srcs=[(datt.data_ptr(), ch, C)], H=1, W=L, ks=1, N=L, b_mode=1, ldb=3 * C, heads=heads,
                               bmat=qp + 4 * 2 * ch, kind="attn", a_strides=(L * C, ch), b_strides=(L * 3 * C, 3 * ch),
                               out=dP, out_ld=L, o_strides=(heads * L * L, L * L))
                    # dV_h = P^T d(att)_h
                    tr = TransposeArgs()
                    tr.inp, tr.out, tr.Z, tr.L = Pm.data_ptr(), T1.data_ptr(), Z, L
                    self.add(_lib.OP_TRANSPOSE, tr)
                    self.igemm(srcs=[(T1.data_ptr(), L, L)], H=1, W=L, ks=1, N=ch, b_mode=2, ldb=C, heads=heads,
                               bmat=datt.data_ptr(), kind="attn", a_strides=(heads * L * L, L * L), b_strides=(L * C, ch),
                               out=dq + 4 * 2 * ch, out_ld=3 * C, o_strides=(L * 3 * C, 3 * ch))
                    # dS = P o (dP - rowsum(dP o P)), in place
                    sb = SoftmaxBwdArgs()
                    sb.p, sb.dp, sb.rows, sb.L = Pm.data_ptr(), dP.data_ptr(), Z * L, L
                    self.add(_lib.OP_SOFTMAX_BWD, sb)
                    # dQ_h = alpha dS k_h ; dK_h = alpha dS^T q_h
                    self.igemm(srcs=[(dP.data_ptr(), L, L)], H=1, W=L, ks=1, N=ch, b_mode=2, ldb=3 * C, heads=heads,
                               bmat=qp + 4 * ch, alpha=alpha, kind="attn", a_strides=(heads * L * L, L * L),
                               b_strides=(L * 3 * C, 3 * ch), out=dq, out_ld=3 * C, o_strides=(L * 3 * C, 3 * ch))
                    tr2 = TransposeArgs()
                    tr2.inp, tr2.out, tr2.Z, tr2.L = dP.data_ptr(), T1.data_ptr(), Z, L
                    self.add(_lib.OP_TRANSPOSE, tr2)
                    self.igemm(srcs=[(T1.data_ptr(), L, L)], H=1, W=L, ks=1, N=ch, b_mode=2, ldb=3 * C, heads=heads,
                               bmat=qp, alpha=alpha, kind="attn", a_strides=(heads * L * L, L * L),
                               b_strides=(L * 3 * C, 3 * ch), out=dq + 4 * ch, out_ld=3 * C, o_strides=(L * 3 * C, 3 * ch))
                    # to_qkv: dW (input = GroupNorm(x), no SiLU), d(normed x), GroupNorm backward + the residual
                    self.wgrad1([(x, C)], L, g, 0, dqkv, 3 * C, 3 * C, prefix + ".to_qkv.weight", prefix + ".to_qkv.bias")
                    da = self.buf(B, L, C)
                    self.igemm(srcs=[(dqkv, 3 * C)], H=Hc, W=Hc, ks=1, N=C, kind="qkvproj",
                               bmat=self.pack(prefix + ".to_qkv.weight", 2, bwd=1, k0=0, kc=C), out=da)
                    self.gn_bwd([(x, C)], Hc, da, L, g, prefix + ".norm", 0, 0, dres=gy)
            self._bw.append(bwd)
            return y

        def run(blks, srcs, Hc):
            for (prefix, kind, cin, cout, resample) in blks:
                if kind == "stem":
                    h0 = self.buf(B, S * S, cout)
                    self.stem = StemArgs()
                    self.stem.x = None
                    self.stem.w = self.pack(prefix + ".weight", 3).data_ptr()
                    self.stem.bias = bias(prefix + ".bias")
                    self.stem.out = h0.data_ptr()
                    self.stem.B, self.stem.H, self.stem.W, self.stem.Cin, self.stem.Cout = B, S, S, cin, cout
                    rows = lib().anoddpm_stem_stats_rows(S, S, cin, cout)
                    if rows > 0 and os.environ.get("ANODDPM_NO_STEM_STATS", "0") != "1":
                        # GroupNorm partial sums of the stem output from the stem kernel itself (no chan_stats pass over it)
                        sstats = self.buf(B, rows, cout, 2)
                        self.stem.stats, self.stem.stats_rows = sstats.data_ptr(), rows
                        self.stats_of[h0.data_ptr()] = ("rows", sstats, rows)
                    self.add(_lib.OP_STEM, self.stem)
                    srcs = [(h0, cout)]

                    def stem_bwd(prefix=prefix, h0=h0, cin=cin, cout=cout):
                        sb = StemBwdArgs()
                        self.stem_bwd_args = sb
                        sb.x, sb.w, sb.dy = None, self.W(prefix + ".weight"), self.G(h0).data_ptr()
                        sb.dw, sb.db = self.dW(prefix + ".weight"), self.dW(prefix + ".bias")
                        self.dx = self.buf(B, cin, S, S) if self.want_dx else None
                        sb.dx = self.dx.data_ptr() if self.want_dx else None
                        nblk = B * -(-(S * S) // 1024)
                        sb.ws_floats = nblk * (cin * 9 + 1) * cout
                        self.tws(sb, "ws", sb.ws_floats)
                        sb.B, sb.H, sb.W, sb.Cin, sb.Cout = B, S, S, cin, cout
                        self.badd(_lib.OP_STEM_BWD, sb)
                    self._bw.append(stem_bwd)
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

        def resample_layer(prefix, kind, x, Hc, C, conv):
            """Downsample / Upsample of the biggan_updown=False topology (UNet.py:60-92): raw activations in (no norm, no activation).
            Forward as the inference plan (unet._Plan); backward: the weight gradient on the plain operand, the data gradient as
            the forward kernel on the flipped weights, and the adjoint of the resampling around it."""
            def rs(inp, Hin, mode, out, scale=1.0, acc=0):
                st = ResampleArgs()
                st.inp, st.out = inp.data_ptr(), out.data_ptr()
                st.B, st.H, st.W, st.C, st.mode, st.scale, st.accumulate = B, Hin, Hin, C, mode, scale, acc
                self.add(_lib.OP_RESAMPLE, st)

            def into_gx(dy, H, wkey):
                """dx (+)= conv3x3(dy, flipped weights), straight into the gradient buffer of x"""
                gx = self.G(x)
                acc = self.gacc(x)
                self.igemm(srcs=[(dy, C)], H=H, W=H, ks=3, N=C, bmat=lambda: self.pack(wkey, 0, bwd=1),
                           wino=lambda: self.pack(wkey, 1, bwd=1), wino43=lambda: self.pack(wkey, 5, bwd=1),
                           res=(gx if acc else None), out=gx)

            if kind == "downsample":
                Ho = Hc // 2
                out = self.buf(B, Ho * Ho, C)
                if not conv:
                    rs(x, Hc, 2, out)                                   # nn.AvgPool2d(2, 2)

                    def bwd():
                        with self.in_backward():
                            rs(self.G(out), Ho, 1, self.G(x), 0.25, self.gacc(x))
                    self._bw.append(bwd)
                    return out, Ho
                wk, bk = prefix + ".downsample.weight", prefix + ".downsample.bias"
                full = self.buf(B, Hc * Hc, C)                          # the stride-1 result; the stride-2 output = its even pixels
                self.igemm(srcs=[(x, C)], H=Hc, W=Hc, ks=3, N=C, act=0, bmat=lambda: self.pack(wk, 0), wino=lambda: self.pack(wk, 1),
                           wino43=lambda: self.pack(wk, 5), bias=bias(bk), out=full)
                rs(full, Hc, 3, out)

                def bwd():
                    gfull = self.buf(B, Hc * Hc, C)
                    with self.in_backward():
                        rs(self.G(out), Ho, 4, gfull)                   # gradient on the stride-1 grid: zeros off the even pixels
                        self.wgrad3([(x, C)], Hc, Hc, None, 0, gfull, C, wk, bk)
                        into_gx(gfull, Hc, wk)
                self._bw.append(bwd)
                return out, Ho
            Ho = Hc * 2
            out = self.buf(B, Ho * Ho, C)
            if not conv:
                rs(x, Hc, 1, out)                                       # F.interpolate(scale_factor=2, mode="nearest")

                def bwd():
                    with self.in_backward():
                        rs(self.G(out), Ho, 2, self.G(x), 4.0, self.gacc(x))
                self._bw.append(bwd)
                return out, Ho
            wk, bk = prefix + ".conv.weight", prefix + ".conv.bias"
            self.igemm(srcs=[(x, C)], H=Ho, W=Ho, ks=3, N=C, act=0, a_mode=1, bmat=lambda: self.pack(wk, 0), wino=lambda: self.pack(wk, 1),
                       wino43=lambda: self.pack(wk, 5), bias=bias(bk), out=out, want_stats=True)

            def bwd():
                with self.in_backward():
                    self.wgrad3([(x, C)], Hc, Ho, None, 1, self.G(out), C, wk, bk)
                    dup = dgrad3(self.G(out), Ho, C, C, wk)             # gradient of the upsampled operand, at the output resolution
                    rs(dup, Ho, 2, self.G(x), 4.0, self.gacc(x))        # nearest x2 -> the sum of the four children
            self._bw.append(bwd)
            return out, Ho

        Hc = S
        srcs = None
        skips = []
        for blk in down:
            srcs, Hc = run(blk, srcs, Hc)
            skips.append(srcs[0])
        srcs, Hc = run(middle, srcs, Hc)
        for blk in up:
            srcs, Hc = run(blk, [srcs[0], skips.pop()], Hc)

        # --- head (UNet.py:384-388, 405)
        hfin, cfin = srcs[0]
        gh = self.gn_t([(hfin, cfin)], S * S, "out.0")
        nout = m.in_channels
        self.y = self.buf(B, nout, S, S)
        st = HeadArgs()
        st.x, st.w, st.bias = hfin.data_ptr(), self.pack("out.2.weight", 3).data_ptr(), bias("out.2.bias")
        st.gn_scale, st.gn_shift, st.out = gh[0].data_ptr(), gh[1].data_ptr(), self.y.data_ptr()
        st.B, st.H, st.W, st.C, st.Cout = B, S, S, cfin, nout
        self.add(_lib.OP_HEAD, st)
        self.dy = self.buf(B, nout, S, S)

        def head_bwd():
            hb = HeadBwdArgs()
            da = self.buf(B, S * S, cfin)
            hb.x, hb.gn_scale, hb.gn_shift = hfin.data_ptr(), gh[0].data_ptr(), gh[1].data_ptr()
            hb.w, hb.dy, hb.da = self.W("out.2.weight"), self.dy.data_ptr(), da.data_ptr()
            hb.dw, hb.db = self.dW("out.2.weight"), self.dW("out.2.bias")
            nblk = B * -(-(S * S) // 512)
            hb.ws_floats = nblk * 10 * nout * cfin
            self.tws(hb, "ws", hb.ws_floats)
            hb.B, hb.H, hb.W, hb.C, hb.Cout = B, S, S, cfin, nout
            self.badd(_lib.OP_HEAD_BWD, hb)
            self.gn_bwd([(hfin, cfin)], S, da, S * S, gh, "out.0", 1, 0)
        self._bw.append(head_bwd)

        # --- backward list: the stages in reverse
        for fn in reversed(self._bw):
            self._touched = set()
            fn()
            self.bwd_marks.append((len(self.bops), sorted(self._touched)))
        # split-K workspace of the inference emitters (forward and backward igemm launches) + the training workspace
        if self._ws_need:
            ws = self.buf(self._ws_need)
            for code, st in self.ops + self.bops:
                if code == _lib.OP_IGEMM and st.ksplit > 1:
                    st.ws = ws.data_ptr()
        if self._tws_need:
            tws = self.buf(self._tws_need)
            for st, field in self._ws_patch:
                setattr(st, field, tws.data_ptr())

    # ------------------------------------------------------------------ execution
    def run_forward(self, x, t):
        """x: contiguous fp32 [B,C,S,S] on self.device (kept alive by the caller until the backward); t: int64 [B]."""
        self.stem.x = x.data_ptr()
        self.posemb.t = t.data_ptr()
        if self._drop_ops:
            # a fresh mask per forward and per layer: the seed is one draw from torch's CPU generator (so torch.manual_seed
            # between steps matters, as it does for nn.Dropout, and the stream advances per forward) mixed with the
            # data-parallel rank (replicas seeded alike must not drop the same units) and the layer index; the forward launch
            # and the backward launch of a layer get the same value
            draw = int(torch.randint(0, 2 ** 62, (1,)).item())
            rank = 0
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                rank = torch.distributed.get_rank()
            base = (draw * 0x9E3779B97F4A7C15 + (rank + 1) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
            for fwd, bwd, idx in self._drop_ops:
                fwd.seed = (base + idx * 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF
                if bwd is not None:
                    bwd.seed = fwd.seed
        check(lib().anoddpm_run_ops(self.fwd_array, len(self.fwd_list), _lib.current_stream()), "UNet training forward")
        return self.y

    def run_backward(self, x, dy):
        """Accumulates every parameter gradient into its destination; returns d(x) when the plan was built with want_dx."""
        fresh = [k for k, p in self.named.items() if p.grad is None]
        if fresh:
            if len(fresh) == len(self.named):
                self.arena.zero_()
            else:
                for k in fresh:
                    self.gview[k].zero_()
        self.dy.copy_(dy.reshape(self.dy.shape))
        self.stem_bwd_args.x = x.data_ptr()
        from .training import reducer_of
        red = reducer_of(self.model)
        if red is not None and any(self.named[k].requires_grad for k in fresh):
            # (frozen parameters never get a .grad and are not in the flat buffers: they do not count)
            # every rank must cut its backward at the same places (the collectives are issued in bucket order): with a reducer
            # attached the gradients are bound to the flat views before the forward (UNetModel._forward_autograd), so a
            # missing .grad here means they were dropped between forward and backward
            raise _lib.AnoddpmError("UNetModel training plan: .grad of some parameters was set to None between forward and backward "
                                    "while a GradAllReducer is attached")
        if red is not None:
            # data parallel: the op list is cut where a gradient bucket becomes final, and the bucket's all-reduce is enqueued
            # right there -- RCCL's stream picks it up behind the ops already launched and runs beside the rest of the backward
            lo = 0
            for hi, buckets in self.reduce_schedule(red):
                if hi > lo:
                    check(lib().anoddpm_run_ops(self._bwd_slice(lo), hi - lo, _lib.current_stream()),
                          "UNet training backward")
                for b in buckets:
                    red.launch_bucket(b, ops_done=hi)
                lo = hi
            if lo < len(self.bops):
                check(lib().anoddpm_run_ops(self._bwd_slice(lo), len(self.bops) - lo,
                                            _lib.current_stream()), "UNet training backward")
        else:
            check(lib().anoddpm_run_ops(self.bwd_array, len(self.bops), _lib.current_stream()), "UNet training backward")
        for k in fresh:
            if self.named[k].requires_grad:                  # frozen parameters: their gradient stays plan-owned scratch
                self.named[k].grad = self.gview[k]
        return self.dx

    def _bwd_slice(self, lo):
        return ctypes.cast(ctypes.addressof(self.bwd_array) + lo * ctypes.sizeof(Op), ctypes.POINTER(Op))

    def reduce_schedule(self, red):
        """[(backward op count, [bucket ids])]: after that many ops every gradient of those buckets of `red`
        (training.GradAllReducer) is final.  A parameter's gradient is final after the LAST backward stage that wrote it
        (bwd_marks); parameters no stage writes (unused) are final from the start."""
        key = (red.bounds, tuple(red.flat.names))      # not id(red): CPython reuses ids, a stale schedule would reduce too early
        hit = getattr(self, "_sched", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        final = {}
        for end, keys in self.bwd_marks:
            for k in keys:
                final[k] = end
        names = red.flat.names
        at = {}
        for b, (_, _, members) in enumerate(red.buckets):
            end = max([final.get(names[i], 0) for i in members] + [0])
            at.setdefault(end, []).append(b)
        sched = sorted(at.items())
        self._sched = (key, sched)
        return sched


class TrainPlanFunction(torch.autograd.Function):
    """y = UNet(x, t) on the training plan.  `anchor` is any tensor that requires grad (a parameter): it makes autograd
    call backward() even when x needs no gradient; parameter gradients are accumulated in place by the kernels, so
    backward() returns no gradient for it."""

    @staticmethod
    def forward(ctx, x, t, anchor, plan):
        xin = x.detach()
        if xin.dtype != torch.float32 or not xin.is_contiguous():
            xin = xin.float().contiguous()
        if xin.data_ptr() % 16:
            xin = xin.clone()                        # a view at an odd storage offset: the stem kernel reads 16-byte rows
        tt = t.detach()
        if tt.dtype != torch.int64 or tt.device != x.device or not tt.is_contiguous():
            tt = tt.to(device=x.device, dtype=torch.int64).contiguous()
        y = plan.run_forward(xin, tt)
        plan.fwd_epoch = getattr(plan, "fwd_epoch", 0) + 1
        ctx.plan, ctx.xin, ctx.epoch, ctx.x_dtype = plan, xin, plan.fwd_epoch, x.dtype
        return y.clone().to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        plan = ctx.plan
        if ctx.epoch != plan.fwd_epoch:
            raise _lib.AnoddpmError("UNetModel training plan: backward() of a forward whose activations were overwritten by a "
                                    "later forward of the same (batch, size); run one forward/backward pair at a time")
        if not plan.params_match():
            raise _lib.AnoddpmError("UNetModel training plan: parameters or their .grad tensors moved between forward and backward")
        dx = plan.run_backward(ctx.xin, dy.detach().float().contiguous())
        return (dx.clone().to(ctx.x_dtype) if (ctx.needs_input_grad[0] and dx is not None) else None), None, None, None
