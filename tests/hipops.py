"""Thin test-side wrappers that call single C-ABI ops (include/anoddpm_hip.h) on torch CUDA tensors."""
import ctypes
import math

import torch

from anoddpm_amd import _lib
from anoddpm_amd._lib import (HeadArgs, ChanStatsArgs, GnFinalizeArgs, GnArgs, IgemmArgs, LinearArgs, PosembArgs, ResampleArgs, SoftmaxArgs, StemArgs,
                              check, current_stream, lib)


# ---- host-side reference packers (fp64 transforms, torch ops): what the device packer anoddpm_pack_weights must reproduce
def _pack_conv(w):
    """OIHW / OI1 -> [taps][I/4][O][4] fp32 (B operand layout of the implicit GEMM)."""
    if w.dim() == 3:
        w = w.unsqueeze(-1)
    o, i, kh, kw = w.shape
    return (w.detach().float().permute(2, 3, 1, 0).reshape(kh * kw, i // 4, 4, o)
            .permute(0, 1, 3, 2).contiguous())


_WINO_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])


def _pack_wino(w):
    """OIHW 3x3 -> Winograd F(2x2,3x3) weights U = G g G^T, laid out [16][I/4][O][4] (xi = 4*u + v)."""
    o, i, kh, kw = w.shape
    assert kh == 3 and kw == 3
    G = _WINO_G.to(device=w.device, dtype=torch.float64)
    U = torch.einsum("ua,oiab,vb->uvio", G, w.detach().double(), G).float()          # [4][4][I][O]
    return U.reshape(16, i // 4, 4, o).permute(0, 1, 3, 2).contiguous()


_WINO43_G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                          [0, 0, 1]], dtype=torch.float64)


def _pack_wino43(w):
    """OIHW 3x3 -> Winograd F(4x4,3x3) weights U = G g G^T (G 6x3, fp64, rounded once), laid out [36][I/4][O][4] (xi = 6*u + v)."""
    o, i, kh, kw = w.shape
    assert kh == 3 and kw == 3
    G = _WINO43_G.to(device=w.device)
    U = torch.einsum("ua,oiab,vb->uvio", G, w.detach().double(), G).float()          # [6][6][I][O]
    return U.reshape(36, i // 4, 4, o).permute(0, 1, 3, 2).contiguous()



def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def gn_affine(srcs, gamma, beta, nslab=None, eps=1e-5):
    """srcs: list of NHWC [B,H,W,C] cuda tensors (1 or 2).  Returns scale, shift [B, Ctot]."""
    B, H, W, c0 = srcs[0].shape
    c1 = srcs[1].shape[3] if len(srcs) > 1 else 0
    C, P = c0 + c1, H * W
    nslab = nslab or max(1, min(256, (P * (C // 4)) // 4096))
    st = GnArgs()
    st.a0 = srcs[0].data_ptr()
    st.a1 = srcs[1].data_ptr() if c1 else None
    scale = torch.empty(B, C, device=srcs[0].device)
    shift = torch.empty(B, C, device=srcs[0].device)
    part = torch.empty(B * nslab * 64, dtype=torch.float64, device=srcs[0].device)
    st.gamma, st.beta, st.scale, st.shift, st.partial = gamma.data_ptr(), beta.data_ptr(), scale.data_ptr(), shift.data_ptr(), part.data_ptr()
    st.a0_bs, st.a1_bs, st.c0, st.c1, st.a0_ld, st.a1_ld = P * c0, P * c1, c0, c1, c0, max(c1, 4)
    st.P, st.B, st.groups, st.nslab, st.eps = P, B, 32, nslab, eps
    check(lib().anoddpm_gn_stats(ctypes.byref(st), current_stream()), "gn_stats")
    return scale, shift


def chan_stats(x, nslab):
    """x NHWC -> stats [B][nslab][C][2]."""
    B, H, W, C = x.shape
    stats = torch.full((B, nslab, C, 2), float("nan"), device=x.device)
    st = ChanStatsArgs()
    st.a, st.stats, st.a_bs, st.C, st.a_ld, st.P, st.B, st.nslab = x.data_ptr(), stats.data_ptr(), H * W * C, C, C, H * W, B, nslab
    check(lib().anoddpm_chan_stats(ctypes.byref(st), current_stream()), "chan_stats")
    return stats


def gn_finalize(stats, gamma, beta, P, want_mean_rstd=False):
    """stats: list of 1-2 tensors [B][rows][c][2] -> scale, shift [B][Ctot] (+ mean, rstd [B][32])."""
    B = stats[0].shape[0]
    c0 = stats[0].shape[2]
    c1 = stats[1].shape[2] if len(stats) > 1 else 0
    scale = torch.empty(B, c0 + c1, device=stats[0].device)
    shift = torch.empty(B, c0 + c1, device=stats[0].device)
    st = GnFinalizeArgs()
    st.stats0, st.rows0 = stats[0].data_ptr(), stats[0].shape[1]
    st.stats1, st.rows1 = (stats[1].data_ptr(), stats[1].shape[1]) if c1 else (None, 0)
    st.gamma, st.beta, st.scale, st.shift = gamma.data_ptr(), beta.data_ptr(), scale.data_ptr(), shift.data_ptr()
    st.c0, st.c1, st.P, st.B, st.groups, st.eps = c0, c1, P, B, 32, 1e-5
    mean = rstd = None
    if want_mean_rstd:
        mean = torch.full((B, 32), float("nan"), device=stats[0].device)
        rstd = torch.full((B, 32), float("nan"), device=stats[0].device)
        st.mean_out, st.rstd_out = mean.data_ptr(), rstd.data_ptr()
    check(lib().anoddpm_gn_finalize(ctypes.byref(st), current_stream()), "gn_finalize")
    return (scale, shift, mean, rstd) if want_mean_rstd else (scale, shift)


LAST_IGEMM = None
_LAST_KEEP = None


def conv_igemm(srcs, w, bias=None, *, Hout, ks, gn=None, act=0, a_mode=0, temb=None, res=None, cfg=0, ksplit=1,
               stats_out=None, gn_tail=None, fold=None, res_up=False, csum_out=None, gnb=None):
    """srcs: NHWC sources; w: OIHW weights.  Returns NHWC [B,Hout,Hout,N].  res_up: `res` is [B,Hout/2,Hout/2,N] and is repeated
    2x2 on the read (cfg 3, res_mode 1)."""
    dev = srcs[0].device
    B = srcs[0].shape[0]
    c0 = srcs[0].shape[3]
    c1 = srcs[1].shape[3] if len(srcs) > 1 else 0
    N = w.shape[0]
    P = Hout * Hout
    Pin = srcs[0].shape[1] * srcs[0].shape[2]
    st = IgemmArgs()
    if cfg == 7:                                          # F(4x4,3x3) weights as three bf16 planes (the device packer is the only one)
        wp = torch.empty(54 * N * (c0 + c1), device=dev)
        wc = w.detach().float().contiguous()
        check(lib().anoddpm_pack_wino43_bf16x3(wc.data_ptr(), wp.data_ptr(), N, c0 + c1, current_stream()), "pack_wino43_bf16x3")
    else:
        wp = _pack_wino43(w) if cfg == 3 else (_pack_wino(w) if cfg in (2, 6) else _pack_conv(w))
    out = torch.full((B, Hout, Hout, N), float("nan"), device=dev)
    st.a0, st.a1 = srcs[0].data_ptr(), srcs[1].data_ptr() if c1 else None
    st.a0_ld, st.a1_ld, st.c0, st.c1 = c0, max(c1, 4), c0, c1
    st.a0_bs, st.a1_bs = Pin * c0, Pin * c1
    st.gn_scale = gn[0].data_ptr() if gn else None
    st.gn_shift = gn[1].data_ptr() if gn else None
    st.gn_ld = c0 + c1
    st.bmat = wp.data_ptr()
    st.bias = bias.data_ptr() if bias is not None else None
    st.temb = temb.data_ptr() if temb is not None else None
    st.temb_ld = N
    st.res = res.data_ptr() if res is not None else None
    st.out, st.out_ld, st.res_ld = out.data_ptr(), N, N
    st.o_bs = st.r_bs = P * N
    if res_up:
        st.res_mode, st.r_bs = 1, (P // 4) * N
    st.H, st.W, st.ks, st.a_mode, st.act = Hout, Hout, ks, a_mode, act
    st.b_mode, st.ldb, st.N, st.B, st.heads, st.alpha = 0, 0, N, B, 1, 1.0
    st.cfg, st.ksplit = cfg, ksplit
    ws = torch.empty(max(1, ksplit * B * P * N), device=dev) if ksplit > 1 else None
    st.ws = ws.data_ptr() if ws is not None else None
    if fold is not None:
        # cfg 5: the operand's GroupNorm finished in the kernel's prologue; fold = dict(stats=[(tensor, fmt)] per source, gamma, beta)
        st.fold_gamma, st.fold_beta, st.fold_groups, st.fold_eps = fold["gamma"].data_ptr(), fold["beta"].data_ptr(), 32, 1e-5
        (s0, f0) = fold["stats"][0]
        st.fold_stats0, st.fold_rows0, st.fold_fmt0 = s0.data_ptr(), (1 if f0 else s0.shape[1]), f0
        if c1:
            (s1, f1) = fold["stats"][1]
            st.fold_stats1, st.fold_rows1, st.fold_fmt1 = s1.data_ptr(), (1 if f1 else s1.shape[1]), f1
    if csum_out is not None:
        # cfg 3: per-channel fp64 {sum, sum of squares} of the output accumulated atomically (anoddpm_igemm_args.stats_csum); the
        # caller passes a ZEROED [B, N, 2] float64 tensor (or a list to receive a fresh one)
        if isinstance(csum_out, list):
            csum_out.append(torch.zeros(B, N, 2, dtype=torch.float64, device=dev))
            csum_out = csum_out[-1]
        st.stats_csum = csum_out.data_ptr()
    if cfg in (5, 6) and stats_out is not None:
        tm = 64 if cfg == 6 else 16 * (lib().anoddpm_smallmap_tile(ks, Hout, Hout, c0 + c1, c0, N, B) >> 4)
        stats = torch.full((B, P // tm, N, 2), float("nan"), device=dev)
        st.stats = stats.data_ptr()
        stats_out.append(stats)
    elif gn_tail is not None:
        # split-K tail with the consumer GroupNorm folded in: gn_tail = dict(gamma, beta[, other_csum], want_mean) -> filled with
        # csum / scale / shift / mean / rstd
        assert ksplit > 1
        c1t = gn_tail["other_csum"].shape[1] if gn_tail.get("other_csum") is not None else 0
        csum = torch.full((B, N, 2), float("nan"), device=dev, dtype=torch.float64)
        st.tail_csum, st.tail_c1, st.tail_groups, st.tail_eps = csum.data_ptr(), c1t, 32, 1e-5
        gn_tail["csum"] = csum
        if gn_tail.get("gamma") is not None:
            sc = torch.full((B, N + c1t), float("nan"), device=dev)
            sh = torch.full((B, N + c1t), float("nan"), device=dev)
            st.tail_gamma, st.tail_beta = gn_tail["gamma"].data_ptr(), gn_tail["beta"].data_ptr()
            st.tail_scale, st.tail_shift = sc.data_ptr(), sh.data_ptr()
            st.tail_other = gn_tail["other_csum"].data_ptr() if c1t else None
            gn_tail["scale"], gn_tail["shift"] = sc, sh
            if gn_tail.get("want_mean"):
                gn_tail["mean"] = torch.full((B, 32), float("nan"), device=dev)
                gn_tail["rstd"] = torch.full((B, 32), float("nan"), device=dev)
                st.tail_mean, st.tail_rstd = gn_tail["mean"].data_ptr(), gn_tail["rstd"].data_ptr()
    elif stats_out is not None and ksplit > 1:
        nslab = 3 if P > 16 else 1
        stats = torch.full((B, nslab, N, 2), float("nan"), device=dev)
        st.stats, st.stats_rows = stats.data_ptr(), nslab
        stats_out.append(stats)
    elif stats_out is not None:
        bm = 128 if cfg == 0 else 64
        if cfg in (2, 3, 7):
            tiles = (Hout // 16) ** 2
        else:
            tiles = -(-P // bm) if ks == 1 else (Hout // min(Hout, 32)) * -(-Hout // (bm // min(Hout, 32)))
        stats = torch.full((B, tiles * {2: 4, 3: 1, 7: 1}.get(cfg, 2), N, 2), float("nan"), device=dev)
        st.stats = stats.data_ptr()
        stats_out.append(stats)
    if gnb is not None:
        # data-gradient launch that also writes the partial sums of the GroupNorm + SiLU backward's reduction pass
        # (anoddpm_igemm_args.gnb_*): gnb = dict(srcs=[NHWC x sources], gamma, beta, mean, rstd) -> gnb["partial"] = [B, tiles, N, 2] fp64
        xs = gnb["srcs"]
        xc0 = xs[0].shape[3]
        xc1 = xs[1].shape[3] if len(xs) > 1 else 0
        part = torch.full((B, (Hout // 16) ** 2, N, 2), float("nan"), dtype=torch.float64, device=dev)
        st.gnb_partial = part.data_ptr()
        st.gnb_x0, st.gnb_x1 = xs[0].data_ptr(), (xs[1].data_ptr() if xc1 else None)
        st.gnb_gamma, st.gnb_beta = gnb["gamma"].data_ptr(), gnb["beta"].data_ptr()
        st.gnb_mean, st.gnb_rstd = gnb["mean"].data_ptr(), gnb["rstd"].data_ptr()
        st.gnb_x0_bs, st.gnb_x1_bs = P * xc0, P * xc1
        st.gnb_c0, st.gnb_x0_ld, st.gnb_x1_ld, st.gnb_groups = xc0, xc0, max(xc1, 4), 32
        gnb["partial"] = part
    check(lib().anoddpm_igemm(ctypes.byref(st), current_stream()), "igemm")
    torch.cuda.synchronize()
    global LAST_IGEMM, _LAST_KEEP
    LAST_IGEMM, _LAST_KEEP = st, (srcs, wp, out, gn, bias, temb, res, ws, gn_tail, fold)        # tools/bench_conv.py re-launches the prepared call
    return out


def attention(qkv, heads, cfg=1):
    """qkv: [B, L, 3C] (legacy per-head q|k|v channel order).  Returns [B, L, C]."""
    dev = qkv.device
    B, L, C3 = qkv.shape
    C = C3 // 3
    ch = C // heads
    S = torch.full((B * heads, L, L), float("nan"), device=dev)
    st = IgemmArgs()
    st.a0, st.a0_ld, st.c0, st.c1 = qkv.data_ptr(), C3, ch, 0
    st.a0_bs, st.a0_hs = L * C3, 3 * ch
    st.bmat, st.b_mode, st.ldb, st.b_bs, st.b_hs = qkv.data_ptr() + 4 * ch, 1, C3, L * C3, 3 * ch
    st.out, st.out_ld, st.o_bs, st.o_hs = S.data_ptr(), L, heads * L * L, L * L
    st.H, st.W, st.ks, st.N, st.B, st.heads, st.alpha = 1, L, 1, L, B, heads, 1.0 / math.sqrt(ch)
    st.cfg, st.ksplit, st.a1_ld, st.res_ld, st.temb_ld, st.gn_ld = cfg, 1, 4, L, 0, ch
    check(lib().anoddpm_igemm(ctypes.byref(st), current_stream()), "igemm qk")
    sm = SoftmaxArgs()
    sm.x, sm.rows, sm.L = S.data_ptr(), B * heads * L, L
    check(lib().anoddpm_softmax_rows(ctypes.byref(sm), current_stream()), "softmax")
    out = torch.full((B, L, C), float("nan"), device=dev)
    st2 = IgemmArgs()
    st2.a0, st2.a0_ld, st2.c0, st2.c1 = S.data_ptr(), L, L, 0
    st2.a0_bs, st2.a0_hs = heads * L * L, L * L
    st2.bmat, st2.b_mode, st2.ldb, st2.b_bs, st2.b_hs = qkv.data_ptr() + 8 * ch, 2, C3, L * C3, 3 * ch
    st2.out, st2.out_ld, st2.o_bs, st2.o_hs = out.data_ptr(), C, L * C, ch
    st2.H, st2.W, st2.ks, st2.N, st2.B, st2.heads, st2.alpha = 1, L, 1, ch, B, heads, 1.0
    st2.cfg, st2.ksplit, st2.a1_ld, st2.res_ld, st2.temb_ld, st2.gn_ld = cfg, 1, 4, C, 0, L
    check(lib().anoddpm_igemm(ctypes.byref(st2), current_stream()), "igemm pv")
    torch.cuda.synchronize()
    return out, S


def attention_fused(qkv, heads, want_probs=False):
    """qkv: [B, L, 3C] (legacy per-head q|k|v channel order) -> ([B, L, C], probs or None) through anoddpm_attention."""
    from anoddpm_amd._lib import AttentionArgs
    B, L, C3 = qkv.shape
    C = C3 // 3
    ch = C // heads
    out = torch.full((B, L, C), float("nan"), device=qkv.device)
    probs = torch.full((B * heads, L, L), float("nan"), device=qkv.device) if want_probs else None
    st = AttentionArgs()
    st.qkv, st.out, st.probs = qkv.data_ptr(), out.data_ptr(), probs.data_ptr() if want_probs else None
    st.B, st.L, st.heads, st.ch, st.scale = B, L, heads, ch, 1.0 / math.sqrt(ch)
    check(lib().anoddpm_attention(ctypes.byref(st), current_stream()), "attention")
    torch.cuda.synchronize()
    return out, probs


def resample(x, mode, gn=None):
    """x NHWC.  mode 1 nearest x2, 2 average 2x2; with gn = (scale, shift) [B][C] mode 2 also returns the pooled ACTIVATED tensor."""
    B, H, W, C = x.shape
    Ho = H * 2 if mode == 1 else H // 2
    out = torch.full((B, Ho, Ho, C), float("nan"), device=x.device)
    st = ResampleArgs()
    st.inp, st.out, st.B, st.H, st.W, st.C, st.mode = x.data_ptr(), out.data_ptr(), B, H, W, C, mode
    act = None
    if gn is not None:
        act = torch.full((B, Ho, Ho, C), float("nan"), device=x.device)
        st.gn_scale, st.gn_shift, st.out_act = gn[0].data_ptr(), gn[1].data_ptr(), act.data_ptr()
    check(lib().anoddpm_resample2x(ctypes.byref(st), current_stream()), "resample")
    torch.cuda.synchronize()
    return out if gn is None else (out, act)


def linear(x, w, b, act_in=0, act_out=0):
    B, K = x.shape
    N = w.shape[0]
    out = torch.full((B, N), float("nan"), device=x.device)
    st = LinearArgs()
    st.inp, st.w, st.bias, st.out = x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, out.data_ptr()
    st.B, st.K, st.N, st.act_in, st.act_out = B, K, N, act_in, act_out
    check(lib().anoddpm_linear_small(ctypes.byref(st), current_stream()), "linear")
    torch.cuda.synchronize()
    return out


def posemb(t, freqs, dim):
    B = t.numel()
    out = torch.full((B, dim), float("nan"), device=t.device)
    st = PosembArgs()
    st.t, st.freqs, st.out, st.B, st.dim, st.scale = t.data_ptr(), freqs.data_ptr(), out.data_ptr(), B, dim, 1.0
    check(lib().anoddpm_posemb(ctypes.byref(st), current_stream()), "posemb")
    torch.cuda.synchronize()
    return out


def stem(x, w, b, with_stats=False):
    """-> NHWC output [, fused GroupNorm partial sums [B][rows][Cout][2] (None when the shape has no fused form)]"""
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    wp = w.permute(2, 3, 1, 0).reshape(-1, Cout).contiguous()
    out = torch.full((B, H, W, Cout), float("nan"), device=x.device)
    st = StemArgs()
    st.x, st.w, st.bias, st.out = x.data_ptr(), wp.data_ptr(), b.data_ptr(), out.data_ptr()
    st.B, st.H, st.W, st.Cin, st.Cout = B, H, W, Cin, Cout
    stats = None
    if with_stats:
        rows = lib().anoddpm_stem_stats_rows(H, W, Cin, Cout)
        if rows > 0:
            stats = torch.full((B, rows, Cout, 2), float("nan"), device=x.device)
            st.stats, st.stats_rows = stats.data_ptr(), rows
    check(lib().anoddpm_conv_stem(ctypes.byref(st), current_stream()), "stem")
    torch.cuda.synchronize()
    return (out, stats) if with_stats else out


def head(x, w, b, scale, shift):
    """x NHWC, w OIHW (O <= 4) -> NCHW output."""
    B, H, W, C = x.shape
    Cout = w.shape[0]
    wp = w.permute(2, 3, 1, 0).reshape(-1, Cout).contiguous()
    out = torch.full((B, Cout, H, W), float("nan"), device=x.device)
    st = HeadArgs()
    st.x, st.w, st.bias, st.gn_scale, st.gn_shift, st.out = x.data_ptr(), wp.data_ptr(), b.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr()
    st.B, st.H, st.W, st.C, st.Cout = B, H, W, C, Cout
    check(lib().anoddpm_conv_head(ctypes.byref(st), current_stream()), "conv_head")
    torch.cuda.synchronize()
    return out


def conv_wgrad(srcs, dy, *, gn=None, act=0, a_mode=0, band=None, accumulate_into=None, colsum_out=None, algo=0, bias_out=None):
    """dW (OIHW) of a 3x3 conv whose input is the fused operand load of `conv_igemm` (anoddpm_conv3x3_wgrad).
    srcs: NHWC sources; dy: NHWC [B,H,W,N]."""
    from anoddpm_amd._lib import WgradArgs
    dev = dy.device
    B, H, W, N = dy.shape
    c0 = srcs[0].shape[3]
    c1 = srcs[1].shape[3] if len(srcs) > 1 else 0
    K = c0 + c1
    Pin = srcs[0].shape[1] * srcs[0].shape[2]
    TW = next(t for t in (32, 16, 8, 4, 2) if W % t == 0)
    band = band or max(1, H // 4)
    nitems = B * (W // TW) * (-(-H // band))
    if algo == 1:                       # Winograd-domain weight gradient: PG slabs of [36][K][N]; column sums per 16x8 output patch
        ws = torch.empty(lib().anoddpm_wgrad43_groups(K, N, B, H, W) * 9 * K * N, device=dev)
        nitems = B * lib().anoddpm_wgrad43_colsum_items(K, N, B, H, W)
    else:
        ws = torch.empty(nitems * 9 * K * N, device=dev)
    dw = accumulate_into if accumulate_into is not None else torch.full((N, K, 3, 3), float("nan"), device=dev)
    st = WgradArgs()
    st.a0, st.a1 = srcs[0].data_ptr(), srcs[1].data_ptr() if c1 else None
    st.gn_scale = gn[0].data_ptr() if gn else None
    st.gn_shift = gn[1].data_ptr() if gn else None
    st.dy, st.dw, st.ws, st.ws_floats = dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel()
    st.a0_bs, st.a1_bs, st.dy_bs = Pin * c0, Pin * c1, H * W * N
    st.c0, st.c1, st.a0_ld, st.a1_ld, st.dy_ld = c0, c1, c0, max(c1, 4), N
    st.H, st.W, st.N, st.B = H, W, N, B
    st.a_mode, st.act, st.gn_ld, st.band = a_mode, act, K, band
    st.accumulate = 1 if accumulate_into is not None else 0
    st.algo = algo
    if colsum_out is not None:
        cs = torch.full((B, nitems // B, N), float("nan"), device=dev)
        st.colsum = cs.data_ptr()
        colsum_out.append(cs)
    if bias_out is not None:
        # algo 1: the fold launch also folds the column sums -> bias_out = {"dimg": [B, N] per-image sums of dy, "dbias": [N] += their sum}
        assert algo == 1 and colsum_out is not None
        bias_out["dimg"] = torch.full((B, N), float("nan"), device=dev)
        bias_out.setdefault("dbias", torch.zeros(N, device=dev))
        st.dimg, st.dbias = bias_out["dimg"].data_ptr(), bias_out["dbias"].data_ptr()
    check(lib().anoddpm_conv3x3_wgrad(ctypes.byref(st), current_stream()), "conv3x3_wgrad")
    torch.cuda.synchronize()
    return dw


def gn_silu_backward(srcs, da, gamma, beta, mean, rstd, *, act=1, a_mode=0, acc_into=None, nslab=None, partial=None):
    """anoddpm_gn_silu_backward.  srcs: 1-2 NHWC sources; da: NHWC gradient w.r.t. the tensor the conv read.
    partial: [B, rows, C, 2] fp64 written by the data-gradient launch (conv_igemm(gnb=...)): no reduction launch.
    Returns (dx list, dgamma, dbeta)."""
    from anoddpm_amd._lib import GnBwdArgs
    dev = srcs[0].device
    da = da.clone()                 # the kernel may use da as scratch (act != 0, a_mode == 0): keep the caller's tensor intact
    B, Hs, Ws, c0 = srcs[0].shape
    c1 = srcs[1].shape[3] if len(srcs) > 1 else 0
    C, P = c0 + c1, Hs * Ws
    nslab = partial.shape[1] if partial is not None else (nslab or max(1, min(64, P // 64)))
    dx = acc_into if acc_into is not None else [torch.full_like(s_, float("nan")) for s_ in srcs]
    dgamma, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    part = partial if partial is not None else torch.empty(B * nslab * C * 2, dtype=torch.float64, device=dev)
    coef = torch.empty(B * C * 4, device=dev)
    st = GnBwdArgs()
    st.x0, st.x1 = srcs[0].data_ptr(), srcs[1].data_ptr() if c1 else None
    st.da, st.gamma, st.beta, st.mean, st.rstd = da.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    st.dx0, st.dx1 = dx[0].data_ptr(), dx[1].data_ptr() if c1 else None
    st.dgamma, st.dbeta, st.partial, st.coef = dgamma.data_ptr(), dbeta.data_ptr(), part.data_ptr(), coef.data_ptr()
    Pa = da.shape[1] * da.shape[2]
    st.x0_bs, st.x1_bs, st.da_bs, st.dx0_bs, st.dx1_bs = P * c0, P * c1, Pa * C, P * c0, P * c1
    st.c0, st.c1, st.x0_ld, st.x1_ld, st.da_ld, st.dx0_ld, st.dx1_ld = c0, c1, c0, max(c1, 4), C, c0, max(c1, 4)
    st.Hs, st.Ws, st.B, st.groups, st.nslab = Hs, Ws, B, 32, nslab
    st.act, st.a_mode, st.acc_dx = act, a_mode, 3 if acc_into is not None else 0
    st.partial_ready = 1 if partial is not None else 0
    check(lib().anoddpm_gn_silu_backward(ctypes.byref(st), current_stream()), "gn_silu_backward")
    torch.cuda.synchronize()
    return dx, dgamma, dbeta
