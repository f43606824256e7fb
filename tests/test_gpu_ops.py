"""-m gpu: every C-ABI UNet op against the stock-PyTorch CPU statement of the same op (the oracle's
building blocks, oracle/unet_oracle.py).  Tolerance: 1e-3 relative to the tensor's max magnitude is the
north-star bar; these ops are plain fp32 so the tests ask for 2e-5."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


def dev():
    return torch.device("cuda:0")


def relerr(got, ref):
    ref = ref.float().cpu()
    got = got.float().cpu()
    assert torch.isfinite(got).all(), "non-finite output (unwritten region?)"
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 0), (1, 8, 8, 1024, 0), (2, 32, 32, 96, 0), (1, 64, 64, 128, 0),
                                   (2, 16, 16, 256, 128), (1, 8, 8, 64, 32), (3, 4, 4, 96, 0), (1, 16, 16, 1536, 0)])
def test_gn_affine(shape):
    import hipops
    B, H, W, c0, c1 = shape
    C = c0 + c1
    x = rnd(B, C, H, W, seed=1) * 2 + 0.7
    gamma, beta = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    ref = F.group_norm(x, 32, gamma, beta, eps=1e-5)
    xs = hipops.nhwc(x.to(dev()))
    srcs = [xs[..., :c0].contiguous()] + ([xs[..., c0:].contiguous()] if c1 else [])
    for nslab in (None, 1, 3):
        sc, sh = hipops.gn_affine(srcs, gamma.to(dev()), beta.to(dev()), nslab=nslab)
        got = xs * sc[:, None, None, :] + sh[:, None, None, :]
        assert relerr(hipops.nchw(got), ref) < TOL


@pytest.mark.parametrize("case", [(2, 128, 32, 3, 0), (1, 64, 64, 3, 0), (2, 96, 16, 3, 1), (3, 64, 4, 3, 1), (1, 256, 16, 1, 1), (1, 128, 128, 1, 0)])
def test_fused_groupnorm_statistics(case):
    """GroupNorm affine from the igemm epilogue's per-channel sums == group_norm of the conv output; and the
    two-source fold (virtual concat) with one source coming from the stand-alone chan_stats kernel."""
    import hipops
    B, C, H, ks, cfg = case
    x = rnd(B, 32, H, H, seed=61)
    w, b = rnd(C, 32, ks, ks, seed=62, scale=0.2), rnd(C, seed=63)
    y = F.conv2d(x, w, b, padding=ks // 2)
    gamma, beta = 1 + 0.1 * rnd(C, seed=64), 0.1 * rnd(C, seed=65)
    st = []
    got = hipops.conv_igemm([hipops.nhwc(x.to(dev()))], w.to(dev()), b.to(dev()), Hout=H, ks=ks, cfg=cfg, stats_out=st)
    assert relerr(hipops.nchw(got), y) < TOL
    # same layer through split-K: the reduction kernel emits the statistics
    st_sk = []
    got_sk = hipops.conv_igemm([hipops.nhwc(x.to(dev()))], w.to(dev()), b.to(dev()), Hout=H, ks=ks, cfg=cfg, ksplit=1 if ks == 1 else 1, stats_out=[])
    if C % 4 == 0:
        x64 = rnd(B, 64, H, H, seed=69)
        w64 = rnd(C, 64, ks, ks, seed=70, scale=0.1)
        y64 = F.conv2d(x64, w64, b, padding=ks // 2)
        got_sk = hipops.conv_igemm([hipops.nhwc(x64.to(dev()))], w64.to(dev()), b.to(dev()), Hout=H, ks=ks, cfg=cfg, ksplit=2, stats_out=st_sk)
        assert relerr(hipops.nchw(got_sk), y64) < TOL
        sc2, sh2 = hipops.gn_finalize(st_sk, gamma.to(dev()), beta.to(dev()), H * H)
        assert relerr(hipops.nchw(got_sk * sc2[:, None, None, :] + sh2[:, None, None, :]), F.group_norm(y64, 32, gamma, beta, eps=1e-5)) < TOL
    sc, sh = hipops.gn_finalize(st, gamma.to(dev()), beta.to(dev()), H * H)
    assert relerr(hipops.nchw(got * sc[:, None, None, :] + sh[:, None, None, :]), F.group_norm(y, 32, gamma, beta, eps=1e-5)) < TOL
    # concat [y, skip]: skip statistics from the stand-alone kernel
    skip = rnd(B, 32, H, H, seed=66) * 3 - 1
    st2 = hipops.chan_stats(hipops.nhwc(skip.to(dev())), nslab=3 if H > 4 else 1)
    g2, b2 = 1 + 0.1 * rnd(C + 32, seed=67), 0.1 * rnd(C + 32, seed=68)
    sc, sh = hipops.gn_finalize([st[0], st2], g2.to(dev()), b2.to(dev()), H * H)
    cat = torch.cat([y, skip], 1)
    gotc = hipops.nhwc(cat.to(dev())) * sc[:, None, None, :] + sh[:, None, None, :]
    assert relerr(hipops.nchw(gotc), F.group_norm(cat, 32, g2, b2, eps=1e-5)) < TOL


CONV_CASES = [
    # B, Cin(c0,c1), Cout, Hout, ks, a_mode, gn, act, temb, res, cfg, ksplit
    (1, (64, 0), 128, 32, 3, 0, True, 1, True, True, 0, 1),
    (2, (32, 0), 64, 16, 3, 0, True, 1, False, False, 1, 1),
    (1, (128, 0), 128, 64, 3, 0, False, 0, False, False, 0, 1),
    (2, (64, 64), 128, 16, 3, 0, True, 1, True, True, 0, 1),       # virtual concat
    (2, (96, 32), 96, 8, 3, 0, True, 1, False, True, 1, 4),        # concat + split-K + N tail
    (1, (64, 0), 64, 32, 3, 1, True, 1, False, False, 0, 1),       # fused nearest x2
    (1, (64, 0), 64, 16, 3, 2, True, 1, True, False, 1, 2),        # fused 2x2 avg pool
    (2, (256, 0), 256, 8, 3, 0, True, 1, True, True, 1, 8),        # 8x8 level, split-K
    (3, (64, 0), 32, 4, 3, 0, True, 1, False, True, 1, 2),         # 4x4 level (partial tile)
    (1, (128, 0), 1, 32, 3, 0, True, 1, False, False, 1, 1),       # head conv, N = 1
    (1, (128, 0), 3, 16, 3, 0, True, 1, False, False, 1, 1),       # head conv, N = 3
    (2, (64, 32), 128, 16, 1, 0, False, 0, False, False, 0, 1),    # 1x1 skip over concat
    (1, (128, 0), 384, 16, 1, 0, True, 0, False, False, 1, 2),     # qkv projection (GN, no SiLU)
    (2, (64, 0), 64, 8, 1, 0, False, 0, False, True, 1, 1),        # proj_out + residual
    (1, (32, 0), 32, 128, 3, 0, True, 1, True, True, 1, 1),        # wide image, narrow channels
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_igemm_conv(case):
    import hipops
    B, (c0, c1), N, Hout, ks, a_mode, use_gn, act, use_temb, use_res, cfg, ksplit = case
    C = c0 + c1
    Hin = Hout if a_mode == 0 else (Hout // 2 if a_mode == 1 else Hout * 2)
    x = rnd(B, C, Hin, Hin, seed=11)
    w = rnd(N, C, ks, ks, seed=12, scale=1.0 / math.sqrt(C * ks * ks))
    b = rnd(N, seed=13, scale=0.1)
    gamma, beta = 1 + 0.1 * rnd(C, seed=14), 0.1 * rnd(C, seed=15)
    temb = rnd(B, N, seed=16) if use_temb else None
    res = rnd(B, N, Hout, Hout, seed=17) if use_res else None
    h = x
    if use_gn:
        h = F.group_norm(h, 32, gamma, beta, eps=1e-5)
    if act:
        h = F.silu(h)
    if a_mode == 1:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
    elif a_mode == 2:
        h = F.avg_pool2d(h, 2, 2)
    ref = F.conv2d(h, w, b, padding=ks // 2)
    if temb is not None:
        ref = ref + temb[:, :, None, None]
    if res is not None:
        ref = ref + res
    xs = hipops.nhwc(x.to(dev()))
    srcs = [xs[..., :c0].contiguous()] + ([xs[..., c0:].contiguous()] if c1 else [])
    gn = hipops.gn_affine(srcs, gamma.to(dev()), beta.to(dev())) if use_gn else None
    got = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), Hout=Hout, ks=ks, gn=gn, act=act, a_mode=a_mode,
                            temb=temb.to(dev()) if temb is not None else None,
                            res=hipops.nhwc(res.to(dev())) if res is not None else None, cfg=cfg, ksplit=ksplit)
    assert relerr(hipops.nchw(got), ref) < TOL


@pytest.mark.parametrize("case", [(2, 64, 64, 2), (1, 256, 512, 2), (2, 16, 96, 2), (1, 1024, 64, 1), (1, 64, 128, 4), (1, 256, 512, 1)])
def test_attention_legacy_order(case):
    import hipops
    B, L, C, heads = case
    qkv = rnd(B, 3 * C, L, seed=21)
    ch = C // heads
    q, k, v = qkv.reshape(B * heads, 3 * ch, L).split(ch, dim=1)          # UNet.py:146
    s = 1 / math.sqrt(math.sqrt(ch))
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s).float(), dim=-1)
    ref = torch.einsum("bts,bcs->bct", wgt, v).reshape(B, C, L)
    for cfg in ((1, 0) if L >= 128 else (1,)):
        got, S = hipops.attention(qkv.permute(0, 2, 1).contiguous().to(dev()), heads, cfg=cfg)
        assert relerr(S, wgt) < TOL
        assert relerr(got.permute(0, 2, 1), ref) < TOL


FUSED_ATTN_CASES = [
    # B, L, C, heads
    (2, 64, 64, 2),       # ch 32: two channel tiles, waves 2 and 3 idle in the P v phase
    (1, 256, 512, 2),     # config 2 at 16^2
    (4, 64, 512, 2),      # config 2 at 8^2
    (1, 1024, 256, 2),    # config 5 at 32^2: the largest resident score block
    (2, 16, 32, 2),       # one key tile, ch 16
    (1, 64, 512, 1),      # ch 512
    (3, 48, 64, 1),       # L not a multiple of 64: ragged split of the key tiles over the waves
]


@pytest.mark.parametrize("case", FUSED_ATTN_CASES)
def test_fused_attention(case):
    """anoddpm_attention (csrc/attention.hip) against QKVAttentionLegacy's expression (UNet.py:137-153) in fp64, and against
    the three-launch form; the optional probability output is the softmax itself."""
    import hipops
    B, L, C, heads = case
    qkv = rnd(B, 3 * C, L, seed=121) * 1.5
    ch = C // heads
    q, k, v = qkv.double().reshape(B * heads, 3 * ch, L).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1)
    ref = torch.einsum("bts,bcs->bct", wgt, v).reshape(B, C, L)
    d = qkv.permute(0, 2, 1).contiguous().to(dev())
    got, P = hipops.attention_fused(d, heads, want_probs=True)
    assert relerr(P, wgt.float()) < TOL
    assert relerr(got.permute(0, 2, 1), ref.float()) < TOL
    assert torch.allclose(P.sum(-1).cpu(), torch.ones(B * heads, L), atol=1e-5)
    got2, none = hipops.attention_fused(d, heads)
    assert none is None and torch.equal(got, got2)                       # the optional output does not change the result
    old, _ = hipops.attention(d, heads, cfg=1)
    assert relerr(got, old) < TOL


def test_fused_attention_rejects_unsupported_shapes():
    import hipops
    with pytest.raises(Exception, match="attention"):
        hipops.attention_fused(rnd(1, 40, 3 * 32, seed=1).to(dev()), 1)       # L % 16 != 0
    with pytest.raises(Exception, match="attention"):
        hipops.attention_fused(rnd(1, 64, 3 * 48, seed=1).to(dev()), 1)       # head width 48


def test_softmax_spike_rows():
    import hipops, ctypes
    from anoddpm_amd._lib import SoftmaxArgs, check, lib, current_stream
    x = rnd(37, 200, seed=5) * 30
    x[3, 7] = 500.0
    d = x.to(dev()).clone()
    sm = SoftmaxArgs()
    sm.x, sm.rows, sm.L = d.data_ptr(), 37, 200
    check(lib().anoddpm_softmax_rows(ctypes.byref(sm), current_stream()))
    torch.cuda.synchronize()
    assert relerr(d, torch.softmax(x, -1)) < TOL


@pytest.mark.parametrize("shape", [(2, 64, 16), (1, 128, 64), (3, 96, 8), (2, 512, 4), (1, 32, 34)])
@pytest.mark.parametrize("mode", [1, 2])
def test_resample(mode, shape):
    """nearest x2 / 2x2 average, bit-identical to ATen."""
    import hipops
    B, C, H = shape
    x = rnd(B, C, H, H, seed=31)
    ref = F.interpolate(x, scale_factor=2, mode="nearest") if mode == 1 else F.avg_pool2d(x, 2, 2)
    got = hipops.resample(hipops.nhwc(x.to(dev())), mode)
    assert torch.equal(hipops.nchw(got).cpu(), ref)


@pytest.mark.parametrize("shape", [(2, 64, 16), (1, 128, 64), (3, 96, 8)])
def test_resample_pooled_activated_operand(shape):
    """Mode 2 with a GroupNorm affine also emits mean_2x2(silu(x * scale + shift)) -- the operand of a down block's first
    convolution (UNet.py:175-178) -- in the same pass; row kernel == generic kernel bit for bit."""
    import hipops
    from anoddpm_amd._lib import lib
    B, C, H = shape
    x = rnd(B, C, H, H, seed=32)
    sc, sh = 1 + 0.2 * rnd(B, C, seed=33), 0.3 * rnd(B, C, seed=34)
    xs = hipops.nhwc(x.to(dev()))
    out, act = hipops.resample(xs, 2, gn=(sc.to(dev()), sh.to(dev())))
    assert torch.equal(hipops.nchw(out).cpu(), F.avg_pool2d(x, 2, 2))
    ref = F.avg_pool2d(F.silu(x * sc[:, :, None, None] + sh[:, :, None, None]), 2, 2)
    assert relerr(hipops.nchw(act), ref) < TOL


@pytest.mark.parametrize("case", [(1, 128, 512, 0, 1), (4, 512, 512, 0, 0), (4, 512, 1000, 1, 0), (13, 64, 36, 1, 1)])
def test_linear_small(case):
    import hipops
    B, K, N, ai, ao = case
    x, w, b = rnd(B, K, seed=41), rnd(N, K, seed=42, scale=1 / math.sqrt(K)), rnd(N, seed=43)
    ref = F.linear(F.silu(x) if ai else x, w, b)
    ref = F.silu(ref) if ao else ref
    assert relerr(hipops.linear(x.to(dev()), w.to(dev()), b.to(dev()), ai, ao), ref) < TOL


def test_posemb_matches_reference_expression():
    import hipops
    from anoddpm_amd.unet import _posemb_freqs
    t = torch.tensor([0, 1, 17, 500, 999])
    for dim in (32, 128):
        fr = _posemb_freqs(dim // 2)
        arg = torch.outer(t * 1, fr)
        ref = torch.cat((arg.sin(), arg.cos()), -1)
        got = hipops.posemb(t.to(dev()), fr.to(dev()), dim).cpu()
        assert (got - ref).abs().max() < 2e-6


@pytest.mark.parametrize("case", [(2, 1, 32, 64), (1, 3, 64, 128), (1, 1, 16, 32)])
def test_stem(case):
    import hipops
    B, Cin, H, Cout = case
    x, w, b = rnd(B, Cin, H, H, seed=51), rnd(Cout, Cin, 3, 3, seed=52, scale=0.3), rnd(Cout, seed=53)
    ref = F.conv2d(x, w, b, padding=1)
    got = hipops.stem(x.to(dev()), w.to(dev()), b.to(dev()))
    assert relerr(hipops.nchw(got), ref) < TOL


@pytest.mark.parametrize("case", [(2, 1, 32, 64), (4, 1, 256, 128), (1, 2, 64, 32), (1, 1, 512, 128), (1, 1, 16, 1024), (1, 3, 64, 128)])
def test_stem_fused_groupnorm_sums(case):
    """The stem kernel's own GroupNorm partial sums (one row per workgroup range, the contraction epilogues' row format):
    rows sum to the per-channel sum / sum of squares of the output it wrote; shapes without a fused form report 0 rows."""
    import hipops
    from anoddpm_amd._lib import lib
    B, Cin, H, Cout = case
    x, w, b = rnd(B, Cin, H, H, seed=54), rnd(Cout, Cin, 3, 3, seed=55, scale=0.3), rnd(Cout, seed=56)
    got, stats = hipops.stem(x.to(dev()), w.to(dev()), b.to(dev()), with_stats=True)
    ref = F.conv2d(x, w, b, padding=1)
    assert relerr(hipops.nchw(got), ref) < TOL
    rows = lib().anoddpm_stem_stats_rows(H, H, Cin, Cout)
    if Cin > 2:
        assert rows == 0 and stats is None
        return
    assert 0 < rows <= 1024 and stats.shape == (B, rows, Cout, 2) and torch.isfinite(stats).all()
    o = got.double().reshape(B, rows, -1, Cout)                     # a row = one contiguous pixel range of one image
    assert (stats[..., 0].double() - o.sum(2)).abs().max() <= 1e-5 * o.abs().sum(2).max()
    assert (stats[..., 1].double() - (o * o).sum(2)).abs().max() <= 1e-5 * (o * o).sum(2).max()


@pytest.mark.parametrize("case", [(2, 128, 32, 1), (1, 64, 64, 3), (1, 32, 8, 2), (1, 128, 256, 1),
                                  (1, 256, 16, 1), (2, 128, 24, 4), (1, 64, 40, 2), (1, 96, 16, 1),    # matrix-pipe widths, 96: tap kernel
                                  (1, 64, 40, 1), (3, 128, 24, 1), (1, 128, 72, 1)])                    # 16 x 28 tiles: ragged edges
def test_head_conv(case):
    import hipops
    B, C, H, Cout = case
    x = rnd(B, C, H, H, seed=71)
    w, b = rnd(Cout, C, 3, 3, seed=72, scale=0.05), rnd(Cout, seed=73)
    gamma, beta = 1 + 0.1 * rnd(C, seed=74), 0.1 * rnd(C, seed=75)
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), w, b, padding=1)
    xs = hipops.nhwc(x.to(dev()))
    sc, sh = hipops.gn_affine([xs], gamma.to(dev()), beta.to(dev()))
    got = hipops.head(xs, w.to(dev()), b.to(dev()), sc, sh)
    assert relerr(got, ref) < TOL


WINO_CASES = [
    # B, (c0, c1), Cout, Hout, a_mode, gn, act, temb, res
    (1, (32, 0), 64, 16, 0, False, 0, False, False),       # bare transform check, one workgroup
    (2, (64, 0), 128, 32, 0, True, 1, True, True),         # full ResBlock conv: GN + SiLU + temb + residual
    (1, (128, 0), 128, 64, 0, True, 1, False, False),
    (2, (64, 64), 128, 16, 0, True, 1, True, True),        # virtual concat
    (1, (96, 32), 96, 32, 0, True, 1, False, True),        # N tail (96 = 64 + 32)
    (1, (64, 0), 64, 32, 1, True, 1, False, False),        # fused nearest x2
    (1, (16, 0), 32, 48, 0, False, 0, False, False),       # non power-of-two image, single K iteration
    # split-K over 16-channel chunks (small maps): last element = ksplit
    (2, (256, 0), 128, 32, 0, True, 1, True, True, 4),
    (1, (128, 64), 64, 16, 0, True, 1, False, True, 3),    # virtual concat, split inside and across the sources
    (1, (160, 0), 96, 16, 0, True, 1, True, False, 4),     # 10 chunks over 4 blocks: ragged last block, N tail
    (2, (64, 0), 64, 32, 1, True, 1, False, False, 2),     # fused nearest x2
]


F43_CASES = [
    # B, (c0, c1), Cout, Hout, a_mode, gn, act, temb, res
    (1, (16, 0), 128, 16, 0, False, 0, False, False),      # one workgroup, one K iteration: bare transform check
    (2, (64, 0), 128, 32, 0, True, 1, True, True),         # full ResBlock conv: GN + SiLU + temb + residual
    (1, (128, 0), 256, 64, 0, True, 1, False, False),      # two channel blocks
    (2, (64, 64), 128, 32, 0, True, 1, True, True),        # virtual concat
    (1, (64, 0), 128, 32, 1, True, 1, False, False),       # fused nearest x2
    (1, (96, 0), 128, 48, 0, True, 1, False, True),        # non power-of-two image, six K iterations
    (1, (128, 0), 128, 32, 0, False, 0, False, False),     # data-gradient form: no GroupNorm / activation
    (4, (64, 0), 256, 64, 0, True, 1, True, True),         # 64-channel workgroups (128 of the 128-channel ones would not fill the chip)
    (2, (32, 32), 64, 32, 0, True, 1, False, True),        # N = 64: only the 64-channel variant applies; virtual concat
    (1, (128, 0), 128, 32, 0, True, 1, True, True, 2),     # split-K 2 on the channel-sliced kernel + the split-K tail
    (2, (96, 64), 256, 32, 1, True, 1, False, True, 3),    # split-K 3: ragged slices (10 chunks), concat, fused nearest x2
]


STREAM1X1_CASES = [
    # B, (c0, c1), Cout, H, temb, res
    (1, (128, 0), 128, 32, False, False),      # one chunk group, 128-channel workgroups
    (2, (128, 128), 128, 64, False, False),    # the 256^2 skip shape in small: virtual concat, several tiles per wave
    (4, (256, 0), 128, 96, True, True),        # tiles do not divide over the waves; all epilogue operands
    (2, (256, 128), 128, 32, False, True),     # K = 384: 64-channel workgroups
    (1, (512, 0), 256, 32, False, False),      # K = 512, four channel blocks
    (2, (128, 0), 64, 32, False, True),        # N = 64
    (3, (128, 0), 256, 16, False, False),      # 8 tiles per image: fewer tiles than waves
]


@pytest.mark.parametrize("case", STREAM1X1_CASES)
def test_streaming_pointwise_conv(case):
    """cfg = 4: the streaming 1x1 convolution (csrc/pointwise.hip; UNet.py:200 skip_connection on large maps) against fp64
    conv2d.  Same fp32 MFMA chain as the direct kernel in a permuted k order: 1e-5 of the tensor magnitude."""
    import hipops
    B, (c0, c1), N, H, use_temb, use_res = case
    C = c0 + c1
    x = rnd(B, C, H, H, seed=191)
    w = rnd(N, C, 1, 1, seed=192, scale=1.0 / math.sqrt(C))
    b = rnd(N, seed=193, scale=0.1)
    temb = rnd(B, N, seed=196) if use_temb else None
    res = rnd(B, N, H, H, seed=197) if use_res else None
    ref = F.conv2d(x.double(), w.double(), b.double())
    if temb is not None:
        ref = ref + temb[:, :, None, None].double()
    if res is not None:
        ref = ref + res.double()
    xs = hipops.nhwc(x.to(dev()))
    srcs = [xs[..., :c0].contiguous()] + ([xs[..., c0:].contiguous()] if c1 else [])
    got = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), Hout=H, ks=1, temb=temb.to(dev()) if temb is not None else None,
                            res=hipops.nhwc(res.to(dev())) if res is not None else None, cfg=4)
    err = relerr(hipops.nchw(got), ref.float())
    assert err < 1e-5, err
    direct = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), Hout=H, ks=1, temb=temb.to(dev()) if temb is not None else None,
                               res=hipops.nhwc(res.to(dev())) if res is not None else None, cfg=1)
    assert relerr(got, direct) < 1e-5


def test_streaming_pointwise_rejects_fused_operands():
    """cfg = 4 has no GroupNorm / activation / statistics path: the entry point says so instead of ignoring them."""
    import hipops
    x = hipops.nhwc(rnd(1, 128, 32, 32, seed=5).to(dev()))
    w = rnd(128, 128, 1, 1, seed=6).to(dev())
    with pytest.raises(Exception, match="pointwise stream"):
        hipops.conv_igemm([x], w, None, Hout=32, ks=1, act=1, cfg=4)
    with pytest.raises(Exception, match="pointwise stream"):
        hipops.conv_igemm([hipops.nhwc(rnd(1, 64, 32, 32, seed=5).to(dev()))], rnd(128, 64, 1, 1, seed=6).to(dev()), None,
                          Hout=32, ks=1, cfg=4)


@pytest.fixture(params=[0, 3, 6, 7], ids=["auto", "channel-sliced", "one-wave-per-simd", "one-wave-per-simd-tsplit"])
def f43_variant(request):
    """0: the launcher's choice (the 64-channel position-sliced kernel on these small grids); 3: force the channel-sliced
    kernel (csrc/winograd43r.hip, what the 128-channel grids of the real layers run) wherever N % 128 == 0; 6 / 7: the
    one-wave-per-SIMD form of that workgroup (csrc/winograd43w.hip, round 6 measurement kernel; 7 = transform interleaved with the
    MFMA stream) wherever N % 128 == 0 and the launch is not split-K."""
    from anoddpm_amd._lib import lib
    lib().anoddpm_internal_variant(5, request.param)
    yield request.param
    lib().anoddpm_internal_variant(5, 0)


@pytest.mark.parametrize("case", [(2, 64, 128, 32), (1, 128, 128, 64), (4, 64, 256, 64), (3, 96, 128, 48)])
def test_winograd_f43_half_resolution_residual(case, f43_variant):
    """cfg 3 with res_mode 1: the residual is the block input at half resolution, repeated 2x2 on the read (the x_upd path of an
    up-sampling ResBlock, UNet.py:196-198) -- must equal the same launch on the materialised nearest-x2 tensor bit for bit."""
    import hipops
    B, C, N, Hout = case
    x = rnd(B, C, Hout, Hout, seed=41)
    w = rnd(N, C, 3, 3, seed=42, scale=1.0 / math.sqrt(C * 9))
    b = rnd(N, seed=43, scale=0.1)
    gamma, beta = 1 + 0.1 * rnd(C, seed=44), 0.1 * rnd(C, seed=45)
    half = rnd(B, N, Hout // 2, Hout // 2, seed=46)
    full = F.interpolate(half, scale_factor=2, mode="nearest")
    srcs = [hipops.nhwc(x.to(dev()))]
    gn = hipops.gn_affine(srcs, gamma.to(dev()), beta.to(dev()))
    kw = dict(Hout=Hout, ks=3, gn=gn, act=1, cfg=3)
    st0, st1 = [], []
    want = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), res=hipops.nhwc(full.to(dev())).contiguous(), stats_out=st0, **kw)
    got = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), res=hipops.nhwc(half.to(dev())).contiguous(), res_up=True, stats_out=st1, **kw)
    assert torch.equal(got, want)
    assert torch.equal(st0[0], st1[0])
    with pytest.raises(Exception, match="res_mode"):
        hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), res=hipops.nhwc(half.to(dev())).contiguous(), res_up=True, Hout=Hout, ks=3, gn=gn, act=1, cfg=2)


@pytest.mark.parametrize("case", F43_CASES)
def test_winograd_f43_conv(case, f43_variant):
    """cfg = 3: Winograd F(4x4,3x3) (csrc/winograd43.hip) against the direct 3x3 convolution.  fp32 with wider transforms:
    measured ~1e-5 of the tensor magnitude per layer; asserted 1e-4 (north star for whole-model activations: 1e-3)."""
    import hipops
    B, (c0, c1), N, Hout, a_mode, use_gn, act, use_temb, use_res = case[:9]
    ksplit = case[9] if len(case) > 9 else 1
    C = c0 + c1
    Hin = Hout if a_mode == 0 else Hout // 2
    x = rnd(B, C, Hin, Hin, seed=91)
    w = rnd(N, C, 3, 3, seed=92, scale=1.0 / math.sqrt(C * 9))
    b = rnd(N, seed=93, scale=0.1)
    gamma, beta = 1 + 0.1 * rnd(C, seed=94), 0.1 * rnd(C, seed=95)
    temb = rnd(B, N, seed=96) if use_temb else None
    res = rnd(B, N, Hout, Hout, seed=97) if use_res else None
    hh = x
    if use_gn:
        hh = F.group_norm(hh, 32, gamma, beta, eps=1e-5)
    if act:
        hh = F.silu(hh)
    if a_mode == 1:
        hh = F.interpolate(hh, scale_factor=2, mode="nearest")
    ref = F.conv2d(hh.double(), w.double(), b.double(), padding=1)
    if temb is not None:
        ref = ref + temb[:, :, None, None].double()
    if res is not None:
        ref = ref + res.double()
    xs = hipops.nhwc(x.to(dev()))
    srcs = [xs[..., :c0].contiguous()] + ([xs[..., c0:].contiguous()] if c1 else [])
    gn = hipops.gn_affine(srcs, gamma.to(dev()), beta.to(dev())) if use_gn else None
    st = []
    got = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), Hout=Hout, ks=3, gn=gn, act=act, a_mode=a_mode,
                            temb=temb.to(dev()) if temb is not None else None,
                            res=hipops.nhwc(res.to(dev())) if res is not None else None, cfg=3, stats_out=st, ksplit=ksplit)
    err = relerr(hipops.nchw(got), ref.float())
    assert err < 1e-4, err
    # fused statistics: one row per workgroup, sums over its 256 pixels (split-K: the tail's pixel slabs)
    o = hipops.nchw(got).double().cpu()
    s = st[0].double().cpu()
    assert s.shape[1] == ((Hout // 16) ** 2 if ksplit == 1 else 3)
    tot = s.sum(dim=1)                                                # [B][N][2]
    assert torch.allclose(tot[..., 0], o.sum(dim=(2, 3)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(tot[..., 1], (o * o).sum(dim=(2, 3)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("case", [
    # B, (c0, c1), Cout, Hout, a_mode
    (2, (64, 0), 128, 32, 0),          # one source, 2 channels per group... K = 64: cpg 2 (a thread's four channels span two groups)
    (4, (128, 0), 128, 64, 0),         # the 64-channel grid (128 workgroups would not fill the chip): 16 workgroups add to every pair
    (2, (256, 128), 128, 32, 0),       # virtual concat, 12 channels per group: a group straddles the two sources
    (1, (128, 0), 128, 32, 1),         # nearest x2 operand: the statistics are those of the half-resolution source
    (3, (96, 32), 256, 48, 0),         # concat with a narrow second source, three images, two channel blocks
])
def test_winograd_f43_groupnorm_from_atomic_sums(case, f43_variant):
    """Round 6: two F(4x4,3x3) layers with NO finalize launch between them.  The producer adds its output's per-channel
    {sum, sum of squares} to a zeroed [B][N][2] fp64 buffer with device-scope atomics (anoddpm_igemm_args.stats_csum -- every
    workgroup of every XCD must be counted); the consumer finishes nn.GroupNorm(32, C) (UNet.py:409-411) in its prologue from those
    sums (fold_* with fmt 1) and must give what the same launch gives with the affine from anoddpm_gn_finalize, and what fp64
    group_norm -> SiLU -> conv2d gives."""
    import hipops
    B, (c0, c1), N, Hout, a_mode = case
    C = c0 + c1
    Hin = Hout if a_mode == 0 else Hout // 2
    d = dev()
    # producers: one F(4x4) launch per source (16 input channels are enough), output = the consumer's operand
    prod, sums = [], []
    for i, c in enumerate([c0] + ([c1] if c1 else [])):
        if c % 64:                                          # not an F(4x4) width: sums from a reduction instead (tail_csum format)
            t = hipops.nhwc(rnd(B, c, Hin, Hin, seed=300 + i).to(d))
            prod.append(t)
            td = t.double()
            sums.append(torch.stack([td.sum(dim=(1, 2)), (td * td).sum(dim=(1, 2))], dim=-1).contiguous())
            continue
        xin = hipops.nhwc(rnd(B, 16, Hin, Hin, seed=310 + i).to(d))
        wp = rnd(c, 16, 3, 3, seed=320 + i, scale=1.0 / 12).to(d)
        got_sums = []
        y = hipops.conv_igemm([xin], wp, rnd(c, seed=330 + i, scale=0.3).to(d), Hout=Hin, ks=3, cfg=3, csum_out=got_sums)
        prod.append(y)
        sums.append(got_sums[0])
        yd = y.double()
        want = torch.stack([yd.sum(dim=(1, 2)), (yd * yd).sum(dim=(1, 2))], dim=-1)
        assert torch.allclose(got_sums[0], want, rtol=2e-5, atol=2e-3), float((got_sums[0] - want).abs().max())
    gamma, beta = (1 + 0.1 * rnd(C, seed=94)).to(d), (0.1 * rnd(C, seed=95)).to(d)
    w = rnd(N, C, 3, 3, seed=92, scale=1.0 / math.sqrt(C * 9)).to(d)
    b = rnd(N, seed=93, scale=0.1).to(d)
    fold = dict(stats=[(s_, 1) for s_ in sums], gamma=gamma, beta=beta)
    got = hipops.conv_igemm(prod, w, b, Hout=Hout, ks=3, act=1, a_mode=a_mode, cfg=3, fold=fold)
    gn = hipops.gn_affine(prod, gamma, beta)
    base = hipops.conv_igemm(prod, w, b, Hout=Hout, ks=3, gn=gn, act=1, a_mode=a_mode, cfg=3)
    assert relerr(got, base) < 1e-5, relerr(got, base)                 # same kernel, affine from the two GroupNorm routes
    x = torch.cat([hipops.nchw(t) for t in prod], dim=1).double().cpu()
    hh = F.silu(F.group_norm(x, 32, gamma.double().cpu(), beta.double().cpu(), eps=1e-5))
    if a_mode == 1:
        hh = F.interpolate(hh, scale_factor=2, mode="nearest")
    ref = F.conv2d(hh, w.double().cpu(), b.double().cpu(), padding=1)
    assert relerr(hipops.nchw(got).cpu(), ref.float()) < 1e-4


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_conv(case):
    """cfg = 2: Winograd F(2x2,3x3) on the matrix pipe must equal the direct 3x3 convolution (fp32; the
    transform adds a few ulps: tolerance 1e-4 of the tensor magnitude, north star 1e-3)."""
    import hipops
    B, (c0, c1), N, Hout, a_mode, use_gn, act, use_temb, use_res = case[:9]
    ksplit = case[9] if len(case) > 9 else 1
    C = c0 + c1
    Hin = Hout if a_mode == 0 else Hout // 2
    x = rnd(B, C, Hin, Hin, seed=81)
    w = rnd(N, C, 3, 3, seed=82, scale=1.0 / math.sqrt(C * 9))
    b = rnd(N, seed=83, scale=0.1)
    gamma, beta = 1 + 0.1 * rnd(C, seed=84), 0.1 * rnd(C, seed=85)
    temb = rnd(B, N, seed=86) if use_temb else None
    res = rnd(B, N, Hout, Hout, seed=87) if use_res else None
    hh = x
    if use_gn:
        hh = F.group_norm(hh, 32, gamma, beta, eps=1e-5)
    if act:
        hh = F.silu(hh)
    if a_mode == 1:
        hh = F.interpolate(hh, scale_factor=2, mode="nearest")
    ref = F.conv2d(hh, w, b, padding=1)
    if temb is not None:
        ref = ref + temb[:, :, None, None]
    if res is not None:
        ref = ref + res
    xs = hipops.nhwc(x.to(dev()))
    srcs = [xs[..., :c0].contiguous()] + ([xs[..., c0:].contiguous()] if c1 else [])
    gn = hipops.gn_affine(srcs, gamma.to(dev()), beta.to(dev())) if use_gn else None
    st = []
    got = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), Hout=Hout, ks=3, gn=gn, act=act, a_mode=a_mode,
                            temb=temb.to(dev()) if temb is not None else None,
                            res=hipops.nhwc(res.to(dev())) if res is not None else None, cfg=2, ksplit=ksplit,
                            stats_out=st)
    err = relerr(hipops.nchw(got), ref)
    assert err < 1e-4, err
    # fused statistics of the Winograd epilogue
    if N % 32 == 0:
        g2, b2 = 1 + 0.1 * rnd(N, seed=88), 0.1 * rnd(N, seed=89)
        sc, sh = hipops.gn_finalize(st, g2.to(dev()), b2.to(dev()), Hout * Hout)
        assert relerr(hipops.nchw(got * sc[:, None, None, :] + sh[:, None, None, :]), F.group_norm(ref, 32, g2, b2, eps=1e-5)) < 2e-4


# ---------------------------------------------------------------------------- backward twins of the 3x3 convolution
BWD_CASES = [
    # B, (c0, c1), Cout, Hout, a_mode, gn+act
    (2, (64, 0), 64, 32, 0, True),
    (1, (128, 0), 128, 64, 0, True),
    (2, (64, 64), 128, 16, 0, True),        # virtual concat
    (1, (96, 32), 96, 32, 0, True),         # channel tails on both sides (96 = 64 + 32)
    (1, (64, 0), 64, 32, 1, True),          # fused nearest x2
    (1, (32, 0), 64, 48, 0, False),         # plain input, 16-pixel segments (W = 48), ragged bands
    (2, (64, 0), 64, 16, 2, True),          # fused 2x2 average of the activated input (down blocks)
    (2, (128, 0), 64, 8, 0, True),          # 8x8 map: 8-pixel segments
    (2, (64, 0), 128, 4, 0, True),          # 4x4 map (bottom of the 32x32 models)
]


def _bwd_reference(case):
    B, (c0, c1), N, Hout, a_mode, fused = case
    C = c0 + c1
    Hin = Hout if a_mode == 0 else (Hout // 2 if a_mode == 1 else Hout * 2)
    x = rnd(B, C, Hin, Hin, seed=91)
    w = rnd(N, C, 3, 3, seed=92, scale=1.0 / math.sqrt(C * 9)).requires_grad_(True)
    gamma, beta = 1 + 0.1 * rnd(C, seed=93), 0.1 * rnd(C, seed=94)
    dy = rnd(B, N, Hout, Hout, seed=95)
    hh = x
    if fused:
        hh = F.silu(F.group_norm(hh, 32, gamma, beta, eps=1e-5))
    if a_mode == 1:
        hh = F.interpolate(hh, scale_factor=2, mode="nearest")
    elif a_mode == 2:
        hh = F.avg_pool2d(hh, 2)
    hh = hh.detach().requires_grad_(True)
    y = F.conv2d(hh.double(), w.double(), None, padding=1)
    y.backward(dy.double())
    return x, w.detach(), gamma, beta, dy, w.grad.float(), hh.grad.float()


@pytest.mark.parametrize("case", BWD_CASES)
def test_conv3x3_wgrad(case):
    """anoddpm_conv3x3_wgrad against autograd (fp64) of F.conv2d on the same fused input."""
    import hipops
    B, (c0, c1), N, Hout, a_mode, fused = case
    x, w, gamma, beta, dy, dw_ref, _ = _bwd_reference(case)
    xs = hipops.nhwc(x.to(dev()))
    srcs = [xs[..., :c0].contiguous()] + ([xs[..., c0:].contiguous()] if c1 else [])
    gn = hipops.gn_affine(srcs, gamma.to(dev()), beta.to(dev())) if fused else None
    dyn = hipops.nhwc(dy.to(dev())).contiguous()
    cs = []
    got = hipops.conv_wgrad(srcs, dyn, gn=gn, act=1 if fused else 0, a_mode=a_mode, band=7 if Hout == 48 else None, colsum_out=cs)
    assert got.shape == dw_ref.shape
    # the fused column sums of dY (bias / embedding gradients): per image sum over pixels
    assert relerr(cs[0].sum(dim=1), dy.sum(dim=(2, 3))) < 1e-5
    assert relerr(got, dw_ref) < 2e-5, relerr(got, dw_ref)
    # accumulate mode adds into an existing gradient (two micro-batches of the same data = 2x)
    acc = got.clone()
    hipops.conv_wgrad(srcs, dyn, gn=gn, act=1 if fused else 0, a_mode=a_mode, accumulate_into=acc)
    assert relerr(acc, 2 * dw_ref) < 2e-5


WGRAD43_CASES = [
    # B, (c0, c1), Cout, Hout, a_mode, gn+act
    (1, (32, 0), 64, 16, 0, True),          # one (32, 64) block, two patches
    (2, (64, 0), 64, 32, 0, True),          # two input-channel blocks, images interleaved over the patch groups
    (1, (128, 0), 128, 64, 0, True),        # 8 blocks x 32 patches
    (2, (64, 64), 128, 32, 0, True),        # virtual concat
    (1, (64, 0), 64, 32, 1, True),          # fused nearest x2
    (1, (96, 32), 64, 48, 0, True),         # three patches per row (not a power of two), sources split at 96
    (3, (32, 0), 128, 16, 0, True),         # fewer patches than the 256 / blocks group count
]


@pytest.mark.parametrize("case", WGRAD43_CASES)
def test_conv3x3_wgrad_winograd_domain(case):
    """algo = 1 of anoddpm_conv3x3_wgrad (csrc/wgrad43.hip): dU = sum_tiles V (.) Z, dg = G^T dU G -- the adjoint of the
    F(4x4,3x3) forward kernel -- against fp64 autograd of F.conv2d and against the direct kernel.  fp32 with the wide
    transforms: asserted 1e-4 of the gradient's magnitude (direct kernel: 2e-5)."""
    import hipops
    B, (c0, c1), N, Hout, a_mode, fused = case
    x, w, gamma, beta, dy, dw_ref, _ = _bwd_reference(case)
    xs = hipops.nhwc(x.to(dev()))
    srcs = [xs[..., :c0].contiguous()] + ([xs[..., c0:].contiguous()] if c1 else [])
    gn = hipops.gn_affine(srcs, gamma.to(dev()), beta.to(dev()))
    dyn = hipops.nhwc(dy.to(dev())).contiguous()
    cs = []
    got = hipops.conv_wgrad(srcs, dyn, gn=gn, act=1, a_mode=a_mode, colsum_out=cs, algo=1)
    assert relerr(got, dw_ref) < 1e-4, relerr(got, dw_ref)
    direct = hipops.conv_wgrad(srcs, dyn, gn=gn, act=1, a_mode=a_mode)
    assert relerr(got, direct) < 1e-4
    # column sums of dY: one row per workgroup set and tile row of the kernel (it sums over its patches of an image; sets without a
    # patch in an image write zeros) -> per image sums over pixels (embedding gradient) and, over the batch, the bias gradient
    from anoddpm_amd._lib import lib
    K = c0 + c1
    items = lib().anoddpm_wgrad43_colsum_items(K, N, B, Hout, Hout)
    assert items == 2 * lib().anoddpm_wgrad43_groups(K, N, B, Hout, Hout)
    assert cs[0].shape == (B, items, N) and torch.isfinite(cs[0]).all()
    assert relerr(cs[0].sum(dim=1), dy.sum(dim=(2, 3))) < 1e-5
    # the same sums folded by the extra workgroups of the fold launch (dimg / dbias): no anoddpm_colsum_fold launch
    bo = {"dbias": torch.ones(N, device=dev())}
    got2 = hipops.conv_wgrad(srcs, dyn, gn=gn, act=1, a_mode=a_mode, colsum_out=[], algo=1, bias_out=bo)
    assert torch.equal(got2, got)
    assert relerr(bo["dimg"], dy.sum(dim=(2, 3))) < 1e-5
    assert relerr(bo["dbias"], 1.0 + dy.sum(dim=(0, 2, 3))) < 1e-5
    acc = got.clone()
    hipops.conv_wgrad(srcs, dyn, gn=gn, act=1, a_mode=a_mode, accumulate_into=acc, algo=1)
    assert relerr(acc, 2 * dw_ref) < 1e-4


def test_conv3x3_wgrad_winograd_domain_rejects_other_shapes():
    import hipops
    x = hipops.nhwc(rnd(1, 32, 16, 16, seed=3).to(dev()))
    dyn = hipops.nhwc(rnd(1, 64, 16, 16, seed=4).to(dev())).contiguous()
    with pytest.raises(Exception, match="Winograd"):
        hipops.conv_wgrad([x], dyn, algo=1)                              # no fused GroupNorm + SiLU operand


@pytest.mark.parametrize("case", [c for c in BWD_CASES if c[4] == 0])
def test_conv3x3_dgrad_is_forward_kernel_on_flipped_weights(case):
    """The data gradient w.r.t. the (activated) conv input is anoddpm_igemm on dY with the spatially flipped,
    channel-transposed weights -- direct and Winograd kernels."""
    import hipops
    B, (c0, c1), N, Hout, a_mode, fused = case
    x, w, gamma, beta, dy, _, da_ref = _bwd_reference(case)
    wt = w.flip(2, 3).transpose(0, 1).contiguous()                      # [Cin][Cout][3][3]
    dyn = hipops.nhwc(dy.to(dev())).contiguous()
    pow2 = (Hout & (Hout - 1)) == 0                                     # the direct 3x3 kernel needs a power-of-two width
    for cfg in ([1] if pow2 else []) + ([2] if Hout % 16 == 0 else []):
        got = hipops.conv_igemm([dyn], wt.to(dev()), None, Hout=Hout, ks=3, cfg=cfg)
        assert relerr(hipops.nchw(got), da_ref) < (1e-4 if cfg == 2 else 1e-5), (cfg, relerr(hipops.nchw(got), da_ref))


GN_BWD_CASES = [
    # B, (c0, c1), Hs, act, a_mode
    (2, (64, 0), 16, 1, 0),
    (1, (128, 0), 32, 1, 0),
    (2, (96, 32), 8, 1, 0),          # concat, group straddles nothing (cpg 4)
    (1, (256, 128), 8, 1, 0),        # concat where groups of 12 channels straddle the two sources
    (2, (64, 0), 8, 1, 1),           # forward upsampled the activation: da at 2x resolution
    (2, (64, 0), 16, 1, 2),          # forward average-pooled the activation: da at half resolution
    (1, (64, 0), 16, 0, 0),          # GroupNorm alone (attention blocks)
]


@pytest.mark.parametrize("case", GN_BWD_CASES)
def test_gn_silu_backward(case):
    """anoddpm_gn_silu_backward against autograd (fp64) of [resample](silu(group_norm(cat(x))))."""
    import hipops
    B, (c0, c1), Hs, act, a_mode = case
    C = c0 + c1
    x = rnd(B, C, Hs, Hs, seed=101)
    gamma, beta = 1 + 0.1 * rnd(C, seed=102), 0.1 * rnd(C, seed=103)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.group_norm(xd, 32, gd, bd, eps=1e-5)
    if act:
        a = F.silu(a)
    if a_mode == 1:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
    elif a_mode == 2:
        a = F.avg_pool2d(a, 2)
    da = rnd(*a.shape, seed=104)
    a.backward(da.double())
    xs = hipops.nhwc(x.to(dev()))
    srcs = [xs[..., :c0].contiguous()] + ([xs[..., c0:].contiguous()] if c1 else [])
    nslab = 4
    stats = [hipops.chan_stats(s_, nslab) for s_ in srcs]
    sc, sh, mean, rstd = hipops.gn_finalize(stats, gamma.to(dev()), beta.to(dev()), Hs * Hs, want_mean_rstd=True)
    ref_mean = x.reshape(B, 32, -1).double().mean(dim=2)
    assert relerr(mean, ref_mean.float()) < 1e-5
    dan = hipops.nhwc(da.to(dev())).contiguous()
    dx, dgamma, dbeta = hipops.gn_silu_backward(srcs, dan, gamma.to(dev()), beta.to(dev()), mean, rstd, act=act, a_mode=a_mode)
    got = hipops.nchw(torch.cat(dx, dim=3))
    assert relerr(got, xd.grad.float()) < 2e-5, relerr(got, xd.grad.float())
    assert relerr(dgamma, gd.grad.float()) < 2e-5 and relerr(dbeta, bd.grad.float()) < 2e-5
    # fan-in: accumulate into an existing gradient
    base = [torch.ones_like(d) for d in dx]
    dx2, _, _ = hipops.gn_silu_backward(srcs, dan, gamma.to(dev()), beta.to(dev()), mean, rstd, act=act, a_mode=a_mode, acc_into=base)
    assert relerr(hipops.nchw(torch.cat(dx2, dim=3)), xd.grad.float() + 1.0) < 2e-5


@pytest.mark.parametrize("case", [(2, (128, 0), 64, 32, 3), (1, (128, 128), 128, 32, 3), (3, (128, 0), 128, 128, 0), (5, (256, 128), 128, 64, 0)])
def test_f43_data_gradient_writes_the_gn_backward_partials(case):
    """Round 6: the data-gradient launch on the channel-sliced F(4x4,3x3) kernel with anoddpm_igemm_args.gnb_* set also performs
    the reduction pass of anoddpm_gn_silu_backward (x read in the epilogue's residual slot).  da must not change, and the
    GroupNorm backward fed with those partial rows (partial_ready = 1: fold + elementwise launches only) must equal the
    three-launch form and fp64 autograd.  Selector 3 forces the kernel on the small grids; 0 is the launcher's own choice."""
    import hipops
    from anoddpm_amd._lib import lib
    B, (c0, c1), N, H, sel = case
    C = c0 + c1
    x = rnd(B, C, H, H, seed=201)
    gamma, beta = 1 + 0.1 * rnd(C, seed=202), 0.1 * rnd(C, seed=203)
    dy = rnd(B, N, H, H, seed=204)
    wt = rnd(C, N, 3, 3, seed=205, scale=1.0 / math.sqrt(N * 9))          # the data gradient's (flipped, transposed) weights: N -> C
    xs = hipops.nhwc(x.to(dev()))
    srcs = [xs[..., :c0].contiguous()] + ([xs[..., c0:].contiguous()] if c1 else [])
    stats = [hipops.chan_stats(s_, 4) for s_ in srcs]
    _, _, mean, rstd = hipops.gn_finalize(stats, gamma.to(dev()), beta.to(dev()), H * H, want_mean_rstd=True)
    dyn = hipops.nhwc(dy.to(dev())).contiguous()
    g, bt = gamma.to(dev()), beta.to(dev())
    lib().anoddpm_internal_variant(5, sel)
    try:
        assert lib().anoddpm_f43_channel_sliced(H, H, C, B) == 1
        plain = hipops.conv_igemm([dyn], wt.to(dev()), None, Hout=H, ks=3, cfg=3)
        gnb = dict(srcs=srcs, gamma=g, beta=bt, mean=mean, rstd=rstd)
        fused = hipops.conv_igemm([dyn], wt.to(dev()), None, Hout=H, ks=3, cfg=3, gnb=gnb)
    finally:
        lib().anoddpm_internal_variant(5, 0)
    assert torch.equal(plain, fused)
    part = gnb["partial"]
    assert part.shape == (B, (H // 16) ** 2, C, 2) and torch.isfinite(part).all()
    dx0, dg0, db0 = hipops.gn_silu_backward(srcs, plain, g, bt, mean, rstd, act=1, a_mode=0)
    dx1, dg1, db1 = hipops.gn_silu_backward(srcs, plain, g, bt, mean, rstd, act=1, a_mode=0, partial=part)
    assert relerr(torch.cat(dx1, dim=3), torch.cat(dx0, dim=3)) < 1e-5
    assert relerr(dg1, dg0) < 1e-5 and relerr(db1, db0) < 1e-5
    # and against fp64 autograd of silu(group_norm(x)) with the same upstream gradient
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    F.silu(F.group_norm(xd, 32, gd, bd, eps=1e-5)).backward(hipops.nchw(plain).double().cpu())
    assert relerr(hipops.nchw(torch.cat(dx1, dim=3)), xd.grad.float()) < 2e-5
    assert relerr(dg1, gd.grad.float()) < 2e-5 and relerr(db1, bd.grad.float()) < 2e-5
    # a launch that does not run on the channel-sliced kernel refuses the request instead of ignoring it
    if lib().anoddpm_f43_channel_sliced(32, 32, 128, 1) == 0:
        small = hipops.nhwc(rnd(1, 64, 32, 32, seed=206).to(dev())).contiguous()
        xs2 = [hipops.nhwc(rnd(1, 128, 32, 32, seed=207).to(dev())).contiguous()]
        m2, r2 = torch.zeros(1, 32, device=dev()), torch.ones(1, 32, device=dev())
        with pytest.raises(Exception, match="gnb_partial"):
            hipops.conv_igemm([small], rnd(128, 64, 3, 3, seed=208).to(dev()), None, Hout=32, ks=3, cfg=3,
                              gnb=dict(srcs=xs2, gamma=g[:128].contiguous(), beta=bt[:128].contiguous(), mean=m2, rstd=r2))


@pytest.mark.parametrize("N,K", [(64, 32), (128, 256), (96, 36)])
def test_device_weight_packing_matches_host_packers(N, K):
    """anoddpm_pack_conv3x3 (training re-packs on the device) against the host packers used by the inference plan."""
    from hipops import _pack_conv, _pack_wino
    from anoddpm_amd._lib import check, current_stream, lib
    w = rnd(N, K, 3, 3, seed=111).to(dev())
    for bwd in (0, 1):
        src = w.flip(2, 3).transpose(0, 1).contiguous() if bwd else w
        if (N if bwd else K) % 4:
            continue
        for mode, host in ((0, _pack_conv), (1, _pack_wino)):
            out = torch.full(((16 if mode else 9) * N * K,), float("nan"), device=dev())
            check(lib().anoddpm_pack_conv3x3(w.data_ptr(), out.data_ptr(), N, K, mode, bwd, current_stream()), "pack")
            ref = host(src).reshape(-1)
            assert out.shape == ref.shape
            assert torch.allclose(out, ref, rtol=1e-6, atol=1e-7), (mode, bwd, (out - ref).abs().max().item())


@pytest.mark.parametrize("cfg,ks,H,K,N,c1,ksplit", [(2, 3, 16, 256, 256, 0, 4), (2, 3, 32, 128, 128, 256, 2), (1, 3, 8, 256, 512, 0, 8),
                                                    (1, 1, 16, 512, 256, 128, 4), (2, 3, 16, 128, 256, 384, 2)])
def test_splitk_tail_with_fused_groupnorm(cfg, ks, H, K, N, c1, ksplit):
    """anoddpm_igemm's group-partitioned split-K tail (tail_csum / tail_gamma ...): same output as the slab tail, folded per-channel
    sums, and the consumer GroupNorm(32, N + c1) over the virtual concat [out, other] (UNet.py:402, 409-411) equal to
    torch.nn.functional.group_norm's statistics -- incl. a group that straddles the two sources (N = 256, c1 = 128: 12 per group)."""
    import hipops as H_
    DEV = "cuda:0"
    torch.manual_seed(7)
    B = 3
    x = torch.randn(B, H, H, K, device=DEV)
    w = torch.randn(N, K, ks, ks, device=DEV) * (1.0 / (K * ks * ks) ** 0.5)
    bias = torch.randn(N, device=DEV)
    temb = torch.randn(B, N, device=DEV)
    res = torch.randn(B, H, H, N, device=DEV)
    ref = H_.conv_igemm([x], w, bias, Hout=H, ks=ks, temb=temb, res=res, cfg=cfg, ksplit=ksplit)
    other = torch.randn(B, H, H, c1, device=DEV) * 1.7 + 0.3 if c1 else None
    other_csum = None
    if c1:
        of = other.double().reshape(B, -1, c1)
        other_csum = torch.stack([of.sum(1), (of * of).sum(1)], dim=-1).contiguous()
    C = N + c1
    gamma, beta = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    tail = dict(gamma=gamma, beta=beta, other_csum=other_csum, want_mean=True)
    out = H_.conv_igemm([x], w, bias, Hout=H, ks=ks, temb=temb, res=res, cfg=cfg, ksplit=ksplit, gn_tail=tail)
    assert (out - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()        # same slab order; the compilers' FMA choices may differ
    of = out.double().reshape(B, -1, N)
    np.testing.assert_allclose(tail["csum"][..., 0].cpu().numpy(), of.sum(1).cpu().numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(tail["csum"][..., 1].cpu().numpy(), (of * of).sum(1).cpu().numpy(), rtol=1e-9)
    cat = torch.cat([out, other], dim=-1) if c1 else out
    xn = cat.permute(0, 3, 1, 2).double()
    g = xn.reshape(B, 32, -1)
    mean, var = g.mean(-1), g.var(-1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    np.testing.assert_allclose(tail["mean"].cpu().numpy(), mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tail["rstd"].cpu().numpy(), rstd.cpu().numpy(), rtol=1e-5)
    y = torch.nn.functional.group_norm(xn, 32, gamma.double(), beta.double(), eps=1e-5).permute(0, 2, 3, 1)
    got = cat.double() * tail["scale"].double()[:, None, None, :] + tail["shift"].double()[:, None, None, :]
    assert ((got - y).abs().max() / y.abs().max()).item() < 1e-5
    # folded sums only (no consumer GroupNorm attached)
    tail2 = dict(gamma=None)
    out2 = H_.conv_igemm([x], w, bias, Hout=H, ks=ks, temb=temb, res=res, cfg=cfg, ksplit=ksplit, gn_tail=tail2)
    assert torch.equal(out2, out)
    np.testing.assert_allclose(tail2["csum"].cpu().numpy(), tail["csum"].cpu().numpy(), rtol=1e-12)      # another partition: another fp64 summation order
    # anoddpm_gn_finalize on the folded sums (the launch a second consumer GroupNorm of the same tensor takes)
    from anoddpm_amd._lib import GnFinalizeArgs, check, current_stream, lib
    import ctypes
    st = GnFinalizeArgs()
    sc, sh = torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV)
    st.stats0, st.rows0, st.fmt0 = tail["csum"].data_ptr(), 1, 1
    st.stats1, st.rows1, st.fmt1 = (other_csum.data_ptr(), 1, 1) if c1 else (None, 0, 0)
    st.gamma, st.beta, st.scale, st.shift = gamma.data_ptr(), beta.data_ptr(), sc.data_ptr(), sh.data_ptr()
    st.c0, st.c1, st.P, st.B, st.groups, st.eps = N, c1, H * H, B, 32, 1e-5
    check(lib().anoddpm_gn_finalize(ctypes.byref(st), current_stream()), "gn_finalize")
    assert (sc - tail["scale"]).abs().max().item() < 1e-6 * tail["scale"].abs().max().item()
    assert (sh - tail["shift"]).abs().max().item() < 1e-5 * max(tail["shift"].abs().max().item(), 1.0)


# ---- cfg 5: small maps without split-K (csrc/smallmap.hip), GroupNorm of the operand finished in the prologue -----------------
SMALL_CASES = [
    # B, (c0, c1), N, H, ks, gn ("no" / "given" / "fold" / "fold64": folded from fp64 sums), act, temb, res
    (4, (512, 0), 512, 8, 3, "fold", 1, True, False),        # 8x8 ResBlock conv1: 16 x 32 tiles, two K chunks
    (4, (512, 0), 512, 8, 3, "fold", 1, False, True),        # conv2 + residual
    (4, (512, 512), 512, 8, 3, "fold", 1, True, False),      # up path: virtual concat, four chunks
    (4, (512, 512), 512, 8, 1, "no", 0, False, False),       # 1x1 skip over the concat
    (4, (512, 0), 1536, 8, 1, "fold", 0, False, False),      # to_qkv: GroupNorm, no SiLU
    (4, (512, 0), 512, 8, 1, "no", 0, False, True),          # proj_out + residual
    (4, (512, 0), 1536, 16, 1, "given", 0, False, False),    # 16x16 to_qkv: 32 x 96 tiles, precomputed affine
    (4, (512, 0), 512, 16, 1, "no", 0, False, True),         # 16x16 proj_out
    (4, (768, 256), 512, 16, 1, "no", 0, False, False),      # 16x16 skip over a concat whose first source is not a chunk multiple
    (2, (96, 32), 96, 8, 3, "fold", 1, True, True),          # narrow test-model widths: partial chunk, K tail inside a wave's slice
    (3, (64, 0), 32, 4, 3, "fold64", 1, False, True),        # 4x4 level: one tile per image; statistics given as fp64 sums
    (2, (128, 0), 384, 16, 1, "fold", 0, False, False),      # 16x16 qkv at test width
    (1, (256, 0), 256, 16, 3, "given", 1, True, True),       # batch 1 (configuration 5): 16x16 3x3, 16-row tiles
    (2, (32, 0), 64, 8, 3, "fold", 1, False, False),         # K = 32: seven of eight waves idle
]


@pytest.mark.parametrize("case", SMALL_CASES)
def test_smallmap_contraction(case):
    """anoddpm_igemm cfg 5 against torch on the same fused expression: GroupNorm (folded from statistics rows, from fp64 sums, or
    given) -> SiLU -> conv / 1x1 over one or two sources -> + bias + temb + residual; and its output statistics rows."""
    import hipops
    from anoddpm_amd._lib import lib
    B, (c0, c1), N, H, ks, gn_mode, act, use_temb, use_res = case
    C = c0 + c1
    x = rnd(B, C, H, H, seed=21) * 1.5 + 0.3
    w = rnd(N, C, ks, ks, seed=22, scale=1.0 / math.sqrt(C * ks * ks))
    b = rnd(N, seed=23, scale=0.1)
    gamma, beta = 1 + 0.1 * rnd(C, seed=24), 0.1 * rnd(C, seed=25)
    temb = rnd(B, N, seed=26) if use_temb else None
    res = rnd(B, N, H, H, seed=27) if use_res else None
    h = x
    if gn_mode != "no":
        h = F.group_norm(h, 32, gamma, beta, eps=1e-5)
    if act:
        h = F.silu(h)
    ref = F.conv2d(h, w, b, padding=ks // 2)
    if temb is not None:
        ref = ref + temb[:, :, None, None]
    if res is not None:
        ref = ref + res
    xs = hipops.nhwc(x.to(dev()))
    srcs = [xs[..., :c0].contiguous()] + ([xs[..., c0:].contiguous()] if c1 else [])
    assert lib().anoddpm_smallmap_tile(ks, H, H, C, c0, N, B) != 0
    gn = fold = None
    if gn_mode == "given":
        gn = hipops.gn_affine(srcs, gamma.to(dev()), beta.to(dev()))
    elif gn_mode in ("fold", "fold64"):
        stats = []
        for i, s_ in enumerate(srcs):
            rows = hipops.chan_stats(s_, nslab=(3 if H > 4 else 1) + i)          # different row counts per source
            if gn_mode == "fold64":
                stats.append((rows.double().sum(1).contiguous(), 1))                 # [B][c][2] fp64, as a split-K tail's tail_csum
            else:
                stats.append((rows, 0))
        fold = dict(stats=stats, gamma=gamma.to(dev()), beta=beta.to(dev()))
    st = []
    got = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), Hout=H, ks=ks, gn=gn, act=act, fold=fold,
                            temb=temb.to(dev()) if temb is not None else None,
                            res=hipops.nhwc(res.to(dev())) if res is not None else None, cfg=5, stats_out=st)
    assert relerr(hipops.nchw(got), ref) < TOL
    # statistics rows of the output: their GroupNorm equals torch's on the result
    g2, b2 = 1 + 0.1 * rnd(N, seed=28), 0.1 * rnd(N, seed=29)
    sc, sh = hipops.gn_finalize(st, g2.to(dev()), b2.to(dev()), H * H)
    assert relerr(hipops.nchw(got * sc[:, None, None, :] + sh[:, None, None, :]), F.group_norm(ref, 32, g2, b2, eps=1e-5)) < TOL
    # deterministic: the cross-wave fold has a fixed order
    got2 = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), Hout=H, ks=ks, gn=gn, act=act, fold=fold,
                             temb=temb.to(dev()) if temb is not None else None,
                             res=hipops.nhwc(res.to(dev())) if res is not None else None, cfg=5)
    assert torch.equal(got, got2)


def test_smallmap_rejects_what_it_does_not_take():
    import hipops
    from anoddpm_amd._lib import AnoddpmError, lib
    assert lib().anoddpm_smallmap_tile(3, 32, 32, 256, 256, 256, 4) == 0          # > 256 pixels
    assert lib().anoddpm_smallmap_tile(3, 8, 8, 512, 512, 500, 4) == 0            # N % 32
    assert lib().anoddpm_smallmap_tile(1, 8, 8, 2048, 2048, 512, 4) == 0          # K > 1024
    x = hipops.nhwc(rnd(1, 64, 32, 32, seed=1).to(dev()))
    with pytest.raises(AnoddpmError):
        hipops.conv_igemm([x], rnd(64, 64, 3, 3, seed=2).to(dev()), None, Hout=32, ks=3, cfg=5)
    x8 = hipops.nhwc(rnd(1, 64, 8, 8, seed=1).to(dev()))
    with pytest.raises(AnoddpmError):                                              # the fold is a cfg 5 feature
        hipops.conv_igemm([x8], rnd(64, 64, 3, 3, seed=2).to(dev()), None, Hout=8, ks=3, cfg=1,
                          fold=dict(stats=[(hipops.chan_stats(x8, 1), 0)], gamma=torch.ones(64, device=dev()), beta=torch.zeros(64, device=dev())))


# ---- cfg 6: Winograd F(2x2,3x3) on 16x16 / 32x32 maps without split-K (csrc/wino23s.hip) -------------------------------------
WINO23S_CASES = [
    # B, (c0, c1), N, H, a_mode, gn ("no" / "given" / "fold" / "fold64"), act, temb, res
    (4, (512, 0), 512, 16, 0, "fold", 1, True, False),       # 16x16 ResBlock conv1: 32-channel workgroups, 16 chunks
    (4, (512, 0), 512, 16, 0, "fold", 1, False, True),       # conv2 + residual
    (4, (512, 512), 512, 16, 0, "fold64", 1, True, False),   # up path: virtual concat; statistics as fp64 sums (a split-K tail's)
    (4, (256, 0), 256, 32, 0, "fold", 1, True, True),        # 32x32: 64-channel workgroups
    (4, (512, 256), 256, 32, 0, "given", 1, True, False),    # 32x32 concat, precomputed affine
    (4, (512, 0), 512, 16, 1, "fold", 1, True, False),       # up block: nearest x2 from 8x8 fused into the operand load
    (12, (64, 32), 96, 16, 0, "fold", 1, False, True),       # narrow test-model widths: N = 96 -> 32-channel tiles, cpg = 3
    (2, (128, 0), 128, 32, 0, "no", 0, False, False),        # plain operand (Upsample-style conv)
    (4, (64, 0), 64, 32, 1, "given", 1, False, False),       # nearest x2 from 16x16, two chunks only
]


@pytest.mark.parametrize("case", WINO23S_CASES)
def test_wino23s_conv(case):
    """anoddpm_igemm cfg 6 against torch: GroupNorm (folded / given) -> SiLU -> [nearest x2] -> 3x3 conv over one or two sources
    -> + bias + temb + residual, and its output statistics rows.  Winograd F(2x2,3x3) rounding: 1e-4 of the output's magnitude
    (the bar of WINO_CASES)."""
    import hipops
    from anoddpm_amd._lib import lib
    B, (c0, c1), N, H, a_mode, gn_mode, act, use_temb, use_res = case
    C = c0 + c1
    Hin = H // 2 if a_mode == 1 else H
    x = rnd(B, C, Hin, Hin, seed=21) * 1.5 + 0.3
    w = rnd(N, C, 3, 3, seed=22, scale=1.0 / math.sqrt(C * 9))
    b = rnd(N, seed=23, scale=0.1)
    gamma, beta = 1 + 0.1 * rnd(C, seed=24), 0.1 * rnd(C, seed=25)
    temb = rnd(B, N, seed=26) if use_temb else None
    res = rnd(B, N, H, H, seed=27) if use_res else None
    h = x
    if gn_mode != "no":
        h = F.group_norm(h, 32, gamma, beta, eps=1e-5)
    if act:
        h = F.silu(h)
    if a_mode == 1:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
    ref = F.conv2d(h, w, b, padding=1)
    if temb is not None:
        ref = ref + temb[:, :, None, None]
    if res is not None:
        ref = ref + res
    xs = hipops.nhwc(x.to(dev()))
    srcs = [xs[..., :c0].contiguous()] + ([xs[..., c0:].contiguous()] if c1 else [])
    assert lib().anoddpm_wino23s_tile(H, H, C, c0, N, B, a_mode) != 0
    gn = fold = None
    if gn_mode == "given":
        gn = hipops.gn_affine(srcs, gamma.to(dev()), beta.to(dev()))
    elif gn_mode in ("fold", "fold64"):
        stats = []
        for i, s_ in enumerate(srcs):
            rows = hipops.chan_stats(s_, nslab=3 + i)
            stats.append((rows.double().sum(1).contiguous(), 1) if gn_mode == "fold64" else (rows, 0))
        fold = dict(stats=stats, gamma=gamma.to(dev()), beta=beta.to(dev()))
    st = []
    kw = dict(Hout=H, ks=3, gn=gn, act=act, a_mode=a_mode, fold=fold, temb=temb.to(dev()) if temb is not None else None,
              res=hipops.nhwc(res.to(dev())) if res is not None else None, cfg=6)
    got = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), stats_out=st, **kw)
    assert relerr(hipops.nchw(got), ref) < 1e-4
    g2, b2 = 1 + 0.1 * rnd(N, seed=28), 0.1 * rnd(N, seed=29)
    sc, sh = hipops.gn_finalize(st, g2.to(dev()), b2.to(dev()), H * H)
    assert relerr(hipops.nchw(got * sc[:, None, None, :] + sh[:, None, None, :]), F.group_norm(ref, 32, g2, b2, eps=1e-5)) < 2e-4
    got2 = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), **kw)
    assert torch.equal(got, got2)


def test_wino23s_rejects_what_it_does_not_take():
    from anoddpm_amd._lib import lib
    assert lib().anoddpm_wino23s_tile(64, 64, 256, 256, 256, 4, 0) == 0           # only 16x16 / 32x32 maps
    assert lib().anoddpm_wino23s_tile(16, 16, 48, 48, 64, 4, 0) == 0              # K % 32
    assert lib().anoddpm_wino23s_tile(16, 16, 512, 512, 512, 4, 2) == 0           # pooled operand
    assert lib().anoddpm_wino23s_tile(16, 16, 512, 512, 512, 1, 0) == 0           # batch 1: 64 workgroups, the split-K kernel fills the chip
    assert lib().anoddpm_wino23s_tile(32, 32, 256, 256, 256, 4, 0) == 4 and lib().anoddpm_wino23s_tile(16, 16, 512, 512, 512, 4, 0) == 2


# ---- cfg 7: F(4x4,3x3) with split-bf16 products (csrc/winograd43b.hip) -- the opt-in side configuration ------------------------
BF3_CASES = [
    # B, (c0, c1), Cout, Hout, a_mode, gn, act, temb, res
    (1, (32, 0), 128, 16, 0, False, 0, False, False),      # one workgroup, one K iteration
    (2, (64, 0), 128, 32, 0, True, 1, True, True),         # full ResBlock conv: GN + SiLU + temb + residual
    (1, (128, 0), 256, 64, 0, True, 1, False, False),      # two channel blocks
    (2, (64, 64), 128, 32, 0, True, 1, True, True),        # virtual concat
    (1, (64, 0), 128, 32, 1, True, 1, False, False),       # fused nearest x2
    (1, (96, 0), 128, 48, 0, True, 1, False, True),        # non power-of-two image, three K iterations
    (1, (128, 0), 128, 32, 0, False, 0, False, False),     # no GroupNorm / activation
]


@pytest.mark.parametrize("case", BF3_CASES)
def test_winograd_f43_bf16split3(case):
    """The split-bf16 side configuration against fp64 on the fused layer: it must stay in the accuracy class of the fp32 F(4x4,3x3)
    kernel (three bf16 pieces = 24 bits; the products dropped are below 2^-24) -- asserted at 2x the fp32 kernel's own error, and
    bit-reproducible."""
    import hipops
    B, (c0, c1), N, Hout, a_mode, use_gn, act, use_temb, use_res = case
    C = c0 + c1
    Hin = Hout if a_mode == 0 else Hout // 2
    x = rnd(B, C, Hin, Hin, seed=11)
    w = rnd(N, C, 3, 3, seed=12, scale=1.0 / math.sqrt(C * 9))
    b = rnd(N, seed=13, scale=0.1)
    gamma, beta = 1 + 0.1 * rnd(C, seed=14), 0.1 * rnd(C, seed=15)
    temb = rnd(B, N, seed=16) if use_temb else None
    res = rnd(B, N, Hout, Hout, seed=17) if use_res else None
    h = x.double()
    if use_gn:
        h = F.group_norm(h, 32, gamma.double(), beta.double(), eps=1e-5)
    if act:
        h = F.silu(h)
    if a_mode == 1:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
    ref = F.conv2d(h, w.double(), b.double(), padding=1)
    if temb is not None:
        ref = ref + temb.double()[:, :, None, None]
    if res is not None:
        ref = ref + res.double()
    xs = hipops.nhwc(x.to(dev()))
    srcs = [xs[..., :c0].contiguous()] + ([xs[..., c0:].contiguous()] if c1 else [])
    gn = hipops.gn_affine(srcs, gamma.to(dev()), beta.to(dev())) if use_gn else None
    kw = dict(Hout=Hout, ks=3, gn=gn, act=act, a_mode=a_mode, temb=temb.to(dev()) if temb is not None else None,
              res=hipops.nhwc(res.to(dev())) if res is not None else None)
    st = []
    got = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), cfg=7, stats_out=st, **kw)
    f32 = hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), cfg=3, **kw)

    def err(t):
        return ((hipops.nchw(t).double().cpu() - ref).abs().max() / ref.abs().max()).item()
    e7, e3 = err(got), err(f32)
    assert e7 < max(2 * e3, 2e-5), (e7, e3)
    g2, b2 = 1 + 0.1 * rnd(N, seed=28), 0.1 * rnd(N, seed=29)
    sc, sh = hipops.gn_finalize(st, g2.to(dev()), b2.to(dev()), Hout * Hout)
    assert relerr(hipops.nchw(got * sc[:, None, None, :] + sh[:, None, None, :]), F.group_norm(ref.float(), 32, g2, b2, eps=1e-5)) < 2e-4
    assert torch.equal(got, hipops.conv_igemm(srcs, w.to(dev()), b.to(dev()), cfg=7, **kw))
