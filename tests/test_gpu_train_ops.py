"""-m gpu: the backward twins of the small / pointwise UNet operators (csrc/train_kernels.hip) through the C ABI, each
against torch autograd of the expression it replaces (fp64 on the CPU where cheap)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def L():
    from anoddpm_amd import _lib
    return _lib, _lib.lib()


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-12)).item()


@pytest.mark.parametrize("B,P,c0,c1,N,gn,act,span", [(2, 256, 64, 0, 96, False, 0, 64), (3, 100, 96, 32, 160, True, 0, 32),
                                                     (1, 1024, 256, 128, 128, True, 1, 256), (2, 16, 32, 0, 32, True, 0, 32)])
def test_wgrad_pointwise(B, P, c0, c1, N, gn, act, span):
    _lib, lib = L()
    torch.manual_seed(0)
    K = c0 + c1
    a0 = torch.randn(B, P, c0, device=DEV)
    a1 = torch.randn(B, P, c1, device=DEV) if c1 else None
    dy = torch.randn(B, P, N, device=DEV)
    sc = (torch.rand(B, K, device=DEV) + 0.5) if gn else None
    sh = torch.randn(B, K, device=DEV) if gn else None
    dw = torch.full((N, K), 0.5, device=DEV)
    db = torch.full((N,), -1.0, device=DEV)
    nitems = B * -(-P // span)
    ws = torch.empty(nitems * (K * N + N), device=DEV)
    st = _lib.Wgrad1Args()
    st.a0, st.a1 = a0.data_ptr(), (a1.data_ptr() if c1 else None)
    st.gn_scale, st.gn_shift = (sc.data_ptr(), sh.data_ptr()) if gn else (None, None)
    st.dy, st.dw, st.dbias, st.ws, st.ws_floats = dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel()
    st.a0_bs, st.a1_bs, st.dy_bs = P * c0, P * c1, P * N
    st.c0, st.c1, st.a0_ld, st.a1_ld, st.dy_ld = c0, c1, c0, (c1 or 4), N
    st.P, st.N, st.B, st.act, st.gn_ld, st.span, st.accumulate = P, N, B, act, K, span, 1
    _lib.check(lib.anoddpm_wgrad_pointwise(ctypes.byref(st), _lib.current_stream()))
    A = torch.cat([a0, a1], dim=2) if c1 else a0
    A = A.double().cpu()
    if gn:
        A = A * sc.double().cpu()[:, None, :] + sh.double().cpu()[:, None, :]
    if act:
        A = F.silu(A)
    ref = torch.einsum("bpn,bpk->nk", dy.double().cpu(), A) + 0.5
    assert rel(dw, ref) < 2e-5
    assert rel(db, dy.double().cpu().sum(dim=(0, 1)) - 1.0) < 2e-5


def test_pack_pointwise_and_small_conv():
    _lib, lib = L()
    from hipops import _pack_conv
    torch.manual_seed(1)
    N, K = 24, 40
    w = torch.randn(N, K, device=DEV)
    out = torch.empty(N * K, device=DEV)
    st = _lib.PackArgs()
    st.w, st.out, st.N, st.K, st.kind, st.bwd, st.k0, st.kc = w.data_ptr(), out.data_ptr(), N, K, 2, 0, 0, 0
    _lib.check(lib.anoddpm_pack_weights(ctypes.byref(st), _lib.current_stream()))
    assert torch.equal(out.view(1, K // 4, N, 4), _pack_conv(w.view(N, K, 1, 1)))
    # data-gradient matrix of the column range [8, 8 + 16): W'[i = n][o] = w[n][8 + o]
    out2 = torch.empty(N * 16, device=DEV)
    st.out, st.bwd, st.k0, st.kc = out2.data_ptr(), 1, 8, 16
    _lib.check(lib.anoddpm_pack_weights(ctypes.byref(st), _lib.current_stream()))
    wt = w[:, 8:24].t().contiguous()                      # [O = 16][I = N]
    assert torch.equal(out2.view(1, N // 4, 16, 4), _pack_conv(wt.view(16, N, 1, 1)))
    w3 = torch.randn(6, 5, 3, 3, device=DEV)
    out3 = torch.empty(9 * 5 * 6, device=DEV)
    st.w, st.out, st.N, st.K, st.kind, st.bwd = w3.data_ptr(), out3.data_ptr(), 6, 5, 3, 0
    _lib.check(lib.anoddpm_pack_weights(ctypes.byref(st), _lib.current_stream()))
    assert torch.equal(out3.view(9, 5, 6), w3.permute(2, 3, 1, 0).reshape(9, 5, 6))


@pytest.mark.parametrize("Z,Lq", [(3, 64), (2, 16), (1, 100)])
def test_softmax_backward_and_transpose(Z, Lq):
    _lib, lib = L()
    torch.manual_seed(2)
    s = torch.randn(Z, Lq, Lq, device=DEV, dtype=torch.float64, requires_grad=True)
    p = torch.softmax(s, dim=-1)
    dp = torch.randn(Z, Lq, Lq, device=DEV, dtype=torch.float64)
    p.backward(dp)
    p32, d32 = p.detach().float().contiguous(), dp.float().contiguous()
    st = _lib.SoftmaxBwdArgs()
    st.p, st.dp, st.rows, st.L = p32.data_ptr(), d32.data_ptr(), Z * Lq, Lq
    _lib.check(lib.anoddpm_softmax_rows_backward(ctypes.byref(st), _lib.current_stream()))
    assert rel(d32, s.grad) < 1e-5
    out = torch.empty_like(p32)
    tr = _lib.TransposeArgs()
    tr.inp, tr.out, tr.Z, tr.L = p32.data_ptr(), out.data_ptr(), Z, Lq
    _lib.check(lib.anoddpm_transpose_square(ctypes.byref(tr), _lib.current_stream()))
    assert torch.equal(out, p32.transpose(1, 2).contiguous())


@pytest.mark.parametrize("B,K,N,act_in", [(4, 128, 512, 0), (2, 512, 96, 1), (16, 64, 33, 1)])
def test_linear_small_backward(B, K, N, act_in):
    _lib, lib = L()
    torch.manual_seed(3)
    x = torch.randn(B, K, device=DEV, requires_grad=True)
    w = torch.randn(N, K, device=DEV, requires_grad=True)
    b = torch.randn(N, device=DEV, requires_grad=True)
    dy = torch.randn(B, N, device=DEV)
    y = F.linear(F.silu(x) if act_in else x, w, b)
    y.backward(dy)
    dw = torch.full((N, K), 2.0, device=DEV)
    db = torch.full((N,), 3.0, device=DEV)
    dx = torch.full((B, K), -1.0, device=DEV)
    st = _lib.LinearBwdArgs()
    st.x, st.w, st.dy, st.dw, st.db, st.dx = x.data_ptr(), w.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), dx.data_ptr()
    st.B, st.K, st.N, st.act_in, st.acc_w, st.acc_x = B, K, N, act_in, 1, 1
    _lib.check(lib.anoddpm_linear_small_backward(ctypes.byref(st), _lib.current_stream()))
    assert rel(dw, w.grad + 2.0) < 1e-5 and rel(db, b.grad + 3.0) < 1e-5 and rel(dx, x.grad - 1.0) < 1e-5
    st.acc_w, st.acc_x = 0, 0
    _lib.check(lib.anoddpm_linear_small_backward(ctypes.byref(st), _lib.current_stream()))
    assert rel(dw, w.grad) < 1e-5 and rel(db, b.grad) < 1e-5 and rel(dx, x.grad) < 1e-5


@pytest.mark.parametrize("B,S,Cin,Cout", [(2, 32, 1, 128), (1, 40, 3, 64)])
def test_conv_stem_backward(B, S, Cin, Cout):
    _lib, lib = L()
    torch.manual_seed(4)
    x = torch.randn(B, Cin, S, S, device=DEV, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.2).requires_grad_(True)
    b = torch.randn(Cout, device=DEV, requires_grad=True)
    dy_nchw = torch.randn(B, Cout, S, S, device=DEV)
    F.conv2d(x, w, b, padding=1).backward(dy_nchw)
    dy = dy_nchw.permute(0, 2, 3, 1).contiguous()
    dw, db, dx = torch.zeros_like(w), torch.zeros_like(b), torch.empty_like(x)
    nblk = B * -(-(S * S) // 1024)
    ws = torch.empty(nblk * (Cin * 9 + 1) * Cout, device=DEV)
    st = _lib.StemBwdArgs()
    st.x, st.w, st.dy, st.dw, st.db, st.dx = x.data_ptr(), w.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), dx.data_ptr()
    st.ws, st.ws_floats, st.B, st.H, st.W, st.Cin, st.Cout = ws.data_ptr(), ws.numel(), B, S, S, Cin, Cout
    _lib.check(lib.anoddpm_conv_stem_backward(ctypes.byref(st), _lib.current_stream()))
    assert rel(dw, w.grad) < 2e-5 and rel(db, b.grad) < 2e-5 and rel(dx, x.grad) < 2e-5


@pytest.mark.parametrize("B,S,C,Cout", [(2, 32, 128, 1), (1, 24, 64, 3)])
def test_conv_head_backward(B, S, C, Cout):
    _lib, lib = L()
    torch.manual_seed(5)
    x = torch.randn(B, S * S, C, device=DEV)
    sc, sh = torch.rand(B, C, device=DEV) + 0.5, torch.randn(B, C, device=DEV)
    a = F.silu(x * sc[:, None] + sh[:, None]).view(B, S, S, C).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    w = (torch.randn(Cout, C, 3, 3, device=DEV) * 0.1).requires_grad_(True)
    b = torch.randn(Cout, device=DEV, requires_grad=True)
    dy = torch.randn(B, Cout, S, S, device=DEV)
    F.conv2d(a, w, b, padding=1).backward(dy)
    da = torch.empty(B, S * S, C, device=DEV)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    nblk = B * -(-(S * S) // 512)
    ws = torch.empty(nblk * 10 * Cout * C, device=DEV)
    st = _lib.HeadBwdArgs()
    st.x, st.gn_scale, st.gn_shift, st.w, st.dy = x.data_ptr(), sc.data_ptr(), sh.data_ptr(), w.data_ptr(), dy.data_ptr()
    st.da, st.dw, st.db, st.ws, st.ws_floats = da.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel()
    st.B, st.H, st.W, st.C, st.Cout = B, S, S, C, Cout
    _lib.check(lib.anoddpm_conv_head_backward(ctypes.byref(st), _lib.current_stream()))
    assert rel(da, a.grad.permute(0, 2, 3, 1).reshape(B, S * S, C)) < 2e-5
    assert rel(dw, w.grad) < 5e-5 and rel(db, b.grad) < 2e-5


def test_resample_backward_modes_and_colsum_fold():
    _lib, lib = L()
    torch.manual_seed(6)
    B, H, C = 2, 8, 16
    x = torch.randn(B, C, H, H, device=DEV, requires_grad=True)
    up = F.interpolate(x, scale_factor=2, mode="nearest")
    g_up = torch.randn_like(up)
    up.backward(g_up)
    g_nhwc = g_up.permute(0, 2, 3, 1).contiguous()
    out = torch.full((B, H, H, C), 1.0, device=DEV)
    st = _lib.ResampleArgs()
    st.inp, st.out, st.B, st.H, st.W, st.C, st.mode, st.scale, st.accumulate = g_nhwc.data_ptr(), out.data_ptr(), B, 2 * H, 2 * H, C, 2, 4.0, 1
    _lib.check(lib.anoddpm_resample2x(ctypes.byref(st), _lib.current_stream()))
    assert rel(out, x.grad.permute(0, 2, 3, 1) + 1.0) < 1e-6
    x.grad = None
    dn = F.avg_pool2d(x, 2, 2)
    g_dn = torch.randn_like(dn)
    dn.backward(g_dn)
    g2 = g_dn.permute(0, 2, 3, 1).contiguous()
    out2 = torch.empty((B, H, H, C), device=DEV)
    st.inp, st.out, st.H, st.W, st.mode, st.scale, st.accumulate = g2.data_ptr(), out2.data_ptr(), H // 2, H // 2, 1, 0.25, 0
    _lib.check(lib.anoddpm_resample2x(ctypes.byref(st), _lib.current_stream()))
    assert rel(out2, x.grad.permute(0, 2, 3, 1)) < 1e-6
    cs = torch.randn(3, 7, 40, device=DEV)
    dimg, dbias = torch.empty(3, 40, device=DEV), torch.ones(40, device=DEV)
    cf = _lib.ColsumFoldArgs()
    cf.colsum, cf.dimg, cf.dbias, cf.B, cf.ipb, cf.N = cs.data_ptr(), dimg.data_ptr(), dbias.data_ptr(), 3, 7, 40
    _lib.check(lib.anoddpm_colsum_fold(ctypes.byref(cf), _lib.current_stream()))
    assert rel(dimg, cs.sum(dim=1)) < 1e-6 and rel(dbias, cs.sum(dim=(0, 1)) + 1.0) < 1e-6
