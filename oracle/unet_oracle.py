"""oracle/unet_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional stock-PyTorch (CPU, fp32) restatement of the reference denoiser `UNetModel.forward`
(UNet.py:220-406) over a plain state-dict.  It is the checker for the HIP UNet and the
"port" CPU baseline timed by bench.py; nothing under anoddpm_amd/ imports it.

Parity status: PINNED.  tests/test_oracle_unet.py compares it with tests/golden/unet_*.npz,
produced by running the imported reference `UNetModel` on the same deterministic weights
(tests/golden/make_golden.py); tolerance there is 1e-5 abs on O(1) activations (same ATen
kernels in a slightly different call order).

Restated pieces (reference file:line):
  layout()                 UNet.py:239-254, 278-388  (constructor: which blocks exist)
  timestep_features()      UNet.py:50-57             (PositionalEmbedding)
  res_block()              UNet.py:202-217 with :169-200 (ResBlock, BigGAN up/down variant)
  resample_layer()         UNet.py:60-92             (Downsample / Upsample of biggan_updown=False)
  attention_block()        UNet.py:119-125, 137-153  (AttentionBlock + legacy QKVAttention)
  forward()                UNet.py:390-406
  fill_deterministic()     SURVEY.md 8c/8d recipe: RandomState(crc32(key)) parameter fill
"""
import math
import zlib

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_MULTS = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4),
                 64: (1, 2, 3, 4), 32: (1, 2, 3, 4)}


def layout(img_size, base_channels, channel_mults="", num_res_blocks=2,
           attention_resolutions="32,16,8", in_channels=1, biggan_updown=True, conv_resample=True):
    """Block list of the network as (prefix, kind, cin, cout, resample) tuples.

    kind: 'stem' | 'res' | 'attn' | 'downsample' | 'upsample' (the last two: biggan_updown=False, UNet.py:318-320,
    377-379); resample: None | 'down' | 'up' for ResBlocks, 'conv' | None for Downsample / Upsample (conv_resample).
    Returns dict(down=[[...],...], middle=[...], up=[[...],...], out_ch=int).
    """
    if channel_mults == "":
        if img_size not in DEFAULT_MULTS:
            raise ValueError(f"unsupported image size: {img_size}")
        channel_mults = DEFAULT_MULTS[img_size]
    attn_ds = [img_size // int(r) for r in attention_resolutions.split(",")]
    ch = int(channel_mults[0] * base_channels)
    down = [[("down.0.0", "stem", in_channels, base_channels, None)]]
    skip_ch = [ch]
    ds = 1
    for level, mult in enumerate(channel_mults):
        for _ in range(num_res_blocks):
            n = len(down)
            cout = int(base_channels * mult)
            blk = [(f"down.{n}.0", "res", ch, cout, None)]
            ch = cout
            if ds in attn_ds:
                blk.append((f"down.{n}.1", "attn", ch, ch, None))
            down.append(blk)
            skip_ch.append(ch)
        if level != len(channel_mults) - 1:
            n = len(down)
            if biggan_updown:
                down.append([(f"down.{n}.0", "res", ch, ch, "down")])
            else:
                down.append([(f"down.{n}.0", "downsample", ch, ch, "conv" if conv_resample else None)])
            ds *= 2
            skip_ch.append(ch)
    middle = [("middle.0", "res", ch, ch, None), ("middle.1", "attn", ch, ch, None),
              ("middle.2", "res", ch, ch, None)]
    up = []
    for level, mult in reversed(list(enumerate(channel_mults))):
        for j in range(num_res_blocks + 1):
            n = len(up)
            cin = ch + skip_ch.pop()
            cout = int(base_channels * mult)
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
    return dict(down=down, middle=middle, up=up, out_ch=ch,
                final_cin=int(base_channels * channel_mults[0]))


def param_shapes(img_size, base_channels, channel_mults="", num_res_blocks=2,
                 attention_resolutions="32,16,8", in_channels=1, biggan_updown=True, conv_resample=True):
    """Ordered {key: shape} of every parameter of the reference module for this config."""
    lay = layout(img_size, base_channels, channel_mults, num_res_blocks, attention_resolutions,
                 in_channels, biggan_updown, conv_resample)
    ted = base_channels * 4
    shapes = {}

    def lin(p, o, i):
        shapes[p + ".weight"] = (o, i)
        shapes[p + ".bias"] = (o,)

    def conv(p, o, i, k):
        shapes[p + ".weight"] = (o, i, k, k)
        shapes[p + ".bias"] = (o,)

    def gn(p, c):
        shapes[p + ".weight"] = (c,)
        shapes[p + ".bias"] = (c,)

    def block(b):
        p, kind, cin, cout, _ = b
        if kind == "stem":
            conv(p, cout, cin, 3)
        elif kind == "res":
            gn(p + ".in_layers.0", cin)
            conv(p + ".in_layers.2", cout, cin, 3)
            lin(p + ".embed_layers.1", cout, ted)
            gn(p + ".out_layers.0", cout)
            conv(p + ".out_layers.3", cout, cout, 3)
            if cin != cout:
                conv(p + ".skip_connection", cout, cin, 1)
        elif kind == "downsample":
            if b[4] == "conv":
                conv(p + ".downsample", cout, cin, 3)
        elif kind == "upsample":
            if b[4] == "conv":
                conv(p + ".conv", cout, cin, 3)
        else:
            gn(p + ".norm", cin)
            shapes[p + ".to_qkv.weight"] = (3 * cin, cin, 1)
            shapes[p + ".to_qkv.bias"] = (3 * cin,)
            shapes[p + ".proj_out.weight"] = (cin, cin, 1)
            shapes[p + ".proj_out.bias"] = (cin,)

    lin("time_embedding.1", ted, base_channels)
    lin("time_embedding.3", ted, ted)
    for blk in lay["down"]:
        for b in blk:
            block(b)
    for b in lay["middle"]:
        block(b)
    for blk in lay["up"]:
        for b in blk:
            block(b)
    gn("out.0", lay["out_ch"])
    conv("out.2", in_channels, lay["final_cin"], 3)
    return shapes


def fill_deterministic(shapes, sigma=0.02):
    """SURVEY 8c/8d fill: per-key legacy RandomState(crc32(key)); conv/linear weights N(0,sigma)
    (including the reference's zero-initialised ones), biases small non-zero, GN affine near 1/0.
    Reproducible anywhere without the reference."""
    sd = {}
    for key, shape in shapes.items():
        rs = np.random.RandomState(zlib.crc32(key.encode()) & 0xFFFFFFFF)
        v = rs.standard_normal(shape).astype(np.float32)
        is_norm = (".in_layers.0." in key or ".out_layers.0." in key or ".norm." in key
                   or key.startswith("out.0."))
        if is_norm:
            v = (1.0 + 0.1 * v) if key.endswith("weight") else 0.1 * v
        elif key.endswith("bias"):
            v = 0.05 * v
        else:
            fan_in = int(np.prod(shape[1:]))
            v = v * max(sigma, 1.0 / math.sqrt(fan_in))
        sd[key] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def timestep_features(t, dim):
    half = dim // 2
    step = np.log(10000) / half
    freqs = torch.exp(torch.arange(half) * -step).to(t.device)     # formed on the host as UNet.py:53-54 does on a CPU run
    arg = torch.outer(t * 1, freqs)
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def _gn(sd, p, x):
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def res_block(sd, p, x, temb, resample, record=None, dropout=None):
    h = F.silu(_gn(sd, p + ".in_layers.0", x))
    if resample == "down":
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    elif resample == "up":
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(temb), sd[p + ".embed_layers.1.weight"], sd[p + ".embed_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = F.silu(_gn(sd, p + ".out_layers.0", h))
    if dropout is not None:                      # nn.Dropout(p) of UNet.py:192 with an INJECTED keep mask: dropout(p, prefix, h) -> h * mask / (1 - p)
        h = dropout(p, h)
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def attention_block(sd, p, x, n_heads, n_head_channels):
    b, c, hh, ww = x.shape
    heads = n_heads if n_head_channels == -1 else c // n_head_channels
    xf = x.reshape(b, c, hh * ww)
    qkv = F.conv1d(_gn(sd, p + ".norm", xf), sd[p + ".to_qkv.weight"], sd[p + ".to_qkv.bias"])
    ch = c // heads
    q, k, v = qkv.reshape(b * heads, 3 * ch, hh * ww).split(ch, dim=1)   # legacy per-head q|k|v
    s = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s).float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, c, hh * ww)
    a = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + a).reshape(b, c, hh, ww)


def resample_layer(sd, p, kind, conv, x):
    """Downsample / Upsample (UNet.py:60-92): stride-2 3x3 conv or 2x2 average pool; nearest x2 (+ 3x3 conv)."""
    if kind == "downsample":
        if conv:
            return F.conv2d(x, sd[p + ".downsample.weight"], sd[p + ".downsample.bias"], stride=2, padding=1)
        return F.avg_pool2d(x, 2, 2)
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    if conv:
        x = F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
    return x


def forward_autograd(sd, x, t, img_size, base_channels, channel_mults="", num_res_blocks=2,
                     attention_resolutions="32,16,8", in_channels=1, n_heads=1, n_head_channels=-1,
                     record=None, biggan_updown=True, conv_resample=True, dropout=None):
    """Returns the model output; if `record` is a dict it receives per-block activations
    keyed by block prefix (NCHW fp32) for layer-wise parity checks.  Autograd is left on: with
    `requires_grad` leaves in `sd` this is the CPU checker for the training gradients
    (diffusion_training.py:102, pinned by tests/golden/train_*.npz)."""
    lay = layout(img_size, base_channels, channel_mults, num_res_blocks, attention_resolutions,
                 in_channels, biggan_updown, conv_resample)
    temb = timestep_features(t, base_channels)
    temb = F.linear(temb, sd["time_embedding.1.weight"], sd["time_embedding.1.bias"])
    temb = F.linear(F.silu(temb), sd["time_embedding.3.weight"], sd["time_embedding.3.bias"])
    if record is not None:
        record["time_embed"] = temb.detach().clone()

    def run(blocks, h):
        for (p, kind, cin, cout, resample) in blocks:
            if kind == "stem":
                h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
            elif kind == "res":
                h = res_block(sd, p, h, temb, resample, dropout=dropout)
            elif kind in ("downsample", "upsample"):
                h = resample_layer(sd, p, kind, resample == "conv", h)
            else:
                h = attention_block(sd, p, h, n_heads, n_head_channels)
            if record is not None:
                record[p] = h.detach().clone()
        return h

    h = x.float()
    skips = []
    for blk in lay["down"]:
        h = run(blk, h)
        skips.append(h)
    h = run(lay["middle"], h)
    for blk in lay["up"]:
        h = run(blk, torch.cat([h, skips.pop()], dim=1))
    h = F.silu(_gn(sd, "out.0", h))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


@torch.no_grad()
def forward(sd, x, t, *a, **k):
    """Inference form of `forward_autograd` (no autograd graph)."""
    return forward_autograd(sd, x, t, *a, **k)


def perturb(sd, scale=0.02, salt="train"):
    """Deterministic perturbation of a filled state-dict (RandomState(crc32(key + salt))): moves every tensor off
    special values so that each parameter receives a non-trivial gradient in the training fixtures."""
    out = {}
    for key, v in sd.items():
        rs = np.random.RandomState(zlib.crc32((key + salt).encode()) & 0xFFFFFFFF)
        out[key] = v + torch.from_numpy((scale * rs.standard_normal(tuple(v.shape))).astype(np.float32))
    return out


def flops_per_image(img_size, base_channels, channel_mults="", num_res_blocks=2,
                    attention_resolutions="32,16,8", in_channels=1, n_heads=1, n_head_channels=-1,
                    biggan_updown=True, conv_resample=True):
    """Algorithmic FLOPs of one forward for one image by the SURVEY 8d counting rule:
    2*Cin*Cout*k^2*Hout*Wout per conv, 2*in*out per linear, 4*C*L^2 per attention."""
    lay = layout(img_size, base_channels, channel_mults, num_res_blocks, attention_resolutions,
                 in_channels, biggan_updown, conv_resample)
    tot = dict(conv3=0.0, conv1=0.0, qkvproj=0.0, attn=0.0, linear=0.0)
    ted = 4 * base_channels
    tot["linear"] += 2 * base_channels * ted + 2 * ted * ted
    res = img_size

    def block(b):
        nonlocal res
        p, kind, cin, cout, rs = b
        if kind == "stem":
            tot["conv3"] += 2 * cin * cout * 9 * res * res
        elif kind == "res":
            if rs == "down":
                res //= 2
            elif rs == "up":
                res *= 2
            tot["conv3"] += 2 * cin * cout * 9 * res * res + 2 * cout * cout * 9 * res * res
            tot["linear"] += 2 * ted * cout
            if cin != cout:
                tot["conv1"] += 2 * cin * cout * res * res
        elif kind in ("downsample", "upsample"):
            res = res // 2 if kind == "downsample" else res * 2
            if rs == "conv":
                tot["conv3"] += 2 * cin * cout * 9 * res * res
        else:
            L = res * res
            tot["qkvproj"] += 2 * cin * 3 * cin * L + 2 * cin * cin * L
            tot["attn"] += 4 * cin * L * L
    for blk in lay["down"]:
        for b in blk:
            block(b)
    for b in lay["middle"]:
        block(b)
    for blk in lay["up"]:
        for b in blk:
            block(b)
    tot["conv3"] += 2 * lay["final_cin"] * in_channels * 9 * res * res
    tot["total"] = sum(tot.values())
    return tot
