"""Pins oracle/unet_oracle.py against outputs of the reference UNetModel (UNet.py:220-406) on
the deterministic parameter fill.  CPU only; the 256^2 case runs one 1.8 s forward."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as uo

from conftest import GOLDEN

CASES = {
    "i32_b32_h1": dict(img_size=32, base_channels=32),
    "i32_b32_h2_a16_8": dict(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8"),
    "i64_b32_hc32": dict(img_size=64, base_channels=32, n_head_channels=32, attention_resolutions="16,8"),
    "i64_b64_c3": dict(img_size=64, base_channels=64, n_heads=2, in_channels=3),
    "i128_b32_h2": dict(img_size=128, base_channels=32, n_heads=2, attention_resolutions="16,8"),
    "c5like_i128_b32": dict(img_size=128, base_channels=32, n_heads=2, channel_mults=(1, 1, 2, 2, 4, 4),
                            attention_resolutions="32,16,8"),
    "c2_256_b128": dict(img_size=256, base_channels=128, n_heads=2, attention_resolutions="16,8"),
    # biggan_updown=False: Downsample / Upsample layers (UNet.py:60-92), with and without their convolutions
    "convrs_i64_b32": dict(img_size=64, base_channels=32, n_heads=2, attention_resolutions="16,8", biggan_updown=False,
                           conv_resample=True),
    "poolrs_i32_b32": dict(img_size=32, base_channels=32, biggan_updown=False, conv_resample=False),
}


def shapes_of(kw):
    return uo.param_shapes(kw["img_size"], kw["base_channels"], kw.get("channel_mults", ""), 2,
                           kw.get("attention_resolutions", "32,16,8"), kw.get("in_channels", 1), kw.get("biggan_updown", True),
                           kw.get("conv_resample", True))


@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_reference(name):
    kw = CASES[name]
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    shapes = shapes_of(kw)
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(g["n_params"])
    assert len(shapes) == int(g["n_tensors"])
    if "keys" in g.files:
        assert list(shapes.keys()) == g["keys"].tolist()
        assert [",".join(map(str, s)) for s in shapes.values()] == g["key_shapes"].tolist()
    sd = uo.fill_deterministic(shapes)
    rec = {}
    y = uo.forward(sd, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), record=rec, **kw)
    assert y.shape == g["y"].shape
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=2e-5)
    # per-block probes recorded from the reference's forward hooks
    for k in g.files:
        if k.startswith("probe/"):
            key = k[len("probe/"):]
            ok = {"time_embed": "time_embed"}.get(key, key)
            v = rec[ok].flatten()
            stride = max(1, v.numel() // 256)
            np.testing.assert_allclose(v[::stride][:256].numpy(), g[k], rtol=0, atol=2e-5, err_msg=key)


def test_config2_batch4_matches_reference():
    """The oracle on BASELINE config 2 at its benchmarked batch (4 images, four timesteps) against the reference model's output."""
    kw = CASES["c2_256_b128"]
    g = np.load(os.path.join(GOLDEN, "unet_c2_256_b128_batch4.npz"))
    sd = uo.fill_deterministic(shapes_of(kw))
    y = uo.forward(sd, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), **kw)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=2e-5)


def test_c2_param_count_and_flops():
    kw = CASES["c2_256_b128"]
    shapes = shapes_of(kw)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 130331393     # SURVEY: 130.3 M
    assert len(shapes) == 536                                            # SURVEY: 536 tensors
    fl = uo.flops_per_image(**kw)
    assert abs(fl["total"] / 1e9 + 1.6 - 556.1) < 0.5                     # 556.1 GFLOP incl. 1.6 GF of GroupNorm
    assert abs(fl["conv3"] / 1e9 - 528.5) < 1.0


def test_zero_init_convention_not_used():
    # the reference zero-initialises out-convs; the deterministic fill must not, or parity is vacuous
    sd = uo.fill_deterministic(shapes_of(CASES["i32_b32_h1"]))
    assert sd["out.2.weight"].abs().sum() > 0 and sd["down.1.0.out_layers.3.weight"].abs().sum() > 0


def test_unsupported_size():
    with pytest.raises(ValueError):
        uo.layout(48, 32)
