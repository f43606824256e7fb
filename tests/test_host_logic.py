"""Host-side behaviour of the drop-in modules that needs no GPU: schedule tables, state-dict layout,
RNG consumption, error behaviour, config helpers, refusal of CPU tensors.  CPU only."""
import copy
import json
import os
import sys

import numpy as np
import pytest
import torch

import GaussianDiffusion as GD
import UNet as UN
import helpers as HP
import simplex as SX
from anoddpm_amd._lib import AnoddpmError

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def dkat():
    return np.load(os.path.join(GOLDEN, "diffusion_kat.npz"))


@pytest.mark.parametrize("name", ["linear", "cosine"])
def test_schedule_tables_bit_exact(dkat, name):
    betas = GD.get_beta_schedule(1000, name)
    assert (betas.view(np.uint64) == dkat[f"{name}_betas"].view(np.uint64)).all()
    d = GD.GaussianDiffusionModel([16, 16], betas)
    for k in ("sqrt_alphas", "sqrt_betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
              "posterior_mean_coef1", "posterior_mean_coef2"):
        assert (getattr(d, k).view(np.uint64) == dkat[f"{name}_{k}"].view(np.uint64)).all(), k
    assert d.num_timesteps == 1000 and d.img_size == [16, 16] and d.img_channels == 1


def test_schedule_errors_and_attrs():
    with pytest.raises(NotImplementedError):
        GD.get_beta_schedule(10, "sigmoid")
    d = GD.GaussianDiffusionModel([8, 8], GD.get_beta_schedule(50, "linear"), loss_weight="prop-t", noise="simplex")
    assert d.weights[0] == 50 and d.weights[-1] == 1 and hasattr(d, "simplex") and callable(d.noise_fn)
    d2 = GD.GaussianDiffusionModel([8, 8], GD.get_beta_schedule(50, "linear"), noise="anything-else")
    assert isinstance(d2.simplex, SX.Simplex_CLASS)          # unknown strings silently mean simplex
    with pytest.raises(AssertionError):
        d.forward_backward(None, torch.zeros(1, 1, 8, 8), see_whole_sequence="quarter")
    x = torch.zeros(1, 1, 8, 8)
    assert d.forward_backward(None, x, t_distance=0) is not None        # t_distance == 0 -> x.detach()


def test_cpu_tensors_are_refused_not_emulated():
    d = GD.GaussianDiffusionModel([8, 8], GD.get_beta_schedule(50, "linear"), noise="simplex")
    x = torch.zeros(2, 1, 8, 8)
    t = torch.tensor([1, 2])
    with pytest.raises(AnoddpmError):
        d.sample_q(x, t, x)
    with pytest.raises(AnoddpmError):
        d.sample_p(lambda a, b: a, x, t)
    with pytest.raises(AnoddpmError):
        d.noise_fn(x, t)
    m = UN.UNetModel(32, 32)
    with pytest.raises(AnoddpmError):
        with torch.no_grad():
            m(torch.zeros(1, 1, 32, 32), torch.tensor([3]))
    with pytest.raises(AnoddpmError):
        m(torch.zeros(1, 1, 32, 32), torch.tensor([3]))      # autograd path is device-only as well


@pytest.mark.parametrize("name", ["i32_b32_h1", "i64_b32_hc32", "i64_b64_c3", "convrs_i64_b32", "poolrs_i32_b32"])
def test_state_dict_layout_matches_reference(name):
    cases = {"i32_b32_h1": dict(img_size=32, base_channels=32),
             "i64_b32_hc32": dict(img_size=64, base_channels=32, n_head_channels=32, attention_resolutions="16,8"),
             "i64_b64_c3": dict(img_size=64, base_channels=64, n_heads=2, in_channels=3),
             # biggan_updown=False: Downsample / Upsample layers with (`.downsample.*`, `.conv.*`) and without parameters
             "convrs_i64_b32": dict(img_size=64, base_channels=32, n_heads=2, attention_resolutions="16,8", biggan_updown=False,
                                    conv_resample=True),
             "poolrs_i32_b32": dict(img_size=32, base_channels=32, biggan_updown=False, conv_resample=False)}
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    m = UN.UNetModel(**cases[name])
    sd = m.state_dict()
    assert list(sd.keys()) == g["keys"].tolist()
    assert [",".join(map(str, v.shape)) for v in sd.values()] == g["key_shapes"].tolist()
    assert sum(v.numel() for v in sd.values()) == int(g["n_params"])
    # zero-initialised modules (UNet.py:117,193,387,414-420)
    assert sd["out.2.weight"].abs().sum() == 0 and sd["down.1.0.out_layers.3.weight"].abs().sum() == 0
    assert sd["middle.1.proj_out.weight"].abs().sum() == 0 and sd["down.1.0.in_layers.2.weight"].abs().sum() > 0


def test_fused_attention_shape_policy(monkeypatch):
    """unet.fused_attention_ok: the shapes anoddpm_attention takes (score rows resident in LDS, power-of-two head width)."""
    from anoddpm_amd.unet import fused_attention_ok as ok
    assert ok(256, 256) and ok(64, 256) and ok(1024, 128) and ok(16, 16) and ok(64, 512)
    assert not ok(4096, 128) and not ok(40, 32) and not ok(64, 48) and not ok(64, 8) and not ok(64, 1024)
    monkeypatch.setenv("ANODDPM_NO_FUSED_ATTENTION", "1")
    assert not ok(256, 256)


def test_module_protocol():
    m = UN.UNetModel(32, 32, n_heads=2)
    e = copy.deepcopy(m)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), e.state_dict().values()))
    with torch.no_grad():
        for p in m.parameters():
            p.add_(1.0)
    UN.update_ema_params(e, m, 0.5)
    k = "time_embedding.1.bias"
    assert torch.allclose(e.state_dict()[k], m.state_dict()[k] - 0.5)
    e.load_state_dict(m.state_dict())
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    assert len(opt.param_groups[0]["params"]) == len(list(m.parameters()))
    with pytest.raises(ValueError):
        UN.UNetModel(48, 32)
    with pytest.raises(AssertionError):
        UN.UNetModel(32, 32, n_head_channels=7)
    m.eval(); m.train()


def test_newseed_consumes_numpy_stream_like_reference():
    from oracle.simplex_oracle import OracleSimplex
    np.random.seed(1234)
    a = SX.Simplex_CLASS()
    a.newSeed()
    st = np.random.get_state()[1][:4].copy()
    np.random.seed(1234)
    b = OracleSimplex(None)
    b.newSeed()
    assert (np.random.get_state()[1][:4] == st).all()
    assert (a._perm == b._perm).all() and (a._perm_grad_index3 == b._perm_grad_index3).all()
    a.newSeed(3)
    assert a._perm[:6].tolist() == [164, 187, 231, 144, 104, 73] and a._perm.dtype == np.int64
    with pytest.raises(AssertionError):
        a.rand_3d_octaves((4, 4), 1)
    with pytest.raises(AssertionError):
        a.rand_3d_fixed_T_octaves((4, 4, 4), np.array([1]))
    with pytest.raises(AssertionError):
        a.rand_2d_octaves((4, 4, 4))
    with pytest.raises(ValueError):                      # upstream's (W,H) + (H,W) broadcast error, simplex.py:69
        a.rand_2d_octaves((4, 6))


def test_helpers_surface(tmp_path):
    dd = HP.defaultdict_from_json({"a": 1})
    assert dd["a"] == 1 and dd["missing"] == ""
    assert GD.torch is torch and GD.os is os and GD.json is json          # leaked star-imports
    img = torch.linspace(-1, 1, 2 * 1 * 4 * 4).reshape(2, 1, 4, 4)
    grid = HP.gridify_output(img, 2)
    assert grid.dtype == torch.uint8 and grid.shape[-1] == 3
    for n in (26, 28, 1001, 1002, 1003, 1005):
        args = HP.defaultdict_from_json(json.load(open(os.path.join(ROOT, "test_args", f"args{n}.json"))))
        assert args["T"] == 1000 and args["channels"] == "" and args["noise_fn"] in ("gauss", "simplex")
    # helpers.py:26-93: checkpoint discovery + argv resolution, the checkpoint dict of diffusion_training.py:169-189 untouched
    cwd, argv = os.getcwd(), sys.argv
    try:
        os.chdir(tmp_path)
        os.makedirs("model/diff-params-ARGS=28/checkpoint")
        ck = {"n_epoch": 3, "model_state_dict": {"w": torch.ones(2)}, "optimizer_state_dict": {}, "ema": {"w": torch.zeros(2)},
              "args": HP.defaultdict_from_json({"arg_num": "28", "T": 1000})}
        torch.save(ck, "model/diff-params-ARGS=28/params-final.pt")
        torch.save(dict(ck, n_epoch=1000), "model/diff-params-ARGS=28/checkpoint/diff_epoch=1000.pt")
        torch.save(dict(ck, n_epoch=2000), "model/diff-params-ARGS=28/checkpoint/diff_epoch=2000.pt")
        open("model/diff-params-ARGS=28/checkpoint/diff_epoch=3000.pt", "wb").write(b"PK\x03\x04 truncated")   # corrupt: skipped
        assert HP.load_checkpoint("28", False, "cpu")["n_epoch"] == 3
        assert HP.load_checkpoint("28", True, "cpu")["n_epoch"] == 2000             # newest readable one
        for spec in (["28"], ["args28"], ["args28.json"]):
            sys.argv = ["detection.py"] + spec
            args, out = HP.load_parameters("cpu")
            assert out["n_epoch"] == 3 and args["noise_fn"] == "gauss" and args["missing"] == ""
        sys.argv = ["detection.py", "CHECKPOINT", "28"]
        assert HP.load_parameters("cpu")[1]["n_epoch"] == 2000
        sys.argv = ["detection.py"]                                             # no argv: the entries of ./model
        with pytest.raises(ValueError):
            HP.load_parameters("cpu")                                           # "diff-params-ARGS=28" is not a valid spec upstream either
        sys.argv = ["detection.py", "bogus"]
        with pytest.raises(ValueError):
            HP.load_parameters("cpu")
        # round-5 advisor finding: the safe (weights-only) unpickler is what reads checkpoints; a pickle that names other
        # globals is refused unless the user opts in for their own files
        import pickle

        class Foreign:
            pass
        import __main__
        __main__.Foreign = Foreign
        Foreign.__module__, Foreign.__qualname__ = "__main__", "Foreign"
        torch.save(dict(ck, extra=Foreign()), "model/diff-params-ARGS=28/params-final.pt")
        with pytest.raises(pickle.UnpicklingError):
            HP.load_checkpoint("28", False, "cpu")
        os.environ["ANODDPM_UNSAFE_CHECKPOINTS"] = "1"
        try:
            assert HP.load_checkpoint("28", False, "cpu")["n_epoch"] == 3
        finally:
            del os.environ["ANODDPM_UNSAFE_CHECKPOINTS"]
            del __main__.Foreign
    finally:
        os.chdir(cwd)
        sys.argv = argv


def test_conv_launch_policy_on_config2_shapes():
    """unet.choose_conv_cfg (shared by the inference plan and the training operators): the configurations the round-1
    measurements were taken with (profiles/r1e_igemm_by_layer.csv)."""
    from anoddpm_amd.unet import choose_conv_cfg as pick
    Z = 4
    assert pick(256, 256, 128, 128, Z) == (2, 1)                    # Winograd, one K slice
    assert pick(64, 64, 256, 256, Z) == (2, 1)
    assert pick(32, 32, 256, 256, Z) == (2, 4)                      # small map: Winograd split-K
    assert pick(16, 16, 512, 512, Z) == (2, 8)
    assert pick(8, 8, 512, 512, Z) == (1, 16)                       # H % 16 != 0: direct kernel, split-K
    assert pick(128, 128, 128, 128, Z, a_mode=2) == (0, 1)          # pool-fused operand: direct 128x128 tiles
    assert pick(256, 256, 256, 128, Z, ks=1) == (0, 1)              # 1x1 skip convolution
    # the deep 32x32 layers on F(4x4) + split-K (round 5: opt-in; round 6: on by default after a repeatable -0.7 % on the step);
    # ANODDPM_F43_32=0 restores F(2x2) + split-K there
    assert pick(32, 32, 512, 512, Z, f43=True) == (3, 4) and pick(32, 32, 512, 256, Z, f43=True) == (3, 8)
    assert pick(32, 32, 768, 256, Z, f43=True) == (3, 8) and pick(32, 32, 256, 256, Z, f43=True) == (2, 4)
    assert pick(16, 16, 512, 512, Z, f43=True) == (2, 8)
    assert pick(32, 32, 512, 512, Z) == (2, 2)                      # (a caller without F(4x4) weights keeps F(2x2))
    assert pick(32, 32, 512, 512, 1, f43=True)[0] == 2              # batch 1 (config 5): too few workgroups, F(2x2) + split-K stays
    assert pick(32, 32, 512, 512, 16, f43=True) == (3, 1)           # the detection loop's 16 slots: one K slice
    os.environ["ANODDPM_F43_32"] = "0"
    try:
        assert pick(32, 32, 512, 512, Z, f43=True) == (2, 2) and pick(32, 32, 768, 256, Z, f43=True) == (2, 4)
    finally:
        del os.environ["ANODDPM_F43_32"]
    assert pick(16, 16, 512, 1536, Z, ks=1) == (1, 2)               # qkv projection
    assert pick(256, 256, 128, 128, Z, wino=False)[0] in (0, 1)
    # Winograd F(4x4,3x3): only when the caller can supply its weights, on maps >= 128^2 with Cout % 128 == 0
    assert pick(256, 256, 128, 128, Z, f43=True) == (3, 1) and pick(128, 128, 384, 128, Z, c0=256, c1=128, f43=True) == (3, 1)
    assert pick(128, 128, 256, 256, Z, a_mode=1, f43=True) == (3, 1)                # fused nearest x2
    assert pick(64, 64, 256, 256, Z, f43=True) == (3, 1) and pick(128, 128, 128, 96, Z, f43=True)[0] == 2
    assert pick(32, 32, 256, 256, Z, f43=True) == (2, 4) and pick(64, 64, 128, 64, 1, f43=True)[0] != 3      # too few workgroups
    assert pick(128, 128, 128, 128, Z, a_mode=2, f43=True) == (0, 1)
    # streaming 1x1 (cfg 4): plain operands only, enough 32-pixel tiles x channel blocks for the chip's 2048 waves
    assert pick(256, 256, 256, 128, Z, ks=1, c0=128, c1=128, plain=True) == (4, 1)
    assert pick(128, 128, 384, 128, Z, ks=1, c0=256, c1=128, plain=True) == (4, 1)
    assert pick(64, 64, 512, 256, Z, ks=1, plain=True) == (4, 1)
    assert pick(32, 32, 512, 256, Z, ks=1, plain=True)[0] != 4 and pick(16, 16, 512, 512, Z, ks=1, plain=True)[0] != 4
    assert pick(256, 256, 256, 128, Z, ks=1)[0] == 0 and pick(256, 256, 192, 128, Z, ks=1, plain=True)[0] == 0
    cfg, ks = pick(32, 32, 768, 256, Z)
    assert cfg == 2 and (768 // 16) % 1 == 0 and (ks - 1) * -(-(768 // 16) // ks) < 768 // 16      # no empty K slice


def test_small_map_launch_policy_round4():
    """The inference plan's choices for the <= 32x32 maps (small=True): no-split-K small-map kernel (cfg 5) by rows over the batch,
    F(2x2,3x3) without split-K (cfg 6) for K <= 256, split-K + tail otherwise; streaming 1x1 from 1024 tile x block items."""
    from anoddpm_amd.unet import choose_conv_cfg as pick
    assert pick(8, 8, 512, 512, 4, small=True) == (5, 1)                       # 8x8 at batch 4: 256 rows
    assert pick(8, 8, 1024, 512, 4, c0=512, c1=512, small=True) == (5, 1)      # up path: virtual concat
    assert pick(16, 16, 512, 512, 1, small=True) == (5, 1)                     # 16x16 at batch 1 (config 5): the same 256 rows
    assert pick(16, 16, 512, 512, 4, small=True) == (2, 8)                     # 1024 rows: Winograd + split-K + tail
    assert pick(8, 8, 512, 512, 5, small=True) == (5, 1)                       # the detection loop's batch: map of <= 64 pixels
    assert pick(16, 16, 512, 1536, 4, ks=1, small=True) == (5, 1)              # qkv projection
    assert pick(16, 16, 256, 256, 4, small=True) == (6, 1) and pick(32, 32, 256, 256, 4, small=True) == (6, 1)
    assert pick(32, 32, 512, 256, 4, small=True)[0] == 2                       # K > 256: split-K
    assert pick(8, 8, 512, 512, 4) == (1, 16)                                  # the training plan does not ask for the small-map kernels here
    assert pick(64, 64, 128, 256, 4, ks=1, plain=True, small=True) == (4, 1)   # 1024 items: streaming 1x1, no split-K tail
    assert pick(32, 32, 512, 256, 4, ks=1, plain=True, small=True)[0] != 4     # 512 items: direct kernel


def test_f43_is_not_chosen_beyond_its_groupnorm_table():
    """Round-5 advisor finding: the F(4x4,3x3) kernels keep the image's GroupNorm affine in an LDS table of 1024 input channels and
    refuse wider GroupNorm-fused launches; the policy must not hand them one (a wider custom model, or a lowered
    ANODDPM_F43_MIN_PIXELS / ANODDPM_F43_32 on the deep layers of a base-256 model)."""
    from anoddpm_amd.unet import choose_conv_cfg as pick, F43_MAX_GN_K
    assert F43_MAX_GN_K == 1024
    assert pick(64, 64, 1024, 256, 4, f43=True) == (3, 1)                      # the widest shipped shape class still takes F(4x4)
    assert pick(64, 64, 1280, 256, 4, c0=768, c1=512, f43=True)[0] == 2        # K = 1280 with a fused GroupNorm: F(2x2)
    assert pick(64, 64, 1280, 256, 4, c0=768, c1=512, f43=True, plain=True) == (3, 1)      # no affine to table: not limited
    assert pick(32, 32, 1280, 256, 4, c0=768, c1=512, f43=True)[0] == 2          # the deep 32x32 layers (F(4x4) + split-K by default)
    assert pick(32, 32, 1024, 256, 4, f43=True)[0] == 3


def test_host_side_shape_helpers_of_the_library():
    """Pure host functions of the C ABI (no device needed): statistics rows of the fused stem sums, patch / group counts of
    the Winograd-domain weight gradient -- the values the plans size their buffers with."""
    from anoddpm_amd._lib import lib
    L = lib()
    # stem: one row per workgroup range; 8 strips of 8 pixels per workgroup trip at Cout = 128, at most 1024 rows per image; round 6:
    # two prefetched trips per workgroup for one input channel while that leaves >= 256 rows (256^2: 512 rows)
    assert L.anoddpm_stem_stats_rows(256, 256, 1, 128) == 512 and L.anoddpm_stem_stats_rows(512, 512, 1, 128) == 1024
    assert L.anoddpm_stem_stats_rows(256, 256, 2, 128) == 1024
    assert L.anoddpm_stem_stats_rows(128, 128, 1, 128) == 256 and L.anoddpm_stem_stats_rows(64, 64, 2, 64) == 32
    assert L.anoddpm_stem_stats_rows(64, 64, 3, 128) == 0 and L.anoddpm_stem_stats_rows(60, 60, 1, 128) == 0     # no fused form
    for (H, C) in ((256, 128), (512, 128), (64, 64), (16, 1024)):
        rows = L.anoddpm_stem_stats_rows(H, H, 1, C)
        assert rows > 0 and (H * H) % rows == 0 and (H * H // rows) % (8 * (256 // (C // 4))) == 0     # whole workgroup trips
    # weight gradient, algo 1: groups x blocks fill the 256 CUs; column-sum rows per image = one per group and tile row of its patches
    assert L.anoddpm_wgrad43_colsum_items(128, 128, 4, 256, 256) == 64 and L.anoddpm_wgrad43_colsum_items(512, 512, 4, 32, 32) == 4   # 2 rows per set
    assert L.anoddpm_wgrad43_groups(128, 128, 4, 256, 256) == 32 and L.anoddpm_wgrad43_groups(512, 512, 4, 32, 32) == 2
    assert L.anoddpm_wgrad43_groups(32, 64, 1, 16, 16) == 2                                       # fewer patches than groups
    # which cfg-3 launches run on the channel-sliced 128-channel kernel (the one whose epilogue can carry the GroupNorm-backward
    # reduction, anoddpm_igemm_args.gnb_*): grids that fill the chip with 128-channel workgroups; 64-channel workgroups otherwise
    assert L.anoddpm_f43_channel_sliced(256, 256, 128, 4) == 1 and L.anoddpm_f43_channel_sliced(128, 128, 128, 4) == 1
    assert L.anoddpm_f43_channel_sliced(128, 128, 256, 4) == 1 and L.anoddpm_f43_channel_sliced(64, 64, 256, 4) == 0      # 128 workgroups: halves
    assert L.anoddpm_f43_channel_sliced(64, 64, 256, 5) == 1 and L.anoddpm_f43_channel_sliced(64, 64, 192, 8) == 0        # N % 128
    assert L.anoddpm_f43_channel_sliced(60, 64, 128, 4) == 0 and L.anoddpm_f43_channel_sliced(0, 64, 128, 4) == 0


def test_reverse_chain_reuse_keys():
    """Which reverse chains may be restarted on their buffers (ReverseChain.reset): keyed by the noise source the captured
    graph contains -- 'gauss' / 'random' / a gaussian default noise_fn share one key, every simplex parameter set has its own,
    a host-RNG callable has none (it is never captured)."""
    import GaussianDiffusion as GD
    from anoddpm_amd.diffusion import ReverseChain
    d = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(100, "linear"), noise="simplex")
    g = GD.GaussianDiffusionModel([32, 32], GD.get_beta_schedule(100, "linear"), noise="gauss")
    key = ReverseChain._reuse_key_of
    assert key(d, "gauss") == key(d, "random") == key(g, "noise_fn") == ("gauss",)
    fa = GD.SimplexNoiseFn(d.simplex, octave=4, persistence=0.6, frequency=16)
    fb = GD.SimplexNoiseFn(d.simplex, octave=4, persistence=0.6, frequency=32)
    assert key(d, fa) == key(d, GD.SimplexNoiseFn(d.simplex, octave=4, persistence=0.6, frequency=16)) != key(d, fb)
    assert key(d, "noise_fn")[0] == "simplex" and key(d, "noise_fn") == key(d, "simplex")        # both: the :179-181 / :310 defaults
    assert key(d, lambda x, t: x) is None
    d.noise_fn = lambda x, t: x                                                                   # user-replaced noise_fn: host side, eager
    assert key(d, "noise_fn") is None


def test_chain_slot_schedule():
    """diffusion.plan_chain_slots: the host-side schedule behind the batched detection loops."""
    from anoddpm_amd.diffusion import plan_chain_slots
    for lengths, slots in (([d for d in range(50, 800, 50) for _ in range(5)], 16), ([d for d in range(50, 600, 50) for _ in range(2)] * 7, 16),
                           ([7, 3, 3, 1], 2), ([5], 4), ([], 3)):
        makespan, place = plan_chain_slots(lengths, slots)
        assert len(place) == len(lengths)
        busy = {}
        for L, (slot, start) in zip(lengths, place):
            assert 0 <= slot < slots and start >= 0 and start + L <= makespan
            for k in range(start, start + L):
                assert (slot, k) not in busy                      # one chain per slot and step
                busy[(slot, k)] = 1
        total = sum(lengths)
        assert len(busy) == total
        if lengths:
            assert makespan >= -(-total // slots) and makespan <= -(-total // slots) + max(lengths) - 1 + (1 if total % slots else 0)
            assert makespan >= max(lengths)
    # the detection_B gaussian sweep (15 settings x 5 chains) keeps 16 slots 96 % busy
    m, _ = plan_chain_slots([d for d in range(50, 800, 50) for _ in range(5)], 16)
    assert m == 1950
    with pytest.raises(ValueError):
        plan_chain_slots([3, 0], 2)
    with pytest.raises(ValueError):
        plan_chain_slots([3], 0)
