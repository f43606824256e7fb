#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the UPSTREAM REFERENCE itself.

Run in the build container only (needs /root/reference; the GPU box never has it):
    python tests/golden/make_golden.py
The reference is imported read-only with stubbed third-party modules (tests/golden/_refimport.py,
recipe from SURVEY.md 8c).  Outputs are DATA (inputs + expected outputs), never reference source:
    simplex_kat.npz      _init tables, noise3 point KATs (bit patterns), octave fields, C4 crops
    diffusion_kat.npz    schedule tables (linear/cosine), sample_q / p_mean_variance / sample_p
    unet_<name>.npz      UNetModel.forward outputs (+ per-block activation probes); unet_c2_256_b128_batch4.npz: BASELINE
                         config 2 at its benchmarked batch 4 with four timesteps, output only
    metrics_kat.npz      evaluation.py metrics + the mean / mse / threshold images of detection_A/B
    simplex2_kat.npz     2-D noise2 point KATs (bit patterns), a coordinate grid, octave fields
    vlb_kat.npz          calc_vlb_xt (KL and decoder-NLL branches) and the MSE curves of calc_total_vlb
    vlb_total_kat.npz    calc_total_vlb itself (T = 100, injected randn_like draws, analytic eps-model): all five returned curves
    loss_kat.npz         p_loss / calc_loss for l1, l2, hybrid (uniform and prop-t weights): per-sample terms, the scalar,
                         and d(scalar)/d(model output) from the reference's own autograd
    train_<name>.npz     two optimiser steps of the reference loop body (diffusion_training.py:99-107): p_loss scalars,
                         gradient probes / norms of every parameter, parameter + EMA probes after each step
    detection_fixedT.npz detection_A_fixedT (GaussianDiffusion.py:596-623) on a 32^2 model, seeded numpy stream
    detection_loops_kat.npz  detection_B ("gauss", "octave") and detection_A (:480-594) on a 32^2 model, T = 200: per setting the
                         tensor upstream plots (x_0, chains, mean, mse, threshold, mask), keyed randn_like draws (keyed.py)
    mri_loader.npz       MRIDataset normalisation + slice (dataset.py:585-594, 621-625) and the deterministic part of
                         its transform (centre-crop 235, bilinear resize, scale; torchvision -> PIL) on a synthetic volume
"""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refimport  # noqa: E402
from oracle import unet_oracle  # noqa: E402  (only for the deterministic parameter fill)

ref_simplex, ref_unet, ref_gd = _refimport.load()


def classify(x, y, z):
    """Which branch family of simplex.py:354/469/587 a point exercises (for coverage stats)."""
    s = (x + y + z) * (-1.0 / 6)
    xs, ys, zs = x + s, y + s, z + s
    xi, yi, zi = xs - np.floor(xs), ys - np.floor(ys), zs - np.floor(zs)
    t = xi + yi + zi
    return 0 if t <= 1 else (1 if t >= 2 else 2)


def gen_simplex():
    out = {}
    seeds = [3, 12345, -9999999999, 9999999999, 9726117423]
    out["init_seeds"] = np.array(seeds, dtype=np.int64)
    perms, pgis = [], []
    for s in seeds:
        p, g = ref_simplex._init(s)
        perms.append(p)
        pgis.append(g)
    out["init_perm"] = np.stack(perms)
    out["init_pgi3"] = np.stack(pgis)

    rng = np.random.RandomState(20260926)
    pts = np.concatenate([
        rng.uniform(-60, 60, (5000, 3)),
        rng.uniform(-2, 2, (3000, 3)),
        rng.randint(-30, 30, (1500, 3)).astype(np.float64),          # exact lattice points
        rng.randint(-60, 60, (1500, 3)) / 2.0,                       # f=0.5 style coordinates
        rng.randint(0, 256, (1500, 3)) / 64.0,                       # octave-0 style coordinates
        np.array([[0, 0, 0], [0.1, 0.2, 0.3], [1.5, 2.25, 0.15625],
                  [3.984375, 3.984375, 3.890625], [510, 510, 1998]], dtype=np.float64),
    ])
    out["points"] = pts
    hist = np.zeros(3, dtype=np.int64)
    for p in pts:
        hist[classify(*p)] += 1
    out["points_region_hist"] = hist
    for s in (3, 12345):
        S = ref_simplex.Simplex_CLASS()
        S.newSeed(s)
        out[f"points_val_seed{s}"] = np.array([S.noise3(*p) for p in pts], dtype=np.float64)

    S = ref_simplex.Simplex_CLASS()
    S.newSeed(9726117423)
    ts = [0, 1, 249, 999]
    out["fixedT_t"] = np.array(ts, dtype=np.int64)
    out["fixedT_seed"] = np.int64(9726117423)
    out["fixedT_64x64_o6"] = np.stack(
        [S.rand_3d_fixed_T_octaves((64, 64), np.array([t]), 6, 0.8, 64)[0] for t in ts])
    out["fixedT_40x24_o8_f32"] = np.stack(
        [S.rand_3d_fixed_T_octaves((40, 24), np.array([t]), 8, 0.7, 32)[0] for t in ts])

    # crops of the BASELINE config-4 volume rand_3d_octaves((1000,256,256), 8, 0.8, 64), seed 12345
    S.newSeed(12345)
    zs = [0, 1, 499, 999]
    crops = []
    for (y0, x0) in ((0, 0), (224, 224)):
        acc = np.zeros((len(zs), 32, 32))
        f, a = 64, 1
        for _ in range(8):
            acc += a * S.noise3array(np.arange(x0, x0 + 32) / f, np.arange(y0, y0 + 32) / f,
                                     np.array(zs) / f)
            f /= 2
            a *= 0.8
        crops.append(acc)
    out["c4_seed"] = np.int64(12345)
    out["c4_z"] = np.array(zs, dtype=np.int64)
    out["c4_crop_origin_yx"] = np.array([[0, 0], [224, 224]], dtype=np.int64)
    out["c4_crops"] = np.stack(crops)
    # a full tiny volume through the reference entry point itself
    out["vol_5x12x20_o3"] = S.rand_3d_octaves((5, 12, 20), 3, 0.5, 8)
    np.savez_compressed(os.path.join(HERE, "simplex_kat.npz"), **out)
    print("simplex_kat.npz", {k: getattr(v, "shape", ()) for k, v in out.items()})


def gen_diffusion():
    out = {}
    for name in ("linear", "cosine"):
        betas = ref_gd.get_beta_schedule(1000, name)
        d = ref_gd.GaussianDiffusionModel([16, 16], betas, noise="gauss")
        out[f"{name}_betas"] = betas
        for k in ("sqrt_alphas", "sqrt_betas", "alphas_cumprod", "alphas_cumprod_prev",
                  "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                  "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                  "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                  "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
            out[f"{name}_{k}"] = getattr(d, k)
    out["linear_T250_betas"] = ref_gd.get_beta_schedule(250, "linear")

    g = torch.Generator().manual_seed(7)
    x = torch.rand(4, 1, 16, 16, generator=g) * 2 - 1
    eps = torch.randn(4, 1, 16, 16, generator=g)
    noise = torch.randn(4, 1, 16, 16, generator=g)
    t = torch.tensor([0, 1, 500, 999])
    out["x"], out["eps"], out["noise"], out["t"] = x.numpy(), eps.numpy(), noise.numpy(), t.numpy()
    for name in ("linear", "cosine"):
        d = ref_gd.GaussianDiffusionModel([16, 16], ref_gd.get_beta_schedule(1000, name), noise="gauss")
        out[f"{name}_sample_q"] = d.sample_q(x, t, noise).numpy()
        out[f"{name}_sample_q_gradual"] = d.sample_q_gradual(x, t, noise).numpy()
        pmv = d.p_mean_variance(None, x, t, estimate_noise=eps)
        for k, v in pmv.items():
            out[f"{name}_pmv_{k}"] = v.contiguous().numpy()
        sp = d.sample_p(lambda a, b: eps, x, t, denoise_fn=lambda a, b: noise)
        out[f"{name}_sample_p_sample"] = sp["sample"].numpy()
        out[f"{name}_sample_p_pred_x_0"] = sp["pred_x_0"].numpy()
        out[f"{name}_predict_eps_from_x_0"] = d.predict_eps_from_x_0(x, t, pmv["pred_x_0"]).numpy()

    # forward_backward structure with an injected model / noise (lengths + a short chain)
    d = ref_gd.GaussianDiffusionModel([16, 16], ref_gd.get_beta_schedule(1000, "linear"), noise="gauss")
    x1 = x[:1]
    fixed = noise[:1]
    d.noise_fn = lambda a, b: fixed
    model = lambda a, b: 0.3 * a - 0.1
    seq_half = d.forward_backward(model, x1, "half", 5, denoise_fn=lambda a, b: 0.5 * fixed)
    seq_whole = d.forward_backward(model, x1, "whole", 4, denoise_fn=lambda a, b: 0.5 * fixed)
    final = d.forward_backward(model, x1, None, 5, denoise_fn=lambda a, b: 0.5 * fixed)
    out["fb_half_len"] = np.int64(len(seq_half))
    out["fb_whole_len"] = np.int64(len(seq_whole))
    out["fb_half_seq"] = torch.stack(seq_half).numpy()
    out["fb_whole_seq"] = torch.stack(seq_whole).numpy()
    out["fb_final"] = final.numpy()

    # losses with injected t / noise
    for lt in ("l1", "l2", "hybrid"):
        d = ref_gd.GaussianDiffusionModel([16, 16], ref_gd.get_beta_schedule(1000, "linear"),
                                          loss_type=lt, noise="gauss")
        d.noise_fn = lambda a, b: noise
        loss, x_t, est = d.calc_loss(model, x, t)
        out[f"loss_{lt}"] = loss["loss"].numpy()
        if lt == "hybrid":
            out["loss_hybrid_vlb"] = loss["vlb"].numpy()
    np.savez_compressed(os.path.join(HERE, "diffusion_kat.npz"), **out)
    print("diffusion_kat.npz", len(out), "arrays")


UNET_CASES = {
    # name: (ctor kwargs, batch, timesteps)
    "i32_b32_h1": (dict(img_size=32, base_channels=32), 2, [3, 977]),
    "i32_b32_h2_a16_8": (dict(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8"), 1, [500]),
    "i64_b32_hc32": (dict(img_size=64, base_channels=32, n_head_channels=32, attention_resolutions="16,8"), 2, [0, 999]),
    "i64_b64_c3": (dict(img_size=64, base_channels=64, n_heads=2, in_channels=3), 1, [42]),
    "i128_b32_h2": (dict(img_size=128, base_channels=32, n_heads=2, attention_resolutions="16,8"), 1, [250]),
    # BASELINE config 5's topology at a quarter of its size: explicit mults (1,1,2,2,4,4), attention "32,16,8"
    # (sequence lengths 1024 / 256 / 64 as in the 512^2 model)
    "c5like_i128_b32": (dict(img_size=128, base_channels=32, n_heads=2, channel_mults=(1, 1, 2, 2, 4, 4),
                             attention_resolutions="32,16,8"), 2, [7, 640]),
    # biggan_updown=False: Downsample / Upsample layers between the levels (UNet.py:60-92) with and without the convolutions
    "convrs_i64_b32": (dict(img_size=64, base_channels=32, n_heads=2, attention_resolutions="16,8", biggan_updown=False,
                            conv_resample=True), 2, [11, 870]),
    "poolrs_i32_b32": (dict(img_size=32, base_channels=32, biggan_updown=False, conv_resample=False), 1, [333]),
}


def run_unet_case(name, kw, batch, ts, probes=True):
    kw = dict(kw)
    model = ref_unet.UNetModel(**kw)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    oshapes = unet_oracle.param_shapes(
        kw["img_size"], kw["base_channels"], kw.get("channel_mults", ""), 2,
        kw.get("attention_resolutions", "32,16,8"), kw.get("in_channels", 1), kw.get("biggan_updown", True),
        kw.get("conv_resample", True))
    assert list(shapes.items()) == list(oshapes.items()), "state-dict layout mismatch vs oracle.param_shapes"
    sd = unet_oracle.fill_deterministic(shapes)
    model.load_state_dict(sd)
    model.eval()
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    x = torch.rand(batch, kw.get("in_channels", 1), kw["img_size"], kw["img_size"], generator=g) * 2 - 1
    t = torch.tensor(ts)
    rec = {}
    hooks = []
    if probes:
        def mk(key):
            def hook(mod, inp, outp):
                rec[key] = outp.detach()
            return hook
        for grp in ("down", "up"):
            for i, seq in enumerate(getattr(model, grp)):
                for j, m in enumerate(seq):
                    hooks.append(m.register_forward_hook(mk(f"{grp}.{i}.{j}")))
        for j, m in enumerate(model.middle):
            hooks.append(m.register_forward_hook(mk(f"middle.{j}")))
        hooks.append(model.time_embedding.register_forward_hook(mk("time_embed")))
    with torch.no_grad():
        y = model(x, t)
    for h in hooks:
        h.remove()
    out = {"x": x.numpy(), "t": t.numpy(), "y": y.numpy(),
           "n_params": np.int64(sum(int(np.prod(s)) for s in shapes.values())),
           "n_tensors": np.int64(len(shapes))}
    for k, v in rec.items():
        f = v.flatten()
        stride = max(1, f.numel() // 256)
        out["probe/" + k] = f[::stride][:256].numpy().copy()
        out["stat/" + k] = np.array([v.mean().item(), v.abs().mean().item(), v.std().item()], dtype=np.float64)
        out["shape/" + k] = np.array(v.shape, dtype=np.int64)
    out["keys"] = np.array(list(shapes.keys()))
    out["key_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
    return out


def gen_unet(only=None):
    for name, (kw, batch, ts) in UNET_CASES.items():
        if only and name not in only:
            continue
        out = run_unet_case(name, kw, batch, ts)
        np.savez_compressed(os.path.join(HERE, f"unet_{name}.npz"), **out)
        print(f"unet_{name}.npz", "y", out["y"].shape, "params", int(out["n_params"]),
              "|y| mean", float(np.abs(out["y"]).mean()))
    if only and "c2_256_b128" not in only:
        return
    # BASELINE config-2 shaped forward (256^2, base 128, heads 2, attn 16,8), batch 1
    kw = dict(img_size=256, base_channels=128, n_heads=2, attention_resolutions="16,8")
    out = run_unet_case("c2_256_b128", kw, 1, [123], probes=True)
    out.pop("keys"); out.pop("key_shapes")
    np.savez_compressed(os.path.join(HERE, "unet_c2_256_b128.npz"), **out)
    print("unet_c2_256_b128.npz params", int(out["n_params"]), "|y| mean", float(np.abs(out["y"]).mean()))


def gen_unet_c2_batch4():
    """BASELINE config 2 exactly as benchmarked: 256^2, base 128, heads 2, attn 16,8 at BATCH 4 with four different timesteps (the
    build picks other kernels at batch 4 than at batch 1: 128-channel F(4x4) grids, other split-K factors).  Output only."""
    kw = dict(img_size=256, base_channels=128, n_heads=2, attention_resolutions="16,8")
    out = run_unet_case("c2_256_b128_batch4", kw, 4, [0, 249, 500, 999], probes=False)
    out.pop("keys"); out.pop("key_shapes")
    np.savez_compressed(os.path.join(HERE, "unet_c2_256_b128_batch4.npz"), **out)
    print("unet_c2_256_b128_batch4.npz |y| mean", float(np.abs(out["y"]).mean()))


def gen_unet_c5():
    """BASELINE config 5: 512^2, base 128, mults (1,1,2,2,4,4), attention at 32/16/8, two heads; batch 1."""
    kw = dict(img_size=512, base_channels=128, n_heads=2, channel_mults=(1, 1, 2, 2, 4, 4), attention_resolutions="32,16,8")
    out = run_unet_case("c5_512_b128", kw, 1, [777], probes=True)
    out.pop("keys"); out.pop("key_shapes")
    out["x"] = out["x"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "unet_c5_512_b128.npz"), **out)
    print("unet_c5_512_b128.npz params", int(out["n_params"]), "|y| mean", float(np.abs(out["y"]).mean()))


TRAIN_CASES = {
    # name: (ctor kwargs, batch, optimiser kwargs, [t per step])
    # base 64 -> two channels per GroupNorm group at the first level: no parameter has an identically-zero gradient
    "i64_b64_h2": (dict(img_size=64, base_channels=64, n_heads=2, attention_resolutions="16,8"), 2,
                   dict(lr=1e-4, weight_decay=0.0), [[17, 640], [799, 3]]),
    "i128_b32_hc32": (dict(img_size=128, base_channels=32, n_head_channels=32, attention_resolutions="16,8"), 2,
                      dict(lr=2e-4, weight_decay=0.01), [[250, 0], [31, 555]]),
    # the shapes that reach the Winograd F(4x4,3x3) forward / data-gradient kernels (both variants: 128-channel grids of >= 200
    # workgroups at 128^2, 64-channel workgroups at 64^2) and the Winograd-domain weight gradient: base 128, batch 4; one step
    "i128_b128_f43": (dict(img_size=128, base_channels=128, n_heads=2, attention_resolutions="16,8"), 4,
                      dict(lr=1e-4, weight_decay=0.0), [[17, 640, 3, 799]]),
}


def _probe(v, n=256):
    f = v.detach().flatten()
    stride = max(1, f.numel() // n)
    return f[::stride][:n].numpy().copy()


def gen_training(only=None):
    """The reference training loop body, run by the reference's own classes (diffusion_training.py:99-107):
       loss, est = diffusion.p_loss(model, x, args); optimiser.zero_grad(); loss.backward();
       clip_grad_norm_(model.parameters(), 1); optimiser.step(); update_ema_params(ema, model)
    with the three random draws injected: `t` (torch.randint inside p_loss is patched), the forward noise
    (diffusion.noise_fn) and the data batch."""
    import copy
    from unittest import mock
    for name, (kw, B, okw, tsteps) in TRAIN_CASES.items():
        if only and name not in only:
            continue
        S = kw["img_size"]
        model = ref_unet.UNetModel(**kw)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = unet_oracle.perturb(unet_oracle.fill_deterministic(shapes))
        model.load_state_dict(sd)
        model.train()
        ema = copy.deepcopy(model)
        opt = torch.optim.AdamW(model.parameters(), betas=(0.9, 0.999), **okw)          # diffusion_training.py:75
        diff = ref_gd.GaussianDiffusionModel([S, S], ref_gd.get_beta_schedule(1000, "linear"), loss_type="l2", noise="gauss")
        args = {"train_start": True, "sample_distance": 800, "Batch_Size": B}
        g = torch.Generator().manual_seed(zlib.crc32(("train" + name).encode()))
        out = {"lr": np.float64(okw["lr"]), "weight_decay": np.float64(okw["weight_decay"]),
               "keys": np.array(list(shapes.keys()))}
        for step, ts in enumerate(tsteps):
            x0 = torch.rand(B, 1, S, S, generator=g) * 2 - 1
            noise = torch.randn(B, 1, S, S, generator=g)
            t = torch.tensor(ts)
            diff.noise_fn = lambda a, b, _n=noise: _n
            with mock.patch.object(torch, "randint", lambda *a, **k: t.clone()):
                loss, est = diff.p_loss(model, x0, args)
            opt.zero_grad()
            loss.backward()
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
            norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1)
            opt.step()
            ref_unet.update_ema_params(ema, model)
            out[f"s{step}/x0"], out[f"s{step}/noise"], out[f"s{step}/t"] = x0.numpy(), noise.numpy(), t.numpy()
            out[f"s{step}/loss"] = np.float64(loss.item())
            out[f"s{step}/x_t"] = _probe(est[1], 1024)
            out[f"s{step}/eps"] = _probe(est[2], 1024)
            out[f"s{step}/grad_norm"] = np.float64(norm.item())
            out[f"s{step}/gnorm"] = np.array([grads[k].double().norm().item() for k in shapes], dtype=np.float64)
            out[f"s{step}/gprobe"] = np.stack([np.resize(_probe(grads[k]), 256) for k in shapes])
            params, emas = dict(model.named_parameters()), dict(ema.named_parameters())
            out[f"s{step}/pprobe"] = np.stack([np.resize(_probe(params[k]), 256) for k in shapes])
            out[f"s{step}/eprobe"] = np.stack([np.resize(_probe(emas[k]), 256) for k in shapes])
            out[f"s{step}/psum"] = np.array([params[k].detach().double().sum().item() for k in shapes], dtype=np.float64)
            print(f"train_{name} step {step}: loss {loss.item():.6f} |g| {norm.item():.4f}")
        out["p0probe"] = np.stack([np.resize(_probe(sd[k]), 256) for k in shapes])
        np.savez_compressed(os.path.join(HERE, f"train_{name}.npz"), **out)
        print(f"train_{name}.npz", len(out), "arrays")


def gen_detection():
    """detection_A_fixedT (GaussianDiffusion.py:596-623) run by the reference on a 32^2 model: 250-step chains with
    simplex noise at frequencies 2 and 4 in both directions; every random draw comes from the seeded global numpy
    stream (Simplex_CLASS.newSeed), so the routine is reproducible draw for draw."""
    kw = dict(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8")
    model = ref_unet.UNetModel(**kw)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(unet_oracle.fill_deterministic(shapes))
    model.eval()
    d = ref_gd.GaussianDiffusionModel([32, 32], ref_gd.get_beta_schedule(1000, "linear"), noise="simplex")
    g = torch.Generator().manual_seed(4242)
    x0 = (torch.rand(1, 1, 32, 32, generator=g) * 2 - 1)
    mask = (torch.rand(1, 1, 32, 32, generator=g) > 0.8).float()
    np.random.seed(20260927)
    out = d.detection_A_fixedT(model, x0, {"img_size": [32, 32]}, mask, end_freq=2)
    after = np.random.randint(-10000000000, 10000000000)          # position of the numpy stream after the call
    np.savez_compressed(os.path.join(HERE, "detection_fixedT.npz"), x0=x0.numpy(), mask=mask.numpy(),
                        output=out.numpy(), np_seed=np.int64(20260927), end_freq=np.int64(2),
                        next_randint=np.int64(after))
    print("detection_fixedT.npz", out.shape, float(out.abs().mean()))


def gen_detection_loops():
    """detection_B ("gauss" and "octave") and detection_A (GaussianDiffusion.py:480-594) run by the REFERENCE on a 32^2 model with
    T = 200: the serial (t_distance, avg) loops, their forward-noise draw order, `output[avg]` placement and the
    mean -> mse -> threshold post-processing.  `torch.randn_like` is replaced by a keyed stream (tests/golden/keyed.py: the value
    depends on (chain, t), chain = index in upstream's loop order), the simplex forward noise comes from the seeded numpy stream;
    the figure calls are stubbed and what upstream hands to `gridify_output` (cat[x_0, output[:3], mean, mse, threshold, mask]) is
    recorded per setting."""
    import tempfile
    import keyed
    kw = dict(img_size=32, base_channels=32, n_heads=2, attention_resolutions="16,8")
    model = ref_unet.UNetModel(**kw)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(unet_oracle.fill_deterministic(shapes))
    model.eval()
    T = 200
    args = {"T": T, "img_size": [32, 32], "arg_num": 0}
    g = torch.Generator().manual_seed(777)
    x0 = torch.rand(1, 1, 32, 32, generator=g) * 2 - 1
    mask = (torch.rand(1, 1, 32, 32, generator=g) > 0.8).float()
    out = {"x0": x0.numpy(), "mask": mask.numpy(), "T": np.int64(T)}

    state = {"chain": 0, "t": None}
    grids = []

    def run(fn_name, np_seed, **kwargs):
        d = ref_gd.GaussianDiffusionModel([32, 32], ref_gd.get_beta_schedule(T, "linear"), noise="simplex")
        state["chain"], state["t"] = 0, None
        grids.clear()
        real_q, real_p = d.sample_q, d.sample_p

        def sample_q(x_0, t, noise):                    # one call per chain, after its forward noise was drawn
            r = real_q(x_0, t, noise)
            return r

        def sample_p(m, x_t, t, denoise_fn="gauss"):
            state["t"] = int(t[0])
            r = real_p(m, x_t, t, denoise_fn)
            if int(t[0]) == 0:
                state["chain"] += 1                     # the chain's last reverse step
            return r

        def randn_like(x, *a, **k):
            if state["t"] is None:                      # detection_B "gauss": the chain's forward noise (:552)
                return keyed.keyed_normal(state["chain"], keyed.FORWARD, x.shape)
            t, state["t"] = state["t"], None
            return keyed.keyed_normal(state["chain"], t, x.shape)

        d.sample_q, d.sample_p = sample_q, sample_p
        saved = (torch.randn_like, ref_gd.gridify_output, ref_gd.plt.imshow, ref_gd.plt.savefig, ref_gd.plt.axis, ref_gd.plt.clf,
                 ref_gd.evaluation.heatmap)
        torch.randn_like = randn_like
        ref_gd.gridify_output = lambda img, *a, **k: (grids.append(img.detach().clone()), np.zeros((4, 4)))[1]
        ref_gd.plt.imshow = ref_gd.plt.savefig = ref_gd.plt.axis = ref_gd.plt.clf = lambda *a, **k: None
        ref_gd.evaluation.heatmap = lambda *a, **k: None
        cwd = os.getcwd()
        np.random.seed(np_seed)
        try:
            with tempfile.TemporaryDirectory() as tmp:
                os.chdir(tmp)
                ret = getattr(d, fn_name)(model, x0, args, ("f", "0"), mask, **kwargs)
        finally:
            os.chdir(cwd)
            (torch.randn_like, ref_gd.gridify_output, ref_gd.plt.imshow, ref_gd.plt.savefig, ref_gd.plt.axis, ref_gd.plt.clf,
             ref_gd.evaluation.heatmap) = saved
        after = np.random.randint(-10000000000, 10000000000)
        return ret, torch.stack(grids).numpy(), state["chain"], after

    ret, gr, nchain, after = run("detection_B", 101, denoise_fn="gauss", total_avg=3)
    assert ret == [None] * 3 and nchain == 9 and gr.shape == (3, 8, 1, 32, 32)
    out.update(B_gauss=gr, B_gauss_total_avg=np.int64(3), B_gauss_np_seed=np.int64(101), B_gauss_next_randint=np.int64(after))
    ret, gr, nchain, after = run("detection_B", 202, denoise_fn="octave", total_avg=3)
    assert ret == [None] * 2 and nchain == 6 and gr.shape == (2, 8, 1, 32, 32)
    out.update(B_octave=gr, B_octave_total_avg=np.int64(3), B_octave_np_seed=np.int64(202), B_octave_next_randint=np.int64(after))
    ret, gr, nchain, after = run("detection_A", 303, total_avg=2)
    assert ret is None and nchain == 28 and gr.shape == (14, 7, 1, 32, 32)
    out.update(A=gr, A_total_avg=np.int64(2), A_np_seed=np.int64(303), A_next_randint=np.int64(after))
    np.savez_compressed(os.path.join(HERE, "detection_loops_kat.npz"), **out)
    print("detection_loops_kat.npz:", {k: getattr(v, "shape", ()) for k, v in out.items()})


def gen_loader():
    """MRIDataset (dataset.py:575-643) run by the reference on a synthetic NIfTI-shaped volume (nibabel stubbed: `nib.load`
    returns an object whose get_fdata() is the array): the normalised .npy cache it writes and the slices it cuts.  The
    default transform needs torchvision, which is not installed; its deterministic stages are Pillow calls, so the fixture
    holds Pillow's own outputs for centre-crop-235 + BILINEAR resize (+ Normalize) and for AFFINE / NEAREST with matrices
    built by oracle/loader_oracle.py's restatement of torchvision's parameter glue."""
    import importlib
    import tempfile
    import types
    from PIL import Image
    from oracle import loader_oracle as lo
    vol = lo.synthetic_volume()
    saved = {k: sys.modules.get(k) for k in ("cv2", "nibabel", "torchvision", "dataset")}
    nib = types.ModuleType("nibabel")
    nib.load = lambda path: types.SimpleNamespace(get_fdata=lambda: vol)
    sys.modules["nibabel"] = nib
    sys.modules["cv2"] = types.ModuleType("cv2")
    tv = types.ModuleType("torchvision")
    tv.datasets, tv.transforms = types.ModuleType("torchvision.datasets"), types.ModuleType("torchvision.transforms")
    sys.modules["torchvision"] = tv
    sys.path.insert(0, _refimport.REF)
    try:
        sys.modules.pop("dataset", None)
        ref_ds = importlib.import_module("dataset")
    finally:
        sys.path.remove(_refimport.REF)
    out = {"volume_probe": vol.flatten()[::9973].copy()}     # the volume itself is regenerated by loader_oracle.synthetic_volume()
    with tempfile.TemporaryDirectory() as root:
        os.makedirs(os.path.join(root, "vol0"))
        ds = ref_ds.MRIDataset(root, transform=lambda a: a, img_size=(64, 64), random_slice=False)
        s80 = ds[0]["image"]                                 # writes vol0/vol0.npy, returns the slice at 80
        npy = np.load(os.path.join(root, "vol0", "vol0.npy"))
        out["npy_shape"] = np.array(npy.shape, dtype=np.int64)
        out["npy_probe"] = npy.flatten()[::97].copy()
        out["npy_stats"] = np.array([npy.astype(np.float64).mean(), npy.astype(np.float64).std(), npy.min(), npy.max()])
        out["slice80"] = s80
        import random
        random.seed(5)
        ds2 = ref_ds.MRIDataset(root, transform=lambda a: a, img_size=(64, 64), random_slice=True)
        sl = [ds2[0]["image"] for _ in range(3)]
        random.seed(5)
        out["random_slices_idx"] = np.array([random.randint(40, 100) for _ in range(3)], dtype=np.int64)
        out["random_slices"] = np.stack(sl)
    for k, v in saved.items():
        if v is not None:
            sys.modules[k] = v
        else:
            sys.modules.pop(k, None)
    # deterministic transform stages through Pillow itself
    img = Image.fromarray(s80, mode="F")                     # torchvision ToPILImage of a float32 2-D array
    w, h = img.size
    pl, pt, ct, cl = lo.center_crop_geometry(h, w, 235)
    padded = Image.new("F", (w + pl + (235 - w + 1) // 2 if 235 > w else w, h), 0.0)
    padded.paste(img, (pl, pt))
    cropped = padded.crop((cl, ct, cl + 235, ct + 235))
    out["crop235"] = np.array(cropped)
    for (oh, ow) in ((64, 64), (256, 256), (32, 48)):
        r = np.array(cropped.resize((ow, oh), Image.BILINEAR))
        out[f"resized_{oh}x{ow}"] = r
        out[f"final_{oh}x{ow}"] = ((torch.from_numpy(r) - 0.5) / 0.5).numpy()[None]           # ToTensor + Normalize(0.5, 0.5)
    params = [(2.3, (3, -17)), (-2.9, (-2, 20)), (0.7, (0, 5))]
    out["affine_angle"] = np.array([p[0] for p in params])
    out["affine_translate"] = np.array([p[1] for p in params], dtype=np.int64)
    for i, (ang, tr) in enumerate(params):
        m = lo.inverse_affine_matrix((w * 0.5, h * 0.5), ang, tr)
        a = img.transform((w, h), Image.AFFINE, m, Image.NEAREST, fillcolor=0)
        out[f"affine{i}"] = np.array(a)
        pa = Image.new("F", padded.size, 0.0)
        pa.paste(a, (pl, pt))
        out[f"affine{i}_final_64x64"] = ((torch.from_numpy(np.array(pa.crop((cl, ct, cl + 235, ct + 235)).resize((64, 64), Image.BILINEAR))) - 0.5) / 0.5).numpy()[None]
    np.savez_compressed(os.path.join(HERE, "mri_loader.npz"), **out)
    print("mri_loader.npz", {k: getattr(v, "shape", ()) for k, v in out.items()})


def gen_metrics():
    """Anomaly-map arithmetic and segmentation metrics: the reference's evaluation.py functions (called) and the
    inline tensor expressions of GaussianDiffusion.py:572, 581-583 / detection.py:229-232 (evaluated with torch CPU)."""
    import torch
    ev = ref_gd.evaluation
    rng = np.random.RandomState(77)
    out = {}
    for case, (B, H, navg) in {"one": (1, 32, 5), "batch": (3, 24, 1)}.items():
        real = (rng.rand(B, 1, H, H).astype(np.float32) * 2 - 1)
        recon = np.clip(real[None] + rng.randn(navg, B, 1, H, H).astype(np.float32) * 0.6, -1, 1).astype(np.float32)
        mask = (rng.rand(B, 1, H, H) > 0.7).astype(np.float32)
        x0, outp, mk = torch.from_numpy(real), torch.from_numpy(recon), torch.from_numpy(mask)
        mean = torch.mean(outp, dim=[0]).reshape(B, 1, H, H)                   # GaussianDiffusion.py:572
        mse_img = ((mean - x0).square() * 2) - 1                                # :581
        thr_img = ((mse_img > 0).float() * 2) - 1                               # :582-583
        se = (x0 - mean).square()                                               # detection.py:229
        pred = (se > 0.5).float()                                               # detection.py:232
        out[f"{case}_real"], out[f"{case}_recon"], out[f"{case}_mask"] = real, recon, mask
        out[f"{case}_mean"], out[f"{case}_mse_img"], out[f"{case}_thr_img"] = mean.numpy(), mse_img.numpy(), thr_img.numpy()
        out[f"{case}_sqerr"], out[f"{case}_pred"] = se.numpy(), pred.numpy()
        out[f"{case}_dice"] = ev.dice_coeff(x0, mean, mk).numpy()
        out[f"{case}_dice_mse"] = ev.dice_coeff(x0, mean, mk, mse=pred).numpy()
        out[f"{case}_precision"] = ev.precision(mk, pred).numpy()
        out[f"{case}_recall"] = ev.recall(mk, pred).numpy()
        out[f"{case}_FPR"] = ev.FPR(mk, pred).numpy()
        out[f"{case}_IoU"] = np.float64(ev.IoU(mk, pred))
        out[f"{case}_PSNR"] = ev.PSNR(mean, x0)
    np.savez_compressed(os.path.join(HERE, "metrics_kat.npz"), **out)
    print("metrics_kat.npz:", len(out), "arrays")


def gen_vlb():
    """calc_vlb_xt (incl. the t == 0 decoder-NLL branch and both |x_0| > 0.999 branches) and the per-step MSE
    curves of calc_total_vlb, from the reference with injected eps / noise."""
    g = torch.Generator().manual_seed(19)
    B, H = 6, 16
    x0 = torch.rand(B, 1, H, H, generator=g) * 2 - 1
    x0[:, :, 0, :4] = -1.0                      # exact background value: the x < -0.999 branch
    x0[:, :, 1, :4] = 1.0                       # the x > 0.999 branch
    eps = torch.randn(B, 1, H, H, generator=g)
    noise = torch.randn(B, 1, H, H, generator=g)
    out = {"x0": x0.numpy(), "eps": eps.numpy(), "noise": noise.numpy()}
    for name in ("linear", "cosine"):
        d = ref_gd.GaussianDiffusionModel([H, H], ref_gd.get_beta_schedule(1000, name), noise="gauss")
        for tag, t in (("mixed", torch.tensor([0, 1, 2, 500, 998, 999])), ("zero", torch.zeros(B, dtype=torch.int64))):
            x_t = d.sample_q(x0, t, noise)
            r = d.calc_vlb_xt(None, x0, x_t, t, estimate_noise=eps)
            pred = r["pred_x_0"]
            out[f"{name}_{tag}_t"] = t.numpy()
            out[f"{name}_{tag}_x_t"] = x_t.numpy()
            out[f"{name}_{tag}_vlb"] = r["output"].numpy()
            out[f"{name}_{tag}_pred_x_0"] = pred.numpy()
            out[f"{name}_{tag}_x_0_mse"] = ref_gd.mean_flat((pred - x0) ** 2).numpy()              # :463
            e2 = d.predict_eps_from_x_0(x_t, t, pred)                                               # :464
            out[f"{name}_{tag}_mse"] = ref_gd.mean_flat((e2 - noise) ** 2).numpy()                  # :465
    np.savez_compressed(os.path.join(HERE, "vlb_kat.npz"), **out)
    print("vlb_kat.npz:", len(out), "arrays")


def gen_vlb_total():
    """calc_total_vlb itself (GaussianDiffusion.py:445-478): the reference's loop over all T steps with an analytic eps-model and
    its `torch.randn_like` draws injected (T pre-generated fields, popped in call order), so that the build can replay the same
    draws: vb / x_0_mse / mse curves, prior_vlb, total_vlb."""
    T, B, H = 100, 3, 16
    g = torch.Generator().manual_seed(29)
    x0 = torch.rand(B, 1, H, H, generator=g) * 2 - 1
    x0[:, :, 0, :4] = -1.0
    x0[:, :, 1, :4] = 1.0
    draws = torch.randn(T, B, 1, H, H, generator=g)
    out = {"x0": x0.numpy(), "draws": draws.numpy(), "T": np.int64(T)}
    model = lambda x, t: 0.3 * x - 0.05 * t.view(-1, 1, 1, 1).float() / T
    for name in ("linear", "cosine"):
        d = ref_gd.GaussianDiffusionModel([H, H], ref_gd.get_beta_schedule(T, name), noise="gauss")
        it = iter(draws)
        real = torch.randn_like
        torch.randn_like = lambda x, *a, **k: next(it).clone()
        try:
            r = d.calc_total_vlb(x0, model, {"Batch_Size": B})
        finally:
            torch.randn_like = real
        for k in ("total_vlb", "prior_vlb", "vb", "x_0_mse", "mse"):
            out[f"{name}_{k}"] = r[k].numpy()
    np.savez_compressed(os.path.join(HERE, "vlb_total_kat.npz"), **out)
    print("vlb_total_kat.npz:", len(out), "arrays")


def gen_loss():
    """p_loss (GaussianDiffusion.py:419-434) of the reference with the model output, the noise and t injected: the loss dict,
    the scalar and -- through the reference's own autograd graph -- d(scalar)/d(estimate_noise), for every loss type, with
    uniform draws (`loss_weight="none"`) and the weighted numpy draw ("prop-t")."""
    g = torch.Generator().manual_seed(23)
    B, H = 6, 16
    x0 = torch.rand(B, 1, H, H, generator=g) * 2 - 1
    x0[:, :, 0, :4] = -1.0
    x0[:, :, 1, :4] = 1.0
    eps0 = torch.randn(B, 1, H, H, generator=g)
    eps0[:, :, 2, :3] = 30.0                       # drives pred_x_0 into the clamp (zero gradient through the VLB term there)
    noise = torch.randn(B, 1, H, H, generator=g)
    noise[:, :, 3, :2] = eps0[:, :, 3, :2]         # exact zeros of eps - noise: sign(0) = 0 in the l1 gradient
    out = {"x0": x0.numpy(), "eps": eps0.numpy(), "noise": noise.numpy()}
    real_randint = torch.randint
    for lt in ("l1", "l2", "hybrid"):
        for lw in ("none", "prop-t"):
            d = ref_gd.GaussianDiffusionModel([H, H], ref_gd.get_beta_schedule(1000, "linear"), loss_type=lt, loss_weight=lw,
                                              noise="gauss")
            d.noise_fn = lambda a, b: noise
            eps = eps0.clone().requires_grad_(True)
            t_inj = torch.tensor([0, 1, 2, 500, 998, 0])
            np.random.seed(5)
            torch.randint = lambda *a, **k: t_inj.clone()
            try:
                total, (ld, x_t, e_out) = d.p_loss(lambda a, b: eps, x0, {"train_start": True, "sample_distance": 800})
            finally:
                torch.randint = real_randint
            total.backward()
            tag = f"{lt}_{lw}"
            if lw == "none":
                t_used, w_used = t_inj, torch.ones(B)
            else:
                np.random.seed(5)
                t_used, w_used = d.sample_t_with_weights(B, "cpu")
            out[f"{tag}_t"] = t_used.numpy()
            out[f"{tag}_weights"] = w_used.numpy()
            out[f"{tag}_x_t"] = x_t.detach().numpy()
            out[f"{tag}_loss"] = ld["loss"].detach().numpy()
            if "vlb" in ld:
                out[f"{tag}_vlb"] = ld["vlb"].detach().numpy()
            out[f"{tag}_total"] = np.float32(total.item())
            out[f"{tag}_d_eps"] = eps.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "loss_kat.npz"), **out)
    print("loss_kat.npz:", len(out), "arrays")


def gen_simplex2():
    """2-D OpenSimplex (simplex.py:211-318, 56-73): point values as bit patterns, a coordinate grid, octave fields."""
    out = {}
    rng = np.random.RandomState(202)
    pts = np.concatenate([rng.uniform(-40, 40, (1500, 2)), rng.uniform(-2, 2, (500, 2)),
                          np.array([[0, 0], [0.5, 0.5], [1, 1], [-1.25, 3.5], [510.0, 510.0], [1e-9, -1e-9]])])
    out["points"] = pts
    for seed in (3, 12345, -9999999999):
        s = ref_simplex.Simplex_CLASS()
        s.newSeed(seed)
        out[f"s{seed}_values"] = np.array([s.noise2(x, y) for x, y in pts], dtype=np.float64)
        out[f"s{seed}_grid"] = s.noise2array(np.arange(12) / 3.0 - 1.0, np.arange(12) / 5.0 + 0.25)
        out[f"s{seed}_oct_32_4_07_16"] = s.rand_2d_octaves((32, 32), 4, 0.7, 16)
        out[f"s{seed}_oct_64_6_08_64"] = s.rand_2d_octaves((64, 64), 6, 0.8, 64)
    out["grid_x"], out["grid_y"] = np.arange(12) / 3.0 - 1.0, np.arange(12) / 5.0 + 0.25
    np.savez_compressed(os.path.join(HERE, "simplex2_kat.npz"), **out)
    print("simplex2_kat.npz:", len(out), "arrays")


if __name__ == "__main__":
    which = sys.argv[1:] or ["simplex", "diffusion", "unet", "metrics", "vlb", "vlb_total", "loss", "simplex2", "unet_c5", "unet_c2_batch4", "training",
                             "detection", "loader", "detection_loops"]
    torch.set_num_threads(8)
    if "simplex" in which:
        gen_simplex()
    if "diffusion" in which:
        gen_diffusion()
    if "unet" in which:
        gen_unet()
    if any(w.startswith("unet:") for w in which):
        gen_unet([w.split(":", 1)[1] for w in which if w.startswith("unet:")])
    if "metrics" in which:
        gen_metrics()
    if "vlb" in which:
        gen_vlb()
    if "vlb_total" in which:
        gen_vlb_total()
    if "loss" in which:
        gen_loss()
    if "simplex2" in which:
        gen_simplex2()
    if "unet_c5" in which:
        gen_unet_c5()
    if "unet_c2_batch4" in which:
        gen_unet_c2_batch4()
    if "training" in which:
        gen_training()
    if any(w.startswith("training:") for w in which):
        gen_training([w.split(":", 1)[1] for w in which if w.startswith("training:")])
    if "detection" in which:
        gen_detection()
    if "loader" in which:
        gen_loader()
    if "detection_loops" in which:
        gen_detection_loops()
