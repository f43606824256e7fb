"""-m gpu: bench.py's output contract (one JSON line with the fields the driver and the judge read), on the small
config so it runs in seconds; and __graft_entry__.smoke()."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", *extra],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_small_config():
    d = run_bench("--config", "c1", "--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["config"]["output_finite"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] == "TFLOP/s" and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] <= 1.0
    assert r["algorithmic_tflops"] >= r["achieved"] - 1e-9
    assert r["traffic"] is None                                   # no committed PMC pass for this (test-sized) configuration
    # the memory-bound kernels against the HBM roofline: stem, head, resample, stem statistics, the fused reverse update
    names = " ".join(k["kernel"] for k in r["hbm_kernels"])
    assert all(n in names for n in ("conv_stem", "conv_head", "resample2x", "p_update"))
    assert all(0 < k["frac_of_8TBps"] < 1 and abs(k["frac_of_8TBps"] - k["GBps"] / 8000.0) < 1e-9 for k in r["hbm_kernels"])
    assert all(abs(k["frac_of_6.3TBps_achievable"] - k["GBps"] / 6300.0) < 1e-9 for k in r["hbm_kernels"])
    pu = [k for k in r["hbm_kernels"] if "p_update" in k["kernel"]][0]
    assert pu["launches_per_step"] == 1.0                           # timed inside the instrumented steps, one launch per step
    assert r["kernel_launches_per_step"] > 50 and r["class_ms_per_step"]["simplex"] > 0 and r["class_ms_per_step"]["p_update"] > 0


def test_bench_cpu_baseline_object():
    d = run_bench("--config", "c1")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "images/s" and c["value"] > 0 and "sample" in c
    assert d["value"] > c["value"]
    # the host the baseline ran on
    assert c["host_cpu_count"] >= c["cores"] and c["host_cpus_available"] >= 1 and isinstance(c["cpu_model"], (str, type(None)))
    # work outside the timed region is reported: the chain's pre-drawn simplex tables
    assert d["config"]["table_setup_ms"] > 0 and d["config"]["table_setup_us_per_step"] > 0


def test_bench_line_simplex_and_training_configs():
    d = run_bench("--config", "c4", "--no-cpu-baseline")
    assert d["unit"] == "voxels/s" and d["dtype"] == "f64" and d["value"] > 1e9
    r = d["roofline"]
    assert r["bound"] == "fp64-alu" and 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    d = run_bench("--config", "c3", "--batch", "1", "--no-cpu-baseline")
    assert d["unit"] == "images/s" and d["config"]["loss_finite"] is True and d["value"] > 0
    r = d["roofline"]
    # the dominant contraction class of the training step (forward + data-gradient F(4x4,3x3), or one of the weight-gradient
    # kernels) priced by the FLOPs the matrix pipe EXECUTES: a pipe utilisation, never above 1
    # (at batch 1 the small-map GEMMs can be the largest class, at batch 4 it is the F(4x4,3x3) kernel)
    assert any(k in r["kernel"] for k in ("wino", "wgrad", "igemm")) and 0 < r["frac"] <= 1.0 and r["launches_per_step"] > 20
    kernels = [r["kernel"]] + [k["kernel"] for k in r["other_contraction_kernels"]]
    assert any("wgrad43_kernel" in k for k in kernels) and all(0 < k["achieved"] <= r["peak"] for k in r["other_contraction_kernels"])


def test_bench_line_detection_sweep():
    """`--config det`: one image's detection_B sweep end to end (every (t_distance, avg) chain slot-batched + on-device anomaly
    maps), a side line.  Shortened sweep here: range(50, 150, 50) x 5 chains."""
    d = run_bench("--config", "det", "--no-cpu-baseline", "--det-end", "150")
    assert d["unit"] == "chain-steps/s" and d["value"] > 0 and d["config"]["output_finite"] is True
    cf = d["config"]
    assert cf["chains_per_setting"] == 5 and cf["settings"] == [50, 100] and cf["chains"] == 10 and cf["chain_steps_per_image"] == 750
    assert cf["slots"] * cf["batched_steps_per_image"] >= 750 and 0 < cf["slot_utilisation"] <= 1
    assert abs(d["value"] - 750 / (d["ms_per_step"] / 1000.0)) < 1e-6 * d["value"]
    # round 6: the line says what a batched step of the sweep is made of (the plan at batch = slots): class table + the dominant
    # contraction class against the fp32 matrix peak
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["batch"] == cf["slots"] and 0 < r["frac"] <= 1.0 and cf["metric_version"] == 2
    assert r["class_ms_per_step"]["winograd_f43"] > 0 and "peak_at_sustained_clock" in r
    d = run_bench("--config", "c4", "--no-cpu-baseline", "--steps", "3", "--warmup", "1")
    assert abs(d["roofline"]["peak_no_fma"] - d["roofline"]["peak"] / 2) < 1e-9 and 0 < d["roofline"]["frac_of_peak_no_fma"] <= 1.0


def test_bench_under_torchrun_one_rank_uses_rccl():
    """The driver launches N > 1 as `python -m torch.distributed.run ... bench.py --gpus N`; with one GPU here the same
    launcher at --nproc-per-node 1 plus ANODDPM_BENCH_FORCE_DIST=1 executes the process-group / barrier / all-reduce path."""
    env = dict(os.environ, ANODDPM_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29544", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "3", "--warmup", "1", "--config", "c1", "--no-prof"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "cpu_baseline" in d


def test_smoke_entry():
    import importlib
    sys.path.insert(0, ROOT)
    g = importlib.import_module("__graft_entry__")
    g.smoke()
