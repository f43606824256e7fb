#!/usr/bin/env python3
"""Per-layer HIP-event time of the Winograd-domain weight-gradient ops INSIDE the config-3 training step (executor profiler,
launch order matched to the plan's backward op list):  ANODDPM_LIB_TAG=<tag> python tools/wgrad_by_layer.py [steps]"""
import argparse
import ctypes
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from anoddpm_amd import _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
args = argparse.Namespace(gpus=1, steps=steps, warmup=2, batch=0, no_prof=True, config="c3")
c = bench.Ctx()
c.world, c.rank, c.local_rank, c.dist, c.shared, c.dev = 1, 0, 0, None, False, torch.device("cuda", 0)
torch.cuda.set_device(0)
hold = {}
orig = bench.timed


def timed(c_, a_, step_fn):                       # keep the step closure: the instrumented pass below re-runs it
    hold["step"] = step_fn
    return orig(c_, a_, step_fn)


bench.timed = timed
out, _, _ = bench.run_train(c, args, dict(bench.CONFIGS["c3"]))
print("c3 step ms", round(out["ms_per_step"], 3))
L = _lib.lib()
import gc
plan = None
for o in gc.get_objects():
    if type(o).__name__ == "TrainPlan":
        plan = o
wg = [st for code, st in plan.bops if code == _lib.OP_WGRAD3 and st.algo == 1]
L.anoddpm_prof_enable(1)
torch.cuda.synchronize()
for _ in range(steps):
    hold["step"]()
torch.cuda.synchronize()
ms = (ctypes.c_double * _lib.OP_MAX)()
cnt = (ctypes.c_int64 * _lib.OP_MAX)()
L.anoddpm_prof_collect(ms, cnt)
n = L.anoddpm_prof_list(None, None, 0)
codes = (ctypes.c_int32 * n)()
msv = (ctypes.c_float * n)()
L.anoddpm_prof_list(codes, msv, n)
L.anoddpm_prof_enable(0)
mine = [msv[i] for i in range(n) if codes[i] == 15]
assert len(mine) == len(wg) * steps, (len(mine), len(wg), steps)
agg = defaultdict(lambda: [0, 0.0])
for s in range(steps):
    for i, st in enumerate(wg):
        a = agg[(st.H, st.c0 + st.c1, st.N, st.a_mode, int(bool(st.colsum)))]
        a[0] += 1
        a[1] += 1000.0 * mine[s * len(wg) + i]
print("H,K,N,a_mode,colsum,launches_per_step,avg_us (kernel + fold),total_us_per_step")
tot = 0.0
for k, (m, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(",".join(str(v) for v in k) + f",{m // steps},{us / m:.1f},{us / steps:.0f}")
    tot += us / steps
print("total_us_per_step", round(tot))
