#!/usr/bin/env python3
"""Join `bench.py --dump-plan` (the igemm launch list of one step, in order) with a rocprofv3 kernel trace CSV of eager launches
(ANODDPM_NO_GRAPH=1): python tools/by_layer.py plan.json kernel_trace.csv [steps_to_skip] -> per layer class: shape, config, us, TFLOP/s.
Launches are matched by order: every contraction launch of the plan is one igemm_kernel / wino_kernel / wino43_kernel / wino43r_kernel / pointwise_stream_kernel dispatch."""
import csv
import json
import sys
from collections import defaultdict

plan = json.load(open(sys.argv[1]))
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = [r for r in csv.DictReader(open(sys.argv[2]))
        if any(k in r["Kernel_Name"] for k in ("igemm_kernel", "wino_kernel", "wino43_kernel", "wino43r_kernel", "pointwise_stream_kernel"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(plan)
steps = len(rows) // n
assert steps * n == len(rows), (len(rows), n)
agg = defaultdict(lambda: [0, 0.0, 0.0])
for s in range(skip, steps):
    for i, e in enumerate(plan):
        r = rows[s * n + i]
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (e["kind"], e["H"], e["K"], e["N"], e["ks"], e["a_mode"], e["cfg"], e["ksplit"])
        a = agg[key]
        a[0] += 1
        a[1] += us
        a[2] = e["gflop"]
nst = steps - skip
print("kind,H,K,N,ks,a_mode,cfg,ksplit,launches_per_step,avg_us,algorithmic_TFLOPs,executed_TFLOPs,total_us_per_step")
for k, (cnt, us, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    avg = us / cnt
    ex = {2: 4.0 / 9.0, 3: 0.25}.get(k[6], 1.0)
    print(",".join(str(v) for v in k) + f",{cnt // nst},{avg:.1f},{gf / avg * 1e3:.1f},{gf * ex / avg * 1e3:.1f},{us / nst:.0f}")
