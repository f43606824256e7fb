#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace as a timeline: python tools/timeline.py kernel_trace.csv [step_index] [out.csv]
A step is delimited by the chain_advance kernel that ends it.  Prints per kernel: start offset, duration, gap to the previous
kernel's end; and totals of busy time / gaps per kernel name.  Used to see where the small-map section of the reverse step
loses time (kernel duration vs. dependent-launch gaps)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:48].replace(", ", ";")


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
step = int(sys.argv[2]) if len(sys.argv) > 2 else -2
ends = [i for i, r in enumerate(rows) if "chain_advance" in r["Kernel_Name"]]
if step < 0:
    step += len(ends)
lo = ends[step - 1] + 1 if step > 0 else 0
hi = ends[step] + 1
seg = rows[lo:hi]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
out = []
busy = defaultdict(float)
gaps = defaultdict(float)
cnt = defaultdict(int)
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = short(r["Kernel_Name"])
    gap = (s - prev_end) / 1e3
    out.append((k, (s - t0) / 1e3, (e - s) / 1e3, gap, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"]))
    busy[k] += (e - s) / 1e3
    gaps[k] += gap
    cnt[k] += 1
    prev_end = max(prev_end, e)
total = (prev_end - t0) / 1e3
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as f:
        f.write("kernel,start_us,dur_us,gap_before_us,grid_x,grid_y,grid_z,wg\n")
        for o in out:
            f.write(f"{o[0]},{o[1]:.2f},{o[2]:.2f},{o[3]:.2f},{o[4]},{o[5]},{o[6]},{o[7]}\n")
print(f"step {step}: {len(seg)} kernels, span {total:.1f} us, busy {sum(busy.values()):.1f} us, gaps {sum(gaps.values()):.1f} us")
print("kernel,launches,busy_us,avg_us,gap_before_total_us,avg_gap_us")
for k in sorted(busy, key=lambda k: -busy[k]):
    print(f"{k},{cnt[k]},{busy[k]:.1f},{busy[k] / cnt[k]:.2f},{gaps[k]:.1f},{gaps[k] / cnt[k]:.2f}")
