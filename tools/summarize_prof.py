"""Aggregate rocprofv3 CSV output into the small summaries kept under profiles/.

  python tools/summarize_prof.py trace   <kernel_trace.csv>        -> per (kernel, grid) launch count / avg us
  python tools/summarize_prof.py counter <counter_collection.csv>  -> per (kernel, grid) average counter value

Kernel names are shortened to the part before the argument list."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:60].replace(", ", ";")


def main():
    mode, path = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(path)))
    agg = defaultdict(lambda: [0, 0.0])
    if mode == "trace":
        for r in rows:
            key = (short(r["Kernel_Name"]), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
            a = agg[key]
            a[0] += 1
            a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print("kernel,grid_x,grid_y,grid_z,wg_x,launches,avg_us,total_us")
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(",".join(k) + f",{n},{t / n:.1f},{t:.0f}")
    else:
        for r in rows:
            key = (short(r["Kernel_Name"]), r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"])
            a = agg[key]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        print("kernel,grid_threads,wg_threads,counter,launches,avg_value,total_value")
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(",".join(k) + f",{n},{t / n:.1f},{t:.0f}")


if __name__ == "__main__":
    main()
