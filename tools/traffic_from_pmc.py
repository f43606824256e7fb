"""Turn rocprofv3 PMC summaries (tools/summarize_prof.py counter) into the committed files bench.py reads:

  python tools/traffic_from_pmc.py calib  <calib_fetch_by_kernel.csv> <calib_write_by_kernel.csv> > profiles/rN_hbm_calibration.json
  python tools/traffic_from_pmc.py traffic <calibration.json> <c2_fetch_by_kernel.csv> <c2_write_by_kernel.csv> <per-GPU batch> [tree] > profiles/rN_traffic_c2.json

calib:   FETCH_SIZE / WRITE_SIZE (reported in KB) of tools/hbm_calib.bin, whose kernels move exactly 1 GiB each -> correction
         factors (true bytes / reported bytes) per access pattern.
traffic: per kernel of the bench step, HBM-side bytes per launch = factor * reported, averaged over its launches."""
import csv
import json
import sys
from collections import defaultdict

KNOWN = 1 << 30


def rows(path):
    """summarize_prof.py counter output; kernel names may contain commas, so the six fixed columns are taken from the right."""
    out = []
    with open(path) as f:
        next(f)
        for ln in f:
            parts = ln.rstrip("\n").split(",")
            if len(parts) < 7:
                continue
            out.append({"kernel": ",".join(parts[:-6]), "grid_threads": parts[-6], "wg_threads": parts[-5], "counter": parts[-4],
                        "launches": parts[-3], "avg_value": parts[-2], "total_value": parts[-1]})
    return out


def calib(fetch_csv, write_csv):
    out = {"known_bytes_per_kernel": KNOWN, "unit_of_reported_values": "KB (1024 B)", "kernels": {}}
    for r in rows(fetch_csv):
        if r["kernel"].startswith("read_"):
            rep = float(r["avg_value"]) * 1024
            out["kernels"][r["kernel"]] = {"counter": "FETCH_SIZE", "reported_bytes": rep, "factor": KNOWN / rep}
    for r in rows(write_csv):
        if r["kernel"].startswith("write_"):
            rep = float(r["avg_value"]) * 1024
            out["kernels"][r["kernel"]] = {"counter": "WRITE_SIZE", "reported_bytes": rep, "factor": KNOWN / rep}
    k = out["kernels"]
    # the patch pattern is what the convolution kernels do; stream16 is the guide's own calibration point
    # FETCH_SIZE reports 1/2 of the bytes of 16-byte-per-lane reads (stream16: exactly; the 64-byte-per-pixel patch pattern of the
    # convolution kernels: 0.513 / 0.504 -- the excess over 1/2 is real re-fetching of lines whose two halves are read in different
    # loop iterations); WRITE_SIZE is exact for both store patterns.  Factors applied to the step's counters:
    out["fetch_factor"] = round(k["read_stream16"]["factor"], 3) if "read_stream16" in k else 2.0
    out["patch_pattern_refetch"] = {n: round(2.0 / v["factor"], 4) for n, v in k.items() if n.startswith("read_patch")}
    out["write_factor"] = round(k["write_pixel4"]["factor"], 3) if "write_pixel4" in k else 1.0
    return out


def traffic(cal_json, fetch_csv, write_csv):
    cal = json.load(open(cal_json))
    ff, wf = cal["fetch_factor"], cal["write_factor"]
    agg = defaultdict(lambda: {"launches": 0, "fetch": 0.0, "write": 0.0})
    for path, key, fac in ((fetch_csv, "fetch", ff), (write_csv, "write", wf)):
        for r in rows(path):
            a = agg[r["kernel"]]
            n = int(r["launches"])
            if key == "fetch":
                a["launches"] += n
            a[key] += float(r["total_value"]) * 1024 * fac
    out = {"calibration": {"fetch_factor": ff, "write_factor": wf, "source": cal_json}, "kernels": {}}
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"])):
        n = max(a["launches"], 1)
        out["kernels"][k] = {"launches": a["launches"], "fetch_bytes_per_launch": a["fetch"] / n, "write_bytes_per_launch": a["write"] / n,
                             "bytes_per_launch": (a["fetch"] + a["write"]) / n}
    return out


if __name__ == "__main__":
    if sys.argv[1] == "calib":
        print(json.dumps(calib(sys.argv[2], sys.argv[3]), indent=1))
    else:
        d = traffic(sys.argv[2], sys.argv[3], sys.argv[4])
        d["per_gpu_batch"] = int(sys.argv[5]) if len(sys.argv) > 5 else None
        d["tree"] = sys.argv[6] if len(sys.argv) > 6 else None
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        d["sources_hash"] = bench.tree_hash()          # bench.py flags the file as stale when the kernels / plans change
        d["class_hashes"] = {c: bench.class_hash(c) for c in bench.CLASS_SOURCES}    # ... per kernel class (round 6)
        d["command"] = "ANODDPM_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof"
        print(json.dumps(d, indent=1))
