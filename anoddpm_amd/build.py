"""Build libanoddpm_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

In-tree output (anoddpm_amd/lib/libanoddpm_hip.so) so that it travels to the GPU box with the
repo snapshot.  Per-file objects are cached by mtime; simplex/diffusion are compiled with
-ffp-contract=off because their results must be bit-identical to the reference's arithmetic.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
SO = os.path.join(LIBDIR, "libanoddpm_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "anoddpm_hip.h")

SOURCES = {
    "simplex.hip": ["-ffp-contract=off"],
    "diffusion.hip": ["-ffp-contract=off"],
    "unet_kernels.hip": [],
    "igemm.hip": [],
    "winograd.hip": [],
    "winograd43.hip": [],
    "winograd43r.hip": [],
    "winograd43w.hip": [],
    "winograd43b.hip": [],
    "pointwise.hip": [],
    "smallmap.hip": [],
    "wino23s.hip": [],
    "attention.hip": [],
    "executor.hip": [],
    "optim.hip": [],
    "metrics.hip": ["-ffp-contract=off"],
    "wgrad.hip": [],
    "wgrad43.hip": [],
    "gn_backward.hip": [],
    "train_kernels.hip": [],
    "loader.hip": ["-ffp-contract=off"],
}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False, verbose=False):
    global OBJDIR, SO
    # measurement builds beside the product library: ANODDPM_BUILD_TAG=<tag> writes lib/libanoddpm_hip_<tag>.so (objects in
    # lib/obj_<tag>); ANODDPM_EXTRA_FLAGS="file.hip:-flag1,-flag2;other.hip:-flag" appends per-file compiler flags.  Loaded with
    # ANODDPM_LIB_TAG=<tag> (_lib.py).  The product build (no tag) ignores ANODDPM_EXTRA_FLAGS.
    tag = os.environ.get("ANODDPM_BUILD_TAG", "")
    extra_flags = {}
    if tag:
        OBJDIR = os.path.join(LIBDIR, "obj_" + tag)
        SO = os.path.join(LIBDIR, "libanoddpm_hip_%s.so" % tag)
        for item in filter(None, os.environ.get("ANODDPM_EXTRA_FLAGS", "").split(";")):
            f, _, fl = item.partition(":")
            extra_flags[f.strip()] = [x for x in fl.split(",") if x]
    os.makedirs(OBJDIR, exist_ok=True)
    # ANODDPM_ABLATE=1: measurement build that also contains the timing ablations (kernels that skip work and produce wrong
    # results by design; selected through anoddpm_internal_variant / ANODDPM_DEBUGn).  The product build has none of them.
    flags = list(COMMON) + (["-DANODDPM_ABLATE"] if os.environ.get("ANODDPM_ABLATE", "0") == "1" else [])
    stamp = os.path.join(OBJDIR, "flags.txt")
    stamp_text = " ".join(flags) + "".join(f" | {k}: {' '.join(v)}" for k, v in sorted(extra_flags.items()))
    if not os.path.exists(stamp) or open(stamp).read() != stamp_text:
        force = True
        if os.path.exists(stamp):
            os.remove(stamp)                     # re-written only after EVERY object compiled with the new flags
    # every header under csrc/ is a dependency of every object (simplex_tables.h and pack_items.h carry bit-exact tables)
    deps = [HEADER, os.path.abspath(__file__)] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    dep_m = max(os.path.getmtime(d) for d in deps)
    objs, rebuilt = [], False
    procs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), dep_m):
            cmd = [hipcc(), *flags, *extra, *extra_flags.get(src, []), "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            rebuilt = True
    failed = []
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
            o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
            if os.path.exists(o):
                os.remove(o)                     # never leave an object of the previous flag set behind a failure
        elif verbose and out.strip():
            print(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("\n".join(failed))
    with open(stamp, "w") as f:
        f.write(stamp_text)
    if rebuilt or not os.path.exists(SO):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs]
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
