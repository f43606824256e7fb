#!/usr/bin/env python3
"""Basic-block instruction table of one kernel in a hipcc assembly listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math --cuda-device-only -S -Iinclude anoddpm_amd/csrc/X.hip -o /tmp/X.s
    python tools/isa_blocks.py /tmp/X.s <substring of the mangled kernel name> [min_instructions]

Per basic block (label to label): MFMA, other VALU (of which packed f32 and transcendental), SALU, LDS reads / writes, buffer /
global loads / stores, s_waitcnt (vmcnt / lgkmcnt separately), s_barrier.  Runs of MFMA-free instructions inside a block are
reported as `longest MFMA-free run`: the matrix pipe idles for at least that many issue slots unless another wave covers it."""
import re
import sys


def classify(op):
    for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
        if op.endswith(suf):
            op = op[:-len(suf)]
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "vpk"
    if op in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f64", "v_rsq_f64",
              "v_rcp_iflag_f32"):
        return "vtrans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ldsr"
    if op.startswith("ds_"):
        return "ldsw"
    if op.startswith("buffer_load") or op.startswith("global_load") or op.startswith("flat_load") or op.startswith("scratch_load"):
        return "vmr"
    if op.startswith("buffer_store") or op.startswith("global_store") or op.startswith("flat_store") or op.startswith("scratch_store") \
            or op.startswith("buffer_atomic") or op.startswith("global_atomic"):
        return "vmw"
    if op == "s_waitcnt":
        return "wait"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^([A-Za-z_][\w$.]*):", l)
        if m and key in m.group(1):
            start = i
            name = m.group(1)
            break
    if start is None:
        sys.exit(f"no kernel matching {key!r}")
    blocks, cur, label = [], [], "entry"
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        op = s.split()[0]
        cur.append((op, s))
    blocks.append((label, cur))
    print(f"kernel {name}")
    cols = ["mfma", "valu", "vpk", "vtrans", "salu", "ldsr", "ldsw", "vmr", "vmw", "smem", "barrier"]
    print(f"{'block':>12} {'n':>5} " + " ".join(f"{c:>6}" for c in cols) + "  waits(vm/lgkm)  longest-MFMA-free-run  backedge")
    tot = {c: 0 for c in cols}
    for label, ins in blocks:
        cnt = {c: 0 for c in cols}
        wv = wl = 0
        run = best = 0
        back = ""
        for op, s in ins:
            c = classify(op)
            if c in cnt:
                cnt[c] += 1
            if c == "wait":
                if "vmcnt" in s:
                    wv += 1
                if "lgkmcnt" in s:
                    wl += 1
            if c == "mfma":
                run = 0
            else:
                run += 1
                best = max(best, run)
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = s.split()[-1]
                if tgt == label:
                    back = "LOOP"
        for c in cols:
            tot[c] += cnt[c]
        if len(ins) >= min_n:
            print(f"{label:>12} {len(ins):>5} " + " ".join(f"{cnt[c]:>6}" for c in cols) + f"  {wv:>5}/{wl:<5}  {best:>8}  {back}")
    print(f"{'total':>12} {sum(len(i) for _, i in blocks):>5} " + " ".join(f"{tot[c]:>6}" for c in cols))


if __name__ == "__main__":
    main()
