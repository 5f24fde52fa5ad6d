#!/usr/bin/env python3
"""Summarise the SQ counter passes of tools/profile.sh into profiles/rNN_<tag>_pmc_sq.json.

    python tools/pmc_sq_summary.py gpurun_out/prof_<tag> <frames per launch> <out.json>

Per kernel (mean over its launches): the raw counters, and a few ratios read off them --
instructions per wave, the share of wave time in which a vector instruction issues, the share spent
waiting, LDS bank-conflict share, waves per launch.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles, summed over all waves (MI355X_MICROARCH.md, cycle constants)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(dirname):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = re.sub(r"^void ", "", row["Kernel_Name"])
                name = re.split(r"[<(]", name)[0]
                if name.startswith("k_"):
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def tree_tag():
    """what the summary is tagged with: git HEAD and a hash of the kernel sources (bench.py compares the latter with
    the tree it runs on and says so loudly when a kernel has changed since the counters were collected)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    try:
        head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], text=True).strip()
        dirty = bool(subprocess.check_output(["git", "-C", root, "status", "--porcelain", "jpeg_amd/csrc"], text=True).strip())
    except Exception:
        head, dirty = None, None
    return {"head": head, "csrc_dirty": dirty, "csrc_sha16": bench.csrc_sha16()}


def main():
    root, frames, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    merged = defaultdict(dict)
    for sub in ("sq_insts", "sq_cycles", "sq_lds"):
        for k, d in load(os.path.join(root, sub)).items():
            merged[k].update(d)
    res = {}
    for k, c in sorted(merged.items()):
        r = {n: round(v, 1) for n, v in sorted(c.items())}
        waves = c.get("SQ_WAVES", 0.0)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        d = {}
        if waves:
            for n in ("VALU", "SALU", "LDS", "VMEM_RD", "VMEM_WR", "SMEM"):
                if "SQ_INSTS_" + n in c:
                    d[n.lower() + "_insts_per_wave"] = round(c["SQ_INSTS_" + n] / waves, 1)
        if wc:
            for n, key in (("valu_issue_share", "SQ_ACTIVE_INST_VALU"), ("lds_issue_share", "SQ_ACTIVE_INST_LDS"),
                           ("any_issue_share", "SQ_ACTIVE_INST_ANY"), ("wait_any_share", "SQ_WAIT_ANY"),
                           ("wait_inst_any_share", "SQ_WAIT_INST_ANY"), ("wait_inst_lds_share", "SQ_WAIT_INST_LDS")):
                if key in c:
                    d[n] = round(c[key] / wc, 4)
            if c.get("SQ_BUSY_CYCLES"):
                d["waves_in_flight_per_simd_avg"] = round(wc / c["SQ_BUSY_CYCLES"], 2)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        res[k] = {"counters": r, "derived": d}
    json.dump({"command": "rocprofv3 --pmc <SQ counters, three passes: tools/profile.sh> -- python3 bench.py --steps 1 --warmup 0 "
                          "--no-cpu --no-mjpeg --no-workloads --one-stream", "frames_per_launch": frames, "kernels": res, **tree_tag()},
              open(out, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        print("%-20s %s" % (k, json.dumps(v["derived"])))


if __name__ == "__main__":
    main()
