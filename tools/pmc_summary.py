#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/rNN_x_pmc_traffic.json.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <frames> <out.json>

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters
are in KB; FETCH_SIZE is doubled.  Values are per launch (mean over the launches of a kernel).
"""
import csv, json, re, sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = re.sub(r"^void ", "", row["Kernel_Name"])
            name = re.split(r"[<(]", name)[0]
            acc[name].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


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
    fetch_csv, write_csv, frames, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fe, wr = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith("k_"):
            continue
        fkb, wkb = fe.get(k, 0.0), wr.get(k, 0.0)
        fb, wb = int(round(fkb * 1024 * 2)), int(round(wkb * 1024))
        kernels[k] = {"fetch_size_kb_raw": round(fkb, 1), "write_size_kb_raw": round(wkb, 1),
                      "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb}
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py "
                          "--steps 1 --warmup 0 --no-cpu --no-mjpeg --frames %d" % frames,
               "frames_per_launch": frames, "kernels": kernels, **tree_tag()}, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes"]):
        print("%-20s fetch %12d  write %12d" % (k, v["fetch_bytes"], v["write_bytes"]))


if __name__ == "__main__":
    main()
