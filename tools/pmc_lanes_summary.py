#!/usr/bin/env python3
"""Summarise the counter pass of tools/pmc_util.sh into profiles/rNN_<tag>_pmc_lanes.json:

    python tools/pmc_lanes_summary.py gpurun_out/prof_util <out.json>

per kernel, SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = the share of a wave's 64 lanes that take part in a
vector instruction, and the vector instructions per wave."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_sq_summary import load, tree_tag


def main():
    root, out = sys.argv[1], sys.argv[2]
    k = load(root)
    res = {"command": "rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -- python3 bench.py "
                      "--steps 1 --warmup 0 --no-cpu --no-mjpeg --no-workloads --one-stream --host-synth 4 (tools/pmc_util.sh)"}
    res.update(tree_tag())
    res["kernels"] = {}
    for name, c in sorted(k.items()):
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_WAVES"):
            res["kernels"][name] = {
                "lanes_per_valu_inst": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 3),
                "valu_insts_per_wave": round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)}
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
        f.write("\n")
    for name, v in res["kernels"].items():
        print("%-20s %s" % (name, v))


if __name__ == "__main__":
    main()
