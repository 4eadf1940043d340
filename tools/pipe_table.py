#!/usr/bin/env python3
"""profiles/pipe.json from the committed SQ-counter tables (profiles/rNN_*sq_counters*.txt, made
by tools/sq_counters.sh / tools/profile_round.sh with rocprofv3 --pmc): per kernel, how busy the
SIMDs' vector pipe was over the launch --

    pipe_busy = (SQ_VALU_MFMA_BUSY_CYCLES + 4 x (SQ_INSTS_VALU - SQ_INSTS_MFMA)) / (32 x SQ_BUSY_CYCLES)

An FP64 matrix instruction occupies the pipe for 16 (4x4x4) or 64 (16x16x4) clocks and nothing
co-issues with it (tools/micro/coissue*.hip); another vector instruction takes 4.
SQ_BUSY_CYCLES is summed over the chip's 32 shader engines, and each of the 256 CUs has four
SIMDs: 32 x SQ_BUSY_CYCLES = SIMD cycles of the launch.  bench.py prints the figure as
`roofline.pipe_busy` (static: from these files, not from the timed run).

  python tools/pipe_table.py [round-tag ...]      (default: every round, later rounds win)"""
import glob
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def blocks(path):
    name, vals = None, {}
    for ln in open(path):
        if not ln.strip():
            continue
        if not ln.startswith(" "):
            if name and vals:
                yield name, vals
            name, vals = ln.strip(), {}
            continue
        m = re.match(r"\s+(SQ_[A-Z_]+)\s+([0-9.eE+]+)", ln)
        if m:
            vals[m.group(1)] = float(m.group(2))
    if name and vals:
        yield name, vals


def main():
    tags = sys.argv[1:]
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*sq_counters*.txt")))
    table = {}
    for f in files:
        tag = os.path.basename(f).split("_")[0]
        if tags and tag not in tags:
            continue
        for name, v in blocks(f):
            need = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES", "SQ_WAVES")
            if not all(k in v for k in need):
                continue
            other = v["SQ_INSTS_VALU"] - v["SQ_INSTS_MFMA"]
            simd_cycles = 32.0 * v["SQ_BUSY_CYCLES"]
            key = re.sub(r"^void |miphylo::|\(anonymous namespace\)::|\(miphylo::LikArgs.*$", "", name).strip()
            table[key] = {
                "pipe_busy": (v["SQ_VALU_MFMA_BUSY_CYCLES"] + 4.0 * other) / simd_cycles,
                "mfma_busy": v["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
                "mfma_per_wave": v["SQ_INSTS_MFMA"] / v["SQ_WAVES"],
                "other_valu_per_wave": other / v["SQ_WAVES"],
                "vmem_per_wave": (v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0)) / v["SQ_WAVES"],
                "waves": v["SQ_WAVES"],
                "wait_any_over_wave_cycles": (v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]) if v.get("SQ_WAVE_CYCLES") else None,
                "source": "profiles/" + os.path.basename(f), "round": tag,
            }
    out = os.path.join(REPO, "profiles", "pipe.json")
    json.dump(table, open(out, "w"), indent=1, sort_keys=True)
    for k, e in sorted(table.items()):
        print("%-72s pipe_busy %.3f (matrix %.3f)  %s" % (k[:72], e["pipe_busy"], e["mfma_busy"], e["source"]))


if __name__ == "__main__":
    main()
