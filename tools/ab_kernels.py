#!/usr/bin/env python3
"""A/B timing of builds of libmi_phylo.so on the headline step, interleaved in ONE process
chain on ONE box (kernel times differ by 2-3 % from box to box and run to run: only
back-to-back alternation tells a 2 % change from noise).

  python tools/ab_kernels.py [--rounds 3] [--bench-args "..."] name=path.so[,ENV=value...] ...

Each variant is a library path (or "default") plus optional environment settings; every round
runs bench.py once per variant; the table gives min / median of the dominant kernel's launch
time and of the step."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--bench-args", default="--steps 20 --warmup 3 --headline-only")
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    variants = []
    for v in args.variants:
        name, _, rest = v.partition("=")
        parts = rest.split(",") if rest else ["default"]
        env = {}
        if parts[0] != "default":
            env["MI_PHYLO_LIBRARY"] = os.path.abspath(parts[0])
        for kv in parts[1:]:
            k, _, val = kv.partition("=")
            env[k] = val
        variants.append((name, env))
    res = {name: {"kernel": [], "step": []} for name, _ in variants}
    for _ in range(args.rounds):
        for name, env in variants:
            r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args.bench_args.split(),
                               env=dict(os.environ, **env), capture_output=True, text=True)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not lines:
                print(name, "FAILED", r.stdout[-500:], r.stderr[-1500:])
                continue
            d = json.loads(lines[-1])
            res[name]["kernel"].append(d["step_ms_device"]["kernel_median"])
            res[name]["step"].append(d["step_ms_device"]["median"])
            res[name]["name"] = d["roofline"]["kernel"]
    for name, _ in variants:
        k, s = res[name]["kernel"], res[name]["step"]
        if k:
            print(f"{name:24s} {res[name].get('name', ''):24s} kernel min {min(k):.4f} median {np.median(k):.4f} ms | "
                  f"step min {min(s):.4f} median {np.median(s):.4f} ms   ({len(k)} runs)")


if __name__ == "__main__":
    main()
