#!/usr/bin/env python3
"""Site-pattern compression: device path (mi_site_pattern_compress) against the CPU
restatement, on a C5-shaped alignment (512 taxa x 50 000 columns by default).
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--taxa", type=int, default=512)
    ap.add_argument("--sites", type=int, default=50000)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import libsbn_amd as L
    import oracle_lib as O
    rng = np.random.default_rng(7)
    n, S = args.taxa, args.sites
    # a random tree-like alignment: columns mutate from a few ancestors, ~60 % distinct
    base = rng.integers(0, 4, size=(n, S // 3 + 1)).astype(np.int8)
    codes = base[:, rng.integers(0, base.shape[1], size=S)]
    flip = rng.random(size=codes.shape) < 0.002
    codes = np.where(flip, rng.integers(0, 5, size=codes.shape), codes).astype(np.int8)
    L.site_pattern_compress_device(codes)  # warm-up (module load, allocator)
    best, ms = 1e9, 0.0
    for _ in range(args.reps):
        t0 = time.perf_counter()
        pats, w, kms = L.site_pattern_compress_device(codes)
        dt = time.perf_counter() - t0
        if dt < best:
            best, ms = dt, kms
    # the device-resident door: patterns and weights stay in HBM and feed the engine directly
    best_dev, best_dev_engine = 1e9, 1e9
    spec = L.PhyloModelSpecification("JC69", "weibull+4", "strict")
    for _ in range(args.reps):
        t0 = time.perf_counter()
        dp, _ = L.site_pattern_compress_device(codes, keep_on_device=True)
        t1 = time.perf_counter()
        eng = L.Engine(spec, None, None, device_tips=dp.as_device_tips())
        t2 = time.perf_counter()
        best_dev, best_dev_engine = min(best_dev, t1 - t0), min(best_dev_engine, t2 - t0)
        eng.close()
        dp.release()
    t0 = time.perf_counter()
    eng = L.Engine(spec, pats, w)
    host_engine = time.perf_counter() - t0
    eng.close()
    rows = ["".join("ACGT-"[c] for c in row) for row in codes]
    t0 = time.perf_counter()
    op, ow = O.site_pattern_compress(rows)
    cpu = time.perf_counter() - t0
    assert np.array_equal(pats, op) and np.array_equal(w, ow)
    nbytes = n * S
    print(json.dumps({
        "workload": f"{n} taxa x {S} columns, {pats.shape[1]} distinct patterns",
        "device_end_to_end_ms": best * 1e3, "cpu_oracle_ms": cpu * 1e3,
        "device_resident_ms": best_dev * 1e3,
        "device_resident_plus_engine_ms": best_dev_engine * 1e3,
        "host_arrays_plus_engine_ms": (best + host_engine) * 1e3,
        "hash_kernel_ms": ms, "hash_kernel_GBps": nbytes / (ms * 1e-3) / 1e9 if ms else None,
        "hbm_peak_GBps": 8000.0, "identical_to_cpu": True}))


if __name__ == "__main__":
    main()
