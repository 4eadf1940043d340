#!/bin/bash
# Per-wave instruction mix and wait shares of the walk kernels (rocprofv3 PMC, three passes of
# eight SQ counters each -- counters in their own runs, --kernel-trace only, as the pool's
# gpurun requires).  Run on the GPU box from the repo root:
#   bash tools/sq_counters.sh <out-file> [bench.py arguments...]
# Writes the per-kernel table the profiles/rNN_sq_counters_*.txt files hold.
out=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_MFMA" "SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rm -rf gpurun_out/sq_pass$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/sq_pass$i -- python3 bench.py --steps 3 --warmup 1 --headline-only "$@" > gpurun_out/sq_pass$i.log 2>&1
done
mkdir -p "$(dirname "$out")"
python3 - "$out" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2, 3):
    for f in glob.glob("gpurun_out/sq_pass%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(s in k for s in ("gradient_walk", "gradient_mfma", "loglik_mfma")) and int(r["Grid_Size"]) > 64 * 4000:
                agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1], "w") as fh:
    for k, v in agg.items():
        d = {c: sum(x) / len(x) for c, x in v.items()}
        # SQ_JOBS: (evaluation, tile) jobs of the launch -- since round 3 a wave of the gradient
        # walk takes several tiles, so "per wave" is no longer "per tile"; default: per wave
        import os
        w = float(os.environ.get("SQ_JOBS", 0)) or d["SQ_WAVES"]
        unit = "per tile job" if os.environ.get("SQ_JOBS") else "per wave"
        lines = [k] + ["   %-26s %16.0f  %s %12.1f" % (c, x, unit, x / w) for c, x in sorted(d.items())]
        if "SQ_INSTS_VALU" in d and "SQ_INSTS_MFMA" in d:
            lines.append("   non-MFMA VALU %s: %.1f; SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.2f" % (
                unit, (d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) / w, d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]))
        print("\n".join(lines))
        fh.write("\n".join(lines) + "\n")
PY
