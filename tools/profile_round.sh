#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run from the repo root through
# gpurun); everything lands under gpurun_out/prof_<tag>/ and is summarised into
# gpurun_out/prof_<tag>/summary/, from where the files are copied to profiles/<tag>_*.
#   bash tools/profile_round.sh r03
# Kernel-trace statistics and PMC counters are separate runs (counters with --kernel-trace
# only, FETCH_SIZE and WRITE_SIZE in passes of their own), as the pool's gpurun requires.
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$tag
S=$O/summary
mkdir -p $S
H="--steps 20 --warmup 3 --headline-only"

stats() {  # name, command...
  local name=$1; shift
  rm -rf $O/$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1
  find $O/$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $S/${tag}_${name}_kernel_stats.csv
  grep -h '^{\|^T \|^GTR' $O/$name.log | tail -1 | cut -c1-600 > $S/${tag}_${name}_result.txt
}
pmc() {  # name, counter, command...
  local name=$1 counter=$2; shift 2
  rm -rf $O/$name
  rocprofv3 --kernel-trace --pmc $counter --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1
  find $O/$name -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $S/${tag}_${name}.csv
}

stats gradient python3 bench.py $H
stats loglik python3 bench.py --mode loglik $H
stats small_batch_125 python3 bench.py --trees 125 --steps 50 --warmup 5 --headline-only
stats gtr_full python3 tools/bench_gtr.py 1000 10
stats flua_1 python3 tools/bench_flua.py 1
stats flua_1000 python3 tools/bench_flua.py 1000
stats aa_T1 python3 tools/bench_aa.py --trees 1 --steps 3
stats aa_T8 python3 tools/bench_aa.py --trees 8 --steps 3
# round 6: a rank's pattern block of BASELINE configs[4] under 8-way sharding; the arena shapes
stats aa_shard_T1 python3 tools/bench_aa.py --trees 1 --steps 10 --patterns 6250
stats ts_64x1008 python3 tools/bench_tree_size.py 64 1008
stats ts_36x1812 python3 tools/bench_tree_size.py 36 1812
stats ts_50x378 python3 tools/bench_tree_size.py 50 378
python3 tools/bench_shapes_ab.py default 2>&1 | grep -v amdgpu.ids > $S/${tag}_tree_size.txt
python3 tools/bench_shapes_ab.py --shapes 36x1812x4,41x1137x4,50x378x4,50x1133x4,59x1824x4,64x1008x4,100x500x4,69x238x1,45x1000x1,45x1000x2 default gen2=MI_PHYLO_GRADIENT_WALK=v2 lds=MI_PHYLO_GRADIENT_STORE=lds 2>&1 | grep -v amdgpu.ids > $S/${tag}_arena_ab.txt
python3 tools/bench_small_step.py --rounds 2 default=MI_PHYLO_FUSED_FENCE=l1 none=MI_PHYLO_FUSED_FENCE=none none_id_order=MI_PHYLO_FUSED_FENCE=none,MI_PHYLO_FUSED_COLOCATE=0 agent=MI_PHYLO_FUSED_FENCE=agent four_launches=MI_PHYLO_FUSED_SETUP=0 2>&1 | grep -v amdgpu.ids > $S/${tag}_small_step_handoff.txt

pmc pmc_fetch_gradient FETCH_SIZE python3 bench.py --steps 3 --warmup 1 --headline-only
pmc pmc_write_gradient WRITE_SIZE python3 bench.py --steps 3 --warmup 1 --headline-only
pmc pmc_fetch_loglik FETCH_SIZE python3 bench.py --mode loglik --steps 3 --warmup 1 --headline-only
pmc pmc_write_loglik WRITE_SIZE python3 bench.py --mode loglik --steps 3 --warmup 1 --headline-only
for T in 1 8; do
  pmc aa_T${T}_pmc_fetch FETCH_SIZE python3 tools/bench_aa.py --trees $T --steps 2
  pmc aa_T${T}_pmc_write WRITE_SIZE python3 tools/bench_aa.py --trees $T --steps 2
done

SQ_JOBS=78000 bash tools/sq_counters.sh $S/${tag}_sq_counters_gradient_walk.txt > /dev/null 2>&1
SQ_JOBS=59000 bash tools/sq_counters.sh $S/${tag}_sq_counters_loglik.txt --mode loglik > /dev/null 2>&1

# SQ counters of the 20-state walk kernels (one tree and eight trees per launch)
for T in 1 8; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_MFMA" "SQ_WAVES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    rm -rf $O/aa_sq${T}_$i
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/aa_sq${T}_$i -- python3 tools/bench_aa.py --trees $T --steps 2 > $O/aa_sq${T}_$i.log 2>&1
  done
  python3 - $O $T > $S/${tag}_aa_T${T}_sq_counters.txt <<'PY'
import csv, glob, collections, sys
O, T = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2, 3):
    for f in glob.glob("%s/aa_sq%s_%d/**/*counter_collection.csv" % (O, T, i), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "aa_post" in k or "aa_pre" in k:
                key = ("aa_post_wg_kernel<2,true> (gradient)" if "true" in k else "aa_post_wg_kernel<2,false> (log_likelihoods)") if "aa_post" in k else "aa_pre_wg_kernel<2>"
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, "--", T, "tree(s) per launch")
    d = {c: sum(x) / len(x) for c, x in v.items()}
    w = d["SQ_WAVES"]
    for c, x in sorted(d.items()):
        print("   %-26s %16.0f  per wave %12.1f" % (c, x, x / w))
PY
done

# per-wave timeline of the gradient walk (diagnostic build, made here if absent)
[ -f libsbn_amd/variants/timeline.so ] || make -C libsbn_amd/csrc timeline > $O/timeline_build.log 2>&1
for T in 125 1000; do python3 tools/walk_timeline.py $T 2>&1 | grep -v amdgpu.ids; done > $S/${tag}_walk_timeline_tiles.txt

./build_tools/fp64_peak_probe > $S/${tag}_fp64_peak_probe.txt 2>&1
./build_tools/lds_dma_probe > $S/${tag}_lds_dma_probe.txt 2>&1

# HBM-side bytes per launch: (2 FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts
# 128-byte requests as 64 bytes -- MI355X_MICROARCH.md, HBM / rocprofv3)
python3 - $S $tag > $S/${tag}_traffic.json <<'PY'
import csv, glob, collections, json, sys
S, tag = sys.argv[1], sys.argv[2]
def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
out = {}
def add(fetch_csv, write_csv, launch, suffix="", only=None):
    try:
        f, w = per_kernel(fetch_csv), per_kernel(write_csv)
    except OSError:
        return
    for k in f:
        if only and not any(o in k for o in only):
            continue
        if "miphylo" not in k:
            continue
        name = k.replace("void ", "").replace("(anonymous namespace)::", "").replace("miphylo::", "")
        name = name.split("(")[0].strip()  # e.g. gradient_walk_kernel<3, false, false, false>
        out[name + suffix] = {"FETCH_SIZE_KB_raw": f[k], "WRITE_SIZE_KB_raw": w.get(k, 0.0),
                              "hbm_bytes_per_launch": int((2 * f[k] + w.get(k, 0.0)) * 1024),
                              "launch": launch, "round": tag}
add("%s/%s_pmc_fetch_gradient.csv" % (S, tag), "%s/%s_pmc_write_gradient.csv" % (S, tag),
    "1000 DS1 trees, JC69+weibull+4, bench.py --steps 3 --warmup 1 (gradient)")
add("%s/%s_pmc_fetch_loglik.csv" % (S, tag), "%s/%s_pmc_write_loglik.csv" % (S, tag),
    "1000 DS1 trees, JC69+weibull+4, bench.py --mode loglik --steps 3 --warmup 1", only=["loglik"])
for T in (1, 8):
    add("%s/%s_aa_T%d_pmc_fetch.csv" % (S, tag, T), "%s/%s_aa_T%d_pmc_write.csv" % (S, tag, T),
        "S-WAG 512 x 50 000 x 4, %d tree(s) per launch (tools/bench_aa.py --trees %d --steps 2)" % (T, T),
        suffix="|T=%d" % T, only=["aa_post", "aa_pre"])
print(json.dumps(out, indent=1))
PY
ls -la $S
find $O -mindepth 1 -maxdepth 1 -not -name summary -exec rm -rf {} +
