#!/bin/bash
# The randomised parity sweeps of a round, run in parallel on the GPU box's cores (the CPU oracle
# is most of their time); summaries -- trial counts, mismatches, the histogram of the paths the
# trials took (mi_engine_last_call_path) -- land in gpurun_out/<tag>_stress_*.txt, from where
# they are copied to profiles/.
#   bash tools/stress_round.sh r05      (STRESS_SEED_SHIFT=<k>: every sweep's seed + k -- another pass)
tag=${1:-rXX}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {  # name, tool, seed, trials, [ENV=value ...]
  local name=$1 tool=$2 seed=$(($3 + ${STRESS_SEED_SHIFT:-0})) trials=$4; shift 4
  env STRESS_SEED=$seed STRESS_TRIALS=$trials "$@" timeout 3000 python3 tools/stress_$tool.py > gpurun_out/stress_$name.log 2>&1
  {
    echo "stress_$tool ($name) seed $seed trials $trials $* rc=$?"
    grep -c "MISMATCH" gpurun_out/stress_$name.log | sed 's/^/mismatch lines: /'
    sed -n '/^trials\|^rooted trials/,$p' gpurun_out/stress_$name.log
  } > gpurun_out/${tag}_stress_$name.txt
}
run parity_a parity 3001 1500 &
run parity_b parity 3002 1500 &
run parity_fused parity 3003 600 STRESS_FOCUS=fused &
run parity_arena parity 3004 400 STRESS_FOCUS=arena &
run parity_arena_v2 parity 3012 200 STRESS_FOCUS=arena MI_PHYLO_WALK3_ARENA=0 &
# (a batch of 150 small-pattern trees often fits the chip at once and keeps its vectors in LDS:
# the arena variants forced, so that every gradient call of the sweep takes them)
run parity_arena_forced parity 3013 300 STRESS_FOCUS=arena MI_PHYLO_GRADIENT_STORE=arena &
run rooted_arena_forced rooted 3014 300 MI_PHYLO_GRADIENT_STORE=arena &
run parity_v2 parity 3005 600 MI_PHYLO_GRADIENT_WALK=v2 &
# (wide pattern tiles -- engines whose batches take the arena get them by tile count -- on every
# engine the look-up walk takes, and with the arena forced)
run parity_wide parity 3015 600 MI_PHYLO_WALK_TILE_REGS=4 &
run parity_wide_arena parity 3016 300 STRESS_FOCUS=arena MI_PHYLO_WALK_TILE_REGS=4 MI_PHYLO_GRADIENT_STORE=arena &
run rooted_wide rooted 3017 300 MI_PHYLO_WALK_TILE_REGS=4 &
run parity_unfused parity 3007 400 MI_PHYLO_FUSED_SETUP=0 STRESS_FOCUS=fused &
run parity_analytic parity 3008 400 MI_PHYLO_SUBST_GRADIENT=analytic &
run rooted rooted 3009 800 &
run aa aa 3011 500 &
run aa_b aa 5012 500 &
# (the sweeps' problems are small: the launchers would never choose a ring for them)
run aa_ring1 aa 6001 400 MI_PHYLO_AA_RING=1 MI_PHYLO_AA_PRE_RING=2 &
run aa_ring2 aa 6002 400 MI_PHYLO_AA_RING=2 MI_PHYLO_AA_PRE_RING=0 &
run aa_ring4 aa 6003 400 MI_PHYLO_AA_RING=4 MI_PHYLO_AA_POST_TILES=4 &
run aa_tiles1 aa 6004 400 MI_PHYLO_AA_POST_TILES=1 MI_PHYLO_AA_RING=2 &
wait
cat gpurun_out/${tag}_stress_*.txt
