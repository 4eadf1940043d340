// Host side of the C ABI declared in include/mi_phylo.h: device memory, the
// per-call launch sequence, error reporting.  Compiled with hipcc; no torch.
//
// Per call (all on one HIP stream, no host synchronisation in the *_device path):
//   tree_setup -> model_setup -> transition -> {loglik_* | gradient_mfma | gradient_hbm}*
//   -> reduce_tiles -> finalize
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mi_phylo.h"
#include "mi_phylo_kernels.h"

using namespace miphylo;

namespace {
thread_local std::string g_error;
}

namespace miphylo {
// sets mi_last_error(); shared with mi_site_pattern.hip
int fail(const std::string& msg) {
  g_error = msg;
  return 1;
}
void set_last_error(const std::string& msg) { g_error = msg; }
}  // namespace miphylo

#include "mi_phylo_engine.h"

namespace {

struct CallShape {
  int T, E, Eg, M, models_per_tree;
  bool gradient, gtr, site_fused, site_separate;
};

// analytic: the opt-in analytic substitution gradient replaces the 16 finite-difference
// evaluations (and with them the perturbed-model site pass): one gradient evaluation per
// tree, as for JC69.
// (also for a GTR gradient call that asks for neither the substitution nor the site gradient
// -- `light`, run_device: the finite-difference passes and the perturbed-model site pass would
// be computed for nobody)
CallShape call_shape(const mi_engine* e, int T, bool gradient, bool analytic = false) {
  CallShape c{};
  c.T = T;
  c.gradient = gradient;
  c.gtr = e->spec.subst_model == MI_SUBST_GTR;
  const bool fd = gradient && c.gtr && !analytic;
  c.site_fused = gradient && e->K > 1 && !fd;
  c.site_separate = gradient && e->K > 1 && fd;
  c.models_per_tree = fd ? kFdModels : 1;
  c.M = T * c.models_per_tree;
  c.E = T;
  c.Eg = gradient ? T : 0;
  if (fd) c.E += 16 * T;
  if (c.site_separate) {
    c.E += T;
    c.Eg += T;
  }
  return c;
}

// (waves: one-wave workgroups of a gradient launch; default: a large batch)
bool walk3_possible(const mi_engine* e);
bool use_arena(const mi_engine* e, bool rescale, bool subst, size_t waves = (size_t)-1, int regs = 0) {
  // (the look-up walk's arena variant starts one step earlier: gradient_walk_use_arena)
  const bool lut = walk3_possible(e) && e->walk3_arena && !subst && gradient_mfma_groups(e->K) == 1;
  return gradient_walk_use_arena(e->n, e->K, rescale, subst, waves, lut, regs);
}
bool walk_fits(const mi_engine* e, bool rescale) { return gradient_walk_fits(e->n, e->K, rescale); }
// (engine creation, tips in mask form on the device: the log-likelihood kernel's pre-tiled copy)
int engine_tile_regs(mi_engine* e);
int build_tip_tiles(mi_engine* e) {
  if (!e->have_tip_masks || e->K > kMaxCategories) return 0;
  if (e->tip_tiles.ensure(loglik_tip_tiles_bytes(e->n, e->P, e->K))) return 1;
  launch_tip_tiles(e->tip_masks.as<uint8_t>(), e->tip_tiles.as<uint8_t>(), e->n, e->P, e->K, e->stream);
  // ... and the look-up walk's, for the engine's tile width
  if (walk3_possible(e) && gradient_mfma_groups(e->K) == 1) {
    const int regs = engine_tile_regs(e);
    if (e->tip_code_tiles.ensure(tip_code_tiles_bytes(e->n, e->P, e->K, regs))) return 1;
    launch_tip_code_tiles(e->tip_codes.as<uint8_t>(), e->tip_code_tiles.as<uint8_t>(), e->n, e->P, e->K, regs, e->stream);
  }
  return 0;
}
// The look-up walk's tile width for this engine (kernels_walk3.hip, RR; gradient_walk_tile_regs):
// wide tiles pay in the arena variant, so an engine gets them if its batches take the arena --
// and then for every look-up-walk call: sums over patterns are formed tile by tile, and a
// tree's outputs must not depend on the size of the batch it came in.  (Its calls of a few
// trees keep every vector in LDS with the same wide tiles, one wave per SIMD:
// gradient_walk_use_arena.)
int engine_tile_regs(mi_engine* e) {
  if (e->tile_regs < 0) {
    const bool lut = walk3_possible(e) && e->walk3_arena && gradient_mfma_groups(e->K) == 1;
    const bool forced = getenv("MI_PHYLO_WALK_TILE_REGS") != nullptr;
    const int r = lut && (forced || gradient_walk_batches_take_arena(e->n, e->K, true)) ? gradient_walk_tile_regs(e->n, e->P, e->K) : 0;
    e->tile_regs = r > kLlR ? r : 0;
  }
  return e->tile_regs;
}

// which log-likelihood kernel a call uses (also decides who fills the tip tables)
bool loglik_kernel_is_valu(const mi_engine* e, bool rescaling) {
  LikArgs probe{};
  probe.n = e->n;
  probe.K = e->K;
  probe.tip_masks = e->have_tip_masks ? e->tip_masks.as<uint8_t>() : nullptr;
  return std::string(loglik_kernel_name(probe, rescaling, e->max_slots)) == "loglik_onchip_kernel";
}
// Does a gradient call run on the matrix-core walk kernel?  ONE predicate for run_device and
// mi_engine_reserve (a reserve that guesses differently leaves a later *_device call to
// allocate -- inside a hipGraph capture, for instance).  (K > 4: the kernel takes the site
// likelihoods from a pass of the matrix-core log-likelihood kernel; if that one cannot run,
// neither can it.)
bool matrix_core_gradient(const mi_engine* e, bool rescaling) {
  return e->allow_onchip_gradient && e->have_tip_masks && walk_fits(e, rescaling) &&
         reduce_tiles_fits(e->N) &&
         (gradient_mfma_groups(e->K) == 1 || !loglik_kernel_is_valu(e, rescaling));
}

// Can calls of this engine take the third-generation walk (kernels_walk3.hip)?  (Per call it
// also needs the stored vectors in LDS and no analytic substitution gradient.)
bool walk3_possible(const mi_engine* e) {
  return e->walk3 && e->have_tip_codes && gradient_walk_lut_applies(e->K);
}

size_t plv_bytes_per_eval(const mi_engine* e) {
  return (size_t)(e->n - 1) * e->K * e->tiles * kTile * 4 * sizeof(double);
}

int reserve(mi_engine* e, int T, bool gradient, bool need_hbm_path = true,
            bool analytic = false, bool light = false) {
  const CallShape c = call_shape(e, T, gradient, analytic || light);
  const int n = e->n, N = e->N;
  if (e->tree_scratch.ensure(sizeof(int32_t) * (size_t)T * 13 * N)) return 1;
  if (e->sched.ensure(sizeof(SchedEntry) * (size_t)T * (n - 1))) return 1;
  if (e->macros.ensure(sizeof(MacroEntry) * (size_t)T * macro_stride(n))) return 1;
  if (e->macro_count.ensure(sizeof(int32_t) * (size_t)T)) return 1;
  if (e->bl_eff.ensure(sizeof(double) * (size_t)T * N)) return 1;
  if (e->models.ensure(sizeof(DevModel) * (size_t)c.M)) return 1;
  if (e->mats.ensure(sizeof(double) * (size_t)c.E * (N - 1) * e->K * 16)) return 1;
  if (e->tip_tables.ensure(sizeof(double) * (size_t)c.E * n * e->K * 20)) return 1;
  if (gradient) {
    // matrices in the walk's order, per gradient evaluation (kernels_walk.hip)
    const size_t per = std::max(gradient_walk_mats_bytes_per_eval(n, e->K),
                                walk3_possible(e) ? gradient_walk_lut_mats_bytes_per_eval(n) : 0);
    if (e->mmats.ensure(per * (size_t)c.Eg)) return 1;
    if (analytic && e->mphi.ensure(per / 2 * (size_t)c.Eg)) return 1;
  }
  if (analytic && e->x_sum.ensure(sizeof(double) * (size_t)c.Eg * kSubstExtra)) return 1;
  if (e->ll_part.ensure(sizeof(double) * (size_t)c.E * e->ll_stride)) return 1;
  if (e->fin_scratch.ensure(sizeof(double) * (size_t)T * 6 * n)) return 1;
  if (e->status.ensure(sizeof(int32_t) * kStatusWords)) return 1;
  if (gradient && e->fused_setup && walk3_possible(e) && e->ready.bytes < sizeof(int32_t) * kReadyStride * (size_t)T) {
    // hand-off words of the one-launch small call: zero whenever no such call is running
    if (e->ready.ensure(sizeof(int32_t) * kReadyStride * (size_t)T)) return 1;
    HIP_TRY(hipMemset(e->ready.ptr, 0, e->ready.bytes));
    HIP_TRY(hipDeviceSynchronize());
  }
  if (e->ll_sum.ensure(sizeof(double) * (size_t)c.E)) return 1;
  if (gradient && e->g_sum.ensure(sizeof(double) * (size_t)c.Eg * 2 * N)) return 1;
  if (gradient) {
    // the HBM-streamed kernel is the fallback for rescaling / trees that do not fit
    // in LDS; its arena is only allocated when that path can be taken
    const size_t per = plv_bytes_per_eval(e);
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(c.Eg, e->plv_budget / per));
    if (need_hbm_path && e->plv.ensure(per * chunk)) return 1;
    // the arena variant of the matrix-core kernel keeps its stored vectors in the same buffer
    if (!need_hbm_path && (engine_tile_regs(e) || use_arena(e, false, true) || use_arena(e, true, true) ||
                           use_arena(e, false, false) || use_arena(e, true, false))) {
      const size_t aper = gradient_arena_bytes_per_eval(n, e->P, e->K);
      const size_t achunk = std::max<size_t>(1, std::min<size_t>(c.Eg, e->plv_budget / aper));
      if (e->plv.ensure(aper * achunk)) return 1;
      if (e->arena_macros.ensure(sizeof(MacroEntry) * (size_t)T * macro_stride(n))) return 1;
      if (e->slot_need.ensure(sizeof(int32_t) * (size_t)T)) return 1;
    }
    const size_t g_width = std::max<size_t>(2 * (size_t)N, (size_t)gradient_mfma_width(n, true));
    if (e->g_part.ensure(sizeof(double) * (size_t)c.Eg * e->ll_stride * gradient_mfma_groups(e->K) *
                         g_width))
      return 1;
    if (e->site_lik.ensure(sizeof(double) * (size_t)c.Eg * e->tiles * kTile)) return 1;
    if (e->site_exp.ensure(sizeof(int32_t) * (size_t)c.Eg * e->tiles * kTile)) return 1;
  }
  return 0;
}

// Enqueue one engine call; every pointer in `d` is a device pointer.
int run_device(mi_engine* e, hipStream_t s, const DeviceCall& d) {
  // (a caller driving several GPUs from one thread may have another device current)
  HIP_TRY(hipSetDevice(e->spec.device));
  if (e->s == kAa) return aa_run_device(e, s, d);
  if (d.T <= 0) return fail("tree_count must be positive");
  if (!d.parent_ids || !d.bl || !d.out_ll) return fail("null tree / output pointer");
  if (e->param_count > 0 && !d.params) return fail("null parameter matrix");
  // on-chip gradient kernels: the matrix-core one (K <= 4; rescaling supported) or the
  // VALU one (no rescaling); everything else takes the HBM-streamed kernel
  // which log-likelihood kernel runs (also decides who fills the tip tables, below)
  const bool loglik_is_valu = loglik_kernel_is_valu(e, d.rescaling);
  const bool mfma = d.gradient && matrix_core_gradient(e, d.rescaling);
  const bool onchip = mfma;  // the only on-chip gradient kernel; everything else streams PLVs
  const int groups = mfma ? gradient_mfma_groups(e->K) : 1;
  const bool analytic = e->analytic_subst && mfma && e->spec.subst_model == MI_SUBST_GTR;
  // (a wide-tile engine: every call the look-up walk can take runs it, with wide tiles)
  const int tile_regs = mfma && !analytic && groups == 1 ? engine_tile_regs(e) : 0;
  const int g_tiles = mfma ? gradient_mfma_tiles(e->P, e->K, tile_regs) * groups : e->tiles;
  // A GTR gradient call whose caller wants neither the substitution-model nor the site-model
  // gradient (BASELINE configs[2] as worded: log-likelihood + branch-length gradient) is ONE
  // evaluation per tree with the tree's own model, exactly like a JC69 call: no perturbed
  // model instances, no finite-difference passes -- and it can take the one-launch path.
  // What it delivers is bit-identical to the full call's.
  const bool light = d.gradient && mfma && !analytic && e->spec.subst_model == MI_SUBST_GTR &&
                     !d.out_subst && !d.out_site;
  if (reserve(e, d.T, d.gradient, !onchip, analytic, light)) return 1;
  const CallShape c = call_shape(e, d.T, d.gradient, analytic || light);
  const int n = e->n, N = e->N, T = d.T;
  // (the status word is sticky: cleared when it is read, check_status -- not per call: one
  // dispatch less on the small-batch path)

  TreeSetupArgs ts{};
  ts.n = n;
  ts.T = T;
  ts.rooted = d.rooted;
  ts.parent_ids = d.parent_ids;
  ts.bl = d.bl;
  // rooted trees: LogLikelihood/Gradient scale by rates (fat_beagle.cpp:96-101,507-511);
  // UnrootedLogLikelihood(RootedTree) does not (:78-80).
  ts.rates = (d.rooted && (d.gradient || d.with_jacobian)) ? d.rates : nullptr;
  ts.scratch = e->tree_scratch.as<int32_t>();
  ts.sched = e->sched.as<SchedEntry>();
  // (the gradient schedule is only built for gradient calls that walk it)
  ts.macros = (d.gradient && mfma) ? e->macros.as<MacroEntry>() : nullptr;
  ts.macro_count = e->macro_count.as<int32_t>();
  ts.bl_eff = e->bl_eff.as<double>();
  ts.status = e->status.as<int32_t>();
  ts.max_slots = e->max_slots;
  // the Sethi-Ullman schedule with LDS slots is what the log-likelihood kernels walk
  ts.need_slots = !(d.gradient && mfma && groups == 1 && (!c.gtr || analytic || light));
  ModelSetupArgs ms{};
  ms.T = T;
  ms.models_per_tree = c.models_per_tree;
  ms.subst = e->spec.subst_model;
  ms.site = e->spec.site_model;
  ms.K = e->K;
  ms.param_count = e->param_count;
  ms.rates_off = e->rates_off;
  ms.freqs_off = e->freqs_off;
  ms.shape_off = e->shape_off;
  ms.params = d.params;
  ms.models = e->models.as<DevModel>();
  ms.status = e->status.as<int32_t>();
  ms.weibull_x = e->weibull_x.as<double>();
  const bool prof = e->prof_used < e->prof_capacity;
  const bool marks = prof && e->prof_phases;
  PROF_MARK(e, marks, 0, s);
  // (a call of a few trees keeps its stored vectors in LDS however large the tree)
  const bool arena = mfma && use_arena(e, d.rescaling, analytic, (size_t)T * (size_t)g_tiles, tile_regs);
  const bool walk2 = mfma;  // (every matrix-core call: the first generation was retired in round 6)
  // the third-generation (look-up) walk: stored vectors in LDS or, since round 6, in the arena
  constexpr int kMaxEvals = 32768;
  // The one-launch call (kernels_walk3.hip): tree set-up, model instances and operand records
  // ride in the walk's launch.  One evaluation and one model instance per tree (JC69-type
  // calls), trees of at most 64 nodes, one walk launch, nobody else reads the schedule's LDS
  // slots.
  static const bool fuse_allowed =
      !(getenv("MI_PHYLO_FUSE_FINALIZE") && std::string(getenv("MI_PHYLO_FUSE_FINALIZE")) == "0");
  // Up to 512 trees: the set-up waves take wave slots the walk would use (four waves of ~10
  // microseconds per tree, a GTR eigensystem on one lane of each) -- measured, DS1
  // (tools/bench_fused_scan.py): one launch / four launches 0.91 at 1-8 trees,
  // 0.98 at 250-500, 0.99 at 1000 (JC69; GTR 1.00), 1.01 beyond.  MI_PHYLO_FUSED_MAX_TREES
  // moves the cross-over (testing).
  static const int fuse_max_trees =
      getenv("MI_PHYLO_FUSED_MAX_TREES") ? atoi(getenv("MI_PHYLO_FUSED_MAX_TREES")) : 512;
  const bool fuse_possible = mfma && walk3_possible(e) && !analytic && groups == 1 && !arena && !tile_regs && e->fused_setup &&
                             fuse_allowed && c.E == T && c.models_per_tree == 1 && !ts.need_slots &&
                             T <= fuse_max_trees && e->ready.ptr && gradient_walk_lut_fused_applies(n, e->K);
  // One rate category with the stored vectors in LDS.  The second generation, whose waves take
  // several tiles of a tree in a row, was 5-7 % ahead on a large batch (DS1 x 1000 with the
  // constant site model 0.281 against 0.297 ms) and the look-up walk, with its one-launch call,
  // 6-20 % ahead up to 500 trees (profiles/r06_k1_small_batches.txt): so the rule was "look-up
  // walk where the one-launch call applies".  With the tip codes pre-tiled (a one-category wave
  // regrouped 16 columns of fields per tip: a quarter of its vector instructions) the look-up
  // walk is level on large batches too -- 1000 / 4000 trees, second generation / look-up walk:
  // DS1's shape 0.282 / 0.281 and 0.970 / 0.976 ms, 31 x 1000 0.360 / 0.365 and 1.26 / 1.31, 16 x
  // 500 0.112 / 0.106 and 0.373 / 0.364, 29 x 1195 0.362 / 0.361 and 1.39 / 1.28
  // (profiles/r06_k1_large_batches.txt) -- and takes every one-category call
  // (MI_PHYLO_WALK3_K1=0: the old rule).  (Two and three categories: look-up walk 0.462 / 0.490
  // and 0.831 / 0.868 ms per 1000 DS1 trees.)
  const bool walk3 = walk2 && walk3_possible(e) && !analytic && groups == 1 &&
                     (tile_regs || (arena ? e->walk3_arena : (e->K > 1 || e->walk3_k1_lds || fuse_possible)));
  const bool fuse_setup = walk3 && fuse_possible;
  // Beyond the one-launch call's size the same set-up waves CAN run as one launch in front of the
  // walk's (round 6, MI_PHYLO_SETUP_RECORDS=1): trees, model instances and operand records --
  // instead of the tree set-up launch and the record launch, 13 + 18 us of a 1000-tree DS1 step.
  // Built, bit-identical (test), and not the default: the four quarter-waves of a tree each build
  // the tree, and 4 000 of them take what the two launches take -- the replayed headline step
  // 0.7826 against 0.7823 ms (tools/ab_kernels.py, four rounds), direct launches -1 %.
  const bool setup_records = walk3 && !fuse_setup && !arena && !tile_regs && !analytic && groups == 1 && e->fused_setup &&
                             c.E == T && c.models_per_tree == 1 && !ts.need_slots && T <= kMaxEvals &&
                             gradient_walk_lut_fused_applies(n, e->K) &&
                             getenv("MI_PHYLO_SETUP_RECORDS") && getenv("MI_PHYLO_SETUP_RECORDS")[0] == '1';
  // (arena calls: the slot assignment rides in the set-up launch where a workgroup builds the tree)
  ts.arena_macros = arena ? e->arena_macros.as<MacroEntry>() : nullptr;
  ts.slot_need = e->slot_need.as<int32_t>();
  if (setup_records) {
    FusedSetupArgs fs{};
    fs.ts = ts;
    fs.ms = ms;
    fs.mmats = e->mmats.as<double>();
    fs.colocate = e->fused_colocate;
    launch_setup_records(fs, T, s);
  }
  const bool slots_done = !fuse_setup && !setup_records && launch_setup(ts, ms, s);  // tree schedules and model instances, one launch
  if (arena && !slots_done)
    launch_macro_slots(e->macros.as<MacroEntry>(), e->arena_macros.as<MacroEntry>(),
                       e->macro_count.as<int32_t>(), n, T, e->slot_need.as<int32_t>(),
                       e->status.as<int32_t>(), s);

  const EvalMap map{T, c.models_per_tree};
  TransitionArgs tr{};
  tr.E = c.E;
  tr.N = N;
  tr.K = e->K;
  tr.map = map;
  tr.models = e->models.as<DevModel>();
  tr.bl_eff = e->bl_eff.as<double>();
  tr.mats = e->mats.as<double>();
  // the per-state tip tables feed the VALU walk kernels only
  // (only the VALU log-likelihood kernel reads them)
  // The C ABI's outputs are optional, and work nobody reads is not done: without a
  // substitution-gradient output the 16 finite-difference log-likelihood passes of a GTR
  // call are skipped, without a site-gradient output the extra gradient pass under the
  // perturbed model (section 8 of DESIGN.md) too.  The evaluations keep their numbers; what
  // is delivered is bit-identical to the full call.
  const bool fd_pass = d.gradient && c.gtr && !analytic && d.out_subst != nullptr;
  const bool site_pass = c.site_separate && d.out_site != nullptr;
  const bool loglik_runs = !d.gradient || fd_pass || (mfma && groups > 1);
  const bool need_tip_tables = loglik_runs && loglik_is_valu;
  tr.tip_tables = need_tip_tables ? e->tip_tables.as<double>() : nullptr;
  tr.n = n;
  // evaluations nobody walks need no matrices at all
  tr.ev_skip_begin = tr.ev_skip_end = 0;
  if (d.gradient && c.gtr && !analytic && !light && !fd_pass) {
    tr.ev_skip_begin = T;
    tr.ev_skip_end = site_pass ? 17 * T : c.E;
  }
  if (walk2 && groups == 1) {
    // the second-generation walk reads its matrices in macro order (below); node-ordered
    // ones are only needed by the evaluations a log-likelihood kernel walks: the
    // finite-difference passes [T, 17 T) of a GTR call
    if (fd_pass) {
      tr.eval_base = T;
      tr.E = 16 * T;
      launch_transition(tr, s);
    }
  } else {
    launch_transition(tr, s);
  }
  const MacroEntry* walk_macros = arena ? e->arena_macros.as<MacroEntry>() : e->macros.as<MacroEntry>();
  auto macro_matrices = [&](int eval_begin, int grad_begin, int count) {
    TransitionMacroArgs tm{};
    tm.n = n;
    tm.N = N;
    tm.K = e->K;
    tm.count = count;
    tm.eval_begin = eval_begin;
    tm.map = map;
    tm.models = e->models.as<DevModel>();
    tm.bl_eff = e->bl_eff.as<double>();
    tm.macros = walk_macros;
    tm.macro_count = e->macro_count.as<int32_t>();
    if (walk3) {
      tm.mmats = e->mmats.as<double>() +
                 (size_t)grad_begin * (gradient_walk_lut_mats_bytes_per_eval(n) / sizeof(double));
      tm.mphi = nullptr;
      launch_transition_lut(tm, s);
      return;
    }
    const size_t per = gradient_walk_mats_bytes_per_eval(n, e->K) / sizeof(double);
    tm.mmats = e->mmats.as<double>() + (size_t)grad_begin * per;
    tm.mphi = analytic ? e->mphi.as<double>() + (size_t)grad_begin * (per / 2) : nullptr;
    launch_transition_macro(tm, s);
  };
  if (walk2 && !fuse_setup && !setup_records) {
    macro_matrices(0, 0, T);
    if (site_pass) macro_matrices(17 * T, T, T);
  }

  LikArgs la{};
  la.n = n;
  la.N = N;
  la.P = e->P;
  la.K = e->K;
  la.tiles = e->tiles;
  la.ll_tiles = e->ll_stride;
  la.g_tiles = g_tiles;
  la.map = map;
  la.models = e->models.as<DevModel>();
  la.sched = e->sched.as<SchedEntry>();
  la.macros = arena ? e->arena_macros.as<MacroEntry>() : e->macros.as<MacroEntry>();
  la.macro_count = e->macro_count.as<int32_t>();
  la.mats = e->mats.as<double>();
  la.tip_tables = e->tip_tables.as<double>();
  la.mmats = e->mmats.as<double>();
  la.mphi = e->mphi.as<double>();
  la.tip_states = e->tip_states.as<int8_t>();
  la.tip_masks = e->have_tip_masks ? e->tip_masks.as<uint8_t>() : nullptr;
  {  // (the look-up walk's pre-tiled codes were made for the engine's tile width)
    const char* off = getenv("MI_PHYLO_TIP_TILES");
    la.tip_code_tiles = e->have_tip_codes && e->tip_code_tiles.ptr && !(off && off[0] == '0') &&
                                engine_tile_regs(e) == tile_regs
                            ? e->tip_code_tiles.as<uint8_t>()
                            : nullptr;
  }
  // (MI_PHYLO_TIP_TILES=0: the kernels stage their tip bytes from tip_masks / tip_codes themselves -- A/B, tests)
  la.tip_tiles = e->have_tip_masks && e->tip_tiles.ptr && !(getenv("MI_PHYLO_TIP_TILES") && getenv("MI_PHYLO_TIP_TILES")[0] == '0')
                     ? e->tip_tiles.as<uint8_t>() : nullptr;
  la.tip_codes = e->have_tip_codes ? e->tip_codes.as<uint8_t>() : nullptr;
  la.tip_partials = e->spec.use_tip_states ? nullptr : e->tip_partials.as<double>();
  la.weights = e->weights.as<double>();
  la.ll_part = e->ll_part.as<double>();
  la.plv = e->plv.as<double>();
  la.g_part = e->g_part.as<double>();

  la.site_lik = nullptr;
  la.status = e->status.as<int32_t>();
  la.slot_need = e->slot_need.as<int32_t>();
  la.store = mfma ? (arena ? 2 : 1) : 0;  // (the launchers follow the choice the schedules were made for)
  la.tile_regs = tile_regs;
  // one launch covers at most kMaxEvals evaluations (grid y dimension: 65535; a multiple
  // of 8 keeps whole evaluations per XCD)
  int walk_launches = 0;
  auto loglik_range = [&](int eval_begin, int count) {
    for (int done = 0; done < count; done += kMaxEvals) {
      walk_launches++;
      LikArgs l = la;
      l.eval_offset = eval_begin + done;
      launch_loglik(l, std::min(kMaxEvals, count - done), d.rescaling, e->max_slots, s);
    }
  };
  auto grad_range = [&](int eval_begin, int grad_begin, int count) {
    if (mfma) {
      // (arena variant: a launch covers what its HBM arena holds)
      const int max_part =
          arena ? (int)std::max<size_t>(
                      1, std::min<size_t>(kMaxEvals, e->plv.bytes / gradient_arena_bytes_per_eval(
                                                                        n, e->P, e->K)))
                : kMaxEvals;
      for (int done = 0; done < count; done += max_part) {
        const int part = std::min(max_part, count - done);
        walk_launches++;
        LikArgs g = la;
        g.eval_offset = eval_begin + done;
        g.grad_offset = grad_begin + done;
        if (groups > 1) {
          // K > 4: the site likelihoods (and logL) come from a log-likelihood pass
          g.site_lik = e->site_lik.as<double>();
          g.site_exp = e->site_exp.as<int32_t>();
          launch_loglik(g, part, d.rescaling, e->max_slots, s);
        }
        if (fuse_setup) {
          FusedSetupArgs fs{};
          fs.ts = ts;
          fs.ms = ms;
          fs.mmats = e->mmats.as<double>();
          fs.ready = e->ready.as<int32_t>();
          fs.debug_skip = e->fused_debug_skip;
          fs.spin_ticks = e->fused_spin_ticks;
          fs.fence = e->fused_fence;
          fs.colocate = e->fused_colocate;
          launch_gradient_walk_lut_fused(g, fs, part, d.rescaling, s);
        } else if (walk3) launch_gradient_walk_lut(g, part, d.rescaling, s);
        else launch_gradient_walk(g, part, d.rescaling, analytic, s);
      }
      return;
    }
    const size_t per = plv_bytes_per_eval(e);
    const int chunk = (int)std::max<size_t>(
        1, std::min<size_t>(std::min(count, kMaxEvals), e->plv.bytes / per));
    for (int done = 0; done < count; done += chunk) {
      walk_launches++;
      LikArgs g = la;
      g.eval_offset = eval_begin + done;
      g.grad_offset = grad_begin + done;
      launch_gradient_hbm(g, std::min(chunk, count - done), d.rescaling, s);
    }
  };

  if (prof) HIP_TRY(hipEventRecord(prof_event(e, 0), s));
  PROF_MARK(e, marks, 1, s);
  PROF_MARK(e, marks, 2, s);
  if (!d.gradient) {
    loglik_range(0, T);
    e->dominant = loglik_kernel_name(la, d.rescaling, e->max_slots);
    if (prof) HIP_TRY(hipEventRecord(prof_event(e, 1), s));
    PROF_MARK(e, marks, 3, s);
  } else {
    grad_range(0, 0, T);
    if (prof) HIP_TRY(hipEventRecord(prof_event(e, 1), s));
    PROF_MARK(e, marks, 3, s);
    if (fd_pass) loglik_range(T, 16 * T);
    if (site_pass) grad_range(17 * T, T, T);
    e->dominant = fuse_setup ? gradient_walk_lut_fused_kernel_name()
                  : walk3 ? gradient_walk_lut_kernel_name()
                  : walk2 ? gradient_walk_kernel_name()
                        : gradient_kernel_name();
  }
  {  // which path the call took, for diagnostics (mi_engine_last_call_path)
    std::string path = e->dominant;
    if (d.gradient)
      path += !mfma ? " store=hbm" : (arena ? " store=arena" : " store=lds");
    path += fuse_setup ? " setup=in-walk" : (setup_records ? " setup=with-records" : " setup=own-launch");
    if (tile_regs > kLlR) path += " tile=wide";
    if (d.gradient && fd_pass) path += " fd=16";
    if (d.gradient && site_pass) path += " site-pass";
    if (light) path += " light";
    if (analytic) path += " analytic";
    if (d.rescaling) path += " rescaled";
    if (d.rooted) path += " rooted";
    path += " K=" + std::to_string(e->K);
    e->last_path = path;
  }
  e->prof_first_launch_evals = T;
  e->last_evals = c.E;
  e->last_grad_evals = c.Eg;
  e->last_walk_launches = walk_launches;

  FinalizeArgs fa{};
  fa.n = n;
  fa.N = N;
  fa.T = T;
  fa.K = e->K;
  fa.tiles = e->tiles;
  fa.ll_tiles = e->ll_stride;
  fa.g_tiles = g_tiles;
  fa.ll_part = e->ll_part.as<double>();
  fa.g_part = e->g_part.as<double>();
  // logL partial sums each evaluation's walk kernel wrote (no memset of ll_part: the
  // consumers sum exactly these): the log-likelihood kernel in use tiles the patterns its
  // way, the gradient kernels theirs; K > 4 takes the gradient evaluations' logL from the
  // log-likelihood pass
  const int ll_kernel_count = loglik_is_valu ? e->tiles : loglik_mfma_tiles(e->P, e->K);
  const int grad_kernel_count =
      mfma ? (groups > 1 ? ll_kernel_count : g_tiles) : e->tiles;
  LlCounts ll_used{d.gradient ? grad_kernel_count : ll_kernel_count, ll_kernel_count, 0, 0};
  if (d.gradient && c.gtr && !analytic && !light) {
    ll_used.mid_lo = T;
    ll_used.mid_hi = 17 * T;
  }
  fa.ll_used = ll_used;
  bool fused = false;
  ReduceArgs fused_ra{};
  if (reduce_tiles_fits(N)) {
    // sum the per-tile partials with one workgroup per evaluation first
    ReduceArgs ra{};
    ra.N = N;
    ra.E = c.E;
    ra.Eg = c.Eg;
    ra.ll_tiles = e->ll_stride;
    ra.ll_used = ll_used;
    ra.g_tiles = g_tiles;
    ra.ll_part = e->ll_part.as<double>();
    ra.g_part = e->g_part.as<double>();
    ra.ll_sum = e->ll_sum.as<double>();
    ra.g_sum = e->g_sum.as<double>();
    ra.g_width = (d.gradient && mfma) ? gradient_mfma_width(n, analytic) : 0;
    ra.extra = analytic ? kSubstExtra : 0;
    ra.x_sum = e->x_sum.as<double>();
    ra.n = n;
    ra.T = T;
    ra.macros = arena ? e->arena_macros.as<MacroEntry>() : e->macros.as<MacroEntry>();
    ra.macro_count = e->macro_count.as<int32_t>();
    // one evaluation per tree (JC69-type models, the analytic GTR gradient; log-likelihood
    // calls too): tile reduction and finalize step in ONE launch, a workgroup per tree
    fused = fuse_allowed && c.E == T;
    if (!fused) launch_reduce_tiles(ra, s);
    fused_ra = ra;
    fa.ll_tiles = 1;
    fa.ll_used = LlCounts{1, 1, 0, 0};
    fa.g_tiles = 1;
    fa.ll_part = ra.ll_sum;
    fa.g_part = ra.g_sum;
  }
  fa.gradient = d.gradient;
  fa.rooted = d.rooted;
  fa.with_jacobian = d.with_jacobian;
  fa.gtr = c.gtr && !analytic && !light;  // finite-difference assembly of the substitution gradient
  fa.site_fused = c.site_fused;
  fa.site_separate = c.site_separate;
  fa.bl_eff = e->bl_eff.as<double>();
  fa.bl_raw = d.bl;
  fa.rates = d.rates;
  fa.rate_counts = d.rate_counts;
  fa.node_heights = d.heights;
  fa.node_bounds = d.bounds;
  fa.height_ratios = d.ratios;
  fa.sched = e->sched.as<SchedEntry>();
  fa.scratch = e->fin_scratch.as<double>();
  fa.out_ll = d.out_ll;
  fa.out_branch = d.out_branch;
  fa.out_ratios = d.out_ratios;
  fa.out_clock = d.out_clock;
  fa.out_site = d.out_site;
  fa.out_subst = d.out_subst;
  fa.status = e->status.as<int32_t>();
  fa.clear_ready = fuse_setup ? e->ready.as<int32_t>() : nullptr;
  if (fuse_setup && !fused) return fail("internal error: the one-launch call needs the fused reduction");
  if (fused) launch_reduce_finalize(fused_ra, fa, s);
  else launch_finalize(fa, s);
  if (analytic && d.out_subst) {
    SubstGradArgs sg{};
    sg.T = T;
    sg.param_count = e->param_count;
    sg.rates_off = e->rates_off;
    sg.freqs_off = e->freqs_off;
    sg.params = d.params;
    sg.models = e->models.as<DevModel>();
    sg.x_sum = e->x_sum.as<double>();
    sg.out_subst = d.out_subst;
    launch_subst_gradient(sg, s);
  }
  PROF_MARK(e, marks, 4, s);
  if (prof) e->prof_used++;
  HIP_TRY(hipGetLastError());
  return 0;
}

int check_status(mi_engine* e, hipStream_t s) {
  HIP_TRY(hipSetDevice(e->spec.device));
  int32_t st[kStatusWords] = {};
  HIP_TRY(hipMemcpyAsync(st, e->status.ptr, sizeof st, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (st[2] != 0) {
    // A walk wave of the one-launch call waited in vain for its tree's set-up waves (workgroups
    // not dispatched in id order? the device held by another process's kernels for longer than
    // the poll's budget?).  The time-out has a status word of its own, so that an input error
    // in the same batch -- status[0] keeps the FIRST code -- cannot hide it.  This engine takes
    // the four-launch sequence from now on (a hipGraph captured BEFORE this point still replays
    // the one-launch kernel: re-capture it); a host-pointer entry point runs the call again at
    // once and returns its results (finish_host_call), a *_device caller gets the message
    // below.  (reduce_finalize has cleared every tree's hand-off word already; the memset makes
    // the "zero between calls" invariant independent of that.)
    e->fused_setup = false;
    e->fused_timed_out = true;
    if (e->ready.ptr) HIP_TRY(hipMemsetAsync(e->ready.ptr, 0, e->ready.bytes, s));
    if (st[0] == 0) {
      st[0] = kFusedTimeout;
      st[1] = st[2] - 1;
    }
  }
  if (st[0] != 0) {  // reported once: the first error since the last check
    HIP_TRY(hipMemsetAsync(e->status.ptr, 0, sizeof(int32_t) * kStatusWords, s));
    HIP_TRY(hipStreamSynchronize(s));
    // (a shard of a sharded handle reports the caller's tree index, not its own)
    return fail(std::string(status_message(st[0])) + " (tree " +
                std::to_string(st[1] + e->status_tree_offset) + ")");
  }
  return 0;
}

template <typename T>
int upload(Buffer& b, const T* host, size_t count, hipStream_t s) {
  if (b.ensure(sizeof(T) * std::max<size_t>(count, 1))) return 1;
  if (count) HIP_TRY(hipMemcpyAsync(b.ptr, host, sizeof(T) * count, hipMemcpyHostToDevice, s));
  return 0;
}

template <typename T>
int upload_staged(mi_engine* e, Buffer& b, const T* host, size_t count) {
  if (b.ensure(sizeof(T) * std::max<size_t>(count, 1))) return 1;
  if (!count) return 0;
  void* p = e->pinned.alloc(sizeof(T) * count, e->stream);
  if (!p) return fail("pinned staging allocation failed");
  memcpy(p, host, sizeof(T) * count);
  HIP_TRY(hipMemcpyAsync(b.ptr, p, sizeof(T) * count, hipMemcpyHostToDevice, e->stream));
  return 0;
}

int download(mi_engine* e, double* host, const Buffer& b, size_t count) {
  if (!host || !count) return 0;
  void* p = e->pinned.alloc(sizeof(double) * count, e->stream);
  if (!p) return fail("pinned staging allocation failed");
  HIP_TRY(hipMemcpyAsync(p, b.ptr, sizeof(double) * count, hipMemcpyDeviceToHost, e->stream));
  e->pinned.pending.push_back({host, p, sizeof(double) * count});
  return 0;
}

void add_block(std::map<std::string, std::pair<int, int>>& m, const std::string& k, int start,
               int len) {
  m[k] = {start, len};
}

hipStream_t pick_stream(mi_engine* e, void* stream) {
  return stream ? static_cast<hipStream_t>(stream) : e->stream;
}

}  // namespace

namespace {
const char kShardedDeviceCall[] =
    "device-pointer entry points need a single-device engine: one engine per device (one "
    "process per GPU), or the host-pointer entry points";
}

extern "C" {

int32_t mi_abi_version(void) { return MI_PHYLO_ABI_VERSION; }
const char* mi_last_error(void) { return g_error.c_str(); }
int32_t mi_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

// (device_tips / device_weights: the tips already on the device, mi_engine_create_device_tips)
static int32_t create_engine(const mi_engine_spec* spec, const double* exchangeabilities,
                             const double* frequencies, const int32_t* tip_states,
                             const double* tip_partials, const double* pattern_weights,
                             mi_engine** out_engine, const int32_t* device_tips = nullptr,
                             const double* device_weights = nullptr);

int32_t mi_engine_create_device_tips(const mi_engine_spec* spec, const int32_t* device_tip_states,
                                     const double* device_pattern_weights,
                                     mi_engine** out_engine) {
  if (!device_tip_states || !device_pattern_weights) return fail("null device pointer");
  return create_engine(spec, nullptr, nullptr, nullptr, nullptr, nullptr, out_engine,
                       device_tip_states, device_pattern_weights);
}

// Everything engine creation derives from the compact tip states, on the device: the int8
// states (rows padded with gaps to `stride`), and for 4-state engines the state masks, the
// table offsets of the third-generation walk and (use_tip_states == 0) the 0/1 partial vectors
// of SitePattern::GetPartials (site_pattern.cpp:117-131).
__global__ __launch_bounds__(256) void tips_prepare_kernel(const int32_t* in, int n, int P, int states,
                                                           int stride, int8_t* st8, uint8_t* masks,
                                                           uint8_t* codes, double* partials) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n * stride) return;
  const int x = (int)(idx / stride), p = (int)(idx - (size_t)x * stride);
  int c = states;
  if (p < P) {
    const int32_t v = in[(size_t)x * P + p];
    c = (v >= 0 && v < states) ? v : states;
  }
  st8[idx] = (int8_t)c;
  if (p >= P) return;
  const size_t i = (size_t)x * P + p;
  if (masks) masks[i] = c >= kStates ? 0xF : (uint8_t)(1u << c);
  if (codes) codes[i] = c >= kStates ? 64 : (uint8_t)(16 * c);
  if (partials)
    for (int k = 0; k < kStates; k++) partials[i * kStates + k] = (c >= kStates || c == k) ? 1.0 : 0.0;
}

int32_t mi_engine_create(const mi_engine_spec* spec, const int32_t* tip_states,
                         const double* tip_partials, const double* pattern_weights,
                         mi_engine** out_engine) {
  return create_engine(spec, nullptr, nullptr, tip_states, tip_partials, pattern_weights,
                       out_engine);
}

int32_t mi_engine_create_reversible(const mi_engine_spec* spec, const double* exchangeabilities,
                                    const double* frequencies, const int32_t* tip_states,
                                    const double* tip_partials, const double* pattern_weights,
                                    mi_engine** out_engine) {
  if (spec && spec->subst_model != MI_SUBST_REVERSIBLE)
    return fail("mi_engine_create_reversible needs subst_model == MI_SUBST_REVERSIBLE");
  if ((exchangeabilities == nullptr) != (frequencies == nullptr))
    return fail("pass both exchangeabilities and frequencies, or neither (built-in WAG)");
  return create_engine(spec, exchangeabilities, frequencies, tip_states, tip_partials,
                       pattern_weights, out_engine);
}

static int32_t create_engine(const mi_engine_spec* spec, const double* exchangeabilities,
                             const double* frequencies, const int32_t* tip_states,
                             const double* tip_partials, const double* pattern_weights,
                             mi_engine** out_engine, const int32_t* device_tips,
                             const double* device_weights) {
  if (!spec || !out_engine) return fail("null spec / out_engine");
  *out_engine = nullptr;
  if (spec->taxon_count < 3) return fail("need at least 3 taxa");
  if (spec->pattern_count < 1) return fail("need at least one site pattern");
  if (spec->state_count != kStates && spec->state_count != kAa)
    return fail("state_count must be 4 (DNA, as in the reference: substitution_model.cpp:6-15) "
                "or 20 (amino acids)");
  const int states = spec->state_count;
  if (states == kStates && spec->subst_model != MI_SUBST_JC69 && spec->subst_model != MI_SUBST_GTR)
    return fail("Substitution model not known");
  if (states == kAa && spec->subst_model != MI_SUBST_REVERSIBLE)
    return fail("a 20-state engine takes subst_model MI_SUBST_REVERSIBLE (an empirical model "
                "given as data; WAG when none is passed)");
  if (spec->site_model != MI_SITE_CONSTANT && spec->site_model != MI_SITE_WEIBULL)
    return fail("Site model not known");
  if (spec->clock_model != MI_CLOCK_NONE && spec->clock_model != MI_CLOCK_STRICT)
    return fail("Clock model not known");
  if (spec->site_model == MI_SITE_CONSTANT && spec->category_count != 1)
    return fail("the constant site model has exactly one rate category");
  if (spec->category_count < 1 || spec->category_count > kMaxCategories)
    return fail("category_count out of range (1..64)");
  if (!device_tips && !tip_states && !(spec->use_tip_states == 0 && tip_partials))
    return fail("tip_states is required");
  if (!device_tips && !pattern_weights) return fail("pattern_weights is required");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    return fail("no HIP device available: the MI355X engine has no CPU fallback");
  if (spec->device >= count) return fail("device ordinal out of range");
  if (spec->device >= 0) HIP_TRY(hipSetDevice(spec->device));

  mi_engine* e = new mi_engine();
  e->spec = *spec;
  if (spec->device < 0) HIP_TRY(hipGetDevice(&e->spec.device));
  e->n = spec->taxon_count;
  e->N = 2 * e->n - 1;
  e->P = spec->pattern_count;
  e->K = spec->category_count;
  e->s = states;
  if (states == kAa) {
    e->tiles = aa_tiles(e->P);
    e->ll_stride = aa_ll_blocks(e->P);
    // 32 MB per vector at 50 000 patterns x 4 categories, 16.4 GB per gradient tree of 512
    // taxa: the arena gets half of what the device has free now (an MI355X has 288 GB; more
    // trees per launch fill its 256 CUs better: 8 such trees in one launch instead of 3 + 3 + 2
    // are 8 % quicker), never less than 8 GB; aa_reserve backs off if that cannot be had
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)96 << 30;
    e->plv_budget = std::max<size_t>(free_b / 2, (size_t)8 << 30);
  } else {
    e->tiles = (e->P + kTile - 1) / kTile;
    e->ll_stride =
        std::max({e->tiles, loglik_mfma_tiles(e->P, e->K), gradient_mfma_tiles(e->P, e->K)});
  }
  int lg = 0;
  while ((2 << lg) <= e->n) lg++;
  e->max_slots = lg + 1;
  if (const char* env = getenv("MI_PHYLO_PLV_BYTES")) e->plv_budget = strtoull(env, nullptr, 10);
  if (const char* env = getenv("MI_PHYLO_SUBST_GRADIENT"))
    e->analytic_subst = std::string(env) == "analytic";
  // Which generation of the matrix-core gradient walk a call takes: the third (kernels_walk3.hip:
  // tip children looked up; one-hot / all-ones tips, at most four rate categories, no analytic
  // substitution gradient -- everything the reference produces) wherever it applies, else the
  // second (kernels_walk.hip: mask tips, any category count, analytic gradient).
  // MI_PHYLO_GRADIENT_WALK=v2 keeps every call on the second.  (The first generation,
  // gradient_mfma_kernel, was retired in round 6: the second had been ahead of it on every shape
  // but the arena shapes with fewer than three categories and a handful of tiles -- fluA: 0.321
  // against 0.335 ms per 1000 trees -- and those now take the third: 0.305 -> 0.29.)
  if (const char* env = getenv("MI_PHYLO_GRADIENT_WALK")) e->walk3 = std::string(env) != "v2";
  if (const char* env = getenv("MI_PHYLO_FUSED_SETUP")) e->fused_setup = std::string(env) != "0";
  if (const char* env = getenv("MI_PHYLO_DEBUG_FUSED_SKIP")) e->fused_debug_skip = atoi(env);
  // the one-launch call's hand-off (kernels_walk3.hip, walk_lut_body): none | l1 (default) | agent
  if (const char* env = getenv("MI_PHYLO_FUSED_FENCE"))
    e->fused_fence = std::string(env) == "none" ? 0 : (std::string(env) == "agent" ? 2 : 1);
  if (const char* env = getenv("MI_PHYLO_FUSED_COLOCATE")) e->fused_colocate = std::string(env) != "0";
  // how long a walk wave of the one-launch call polls before it gives up (testing; default 1 s)
  if (const char* env = getenv("MI_PHYLO_FUSED_SPIN_MS"))
    e->fused_spin_ticks = (int)std::min(2.0e9, std::max(0.01, atof(env)) * 1.0e5);
  // MI_PHYLO_WALK3_ARENA=0: arena-variant calls stay with the second / first generation (A/B)
  if (const char* env = getenv("MI_PHYLO_WALK3_ARENA")) e->walk3_arena = std::string(env) != "0";
  // MI_PHYLO_WALK3_K1=0: one-category calls with the vectors in LDS take the third generation only where the one-launch call applies (A/B)
  if (const char* env = getenv("MI_PHYLO_WALK3_K1")) e->walk3_k1_lds = std::string(env) != "0";
  if (const char* env = getenv("MI_PHYLO_GRADIENT_PATH")) {  // force one gradient kernel
    const std::string v(env);
    e->gradient_path = v == "hbm" ? 2 : v == "mfma" ? 3 : 0;
    e->allow_onchip_gradient = v != "hbm";
  }

  // BlockSpecification (block_specification.cpp:11-50, phylo_model.cpp:13-15)
  std::map<std::string, std::pair<int, int>> bm;
  int off = 0;
  e->rates_off = e->freqs_off = e->shape_off = e->clock_off = -1;
  if (spec->subst_model == MI_SUBST_GTR) {
    e->rates_off = off;
    add_block(bm, "GTR rates", off, 6);
    off += 6;
    e->freqs_off = off;
    add_block(bm, "frequencies", off, 4);
    off += 4;
  }
  add_block(bm, "entire substitution", 0, off);
  const int site_start = off;
  if (spec->site_model == MI_SITE_WEIBULL) {
    e->shape_off = off;
    add_block(bm, "Weibull shape", off, 1);
    off += 1;
  }
  add_block(bm, "entire site", site_start, off - site_start);
  const int clock_start = off;
  if (spec->clock_model == MI_CLOCK_STRICT) {
    e->clock_off = off;
    add_block(bm, "clock rate", off, 1);
    off += 1;
  }
  add_block(bm, "entire clock", clock_start, off - clock_start);
  add_block(bm, "entire", 0, off);
  e->param_count = off;
  for (const auto& kv : bm) e->blocks.push_back({kv.first, kv.second.first, kv.second.second});

  auto cleanup_fail = [&](int) {
    mi_engine_destroy(e);
    return 1;
  };
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
    return cleanup_fail(fail("hipStreamCreate failed"));
  if (e->status.ensure(sizeof(int32_t) * kStatusWords) ||
      hipMemsetAsync(e->status.ptr, 0, sizeof(int32_t) * kStatusWords, e->stream) != hipSuccess)
    return cleanup_fail(fail("status word allocation failed"));
  if (spec->site_model == MI_SITE_WEIBULL) {
    // what the Weibull site model needs of its quantiles, once per engine (kernels_setup.hip)
    if (e->weibull_x.ensure(sizeof(double) * 2 * (size_t)e->K)) return cleanup_fail(1);
    launch_weibull_table(e->K, e->weibull_x.as<double>(), e->stream);
  }
  const size_t np = (size_t)e->n * e->P;
  if (device_tips) {
    // (the two arrays must be device memory of THIS engine's device: a host pointer or another
    // device's memory would fault inside the preparation kernel, or worse, not fault)
    for (const void* p : {static_cast<const void*>(device_tips), static_cast<const void*>(device_weights)}) {
      hipPointerAttribute_t at{};
      if (hipPointerGetAttributes(&at, p) != hipSuccess || at.type != hipMemoryTypeDevice ||
          at.device != e->spec.device) {
        (void)hipGetLastError();
        return cleanup_fail(fail("device_tip_states / device_pattern_weights must be device memory of "
                                 "the engine's device"));
      }
    }
    // the tips are on the device already: one kernel derives what the host loops below derive
    const int stride = states == kAa ? e->tiles * kAaTile : e->P;
    const bool dna = states == kStates;
    if (e->tip_states.ensure((size_t)e->n * stride + (states == kAa ? kAaTipSlack : 0))) return cleanup_fail(1);
    if (states == kAa &&
        hipMemsetAsync(e->tip_states.as<int8_t>() + (size_t)e->n * stride, states, kAaTipSlack, e->stream) != hipSuccess)
      return cleanup_fail(fail("hipMemset failed"));
    if (dna && (e->tip_masks.ensure(np) || e->tip_codes.ensure(np + 16))) return cleanup_fail(1);
    if (dna && !spec->use_tip_states && e->tip_partials.ensure(sizeof(double) * np * kStates))
      return cleanup_fail(1);
    if (dna && hipMemsetAsync(e->tip_codes.ptr, 64, np + 16, e->stream) != hipSuccess)
      return cleanup_fail(fail("hipMemset failed"));
    const size_t total = (size_t)e->n * stride;
    hipLaunchKernelGGL(tips_prepare_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       e->stream, device_tips, e->n, e->P, states, stride,
                       e->tip_states.as<int8_t>(), dna ? e->tip_masks.as<uint8_t>() : nullptr,
                       dna ? e->tip_codes.as<uint8_t>() : nullptr,
                       dna && !spec->use_tip_states ? e->tip_partials.as<double>() : nullptr);
    e->have_tip_masks = e->have_tip_codes = dna;
    if (build_tip_tiles(e)) return cleanup_fail(1);
    if (states == kAa && aa_engine_init(e, exchangeabilities, frequencies)) return cleanup_fail(1);
    if (e->weights.ensure(sizeof(double) * (size_t)e->P) ||
        hipMemcpyAsync(e->weights.ptr, device_weights, sizeof(double) * (size_t)e->P,
                       hipMemcpyDeviceToDevice, e->stream) != hipSuccess)
      return cleanup_fail(fail("copy of the pattern weights failed"));
    if (hipStreamSynchronize(e->stream) != hipSuccess || hipGetLastError() != hipSuccess)
      return cleanup_fail(fail("preparation of the device-resident tips failed"));
    *out_engine = e;
    return 0;
  }
  std::vector<int8_t> st8(np, (int8_t)states);
  if (tip_states)
    for (size_t i = 0; i < np; i++) {
      const int32_t v = tip_states[i];
      st8[i] = (v >= 0 && v < states) ? (int8_t)v : (int8_t)states;
    }
  if (states == kAa) {
    // the 20-state kernels read compact states only; tip partials are accepted in the form
    // SitePattern::GetPartials produces (one-hot, or all ones for a gap: site_pattern.cpp:117-131)
    if (!spec->use_tip_states && tip_partials) {
      for (size_t i = 0; i < np; i++) {
        int ones = 0, last = 0;
        for (int x = 0; x < states; x++) {
          const double v = tip_partials[i * states + x];
          if (v == 1.0) { ones++; last = x; }
          else if (v != 0.0) ones = -states - 1;
        }
        if (ones == 1) st8[i] = (int8_t)last;
        else if (ones == states) st8[i] = (int8_t)states;
        else {
          mi_engine_destroy(e);
          return fail("20-state engine: tip partials must be one-hot or all ones");
        }
      }
    }
    {  // rows padded with gaps to whole 16-pattern tiles: the walk kernels read them unmasked
      const size_t stride = (size_t)e->tiles * kAaTile;
      // (+ a tile group of slack behind the last row: the workgroup kernels fetch the states of
      // their whole pattern range, padding waves included, with one LDS-DMA)
      std::vector<int8_t> padded((size_t)e->n * stride + kAaTipSlack, (int8_t)states);
      for (int x = 0; x < e->n; x++)
        std::copy(st8.begin() + (size_t)x * e->P, st8.begin() + (size_t)(x + 1) * e->P,
                  padded.begin() + (size_t)x * stride);
      if (upload(e->tip_states, padded.data(), padded.size(), e->stream)) return cleanup_fail(1);
    }
    if (aa_engine_init(e, exchangeabilities, frequencies)) return cleanup_fail(1);
  } else {
  if (upload(e->tip_states, st8.data(), np, e->stream)) return cleanup_fail(1);
  if (!spec->use_tip_states) {
    std::vector<double> tp(np * kStates);
    if (tip_partials) {
      std::copy(tip_partials, tip_partials + np * kStates, tp.begin());
    } else {
      for (size_t i = 0; i < np; i++)
        for (int x = 0; x < kStates; x++)
          tp[i * kStates + x] = (st8[i] >= kStates || st8[i] == x) ? 1.0 : 0.0;
    }
    if (upload(e->tip_partials, tp.data(), tp.size(), e->stream)) return cleanup_fail(1);
  }
  {
    // Tip vectors as state masks (bit s = compatible with state s) for the matrix-core
    // kernel: from the compact states, or from the partials when every entry is exactly 0
    // or 1 (what SitePattern::GetPartials produces, site_pattern.cpp:117-131); real-valued
    // partials have no mask form and take the HBM-streamed kernel.
    std::vector<uint8_t> masks(np);
    e->have_tip_masks = true;
    if (spec->use_tip_states || !tip_partials) {
      for (size_t i = 0; i < np; i++) masks[i] = st8[i] >= kStates ? 0xF : (uint8_t)(1u << st8[i]);
    } else {
      for (size_t i = 0; i < np && e->have_tip_masks; i++) {
        uint8_t m = 0;
        for (int x = 0; x < kStates; x++) {
          const double v = tip_partials[i * kStates + x];
          if (v == 1.0) m |= (uint8_t)(1u << x);
          else if (v != 0.0) e->have_tip_masks = false;
        }
        masks[i] = m;
      }
    }
    if (e->have_tip_masks && upload(e->tip_masks, masks.data(), np, e->stream)) return cleanup_fail(1);
    // The same tips as byte offsets of a state's entry in the third-generation walk's tip
    // tables (kernels_walk3.hip): 16 x state, 64 for the all-ones vector.  Only the five
    // vectors SitePattern produces have that form (site_pattern.cpp:117-131); an engine with
    // any other 0/1 vector keeps the mask kernels.  (+16 bytes: the kernel reads 12-byte
    // groups as three words.)
    if (e->have_tip_masks) {
      std::vector<uint8_t> codes(np + 16, 64);
      e->have_tip_codes = true;
      for (size_t i = 0; i < np && e->have_tip_codes; i++) {
        switch (masks[i]) {
          case 1: codes[i] = 0; break;
          case 2: codes[i] = 16; break;
          case 4: codes[i] = 32; break;
          case 8: codes[i] = 48; break;
          case 15: codes[i] = 64; break;
          default: e->have_tip_codes = false;
        }
      }
      if (e->have_tip_codes && upload(e->tip_codes, codes.data(), codes.size(), e->stream))
        return cleanup_fail(1);
    }
  }
  }
  if (build_tip_tiles(e)) return cleanup_fail(1);
  if (upload(e->weights, pattern_weights, (size_t)e->P, e->stream)) return cleanup_fail(1);
  if (hipStreamSynchronize(e->stream) != hipSuccess)
    return cleanup_fail(fail("upload of tips failed"));
  *out_engine = e;
  return 0;
}

void mi_engine_destroy(mi_engine* e) {
  if (!e) return;
  for (mi_engine* shard : e->shards) mi_engine_destroy(shard);
  e->shards.clear();
  if (e->stream || e->tip_states.ptr) (void)hipSetDevice(e->spec.device);
  if (e->stream) {
    (void)hipStreamSynchronize(e->stream);
  }
  for (Buffer* b :
       {&e->tip_states, &e->tip_partials, &e->tip_masks, &e->tip_tiles, &e->tip_code_tiles, &e->tip_codes, &e->weights, &e->tree_scratch, &e->sched, &e->macros,
        &e->arena_macros, &e->slot_need,
        &e->macro_count, &e->tip_tables, &e->mmats, &e->mphi, &e->x_sum, &e->bl_eff,
        &e->models, &e->mats, &e->ll_part, &e->plv, &e->g_part, &e->site_lik, &e->site_exp, &e->fin_scratch,
        &e->ll_sum, &e->g_sum, &e->status, &e->ready, &e->weibull_x, &e->aa_model, &e->aa_matP, &e->aa_matPT,
        &e->aa_tipP, &e->aa_tipPQ, &e->aa_exp_cum, &e->aa_exp_loc, &e->aa_root_val,
        &e->aa_root_exp, &e->aa_root_scale, &e->in_index, &e->in_weights, &e->out_reduced,
        &e->red_ll, &e->red_g, &e->red_site, &e->red_sort,
        &e->in_parent, &e->in_bl, &e->in_params, &e->in_rates, &e->in_rate_counts,
        &e->in_heights, &e->in_bounds, &e->in_ratios, &e->out_ll, &e->out_a, &e->out_b,
        &e->out_site, &e->out_subst, &e->in_pack, &e->out_pack})
    b->release();
  e->pinned.release();
  for (hipEvent_t ev : e->prof_events) (void)hipEventDestroy(ev);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int32_t mi_engine_param_count(const mi_engine* e) { return e ? e->param_count : -1; }
int32_t mi_engine_block_count(const mi_engine* e) { return e ? (int32_t)e->blocks.size() : -1; }
int32_t mi_engine_block(const mi_engine* e, int32_t index, const char** name, int32_t* start,
                        int32_t* length) {
  if (!e || index < 0 || index >= (int32_t)e->blocks.size()) return fail("block index out of range");
  if (name) *name = e->blocks[index].name.c_str();
  if (start) *start = e->blocks[index].start;
  if (length) *length = e->blocks[index].length;
  return 0;
}

int32_t mi_engine_reserve(mi_engine* e, int32_t tree_count, int32_t for_gradients) {
  if (!e) return fail("null engine");
  if (tree_count <= 0) return fail("tree_count must be positive");
  if (!e->shards.empty()) {
    const int D = (int)e->shards.size();
    for (int i = 0; i < D; i++) {
      int32_t b = 0, c = tree_count;
      if (e->shard_mode == MI_SHARD_TREES) mi_shard_range(tree_count, D, i, &b, &c);
      if (c > 0 && mi_engine_reserve(e->shards[i], c, for_gradients)) return 1;
    }
    return 0;
  }
  HIP_TRY(hipSetDevice(e->spec.device));
  if (e->s == kAa) {
    // a gradient engine may be asked for log-likelihoods too: those calls run more evaluations
    // per launch (fewer vectors each) and size the per-launch operand buffers accordingly
    // A back-off of the arena budget inside either reservation RELEASES every buffer that
    // scales with the budget -- those of the other shape too -- so both are repeated with the
    // reduced budget until a whole pass allocates without backing off: a later *_device call
    // then allocates nothing.  (A back-off also invalidates hipGraphs captured earlier on
    // this engine: their kernels point at released buffers.  include/mi_phylo.h says so.)
    for (int pass = 0; pass < 64; pass++) {
      const int before = e->aa_backoffs;
      if (for_gradients && aa_reserve(e, tree_count, true)) return 1;
      if (aa_reserve(e, tree_count, false)) return 1;
      if (e->aa_backoffs == before) return 0;
    }
    return fail("the partial-vector arena could not be reserved: the budget kept shrinking");
  }
  // Everything a later *_device call over `tree_count` trees can need -- with or without
  // rescaling, with the engine's substitution-gradient setting --, so that such a call
  // allocates nothing (it can then be captured in a hipGraph): the union of both rescaling
  // settings' workspaces (the HBM arena when either of them cannot use an on-chip kernel).
  const bool grad = for_gradients != 0;
  const bool onchip_plain = matrix_core_gradient(e, false);
  const bool onchip_rescaled = matrix_core_gradient(e, true);
  if (grad) {  // the fused reductions' per-tree buffers (mi_engine_gradients_unrooted_reduced*)
    if (e->red_ll.ensure(sizeof(double) * tree_count)) return 1;
    if (e->red_g.ensure(sizeof(double) * (size_t)tree_count * e->N)) return 1;
    if (e->red_site.ensure(sizeof(double) * tree_count)) return 1;
  }
  const bool analytic = e->analytic_subst && e->spec.subst_model == MI_SUBST_GTR;
  if (reserve(e, tree_count, grad, !onchip_plain, analytic && onchip_plain)) return 1;
  if (grad && onchip_plain != onchip_rescaled &&
      reserve(e, tree_count, grad, !onchip_rescaled, analytic && onchip_rescaled))
    return 1;
  return 0;
}

int32_t mi_engine_check_status(mi_engine* e, void* stream) {
  if (!e) return fail("null engine");
  if (!e->shards.empty()) {
    for (mi_engine* shard : e->shards) {
      const int rc = check_status(shard, shard->stream);
      shard->fused_timed_out = false;  // (reported to the caller: not a host-pointer call's to repeat)
      if (rc) return 1;
    }
    return 0;
  }
  const int rc = check_status(e, pick_stream(e, stream));
  e->fused_timed_out = false;  // (a device-pointer caller repeats the call itself)
  return rc;
}

static int32_t profile_begin(mi_engine* e, int32_t max_calls, bool phases) {
  if (!e) return fail("null engine");
  if (!e->shards.empty()) return profile_begin(e->shards[0], max_calls, phases);
  if (max_calls < 0) return fail("max_calls must be >= 0");
  HIP_TRY(hipSetDevice(e->spec.device));
  while ((int)e->prof_events.size() < kProfEvents * max_calls) {
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    e->prof_events.push_back(ev);
  }
  e->prof_capacity = max_calls;
  e->prof_used = 0;
  e->prof_phases = phases;
  return 0;
}

int32_t mi_engine_profile_begin(mi_engine* e, int32_t max_calls) {
  return profile_begin(e, max_calls, false);
}
int32_t mi_engine_profile_begin_phases(mi_engine* e, int32_t max_calls) {
  return profile_begin(e, max_calls, true);
}

int32_t mi_engine_profile_collect(mi_engine* e, double* out_ms, int32_t capacity,
                                  int32_t* out_count) {
  return mi_engine_profile_collect_phases(e, out_ms, nullptr, capacity, out_count, nullptr);
}

int32_t mi_engine_profile_collect_phases(mi_engine* e, double* out_ms, double* out_phase_ms,
                                         int32_t capacity, int32_t* out_count,
                                         int32_t* out_first_launch_evaluations) {
  if (!e) return fail("null engine");
  if (!e->shards.empty())
    return mi_engine_profile_collect_phases(e->shards[0], out_ms, out_phase_ms, capacity,
                                            out_count, out_first_launch_evaluations);
  if (out_phase_ms && !e->prof_phases)
    return fail("phase times were not recorded: use mi_engine_profile_begin_phases");
  const int count = std::min(e->prof_used, capacity);
  for (int i = 0; i < count; i++) {
    hipEvent_t* ev = &e->prof_events[(size_t)kProfEvents * i];
    HIP_TRY(hipEventSynchronize(ev[1]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
    if (out_ms) out_ms[i] = ms;
    if (out_phase_ms) {
      HIP_TRY(hipEventSynchronize(ev[6]));
      for (int k = 0; k < 4; k++) {
        HIP_TRY(hipEventElapsedTime(&ms, ev[2 + k], ev[3 + k]));
        out_phase_ms[4 * i + k] = ms;
      }
    }
  }
  if (out_count) *out_count = count;
  if (out_first_launch_evaluations) *out_first_launch_evaluations = e->prof_first_launch_evals;
  e->prof_capacity = 0;
  e->prof_used = 0;
  e->prof_phases = false;
  return 0;
}

int32_t mi_engine_last_call_info(const mi_engine* e, const char** dominant_kernel,
                                 int64_t* evaluations, int64_t* gradient_evaluations) {
  if (!e) return fail("null engine");
  if (!e->shards.empty())
    return mi_engine_last_call_info(e->shards[0], dominant_kernel, evaluations,
                                    gradient_evaluations);
  if (dominant_kernel) *dominant_kernel = e->dominant;
  if (evaluations) *evaluations = e->last_evals;
  if (gradient_evaluations) *gradient_evaluations = e->last_grad_evals;
  return 0;
}

const char* mi_engine_last_call_path(const mi_engine* e) {
  if (!e) return "";
  if (!e->shards.empty()) return mi_engine_last_call_path(e->shards[0]);
  return e->last_path.c_str();
}

int32_t mi_engine_last_call_launches(const mi_engine* e, int32_t* walk_launches,
                                     int32_t* arena_backoffs) {
  if (!e) return fail("null engine");
  if (!e->shards.empty())
    return mi_engine_last_call_launches(e->shards[0], walk_launches, arena_backoffs);
  if (walk_launches) *walk_launches = e->last_walk_launches;
  if (arena_backoffs) *arena_backoffs = e->aa_backoffs;
  return 0;
}

/* ---- device-pointer entry points ---------------------------------------- */

int32_t mi_engine_log_likelihoods_unrooted_device(mi_engine* e, void* stream, int32_t T,
                                                  const int32_t* parent_ids, const double* bl,
                                                  const double* params, int32_t rescaling,
                                                  double* out_ll) {
  if (!e) return fail("null engine");
  if (!e->shards.empty()) return fail(kShardedDeviceCall);
  DeviceCall d;
  d.T = T;
  d.rescaling = rescaling != 0;
  d.parent_ids = parent_ids;
  d.bl = bl;
  d.params = params;
  d.out_ll = out_ll;
  return run_device(e, pick_stream(e, stream), d);
}

int32_t mi_engine_gradients_unrooted_device(mi_engine* e, void* stream, int32_t T,
                                            const int32_t* parent_ids, const double* bl,
                                            const double* params, int32_t rescaling,
                                            double* out_ll, double* out_branch,
                                            double* out_site, double* out_subst) {
  if (!e) return fail("null engine");
  if (!e->shards.empty()) return fail(kShardedDeviceCall);
  if (!out_branch) return fail("null branch-gradient output");
  DeviceCall d;
  d.gradient = true;
  d.T = T;
  d.rescaling = rescaling != 0;
  d.parent_ids = parent_ids;
  d.bl = bl;
  d.params = params;
  d.out_ll = out_ll;
  d.out_branch = out_branch;
  d.out_site = out_site;
  d.out_subst = out_subst;
  return run_device(e, pick_stream(e, stream), d);
}

int32_t mi_engine_log_likelihoods_rooted_device(mi_engine* e, void* stream, int32_t T,
                                                const int32_t* parent_ids, const double* bl,
                                                const double* params, const double* rates,
                                                const double* heights, const double* bounds,
                                                int32_t with_jacobian, int32_t rescaling,
                                                double* out_ll) {
  if (!e) return fail("null engine");
  if (!e->shards.empty()) return fail(kShardedDeviceCall);
  if (with_jacobian && (!rates || !heights || !bounds))
    return fail("Attempted access of a time tree member that requires the time tree to be "
                "initialized. Have you set dates for your time trees, and initialized the "
                "time trees?");
  DeviceCall d;
  d.rooted = true;
  d.with_jacobian = with_jacobian != 0;
  d.T = T;
  d.rescaling = rescaling != 0;
  d.parent_ids = parent_ids;
  d.bl = bl;
  d.params = params;
  d.rates = rates;
  d.heights = heights;
  d.bounds = bounds;
  d.out_ll = out_ll;
  return run_device(e, pick_stream(e, stream), d);
}

int32_t mi_engine_gradients_rooted_device(mi_engine* e, void* stream, int32_t T,
                                          const int32_t* parent_ids, const double* bl,
                                          const double* params, const double* rates,
                                          const int32_t* rate_counts, const double* heights,
                                          const double* bounds, const double* ratios,
                                          int32_t rescaling, double* out_ll, double* out_ratios,
                                          double* out_clock, double* out_site,
                                          double* out_subst) {
  if (!e) return fail("null engine");
  if (!e->shards.empty()) return fail(kShardedDeviceCall);
  if (!rates || !rate_counts || !heights || !bounds || !ratios)
    return fail("Attempted access of a time tree member that requires the time tree to be "
                "initialized. Have you set dates for your time trees, and initialized the "
                "time trees?");
  if (!out_ratios || !out_clock) return fail("null gradient output");
  DeviceCall d;
  d.gradient = true;
  d.rooted = true;
  d.T = T;
  d.rescaling = rescaling != 0;
  d.parent_ids = parent_ids;
  d.bl = bl;
  d.params = params;
  d.rates = rates;
  d.rate_counts = rate_counts;
  d.heights = heights;
  d.bounds = bounds;
  d.ratios = ratios;
  d.out_ll = out_ll;
  d.out_ratios = out_ratios;
  d.out_clock = out_clock;
  d.out_site = out_site;
  d.out_subst = out_subst;
  return run_device(e, pick_stream(e, stream), d);
}

// rocPRIM is asked for its temporary-storage size once per (entries, key bits), not per call
static size_t reduce_workspace_bytes(mi_engine* e, int32_t T, int32_t index_count) {
  const long entries = (long)T * e->N;
  int bits = 1;
  while (bits < 32 && (1u << bits) <= (uint32_t)index_count) bits++;
  if (e->red_ws_entries != entries || e->red_ws_bits != bits) {
    e->red_ws_bytes = vi_reduce_workspace_bytes(entries, index_count);
    e->red_ws_entries = entries;
    e->red_ws_bits = bits;
  }
  return e->red_ws_bytes;
}

int32_t mi_engine_reserve_reduced(mi_engine* e, int32_t tree_count, int32_t index_count) {
  if (!e) return fail("null engine");
  if (!e->shards.empty()) return mi_engine_reserve(e, tree_count, 1);  // (host-pointer calls only)
  if (index_count < 0) return fail("index_count must be >= 0");
  if (mi_engine_reserve(e, tree_count, 1)) return 1;
  HIP_TRY(hipSetDevice(e->spec.device));
  if ((uint64_t)tree_count * (uint64_t)e->N > 0xffffffffull)
    return fail("the fused reductions number their (tree, node) entries in 32 bits: tree_count x "
                "(2 taxa - 1) must stay below 2^32");
  return e->red_sort.ensure(reduce_workspace_bytes(e, tree_count, index_count));
}

/* Engine::Gradients followed by the caller-side reductions of one variational-inference
 * step (vip/burrito.py:143-166, vip/branch_model.py:104-133,
 * src/unrooted_sbn_instance.cpp:176-198), fused behind the call: see include/mi_phylo.h. */
int32_t mi_engine_gradients_unrooted_reduced_device(
    mi_engine* e, void* stream, int32_t T, const int32_t* parent_ids, const double* bl,
    const double* params, int32_t rescaling, const int32_t* branch_index,
    const double* tree_weights, int32_t index_count, double* out_sums,
    double* out_index_gradient, double* out_ll) {
  if (!e) return fail("null engine");
  if (!e->shards.empty())
    return fail("device-pointer entry points need a single-device engine: one engine per "
                "device (one process per GPU), or the host-pointer entry points");
  if (!branch_index || !out_sums || index_count < 0 || (index_count > 0 && !out_index_gradient))
    return fail("null output / index");
  hipStream_t s = pick_stream(e, stream);
  HIP_TRY(hipSetDevice(e->spec.device));
  // per-tree results into the engine's own buffers unless the caller wants logL too
  if (e->red_ll.ensure(sizeof(double) * std::max(T, 1))) return 1;
  if (e->red_g.ensure(sizeof(double) * (size_t)std::max(T, 1) * e->N)) return 1;
  if (e->red_site.ensure(sizeof(double) * std::max(T, 1))) return 1;
  double* ll = out_ll ? out_ll : e->red_ll.as<double>();
  DeviceCall d;
  d.gradient = true;
  d.T = T;
  d.rescaling = rescaling != 0;
  d.parent_ids = parent_ids;
  d.bl = bl;
  d.params = params;
  d.out_ll = ll;
  d.out_branch = e->red_g.as<double>();
  d.out_site = e->K > 1 ? e->red_site.as<double>() : nullptr;
  d.out_subst = nullptr;  // (the 16 finite-difference passes are not part of this reduction)
  if (run_device(e, s, d)) return 1;
  ViReduceArgs ra{};
  ra.T = T;
  ra.N = e->N;
  ra.index_count = index_count;
  ra.ll = ll;
  ra.branch = e->red_g.as<double>();
  ra.site = e->K > 1 ? e->red_site.as<double>() : nullptr;
  ra.branch_index = branch_index;
  ra.tree_weights = tree_weights;
  ra.out_sums = out_sums;
  ra.out_index_gradient = out_index_gradient;
  if ((uint64_t)T * (uint64_t)e->N > 0xffffffffull)
    return fail("the fused reductions number their (tree, node) entries in 32 bits: tree_count x "
                "(2 taxa - 1) must stay below 2^32");
  const size_t ws = reduce_workspace_bytes(e, T, index_count);
  if (e->red_sort.ensure(ws)) return 1;
  if (launch_vi_reduce(ra, e->red_sort.ptr, ws, s)) return fail("the index sort of the reduction could not be launched");
  HIP_TRY(hipGetLastError());
  return 0;
}

int32_t mi_engine_create_sharded(const mi_engine_spec* spec, int32_t shard_count,
                                 const int32_t* devices, int32_t shard_mode,
                                 const double* exchangeabilities, const double* frequencies,
                                 const int32_t* tip_states, const double* tip_partials,
                                 const double* pattern_weights, mi_engine** out_engine) {
  if (!spec || !out_engine) return fail("null spec / out_engine");
  *out_engine = nullptr;
  if (shard_count <= 0) return fail("Thread count needs to be strictly positive.");  // engine.cpp:14-16
  if (shard_mode != MI_SHARD_TREES && shard_mode != MI_SHARD_PATTERNS)
    return fail("unknown shard mode");
  if (shard_mode == MI_SHARD_PATTERNS && shard_count > spec->pattern_count)
    return fail("more pattern shards than site patterns");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    return fail("no HIP device available: the MI355X engine has no CPU fallback");
  int current = 0;
  if (hipGetDevice(&current) != hipSuccess) current = 0;
  mi_engine* front = new mi_engine();
  front->spec = *spec;
  front->shard_mode = shard_mode;
  front->n = spec->taxon_count;
  front->N = 2 * front->n - 1;
  front->P = spec->pattern_count;
  front->K = spec->category_count;
  const int n = spec->taxon_count, s = spec->state_count;
  for (int i = 0; i < shard_count; i++) {
    mi_engine_spec sub = *spec;
    // NULL: round-robin over the visible devices STARTING AT THE CALLER'S CURRENT DEVICE (a
    // one-process-per-GPU launch that selected its device with hipSetDevice keeps it; a
    // negative ordinal -1 - k names "the k-th device from the current one" explicitly)
    int want = devices ? devices[i] : -1 - i;
    if (want < 0) want = (current + (-1 - want)) % count;
    sub.device = want;
    int32_t b = 0, c = spec->pattern_count;
    if (shard_mode == MI_SHARD_PATTERNS) mi_shard_range(spec->pattern_count, shard_count, i, &b, &c);
    sub.pattern_count = c;
    std::vector<int32_t> tips;
    std::vector<double> parts;
    if (shard_mode == MI_SHARD_PATTERNS) {  // columns [b, b + c) of the [taxon][pattern] arrays
      if (tip_states) {
        tips.resize((size_t)n * c);
        for (int x = 0; x < n; x++)
          std::copy(tip_states + (size_t)x * spec->pattern_count + b,
                    tip_states + (size_t)x * spec->pattern_count + b + c, tips.begin() + (size_t)x * c);
      }
      if (tip_partials) {
        parts.resize((size_t)n * c * s);
        for (int x = 0; x < n; x++)
          std::copy(tip_partials + ((size_t)x * spec->pattern_count + b) * s,
                    tip_partials + ((size_t)x * spec->pattern_count + b + c) * s,
                    parts.begin() + (size_t)x * c * s);
      }
    }
    mi_engine* shard = nullptr;
    const int rc = create_engine(
        &sub, exchangeabilities, frequencies,
        shard_mode == MI_SHARD_PATTERNS ? (tip_states ? tips.data() : nullptr) : tip_states,
        shard_mode == MI_SHARD_PATTERNS ? (tip_partials ? parts.data() : nullptr) : tip_partials,
        pattern_weights + b, &shard);
    if (rc) {
      mi_engine_destroy(front);
      return 1;
    }
    front->shards.push_back(shard);
  }
  const mi_engine* first = front->shards[0];
  front->param_count = first->param_count;
  front->blocks = first->blocks;
  front->s = first->s;
  *out_engine = front;
  return 0;
}

int32_t mi_engine_shard_count(const mi_engine* e) {
  return e ? std::max<int32_t>(1, (int32_t)e->shards.size()) : -1;
}

int32_t mi_engine_shard_device(const mi_engine* e, int32_t shard) {
  if (!e) return -1;
  if (e->shards.empty()) return shard == 0 ? e->spec.device : -1;
  if (shard < 0 || shard >= (int32_t)e->shards.size()) return -1;
  return e->shards[shard]->spec.device;
}

/* ---- host-pointer entry points ------------------------------------------ */

}  // extern "C"

namespace {

// One host-pointer call; `begin` stages the inputs, enqueues the device call and the
// downloads on the engine's stream, `finish_host_call` synchronises once and hands the
// staged outputs over.  A sharded handle begins the call on every shard before it finishes
// any, so the devices work side by side.
struct HostCall {
  bool gradient = false, rooted = false;
  int T = 0, rescaling = 0, with_jacobian = 0;
  const int32_t* parent_ids = nullptr;
  const double* bl = nullptr;
  const double* params = nullptr;
  const double* rates = nullptr;
  const int32_t* rate_counts = nullptr;
  const double* heights = nullptr;
  const double* bounds = nullptr;
  const double* ratios = nullptr;
  double* out_ll = nullptr;
  double* out_a = nullptr;  // branch gradient [T][N] (unrooted) / ratios [T][n-1] (rooted)
  double* out_b = nullptr;  // clock gradient [T][N-1] (rooted)
  double* out_site = nullptr;
  double* out_subst = nullptr;
  // fused reductions of a variational-inference step (mi_engine_gradients_unrooted_reduced)
  bool reduced = false;
  const int32_t* branch_index = nullptr;  // [T][N]
  const double* tree_weights = nullptr;   // [T] or null
  int index_count = 0;
  double* out_sum = nullptr;         // [2]: sum w logL, sum w site gradient
  double* out_index_grad = nullptr;  // [index_count]
};

// One DMA each way per host-pointer call (round 5; until then one per array: a memset, three to
// eight uploads and three to five downloads, each its own submission and its own turn on the
// stream -- 60 to 80 of the 930 microseconds of a 1000-tree DS1 call).  Inputs are packed into
// one pinned block and copied to one device block; the outputs live in one device block and
// come back as one copy, the pieces handed to the caller's arrays after the call's one
// synchronisation.  Pieces are 256-byte aligned.
struct InPiece {
  const void* host;
  size_t bytes;
  const void** dev;
};
struct OutPiece {
  double* host;  // may be null: not wanted
  size_t count;  // doubles
  double** dev;
};
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int upload_pack(mi_engine* e, std::initializer_list<InPiece> pieces) {
  size_t total = 0;
  for (const InPiece& p : pieces) total += p.host ? align256(p.bytes) : 0;
  if (e->in_pack.ensure(std::max<size_t>(total, 256))) return 1;
  char* pin = total ? static_cast<char*>(e->pinned.alloc(total, e->stream)) : nullptr;
  if (total && !pin) return fail("pinned staging allocation failed");
  size_t off = 0;
  for (const InPiece& p : pieces) {
    if (!p.host) {
      *p.dev = nullptr;
      continue;
    }
    memcpy(pin + off, p.host, p.bytes);
    *p.dev = static_cast<char*>(e->in_pack.ptr) + off;
    off += align256(p.bytes);
  }
  if (total) HIP_TRY(hipMemcpyAsync(e->in_pack.ptr, pin, total, hipMemcpyHostToDevice, e->stream));
  return 0;
}
// device addresses of the outputs (before the kernels are enqueued) ...
int place_out_pack(mi_engine* e, std::initializer_list<OutPiece> pieces) {
  size_t total = 0;
  for (const OutPiece& p : pieces) total += align256(sizeof(double) * p.count);
  if (e->out_pack.ensure(std::max<size_t>(total, 256))) return 1;
  size_t off = 0;
  for (const OutPiece& p : pieces) {
    *p.dev = reinterpret_cast<double*>(static_cast<char*>(e->out_pack.ptr) + off);
    off += align256(sizeof(double) * p.count);
  }
  return 0;
}
// ... and their one copy back (after them): the wanted pieces are delivered by finish_host_call
int download_pack(mi_engine* e, std::initializer_list<OutPiece> pieces) {
  size_t total = 0;
  for (const OutPiece& p : pieces) total += align256(sizeof(double) * p.count);
  if (!total) return 0;
  char* pin = static_cast<char*>(e->pinned.alloc(total, e->stream));
  if (!pin) return fail("pinned staging allocation failed");
  HIP_TRY(hipMemcpyAsync(pin, e->out_pack.ptr, total, hipMemcpyDeviceToHost, e->stream));
  size_t off = 0;
  for (const OutPiece& p : pieces) {
    if (p.host && p.count) e->pinned.pending.push_back({p.host, pin + off, sizeof(double) * p.count});
    off += align256(sizeof(double) * p.count);
  }
  return 0;
}

int begin_host_call(mi_engine* e, const HostCall& h) {
  const int T = h.T, n = e->n, N = e->N;
  if (T <= 0) return fail("tree_count must be positive");
  if (!h.parent_ids || !h.bl) return fail("null tree arrays");
  if (e->param_count > 0 && !h.params) return fail("null parameter matrix");
  HIP_TRY(hipSetDevice(e->spec.device));
  e->fused_timed_out = false;  // (what an earlier device-pointer call left unread is not this call's)
  e->pinned.reset();  // nothing of an earlier (possibly failed) call is delivered late
  // The status word is sticky (the *_device calls never clear it).  A host-pointer call
  // reports ITS OWN errors only: whatever an earlier device-pointer call left unread on this
  // engine's stream is dropped here, not blamed on this batch.
  HIP_TRY(hipMemsetAsync(e->status.ptr, 0, sizeof(int32_t) * kStatusWords, e->stream));
  const size_t np = h.rooted ? 2 * n - 2 : 2 * n - 3, nb = np + 1;
  const bool tt = h.rooted && h.rates && h.heights && h.bounds;
  const bool gtr = e->spec.subst_model == MI_SUBST_GTR;
  if (h.rooted && h.gradient) {
    if (!tt || !h.rate_counts || !h.ratios) return fail("null time-tree arrays");
    for (int t = 0; t < T; t++)
      if (h.rate_counts[t] != 1 && h.rate_counts[t] != N - 1)
        return fail(status_message(kBadRateCount));
  }
  const void *d_parent, *d_bl, *d_params, *d_rates, *d_heights, *d_bounds, *d_counts, *d_ratios,
      *d_index, *d_weights;
  const bool time_tree = tt && (h.gradient || true);
  if (upload_pack(e, {{h.parent_ids, sizeof(int32_t) * (size_t)T * np, &d_parent},
                      {h.bl, sizeof(double) * (size_t)T * nb, &d_bl},
                      {e->param_count > 0 ? h.params : nullptr, sizeof(double) * (size_t)T * e->param_count, &d_params},
                      {time_tree ? h.rates : nullptr, sizeof(double) * (size_t)T * (N - 1), &d_rates},
                      {time_tree ? h.heights : nullptr, sizeof(double) * (size_t)T * N, &d_heights},
                      {time_tree ? h.bounds : nullptr, sizeof(double) * (size_t)T * N, &d_bounds},
                      {h.rooted && h.gradient ? h.rate_counts : nullptr, sizeof(int32_t) * (size_t)T, &d_counts},
                      {h.rooted && h.gradient ? h.ratios : nullptr, sizeof(double) * (size_t)T * (n - 1), &d_ratios},
                      {h.reduced ? h.branch_index : nullptr, sizeof(int32_t) * (size_t)T * N, &d_index},
                      {h.reduced ? h.tree_weights : nullptr, sizeof(double) * (size_t)T, &d_weights}}))
    return 1;
  // (an engine without parameters still hands the kernels a valid pointer)
  if (!d_params) d_params = e->in_pack.ptr;
  auto P32 = [](const void* p) { return static_cast<const int32_t*>(p); };
  auto F64 = [](const void* p) { return static_cast<const double*>(p); };
  double *o_ll, *o_a, *o_b, *o_site, *o_subst, *o_sum, *o_index;
  if (!h.gradient) {
    const std::initializer_list<OutPiece> outs = {{h.out_ll, (size_t)T, &o_ll}};
    if (place_out_pack(e, outs)) return 1;
    int rc;
    if (!h.rooted)
      rc = mi_engine_log_likelihoods_unrooted_device(e, e->stream, T, P32(d_parent), F64(d_bl),
                                                     F64(d_params), h.rescaling, o_ll);
    else
      rc = mi_engine_log_likelihoods_rooted_device(e, e->stream, T, P32(d_parent), F64(d_bl),
                                                   F64(d_params), F64(d_rates), F64(d_heights),
                                                   F64(d_bounds), h.with_jacobian, h.rescaling, o_ll);
    if (rc) return 1;
    return download_pack(e, outs);
  }
  const bool site = e->K > 1, want_site = site && h.out_site, want_subst = gtr && h.out_subst;
  if (!h.rooted && h.reduced) {
    const std::initializer_list<OutPiece> outs = {{h.out_ll, (size_t)T, &o_ll},
                                                  {h.out_sum, 2, &o_sum},
                                                  {h.out_index_grad, (size_t)h.index_count, &o_index}};
    if (place_out_pack(e, outs)) return 1;
    if (mi_engine_gradients_unrooted_reduced_device(e, e->stream, T, P32(d_parent), F64(d_bl), F64(d_params),
                                                    h.rescaling, P32(d_index), F64(d_weights),
                                                    h.index_count, o_sum, o_index, o_ll))
      return 1;
    return download_pack(e, outs);
  }
  // (outputs nobody wants are neither computed -- NULL skips their work -- nor copied)
  const size_t a_count = h.rooted ? (size_t)T * (n - 1) : (size_t)T * N;
  const std::initializer_list<OutPiece> outs = {{h.out_ll, (size_t)T, &o_ll},
                                                {h.out_a, a_count, &o_a},
                                                {h.out_b, h.rooted ? (size_t)T * (N - 1) : 0, &o_b},
                                                {h.out_site, want_site ? (size_t)T : 0, &o_site},
                                                {h.out_subst, want_subst ? (size_t)T * 8 : 0, &o_subst}};
  if (place_out_pack(e, outs)) return 1;
  int rc;
  if (!h.rooted)
    rc = mi_engine_gradients_unrooted_device(e, e->stream, T, P32(d_parent), F64(d_bl), F64(d_params),
                                             h.rescaling, o_ll, o_a, want_site ? o_site : nullptr,
                                             want_subst ? o_subst : nullptr);
  else
    rc = mi_engine_gradients_rooted_device(e, e->stream, T, P32(d_parent), F64(d_bl), F64(d_params),
                                           F64(d_rates), P32(d_counts), F64(d_heights), F64(d_bounds),
                                           F64(d_ratios), h.rescaling, o_ll, o_a, o_b,
                                           want_site ? o_site : nullptr, want_subst ? o_subst : nullptr);
  if (rc) return 1;
  return download_pack(e, outs);
}

// End of a host-pointer call: one synchronisation (inside check_status), then the staged
// outputs are copied to the caller's buffers.  A time-out of the one-launch call (see
// check_status) does not reach the caller: the call is run again, now through the four-launch
// sequence -- fresh launches in the same process, nothing else is restarted -- and ITS results
// and errors are what the caller gets (the reference never fails spuriously:
// src/engine.cpp:54-92).
int finish_host_call(mi_engine* e, const HostCall& h) {
  int rc = check_status(e, e->stream);
  if (e->fused_timed_out) {
    e->fused_timed_out = false;
    e->fused_fallbacks++;
    e->pinned.reset();
    rc = begin_host_call(e, h);
    if (rc == 0) rc = check_status(e, e->stream);
    e->fused_timed_out = false;
  }
  if (rc == 0) e->pinned.flush();
  e->pinned.reset();
  return rc;
}

// A sharded handle: trees dealt to the shards in contiguous blocks (what
// FatBeagleParallelize's work queue does with thread_count FatBeagles,
// fat_beagle.hpp:119-149), or -- few trees, very long alignments -- every shard evaluates
// all trees on its own block of site patterns and the per-tree results, sums over
// patterns every one of them, are added in shard order.
int run_sharded(mi_engine* e, const HostCall& h) {
  const int D = (int)e->shards.size(), T = h.T;
  const int n = e->n, N = e->N;
  if (T <= 0) return fail("tree_count must be positive");
  if (e->shard_mode == MI_SHARD_TREES) {
    std::vector<int> started;
    std::vector<HostCall> calls(D);
    int rc = 0;
    for (int i = 0; i < D && !rc; i++) {
      int32_t b = 0, c = 0;
      mi_shard_range(T, D, i, &b, &c);
      if (c == 0) continue;
      HostCall s = h;
      s.T = c;
      const size_t np = h.rooted ? 2 * n - 2 : 2 * n - 3, nb = np + 1;
      s.parent_ids = h.parent_ids + (size_t)b * np;
      s.bl = h.bl + (size_t)b * nb;
      if (h.params) s.params = h.params + (size_t)b * e->param_count;
      if (h.rates) s.rates = h.rates + (size_t)b * (N - 1);
      if (h.rate_counts) s.rate_counts = h.rate_counts + b;
      if (h.heights) s.heights = h.heights + (size_t)b * N;
      if (h.bounds) s.bounds = h.bounds + (size_t)b * N;
      if (h.ratios) s.ratios = h.ratios + (size_t)b * (n - 1);
      if (h.out_ll) s.out_ll = h.out_ll + b;
      if (h.out_a) s.out_a = h.out_a + (size_t)b * (h.rooted ? n - 1 : N);
      if (h.out_b) s.out_b = h.out_b + (size_t)b * (N - 1);
      if (h.out_site) s.out_site = h.out_site + b;
      if (h.out_subst) s.out_subst = h.out_subst + (size_t)b * 8;
      if (h.reduced) {
        s.branch_index = h.branch_index + (size_t)b * N;
        if (h.tree_weights) s.tree_weights = h.tree_weights + b;
        e->shard_sums.resize((size_t)D * (2 + h.index_count));
        s.out_sum = e->shard_sums.data() + (size_t)i * (2 + h.index_count);
        s.out_index_grad = s.out_sum + 2;
      }
      e->shards[i]->status_tree_offset = b;
      calls[i] = s;
      rc = begin_host_call(e->shards[i], s);
      started.push_back(i);
    }
    for (int i : started) rc |= finish_host_call(e->shards[i], calls[i]);
    if (rc) return 1;
    if (h.reduced) {  // partial sums added in shard order: deterministic
      h.out_sum[0] = h.out_sum[1] = 0;
      for (int k = 0; k < h.index_count; k++) h.out_index_grad[k] = 0;
      for (int i : started) {
        const double* s = e->shard_sums.data() + (size_t)i * (2 + h.index_count);
        h.out_sum[0] += s[0];
        h.out_sum[1] += s[1];
        for (int k = 0; k < h.index_count; k++) h.out_index_grad[k] += s[2 + k];
      }
    }
    return 0;
  }
  // pattern shards: only what is a plain sum over site patterns
  if (h.rooted)
    return fail("pattern-sharded engines evaluate unrooted calls only (the log-det-Jacobian "
                "and the rooted chain rule are not sums over site patterns)");
  const size_t per = (size_t)T * (1 + (h.gradient ? N + 1 + 8 : 0)) + 2 + h.index_count;
  e->shard_sums.assign((size_t)D * per, 0.0);
  int rc = 0, started = 0;
  std::vector<HostCall> calls(D);
  for (int i = 0; i < D && !rc; i++, started++) {
    double* base = e->shard_sums.data() + (size_t)i * per;
    HostCall s = h;
    s.out_ll = base;
    if (h.gradient) {
      s.out_a = base + T;
      s.out_site = h.out_site ? base + (size_t)T * (1 + N) : nullptr;
      s.out_subst = h.out_subst ? base + (size_t)T * (2 + N) : nullptr;
    }
    if (h.reduced) {
      s.out_sum = base + (size_t)T * (1 + (h.gradient ? N + 1 + 8 : 0));
      s.out_index_grad = s.out_sum + 2;
    }
    calls[i] = s;
    rc = begin_host_call(e->shards[i], s);
  }
  for (int i = 0; i < started; i++) rc |= finish_host_call(e->shards[i], calls[i]);
  if (rc) return 1;
  auto add = [&](double* out, size_t off, size_t count) {
    if (!out) return;
    for (size_t k = 0; k < count; k++) {
      double sum = 0;
      for (int i = 0; i < D; i++) sum += e->shard_sums[(size_t)i * per + off + k];
      out[k] = sum;
    }
  };
  add(h.out_ll, 0, T);
  if (h.gradient) {
    add(h.out_a, T, (size_t)T * N);
    if (e->K > 1) add(h.out_site, (size_t)T * (1 + N), T);
    if (e->spec.subst_model == MI_SUBST_GTR) add(h.out_subst, (size_t)T * (2 + N), (size_t)T * 8);
  }
  if (h.reduced) {
    const size_t off = (size_t)T * (1 + (h.gradient ? N + 1 + 8 : 0));
    add(h.out_sum, off, 2);
    add(h.out_index_grad, off + 2, h.index_count);
  }
  return 0;
}

int run_host(mi_engine* e, const HostCall& h) {
  if (!e) return fail("null engine");
  if (!e->shards.empty()) return run_sharded(e, h);
  if (begin_host_call(e, h)) {
    e->pinned.reset();
    return 1;
  }
  return finish_host_call(e, h);
}

}  // namespace

extern "C" {

int32_t mi_shard_range(int32_t total, int32_t shard_count, int32_t shard, int32_t* begin,
                       int32_t* count) {
  if (total < 0 || shard_count <= 0 || shard < 0 || shard >= shard_count)
    return fail("mi_shard_range: bad arguments");
  // sizes differ by at most one, the larger blocks first (libsbn_amd/sharding.py: tree_shard)
  const int32_t base = total / shard_count, extra = total % shard_count;
  if (begin) *begin = shard * base + (shard < extra ? shard : extra);
  if (count) *count = base + (shard < extra ? 1 : 0);
  return 0;
}

int32_t mi_engine_log_likelihoods_unrooted(mi_engine* e, int32_t T, const int32_t* parent_ids,
                                           const double* bl, const double* params,
                                           int32_t rescaling, double* out_ll) {
  if (!out_ll) return fail("null output");
  HostCall h;
  h.T = T;
  h.rescaling = rescaling;
  h.parent_ids = parent_ids;
  h.bl = bl;
  h.params = params;
  h.out_ll = out_ll;
  return run_host(e, h);
}

int32_t mi_engine_gradients_unrooted(mi_engine* e, int32_t T, const int32_t* parent_ids,
                                     const double* bl, const double* params, int32_t rescaling,
                                     double* out_ll, double* out_branch, double* out_site,
                                     double* out_subst) {
  if (!out_ll || !out_branch) return fail("null output");
  HostCall h;
  h.gradient = true;
  h.T = T;
  h.rescaling = rescaling;
  h.parent_ids = parent_ids;
  h.bl = bl;
  h.params = params;
  h.out_ll = out_ll;
  h.out_a = out_branch;
  h.out_site = out_site;
  h.out_subst = out_subst;
  return run_host(e, h);
}

int32_t mi_engine_gradients_unrooted_reduced(mi_engine* e, int32_t T, const int32_t* parent_ids,
                                             const double* bl, const double* params,
                                             int32_t rescaling, const int32_t* branch_index,
                                             const double* tree_weights, int32_t index_count,
                                             double* out_sums, double* out_index_gradient,
                                             double* out_ll) {
  if (!out_sums || !branch_index || index_count < 0 || (index_count > 0 && !out_index_gradient))
    return fail("null output / index");
  HostCall h;
  h.gradient = true;
  h.reduced = true;
  h.T = T;
  h.rescaling = rescaling;
  h.parent_ids = parent_ids;
  h.bl = bl;
  h.params = params;
  h.branch_index = branch_index;
  h.tree_weights = tree_weights;
  h.index_count = index_count;
  h.out_sum = out_sums;
  h.out_index_grad = out_index_gradient;
  h.out_ll = out_ll;
  return run_host(e, h);
}

int32_t mi_engine_log_likelihoods_rooted(mi_engine* e, int32_t T, const int32_t* parent_ids,
                                         const double* bl, const double* params,
                                         const double* rates, const double* heights,
                                         const double* bounds, int32_t with_jacobian,
                                         int32_t rescaling, double* out_ll) {
  if (!out_ll) return fail("null output");
  HostCall h;
  h.rooted = true;
  h.T = T;
  h.rescaling = rescaling;
  h.with_jacobian = with_jacobian;
  h.parent_ids = parent_ids;
  h.bl = bl;
  h.params = params;
  h.rates = rates;
  h.heights = heights;
  h.bounds = bounds;
  h.out_ll = out_ll;
  return run_host(e, h);
}

int32_t mi_engine_gradients_rooted(mi_engine* e, int32_t T, const int32_t* parent_ids,
                                   const double* bl, const double* params, const double* rates,
                                   const int32_t* rate_counts, const double* heights,
                                   const double* bounds, const double* ratios,
                                   int32_t rescaling, double* out_ll, double* out_ratios,
                                   double* out_clock, double* out_site, double* out_subst) {
  if (!out_ll || !out_ratios || !out_clock) return fail("null output");
  if (!rates || !rate_counts || !heights || !bounds || !ratios)
    return fail("Attempted access of a time tree member that requires the time tree to be "
                "initialized. Have you set dates for your time trees, and initialized the "
                "time trees?");
  HostCall h;
  h.gradient = true;
  h.rooted = true;
  h.T = T;
  h.rescaling = rescaling;
  h.parent_ids = parent_ids;
  h.bl = bl;
  h.params = params;
  h.rates = rates;
  h.rate_counts = rate_counts;
  h.heights = heights;
  h.bounds = bounds;
  h.ratios = ratios;
  h.out_ll = out_ll;
  h.out_a = out_ratios;
  h.out_b = out_clock;
  h.out_site = out_site;
  h.out_subst = out_subst;
  return run_host(e, h);
}

}  // extern "C"
