// 20-state call sequence of the engine (BASELINE.json configs[4]); kernels in kernels_aa.hip.
//
// Per call, on one stream, no host synchronisation:
//   setup (tree schedules + site-model instances, shared with the 4-state path)
//   per chunk of evaluations that fits the arena budget:
//     aa_transition -> aa_post -> aa_root [-> aa_pre] -> aa_reduce
//   finalize (shared: PhyloGradient assembly, rooted chain rule)
// The substitution model is engine-level data (an empirical model has no free parameters),
// decomposed once at engine creation by aa_model_setup_kernel.
#include <cstdlib>

#include "aa_tables.h"
#include "mi_phylo_engine.h"

namespace {

size_t aa_arena_bytes_per_eval(const mi_engine* e, bool gradient) {
  const size_t nodes = gradient ? e->n - 1 : e->max_slots;
  return nodes * e->K * (size_t)e->tiles * kAaTileDoubles * sizeof(double);
}

int aa_chunk(const mi_engine* e, int T, bool gradient) {
  const size_t per = aa_arena_bytes_per_eval(e, gradient);
  return (int)std::max<size_t>(1, std::min<size_t>(T, e->plv_budget / per));
}

}  // namespace

int aa_engine_init(mi_engine* e, const double* exchangeabilities, const double* frequencies) {
  // upper triangle row by row, as the reference's GTR rates (substitution_model.cpp:39-55)
  std::vector<double> ex(kAa * (kAa - 1) / 2), fr(kAa);
  if (exchangeabilities && frequencies) {
    std::copy(exchangeabilities, exchangeabilities + ex.size(), ex.begin());
    std::copy(frequencies, frequencies + kAa, fr.begin());
  } else {
    mi_wag_model(ex.data(), fr.data());
  }
  for (double x : ex)
    if (!(x >= 0)) return fail("exchangeabilities must be non-negative");
  for (double x : fr)
    if (!(x > 0)) return fail("frequencies must be positive");
  Buffer dex, dfr;
  if (dex.ensure(sizeof(double) * ex.size()) || dfr.ensure(sizeof(double) * kAa)) return 1;
  if (e->aa_model.ensure(sizeof(AaModel)) || e->status.ensure(sizeof(int32_t) * 2)) return 1;
  HIP_TRY(hipMemcpyAsync(dex.ptr, ex.data(), sizeof(double) * ex.size(), hipMemcpyHostToDevice,
                         e->stream));
  HIP_TRY(hipMemcpyAsync(dfr.ptr, fr.data(), sizeof(double) * kAa, hipMemcpyHostToDevice,
                         e->stream));
  HIP_TRY(hipMemsetAsync(e->status.ptr, 0, sizeof(int32_t) * 2, e->stream));
  launch_aa_model_setup(dex.as<double>(), dfr.as<double>(), e->aa_model.as<AaModel>(),
                        e->status.as<int32_t>(), e->stream);
  int32_t st[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(st, e->status.ptr, sizeof st, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  dex.release();
  dfr.release();
  if (st[0] != 0) return fail("model frequencies do not sum to 1 +/- 0.001!");
  return 0;
}

int aa_reserve(mi_engine* e, int T, bool gradient) {
  const int n = e->n, N = e->N, K = e->K;
  const size_t tiles = e->tiles, tp = tiles * kAaTile;
  if (e->tree_scratch.ensure(sizeof(int32_t) * (size_t)T * 13 * N)) return 1;
  if (e->sched.ensure(sizeof(SchedEntry) * (size_t)T * (n - 1))) return 1;
  if (e->macro_count.ensure(sizeof(int32_t) * (size_t)T)) return 1;
  if (e->bl_eff.ensure(sizeof(double) * (size_t)T * N)) return 1;
  if (e->models.ensure(sizeof(DevModel) * (size_t)T)) return 1;
  if (e->ll_part.ensure(sizeof(double) * (size_t)T * e->ll_stride)) return 1;
  if (e->fin_scratch.ensure(sizeof(double) * (size_t)T * 6 * n)) return 1;
  if (e->status.ensure(sizeof(int32_t) * 2)) return 1;
  if (e->ll_sum.ensure(sizeof(double) * (size_t)T)) return 1;
  // The partial-vector arena and what scales with it: as many evaluations per launch as the
  // budget allows (default: half of the device memory that was free when the engine was made;
  // MI_PHYLO_PLV_BYTES).  If the device cannot give that much after all -- other engines,
  // other users of the GPU -- the budget is halved until the allocation succeeds.
  auto alloc_chunk = [&](size_t chunk) -> int {
    if (e->plv.ensure(aa_arena_bytes_per_eval(e, gradient) * chunk)) return 1;
    const size_t nodes = gradient ? n - 1 : e->max_slots;
    if (e->aa_exp_cum.ensure(sizeof(int32_t) * chunk * nodes * K * tp)) return 1;
    if (e->aa_matP.ensure(sizeof(double) * chunk * (n - 1) * K * kAaPack)) return 1;
    // (+1 KB: the LDS-DMA of the last tip table reads a whole number of 1 KB pieces)
    if (e->aa_tipP.ensure(sizeof(double) * (chunk * n * K * kAaTipTable + 128))) return 1;
    if (e->aa_root_val.ensure(sizeof(double) * chunk * K * tp)) return 1;
    if (e->aa_root_exp.ensure(sizeof(int32_t) * chunk * K * tp)) return 1;
    if (gradient) {
      if (e->aa_exp_loc.ensure(sizeof(int32_t) * chunk * (n - 1) * K * tp)) return 1;
      if (e->aa_matPT.ensure(sizeof(double) * chunk * (n - 1) * K * kAaPack)) return 1;
      if (e->aa_tipPQ.ensure(sizeof(double) * (chunk * n * K * kAaTipTable + 128))) return 1;
      if (e->aa_root_scale.ensure(sizeof(double) * chunk * K * tp)) return 1;
      if (e->g_part.ensure(sizeof(double) * chunk * K * (tiles / kAaPreTiles) * N)) return 1;
    }
    return 0;
  };
  // (Buffer::ensure only grows: before a retry with a smaller budget everything that scales
  // with the chunk is given back, or an arena allocated at the larger size would stay and the
  // retry could not succeed.)
  auto release_chunk = [&]() {
    for (Buffer* b : {&e->plv, &e->aa_exp_cum, &e->aa_matP, &e->aa_tipP, &e->aa_root_val,
                      &e->aa_root_exp, &e->aa_exp_loc, &e->aa_matPT, &e->aa_tipPQ,
                      &e->aa_root_scale, &e->g_part})
      b->release();
  };
  bool backed_off = false;
  const std::string error_before = mi_last_error();
  for (;;) {
    const size_t chunk = aa_chunk(e, T, gradient);
    if (alloc_chunk(chunk) == 0) break;
    (void)hipGetLastError();  // (out of memory is not sticky)
    if (chunk <= 1) return 1;
    release_chunk();
    backed_off = true;
    // what the device can give NOW bounds the next attempt as well (other engines of this
    // process, torch's or RCCL's allocators may have taken memory since the engine was made)
    size_t free_b = 0, total_b = 0;
    size_t next = aa_arena_bytes_per_eval(e, gradient) * (chunk / 2);
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 2 >= aa_arena_bytes_per_eval(e, gradient))
      next = std::min(next, free_b / 2);
    e->plv_budget = next;
    e->aa_backoffs++;
  }
  if (backed_off) miphylo::set_last_error(error_before);  // the failed attempts are not the call's error
  if (gradient && e->g_sum.ensure(sizeof(double) * (size_t)T * 2 * N)) return 1;
  return 0;
}

int aa_run_device(mi_engine* e, hipStream_t s, const DeviceCall& d) {
  if (d.T <= 0) return fail("tree_count must be positive");
  if (!d.parent_ids || !d.bl || !d.out_ll) return fail("null tree / output pointer");
  if (e->param_count > 0 && !d.params) return fail("null parameter matrix");
  const int n = e->n, N = e->N, T = d.T;
  if (aa_reserve(e, T, d.gradient)) return 1;

  TreeSetupArgs ts{};
  ts.n = n;
  ts.T = T;
  ts.rooted = d.rooted;
  ts.parent_ids = d.parent_ids;
  ts.bl = d.bl;
  ts.rates = (d.rooted && (d.gradient || d.with_jacobian)) ? d.rates : nullptr;
  ts.scratch = e->tree_scratch.as<int32_t>();
  ts.sched = e->sched.as<SchedEntry>();
  ts.macros = nullptr;
  ts.macro_count = e->macro_count.as<int32_t>();
  ts.bl_eff = e->bl_eff.as<double>();
  ts.status = e->status.as<int32_t>();
  ts.max_slots = e->max_slots;
  ts.need_slots = 1;
  ModelSetupArgs ms{};
  ms.T = T;
  ms.models_per_tree = 1;
  ms.subst = 0;  // the 4-state part of DevModel is unused here; only the site model is read
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
  launch_setup(ts, ms, s);

  if (prof) HIP_TRY(hipEventRecord(prof_event(e, 0), s));
  const int chunk = aa_chunk(e, T, d.gradient);
  int post_ring = 0, post_tiles = 0;
  bool first_chunk = true;
  e->prof_first_launch_evals = std::min(chunk, T);
  for (int off = 0; off < T; off += chunk) {
    const int evals = std::min(chunk, T - off);
    AaTransitionArgs tr{};
    tr.n = n;
    tr.N = N;
    tr.K = e->K;
    tr.eval_offset = off;
    tr.evals = evals;
    tr.gradient = d.gradient;
    tr.model = e->aa_model.as<AaModel>();
    tr.models = e->models.as<DevModel>();
    tr.bl_eff = e->bl_eff.as<double>();
    tr.matP = e->aa_matP.as<double>();
    tr.matPT = e->aa_matPT.as<double>();
    tr.tipP = e->aa_tipP.as<double>();
    tr.tipPQ = e->aa_tipPQ.as<double>();
    launch_aa_transition(tr, s);

    AaWalkArgs w{};
    w.n = n;
    w.N = N;
    w.P = e->P;
    w.K = e->K;
    w.tiles = e->tiles;
    w.eval_offset = off;
    w.evals = evals;
    w.gradient = d.gradient;
    w.slots = e->max_slots;
    w.ll_stride = e->ll_stride;
    w.sched = e->sched.as<SchedEntry>();
    w.model = tr.model;
    w.models = tr.models;
    w.matP = tr.matP;
    w.matPT = tr.matPT;
    w.tipP = tr.tipP;
    w.tipPQ = tr.tipPQ;
    w.tip_states = e->tip_states.as<int8_t>();
    w.weights = e->weights.as<double>();
    w.arena = e->plv.as<double>();
    w.exp_cum = e->aa_exp_cum.as<int32_t>();
    w.exp_loc = e->aa_exp_loc.as<int32_t>();
    w.root_val = e->aa_root_val.as<double>();
    w.root_exp = e->aa_root_exp.as<int32_t>();
    w.root_scale = d.gradient ? e->aa_root_scale.as<double>() : nullptr;
    w.ll_part = e->ll_part.as<double>();
    w.g_part = e->g_part.as<double>();
    w.ll_sum = e->ll_sum.as<double>();
    w.g_sum = e->g_sum.as<double>();
    PROF_MARK(e, marks && off == 0, 1, s);
    if (first_chunk) {  // (the path string describes the first -- largest -- launch of the call)
      post_ring = aa_post_ring_entries(w);
      post_tiles = aa_post_tiles_per_wave(w);
      first_chunk = false;
    }
    launch_aa_post(w, s);
    launch_aa_root(w, s);
    PROF_MARK(e, marks && off == 0, 2, s);
    if (d.gradient) launch_aa_pre(w, s);
    PROF_MARK(e, marks && off == 0, 3, s);
    launch_aa_reduce(w, s);
  }
  if (prof) HIP_TRY(hipEventRecord(prof_event(e, 1), s));
  e->dominant = d.gradient ? aa_pre_kernel_name() : aa_post_kernel_name();
  // (the stack tops of the walks live in LDS rings: kernels_aa.hip)
  e->last_path = std::string(e->dominant) + " store=hbm-arena" +
                 " post-tiles=" + std::to_string(post_tiles) + " post-ring=" + std::to_string(post_ring) +
                 (d.gradient ? " pre-ring=" + std::to_string(aa_pre_ring_entries()) : std::string()) +
                 " states=20 K=" + std::to_string(e->K);
  e->last_evals = T;
  e->last_grad_evals = d.gradient ? T : 0;
  e->last_walk_launches = (T + chunk - 1) / chunk;

  FinalizeArgs fa{};
  fa.n = n;
  fa.N = N;
  fa.T = T;
  fa.K = e->K;
  fa.tiles = e->tiles;
  fa.ll_tiles = 1;
  fa.ll_used = LlCounts{1, 1, 0, 0};
  fa.g_tiles = 1;
  fa.ll_part = e->ll_sum.as<double>();
  fa.g_part = e->g_sum.as<double>();
  fa.gradient = d.gradient;
  fa.rooted = d.rooted;
  fa.with_jacobian = d.with_jacobian;
  fa.gtr = 0;
  fa.site_fused = d.gradient && e->K > 1;
  fa.site_separate = 0;
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
  fa.out_subst = nullptr;
  fa.status = e->status.as<int32_t>();
  launch_finalize(fa, s);
  PROF_MARK(e, marks, 4, s);
  if (prof) e->prof_used++;
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int32_t mi_wag_model(double* exchangeabilities, double* frequencies) {
  if (!exchangeabilities || !frequencies) return fail("null output");
  // PAML's lower triangle S[i][j], i > j  ->  upper triangle row by row
  int li = 0;
  double S[kAa][kAa] = {};
  for (int i = 1; i < kAa; i++)
    for (int j = 0; j < i; j++) S[j][i] = kWagLower[li++];
  int ui = 0;
  for (int i = 0; i < kAa; i++)
    for (int j = i + 1; j < kAa; j++) exchangeabilities[ui++] = S[i][j];
  double sum = 0;
  for (int i = 0; i < kAa; i++) sum += kWagFreqs[i];
  for (int i = 0; i < kAa; i++) frequencies[i] = kWagFreqs[i] / sum;
  return 0;
}
