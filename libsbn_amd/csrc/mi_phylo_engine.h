// Private to libmi_phylo.so: the engine object behind include/mi_phylo.h and the helpers
// its translation units share (mi_phylo_engine.cpp: 4-state call sequence and the C ABI;
// mi_phylo_engine_aa.cpp: 20-state call sequence).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi_phylo.h"
#include "mi_phylo_kernels.h"

namespace miphylo {
int fail(const std::string& msg);  // sets mi_last_error()
void set_last_error(const std::string& msg);
}
using miphylo::fail;
using namespace miphylo;

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t err__ = (expr);                                                     \
    if (err__ != hipSuccess)                                                       \
      return fail(std::string("HIP error: ") + hipGetErrorString(err__) + " at " + \
                  __FILE__ + ":" + std::to_string(__LINE__));                      \
  } while (0)

// A device buffer that only ever grows.
struct Buffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return 0;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
    HIP_TRY(hipMalloc(&ptr, need));
    bytes = need;
    return 0;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
  template <typename T>
  T* as() const { return static_cast<T*>(ptr); }
};

// Pinned host staging for the host-pointer entry points: inputs are copied into it and
// DMA'd from there, outputs are DMA'd into it and copied out after the call's one
// synchronisation.  (hipMemcpyAsync on pageable memory is staged by the runtime, copy by
// copy and mostly synchronously: ~0.5 ms per call for 1000 DS1 trees, against ~0.1 ms.)
struct PinnedArena {
  char* ptr = nullptr;
  size_t bytes = 0, used = 0;
  struct Pending {
    void* host;
    const void* staged;
    size_t bytes;
  };
  std::vector<Pending> pending;
  // returns nullptr on failure; may synchronise `s` when it has to grow
  void* alloc(size_t need, hipStream_t s) {
    need = (need + 255) & ~(size_t)255;
    if (used + need > bytes) {
      // copies already issued from / into the old block must finish before it goes away;
      // pending outputs are delivered first
      if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
      flush();
      if (ptr) (void)hipHostFree(ptr);
      ptr = nullptr;
      bytes = 0;
      const size_t want = std::max<size_t>(2 * (used + need), 1 << 20);
      if (hipHostMalloc(reinterpret_cast<void**>(&ptr), want, hipHostMallocDefault) != hipSuccess)
        return nullptr;
      bytes = want;
      used = 0;
    }
    void* p = ptr + used;
    used += need;
    return p;
  }
  void flush() {  // after a synchronisation: hand the staged outputs to the caller
    for (const Pending& q : pending) memcpy(q.host, q.staged, q.bytes);
    pending.clear();
  }
  void reset() {
    pending.clear();
    used = 0;
  }
  void release() {
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    bytes = used = 0;
    pending.clear();
  }
};

struct Block {
  std::string name;
  int start, length;
};

inline const char* status_message(int code) {
  switch (code) {
    case kBadParentIds: return "parent id vector is not in the reference's post-order id form";
    case kNotBifurcating: return "expected a bifurcating tree (node.cpp:198,240)";
    case kNotTrifurcatingRoot:
      return "UnrootedTree::Detrifurcate given a non-trifurcating tree.";
    case kGtrFrequencies: return "GTR frequencies do not sum to 1 +/- 0.001!";
    case kGtrRates: return "GTR rates do not sum to 1 +/- 0.001!";
    case kBadRateCount:
      return "The number of rates should be equal to 1 (i.e. strict clock) or equal to the "
             "number of branches.";
    case kTooManySlots: return "internal error: evaluation schedule needs too many LDS slots";
    case kFusedTimeout:
      return "internal error: the walk waves of the one-launch call waited in vain for their "
             "tree's set-up waves (MI_PHYLO_FUSED_SETUP=0 selects the four-launch sequence)";
    default: return "unknown device status";
  }
}

constexpr int kStatusWords = 4;  // code, tree, 1 + tree of a one-launch time-out, spare

struct mi_engine {
  mi_engine_spec spec;
  int n, N, P, K, tiles, max_slots, ll_stride;
  int s = 4;  // states: 4 (kernels_{loglik,gradient}.hip) or 20 (kernels_aa.hip)
  int param_count, rates_off, freqs_off, shape_off, clock_off;
  std::vector<Block> blocks;
  hipStream_t stream = nullptr;
  // static device data
  Buffer tip_states, tip_partials, tip_masks, tip_codes, weights;
  Buffer tip_code_tiles;  // tip_codes by pattern tile of the look-up walk (launch_tip_code_tiles), if have_tip_codes and K <= 4
  Buffer tip_tiles;  // tip_masks by pattern tile of the matrix-core log-likelihood kernel (launch_tip_tiles), if have_tip_masks
  bool have_tip_masks = false;  // every tip vector is 0/1: the matrix-core kernel can run
  bool have_tip_codes = false;  // ... and one-hot or all ones: the third-generation walk can run
  // per-call workspace
  Buffer arena_macros, slot_need, tree_scratch, sched, macros, macro_count, bl_eff, models, mats, tip_tables, mmats, mphi, x_sum, ll_part, plv, g_part, site_lik, site_exp, fin_scratch,
      ll_sum, g_sum, status;
  Buffer weibull_x;  // [K][2] {x_k, log x_k} of the Weibull quantiles (once per engine)
  Buffer ready;  // [T] hand-off words of the one-launch small call (zero between calls)
  bool fused_setup = true;  // MI_PHYLO_FUSED_SETUP=0: always the four-launch sequence
  bool fused_timed_out = false;  // check_status found the one-launch call's time-out word set
  int fused_fallbacks = 0;       // host-pointer calls that were run again through four launches
  int fused_fence = 1;           // MI_PHYLO_FUSED_FENCE=none|l1|agent: 0 | 1 | 2 (FusedSetupArgs::fence)
  bool fused_colocate = true;    // MI_PHYLO_FUSED_COLOCATE=0: set-up waves in id order (round 5)
  int fused_spin_ticks = 0;      // MI_PHYLO_FUSED_SPIN_MS (testing): the walk waves' poll budget, 100 MHz ticks
  int fused_debug_skip = 0;      // MI_PHYLO_DEBUG_FUSED_SKIP=t+1: tree t's set-up never reports (testing)
  // 20-state path: the engine's eigensystem and the streamed workspace (the arena is `plv`)
  Buffer aa_model, aa_matP, aa_matPT, aa_tipP, aa_tipPQ, aa_exp_cum, aa_exp_loc,
      aa_root_val, aa_root_exp, aa_root_scale;
  bool aa_reserved_gradient = false;
  PinnedArena pinned;
  bool allow_onchip_gradient = true;
  bool analytic_subst = false;       // MI_PHYLO_SUBST_GRADIENT=analytic (opt-in)
  int gradient_path = 0;  // 0 auto, 2 hbm, 3 mfma (MI_PHYLO_GRADIENT_PATH)
  bool walk3 = true;      // third generation where it applies (tip children looked up; MI_PHYLO_GRADIENT_WALK=v2: off)
  bool walk3_k1_lds = true;  // one category, vectors in LDS: third generation for every batch size (MI_PHYLO_WALK3_K1=0: only where the one-launch call applies, the second generation beyond -- the rule until the tip codes were pre-tiled)
  bool walk3_arena = true;  // ... for arena-variant calls too (MI_PHYLO_WALK3_ARENA=0: off)
  int tile_regs = -1;  // look-up walk: the engine's tile width (0: default, 4: wide; -1: not decided yet -- engine_tile_regs)
  // a sharded handle (mi_engine_create_sharded): the per-device / per-shard engines it
  // drives; such a handle owns no device memory itself
  std::vector<mi_engine*> shards;
  int shard_mode = 0;
  std::vector<double> shard_sums;  // per-shard partial results (pattern shards, fused sums)
  // fused reductions (mi_engine_gradients_unrooted_reduced*)
  Buffer in_index, in_weights, out_reduced, red_ll, red_g, red_site, red_sort;
  long red_ws_entries = -1;  // what red_ws_bytes (the sort's workspace size) was computed for
  int red_ws_bits = 0;
  size_t red_ws_bytes = 0;
  // staging for the host-pointer entry points
  Buffer in_parent, in_bl, in_params, in_rates, in_rate_counts, in_heights, in_bounds,
      in_ratios, out_ll, out_a, out_b, out_site, out_subst;
  Buffer in_pack, out_pack;  // one block each way per host-pointer call (begin_host_call)
  size_t plv_budget = (size_t)8 << 30;
  // kernel timing (bench.py)
  std::vector<hipEvent_t> prof_events;  // kProfEvents per call: [begin, end, mark 0..4]
  int prof_capacity = 0, prof_used = 0;
  bool prof_phases = false;   // also record the phase marks (mi_engine_profile_begin_phases)
  int prof_first_launch_evals = 0;  // evaluations in the first walk launch of the last call
  // last-call info
  const char* dominant = "";
  std::string last_path;  // mi_engine_last_call_path
  int64_t last_evals = 0, last_grad_evals = 0;
  int status_tree_offset = 0;  // a shard's first tree in the caller's batch (error messages)
  int last_walk_launches = 1;  // chunks of evaluations the last call's walk kernels ran over
  int aa_backoffs = 0;         // times the 20-state arena budget was reduced (aa_reserve)
};


// Events per profiled call: the pair around the dominant kernel(s) (all launches of a chunked
// call) and, when phases are asked for, five marks: call start | first walk launch starts |
// its post-order part done | its pre-order / main part done | call end.
constexpr int kProfEvents = 7;
inline hipEvent_t prof_event(mi_engine* e, int which) {
  return e->prof_events[(size_t)kProfEvents * e->prof_used + which];
}
#define PROF_MARK(e, on, which, s)                                   \
  do {                                                               \
    if (on) HIP_TRY(hipEventRecord(prof_event(e, 2 + (which)), s)); \
  } while (0)

// One engine call with every pointer a device pointer (what the *_device entry points build).
struct DeviceCall {
  bool gradient = false, rooted = false, with_jacobian = false, rescaling = false;
  int T = 0;
  const int32_t* parent_ids = nullptr;
  const double* bl = nullptr;
  const double* params = nullptr;
  const double* rates = nullptr;
  const int32_t* rate_counts = nullptr;
  const double* heights = nullptr;
  const double* bounds = nullptr;
  const double* ratios = nullptr;
  double* out_ll = nullptr;
  double* out_branch = nullptr;
  double* out_ratios = nullptr;
  double* out_clock = nullptr;
  double* out_site = nullptr;
  double* out_subst = nullptr;
};

// mi_phylo_engine_aa.cpp
int aa_engine_init(mi_engine* e, const double* exchangeabilities, const double* frequencies);
int aa_reserve(mi_engine* e, int T, bool gradient);
int aa_run_device(mi_engine* e, hipStream_t s, const DeviceCall& d);
