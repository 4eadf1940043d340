// Launch interface between the host engine (mi_phylo_engine.cpp) and the HIP
// kernels (kernels_*.hip).  Plain structs of device pointers.
#pragma once
#include <hip/hip_runtime.h>

#include "mi_phylo_device.h"

namespace miphylo {

// Evaluations ("virtual trees") of one engine call, numbered
//   [0, T)            main evaluation of tree t with its own model
//   [T, 17T)          GTR gradient only: 16 finite-difference log-likelihoods
//                     per tree (fat_beagle.cpp:400-465), eval T + 16 t + (j-1)
//                     uses model 17 t + j
//   [17T, 18T)        GTR + K>1 gradient only: site-model pass with the model the
//                     reference is left in after the finite differences (j = 16)
struct EvalMap {
  int T;
  int models_per_tree;  // 1, or kFdModels for a GTR gradient call
  __host__ __device__ void decode(int e, int& tree, int& model) const {
    if (e < T) {
      tree = e;
      model = e * models_per_tree;
    } else if (e < 17 * T) {
      tree = (e - T) / 16;
      model = tree * kFdModels + 1 + (e - T) % 16;
    } else {
      tree = e - 17 * T;
      model = tree * kFdModels + 16;
    }
  }
};

struct TreeSetupArgs {
  int n, T, rooted;
  const int32_t* parent_ids;  // [T][2n-3] unrooted, [T][2n-2] rooted
  const double* bl;           // [T][2n-2] unrooted, [T][2n-1] rooted
  const double* rates;        // [T][2n-2] or nullptr: branch length x rate
  int32_t* scratch;           // [T][13 N]
  SchedEntry* sched;          // [T][n-1]
  MacroEntry* macros;         // [T][max_macros(n)] (may be nullptr)
  int32_t* macro_count;       // [T]
  double* bl_eff;             // [T][N]
  int32_t* status;            // [2]: code, tree
  int max_slots;
  int use_lds;     // set by the launcher
  int need_slots;  // 0: no log-likelihood kernel will run on this batch (node-id order, no LDS slots)
  // arena calls: the slot assignment of launch_macro_slots done by the tree's set-up workgroup
  // (tree_setup_wg_kernel only; launch_setup says whether it was): outputs, or nullptr
  MacroEntry* arena_macros;  // [T][max_macros(n)]
  int32_t* slot_need;        // [T]
  int arena_sure;            // set by the launcher
};

struct ModelSetupArgs {
  int T, models_per_tree;
  int subst, site, K;
  int param_count, rates_off, freqs_off, shape_off;
  const double* params;  // [T][param_count]
  DevModel* models;      // [T * models_per_tree]
  int32_t* status;
  const double* weibull_x;  // [K][2]: {x_k, log x_k} of the Weibull quantiles (launch_weibull_table), or nullptr
};

struct TransitionArgs {
  int E, N, K;
  EvalMap map;
  const DevModel* models;
  const double* bl_eff;  // [T][N]
  double* mats;          // [E][N-1][K][16]
  double* tip_tables;    // [E][n][K][5][4]: per tip edge and state (4 = gap), the column of P; may be nullptr
  int n;
  int ev_skip_begin, ev_skip_end;  // evaluations [begin, end) are not walked at all: no matrices
  int eval_base;                   // this launch covers evaluations [eval_base, eval_base + E)
};

// Matrices in the order the second-generation gradient walk consumes them (kernels_walk.hip):
// gradient evaluations [eval_begin, eval_begin + count) of the call, written to
// mmats[0 .. count) (the caller passes the pointer of the first one).
struct TransitionMacroArgs {
  int n, N, K, count, eval_begin;
  EvalMap map;
  const DevModel* models;
  const double* bl_eff;        // [T][N]
  const MacroEntry* macros;    // [T][max_macros(n)] (the order the walk uses)
  const int32_t* macro_count;  // [T]
  double* mmats;               // [count][max_macros(n)][6][K][16]{f, tr}
  double* mphi;                // [count][max_macros(n)][6][K][16] or nullptr
};

struct LikArgs {
  int n, N, P, K, tiles;
  int lds_slots;    // PLV slots in LDS (set by the launcher)
  // second-generation gradient walk (set by its launcher): the launch's evaluations, how many
  // of them (the first ones) are walked by `walk_groups` waves that take several pattern
  // tiles each (tile g, g + walk_groups, ...); the others get a wave per tile
  int walk_evals, walk_big_evals, walk_groups;
  int store;           // matrix-core gradient kernels: 0 = the launcher decides, 1 = stored vectors in LDS, 2 = arena (the engine decides: its schedules must match)
  const uint8_t* tip_code_tiles;  // look-up walk: tip_codes by pattern tile in the form its LDS holds them (launch_tip_code_tiles), or nullptr
  const uint8_t* tip_tiles;  // loglik_mfma_kernel: tip_masks by pattern tile in the kernel's LDS layout, or nullptr
  int tile_regs;       // look-up walk: registers per vector = tile width (0: the default, kLlR; 4: wide tiles -- gradient_walk_tile_regs)
  int evals_per_wave;  // loglik_mfma_kernel: consecutive evaluations of one tree per wave (launcher)
  int kp;           // MFMA path: categories per instruction (1, 2 or 4; set by the launcher)
  int cat_groups;   // matrix-core gradient: groups of four categories (K > 4; set by the launcher)
  int ll_tiles;     // stride of ll_part per evaluation (>= partial sums any kernel writes)
  int g_tiles;      // stride of g_part per gradient evaluation = tiles of the gradient kernel used
  int eval_offset;  // first evaluation of this launch
  int grad_offset;  // gradient-workspace index of that evaluation
  EvalMap map;
  const DevModel* models;
  const SchedEntry* sched;
  const MacroEntry* macros;    // [T][max_macros(n)]
  const int32_t* macro_count;  // [T]
  const double* mats;
  const double* tip_tables;    // see TransitionArgs
  const double* mmats;         // see TransitionMacroArgs (indexed by gradient evaluation)
  const double* mphi;
  const int8_t* tip_states;    // [n][P]
  const uint8_t* tip_masks;    // [n][P] bit s: compatible with state s (matrix-core gradient)
  const uint8_t* tip_codes;    // [n][P] 16 x state, 64 = gap (third-generation walk: byte offset of the state's table entry)
  const double* tip_partials;  // [n][P][4] or nullptr
  const double* weights;       // [P]
  double* ll_part;             // [E][tiles]
  double* plv;                 // [Eg][n-1][K][tiles*64][4]   (gradient, v1)
  double* g_part;              // [Eg][tiles][2][N]
  double* site_lik;            // [Eg][tiles*64] per-pattern site likelihood (K > 4 gradient: from the logL pass)
  int32_t* site_exp;           // [Eg][tiles*64] its power of two when rescaling
  int32_t* status;             // [4]: code, tree (schedule does not fit the kernel's LDS slots), [2]: 1 + tree of a walk wave of the one-launch call that waited in vain
  const int32_t* slot_need;    // [T] arena gradient kernel: LDS slots each tree's schedule uses
  int lds_lo;                  // arena gradient kernel: this launch takes trees with lds_lo < need <= lds_slots
};

// How many logL partial sums each evaluation's walk kernel wrote (the kernels tile the
// patterns differently): `mid` for evaluations [mid_lo, mid_hi) -- the finite-difference
// passes of a GTR gradient call --, `ends` for the others.  Consumers sum exactly those.
struct LlCounts {
  int ends, mid, mid_lo, mid_hi;
  __host__ __device__ int of(long e) const { return (e >= mid_lo && e < mid_hi) ? mid : ends; }
};

struct FinalizeArgs {
  int n, N, T, K, tiles;
  int ll_tiles;  // stride of ll_part per evaluation
  LlCounts ll_used;
  int g_tiles;   // gradient partial sums per gradient evaluation
  int gradient, rooted, with_jacobian;
  int gtr, site_fused, site_separate;
  const double* ll_part;
  const double* g_part;
  const double* bl_eff;        // [T][N]
  const double* bl_raw;        // rooted: [T][N]
  const double* rates;         // rooted: [T][N-1]
  const int32_t* rate_counts;  // rooted gradient: [T]
  const double* node_heights;  // [T][N]
  const double* node_bounds;   // [T][N]
  const double* height_ratios; // [T][n-1]
  const SchedEntry* sched;     // [T][n-1]
  double* scratch;             // [T][6 n]
  double* out_ll;              // [T]
  double* out_branch;          // unrooted gradient: [T][N]
  double* out_ratios;          // rooted gradient: [T][n-1]
  double* out_clock;           // rooted gradient: [T][N-1]
  double* out_site;            // [T] or nullptr
  double* out_subst;           // [T][8] or nullptr
  int32_t* status;
  int use_lds;  // set by the launcher
  int32_t* clear_ready;  // reduce_finalize: the one-launch call's hand-off words, zeroed per tree (or nullptr)
};

// tree schedules (one wave per tree) and model instances (one thread each) in one launch
// (returns true when the arena's slot assignment was done in the same launch: a.arena_macros)
bool launch_setup(const TreeSetupArgs& a, const ModelSetupArgs& ms, hipStream_t s);
bool launch_tree_setup(const TreeSetupArgs& a, hipStream_t s);  // (as launch_setup)  // trees only
void launch_weibull_table(int K, double* table, hipStream_t s);  // once per engine: ModelSetupArgs::weibull_x
void launch_transition(const TransitionArgs& a, hipStream_t s);
// On-chip (LDS-resident) log-likelihood: evaluations [eval_offset, eval_offset+count)
void launch_loglik(const LikArgs& a, int count, bool rescale, int max_slots, hipStream_t s);
int loglik_mfma_tiles(int P, int K);
// the matrix-core log-likelihood kernel's tip bytes, pre-tiled once per engine (LikArgs::tip_tiles)
size_t loglik_tip_tiles_bytes(int n, int P, int K);
void launch_tip_tiles(const uint8_t* masks, uint8_t* tiles, int n, int P, int K, hipStream_t s);
int gradient_mfma_tiles(int P, int K, int regs = 0);  // regs: registers per vector (0: kLlR)
// the look-up walk's tip codes pre-tiled once per engine for its tile width (LikArgs::tip_code_tiles)
size_t tip_code_tiles_bytes(int n, int P, int K, int regs);
void launch_tip_code_tiles(const uint8_t* codes, uint8_t* out, int n, int P, int K, int regs, hipStream_t s);
// tile width of the look-up walk for an engine whose batches take the arena (kLlR or 4:
// kernels_walk3.hip, RR)
int gradient_walk_tile_regs(int n, int P, int K);
// Gradient, partial-likelihood vectors streamed through HBM (any tree size, rescaling)
void launch_gradient_hbm(const LikArgs& a, int count, bool rescale, hipStream_t s);
// The matrix-core gradient walks (kernels_walk.hip: second generation, kernels_walk3.hip: third;
// the first, gradient_mfma_kernel, was retired in round 6): all categories of a group of four
// per instruction; they also write the log-likelihood partial sums, so no separate logL pass is
// needed (K > 4: the site likelihoods come from a pass of the log-likelihood kernel).
int gradient_mfma_groups(int K);  // waves per pattern tile (category groups of four)
// subst: analytic substitution gradient statistics appended (kSubstExtra doubles)
int gradient_mfma_width(int n, bool subst = false);  // doubles per (gradient evaluation, tile) of its partial sums
// arena variant of the walks (stored post-order vectors in HBM, LDS slots reused): its slot
// assignment pass, and its HBM need per evaluation
void launch_macro_slots(const MacroEntry* macros_in, MacroEntry* macros_out,
                        const int32_t* macro_count, int n, int T, int32_t* need, int32_t* status,
                        hipStream_t s);
size_t gradient_arena_bytes_per_eval(int n, int P, int K);
int device_compute_units();  // of the current device
bool arena_single_launch(size_t lds_bytes, size_t waves);  // all waves resident at once?
int gradient_arena_slots_sure(int n);
int gradient_arena_slots_usual(int n);
// second generation of the same walk (kernels_walk.hip): macro-ordered operand streams
void launch_transition_macro(const TransitionMacroArgs& a, hipStream_t s);
void launch_gradient_walk(const LikArgs& a, int count, bool rescale, bool subst, hipStream_t s);
bool gradient_walk_fits(int n, int K, bool rescale);
// lut: the call runs the third-generation (look-up) walk, whose arena variant pays one step earlier
bool gradient_walk_use_arena(int n, int K, bool rescale, bool subst, size_t waves = (size_t)-1, bool lut = false,
                             int regs = 0);  // regs: the engine's tile width (0: default)
bool gradient_walk_batches_take_arena(int n, int K, bool lut);  // (large batch, no rescaling, whatever MI_PHYLO_GRADIENT_STORE says)
size_t gradient_walk_lds_bytes(int n, int K, bool rescale, bool subst, int regs = 0);
size_t gradient_walk_lds_bytes_for(int n, int K, bool rescale, bool subst, int slots, int regs = 0);
size_t gradient_walk_mats_bytes_per_eval(int n, int K);
const char* gradient_walk_kernel_name();
// kernels_walk3.hip: the third-generation walk (tip children are table look-ups)
size_t gradient_walk_lut_mats_bytes_per_eval(int n);
void launch_transition_lut(const TransitionMacroArgs& a, hipStream_t s);
bool gradient_walk_lut_applies(int K);
void launch_gradient_walk_lut(const LikArgs& a, int count, bool rescale, hipStream_t s);
// The one-launch small call (round 5): tree set-up, model instances and operand records as the
// FIRST workgroups of the walk's launch (four waves per tree), the walk waves wait for their
// tree's hand-off word `ready[t]` (zero before the launch; reduce_finalize clears it again).
// One model instance per tree, one evaluation per tree, trees of at most 64 nodes.
// (a word per 128-byte line: the 78 walk waves of a tree poll their tree's word, and all the
// words of a batch in a handful of lines made ONE memory channel serve every poll of the chip --
// the set-up waves' adds then queued behind them for 6 microseconds)
constexpr int kReadyStride = 32;
struct FusedSetupArgs {
  TreeSetupArgs ts;
  ModelSetupArgs ms;
  double* mmats;     // operand records of gradient evaluation 0 of the launch
  int32_t* ready;    // [T][kReadyStride]
  int setup_blocks;  // 4 T
  int spin_ticks;    // how long a walk wave polls its tree's word (100 MHz ticks; 0: the launcher's default, one second)
  int debug_skip;    // testing: tree debug_skip - 1 never becomes ready (0: off)
  int fence;         // hand-off: 0 none (round 5), 1 L1 + scalar-cache invalidate (default), 2 agent-scope release / acquire
  int colocate;      // set-up waves of a tree on the XCD of its walk waves
  int xcd_base;      // id of the first walk workgroup of the launch that will walk these trees, mod 8 (set by the launchers)
};
bool gradient_walk_lut_fused_applies(int n, int K);
void launch_gradient_walk_lut_fused(const LikArgs& a, const FusedSetupArgs& f, int count, bool rescale,
                                    hipStream_t s);
// The same set-up waves as a launch of their own (round 6): trees, model instances and operand
// records in ONE launch in front of the walk's, for batches beyond the one-launch call's size
// (instead of launch_setup + launch_transition_lut).  f.ready may be nullptr.
void launch_setup_records(const FusedSetupArgs& f, int count, hipStream_t s);
const char* gradient_walk_lut_kernel_name();
const char* gradient_walk_lut_fused_kernel_name();
// waves per CU the walks' LDS footprint allows for this tree size and category count
int gradient_walk_waves_per_cu(int n, int K);
// Sum of the per-tile partials: ll_sum[e] = sum_i ll_part[e][i] for e < E,
// g_sum[gi][2N] = sum_i g_part[gi][i][2N] for gi < Eg (fixed order: deterministic).
struct ReduceArgs {
  int N, E, Eg, ll_tiles, g_tiles;
  LlCounts ll_used;
  const double* ll_part;
  const double* g_part;
  double* ll_sum;
  double* g_sum;
  // positional partial sums (matrix-core gradient kernel): g_part is
  // [Eg][g_tiles][g_width] indexed by (macro, position, {branch, site}); the schedule maps
  // positions to node ids.  g_width == 0: g_part is [Eg][g_tiles][2N] indexed by node id.
  int g_width, n, T;
  const MacroEntry* macros;
  const int32_t* macro_count;
  // the last `extra` doubles of each positional row are plain sums (analytic substitution
  // gradient: 64 + 4), reduced into x_sum[Eg][extra]
  int extra;
  double* x_sum;
  int keep_offset;  // reduce_finalize_kernel (set by its launcher): where in LDS the reduced sums stay, in doubles
};

// Analytic substitution-model gradient (opt-in): one thread per tree turns the reduced
// statistics into d logL / d (stick-breaking coordinates), 5 rate then 3 frequency ones.
struct SubstGradArgs {
  int T, param_count, rates_off, freqs_off;
  const double* params;   // [T][param_count]
  const DevModel* models; // [T] (one model per tree in this mode)
  const double* x_sum;    // [T][68]
  double* out_subst;      // [T][8]
};
void launch_subst_gradient(const SubstGradArgs& a, hipStream_t s);
constexpr int kSubstExtra = 68;
// Reductions of a variational-inference step over the per-tree results of a gradient call:
//   out_sums[0] = sum_t w_t logL_t, out_sums[1] = sum_t w_t site_gradient_t,
//   out_index_gradient[k] = sum over (t, v) with branch_index[t][v] == k of w_t g[t][v]
// (w_t = 1 without tree_weights; negative indices are skipped), every sum in (t, v) order.
struct ViReduceArgs {
  int T, N, index_count;
  const double* ll;                // [T]
  const double* branch;            // [T][N]
  const double* site;              // [T] or nullptr
  const int32_t* branch_index;     // [T][N]
  const double* tree_weights;      // [T] or nullptr
  double* out_sums;                // [2]
  double* out_index_gradient;      // [index_count]
};
// mi_vi_reduce.hip: stable sort of the entries by index + one ordered sum per run.  The
// workspace (vi_reduce_workspace_bytes) holds the keys, the entry numbers and the sort's
// temporary storage; returns nonzero if the sort could not be enqueued.
size_t vi_reduce_workspace_bytes(long entries, int index_count);
int launch_vi_reduce(const ViReduceArgs& a, void* workspace, size_t workspace_bytes, hipStream_t s);
bool reduce_tiles_fits(int N);
void launch_reduce_tiles(const ReduceArgs& a, hipStream_t s);
void launch_finalize(const FinalizeArgs& a, hipStream_t s);
// both in one launch (a workgroup per tree) for calls with one evaluation per tree
void launch_reduce_finalize(const ReduceArgs& ra, const FinalizeArgs& fa, hipStream_t s);

// ------------------------------------------------------------------------
// 20-state path (kernels_aa.hip): partial-likelihood vectors streamed through HBM in
// 16-pattern matrix-core tiles, one wave per (evaluation, category, pattern block).
// ------------------------------------------------------------------------
constexpr int kAa = 20;              // states
constexpr int kAaTile = 16;          // site patterns per matrix-core tile
constexpr int kAaTipSlack = 512;     // bytes behind the tip-state rows (a workgroup fetches its whole pattern range)
constexpr int kAaTileDoubles = 320;  // one tile of one category: 5 registers x 64 lanes
// one 20x20 matrix as matrix-core A operands: rows 0..15 as five registers x 64 lanes (320
// doubles), rows 16..19 as five blocks of 16 values (a 4x4x4 A operand repeats its 16 values
// in each of the instruction's four blocks: round 5 stores them once) -- 400 doubles, in a
// stride of 512 = four 1 KB pieces of LDS-DMA (until round 5: 10 x 64 = 640, five pieces)
constexpr int kAaPack = 512;
constexpr int kAaPackRows16 = 320;   // offset of the rows-16..19 part
constexpr int kAaTipTable = 21 * 20; // per tip edge and category: column of the matrix per state, 20 = gap
constexpr int kAaPreTiles = 2;       // tiles a wave of the pre-order kernel takes (post-order: 2 or 4)

// The substitution model of a 20-state engine: one eigensystem per engine (an empirical
// model has no free parameters); the per-tree part is the site model in DevModel.
struct AaModel {
  double pi[kAa];
  double lambda[kAa];
  double Q[kAa * kAa];     // row-major, normalised to one expected substitution per unit time
  double V[kAa * kAa];     // eigenvectors
  double Vinv[kAa * kAa];  // inverse eigenvectors
  double Qpack[kAaPack];   // Q as matrix-instruction A operands (the pack layout above)
};

struct AaTransitionArgs {
  int n, N, K;
  int eval_offset, evals;   // this chunk of evaluations (evaluation == tree)
  int gradient;             // also the pre-order matrices
  const AaModel* model;
  const DevModel* models;   // [T]: category rates
  const double* bl_eff;     // [T][N]
  double* matP;             // [evals][n-1][K][kAaPack]  P of internal edges
  double* matPT;            // same: P^T (pre-order propagation)
  double* tipP;             // [evals][n][K][21][20]: tipP[x][i] = P[i][x], x = 20: 1
  double* tipPQ;            // same for P Q (x = 20: 0); internal edges need no P Q: the
                            // derivative uses Q (P L) with the model's one Q operand
};

struct AaWalkArgs {
  int n, N, P, K;
  int tiles;               // 16-pattern tiles, rounded up to a multiple of what one wave takes
  int eval_offset, evals;  // this chunk
  int gradient;            // post-order: every internal vector is kept, indexed by node
  int slots;               // log-likelihood only: vectors are kept by schedule slot
  int ring_slots;          // post-order kernel, both forms (set by the launcher): stack entries a wave keeps in LDS
  int pre_ring_slots;      // pre-order kernel (set by the launcher): parked vectors a wave keeps in LDS
  int ll_stride;           // stride of ll_part per evaluation
  const SchedEntry* sched; // [T][n-1]
  const AaModel* model;
  const DevModel* models;  // [T]
  const double* matP;
  const double* matPT;
  const double* tipP;
  const double* tipPQ;
  const int8_t* tip_states;  // [n][P], 20 = gap
  const double* weights;     // [P]
  double* arena;       // [evals][n-1 | slots][K][tiles][kAaTileDoubles]
  int32_t* exp_cum;    // [evals][n-1 | slots][K][tiles*16]: power of two removed below and at the node
  int32_t* exp_loc;    // [evals][n-1][K][tiles*16]: power of two removed at the node (gradient)
  double* root_val;    // [evals][K][tiles*16]: sum_i pi_i L_root[i] (scaled)
  int32_t* root_exp;   // [evals][K][tiles*16]
  double* root_scale;  // [evals][K][tiles*16]: w_p cw_k 2^(E_k - Emax) / site_p, the root's pre-order weight
  double* ll_part;     // [T][ll_stride]
  double* g_part;      // [evals][K][blocks][N]: per-edge sums of one wave
  double* ll_sum;      // [T]
  double* g_sum;       // [T][2][N]
};

void launch_aa_model_setup(const double* exchangeabilities, const double* freqs, AaModel* model,
                           int32_t* status, hipStream_t s);
void launch_aa_transition(const AaTransitionArgs& a, hipStream_t s);
int aa_tiles(int P);                 // padded tile count
int aa_ll_blocks(int P);             // partial log-likelihood sums per evaluation
void launch_aa_post(const AaWalkArgs& a, hipStream_t s);
void launch_aa_root(const AaWalkArgs& a, hipStream_t s);
void launch_aa_pre(const AaWalkArgs& a, hipStream_t s);
// entries of the LDS rings the two launchers above will use for these arguments
int aa_post_ring_entries(const AaWalkArgs& a);
int aa_post_tiles_per_wave(const AaWalkArgs& a);  // 16-pattern tiles a wave of the post-order kernel takes (1, 2 or 4)
int aa_pre_ring_entries();
void launch_aa_reduce(const AaWalkArgs& a, hipStream_t s);
const char* aa_post_kernel_name();
const char* aa_pre_kernel_name();

const char* loglik_kernel_name(const LikArgs& a, bool rescale, int max_slots);
const char* gradient_kernel_name();

}  // namespace miphylo
