/* mi_phylo.h -- C ABI of the MI355X-native phylogenetic likelihood + gradient
 * engine.  This is the drop-in boundary for libsbn's Engine / FatBeagle path:
 * plain pointers and sizes, no C++ or torch types.
 *
 * What it replaces in the reference (paths relative to the libsbn repo root):
 *   Engine::Engine / FatBeagle::FatBeagle / CreateInstance / SetTipStates /
 *   SetTipPartials                     src/engine.cpp:10-46, src/fat_beagle.cpp:13-29,207-271
 *   Engine::LogLikelihoods (unrooted)  src/engine.cpp:54-60  -> fat_beagle.cpp:72-76
 *   Engine::LogLikelihoods (rooted)    src/engine.cpp:62-68  -> fat_beagle.cpp:96-104
 *   Engine::UnrootedLogLikelihoods     src/engine.cpp:70-76  -> fat_beagle.cpp:78-80
 *   Engine::Gradients (unrooted)       src/engine.cpp:78-84  -> fat_beagle.cpp:467-503
 *   Engine::Gradients (rooted)         src/engine.cpp:86-92  -> fat_beagle.cpp:505-545
 *   Engine::GetPhyloModelBlockSpecification  src/engine.cpp:48-52
 * together with everything those call: the 16 BEAGLE C-API calls
 * (SURVEY.md section 2.1), PhyloModel::SetParameters (src/phylo_model.cpp:26-31),
 * the GTR/JC69/Weibull models, Detrifurcate, SlideRootPosition, the rooted
 * chain rule (src/rooted_gradient_transforms.cpp) and the finite-difference
 * substitution gradient (src/fat_beagle.cpp:400-465).
 *
 * Conventions
 *   n = taxon_count, N = 2n-1, P = pattern_count, s = state_count, K = category_count.
 *   Trees arrive as the reference's own flat topology form, the parent-id vector
 *   (Node::ParentIdVector, src/node.hpp:153): leaves 0..n-1, internal nodes in
 *   post-order, root last and without an entry.  Children are ordered by max
 *   leaf id as in src/node.cpp:32-59.
 *     unrooted tree: 2n-2 nodes (trifurcating root), parent_ids[2n-3],
 *                    branch_lengths[2n-2] (entry of the root unused).
 *     rooted tree:   2n-1 nodes, parent_ids[2n-2], branch_lengths[2n-1].
 *   Parameter rows follow BlockSpecification (src/block_specification.cpp:11-50,
 *   src/phylo_model.cpp:13-15): [GTR rates(6) | frequencies(4)] [Weibull shape]
 *   [clock rate]; query with mi_engine_block().
 *   All arrays are row-major, FP64 / int32, caller-owned.  Results come back in
 *   tree order.  Every function returns 0 on success; otherwise nonzero and
 *   mi_last_error() holds the message (the C++ adapter rethrows it as
 *   std::runtime_error, mirroring Failwith, src/sugar.hpp:67-78).
 *   An engine is not re-entrant (neither is the reference's, engine.hpp:26-54).
 *
 * Two families of entry points:
 *   mi_engine_*          host pointers in / host pointers out (what the cgo-like
 *                        binding in the reference would call)
 *   mi_engine_*_device   device pointers in / out, asynchronous on `stream`
 *                        (a hipStream_t passed as void*; NULL = the engine's own
 *                        stream).  Used by bench.py with inputs resident in HBM,
 *                        and by multi-GPU callers that hand results to RCCL.
 */
#ifndef MI_PHYLO_H_
#define MI_PHYLO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_PHYLO_ABI_VERSION 2

typedef struct mi_engine mi_engine;

enum { MI_SUBST_JC69 = 0, MI_SUBST_GTR = 1,         /* src/substitution_model.cpp:6-15 */
       MI_SUBST_REVERSIBLE = 2 };                     /* 20 states: empirical model given as data */
enum { MI_SITE_CONSTANT = 0, MI_SITE_WEIBULL = 1 }; /* src/site_model.cpp:10-25 */
enum { MI_CLOCK_NONE = 0, MI_CLOCK_STRICT = 1 };    /* src/clock_model.cpp:6-15 */

typedef struct {
  int32_t taxon_count;    /* n >= 3 */
  int32_t pattern_count;  /* P >= 1 */
  int32_t state_count;    /* s: 4 (DNA, everything the reference has) or 20 (amino acids) */
  int32_t category_count; /* K: 1 for constant, K of "weibull+K" */
  int32_t subst_model;    /* MI_SUBST_* */
  int32_t site_model;     /* MI_SITE_* */
  int32_t clock_model;    /* MI_CLOCK_* */
  int32_t use_tip_states; /* EngineSpecification::use_tip_states_, engine.hpp:20-24 */
  int32_t device;         /* HIP device ordinal; -1 = current device */
  int32_t reserved;
} mi_engine_spec;

int32_t mi_abi_version(void);
const char* mi_last_error(void);
/* visible HIP devices (0 without a GPU); what thread_count is capped by in the adapter */
int32_t mi_device_count(void);

/* Engine::Engine + FatBeagle::SetTipStates/SetTipPartials/SetPatternWeights.
 * tip_states[n*P]: 0..s-1, >= s means gap (src/site_pattern.cpp:16-46).
 * tip_partials[n*P*s] is read when use_tip_states == 0 (may be NULL: then the
 * partials are derived from tip_states exactly as SitePattern::GetPartials does,
 * src/site_pattern.cpp:117-131). */
int32_t mi_engine_create(const mi_engine_spec* spec, const int32_t* tip_states,
                         const double* tip_partials, const double* pattern_weights,
                         mi_engine** out_engine);
/* 20-state engines (state_count == 20, subst_model == MI_SUBST_REVERSIBLE; not in the
 * reference, whose factory src/substitution_model.cpp:6-15 knows JC69 and GTR only -- this is
 * the entry a third branch `"WAG"` of that factory would call).  The model is DATA with no
 * free parameters: 190 exchangeabilities (upper triangle, row by row, like the reference's
 * GTR rates, substitution_model.cpp:39-55) and 20 frequencies, amino-acid order
 * ARNDCQEGHILKMFPSTWYV; Q is built and normalised by the reference's GTR recipe and
 * eigendecomposed on the device once.  NULL tables = the built-in WAG table
 * (mi_engine_create does the same for such a spec).  tip_states: 0..19, >= 20 = gap /
 * ambiguous.  Parameter rows then hold only the site and clock blocks, as for JC69. */
int32_t mi_engine_create_reversible(const mi_engine_spec* spec, const double* exchangeabilities,
                                    const double* frequencies, const int32_t* tip_states,
                                    const double* tip_partials, const double* pattern_weights,
                                    mi_engine** out_engine);
/* One handle driving several devices -- what Engine's thread_count FatBeagles are
 * (src/engine.cpp:14-27; FatBeagleParallelize, src/fat_beagle.hpp:119-149): `shard_count`
 * engines, shard i on HIP device devices[i] (an ordinal may repeat, which puts several logical
 * shards on one device; a negative ordinal -1 - k means "the k-th device counting from the
 * caller's current HIP device, wrapping around"; NULL = {-1, -2, ...}: round-robin starting
 * at the current device, so that a one-process-per-GPU launch which selected its device with
 * hipSetDevice stays on it), tips and weights resident on each.  shard_mode MI_SHARD_TREES: the host-pointer calls deal the trees to
 * the shards in contiguous blocks (mi_shard_range), all devices work side by side, results
 * come back in tree order -- bit-identical to a single engine's.  MI_SHARD_PATTERNS (few
 * trees, very long alignments): shard i holds the site patterns mi_shard_range(P, count, i),
 * evaluates every tree on them, and the per-tree results of the unrooted calls (sums over
 * patterns, all of them) are added in shard order.  spec->device is ignored.
 * exchangeabilities / frequencies: as mi_engine_create_reversible (NULL for 4-state models).
 * The *_device entry points need a single-device engine (one engine per device, e.g. one
 * process per GPU under torch.distributed: INTEGRATION.md). */
enum { MI_SHARD_TREES = 0, MI_SHARD_PATTERNS = 1 };
int32_t mi_engine_create_sharded(const mi_engine_spec* spec, int32_t shard_count,
                                 const int32_t* devices, int32_t shard_mode,
                                 const double* exchangeabilities, const double* frequencies,
                                 const int32_t* tip_states, const double* tip_partials,
                                 const double* pattern_weights, mi_engine** out_engine);
int32_t mi_engine_shard_count(const mi_engine* engine);
/* The HIP device ordinal shard `shard` was placed on (a single-device engine: shard 0, its
 * device); -1 for an invalid argument.  What the -1 - k / NULL forms of `devices` resolved to:
 * the reference's counterpart is the thread a FatBeagle is pinned to, engine.cpp:23-27. */
int32_t mi_engine_shard_device(const mi_engine* engine, int32_t shard);
/* Contiguous block of shard `shard` of `shard_count` over `total` units: block sizes differ
 * by at most one, the larger blocks first. */
int32_t mi_shard_range(int32_t total, int32_t shard_count, int32_t shard, int32_t* begin,
                       int32_t* count);
/* the built-in WAG table in that form: exchangeabilities[190], frequencies[20] (sum 1) */
int32_t mi_wag_model(double* exchangeabilities, double* frequencies);
void mi_engine_destroy(mi_engine* engine);

/* Engine::GetPhyloModelBlockSpecification: blocks in std::map (ASCII) order. */
int32_t mi_engine_param_count(const mi_engine* engine);
int32_t mi_engine_block_count(const mi_engine* engine);
int32_t mi_engine_block(const mi_engine* engine, int32_t index, const char** name,
                        int32_t* start, int32_t* length);

/* ---- host-pointer entry points ------------------------------------------ */

/* Engine::LogLikelihoods(const UnrootedTreeCollection&, params, rescaling) */
int32_t mi_engine_log_likelihoods_unrooted(mi_engine* engine, int32_t tree_count,
                                           const int32_t* parent_ids,     /* [T][2n-3] */
                                           const double* branch_lengths,  /* [T][2n-2] */
                                           const double* params,          /* [T][param_count] */
                                           int32_t rescaling,
                                           double* out_log_likelihoods /* [T] */);

/* Engine::Gradients(const UnrootedTreeCollection&, ...): PhyloGradient per tree.
 * out_branch_gradient[T][2n-1] = gradient_["branch_lengths"] (last two entries 0);
 * out_site_gradient[T] = gradient_["site_model"] (K > 1, else untouched; may be NULL);
 * out_subst_gradient[T][8] = gradient_["substitution_model"] (GTR; may be NULL).
 * A NULL output also skips the work only it needs: without out_subst_gradient the 16
 * finite-difference log-likelihood passes of a GTR call (fat_beagle.cpp:400-465), without
 * out_site_gradient the separate site-model pass (fat_beagle.cpp:488-496); the outputs that
 * are delivered are bit-identical to those of the full call. */
int32_t mi_engine_gradients_unrooted(mi_engine* engine, int32_t tree_count,
                                     const int32_t* parent_ids, const double* branch_lengths,
                                     const double* params, int32_t rescaling,
                                     double* out_log_likelihoods, double* out_branch_gradient,
                                     double* out_site_gradient, double* out_subst_gradient);

/* Engine::Gradients(const UnrootedTreeCollection&, ...) followed, on the device, by the
 * reductions a variational-inference step applies to its result (vip/burrito.py:143-166,
 * vip/branch_model.py:104-133, src/unrooted_sbn_instance.cpp:176-198):
 *   out_sums[0] = sum_t w_t logL_t          out_sums[1] = sum_t w_t gradient_["site_model"]_t
 *   out_index_gradient[k] = sum over (t, v) with branch_index[t][v] == k of
 *                           w_t gradient_["branch_lengths"]_t[v]        (k < index_count)
 * branch_index[T][2n-1]: the parameter (split) index of every node's branch, negative = not a
 * parameter (the root and the fixed node always carry a zero gradient); tree_weights[T] or
 * NULL (w_t = 1).  Sums run in (t, v) order: deterministic.  out_log_likelihoods[T] may be
 * NULL.  A multi-GPU step then needs ONE all-reduce of 2 + index_count doubles.  The
 * substitution-model block (16 finite-difference passes for GTR) is not computed. */
int32_t mi_engine_gradients_unrooted_reduced(mi_engine* engine, int32_t tree_count,
                                             const int32_t* parent_ids,
                                             const double* branch_lengths, const double* params,
                                             int32_t rescaling, const int32_t* branch_index,
                                             const double* tree_weights, int32_t index_count,
                                             double* out_sums /* [2] */,
                                             double* out_index_gradient /* [index_count] */,
                                             double* out_log_likelihoods /* [T] or NULL */);

/* Engine::LogLikelihoods(const RootedTreeCollection&) when with_jacobian != 0
 * (branch lengths x rates, + log-det-Jacobian, fat_beagle.cpp:82-104), or
 * Engine::UnrootedLogLikelihoods(const RootedTreeCollection&) when 0
 * (fat_beagle.cpp:78-80: raw branch lengths, no Jacobian; rates/heights/bounds
 * may then be NULL). */
int32_t mi_engine_log_likelihoods_rooted(mi_engine* engine, int32_t tree_count,
                                         const int32_t* parent_ids,    /* [T][2n-2] */
                                         const double* branch_lengths, /* [T][2n-1] */
                                         const double* params,
                                         const double* rates,        /* [T][2n-2] RootedTree::rates_ */
                                         const double* node_heights, /* [T][2n-1] */
                                         const double* node_bounds,  /* [T][2n-1] */
                                         int32_t with_jacobian, int32_t rescaling,
                                         double* out_log_likelihoods);

/* Engine::Gradients(const RootedTreeCollection&, ...).
 * out_log_likelihoods excludes the Jacobian (rooted_sbn_instance.hpp:285).
 * out_ratios_root_height[T][n-1], out_clock_gradient[T][2n-2] (strict clock,
 * rate_counts[t]==1: entry 0 holds the single value, rest 0; per-branch clock,
 * rate_counts[t]==2n-2: all entries; anything else is an error,
 * fat_beagle.cpp:375-386). */
int32_t mi_engine_gradients_rooted(mi_engine* engine, int32_t tree_count,
                                   const int32_t* parent_ids, const double* branch_lengths,
                                   const double* params, const double* rates,
                                   const int32_t* rate_counts, /* [T] */
                                   const double* node_heights, const double* node_bounds,
                                   const double* height_ratios, /* [T][n-1] */
                                   int32_t rescaling, double* out_log_likelihoods,
                                   double* out_ratios_root_height, double* out_clock_gradient,
                                   double* out_site_gradient, double* out_subst_gradient);

/* ---- device-pointer entry points (asynchronous) -------------------------- */
/* Same arguments and meaning, but every array pointer is a device pointer on
 * the engine's device and nothing is synchronised: the work is enqueued on
 * `stream`.  Per-tree input errors (malformed parent ids, GTR sums off by
 * >= 1e-3, bad rate_count) set a device-side status word that
 * mi_engine_check_status() reads back (it synchronises the stream).
 *
 * A call never fails spuriously (the reference's Engine does not: src/engine.cpp:54-92).  A
 * gradient call of up to 512 small trees runs tree set-up and walk as ONE launch whose walk
 * waves wait -- bounded: one second of wall clock -- for their tree's set-up waves.  Should
 * that wait ever run out, the HOST-pointer entry points above run the call again at once
 * through the four-launch sequence and return those results (mi_engine_last_call_path then
 * says "setup=own-launch", and the engine keeps four launches from then on); a caller of the
 * *_device entry points finds the message at mi_engine_check_status and repeats the call --
 * a hipGraph captured before that still holds the one-launch kernel and must be captured
 * again.  (MI_PHYLO_FUSED_SETUP=0 selects four launches from the start.) */

int32_t mi_engine_log_likelihoods_unrooted_device(mi_engine* engine, void* stream,
                                                  int32_t tree_count, const int32_t* parent_ids,
                                                  const double* branch_lengths,
                                                  const double* params, int32_t rescaling,
                                                  double* out_log_likelihoods);
int32_t mi_engine_gradients_unrooted_device(mi_engine* engine, void* stream, int32_t tree_count,
                                            const int32_t* parent_ids,
                                            const double* branch_lengths, const double* params,
                                            int32_t rescaling, double* out_log_likelihoods,
                                            double* out_branch_gradient,
                                            double* out_site_gradient,
                                            double* out_subst_gradient);
int32_t mi_engine_log_likelihoods_rooted_device(mi_engine* engine, void* stream,
                                                int32_t tree_count, const int32_t* parent_ids,
                                                const double* branch_lengths,
                                                const double* params, const double* rates,
                                                const double* node_heights,
                                                const double* node_bounds,
                                                int32_t with_jacobian, int32_t rescaling,
                                                double* out_log_likelihoods);
int32_t mi_engine_gradients_rooted_device(mi_engine* engine, void* stream, int32_t tree_count,
                                          const int32_t* parent_ids,
                                          const double* branch_lengths, const double* params,
                                          const double* rates, const int32_t* rate_counts,
                                          const double* node_heights, const double* node_bounds,
                                          const double* height_ratios, int32_t rescaling,
                                          double* out_log_likelihoods,
                                          double* out_ratios_root_height,
                                          double* out_clock_gradient, double* out_site_gradient,
                                          double* out_subst_gradient);

int32_t mi_engine_gradients_unrooted_reduced_device(
    mi_engine* engine, void* stream, int32_t tree_count, const int32_t* parent_ids,
    const double* branch_lengths, const double* params, int32_t rescaling,
    const int32_t* branch_index, const double* tree_weights, int32_t index_count,
    double* out_sums, double* out_index_gradient, double* out_log_likelihoods);

/* Make sure the workspace for `tree_count` trees exists (so that a following
 * *_device call allocates nothing and can be captured in a hipGraph).  20-state engines: if
 * the device cannot give the partial-vector arena its budget (MI_PHYLO_PLV_BYTES), the budget
 * is reduced and every buffer that scales with it is released and allocated again
 * (mi_engine_last_call_launches counts these back-offs) -- which invalidates graphs captured
 * on this engine BEFORE that: capture after the reservation, and again after any call that
 * reports a new back-off.  The fused-reduction entry points additionally need a sort
 * workspace that depends on index_count; it is allocated by their first call. */
int32_t mi_engine_reserve(mi_engine* engine, int32_t tree_count, int32_t for_gradients);
/* The same, plus the per-tree buffers and the index sort's workspace of
 * mi_engine_gradients_unrooted_reduced[_device] for up to `index_count` indices: a reduced
 * *_device call of at most that size then allocates nothing either (it used to
 * size and allocate its sort workspace inside the call). */
int32_t mi_engine_reserve_reduced(mi_engine* engine, int32_t tree_count, int32_t index_count);
/* Synchronise `stream` and report the first per-tree error since the last check (the status
 * word is sticky and cleared when an error is reported: calls themselves never clear it). */
int32_t mi_engine_check_status(mi_engine* engine, void* stream);

/* Introspection for the bench / profiles: name and launch count of the dominant
 * kernel of the last call, and the algorithmic bytes it accounts for
 * (DESIGN.md "Measurement"). */
int32_t mi_engine_last_call_info(const mi_engine* engine, const char** dominant_kernel,
                                 int64_t* evaluations, int64_t* gradient_evaluations);
/* One line saying which path the last call took: the dominant kernel, where the partial
 * vectors were kept (store=lds | arena | hbm), whether the tree set-up rode in the walk's launch
 * (setup=in-walk) or had its own, and the call's flavour (fd=16, site-pass, light, analytic,
 * rescaled, rooted, K=...).  Diagnostics for the randomised sweeps (tools/stress_*.py print a
 * histogram of it); the string is the engine's and lives until the next call. */
const char* mi_engine_last_call_path(const mi_engine* engine);

/* How the last call was cut up: the number of launches of the walk kernel(s) over chunks of
 * evaluations (the partial-vector arena of large trees holds only as many evaluations as
 * MI_PHYLO_PLV_BYTES allows; 1 when everything went in one launch), and how many times since
 * the engine was made the arena budget had to be reduced because the device could not give
 * the memory.  Either pointer may be NULL. */
int32_t mi_engine_last_call_launches(const mi_engine* engine, int32_t* walk_launches,
                                     int32_t* arena_backoffs);

/* Kernel timing for bench.py: after mi_engine_profile_begin(engine, max_calls)
 * every call brackets its dominant kernel launch(es) with a pair of HIP events on
 * the call's stream (no synchronisation).  mi_engine_profile_collect()
 * synchronises on the recorded events, writes one duration in milliseconds per
 * profiled call and stops profiling. */
int32_t mi_engine_profile_begin(mi_engine* engine, int32_t max_calls);
int32_t mi_engine_profile_collect(mi_engine* engine, double* out_ms, int32_t capacity,
                                  int32_t* out_count);
/* The same with the call cut into four phases by five more events (a few microseconds of
 * device time per call: not for a timed region whose step time is reported):
 *   [0] set-up (tree schedules, model instances; 20 states: + the transition matrices of
 *       the first launch)          [1] post-order walk (+ root) of the FIRST walk launch (20
 *       states; 0 for the 4-state kernels, whose walks are one kernel)
 *   [2] pre-order walk of the first launch (20 states) / the main walk pass (4 states)
 *   [3] everything after it (further launches of a chunked call, finite-difference and
 *       site passes, tile reduction, finalize).
 * out_phase_ms: [capacity][4]; *out_first_launch_evaluations: evaluations the first walk
 * launch covered (what a per-launch roofline divides by). */
int32_t mi_engine_profile_begin_phases(mi_engine* engine, int32_t max_calls);
int32_t mi_engine_profile_collect_phases(mi_engine* engine, double* out_ms, double* out_phase_ms,
                                         int32_t capacity, int32_t* out_count,
                                         int32_t* out_first_launch_evaluations);

/* ---- site-pattern compression on the device (widening beyond the Engine boundary) ----
 *
 * Replaces SitePattern::Compress (src/site_pattern.cpp:77-115; called from the
 * SitePattern constructor :9-14): the distinct columns of an alignment, their
 * multiplicities, in the iteration order of the reference's
 * std::unordered_map<std::vector<int>, double, IntVectorHasher> -- bit-exact.
 *
 *   codes        [taxon_count][site_count] symbol codes 0..3, 4 = gap/ambiguous
 *                (SitePattern::SymbolVectorOf, site_pattern.cpp:16-46), rows in taxon-id
 *                order, host memory
 *   out_patterns [taxon_count][*out_pattern_count] (dense; capacity taxon_count *
 *                site_count int32), out_weights [*out_pattern_count] (capacity site_count)
 *   out_hash_kernel_ms  optional: HIP-event time of the column-hashing kernel
 * Hashing, sorting and grouping run on the GPU; the P insertions that fix the order are
 * replayed on the host.  A 64-bit grouping-hash collision (verified, never assumed absent)
 * is retried with another hash seed, so a returned result is always exact.  Fails without
 * a GPU (libmi_phylo_host.so holds the CPU implementation). */
int32_t mi_site_pattern_compress(int32_t device, int32_t taxon_count, int64_t site_count,
                                 const int8_t* codes, int32_t* out_pattern_count,
                                 int32_t* out_patterns, double* out_weights,
                                 double* out_hash_kernel_ms);
/* The device-resident form (round 4): the same compression, but the pattern matrix
 * [taxon_count][*out_pattern_count] (int32, dense) and the weights stay in device memory --
 * for a 512 x 50 000 alignment that is 82 MB that need not come down only to go up again.
 * The two arrays are allocated by the library on `device` and handed to the caller, who
 * passes them to mi_engine_create_device_tips and releases them with mi_device_free (the
 * engine keeps its own copies). */
int32_t mi_site_pattern_compress_device(int32_t device, int32_t taxon_count, int64_t site_count,
                                        const int8_t* codes, int32_t* out_pattern_count,
                                        int32_t** out_device_patterns,
                                        double** out_device_weights, double* out_hash_kernel_ms);
void mi_device_free(void* device_pointer);
/* mi_engine_create / mi_engine_create_reversible (by spec->state_count; 20 states: the
 * built-in WAG table) with the tips ALREADY ON THE DEVICE: device_tip_states[n*P] int32 (as
 * tip_states above: >= s means gap) and device_pattern_weights[P], on spec->device (the
 * current device for -1).  use_tip_states == 0 derives the 0/1 partial vectors on the device
 * as SitePattern::GetPartials does.  Results are bit-identical to an engine made from the
 * same arrays in host memory. */
int32_t mi_engine_create_device_tips(const mi_engine_spec* spec, const int32_t* device_tip_states,
                                     const double* device_pattern_weights,
                                     mi_engine** out_engine);

#ifdef __cplusplus
}
#endif
#endif /* MI_PHYLO_H_ */
