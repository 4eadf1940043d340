/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the libsbn Engine/FatBeagle hot path.
 *
 * A plain-C, FP64 restatement of what the reference computes on the path
 *   Engine::{LogLikelihoods,Gradients}  (src/engine.cpp:54-92)
 *     -> FatBeagle::{LogLikelihood,Gradient} (src/fat_beagle.cpp:50-175,467-545)
 *       -> 16 BEAGLE C-API calls (beagle-dev/beagle-lib, branch hmc-clock, NOT
 *          vendored in the reference and not installed in this image).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (libsbn_amd/) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - integer half (site patterns, node ids, Detrifurcate, traversal orders):
 *     pinned bit-exactly against oracle/_ref/ref_dump, which is built from the
 *     reference's own unmodified sources.
 *   - floating-point half: BEAGLE/Eigen are unbuildable here, so the arithmetic
 *     is restated from BEAGLE's published algorithm and pinned against every
 *     known-answer value the reference's tests hold for this path
 *     (src/unrooted_sbn_instance.hpp:206-335, src/rooted_sbn_instance.hpp:246-378,
 *     vip/test/test_burrito.py:48, src/site_model.hpp:84-108,
 *     src/substitution_model.hpp:97-131) at the tolerances those tests state
 *     (1e-6 .. 1e-3).  Nothing in the reference pins it tighter than that.
 */
#ifndef PHYLO_ORACLE_H_
#define PHYLO_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_STATES 20
#define ORC_MAX_CATEGORIES 64

enum { ORC_SUBST_JC69 = 0, ORC_SUBST_GTR = 1, ORC_SUBST_REVERSIBLE = 2 };
enum { ORC_SITE_CONSTANT = 0, ORC_SITE_WEIBULL = 1 };
enum { ORC_CLOCK_NONE = 0, ORC_CLOCK_STRICT = 1 };

typedef struct {
  int32_t taxon_count;    /* n */
  int32_t pattern_count;  /* P */
  int32_t state_count;    /* s (4 for everything the reference supports) */
  int32_t category_count; /* K */
  int32_t subst_model;
  int32_t site_model;
  int32_t clock_model;
  int32_t use_tip_states; /* 1: compact states, 0: tip partials (same results) */
} orc_spec_t;

typedef struct {
  int s, K;
  double pi[ORC_MAX_STATES];
  double Q[ORC_MAX_STATES * ORC_MAX_STATES];    /* row-major */
  double V[ORC_MAX_STATES * ORC_MAX_STATES];    /* eigenvectors, row-major */
  double Vinv[ORC_MAX_STATES * ORC_MAX_STATES]; /* inverse eigenvectors */
  double lambda[ORC_MAX_STATES];
  double gtr_rates[ORC_MAX_STATES * (ORC_MAX_STATES - 1) / 2];
  int n_gtr_rates; /* 0 for JC69 */
  double cat_rates[ORC_MAX_CATEGORIES];
  double cat_weights[ORC_MAX_CATEGORIES];
  double cat_rate_derivs[ORC_MAX_CATEGORIES]; /* d rate_k / d shape */
} orc_model_t;

const char* orc_last_error(void);
/* 0 (default): BEAGLE's P = V exp(L t) V^-1;  1: I + V expm1(L t) V^-1 (see .c) */
void orc_set_transition_mode(int mode);
/* 1: s == 4 also runs through the s-generic loops (see .c) */
void orc_set_generic_states(int on);
/* table of the ORC_SUBST_REVERSIBLE model: s(s-1)/2 exchangeabilities (upper triangle,
 * row by row) and s frequencies; no free parameters.  Process-wide. */
int orc_set_reversible_model(int s, const double* exchangeabilities, const double* freqs);

/* ---- integer half ------------------------------------------------------- */

/* SitePattern::Compress (site_pattern.cpp:77-115) incl. the libstdc++
 * unordered_map iteration order.  seqs: n rows of L characters, row i = taxon
 * id i.  patterns_out: n*L ints (row stride L, first *P_out valid per row);
 * weights_out: L doubles.  Returns 0, or 1 for an unknown symbol. */
int orc_site_pattern_compress(int n, int L, const char* seqs, int32_t* patterns_out,
                              double* weights_out, int32_t* P_out);

/* children (sorted by max leaf id, node.cpp:32-59) from a parent-id vector of a
 * tree whose ids follow Node::Polish (node.cpp:341-357).  node_count = number
 * of nodes including the root (root id = node_count-1, has no entry in
 * parent_ids).  child_start/child_list: CSR over nodes. Returns 0 or 1. */
int orc_children_of_parent_ids(int node_count, int n_leaves, const int32_t* parent_ids,
                               int32_t* child_start, int32_t* child_list);

/* UnrootedTree::Detrifurcate (unrooted_tree.cpp:27-37) on flat arrays.
 * in: parent_ids[2n-3], bl[2n-2]; out: child0/child1[n-1] (internal node
 * n+i), bl_out[2n-1]. */
int orc_detrifurcate(int n, const int32_t* parent_ids, const double* bl, int32_t* child0,
                     int32_t* child1, double* bl_out);

/* Node::BinaryIdPostorder / TripleIdPreorderBifurcating (node.cpp:190-261) as
 * flat triple lists, for checking against ref_dump. */
void orc_postorder_triples(int n, const int32_t* child0, const int32_t* child1,
                           int32_t* triples /* 3*(n-1) */);
void orc_preorder_triples(int n, const int32_t* child0, const int32_t* child1,
                          int32_t* triples /* 3*(2n-2) */);

/* ---- models -------------------------------------------------------------- */

int orc_param_count(const orc_spec_t* spec);
/* offsets inside a parameter row (block_specification.cpp:11-50, phylo_model.cpp:13-15);
 * -1 when the block does not exist. */
void orc_param_layout(const orc_spec_t* spec, int* gtr_rates_off, int* freqs_off,
                      int* shape_off, int* clock_off);
int orc_model_set(const orc_spec_t* spec, const double* params, orc_model_t* model);
void orc_stick_breaking(int K, const double* y, double* x);         /* y[K-1] -> x[K] */
void orc_stick_breaking_inverse(int K, const double* x, double* y); /* x[K] -> y[K-1] */
void orc_weibull_rates(int K, double shape, double* rates, double* weights, double* derivs);

/* ---- core (bifurcating tree, node ids as in the reference) ---------------- */

double orc_core_log_likelihood(const orc_spec_t* spec, const orc_model_t* model,
                               const int32_t* tip_states, const double* pattern_weights,
                               const int32_t* child0, const int32_t* child1,
                               const double* bl, int rescaling);
/* BranchGradientInternals (fat_beagle.cpp:119-175): dscale[K] multiplies Q per
 * category (category rates, or d rate/d shape). grad has 2n-1 entries. */
double orc_core_branch_gradient(const orc_spec_t* spec, const orc_model_t* model,
                                const int32_t* tip_states, const double* pattern_weights,
                                const int32_t* child0, const int32_t* child1,
                                const double* bl, const double* dscale, int rescaling,
                                double* grad);

/* ---- Engine-level (tree collections; nthreads mirrors thread_count) ------- */

int orc_unrooted_log_likelihoods(const orc_spec_t* spec, const int32_t* tip_states,
                                 const double* pattern_weights, int T,
                                 const int32_t* parent_ids /* T*(2n-3) */,
                                 const double* bl /* T*(2n-2) */,
                                 const double* params /* T*param_count */, int rescaling,
                                 int nthreads, double* out_logl);
int orc_unrooted_gradients(const orc_spec_t* spec, const int32_t* tip_states,
                           const double* pattern_weights, int T, const int32_t* parent_ids,
                           const double* bl, const double* params, int rescaling,
                           int nthreads, double* out_logl, double* out_branch /* T*(2n-1) */,
                           double* out_site /* T or NULL */,
                           double* out_subst /* T*8 or NULL */);

/* RootedTree time-tree state from tip dates + branch lengths
 * (rooted_tree.cpp:20-81).  Returns 1 if not clock-like (tolerance 1e-4). */
int orc_time_tree_init(int n, const int32_t* parent_ids /* 2n-2 */,
                       const double* bl /* 2n-1 */, const double* tip_dates /* n */,
                       double* node_heights /* 2n-1 */, double* node_bounds /* 2n-1 */,
                       double* height_ratios /* n-1 */);

int orc_rooted_log_likelihoods(const orc_spec_t* spec, const int32_t* tip_states,
                               const double* pattern_weights, int T,
                               const int32_t* parent_ids /* T*(2n-2) */,
                               const double* bl /* T*(2n-1) */, const double* params,
                               const double* rates /* T*(2n-2) */,
                               const double* node_heights /* T*(2n-1) */,
                               const double* node_bounds /* T*(2n-1) */, int with_jacobian,
                               int rescaling, int nthreads, double* out_logl);
int orc_rooted_gradients(const orc_spec_t* spec, const int32_t* tip_states,
                         const double* pattern_weights, int T, const int32_t* parent_ids,
                         const double* bl, const double* params, const double* rates,
                         const int32_t* rate_counts /* T */, const double* node_heights,
                         const double* node_bounds, const double* height_ratios /* T*(n-1) */,
                         int rescaling, int nthreads, double* out_logl,
                         double* out_ratios_root_height /* T*(n-1) */,
                         double* out_clock /* T*(2n-2); strict: entry 0 only */,
                         double* out_site /* T or NULL */, double* out_subst /* T*8 or NULL */);

#ifdef __cplusplus
}
#endif
#endif /* PHYLO_ORACLE_H_ */
