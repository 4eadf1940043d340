/* TEST INFRASTRUCTURE ONLY -- see phylo_oracle.h for scope and pinning status.
 *
 * Every function cites the reference file:line (under /root/reference/src unless
 * stated) whose behaviour it restates.  The BEAGLE kernels themselves are not in
 * the reference tree (README.md:16, SConstruct:187-196: libhmsbeagle, branch
 * hmc-clock, no commit pinned); their semantics are restated from BEAGLE's
 * published algorithm (Ayres et al. 2019, Syst. Biol. 68:1052; Ji et al. 2020,
 * MBE 37:3047 for the pre-order/edge-derivative calls) as used by the call sites
 * in fat_beagle.cpp.
 */
#define _GNU_SOURCE
#include "phylo_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Internal arithmetic type.  The default build (liboracle.so) is FP64 like the
 * reference; `make liboracle_ld.so` (-DORC_LONG_DOUBLE) runs the identical code in
 * x87 80-bit long double (64-bit mantissa) as the extended-precision check that
 * backs the 1e-10 parity claim.  The C API stays double in both builds. */
#ifdef ORC_LONG_DOUBLE
typedef long double real;
#define R_EXP expl
#define R_LOG logl
#define R_POW powl
#define R_SQRT sqrtl
#define R_FABS fabsl
#define R_EXPM1 expm1l
#else
typedef double real;
#define R_EXP exp
#define R_LOG log
#define R_POW pow
#define R_SQRT sqrt
#define R_FABS fabs
#define R_EXPM1 expm1
#endif

/* Transition-matrix formula: 0 = P = V diag(exp(l r t)) V^-1 exactly as BEAGLE
 * evaluates it (default, the reference's behaviour); 1 = I + V diag(expm1(l r t))
 * V^-1, algebraically identical but free of the cancellation that costs the
 * BEAGLE form ~1e-16/(r t) relative accuracy in the off-diagonal entries when
 * r t is small.  Mode 1 is what the HIP engine uses; tests use it to bound the
 * noise floor of mode 0 (see DESIGN.md "Accuracy"). */
static int g_transition_mode = 0;
void orc_set_transition_mode(int mode) { g_transition_mode = mode; }

/* 1: run the sweeps through the s-generic loops also when s == 4 (the default
 * lets the compiler unroll a constant-s copy of the SAME code for s == 4).  Used by
 * tests/test_oracle_kats.py to show that the code path the 20-state parity tests rely
 * on reproduces every known-answer value of the reference at s == 4. */
static int g_generic_states = 0;
void orc_set_generic_states(int on) { g_generic_states = on; }

/* ORC_SUBST_REVERSIBLE: a general time-reversible model given as DATA (s(s-1)/2
 * exchangeabilities in upper-triangle row order + s frequencies), with no free
 * parameters -- what an empirical amino-acid model (WAG, LG, ...) is.  Not in the
 * reference (substitution_model.cpp:6-15 knows JC69 and GTR only); it is built by the
 * reference's own GTR recipe (substitution_model.cpp:39-80) from the table. */
static double g_rev_rates[ORC_MAX_STATES * (ORC_MAX_STATES - 1) / 2];
static double g_rev_freqs[ORC_MAX_STATES];
static int g_rev_states = 0;
int orc_set_reversible_model(int s, const double* exchangeabilities, const double* freqs) {
  if (s < 2 || s > ORC_MAX_STATES) return 1;
  for (int i = 0; i < s * (s - 1) / 2; i++) g_rev_rates[i] = exchangeabilities[i];
  for (int i = 0; i < s; i++) g_rev_freqs[i] = freqs[i];
  g_rev_states = s;
  return 0;
}

static _Thread_local char g_err[512];
const char* orc_last_error(void) { return g_err; }
static int fail(const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return 1;
}

/* ======================================================================== *
 *  Integer half
 * ======================================================================== */

/* site_pattern.cpp:16-46 (GetSymbolTable). */
static int symbol_code(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case '-': case 'N': case 'X': case '?':
    case 'B': case 'D': case 'H': case 'K': case 'M': case 'R':
    case 'S': case 'U': case 'V': case 'W': case 'Y':
      return 4;
    default: return -1; /* site_pattern.cpp:48-56: "Symbol '%c' not known." */
  }
}

/* site_pattern.cpp:67-75 (IntVectorHasher).  `int hash`; 0x9e3779b9 is an
 * unsigned literal so the sum is unsigned, >> on the int is arithmetic, and the
 * int result is widened (sign-extended) to size_t by std::unordered_map. */
static uint64_t column_hash(const int32_t* codes, int n, int L, int site) {
  int32_t h = codes[site];
  for (int i = 1; i < n; i++) {
    uint32_t v = (uint32_t)codes[(size_t)i * L + site];
    uint32_t t = v + 0x9e3779b9u + ((uint32_t)h << 6) + (uint32_t)(h >> 2);
    h = (int32_t)((uint32_t)h ^ t);
  }
  return (uint64_t)(int64_t)h;
}

/* Bucket counts libstdc++'s _Prime_rehash_policy walks through when elements
 * are inserted one at a time into an empty std::unordered_map (max load 1.0,
 * growth 2).  MEASURED in this image (g++ 11.4, the compiler oracle/_ref is built
 * with) by inserting 6e7 keys and printing bucket_count() changes; a rehash to
 * the next entry happens when size()+1 exceeds the current entry. */
static const uint64_t k_bucket_seq[] = {
    1,       13,      29,      59,       127,      257,      541,     1109,
    2357,    5087,    10273,   20753,    42043,    85229,    172933,  351061,
    712697,  1447153, 2938679, 5967347,  12117689, 24607243, 49969847, 101473717};

/* site_pattern.cpp:77-115.  The pattern order is the iteration order of
 * std::unordered_map<SymbolVector,double,IntVectorHasher>; we replay
 * libstdc++'s _Hashtable: singly linked node list, a bucket stores the node
 * *before* its first node, new nodes go to the front of their bucket (or to the
 * very front of the list if the bucket is empty), rehash re-threads nodes in
 * list order with the same rule (hashtable.h: _M_insert_bucket_begin,
 * _M_rehash_aux(unique)). */
int orc_site_pattern_compress(int n, int L, const char* seqs, int32_t* patterns_out,
                              double* weights_out, int32_t* P_out) {
  int32_t* codes = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * L);
  for (int i = 0; i < n; i++)
    for (int p = 0; p < L; p++) {
      int c = symbol_code(seqs[(size_t)i * L + p]);
      if (c < 0) {
        snprintf(g_err, sizeof g_err, "Symbol '%c' not known.", seqs[(size_t)i * L + p]);
        free(codes);
        return 1;
      }
      codes[(size_t)i * L + p] = c;
    }
  /* node arrays */
  int* node_site = (int*)malloc(sizeof(int) * L);
  uint64_t* node_hash = (uint64_t*)malloc(sizeof(uint64_t) * L);
  double* node_w = (double*)malloc(sizeof(double) * L);
  int* next = (int*)malloc(sizeof(int) * L);
  int count = 0, head = -1; /* head = before_begin.next */
  int seq_i = 0;
  uint64_t nb = k_bucket_seq[0];
  /* bucket value: -1 empty, -2 = before_begin, else node index that precedes */
  int* bucket = (int*)malloc(sizeof(int) * nb);
  bucket[0] = -1;
#define NEXT_OF(b) ((b) == -2 ? head : next[(b)])
  for (int site = 0; site < L; site++) {
    uint64_t h = column_hash(codes, n, L, site);
    uint64_t bkt = h % nb;
    int found = -1;
    if (bucket[bkt] != -1) {
      int node = NEXT_OF(bucket[bkt]);
      while (node != -1 && node_hash[node] % nb == bkt) {
        int same = 1;
        for (int i = 0; i < n && same; i++)
          same = codes[(size_t)i * L + site] == codes[(size_t)i * L + node_site[node]];
        if (same) { found = node; break; }
        node = next[node];
      }
    }
    if (found >= 0) { node_w[found] += 1.; continue; }
    /* rehash check (hashtable_policy: _M_need_rehash) */
    if ((uint64_t)count + 1 > nb) {
      uint64_t nnb = k_bucket_seq[++seq_i];
      int* nbucket = (int*)malloc(sizeof(int) * nnb);
      for (uint64_t b = 0; b < nnb; b++) nbucket[b] = -1;
      int p = head;
      head = -1;
      uint64_t bbegin_bkt = 0;
      while (p != -1) {
        int pn = next[p];
        uint64_t b = node_hash[p] % nnb;
        if (nbucket[b] == -1) {
          next[p] = head;
          head = p;
          nbucket[b] = -2;
          if (next[p] != -1) nbucket[bbegin_bkt] = p;
          bbegin_bkt = b;
        } else {
          int before = nbucket[b];
          if (before == -2) { next[p] = head; head = p; }
          else { next[p] = next[before]; next[before] = p; }
        }
        p = pn;
      }
      free(bucket);
      bucket = nbucket;
      nb = nnb;
      bkt = h % nb;
    }
    int node = count++;
    node_site[node] = site;
    node_hash[node] = h;
    node_w[node] = 1.;
    if (bucket[bkt] != -1) {
      int before = bucket[bkt];
      if (before == -2) { next[node] = head; head = node; }
      else { next[node] = next[before]; next[before] = node; }
    } else {
      next[node] = head;
      head = node;
      if (next[node] != -1) bucket[node_hash[next[node]] % nb] = node;
      bucket[bkt] = -2;
    }
  }
#undef NEXT_OF
  int P = 0;
  for (int node = head; node != -1; node = next[node], P++) {
    for (int i = 0; i < n; i++)
      patterns_out[(size_t)i * L + P] = codes[(size_t)i * L + node_site[node]];
    weights_out[P] = node_w[node];
  }
  *P_out = P;
  free(codes); free(node_site); free(node_hash); free(node_w); free(next); free(bucket);
  return 0;
}

/* node.cpp:32-59 (children ordered by max leaf id), node.cpp:341-357 (ids:
 * leaves keep their ids, internal nodes are numbered in post-order, so every
 * child id is smaller than its parent's id). */
int orc_children_of_parent_ids(int node_count, int n_leaves, const int32_t* parent_ids,
                               int32_t* child_start, int32_t* child_list) {
  int* maxleaf = (int*)malloc(sizeof(int) * node_count);
  int* deg = (int*)calloc(node_count, sizeof(int));
  for (int v = 0; v < node_count; v++) maxleaf[v] = v < n_leaves ? v : -1;
  for (int v = 0; v < node_count - 1; v++) {
    int par = parent_ids[v];
    if (par <= v || par >= node_count || par < n_leaves) {
      free(maxleaf); free(deg);
      return fail("parent id vector is not in post-order id form");
    }
    if (maxleaf[v] > maxleaf[par]) maxleaf[par] = maxleaf[v];
    deg[par]++;
  }
  child_start[0] = 0;
  for (int v = 0; v < node_count; v++) child_start[v + 1] = child_start[v] + deg[v];
  int* fill = (int*)calloc(node_count, sizeof(int));
  for (int v = 0; v < node_count - 1; v++) {
    int par = parent_ids[v];
    /* insertion sort by maxleaf */
    int base = child_start[par], k = fill[par]++;
    while (k > 0 && maxleaf[child_list[base + k - 1]] > maxleaf[v]) {
      child_list[base + k] = child_list[base + k - 1];
      k--;
    }
    child_list[base + k] = v;
  }
  free(maxleaf); free(deg); free(fill);
  return 0;
}

/* unrooted_tree.cpp:27-37: (c0,c1,c2) at root id r -> node r=(c1,c2), bl[r]=0;
 * new root r+1=(c0,r), bl[r+1]=0. */
int orc_detrifurcate(int n, const int32_t* parent_ids, const double* bl, int32_t* child0,
                     int32_t* child1, double* bl_out) {
  int nc = 2 * n - 2; /* nodes in the trifurcating tree */
  int32_t* cs = (int32_t*)malloc(sizeof(int32_t) * (nc + 1));
  int32_t* cl = (int32_t*)malloc(sizeof(int32_t) * nc);
  if (orc_children_of_parent_ids(nc, n, parent_ids, cs, cl)) { free(cs); free(cl); return 1; }
  int root = nc - 1;
  for (int v = n; v < nc; v++) {
    int d = cs[v + 1] - cs[v];
    if ((v != root && d != 2) || (v == root && d != 3)) {
      free(cs); free(cl);
      return fail("UnrootedTree::Detrifurcate given a non-trifurcating tree.");
    }
  }
  for (int v = n; v < root; v++) {
    child0[v - n] = cl[cs[v]];
    child1[v - n] = cl[cs[v] + 1];
  }
  child0[root - n] = cl[cs[root] + 1];
  child1[root - n] = cl[cs[root] + 2];
  child0[root + 1 - n] = cl[cs[root]];
  child1[root + 1 - n] = root;
  memcpy(bl_out, bl, sizeof(double) * nc);
  bl_out[root] = 0.;
  bl_out[root + 1] = 0.;
  free(cs); free(cl);
  return 0;
}

static int rooted_children(int n, const int32_t* parent_ids, int32_t* child0,
                           int32_t* child1) {
  int nc = 2 * n - 1;
  int32_t* cs = (int32_t*)malloc(sizeof(int32_t) * (nc + 1));
  int32_t* cl = (int32_t*)malloc(sizeof(int32_t) * nc);
  if (orc_children_of_parent_ids(nc, n, parent_ids, cs, cl)) { free(cs); free(cl); return 1; }
  for (int v = n; v < nc; v++) {
    if (cs[v + 1] - cs[v] != 2) {
      free(cs); free(cl);
      return fail("expected a bifurcating rooted tree");
    }
    child0[v - n] = cl[cs[v]];
    child1[v - n] = cl[cs[v] + 1];
  }
  free(cs); free(cl);
  return 0;
}

/* node.cpp:209-211 with node.cpp:194-203: post-order over internal nodes ==
 * increasing internal id. */
void orc_postorder_triples(int n, const int32_t* child0, const int32_t* child1,
                           int32_t* triples) {
  for (int i = 0; i < n - 1; i++) {
    triples[3 * i] = n + i;
    triples[3 * i + 1] = child0[i];
    triples[3 * i + 2] = child1[i];
  }
}

/* node.cpp:226-261 (explicit-stack double visit). */
void orc_preorder_triples(int n, const int32_t* child0, const int32_t* child1,
                          int32_t* triples) {
  int cap = 4 * n, top = 0, out = 0;
  int* st_node = (int*)malloc(sizeof(int) * cap);
  int* st_vis = (int*)malloc(sizeof(int) * cap);
  st_node[top] = 2 * n - 2; st_vis[top++] = 0;
  while (top) {
    int node = st_node[--top], visited = st_vis[top];
    int c0 = child0[node - n], c1 = child1[node - n];
    if (visited) {
      triples[out++] = c1; triples[out++] = c0; triples[out++] = node;
      if (c1 >= n) { st_node[top] = c1; st_vis[top++] = 0; }
    } else {
      triples[out++] = c0; triples[out++] = c1; triples[out++] = node;
      st_node[top] = node; st_vis[top++] = 1;
      if (c0 >= n) { st_node[top] = c0; st_vis[top++] = 0; }
    }
  }
  free(st_node); free(st_vis);
}

/* ======================================================================== *
 *  Models
 * ======================================================================== */

/* block_specification.cpp:11-50 + phylo_model.cpp:13-15: substitution block,
 * then site block, then clock block; inside GTR std::map order puts
 * "GTR rates" before "frequencies". */
void orc_param_layout(const orc_spec_t* spec, int* gtr_rates_off, int* freqs_off,
                      int* shape_off, int* clock_off) {
  int off = 0, s = spec->state_count;
  *gtr_rates_off = *freqs_off = *shape_off = *clock_off = -1;
  if (spec->subst_model == ORC_SUBST_GTR) {
    *gtr_rates_off = off; off += s * (s - 1) / 2;
    *freqs_off = off; off += s;
  }
  if (spec->site_model == ORC_SITE_WEIBULL) { *shape_off = off; off += 1; }
  if (spec->clock_model == ORC_CLOCK_STRICT) { *clock_off = off; off += 1; }
}

int orc_param_count(const orc_spec_t* spec) {
  int a, b, c, d, s = spec->state_count, count = 0;
  orc_param_layout(spec, &a, &b, &c, &d);
  if (a >= 0) count += s * (s - 1) / 2 + s;
  if (c >= 0) count += 1;
  if (d >= 0) count += 1;
  return count;
}

/* internal model, in the working precision */
typedef struct {
  int s, K;
  real pi[ORC_MAX_STATES];
  real Q[ORC_MAX_STATES * ORC_MAX_STATES];
  real V[ORC_MAX_STATES * ORC_MAX_STATES];
  real Vinv[ORC_MAX_STATES * ORC_MAX_STATES];
  real lambda[ORC_MAX_STATES];
  real gtr_rates[ORC_MAX_STATES * (ORC_MAX_STATES - 1) / 2];
  int n_gtr_rates;
  real cat_rates[ORC_MAX_CATEGORIES];
  real cat_weights[ORC_MAX_CATEGORIES];
  real cat_rate_derivs[ORC_MAX_CATEGORIES];
} model_t;

/* stick_breaking_transform.cpp:20-32 */
static void sb_forward(int K, const real* y, real* x) {
  real stick = 1.0;
  for (int k = 0; k < K - 1; k++) {
    real z = 1.0 / (1 + R_EXP(-(y[k] - R_LOG((real)(K - k - 1)))));
    x[k] = stick * z;
    stick -= x[k];
  }
  x[K - 1] = stick;
}

/* stick_breaking_transform.cpp:34-43 */
static void sb_inverse(int K, const real* x, real* y) {
  real sum = 0;
  for (int k = 0; k < K - 1; k++) {
    real z = x[k] / (1.0 - sum);
    y[k] = R_LOG(z / (1.0 - z)) + R_LOG((real)(K - k - 1));
    sum += x[k];
  }
}

void orc_stick_breaking(int K, const double* y, double* x) {
  real yy[ORC_MAX_STATES * ORC_MAX_STATES], xx[ORC_MAX_STATES * ORC_MAX_STATES];
  for (int i = 0; i < K - 1; i++) yy[i] = y[i];
  sb_forward(K, yy, xx);
  for (int i = 0; i < K; i++) x[i] = (double)xx[i];
}

void orc_stick_breaking_inverse(int K, const double* x, double* y) {
  real yy[ORC_MAX_STATES * ORC_MAX_STATES], xx[ORC_MAX_STATES * ORC_MAX_STATES];
  for (int i = 0; i < K; i++) xx[i] = x[i];
  sb_inverse(K, xx, yy);
  for (int i = 0; i < K - 1; i++) y[i] = (double)yy[i];
}

/* site_model.cpp:37-62 */
static void weibull(int K, real shape, real* rates, real* weights, real* derivs) {
  real mean_rate = 0, mean_deriv = 0, du[ORC_MAX_CATEGORIES];
  for (int i = 0; i < K; i++) {
    real quantile = (2.0 * i + 1.0) / (2.0 * K);
    rates[i] = R_POW(-R_LOG(1.0 - quantile), 1.0 / shape);
    mean_rate += rates[i];
    du[i] = -rates[i] * R_LOG(-R_LOG(1.0 - quantile)) / (shape * shape);
    mean_deriv += du[i];
  }
  mean_rate /= K;
  mean_deriv /= K;
  for (int i = 0; i < K; i++) {
    derivs[i] = (du[i] * mean_rate - rates[i] * mean_deriv) / (mean_rate * mean_rate);
    rates[i] /= mean_rate;
    weights[i] = (real)1.0 / K;
  }
}

void orc_weibull_rates(int K, double shape, double* rates, double* weights, double* derivs) {
  real r[ORC_MAX_CATEGORIES], w[ORC_MAX_CATEGORIES], d[ORC_MAX_CATEGORIES];
  weibull(K, shape, r, w, d);
  for (int i = 0; i < K; i++) {
    rates[i] = (double)r[i];
    weights[i] = (double)w[i];
    derivs[i] = (double)d[i];
  }
}

/* Cyclic Jacobi for a symmetric s x s matrix (lower triangle is authoritative,
 * as in Eigen::SelfAdjointEigenSolver which substitution_model.cpp:69 calls).
 * Eigenvalues ascending; U columns are eigenvectors.  Eigen is an empty
 * submodule in the reference (lib/eigen), so this is an independent solver: the
 * decomposition is unique only up to sign/rotation inside eigenspaces, and
 * P(t) = V exp(Lt) V^-1 does not depend on that choice. */
static void jacobi_eigh(int s, const real* A_in, real* evals, real* U) {
  real A[ORC_MAX_STATES * ORC_MAX_STATES];
  for (int i = 0; i < s; i++)
    for (int j = 0; j < s; j++) A[i * s + j] = i >= j ? A_in[i * s + j] : A_in[j * s + i];
  for (int i = 0; i < s; i++)
    for (int j = 0; j < s; j++) U[i * s + j] = i == j;
  for (int sweep = 0; sweep < 100; sweep++) {
    real off = 0, diag = 0;
    for (int i = 0; i < s; i++)
      for (int j = 0; j < s; j++) {
        if (i != j) off += A[i * s + j] * A[i * s + j];
        else diag += A[i * s + j] * A[i * s + j];
      }
    if (off <= 1e-45 * diag || off == 0.) break;
    for (int p = 0; p < s - 1; p++)
      for (int q = p + 1; q < s; q++) {
        real apq = A[p * s + q];
        if (apq == 0.) continue;
        real theta = (A[q * s + q] - A[p * s + p]) / (2. * apq);
        real t = (theta >= 0 ? 1. : -1.) / (R_FABS(theta) + R_SQRT(theta * theta + 1.));
        real c = 1. / R_SQRT(t * t + 1.), sn = t * c;
        for (int k = 0; k < s; k++) {
          real akp = A[k * s + p], akq = A[k * s + q];
          A[k * s + p] = c * akp - sn * akq;
          A[k * s + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < s; k++) {
          real apk = A[p * s + k], aqk = A[q * s + k];
          A[p * s + k] = c * apk - sn * aqk;
          A[q * s + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < s; k++) {
          real ukp = U[k * s + p], ukq = U[k * s + q];
          U[k * s + p] = c * ukp - sn * ukq;
          U[k * s + q] = sn * ukp + c * ukq;
        }
      }
  }
  for (int i = 0; i < s; i++) evals[i] = A[i * s + i];
  /* ascending selection sort, permuting columns of U */
  for (int i = 0; i < s - 1; i++) {
    int m = i;
    for (int j = i + 1; j < s; j++)
      if (evals[j] < evals[m]) m = j;
    if (m != i) {
      real t = evals[i]; evals[i] = evals[m]; evals[m] = t;
      for (int k = 0; k < s; k++) {
        real u = U[k * s + i]; U[k * s + i] = U[k * s + m]; U[k * s + m] = u;
      }
    }
  }
}

/* substitution_model.cpp:39-80 (UpdateQMatrix + Update), generalised from 4 to
 * s states (rate order = upper triangle row by row: AC,AG,AT,CG,CT,GT). */
static void gtr_update(model_t* m) {
  int s = m->s;
  real* Q = m->Q;
  int ri = 0;
  for (int i = 0; i < s; i++)
    for (int j = i + 1; j < s; j++) {
      real rate = m->gtr_rates[ri++];
      Q[i * s + j] = rate * m->pi[j];
      Q[j * s + i] = rate * m->pi[i];
    }
  real total = 0;
  for (int i = 0; i < s; i++) {
    real row_sum = 0;
    for (int j = 0; j < s; j++)
      if (i != j) row_sum += Q[i * s + j];
    Q[i * s + i] = -row_sum;
    total += row_sum * m->pi[i];
  }
  for (int i = 0; i < s * s; i++) Q[i] /= total;
  real sq[ORC_MAX_STATES], S[ORC_MAX_STATES * ORC_MAX_STATES],
      U[ORC_MAX_STATES * ORC_MAX_STATES];
  for (int i = 0; i < s; i++) sq[i] = R_SQRT(m->pi[i]);
  for (int i = 0; i < s; i++)
    for (int j = 0; j < s; j++) S[i * s + j] = sq[i] * Q[i * s + j] * (1.0 / sq[j]);
  jacobi_eigh(s, S, m->lambda, U);
  for (int i = 0; i < s; i++)
    for (int j = 0; j < s; j++) {
      m->V[i * s + j] = (1.0 / sq[i]) * U[i * s + j];
      m->Vinv[i * s + j] = U[j * s + i] * sq[j];
    }
}

/* substitution_model.hpp:59-74: the hard-coded JC69 eigensystem. */
static void jc69_set(model_t* m) {
  static const double V[16] = {1.0, 2.0, 0.0, 0.5, 1.0, -2.0, 0.5, 0.0,
                               1.0, 2.0, 0.0, -0.5, 1.0, -2.0, -0.5, 0.0};
  static const double Vi[16] = {0.25, 0.25, 0.25, 0.25, 0.125, -0.125, 0.125, -0.125,
                                0.0,  1.0,  0.0,  -1.0, 1.0,   0.0,    -1.0,  0.0};
  /* the reference writes the literal -1.3333333333333333 (a double) */
  static const double ev[4] = {0.0, -1.3333333333333333, -1.3333333333333333,
                               -1.3333333333333333};
  for (int i = 0; i < 4; i++) {
    m->pi[i] = 0.25;
    m->lambda[i] = ev[i];
    for (int j = 0; j < 4; j++) m->Q[i * 4 + j] = i == j ? -1.0 : 1.0 / 3.0;
  }
  for (int i = 0; i < 16; i++) {
    m->V[i] = V[i];
    m->Vinv[i] = Vi[i];
  }
  m->n_gtr_rates = 0;
}

/* phylo_model.cpp:26-31 -> substitution_model.cpp:17-37, site_model.cpp:27-32.
 * The clock block is parsed but never used by the likelihood
 * (fat_beagle.cpp:296-300, "Issue #146"). */
static int model_set(const orc_spec_t* spec, const real* params, model_t* m) {
  int ro, fo, so, co, s = spec->state_count, K = spec->category_count;
  if (s > ORC_MAX_STATES || K > ORC_MAX_CATEGORIES) return fail("too many states/categories");
  orc_param_layout(spec, &ro, &fo, &so, &co);
  m->s = s;
  m->K = K;
  if (spec->subst_model == ORC_SUBST_JC69) {
    if (s != 4) return fail("JC69 is a 4-state model");
    jc69_set(m);
  } else if (spec->subst_model == ORC_SUBST_GTR) {
    int nr = s * (s - 1) / 2;
    real fsum = 0, rsum = 0;
    m->n_gtr_rates = nr;
    for (int i = 0; i < nr; i++) rsum += (m->gtr_rates[i] = params[ro + i]);
    for (int i = 0; i < s; i++) fsum += (m->pi[i] = params[fo + i]);
    if (R_FABS(fsum - 1.) >= 0.001) return fail("GTR frequencies do not sum to 1 +/- 0.001!");
    if (R_FABS(rsum - 1.) >= 0.001) return fail("GTR rates do not sum to 1 +/- 0.001!");
    gtr_update(m);
  } else if (spec->subst_model == ORC_SUBST_REVERSIBLE) {
    if (g_rev_states != s) return fail("orc_set_reversible_model was not called for this state count");
    m->n_gtr_rates = s * (s - 1) / 2;
    for (int i = 0; i < m->n_gtr_rates; i++) m->gtr_rates[i] = g_rev_rates[i];
    for (int i = 0; i < s; i++) m->pi[i] = g_rev_freqs[i];
    gtr_update(m);
  } else {
    return fail("Substitution model not known");
  }
  if (spec->site_model == ORC_SITE_CONSTANT) {
    if (K != 1) return fail("constant site model has one category");
    m->cat_rates[0] = 1.;
    m->cat_weights[0] = 1.;
    m->cat_rate_derivs[0] = 0.;
  } else if (spec->site_model == ORC_SITE_WEIBULL) {
    weibull(K, params[so], m->cat_rates, m->cat_weights, m->cat_rate_derivs);
  } else {
    return fail("Site model not known");
  }
  return 0;
}

static real* to_real(const double* x, size_t count) {
  real* r = (real*)malloc(sizeof(real) * (count ? count : 1));
  for (size_t i = 0; i < count; i++) r[i] = x[i];
  return r;
}

static void model_to_public(const model_t* m, orc_model_t* o) {
  int s = m->s;
  o->s = s;
  o->K = m->K;
  o->n_gtr_rates = m->n_gtr_rates;
  for (int i = 0; i < s; i++) { o->pi[i] = (double)m->pi[i]; o->lambda[i] = (double)m->lambda[i]; }
  for (int i = 0; i < s * s; i++) {
    o->Q[i] = (double)m->Q[i]; o->V[i] = (double)m->V[i]; o->Vinv[i] = (double)m->Vinv[i];
  }
  for (int i = 0; i < m->n_gtr_rates; i++) o->gtr_rates[i] = (double)m->gtr_rates[i];
  for (int k = 0; k < m->K; k++) {
    o->cat_rates[k] = (double)m->cat_rates[k];
    o->cat_weights[k] = (double)m->cat_weights[k];
    o->cat_rate_derivs[k] = (double)m->cat_rate_derivs[k];
  }
}

static void model_from_public(const orc_model_t* o, model_t* m) {
  int s = o->s;
  m->s = s;
  m->K = o->K;
  m->n_gtr_rates = o->n_gtr_rates;
  for (int i = 0; i < s; i++) { m->pi[i] = o->pi[i]; m->lambda[i] = o->lambda[i]; }
  for (int i = 0; i < s * s; i++) { m->Q[i] = o->Q[i]; m->V[i] = o->V[i]; m->Vinv[i] = o->Vinv[i]; }
  for (int i = 0; i < o->n_gtr_rates; i++) m->gtr_rates[i] = o->gtr_rates[i];
  for (int k = 0; k < o->K; k++) {
    m->cat_rates[k] = o->cat_rates[k];
    m->cat_weights[k] = o->cat_weights[k];
    m->cat_rate_derivs[k] = o->cat_rate_derivs[k];
  }
}

int orc_model_set(const orc_spec_t* spec, const double* params, orc_model_t* out) {
  model_t m;
  real* pr = to_real(params, (size_t)orc_param_count(spec));
  int rc = model_set(spec, pr, &m);
  free(pr);
  if (!rc) model_to_public(&m, out);
  return rc;
}

/* ======================================================================== *
 *  Core: the BEAGLE calls B4..B11 of SURVEY.md section 2.1
 * ======================================================================== */

/* beagleUpdateTransitionMatrices (fat_beagle.cpp:304-314): P = V diag(exp(l r t)) V^-1
 * per edge and category.  BEAGLE's CPU EigenDecompositionCube clamps negative
 * entries to 0; restated here. out: [edge][k][s][s]. */
static void transition_matrices(const model_t* m, int n_edges, const real* bl, real* out) {
  int s = m->s, K = m->K;
  for (int e = 0; e < n_edges; e++)
    for (int k = 0; k < K; k++) {
      real ex[ORC_MAX_STATES];
      for (int a = 0; a < s; a++) {
        real x = m->lambda[a] * m->cat_rates[k] * bl[e];
        ex[a] = g_transition_mode ? R_EXPM1(x) : R_EXP(x);
      }
      real* P = out + ((size_t)e * K + k) * s * s;
      for (int i = 0; i < s; i++)
        for (int j = 0; j < s; j++) {
          real sum = 0;
          for (int a = 0; a < s; a++) sum += m->V[i * s + a] * ex[a] * m->Vinv[a * s + j];
          if (g_transition_mode && i == j) sum += 1.0;
          P[i * s + j] = sum > 0 ? sum : 0;
        }
    }
}

typedef struct {
  int n, N, P, s, K;
  const int32_t* tips;
  real* post; /* [N][K][P][s] (tip rows filled with one-hot / all-ones) */
  real* pre;  /* [N][K][P][s] or NULL */
  real* mats; /* [N-1][K][s][s] */
  real* cum_log_scale; /* [P] */
  int p0, p1;          /* pattern block the sweeps work on (cache blocking; see below) */
} core_ws_t;

/* Site patterns are independent through the whole tree, so the sweeps run over one
 * block of patterns at a time (post-order, pre-order, derivatives, root), keeping a
 * block's vectors of all nodes in cache.  Sums over patterns are still accumulated
 * pattern by pattern in ascending order, so the results are bitwise those of the
 * unblocked sweeps. */
#define ORC_PATTERN_BLOCK 64

static core_ws_t ws_alloc(const orc_spec_t* spec, const int32_t* tips, int need_pre) {
  core_ws_t w;
  w.n = spec->taxon_count; w.N = 2 * w.n - 1; w.P = spec->pattern_count;
  w.s = spec->state_count; w.K = spec->category_count; w.tips = tips;
  size_t plv = (size_t)w.K * ORC_PATTERN_BLOCK * w.s; /* one pattern block per node */
  w.post = (real*)malloc(sizeof(real) * plv * w.N);
  w.pre = need_pre ? (real*)malloc(sizeof(real) * plv * w.N) : NULL;
  w.mats = (real*)malloc(sizeof(real) * (size_t)(w.N - 1) * w.K * w.s * w.s);
  w.cum_log_scale = (real*)calloc(w.P, sizeof(real));
  w.p0 = 0; w.p1 = 0;
  return w;
}
static void ws_free(core_ws_t* w) { free(w->post); free(w->pre); free(w->mats); free(w->cum_log_scale); }

/* Tip rows of the current pattern block: site_pattern.cpp:117-131 (GetPartials) ==
 * BEAGLE compact states >= s. */
static void fill_tips(core_ws_t* w) {
  const int PB = ORC_PATTERN_BLOCK, p0 = w->p0;
  for (int t = 0; t < w->n; t++)
    for (int k = 0; k < w->K; k++)
      for (int p = w->p0; p < w->p1; p++) {
        real* L = w->post + (((size_t)t * w->K + k) * PB + (p - p0)) * w->s;
        int st = w->tips[(size_t)t * w->P + p];
        for (int i = 0; i < w->s; i++) L[i] = (st >= w->s || st < 0 || st == i) ? 1.0 : 0.0;
      }
}

/* beagleUpdatePartials (fat_beagle.cpp:60-63,139-141) with
 * AddLowerPartialOperation (fat_beagle.cpp:327-342): per-pattern max rescaling
 * writes log(max) into the cumulative buffer when rescaling is on. */
static inline __attribute__((always_inline)) void post_order_impl(
    core_ws_t* w, const int32_t* child0, const int32_t* child1, int rescaling, const int s) {
  int K = w->K, PB = ORC_PATTERN_BLOCK, p0 = w->p0;
  size_t plv = (size_t)K * PB * s;
  for (int v = w->n; v < w->N; v++) {
    int c0 = child0[v - w->n], c1 = child1[v - w->n];
    real* dst = w->post + plv * v;
    for (int k = 0; k < K; k++) {
      const real* M0 = w->mats + ((size_t)c0 * K + k) * s * s;
      const real* M1 = w->mats + ((size_t)c1 * K + k) * s * s;
      for (int p = w->p0; p < w->p1; p++) {
        const real* L0 = w->post + plv * c0 + ((size_t)k * PB + (p - p0)) * s;
        const real* L1 = w->post + plv * c1 + ((size_t)k * PB + (p - p0)) * s;
        real* D = dst + ((size_t)k * PB + (p - p0)) * s;
        for (int i = 0; i < s; i++) {
          real a = 0, b = 0;
          for (int j = 0; j < s; j++) {
            a += M0[i * s + j] * L0[j];
            b += M1[i * s + j] * L1[j];
          }
          D[i] = a * b;
        }
      }
    }
    if (rescaling) {
      for (int p = w->p0; p < w->p1; p++) {
        real mx = 0;
        for (int k = 0; k < K; k++)
          for (int i = 0; i < s; i++) {
            real x = dst[((size_t)k * PB + (p - p0)) * s + i];
            if (x > mx) mx = x;
          }
        if (mx == 0) mx = 1.0;
        for (int k = 0; k < K; k++)
          for (int i = 0; i < s; i++) dst[((size_t)k * PB + (p - p0)) * s + i] /= mx;
        w->cum_log_scale[p] += R_LOG(mx);
      }
    }
  }
}
static void post_order(core_ws_t* w, const int32_t* child0, const int32_t* child1,
                       int rescaling) {
  if (w->s == 4 && !g_generic_states) post_order_impl(w, child0, child1, rescaling, 4); /* unrolled by the compiler */
  else post_order_impl(w, child0, child1, rescaling, w->s);
}

/* beagleCalculateRootLogLikelihoods (fat_beagle.cpp:65-68,170-173). */
static real root_log_likelihood(const core_ws_t* w, const model_t* m, const double* weights,
                                real total) {
  int s = w->s, K = w->K, PB = ORC_PATTERN_BLOCK, p0 = w->p0;
  const real* R = w->post + (size_t)K * PB * s * (w->N - 1);
  for (int p = w->p0; p < w->p1; p++) {
    real site = 0;
    for (int k = 0; k < K; k++) {
      real sk = 0;
      for (int i = 0; i < s; i++) sk += m->pi[i] * R[((size_t)k * PB + (p - p0)) * s + i];
      site += m->cat_weights[k] * sk;
    }
    total += weights[p] * (R_LOG(site) + w->cum_log_scale[p]);
  }
  return total;
}

/* One workspace per worker thread, reused across trees -- the analogue of one
 * BEAGLE instance per FatBeagle (fat_beagle.cpp:207-256). */
static real core_log_likelihood_ws(core_ws_t* w, const model_t* model,
                                   const double* pattern_weights, const int32_t* child0,
                                   const int32_t* child1, const real* bl, int rescaling) {
  memset(w->cum_log_scale, 0, sizeof(real) * w->P); /* beagleResetScaleFactors */
  transition_matrices(model, w->N - 1, bl, w->mats);
  real total = 0;
  for (int pb = 0; pb < w->P; pb += ORC_PATTERN_BLOCK) {
    w->p0 = pb;
    w->p1 = pb + ORC_PATTERN_BLOCK < w->P ? pb + ORC_PATTERN_BLOCK : w->P;
    fill_tips(w);
    post_order(w, child0, child1, rescaling);
    total = root_log_likelihood(w, model, pattern_weights, total);
  }
  return total;
}

static real core_log_likelihood(const orc_spec_t* spec, const model_t* model,
                                const int32_t* tip_states, const double* pattern_weights,
                                const int32_t* child0, const int32_t* child1, const real* bl,
                                int rescaling) {
  core_ws_t w = ws_alloc(spec, tip_states, 0);
  real ll = core_log_likelihood_ws(&w, model, pattern_weights, child0, child1, bl, rescaling);
  ws_free(&w);
  return ll;
}

/* beagleUpdatePrePartials (fat_beagle.cpp:144-151, ops from
 * AddUpperPartialOperation :344-362; root pre-partial = frequencies, :316-325):
 * q_x[j] = sum_i P_x[i][j] * ( q_parent[i] * sum_m P_sis[i][m] L_sis[m] ).
 * Per-node scalers (when rescaling) are written but never accumulated
 * (cumulative index BEAGLE_OP_NONE); they cancel in the derivative ratio. */
static inline __attribute__((always_inline)) void pre_order_impl(
    core_ws_t* w, const model_t* m, const int32_t* tr, int rescaling, const int s) {
  int K = w->K, PB = ORC_PATTERN_BLOCK, p0 = w->p0;
  size_t plv = (size_t)K * PB * s;
  real* root = w->pre + plv * (w->N - 1);
  for (int k = 0; k < K; k++)
    for (int p = w->p0; p < w->p1; p++)
      for (int i = 0; i < s; i++) root[((size_t)k * PB + (p - p0)) * s + i] = m->pi[i];
  for (int op = 0; op < 2 * w->n - 2; op++) {
    int node = tr[3 * op], sis = tr[3 * op + 1], par = tr[3 * op + 2];
    real* dst = w->pre + plv * node;
    for (int k = 0; k < K; k++) {
      const real* Mn = w->mats + ((size_t)node * K + k) * s * s;
      const real* Ms = w->mats + ((size_t)sis * K + k) * s * s;
      for (int p = w->p0; p < w->p1; p++) {
        const real* qp = w->pre + plv * par + ((size_t)k * PB + (p - p0)) * s;
        const real* Ls = w->post + plv * sis + ((size_t)k * PB + (p - p0)) * s;
        real u[ORC_MAX_STATES];
        for (int i = 0; i < s; i++) {
          real a = 0;
          for (int mm = 0; mm < s; mm++) a += Ms[i * s + mm] * Ls[mm];
          u[i] = qp[i] * a;
        }
        real* D = dst + ((size_t)k * PB + (p - p0)) * s;
        for (int j = 0; j < s; j++) {
          real acc = 0;
          for (int i = 0; i < s; i++) acc += Mn[i * s + j] * u[i];
          D[j] = acc;
        }
      }
    }
    if (rescaling) {
      for (int p = w->p0; p < w->p1; p++) {
        real mx = 0;
        for (int k = 0; k < K; k++)
          for (int i = 0; i < s; i++) {
            real x = dst[((size_t)k * PB + (p - p0)) * s + i];
            if (x > mx) mx = x;
          }
        if (mx == 0) mx = 1.0;
        for (int k = 0; k < K; k++)
          for (int i = 0; i < s; i++) dst[((size_t)k * PB + (p - p0)) * s + i] /= mx;
      }
    }
  }
}
static void pre_order(core_ws_t* w, const model_t* m, const int32_t* tr, int rescaling) {
  if (w->s == 4 && !g_generic_states) pre_order_impl(w, m, tr, rescaling, 4);
  else pre_order_impl(w, m, tr, rescaling, w->s);
}

/* beagleCalculateEdgeDerivatives (fat_beagle.cpp:153-166) with the differential
 * matrix of BuildDifferentialMatrices (:107-117): D_k = dscale[k] * Q. */
static inline __attribute__((always_inline)) void edge_derivatives_impl(
    const core_ws_t* w, const model_t* m, const double* weights, const real* dscale, real* grad,
    const int s) {
  int K = w->K, PB = ORC_PATTERN_BLOCK, p0 = w->p0;
  size_t plv = (size_t)K * PB * s;
  for (int node = 0; node < w->N - 1; node++) {
    real g = grad[node]; /* running sum over the pattern blocks, ascending patterns */
    for (int p = w->p0; p < w->p1; p++) {
      real num = 0, den = 0;
      for (int k = 0; k < K; k++) {
        const real* q = w->pre + plv * node + ((size_t)k * PB + (p - p0)) * s;
        const real* L = w->post + plv * node + ((size_t)k * PB + (p - p0)) * s;
        real nk = 0, dk = 0;
        for (int a = 0; a < s; a++) {
          real dl = 0;
          for (int j = 0; j < s; j++) dl += (dscale[k] * m->Q[a * s + j]) * L[j];
          nk += q[a] * dl;
          dk += q[a] * L[a];
        }
        num += m->cat_weights[k] * nk;
        den += m->cat_weights[k] * dk;
      }
      g += weights[p] * (num / den);
    }
    grad[node] = g;
  }
  grad[w->N - 1] = 0.;
}
static void edge_derivatives(const core_ws_t* w, const model_t* m, const double* weights,
                             const real* dscale, real* grad) {
  if (w->s == 4 && !g_generic_states) edge_derivatives_impl(w, m, weights, dscale, grad, 4);
  else edge_derivatives_impl(w, m, weights, dscale, grad, w->s);
}

static real core_branch_gradient_ws(core_ws_t* w, const model_t* model,
                                    const double* pattern_weights, const int32_t* child0,
                                    const int32_t* child1, const real* bl, const real* dscale,
                                    int rescaling, real* grad) {
  memset(w->cum_log_scale, 0, sizeof(real) * w->P);
  transition_matrices(model, w->N - 1, bl, w->mats);
  int32_t* tr = (int32_t*)malloc(sizeof(int32_t) * 3 * (2 * w->n - 2));
  orc_preorder_triples(w->n, child0, child1, tr);
  for (int i = 0; i < w->N; i++) grad[i] = 0;
  real total = 0;
  for (int pb = 0; pb < w->P; pb += ORC_PATTERN_BLOCK) {
    w->p0 = pb;
    w->p1 = pb + ORC_PATTERN_BLOCK < w->P ? pb + ORC_PATTERN_BLOCK : w->P;
    fill_tips(w);
    post_order(w, child0, child1, rescaling);
    pre_order(w, model, tr, rescaling);
    edge_derivatives(w, model, pattern_weights, dscale, grad);
    total = root_log_likelihood(w, model, pattern_weights, total);
  }
  free(tr);
  return total;
}

static real core_branch_gradient(const orc_spec_t* spec, const model_t* model,
                                 const int32_t* tip_states, const double* pattern_weights,
                                 const int32_t* child0, const int32_t* child1, const real* bl,
                                 const real* dscale, int rescaling, real* grad) {
  core_ws_t w = ws_alloc(spec, tip_states, 1);
  real ll = core_branch_gradient_ws(&w, model, pattern_weights, child0, child1, bl, dscale,
                                    rescaling, grad);
  ws_free(&w);
  return ll;
}

double orc_core_log_likelihood(const orc_spec_t* spec, const orc_model_t* model,
                               const int32_t* tip_states, const double* pattern_weights,
                               const int32_t* child0, const int32_t* child1,
                               const double* bl, int rescaling) {
  model_t m;
  model_from_public(model, &m);
  real* b = to_real(bl, (size_t)(2 * spec->taxon_count - 1));
  real ll = core_log_likelihood(spec, &m, tip_states, pattern_weights, child0, child1, b,
                                rescaling);
  free(b);
  return (double)ll;
}

double orc_core_branch_gradient(const orc_spec_t* spec, const orc_model_t* model,
                                const int32_t* tip_states, const double* pattern_weights,
                                const int32_t* child0, const int32_t* child1,
                                const double* bl, const double* dscale, int rescaling,
                                double* grad) {
  int N = 2 * spec->taxon_count - 1;
  model_t m;
  model_from_public(model, &m);
  real* b = to_real(bl, (size_t)N);
  real* ds = to_real(dscale, (size_t)spec->category_count);
  real* g = (real*)malloc(sizeof(real) * N);
  real ll = core_branch_gradient(spec, &m, tip_states, pattern_weights, child0, child1, b, ds,
                                 rescaling, g);
  for (int i = 0; i < N; i++) grad[i] = (double)g[i];
  free(b); free(ds); free(g);
  return (double)ll;
}

/* ======================================================================== *
 *  FatBeagle-level wrappers
 * ======================================================================== */

/* fat_beagle.cpp:400-465: central finite differences (delta 1e-6) in
 * stick-breaking coordinates, frequencies first then rates; output order rates
 * then frequencies.  `f` is the full log-likelihood of the *input* tree.
 * On return *model holds what the reference's model holds afterwards: rates at
 * y - delta in the LAST rate coordinate, original frequencies (the loop's final
 * SetParameters(param_vector) re-applies the minus-perturbed vector,
 * fat_beagle.cpp:433-436).  The site-model pass that follows in
 * FatBeagle::Gradient therefore sees that model; restated as is. */
typedef real (*loglik_fn)(const model_t* model, void* ctx);

static int subst_gradient_fd(const orc_spec_t* spec, const real* params, model_t* model,
                             loglik_fn f, void* ctx, double* out8) {
  int ro, fo, so, co, s = spec->state_count, nr = s * (s - 1) / 2;
  orc_param_layout(spec, &ro, &fo, &so, &co);
  int pc = orc_param_count(spec);
  real* base = (real*)malloc(sizeof(real) * pc);
  real* pv = (real*)malloc(sizeof(real) * pc);
  memcpy(base, params, sizeof(real) * pc);
  /* SubstitutionModelGradient rebuilds param_vector from the model's own
   * frequencies and rates (fat_beagle.cpp:445-452). */
  for (int i = 0; i < nr; i++) base[ro + i] = model->gtr_rates[i];
  for (int i = 0; i < s; i++) base[fo + i] = model->pi[i];
  const real delta = 1.e-6;
  real y[ORC_MAX_STATES * ORC_MAX_STATES];
  int rc = 0;
  for (int which = 0; which < 2 && !rc; which++) { /* 0: frequencies, 1: rates */
    int off = which == 0 ? fo : ro, len = which == 0 ? s : nr;
    double* gout = which == 0 ? out8 + (nr - 1) : out8;
    memcpy(pv, base, sizeof(real) * pc);
    sb_inverse(len, pv + off, y);
    for (int idx = 0; idx < len - 1 && !rc; idx++) {
      real orig = y[idx];
      y[idx] = orig + delta;
      sb_forward(len, y, pv + off);
      if ((rc = model_set(spec, pv, model))) break;
      real lp = f(model, ctx);
      y[idx] = orig - delta;
      sb_forward(len, y, pv + off);
      if ((rc = model_set(spec, pv, model))) break;
      real lm = f(model, ctx);
      gout[idx] = (double)((lp - lm) / (2. * delta));
      y[idx] = orig;
      /* reference: subst_model->SetParameters(param_vector) with the minus
       * vector still in param_vector -> model already in that state. */
    }
  }
  free(base); free(pv);
  return rc;
}

typedef struct {
  const orc_spec_t* spec;
  const int32_t* tips;
  const double* weights;
  const int32_t *child0, *child1;
  const real* bl;
  int rescaling;
  real add; /* log-det-Jacobian for rooted trees */
  core_ws_t* ws;
} ll_ctx_t;

static real ll_of_model(const model_t* model, void* vctx) {
  ll_ctx_t* c = (ll_ctx_t*)vctx;
  return core_log_likelihood_ws(c->ws, model, c->weights, c->child0, c->child1, c->bl,
                                c->rescaling) + c->add;
}

/* fat_beagle.cpp:389-398 */
static real discrete_site_model_gradient(int N, const real* bl, const real* g) {
  real r = 0;
  for (int i = 0; i < N - 1; i++) r += g[i] * bl[i];
  return r;
}

static int detrifurcate_real(int n, const int32_t* parent_ids, const double* bl, int32_t* c0,
                             int32_t* c1, real* b) {
  double* tmp = (double*)malloc(sizeof(double) * (2 * n - 1));
  int rc = orc_detrifurcate(n, parent_ids, bl, c0, c1, tmp);
  for (int i = 0; i < 2 * n - 1; i++) b[i] = tmp[i];
  free(tmp);
  return rc;
}

int orc_unrooted_log_likelihoods(const orc_spec_t* spec, const int32_t* tip_states,
                                 const double* pattern_weights, int T,
                                 const int32_t* parent_ids, const double* bl,
                                 const double* params, int rescaling, int nthreads,
                                 double* out_logl) {
  int n = spec->taxon_count, pc = orc_param_count(spec), rc_all = 0;
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
  {
  core_ws_t ws = ws_alloc(spec, tip_states, 0);
#pragma omp for schedule(dynamic)
  for (int t = 0; t < T; t++) {
    int32_t* c0 = (int32_t*)malloc(sizeof(int32_t) * 2 * (n - 1));
    int32_t* c1 = c0 + (n - 1);
    real* b = (real*)malloc(sizeof(real) * (2 * n - 1));
    real* pr = to_real(params + (size_t)t * pc, (size_t)pc);
    model_t model;
    int rc = detrifurcate_real(n, parent_ids + (size_t)t * (2 * n - 3),
                               bl + (size_t)t * (2 * n - 2), c0, c1, b);
    if (!rc) rc = model_set(spec, pr, &model);
    if (!rc)
      out_logl[t] = (double)core_log_likelihood_ws(&ws, &model, pattern_weights, c0, c1, b,
                                                   rescaling);
    if (rc) {
#pragma omp critical
      { rc_all = 1; }
    }
    free(c0); free(b); free(pr);
  }
  ws_free(&ws);
  }
  if (rc_all && !g_err[0]) fail("a tree failed");
  return rc_all;
}

/* FatBeagle::Gradient(const UnrootedTree&) fat_beagle.cpp:467-503 */
int orc_unrooted_gradients(const orc_spec_t* spec, const int32_t* tip_states,
                           const double* pattern_weights, int T, const int32_t* parent_ids,
                           const double* bl, const double* params, int rescaling,
                           int nthreads, double* out_logl, double* out_branch,
                           double* out_site, double* out_subst) {
  int n = spec->taxon_count, N = 2 * n - 1, pc = orc_param_count(spec), rc_all = 0;
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
  {
  core_ws_t ws = ws_alloc(spec, tip_states, 1);
#pragma omp for schedule(dynamic)
  for (int t = 0; t < T; t++) {
    int32_t* c0 = (int32_t*)malloc(sizeof(int32_t) * 2 * (n - 1));
    int32_t* c1 = c0 + (n - 1);
    real* b = (real*)malloc(sizeof(real) * 3 * N);
    real* g = b + N;
    real* g2 = g + N;
    real* pr = to_real(params + (size_t)t * pc, (size_t)pc);
    model_t model;
    const int32_t* pid = parent_ids + (size_t)t * (2 * n - 3);
    const double* tbl = bl + (size_t)t * (2 * n - 2);
    int rc = detrifurcate_real(n, pid, tbl, c0, c1, b);
    if (!rc) rc = model_set(spec, pr, &model);
    if (!rc) {
      int root = N - 1, root_child = c0[root - n], fixed = c1[root - n];
      /* Tree::SlideRootPosition tree.cpp:72-78 */
      b[root_child] = b[root_child] + b[fixed];
      b[fixed] = 0.0;
      out_logl[t] = (double)core_branch_gradient_ws(&ws, &model, pattern_weights, c0, c1, b,
                                                    model.cat_rates, rescaling, g);
      if (spec->subst_model == ORC_SUBST_GTR && out_subst) {
        /* f = StaticUnrootedLogLikelihood(in_tree): Detrifurcate without the slide */
        real* b0 = (real*)malloc(sizeof(real) * N);
        int32_t* d0 = (int32_t*)malloc(sizeof(int32_t) * 2 * (n - 1));
        detrifurcate_real(n, pid, tbl, d0, d0 + (n - 1), b0);
        ll_ctx_t ctx = {spec, tip_states, pattern_weights, d0, d0 + (n - 1), b0, rescaling, 0.,
                        &ws};
        rc = subst_gradient_fd(spec, pr, &model, ll_of_model, &ctx, out_subst + (size_t)t * 8);
        free(b0); free(d0);
      }
      if (!rc && spec->category_count > 1 && out_site) {
        core_branch_gradient_ws(&ws, &model, pattern_weights, c0, c1, b, model.cat_rate_derivs,
                                rescaling, g2);
        out_site[t] = (double)discrete_site_model_gradient(N, b, g2);
      }
      g[fixed] = 0.; /* fat_beagle.cpp:499 */
      for (int i = 0; i < N; i++) out_branch[(size_t)t * N + i] = (double)g[i];
    }
    if (rc) {
#pragma omp critical
      { rc_all = 1; }
    }
    free(c0); free(b); free(pr);
  }
  ws_free(&ws);
  }
  return rc_all;
}

/* ---- rooted -------------------------------------------------------------- */

/* rooted_tree.cpp:20-81: SetTipDates/SetNodeBoundsUsingDates +
 * InitializeTimeTreeUsingBranchLengths.  (Plain double: this is input
 * preparation done by the caller of the engine, not engine arithmetic.) */
int orc_time_tree_init(int n, const int32_t* parent_ids, const double* bl,
                       const double* tip_dates, double* h, double* bound, double* ratios) {
  int N = 2 * n - 1;
  int32_t* c0 = (int32_t*)malloc(sizeof(int32_t) * 2 * (n - 1));
  int32_t* c1 = c0 + (n - 1);
  if (rooted_children(n, parent_ids, c0, c1)) { free(c0); return 2; }
  for (int i = 0; i < n; i++) h[i] = bound[i] = tip_dates[i];
  int bad = 0;
  for (int v = n; v < N; v++) {
    int a = c0[v - n], b = c1[v - n];
    bound[v] = bound[a] > bound[b] ? bound[a] : bound[b];
    h[v] = h[a] + bl[a];
    if (fabs(h[b] + bl[b] - h[v]) > 1e-4) bad = 1;
  }
  ratios[N - 1 - n] = h[N - 1];
  for (int v = 0; v < N - 1; v++)
    if (v >= n) ratios[v - n] = (h[v] - bound[v]) / (h[parent_ids[v]] - bound[v]);
  free(c0);
  if (bad) return fail("Tree isn't time-calibrated");
  return 0;
}

/* fat_beagle.cpp:82-94 (iteration order = TripleIdPreorderBifurcating). */
static real log_det_jacobian(int n, const int32_t* child0, const int32_t* child1,
                             const double* h, const double* bound) {
  int32_t* tr = (int32_t*)malloc(sizeof(int32_t) * 3 * (2 * n - 2));
  orc_preorder_triples(n, child0, child1, tr);
  real s = 0;
  for (int op = 0; op < 2 * n - 2; op++) {
    int node = tr[3 * op], par = tr[3 * op + 2];
    if (node >= n) s += R_LOG((real)h[par] - (real)bound[node]);
  }
  free(tr);
  return s;
}

int orc_rooted_log_likelihoods(const orc_spec_t* spec, const int32_t* tip_states,
                               const double* pattern_weights, int T,
                               const int32_t* parent_ids, const double* bl,
                               const double* params, const double* rates,
                               const double* node_heights, const double* node_bounds,
                               int with_jacobian, int rescaling, int nthreads,
                               double* out_logl) {
  int n = spec->taxon_count, N = 2 * n - 1, pc = orc_param_count(spec), rc_all = 0;
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
  {
  core_ws_t ws = ws_alloc(spec, tip_states, 0);
#pragma omp for schedule(dynamic)
  for (int t = 0; t < T; t++) {
    int32_t* c0 = (int32_t*)malloc(sizeof(int32_t) * 2 * (n - 1));
    int32_t* c1 = c0 + (n - 1);
    real* b = to_real(bl + (size_t)t * N, (size_t)N);
    real* pr = to_real(params + (size_t)t * pc, (size_t)pc);
    model_t model;
    int rc = rooted_children(n, parent_ids + (size_t)t * (N - 1), c0, c1);
    if (!rc) rc = model_set(spec, pr, &model);
    if (!rc) {
      real add = 0;
      if (with_jacobian) { /* fat_beagle.cpp:96-104; else :78-80 */
        for (int i = 0; i < N - 1; i++) b[i] *= rates[(size_t)t * (N - 1) + i];
        add = log_det_jacobian(n, c0, c1, node_heights + (size_t)t * N, node_bounds + (size_t)t * N);
      }
      out_logl[t] = (double)(core_log_likelihood_ws(&ws, &model, pattern_weights, c0, c1, b,
                                                    rescaling) + add);
    }
    if (rc) {
#pragma omp critical
      { rc_all = 1; }
    }
    free(c0); free(b); free(pr);
  }
  ws_free(&ws);
  }
  return rc_all;
}

/* rooted_gradient_transforms.cpp:17-37.  BinaryIdPreorder visits internal
 * nodes only (node.cpp:194-207), so the "node_id >= leaf_count" test is always
 * true there; the order does not matter for the result. */
static void height_gradient(int n, const int32_t* c0, const int32_t* c1, const double* rates,
                            const real* bg, real* hg) {
  int N = 2 * n - 1, root = N - 1;
  for (int i = 0; i < n - 1; i++) hg[i] = 0;
  for (int v = root; v >= n; v--) {
    if (v != root) hg[v - n] = -bg[v] * rates[v];
    hg[v - n] += bg[c0[v - n]] * rates[c0[v - n]];
    hg[v - n] += bg[c1[v - n]] * rates[c1[v - n]];
  }
}

/* rooted_gradient_transforms.cpp:39-64 */
static real node_partial(int v, int n, const double* h, const double* ratios,
                         const double* bound) {
  return ((real)h[v] - bound[v]) / ratios[v - n];
}
static real epoch_addition(int v, int c, int n, const double* h, const double* ratios,
                           const double* bound, const real* acc) {
  if (c < n) return 0.0;
  if (bound[v] == bound[c]) return acc[c - n] * ratios[c - n] / ratios[v - n];
  return acc[c - n] * ratios[c - n] / ((real)h[v] - bound[c]) *
         node_partial(v, n, h, ratios, bound);
}

/* rooted_gradient_transforms.cpp:78-100 (post-order over internal non-root nodes) */
static void ratio_gradient_unweighted(int n, const int32_t* c0, const int32_t* c1,
                                      const double* h, const double* ratios,
                                      const double* bound, const real* gh, real* out) {
  int N = 2 * n - 1, root = N - 1;
  for (int i = 0; i < n - 1; i++) out[i] = 0;
  for (int v = n; v < N; v++) {
    if (v == root) continue;
    out[v - n] += node_partial(v, n, h, ratios, bound) * gh[v - n];
    out[v - n] += epoch_addition(v, c0[v - n], n, h, ratios, bound, out);
    out[v - n] += epoch_addition(v, c1[v - n], n, h, ratios, bound, out);
  }
}

/* rooted_gradient_transforms.cpp:102-130 */
static real root_height_gradient(int n, const int32_t* c0, const int32_t* c1,
                                 const double* ratios, const real* gh) {
  int N = 2 * n - 1, root = N - 1;
  real* mult = (real*)malloc(sizeof(real) * (n - 1));
  mult[root - n] = 1.0;
  for (int v = root; v >= n; v--) { /* any top-down order */
    int a = c0[v - n], b = c1[v - n];
    if (a >= n) mult[a - n] = ratios[a - n] * mult[v - n];
    if (b >= n) mult[b - n] = ratios[b - n] * mult[v - n];
  }
  real sum = 0;
  for (int i = 0; i < n - 1; i++) sum += gh[i] * mult[i];
  free(mult);
  return sum;
}

/* rooted_gradient_transforms.cpp:132-170 */
static void ratio_gradient_of_branch_gradient(int n, const int32_t* c0, const int32_t* c1,
                                              const double* rates, const double* h,
                                              const double* bound, const double* ratios,
                                              const real* bg, real* out) {
  int root_i = n - 2;
  real* hg = (real*)malloc(sizeof(real) * (n - 1) * 3);
  real* log_time = hg + (n - 1);
  real* jac = log_time + (n - 1);
  height_gradient(n, c0, c1, rates, bg, hg);
  ratio_gradient_unweighted(n, c0, c1, h, ratios, bound, hg, out);
  out[root_i] = root_height_gradient(n, c0, c1, ratios, hg);
  for (int i = 0; i < n - 1; i++) log_time[i] = 0;
  for (int i = 0; i < n - 2; i++) log_time[i] = 1.0 / ((real)h[n + i] - bound[n + i]);
  ratio_gradient_unweighted(n, c0, c1, h, ratios, bound, log_time, jac);
  jac[root_i] = root_height_gradient(n, c0, c1, ratios, log_time);
  for (int i = 0; i < n - 2; i++) out[i] += jac[i] - 1.0 / ratios[i];
  out[root_i] += jac[root_i];
  free(hg);
}

/* FatBeagle::Gradient(const RootedTree&) fat_beagle.cpp:505-545 */
int orc_rooted_gradients(const orc_spec_t* spec, const int32_t* tip_states,
                         const double* pattern_weights, int T, const int32_t* parent_ids,
                         const double* bl, const double* params, const double* rates,
                         const int32_t* rate_counts, const double* node_heights,
                         const double* node_bounds, const double* height_ratios,
                         int rescaling, int nthreads, double* out_logl,
                         double* out_ratios_root_height, double* out_clock, double* out_site,
                         double* out_subst) {
  int n = spec->taxon_count, N = 2 * n - 1, pc = orc_param_count(spec), rc_all = 0;
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
  {
  core_ws_t ws = ws_alloc(spec, tip_states, 1);
#pragma omp for schedule(dynamic)
  for (int t = 0; t < T; t++) {
    int32_t* c0 = (int32_t*)malloc(sizeof(int32_t) * 2 * (n - 1));
    int32_t* c1 = c0 + (n - 1);
    real* b = (real*)malloc(sizeof(real) * 4 * N);
    real* bg = b + N;
    real* g2 = bg + N;
    real* og = g2 + N;
    const double* tb = bl + (size_t)t * N;
    const double* r = rates + (size_t)t * (N - 1);
    const double* h = node_heights + (size_t)t * N;
    const double* bd = node_bounds + (size_t)t * N;
    real* pr = to_real(params + (size_t)t * pc, (size_t)pc);
    model_t model;
    int rc = rooted_children(n, parent_ids + (size_t)t * (N - 1), c0, c1);
    if (!rc) rc = model_set(spec, pr, &model);
    if (!rc && !(rate_counts[t] == 1 || rate_counts[t] == N - 1)) {
      fail("The number of rates should be equal to 1 (i.e. strict clock) or equal to the "
           "number of branches.");
      rc = 1;
    }
    if (!rc) {
      for (int i = 0; i < N; i++) b[i] = tb[i];
      for (int i = 0; i < N - 1; i++) b[i] *= r[i];
      out_logl[t] = (double)core_branch_gradient_ws(&ws, &model, pattern_weights, c0, c1, b,
                                                    model.cat_rates, rescaling, bg);
      if (spec->subst_model == ORC_SUBST_GTR && out_subst) {
        ll_ctx_t ctx = {spec, tip_states, pattern_weights, c0, c1, b, rescaling,
                        log_det_jacobian(n, c0, c1, h, bd), &ws};
        rc = subst_gradient_fd(spec, pr, &model, ll_of_model, &ctx, out_subst + (size_t)t * 8);
      }
      if (!rc && spec->category_count > 1 && out_site) {
        core_branch_gradient_ws(&ws, &model, pattern_weights, c0, c1, b, model.cat_rate_derivs,
                                rescaling, g2);
        out_site[t] = (double)discrete_site_model_gradient(N, b, g2);
      }
      ratio_gradient_of_branch_gradient(n, c0, c1, r, h, bd, height_ratios + (size_t)t * (n - 1),
                                        bg, og);
      for (int i = 0; i < n - 1; i++)
        out_ratios_root_height[(size_t)t * (n - 1) + i] = (double)og[i];
      /* ClockGradient fat_beagle.cpp:367-387: unscaled times */
      double* oc = out_clock + (size_t)t * (N - 1);
      if (rate_counts[t] == 1) {
        real acc = 0;
        for (int i = 0; i < N - 1; i++) acc += bg[i] * tb[i];
        for (int i = 0; i < N - 1; i++) oc[i] = 0;
        oc[0] = (double)acc;
      } else {
        for (int i = 0; i < N - 1; i++) oc[i] = (double)(bg[i] * tb[i]);
      }
    }
    if (rc) {
#pragma omp critical
      { rc_all = 1; }
    }
    free(c0); free(b); free(pr);
  }
  ws_free(&ws);
  }
  return rc_all;
}
