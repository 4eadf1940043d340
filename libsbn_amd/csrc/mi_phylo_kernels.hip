// HIP kernels (gfx950 / CDNA4, wave64) for the phylogenetic likelihood and
// branch-gradient hot path.  See DESIGN.md for the mapping and rooflines.
//
// Mapping used by every compute kernel here: one wavefront = one tile of 64 site
// patterns of one evaluation (tree x model); lane = site pattern.  Site patterns
// are independent through the whole tree (Felsenstein pruning couples nodes, not
// sites), so a wave walks the complete tree for its tile without any
// inter-wave communication.  Transition matrices are wave-uniform and are read
// with scalar loads into SGPRs; FP64 FMAs take them as scalar operands.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>

#include <cstdlib>
#include <string>

#include "mi_phylo_kernels.h"

namespace miphylo {

namespace {

struct D4 {
  double x0, x1, x2, x3;
};

// Read-only data written by an EARLIER kernel (transition matrices, schedules,
// models) is addressed through the constant address space: loads with a
// wave-uniform address then become scalar loads (s_load_*) into SGPRs no matter
// what else the kernel stores.
typedef const double __attribute__((address_space(4))) * cdouble_ptr;
typedef const int __attribute__((address_space(4))) * cint_ptr;
__device__ __forceinline__ cdouble_ptr as_const(const double* p) {
  return (cdouble_ptr)(uintptr_t)p;
}
__device__ __forceinline__ cint_ptr as_const(const int* p) { return (cint_ptr)(uintptr_t)p; }

__device__ __forceinline__ D4 mul4(D4 a, D4 b) {
  return {a.x0 * b.x0, a.x1 * b.x1, a.x2 * b.x2, a.x3 * b.x3};
}

// a_i = sum_j M[i][j] L_j   (M row-major, wave-uniform)
template <typename MP>
__device__ __forceinline__ D4 matvec(MP M, D4 L) {
  D4 a;
  a.x0 = M[0] * L.x0 + M[1] * L.x1 + M[2] * L.x2 + M[3] * L.x3;
  a.x1 = M[4] * L.x0 + M[5] * L.x1 + M[6] * L.x2 + M[7] * L.x3;
  a.x2 = M[8] * L.x0 + M[9] * L.x1 + M[10] * L.x2 + M[11] * L.x3;
  a.x3 = M[12] * L.x0 + M[13] * L.x1 + M[14] * L.x2 + M[15] * L.x3;
  return a;
}

// q_j = sum_i M[i][j] u_i
template <typename MP>
__device__ __forceinline__ D4 matTvec(MP M, D4 u) {
  D4 q;
  q.x0 = M[0] * u.x0 + M[4] * u.x1 + M[8] * u.x2 + M[12] * u.x3;
  q.x1 = M[1] * u.x0 + M[5] * u.x1 + M[9] * u.x2 + M[13] * u.x3;
  q.x2 = M[2] * u.x0 + M[6] * u.x1 + M[10] * u.x2 + M[14] * u.x3;
  q.x3 = M[3] * u.x0 + M[7] * u.x1 + M[11] * u.x2 + M[15] * u.x3;
  return q;
}

__device__ __forceinline__ double dot4(D4 a, D4 b) {
  return a.x0 * b.x0 + a.x1 * b.x1 + a.x2 * b.x2 + a.x3 * b.x3;
}

// Compact tip state -> partial vector: one-hot, or all ones for a gap
// (site_pattern.cpp:117-131; BEAGLE treats compact states >= s the same way).
__device__ __forceinline__ D4 tip_vector(int st) {
  return {(st == 0 || st > 3) ? 1.0 : 0.0, (st == 1 || st > 3) ? 1.0 : 0.0,
          (st == 2 || st > 3) ? 1.0 : 0.0, (st == 3 || st > 3) ? 1.0 : 0.0};
}

// P * tip_vector(st) without arithmetic: column st of P, or 1 (rows of P sum to 1).
// Written as a chain of selects so that it compiles to v_cndmask, never to branches.
__device__ __forceinline__ double select_state(int st, double m0, double m1, double m2,
                                               double m3, double other) {
  double r = other;
  r = st == 3 ? m3 : r;
  r = st == 2 ? m2 : r;
  r = st == 1 ? m1 : r;
  r = st == 0 ? m0 : r;
  return r;
}
template <typename MP>
__device__ __forceinline__ D4 tip_column(MP M, int st) {
  return {select_state(st, M[0], M[1], M[2], M[3], 1.0),
          select_state(st, M[4], M[5], M[6], M[7], 1.0),
          select_state(st, M[8], M[9], M[10], M[11], 1.0),
          select_state(st, M[12], M[13], M[14], M[15], 1.0)};
}

__device__ __forceinline__ D4 load4(const double* __restrict__ ptr) {
  const double2 lo = *reinterpret_cast<const double2*>(ptr);
  const double2 hi = *reinterpret_cast<const double2*>(ptr + 2);
  return {lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ void store4(double* ptr, D4 v) {
  *reinterpret_cast<double2*>(ptr) = double2{v.x0, v.x1};
  *reinterpret_cast<double2*>(ptr + 2) = double2{v.x2, v.x3};
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Power-of-two rescaling: exact, so rescaled and unscaled results agree to the
// last bit wherever the unscaled ones are representable.
__device__ __forceinline__ int max_exponent(double m) { return m > 0.0 ? ilogb(m) : 0; }
__device__ __forceinline__ D4 scale4(D4 a, int e) {
  return {ldexp(a.x0, e), ldexp(a.x1, e), ldexp(a.x2, e), ldexp(a.x3, e)};
}
__device__ __forceinline__ double max4(D4 a) {
  return fmax(fmax(a.x0, a.x1), fmax(a.x2, a.x3));
}

// Workgroups are dealt to the 8 XCDs (each with a private 4 MiB L2) round-robin by
// linear id: ids b and b+8 share an XCD.  The walk kernels launch (tiles x evaluations)
// workgroups; this bijection hands each XCD whole evaluations, so that an evaluation's
// transition matrices and schedule are fetched into ONE L2 instead of all eight.
// Placement is a speed matter only.
struct TileEval {
  int tile, eval;
};
__device__ __forceinline__ TileEval xcd_tile_eval() {
  const int tiles = gridDim.x, count = gridDim.y;
  const int id = blockIdx.x + tiles * blockIdx.y;
  const int full = count & ~7;  // evaluations in complete groups of 8
  TileEval te;
  if (id < full * tiles) {
    const int s = id >> 3;  // s-th workgroup of its XCD
    te.eval = (s / tiles) * 8 + (id & 7);
    te.tile = s % tiles;
  } else {
    te.eval = blockIdx.y;
    te.tile = blockIdx.x;
  }
  return te;
}

__device__ __forceinline__ void set_status(int32_t* status, int code, int tree) {
  if (atomicCAS(status, 0, code) == 0) status[1] = tree;
}

// ------------------------------------------------------------------------
// Tree setup: parent-id vector -> evaluation schedule (one thread per tree).
// Restates node.cpp:32-59 (children ordered by max leaf id),
// unrooted_tree.cpp:27-37 (Detrifurcate), tree.cpp:72-78 (SlideRootPosition, a
// no-op on a detrifurcated tree), fat_beagle.cpp:96-101,507-511 (x rates).
// The schedule lists internal nodes in a post-order chosen by Sethi-Ullman
// labels so that the on-chip kernel needs at most floor(log2 n)+1 live
// partial-likelihood vectors; any post-order gives bitwise the same vectors.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tree_setup_kernel(TreeSetupArgs a) {
  // One wave per tree.  The walk itself is sequential (lane 0); its working set
  // lives in LDS (13 N ints) unless the tree is too large, and the bulk copies
  // (branch lengths, schedule) are done by all 64 lanes.
  extern __shared__ int32_t ts_lds[];
  const int t = blockIdx.x;
  const int lane = threadIdx.x;
  const int n = a.n, N = 2 * n - 1;
  const int nodes_in = a.rooted ? N : N - 1;
  const int32_t* par_in = a.parent_ids + (size_t)t * (nodes_in - 1);
  int32_t* maxleaf = a.use_lds ? ts_lds : a.scratch + (size_t)t * 13 * N;
  int32_t* cnt = maxleaf + N;
  int32_t* kids = cnt + N;  // 3 per node
  int32_t* c0 = kids + 3 * N;
  int32_t* c1 = c0 + N;
  int32_t* label = c1 + N;
  int32_t* slot = label + N;
  int32_t* stack = slot + N;  // 2N
  int32_t* par = stack + 2 * N;  // the parent ids, fetched by the whole wave at once
  __shared__ int ok_flag;
  SchedEntry* sched = a.sched + (size_t)t * (n - 1);
  double* ble = a.bl_eff + (size_t)t * N;

  for (int v = lane; v < N; v += 64) {
    maxleaf[v] = v < n ? v : -1;
    cnt[v] = 0;
    label[v] = 0;
    slot[v] = 0;
    c0[v] = c1[v] = 0;
    if (v < nodes_in - 1) par[v] = par_in[v];
  }
  __syncthreads();
  if (lane == 0) {
    int status = kOk;
    for (int v = 0; v < nodes_in - 1; v++) {
      const int p = par[v];
      if (p <= v || p >= nodes_in || p < n) {
        status = kBadParentIds;
        break;
      }
      if (maxleaf[v] > maxleaf[p]) maxleaf[p] = maxleaf[v];
    }
    for (int v = 0; v < nodes_in - 1 && status == kOk; v++) {
      const int p = par[v];
      int k = cnt[p];
      if (k >= 3) {
        status = a.rooted ? kNotBifurcating : kNotTrifurcatingRoot;
        break;
      }
      while (k > 0 && maxleaf[kids[3 * p + k - 1]] > maxleaf[v]) {
        kids[3 * p + k] = kids[3 * p + k - 1];
        k--;
      }
      kids[3 * p + k] = v;
      cnt[p]++;
    }
    const int root_in = nodes_in - 1;
    for (int v = n; v < nodes_in && status == kOk; v++) {
      const int want = (!a.rooted && v == root_in) ? 3 : 2;
      if (cnt[v] != want) status = a.rooted ? kNotBifurcating : kNotTrifurcatingRoot;
    }
    if (status == kOk) {
      for (int v = n; v < nodes_in; v++) {
        c0[v] = kids[3 * v];
        c1[v] = kids[3 * v + 1];
      }
      if (!a.rooted) {
        // (k0,k1,k2) at root r  ->  r = (k1,k2), r+1 = (k0, r)
        const int r = root_in;
        c0[r] = kids[3 * r + 1];
        c1[r] = kids[3 * r + 2];
        c0[r + 1] = kids[3 * r];
        c1[r + 1] = r;
      }
      // Sethi-Ullman labels (tips cost nothing: they are read in compact form).
      for (int v = n; v < N; v++) {
        const int l0 = label[c0[v]], l1 = label[c1[v]];
        label[v] = l0 == l1 ? l0 + 1 : (l0 > l1 ? l0 : l1);
      }
      // Post-order DFS, heavier child first; slots from a free bitmask.
      uint32_t free_mask = 0xffffffffu;
      int top = 0, out = 0;
      stack[top++] = (N - 1) << 1;
      int used_max = 0;
      while (top) {
        const int item = stack[--top];
        const int v = item >> 1;
        if (item & 1) {
          const int a0 = c0[v], a1 = c1[v];
          if (a0 >= n) free_mask |= 1u << slot[a0];
          if (a1 >= n) free_mask |= 1u << slot[a1];
          const int sl = __ffs(free_mask) - 1;
          free_mask &= ~(1u << sl);
          slot[v] = sl;
          if (sl + 1 > used_max) used_max = sl + 1;
          sched[out++] = {v, a0, a1, sl | (slot[a0] << 8) | (slot[a1] << 16) |
                                        ((a0 < n ? 1 : 0) << 24) | ((a1 < n ? 1 : 0) << 25)};
        } else {
          stack[top++] = (v << 1) | 1;
          const int a0 = c0[v], a1 = c1[v];
          const bool first0 = label[a0] >= label[a1];
          const int lo = first0 ? a1 : a0, hi = first0 ? a0 : a1;
          if (lo >= n) stack[top++] = lo << 1;
          if (hi >= n) stack[top++] = hi << 1;  // popped first
        }
      }
      if (used_max > a.max_slots) status = kTooManySlots;
      // ---- schedule of the on-chip gradient kernel ----
      // A non-root internal node is UNSTORED (2) when all its internal children are
      // stored, else STORED (1): every stored node then has an unstored child, so at
      // most (n-2)/2 nodes need an LDS slot; an unstored node's vector is
      // recomputed from its (stored or tip) children where it is needed.
      if (a.macros) {
        int32_t* cls = cnt;
        int32_t* sslot = maxleaf;
        int stored = 0;
        for (int v = n; v < N - 1; v++) {
          const int a0 = c0[v], a1 = c1[v];
          const bool unstored = (a0 < n || cls[a0] == 1) && (a1 < n || cls[a1] == 1);
          cls[v] = unstored ? 2 : 1;
          sslot[v] = unstored ? 0 : stored++;
        }
        cls[N - 1] = 1;
        sslot[N - 1] = -1;
        MacroEntry* mac = a.macros + (size_t)t * max_macros(n);
        int m = 0;
        for (int v = n; v < N; v++) {
          if (cls[v] != 1) continue;
          MacroEntry me;
          int kind[2];
          me.node = v;
          me.pad = 0;
          me.qslot = sslot[v];
          for (int j = 0; j < 2; j++) {
            const int ch = j ? c1[v] : c0[v];
            me.child[j] = ch;
            kind[j] = ch < n ? 0 : cls[ch];
            me.cslot[j] = (ch >= n && cls[ch] == 1) ? sslot[ch] : 0;
            const bool expand = ch >= n && cls[ch] == 2;
            const int ga = expand ? c0[ch] : 0, gb = expand ? c1[ch] : 0;
            me.grand[2 * j] = ga;
            me.grand[2 * j + 1] = gb;
            me.gslot[2 * j] = ga >= n ? sslot[ga] : 0;
            me.gslot[2 * j + 1] = gb >= n ? sslot[gb] : 0;
          }
          me.shape = macro_shape(kind[0], kind[1], v == N - 1, me.child, me.grand, n);
          mac[m++] = me;
        }
        a.macro_count[t] = m;
        if (stored > max_stored(n)) status = kTooManySlots;
      }
    }
    if (status != kOk) set_status(a.status, status, t);
    ok_flag = status == kOk || status == kTooManySlots;
  }
  __syncthreads();
  if (!ok_flag) {
    if (lane == 0 && a.macro_count) a.macro_count[t] = 0;
    for (int i = lane; i < n - 1; i += 64) sched[i] = {n + i, 0, 1, 0};
    for (int v = lane; v < N; v += 64) ble[v] = 0.0;
    return;
  }
  if (!a.rooted) {
    const double* bl = a.bl + (size_t)t * (N - 1);
    for (int v = lane; v < N; v += 64) ble[v] = v < N - 2 ? bl[v] : 0.0;
  } else {
    const double* bl = a.bl + (size_t)t * N;
    const double* rates = a.rates ? a.rates + (size_t)t * (N - 1) : nullptr;
    for (int v = lane; v < N; v += 64)
      ble[v] = (rates && v < N - 1) ? bl[v] * rates[v] : bl[v];
  }
}

// ------------------------------------------------------------------------
// Tree setup for N <= 64 nodes: the same walk with every per-node array held in ONE
// vector register (lane = node id) and indexed with v_readlane / v_writelane.  The walk
// is sequential and its cost is the latency of each dependent array access: a
// cross-lane read is an order of magnitude quicker than an LDS round trip.  All
// values are wave-uniform, so control flow is scalar.
// ------------------------------------------------------------------------
#define RDL(arr, i) __builtin_amdgcn_readlane((arr), (i))
#define WRL(arr, i, val) (arr) = (lane == (i)) ? (val) : (arr)
// A per-node array of up to NB * 64 entries held in NB vector registers: entry i lives in
// lane i % 64 of register i / 64.  `rd` reads an entry with a wave-uniform index (two or
// four v_readlane and scalar selects, no branch), `own(nb)` is the node id this lane holds
// in register nb.
template <int NB>
struct NodeArray {
  int r[NB];
  __device__ __forceinline__ void fill(int v) {
#pragma unroll
    for (int nb = 0; nb < NB; nb++) r[nb] = v;
  }
  __device__ __forceinline__ int rd(int i) const {
    int out = __builtin_amdgcn_readlane(r[0], i & 63);
#pragma unroll
    for (int nb = 1; nb < NB; nb++) {
      const int x = __builtin_amdgcn_readlane(r[nb], i & 63);
      out = (i >> 6) == nb ? x : out;
    }
    return out;
  }
  // lane-varying index (gather): every lane reads entry idx
  __device__ __forceinline__ int gather(int idx) const {
    int out = __shfl(r[0], idx & 63, 64);
#pragma unroll
    for (int nb = 1; nb < NB; nb++) {
      const int x = __shfl(r[nb], idx & 63, 64);
      out = (idx >> 6) == nb ? x : out;
    }
    return out;
  }
};

template <int NB>
__global__ __launch_bounds__(64) void tree_setup_small_kernel(TreeSetupArgs a) {
  // Branch-free by construction: the tree walks are dependent chains, and on this machine
  // a taken scalar branch costs more than the handful of instructions it would skip, so
  // every loop body is straight-line code (lane selects / scalar selects) and whatever can
  // be done by all lanes at once (child lists, sorting, the macro entries) is.
  using Arr = NodeArray<NB>;
  const int t = blockIdx.x;
  const int lane = threadIdx.x;
  const int n = a.n, N = 2 * n - 1;
  const int nodes_in = a.rooted ? N : N - 1;
  const int root_in = nodes_in - 1;
  const int32_t* par_in = a.parent_ids + (size_t)t * (nodes_in - 1);
  SchedEntry* sched = a.sched + (size_t)t * (n - 1);
  double* ble = a.bl_eff + (size_t)t * N;
  auto own = [&](int nb) { return lane + 64 * nb; };

  Arr par, maxleaf;
  int status = kOk;
  bool bad_parent = false;
#pragma unroll
  for (int nb = 0; nb < NB; nb++) {
    const int v = own(nb);
    par.r[nb] = v < nodes_in - 1 ? par_in[v] : -1;
    maxleaf.r[nb] = v < n ? v : -1;
    bad_parent |= v < nodes_in - 1 && (par.r[nb] <= v || par.r[nb] >= nodes_in || par.r[nb] < n);
  }
  if (__any(bad_parent)) status = kBadParentIds;

  // max leaf id below every node, bottom-up (ids are a post-order: children first)
  if (status == kOk)
    for (int v = 0; v < nodes_in - 1; v++) {
      const int p = par.rd(v), mv = maxleaf.rd(v);
#pragma unroll
      for (int nb = 0; nb < NB; nb++)
        maxleaf.r[nb] = (own(nb) == p && mv > maxleaf.r[nb]) ? mv : maxleaf.r[nb];
    }
  // every lane collects the children (at most three) of the nodes it holds, ascending max
  // leaf id, by looking at each node once
  Arr cnt, k0, k1, k2, m0, m1;
  cnt.fill(0); k0.fill(0); k1.fill(0); k2.fill(0); m0.fill(0); m1.fill(0);
  if (status == kOk)
    for (int v = 0; v < nodes_in - 1; v++) {
      const int p = par.rd(v), mv = maxleaf.rd(v);
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        const bool mine = own(nb) == p;
        const int c = cnt.r[nb];
        // sorted insert of (v, mv) into (k0 | m0), (k1 | m1), k2; equal keys cannot occur
        // (disjoint leaf sets)
        const bool lt0 = c >= 1 && m0.r[nb] > mv, lt1 = c >= 2 && m1.r[nb] > mv;
        const int n0 = c == 0 || lt0 ? v : k0.r[nb];
        const int n1 = c == 0 ? k1.r[nb] : (lt0 ? k0.r[nb] : (c == 1 || lt1 ? v : k1.r[nb]));
        const int n2 = c < 2 ? k2.r[nb] : (lt1 ? k1.r[nb] : v);
        const int nm0 = c == 0 || lt0 ? mv : m0.r[nb];
        const int nm1 = c == 0 ? m1.r[nb] : (lt0 ? m0.r[nb] : (c == 1 || lt1 ? mv : m1.r[nb]));
        k0.r[nb] = mine ? n0 : k0.r[nb];
        k1.r[nb] = mine ? n1 : k1.r[nb];
        k2.r[nb] = (mine && c <= 2) ? n2 : k2.r[nb];
        m0.r[nb] = mine ? nm0 : m0.r[nb];
        m1.r[nb] = mine ? nm1 : m1.r[nb];
        cnt.r[nb] += mine ? 1 : 0;
      }
    }
  if (status == kOk) {
    bool wrong = false;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      const int v = own(nb);
      const int want = (!a.rooted && v == root_in) ? 3 : 2;
      wrong |= v >= n && v < nodes_in && cnt.r[nb] != want;
    }
    if (__any(wrong)) status = a.rooted ? kNotBifurcating : kNotTrifurcatingRoot;
  }
  Arr c0, c1;
  c0.fill(0);
  c1.fill(0);
  // schedule, entry i in lane i % 64 of register i / 64
  Arr s_node, s_c0, s_c1, s_sl;
  s_node.fill(0); s_c0.fill(0); s_c1.fill(0); s_sl.fill(0);
  int macro_total = 0, stored_total = 0;
  MacroEntry me[NB];
  bool is_macro[NB];
  int macro_rank[NB];
#pragma unroll
  for (int nb = 0; nb < NB; nb++) {
    me[nb] = MacroEntry{};
    is_macro[nb] = false;
    macro_rank[nb] = 0;
  }
  if (status == kOk) {
    const int kr0 = a.rooted ? 0 : k0.rd(root_in);
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      const int v = own(nb);
      if (v >= n && v < nodes_in) {
        c0.r[nb] = k0.r[nb];
        c1.r[nb] = k1.r[nb];
      }
      if (!a.rooted) {
        // (k0,k1,k2) at root r  ->  r = (k1,k2), r+1 = (k0, r)
        if (v == root_in) {
          c0.r[nb] = k1.r[nb];
          c1.r[nb] = k2.r[nb];
        }
        if (v == root_in + 1) {
          c0.r[nb] = kr0;
          c1.r[nb] = root_in;
        }
      }
    }
    if (!a.need_slots) {
      // only the matrix-core gradient kernel and finalize will read this tree: the
      // node-id order (already a post-order) with no slot assignment is enough
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        const int node = n + own(nb);
        s_node.r[nb] = node;
        s_c0.r[nb] = c0.gather(node < N ? node : 0);
        s_c1.r[nb] = c1.gather(node < N ? node : 0);
      }
    } else {
      // Sethi-Ullman labels and internal-subtree sizes, bottom-up (tips cost nothing)
      Arr label, size;
      label.fill(0);
      size.fill(0);
      for (int v = n; v < N; v++) {
        const int a0 = c0.rd(v), a1 = c1.rd(v);
        const int l0 = label.rd(a0), l1 = label.rd(a1);
        const int sz = 1 + size.rd(a0) + size.rd(a1);
        const int lb = l0 == l1 ? l0 + 1 : (l0 > l1 ? l0 : l1);
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
          label.r[nb] = own(nb) == v ? lb : label.r[nb];
          size.r[nb] = own(nb) == v ? sz : size.r[nb];
        }
      }
      // position in the post-order that visits the heavier child first: top-down, a
      // node's subtree occupies [start, start + size), the node itself comes last
      Arr first, second, size_first, start;
      start.fill(0);
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        const int lab0 = label.gather(c0.r[nb]), lab1 = label.gather(c1.r[nb]);
        const bool first0 = lab0 >= lab1;
        first.r[nb] = first0 ? c0.r[nb] : c1.r[nb];
        second.r[nb] = first0 ? c1.r[nb] : c0.r[nb];
      }
#pragma unroll
      for (int nb = 0; nb < NB; nb++) size_first.r[nb] = size.gather(first.r[nb]);
      // (the same loop fills node_at[position] = node)
      Arr node_at;
      node_at.fill(0);
      for (int v = N - 1; v >= n; v--) {
        const int st = start.rd(v), f = first.rd(v), sc = second.rd(v), sf = size_first.rd(v);
        const int pos = st + size.rd(v) - 1;
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
          start.r[nb] = own(nb) == f ? st : (own(nb) == sc ? st + sf : start.r[nb]);
          node_at.r[nb] = own(nb) == pos ? v : node_at.r[nb];
        }
      }
      // LDS slots in schedule order from a free bitmask
      Arr slot;
      slot.fill(0);
      uint32_t free_mask = 0xffffffffu;
      int used_max = 0;
      for (int out = 0; out < n - 1; out++) {
        const int v = node_at.rd(out);
        const int a0 = c0.rd(v), a1 = c1.rd(v);
        const int sa0 = slot.rd(a0), sa1 = slot.rd(a1);
        free_mask |= (a0 >= n ? 1u << sa0 : 0u) | (a1 >= n ? 1u << sa1 : 0u);
        const int sl = __ffs(free_mask) - 1;
        free_mask &= ~(1u << sl);
#pragma unroll
        for (int nb = 0; nb < NB; nb++) slot.r[nb] = own(nb) == v ? sl : slot.r[nb];
        used_max = sl + 1 > used_max ? sl + 1 : used_max;
      }
      if (used_max > a.max_slots) status = kTooManySlots;
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        const int node = node_at.r[nb];
        const int ch0 = c0.gather(node), ch1 = c1.gather(node);
        s_node.r[nb] = node;
        s_c0.r[nb] = ch0;
        s_c1.r[nb] = ch1;
        s_sl.r[nb] = slot.gather(node) | (slot.gather(ch0) << 8) | (slot.gather(ch1) << 16) |
                     ((ch0 < n ? 1 : 0) << 24) | ((ch1 < n ? 1 : 0) << 25);
      }
    }
    // ---- schedule of the on-chip gradient kernel (see tree_setup_kernel) ----
    if (a.macros) {
      // stored (1) / unstored (2) classes, bottom-up
      Arr cls;
      cls.fill(0);
      for (int v = n; v < N - 1; v++) {
        const int a0 = c0.rd(v), a1 = c1.rd(v);
        const int k0c = cls.rd(a0), k1c = cls.rd(a1);
        const bool unstored = (a0 < n || k0c == 1) && (a1 < n || k1c == 1);
#pragma unroll
        for (int nb = 0; nb < NB; nb++) cls.r[nb] = own(nb) == v ? (unstored ? 2 : 1) : cls.r[nb];
      }
      Arr sslot;
      int stored_before = 0, macros_before = 0;
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        const int v = own(nb);
        cls.r[nb] = v == N - 1 ? 1 : cls.r[nb];
        const bool internal = v >= n && v < N;
        const bool stored = internal && cls.r[nb] == 1 && v != N - 1;
        const uint64_t stored_mask = __ballot(stored);
        const uint64_t macro_mask = __ballot(internal && cls.r[nb] == 1);
        const uint64_t below = (1ull << lane) - 1;
        // slots and macro indices in node-id order
        sslot.r[nb] = v == N - 1 ? -1 : stored_before + __popcll(stored_mask & below);
        is_macro[nb] = internal && cls.r[nb] == 1;
        macro_rank[nb] = macros_before + __popcll(macro_mask & below);
        stored_before += __popcll(stored_mask);
        macros_before += __popcll(macro_mask);
      }
      stored_total = stored_before;
      macro_total = macros_before;
      // every lane that owns a macro assembles it from its children's lanes (cross-lane
      // reads stay outside lane-dependent conditions: an inactive source lane reads as 0)
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        int kind[2];
        me[nb].node = own(nb);
        me[nb].pad = 0;
        me[nb].qslot = sslot.r[nb];
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int ch = j ? c1.r[nb] : c0.r[nb];
          const int cls_ch = cls.gather(ch), chs = sslot.gather(ch);
          const int ga_ = c0.gather(ch), gb_ = c1.gather(ch);
          const int cc = ch >= n ? cls_ch : 0;
          me[nb].child[j] = ch;
          kind[j] = cc;
          me[nb].cslot[j] = cc == 1 ? chs : 0;
          const bool expand = cc == 2;
          const int ga = expand ? ga_ : 0, gb = expand ? gb_ : 0;
          const int gas = sslot.gather(ga), gbs = sslot.gather(gb);
          me[nb].grand[2 * j] = ga;
          me[nb].grand[2 * j + 1] = gb;
          me[nb].gslot[2 * j] = ga >= n ? gas : 0;
          me[nb].gslot[2 * j + 1] = gb >= n ? gbs : 0;
        }
        me[nb].shape =
            macro_shape(kind[0], kind[1], own(nb) == N - 1, me[nb].child, me[nb].grand, n);
      }
      if (stored_total > max_stored(n)) status = kTooManySlots;
    }
  }
  if (status != kOk && lane == 0) set_status(a.status, status, t);
  const bool ok = status == kOk || status == kTooManySlots;
  if (!ok) {
    if (lane == 0 && a.macro_count) a.macro_count[t] = 0;
    for (int i = lane; i < n - 1; i += 64) sched[i] = {n + i, 0, 1, 0};
    for (int v = lane; v < N; v += 64) ble[v] = 0.0;
    return;
  }
#pragma unroll
  for (int nb = 0; nb < NB; nb++)
    if (own(nb) < n - 1) sched[own(nb)] = {s_node.r[nb], s_c0.r[nb], s_c1.r[nb], s_sl.r[nb]};
  if (a.macros) {
    MacroEntry* mac = a.macros + (size_t)t * max_macros(n);
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
      if (is_macro[nb]) mac[macro_rank[nb]] = me[nb];
    if (lane == 0) a.macro_count[t] = macro_total;
  }
  if (!a.rooted) {
    const double* bl = a.bl + (size_t)t * (N - 1);
    for (int v = lane; v < N; v += 64) ble[v] = v < N - 2 ? bl[v] : 0.0;
  } else {
    const double* bl = a.bl + (size_t)t * N;
    const double* rates = a.rates ? a.rates + (size_t)t * (N - 1) : nullptr;
    for (int v = lane; v < N; v += 64) ble[v] = (rates && v < N - 1) ? bl[v] * rates[v] : bl[v];
  }
}
#undef RDL
#undef WRL

// ------------------------------------------------------------------------
// Model setup (one thread per model instance).
// ------------------------------------------------------------------------
__device__ void jacobi4(const double* A_in, double* evals, double* U) {
  double A[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) A[i * 4 + j] = i >= j ? A_in[i * 4 + j] : A_in[j * 4 + i];
  for (int i = 0; i < 16; i++) U[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0, diag = 0;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        const double x = A[i * 4 + j] * A[i * 4 + j];
        if (i != j) off += x; else diag += x;
      }
    if (off <= 1e-40 * diag || off == 0.) break;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        const double apq = A[p * 4 + q];
        if (apq == 0.) continue;
        const double theta = (A[q * 4 + q] - A[p * 4 + p]) / (2. * apq);
        const double tt = (theta >= 0 ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
        const double c = 1. / sqrt(tt * tt + 1.), sn = tt * c;
        for (int k = 0; k < 4; k++) {
          const double akp = A[k * 4 + p], akq = A[k * 4 + q];
          A[k * 4 + p] = c * akp - sn * akq;
          A[k * 4 + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < 4; k++) {
          const double apk = A[p * 4 + k], aqk = A[q * 4 + k];
          A[p * 4 + k] = c * apk - sn * aqk;
          A[q * 4 + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < 4; k++) {
          const double ukp = U[k * 4 + p], ukq = U[k * 4 + q];
          U[k * 4 + p] = c * ukp - sn * ukq;
          U[k * 4 + q] = sn * ukp + c * ukq;
        }
      }
  }
  for (int i = 0; i < 4; i++) evals[i] = A[i * 4 + i];
  for (int i = 0; i < 3; i++) {
    int m = i;
    for (int j = i + 1; j < 4; j++)
      if (evals[j] < evals[m]) m = j;
    if (m != i) {
      const double tmp = evals[i]; evals[i] = evals[m]; evals[m] = tmp;
      for (int k = 0; k < 4; k++) {
        const double u = U[k * 4 + i]; U[k * 4 + i] = U[k * 4 + m]; U[k * 4 + m] = u;
      }
    }
  }
}

// stick_breaking_transform.cpp:20-43
__device__ void stick_breaking(int K, const double* y, double* x) {
  double stick = 1.0;
  for (int k = 0; k < K - 1; k++) {
    const double z = 1.0 / (1 + exp(-(y[k] - log((double)(K - k - 1)))));
    x[k] = stick * z;
    stick -= x[k];
  }
  x[K - 1] = stick;
}
__device__ void stick_breaking_inverse(int K, const double* x, double* y) {
  double sum = 0;
  for (int k = 0; k < K - 1; k++) {
    const double z = x[k] / (1.0 - sum);
    y[k] = log(z / (1.0 - z)) + log((double)(K - k - 1));
    sum += x[k];
  }
}

__global__ void model_setup_kernel(ModelSetupArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.T * a.models_per_tree) return;
  const int t = idx / a.models_per_tree, j = idx % a.models_per_tree;
  const double* row = a.params + (size_t)t * a.param_count;
  DevModel& m = a.models[idx];
  if (a.subst == 0) {
    // substitution_model.hpp:59-74 (JC69 eigensystem as hard-coded there)
    const double V[16] = {1.0, 2.0, 0.0, 0.5, 1.0, -2.0, 0.5, 0.0,
                          1.0, 2.0, 0.0, -0.5, 1.0, -2.0, -0.5, 0.0};
    const double Vi[16] = {0.25, 0.25, 0.25, 0.25, 0.125, -0.125, 0.125, -0.125,
                           0.0,  1.0,  0.0,  -1.0, 1.0,   0.0,    -1.0,  0.0};
    for (int i = 0; i < 4; i++) {
      m.pi[i] = 0.25;
      m.lambda[i] = i == 0 ? 0.0 : -1.3333333333333333;
      for (int k = 0; k < 4; k++) m.Q[i * 4 + k] = i == k ? -1.0 : 1.0 / 3.0;
    }
    for (int i = 0; i < 16; i++) {
      m.V[i] = V[i];
      m.Vinv[i] = Vi[i];
    }
  } else {
    // substitution_model.cpp:17-80, with the finite-difference perturbation of
    // fat_beagle.cpp:400-438 applied for j > 0: coordinate c of the
    // stick-breaking image of (frequencies | rates), sign +/-.
    double rates[6], freqs[4];
    for (int i = 0; i < 6; i++) rates[i] = row[a.rates_off + i];
    for (int i = 0; i < 4; i++) freqs[i] = row[a.freqs_off + i];
    double fsum = 0, rsum = 0;
    for (int i = 0; i < 4; i++) fsum += freqs[i];
    for (int i = 0; i < 6; i++) rsum += rates[i];
    if (j == 0) {
      if (fabs(fsum - 1.) >= 0.001) set_status(a.status, kGtrFrequencies, t);
      if (fabs(rsum - 1.) >= 0.001) set_status(a.status, kGtrRates, t);
    }
    if (j > 0) {
      const int coord = (j - 1) >> 1;
      const double delta = ((j - 1) & 1) ? -1.e-6 : 1.e-6;
      double y[5];
      if (coord < 3) {
        stick_breaking_inverse(4, freqs, y);
        y[coord] += delta;
        stick_breaking(4, y, freqs);
      } else {
        stick_breaking_inverse(6, rates, y);
        y[coord - 3] += delta;
        stick_breaking(6, y, rates);
      }
    }
    double Q[16];
    int ri = 0;
    for (int i = 0; i < 4; i++)
      for (int k = i + 1; k < 4; k++) {
        const double r = rates[ri++];
        Q[i * 4 + k] = r * freqs[k];
        Q[k * 4 + i] = r * freqs[i];
      }
    double total = 0;
    for (int i = 0; i < 4; i++) {
      double row_sum = 0;
      for (int k = 0; k < 4; k++)
        if (i != k) row_sum += Q[i * 4 + k];
      Q[i * 4 + i] = -row_sum;
      total += row_sum * freqs[i];
    }
    for (int i = 0; i < 16; i++) Q[i] /= total;
    double sq[4], S[16], U[16], ev[4];
    for (int i = 0; i < 4; i++) sq[i] = sqrt(freqs[i]);
    for (int i = 0; i < 4; i++)
      for (int k = 0; k < 4; k++) S[i * 4 + k] = sq[i] * Q[i * 4 + k] * (1.0 / sq[k]);
    jacobi4(S, ev, U);
    for (int i = 0; i < 4; i++) {
      m.pi[i] = freqs[i];
      m.lambda[i] = ev[i];
      for (int k = 0; k < 4; k++) {
        m.Q[i * 4 + k] = Q[i * 4 + k];
        m.V[i * 4 + k] = (1.0 / sq[i]) * U[i * 4 + k];
        m.Vinv[i * 4 + k] = U[k * 4 + i] * sq[k];
      }
    }
  }
  if (a.site == 0) {
    m.cat_rate[0] = 1.0;
    m.cat_weight[0] = 1.0;
    m.cat_drate[0] = 0.0;
  } else {
    // site_model.cpp:37-62
    const int K = a.K;
    const double shape = row[a.shape_off];
    double mean_rate = 0, mean_deriv = 0;
    for (int i = 0; i < K; i++) {
      const double quantile = (2.0 * i + 1.0) / (2.0 * K);
      const double r = pow(-log(1.0 - quantile), 1.0 / shape);
      m.cat_rate[i] = r;
      mean_rate += r;
      const double du = -r * log(-log(1.0 - quantile)) / (shape * shape);
      m.cat_drate[i] = du;
      mean_deriv += du;
    }
    mean_rate /= K;
    mean_deriv /= K;
    for (int i = 0; i < K; i++) {
      m.cat_drate[i] =
          (m.cat_drate[i] * mean_rate - m.cat_rate[i] * mean_deriv) / (mean_rate * mean_rate);
      m.cat_rate[i] /= mean_rate;
      m.cat_weight[i] = 1.0 / K;
    }
  }
}

// ------------------------------------------------------------------------
// Transition matrices (beagleUpdateTransitionMatrices, fat_beagle.cpp:304-314):
// one thread per (evaluation, edge, category).
// BEAGLE evaluates P = V diag(exp(l r t)) V^-1.  We evaluate the algebraically
// identical P = I + V diag(expm1(l r t)) V^-1: for small r t the BEAGLE form
// obtains the O(r t) off-diagonal entries as a difference of O(1) terms and
// loses ~1e-16/(r t) relative accuracy there (5e-14 relative in logL on the
// reference's fluA test, which its 2e-6 finite-difference divisor turns into 1e-3
// of gradient noise); the expm1 form agrees with an 80-bit evaluation to 1e-15.
// See DESIGN.md "Accuracy".
// ------------------------------------------------------------------------
constexpr int kTransitionBlock = 256;
__global__ __launch_bounds__(kTransitionBlock) void transition_kernel(TransitionArgs a) {
  // one thread per matrix; the 128-byte results are staged through LDS (row stride 17:
  // conflict-free) so that the block writes its 32 KB of output as whole cache lines
  __shared__ double stage[kTransitionBlock * 17];
  const long first = (long)blockIdx.x * kTransitionBlock;
  const long idx = first + threadIdx.x;
  const long total = (long)a.E * (a.N - 1) * a.K;
  double Pm[16];
  bool tip_edge = false;
  int mi_keep = 0;
  if (idx < total) {
    const int k = idx % a.K;
    const int edge = (idx / a.K) % (a.N - 1);
    const int e = idx / ((long)a.K * (a.N - 1));
    int t, mi;
    a.map.decode(e, t, mi);
    const DevModel& m = a.models[mi];
    const double bl = a.bl_eff[(size_t)t * a.N + edge];
    const double rt = m.cat_rate[k] * bl;
    double ex[4], W[16];
    for (int x = 0; x < 4; x++) ex[x] = expm1(m.lambda[x] * rt);
    for (int x = 0; x < 4; x++)
      for (int j = 0; j < 4; j++) W[x * 4 + j] = ex[x] * m.Vinv[x * 4 + j];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        double sum = i == j ? 1.0 : 0.0;
        for (int x = 0; x < 4; x++) sum += m.V[i * 4 + x] * W[x * 4 + j];
        Pm[i * 4 + j] = sum > 0 ? sum : 0;  // BEAGLE clamps negative probabilities to 0
        stage[threadIdx.x * 17 + i * 4 + j] = Pm[i * 4 + j];
      }
    tip_edge = edge < a.n;
    mi_keep = mi;
  }
  __syncthreads();
  const long left = total - first;
  const int count = (int)(left < kTransitionBlock ? left : kTransitionBlock) * 16;
  double* out = a.mats + first * 16;
  for (int x = threadIdx.x; x < count; x += kTransitionBlock) out[x] = stage[(x >> 4) * 17 + (x & 15)];
  if (a.tip_tables != nullptr) {
    // Tip edges: what a compact tip state st contributes to the forward sweep is a
    // COLUMN of P; a gap contributes 1 (rows of P sum to 1).  Tabulated per state so
    // that the VALU log-likelihood kernel fetches it with one 32-byte gather instead of
    // spending FP64 issue slots on one-hot vectors: table[st][i] = P[i][st], st = 0..4.
    // Written by the whole block from the staged matrices (contiguous destinations).
    for (int x = threadIdx.x; x < (count >> 4) * 20; x += kTransitionBlock) {
      const int m = x / 20, j = x - m * 20;
      const long id = first + m;
      const int k = id % a.K;
      const int edge = (id / a.K) % (a.N - 1);
      const int e = id / ((long)a.K * (a.N - 1));
      if (edge < a.n)
        a.tip_tables[(((size_t)e * a.n + edge) * a.K + k) * 20 + j] =
            j < 16 ? stage[m * 17 + (j & 3) * 4 + (j >> 2)] : 1.0;
    }
  }
  if (a.tr_mats != nullptr) {
    // Matrix of the matrix-core kernel's pre-order step, per edge: P again for an
    // internal edge (the kernel reads it transposed), and for a tip edge -- whose
    // derivative is (q_parent o sibling) . (P Q) e_state, one product instead of two --
    // (P Q) stored transposed so that the same transposed read yields it in forward layout.
    __syncthreads();
    if (tip_edge) {
      const DevModel& m = a.models[mi_keep];
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
          double pq = 0;
          for (int x = 0; x < 4; x++) pq += Pm[i * 4 + x] * m.Q[x * 4 + j];
          stage[threadIdx.x * 17 + j * 4 + i] = pq;
        }
    }
    __syncthreads();
    double* out2 = a.tr_mats + first * 16;
    for (int x = threadIdx.x; x < count; x += kTransitionBlock)
      out2[x] = stage[(x >> 4) * 17 + (x & 15)];
  }
  if (a.phi != nullptr) {
    // Analytic substitution gradient: d exp(Q tau) = V ((V^-1 dQ V) o Phi) V^-1 with the
    // divided differences Phi_ij = (e^{l_i tau} - e^{l_j tau}) / (l_i - l_j), Phi_ii =
    // tau e^{l_i tau}, tau = r_k t.  Evaluated as tau e^{l_j tau} expm1(x)/x, x = (l_i -
    // l_j) tau, which is stable for close and for equal eigenvalues.
    __syncthreads();
    if (idx < total) {
      const int k = idx % a.K;
      const int edge = (idx / a.K) % (a.N - 1);
      const int e = idx / ((long)a.K * (a.N - 1));
      int t, mi;
      a.map.decode(e, t, mi);
      const DevModel& m = a.models[mi];
      const double tau = m.cat_rate[k] * a.bl_eff[(size_t)t * a.N + edge];
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
          const double x = (m.lambda[i] - m.lambda[j]) * tau;
          const double r = fabs(x) < 1e-5 ? 1.0 + 0.5 * x + x * x * (1.0 / 6.0) : expm1(x) / x;
          stage[threadIdx.x * 17 + i * 4 + j] = tau * exp(m.lambda[j] * tau) * r;
        }
    }
    __syncthreads();
    double* out3 = a.phi + first * 16;
    for (int x = threadIdx.x; x < count; x += kTransitionBlock)
      out3[x] = stage[(x >> 4) * 17 + (x & 15)];
  }
}

// ------------------------------------------------------------------------
// On-chip log-likelihood (B5, B6, B11 of SURVEY.md 2.1).
// One wave per (evaluation, 64-pattern tile); rate categories are walked one
// after the other so that only floor(log2 n)+1 partial-likelihood vectors of one
// category are live, each in a lane-private LDS column (SoA: [slot][state][lane],
// conflict-free ds_read_b64 / ds_write_b64).  HBM traffic: tip states, the
// schedule and the transition matrices only.
// ------------------------------------------------------------------------
template <bool RESCALE, bool TIP_PARTIALS>
__global__ __launch_bounds__(kTile) void loglik_onchip_kernel(LikArgs a) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const TileEval te = xcd_tile_eval();
  const int tile = te.tile;
  const int e = a.eval_offset + te.eval;
  int t, mi;
  a.map.decode(e, t, mi);
  const DevModel* __restrict__ model = a.models + mi;
  const SchedEntry* __restrict__ sched = a.sched + (size_t)t * (a.n - 1);
  const int p = tile * kTile + lane;
  const int pc = p < a.P ? p : a.P - 1;
  const double w = p < a.P ? a.weights[pc] : 0.0;
  const int K = a.K, n = a.n;
  const double* __restrict__ mats_e = a.mats + (size_t)e * (a.N - 1) * K * 16;
  const double* __restrict__ tabs_e = a.tip_tables + (size_t)e * n * K * 20;
  // LDS: PLV columns [slot][state][lane] | this tile's tip states [taxon][lane]
  int8_t* tips = reinterpret_cast<int8_t*>(lds + (size_t)a.lds_slots * 4 * kTile);
  if (!TIP_PARTIALS) {
    for (int i = 0; i < n; i++) tips[i * kTile + lane] = a.tip_states[(size_t)i * a.P + pc];
  }
  __syncthreads();

  auto load_slot = [&](int slot) {
    const double* c = lds + slot * 4 * kTile + lane;
    return D4{c[0], c[kTile], c[2 * kTile], c[3 * kTile]};
  };
  // Addresses are "per-category base + 32-bit byte offset of the node": one scalar
  // multiply per address instead of 64-bit index arithmetic (which was most of the
  // scalar work of a visit).
  const unsigned node_bytes = (unsigned)K * 128u, tab_bytes = (unsigned)K * 160u;
  const char* mats_k = reinterpret_cast<const char*>(mats_e);  // advanced per category
  const char* tabs_k = reinterpret_cast<const char*>(tabs_e);
  auto mat = [&](int node) {
    return as_const(reinterpret_cast<const double*>(
        __builtin_assume_aligned(mats_k + (unsigned)node * node_bytes, 128)));
  };
  // What the tip children of a visit contribute is fetched one visit ahead:
  // compact states -> one 32-byte gather from the tip table (column of P, no
  // arithmetic); tip partials -> the partial vector itself.
  struct TipPre {
    D4 v[2];
  };
  auto fetch_tip = [&](const SchedEntry& s, int k) {
    TipPre d;
    const int c0 = s.child0 < n ? s.child0 : 0, c1 = s.child1 < n ? s.child1 : 0;
    if (TIP_PARTIALS) {
      d.v[0] = load4(a.tip_partials + ((size_t)c0 * a.P + pc) * 4);
      d.v[1] = load4(a.tip_partials + ((size_t)c1 * a.P + pc) * 4);
    } else {
      const int st0 = tips[c0 * kTile + lane], st1 = tips[c1 * kTile + lane];
      d.v[0] = load4(reinterpret_cast<const double*>(
          tabs_k + ((unsigned)c0 * tab_bytes + (unsigned)st0 * 32u)));
      d.v[1] = load4(reinterpret_cast<const double*>(
          tabs_k + ((unsigned)c1 * tab_bytes + (unsigned)st1 * 32u)));
    }
    return d;
  };
  auto touch = [&](const SchedEntry& s, int k) {
    const cint_ptr a0 = (cint_ptr)(uintptr_t)(mats_k + (unsigned)s.child0 * node_bytes);
    const cint_ptr a1 = (cint_ptr)(uintptr_t)(mats_k + (unsigned)s.child1 * node_bytes);
    return a0[0] ^ a0[16] ^ a1[0] ^ a1[16];
  };
  int touched = 0;

  double site = 0.0;
  int site_exp = 0;
  for (int k = 0; k < K; k++, mats_k += 128, tabs_k += 160) {
    int cum_exp = 0;
    D4 L = {0, 0, 0, 0};
    SchedEntry s_cur = sched[0];
    SchedEntry s_nxt = sched[n > 2 ? 1 : 0];
    TipPre td = fetch_tip(s_cur, k);
    for (int i = 0; i < n - 1; i++) {
      const SchedEntry s_nn = sched[i + 2 < n - 1 ? i + 2 : n - 2];
      asm volatile("" ::"s"(touched));
      touched = touch(s_nxt, k);
      const TipPre tdn = fetch_tip(s_nxt, k);
      const cdouble_ptr M0 = mat(s_cur.child0);
      const cdouble_ptr M1 = mat(s_cur.child1);
      const bool tip0 = s_cur.child0 < n, tip1 = s_cur.child1 < n;
      // PLV columns are read unconditionally (slot 0 for a tip), before any branch
      const D4 c0 = load_slot(tip0 ? 0 : (s_cur.slots >> 8) & 0xff);
      const D4 c1 = load_slot(tip1 ? 0 : (s_cur.slots >> 16) & 0xff);
      D4 A, B;
      if (tip0) A = TIP_PARTIALS ? matvec(M0, td.v[0]) : td.v[0];
      else A = matvec(M0, c0);
      if (tip1) B = TIP_PARTIALS ? matvec(M1, td.v[1]) : td.v[1];
      else B = matvec(M1, c1);
      L = mul4(A, B);
      if (RESCALE) {
        const int ex = max_exponent(max4(L));
        L = scale4(L, -ex);
        cum_exp += ex;
      }
      double* dst = lds + (s_cur.slots & 0xff) * 4 * kTile + lane;
      dst[0] = L.x0;
      dst[kTile] = L.x1;
      dst[2 * kTile] = L.x2;
      dst[3 * kTile] = L.x3;
      s_cur = s_nxt;
      s_nxt = s_nn;
      td = tdn;
    }
    // the last schedule entry is the root
    const double sk = model->cat_weight[k] * (model->pi[0] * L.x0 + model->pi[1] * L.x1 +
                                              model->pi[2] * L.x2 + model->pi[3] * L.x3);
    if (RESCALE) {
      if (k == 0) {
        site = sk;
        site_exp = cum_exp;
      } else if (cum_exp > site_exp) {
        site = ldexp(site, site_exp - cum_exp) + sk;
        site_exp = cum_exp;
      } else {
        site += ldexp(sk, cum_exp - site_exp);
      }
    } else {
      site += sk;
    }
  }
  asm volatile("" ::"s"(touched));
  if (!RESCALE && a.site_lik)
    a.site_lik[((size_t)a.grad_offset + te.eval) * a.tiles * kTile + p] = site;
  double ll = log(site);
  if (RESCALE) ll += site_exp * 0.6931471805599453;
  ll = p < a.P ? w * ll : 0.0;
  ll = wave_sum(ll);
  if (lane == 0) a.ll_part[(size_t)e * a.ll_tiles + tile] = ll;
}

// ------------------------------------------------------------------------
// Log-likelihood on the FP64 matrix cores (v_mfma_f64_4x4x4_4b_f64).
//
// The instruction multiplies four independent 4x4 blocks: D_b = A_b * B_b.
// Lane maps MEASURED on gfx950 (scratch probe): with lane =
// 16*hi + 4*b + lo, A_b[i][k] sits at (hi = k, lo = i), B_b[k][j] at (hi = k,
// lo = j) and D_b[i][j] at (hi = i, lo = j).  So a register holds, per lane, one
// state (hi) of one of 16 "columns" (b, lo); a product D is already in the layout
// the next product wants as B: partial-likelihood vectors flow from node to node
// with no data movement.
//   block b  = rate category (K = 4), or further pattern groups when K < 4
//   A        = the child's transition matrices, one element per lane: ONE 8-byte
//              load per lane fetches all categories' matrices (no SGPR traffic,
//              trivially prefetched a visit ahead)
//   R registers per node = R * 16/Kp site patterns per wave, all categories at once
// Measured issue rate: 18 cycles per instruction from one wave, 9 with two waves
// per SIMD (28 MAC/clk/SIMD, 1.8x the FP64 VALU peak), and the VALU stays free
// for the element-wise products.
// ------------------------------------------------------------------------
template <int R, bool RESCALE>
__global__ __launch_bounds__(kTile) void loglik_mfma_kernel(LikArgs a) {
  static_assert(R <= 4, "tip masks of one column group are packed in one 32-bit word");
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int hi = lane >> 4, b = (lane >> 2) & 3, lo = lane & 3;
  const TileEval te = xcd_tile_eval();
  const int e = a.eval_offset + te.eval;
  int t, mi;
  a.map.decode(e, t, mi);
  const DevModel* __restrict__ model = a.models + mi;
  const SchedEntry* __restrict__ sched = a.sched + (size_t)t * (a.n - 1);
  const int K = a.K, n = a.n, Kp = a.kp;       // Kp in {1, 2, 4}: categories per instruction
  const int cat = b % Kp, pgrp = b / Kp, ppr = 16 / Kp;  // ppr = patterns per register
  // K > 4: the categories are walked four at a time (`groups` complete walks that only
  // meet in the per-pattern site likelihood); the per-group lane constants follow
  const int groups = Kp == 4 ? (K + 3) / 4 : 1;
  const char* __restrict__ mats_e =
      reinterpret_cast<const char*>(a.mats + (size_t)e * (a.N - 1) * K * 16);
  unsigned a_off = 0;  // per-lane element of a child's matrix block: A_b[i = lo][k = hi] (bytes)
  double wgt = 0.0;    // category weight x stationary frequency of this lane
  const unsigned node_bytes = (unsigned)K * 128u;
  const int TP = ppr * R, tile_start = te.tile * TP;
  const int col = pgrp * 4 + lo;  // this lane's pattern column; register r adds r * ppr
  int pat[R];
  double pw[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    pat[r] = tile_start + r * ppr + col;
    pw[r] = a.weights[pat[r] < a.P ? pat[r] : a.P - 1];
  }
  // LDS: tip state masks [taxon][column][r] (bit s: compatible with state s; first, so
  // that the ignored mask fetch of an internal node id needs no clamping) | schedule |
  // vectors [slot][r][lane]
  uint8_t* tips = reinterpret_cast<uint8_t*>(lds);
  SchedEntry* sched_l = reinterpret_cast<SchedEntry*>(lds + ((n * ppr * 4 + 7) >> 3));
  double* plv = reinterpret_cast<double*>(sched_l + (n - 1));
  for (int i = lane; i < n - 1; i += kTile) sched_l[i] = sched[i];
  {
    const int tp_shift = TP <= 16 ? 4 : (TP <= 32 ? 5 : 6);
    const int q = lane & ((1 << tp_shift) - 1), group = 64 >> tp_shift;
    const int ppr_shift = Kp == 4 ? 2 : (Kp == 2 ? 3 : 4);
    const int r = q >> ppr_shift, c = q & (ppr - 1);
    if (q < TP) {
      const int pp = tile_start + q < a.P ? tile_start + q : a.P - 1;
      const uint8_t* src = a.tip_masks + pp;
#pragma unroll 4
      for (int taxon = lane >> tp_shift; taxon < n; taxon += group)
        tips[(taxon * ppr + c) * 4 + r] = src[(size_t)taxon * a.P];
    }
  }
  __syncthreads();

  // What a visit needs from memory is requested kAhead visits before it is used (the
  // schedule sits in LDS, so future visits' children are known): the two matrix
  // registers, the two tip words, and the entry itself.  Child ids stay vector
  // registers (multiplicands of per-lane addresses); the `slots` word, which also
  // carries the two is-a-tip flags, is the only scalar.
  constexpr int kAhead = 4;
  struct Ahead {
    double A0, A1;
    uint32_t w0, w1;
    int slots;
  };
  const unsigned lane8 = 8u * lane, col4 = 4u * col;
  auto request = [&](int i) {
    const SchedEntry sv = sched_l[i < n - 1 ? i : n - 2];
    int c0 = sv.child0, c1 = sv.child1;
    asm volatile("" : "+v"(c0), "+v"(c1));  // stay vector operands (see gradient_mfma_kernel)
    Ahead h;
    h.A0 = *reinterpret_cast<const double*>(mats_e + (__umul24((unsigned)c0, node_bytes) + a_off));
    h.A1 = *reinterpret_cast<const double*>(mats_e + (__umul24((unsigned)c1, node_bytes) + a_off));
    h.w0 = *reinterpret_cast<const uint32_t*>(tips + (__umul24((unsigned)c0, (unsigned)(ppr * 4)) + col4));
    h.w1 = *reinterpret_cast<const uint32_t*>(tips + (__umul24((unsigned)c1, (unsigned)(ppr * 4)) + col4));
    h.slots = sv.slots;
    return h;
  };
  auto slot_ptr = [&](int slot) {
    return reinterpret_cast<double*>(reinterpret_cast<char*>(plv) +
                                     ((unsigned)slot * (unsigned)(R * kTile * 8) + lane8));
  };
  double L[R];
  int esum[R];  // RESCALE: power-of-two exponents removed so far in this walk, per pattern
#pragma unroll
  for (int r = 0; r < R; r++) {
    L[r] = 0.0;
    esum[r] = 0;
  }
  auto visit = [&](int i, Ahead& h) {
    const int slots = __builtin_amdgcn_readfirstlane(h.slots);
    double B0[R], B1[R];
    if (slots & (1 << 24)) {
#pragma unroll
      for (int r = 0; r < R; r++) B0[r] = (double)__builtin_amdgcn_ubfe(h.w0, (uint32_t)(8 * r + hi), 1u);
    } else {
      const double* src = slot_ptr((slots >> 8) & 0xff);
#pragma unroll
      for (int r = 0; r < R; r++) B0[r] = src[r * kTile];
    }
    if (slots & (1 << 25)) {
#pragma unroll
      for (int r = 0; r < R; r++) B1[r] = (double)__builtin_amdgcn_ubfe(h.w1, (uint32_t)(8 * r + hi), 1u);
    } else {
      const double* src = slot_ptr((slots >> 16) & 0xff);
#pragma unroll
      for (int r = 0; r < R; r++) B1[r] = src[r * kTile];
    }
    const double A0 = h.A0, A1 = h.A1;
    h = request(i + kAhead);  // refill this ring slot
#pragma unroll
    for (int r = 0; r < R; r++) {
      const double D0 = __builtin_amdgcn_mfma_f64_4x4x4f64(A0, B0[r], 0.0, 0, 0, 0);
      const double D1 = __builtin_amdgcn_mfma_f64_4x4x4f64(A1, B1[r], 0.0, 0, 0, 0);
      L[r] = D0 * D1;
      if (RESCALE) {
        // exact per-pattern power-of-two rescaling (largest entry over states, categories)
        double mx = L[r];
        mx = fmax(mx, __shfl_xor(mx, 16, 64));
        mx = fmax(mx, __shfl_xor(mx, 32, 64));
        if (Kp >= 2) mx = fmax(mx, __shfl_xor(mx, 4, 64));
        if (Kp >= 4) mx = fmax(mx, __shfl_xor(mx, 8, 64));
        const int ex = mx > 0.0 ? ilogb(mx) : 0;
        L[r] = ldexp(L[r], -ex);
        esum[r] += ex;
      }
    }
    double* dst = slot_ptr(slots & 0xff);
#pragma unroll
    for (int r = 0; r < R; r++) dst[r * kTile] = L[r];
  };
  double site[R];
  int site_exp[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    site[r] = 0.0;
    site_exp[r] = 0;
  }
  for (int g = 0; g < groups; g++) {
    const int cat_g = 4 * g + cat;
    const int catc = cat_g < K ? cat_g : K - 1;  // padded category (weight 0) reads a valid matrix
    a_off = 8u * (catc * 16 + lo * 4 + hi);
    wgt = (cat_g < K ? model->cat_weight[cat_g] : 0.0) * model->pi[hi];
    Ahead ring[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; j++) ring[j] = request(j);
#pragma unroll
    for (int r = 0; r < R; r++) esum[r] = 0;
    for (int i = 0; i < n - 1; i += kAhead) {
#pragma unroll
      for (int j = 0; j < kAhead; j++)
        if (i + j < n - 1) visit(i + j, ring[j]);
    }
    // root: this group's share of the site likelihood = sum over its categories (blocks)
    // and the states (hi) of cw * pi * L; every lane of a pattern ends up with the sum
#pragma unroll
    for (int r = 0; r < R; r++) {
      double v = wgt * L[r];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (Kp >= 2) v += __shfl_xor(v, 4, 64);
      if (Kp >= 4) v += __shfl_xor(v, 8, 64);
      if (!RESCALE) {
        site[r] += v;
      } else if (g == 0) {
        site[r] = v;
        site_exp[r] = esum[r];
      } else if (esum[r] > site_exp[r]) {
        site[r] = ldexp(site[r], site_exp[r] - esum[r]) + v;
        site_exp[r] = esum[r];
      } else {
        site[r] += ldexp(v, esum[r] - site_exp[r]);
      }
    }
  }
  double ll = 0.0;
#pragma unroll
  for (int r = 0; r < R; r++) {
    const bool owner = hi == 0 && cat == 0 && pat[r] < a.P;  // one lane per pattern
    if (owner) {
      if (a.site_lik) {
        // per-pattern site likelihood for a following gradient pass (rescaled: the
        // mantissa here, the power of two in site_exp)
        const size_t at = ((size_t)a.grad_offset + te.eval) * a.tiles * kTile + pat[r];
        a.site_lik[at] = site[r];
        if (RESCALE) a.site_exp[at] = site_exp[r];
      }
      ll += pw[r] * (RESCALE ? log(site[r]) + site_exp[r] * 0.69314718055994530942 : log(site[r]));
    }
  }
  ll = wave_sum(ll);
  if (lane == 0) a.ll_part[(size_t)e * a.ll_tiles + te.tile] = ll;
}

// ------------------------------------------------------------------------
// Gradient v1 (B4-B11): post-order, pre-order and edge derivatives in one
// launch, partial-likelihood vectors streamed through HBM in the layout
// [evaluation][node][category][pattern][state] (32 B per lane, a wave reads or
// writes 2 KiB contiguous).  Each lane only ever re-reads what it wrote itself,
// so no inter-wave synchronisation is needed.  The pre-order vector of a node
// overwrites its post-order vector in place once the latter is dead.
// ------------------------------------------------------------------------
template <bool RESCALE, bool TIP_PARTIALS>
__global__ __launch_bounds__(kTile) void gradient_hbm_kernel(LikArgs a) {
  const int lane = threadIdx.x;
  const TileEval te = xcd_tile_eval();
  const int tile = te.tile;
  const int e = a.eval_offset + te.eval;
  const int gi = a.grad_offset + te.eval;
  int t, mi;
  a.map.decode(e, t, mi);
  const DevModel* __restrict__ model = a.models + mi;
  const SchedEntry* __restrict__ sched = a.sched + (size_t)t * (a.n - 1);
  const int p = tile * kTile + lane;
  const int pc = p < a.P ? p : a.P - 1;
  const double w = p < a.P ? a.weights[pc] : 0.0;
  const int K = a.K, n = a.n, N = a.N;
  const size_t ppad = (size_t)a.tiles * kTile;
  const double* __restrict__ mats_e = a.mats + (size_t)e * (N - 1) * K * 16;
  double* plv_e = a.plv + (size_t)te.eval * (n - 1) * K * ppad * 4 + (size_t)p * 4;
  double* gout = a.g_part + ((size_t)gi * a.g_tiles + tile) * 2 * N;

  auto plv_at = [&](int node, int k) { return plv_e + ((size_t)(node - n) * K + k) * ppad * 4; };
  auto tip_L = [&](int node) {
    if (TIP_PARTIALS) return load4(a.tip_partials + ((size_t)node * a.P + pc) * 4);
    return tip_vector(a.tip_states[(size_t)node * a.P + pc]);
  };

  // ---- post-order ----
  int cum_exp = 0;
  double site = 0.0;
  for (int i = 0; i < n - 1; i++) {
    const SchedEntry s = sched[i];
    const bool is_root = i == n - 2;
    double mx = 0.0;
    for (int k = 0; k < K; k++) {
      const double* __restrict__ M0 = mats_e + ((size_t)s.child0 * K + k) * 16;
      const double* __restrict__ M1 = mats_e + ((size_t)s.child1 * K + k) * 16;
      const D4 L0 = s.child0 < n ? tip_L(s.child0) : load4(plv_at(s.child0, k));
      const D4 L1 = s.child1 < n ? tip_L(s.child1) : load4(plv_at(s.child1, k));
      const D4 L = mul4(matvec(M0, L0), matvec(M1, L1));
      if (RESCALE) mx = fmax(mx, max4(L));
      if (is_root && !RESCALE) {
        site += model->cat_weight[k] * (model->pi[0] * L.x0 + model->pi[1] * L.x1 +
                                        model->pi[2] * L.x2 + model->pi[3] * L.x3);
      } else {
        store4(plv_at(s.node, k), L);
      }
    }
    if (RESCALE) {
      // common exponent across categories (the ratio in the edge derivative needs it)
      const int ex = max_exponent(mx);
      cum_exp += ex;
      for (int k = 0; k < K; k++) {
        const D4 L = scale4(load4(plv_at(s.node, k)), -ex);
        if (is_root)
          site += model->cat_weight[k] * (model->pi[0] * L.x0 + model->pi[1] * L.x1 +
                                          model->pi[2] * L.x2 + model->pi[3] * L.x3);
        else
          store4(plv_at(s.node, k), L);
      }
    }
  }
  {
    double ll = log(site);
    if (RESCALE) ll += cum_exp * 0.6931471805599453;
    ll = p < a.P ? w * ll : 0.0;
    ll = wave_sum(ll);
    if (lane == 0) a.ll_part[(size_t)e * a.ll_tiles + tile] = ll;
  }

  // ---- pre-order + edge derivatives, parents before children ----
  for (int i = n - 2; i >= 0; i--) {
    const SchedEntry s = sched[i];
    const bool is_root = i == n - 2;
    double nb0 = 0, ns0 = 0, den0 = 0, nb1 = 0, ns1 = 0, den1 = 0;
    double mx0 = 0, mx1 = 0;
    for (int k = 0; k < K; k++) {
      const double* __restrict__ M0 = mats_e + ((size_t)s.child0 * K + k) * 16;
      const double* __restrict__ M1 = mats_e + ((size_t)s.child1 * K + k) * 16;
      const D4 qv = is_root ? D4{model->pi[0], model->pi[1], model->pi[2], model->pi[3]}
                            : load4(plv_at(s.node, k));
      const D4 L0 = s.child0 < n ? tip_L(s.child0) : load4(plv_at(s.child0, k));
      const D4 L1 = s.child1 < n ? tip_L(s.child1) : load4(plv_at(s.child1, k));
      const D4 A = matvec(M0, L0), B = matvec(M1, L1);
      const D4 q0 = matTvec(M0, mul4(qv, B));
      const D4 q1 = matTvec(M1, mul4(qv, A));
      const double cw = model->cat_weight[k];
      const double n0 = cw * dot4(q0, matvec(model->Q, L0));
      const double n1 = cw * dot4(q1, matvec(model->Q, L1));
      nb0 += model->cat_rate[k] * n0;
      ns0 += model->cat_drate[k] * n0;
      den0 += cw * dot4(q0, L0);
      nb1 += model->cat_rate[k] * n1;
      ns1 += model->cat_drate[k] * n1;
      den1 += cw * dot4(q1, L1);
      if (s.child0 >= n) {
        store4(plv_at(s.child0, k), q0);
        if (RESCALE) mx0 = fmax(mx0, max4(q0));
      }
      if (s.child1 >= n) {
        store4(plv_at(s.child1, k), q1);
        if (RESCALE) mx1 = fmax(mx1, max4(q1));
      }
    }
    if (RESCALE) {
      if (s.child0 >= n) {
        const int ex = max_exponent(mx0);
        for (int k = 0; k < K; k++)
          store4(plv_at(s.child0, k), scale4(load4(plv_at(s.child0, k)), -ex));
      }
      if (s.child1 >= n) {
        const int ex = max_exponent(mx1);
        for (int k = 0; k < K; k++)
          store4(plv_at(s.child1, k), scale4(load4(plv_at(s.child1, k)), -ex));
      }
    }
    const double gb0 = wave_sum(p < a.P ? w * (nb0 / den0) : 0.0);
    const double gs0 = wave_sum(p < a.P ? w * (ns0 / den0) : 0.0);
    const double gb1 = wave_sum(p < a.P ? w * (nb1 / den1) : 0.0);
    const double gs1 = wave_sum(p < a.P ? w * (ns1 / den1) : 0.0);
    if (lane == 0) {
      gout[s.child0] = gb0;
      gout[N + s.child0] = gs0;
      gout[s.child1] = gb1;
      gout[N + s.child1] = gs1;
    }
  }
  if (lane == 0) {
    gout[N - 1] = 0.0;
    gout[N + N - 1] = 0.0;
  }
}

// ------------------------------------------------------------------------
// DPP helper of the matrix-core kernels: v[lane] += v[lane - SHIFT] within each 16-lane
// row (0 shifted in).
// ------------------------------------------------------------------------
template <int SHIFT>
__device__ __forceinline__ double row_shr_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int slo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + SHIFT, 0xf, 0xf, true);
  const int shi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + SHIFT, 0xf, 0xf, true);
  return v + __hiloint2double(shi, slo);
}

// ------------------------------------------------------------------------
// Gradient on the FP64 matrix cores: same half-storage walk as
// superseded VALU kernel of this round, same register layout as loglik_mfma_kernel.
//   * all rate categories sit in the four blocks of one instruction: no category
//     loop, and the per-pattern site likelihood (the derivative's denominator) is
//     computed by this kernel itself at the root -- no separate log-likelihood pass
//   * a node's matrices are ONE register (forward) + ONE register (transposed, or
//     (P Q) for a tip), fetched with a single 8-byte load per lane a whole macro ahead
//   * per edge: n = q (.) (Q L) lane-wise, weighted by w_p cw_k r_k / site_p and
//     summed over the whole wave (states, categories, patterns) by the transposed
//     butterfly; the four sums of an edge pair land in LDS at the pair's POSITION in
//     the schedule (macro index, child/grandchild), reduce_tiles_kernel maps positions
//     to node ids
//   * per-macro bookkeeping costs as much as the arithmetic, so it is
//     kept to a minimum: the schedule entry stays in vector registers (all lanes hold
//     the same values; addresses are one multiply-add per use), only its `shape` word
//     is made scalar for the control flow
// LDS per wave: max_stored(n) * R * 512 B of vectors (+ edge sums, tip state masks).
// ------------------------------------------------------------------------
template <int R, int DBG = 0, bool RESCALE = false, bool SUBST = false>
__global__ __launch_bounds__(kTile, 2) void gradient_mfma_kernel(LikArgs a) {
  static_assert(R <= 4, "tip masks of one column group are packed in one 32-bit word");
  extern __shared__ double glds[];
  const int lane = threadIdx.x;
  const int hi = lane >> 4, b = (lane >> 2) & 3, lo = lane & 3;
  const TileEval te = xcd_tile_eval();
  const int e = a.eval_offset + te.eval;
  const int gi = a.grad_offset + te.eval;
  int t, mi;
  a.map.decode(e, t, mi);
  const DevModel* __restrict__ model = a.models + mi;
  const int K = a.K, n = a.n, N = a.N, Kp = a.kp;
  // K > 4: a wave takes four categories (its category group) of its pattern tile; the
  // per-pattern site likelihood, the one quantity that couples the groups, then comes
  // from a preceding log-likelihood pass (a.site_lik) instead of this wave's own root.
  const int groups = a.cat_groups, tiles_per_group = gridDim.x / groups;
  const int group = te.tile / tiles_per_group, ptile = te.tile - group * tiles_per_group;
  const int cat = 4 * group + b % Kp, pgrp = b / Kp, ppr = 16 / Kp;
  const int catc = cat < K ? cat : K - 1;
  // forward matrices, and per edge the matrix of the pre-order step: P again (read
  // transposed) for an internal edge, (P Q) transposed for a tip edge
  const char* __restrict__ mats_e =
      reinterpret_cast<const char*>(a.mats + (size_t)e * (N - 1) * K * 16);
  const char* __restrict__ trm_e =
      reinterpret_cast<const char*>(a.tr_mats + (size_t)e * (N - 1) * K * 16);
  const char* __restrict__ phi_e =
      SUBST ? reinterpret_cast<const char*>(a.phi + (size_t)e * (N - 1) * K * 16) : nullptr;
  const MacroEntry* __restrict__ macros = a.macros + (size_t)t * max_macros(n);
  const int M = __builtin_amdgcn_readfirstlane(a.macro_count[t]);
  // byte offsets inside one node's K matrices
  const unsigned f_off = 8u * (catc * 16 + lo * 4 + hi);  // forward:    A[i=lo][k=hi] = P[lo][hi]
  const unsigned t_off = 8u * (catc * 16 + hi * 4 + lo);  // transposed: A[i=lo][k=hi] = P[hi][lo]
  const unsigned node_bytes = (unsigned)K * 128u;
  const int TP = ppr * R, tile_start = ptile * TP;
  const int col = pgrp * 4 + lo;  // this lane's pattern column; register r adds r * ppr
  int pat[R], patc[R];
  double pw[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    pat[r] = tile_start + r * ppr + col;
    patc[r] = pat[r] < a.P ? pat[r] : a.P - 1;
    pw[r] = pat[r] < a.P ? a.weights[patc[r]] : 0.0;
  }
  const double pi_l = model->pi[hi];
  const double cw_l = cat < K ? model->cat_weight[cat] : 0.0;
  const double rate_l = model->cat_rate[catc], drate_l = model->cat_drate[catc];
  const double AQ = model->Q[lo * 4 + hi];  // A operand for Q L (same in every block)
  // LDS: tip state masks [taxon][column][r] (one byte each: bit s set when the tip is
  // compatible with state s) | edge sums [macro][position][branch, site] | vectors
  // [slot][r][lane].  The masks come first so that the (ignored) mask fetch of an
  // internal node id lands in valid memory without clamping: N * 4 * ppr bytes from
  // the start is always inside the allocation.
  uint8_t* tips = reinterpret_cast<uint8_t*>(glds);
  const int gwidth = max_macros(n) * kMacroPositions * 2 + (SUBST ? kSubstExtra : 0);
  double* gacc = glds + ((n * ppr * 4 + 7) >> 3);
  double* plv = gacc + gwidth;
  // RESCALE: per (slot, pattern) power-of-two exponent taken out of a stored vector
  int32_t* exps = reinterpret_cast<int32_t*>(plv + (size_t)max_stored(n) * R * kTile);
  {
    // tip staging without divisions: the 64 lanes are (taxon group, pattern column) with
    // the column count rounded up to a power of two (TP = 4R, 8R or 16R, R <= 4)
    const int tp_shift = TP <= 16 ? 4 : (TP <= 32 ? 5 : 6);
    const int q = lane & ((1 << tp_shift) - 1), group = 64 >> tp_shift;
    const int ppr_shift = Kp == 4 ? 2 : (Kp == 2 ? 3 : 4);
    const int r = q >> ppr_shift, c = q & (ppr - 1);  // pattern tile_start + q = r * ppr + c
    if (q < TP) {
      const int pp = tile_start + q < a.P ? tile_start + q : a.P - 1;
      const uint8_t* src = a.tip_masks + pp;
#pragma unroll 4
      for (int taxon = lane >> tp_shift; taxon < n; taxon += group)
        tips[(taxon * ppr + c) * 4 + r] = src[(size_t)taxon * a.P];
    }
  }
  __syncthreads();
  if (M <= 0) return;
  if (DBG & 64) {
    if (lane == 0) a.ll_part[(size_t)e * a.ll_tiles + te.tile] = pw[0] + pi_l + cw_l + rate_l + drate_l + AQ;
    return;
  }

  struct V {
    double v[R];
  };
  // slots arrive as vector registers (the same value in every lane)
  const unsigned lane8 = 8u * lane;
  auto slot_ptr = [&](int slot) {
    // keep the (wave-uniform) slot in its vector register: the compiler would otherwise
    // move it to a scalar register first (a v_readfirstlane per field per macro)
    asm volatile("" : "+v"(slot));
    return reinterpret_cast<double*>(reinterpret_cast<char*>(plv) +
                                     (__umul24((unsigned)slot, (unsigned)(R * kTile * 8)) + lane8));
  };
  auto load_slot = [&](int slot) {
    V x;
    if (DBG & 8) {
#pragma unroll
      for (int r = 0; r < R; r++) x.v[r] = pi_l + slot;
      return x;
    }
    const double* c = slot_ptr(slot);
#pragma unroll
    for (int r = 0; r < R; r++) x.v[r] = c[r * kTile];
    return x;
  };
  auto store_slot = [&](int slot, const V& x) {
    if (DBG & 8) {
      asm volatile("" ::"v"(x.v[0]), "v"(x.v[R - 1]));
      return;
    }
    double* c = slot_ptr(slot);
#pragma unroll
    for (int r = 0; r < R; r++) c[r * kTile] = x.v[r];
  };
  auto mm = [&](double A, const V& x) {  // block-wise matrix product, R instructions
    V y;
#pragma unroll
    for (int r = 0; r < R; r++)
      y.v[r] = (DBG & 1) ? x.v[r] + A : __builtin_amdgcn_mfma_f64_4x4x4f64(A, x.v[r], 0.0, 0, 0, 0);
    return y;
  };
  auto mul = [&](const V& x, const V& y) {
    V z;
#pragma unroll
    for (int r = 0; r < R; r++) z.v[r] = x.v[r] * y.v[r];
    return z;
  };

  // ---- schedule entries: two 32-byte halves, loaded by every lane from one address ----
  struct Ids {  // first half: needed a macro ahead
    int shape, c0, c1, g0, g1, g2, g3;
  };
  struct Slots {  // second half: needed during the macro
    int q, cs0, cs1, gs0, gs1, gs2, gs3;
  };
  auto load_ids = [&](int m) {
    const int4* p = reinterpret_cast<const int4*>(macros + m);
    const int4 x = p[0], y = p[1];
    return Ids{x.x, x.y, x.z, x.w, y.x, y.y, y.z};
  };
  auto load_slots = [&](int m) {
    const int4* p = reinterpret_cast<const int4*>(macros + m) + 2;
    const int4 x = p[0], y = p[1];
    return Slots{x.x, x.y, x.z, x.w, y.x, y.y, y.z};
  };
  // shape word, scalar: bits 0-1 kind0, 2-3 kind1, 4 root, 8.. tip flags
  auto kind0 = [](int s) { return s & 3; };
  auto kind1 = [](int s) { return (s >> 2) & 3; };
  auto is_root = [](int s) { return (s & 16) != 0; };
  auto is_tip = [](int s, int j) { return ((s >> (8 + j)) & 1) != 0; };

  // matrix registers of one macro (children c0,c1 and grandchildren a0,b0,a1,b1) and the
  // tip state masks of this lane's column, fetched a macro ahead
  struct MacroMats {
    double f[6], tr[6];
    double ph[SUBST ? 6 : 1];  // SUBST: divided differences Phi of the six edges
    uint32_t tw[6];
  };
  const unsigned tt_delta = t_off - f_off;
  const unsigned col4 = 4u * col;
  auto fetch_mats = [&](const Ids& id, bool pre) {  // post-order needs the forward ones only
    MacroMats mt;
    int nodes[6] = {id.c0, id.c1, id.g0, id.g1, id.g2, id.g3};
#pragma unroll
    for (int j = 0; j < 6; j++) {
      asm volatile("" : "+v"(nodes[j]));  // stays a vector register (see slot_ptr)
      // (the compiler folds base + node * bytes + lane offset into one 64-bit multiply-add
      // per address; forcing "scalar base + 32-bit lane offset" addressing measured slower)
      const unsigned vf = __umul24((unsigned)nodes[j], node_bytes) + f_off;
      const unsigned vt = vf + tt_delta;
      if (DBG & 16) {
        mt.f[j] = pi_l + nodes[j];
        mt.tr[j] = pi_l - nodes[j];
      } else {
        mt.f[j] = *reinterpret_cast<const double*>(mats_e + vf);
        if (pre) mt.tr[j] = *reinterpret_cast<const double*>(trm_e + vt);
        if (pre && SUBST) mt.ph[j] = *reinterpret_cast<const double*>(phi_e + vt);
      }
      mt.tw[j] = *reinterpret_cast<const uint32_t*>(
          tips + (__umul24((unsigned)nodes[j], (unsigned)(ppr * 4)) + col4));
    }
    return mt;
  };
  auto tip_vector = [&](uint32_t w) {
    V x;
#pragma unroll
    for (int r = 0; r < R; r++)
      x.v[r] = (double)__builtin_amdgcn_ubfe(w, (uint32_t)(8 * r + hi), 1u);
    return x;
  };

  double qroot[R];  // root pre-order vector: pi * category weight * w_p / site likelihood
  int esum[R];      // RESCALE: exponents removed so far, per pattern
#pragma unroll
  for (int r = 0; r < R; r++) esum[r] = 0;
  // Operands of one macro: issued (LDS reads) before the next macro's matrices are
  // requested, consumed afterwards.
  struct Ops {
    V q, x0, y0, x1, y1;  // q: pre-order vector; child 0: x0 (,y0 when unstored); child 1
  };
  auto load_ops = [&](int sh, const Slots& sl, const MacroMats& cm, bool pre) {
    Ops o;
    if (pre) {
      if (is_root(sh)) {
#pragma unroll
        for (int r = 0; r < R; r++) o.q.v[r] = qroot[r];
      } else {
        o.q = load_slot(sl.q);
        if (RESCALE) {
#pragma unroll
          for (int r = 0; r < R; r++)
            o.q.v[r] = ldexp(o.q.v[r], -exps[__umul24((unsigned)sl.q, (unsigned)TP) +
                                              (unsigned)(r * ppr + col)]);
        }
      }
    }
    if (kind0(sh) == 2) {
      o.x0 = is_tip(sh, 2) ? tip_vector(cm.tw[2]) : load_slot(sl.gs0);
      o.y0 = is_tip(sh, 3) ? tip_vector(cm.tw[3]) : load_slot(sl.gs1);
    } else {
      o.x0 = is_tip(sh, 0) ? tip_vector(cm.tw[0]) : load_slot(sl.cs0);
    }
    if (kind1(sh) == 2) {
      o.x1 = is_tip(sh, 4) ? tip_vector(cm.tw[4]) : load_slot(sl.gs2);
      o.y1 = is_tip(sh, 5) ? tip_vector(cm.tw[5]) : load_slot(sl.gs3);
    } else {
      o.x1 = is_tip(sh, 1) ? tip_vector(cm.tw[1]) : load_slot(sl.cs1);
    }
    return o;
  };

  // ================= post-order over the stored nodes (+ root: site likelihood) ====
  auto post_step = [&](int sh, const Slots& sl, const MacroMats& cm, const Ops& o) {
    V L0, L1;
    if (kind0(sh) == 2) L0 = mul(mm(cm.f[2], o.x0), mm(cm.f[3], o.y0));
    else L0 = o.x0;
    if (kind1(sh) == 2) L1 = mul(mm(cm.f[4], o.x1), mm(cm.f[5], o.y1));
    else L1 = o.x1;
    V Lv = mul(mm(cm.f[0], L0), mm(cm.f[1], L1));
    if (!is_root(sh)) {
      if (RESCALE) {
        // Per-pattern power-of-two rescaling of every STORED vector (exact): the
        // exponent of the largest entry over states and categories is removed, summed
        // per pattern for the log-likelihood, and remembered for the pre-order walk.
        // With L_s = L 2^-E (E = exponents removed in the subtree) and q_s = q 2^E,
        // q_s o L_s is scale-free and q_child_s = P^T(q_s o P L_sib_s) 2^-e_parent, so the
        // only place an exponent re-enters is where a stored node's q is read back.
#pragma unroll
        for (int r = 0; r < R; r++) {
          double mx = Lv.v[r];
          mx = fmax(mx, __shfl_xor(mx, 16, 64));
          mx = fmax(mx, __shfl_xor(mx, 32, 64));
          if (Kp >= 2) mx = fmax(mx, __shfl_xor(mx, 4, 64));
          if (Kp >= 4) mx = fmax(mx, __shfl_xor(mx, 8, 64));
          const int ex = mx > 0.0 ? ilogb(mx) : 0;
          Lv.v[r] = ldexp(Lv.v[r], -ex);
          esum[r] += ex;
          exps[__umul24((unsigned)sl.q, (unsigned)TP) + (unsigned)(r * ppr + col)] = ex;
        }
      }
      store_slot(sl.q, Lv);
    } else {
      // root: site likelihood per pattern, log-likelihood partial, derivative weights
      double ll = 0.0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        double v;
        if (groups > 1) {
          const size_t at = ((size_t)a.grad_offset + te.eval) * a.tiles * kTile + patc[r];
          v = a.site_lik[at];
          // rescaled: q_root 2^(E of this group's walk) = pi cw w 2^(E - site_exp) / mantissa
          if (RESCALE) v = ldexp(v, a.site_exp[at] - esum[r]);
        } else {
          v = cw_l * pi_l * Lv.v[r];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          if (Kp >= 2) v += __shfl_xor(v, 4, 64);
          if (Kp >= 4) v += __shfl_xor(v, 8, 64);
        }
        qroot[r] = pi_l * cw_l * (pw[r] / v);  // pw = 0 for padding patterns
        if (hi == 0 && (b % Kp) == 0 && pat[r] < a.P)
          ll += pw[r] * (RESCALE ? log(v) + esum[r] * 0.69314718055994530942 : log(v));
      }
      ll = wave_sum(ll);
      if (lane == 0 && groups == 1) a.ll_part[(size_t)e * a.ll_tiles + te.tile] = ll;
      if (SUBST) {
        // d logL / d pi_c through the root: sum_p w_p sum_k cw_k L_root[c] / site_p;
        // this lane's state is c = hi, the 16 lanes of a row hold (category, pattern)
        double z = 0;
#pragma unroll
        for (int r = 0; r < R; r++) z += qroot[r] * Lv.v[r];  // qroot = pi cw w / site
        z = z / pi_l;
        z = row_shr_add<8>(z);
        z = row_shr_add<4>(z);
        z = row_shr_add<2>(z);
        z = row_shr_add<1>(z);
        if ((lane & 15) == 15) gacc[gwidth - 4 + hi] = z;
      }
    }
  };
  {
    // Two macros per iteration; everything ping-pongs between two register sets, so
    // nothing is copied.  At the top of macro m: ids(m+1) have arrived (requested a macro
    // ago) and give the addresses of the matrices / tip words of macro m+1; slots(m)
    // have arrived too; ids(m+2) and slots(m+1) are requested.
    Ids ia = load_ids(0), ib;
    Slots sa = load_slots(0), sb;
    int sha = __builtin_amdgcn_readfirstlane(ia.shape), shb = 0;
    MacroMats ma = fetch_mats(ia, false), mb;
    ib = load_ids(M > 1 ? 1 : 0);
    for (int m = 0; m < M; m += 2) {
      const Ops oa = load_ops(sha, sa, ma, false);
      shb = __builtin_amdgcn_readfirstlane(ib.shape);
      mb = fetch_mats(ib, false);
      ia = load_ids(m + 2 < M ? m + 2 : M - 1);
      sb = load_slots(m + 1 < M ? m + 1 : M - 1);
      post_step(sha, sa, ma, oa);
      if (m + 1 < M) {
        const Ops ob = load_ops(shb, sb, mb, false);
        sha = __builtin_amdgcn_readfirstlane(ia.shape);
        ma = fetch_mats(ia, false);
        ib = load_ids(m + 3 < M ? m + 3 : M - 1);
        sa = load_slots(m + 2 < M ? m + 2 : M - 1);
        post_step(shb, sb, mb, ob);
      }
    }
  }
  if (DBG & 128) return;
  // ================= pre-order + edge derivatives =================
  // The four sums of an edge pair (branch a, site a, branch b, site b) over the whole wave,
  // on the matrix cores: with the lane sums s (state = hi, pattern = lo) as A operand,
  //   D1[pattern][j] = sum_state s[state][pattern] * coef[state][j]     (coef: rate in column
  //                    0 / 2, d rate in column 1 / 3 for edge a / b; accumulated over both)
  //   D2[i][j]       = sum_pattern D1[pattern][j]                        (A = ones)
  // leaves, in every block, quantity j in the lanes with lo = j; two DPP row shifts add the
  // four blocks.  3 MFMA + 6 VALU instead of a 25-instruction cross-lane butterfly.
  const double coef_a = lo == 0 ? rate_l : (lo == 1 ? drate_l : 0.0);
  const double coef_b = lo == 2 ? rate_l : (lo == 3 ? drate_l : 0.0);
  auto edge_sums = [&](const V& na, const V& nb, int m, int pos_a) {
    // the pattern and category weights ride along in q (linear in the root vector)
    double sa = na.v[0], sb = nb.v[0];
#pragma unroll
    for (int r = 1; r < R; r++) {
      sa += na.v[r];
      sb += nb.v[r];
    }
    double red;
    if (DBG & 2) {
      red = rate_l * sa + drate_l * sb;
    } else {
      double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(sa, coef_a, 0.0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(sb, coef_b, d1, 0, 0, 0);
      red = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, d1, 0.0, 0, 0, 0);
      red = row_shr_add<4>(red);
      red = row_shr_add<8>(red);
    }
    // lanes 12..15 (block 3 of row 0) hold branch a, site a, branch b, site b
    if (lane >= 12 && lane < 16) {
      char* dst = reinterpret_cast<char*>(gacc) + (unsigned)((m * kMacroPositions + pos_a) * 16);
      *reinterpret_cast<double*>(dst + 8u * (unsigned)lo) = red;
    }
  };
  // ---- analytic substitution gradient (SUBST) ----
  // d logL = sum over edges, categories of <G, dP> with G = sum_p u L_c^T (u = q_parent o
  // sibling product, carrying the pattern / category weights).  With P = V e^{L tau} V^-1:
  // <G, dP> = <(V^T G V^-T) o Phi, V^-1 dQ V>, so one 4x4 per category,
  //   H = sum_edges (V^T G V^-T) o Phi,
  // accumulated in ONE register over the whole walk, is all the model gradient needs
  // (subst_gradient_kernel finishes: dlogL/dQ = V^-T H V^T, chain rule to the
  // parameters).  G is a matrix product over patterns, i.e. matrix-core work: operands
  // are the 4x4-block transposes of u and L_c (lane (hi, lo) <-> (lo, hi)).
  const int tr_lane = 16 * lo + 4 * b + hi;
  auto blockT = [&](double x) { return __shfl(x, tr_lane, 64); };
  const double AVt = SUBST ? model->V[hi * 4 + lo] : 0.0;     // A operand V^T: A[i][k] = V[k][i]
  const double AVi = SUBST ? model->Vinv[lo * 4 + hi] : 0.0;  // A operand V^-1
  double Ht = 0.0;  // (hi = i, lo = j) holds H^T[i][j] of this block's category
  auto subst_stats = [&](const V& u, const V& Lc, double phi) {
    double G = 0.0;
#pragma unroll
    for (int r = 0; r < R; r++)
      G = __builtin_amdgcn_mfma_f64_4x4x4f64(blockT(u.v[r]), blockT(Lc.v[r]), G, 0, 0, 0);
    const double R1 = __builtin_amdgcn_mfma_f64_4x4x4f64(AVt, G, 0.0, 0, 0, 0);            // V^T G
    const double R2 = __builtin_amdgcn_mfma_f64_4x4x4f64(AVi, blockT(R1), 0.0, 0, 0, 0);   // (V^T G V^-T)^T
    Ht += R2 * phi;  // Phi is symmetric
  };
  auto pre_step = [&](int sh, const Slots& sl, const MacroMats& cm, const Ops& o, int m) {
    const V& qv = o.q;
    const V &La0 = o.x0, &Lb0 = o.y0, &La1 = o.x1, &Lb1 = o.y1;
    V L0, L1, Ap0, Bp0, Ap1, Bp1;
    if (kind0(sh) == 2) {
      Ap0 = mm(cm.f[2], La0);
      Bp0 = mm(cm.f[3], Lb0);
      L0 = mul(Ap0, Bp0);
    } else {
      L0 = o.x0;
    }
    if (kind1(sh) == 2) {
      Ap1 = mm(cm.f[4], La1);
      Bp1 = mm(cm.f[5], Lb1);
      L1 = mul(Ap1, Bp1);
    } else {
      L1 = o.x1;
    }
    const V A = mm(cm.f[0], L0), B = mm(cm.f[1], L1);
    // Edge of child c below a node with pre-order vector q and sibling product S:
    //   internal child: q_c = P_c^T (q o S), numerator q_c o (Q L_c), q_c kept if stored
    //   tip child:      numerator (q o S) o ((P_c Q) L_c)  -- `trm` is then (P_c Q)
    auto edge = [&](double trm, const V& qs, const V& Lc, bool tip, int slot, bool keep, V& qc,
                    double phi) {
      if (SUBST) subst_stats(qs, Lc, phi);
      if (tip) return mul(qs, mm(trm, Lc));
      qc = mm(trm, qs);
      if (keep) store_slot(slot, qc);
      return mul(qc, mm(AQ, Lc));
    };
    V q0, q1;
    {
      const V n0 = edge(cm.tr[0], mul(qv, B), L0, is_tip(sh, 0), sl.cs0, kind0(sh) == 1, q0,
                        cm.ph[0]);
      const V n1 = edge(cm.tr[1], mul(qv, A), L1, is_tip(sh, 1), sl.cs1, kind1(sh) == 1, q1,
                        cm.ph[SUBST ? 1 : 0]);
      edge_sums(n0, n1, m, 0);
    }
    if (kind0(sh) == 2) {
      V qa, qb;
      const V na = edge(cm.tr[2], mul(q0, Bp0), La0, is_tip(sh, 2), sl.gs0, true, qa,
                        cm.ph[SUBST ? 2 : 0]);
      const V nb = edge(cm.tr[3], mul(q0, Ap0), Lb0, is_tip(sh, 3), sl.gs1, true, qb,
                        cm.ph[SUBST ? 3 : 0]);
      edge_sums(na, nb, m, 2);
    }
    if (kind1(sh) == 2) {
      V qa, qb;
      const V na = edge(cm.tr[4], mul(q1, Bp1), La1, is_tip(sh, 4), sl.gs2, true, qa,
                        cm.ph[SUBST ? 4 : 0]);
      const V nb = edge(cm.tr[5], mul(q1, Ap1), Lb1, is_tip(sh, 5), sl.gs3, true, qb,
                        cm.ph[SUBST ? 5 : 0]);
      edge_sums(na, nb, m, 4);
    }
  };
  {
    Ids ia = load_ids(M - 1), ib;
    Slots sa = load_slots(M - 1), sb;
    int sha = __builtin_amdgcn_readfirstlane(ia.shape), shb = 0;
    MacroMats ma = fetch_mats(ia, true), mb;
    ib = load_ids(M > 1 ? M - 2 : 0);
    for (int m = M - 1; m >= 0; m -= 2) {
      const Ops oa = load_ops(sha, sa, ma, true);
      shb = __builtin_amdgcn_readfirstlane(ib.shape);
      mb = fetch_mats(ib, true);
      ia = load_ids(m >= 2 ? m - 2 : 0);
      sb = load_slots(m >= 1 ? m - 1 : 0);
      pre_step(sha, sa, ma, oa, m);
      if (m >= 1) {
        const Ops ob = load_ops(shb, sb, mb, true);
        sha = __builtin_amdgcn_readfirstlane(ia.shape);
        ma = fetch_mats(ia, true);
        ib = load_ids(m >= 3 ? m - 3 : 0);
        sa = load_slots(m >= 2 ? m - 2 : 0);
        pre_step(shb, sb, mb, ob, m - 1);
      }
    }
  }
  __syncthreads();
  // positions that do not exist in a macro are never written nor read downstream
  double* gout = a.g_part + ((size_t)gi * a.g_tiles + te.tile) * gwidth;
  for (int i = lane; i < M * kMacroPositions * 2; i += kTile) gout[i] = gacc[i];
  if (SUBST) {
    gout[gwidth - kSubstExtra + lane] = Ht;
    if (lane < 4) gout[gwidth - 4 + lane] = gacc[gwidth - 4 + lane];
  }
}

// ------------------------------------------------------------------------
// Tile reduction (one workgroup per evaluation): four waves split the tiles
// (wave w takes tiles w, w+4, ...), combine through LDS in a fixed order.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_tiles_kernel(ReduceArgs a) {
  extern __shared__ double red_lds[];  // [4][W]
  __shared__ double llw[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int N2 = 2 * a.N;
  const int W = a.g_width ? a.g_width : N2;  // doubles per (evaluation, tile)
  double llp = 0;
  for (int i = threadIdx.x; i < a.ll_tiles; i += 256) llp += a.ll_part[(size_t)b * a.ll_tiles + i];
  llp = wave_sum(llp);
  if (lane == 0) llw[wv] = llp;
  const int t = b < a.T ? b : b - a.T;  // gradient evaluations: [0,T) main, [T,2T) site pass
  int used = W;
  if (b < a.Eg && a.g_width) used = a.macro_count[t] * kMacroPositions * 2;
  if (b < a.Eg) {
    const double* src = a.g_part + (size_t)b * a.g_tiles * W;
    const int tail = a.g_width ? W - a.extra : W;  // plain sums after the positional part
    for (int v = lane; v < used + a.extra; v += 64) {
      if (v >= used) v = tail + (v - used);
      double s0 = 0, s1 = 0;
      int i = wv;
      for (; i + 4 < a.g_tiles; i += 8) {
        s0 += src[(size_t)i * W + v];
        s1 += src[(size_t)(i + 4) * W + v];
      }
      if (i < a.g_tiles) s0 += src[(size_t)i * W + v];
      red_lds[wv * W + v] = s0 + s1;
      if (v >= tail) v = used + (v - tail);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) a.ll_sum[b] = (llw[0] + llw[1]) + (llw[2] + llw[3]);
  if (b >= a.Eg) return;
  double* out = a.g_sum + (size_t)b * N2;
  if (!a.g_width) {
    for (int v = threadIdx.x; v < N2; v += 256)
      out[v] = (red_lds[v] + red_lds[W + v]) + (red_lds[2 * W + v] + red_lds[3 * W + v]);
    return;
  }
  // positional: entry (m, pos, q) belongs to the edge above child/grandchild `pos` of macro m
  const MacroEntry* mac = a.macros + (size_t)t * max_macros(a.n);
  if (a.extra)
    for (int v = threadIdx.x; v < a.extra; v += 256) {
      const int c = W - a.extra + v;
      a.x_sum[(size_t)b * a.extra + v] =
          (red_lds[c] + red_lds[W + c]) + (red_lds[2 * W + c] + red_lds[3 * W + c]);
    }
  if (threadIdx.x < 2) out[threadIdx.x * a.N + a.N - 1] = 0.0;  // the root has no edge
  for (int v = threadIdx.x; v < used; v += 256) {
    const int m = v / (kMacroPositions * 2), r = v - m * (kMacroPositions * 2);
    const int pos = r >> 1, q = r & 1;
    const MacroEntry& me = mac[m];
    const bool exists = pos < 2 || ((me.shape >> (2 * ((pos - 2) >> 1))) & 3) == 2;
    if (!exists) continue;
    const int node = pos < 2 ? me.child[pos] : me.grand[pos - 2];
    out[q * a.N + node] =
        (red_lds[v] + red_lds[W + v]) + (red_lds[2 * W + v] + red_lds[3 * W + v]);
  }
}

// ------------------------------------------------------------------------
// Analytic substitution-model gradient, last step (one thread per tree).
// In: H^T per category block (64 doubles, lane order of gradient_mfma_kernel) and the
// root term d logL / d pi (4 doubles).  GTR as built by model_setup_kernel
// (substitution_model.cpp:17-80): Qt_ab = rho_ab pi_b, mu = sum_a pi_a sum_{b != a} Qt_ab,
// Q = Qt / mu.  Out: derivatives w.r.t. the stick-breaking coordinates of the rates (5)
// and of the frequencies (3), the quantities the reference obtains by finite
// differences (fat_beagle.cpp:400-465).
// ------------------------------------------------------------------------
__device__ void stick_breaking_chain(int K, const double* x, const double* g, double* out) {
  // x = stick_breaking(y): x_k = s_k z_k, s_k = prod_{j<k} (1 - z_j), x_{K-1} = s_{K-1};
  // dz_k/dy_k = z_k (1 - z_k)  =>  dL/dy_k = g_k x_k (1 - z_k) - z_k sum_{m>k} g_m x_m
  double tail[8];
  double acc = 0;
  for (int m = K - 1; m >= 0; m--) {
    tail[m] = acc;  // sum_{m' > m} g_m' x_m'
    acc += g[m] * x[m];
  }
  double used = 0;
  for (int k = 0; k < K - 1; k++) {
    const double z = x[k] / (1.0 - used);
    out[k] = g[k] * x[k] * (1.0 - z) - z * tail[k];
    used += x[k];
  }
}

__global__ void subst_gradient_kernel(SubstGradArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.T) return;
  const DevModel& m = a.models[t];
  const double* x = a.x_sum + (size_t)t * kSubstExtra;
  const double* row = a.params + (size_t)t * a.param_count;
  double H[16];  // H[i][j] = sum over blocks of H^T[j][i]
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double sum = 0;
      for (int b = 0; b < 4; b++) sum += x[16 * j + 4 * b + i];
      H[i * 4 + j] = sum;
    }
  // D = V^-T H V^T : D[a][b] = sum_ij Vinv[i][a] H[i][j] V[b][j]
  double HV[16], D[16];
  for (int i = 0; i < 4; i++)
    for (int b = 0; b < 4; b++) {
      double sum = 0;
      for (int j = 0; j < 4; j++) sum += H[i * 4 + j] * m.V[b * 4 + j];
      HV[i * 4 + b] = sum;
    }
  double S = 0;  // <D, Q>
  for (int c = 0; c < 4; c++)
    for (int b = 0; b < 4; b++) {
      double sum = 0;
      for (int i = 0; i < 4; i++) sum += m.Vinv[i * 4 + c] * HV[i * 4 + b];
      D[c * 4 + b] = sum;
      S += sum * m.Q[c * 4 + b];
    }
  double rates[6], pi[4];
  for (int i = 0; i < 6; i++) rates[i] = row[a.rates_off + i];
  for (int i = 0; i < 4; i++) pi[i] = row[a.freqs_off + i];
  double mu = 0;
  {
    int ri = 0;
    for (int i = 0; i < 4; i++)
      for (int k = i + 1; k < 4; k++) {
        const double r = rates[ri++];
        mu += pi[i] * r * pi[k] + pi[k] * r * pi[i];
      }
  }
  // d logL / d Qt_ab (a != b, diagonal follows) = (D_ab - D_aa - S pi_a) / mu
  auto dQt = [&](int c, int b) { return (D[c * 4 + b] - D[c * 4 + c] - S * pi[c]) / mu; };
  double g_rate[6], g_pi[4];
  for (int c = 0; c < 4; c++) g_pi[c] = x[64 + c] + S * m.Q[c * 4 + c];  // root term, explicit pi in mu
  {
    int ri = 0;
    for (int i = 0; i < 4; i++)
      for (int k = i + 1; k < 4; k++) {
        const double r = rates[ri];
        g_rate[ri] = pi[k] * dQt(i, k) + pi[i] * dQt(k, i);
        g_pi[k] += r * dQt(i, k);  // Qt_ik = r pi_k
        g_pi[i] += r * dQt(k, i);  // Qt_ki = r pi_i
        ri++;
      }
  }
  double* out = a.out_subst + (size_t)t * 8;
  stick_breaking_chain(6, rates, g_rate, out);
  stick_breaking_chain(4, pi, g_pi, out + 5);
}

// ------------------------------------------------------------------------
// Finalize (one thread per tree): sum tile partials in a fixed order
// (deterministic), assemble PhyloGradient, rooted chain rule.
// ------------------------------------------------------------------------
__device__ double sum_tiles(const double* part, int tiles) {
  double s = 0;
  for (int i = 0; i < tiles; i++) s += part[i];
  return s;
}

__device__ double node_partial(int v, int n, const double* h, const double* ratios,
                               const double* bound) {
  return (h[v] - bound[v]) / ratios[v - n];
}
__device__ double epoch_addition(int v, int c, int n, const double* h, const double* ratios,
                                 const double* bound, const double* acc) {
  if (c < n) return 0.0;
  if (bound[v] == bound[c]) return acc[c - n] * ratios[c - n] / ratios[v - n];
  return acc[c - n] * ratios[c - n] / (h[v] - bound[c]) * node_partial(v, n, h, ratios, bound);
}

// rooted_gradient_transforms.cpp:78-130 for one input vector gh -> out (+ root entry)
__device__ void ratio_transform(int n, const int32_t* c0, const int32_t* c1, const double* h,
                                const double* ratios, const double* bound, const double* gh,
                                double* mult, double* out) {
  const int N = 2 * n - 1, root = N - 1;
  for (int i = 0; i < n - 1; i++) out[i] = 0;
  for (int v = n; v < root; v++) {
    out[v - n] += node_partial(v, n, h, ratios, bound) * gh[v - n];
    out[v - n] += epoch_addition(v, c0[v - n], n, h, ratios, bound, out);
    out[v - n] += epoch_addition(v, c1[v - n], n, h, ratios, bound, out);
  }
  mult[root - n] = 1.0;
  for (int v = root; v >= n; v--) {
    const int a = c0[v - n], b = c1[v - n];
    if (a >= n) mult[a - n] = ratios[a - n] * mult[v - n];
    if (b >= n) mult[b - n] = ratios[b - n] * mult[v - n];
  }
  double sum = 0;
  for (int i = 0; i < n - 1; i++) sum += gh[i] * mult[i];
  out[root - n] = sum;
}

__global__ __launch_bounds__(64) void finalize_kernel(FinalizeArgs a) {
  // One wave per tree: lanes run over nodes / tiles for the reductions, lane 0
  // walks the O(n) recurrences of the rooted chain rule.  Working set (6n doubles, for
  // rooted trees also the tree's heights, bounds, ratios, rates and the ratio gradient
  // being built: each access of those recurrences is on a dependent chain, and a global
  // load there costs ten LDS reads) in LDS unless the tree is too large.
  extern __shared__ double fin_lds[];
  const int t = blockIdx.x, lane = threadIdx.x;
  const int n = a.n, N = a.N, T = a.T;
  double* base = a.use_lds ? fin_lds : a.scratch + (size_t)t * 6 * n;
  int32_t* c0 = reinterpret_cast<int32_t*>(base);
  int32_t* c1 = c0 + n;
  double* work = base + n;  // 5n doubles
  const SchedEntry* sched = a.sched + (size_t)t * (n - 1);
  for (int i = lane; i < n - 1; i += 64) {
    const SchedEntry se = sched[i];
    c0[se.node - n] = se.child0;
    c1[se.node - n] = se.child1;
  }
  const double* h = a.node_heights ? a.node_heights + (size_t)t * N : nullptr;
  const double* bd = a.node_bounds ? a.node_bounds + (size_t)t * N : nullptr;
  const double* ratios = a.height_ratios ? a.height_ratios + (size_t)t * (n - 1) : nullptr;
  const double* rates = a.rates ? a.rates + (size_t)t * (N - 1) : nullptr;
  double* outr_stage = nullptr;
  if (a.rooted && a.use_lds) {
    double* stage = fin_lds + 6 * n;  // h[N] | bd[N] | rates[N] | ratios[n] | out[n]
    for (int v = lane; v < N; v += 64) {
      if (h) stage[v] = h[v];
      if (bd) stage[N + v] = bd[v];
      if (rates && v < N - 1) stage[2 * N + v] = rates[v];
      if (ratios && v < n - 1) stage[3 * N + v] = ratios[v];
    }
    if (h) h = stage;
    if (bd) bd = stage + N;
    if (rates) rates = stage + 2 * N;
    if (ratios) ratios = stage + 3 * N;
    outr_stage = stage + 3 * N + n;
  }
  __syncthreads();
  __shared__ double sh_ll, sh_jac;
  if (lane == 0) {
    sh_ll = sum_tiles(a.ll_part + (size_t)t * a.ll_tiles, a.ll_tiles);
    double jac = 0.0;
    if (a.rooted && (a.with_jacobian || (a.gradient && a.gtr))) {
      // fat_beagle.cpp:82-94; iteration order of TripleIdPreorderBifurcating
      // (node.cpp:226-261) reproduced with an explicit stack in `work`.
      int32_t* st = reinterpret_cast<int32_t*>(work);
      int top = 0;
      st[top++] = (N - 1) << 1;
      while (top) {
        const int item = st[--top];
        const int v = item >> 1;
        const int a0 = c0[v - n], a1 = c1[v - n];
        if (item & 1) {
          if (a1 >= n) {
            jac += log(h[v] - bd[a1]);
            st[top++] = a1 << 1;
          }
        } else {
          st[top++] = (v << 1) | 1;
          if (a0 >= n) {
            jac += log(h[v] - bd[a0]);
            st[top++] = a0 << 1;
          }
        }
      }
    }
    sh_jac = jac;
  }
  __syncthreads();
  const double ll = sh_ll, jac = sh_jac;
  if (!a.gradient) {
    if (lane == 0) a.out_ll[t] = a.with_jacobian ? ll + jac : ll;
    return;
  }
  if (lane == 0) a.out_ll[t] = ll;
  const double* ble = a.bl_eff + (size_t)t * N;
  // branch gradient of the main evaluation: tile partials summed in tile order
  double* bg = work;  // N doubles (N < 2n)
  for (int v = lane; v < N; v += 64) {
    double sum = 0;
    for (int i = 0; i < a.g_tiles; i++)
      sum += a.g_part[(((size_t)t * a.g_tiles + i) * 2) * N + v];
    bg[v] = sum;
  }
  if (a.out_site && (a.site_fused || a.site_separate)) {
    // DiscreteSiteModelGradient fat_beagle.cpp:389-398
    const size_t gi = a.site_separate ? (size_t)T + t : (size_t)t;
    double r = 0;
    for (int v = lane; v < N - 1; v += 64) {
      double sum = 0;
      for (int i = 0; i < a.g_tiles; i++)
        sum += a.g_part[((gi * a.g_tiles + i) * 2 + 1) * N + v];
      r += sum * ble[v];
    }
    r = wave_sum(r);
    if (lane == 0) a.out_site[t] = r;
  }
  if (a.gtr && a.out_subst && lane < 8) {
    // fat_beagle.cpp:431,455-464: rates (5) then frequencies (3)
    const int coord = lane < 5 ? 3 + lane : lane - 5;
    const size_t ep = (size_t)T + (size_t)t * 16 + 2 * coord;
    double lp = sum_tiles(a.ll_part + ep * a.ll_tiles, a.ll_tiles);
    double lm = sum_tiles(a.ll_part + (ep + 1) * a.ll_tiles, a.ll_tiles);
    if (a.rooted) {
      lp += jac;
      lm += jac;
    }
    a.out_subst[(size_t)t * 8 + lane] = (lp - lm) / (2. * 1.e-6);
  }
  __syncthreads();
  if (!a.rooted) {
    double* ob = a.out_branch + (size_t)t * N;
    // fixed node = second child of the root (fat_beagle.cpp:499); root entry is 0
    for (int v = lane; v < N; v += 64) ob[v] = v < N - 2 ? bg[v] : 0.0;
    return;
  }
  // ---- rooted: clock + ratios/root-height gradients ----
  const double* tb = a.bl_raw + (size_t)t * N;
  double* oc = a.out_clock + (size_t)t * (N - 1);
  const int rc = a.rate_counts[t];
  if (rc == 1) {
    // ClockGradient fat_beagle.cpp:367-387 (strict): sum_i g_i * t_i
    double acc = 0;
    for (int v = lane; v < N - 1; v += 64) acc += bg[v] * tb[v];
    acc = wave_sum(acc);
    for (int v = lane; v < N - 1; v += 64) oc[v] = v == 0 ? acc : 0.0;
  } else if (rc == N - 1) {
    for (int v = lane; v < N - 1; v += 64) oc[v] = bg[v] * tb[v];
  } else {
    if (lane == 0) set_status(a.status, kBadRateCount, t);
    for (int v = lane; v < N - 1; v += 64) oc[v] = 0;
  }
  double* out_global = a.out_ratios + (size_t)t * (n - 1);
  if (outr_stage) {
    // Working set in LDS: everything that is not a recurrence is done by all lanes (height
    // gradient, the per-node coefficients of the ratio transform, the root sums), and the
    // two ratio transforms (height gradient, log-Jacobian) share ONE bottom-up and one
    // top-down chain of multiply-adds (rooted_gradient_transforms.cpp:17-170).
    const int root = N - 1;
    double* rw = outr_stage + n;  // 8n doubles
    double *hg = rw, *aux = rw + n, *Pv = rw + 2 * n, *E0 = rw + 3 * n, *E1 = rw + 4 * n;
    double *outA = rw + 5 * n, *outB = rw + 6 * n, *mult = rw + 7 * n;
    for (int i = lane; i < n - 1; i += 64) {
      const int v = n + i, a0 = c0[i], a1 = c1[i];
      // HeightGradient :17-37
      double x = v != root ? -bg[v] * rates[v] : 0.0;
      x += bg[a0] * rates[a0];
      x += bg[a1] * rates[a1];
      hg[i] = x;
      aux[i] = i < n - 2 ? 1.0 / (h[v] - bd[v]) : 0.0;  // d log|J| / d height
      // out_v = partial_v gh_v + sum over internal children c of out_c * epoch(v, c) :47-64
      const double partial = v != root ? (h[v] - bd[v]) / ratios[i] : 0.0;
      auto epoch = [&](int c) {
        if (c < n || v == root) return 0.0;
        if (bd[v] == bd[c]) return ratios[c - n] / ratios[i];
        return ratios[c - n] / (h[v] - bd[c]) * partial;
      };
      Pv[i] = partial;
      E0[i] = epoch(a0);
      E1[i] = epoch(a1);
    }
    __syncthreads();
    if (lane == 0) {
      for (int i = 0; i < n - 2; i++) {
        const int a0 = c0[i] >= n ? c0[i] - n : 0, a1 = c1[i] >= n ? c1[i] - n : 0;
        outA[i] = Pv[i] * hg[i] + E0[i] * outA[a0] + E1[i] * outA[a1];
        outB[i] = Pv[i] * aux[i] + E0[i] * outB[a0] + E1[i] * outB[a1];
      }
      mult[root - n] = 1.0;  // :102-130
      for (int v = root; v >= n; v--) {
        const int a0 = c0[v - n], a1 = c1[v - n];
        const double m = mult[v - n];
        if (a0 >= n) mult[a0 - n] = ratios[a0 - n] * m;
        if (a1 >= n) mult[a1 - n] = ratios[a1 - n] * m;
      }
    }
    __syncthreads();
    double ra = 0, rb = 0;
    for (int i = lane; i < n - 1; i += 64) {
      ra += hg[i] * mult[i];
      rb += aux[i] * mult[i];
    }
    ra = wave_sum(ra);
    rb = wave_sum(rb);
    for (int i = lane; i < n - 2; i += 64) out_global[i] = outA[i] + (outB[i] - 1.0 / ratios[i]);
    if (lane == 0) out_global[n - 2] = ra + rb;
    return;
  }
  double* outr = out_global;
  if (lane == 0) {
  double* hg = work + 2 * n;    // n-1
  double* aux = work + 3 * n;   // n-1 (log_time)
  double* jacg = work + 4 * n;  // n-1
  // HeightGradient rooted_gradient_transforms.cpp:17-37
  for (int v = N - 1; v >= n; v--) {
    double x = v != N - 1 ? -bg[v] * rates[v] : 0.0;
    x += bg[c0[v - n]] * rates[c0[v - n]];
    x += bg[c1[v - n]] * rates[c1[v - n]];
    hg[v - n] = x;
  }
  double* mult = work;  // bg is dead from here on
  ratio_transform(n, c0, c1, h, ratios, bd, hg, mult, outr);
  for (int i = 0; i < n - 1; i++) aux[i] = 0;
  for (int i = 0; i < n - 2; i++) aux[i] = 1.0 / (h[n + i] - bd[n + i]);
  ratio_transform(n, c0, c1, h, ratios, bd, aux, mult, jacg);
  for (int i = 0; i < n - 2; i++) outr[i] += jacg[i] - 1.0 / ratios[i];
  outr[n - 2] += jacg[n - 2];
  }
}

}  // namespace

// ------------------------------------------------------------------------
// Launch wrappers
// ------------------------------------------------------------------------
// Kernels that want more than 64 KiB of dynamic LDS have to opt in, per function AND per
// device (a process may drive several GPUs): remembered per (function, device).
static void allow_large_lds(const void* func, size_t lds) {
  if (lds <= 64 * 1024) return;
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> configured;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = configured[{func, dev}];
  if (lds > have) {
    (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    have = lds;
  }
}

void launch_tree_setup(const TreeSetupArgs& a_in, hipStream_t s) {
  TreeSetupArgs a = a_in;
  const size_t lds = sizeof(int32_t) * 13 * (size_t)(2 * a.n - 1);
  a.use_lds = lds <= 48 * 1024;
  // MI_PHYLO_TREE_SETUP=lds forces the general kernel (testing)
  static const bool force_lds = [] {
    const char* env = getenv("MI_PHYLO_TREE_SETUP");
    return env && std::string(env) == "lds";
  }();
  const int N = 2 * a.n - 1;
  if (a.n >= 3 && N <= 256 && !force_lds) {
    if (N <= 64) hipLaunchKernelGGL(tree_setup_small_kernel<1>, dim3(a.T), dim3(64), 0, s, a);
    else if (N <= 128) hipLaunchKernelGGL(tree_setup_small_kernel<2>, dim3(a.T), dim3(64), 0, s, a);
    else if (N <= 192) hipLaunchKernelGGL(tree_setup_small_kernel<3>, dim3(a.T), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(tree_setup_small_kernel<4>, dim3(a.T), dim3(64), 0, s, a);
    return;
  }
  hipLaunchKernelGGL(tree_setup_kernel, dim3(a.T), dim3(64), a.use_lds ? lds : 0, s, a);
}
void launch_model_setup(const ModelSetupArgs& a, hipStream_t s) {
  const int total = a.T * a.models_per_tree;
  hipLaunchKernelGGL(model_setup_kernel, dim3((total + 63) / 64), dim3(64), 0, s, a);
}
void launch_transition(const TransitionArgs& a, hipStream_t s) {
  const long total = (long)a.E * (a.N - 1) * a.K;
  hipLaunchKernelGGL(transition_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     a);
}
// which log-likelihood kernel runs by default when both can (measured)
constexpr bool kLoglikMfmaDefault = true;
static size_t loglik_mfma_lds_bytes(int n, int K, int max_slots) {
  const int kp = K == 1 ? 1 : (K == 2 ? 2 : 4);
  const size_t tip_bytes = (((size_t)n * 4 * (16 / kp) + 7) / 8) * 8;
  const size_t bytes = tip_bytes + sizeof(SchedEntry) * (size_t)(n - 1) +
                       sizeof(double) * (size_t)max_slots * kLlR * kTile;
  const size_t reach = (size_t)(2 * n - 1) * 4 * (16 / kp);  // mask fetches of internal ids
  return bytes > reach ? bytes : reach;
}
int loglik_mfma_tiles(int P, int K) {
  const int kp = K == 1 ? 1 : (K == 2 ? 2 : 4);
  const int per_wave = kLlR * (16 / kp);
  return (P + per_wave - 1) / per_wave;
}
bool loglik_mfma_supported(const LikArgs& a, bool rescale) {
  // The matrix-core log-likelihood kernel needs tips in state-mask form (K > 4: the
  // categories are walked four at a time).  MI_PHYLO_LOGLIK_PATH=valu|mfma forces one of the two kernels.
  static const int forced = [] {
    const char* env = getenv("MI_PHYLO_LOGLIK_PATH");
    if (!env) return 0;
    return std::string(env) == "mfma" ? 2 : (std::string(env) == "valu" ? 1 : 0);
  }();
  (void)rescale;
  const bool possible = a.K <= kMaxCategories && a.tip_masks != nullptr;
  if (forced == 1) return false;
  if (forced == 2) return possible;
  return possible && kLoglikMfmaDefault;
}
static bool use_loglik_mfma(const LikArgs& a, bool rescale, int max_slots) {
  return loglik_mfma_supported(a, rescale) &&
         loglik_mfma_lds_bytes(a.n, a.K, max_slots) <= 160 * 1024;
}
static void launch_loglik_mfma(const LikArgs& a_in, int count, bool rescale, int max_slots,
                               hipStream_t s) {
  LikArgs a = a_in;
  a.lds_slots = max_slots;
  a.kp = a.K == 1 ? 1 : (a.K == 2 ? 2 : 4);
  const dim3 grid(loglik_mfma_tiles(a.P, a.K), count), block(kTile);
  const size_t lds = loglik_mfma_lds_bytes(a.n, a.K, max_slots);
  if (rescale) {
    allow_large_lds(reinterpret_cast<const void*>(loglik_mfma_kernel<kLlR, true>), lds);
    hipLaunchKernelGGL((loglik_mfma_kernel<kLlR, true>), grid, block, lds, s, a);
  } else {
    allow_large_lds(reinterpret_cast<const void*>(loglik_mfma_kernel<kLlR, false>), lds);
    hipLaunchKernelGGL((loglik_mfma_kernel<kLlR, false>), grid, block, lds, s, a);
  }
}
void launch_loglik(const LikArgs& a_in, int count, bool rescale, int max_slots, hipStream_t s) {
  if (count <= 0) return;
  if (use_loglik_mfma(a_in, rescale, max_slots)) {
    launch_loglik_mfma(a_in, count, rescale, max_slots, s);
    return;
  }
  LikArgs a = a_in;
  a.lds_slots = max_slots;
  const dim3 grid(a.tiles, count), block(kTile);
  const size_t lds = (size_t)max_slots * 4 * kTile * sizeof(double) + (size_t)a.n * kTile;
  const bool tp = a.tip_partials != nullptr;
  if (rescale) {
    if (tp) hipLaunchKernelGGL((loglik_onchip_kernel<true, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((loglik_onchip_kernel<true, false>), grid, block, lds, s, a);
  } else {
    if (tp) hipLaunchKernelGGL((loglik_onchip_kernel<false, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((loglik_onchip_kernel<false, false>), grid, block, lds, s, a);
  }
}
void launch_gradient_hbm(const LikArgs& a, int count, bool rescale, hipStream_t s) {
  if (count <= 0) return;
  const dim3 grid(a.tiles, count), block(kTile);
  const bool tp = a.tip_partials != nullptr;
  if (rescale) {
    if (tp) hipLaunchKernelGGL((gradient_hbm_kernel<true, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gradient_hbm_kernel<true, false>), grid, block, 0, s, a);
  } else {
    if (tp) hipLaunchKernelGGL((gradient_hbm_kernel<false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gradient_hbm_kernel<false, false>), grid, block, 0, s, a);
  }
}
size_t gradient_mfma_lds_bytes(int n, int K, bool rescale, bool subst) {
  const int kp = K == 1 ? 1 : (K == 2 ? 2 : 4);
  const size_t tip_bytes = (((size_t)n * 4 * (16 / kp) + 7) / 8) * 8;
  size_t bytes = tip_bytes + sizeof(double) * ((size_t)max_stored(n) * kLlR * kTile +
                                               gradient_mfma_width(n, subst));
  if (rescale) bytes += sizeof(int32_t) * (size_t)max_stored(n) * kLlR * (16 / kp);
  const size_t reach = (size_t)(2 * n - 1) * 4 * (16 / kp);  // mask fetches of internal ids
  return bytes > reach ? bytes : reach;
}
int gradient_mfma_width(int n, bool subst) {
  return max_macros(n) * kMacroPositions * 2 + (subst ? kSubstExtra : 0);
}
int gradient_mfma_groups(int K) { return K <= 4 ? 1 : (K + 3) / 4; }
bool gradient_mfma_fits(int n, int K, bool rescale) {
  return n >= 3 && K <= kMaxCategories &&
         gradient_mfma_lds_bytes(n, K, rescale, true) <= 160 * 1024;
}
template <bool RESCALE, bool SUBST>
static void launch_gradient_mfma_variant(const LikArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  allow_large_lds(reinterpret_cast<const void*>(gradient_mfma_kernel<kLlR, 0, RESCALE, SUBST>),
                  lds);
  hipLaunchKernelGGL((gradient_mfma_kernel<kLlR, 0, RESCALE, SUBST>), grid, dim3(kTile), lds, s, a);
}
void launch_gradient_mfma(const LikArgs& a_in, int count, bool rescale, bool subst,
                          hipStream_t s) {
  if (count <= 0) return;
  LikArgs a = a_in;
  a.kp = a.K == 1 ? 1 : (a.K == 2 ? 2 : 4);
  a.cat_groups = gradient_mfma_groups(a.K);
  const size_t lds = gradient_mfma_lds_bytes(a.n, a.K, rescale, subst);
  const dim3 grid(loglik_mfma_tiles(a.P, a.K) * a.cat_groups, count);
  if (rescale || subst) {
    if (rescale && subst) launch_gradient_mfma_variant<true, true>(a, grid, lds, s);
    else if (rescale) launch_gradient_mfma_variant<true, false>(a, grid, lds, s);
    else launch_gradient_mfma_variant<false, true>(a, grid, lds, s);
    return;
  }
  // ablation builds: 1 no matrix products, 2 no cross-lane reductions,
  // 8 no LDS vector traffic, 16 no matrix loads, 64 prologue only, 128 post-order only
  static const int dbg = getenv("MI_PHYLO_DEBUG") ? atoi(getenv("MI_PHYLO_DEBUG")) : 0;
  switch (dbg) {
    case 1: hipLaunchKernelGGL((gradient_mfma_kernel<kLlR, 1>), grid, dim3(kTile), lds, s, a); return;
    case 2: hipLaunchKernelGGL((gradient_mfma_kernel<kLlR, 2>), grid, dim3(kTile), lds, s, a); return;
    case 8: hipLaunchKernelGGL((gradient_mfma_kernel<kLlR, 8>), grid, dim3(kTile), lds, s, a); return;
    case 16: hipLaunchKernelGGL((gradient_mfma_kernel<kLlR, 16>), grid, dim3(kTile), lds, s, a); return;
    case 27: hipLaunchKernelGGL((gradient_mfma_kernel<kLlR, 27>), grid, dim3(kTile), lds, s, a); return;
    case 64: hipLaunchKernelGGL((gradient_mfma_kernel<kLlR, 64>), grid, dim3(kTile), lds, s, a); return;
    case 128: hipLaunchKernelGGL((gradient_mfma_kernel<kLlR, 128>), grid, dim3(kTile), lds, s, a); return;
    default: break;
  }
  launch_gradient_mfma_variant<false, false>(a, grid, lds, s);
}
void launch_subst_gradient(const SubstGradArgs& a, hipStream_t s) {
  if (a.T <= 0) return;
  hipLaunchKernelGGL(subst_gradient_kernel, dim3((a.T + 63) / 64), dim3(64), 0, s, a);
}
bool reduce_tiles_fits(int N) {
  return sizeof(double) * 4 * (size_t)(3 * N + 12 + kSubstExtra) <= 64 * 1024;
}
void launch_reduce_tiles(const ReduceArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  const size_t W = a.g_width ? a.g_width : 2 * (size_t)a.N;
  hipLaunchKernelGGL(reduce_tiles_kernel, dim3(a.E), dim3(256), sizeof(double) * 4 * W, s, a);
}
void launch_finalize(const FinalizeArgs& a_in, hipStream_t s) {
  FinalizeArgs a = a_in;
  // 6n of working set (+ for a rooted tree its staged state, 3N + 2n, and 8n of
  // coefficients and partial results)
  const size_t lds = sizeof(double) * (a.rooted ? 22 * (size_t)a.n : 6 * (size_t)a.n);
  a.use_lds = lds <= 48 * 1024;
  hipLaunchKernelGGL(finalize_kernel, dim3(a.T), dim3(64), a.use_lds ? lds : 0, s, a);
}

const char* loglik_kernel_name(const LikArgs& a, bool rescale, int max_slots) {
  return use_loglik_mfma(a, rescale, max_slots) ? "loglik_mfma_kernel" : "loglik_onchip_kernel";
}
const char* gradient_kernel_name() { return "gradient_hbm_kernel"; }
const char* gradient_mfma_kernel_name() { return "gradient_mfma_kernel"; }

}  // namespace miphylo
