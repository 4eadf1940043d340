// Device code of the per-tree set-up that more than one kernel file needs (each gets its own
// copy): the register-array tree walk for trees of up to 256 nodes (tree_setup_small_kernel;
// round 5: also the set-up role of the one-launch small call, kernels_walk3.hip) and the
// model instances.
#pragma once
#include <hip/hip_runtime.h>

#include "mi_phylo_device_utils.h"
#include "mi_phylo_kernels.h"

namespace miphylo {
namespace dev {

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

// What the walk over one small tree leaves in the wave's registers: the schedule (entry i in
// lane i % 64 of register i / 64), the macro each lane owns (lane = id of the macro's node),
// its index in the tree's macro list, their number, and the status.
template <int NB>
struct SmallTree {
  NodeArray<NB> s_node, s_c0, s_c1, s_sl;
  MacroEntry me[NB];
  bool is_macro[NB];
  int macro_rank[NB];
  int macro_total;
  int status;
};

// (t: the tree; lane: 0..63 of the one wave that builds it; lds: 256 NB ints of LDS of the
// wave's own.  Reads a.parent_ids only.)
constexpr int kSmallTreeLdsInts = 256;  // per NB
template <int NB>
__device__ __forceinline__ void small_tree_build(const TreeSetupArgs& a, const int t, const int lane,
                                                 SmallTree<NB>& out, int* lds) {
  // Whatever can be done by all lanes at once is (child lists, sorting, the macro entries);
  // the bottom-up recurrences (largest leaf id, stored / unstored classes) run in ROUNDS --
  // every node whose children are done, at once -- i.e. as many steps as the tree is high
  // (round 5; until then one node per step: 76 steps of ~30 instructions for a DS1 tree, 5 of
  // the 7 microseconds of the set-up kernel); the Sethi-Ullman walks of the log-likelihood
  // schedule stay sequential, straight-line code (lane selects / scalar selects: a taken
  // scalar branch costs more than the handful of instructions it would skip).
  auto lds_fence = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  using Arr = NodeArray<NB>;
  const int n = a.n, N = 2 * n - 1;
  const int nodes_in = a.rooted ? N : N - 1;
  const int root_in = nodes_in - 1;
  const int32_t* par_in = a.parent_ids + (size_t)t * (nodes_in - 1);
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

  // Every node registers with its parent (an LDS counter per parent hands out the places: the
  // order of arrival is arbitrary, the -- at most three -- children are sorted by largest leaf
  // id below, all nodes at once), then the largest leaf ids bottom-up in rounds.
  Arr cnt, k0, k1, k2;
  cnt.fill(0); k0.fill(0); k1.fill(0); k2.fill(0);
  if (status == kOk) {
    int* cnt_l = lds;            // [64 NB]
    int* kid_l = lds + 64 * NB;  // [64 NB][3]
#pragma unroll
    for (int nb = 0; nb < NB; nb++) cnt_l[own(nb)] = 0;
    lds_fence();
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
      if (own(nb) < nodes_in - 1) {
        const int p = par.r[nb];
        const int place = atomicAdd(&cnt_l[p], 1);
        if (place < 3) kid_l[3 * p + place] = own(nb);
      }
    lds_fence();
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      const int v = own(nb);
      const int c = cnt_l[v];
      cnt.r[nb] = c;
      k0.r[nb] = c >= 1 ? kid_l[3 * v] : 0;
      k1.r[nb] = c >= 2 ? kid_l[3 * v + 1] : 0;
      k2.r[nb] = c >= 3 ? kid_l[3 * v + 2] : 0;
    }
    for (int round = 0; round < nodes_in; round++) {
      bool open = false;
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        const int c = cnt.r[nb];
        const int m0 = maxleaf.gather(k0.r[nb]), m1 = maxleaf.gather(k1.r[nb]), m2 = maxleaf.gather(k2.r[nb]);
        const bool mine = own(nb) >= n && own(nb) < nodes_in && maxleaf.r[nb] < 0;
        const bool done = (c < 1 || m0 >= 0) && (c < 2 || m1 >= 0) && (c < 3 || m2 >= 0);
        int mx = c >= 1 ? m0 : 0;  // (a childless internal node is an error, found below)
        mx = (c >= 2 && m1 > mx) ? m1 : mx;
        mx = (c >= 3 && m2 > mx) ? m2 : mx;
        maxleaf.r[nb] = (mine && done) ? mx : maxleaf.r[nb];
        open |= mine && !done;
      }
      if (!__any(open)) break;
    }
    // ascending largest leaf id (keys of siblings differ: disjoint leaf sets); absent children
    // sort last
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      const int c = cnt.r[nb];
      int a = k0.r[nb], b = k1.r[nb], d = k2.r[nb];
      int ma = maxleaf.gather(a), mb = maxleaf.gather(b), md = maxleaf.gather(d);
      ma = c >= 1 ? ma : 0x7fffffff;
      mb = c >= 2 ? mb : 0x7fffffff;
      md = c >= 3 ? md : 0x7fffffff;
      auto order2 = [](int& x, int& mx, int& y, int& my) {
        const bool swap = mx > my;
        const int tx = x, tm = mx;
        x = swap ? y : x;
        mx = swap ? my : mx;
        y = swap ? tx : y;
        my = swap ? tm : my;
      };
      order2(a, ma, b, mb);
      order2(b, mb, d, md);
      order2(a, ma, b, mb);
      k0.r[nb] = a;
      k1.r[nb] = b;
      k2.r[nb] = d;
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
      Arr cls;  // (-1: not known yet; in rounds, as the largest leaf ids above)
#pragma unroll
      for (int nb = 0; nb < NB; nb++) cls.r[nb] = (own(nb) >= n && own(nb) < N - 1) ? -1 : 0;
      for (int round = 0; round < N; round++) {
        bool open = false;
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
          const int a0 = c0.r[nb], a1 = c1.r[nb];
          const int k0c = cls.gather(a0), k1c = cls.gather(a1);
          const bool mine = cls.r[nb] < 0;
          const bool done = k0c >= 0 && k1c >= 0;
          const bool unstored = (a0 < n || k0c == 1) && (a1 < n || k1c == 1);
          cls.r[nb] = (mine && done) ? (unstored ? 2 : 1) : cls.r[nb];
          open |= mine && !done;
        }
        if (!__any(open)) break;
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
  out.s_node = s_node;
  out.s_c0 = s_c0;
  out.s_c1 = s_c1;
  out.s_sl = s_sl;
#pragma unroll
  for (int nb = 0; nb < NB; nb++) {
    out.me[nb] = me[nb];
    out.is_macro[nb] = is_macro[nb];
    out.macro_rank[nb] = macro_rank[nb];
  }
  out.macro_total = macro_total;
  out.status = status;
}

// Writes what small_tree_build left in registers: schedule, macros and their count, effective
// branch lengths (Detrifurcate zeros / x rates), the status word.
template <int NB>
__device__ __forceinline__ void small_tree_store(const TreeSetupArgs& a, const int t, const int lane,
                                                 const SmallTree<NB>& r) {
  const int n = a.n, N = 2 * n - 1;
  SchedEntry* sched = a.sched + (size_t)t * (n - 1);
  double* ble = a.bl_eff + (size_t)t * N;
  auto own = [&](int nb) { return lane + 64 * nb; };
  const int status = r.status;
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
    if (own(nb) < n - 1) sched[own(nb)] = {r.s_node.r[nb], r.s_c0.r[nb], r.s_c1.r[nb], r.s_sl.r[nb]};
  if (a.macros) {
    MacroEntry* mac = a.macros + (size_t)t * macro_stride(n);
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
      if (r.is_macro[nb]) mac[r.macro_rank[nb]] = r.me[nb];
    if (lane == 0) a.macro_count[t] = r.macro_total;
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
__device__ inline void jacobi4(const double* A_in, double* evals, double* U) {
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
    // (off, diag: SQUARED norms.  Until round 5 the bound was 1e-40 -- an off-diagonal norm of
    // 1e-20 of the diagonal's, which double precision reaches only by accident: the loop ran
    // 40 to 60 sweeps of ~3.4 microseconds on its one lane, 110-140 microseconds of EVERY GTR
    // call's set-up.  1e-31 is an off-diagonal norm of 3e-16 of the diagonal's -- the rounding
    // level; cyclic Jacobi converges quadratically and gets there in 5 or 6 sweeps.)
    if (off <= 1e-31 * diag || off == 0.) break;
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
__device__ inline void stick_breaking(int K, const double* y, double* x) {
  double stick = 1.0;
  for (int k = 0; k < K - 1; k++) {
    const double z = 1.0 / (1 + exp(-(y[k] - log((double)(K - k - 1)))));
    x[k] = stick * z;
    stick -= x[k];
  }
  x[K - 1] = stick;
}
__device__ inline void stick_breaking_inverse(int K, const double* x, double* y) {
  double sum = 0;
  for (int k = 0; k < K - 1; k++) {
    const double z = x[k] / (1.0 - sum);
    y[k] = log(z / (1.0 - z)) + log((double)(K - k - 1));
    sum += x[k];
  }
}

// The substitution model of instance (tree t, perturbation j) into m: frequencies, rate matrix,
// eigensystem.
__device__ inline void subst_model_into(const ModelSetupArgs& a, const int t, const int j, const double* row,
                                        DevModel& m) {
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
}

// site_model.cpp:37-62 in two steps shared by the one-thread and the one-wave form below (explicitly
// rounded operations: both forms must produce the same bits whatever the compiler would
// contract around them): rate and d rate / d shape of category i before the normalisation ...
// (table: {x_i, log x_i} per category, x_i = -log(1 - quantile_i) -- they depend on the
// category count only and are computed once per engine by weibull_table_kernel with these very
// expressions, which leaves the power as the one transcendental of a category; null: inline)
__device__ __forceinline__ void weibull_category(const int K, const double shape, const int i, double& r,
                                                 double& du, const double* table = nullptr) {
  double x, lx;
  if (table) {
    x = table[2 * i];
    lx = table[2 * i + 1];
  } else {
    const double quantile = (2.0 * i + 1.0) / (2.0 * K);
    x = -log(1.0 - quantile);
    lx = log(x);
  }
  r = pow(x, 1.0 / shape);
  du = __ddiv_rn(__dmul_rn(-r, lx), __dmul_rn(shape, shape));
}
// ... and the normalisation by the mean rate
__device__ __forceinline__ void weibull_normalise(const double mean_rate, const double mean_deriv, double& r,
                                                  double& du) {
  du = __ddiv_rn(__dsub_rn(__dmul_rn(du, mean_rate), __dmul_rn(r, mean_deriv)), __dmul_rn(mean_rate, mean_rate));
  r = __ddiv_rn(r, mean_rate);
}

__device__ inline void site_model_into(const ModelSetupArgs& a, const double* row, DevModel& m) {
  if (a.site == 0) {
    m.cat_rate[0] = 1.0;
    m.cat_weight[0] = 1.0;
    m.cat_drate[0] = 0.0;
    return;
  }
  const int K = a.K;
  const double shape = row[a.shape_off];
  double mean_rate = 0, mean_deriv = 0;
  for (int i = 0; i < K; i++) {
    double r, du;
    weibull_category(K, shape, i, r, du, a.weibull_x);
    m.cat_rate[i] = r;
    m.cat_drate[i] = du;
    mean_rate = __dadd_rn(mean_rate, r);
    mean_deriv = __dadd_rn(mean_deriv, du);
  }
  mean_rate = __ddiv_rn(mean_rate, (double)K);
  mean_deriv = __ddiv_rn(mean_deriv, (double)K);
  for (int i = 0; i < K; i++) {
    weibull_normalise(mean_rate, mean_deriv, m.cat_rate[i], m.cat_drate[i]);
    m.cat_weight[i] = 1.0 / K;
  }
}

// model instance idx (tree idx / models_per_tree, perturbation idx % models_per_tree) into m
__device__ inline void model_setup_into(const ModelSetupArgs& a, int idx, DevModel& m) {
  const int t = idx / a.models_per_tree, j = idx % a.models_per_tree;
  const double* row = a.params + (size_t)t * a.param_count;
  subst_model_into(a, t, j, row, m);
  site_model_into(a, row, m);
}

// The same instance (perturbation 0) by ONE WAVE, into LDS or HBM: the categories of the site
// model a lane each (a power and two logarithms per category are most of a JC69 instance's
// cost: 5 of the 6 microseconds one thread takes for four categories), their sums in index
// order; JC69's constants by sixteen lanes; a GTR eigensystem stays one lane's work.  Same
// bits as model_setup_into.
__device__ __forceinline__ void model_setup_wave(const ModelSetupArgs& a, const int t, const int lane, DevModel& m) {
  const double* row = a.params + (size_t)t * a.param_count;
  if (a.subst == 0) {
    if (lane < 16) {
      // substitution_model.hpp:59-74 (the tables of subst_model_into)
      const double V[16] = {1.0, 2.0, 0.0, 0.5, 1.0, -2.0, 0.5, 0.0,
                            1.0, 2.0, 0.0, -0.5, 1.0, -2.0, -0.5, 0.0};
      const double Vi[16] = {0.25, 0.25, 0.25, 0.25, 0.125, -0.125, 0.125, -0.125,
                             0.0,  1.0,  0.0,  -1.0, 1.0,   0.0,    -1.0,  0.0};
      const int i = lane >> 2, k = lane & 3;
      const double v = V[lane], vi = Vi[lane];
      m.V[lane] = v;
      m.Vinv[lane] = vi;
      m.Q[lane] = i == k ? -1.0 : 1.0 / 3.0;
      if (lane < 4) {
        m.pi[lane] = 0.25;
        m.lambda[lane] = lane == 0 ? 0.0 : -1.3333333333333333;
      }
    }
  } else if (lane == 0) {
    subst_model_into(a, t, 0, row, m);
  }
  if (a.site == 0) {
    if (lane == 0) {
      m.cat_rate[0] = 1.0;
      m.cat_weight[0] = 1.0;
      m.cat_drate[0] = 0.0;
    }
    return;
  }
  const int K = a.K;
  const double shape = row[a.shape_off];
  double r = 0, du = 0;
  if (lane < K) weibull_category(K, shape, lane, r, du, a.weibull_x);
  auto lane_value = [](double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                            __builtin_amdgcn_readlane(__double2loint(x), l));
  };
  double mean_rate = 0, mean_deriv = 0;
  for (int i = 0; i < K; i++) {
    mean_rate = __dadd_rn(mean_rate, lane_value(r, i));
    mean_deriv = __dadd_rn(mean_deriv, lane_value(du, i));
  }
  mean_rate = __ddiv_rn(mean_rate, (double)K);
  mean_deriv = __ddiv_rn(mean_deriv, (double)K);
  if (lane < K) {
    weibull_normalise(mean_rate, mean_deriv, r, du);
    m.cat_rate[lane] = r;
    m.cat_drate[lane] = du;
    m.cat_weight[lane] = 1.0 / K;
  }
}
__device__ inline void model_setup_thread(const ModelSetupArgs& a, int idx) {
  if (idx >= a.T * a.models_per_tree) return;
  model_setup_into(a, idx, a.models[idx]);
}

}  // namespace dev
}  // namespace miphylo
