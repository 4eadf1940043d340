// Per-call set-up kernels: tree schedules, model eigensystems, transition matrices.
// (gfx950 / CDNA4, wave64; see DESIGN.md for the mapping and what bounds each kernel.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "mi_phylo_device_utils.h"
#include "mi_phylo_kernels.h"
#include "mi_phylo_setup_device.h"
#include "mi_phylo_macro_slots_device.h"

namespace miphylo {

namespace {
using namespace dev;

// ------------------------------------------------------------------------
// Tree setup: parent-id vector -> evaluation schedule (one thread per tree).
// Restates node.cpp:32-59 (children ordered by max leaf id),
// unrooted_tree.cpp:27-37 (Detrifurcate), tree.cpp:72-78 (SlideRootPosition, a
// no-op on a detrifurcated tree), fat_beagle.cpp:96-101,507-511 (x rates).
// The schedule lists internal nodes in a post-order chosen by Sethi-Ullman
// labels so that the on-chip kernel needs at most floor(log2 n)+1 live
// partial-likelihood vectors; any post-order gives bitwise the same vectors.
// ------------------------------------------------------------------------
// The model instances are independent of the trees: their set-up (one thread per model
// instance, defined below) rides in the same launch as extra workgroups behind the T
// tree workgroups -- one dispatch less per call, and the two run side by side.

__global__ __launch_bounds__(64) void tree_setup_kernel(TreeSetupArgs a, ModelSetupArgs ms) {
  if ((int)blockIdx.x >= a.T) {
    model_setup_thread(ms, ((int)blockIdx.x - a.T) * 64 + (int)threadIdx.x);
    return;
  }
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
        MacroEntry* mac = a.macros + (size_t)t * macro_stride(n);
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
// Tree setup for large trees, one WORKGROUP per tree (a thread per node): the sequential walk
// above takes about 2 000 cycles per node (dependent LDS round trips of one lane), 0.84 ms for
// the 512-taxon trees of the 20-state configuration -- more than their log-likelihood
// kernel takes on one of eight pattern shards.  Every quantity of that walk has a parallel
// form whose depth is the tree's height:
//   * children: each node registers with its parent (LDS atomic), the parent orders its two
//     or three children by their largest leaf id (sibling subtrees are disjoint: no ties);
//   * bottom-up sweep (rounds: a node computes once its children have): largest leaf id,
//     Sethi-Ullman label, number of internal nodes below, stored / unstored class;
//   * top-down sweep from the root: with hi the child evaluated first (the larger label, the
//     first child on a tie) and lo the other,
//         start(hi) = start(v),  start(lo) = start(v) + internal(hi),
//         base(hi)  = base(v),   base(lo)  = base(v) + [hi is internal],
//     node v is entry start(v) + internal(v) - 1 of the post-order and its vector lives in
//     slot base(v): the lowest-free-slot rule of the sequential walk always finds the live
//     slots to be 0 .. base(v)-1 (one per ancestor that is waiting for its second child);
//   * the gradient kernels' macros: a stored node's slot number is a prefix count over node
//     ids (ballots), the entries are written by the nodes' own threads.
// Same schedule, bit for bit, as tree_setup_kernel (tests/test_tree_setup_gpu.py).
// ------------------------------------------------------------------------
constexpr int kSetupMaxThreads = 1024;  // (the launcher takes fewer for small trees)
constexpr int kSetupArrays = 16;         // LDS ints per node
constexpr int kSetupOwn = 3;             // nodes per thread at most (N <= 2 * 3 * 1024)
// What a round of a sweep reads of another node is ONE 64-bit LDS word (a dependent LDS round
// trip costs more than everything else in a round):
//   bottom-up  W: done << 63 | class << 56 | label << 48 | internal-node count << 20 | largest leaf
//   top-down   D: ready << 63 | slot << 32 | start
__device__ __forceinline__ uint64_t up_word(int cls, int label, int isz, int mxl) {
  return (1ull << 63) | ((uint64_t)cls << 56) | ((uint64_t)label << 48) | ((uint64_t)isz << 20) | (uint64_t)mxl;
}
__device__ __forceinline__ int up_cls(uint64_t w) { return (int)(w >> 56) & 3; }
__device__ __forceinline__ int up_label(uint64_t w) { return (int)(w >> 48) & 0xff; }
__device__ __forceinline__ int up_isz(uint64_t w) { return (int)(w >> 20) & 0xfffff; }
__device__ __forceinline__ int up_mxl(uint64_t w) { return (int)w & 0xfffff; }
// (the 1024-thread bound leaves 128 registers per lane and 52 values spill; an instance bounded
// at 256 threads for trees of up to 256 taxa -- 228 registers, no scratch -- measured SLOWER:
// fluA one tree 0.0928 against 0.0896 ms, x 1000 0.315 against 0.310: the spills are not on
// the rounds' dependent chains, the occupancy is)
__global__ __launch_bounds__(kSetupMaxThreads) void tree_setup_wg_kernel(TreeSetupArgs a, ModelSetupArgs ms) {
  const int tid = threadIdx.x, nthreads = blockDim.x;
  if ((int)blockIdx.x >= a.T) {
    model_setup_thread(ms, ((int)blockIdx.x - a.T) * nthreads + tid);
    return;
  }
  extern __shared__ int32_t ts_lds[];
  __shared__ int flag_bad, flag_arity, used_max, chunk_tot[64], ok_flag;
  const int t = blockIdx.x;
  const int n = a.n, N = 2 * n - 1;
  const int nodes_in = a.rooted ? N : N - 1, root_in = nodes_in - 1;
  const int32_t* par_in = a.parent_ids + (size_t)t * (nodes_in - 1);
  // (64-bit and 128-bit arrays first: alignment.  The pointers are TYPED as LDS pointers: a
  // volatile access through a generic pointer stays a flat instruction -- 105 of them in this
  // kernel until round 5, each on the dependent chain of a round)
#define TS_LDS(T) volatile __attribute__((address_space(3))) T*
  typedef __attribute__((address_space(3))) int32_t* ts_lds_base;
  const ts_lds_base ts3 = (ts_lds_base)ts_lds;
  TS_LDS(int4) kid = (TS_LDS(int4))ts3;                 // {k0, k1, k2, count}
  TS_LDS(uint64_t) W = (TS_LDS(uint64_t))(ts3 + 4 * N);  // bottom-up word
  TS_LDS(uint64_t) D = W + N;                            // top-down word
  TS_LDS(int2) cc = (TS_LDS(int2))(D + N);               // {c0, c1 | first0 << 30}
  TS_LDS(int2) up = cc + N;                              // {parent, offsets} of the reshaped tree
  TS_LDS(int32_t) par = (TS_LDS(int32_t))(up + N);
  TS_LDS(int32_t) sslot = par + N;
#undef TS_LDS
  int32_t* kid_i = ts_lds;
  SchedEntry* sched = a.sched + (size_t)t * (n - 1);
  double* ble = a.bl_eff + (size_t)t * N;

  for (int v = tid; v < N; v += nthreads) {
    par[v] = v < nodes_in - 1 ? par_in[v] : -1;
    kid_i[4 * v] = kid_i[4 * v + 1] = kid_i[4 * v + 2] = kid_i[4 * v + 3] = 0;
    W[v] = v < n ? up_word(0, 0, 0, v) : 0;
    D[v] = 0;
    sslot[v] = 0;
  }
  if (tid == 0) flag_bad = flag_arity = used_max = 0;
  __syncthreads();
  for (int v = tid; v < nodes_in - 1; v += nthreads) {
    const int p = par[v];
    if (p <= v || p >= nodes_in || p < n) flag_bad = 1;
  }
  __syncthreads();
  int status = flag_bad ? kBadParentIds : kOk;
  int macro_M = 0;  // (uniform) the tree's macro count, for the arena's slot assignment below
  if (status == kOk) {
    for (int v = tid; v < nodes_in - 1; v += nthreads) {
      const int p = par[v];
      const int k = atomicAdd(&kid_i[4 * p + 3], 1);
      if (k < 3) kid_i[4 * p + k] = v;
    }
    __syncthreads();
    for (int v = n + tid; v < nodes_in; v += nthreads) {
      const int want = (!a.rooted && v == root_in) ? 3 : 2;
      if (kid_i[4 * v + 3] != want) flag_arity = 1;
    }
    __syncthreads();
    if (flag_arity) status = a.rooted ? kNotBifurcating : kNotTrifurcatingRoot;
  }
  if (status == kOk) {
    // ---- bottom-up: largest leaf, label, internal-node count, class.  A thread keeps what
    // is fixed about its (up to kSetupOwn) nodes in registers: a round costs it one LDS round
    // trip (its children's words) and the barrier.
    int own_k0[kSetupOwn], own_k1[kSetupOwn], own_k2[kSetupOwn];
    bool own_todo[kSetupOwn];
#pragma unroll
    for (int i = 0; i < kSetupOwn; i++) {
      const int v = n + tid + i * nthreads;
      own_todo[i] = v < nodes_in;
      const int vv = own_todo[i] ? v : n;
      own_k0[i] = kid[vv].x;
      own_k1[i] = kid[vv].y;
      own_k2[i] = kid[vv].w == 3 ? kid[vv].z : kid[vv].x;
    }
    // (round 4: no barrier per level.  A node's word is ONE 64-bit LDS write carrying its own
    // done bit, so a wave simply polls its nodes' children until they are there -- the waves of
    // the workgroup run side by side and progress independently; one barrier after the sweep.
    // A level costs an LDS round trip instead of a workgroup barrier plus a flag protocol: the
    // 33 levels of a fluA tree went from ~0.5 us each to ~0.15.)
    bool pending;
    do {
      pending = false;
#pragma unroll
      for (int i = 0; i < kSetupOwn; i++) {
        if (!own_todo[i]) continue;
        const int v = n + tid + i * nthreads;
        const int k0 = own_k0[i], k1 = own_k1[i];
        const uint64_t w0 = W[k0], w1 = W[k1], w2 = W[own_k2[i]];
        if (!((w0 & w1 & w2) >> 63)) {
          pending = true;
          continue;
        }
        const int m01 = up_mxl(w0) > up_mxl(w1) ? up_mxl(w0) : up_mxl(w1);
        const int mx = m01 > up_mxl(w2) ? m01 : up_mxl(w2);
        // (label, count and class of the trifurcating root are re-made below)
        const int l0 = up_label(w0), l1 = up_label(w1);
        const bool unstored = (k0 < n || up_cls(w0) == 1) && (k1 < n || up_cls(w1) == 1);
        W[v] = up_word(unstored ? 2 : 1, l0 == l1 ? l0 + 1 : (l0 > l1 ? l0 : l1),
                       up_isz(w0) + up_isz(w1) + 1, mx);
        own_todo[i] = false;
      }
    } while (__any(pending));
    __syncthreads();
    // ---- children in order of their largest leaf id; the root re-shaped (Detrifurcate);
    // what each internal child will need from its parent in the top-down sweep
    auto link = [&](int v, int a0, int a1, uint64_t w0, uint64_t w1) {
      const bool first0 = up_label(w0) >= up_label(w1);
      cc[v].x = a0;
      cc[v].y = a1 | (first0 ? 1 << 30 : 0);
      const int hi = first0 ? a0 : a1, lo = first0 ? a1 : a0;
      const int hi_isz = first0 ? up_isz(w0) : up_isz(w1);
      if (hi >= n) {
        up[hi].x = v;
        up[hi].y = 0;
      }
      if (lo >= n) {
        up[lo].x = v;
        up[lo].y = hi_isz | (hi >= n ? 1 << 24 : 0);  // start += internal(hi), slot += [hi internal]
      }
    };
    auto join = [&](uint64_t w0, uint64_t w1, int a0, int a1, bool root) {
      const int l0 = up_label(w0), l1 = up_label(w1);
      const bool unstored = (a0 < n || up_cls(w0) == 1) && (a1 < n || up_cls(w1) == 1);
      return up_word(root ? 1 : (unstored ? 2 : 1), l0 == l1 ? l0 + 1 : (l0 > l1 ? l0 : l1),
                     up_isz(w0) + up_isz(w1) + 1, 0);
    };
    for (int v = n + tid; v < nodes_in; v += nthreads) {
      int k0 = kid[v].x, k1 = kid[v].y;
      uint64_t w0 = W[k0], w1 = W[k1];
      if (up_mxl(w0) > up_mxl(w1)) {
        const int x = k0; k0 = k1; k1 = x;
        const uint64_t y = w0; w0 = w1; w1 = y;
      }
      if (kid[v].w == 2) {
        link(v, k0, k1, w0, w1);
        if (v == N - 1) W[v] = join(w0, w1, k0, k1, true);  // (rooted: the root's class is 1)
      } else {
        int k2 = kid[v].z;
        uint64_t w2 = W[k2];
        if (up_mxl(w1) > up_mxl(w2)) {
          const int x = k1; k1 = k2; k2 = x;
          const uint64_t y = w1; w1 = w2; w2 = y;
        }
        if (up_mxl(w0) > up_mxl(w1)) {
          const int x = k0; k0 = k1; k1 = x;
          const uint64_t y = w0; w0 = w1; w1 = y;
        }
        // (k0,k1,k2) at root r  ->  r = (k1,k2), r+1 = (k0, r)
        const int r = v;
        const uint64_t wr = join(w1, w2, k1, k2, false);
        W[r] = wr;
        W[r + 1] = join(w0, wr, k0, r, true);
        link(r, k1, k2, w1, w2);
        link(r + 1, k0, r, w0, wr);
      }
    }
    if (tid == 0) D[N - 1] = 1ull << 63;  // the root: ready, slot 0, start 0
    __syncthreads();
    // ---- top-down: position in the post-order and LDS slot; the schedule entries
    int own_p[kSetupOwn], own_off[kSetupOwn];
#pragma unroll
    for (int i = 0; i < kSetupOwn; i++) {
      const int v = n + tid + i * nthreads;
      own_todo[i] = v < N;
      const int vv = own_todo[i] ? v : n;
      own_p[i] = vv == N - 1 ? vv : up[vv].x;  // (the root waits for itself: ready from the start)
      own_off[i] = vv == N - 1 ? 0 : up[vv].y;
    }
    do {
      pending = false;
#pragma unroll
      for (int i = 0; i < kSetupOwn; i++) {
        if (!own_todo[i]) continue;
        const int v = n + tid + i * nthreads;
        const uint64_t dp = D[own_p[i]];
        if (!(dp >> 63)) {
          pending = true;
          continue;
        }
        // (the polling loop only publishes the node's own word: every lane of the wave runs this
        // body whenever ANY of its nodes becomes ready, so the schedule entry -- three more LDS
        // reads, a global store, an atomic -- is written in one pass after the sweep)
        const int off = own_off[i];
        const int b = ((int)(dp >> 32) & 0xffff) + ((off >> 24) & 1);
        const int st = (int)(dp & 0xffffffffu) + (off & 0xffffff);
        if (v != N - 1) D[v] = (1ull << 63) | ((uint64_t)b << 32) | (uint64_t)st;
        own_p[i] = b;      // (the parent's id is done with: the node's slot and start instead)
        own_off[i] = st;
        own_todo[i] = false;
      }
    } while (__any(pending));
    {
      int slots_used = 0;
#pragma unroll
      for (int i = 0; i < kSetupOwn; i++) {
        const int v = n + tid + i * nthreads;
        if (v >= N) continue;
        const int b = own_p[i], st = own_off[i];
        const int a0 = cc[v].x, a1f = cc[v].y, a1 = a1f & 0xfffff;
        const bool first0 = (a1f >> 30) & 1;
        const int hi = first0 ? a0 : a1;
        const int blo = b + (hi >= n ? 1 : 0);
        const int s0 = a0 < n ? 0 : (a0 == hi ? b : blo), s1 = a1 < n ? 0 : (a1 == hi ? b : blo);
        sched[st + up_isz(W[v]) - 1] = {v, a0, a1, b | (s0 << 8) | (s1 << 16) | ((a0 < n ? 1 : 0) << 24) |
                                                      ((a1 < n ? 1 : 0) << 25)};
        slots_used = slots_used > b + 1 ? slots_used : b + 1;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int x = __shfl_xor(slots_used, o, 64);
        slots_used = slots_used > x ? slots_used : x;
      }
      if ((tid & 63) == 0) atomicMax(&used_max, slots_used);
    }
    __syncthreads();
    if (used_max > a.max_slots) status = kTooManySlots;
    // ---- schedule of the on-chip gradient kernel (see tree_setup_kernel)
    if (a.macros) {
      const int lane = tid & 63, wave = tid >> 6, waves = nthreads / 64;
      const int chunks = (N - 1 - n + 63) / 64;  // nodes n .. N-2 in chunks of 64
      for (int c = wave; c < chunks; c += waves) {
        const int v = n + c * 64 + lane;
        const bool st = v < N - 1 && up_cls(W[v < N ? v : N - 1]) == 1;
        const unsigned long long m = __ballot(st);
        if (st) sslot[v] = __popcll(m & ((1ull << lane) - 1));
        if (lane == 0) chunk_tot[c] = __popcll(m);
      }
      __syncthreads();
      if (tid == 0) {
        int run = 0;
        for (int c = 0; c < chunks; c++) {
          const int x = chunk_tot[c];
          chunk_tot[c] = run;
          run += x;
        }
        sslot[N - 1] = run;  // (the root's macro index; its qslot is -1)
      }
      __syncthreads();
      const int stored = sslot[N - 1];
      for (int v = n + tid; v < N - 1; v += nthreads)
        if (up_cls(W[v]) == 1) sslot[v] += chunk_tot[(v - n) >> 6];
      __syncthreads();
      MacroEntry* mac = a.macros + (size_t)t * macro_stride(n);
      for (int v = n + tid; v < N; v += nthreads) {
        if (up_cls(W[v]) != 1) continue;
        const bool root = v == N - 1;
        MacroEntry me;
        int kind[2];
        me.node = v;
        me.pad = 0;
        me.qslot = root ? -1 : sslot[v];
        for (int j = 0; j < 2; j++) {
          const int ch = j ? (cc[v].y & 0xfffff) : cc[v].x;
          const int ccls = ch < n ? 0 : up_cls(W[ch]);
          me.child[j] = ch;
          kind[j] = ccls;
          me.cslot[j] = ccls == 1 ? sslot[ch] : 0;
          const bool expand = ccls == 2;
          const int ga = expand ? cc[ch].x : 0, gb = expand ? (cc[ch].y & 0xfffff) : 0;
          me.grand[2 * j] = ga;
          me.grand[2 * j + 1] = gb;
          me.gslot[2 * j] = ga >= n ? sslot[ga] : 0;
          me.gslot[2 * j + 1] = gb >= n ? sslot[gb] : 0;
        }
        me.shape = macro_shape(kind[0], kind[1], root, me.child, me.grand, n);
        if (sslot[v] < max_macros(n)) mac[sslot[v]] = me;
      }
      macro_M = stored + 1 < max_macros(n) ? stored + 1 : max_macros(n);
      if (tid == 0) a.macro_count[t] = macro_M;
      if (stored > max_stored(n)) status = kTooManySlots;
    }
  }
  if (tid == 0) {
    if (status != kOk) set_status(a.status, status, t);
    ok_flag = status == kOk || status == kTooManySlots;
  }
  __syncthreads();
  if (!ok_flag) {
    macro_M = 0;
    if (tid == 0 && a.macro_count) a.macro_count[t] = 0;
    for (int i = tid; i < n - 1; i += nthreads) sched[i] = {n + i, 0, 1, 0};
    for (int v = tid; v < N; v += nthreads) ble[v] = 0.0;
  } else if (!a.rooted) {
    const double* bl = a.bl + (size_t)t * (N - 1);
    for (int v = tid; v < N; v += nthreads) ble[v] = v < N - 2 ? bl[v] : 0.0;
  } else {
    const double* bl = a.bl + (size_t)t * N;
    const double* rates = a.rates ? a.rates + (size_t)t * (N - 1) : nullptr;
    for (int v = tid; v < N; v += nthreads)
      ble[v] = (rates && v < N - 1) ? bl[v] * rates[v] : bl[v];
  }
  // Arena calls (round 6): the slot assignment of the arena variant -- macro_slots_wg_kernel's
  // work, the same workgroup-per-tree mapping -- on the macro list this workgroup has just
  // written (visible to it after the barrier), in the LDS the tree arrays no longer need.
  if (a.arena_macros) {
    __syncthreads();
    macro_slots_wg_body(ts_lds, a.macros, a.arena_macros, macro_M, n, t, a.slot_need, a.arena_sure, a.status);
  }
}


// ------------------------------------------------------------------------
// Tree setup for N <= 256 nodes: one wave per tree, every per-node array in vector registers
// (mi_phylo_setup_device.h).
// ------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(64) void tree_setup_small_kernel(TreeSetupArgs a, ModelSetupArgs ms) {
  if ((int)blockIdx.x >= a.T) {
    model_setup_thread(ms, ((int)blockIdx.x - a.T) * 64 + (int)threadIdx.x);
    return;
  }
  __shared__ int tree_lds[kSmallTreeLdsInts * NB];
  SmallTree<NB> tree;
  small_tree_build<NB>(a, blockIdx.x, threadIdx.x, tree, tree_lds);
  small_tree_store<NB>(a, blockIdx.x, threadIdx.x, tree);
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
  // (the launch covers evaluations [eval_base, eval_base + E): indices below are global)
  const long per = (long)a.K * (a.N - 1);
  const long first = (long)blockIdx.x * kTransitionBlock + (long)a.eval_base * per;
  const long idx = first + threadIdx.x;
  const long total = (long)(a.eval_base + a.E) * per;
  {
    // whole workgroups inside a range of evaluations that nobody walks have nothing to do
    const long last = (first + kTransitionBlock - 1 < total ? first + kTransitionBlock - 1 : total - 1);
    if (first / per >= a.ev_skip_begin && last / per < a.ev_skip_end) return;
  }
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
        stage[threadIdx.x * 17 + i * 4 + j] = sum > 0 ? sum : 0;  // BEAGLE clamps negative probabilities to 0
      }
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
}

}  // namespace

// ------------------------------------------------------------------------
// Launch wrappers
// ------------------------------------------------------------------------
bool launch_setup(const TreeSetupArgs& a_in, const ModelSetupArgs& ms, hipStream_t s) {
  TreeSetupArgs a = a_in;
  // MI_PHYLO_MACRO_SLOTS=own|seq: the arena's slot assignment stays a launch of its own (A/B, tests)
  const bool fold_slots = getenv("MI_PHYLO_MACRO_SLOTS") == nullptr;  // (read per call)
  a.arena_sure = gradient_arena_slots_sure(a.n);
  const size_t lds = sizeof(int32_t) * 13 * (size_t)(2 * a.n - 1);
  a.use_lds = lds <= 64 * 1024;
  // Three kernels build the same schedule (tests/test_cpp_adapter_gpu.py compares them bit for
  // bit).  Measured, 1000 trees per launch: DS1 (53 nodes) register-array kernel 13 us /
  // workgroup kernel 20 us; fluA (137 nodes) 63 / 41 us; one 512-taxon tree: sequential
  // 840 us / workgroup 38 us.  So: register arrays up to 64 nodes, a workgroup per tree above,
  // the sequential kernel only for trees whose arrays exceed the LDS.
  // MI_PHYLO_TREE_SETUP=small|wg|lds forces one of them where it applies (testing).
  static const std::string forced = getenv("MI_PHYLO_TREE_SETUP") ? getenv("MI_PHYLO_TREE_SETUP") : "";
  const int N = 2 * a.n - 1;
  const size_t wg_lds = sizeof(int32_t) * kSetupArrays * (size_t)N;
  const bool small_ok = a.n >= 3 && N <= 256, wg_ok = a.n >= 3 && wg_lds <= 160 * 1024 - 1024;
  const bool use_small = small_ok && (forced == "small" || (forced.empty() && N <= 64));
  const bool use_wg = !use_small && wg_ok && forced != "lds" && !(forced == "small" && small_ok);
  if (use_small) {
    // T workgroups of trees, then the model instances, 64 per workgroup
    const dim3 grid(a.T + (ms.T * ms.models_per_tree + 63) / 64), block(64);
    if (N <= 64) hipLaunchKernelGGL(tree_setup_small_kernel<1>, grid, block, 0, s, a, ms);
    else if (N <= 128) hipLaunchKernelGGL(tree_setup_small_kernel<2>, grid, block, 0, s, a, ms);
    else if (N <= 192) hipLaunchKernelGGL(tree_setup_small_kernel<3>, grid, block, 0, s, a, ms);
    else hipLaunchKernelGGL(tree_setup_small_kernel<4>, grid, block, 0, s, a, ms);
    return false;
  }
  if (use_wg) {
    // (a thread per internal node, at most kSetupOwn of them per thread: the LDS bound keeps
    // N below 2 600)
    const int threads = a.n >= kSetupMaxThreads ? kSetupMaxThreads : (a.n + 63) / 64 * 64;
    const dim3 wgrid(a.T + (ms.T * ms.models_per_tree + threads - 1) / threads);
    // (the slot assignment takes two macros per thread at most, and its own LDS in place of the
    // tree arrays)
    const size_t slots_lds = macro_slots_wg_lds_bytes(a.n);
    const bool fold = a.arena_macros && a.macros && fold_slots && max_macros(a.n) <= 2 * threads &&
                      slots_lds <= 160 * 1024 - 1024;
    if (!fold) a.arena_macros = nullptr;
    const size_t lds_bytes = fold ? std::max(wg_lds, slots_lds) : wg_lds;
    allow_large_lds(reinterpret_cast<const void*>(tree_setup_wg_kernel), lds_bytes);
    hipLaunchKernelGGL(tree_setup_wg_kernel, wgrid, dim3(threads), lds_bytes, s, a, ms);
    return fold;
  }
  a.arena_macros = nullptr;
  const dim3 grid(a.T + (ms.T * ms.models_per_tree + 63) / 64), block(64);
  hipLaunchKernelGGL(tree_setup_kernel, grid, block, a.use_lds ? lds : 0, s, a, ms);
  return false;
}
namespace {
__global__ void weibull_table_kernel(int K, double* table) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K) return;
  const double quantile = (2.0 * i + 1.0) / (2.0 * K);  // (the expressions of weibull_category)
  const double x = -log(1.0 - quantile);
  table[2 * i] = x;
  table[2 * i + 1] = log(x);
}
}  // namespace
void launch_weibull_table(int K, double* table, hipStream_t s) {
  hipLaunchKernelGGL(weibull_table_kernel, dim3(1), dim3(64), 0, s, K, table);
}
bool launch_tree_setup(const TreeSetupArgs& a, hipStream_t s) {  // trees only
  ModelSetupArgs none{};
  return launch_setup(a, none, s);
}
void launch_transition(const TransitionArgs& a, hipStream_t s) {
  const long total = (long)a.E * (a.N - 1) * a.K;
  hipLaunchKernelGGL(transition_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     a);
}
}  // namespace miphylo
