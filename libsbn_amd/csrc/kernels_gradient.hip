// Gradient kernel with partial vectors streamed through HBM (real-valued tip partials, trees no
// on-chip walk fits), and what the on-chip walks (kernels_walk.hip, kernels_walk3.hip) share:
// tile counts, the arena variant's slot assignment (macro_slots kernels), device queries.
// (The first-generation matrix-core walk, gradient_mfma_kernel, lived here until round 6: the
// second generation covers every call it took, the third the shapes it was still chosen for.)
// (gfx950 / CDNA4, wave64; see DESIGN.md for the mapping and what bounds each kernel.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "mi_phylo_device_utils.h"
#include "mi_phylo_kernels.h"
#include "mi_phylo_macro_slots_device.h"

namespace miphylo {

namespace {
using namespace dev;

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

}  // namespace

// ------------------------------------------------------------------------
// Launch wrappers
// ------------------------------------------------------------------------
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
int gradient_mfma_width(int n, bool subst) {
  return max_macros(n) * kMacroPositions * 2 + (subst ? kSubstExtra : 0);
}
int gradient_mfma_groups(int K) { return K <= 4 ? 1 : (K + 3) / 4; }
int gradient_mfma_tiles(int P, int K, int regs) {
  const int per_wave = (regs > 0 ? regs : kLlR) * (16 / (K == 1 ? 1 : (K == 2 ? 2 : 4)));
  return (P + per_wave - 1) / per_wave;
}
// Tile width of the look-up walk for an engine whose batches take the arena: four registers per
// vector where that leaves enough fewer tiles, else the default.  Measured both ways on 29 + 17
// shapes (profiles/r06_wide_tiles.txt; the second table after the tip codes were pre-tiled): a
// wide tile costs 1.14-1.2 default tiles (a visit's fixed costs are spread over a third more
// columns) -- 36 taxa x 1812 patterns x 4 categories 151 -> 114 tiles, 2.72 -> 2.34 ms per 1000
// trees -- unless its LDS footprint costs a wave per CU (64 taxa and more: eight 2-KB slots and
// the tip words pass 20 KB) AND the tree has few tiles: then 1.3-1.45 (fluA's 69 x 238 x 1, five
// tiles against four: 0.245 -> 0.260).  So: wide where tiles x cost falls, cost 1.16 or 1.33, and
// never below five default tiles.  The width is the ENGINE's (all its look-up calls take it):
// sums over patterns are formed tile by tile, and a tree's outputs must not depend on the size
// of the batch it came in.  MI_PHYLO_WALK_TILE_REGS=3|4 forces a width (read at engine creation).
int gradient_walk_tile_regs(int n, int P, int K) {
  if (kLlR >= 4) return kLlR;
  if (const char* env = getenv("MI_PHYLO_WALK_TILE_REGS")) {
    const int r = atoi(env);
    if (r == 4 || r == kLlR) return r;
  }
  const int t3 = gradient_mfma_tiles(P, K, kLlR), t4 = gradient_mfma_tiles(P, K, 4);
  auto waves = [&](int regs) {
    const size_t lds = gradient_walk_lds_bytes_for(n, K, false, false, gradient_arena_slots_usual(n), regs);
    return std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1));
  };
  const bool cheap = waves(4) == waves(kLlR) || t4 >= 12;
  const bool wide = t3 >= 5 && (cheap ? 116 : 133) * t4 < 100 * t3;
  return wide ? 4 : kLlR;
}
// ---- arena variant: LDS slots of the two launches ----
// Live vectors of the macro walk: each is either live in the node-level Sethi-Ullman walk
// (at most floor(log2 n) at a time) or one of the two children of an unstored node that
// is, so 2 floor(log2 n) always suffice (second launch); floor(log2 n) + 2 is what trees
// need in practice (first launch: random, ladder and balanced trees up to 200 taxa never
// exceeded floor(log2 n) + 1 in simulation).
static int floor_log2(int n) {
  int lg = 0;
  while ((2 << lg) <= n) lg++;
  return lg;
}
int gradient_arena_slots_sure(int n) { return std::max(1, std::min(max_stored(n), 2 * floor_log2(n))); }
int gradient_arena_slots_usual(int n) { return std::min(gradient_arena_slots_sure(n), floor_log2(n) + 2); }
size_t gradient_arena_bytes_per_eval(int n, int P, int K) {
  // (the larger of the two tile widths: the choice is made per call)
  const size_t cols = std::max((size_t)gradient_mfma_tiles(P, K) * kLlR, (size_t)gradient_mfma_tiles(P, K, 4) * 4);
  return cols * gradient_mfma_groups(K) * max_stored(n) * kTile * sizeof(double);
}
// One thread per tree: the macro schedule re-ordered and given reusable LDS slots.
// tree_setup lists the macros by node id -- a post-order, but one that can keep many
// vectors alive.  Here the macro tree (a macro's inputs are the stored nodes among its
// children / grandchildren, up to four) is walked Sethi-Ullman style, the input needing
// the most slots first, and the stored vectors' LDS slots are an interval colouring in
// that order (free the inputs' slots, take the lowest free one for the node) and replace
// the node-unique numbers in the slot fields; the arena index of a stored node's vector
// goes to its `pad` field.  need[t] = slots the tree uses.
// (a wave per tree: the entries are staged in LDS by all lanes, lane 0 does the walk)
__global__ __launch_bounds__(kTile) void macro_slots_kernel(const MacroEntry* macros_in,
                                                            MacroEntry* macros_out,
                                                            const int32_t* macro_count, int n, int T,
                                                            int32_t* need, int sure, int32_t* status) {
  extern __shared__ int32_t slot_scratch[];
  const int t = blockIdx.x, lane = threadIdx.x;
  const int Mmax = max_macros(n), S = max_stored(n);
  MacroEntry* ent = reinterpret_cast<MacroEntry*>(slot_scratch);
  int32_t* mac_of = slot_scratch + (size_t)Mmax * (sizeof(MacroEntry) / 4);  // stored id -> macro, later -> LDS slot
  int32_t* label = mac_of + S;
  int32_t* order = label + Mmax;
  int32_t* stack = order + Mmax;
  const int M = macro_count[t];
  if (M <= 0) {
    if (lane == 0) need[t] = 0;
    return;
  }
  {
    const int32_t* src = reinterpret_cast<const int32_t*>(macros_in + (size_t)t * macro_stride(n));
    for (int i = lane; i < M * 16; i += kTile) slot_scratch[i] = src[i];
  }
  __syncthreads();
  if (lane == 0) {
    // inputs of a macro as macro indices, sorted by label, largest first
    auto sorted_inputs = [&](MacroEntry& e, int* idx) {
      const MacroInputs mi = macro_inputs(e);
      for (int i = 0; i < mi.count; i++) idx[i] = mac_of[macro_field(e, mi.field[i])];
      for (int i = 1; i < mi.count; i++)
        for (int k = i; k > 0 && label[idx[k]] > label[idx[k - 1]]; k--) {
          const int x = idx[k];
          idx[k] = idx[k - 1];
          idx[k - 1] = x;
        }
      return mi.count;
    };
    // 1. labels, bottom-up (the given order is a post-order)
    for (int m = 0; m < M; m++) {
      int idx[4];
      const int k = sorted_inputs(ent[m], idx);
      const bool root = (ent[m].shape & 16) != 0;
      int l = root ? k : (k > 1 ? k : 1);
      for (int i = 0; i < k; i++) l = l > label[idx[i]] + i ? l : label[idx[i]] + i;
      label[m] = l;
      if (!root) mac_of[ent[m].qslot] = m;
    }
    // 2. post-order from the root (the last macro), largest label first
    int top = 0, emitted = 0;
    stack[top++] = (M - 1) << 1;
    while (top) {
      const int item = stack[--top];
      const int m = item >> 1;
      if (item & 1) {
        order[emitted++] = m;
        continue;
      }
      stack[top++] = item | 1;
      int idx[4];
      const int k = sorted_inputs(ent[m], idx);
      for (int i = k - 1; i >= 0; i--) stack[top++] = idx[i] << 1;  // idx[0] is popped first
    }
    // 2b. arena indices: a macro's stored inputs are numbered consecutively in position
    // order from the macro's base (upper half of its shape word); each input is told where
    // its vector goes (pad field)
    int next_index = 0;
    for (int o = 0; o < M; o++) {
      MacroEntry& e = ent[order[o]];
      const MacroInputs mi = macro_inputs(e);
      e.shape |= next_index << 16;
      for (int i = 0; i < mi.count; i++) ent[mac_of[macro_field(e, mi.field[i])]].pad = next_index++;
    }
    // 3. slots in that order (packed in place); mac_of becomes stored id -> LDS slot
    uint64_t free_mask = ~0ull;
    int used = 0;
    for (int o = 0; o < M; o++) {
      MacroEntry& e = ent[order[o]];
      const MacroInputs mi = macro_inputs(e);
      for (int i = 0; i < mi.count; i++) {
        int32_t& f = macro_field(e, mi.field[i]);
        const int s = mac_of[f];
        free_mask |= 1ull << s;
        f = s;
      }
      if (!(e.shape & 16)) {
        const int s = __ffsll((unsigned long long)free_mask) - 1;
        free_mask &= ~(1ull << s);
        mac_of[e.qslot] = s;
        e.qslot = s;
        if (s + 1 > used) used = s + 1;
      }
    }
    need[t] = used;
    if (used > sure || emitted != M) set_status(status, kTooManySlots, t);
  }
  __syncthreads();
  {
    int32_t* dst = reinterpret_cast<int32_t*>(macros_out + (size_t)t * macro_stride(n));
    for (int i = lane; i < M * 16; i += kTile) dst[i] = slot_scratch[order[i >> 4] * 16 + (i & 15)];
  }
}
// ------------------------------------------------------------------------
// The same, a workgroup per tree and a thread per macro, in O(height of the macro tree)
// rounds (the sequential walk above: about 1 700 cycles per macro and pass, 84 us for the 35
// macros of a 69-taxon tree -- a third of a one-tree gradient call).  As in
// tree_setup_wg_kernel every quantity has a closed form:
//   * labels bottom-up (a macro computes once its inputs have), the inputs ordered by label,
//     largest first, ties in field order (what the stable insertion sort above does);
//   * the post-order top-down: input i of a macro starts where input i-1 ended,
//         start(in_0) = start(m),  start(in_i) = start(in_{i-1}) + macros(in_{i-1}),
//     and macro m is entry start(m) + macros(m) - 1;
//   * LDS slots: slot(in_i) = slot(m) + i -- the lowest-free-slot rule always finds slots
//     0 .. slot(m)-1 taken by the ancestors that are waiting and slot(m) .. slot(m)+i-1 by the
//     inputs already evaluated;
//   * arena indices: a prefix sum of the input counts over the post-order.
// One 64-bit LDS word per macro and sweep carries the state flag with the values; one barrier
// per round (three rotating flags).  Bit-identical to the sequential kernel
// (tests/test_cpp_adapter_gpu.py); MI_PHYLO_MACRO_SLOTS=seq selects that one.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void macro_slots_wg_kernel(const MacroEntry* macros_in,
                                                              MacroEntry* macros_out,
                                                              const int32_t* macro_count, int n, int T,
                                                              int32_t* need, int sure, int32_t* status) {
  extern __shared__ int32_t slot_scratch[];
  macro_slots_wg_body(slot_scratch, macros_in, macros_out, macro_count[blockIdx.x], n, blockIdx.x, need, sure,
                      status);
}
void launch_macro_slots(const MacroEntry* macros_in, MacroEntry* macros_out,
                        const int32_t* macro_count, int n, int T, int32_t* need, int32_t* status,
                        hipStream_t s) {
  static const bool seq = getenv("MI_PHYLO_MACRO_SLOTS") && std::string(getenv("MI_PHYLO_MACRO_SLOTS")) == "seq";
  const size_t Mmax = max_macros(n), S = max_stored(n);
  const size_t wg_lds = macro_slots_wg_lds_bytes(n);
  (void)S;
  if (!seq && wg_lds <= 160 * 1024 - 512 && Mmax <= 2 * 1024) {  // (two macros per thread at most)
    const int threads = Mmax >= 1024 ? 1024 : (int)((Mmax + 63) / 64 * 64);
    allow_large_lds(reinterpret_cast<const void*>(macro_slots_wg_kernel), wg_lds);
    hipLaunchKernelGGL(macro_slots_wg_kernel, dim3(T), dim3(threads), wg_lds, s, macros_in, macros_out,
                       macro_count, n, T, need, gradient_arena_slots_sure(n), status);
    return;
  }
  const size_t lds = sizeof(MacroEntry) * (size_t)max_macros(n) +
                     sizeof(int32_t) * ((size_t)max_stored(n) + 3 * (size_t)max_macros(n));
  allow_large_lds(reinterpret_cast<const void*>(macro_slots_kernel), lds);
  hipLaunchKernelGGL(macro_slots_kernel, dim3(T), dim3(kTile), lds, s, macros_in, macros_out,
                     macro_count, n, T, need, gradient_arena_slots_sure(n), status);
}

// Do `waves` single-wave workgroups of `lds` bytes each all fit the device at once?
// compute units of the current device (asked once per device)
int device_compute_units() {
  static int cached[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    cached[dev] = cus;
  }
  return cached[dev];
}
bool arena_single_launch(size_t lds, size_t waves) {
  const int cus = device_compute_units();
  const size_t per_cu = std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1));
  return waves <= (size_t)cus * per_cu;
}
const char* gradient_kernel_name() { return "gradient_hbm_kernel"; }

}  // namespace miphylo
