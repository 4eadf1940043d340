// Gradient kernels: matrix-core on-chip (default) and HBM-streamed.
// (gfx950 / CDNA4, wave64; see DESIGN.md for the mapping and what bounds each kernel.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "mi_phylo_device_utils.h"
#include "mi_phylo_kernels.h"

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
// ARENA (trees whose stored vectors would crowd the LDS): the post-order vectors of the
// stored nodes go to a per-wave arena in HBM as they are produced and are read back from
// there by the pre-order walk; LDS then only holds the vectors that are live at one time
// -- post-order vectors until their consumer's macro, pre-order vectors from the parent's
// macro to their own: the same intervals walked backwards, so one interval colouring
// (macro_slots_kernel) serves both walks.
template <int R, int DBG = 0, bool RESCALE = false, bool SUBST = false, bool ARENA = false>
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
  const MacroEntry* __restrict__ macros = a.macros + (size_t)t * macro_stride(n);
  const int M = __builtin_amdgcn_readfirstlane(a.macro_count[t]);
  if (ARENA) {
    // this launch takes the trees whose live vectors fit its LDS slots (and not the
    // previous, tighter launch's)
    const int need = __builtin_amdgcn_readfirstlane(a.slot_need[t]);
    if (need <= a.lds_lo || need > a.lds_slots) return;
  }
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
  // (in LDS the substitution extras are only the four root sums; H goes out from registers)
  const int lds_width = max_macros(n) * kMacroPositions * 2 + (SUBST ? 4 : 0);
  double* plv = gacc + lds_width;
  // RESCALE: per (slot, pattern) power-of-two exponent taken out of a stored vector
  // (16 bits hold any FP64 exponent; 32-bit entries pushed DS1's footprint over the
  // 20 KB that 8 waves per CU leave)
  int16_t* exps = reinterpret_cast<int16_t*>(
      plv + (size_t)(ARENA ? a.lds_slots : max_stored(n)) * R * kTile);
  // ARENA: this wave's [stored node][r][lane] block of post-order vectors in HBM
  char* const arena =
      ARENA ? reinterpret_cast<char*>(a.plv + ((size_t)te.eval * gridDim.x + te.tile) *
                                                   max_stored(n) * R * kTile)
            : nullptr;
  // (ARENA: the slot fields of the re-ordered schedule are the reusable LDS slots, and a
  // stored node is identified by its arena index `dst`)
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
  auto arena_ptr = [&](int id) {
    asm volatile("" : "+v"(id));
    return reinterpret_cast<double*>(arena + (__umul24((unsigned)id, (unsigned)(R * kTile * 8)) + lane8));
  };
  auto store_arena = [&](int id, const V& x) {
    double* c = arena_ptr(id);
#pragma unroll
    for (int r = 0; r < R; r++) c[r * kTile] = x.v[r];
  };
  // ARENA, pre-order walk: the post-order vectors of a macro's stored inputs sit in the
  // arena at consecutive indices from the macro's base (upper half of its shape word, a
  // scalar), in position order -- so they can be requested a whole macro ahead, before
  // the macro's slot fields are even loaded.
  struct PreL {
    V x0, y0, x1, y1;
  };
  auto arena_at = [&](int k) {  // k is wave-uniform (scalar)
    V x;
    const double* c =
        reinterpret_cast<const double*>(arena + ((unsigned)k * (unsigned)(R * kTile * 8) + lane8));
#pragma unroll
    for (int r = 0; r < R; r++) x.v[r] = c[r * kTile];
    return x;
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
    int dst;  // ARENA: where this node's post-order vector goes in the arena
  };
  auto load_ids = [&](int m) {
    const int4* p = reinterpret_cast<const int4*>(macros + m);
    const int4 x = p[0], y = p[1];
    return Ids{x.x, x.y, x.z, x.w, y.x, y.y, y.z};
  };
  auto load_slots = [&](int m) {
    const int4* p = reinterpret_cast<const int4*>(macros + m) + 2;
    const int4 x = p[0], y = p[1];
    return Slots{x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
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
  auto is_tip_early = [](int s, int j) { return ((s >> (8 + j)) & 1) != 0; };
  auto prefetch_L = [&](int sh) {
    PreL p;
    int k = (int)((unsigned)sh >> 16);
    if ((sh & 3) == 2) {
      if (!is_tip_early(sh, 2)) p.x0 = arena_at(k++);
      if (!is_tip_early(sh, 3)) p.y0 = arena_at(k++);
    } else if ((sh & 3) == 1) {
      p.x0 = arena_at(k++);
    }
    if (((sh >> 2) & 3) == 2) {
      if (!is_tip_early(sh, 4)) p.x1 = arena_at(k++);
      if (!is_tip_early(sh, 5)) p.y1 = arena_at(k++);
    } else if (((sh >> 2) & 3) == 1) {
      p.x1 = arena_at(k++);
    }
    return p;
  };
  auto load_ops = [&](int sh, const Slots& sl, const MacroMats& cm, bool pre, const PreL& pl) {
    Ops o;
    const bool from_arena = ARENA && pre;
    if (pre) {
      if (is_root(sh)) {
#pragma unroll
        for (int r = 0; r < R; r++) o.q.v[r] = qroot[r];
      } else {
        o.q = load_slot(sl.q);
        if (RESCALE) {
#pragma unroll
          for (int r = 0; r < R; r++)
            o.q.v[r] = ldexp(o.q.v[r], -(int)exps[__umul24((unsigned)(ARENA ? sl.dst : sl.q), (unsigned)TP) +
                                              (unsigned)(r * ppr + col)]);
        }
      }
    }
    if (kind0(sh) == 2) {
      o.x0 = is_tip(sh, 2) ? tip_vector(cm.tw[2]) : (from_arena ? pl.x0 : load_slot(sl.gs0));
      o.y0 = is_tip(sh, 3) ? tip_vector(cm.tw[3]) : (from_arena ? pl.y0 : load_slot(sl.gs1));
    } else {
      o.x0 = is_tip(sh, 0) ? tip_vector(cm.tw[0]) : (from_arena ? pl.x0 : load_slot(sl.cs0));
    }
    if (kind1(sh) == 2) {
      o.x1 = is_tip(sh, 4) ? tip_vector(cm.tw[4]) : (from_arena ? pl.x1 : load_slot(sl.gs2));
      o.y1 = is_tip(sh, 5) ? tip_vector(cm.tw[5]) : (from_arena ? pl.y1 : load_slot(sl.gs3));
    } else {
      o.x1 = is_tip(sh, 1) ? tip_vector(cm.tw[1]) : (from_arena ? pl.x1 : load_slot(sl.cs1));
    }
    return o;
  };

  V pend_L;  // ARENA: the last stored vector, on its way to the arena
  int pend_dst = 0;
  bool pend = false;
  auto flush_arena = [&]() {
    // after the fetches: loads issued behind a store cannot be consumed before the store
    // has completed (one in-order counter), so the store goes where the loads behind it
    // are not needed for a whole macro
    if (ARENA && pend) {
      store_arena(pend_dst, pend_L);
      pend = false;
    }
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
          // (the exponent of each category's sum over the states comes from one product
          // with a ones matrix; the largest over the categories is the pattern's scale)
          const double colsum = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, Lv.v[r], 0.0, 0, 0, 0);
          int ex = colsum > 0.0 ? __builtin_amdgcn_frexp_exp(colsum) : -4096;
          if (Kp >= 2) ex = max(ex, __shfl_xor(ex, 4, 64));
          if (Kp >= 4) ex = max(ex, __shfl_xor(ex, 8, 64));
          ex = ex == -4096 ? 0 : ex;
          Lv.v[r] = ldexp(Lv.v[r], -ex);
          esum[r] += ex;
          exps[__umul24((unsigned)(ARENA ? sl.dst : sl.q), (unsigned)TP) + (unsigned)(r * ppr + col)] = (int16_t)ex;
        }
      }
      store_slot(sl.q, Lv);
      if (ARENA) {  // stored to the arena after the next macro's fetches have been issued
        pend_L = Lv;
        pend_dst = sl.dst;
        pend = true;
      }
    } else {
      // root: site likelihood per pattern, log-likelihood partial, derivative weights
      double sitev[R];
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
        sitev[r] = v;
      }
      // every lane of a pattern holds its site likelihood: the lane with state index hi
      // does the division and the logarithm of register r = hi only, and the quotients
      // go back to the pattern's other lanes with one cross-lane read per register
      static_assert(R <= 4, "one register per state index");
      double sv = sitev[0], wv = pw[0];
      int pv = pat[0], ev = esum[0];
#pragma unroll
      for (int j = 1; j < R; j++) {
        sv = hi == j ? sitev[j] : sv;
        wv = hi == j ? pw[j] : wv;
        pv = hi == j ? pat[j] : pv;
        ev = hi == j ? esum[j] : ev;
      }
      const double quot = wv / sv;  // pw = 0 for padding patterns
#pragma unroll
      for (int r = 0; r < R; r++) qroot[r] = pi_l * cw_l * __shfl(quot, (r << 4) | (lane & 15), 64);
      double ll = 0.0;
      if (hi < R && (b % Kp) == 0 && pv < a.P)
        ll = wv * (RESCALE ? log(sv) + ev * 0.69314718055994530942 : log(sv));
      ll = wave_sum(ll);
      if (lane == 0 && groups == 1) a.ll_part[(size_t)e * a.ll_tiles + te.tile] = ll;
      if (SUBST) {
        // d logL / d pi_c through the root: sum_p w_p sum_k cw_k L_root[c] / site_p;
        // this lane's state is c = hi, the 16 lanes of a row hold (category, pattern)
        double z = 0;
#pragma unroll
        for (int r = 0; r < R; r++) z = fma(qroot[r], Lv.v[r], z);  // qroot = pi cw w / site
        z = z / pi_l;
        z = row_shr_add<8>(z);
        z = row_shr_add<4>(z);
        z = row_shr_add<2>(z);
        z = row_shr_add<1>(z);
        if ((lane & 15) == 15) gacc[lds_width - 4 + hi] = z;
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
      const Ops oa = load_ops(sha, sa, ma, false, PreL{});
      shb = __builtin_amdgcn_readfirstlane(ib.shape);
      mb = fetch_mats(ib, false);
      ia = load_ids(m + 2 < M ? m + 2 : M - 1);
      sb = load_slots(m + 1 < M ? m + 1 : M - 1);
      flush_arena();
      post_step(sha, sa, ma, oa);
      if (m + 1 < M) {
        const Ops ob = load_ops(shb, sb, mb, false, PreL{});
        sha = __builtin_amdgcn_readfirstlane(ia.shape);
        ma = fetch_mats(ia, false);
        ib = load_ids(m + 3 < M ? m + 3 : M - 1);
        sa = load_slots(m + 2 < M ? m + 2 : M - 1);
        flush_arena();
        post_step(shb, sb, mb, ob);
      }
    }
    flush_arena();
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
  // block transpose (lane (hi, lo) <-> (lo, hi)) on the matrix cores: a register used as
  // the A operand is read transposed, so A = x against the identity returns x^T in the
  // ordinary layout -- one product instead of two ds_bpermute and their wait
  const double ident = hi == lo ? 1.0 : 0.0;
  auto blockT = [&](double x) { return __builtin_amdgcn_mfma_f64_4x4x4f64(x, ident, 0.0, 0, 0, 0); };
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
    Ht = fma(R2, phi, Ht);  // Phi is symmetric (explicit fma: both walk generations round alike)
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
    PreL la, lb;  // ARENA: stored inputs of the macro after the current one, in flight
    if (ARENA) la = prefetch_L(sha);
    for (int m = M - 1; m >= 0; m -= 2) {
      const Ops oa = load_ops(sha, sa, ma, true, la);
      shb = __builtin_amdgcn_readfirstlane(ib.shape);
      mb = fetch_mats(ib, true);
      if (ARENA) lb = prefetch_L(shb);
      ia = load_ids(m >= 2 ? m - 2 : 0);
      sb = load_slots(m >= 1 ? m - 1 : 0);
      pre_step(sha, sa, ma, oa, m);
      if (m >= 1) {
        const Ops ob = load_ops(shb, sb, mb, true, lb);
        sha = __builtin_amdgcn_readfirstlane(ia.shape);
        ma = fetch_mats(ia, true);
        if (ARENA) la = prefetch_L(sha);
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
    if (lane < 4) gout[gwidth - 4 + lane] = gacc[lds_width - 4 + lane];
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
// LDS bytes of a wave that keeps `slots` vectors (tip masks, edge sums, vectors, exponents)
static size_t gradient_mfma_lds_bytes_for(int n, int K, bool rescale, bool subst, int slots) {
  const int kp = K == 1 ? 1 : (K == 2 ? 2 : 4);
  const size_t tip_bytes = (((size_t)n * 4 * (16 / kp) + 7) / 8) * 8;
  size_t bytes = tip_bytes + sizeof(double) * ((size_t)slots * kLlR * kTile +
                                               max_macros(n) * kMacroPositions * 2 + (subst ? 4 : 0));
  if (rescale) bytes += ((sizeof(int16_t) * (size_t)max_stored(n) * kLlR * (16 / kp) + 7) / 8) * 8;
  const size_t reach = (size_t)(2 * n - 1) * 4 * (16 / kp);  // mask fetches of internal ids
  return bytes > reach ? bytes : reach;
}
size_t gradient_mfma_lds_bytes(int n, int K, bool rescale, bool subst) {
  return gradient_mfma_lds_bytes_for(n, K, rescale, subst, max_stored(n));
}
int gradient_mfma_width(int n, bool subst) {
  return max_macros(n) * kMacroPositions * 2 + (subst ? kSubstExtra : 0);
}
int gradient_mfma_groups(int K) { return K <= 4 ? 1 : (K + 3) / 4; }
int gradient_mfma_tiles(int P, int K) {
  const int per_wave = kLlR * (16 / (K == 1 ? 1 : (K == 2 ? 2 : 4)));
  return (P + per_wave - 1) / per_wave;
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
  return (size_t)gradient_mfma_tiles(P, K) * gradient_mfma_groups(K) * max_stored(n) * kLlR * kTile *
         sizeof(double);
}
// The arena variant runs when keeping every stored vector in LDS would leave 6 or fewer
// waves per CU (the registers allow 8), or would not fit at all: measured cross-over with
// 934 patterns at 31-32 taxa (29 taxa: 1.34 ms per 1000 trees in LDS / 1.47 arena; 32: 1.60 /
// 1.56; 42: 2.79 / 2.03).  MI_PHYLO_GRADIENT_STORE=lds|arena forces one of the two.
bool gradient_mfma_use_arena(int n, int K, bool rescale, bool subst, size_t waves) {
  // (read at every call: tools/audit_paths.py switches it between engines of one process)
  const int forced = [] {
    const char* env = getenv("MI_PHYLO_GRADIENT_STORE");
    if (!env) return 0;
    return std::string(env) == "arena" ? 2 : (std::string(env) == "lds" ? 1 : 0);
  }();
  const size_t lds_all = gradient_mfma_lds_bytes(n, K, rescale, subst);
  const bool lds_fits = lds_all <= 160 * 1024;
  if (forced == 1 && lds_fits) return false;
  if (forced == 2) return true;
  if (lds_fits && arena_single_launch(lds_all, waves)) return false;  // a call of a few trees
  // (round 5, tools/audit_paths.py with the store forced either way: with five and six waves per
  // CU the LDS store still wins -- 31 taxa x 1000 patterns x 4 categories 1.29 against 1.54 ms per
  // 1000 trees, 36 x 200: 0.41 / 0.46, one category 0.36 / 0.41 -- and from four waves down the
  // arena does; the round-1 cross-over "fewer than seven" dated from the first generation)
  return !lds_fits || (160 * 1024) / lds_all < 5;
}
bool gradient_mfma_fits(int n, int K, bool rescale) {
  if (n < 3 || K > kMaxCategories) return false;
  if (gradient_mfma_lds_bytes(n, K, rescale, true) <= 160 * 1024) return true;
  return gradient_mfma_lds_bytes_for(n, K, rescale, true, gradient_arena_slots_sure(n)) <= 160 * 1024;
}

// One thread per tree: the macro schedule re-ordered and given reusable LDS slots.
// tree_setup lists the macros by node id -- a post-order, but one that can keep many
// vectors alive.  Here the macro tree (a macro's inputs are the stored nodes among its
// children / grandchildren, up to four) is walked Sethi-Ullman style, the input needing
// the most slots first, and the stored vectors' LDS slots are an interval colouring in
// that order (free the inputs' slots, take the lowest free one for the node) and replace
// the node-unique numbers in the slot fields; the arena index of a stored node's vector
// goes to its `pad` field.  need[t] = slots the tree uses.
struct MacroInputs {
  int count;
  int field[4];  // which slot field: 0,1 = cslot[j]; 2..5 = gslot[g]
};
__device__ inline MacroInputs macro_inputs(const MacroEntry& e) {  // which inputs are stored nodes
  MacroInputs in{0, {0, 0, 0, 0}};
  for (int j = 0; j < 2; j++) {
    const int kind = (e.shape >> (2 * j)) & 3;
    if (kind == 1) in.field[in.count++] = j;
    if (kind == 2)
      for (int g = 2 * j; g < 2 * j + 2; g++)
        if (!((e.shape >> (10 + g)) & 1)) in.field[in.count++] = 2 + g;
  }
  return in;
}
__device__ inline int32_t& macro_field(MacroEntry& e, int f) { return f < 2 ? e.cslot[f] : e.gslot[f - 2]; }
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
  __shared__ int more[3], used_s, placed_s, wave_tot[16];
  const int t = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
  const int Mmax = max_macros(n), S = max_stored(n);
  MacroEntry* ent = reinterpret_cast<MacroEntry*>(slot_scratch);
  // (typed as LDS pointers: volatile accesses through generic pointers stay flat instructions)
#define MS_LDS(T) volatile __attribute__((address_space(3))) T*
  typedef __attribute__((address_space(3))) int32_t* ms_lds_base;
  const ms_lds_base ms3 = (ms_lds_base)slot_scratch;
  MS_LDS(uint64_t) Lw = (MS_LDS(uint64_t))(ms3 + (size_t)Mmax * 16);
  MS_LDS(uint64_t) Dw = Lw + Mmax;  // ready << 63 | expanded << 62 | slot << 32 | start
  MS_LDS(int32_t) mac_of = (MS_LDS(int32_t))(Dw + Mmax);  // stored id -> macro
  MS_LDS(int32_t) order = mac_of + S;
  MS_LDS(int32_t) base = order + Mmax;  // arena index of a position's first input
#undef MS_LDS
  const int M = macro_count[t];
  if (M <= 0) {
    if (tid == 0) need[t] = 0;
    return;
  }
  {
    const int32_t* src = reinterpret_cast<const int32_t*>(macros_in + (size_t)t * macro_stride(n));
    for (int i = tid; i < M * 16; i += nthreads) slot_scratch[i] = src[i];
  }
  if (tid == 0) more[0] = more[1] = more[2] = used_s = placed_s = 0;
  __syncthreads();
  for (int m = tid; m < M; m += nthreads) {
    Lw[m] = 0;
    Dw[m] = 0;
    if (!(ent[m].shape & 16)) mac_of[ent[m].qslot] = m;
  }
  __syncthreads();
  int round = 0;
  auto end_round = [&]() {
    __syncthreads();
    const int mo = more[round % 3];
    if (tid == 0) more[(round + 2) % 3] = 0;
    round++;
    return mo;
  };
  // What is fixed about a thread's own macros sits in registers (a round then costs one LDS
  // round trip and the barrier); kOwn macros per thread at most.
  constexpr int kOwn = 2;
  int own_k[kOwn], own_in[kOwn][4], own_size[kOwn], own_csize[kOwn][4];
  bool own_todo[kOwn], own_root[kOwn];
#pragma unroll
  for (int j = 0; j < kOwn; j++) {
    const int m = tid + j * nthreads;
    own_todo[j] = m < M;
    const int mm = own_todo[j] ? m : 0;
    const MacroInputs mi = macro_inputs(ent[mm]);
    own_k[j] = mi.count;
    own_root[j] = (ent[mm].shape & 16) != 0;
    own_size[j] = 1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      own_in[j][i] = i < mi.count ? mac_of[macro_field(ent[mm], mi.field[i])] : 0;
      own_csize[j][i] = 0;
    }
  }
  // 1. labels and subtree sizes, bottom-up (word: done << 63 | inputs << 40 | macros below << 16 | label)
  do {
#pragma unroll
    for (int j = 0; j < kOwn; j++) {
      if (!own_todo[j]) continue;
      const int m = tid + j * nthreads, k = own_k[j];
      uint64_t w[4];
      bool ready = true;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        w[i] = i < k ? Lw[own_in[j][i]] : (1ull << 63);
        ready = ready && (w[i] >> 63);
      }
      if (!ready) continue;
      // inputs by label, largest first, ties in field order (stable insertion sort)
#pragma unroll
      for (int i = 1; i < 4; i++)
#pragma unroll
        for (int q = i; q > 0; q--) {
          const bool swap = q < k && (int)(w[q] & 0xffff) > (int)(w[q - 1] & 0xffff);
          const int x0 = own_in[j][q - 1], x1 = own_in[j][q];
          const uint64_t y0 = w[q - 1], y1 = w[q];
          own_in[j][q - 1] = swap ? x1 : x0;
          own_in[j][q] = swap ? x0 : x1;
          w[q - 1] = swap ? y1 : y0;
          w[q] = swap ? y0 : y1;
        }
      int l = own_root[j] ? k : (k > 1 ? k : 1), size = 1;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (i >= k) continue;
        const int li = (int)(w[i] & 0xffff) + i;
        l = l > li ? l : li;
        own_csize[j][i] = (int)(w[i] >> 16) & 0xffffff;
        size += own_csize[j][i];
      }
      own_size[j] = size;
      Lw[m] = (1ull << 63) | ((uint64_t)k << 40) | ((uint64_t)size << 16) | (uint64_t)l;
      own_todo[j] = false;
      more[round % 3] = 1;  // (progress)
    }
  } while (end_round());
  bool all_done = true;
#pragma unroll
  for (int j = 0; j < kOwn; j++) {
    all_done = all_done && !own_todo[j];
    own_todo[j] = tid + j * nthreads < M;
  }
  // 2. post-order positions and LDS slots, top-down from the root (the last macro)
  if (tid == 0) Dw[M - 1] = 1ull << 63;
  __syncthreads();
  do {
#pragma unroll
    for (int j = 0; j < kOwn; j++) {
      if (!own_todo[j]) continue;
      const int m = tid + j * nthreads;
      const uint64_t d = Dw[m];
      if (!(d >> 63)) continue;
      const int k = own_k[j];
      const int b = (int)(d >> 32) & 0xffff;
      int st = (int)(d & 0xffffffffu);
      order[st + own_size[j] - 1] = m;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (i >= k) continue;
        Dw[own_in[j][i]] = (1ull << 63) | ((uint64_t)(b + i) << 32) | (uint64_t)st;
        st += own_csize[j][i];
      }
      if (!own_root[j]) atomicMax(&used_s, b + 1);
      atomicAdd(&placed_s, 1);
      own_todo[j] = false;
      more[round % 3] = 1;
    }
  } while (end_round());
  (void)all_done;
  const bool complete = placed_s == M;
  // 2b. arena indices: exclusive prefix sum of the input counts over the post-order
  if (complete) {
    const int lane = tid & 63, wave = tid >> 6, waves = nthreads >> 6;
    int carry = 0;
    for (int o0 = 0; o0 < M; o0 += nthreads) {
      const int o = o0 + tid;
      const int k = o < M ? (int)(Lw[order[o]] >> 40) & 0xff : 0;
      int incl = k;
      for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(incl, d, 64);
        if (lane >= d) incl += y;
      }
      if (lane == 63) wave_tot[wave] = incl;
      __syncthreads();
      int off = carry;
      for (int w2 = 0; w2 < wave; w2++) off += wave_tot[w2];
      if (o < M) base[o] = off + incl - k;
      int total = 0;
      for (int w2 = 0; w2 < waves; w2++) total += wave_tot[w2];
      carry += total;
      __syncthreads();
    }
    // a macro's stored inputs are numbered in FIELD order from its base; each input learns
    // where its vector goes (pad); slot fields become LDS slots
    for (int o = tid; o < M; o += nthreads) {
      MacroEntry& e = ent[order[o]];
      const MacroInputs mi = macro_inputs(e);
      const int b0 = base[o];
      e.shape |= b0 << 16;
      for (int i = 0; i < mi.count; i++) {
        int32_t& f = macro_field(e, mi.field[i]);
        const int c = mac_of[f];
        ent[c].pad = b0 + i;
        f = (int)(Dw[c] >> 32) & 0xffff;
      }
    }
    __syncthreads();
    for (int m = tid; m < M; m += nthreads)
      if (!(ent[m].shape & 16)) ent[m].qslot = (int)(Dw[m] >> 32) & 0xffff;
  }
  if (tid == 0) {
    need[t] = used_s;
    if (used_s > sure || !complete) set_status(status, kTooManySlots, t);
  }
  __syncthreads();
  if (complete) {
    int32_t* dst = reinterpret_cast<int32_t*>(macros_out + (size_t)t * macro_stride(n));
    for (int i = tid; i < M * 16; i += nthreads) dst[i] = slot_scratch[order[i >> 4] * 16 + (i & 15)];
  }
}
void launch_macro_slots(const MacroEntry* macros_in, MacroEntry* macros_out,
                        const int32_t* macro_count, int n, int T, int32_t* need, int32_t* status,
                        hipStream_t s) {
  static const bool seq = getenv("MI_PHYLO_MACRO_SLOTS") && std::string(getenv("MI_PHYLO_MACRO_SLOTS")) == "seq";
  const size_t Mmax = max_macros(n), S = max_stored(n);
  const size_t wg_lds = sizeof(int32_t) * (Mmax * (16 + 2 + 2 + 1 + 1) + S);
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

template <bool RESCALE, bool SUBST, bool ARENA>
static void launch_gradient_mfma_variant(const LikArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  allow_large_lds(
      reinterpret_cast<const void*>(gradient_mfma_kernel<kLlR, 0, RESCALE, SUBST, ARENA>), lds);
  hipLaunchKernelGGL((gradient_mfma_kernel<kLlR, 0, RESCALE, SUBST, ARENA>), grid, dim3(kTile), lds, s,
                     a);
}
template <bool ARENA>
static void launch_gradient_mfma_store(const LikArgs& a, dim3 grid, size_t lds, bool rescale,
                                       bool subst, hipStream_t s) {
  if (rescale && subst) launch_gradient_mfma_variant<true, true, ARENA>(a, grid, lds, s);
  else if (rescale) launch_gradient_mfma_variant<true, false, ARENA>(a, grid, lds, s);
  else if (subst) launch_gradient_mfma_variant<false, true, ARENA>(a, grid, lds, s);
  else launch_gradient_mfma_variant<false, false, ARENA>(a, grid, lds, s);
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
void launch_gradient_mfma(const LikArgs& a_in, int count, bool rescale, bool subst,
                          hipStream_t s) {
  if (count <= 0) return;
  LikArgs a = a_in;
  a.kp = a.K == 1 ? 1 : (a.K == 2 ? 2 : 4);
  a.cat_groups = gradient_mfma_groups(a.K);
  const dim3 grid(gradient_mfma_tiles(a.P, a.K) * a.cat_groups, count);
  if (a.store ? a.store == 2 : gradient_mfma_use_arena(a.n, a.K, rescale, subst, (size_t)grid.x * grid.y)) {
    // two launches over the same grid: the trees that fit the usual number of LDS slots,
    // then (more LDS per wave) the rest; a wave of the other launch's tree exits at once
    const int usual = gradient_arena_slots_usual(a.n), sure = gradient_arena_slots_sure(a.n);
    a.lds_lo = -1;
    if (arena_single_launch(gradient_mfma_lds_bytes_for(a.n, a.K, rescale, subst, sure),
                            (size_t)grid.x * grid.y)) {
      // few waves (a call of one or a few trees): all of them resident at once even with the
      // larger LDS footprint -- ONE launch that takes every tree
      a.lds_slots = sure;
      launch_gradient_mfma_store<true>(
          a, grid, gradient_mfma_lds_bytes_for(a.n, a.K, rescale, subst, sure), rescale, subst, s);
      return;
    }
    a.lds_slots = usual;
    launch_gradient_mfma_store<true>(
        a, grid, gradient_mfma_lds_bytes_for(a.n, a.K, rescale, subst, usual), rescale, subst, s);
    if (sure > usual) {
      a.lds_lo = usual;
      a.lds_slots = sure;
      launch_gradient_mfma_store<true>(
          a, grid, gradient_mfma_lds_bytes_for(a.n, a.K, rescale, subst, sure), rescale, subst, s);
    }
    return;
  }
  const size_t lds = gradient_mfma_lds_bytes(a.n, a.K, rescale, subst);
  if (rescale || subst) {
    launch_gradient_mfma_store<false>(a, grid, lds, rescale, subst, s);
    return;
  }
#ifdef MI_PHYLO_ABLATION
  // Ablation variants -- kernels that return WRONG numbers by design, so they
  // exist only in a library built with `make ablation` (-DMI_PHYLO_ABLATION), never in the
  // product build: 1 no matrix products, 2 no cross-lane reductions, 8 no LDS vector
  // traffic, 16 no matrix loads, 64 prologue only, 128 post-order only
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
#endif
  launch_gradient_mfma_store<false>(a, grid, lds, false, false, s);
}
int gradient_mfma_waves_per_cu(int n, int K) {
  const size_t lds = gradient_mfma_use_arena(n, K, false, false)
                         ? gradient_mfma_lds_bytes_for(n, K, false, false, gradient_arena_slots_usual(n))
                         : gradient_mfma_lds_bytes(n, K, false, false);
  return (int)std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1));
}
const char* gradient_kernel_name() { return "gradient_hbm_kernel"; }
const char* gradient_mfma_kernel_name() { return "gradient_mfma_kernel"; }

}  // namespace miphylo
