// Log-likelihood kernels: matrix-core (default) and VALU.
// (gfx950 / CDNA4, wave64; see DESIGN.md for the mapping and what bounds each kernel.)
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <type_traits>

#include "mi_phylo_device_utils.h"
#include "mi_phylo_kernels.h"

// (timing builds of the matrix-core kernel: see its visit)
#ifndef LL_ABL_TIPS
#define LL_ABL_TIPS 0
#endif
#ifndef LL_WAVES
#define LL_WAVES 5
#endif

namespace miphylo {

namespace {
using namespace dev;

// v[lane] += v[lane rotated right by SHIFT within its 16-lane row]
template <int SHIFT>
__device__ __forceinline__ double ll_row_ror_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int slo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + SHIFT, 0xf, 0xf, true);
  const int shi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + SHIFT, 0xf, 0xf, true);
  return v + __hiloint2double(shi, slo);
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
template <int R, bool RESCALE, bool MULTI>
// (kTile, 5): LDS allows ~20 waves per CU for typical trees; and with at most 256 registers per lane the compiler keeps the products in
// ordinary vector registers; without the bound it places them in accumulation registers
// and spends two v_accvgpr_read per product to get them back
__global__ __launch_bounds__(kTile, LL_WAVES) void loglik_mfma_kernel(LikArgs a) {
  static_assert(R <= 8, "tip masks of one column group are packed in one 32- or 64-bit word");
  using TipWord = std::conditional_t<(R > 4), uint64_t, uint32_t>;
  constexpr unsigned TB = sizeof(TipWord);  // bytes of tip masks per (taxon, column)
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int hi = lane >> 4, b = (lane >> 2) & 3, lo = lane & 3;
  // A wave walks `evals_per_wave` consecutive evaluations of ONE tree (the finite-difference
  // models of a GTR gradient call: same tree, same tips, other matrices) for its tile: tip
  // masks and schedule are staged once for all of them.  grid.y counts the groups.
  const TileEval te = xcd_tile_eval();
  const int epw = a.evals_per_wave;
  const int e0 = a.eval_offset + te.eval * epw;
  int t, mi0;
  a.map.decode(e0, t, mi0);
  const SchedEntry* __restrict__ sched = a.sched + (size_t)t * (a.n - 1);
  const int K = a.K, n = a.n, Kp = a.kp;       // Kp in {1, 2, 4}: categories per instruction
  const int cat = b % Kp, pgrp = b / Kp, ppr = 16 / Kp;  // ppr = patterns per register
  // K > 4: the categories are walked four at a time (`groups` complete walks that only
  // meet in the per-pattern site likelihood); the per-group lane constants follow
  // (MULTI: a compile-time 1 keeps the site-likelihood accumulators out of the walk's registers)
  const int groups = MULTI ? (K + 3) / 4 : 1;
  const char* __restrict__ mats_e = nullptr;  // this evaluation's matrices (set per evaluation)
  unsigned a_off = 0;  // per-lane element of a child's matrix block: A_b[i = lo][k = hi] (bytes)
  double wgt = 0.0;    // category weight x stationary frequency of this lane
  const unsigned node_bytes = (unsigned)K * 128u;
  const int TP = ppr * R, tile_start = te.tile * TP;
  const int col = pgrp * 4 + lo;  // this lane's pattern column; register r adds r * ppr
  // the epilogue's lane (hi, column) owns register r = hi (+4, ...): its pattern and weight
  constexpr int kOwned = (R + 3) / 4;
  int own_pat[kOwned];
  double own_w[kOwned];
#pragma unroll
  for (int q = 0; q < kOwned; q++) {
    own_pat[q] = tile_start + (4 * q + hi) * ppr + col;
    own_w[q] = a.weights[own_pat[q] < a.P ? own_pat[q] : a.P - 1];
  }
  // LDS: tip state masks [taxon][column][r] (bit s: compatible with state s; first, so
  // that the ignored mask fetch of an internal node id needs no clamping) | schedule |
  // vectors [slot][r][lane]
  uint8_t* tips = reinterpret_cast<uint8_t*>(lds);
  SchedEntry* sched_l = reinterpret_cast<SchedEntry*>(lds + ((n * ppr * TB + 7) >> 3));
  double* plv = reinterpret_cast<double*>(sched_l + (n - 1));
  // In a post-order the visit just before a node's is that of one of its children
  // (unless both are tips): that child -- made child 1 here, the element-wise product
  // commutes -- is taken from the registers the previous visit left it in, and a visit
  // whose successor consumes it that way does not store its vector at all.
  constexpr int kFromPrev = 1 << 26, kStore = 1 << 27;
  for (int i = lane; i < n - 1; i += kTile) {
    SchedEntry v = sched[i];
    const int prev = i > 0 ? sched[i - 1].node : -1;
    if (v.child0 == prev && v.child1 != prev) {
      const int c = v.child0;
      v.child0 = v.child1;
      v.child1 = c;
      const int sl = v.slots;
      v.slots = (sl & 0xff) | ((sl >> 8) & 0xff00) | ((sl & 0xff00) << 8) |
                ((sl >> 1) & (1 << 24)) | ((sl & (1 << 24)) << 1);
    }
    if (v.child1 == prev) v.slots |= kFromPrev;
    // (the root's vector is read from L after the walk)
    const bool consumed =
        i == n - 2 || sched[i + 1].child0 == v.node || sched[i + 1].child1 == v.node;
    if (!consumed) {
      v.slots |= kStore;
      // stored vectors use the low slot numbers only (loglik_mfma_slots); anything else
      // is an internal error that must not pass silently
      if ((v.slots & 0xff) >= a.lds_slots) set_status(a.status, kTooManySlots, t);
    }
    sched_l[i] = v;
  }
  if (a.tip_tiles) {
    // (round 6) the tile's tip bytes in exactly this layout, prepared once per engine
    // (tip_tiles_kernel): 8-byte copies -- the general staging below is ~200 vector instructions
    // of a tile job's ~950, 0.241 -> 0.231 ms per 1000 DS1 log-likelihoods
    const int words = (n * ppr * (int)TB + 7) >> 3;
    const uint64_t* src = reinterpret_cast<const uint64_t*>(a.tip_tiles) + (size_t)te.tile * words;
    for (int q = lane; q < words; q += kTile) reinterpret_cast<uint64_t*>(tips)[q] = src[q];
  } else {
    const int tp_shift = TP <= 16 ? 4 : (TP <= 32 ? 5 : 6);
    const int group = 64 >> tp_shift;
    const int ppr_shift = Kp == 4 ? 2 : (Kp == 2 ? 3 : 4);
    for (int qb = 0; qb < TP; qb += kTile) {  // one trip unless the tile is wider than the wave
      const int q = qb + (lane & ((1 << tp_shift) - 1));
      const int r = q >> ppr_shift, c = q & (ppr - 1);
      if (q < TP) {
        const int pp = tile_start + q < a.P ? tile_start + q : a.P - 1;
        const uint8_t* src = a.tip_masks + pp;
#pragma unroll 4
        for (int taxon = lane >> tp_shift; taxon < n; taxon += group)
          tips[(taxon * ppr + c) * TB + r] = src[(size_t)taxon * a.P];
      }
    }
  }
  __syncthreads();

  // What a visit needs from memory is requested kAhead visits before it is used (the
  // schedule sits in LDS, so future visits' children are known): the two matrix
  // registers, the two tip words, and the entry itself.  Child ids stay vector
  // registers (multiplicands of per-lane addresses); the `slots` word, which also
  // carries the two is-a-tip flags, is the only scalar.
#if LL_ABL_TIPS == 3
  // TIMING build 3: a tip child's products are gathered from its matrix block when the visit
  // is requested (kAhead visits early), in place of the matrix register; no tip bits, no
  // matrix instruction (the state index is made up from the mask byte: wrong results)
  constexpr int kAhead = 3;
  struct Ahead {
    double G0[R], G1[R];  // tip child: its products; internal child: [0] = the matrix register
    int slots;
  };
  const unsigned lane8 = 8u * lane, col4 = TB * col;
  const unsigned row_off = 8u * ((lane >> 2 & 3) * 16 + hi * 4);
  auto request = [&](int i) {
    const SchedEntry sv = sched_l[i < n - 1 ? i : n - 2];
    int c0 = sv.child0, c1 = sv.child1;
    asm volatile("" : "+v"(c0), "+v"(c1));
    const int slots = __builtin_amdgcn_readfirstlane(sv.slots);
    Ahead h;
    h.slots = slots;
    if (slots & (1 << 24)) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(tips + (__umul24((unsigned)c0, (unsigned)ppr * TB) + col4));
      const unsigned g = __umul24((unsigned)c0, node_bytes) + row_off;
#pragma unroll
      for (int r = 0; r < R; r++)
        h.G0[r] = *reinterpret_cast<const double*>(mats_e + (g + ((w >> (8 * r)) & 3u) * 8u));
    } else {
      h.G0[0] = *reinterpret_cast<const double*>(mats_e + (__umul24((unsigned)c0, node_bytes) + a_off));
    }
    if (slots & (1 << 25)) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(tips + (__umul24((unsigned)c1, (unsigned)ppr * TB) + col4));
      const unsigned g = __umul24((unsigned)c1, node_bytes) + row_off;
#pragma unroll
      for (int r = 0; r < R; r++)
        h.G1[r] = *reinterpret_cast<const double*>(mats_e + (g + ((w >> (8 * r)) & 3u) * 8u));
    } else {
      h.G1[0] = *reinterpret_cast<const double*>(mats_e + (__umul24((unsigned)c1, node_bytes) + a_off));
    }
    return h;
  };
#else
  constexpr int kAhead = 4;
  struct Ahead {
    double A0, A1;
    TipWord w0, w1;
    int slots;
#if LL_ABL_TIPS == 2
    unsigned g0, g1;  // (timing build: byte offsets of the children's matrix blocks)
#endif
  };
  const unsigned lane8 = 8u * lane, col4 = TB * col;
  auto request = [&](int i) {
    const SchedEntry sv = sched_l[i < n - 1 ? i : n - 2];
    int c0 = sv.child0, c1 = sv.child1;
    asm volatile("" : "+v"(c0), "+v"(c1));  // stay vector operands (see gradient_mfma_kernel)
    Ahead h;
    h.A0 = *reinterpret_cast<const double*>(mats_e + (__umul24((unsigned)c0, node_bytes) + a_off));
    h.A1 = *reinterpret_cast<const double*>(mats_e + (__umul24((unsigned)c1, node_bytes) + a_off));
    h.w0 = *reinterpret_cast<const TipWord*>(tips + (__umul24((unsigned)c0, (unsigned)ppr * TB) + col4));
    h.w1 = *reinterpret_cast<const TipWord*>(tips + (__umul24((unsigned)c1, (unsigned)ppr * TB) + col4));
    h.slots = sv.slots;
#if LL_ABL_TIPS == 2
    h.g0 = __umul24((unsigned)c0, node_bytes) + 8u * ((lane >> 2 & 3) * 16 + hi * 4);
    h.g1 = __umul24((unsigned)c1, node_bytes) + 8u * ((lane >> 2 & 3) * 16 + hi * 4);
#endif
    return h;
  };
#endif
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
  auto tip_bit = [&](TipWord w, int r) {  // 0.0 / 1.0: is this lane's state compatible with the tip
    const uint32_t half = r < 4 ? (uint32_t)w : (uint32_t)((uint64_t)w >> 32);
    return (double)__builtin_amdgcn_ubfe(half, (uint32_t)(8 * (r & 3) + hi), 1u);
  };
#if LL_ABL_TIPS == 3
  auto visit = [&](int i, Ahead& h) {
    const int slots = h.slots;
    double D0[R];
    if (slots & (1 << 24)) {
#pragma unroll
      for (int r = 0; r < R; r++) D0[r] = h.G0[r];
    } else {
      const double* src = slot_ptr((slots >> 8) & 0xff);
      const double A0 = h.G0[0];
#pragma unroll
      for (int r = 0; r < R; r++) D0[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(A0, src[r * kTile], 0.0, 0, 0, 0);
    }
    if ((slots & (1 << 25)) && !(slots & kFromPrev)) {
#pragma unroll
      for (int r = 0; r < R; r++) L[r] = h.G1[r];
    } else {
      if (!(slots & kFromPrev)) {
        const double* src = slot_ptr((slots >> 16) & 0xff);
#pragma unroll
        for (int r = 0; r < R; r++) L[r] = src[r * kTile];
      }
      const double A1 = h.G1[0];
#pragma unroll
      for (int r = 0; r < R; r++) L[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(A1, L[r], 0.0, 0, 0, 0);
    }
    h = request(i + kAhead);
#pragma unroll
    for (int r = 0; r < R; r++) L[r] = D0[r] * L[r];
    if (slots & kStore) {
      double* dst = slot_ptr(slots & 0xff);
#pragma unroll
      for (int r = 0; r < R; r++) dst[r * kTile] = L[r];
    }
  };
#elif LL_ABL_TIPS
  // TIMING builds only (wrong results): what a tip child costs.  1: its product is a constant
  // (no matrix instruction, no tip bits); 2: its product is gathered from the node's matrix
  // block at the moment of use (the look-up's memory pattern, latency exposed).
  auto visit = [&](int i, Ahead& h) {
    const int slots = __builtin_amdgcn_readfirstlane(h.slots);
    double D0[R];
    const double A0 = h.A0, A1 = h.A1;
    if (slots & (1 << 24)) {
#pragma unroll
      for (int r = 0; r < R; r++) {
#if LL_ABL_TIPS == 1
        D0[r] = 0.25;
#elif LL_ABL_TIPS == 4
        {  // (timing build 4: the product permuted out of the matrix register across lanes)
          const int at = (int)(4u * lane + (((uint32_t)h.w0 >> (8 * r)) & 0xffu));
          D0[r] = __hiloint2double(__builtin_amdgcn_ds_bpermute(at, __double2hiint(A0)),
                                   __builtin_amdgcn_ds_bpermute(at, __double2loint(A0)));
        }
#else
        D0[r] = *reinterpret_cast<const double*>(mats_e + (h.g0 + (((uint32_t)h.w0 >> (8 * r)) & 3u) * 8u));
#endif
      }
    } else {
      const double* src = slot_ptr((slots >> 8) & 0xff);
#pragma unroll
      for (int r = 0; r < R; r++) D0[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(A0, src[r * kTile], 0.0, 0, 0, 0);
    }
    if ((slots & (1 << 25)) && !(slots & kFromPrev)) {
#pragma unroll
      for (int r = 0; r < R; r++) {
#if LL_ABL_TIPS == 1
        L[r] = 0.25;
#elif LL_ABL_TIPS == 4
        {
          const int at = (int)(4u * lane + (((uint32_t)h.w1 >> (8 * r)) & 0xffu));
          L[r] = __hiloint2double(__builtin_amdgcn_ds_bpermute(at, __double2hiint(A1)),
                                  __builtin_amdgcn_ds_bpermute(at, __double2loint(A1)));
        }
#else
        L[r] = *reinterpret_cast<const double*>(mats_e + (h.g1 + (((uint32_t)h.w1 >> (8 * r)) & 3u) * 8u));
#endif
      }
    } else {
      if (!(slots & kFromPrev)) {
        const double* src = slot_ptr((slots >> 16) & 0xff);
#pragma unroll
        for (int r = 0; r < R; r++) L[r] = src[r * kTile];
      }
#pragma unroll
      for (int r = 0; r < R; r++) L[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(A1, L[r], 0.0, 0, 0, 0);
    }
    h = request(i + kAhead);
#pragma unroll
    for (int r = 0; r < R; r++) L[r] = D0[r] * L[r];
    if (slots & kStore) {
      double* dst = slot_ptr(slots & 0xff);
#pragma unroll
      for (int r = 0; r < R; r++) dst[r * kTile] = L[r];
    }
  };
#else
  auto visit = [&](int i, Ahead& h) {
    const int slots = __builtin_amdgcn_readfirstlane(h.slots);
    double B0[R];
    if (slots & (1 << 24)) {
#pragma unroll
      for (int r = 0; r < R; r++) B0[r] = tip_bit(h.w0, r);
    } else {
      const double* src = slot_ptr((slots >> 8) & 0xff);
#pragma unroll
      for (int r = 0; r < R; r++) B0[r] = src[r * kTile];
    }
    // child 1: normally the previous visit's result, still in L
    if (!(slots & kFromPrev)) {
      if (slots & (1 << 25)) {
#pragma unroll
        for (int r = 0; r < R; r++) L[r] = tip_bit(h.w1, r);
      } else {
        const double* src = slot_ptr((slots >> 16) & 0xff);
#pragma unroll
        for (int r = 0; r < R; r++) L[r] = src[r * kTile];
      }
    }
    const double A0 = h.A0, A1 = h.A1;
    h = request(i + kAhead);  // refill this ring slot
#pragma unroll
    for (int r = 0; r < R; r++) {
      const double D0 = __builtin_amdgcn_mfma_f64_4x4x4f64(A0, B0[r], 0.0, 0, 0, 0);
      const double D1 = __builtin_amdgcn_mfma_f64_4x4x4f64(A1, L[r], 0.0, 0, 0, 0);
      L[r] = D0 * D1;
      if (RESCALE) {
        // exact power-of-two rescaling per (pattern, category): the scale is the exponent
        // of the SUM over the four states, which one product with a ones matrix leaves in
        // all four lanes of the column -- no cross-lane traffic, and the matrix pipe has
        // the slack (any power of two works: the mantissas are unchanged; 0 -> exponent 0).
        // The categories' exponents meet at the root.
        const double colsum = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, L[r], 0.0, 0, 0, 0);
        const int ex = __builtin_amdgcn_frexp_exp(colsum);
        L[r] = ldexp(L[r], -ex);
        esum[r] += ex;
      }
    }
    if (slots & kStore) {
      double* dst = slot_ptr(slots & 0xff);
#pragma unroll
      for (int r = 0; r < R; r++) dst[r * kTile] = L[r];
    }
  };
#endif
  for (int ev = 0; ev < epw; ev++) {
  const int e = e0 + ev;
  int t_ev, mi;
  a.map.decode(e, t_ev, mi);
  const DevModel* __restrict__ model = a.models + mi;
  mats_e = reinterpret_cast<const char*>(a.mats + (size_t)e * (a.N - 1) * K * 16);
  double site[R];
  int site_exp[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    site[r] = 0.0;
    site_exp[r] = 0;
    L[r] = 0.0;
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
      int ev = esum[r];  // RESCALE: common exponent of this pattern's categories
      if (RESCALE) {
        if (Kp >= 2) ev = max(ev, __shfl_xor(ev, 4, 64));
        if (Kp >= 4) ev = max(ev, __shfl_xor(ev, 8, 64));
        v = ldexp(v, esum[r] - ev);
      }
      // states: one product with a ones matrix leaves the column sums in every row;
      // categories (four lanes apart in a row): two row rotations -- no LDS round trips
      // (round 3; the ds_bpermute butterfly was a chain of four per register)
      v = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, v, 0.0, 0, 0, 0);
      if (Kp == 4) {
        v = ll_row_ror_add<8>(v);
        v = ll_row_ror_add<4>(v);
      } else if (Kp == 2) {
        v += __shfl_xor(v, 4, 64);
      }
      if (!RESCALE) {
        site[r] += v;
      } else if (g == 0) {
        site[r] = v;
        site_exp[r] = ev;
      } else if (ev > site_exp[r]) {
        site[r] = ldexp(site[r], site_exp[r] - ev) + v;
        site_exp[r] = ev;
      } else {
        site[r] += ldexp(v, ev - site_exp[r]);
      }
    }
  }
  // every lane of a pattern holds its site likelihood: the lane with state index hi
  // takes register r = hi (+4, ...), so a wave evaluates each logarithm once
  double ll = 0.0;
#pragma unroll
  for (int r0 = 0; r0 < R; r0 += 4) {
    double sv = site[r0];
    const double wv = own_w[r0 / 4];
    const int pv = own_pat[r0 / 4];
    int ev = site_exp[r0];
#pragma unroll
    for (int j = 1; j < 4 && r0 + j < R; j++) {
      sv = hi == j ? site[r0 + j] : sv;
      ev = hi == j ? site_exp[r0 + j] : ev;
    }
    const bool owner = r0 + hi < R && cat == 0 && pv < a.P;  // one lane per pattern
    if (owner) {
      if (a.site_lik) {
        // per-pattern site likelihood for a following gradient pass (rescaled: the
        // mantissa here, the power of two in site_exp)
        const size_t at = ((size_t)a.grad_offset + (e - a.eval_offset)) * a.tiles * kTile + pv;
        a.site_lik[at] = sv;
        if (RESCALE) a.site_exp[at] = ev;
      }
      ll += wv * (RESCALE ? log(sv) + ev * 0.69314718055994530942 : log(sv));
    }
  }
  ll = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, ll, 0.0, 0, 0, 0);  // the four rows
  ll = ll_row_ror_add<8>(ll);
  ll = ll_row_ror_add<4>(ll);
  ll = ll_row_ror_add<2>(ll);
  ll = ll_row_ror_add<1>(ll);
  if (lane == 0) a.ll_part[(size_t)e * a.ll_tiles + te.tile] = ll;
  }  // evaluations of this wave
}

}  // namespace

// ------------------------------------------------------------------------
// Launch wrappers
// ------------------------------------------------------------------------
// which log-likelihood kernel runs by default when both can (measured)
constexpr bool kLoglikMfmaDefault = true;
constexpr int kLogR = 4;  // registers (16 columns each) per node in loglik_mfma_kernel: 3 -> 4 is 8 % faster, 6 (64-bit tip words) 12 % slower
// LDS vector slots of the matrix-core kernel.  The schedule numbers its slots for a walk
// that stores every node (max_slots = floor(log2 n) + 1 bounds them); with the chain child
// forwarded in registers the vectors that are really stored -- the first-visited child of a
// node with two internal children -- only ever carry the numbers below max_slots - 2 (a
// stored vector coexists with at least the two nodes of the sibling subtree that end that
// subtree's walk); the kernel checks it.
static int loglik_mfma_slots(int max_slots) { return max_slots > 3 ? max_slots - 2 : 1; }
static size_t loglik_mfma_lds_bytes(int n, int K, int max_slots) {
  const int kp = K == 1 ? 1 : (K == 2 ? 2 : 4);
  const size_t tb = kLogR > 4 ? 8 : 4;  // bytes of tip masks per (taxon, column)
  const size_t tip_bytes = (((size_t)n * tb * (16 / kp) + 7) / 8) * 8;
  const size_t bytes = tip_bytes + sizeof(SchedEntry) * (size_t)(n - 1) +
                       sizeof(double) * (size_t)loglik_mfma_slots(max_slots) * kLogR * kTile;
  const size_t reach = (size_t)(2 * n - 1) * tb * (16 / kp);  // mask fetches of internal ids
  return bytes > reach ? bytes : reach;
}
// The matrix-core kernel's tip bytes per pattern tile, in the layout its LDS wants
// ([taxon][column][register], the block padded to 8 bytes): built once per engine.
namespace {
__global__ __launch_bounds__(256) void tip_tiles_kernel(const uint8_t* masks, uint8_t* tiles, int n, int P, int ppr,
                                                        int tile_count, int block_bytes) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)tile_count * block_bytes;
  if (id >= total) return;
  const int tile = (int)(id / block_bytes), x = (int)(id - (long)tile * block_bytes);
  const int r = x & 3, c = (x >> 2) % ppr, taxon = (x >> 2) / ppr;
  uint8_t v = 0;
  if (taxon < n && r < kLogR) {
    const int p = tile * (kLogR * ppr) + r * ppr + c;
    v = masks[(size_t)taxon * P + (p < P ? p : P - 1)];
  }
  tiles[id] = v;
}
}  // namespace
size_t loglik_tip_tiles_bytes(int n, int P, int K) {
  const int ppr = 16 / (K == 1 ? 1 : (K == 2 ? 2 : 4));
  return (size_t)loglik_mfma_tiles(P, K) * (size_t)(((size_t)n * ppr * 4 + 7) / 8 * 8);
}
void launch_tip_tiles(const uint8_t* masks, uint8_t* tiles, int n, int P, int K, hipStream_t s) {
  static_assert(kLogR <= 4, "four bytes per (taxon, column)");
  const int ppr = 16 / (K == 1 ? 1 : (K == 2 ? 2 : 4));
  const int block_bytes = (int)(((size_t)n * ppr * 4 + 7) / 8 * 8);
  const int tile_count = loglik_mfma_tiles(P, K);
  const long total = (long)tile_count * block_bytes;
  hipLaunchKernelGGL(tip_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, masks, tiles, n, P, ppr,
                     tile_count, block_bytes);
}
int loglik_mfma_tiles(int P, int K) {
  const int kp = K == 1 ? 1 : (K == 2 ? 2 : 4);
  const int per_wave = kLogR * (16 / kp);
  return (P + per_wave - 1) / per_wave;
}
bool loglik_mfma_supported(const LikArgs& a, bool rescale) {
  // The matrix-core log-likelihood kernel needs tips in state-mask form (K > 4: the
  // categories are walked four at a time).  MI_PHYLO_LOGLIK_PATH=valu|mfma forces one of the two kernels.
  const int forced = [] {  // (read per call: tools/audit_paths.py switches it between engines)
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
  a.lds_slots = loglik_mfma_slots(max_slots);
  a.kp = a.K == 1 ? 1 : (a.K == 2 ? 2 : 4);
  // Evaluations per wave: the 16 finite-difference evaluations of a tree ([T, 17 T) of a GTR
  // gradient call, tree-major) share tips and schedule -- a wave takes as many of them in a
  // row as still leaves the device several rounds of waves (MI_PHYLO_LOGLIK_EVALS_PER_WAVE
  // forces 1, 2, 4 or 8; 16 measured the same as 8)
  const int tiles = loglik_mfma_tiles(a.P, a.K);
  int epw = 1;
  if (a.eval_offset >= a.map.T && a.eval_offset + count <= 17 * a.map.T &&
      (a.eval_offset - a.map.T) % 16 == 0 && count % 16 == 0) {
    static const int forced = getenv("MI_PHYLO_LOGLIK_EVALS_PER_WAVE") ? atoi(getenv("MI_PHYLO_LOGLIK_EVALS_PER_WAVE")) : 0;
    for (epw = 8; epw > 1; epw >>= 1)
      if (forced ? epw <= forced : (long)(count / epw) * tiles >= 3L * 256 * 16) break;
  }
  a.evals_per_wave = epw;
  const dim3 grid(tiles, count / epw), block(kTile);
  const size_t lds = loglik_mfma_lds_bytes(a.n, a.K, max_slots);
  auto go = [&](auto kernel) {
    allow_large_lds(reinterpret_cast<const void*>(kernel), lds);
    hipLaunchKernelGGL(kernel, grid, block, lds, s, a);
  };
  const bool multi = a.K > 4;
  if (rescale) {
    if (multi) go(loglik_mfma_kernel<kLogR, true, true>);
    else go(loglik_mfma_kernel<kLogR, true, false>);
  } else {
    if (multi) go(loglik_mfma_kernel<kLogR, false, true>);
    else go(loglik_mfma_kernel<kLogR, false, false>);
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
const char* loglik_kernel_name(const LikArgs& a, bool rescale, int max_slots) {
  return use_loglik_mfma(a, rescale, max_slots) ? "loglik_mfma_kernel" : "loglik_onchip_kernel";
}

}  // namespace miphylo
