// The matrix-core gradient walk, second generation (round 3): the same arithmetic as
// gradient_mfma_kernel (kernels_gradient.hip) -- half storage, macros,
// post-order then pre-order with positional edge sums -- fed by MACRO-ORDERED operand
// streams so that the walk itself does no address arithmetic on node ids:
//
//   * transition_macro_kernel writes, per gradient evaluation, the matrices in the order
//     the walk consumes them: [evaluation][macro][category group][position 0..5][4
//     categories][16]{f, tr}
//     (position = child0, child1, grand0..3; f = P[lo][hi], tr = P[hi][lo] for an internal
//     edge, (P Q)[lo][hi] for a tip edge -- one 16-byte load per lane and position in the
//     pre-order walk, one 8-byte load in the post-order walk), so a visit's six loads are a
//     running scalar base + constants;
//   * the tip state masks of a wave's columns sit in LDS by (macro, position) too: a
//     visit's six words are one ds_read_b128 + one ds_read_b64 at a running address;
//   * what is left of the schedule entry -- the shape word and the LDS slots -- comes in
//     through the scalar cache (s_load_dword / s_load_dwordx8, two visits ahead) straight
//     into scalar registers: no vector loads of the entry, no v_readfirstlane, no node ids
//     in the walk at all;
//   * a child's operand handling is selected by ONE switch on its configuration (tip /
//     stored / unstored with tip-or-stored grandchildren: six cases of straight-line code)
//     instead of a branch per operand;
//   * the edge sums of macro m land in the LDS bytes that held the tip words of macro m
//     (dead by then): 8 waves per CU as before;
//   * a wave takes several pattern tiles of its tree one after the other (the launch's first
//     evaluations; the last ones get a wave per tile and level out the end of the launch):
//     what belongs to the tree stays in registers, the next tile's tip bytes are requested
//     a whole walk ahead (launch_gradient_walk chooses the split).
// Every product and edge sum is the one the first-generation kernel does, in the same order;
// two reductions at the root (site likelihood over states / categories, log-likelihood
// partial) run on the matrix cores and row rotations instead of an LDS butterfly, so the
// two generations agree to the last bits, not bitwise
// (tests/test_gpu_parity.py::test_walk_kernels_agree).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "mi_phylo_device_utils.h"
#include "mi_phylo_kernels.h"

namespace miphylo {

#ifdef MI_WALK_TIMELINE
// diagnostic build (make timeline; tools/walk_timeline.py): eight words per wave
__device__ long long g_walk_timeline[65536 * 8];
#endif

namespace {
using namespace dev;

template <int SHIFT>
__device__ __forceinline__ double row_shr_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int slo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + SHIFT, 0xf, 0xf, true);
  const int shi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + SHIFT, 0xf, 0xf, true);
  return v + __hiloint2double(shi, slo);
}

// v[lane] += v[lane rotated right by SHIFT within its 16-lane row]
template <int SHIFT>
__device__ __forceinline__ double row_ror_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int slo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + SHIFT, 0xf, 0xf, true);
  const int shi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + SHIFT, 0xf, 0xf, true);
  return v + __hiloint2double(shi, slo);
}

// bytes per (macro, column) of the tip words in LDS: six words, padded to 32 (one ds_read_b128
// + one ds_read_b64; dense with three ds_read_b64 measured the same)
constexpr unsigned kTwCol = 32;
// ... and in the COMPACT form (fewer than three rate categories: a wave then has 8 or 16 pattern
// columns, and 32 bytes per column and macro cost waves per CU -- fluA, K = 1: 17 KB of tip
// words, 5 waves instead of 8): six 16-bit fields per column, three 4-bit masks (one per
// register r) each -- 12 bytes; a tip vector is then v_bfe at (16 (position & 1) + 4 r + hi) of
// word (position >> 1).  (Round 4; K < 3 engines ran the first-generation kernel before.)
constexpr unsigned kTwColCompact = 12;

// configuration of one child of a macro (from the shape word)
enum ChildCfg { kTip = 0, kStored = 1, kUss = 2, kUts = 3, kUst = 4, kUtt = 5 };
__device__ __forceinline__ int child_cfg(int sh, int j) {
  const int kind = (sh >> (2 * j)) & 3;
  const int tips = (sh >> (10 + 2 * j)) & 3;  // bit 0: first grandchild is a tip, bit 1: second
  return kind < 2 ? kind : 2 + tips;
}

template <int R, bool RESCALE, bool SUBST, bool ARENA, bool COMPACT>
__global__ __launch_bounds__(kTile, 2) void gradient_walk_kernel(LikArgs a) {
  static_assert(R <= 4, "tip masks of one column group are packed in one 32-bit word");
  static_assert(!COMPACT || R <= 4, "R 4-bit masks per 16-bit field");
  constexpr unsigned kCol = COMPACT ? kTwColCompact : kTwCol;
  extern __shared__ double wlds[];
  const int lane = threadIdx.x;
  const int hi = lane >> 4, b = (lane >> 2) & 3, lo = lane & 3;
#ifdef MI_WALK_STAMPS
  const long long st0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef MI_WALK_TIMELINE
  const long long tl0 = __builtin_amdgcn_s_memtime();
  const long long rt0 = __builtin_amdgcn_s_memrealtime();
  long long tl1 = 0, tl2 = 0;
#endif
  // This wave's jobs: pattern tiles first, first + step, ... of ONE evaluation.  The launch's
  // first walk_big_evals evaluations are walked by walk_groups waves each, every wave taking
  // several tiles one after the other -- everything that belongs to the tree stays, the next
  // tile's tip bytes are requested a whole walk ahead, and no wave slot stands empty between
  // two of its jobs; the evaluations after them get a wave per tile: the small jobs come last
  // (workgroups start in id order) and fill the end of the launch.
  int job_eval, job_first, job_step;
  {
    const int id = blockIdx.x, G = a.walk_groups, nb = a.walk_big_evals * G;
    if (id < nb) {
      const TileEval te = xcd_map(id, G, a.walk_big_evals);
      job_eval = te.eval;
      job_first = te.tile;
      job_step = G;
    } else {
      const TileEval te = xcd_map(id - nb, a.g_tiles, a.walk_evals - a.walk_big_evals);
      job_eval = a.walk_big_evals + te.eval;
      job_first = te.tile;
      job_step = a.g_tiles;  // (one job)
    }
  }
  const int e = a.eval_offset + job_eval;
  const int gi = a.grad_offset + job_eval;
  int t, mi;
  a.map.decode(e, t, mi);
  const DevModel* __restrict__ model = a.models + mi;
  const int K = a.K, n = a.n, Kp = a.kp;
  const int gtiles = a.g_tiles;  // pattern tiles x category groups
  const int groups = a.cat_groups, tiles_per_group = gtiles / groups;
  const int Mmax = max_macros(n);
  const MacroEntry* __restrict__ macros = a.macros + (size_t)t * macro_stride(n);
  const cint_ptr mw = as_const(reinterpret_cast<const int*>(macros));  // scalar loads
  // Tip staging starts here, before anything else of the prologue: this lane's (macro,
  // position) pairs j = lane, lane + 64 -- their node ids are the first link of the chain
  // node id -> tip bytes -> LDS, the longest latency of a wave's life (entries beyond the
  // tree's macro count hold no valid node: guarded below, they are inside the allocation)
  const int* mwv = reinterpret_cast<const int*>(macros);
  const int jmax = Mmax * 6;
  int node_j[2] = {-1, -1};
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int j = lane + 64 * u;
    if (j < jmax) node_j[u] = mwv[(j / 6) * 16 + 1 + (j % 6)];
  }
  const int M = __builtin_amdgcn_readfirstlane(a.macro_count[t]);
  if (ARENA) {
    const int need = __builtin_amdgcn_readfirstlane(a.slot_need[t]);
    if (need <= a.lds_lo || need > a.lds_slots) return;
  }
  if (M <= 0) return;
  const int ppr = 16 / Kp, TP = ppr * R;
  // four categories, twelve whole columns: the 12 tip bytes of a (macro, position) pair come
  // as three (unaligned) words; those of a wave's NEXT tile are requested when a tile starts
  struct __attribute__((packed)) Bytes12 {
    uint32_t d0, d1, d2;
  };
  auto tile_first_pattern = [&](int tile) { return (tile % tiles_per_group) * TP; };
  auto whole_words = [&](int tile) { return R == 3 && Kp == 4 && tile_first_pattern(tile) + 12 <= a.P; };
  auto request_bytes = [&](int tile, Bytes12 (&w)[2]) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int j = lane + 64 * u, node = node_j[u];
      if (j < jmax && (unsigned)node < (unsigned)n)
        w[u] = *reinterpret_cast<const Bytes12*>(a.tip_masks + (size_t)node * a.P + tile_first_pattern(tile));
    }
  };
  Bytes12 bytes_next[2] = {};
  bool have_next = whole_words(job_first);
  if (have_next) request_bytes(job_first, bytes_next);

  // (a wave that takes several tiles stays in ONE category group: the launcher gives every
  // tile its own wave when there are several groups)
  const int group = job_first / tiles_per_group;
  const int cat = 4 * group + b % Kp, pgrp = b / Kp;
  const int catc = cat < K ? cat : K - 1;
  // macro-ordered matrices of this gradient evaluation and category group: a position is
  // four categories x 16 {f, tr} pairs = 1 KB whatever K is (unused categories are never
  // read: a lane beyond K reads category K - 1, with weight zero), a visit 6 KB
  constexpr unsigned kPosBytes = 1024u, kVisitBytes = 6u * kPosBytes;
  const unsigned lane_moff = 16u * (unsigned)((catc - 4 * group) * 16 + lo * 4 + hi);
  const unsigned visit_stride = (unsigned)groups * kVisitBytes;
  const char* __restrict__ mm_g = reinterpret_cast<const char*>(a.mmats) +
                                  ((size_t)gi * Mmax * groups + group) * kVisitBytes;
  const char* __restrict__ ph_g =
      SUBST ? reinterpret_cast<const char*>(a.mphi) + ((size_t)gi * Mmax * groups + group) * (kVisitBytes / 2)
            : nullptr;
  const int col = pgrp * 4 + lo;  // this lane's pattern column; register r adds r * ppr
  const double pi_l = model->pi[hi];
  const double cw_l = cat < K ? model->cat_weight[cat] : 0.0;
  const double rate_l = model->cat_rate[catc], drate_l = model->cat_drate[catc];
  const double AQ = model->Q[lo * 4 + hi];  // A operand for Q L (same in every block)

  // ---- what a visit needs from memory, fetched a visit ahead ----
  struct Mats {
    double f[6], tr[6];
    double ph[SUBST ? 6 : 1];
  };
  struct Slots {  // scalars (s_load_dwordx8)
    int q, c[2], g[4], dst;
  };
  auto load_shape = [&](int m) { return mw[m * 16]; };
  auto load_slots = [&](int m) {
    const cint_ptr p = mw + m * 16 + 8;
    return Slots{p[0], {p[1], p[2]}, {p[3], p[4], p[5], p[6]}, p[7]};
  };
  const unsigned tw_lane = (unsigned)col * kCol;
  auto fetch = [&](int m, bool pre) {  // m: scalar
    Mats mt;
    // scalar base of the visit + this lane's 32-bit offset + constants (the offset is made
    // opaque so that it is not folded into a 64-bit per-lane pointer: that would cost two
    // vector instructions per load instead of none)
    // (positions 4 and 5 lie beyond the 4095-byte immediate: a second scalar base)
    const char* sb = mm_g + (size_t)((unsigned)m * visit_stride);
    unsigned off4 = 4 * kPosBytes;
    asm volatile("" : "+s"(off4));
    const char* sb4 = sb + off4;
    unsigned voff = lane_moff;
    asm volatile("" : "+v"(voff));
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const char* at = (j < 4 ? sb + j * kPosBytes : sb4 + (j - 4) * kPosBytes) + (size_t)voff;
      if (pre) {
        const double2 x = *reinterpret_cast<const double2*>(at);
        mt.f[j] = x.x;
        mt.tr[j] = x.y;
      } else {
        mt.f[j] = *reinterpret_cast<const double*>(at);
      }
    }
    if (pre && SUBST) {
      const char* sp = ph_g + (size_t)((unsigned)m * (visit_stride / 2));
      unsigned vph = lane_moff / 2;
      asm volatile("" : "+v"(vph));
#pragma unroll
      for (int j = 0; j < 6; j++)
        mt.ph[j] = *reinterpret_cast<const double*>(sp + j * (kPosBytes / 2) + (size_t)vph);
    }
    return mt;
  };
  // LDS: [macro][column][8 words: tip masks of positions 0..5, one byte per register r]
  // -- re-used, macro by macro, for that macro's edge sums [position][branch, site] once its
  // tip words are in registers -- | SUBST: four root sums | vectors [slot][r][lane] |
  // RESCALE: exponents
  const unsigned tstride = (unsigned)ppr * kCol;  // bytes per macro (>= 96: it also takes the macro's edge sums)
  char* const lds0 = reinterpret_cast<char*>(wlds);
  const unsigned tips_bytes = (unsigned)Mmax * tstride + (SUBST ? 32u : 0u);
  double* const xroot = reinterpret_cast<double*>(lds0 + (unsigned)Mmax * tstride);
  char* const plv = lds0 + tips_bytes;
  int16_t* exps = reinterpret_cast<int16_t*>(
      plv + (size_t)(ARENA ? a.lds_slots : max_stored(n)) * R * kTile * 8);
  // the first visit's scalars and matrices are on their way while the tip words are staged
  // (a round trip to L2 less on the prologue's chain; M <= 0: macro 0 is inside the allocation)
  const int M1 = M - 1;  // the root's macro is the last one; visits 0 .. M1 - 1 are stored nodes
  const int sh_first = load_shape(0), sh_second = load_shape(max(min(1, M1), 0));
  const Slots sl_first = load_slots(0);
  const Mats mt_first = fetch(0, false);

  for (int tile = job_first; tile < gtiles; tile += job_step) {
  const int tile_start = tile_first_pattern(tile);
  int pat[R], patc[R];
  double pw[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    pat[r] = tile_start + r * ppr + col;
    patc[r] = pat[r] < a.P ? pat[r] : a.P - 1;
    pw[r] = pat[r] < a.P ? a.weights[patc[r]] : 0.0;
  }
  char* const arena =
      ARENA ? reinterpret_cast<char*>(a.plv + ((size_t)job_eval * gtiles + tile) *
                                                   max_stored(n) * R * kTile)
            : nullptr;
  {
    // tip state masks of this wave's columns, by (macro, position): a lane takes the (macro,
    // position) pairs whose node is a tip and copies their TP bytes
    const int ppr_shift = Kp == 4 ? 2 : (Kp == 2 ? 3 : 4);
    auto stage_bytes = [&](int j, int node) {  // any layout, columns clamped to the last pattern
      const int m = j / 6, pos = j - m * 6;
      const uint8_t* src = a.tip_masks + (size_t)node * a.P;
      if (COMPACT) {
        // one 16-bit field per column: the masks of its R patterns (ppr apart), 4 bits each
        char* dst = lds0 + (unsigned)m * tstride + (unsigned)pos * 2u;
        if (R == 3 && Kp == 1 && tile_start + 48 <= a.P) {  // whole tile: its 48 bytes as twelve words
          uint32_t d[12];
#pragma unroll
          for (int i = 0; i < 12; i++) d[i] = *reinterpret_cast<const uint32_t*>(src + tile_start + 4 * i);
#pragma unroll
          for (int c = 0; c < 16; c++) {
            uint32_t f = 0;
#pragma unroll
            for (int r = 0; r < 3; r++) f |= ((d[4 * r + (c >> 2)] >> (8 * (c & 3))) & 0xfu) << (4 * r);
            *reinterpret_cast<uint16_t*>(dst + c * kTwColCompact) = (uint16_t)f;
          }
          return;
        }
        for (int c = 0; c < ppr; c++) {
          uint32_t f = 0;
#pragma unroll
          for (int r = 0; r < R; r++) {
            const int q = tile_start + r * ppr + c;
            f |= ((uint32_t)src[q < a.P ? q : a.P - 1] & 0xfu) << (4 * r);
          }
          *reinterpret_cast<uint16_t*>(dst + c * kTwColCompact) = (uint16_t)f;
        }
        return;
      }
      char* dst = lds0 + (unsigned)m * tstride + (unsigned)pos * 4u;
      for (int q = 0; q < TP; q++) {
        const int pp = tile_start + q < a.P ? tile_start + q : a.P - 1;
        dst[(q & (ppr - 1)) * kTwCol + (q >> ppr_shift)] = (char)src[pp];
      }
    };
    const bool whole = have_next;  // (requested for this tile when the previous one started)
    Bytes12 bytes_now[2] = {bytes_next[0], bytes_next[1]};
    {
      const int next_tile = tile + job_step;
      have_next = next_tile < gtiles && whole_words(next_tile);
      if (have_next) request_bytes(next_tile, bytes_next);
    }
    if (whole) {
      // regrouped into the four columns' words
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int j = lane + 64 * u, node = node_j[u];
        if (j < jmax && (unsigned)node < (unsigned)n) {
          const Bytes12 w = bytes_now[u];
          const int m = j / 6, pos = j - m * 6;
          char* dst = lds0 + (unsigned)m * tstride + (unsigned)pos * 4u;
#pragma unroll
          for (int c = 0; c < 4; c++)
            *reinterpret_cast<uint32_t*>(dst + c * kTwCol) =
                ((w.d0 >> (8 * c)) & 0xffu) | (((w.d1 >> (8 * c)) & 0xffu) << 8) |
                (((w.d2 >> (8 * c)) & 0xffu) << 16);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int j = lane + 64 * u, node = node_j[u];
        if (j < jmax && (unsigned)node < (unsigned)n) stage_bytes(j, node);
      }
    }
    for (int j = lane + 128; j < jmax; j += kTile) {  // larger trees: the rest
      const int node = mwv[(j / 6) * 16 + 1 + (j % 6)];
      if ((unsigned)node < (unsigned)n) stage_bytes(j, node);
    }
  }
  __syncthreads();
#ifdef MI_WALK_TIMELINE
  tl1 = __builtin_amdgcn_s_memtime();
#endif
#ifdef MI_WALK_STAMPS
  const long long st1 = __builtin_amdgcn_s_memtime();
#endif

  struct V {
    double v[R];
  };
  const unsigned lane8 = 8u * lane;
  // slot (a scalar) -> LDS address in ONE vector instruction: v_mad_u32_u24 with the stride
  // in a vector register (a literal stride would cost a scalar multiply on top of the add)
  unsigned slot_stride = R * kTile * 8;
  asm volatile("" : "+v"(slot_stride));
  const unsigned plv_lane = (unsigned)(plv - lds0) + lane8;
  auto slot_ptr = [&](int slot) {  // slot: scalar
    return reinterpret_cast<double*>(lds0 + (__umul24((unsigned)slot, slot_stride) + plv_lane));
  };
  auto load_slot = [&](int slot) {
    V x;
    const double* c = slot_ptr(slot);
#pragma unroll
    for (int r = 0; r < R; r++) x.v[r] = c[r * kTile];
    return x;
  };
  auto store_slot = [&](int slot, const V& x) {
    double* c = slot_ptr(slot);
#pragma unroll
    for (int r = 0; r < R; r++) c[r * kTile] = x.v[r];
  };
  auto store_arena = [&](int id, const V& x) {
    double* c = reinterpret_cast<double*>(arena + ((unsigned)id * (unsigned)(R * kTile * 8) + lane8));
#pragma unroll
    for (int r = 0; r < R; r++) c[r * kTile] = x.v[r];
  };
  auto arena_at = [&](int k) {
    V x;
    const double* c =
        reinterpret_cast<const double*>(arena + ((unsigned)k * (unsigned)(R * kTile * 8) + lane8));
#pragma unroll
    for (int r = 0; r < R; r++) x.v[r] = c[r * kTile];
    return x;
  };
  auto mm = [&](double A, const V& x) {
    V y;
#pragma unroll
    for (int r = 0; r < R; r++) y.v[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(A, x.v[r], 0.0, 0, 0, 0);
    return y;
  };
  auto mul = [&](const V& x, const V& y) {
    V z;
#pragma unroll
    for (int r = 0; r < R; r++) z.v[r] = x.v[r] * y.v[r];
    return z;
  };
  struct Tw {
    uint32_t w[6];  // (COMPACT: w[0..2], two positions per word)
  };
  // the 0/1 vector of the tip at position POS of a visit
  auto tipv = [&](const Tw& tw, auto pos_tag) {
    constexpr int POS = decltype(pos_tag)::value;
    V x;
#pragma unroll
    for (int r = 0; r < R; r++)
      x.v[r] = COMPACT ? (double)__builtin_amdgcn_ubfe(tw.w[POS >> 1], (uint32_t)(16 * (POS & 1) + 4 * r + hi), 1u)
                       : (double)__builtin_amdgcn_ubfe(tw.w[POS], (uint32_t)(8 * r + hi), 1u);
    return x;
  };

  auto fetch_tw = [&](int m) {  // the six tip words of visit m (LDS)
    Tw t;
    const char* twp = lds0 + ((unsigned)m * tstride + tw_lane);
    if (COMPACT) {
#pragma unroll
      for (int j = 0; j < 3; j++) t.w[j] = *reinterpret_cast<const uint32_t*>(twp + 4 * j);
      t.w[3] = t.w[4] = t.w[5] = 0;
    } else if (kTwCol == 32) {
      const uint4 w4 = *reinterpret_cast<const uint4*>(twp);
      const uint2 w2 = *reinterpret_cast<const uint2*>(twp + 16);
      t.w[0] = w4.x;
      t.w[1] = w4.y;
      t.w[2] = w4.z;
      t.w[3] = w4.w;
      t.w[4] = w2.x;
      t.w[5] = w2.y;
    } else {
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const uint2 w2 = *reinterpret_cast<const uint2*>(twp + 8 * j);
        t.w[2 * j] = w2.x;
        t.w[2 * j + 1] = w2.y;
      }
    }
    return t;
  };

  double qroot[R];  // root pre-order vector: pi * category weight * w_p / site likelihood
  int esum[R];      // RESCALE: exponents removed so far, per pattern
#pragma unroll
  for (int r = 0; r < R; r++) esum[r] = 0;

  // One child (J = 0, 1) of a visit: its vector L -- tip: expanded from its state masks;
  // stored: from its LDS slot; unstored: recomputed from its two children, which are tips or
  // stored nodes.  ONE decision tree on the shape word's bits (kind, then the two
  // grandchild tip flags), straight-line code at the leaves.  pre && ARENA: stored inputs
  // arrive from the arena (pa, pb), requested a visit ahead.
  struct Child {
    V L, xa, xb, Ap, Bp;
  };
  auto child_L = [&](int sh, auto jtag, const Mats& mt, const Tw& tw, const Slots& sl, Child& c,
                     bool pre, const V& pa, const V& pb) {
    constexpr int J = decltype(jtag)::value;
    const bool fa = ARENA && pre;
    const int kind = (sh >> (2 * J)) & 3;
    if (kind == 2) {
      if (sh & (1 << (10 + 2 * J))) c.xa = tipv(tw, std::integral_constant<int, 2 + 2 * J>{});
      else c.xa = fa ? pa : load_slot(sl.g[2 * J]);
      if (sh & (1 << (11 + 2 * J))) c.xb = tipv(tw, std::integral_constant<int, 3 + 2 * J>{});
      else c.xb = fa ? pb : load_slot(sl.g[2 * J + 1]);
      c.Ap = mm(mt.f[2 + 2 * J], c.xa);
      c.Bp = mm(mt.f[3 + 2 * J], c.xb);
      c.L = mul(c.Ap, c.Bp);
    } else if (kind == 1) {
      c.L = fa ? pa : load_slot(sl.c[J]);
    } else {
      c.L = tipv(tw, std::integral_constant<int, J>{});
    }
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;

  V pend_L;  // ARENA: the last stored vector, on its way to the arena
  int pend_dst = 0;
  bool pend = false;
  auto flush_arena = [&]() {
    if (ARENA && pend) {
      store_arena(pend_dst, pend_L);
      pend = false;
    }
  };

  // ================= post-order over the stored nodes, then the root (site likelihood) ====
  // A visit ends with the requests for the coming visits: tip words of the next one (LDS) and
  // the scalar loads (slots of the next visit, shape of the one after).  LDS and scalar loads
  // share one counter and a scalar load may return out of order: issued earlier they turn
  // every LDS wait of the visit into a wait for them too (measured: 3 % slower mid-visit).
  auto post_visit = [&](auto root_tag, int sh, const Slots& sl, const Mats& mt, const Tw& tw,
                        auto&& requests) {
    constexpr bool ROOT = decltype(root_tag)::value;
    Child c0, c1;
    const V none{};
    child_L(sh, J0{}, mt, tw, sl, c0, false, none, none);
    child_L(sh, J1{}, mt, tw, sl, c1, false, none, none);
    V Lv = mul(mm(mt.f[0], c0.L), mm(mt.f[1], c1.L));
    if (!ROOT) {
      if (RESCALE) {
#pragma unroll
        for (int r = 0; r < R; r++) {
          const double colsum = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, Lv.v[r], 0.0, 0, 0, 0);
          int ex = colsum > 0.0 ? __builtin_amdgcn_frexp_exp(colsum) : -4096;
          if (Kp >= 2) ex = max(ex, __shfl_xor(ex, 4, 64));
          if (Kp >= 4) ex = max(ex, __shfl_xor(ex, 8, 64));
          ex = ex == -4096 ? 0 : ex;
          Lv.v[r] = ldexp(Lv.v[r], -ex);
          esum[r] += ex;
          exps[(unsigned)(ARENA ? sl.dst : sl.q) * (unsigned)TP + (unsigned)(r * ppr + col)] = (int16_t)ex;
        }
      }
      store_slot(sl.q, Lv);
      if (ARENA) {
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
          const size_t at = ((size_t)a.grad_offset + job_eval) * a.tiles * kTile + patc[r];
          v = a.site_lik[at];
          if (RESCALE) v = ldexp(v, a.site_exp[at] - esum[r]);
        } else {
          v = cw_l * pi_l * Lv.v[r];
          // states: one product with a ones matrix leaves the column sums in every row; the
          // four categories of a pattern sit 4 lanes apart in a row: two row rotations
          // (no LDS round trips on this chain; sums in another order than the first
          // generation's butterfly: last-bit differences in the site likelihoods)
          v = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, v, 0.0, 0, 0, 0);
          if (Kp == 4) {
            v = row_ror_add<8>(v);
            v = row_ror_add<4>(v);
          } else if (Kp == 2) {
            v += __shfl_xor(v, 4, 64);
          }
        }
        sitev[r] = v;
      }
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
      // row r of the quotients to every row: a product with the selector matrix e_r e_r^T ...
      // (A[i][k] = [k == r] for all i: D[i][j] = quot[r][j], exact)
#pragma unroll
      for (int r = 0; r < R; r++)
        qroot[r] = pi_l * cw_l * __builtin_amdgcn_mfma_f64_4x4x4f64(hi == r ? 1.0 : 0.0, quot, 0.0, 0, 0, 0);
      double ll = 0.0;
      if (hi < R && (b % Kp) == 0 && pv < a.P)
        ll = wv * (RESCALE ? log(sv) + ev * 0.69314718055994530942 : log(sv));
      ll = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, ll, 0.0, 0, 0, 0);  // rows
      ll = row_ror_add<8>(ll);
      ll = row_ror_add<4>(ll);
      ll = row_ror_add<2>(ll);
      ll = row_ror_add<1>(ll);
      if (lane == 0 && groups == 1) a.ll_part[(size_t)e * a.ll_tiles + tile] = ll;
      if (SUBST) {
        double z = 0;
#pragma unroll
        for (int r = 0; r < R; r++) z = fma(qroot[r], Lv.v[r], z);
        z = z / pi_l;
        z = row_shr_add<8>(z);
        z = row_shr_add<4>(z);
        z = row_shr_add<2>(z);
        z = row_shr_add<1>(z);
        if ((lane & 15) == 15) xroot[hi] = z;
      }
    }
    requests(0);
    requests(1);
  };
  using Inner = std::false_type;
  using Root = std::true_type;
  {
    // Two visits per iteration, two register sets (A, B), nothing copied.  At the top of
    // visit m: shape, slots, matrices and tip words of m are there and the matrices of m + 1
    // are requested.
    int sha = sh_first, shb = sh_second;
    Slots sa = sl_first, sb;
    Mats ma = mt_first, mb;
    Tw ta = fetch_tw(0), tb;
    for (int m = 0; m < M1; m += 2) {
      mb = fetch(m + 1, false);
      flush_arena();
      int sh_next;
      post_visit(Inner{}, sha, sa, ma, ta, [&](int part) {
        if (part) {
          tb = fetch_tw(m + 1);
        } else {
          sb = load_slots(m + 1);
          sh_next = load_shape(min(m + 2, M1));
        }
      });
      sha = sh_next;
      if (m + 1 < M1) {
        ma = fetch(m + 2, false);
        flush_arena();
        post_visit(Inner{}, shb, sb, mb, tb, [&](int part) {
          if (part) {
            ta = fetch_tw(m + 2);
          } else {
            sa = load_slots(m + 2);
            sh_next = load_shape(min(m + 3, M1));
          }
        });
        shb = sh_next;
      }
    }
    flush_arena();
    if (M1 & 1) {  // the root's operands arrived in set B
      sha = shb;
      sa = sb;
      ma = mb;
      ta = tb;
    }
#ifdef MI_WALK_STAMPS
    const long long st2r = __builtin_amdgcn_s_memtime();
#endif
    post_visit(Root{}, sha, sa, ma, ta, [&](int) {});
#ifdef MI_WALK_STAMPS
    const long long st2 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && (job_eval % 250) == 3 && (tile % 39) == 5)
      printf("walk stamps eval %d tile %d: prologue %lld, post-order (inner) %lld, root %lld (s_memtime ticks, 100 MHz)\n",
             job_eval, tile, st1 - st0, st2r - st1, st2 - st2r);
#endif
  }

#ifdef MI_WALK_TIMELINE
  tl2 = __builtin_amdgcn_s_memtime();
#endif
  // ================= pre-order + edge derivatives =================
  const double coef_a = lo == 0 ? rate_l : (lo == 1 ? drate_l : 0.0);
  const double coef_b = lo == 2 ? rate_l : (lo == 3 ? drate_l : 0.0);
  const unsigned sum_lane = 8u * (unsigned)lo;
  auto edge_sums = [&](const V& na, const V& nb, int m, int pos_a) {
    double sa = na.v[0], sb = nb.v[0];
#pragma unroll
    for (int r = 1; r < R; r++) {
      sa += na.v[r];
      sb += nb.v[r];
    }
    double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(sa, coef_a, 0.0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(sb, coef_b, d1, 0, 0, 0);
    double red = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, d1, 0.0, 0, 0, 0);
    red = row_shr_add<4>(red);
    red = row_shr_add<8>(red);
    // lanes 12..15 (block 3 of row 0) hold branch a, site a, branch b, site b
    if (lane >= 12 && lane < 16)
      *reinterpret_cast<double*>(lds0 + ((unsigned)m * tstride + (unsigned)pos_a * 16u + sum_lane)) = red;
  };
  const double ident = hi == lo ? 1.0 : 0.0;
  auto blockT = [&](double x) { return __builtin_amdgcn_mfma_f64_4x4x4f64(x, ident, 0.0, 0, 0, 0); };
  const double AVt = SUBST ? model->V[hi * 4 + lo] : 0.0;
  const double AVi = SUBST ? model->Vinv[lo * 4 + hi] : 0.0;
  double Ht = 0.0;
  auto subst_stats = [&](const V& u, const V& Lc, double phi) {
    double G = 0.0;
#pragma unroll
    for (int r = 0; r < R; r++)
      G = __builtin_amdgcn_mfma_f64_4x4x4f64(blockT(u.v[r]), blockT(Lc.v[r]), G, 0, 0, 0);
    const double R1 = __builtin_amdgcn_mfma_f64_4x4x4f64(AVt, G, 0.0, 0, 0, 0);
    const double R2 = __builtin_amdgcn_mfma_f64_4x4x4f64(AVi, blockT(R1), 0.0, 0, 0, 0);
    Ht = fma(R2, phi, Ht);  // (explicit fma: both walk generations round alike)
  };
  // Edge of child c below a node with pre-order vector q and sibling product S (qs = q o S):
  //   internal child: q_c = P_c^T qs, numerator q_c o (Q L_c), q_c kept if stored
  //   tip child:      numerator qs o ((P_c Q) L_c)  -- `tr` is then (P_c Q)
  auto tip_edge = [&](double trm, const V& qs, const V& Lc, double phi) {
    if (SUBST) subst_stats(qs, Lc, phi);
    return mul(qs, mm(trm, Lc));
  };
  auto inner_edge = [&](double trm, const V& qs, const V& Lc, double phi, V& qc) {
    if (SUBST) subst_stats(qs, Lc, phi);
    qc = mm(trm, qs);
    return mul(qc, mm(AQ, Lc));
  };
  struct PreL {  // ARENA: stored inputs of a visit, in position order
    V x[4];
  };
  auto prefetch_L = [&](int sh) {
    PreL p;
    int k = (int)((unsigned)sh >> 16);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int cfg = child_cfg(sh, j);
      if (cfg == kStored) p.x[2 * j] = arena_at(k++);
      if (cfg == kUss || cfg == kUst) p.x[2 * j] = arena_at(k++);
      if (cfg == kUss || cfg == kUts) p.x[2 * j + 1] = arena_at(k++);
    }
    return p;
  };
  // the edges below child J: its own (numerator n) and, for an unstored child, its two
  // children's, whose four sums are reduced right here -- one decision tree per child again
  auto child_edges = [&](int sh, auto jtag, const Mats& mt, const Slots& sl, const Child& c,
                         const V& qs, int m, V& n) {
    constexpr int J = decltype(jtag)::value;
    const int kind = (sh >> (2 * J)) & 3;
    if (kind == 0) {
      n = tip_edge(mt.tr[J], qs, c.L, mt.ph[SUBST ? J : 0]);
      return;
    }
    V qc;
    n = inner_edge(mt.tr[J], qs, c.L, mt.ph[SUBST ? J : 0], qc);
    if (kind == 1) {
      store_slot(sl.c[J], qc);
      return;
    }
    V na, nb, qa, qb;
    const V qsa = mul(qc, c.Bp), qsb = mul(qc, c.Ap);
    if (sh & (1 << (10 + 2 * J))) {
      na = tip_edge(mt.tr[2 + 2 * J], qsa, c.xa, mt.ph[SUBST ? 2 + 2 * J : 0]);
    } else {
      na = inner_edge(mt.tr[2 + 2 * J], qsa, c.xa, mt.ph[SUBST ? 2 + 2 * J : 0], qa);
      store_slot(sl.g[2 * J], qa);
    }
    if (sh & (1 << (11 + 2 * J))) {
      nb = tip_edge(mt.tr[3 + 2 * J], qsb, c.xb, mt.ph[SUBST ? 3 + 2 * J : 0]);
    } else {
      nb = inner_edge(mt.tr[3 + 2 * J], qsb, c.xb, mt.ph[SUBST ? 3 + 2 * J : 0], qb);
      store_slot(sl.g[2 * J + 1], qb);
    }
    edge_sums(na, nb, m, 2 + 2 * J);
  };
  auto pre_visit = [&](auto root_tag, int sh, const Slots& sl, const Mats& mt, const Tw& tw, int m,
                       const PreL& pl, auto&& requests) {
    constexpr bool ROOT = decltype(root_tag)::value;
    V qv;
    if (ROOT) {
#pragma unroll
      for (int r = 0; r < R; r++) qv.v[r] = qroot[r];
    } else {
      qv = load_slot(sl.q);
      if (RESCALE) {
#pragma unroll
        for (int r = 0; r < R; r++)
          qv.v[r] = ldexp(qv.v[r], -(int)exps[(unsigned)(ARENA ? sl.dst : sl.q) * (unsigned)TP +
                                              (unsigned)(r * ppr + col)]);
      }
    }
    Child c0, c1;
    child_L(sh, J0{}, mt, tw, sl, c0, true, pl.x[0], pl.x[1]);
    child_L(sh, J1{}, mt, tw, sl, c1, true, pl.x[2], pl.x[3]);
    const V A = mm(mt.f[0], c0.L), B = mm(mt.f[1], c1.L);
    V n0, n1;
    child_edges(sh, J0{}, mt, sl, c0, mul(qv, B), m, n0);
    child_edges(sh, J1{}, mt, sl, c1, mul(qv, A), m, n1);
    edge_sums(n0, n1, m, 0);
    requests(0);
    requests(1);
  };
  {
    // the root's visit first (set A), then the stored nodes downwards, B and A in turn
    int sha = load_shape(M1), shb = load_shape(max(M1 - 1, 0));
    Slots sa = load_slots(M1), sb = load_slots(max(M1 - 1, 0));
    Mats ma = fetch(M1, true), mb = fetch(max(M1 - 1, 0), true);
    Tw ta = fetch_tw(M1), tb;
    PreL la, lb;
    if (ARENA) la = prefetch_L(sha);
    if (ARENA) lb = prefetch_L(shb);
    {
      int sh_next;
      pre_visit(Root{}, sha, sa, ma, ta, M1, la, [&](int part) {
        if (part) tb = fetch_tw(max(M1 - 1, 0));
        else sh_next = load_shape(max(M1 - 2, 0));
      });
      sha = sh_next;
    }
    for (int m = M1 - 1; m >= 0; m -= 2) {
      ma = fetch(max(m - 1, 0), true);
      if (ARENA) la = prefetch_L(sha);
      int sh_next;
      pre_visit(Inner{}, shb, sb, mb, tb, m, lb, [&](int part) {
        if (part) {
          ta = fetch_tw(max(m - 1, 0));
        } else {
          sa = load_slots(max(m - 1, 0));
          sh_next = load_shape(max(m - 2, 0));  // (set B's next visit)
        }
      });
      shb = sh_next;
      if (m >= 1) {
        mb = fetch(max(m - 2, 0), true);
        if (ARENA) lb = prefetch_L(shb);
        pre_visit(Inner{}, sha, sa, ma, ta, m - 1, la, [&](int part) {
          if (part) {
            tb = fetch_tw(max(m - 2, 0));
          } else {
            sb = load_slots(max(m - 2, 0));
            sh_next = load_shape(max(m - 3, 0));  // (set A's next visit)
          }
        });
        sha = sh_next;
      }
    }
  }
  __syncthreads();
  // positions that do not exist in a macro are never written nor read downstream
  const int gwidth = Mmax * kMacroPositions * 2 + (SUBST ? kSubstExtra : 0);
  double* gout = a.g_part + ((size_t)gi * a.g_tiles + tile) * gwidth;
  for (int i = lane; i < M * kMacroPositions * 2; i += kTile) {
    const int m = i / (kMacroPositions * 2), r = i - m * (kMacroPositions * 2);
    gout[i] = *reinterpret_cast<const double*>(lds0 + (unsigned)m * tstride + (unsigned)r * 8u);
  }
  if (SUBST) {
    gout[gwidth - kSubstExtra + lane] = Ht;
    if (lane < 4) gout[gwidth - 4 + lane] = xroot[lane];
  }
  // (one tile per wave in the arena and analytic-substitution variants, which have no
  // registers to spare for what would have to live from tile to tile: no back edge for them)
  if (ARENA || SUBST) break;
  __syncthreads();  // (the next tile's tip words go where these sums were read from)
  }  // tiles of this wave
#ifdef MI_WALK_TIMELINE
  {
    const long long tl3 = __builtin_amdgcn_s_memtime();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned id = blockIdx.x;
    if (lane == 0 && id < 65536u) {
      long long* o = g_walk_timeline + 8 * id;
      o[0] = tl0; o[1] = tl1; o[2] = tl2; o[3] = tl3; o[4] = hwid; o[5] = xcc; o[6] = rt0; o[7] = __builtin_amdgcn_s_memrealtime();
    }
  }
#endif
}

// ------------------------------------------------------------------------
// Transition matrices in the order the walk consumes them.  One thread per (gradient
// evaluation, macro, position, category): P = I + V expm1(L r t) V^-1 of the position's
// node (DESIGN.md "Accuracy"; negative entries clamped as BEAGLE does) and the matrix of the
// pre-order step -- P^T for an internal edge, P Q for a tip edge -- interleaved {f, tr} per
// lane slot (lo, hi): f = P[lo][hi], tr = P[hi][lo] | (P Q)[lo][hi].  Staged through LDS so
// that a block writes whole cache lines.  SUBST: the divided differences Phi[hi][lo] too.
// ------------------------------------------------------------------------
constexpr int kTmBlock = 128;
__global__ __launch_bounds__(kTmBlock) void transition_macro_kernel(TransitionMacroArgs a) {
  // One thread per (node, category) of ONE gradient evaluation (blockIdx.x): as many threads
  // as matrices, none idle on positions that do not exist.  Where a node's matrices go -- its
  // (macro, position) -- comes from a map the block builds from the tree's macro entries.
  // The 16-double matrices are staged one per thread (row stride 17: conflict-free) and
  // written out as the f halves, then the tr halves, of the 256-byte records.
  __shared__ double stage[kTmBlock * 17];
  __shared__ int rec_of[kTmBlock];
  extern __shared__ int slot_of[];  // [N - 1]: node -> macro * 6 + position
  const int Mmax = max_macros(a.n), groups = (a.K + 3) / 4;
  const int ge = blockIdx.x;  // gradient evaluation of this launch (grid.x: no 65535 limit)
  int t, mi;
  a.map.decode(a.eval_begin + ge, t, mi);
  const MacroEntry* mac = a.macros + (size_t)t * macro_stride(a.n);
  const int M = a.macro_count[t];
  for (int j = threadIdx.x; j < M * 6; j += kTmBlock) {
    const int m = j / 6, pos = j - m * 6;
    const MacroEntry& me = mac[m];
    if (pos < 2 || ((me.shape >> (2 * ((pos - 2) >> 1))) & 3) == 2)
      slot_of[pos < 2 ? me.child[pos] : me.grand[pos - 2]] = j;
  }
  __syncthreads();
  const int idx = blockIdx.y * kTmBlock + threadIdx.x;  // node * K + category
  const int node = idx / a.K, k = idx - node * a.K;
  const bool live = node < a.N - 1 && M > 0;
  double Pm[16];
  const DevModel& md = a.models[mi];
  bool tip = false;
  double tau = 0;
  if (live) {
    tip = node < a.n;
    tau = md.cat_rate[k] * a.bl_eff[(size_t)t * a.N + node];
    double ex[4], W[16];
    for (int x = 0; x < 4; x++) ex[x] = expm1(md.lambda[x] * tau);
    for (int x = 0; x < 4; x++)
      for (int j = 0; j < 4; j++) W[x * 4 + j] = ex[x] * md.Vinv[x * 4 + j];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        double sum = i == j ? 1.0 : 0.0;
        for (int x = 0; x < 4; x++) sum += md.V[i * 4 + x] * W[x * 4 + j];
        Pm[i * 4 + j] = sum > 0 ? sum : 0;
        stage[threadIdx.x * 17 + i * 4 + j] = Pm[i * 4 + j];
      }
    const int slot = slot_of[node], m = slot / 6, pos = slot - m * 6;
    // record index: (evaluation, macro, category group, position, category in the group)
    rec_of[threadIdx.x] = ((((ge * Mmax + m) * groups + (k >> 2)) * 6 + pos) << 2) + (k & 3);
  } else {
    rec_of[threadIdx.x] = -1;
  }
  __syncthreads();
  for (int x = threadIdx.x; x < kTmBlock * 16; x += kTmBlock) {
    const int rec = rec_of[x >> 4];
    if (rec >= 0) a.mmats[(size_t)rec * 32 + (x & 15) * 2] = stage[(x >> 4) * 17 + (x & 15)];
  }
  __syncthreads();
  if (live) {
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        double trv;
        if (tip) {
          trv = 0;
          for (int x = 0; x < 4; x++) trv += Pm[i * 4 + x] * md.Q[x * 4 + j];
        } else {
          trv = Pm[j * 4 + i];
        }
        stage[threadIdx.x * 17 + i * 4 + j] = trv;
      }
  }
  __syncthreads();
  for (int x = threadIdx.x; x < kTmBlock * 16; x += kTmBlock) {
    const int rec = rec_of[x >> 4];
    if (rec >= 0) a.mmats[(size_t)rec * 32 + (x & 15) * 2 + 1] = stage[(x >> 4) * 17 + (x & 15)];
  }
  if (a.mphi != nullptr) {
    __syncthreads();
    if (live) {
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)  // slot (lo = i, hi = j) holds Phi[hi][lo]
          stage[threadIdx.x * 17 + i * 4 + j] = phi_divided_difference(md.lambda[j], md.lambda[i], tau);
    }
    __syncthreads();
    for (int x = threadIdx.x; x < kTmBlock * 16; x += kTmBlock) {
      const int rec = rec_of[x >> 4];
      if (rec >= 0) a.mphi[(size_t)rec * 16 + (x & 15)] = stage[(x >> 4) * 17 + (x & 15)];
    }
  }
}

}  // namespace

size_t gradient_walk_lds_bytes_for(int n, int K, bool rescale, bool subst, int slots, int regs) {
  const int kp = K == 1 ? 1 : (K == 2 ? 2 : 4);
  const int R = regs > 0 ? regs : kLlR;
  // (compact tip words for fewer than three categories -- not in the analytic variant)
  const unsigned col = (kp < 4 && !subst) ? kTwColCompact : kTwCol;
  size_t bytes = (size_t)max_macros(n) * (16 / kp) * col + (subst ? 32 : 0) +
                 sizeof(double) * (size_t)slots * R * kTile;
  if (rescale) bytes += ((sizeof(int16_t) * (size_t)max_stored(n) * R * (16 / kp) + 7) / 8) * 8;
  return bytes;
}
size_t gradient_walk_lds_bytes(int n, int K, bool rescale, bool subst, int regs) {
  return gradient_walk_lds_bytes_for(n, K, rescale, subst, max_stored(n), regs);
}
size_t gradient_walk_mats_bytes_per_eval(int n, int K) {
  return (size_t)max_macros(n) * ((K + 3) / 4) * 24 * 32 * sizeof(double);
}

void launch_transition_macro(const TransitionMacroArgs& a, hipStream_t s) {
  if (a.count <= 0) return;
  const int per_eval = (a.N - 1) * a.K;
  const dim3 grid(a.count, (per_eval + kTmBlock - 1) / kTmBlock);
  hipLaunchKernelGGL(transition_macro_kernel, grid, dim3(kTmBlock), sizeof(int) * (size_t)(a.N - 1), s, a);
}

template <bool RESCALE, bool SUBST, bool ARENA, bool COMPACT>
static void launch_walk_variant(const LikArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  allow_large_lds(reinterpret_cast<const void*>(gradient_walk_kernel<kLlR, RESCALE, SUBST, ARENA, COMPACT>), lds);
  hipLaunchKernelGGL((gradient_walk_kernel<kLlR, RESCALE, SUBST, ARENA, COMPACT>), grid, dim3(kTile), lds, s, a);
}
template <bool ARENA>
static void launch_walk_store(const LikArgs& a, dim3 grid, size_t lds, bool rescale, bool subst,
                              hipStream_t s) {
  const bool compact = a.kp < 4 && !subst;  // (as gradient_walk_lds_bytes_for sizes the tip words)
  if (rescale && subst) launch_walk_variant<true, true, ARENA, false>(a, grid, lds, s);
  else if (subst) launch_walk_variant<false, true, ARENA, false>(a, grid, lds, s);
  else if (rescale && compact) launch_walk_variant<true, false, ARENA, true>(a, grid, lds, s);
  else if (rescale) launch_walk_variant<true, false, ARENA, false>(a, grid, lds, s);
  else if (compact) launch_walk_variant<false, false, ARENA, true>(a, grid, lds, s);
  else launch_walk_variant<false, false, ARENA, false>(a, grid, lds, s);
}
void launch_gradient_walk(const LikArgs& a_in, int count, bool rescale, bool subst, hipStream_t s) {
  if (count <= 0) return;
  LikArgs a = a_in;
  a.kp = a.K == 1 ? 1 : (a.K == 2 ? 2 : 4);
  a.cat_groups = gradient_mfma_groups(a.K);
  // Jobs.  The launch's first `big` evaluations (a multiple of 8: whole XCD groups) are
  // walked by waves that take `tpw` pattern tiles each, one after the other (what belongs to
  // the tree stays in registers, the next tile's tip bytes arrive during the current walk, no
  // wave slot stands empty in between: -6 % per 1000 DS1 trees); the evaluations after them
  // get a wave per tile.  Workgroups start in id order, so the small jobs come last and level
  // out the end of the launch: the slots finish their big jobs up to one big job apart, which
  // takes about (tpw / 2 + 1) rounds of small jobs to fill.  tpw maximises the share of the
  // work that runs as the second or later tile of a wave.  (A wave stays in one category
  // group; the arena and analytic variants take one tile per wave: kernel comment.)
  // MI_PHYLO_WALK_TILES_PER_WAVE=k forces k (1: every tile its own wave, as until round 3).
  static const int forced_tpw = [] {
    const char* env = getenv("MI_PHYLO_WALK_TILES_PER_WAVE");
    return env ? std::max(1, atoi(env)) : 0;
  }();
  const int gtiles = gradient_mfma_tiles(a.P, a.K) * a.cat_groups;
  const bool arena_variant =
      a.store ? a.store == 2
              : gradient_walk_use_arena(a.n, a.K, rescale, subst, (size_t)gtiles * (size_t)count);
  int tpw = 1, big = 0;
  if (a.cat_groups == 1 && !arena_variant && !subst) {
    const double slots = (double)device_compute_units() * gradient_walk_waves_per_cu(a.n, a.K);
    double best = 0;
    for (int k = forced_tpw ? forced_tpw : 2; k <= (forced_tpw ? forced_tpw : 8); k++) {
      // (measured, 1000 and 125 DS1 trees: fewer small jobs -- k / 4 + 1, k / 8 + 1/2 rounds --
      // lose more at the end of the launch than the larger share of big jobs gains)
      const int small = (int)std::ceil((0.5 * k + 1.0) * slots / gtiles);
      const int b = small < count ? (count - small) & ~7 : 0;
      const double gain = (double)b / count * (1.0 - 1.0 / k);
      if (gain > best) {
        best = gain;
        tpw = k;
        big = b;
      }
    }
  }
  a.walk_evals = count;
  a.walk_groups = (gtiles + tpw - 1) / tpw;
  a.walk_big_evals = big;
  const dim3 grid((unsigned)((size_t)a.walk_big_evals * a.walk_groups +
                             (size_t)(count - a.walk_big_evals) * gtiles));
  if (arena_variant) {
    const int usual = gradient_arena_slots_usual(a.n), sure = gradient_arena_slots_sure(a.n);
    a.lds_lo = -1;
    if (arena_single_launch(gradient_walk_lds_bytes_for(a.n, a.K, rescale, subst, sure), grid.x)) {
      // (few waves: one launch with the larger footprint takes every tree)
      a.lds_slots = sure;
      launch_walk_store<true>(a, grid, gradient_walk_lds_bytes_for(a.n, a.K, rescale, subst, sure),
                              rescale, subst, s);
      return;
    }
    a.lds_slots = usual;
    launch_walk_store<true>(a, grid, gradient_walk_lds_bytes_for(a.n, a.K, rescale, subst, usual),
                            rescale, subst, s);
    if (sure > usual) {
      a.lds_lo = usual;
      a.lds_slots = sure;
      launch_walk_store<true>(a, grid, gradient_walk_lds_bytes_for(a.n, a.K, rescale, subst, sure),
                              rescale, subst, s);
    }
    return;
  }
  launch_walk_store<false>(a, grid, gradient_walk_lds_bytes(a.n, a.K, rescale, subst), rescale,
                           subst, s);
}

bool gradient_walk_use_arena(int n, int K, bool rescale, bool subst, size_t waves, bool lut, int regs) {
  // (read at every call: tools/audit_paths.py switches it between engines of one process)
  const int forced = [] {
    const char* env = getenv("MI_PHYLO_GRADIENT_STORE");
    if (!env) return 0;
    return std::string(env) == "arena" ? 2 : (std::string(env) == "lds" ? 1 : 0);
  }();
  if (regs > kLlR) {
    // A wide-tile engine (look-up walk, kernels_walk3.hip): wide tiles pay in the arena; the form
    // with every vector in LDS runs at one wave per SIMD (registers) and takes the calls whose
    // waves are all resident at once at that occupancy -- 16 trees of 45 taxa x 200 patterns
    // 0.052 (default tiles in LDS) / 0.068 ms (wide, arena) before it existed.  The tile width
    // never depends on the call: a tree's outputs do not depend on the batch it came in.
    const size_t lds_w = gradient_walk_lds_bytes(n, K, rescale, subst, regs);
    const bool fits = lds_w <= 160 * 1024;
    if (forced == 1 && fits) return false;
    if (forced == 2 || !fits) return true;
    const size_t per_cu = std::min<size_t>(4, (160 * 1024) / lds_w);
    return waves > (size_t)device_compute_units() * per_cu;
  }
  const size_t lds_all = gradient_walk_lds_bytes(n, K, rescale, subst);
  const bool lds_fits = lds_all <= 160 * 1024;
  if (forced == 1 && lds_fits) return false;
  if (forced == 2) return true;
  if (lds_fits && arena_single_launch(lds_all, waves)) return false;  // a call of a few trees
  // (round 5, tools/audit_paths.py with the store forced either way: with five and six waves per
  // CU the LDS store still wins -- 31 taxa x 1000 patterns x 4 categories 1.29 against 1.54 ms per
  // 1000 trees, 36 x 200: 0.41 / 0.46, one category 0.36 / 0.41 -- and from four waves down the
  // arena does; the round-1 cross-over "fewer than seven" dated from the first generation)
  // (round 6: the look-up walk's arena variant -- stored vectors back from the arena in 16-byte
  // accesses, requested with the operands -- wins one step earlier: at five waves per CU the
  // arena's eight take 36 taxa x 1812 patterns in 2.68 against 2.87 ms per 1000 trees, 41 x 1137
  // in 1.96 against 2.07; at six -- 29 to 35 taxa -- the LDS store keeps 1.21 against 1.38)
  return !lds_fits || (160 * 1024) / lds_all < (lut ? 6 : 5);
}
// the same rule for a large batch with no store forced: does this engine's tree size take the arena?
bool gradient_walk_batches_take_arena(int n, int K, bool lut) {
  const size_t lds_all = gradient_walk_lds_bytes(n, K, false, false);
  return lds_all > 160 * 1024 || (160 * 1024) / lds_all < (size_t)(lut ? 6 : 5);
}
int gradient_walk_waves_per_cu(int n, int K);
bool gradient_walk_fits(int n, int K, bool rescale) {
  if (n < 3 || K > kMaxCategories) return false;
  if (gradient_walk_lds_bytes(n, K, rescale, true) <= 160 * 1024) return true;
  return gradient_walk_lds_bytes_for(n, K, rescale, true, gradient_arena_slots_sure(n)) <= 160 * 1024;
}
// waves per CU the kernel's LDS footprint allows (the registers allow 8)
int gradient_walk_waves_per_cu(int n, int K) {
  const size_t lds = gradient_walk_use_arena(n, K, false, false, (size_t)-1, false)
                         ? gradient_walk_lds_bytes_for(n, K, false, false, gradient_arena_slots_usual(n))
                         : gradient_walk_lds_bytes(n, K, false, false);
  return (int)std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1));
}
const char* gradient_walk_kernel_name() { return "gradient_walk_kernel"; }

}  // namespace miphylo

#ifdef MI_WALK_TIMELINE
extern "C" __attribute__((visibility("default"))) int mi_debug_walk_timeline(long long* out, int waves) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(miphylo::g_walk_timeline), (size_t)waves * 64);
}
#endif
