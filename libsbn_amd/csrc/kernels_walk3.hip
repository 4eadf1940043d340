// The matrix-core gradient walk, third generation (round 4): TIP CHILDREN COST NO MATRIX
// PRODUCT.  Same walk as gradient_walk_kernel (kernels_walk.hip; DESIGN.md 4.1: half storage,
// macros, post-order then pre-order, positional edge sums, macro-ordered operand streams,
// scalar schedule descriptors) -- but where that kernel expands a tip's state masks to a 0/1
// vector (v_bfe + v_cvt per register) and multiplies it by the edge's matrix on the matrix
// cores, this one LOOKS THE PRODUCT UP: P e_s is column s of P, so transition_lut_kernel
// writes, for a tip position, the table [category][row][state 0..3, gap]{P, P Q} (gap: the row
// sums) and a lane fetches its entries with one gather per register -- a global load at
// (scalar base of the visit) + (lane constant + the state's byte offset, ONE vector add).  In a
// 27-taxon tree 243 of the 628 matrix instructions of a tile job and 324 of its vector
// instructions were tip work (three products per tip: post-order, sibling message, derivative).
//
//   * records: [gradient evaluation][macro][position 0..5] x 1280 bytes;
//       internal position: [4 categories][16 slots (lo, hi)]{f = P[lo][hi], tr = P[hi][lo]}
//                          (bytes 0..1023, as in the second generation)
//       tip position:      [4 categories][4 rows hi][5 states]{P[hi][s], (P Q)[hi][s]}, state 4 =
//                          gap: {sum_s P[hi][s], sum_s (P Q)[hi][s]}   (320 bytes per category)
//   * tip codes: one byte per (taxon, pattern) = 16 x state (0, 16, 32, 48; 64 for a gap) -- the
//     byte offset of the state's {P, P Q} pair in its row; staged in LDS by (macro, position,
//     column) exactly as the second generation stages its mask bytes;
//   * a visit's operands are requested a visit AHEAD (into the other of two register sets):
//     the gathers need the NEXT visit's tip words and shape, so tip words are read two visits
//     ahead (right after the gathers are issued) and shapes three.
// Every other product, edge sum and reduction is the second generation's, in the same order;
// a looked-up product is the very number the matrix instruction forms (one term; a gap's row
// sum is added in index order by the table builder as by the instruction): results are
// BIT-IDENTICAL to the second generation's (tests/test_gpu_parity.py::test_walk_kernels_agree).
//
// Scope (round 6: widened from "three or four categories, stored vectors in LDS"): engines
// whose tip vectors are the five of SitePattern (one-hot / all ones: src/site_pattern.cpp:117-131
// -- everything the reference produces), one to four rate categories (one category group; KP =
// categories per matrix instruction), stored vectors in LDS or -- batches of trees of 36 taxa and
// more -- in a per-wave HBM arena (ARENA), no analytic substitution gradient.  The second
// generation keeps: more than four categories, the analytic gradient, 0/1 tip vectors that are
// not one-hot, and one-category engines whose vectors fit LDS (its waves take several tiles in a
// row there).  The first generation (gradient_mfma_kernel) was retired in round 6.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "mi_phylo_device_utils.h"
#include "mi_phylo_kernels.h"
#include "mi_phylo_setup_device.h"

namespace miphylo {
namespace {
using namespace dev;

template <int SHIFT>
__device__ __forceinline__ double row_shr_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int slo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + SHIFT, 0xf, 0xf, true);
  const int shi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + SHIFT, 0xf, 0xf, true);
  return v + __hiloint2double(shi, slo);
}
template <int SHIFT>
__device__ __forceinline__ double row_ror_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int slo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + SHIFT, 0xf, 0xf, true);
  const int shi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + SHIFT, 0xf, 0xf, true);
  return v + __hiloint2double(shi, slo);
}

constexpr int kRegs = kLlR;              // registers (16 columns each) per vector: the default tile width
constexpr int kRegsWide = 4;             // the arena variant's wide tile (round 6: walk_lut_body, RR)
// Tip codes in LDS, by (macro, column).  Four categories per instruction (KP = 4: three or four
// rate categories, four pattern columns per register): six words per column -- one per position,
// one byte per register: the byte offset of the state's table entry -- padded to 32 bytes.
// Fewer (KP = 1, 2: a register then holds 16 / 8 pattern columns, and 32 bytes per column and
// macro would cost waves per CU -- fluA, K = 1: 17 KB of them): the COMPACT form of the second
// generation, six 16-bit fields per column with a 4-bit state number (0..3, 4 = gap) per register;
// the byte offset is then one v_bfe + one v_lshl_add instead of one SDWA add.
constexpr unsigned kTwCol = 32, kTwColCompact = 12;
constexpr unsigned kPos = 1280u, kVisit = 6u * kPos;
constexpr unsigned kTipCat = 320u, kTipRow = 80u;
typedef double dbl2 __attribute__((ext_vector_type(2)));

// operands of one child of a visit (see fetch_child)
// (RR registers per vector: a tip position takes RR W doubles, W = 1 post-order / 2 pre-order;
// the child itself, or its first grandchild, and its second grandchild: (2 RR + 1) W)
template <bool PRE, int RR>
struct Ops {
  double x[(2 * RR + 1) * (PRE ? 2 : 1)];
};

// Hand-off word of the one-launch small call (round 5), one per tree: the set-up waves of a
// tree add kReadyQuarter each when their share of the tree's operand records (wave 0 also the
// macro list and the model instance) has been written through to memory, wave 0 adds the macro
// count in the low byte; the walk waves of the tree wait for kSetupQuarters of them.
constexpr int kSetupQuarters = 4, kReadyQuarter = 1 << 8;

// block: this wave's number among the walk waves of the launch.
//   FUSED: `ready` is the hand-off word array (above), the launch's first workgroups are set-up
//          waves; `spin_ticks`: how long a walk wave polls (100 MHz ticks) before it gives up.
//   ARENA (round 6): trees of more than ~45 taxa -- the stored post-order vectors go to a per-wave
//          HBM arena as well as to a handful of recycled LDS slots (schedule of macro_slots_kernel:
//          Sethi-Ullman order, the arena index of a node in its `pad` word, of a macro's first
//          stored input in the upper half of its shape word); the pre-order walk takes them back
//          from the arena, requested a visit ahead WITH the visit's operands and INTO the operand
//          group of the child they belong to (a stored input's group has room: see fetch_child),
//          and recycles the LDS slots for its own vectors.  Same products in the same order as
//          with every vector in LDS: bit-identical.
//   KP:    rate categories per matrix instruction (4: K = 3, 4; 2: K = 2; 1: K = 1); a register
//          holds 16 / KP pattern columns.
//   NT:    ARENA only -- the arena is written and read with non-temporal accesses.  A vector is
//          written once and read once; whether it should stay in the XCD's L2 in between depends
//          on what else wants that L2 (measured, round 6, ms per 1000 trees, plain / non-temporal):
//          with few pattern tiles per tree many trees are in flight per XCD and their operand
//          records -- shared by all waves of a tree -- are what the L2 should hold: 50 taxa x 378
//          patterns (32 tiles) 0.983 / 0.937, fluA-sized 69 x 238 x 1 category (5 tiles) 0.271 /
//          0.258; with many tiles per tree the records fit anyway and the pre-order walk finds
//          the vectors written last still in the L2 (it reads them in reverse order): 36 x 1812
//          (151 tiles) 2.72 / 2.82, 59 x 1824 4.50 / 4.71, 64 x 1008 (84) 2.81 / 2.89; 100 x 500
//          (42) 2.40 / 2.40.  The launcher takes NT up to 48 tiles per tree.
//   RR:    registers per vector = the tile width, 16 / KP pattern columns each.  Three by default
//          (MI_LLR); FOUR (ARENA only) for engines whose tile counts say so (round 6).  With
//          the vectors in LDS a wider tile costs waves per CU (DS1: 0.81 -> 0.84 ms); the arena
//          variant's LDS holds a handful of recycled slots only, and a visit's fixed costs -- the
//          operand requests, the arena round trip, the schedule words -- are spread over a third
//          more columns: 36 taxa x 1812 patterns x 4 categories 2.72 -> 2.34 ms per 1000 trees,
//          64 x 1008 2.74 -> 2.44, 59 x 1824 4.37 -> 3.85, 50 x 378 0.922 -> 0.819, two categories
//          64 x 1008 1.42 -> 1.29, one 36 x 1812 0.776 -> 0.702 (profiles/r06_wide_tiles.txt; the
//          rule: gradient_walk_tile_regs).  248 registers, no scratch; with every vector in LDS
//          the wide kernel needs ~280, so that form is built for one wave per SIMD and a
//          wide-tile engine takes the arena unless the call's waves are all resident at that
//          occupancy (gradient_walk_use_arena).
template <bool RESCALE, bool FUSED, bool ARENA, int KP, bool NT = false, int RR = kRegs>
__device__ __forceinline__ void walk_lut_body(const LikArgs& a, double* wlds, const int block,
                                              const int32_t* ready, const int spin_ticks, const int fence) {
  constexpr int R = RR;
  constexpr unsigned kVecBytes = R * kTile * 8;  // one stored vector of a wave (LDS slot, arena entry)
  static_assert(R >= 1 && R <= 4, "a tip word holds one byte / one 4-bit field per register");
  static_assert(KP == 1 || KP == 2 || KP == 4, "categories per matrix instruction");
  static_assert(!(FUSED && ARENA), "the one-launch call keeps its stored vectors in LDS");
  constexpr bool COMPACT = KP < 4;
  constexpr int ppr = 16 / KP, TP = ppr * R;
  constexpr unsigned kCol = COMPACT ? kTwColCompact : kTwCol;
  constexpr unsigned kTStride = (unsigned)ppr * kCol;  // per macro (>= 96: it also takes the macro's edge sums)
  static_assert(kTStride >= 96, "a macro's edge sums go where its tip words were");
  const int lane = threadIdx.x;
  const int hi = lane >> 4, b = (lane >> 2) & 3, lo = lane & 3;
  // one tile per wave (several tiles per wave, as the second generation has them, were
  // measured here: what must live from tile to tile pushes the 216 registers of this kernel
  // past 256 -- 176 B of scratch per lane, 0.78 -> 0.95 ms)
  const TileEval te = xcd_map(block, a.g_tiles, a.walk_evals);
  const int job_eval = te.eval, tile = te.tile;
  const int e = a.eval_offset + job_eval;
  int gi = a.grad_offset + job_eval;
  int t, mi;
  a.map.decode(e, t, mi);
  int M_ready = 0;
#ifdef W3_SETUP_STAMPS
  const long long w_in = __builtin_amdgcn_s_memrealtime();
#endif
  if (FUSED) {
    // Wait for this tree's set-up waves: poll the tree's word (relaxed load at agent scope: served
    // by the memory side, never by this CU's L1), bounded by WALL-CLOCK time (s_memrealtime,
    // 100 MHz: a count of polls would shrink under a profiler or pre-emption).  What makes the
    // plain loads behind the poll see the handed-over bytes -- `fence` (FusedSetupArgs::fence):
    //   1 (default): everything handed over lies in cache lines of its own per tree
    //      (macro_stride, kVisit, alignas(128) DevModel) that no wave of this launch reads before
    //      the word says so (control dependency on the polled value + the redefinition of the
    //      indices below), was stored write-through (sc1) and waited for (vmcnt(0)) before the
    //      word was added to, and the caches held nothing of these lines when the kernel started
    //      (the dispatch packet's acquire); on top of that this wave invalidates its CU's vector
    //      L1 (buffer_inv sc0) and the scalar cache (s_dcache_inv) behind the poll, so that a
    //      line touched early by a wave that timed out cannot be served stale from them, and the
    //      set-up waves of a tree run on the XCD of its walk waves (fused_setup_role), whose L2
    //      is the one coherence point of both;
    //   2 (MI_PHYLO_FUSED_FENCE=agent): a formal release / acquire pair at agent scope
    //       -- on gfx950 that is buffer_wbl2 sc1 in every set-up wave and buffer_inv sc1
    //      in every walk wave: a write-back / invalidate of the XCD's WHOLE L2 per wave (the
    //      eight L2s of the chip are not coherent with each other), measured +42 % on the
    //      125-tree step (0.1326 -> 0.188 ms, round 6) -- which is why it is not the default;
    //   0 (MI_PHYLO_FUSED_FENCE=none): the round-5 form, the argument of (1) alone.
    int v = 0;
    const long long t_in = __builtin_amdgcn_s_memrealtime();
    for (;;) {
      v = __builtin_amdgcn_readfirstlane(
          __hip_atomic_load(ready + (size_t)t * kReadyStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if ((v >> 8) >= kSetupQuarters) break;
      if ((long long)__builtin_amdgcn_s_memrealtime() - t_in > (long long)spin_ticks) break;
      __builtin_amdgcn_s_sleep(8);
    }
    if (fence == 2) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_dcache_inv" ::: "memory");
    } else if (fence == 1) {
      asm volatile("buffer_inv sc0\n\ts_dcache_inv" ::: "memory");
    }
    // (-1: waited in vain.  The wave leaves through the one exit below -- a second `return`
    // up here, with its status store, changed how the WHOLE walk is compiled: 224 registers
    // instead of 214 and a wait in front of single operand loads, +34 % time per 1000 trees)
    M_ready = (v >> 8) < kSetupQuarters ? -1 : (v & 0xff);
    // No load of the handed-over data may move above the poll -- the scalar loads of the
    // macro list are loads from the constant address space, which the compiler may hoist over
    // anything: the indices every such address is formed from are redefined HERE.
    asm volatile("" : "+s"(t), "+s"(mi), "+s"(gi) : : "memory");
  }
#ifdef W3_SETUP_STAMPS
  if (lane == 0 && (block % 1499) == 0)
    printf("walk block %d (tree %d tile %d): in %lld polled %lld\n", block, t, tile, w_in, (long long)__builtin_amdgcn_s_memrealtime());
#endif
  const DevModel* __restrict__ model = a.models + mi;
  const int K = a.K, n = a.n;
  const int Mmax = max_macros(n);
  const MacroEntry* __restrict__ macros = a.macros + (size_t)t * macro_stride(n);
  const cint_ptr mw = as_const(reinterpret_cast<const int*>(macros));  // scalar loads
  // tip staging starts here (node ids of this lane's (macro, position) pairs): the first link
  // of the chain node id -> tip bytes -> LDS, the longest latency of a wave's life
  const int* mwv = reinterpret_cast<const int*>(macros);
  const int jmax = Mmax * 6;
  // (pairs per lane whose node id and tip bytes are requested up front: two cover 42 taxa; the
  // arena variant -- larger trees -- takes four, 86 taxa: round 6, until then a lane's third
  // pair went through two more dependent round trips and byte-by-byte copies; +0.2-0.5 % on
  // the 50-64-taxon shapes)
  constexpr int U = ARENA ? 4 : 2;
  int node_j[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int j = lane + 64 * u;
    node_j[u] = -1;
    if (j < jmax) node_j[u] = mwv[(j / 6) * 16 + 1 + (j % 6)];
  }
  const int M = FUSED ? M_ready : __builtin_amdgcn_readfirstlane(a.macro_count[t]);
  if (ARENA) {  // (two launches over one grid: this one takes the trees whose schedule fits its LDS slots)
    const int need = __builtin_amdgcn_readfirstlane(a.slot_need[t]);
    if (need <= a.lds_lo || need > a.lds_slots) return;
  }
  if (M <= 0) {
    // (the time-out has a word of its own, status[2]: an input error of another tree, which
    // keeps the FIRST code in status[0], must not hide it from the host's fallback)
    if (FUSED && M < 0 && lane == 0) a.status[2] = 1 + t;
    return;
  }
  struct __attribute__((packed)) BytesTP {  // (the pair's codes of this tile)
    uint32_t d[TP / 4];
  };
  static_assert(TP % 4 == 0, "whole words of tip codes per tile");
  const int tile_start = tile * TP;
  const bool whole = tile_start + TP <= a.P;
  // (round 6) the tile's codes of a tip in exactly the form the LDS wants -- per column one word
  // (byte r: register r's code) or, COMPACT, one 16-bit field (4 bits per register) -- laid out
  // per tile at engine creation (launch_tip_code_tiles): a copy instead of TP bytes regrouped by
  // every wave (one category: 16 columns x R fields, ~150 vector instructions per tip pair and
  // round, a quarter of a fluA tile job's vector instructions)
  struct TileCodes {
    uint32_t d[ppr * (COMPACT ? 2 : 4) / 4];
  };
  const bool tiled = a.tip_code_tiles != nullptr;
  const TileCodes* __restrict__ tile_codes =
      reinterpret_cast<const TileCodes*>(a.tip_code_tiles) + (size_t)tile * (size_t)n;
  TileCodes codes_now[U] = {};
  if (tiled) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int j = lane + 64 * u, node = node_j[u];
      if (j < jmax && (unsigned)node < (unsigned)n) codes_now[u] = tile_codes[node];
    }
  }
  BytesTP bytes_now[U] = {};
  if (!tiled && whole && !COMPACT) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int j = lane + 64 * u, node = node_j[u];
      if (j < jmax && (unsigned)node < (unsigned)n)
        bytes_now[u] = *reinterpret_cast<const BytesTP*>(a.tip_codes + (size_t)node * a.P + tile_start);
    }
  }
  const int cat = b % KP, pgrp = b / KP;
  const int catc = cat < K ? cat : K - 1;
  // this lane's constant offsets into a position's record: internal (16-byte {f, tr} slot of
  // (category, lo, hi)) and tip (row hi of the category's table; the state's offset is added)
  const unsigned lane_moff = 16u * (unsigned)(catc * 16 + lo * 4 + hi);
  const unsigned lane_tip = (unsigned)catc * kTipCat + (unsigned)hi * kTipRow;
  const char* __restrict__ mm_g = reinterpret_cast<const char*>(a.mmats) + (size_t)gi * Mmax * kVisit;
  const int col = pgrp * 4 + lo;  // this lane's pattern column; register r adds r * ppr
  const double pi_l = model->pi[hi];
  const double cw_l = cat < K ? model->cat_weight[cat] : 0.0;
  const double rate_l = model->cat_rate[catc], drate_l = model->cat_drate[catc];
  const double AQ = model->Q[lo * 4 + hi];  // A operand for Q L (same in every block)

  struct Tw {
    uint32_t w[COMPACT ? 3 : 6];  // (COMPACT: two positions per word)
  };
  struct Slots {  // scalars (s_load_dwordx8)
    int q, c[2], g[4], dst;
  };
  auto load_shape = [&](int m) { return mw[m * 16]; };
  auto load_slots = [&](int m) {
    const cint_ptr p = mw + m * 16 + 8;
    return Slots{p[0], {p[1], p[2]}, {p[3], p[4], p[5], p[6]}, p[7]};
  };
  // LDS: [macro][column][tip codes of positions 0..5] -- re-used, macro by macro, for that
  // macro's edge sums once its tip words were consumed -- | vectors [slot][r][lane] |
  // RESCALE: exponents
  char* const lds0 = reinterpret_cast<char*>(wlds);
  char* const plv = lds0 + (unsigned)Mmax * kTStride;
  int16_t* exps = reinterpret_cast<int16_t*>(plv + (size_t)(ARENA ? a.lds_slots : max_stored(n)) * kVecBytes);
  const unsigned tw_lane = (unsigned)col * kCol;
  auto fetch_tw = [&](int m) {  // the tip words of visit m (LDS)
    Tw t;
    const char* twp = lds0 + ((unsigned)m * kTStride + tw_lane);
    if constexpr (COMPACT) {
#pragma unroll
      for (int j = 0; j < 3; j++) t.w[j] = *reinterpret_cast<const uint32_t*>(twp + 4 * j);
    } else {
      const uint4 w4 = *reinterpret_cast<const uint4*>(twp);
      const uint2 w2 = *reinterpret_cast<const uint2*>(twp + 16);
      t.w[0] = w4.x;
      t.w[1] = w4.y;
      t.w[2] = w4.z;
      t.w[3] = w4.w;
      t.w[4] = w2.x;
      t.w[5] = w2.y;
    }
    return t;
  };
  // byte offset of the table entry of register r's pattern at tip position POS
  auto tip_offset = [&](const Tw& tw, auto pos_tag, int r) {
    constexpr int POS = decltype(pos_tag)::value;
    if constexpr (COMPACT)
      return lane_tip + (__builtin_amdgcn_ubfe(tw.w[POS >> 1], (uint32_t)(16 * (POS & 1) + 4 * r), 3u) << 4);
    else
      return lane_tip + ((tw.w[POS] >> (8 * r)) & 0xffu);
  };

  // ARENA: this wave's arena, [stored node in order of consumption] x kVecBytes; a vector lies
  // there as register pairs interleaved by lane (16-byte accesses: one instruction moves two
  // registers -- every vector-memory instruction costs a wave ~45 clocks of issue)
  // followed by the odd register
  char* const arena =
      ARENA ? reinterpret_cast<char*>(a.plv + ((size_t)job_eval * a.g_tiles + tile) * max_stored(n) * (R * kTile))
            : nullptr;
  const unsigned lane16 = 16u * lane, lane8 = 8u * lane;

  // ---- operands of one child (J = 0, 1) of a visit, requested a visit ahead ----
  // W doubles per internal position ({f} post-order, {f, tr} pre-order), R W per tip position
  // ({P} / {P, P Q} per register).  One register group per child, laid out by its kind:
  //   tip:      x[0 .. RW)                      stored: x[0 .. W)
  //   unstored: x[0 .. W), first grandchild at x[W ..), second at x[(R+1)W ..) (tip: RW, else W)
  // (destinations are compile-time indices into the group: the groups live in registers)
  // ARENA, pre-order: a stored input's post-order vector comes from the arena into the SAME
  // group, behind the input's W doubles -- R more, and a stored input leaves (R - 1) W >= R of
  // its place unused: stored child x[W .. W + R), grandchildren x[2W ..) and x[(R+2)W ..)
  static_assert(R >= 2 && R <= 4, "an arena vector fits the unused part of its operand group ((R - 1) W >= R doubles)");
  auto load_internal = [&](auto pre_tag, const char* at, auto& o, auto off_tag) {
    constexpr bool PRE = decltype(pre_tag)::value;
    constexpr int OFF = decltype(off_tag)::value;
    unsigned voff = lane_moff;
    asm volatile("" : "+v"(voff));  // (kept a 32-bit offset beside the scalar base)
    if (PRE) {
      const double2 v = *reinterpret_cast<const double2*>(at + (size_t)voff);
      o.x[OFF] = v.x;
      o.x[OFF + 1] = v.y;
    } else {
      o.x[OFF] = *reinterpret_cast<const double*>(at + (size_t)voff);
    }
  };
  auto load_tip = [&](auto pre_tag, const char* at, const Tw& tw, auto pos_tag, auto& o, auto off_tag) {
    constexpr bool PRE = decltype(pre_tag)::value;
    constexpr int OFF = decltype(off_tag)::value;
#pragma unroll
    for (int r = 0; r < R; r++) {
#ifdef W3_ABL_FEWER_LOADS
      // (timing experiment, WRONG results, never the product: what tables by state PAIR could
      // save at best -- a post-order tip position issues two gathers instead of three; register 1
      // takes register 0's value.  DESIGN.md 4.1)
      if (!PRE && r == 1) {  // (a constant, not a copy: a copy would wait for the load right here)
        o.x[OFF + 1] = 0.25;
        continue;
      }
#endif
      unsigned voff = tip_offset(tw, pos_tag, r);
      asm volatile("" : "+v"(voff));
      if (PRE) {
        const double2 v = *reinterpret_cast<const double2*>(at + (size_t)voff);
        o.x[OFF + 2 * r] = v.x;
        o.x[OFF + 2 * r + 1] = v.y;
      } else {
        o.x[OFF + r] = *reinterpret_cast<const double*>(at + (size_t)voff);
      }
    }
  };
  auto load_arena = [&](int k, auto& o, auto off_tag) {  // k: scalar
    constexpr int OFF = decltype(off_tag)::value;
    const char* at = arena + (size_t)((unsigned)k * kVecBytes);
    unsigned v16 = lane16, v8 = lane8;
    asm volatile("" : "+v"(v16), "+v"(v8));
#pragma unroll
    for (int p = 0; p < R / 2; p++) {
      const dbl2* src = reinterpret_cast<const dbl2*>(at + p * (kTile * 16) + (size_t)v16);
      const dbl2 v = NT ? __builtin_nontemporal_load(src) : *src;
      o.x[OFF + 2 * p] = v.x;
      o.x[OFF + 2 * p + 1] = v.y;
    }
    if (R & 1) {
      const double* src = reinterpret_cast<const double*>(at + (R / 2) * (kTile * 16) + (size_t)v8);
      o.x[OFF + R - 1] = NT ? __builtin_nontemporal_load(src) : *src;
    }
  };
  auto fetch_child = [&](auto pre_tag, auto jtag, int sh, const Tw& tw, const char* sb, const char* sb4,
                         auto& o, int& ak) {
    constexpr bool PRE = decltype(pre_tag)::value;
    constexpr int J = decltype(jtag)::value;
    constexpr int W = PRE ? 2 : 1;
    constexpr bool AR = ARENA && PRE;
    using O0 = std::integral_constant<int, 0>;
    using OA = std::integral_constant<int, W>;
    using OB = std::integral_constant<int, (R + 1) * W>;
    using P0 = std::integral_constant<int, J>;
    using PA = std::integral_constant<int, 2 + 2 * J>;
    using PB = std::integral_constant<int, 3 + 2 * J>;
    const int kind = (sh >> (2 * J)) & 3;
    if (kind == 0) {
      load_tip(pre_tag, sb + J * kPos, tw, P0{}, o, O0{});
      return;
    }
    load_internal(pre_tag, sb + J * kPos, o, O0{});
    if (kind != 2) {
      if constexpr (AR) load_arena(ak++, o, std::integral_constant<int, W>{});
      return;
    }
    // positions 2, 3 (J = 0) lie below the 4095-byte immediate, 4, 5 (J = 1) beyond it
    const char* ga = J == 0 ? sb + 2 * kPos : sb4;
    const char* gb = J == 0 ? sb + 3 * kPos : sb4 + kPos;
    if (sh & (1 << (10 + 2 * J))) {
      load_tip(pre_tag, ga, tw, PA{}, o, OA{});
    } else {
      load_internal(pre_tag, ga, o, OA{});
      if constexpr (AR) load_arena(ak++, o, std::integral_constant<int, 2 * W>{});
    }
    if (sh & (1 << (11 + 2 * J))) {
#if defined(W3_ABL_FEWER_LOADS) && W3_ABL_FEWER_LOADS >= 2
      // (timing experiment, WRONG results: what CHERRY tables could save at best -- the second
      // tip of an unstored child with two tip children costs no gather in the post-order walk)
      if (!PRE && (sh & (1 << (10 + 2 * J)))) {
#pragma unroll
        for (int r = 0; r < R; r++) o.x[(R + 1) * W + r] = 0.25;
        return;
      }
#endif
      load_tip(pre_tag, gb, tw, PB{}, o, OB{});
    } else {
      load_internal(pre_tag, gb, o, OB{});
      if constexpr (AR) load_arena(ak++, o, std::integral_constant<int, (R + 2) * W>{});
    }
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using Post = std::false_type;
  using Pre = std::true_type;
#ifdef W3_STAMPS
  long long stamp_fetch = 0, stamp_visits = 0;
#endif
  auto fetch = [&](auto pre_tag, int m, int sh, const Tw& tw, auto& o0, auto& o1) {
#ifdef W3_STAMPS
    const long long f0 = __builtin_amdgcn_s_memtime();
#endif
    const char* sb = mm_g + (size_t)((unsigned)m * kVisit);
    unsigned off4 = 4 * kPos;
    asm volatile("" : "+s"(off4));
    const char* sb4 = sb + off4;
    int ak = (int)((unsigned)sh >> 16);  // ARENA: arena index of the visit's first stored input
    fetch_child(pre_tag, J0{}, sh, tw, sb, sb4, o0, ak);
    fetch_child(pre_tag, J1{}, sh, tw, sb, sb4, o1, ak);
#ifdef W3_STAMPS
    stamp_fetch += __builtin_amdgcn_s_memtime() - f0;
    stamp_visits++;
#endif
  };

  // ONE wait per visit for operands.  How many loads a fetch issues depends on the visit's
  // shape, so the compiler cannot count them: left alone it waits with vmcnt(0) at the first
  // use of a visit's operands -- i.e. also for the NEXT visit's operands requested just before,
  // and the prefetch is gone (first version: 0.83 ms, a third of a wave's life in s_waitcnt).
  // So, at the top of a visit: "use" every register of the visit's groups in an empty asm (the
  // compiler puts its vmcnt(0) there: everything older has had a whole visit to arrive), THEN
  // request the next visit's operands; what follows reads asm results and waits no more.
#ifdef W3_STAMPS
  long long stamp_wait = 0, stamp_t0 = __builtin_amdgcn_s_memtime();
#endif
  auto settle = [&](auto& o0, auto& o1) {
    constexpr int n = sizeof(o0.x) / sizeof(double);
#ifdef W3_STAMPS
    const long long w0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp_wait += __builtin_amdgcn_s_memtime() - w0;
#endif
#pragma unroll
    for (int i = 0; i < n; i++) asm volatile("" : "+v"(o0.x[i]));
#pragma unroll
    for (int i = 0; i < n; i++) asm volatile("" : "+v"(o1.x[i]));
  };

  const int M1 = M - 1;  // the root's macro is the last one; visits 0 .. M1 - 1 are stored nodes
  // the first visits' scalars are on their way while the tip codes are staged
  const int sh_a = load_shape(0), sh_b = load_shape(min(1, M1)), sh_c = load_shape(min(2, M1));
  const Slots sl_a = load_slots(0);

  int pat[R], patc[R];
  double pw[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    pat[r] = tile_start + r * ppr + col;
    patc[r] = pat[r] < a.P ? pat[r] : a.P - 1;
    pw[r] = pat[r] < a.P ? a.weights[patc[r]] : 0.0;
  }
  {
    // tip codes of this wave's columns, by (macro, position): a lane takes the (macro,
    // position) pairs whose node is a tip and copies their TP bytes
    auto stage_bytes = [&](int j, int node) {  // columns clamped to the last pattern
      const int m = j / 6, pos = j - m * 6;
      const uint8_t* src = a.tip_codes + (size_t)node * a.P;
      if constexpr (COMPACT) {
        // one 16-bit field per column: the state numbers (code / 16) of its R patterns (ppr apart)
        char* dst = lds0 + (unsigned)m * kTStride + (unsigned)pos * 2u;
        if (whole) {  // the tile's TP bytes as whole words
          uint32_t d[TP / 4];
#pragma unroll
          for (int i = 0; i < TP / 4; i++) d[i] = *reinterpret_cast<const uint32_t*>(src + tile_start + 4 * i);
#pragma unroll
          for (int c = 0; c < ppr; c++) {
            uint32_t f = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
              const int q = r * ppr + c;
              f |= ((d[q >> 2] >> (8 * (q & 3) + 4)) & 0x7u) << (4 * r);
            }
            *reinterpret_cast<uint16_t*>(dst + c * kCol) = (uint16_t)f;
          }
          return;
        }
        for (int c = 0; c < ppr; c++) {
          uint32_t f = 0;
#pragma unroll
          for (int r = 0; r < R; r++) {
            const int q = tile_start + r * ppr + c;
            f |= (((uint32_t)src[q < a.P ? q : a.P - 1] >> 4) & 0x7u) << (4 * r);
          }
          *reinterpret_cast<uint16_t*>(dst + c * kCol) = (uint16_t)f;
        }
      } else {
        char* dst = lds0 + (unsigned)m * kTStride + (unsigned)pos * 4u;
        for (int q = 0; q < TP; q++) {
          const int pp = tile_start + q < a.P ? tile_start + q : a.P - 1;
          dst[(q & (ppr - 1)) * kTwCol + (q >> 2)] = (char)src[pp];
        }
      }
    };
    // the packed form: a pair's TP bytes as whole words, regrouped into the four columns' words
    auto stage_packed = [&](int j, const BytesTP& w) {
      const int m = j / 6, pos = j - m * 6;
      char* dst = lds0 + (unsigned)m * kTStride + (unsigned)pos * 4u;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        uint32_t word = 0;  // byte r: the code of column c in register r
#pragma unroll
        for (int r = 0; r < R; r++) word |= ((w.d[r] >> (8 * c)) & 0xffu) << (8 * r);
        *reinterpret_cast<uint32_t*>(dst + c * kTwCol) = word;
      }
    };
    auto stage_tiled = [&](int j, const TileCodes& w) {
      const int m = j / 6, pos = j - m * 6;
      if constexpr (COMPACT) {
        char* dst = lds0 + (unsigned)m * kTStride + (unsigned)pos * 2u;
#pragma unroll
        for (int c = 0; c < ppr; c++)
          *reinterpret_cast<uint16_t*>(dst + c * kCol) = (uint16_t)(w.d[c >> 1] >> (16 * (c & 1)));
      } else {
        char* dst = lds0 + (unsigned)m * kTStride + (unsigned)pos * 4u;
#pragma unroll
        for (int c = 0; c < 4; c++) *reinterpret_cast<uint32_t*>(dst + c * kTwCol) = w.d[c];
      }
    };
    if (tiled) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int j = lane + 64 * u, node = node_j[u];
        if (j < jmax && (unsigned)node < (unsigned)n) stage_tiled(j, codes_now[u]);
      }
      for (int j = lane + 64 * U; j < jmax; j += kTile) {  // larger trees: the rest
        const int node = mwv[(j / 6) * 16 + 1 + (j % 6)];
        if ((unsigned)node < (unsigned)n) stage_tiled(j, tile_codes[node]);
      }
    } else if (whole && !COMPACT) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int j = lane + 64 * u, node = node_j[u];
        if (j < jmax && (unsigned)node < (unsigned)n) stage_packed(j, bytes_now[u]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int j = lane + 64 * u, node = node_j[u];
        if (j < jmax && (unsigned)node < (unsigned)n) stage_bytes(j, node);
      }
    }
    for (int j = lane + 64 * U; !tiled && j < jmax; j += kTile) {  // larger trees: the rest
      const int node = mwv[(j / 6) * 16 + 1 + (j % 6)];
      if ((unsigned)node < (unsigned)n) {
        if (whole && !COMPACT)
          stage_packed(j, *reinterpret_cast<const BytesTP*>(a.tip_codes + (size_t)node * a.P + tile_start));
        else
          stage_bytes(j, node);
      }
    }
  }
  __syncthreads();

  struct V {
    double v[R];
  };
  unsigned slot_stride = kVecBytes;
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
  auto store_arena = [&](int k, const V& x) {  // (the layout load_arena reads)
    char* at = arena + (size_t)((unsigned)k * kVecBytes);
    unsigned v16 = lane16, v8 = lane8;
    asm volatile("" : "+v"(v16), "+v"(v8));
#pragma unroll
    for (int p = 0; p < R / 2; p++) {
      dbl2* dst = reinterpret_cast<dbl2*>(at + p * (kTile * 16) + (size_t)v16);
      const dbl2 v = {x.v[2 * p], x.v[2 * p + 1]};
      if (NT) __builtin_nontemporal_store(v, dst);
      else *dst = v;
    }
    if (R & 1) {
      double* dst = reinterpret_cast<double*>(at + (R / 2) * (kTile * 16) + (size_t)v8);
      if (NT) __builtin_nontemporal_store(x.v[R - 1], dst);
      else *dst = x.v[R - 1];
    }
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
  // the looked-up products of a tip position: P e_s (stride W) and (P Q) e_s
  auto tip_p = [&](auto pre_tag, const auto& o, auto off_tag) {
    constexpr int W = decltype(pre_tag)::value ? 2 : 1;
    constexpr int OFF = decltype(off_tag)::value;
    V y;
#pragma unroll
    for (int r = 0; r < R; r++) y.v[r] = o.x[OFF + W * r];
    return y;
  };
  auto tip_pq = [&](const Ops<true, R>& o, auto off_tag) {
    constexpr int OFF = decltype(off_tag)::value;
    V y;
#pragma unroll
    for (int r = 0; r < R; r++) y.v[r] = o.x[OFF + 2 * r + 1];
    return y;
  };
  // ARENA, pre-order: the post-order vector of a stored input, as load_arena left it in the group
  auto arena_vec = [&](const auto& o, auto off_tag) {
    constexpr int OFF = decltype(off_tag)::value;
    V y;
#pragma unroll
    for (int r = 0; r < R; r++) y.v[r] = o.x[OFF + r];
    return y;
  };

  double qroot[R];  // root pre-order vector: pi * category weight * w_p / site likelihood
  int esum[R];      // RESCALE: exponents removed so far, per pattern
#pragma unroll
  for (int r = 0; r < R; r++) esum[r] = 0;

  // What a visit knows of one child: S = P_c L_c (the message to the parent / the sibling),
  // and for an internal child its vector L (stored: from its slot -- ARENA, pre-order: from the
  // arena --; unstored: Ap o Bp, the products of its two children, whose vectors xa / xb are
  // read only where they are stored nodes).
  struct Child {
    V S, L, xa, xb, Ap, Bp;
  };
  auto child_S = [&](auto pre_tag, auto jtag, int sh, const auto& o, const Slots& sl, Child& c) {
    constexpr bool PRE = decltype(pre_tag)::value;
    constexpr int J = decltype(jtag)::value;
    constexpr int W = PRE ? 2 : 1;
    constexpr bool AR = ARENA && PRE;
    using O0 = std::integral_constant<int, 0>;
    using OA = std::integral_constant<int, W>;
    using OB = std::integral_constant<int, (R + 1) * W>;
    const int kind = (sh >> (2 * J)) & 3;
    if (kind == 0) {
      c.S = tip_p(pre_tag, o, O0{});
      return;
    }
    if (kind == 1) {
      if constexpr (AR) c.L = arena_vec(o, std::integral_constant<int, W>{});
      else c.L = load_slot(sl.c[J]);
    } else {
      if (sh & (1 << (10 + 2 * J))) {
        c.Ap = tip_p(pre_tag, o, OA{});
      } else {
        if constexpr (AR) c.xa = arena_vec(o, std::integral_constant<int, 2 * W>{});
        else c.xa = load_slot(sl.g[2 * J]);
        c.Ap = mm(o.x[W], c.xa);
      }
      if (sh & (1 << (11 + 2 * J))) {
        c.Bp = tip_p(pre_tag, o, OB{});
      } else {
        if constexpr (AR) c.xb = arena_vec(o, std::integral_constant<int, (R + 2) * W>{});
        else c.xb = load_slot(sl.g[2 * J + 1]);
        c.Bp = mm(o.x[(R + 1) * W], c.xb);
      }
      c.L = mul(c.Ap, c.Bp);
    }
    c.S = mm(o.x[0], c.L);
  };

  // ARENA: the last stored vector, on its way to the arena -- stored at the top of the NEXT
  // visit, behind that visit's operand wait and requests, so that the store has a whole visit
  // before the next wait (stores count on vmcnt like loads)
  V pend_L;
  int pend_dst = 0;
  bool pend = false;
  auto flush_arena = [&]() {
    if (ARENA && pend) {
      store_arena(pend_dst, pend_L);
      pend = false;
    }
  };

  // Post-order (round 6): the message S = P_c L_c of child J is handed to its consumer `use`
  // INSIDE the arm that knows where it lives -- a tip's S is the registers its gathers filled, an
  // internal child's the matrix instruction's result -- and an unstored child's vector is formed
  // at the four leaves of ONE decision tree on its grandchildren's tip flags.  Written as
  // "compute S under the child's branches, then multiply" (as the pre-order walk's child_S still
  // is), the compiler joined S into common registers behind the branches and every tip arm
  // paid R 64-bit moves on the visit's dependent chain (found by a census of the kernel's
  // instructions, profiles/r06_walk_isa_census.txt: 17 % of the visit code's vector
  // instructions were moves): kernel 0.794 -> 0.774 ms per 1000 DS1 trees (-2.5 %), bit-identical
  // (the same products in the same order).  The consumer's code is inlined into both arms of
  // child 0 (child 1's code twice): 3 047 -> 3 411 static instructions, 214 registers as before.
  auto post_child = [&](auto jtag, int sh, const Ops<false, R>& o, const Slots& sl, auto&& use) {
    constexpr int J = decltype(jtag)::value;
    using O0 = std::integral_constant<int, 0>;
    using OA = std::integral_constant<int, 1>;
    using OB = std::integral_constant<int, R + 1>;
    const int kind = (sh >> (2 * J)) & 3;
    if (kind == 0) {
      use(tip_p(Post{}, o, O0{}));
      return;
    }
    V L;
    if (kind == 1) {
      L = load_slot(sl.c[J]);
    } else if (sh & (1 << (10 + 2 * J))) {
      if (sh & (1 << (11 + 2 * J))) L = mul(tip_p(Post{}, o, OA{}), tip_p(Post{}, o, OB{}));
      else L = mul(tip_p(Post{}, o, OA{}), mm(o.x[R + 1], load_slot(sl.g[2 * J + 1])));
    } else {
      const V Ap = mm(o.x[1], load_slot(sl.g[2 * J]));
      if (sh & (1 << (11 + 2 * J))) L = mul(Ap, tip_p(Post{}, o, OB{}));
      else L = mul(Ap, mm(o.x[R + 1], load_slot(sl.g[2 * J + 1])));
    }
    use(mm(o.x[0], L));
  };

  // ================= post-order over the stored nodes, then the root (site likelihood) ====
  auto post_visit = [&](auto root_tag, int sh, const Slots& sl, const Ops<false, R>& o0, const Ops<false, R>& o1,
                        int tile_for_ll) {
    constexpr bool ROOT = decltype(root_tag)::value;
    V Lv;
    post_child(J0{}, sh, o0, sl, [&](const V& S0) {
      post_child(J1{}, sh, o1, sl, [&](const V& S1) { Lv = mul(S0, S1); });
    });
    if (!ROOT) {
      if (RESCALE) {
#pragma unroll
        for (int r = 0; r < R; r++) {
          const double colsum = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, Lv.v[r], 0.0, 0, 0, 0);
          int ex = colsum > 0.0 ? __builtin_amdgcn_frexp_exp(colsum) : -4096;
          if (KP >= 2) ex = max(ex, __shfl_xor(ex, 4, 64));
          if (KP >= 4) ex = max(ex, __shfl_xor(ex, 8, 64));
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
        double v = cw_l * pi_l * Lv.v[r];
        v = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, v, 0.0, 0, 0, 0);  // states
        if (KP == 4) {                                                   // categories
          v = row_ror_add<8>(v);
          v = row_ror_add<4>(v);
        } else if (KP == 2) {
          v += __shfl_xor(v, 4, 64);
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
#pragma unroll
      for (int r = 0; r < R; r++)
        qroot[r] = pi_l * cw_l * __builtin_amdgcn_mfma_f64_4x4x4f64(hi == r ? 1.0 : 0.0, quot, 0.0, 0, 0, 0);
      double ll = 0.0;
      if (hi < R && (b % KP) == 0 && pv < a.P)
        ll = wv * (RESCALE ? log(sv) + ev * 0.69314718055994530942 : log(sv));
      ll = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, ll, 0.0, 0, 0, 0);  // rows
      ll = row_ror_add<8>(ll);
      ll = row_ror_add<4>(ll);
      ll = row_ror_add<2>(ll);
      ll = row_ror_add<1>(ll);
      if (lane == 0) a.ll_part[(size_t)e * a.ll_tiles + tile_for_ll] = ll;
    }
  };
  using Inner = std::false_type;
  using Root = std::true_type;
  {
    // Two visits per iteration, two operand sets (A, B), nothing copied.  At the top of visit
    // m: shape (s0), slots and operands of m are there; shape (s1) and tip words of m + 1 too,
    // so its operands are requested into the other set; then the tip words of m + 2 are read
    // (LDS).  The visit ends with the scalar loads: slots of m + 1, shape of m + 3 (scalar
    // loads return out of order: any wait for LDS data with one of them in flight is a wait
    // for everything -- at the visit's top or in its middle they cost 12-14 %, round 4).
    Ops<false, R> a0, a1, b0, b1;
    int s0 = sh_a, s1 = sh_b, s2 = sh_c;
    Slots la = sl_a, lb;
    Tw tw = fetch_tw(0);
    fetch(Post{}, 0, s0, tw, a0, a1);
    tw = fetch_tw(min(1, M1));
    for (int m = 0; m < M1; m += 2) {
      // ---- visit m (set A) ----
      settle(a0, a1);
      fetch(Post{}, min(m + 1, M1), s1, tw, b0, b1);
      tw = fetch_tw(min(m + 2, M1));
      flush_arena();
      post_visit(Inner{}, s0, la, a0, a1, 0);
      lb = load_slots(min(m + 1, M1));
      const int s3 = load_shape(min(m + 3, M1));
      s0 = s1;
      s1 = s2;
      s2 = s3;
      if (m + 1 < M1) {
        // ---- visit m + 1 (set B) ----
        settle(b0, b1);
        fetch(Post{}, min(m + 2, M1), s1, tw, a0, a1);
        tw = fetch_tw(min(m + 3, M1));
        flush_arena();
        post_visit(Inner{}, s0, lb, b0, b1, 0);
        la = load_slots(min(m + 2, M1));
        const int s4 = load_shape(min(m + 4, M1));
        s0 = s1;
        s1 = s2;
        s2 = s4;
      } else {  // M1 odd: the root's operands arrived in set B
        a0 = b0;
        a1 = b1;
        la = lb;
      }
    }
    settle(a0, a1);
    flush_arena();
    post_visit(Root{}, s0, la, a0, a1, tile);
  }

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
      *reinterpret_cast<double*>(lds0 + ((unsigned)m * kTStride + (unsigned)pos_a * 16u + sum_lane)) = red;
  };
  // Edge of child c below a node with pre-order vector q and sibling message S (qs = q o S):
  //   internal child: q_c = P_c^T qs, numerator q_c o (Q L_c), q_c kept if stored
  //   tip child:      numerator qs o ((P_c Q) e_state)  -- looked up
  auto inner_edge = [&](double trm, const V& qs, const V& Lc, V& qc) {
    qc = mm(trm, qs);
    return mul(qc, mm(AQ, Lc));
  };
  auto child_edges = [&](auto jtag, int sh, const Ops<true, R>& o, const Slots& sl, const Child& c, const V& qs,
                         int m, V& nout) {
    constexpr int J = decltype(jtag)::value;
    const int kind = (sh >> (2 * J)) & 3;
    if (kind == 0) {
      nout = mul(qs, tip_pq(o, std::integral_constant<int, 0>{}));
      return;
    }
    V qc;
    nout = inner_edge(o.x[1], qs, c.L, qc);
    if (kind == 1) {
      store_slot(sl.c[J], qc);
      return;
    }
    V na, nb, qa, qb;
    const V qsa = mul(qc, c.Bp), qsb = mul(qc, c.Ap);
    if (sh & (1 << (10 + 2 * J))) {
      na = mul(qsa, tip_pq(o, std::integral_constant<int, 2>{}));
    } else {
      na = inner_edge(o.x[3], qsa, c.xa, qa);
      store_slot(sl.g[2 * J], qa);
    }
    if (sh & (1 << (11 + 2 * J))) {
      nb = mul(qsb, tip_pq(o, std::integral_constant<int, 2 * (R + 1)>{}));
    } else {
      nb = inner_edge(o.x[2 * (R + 1) + 1], qsb, c.xb, qb);
      store_slot(sl.g[2 * J + 1], qb);
    }
    edge_sums(na, nb, m, 2 + 2 * J);
  };
  auto pre_visit = [&](auto root_tag, int sh, const Slots& sl, const Ops<true, R>& o0, const Ops<true, R>& o1, int m) {
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
    V n0, n1;
    // (two treatments of the pre-order walk in the same spirit were built and measured in round 6
    // and not kept: every use of a child's S, Ap, Bp branching on the shape bit again and reading
    // a tip's values in place -- +0.5 % alone, and it took back 1 % of the post-order form's
    // gain: more branches than moves saved --; and q o S formed inside the arm that produced S,
    // no new branch -- level, 0.7716 against 0.7743 ms.  profiles/r06_ab_post_cps.txt, r06_ab_pre_qs.txt)
    child_S(Pre{}, J0{}, sh, o0, sl, c0);
    child_S(Pre{}, J1{}, sh, o1, sl, c1);
    child_edges(J0{}, sh, o0, sl, c0, mul(qv, c1.S), m, n0);
    child_edges(J1{}, sh, o1, sl, c1, mul(qv, c0.S), m, n1);
    edge_sums(n0, n1, m, 0);
  };
  {
    // the root's visit first (set A), then the stored nodes downwards, B and A in turn; the
    // same pipeline backwards: at the top of visit m the operands of m - 1 are requested,
    // then the tip words of m - 2 are read; the visit ends with the slots of m - 1 and the
    // shape of m - 3
    auto dn = [&](int m) { return max(m, 0); };
    Ops<true, R> a0, a1, b0, b1;
    int s0 = load_shape(M1), s1 = load_shape(dn(M1 - 1)), s2 = load_shape(dn(M1 - 2));
    Slots la = load_slots(M1), lb;
    Tw tw = fetch_tw(M1);
    fetch(Pre{}, M1, s0, tw, a0, a1);
    tw = fetch_tw(dn(M1 - 1));
    {  // ---- visit M1 (the root, set A) ----
      settle(a0, a1);
      fetch(Pre{}, dn(M1 - 1), s1, tw, b0, b1);
      tw = fetch_tw(dn(M1 - 2));
      pre_visit(Root{}, s0, la, a0, a1, M1);
      lb = load_slots(dn(M1 - 1));
      const int s3 = load_shape(dn(M1 - 3));
      s0 = s1;
      s1 = s2;
      s2 = s3;
    }
    for (int m = M1 - 1; m >= 0; m -= 2) {
      // ---- visit m (set B) ----
      settle(b0, b1);
      fetch(Pre{}, dn(m - 1), s1, tw, a0, a1);
      tw = fetch_tw(dn(m - 2));
      pre_visit(Inner{}, s0, lb, b0, b1, m);
      la = load_slots(dn(m - 1));
      const int s3 = load_shape(dn(m - 3));
      s0 = s1;
      s1 = s2;
      s2 = s3;
      if (m >= 1) {
        // ---- visit m - 1 (set A) ----
        settle(a0, a1);
        fetch(Pre{}, dn(m - 2), s1, tw, b0, b1);
        tw = fetch_tw(dn(m - 3));
        pre_visit(Inner{}, s0, la, a0, a1, m - 1);
        lb = load_slots(dn(m - 2));
        const int s4 = load_shape(dn(m - 4));
        s0 = s1;
        s1 = s2;
        s2 = s4;
      }
    }
  }
#ifdef W3_STAMPS
  if (lane == 0 && (job_eval % 250) == 3 && (tile % 39) == 5)
    printf("walk3 stamps eval %d tile %d: operand waits %lld, operand requests %lld (%lld visits) of %lld shader clocks\n", job_eval, tile,
           stamp_wait, stamp_fetch, stamp_visits, (long long)__builtin_amdgcn_s_memtime() - stamp_t0);
#endif
  __syncthreads();
  // positions that do not exist in a macro are never written nor read downstream
  const int gwidth = Mmax * kMacroPositions * 2;
  double* gout = a.g_part + ((size_t)gi * a.g_tiles + tile) * gwidth;
  for (int i = lane; i < M * kMacroPositions * 2; i += kTile) {
    const int m = i / (kMacroPositions * 2), r = i - m * (kMacroPositions * 2);
    gout[i] = *reinterpret_cast<const double*>(lds0 + (unsigned)m * kTStride + (unsigned)r * 8u);
  }
}

// (R = 2 -- 8 patterns per wave, 14 KB of LDS -- is built with three waves per SIMD:
// `make EXTRA_LLVM=-DMI_LLR=2`, an experiment of round 4)
// (WAVES: waves per SIMD the registers are budgeted for.  Two -- 256 registers -- everywhere but in
// the wide-tile form with every vector in LDS, which needs ~280: that one is built for ONE wave
// per SIMD and serves the calls of a wide-tile engine whose waves are all resident at that
// occupancy, a handful of trees; larger calls take the arena.)
template <bool RESCALE, bool ARENA, int KP, bool NT = false, int RR = kRegs, int WAVES = (RR == 2 ? 3 : 2)>
__global__ __launch_bounds__(kTile, WAVES) void gradient_walk_lut_kernel(LikArgs a) {
  extern __shared__ double wlds[];
  walk_lut_body<RESCALE, false, ARENA, KP, NT, RR>(a, wlds, blockIdx.x, nullptr, 0, 0);
}

// ------------------------------------------------------------------------
// The operand records of the third-generation walk.  One thread per (node, category) of ONE
// gradient evaluation (blockIdx.x), as transition_macro_kernel: P = I + V expm1(L r t) V^-1
// (negative entries clamped as BEAGLE does); an internal node's record is the second
// generation's ({f, tr} per (lo, hi) slot), a tip's is the table [row][state 0..3, gap]{P, P Q}.
// Staged through LDS (row stride 41) so that a block writes runs of whole records.
// ------------------------------------------------------------------------
// One (node, category) of a record: P = I + V expm1(L tau) V^-1 (negative entries clamped), as
// the {f, tr} slots of an internal position (32 doubles) or the table of a tip position (40).
__device__ __forceinline__ void lut_record(const DevModel& md, const double tau, const bool tip, double* st) {
  double ex[4], W[16], Pm[16];
  for (int x = 0; x < 4; x++) ex[x] = expm1(md.lambda[x] * tau);
  for (int x = 0; x < 4; x++)
    for (int j = 0; j < 4; j++) W[x * 4 + j] = ex[x] * md.Vinv[x * 4 + j];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double sum = i == j ? 1.0 : 0.0;
      for (int x = 0; x < 4; x++) sum += md.V[i * 4 + x] * W[x * 4 + j];
      Pm[i * 4 + j] = sum > 0 ? sum : 0;
    }
  if (tip) {
    for (int i = 0; i < 4; i++) {
      double ps = 0, qs = 0;
      for (int j = 0; j < 4; j++) {
        double pq = 0;
        for (int x = 0; x < 4; x++) pq += Pm[i * 4 + x] * md.Q[x * 4 + j];
        st[(i * 5 + j) * 2] = Pm[i * 4 + j];
        st[(i * 5 + j) * 2 + 1] = pq;
        ps += Pm[i * 4 + j];
        qs += pq;
      }
      st[(i * 5 + 4) * 2] = ps;
      st[(i * 5 + 4) * 2 + 1] = qs;
    }
  } else {
    for (int l = 0; l < 4; l++)  // slot (lo = l, hi = h): f = P[lo][hi], tr = P[hi][lo]
      for (int h = 0; h < 4; h++) {
        st[(l * 4 + h) * 2] = Pm[l * 4 + h];
        st[(l * 4 + h) * 2 + 1] = Pm[h * 4 + l];
      }
  }
}

constexpr int kTlBlock = 128;
__global__ __launch_bounds__(kTlBlock) void transition_lut_kernel(TransitionMacroArgs a) {
  __shared__ double stage[kTlBlock * 41];
  __shared__ int rec_of[kTlBlock];  // byte offset / 8 of the thread's record, -1: none
  __shared__ int len_of[kTlBlock];  // doubles in it (32: internal, 40: tip)
  extern __shared__ int slot_of[];  // [N - 1]: node -> macro * 6 + position
  const int Mmax = max_macros(a.n);
  const int ge = blockIdx.x;
  int t, mi;
  a.map.decode(a.eval_begin + ge, t, mi);
  const MacroEntry* mac = a.macros + (size_t)t * macro_stride(a.n);
  const int M = a.macro_count[t];
  for (int j = threadIdx.x; j < M * 6; j += kTlBlock) {
    const int m = j / 6, pos = j - m * 6;
    const MacroEntry& me = mac[m];
    if (pos < 2 || ((me.shape >> (2 * ((pos - 2) >> 1))) & 3) == 2)
      slot_of[pos < 2 ? me.child[pos] : me.grand[pos - 2]] = j;
  }
  __syncthreads();
  const int idx = blockIdx.y * kTlBlock + threadIdx.x;  // node * K + category
  const int node = idx / a.K, k = idx - node * a.K;
  const bool live = node < a.N - 1 && M > 0;
  const DevModel& md = a.models[mi];
  double* st = stage + threadIdx.x * 41;
  if (live) {
    const bool tip = node < a.n;
    const double tau = md.cat_rate[k] * a.bl_eff[(size_t)t * a.N + node];
    lut_record(md, tau, tip, st);
    const int slot = slot_of[node], m = slot / 6, pos = slot - m * 6;
    const int rec8 = (int)((((size_t)ge * Mmax + m) * 6 + pos) * (kPos / 8));
    rec_of[threadIdx.x] = rec8 + k * (tip ? (int)(kTipCat / 8) : 32);
    len_of[threadIdx.x] = tip ? 40 : 32;
  } else {
    rec_of[threadIdx.x] = -1;
    len_of[threadIdx.x] = 0;
  }
  __syncthreads();
  // eight threads per record, five doubles each: consecutive threads write consecutive bytes
  for (int x = threadIdx.x; x < kTlBlock * 8; x += kTlBlock) {
    const int th = x >> 3, part = x & 7;
    const int rec = rec_of[th], len = len_of[th];
    if (rec < 0) continue;
#pragma unroll
    for (int q = 0; q < 5; q++) {
      const int d = q * 8 + part;
      if (d < len) a.mmats[(size_t)rec + d] = stage[th * 41 + d];
    }
  }
}

// ------------------------------------------------------------------------
// The one-launch small call (round 5).  A `phylo_gradients` call of a
// JC69-type engine used to be four dependent launches: tree set-up (+ model instances) ->
// operand records -> walk -> reduce/finalize; on a batch of 125 trees -- the share of one GPU
// of eight under strong scaling -- the two set-up kernels and their launch boundaries were
// 20 of the step's 145 microseconds, with most of the chip idle.  Here they are the FIRST
// 4 T one-wave workgroups of the walk's own launch:
//   * set-up wave (tree t, quarter q): builds the tree's schedule in its registers
//     (small_tree_build -- all four quarters do, redundantly: it costs no memory traffic and no
//     hand-off between them), the model instance in LDS, then writes its quarter of the tree's
//     (node, category) operand records; quarter 0 also writes the macro list, the model, the
//     log-likelihood schedule and the effective branch lengths.  What walk waves read is stored
//     write-through (sc1); then s_waitcnt vmcnt(0) and ONE agent-scope add to ready[t].
//   * walk waves (walk_lut_body<.., true>) poll ready[t] (sc1 load, s_sleep in between, bounded).
// Deadlock-freedom rests on workgroups being dispatched in id order (observed, not promised by
// HIP): the set-up waves are resident (or done) before any walk wave is, and they wait for
// nothing.  Should a walk wave ever wait in vain, it gives up after `spin_ticks` of wall-clock
// time and raises the time-out word status[2]: a host-pointer entry point then runs the call
// again through the four-launch sequence and returns ITS results (round 6: a time-out never
// reaches such a caller); a *_device caller finds the sticky status kFusedTimeout.
// The hand-off is a release / acquire pair at agent scope (round 6): the set-up wave
// passes a RELEASE fence between its (write-through, waited-for) stores and the add, the walk
// wave an ACQUIRE fence and a scalar-cache invalidate between its poll and its first read.
// ------------------------------------------------------------------------
// Write-through (sc1) stores of 16 bytes.  Inline assembly: the compiler has no 16-byte store
// with a scope, and does not count these -- the role ends with an explicit s_waitcnt vmcnt(0).
typedef int v4i32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_sc1(void* p, const v4i32 v) {
  // (s_nop: a store of more than 64 bits reads its data registers a cycle or two after it
  // issues, and the compiler's hazard recogniser, which pads its own stores, does not see this
  // one -- without the wait states the next instruction overwrote a data register of the macro
  // list's stores now and then: garbage macros, wild tip reads in the walk waves)
  // (no "memory" clobber: volatile asm statements keep their order among themselves -- the
  // role's final wait is one --, and the LDS reads of the staged records may be batched across
  // these stores)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v));
}
__device__ __forceinline__ void store16_sc1(double* p, const double x, const double y) {
  store16_sc1(p, v4i32{__double2loint(x), __double2hiint(x), __double2loint(y), __double2hiint(y)});
}

// LDS of a set-up wave: staged records [per_q][41] | model instance | rec_of, len_of [64] | slot_of [N - 1]
__host__ __device__ inline unsigned fused_setup_lds(int n, int K) {
  const unsigned per_q = (unsigned)(((2 * n - 2) * K + kSetupQuarters - 1) / kSetupQuarters);
  return per_q * 41u * 8u + (unsigned)sizeof(DevModel) + 2u * 64u * 4u + (unsigned)(2 * n - 2) * 4u;
}

__device__ __forceinline__ void fused_setup_role(const FusedSetupArgs& f, const int bid, char* lds) {
  const TreeSetupArgs& a = f.ts;
  // Which (tree, quarter): workgroups are dealt to the eight XCDs round-robin by id, and the walk
  // waves of tree t (t in a whole group of eight: xcd_map) all run on XCD (setup_blocks + t) & 7 --
  // its set-up waves are placed there too (round 6): what they write passes through the ONE L2
  // both sides use (no reliance on another XCD's L2 having never seen the lines), and the walk
  // waves find the records in that L2 instead of fetching them from the memory side.
  // f.colocate == 0: bid / 4, bid % 4 as in round 5 (A/B).
  int t, q;
  {
    const int T = f.setup_blocks / kSetupQuarters, full = T & ~7;
    if (f.colocate && bid < kSetupQuarters * full) {
      const int j = bid >> 3, x = bid & 7;
      t = 8 * (j / kSetupQuarters) + ((x - f.xcd_base) & 7);
      q = j % kSetupQuarters;
    } else {
      t = bid / kSetupQuarters;
      q = bid - t * kSetupQuarters;
    }
  }
  const int lane = threadIdx.x;
  const int n = a.n, N = 2 * n - 1, K = f.ms.K;
  const int per_q = ((N - 1) * K + kSetupQuarters - 1) / kSetupQuarters;
  double* stage = reinterpret_cast<double*>(lds);
  DevModel* md = reinterpret_cast<DevModel*>(lds + (unsigned)per_q * 41u * 8u);
  int* rec_of = reinterpret_cast<int*>(reinterpret_cast<char*>(md) + sizeof(DevModel));
  int* len_of = rec_of + 64;
  int* slot_of = len_of + 64;

#ifdef W3_SETUP_STAMPS
  long long st0 = __builtin_amdgcn_s_memrealtime(), st1, st2, st3, st4, st5;
#define XSTAMP(x) x = __builtin_amdgcn_s_memrealtime()
#else
#define XSTAMP(x)
#endif
  // the model instance of the tree (one per tree in these calls)
  model_setup_wave(f.ms, t, lane, *md);
  XSTAMP(st1);
  SmallTree<1> tree;
  small_tree_build<1>(a, t, lane, tree, reinterpret_cast<int*>(lds));  // (the staging area is not in use yet)
  const bool ok = tree.status == kOk;
  const int M = ok ? tree.macro_total : 0;
  XSTAMP(st2);
  if (q == 0) small_tree_store<1>(a, t, lane, tree);  // (plain stores: later kernels read these)
  if (ok) {
    // node -> macro * 6 + position (the lane that owns a macro knows its children)
    if (tree.is_macro[0]) {
      const MacroEntry& me = tree.me[0];
      const int base = tree.macro_rank[0] * 6;
      slot_of[me.child[0]] = base;
      slot_of[me.child[1]] = base + 1;
#pragma unroll
      for (int j = 0; j < 2; j++)
        if (((me.shape >> (2 * j)) & 3) == 2) {
          slot_of[me.grand[2 * j]] = base + 2 + 2 * j;
          slot_of[me.grand[2 * j + 1]] = base + 3 + 2 * j;
        }
    }
    if (q == 0) {
      // the macro list, write-through: 16 words per entry
      MacroEntry* mac = a.macros + (size_t)t * macro_stride(n);
      if (tree.is_macro[0]) {
        char* dst = reinterpret_cast<char*>(mac + tree.macro_rank[0]);
        const MacroEntry& me = tree.me[0];
        static_assert(sizeof(MacroEntry) == 64, "sixteen words, in this order");
        store16_sc1(dst, v4i32{me.shape, me.child[0], me.child[1], me.grand[0]});
        store16_sc1(dst + 16, v4i32{me.grand[1], me.grand[2], me.grand[3], me.node});
        store16_sc1(dst + 32, v4i32{me.qslot, me.cslot[0], me.cslot[1], me.gslot[0]});
        store16_sc1(dst + 48, v4i32{me.gslot[1], me.gslot[2], me.gslot[3], me.pad});
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its LDS operations are in order)
  if (ok) {
    if (q == 0) {  // the model instance, write-through (the walk reads frequencies, rates, weights)
      double* dst = reinterpret_cast<double*>(f.ms.models + t);
      const double* src = reinterpret_cast<const double*>(md);
      static_assert(sizeof(DevModel) % 16 == 0, "whole 16-byte pieces");
      for (int i = lane; i < (int)(sizeof(DevModel) / 16); i += 64) store16_sc1(dst + 2 * i, src[2 * i], src[2 * i + 1]);
    }
    const int idx = q * per_q + lane;  // node * K + category
    const int node = idx / K, k = idx - node * K;
    const bool live = lane < per_q && node < N - 1;
    if (live) {
      // effective branch length as small_tree_store forms it
      double bl;
      if (!a.rooted) {
        bl = node < N - 2 ? a.bl[(size_t)t * (N - 1) + node] : 0.0;
      } else {
        bl = a.bl[(size_t)t * N + node];
        if (a.rates) bl *= a.rates[(size_t)t * (N - 1) + node];
      }
      const bool tip = node < n;
      lut_record(*md, md->cat_rate[k] * bl, tip, stage + lane * 41);
      const int slot = slot_of[node], m = slot / 6, pos = slot - m * 6;
      const int rec8 = (int)((((size_t)t * max_macros(n) + m) * 6 + pos) * (kPos / 8));
      rec_of[lane] = rec8 + k * (tip ? (int)(kTipCat / 8) : 32);
      len_of[lane] = tip ? 40 : 32;
    } else {
      rec_of[lane] = -1;
      len_of[lane] = 0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    XSTAMP(st3);
    // eight lanes per record, up to three 16-byte pieces each: consecutive lanes write
    // consecutive bytes.  Every LDS read first, unconditionally (rows clamped), then the
    // stores under their conditions: with a read and its wait inside each condition the 24
    // stores went out one LDS round trip apart (1.9 of the role's 11 microseconds).
    {
      int rec[8], len[8];
      double va[24], vb[24];
      const int part = lane & 7;
#pragma unroll
      for (int x8 = 0; x8 < 8; x8++) {
        const int th = (lane >> 3) + 8 * x8;
        rec[x8] = rec_of[th];
        len[x8] = len_of[th];
        const int row = min(th, per_q - 1) * 41;
#pragma unroll
        for (int p3 = 0; p3 < 3; p3++) {
          const int d = 2 * (p3 * 8 + part);
          va[3 * x8 + p3] = stage[row + min(d, 38)];
          vb[3 * x8 + p3] = stage[row + min(d, 38) + 1];
        }
      }
#pragma unroll
      for (int i = 0; i < 24; i++) asm volatile("" : "+v"(va[i]), "+v"(vb[i]));
#pragma unroll
      for (int x8 = 0; x8 < 8; x8++)
#pragma unroll
        for (int p3 = 0; p3 < 3; p3++) {
          const int d = 2 * (p3 * 8 + part);
          if (rec[x8] >= 0 && d < len[x8])
            store16_sc1(f.mmats + (size_t)rec[x8] + d, va[3 * x8 + p3], vb[3 * x8 + p3]);
        }
    }
  }
  XSTAMP(st4);
  // everything above has reached memory before the word says so
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  XSTAMP(st5);
#ifdef W3_SETUP_STAMPS
  if (lane == 0 && (t == 0 || t == 60 || t == 124) && q < 2)
    printf("setup role t %d q %d: start %lld end %lld model %lld tree %lld records %lld stores issued %lld landed %lld (10 ns ticks)\n", t, q, st0, (long long)__builtin_amdgcn_s_memrealtime(),
           st1 - st0, st2 - st1, st3 - st2, st4 - st3, st5 - st4);
#endif
  if (f.fence == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  // (f.debug_skip: testing only -- quarter 1 of tree debug_skip - 1 never reports, its walk waves
  // time out: tests/test_fused_setup_gpu.py forces the host's fallback with it)
  if (lane == 0 && f.ready && !(f.debug_skip == t + 1 && q == 1))
    __hip_atomic_fetch_add(f.ready + (size_t)t * kReadyStride, kReadyQuarter + (q == 0 ? M : 0), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// The set-up waves alone (launch_setup_records).
__global__ __launch_bounds__(kTile) void setup_records_kernel(FusedSetupArgs f) {
  extern __shared__ double wlds[];
  fused_setup_role(f, blockIdx.x, reinterpret_cast<char*>(wlds));
}

template <bool RESCALE, int KP>
__global__ __launch_bounds__(kTile, kRegs == 2 ? 3 : 2) void gradient_walk_lut_fused_kernel(LikArgs a, FusedSetupArgs f) {
  extern __shared__ double wlds[];
  if ((int)blockIdx.x < f.setup_blocks) {
    fused_setup_role(f, blockIdx.x, reinterpret_cast<char*>(wlds));
    return;
  }
  walk_lut_body<RESCALE, true, false, KP>(a, wlds, (int)blockIdx.x - f.setup_blocks, f.ready, f.spin_ticks, f.fence);
#ifdef W3_SETUP_STAMPS
  if (threadIdx.x == 0 && (((int)blockIdx.x - f.setup_blocks) % 1499) == 0)
    printf("walk block %d: out %lld\n", (int)blockIdx.x - f.setup_blocks, (long long)__builtin_amdgcn_s_memrealtime());
#endif
}

}  // namespace

size_t gradient_walk_lut_mats_bytes_per_eval(int n) { return (size_t)max_macros(n) * kVisit; }

// categories per matrix instruction for K rate categories
static int lut_kp(int K) { return K == 1 ? 1 : (K == 2 ? 2 : 4); }

// trees whose arrays fit one register per lane (64 nodes), a quarter of the (node, category)
// pairs per set-up wave
bool gradient_walk_lut_fused_applies(int n, int K) {
  return gradient_walk_lut_applies(K) && 2 * n - 1 <= 64 && ((2 * n - 2) * K + 3) / 4 <= 64;
}

template <bool RESCALE, int KP>
static void launch_fused_variant(const LikArgs& a, const FusedSetupArgs& f, dim3 grid, size_t lds, hipStream_t s) {
  allow_large_lds(reinterpret_cast<const void*>(gradient_walk_lut_fused_kernel<RESCALE, KP>), lds);
  hipLaunchKernelGGL((gradient_walk_lut_fused_kernel<RESCALE, KP>), grid, dim3(kTile), lds, s, a, f);
}
void launch_gradient_walk_lut_fused(const LikArgs& a_in, const FusedSetupArgs& f_in, int count, bool rescale,
                                    hipStream_t s) {
  if (count <= 0) return;
  LikArgs a = a_in;
  FusedSetupArgs f = f_in;
  a.kp = lut_kp(a.K);
  a.cat_groups = 1;
  a.walk_evals = count;
  f.setup_blocks = kSetupQuarters * count;
  f.xcd_base = f.setup_blocks & 7;  // (the walk workgroups follow the set-up waves in this launch)
  // how long a walk wave polls before it gives up: one second of the 100 MHz clock by default
  if (f.spin_ticks <= 0) f.spin_ticks = 100 * 1000 * 1000;
  const int gtiles = gradient_mfma_tiles(a.P, a.K);
  const dim3 grid((unsigned)((size_t)count * gtiles + f.setup_blocks));
  const size_t lds = std::max<size_t>(gradient_walk_lds_bytes(a.n, a.K, rescale, false), fused_setup_lds(a.n, a.K));
  switch ((rescale ? 8 : 0) | a.kp) {
    case 1: launch_fused_variant<false, 1>(a, f, grid, lds, s); break;
    case 2: launch_fused_variant<false, 2>(a, f, grid, lds, s); break;
    case 4: launch_fused_variant<false, 4>(a, f, grid, lds, s); break;
    case 9: launch_fused_variant<true, 1>(a, f, grid, lds, s); break;
    case 10: launch_fused_variant<true, 2>(a, f, grid, lds, s); break;
    default: launch_fused_variant<true, 4>(a, f, grid, lds, s); break;
  }
}

void launch_setup_records(const FusedSetupArgs& f_in, int count, hipStream_t s) {
  if (count <= 0) return;
  FusedSetupArgs f = f_in;
  f.setup_blocks = kSetupQuarters * count;
  f.xcd_base = 0;  // (the walk's launch numbers its workgroups from zero)
  f.ready = nullptr;
  f.fence = 0;
  f.debug_skip = 0;
  const size_t lds = fused_setup_lds(f.ts.n, f.ms.K);
  allow_large_lds(reinterpret_cast<const void*>(setup_records_kernel), lds);
  hipLaunchKernelGGL(setup_records_kernel, dim3((unsigned)f.setup_blocks), dim3(kTile), lds, s, f);
}

void launch_transition_lut(const TransitionMacroArgs& a, hipStream_t s) {
  if (a.count <= 0) return;
  const int per_eval = (a.N - 1) * a.K;
  const dim3 grid(a.count, (per_eval + kTlBlock - 1) / kTlBlock);
  hipLaunchKernelGGL(transition_lut_kernel, grid, dim3(kTlBlock), sizeof(int) * (size_t)(a.N - 1), s, a);
}

// The third generation takes every call of an engine whose tips are one-hot / all ones with one
// category group (K <= 4) and no analytic substitution gradient: stored vectors in LDS or, for
// the larger trees, in the arena (round 6; until then K = 3, 4 with the vectors in LDS only).
// (R = 1 -- four patterns per wave -- is not worth the look-up walk's per-visit cost: the
// engine's tile-width choice never pairs them)
bool gradient_walk_lut_applies(int K) { return K >= 1 && K <= 4 && kRegs >= 2; }

template <bool RESCALE, bool ARENA, int KP, int RR>
static void launch_lut_variant(const LikArgs& a, dim3 grid, size_t lds, hipStream_t s, bool resident) {
  // (non-temporal arena accesses up to 48 default-width pattern tiles per tree, and for a call
  // whose waves are all resident at once -- 64 taxa x 1000 x 4 categories, 16 trees: 0.0990
  // against 0.1025 ms, tools/audit_paths.py: walk_lut_body; MI_PHYLO_ARENA_NT=0|1 forces plain /
  // non-temporal)
  // (read per launch: tests switch it between calls of one process)
  const char* nt_env = getenv("MI_PHYLO_ARENA_NT");
  const int forced = nt_env ? atoi(nt_env) : -1;
  if (ARENA && (forced < 0 ? (a.g_tiles * RR <= 48 * kRegs || resident) : forced != 0)) {
    allow_large_lds(reinterpret_cast<const void*>(gradient_walk_lut_kernel<RESCALE, ARENA, KP, ARENA, RR>), lds);
    hipLaunchKernelGGL((gradient_walk_lut_kernel<RESCALE, ARENA, KP, ARENA, RR>), grid, dim3(kTile), lds, s, a);
    return;
  }
  allow_large_lds(reinterpret_cast<const void*>(gradient_walk_lut_kernel<RESCALE, ARENA, KP, false, RR>), lds);
  hipLaunchKernelGGL((gradient_walk_lut_kernel<RESCALE, ARENA, KP, false, RR>), grid, dim3(kTile), lds, s, a);
}
template <bool ARENA, int RR = kRegs>
static void launch_lut_store(const LikArgs& a, dim3 grid, size_t lds, bool rescale, hipStream_t s,
                             bool resident = false) {
  switch ((rescale ? 8 : 0) | a.kp) {
    case 1: launch_lut_variant<false, ARENA, 1, RR>(a, grid, lds, s, resident); break;
    case 2: launch_lut_variant<false, ARENA, 2, RR>(a, grid, lds, s, resident); break;
    case 4: launch_lut_variant<false, ARENA, 4, RR>(a, grid, lds, s, resident); break;
    case 9: launch_lut_variant<true, ARENA, 1, RR>(a, grid, lds, s, resident); break;
    case 10: launch_lut_variant<true, ARENA, 2, RR>(a, grid, lds, s, resident); break;
    default: launch_lut_variant<true, ARENA, 4, RR>(a, grid, lds, s, resident); break;
  }
}

void launch_gradient_walk_lut(const LikArgs& a_in, int count, bool rescale, hipStream_t s) {
  if (count <= 0) return;
  LikArgs a = a_in;
  a.kp = lut_kp(a.K);
  a.cat_groups = 1;
  a.walk_evals = count;
  // (the engine chose the store and the tile width: a.g_tiles follows them)
  const bool wide = kRegs < kRegsWide && a.tile_regs == kRegsWide;
  const int regs = wide ? kRegsWide : kRegs;
  const int gtiles = gradient_mfma_tiles(a.P, a.K, regs);
  const bool arena_variant =
      a.store ? a.store == 2
              : gradient_walk_use_arena(a.n, a.K, rescale, false, (size_t)gtiles * (size_t)count, true, regs);
  const dim3 grid((unsigned)((size_t)count * gtiles));
  if (arena_variant) {
    // two launches over one grid, as the second generation's arena variant: the trees whose
    // schedule fits the usual number of LDS slots, then (more LDS per wave) the rest
    const int usual = gradient_arena_slots_usual(a.n), sure = gradient_arena_slots_sure(a.n);
    auto lds_for = [&](int slots) { return gradient_walk_lds_bytes_for(a.n, a.K, rescale, false, slots, regs); };
    auto launch = [&](int slots, bool resident) {
      if (wide) launch_lut_store<true, kRegsWide>(a, grid, lds_for(slots), rescale, s, resident);
      else launch_lut_store<true>(a, grid, lds_for(slots), rescale, s, resident);
    };
    a.lds_lo = -1;
    // (ONE launch with the larger footprint only when all waves are resident at once anyway -- a
    // call of a few trees.  Round 6 tried it wherever the larger footprint costs no wave per CU,
    // up to ~60 taxa, to save the second launch's 35-50 us of waves that exit at once: no gain --
    // 59 x 1824: 4.51 against 4.34-4.40 ms, the others level)
    if (arena_single_launch(lds_for(sure), grid.x)) {
      a.lds_slots = sure;
      launch(sure, true);
      return;
    }
    a.lds_slots = usual;
    const bool resident = arena_single_launch(lds_for(usual), grid.x);
    launch(usual, resident);
    if (sure > usual) {
      a.lds_lo = usual;
      a.lds_slots = sure;
      launch(sure, resident);
    }
    return;
  }
  if (wide) {
    // (a wide-tile engine's call of a few trees: every vector in LDS, one wave per SIMD)
    const size_t lds = gradient_walk_lds_bytes(a.n, a.K, rescale, false, regs);
    auto go = [&](auto kernel) {
      allow_large_lds(reinterpret_cast<const void*>(kernel), lds);
      hipLaunchKernelGGL(kernel, grid, dim3(kTile), lds, s, a);
    };
    switch ((rescale ? 8 : 0) | a.kp) {
      case 1: go(gradient_walk_lut_kernel<false, false, 1, false, kRegsWide, 1>); break;
      case 2: go(gradient_walk_lut_kernel<false, false, 2, false, kRegsWide, 1>); break;
      case 4: go(gradient_walk_lut_kernel<false, false, 4, false, kRegsWide, 1>); break;
      case 9: go(gradient_walk_lut_kernel<true, false, 1, false, kRegsWide, 1>); break;
      case 10: go(gradient_walk_lut_kernel<true, false, 2, false, kRegsWide, 1>); break;
      default: go(gradient_walk_lut_kernel<true, false, 4, false, kRegsWide, 1>); break;
    }
    return;
  }
  launch_lut_store<false>(a, grid, gradient_walk_lds_bytes(a.n, a.K, rescale, false), rescale, s);
}
// The look-up walk's tip codes per pattern tile, in the form its LDS holds them (walk_lut_body,
// TileCodes): per (tile, taxon) one word per column (byte r: the code of register r's pattern) for
// three / four categories, one 16-bit field per column (4 bits per register: code / 16) for one /
// two.  Built once per engine for its tile width; columns past the last pattern repeat it.
namespace {
__global__ __launch_bounds__(256) void tip_code_tiles_kernel(const uint8_t* codes, uint8_t* out, int n, int P, int ppr,
                                                             int regs, int tile_count) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)tile_count * n * ppr) return;
  const int c = (int)(id % ppr), taxon = (int)((id / ppr) % n), tile = (int)(id / ((long)ppr * n));
  uint32_t word = 0, field = 0;
  for (int r = 0; r < regs; r++) {
    const int p = tile * (regs * ppr) + r * ppr + c;
    const uint32_t code = codes[(size_t)taxon * P + (p < P ? p : P - 1)];
    word |= code << (8 * r);
    field |= ((code >> 4) & 0x7u) << (4 * r);
  }
  if (ppr == 4) reinterpret_cast<uint32_t*>(out)[id] = word;
  else reinterpret_cast<uint16_t*>(out)[id] = (uint16_t)field;
}
}  // namespace
size_t tip_code_tiles_bytes(int n, int P, int K, int regs) {
  const int ppr = 16 / lut_kp(K);
  return (size_t)gradient_mfma_tiles(P, K, regs) * (size_t)n * ppr * (ppr == 4 ? 4 : 2);
}
void launch_tip_code_tiles(const uint8_t* codes, uint8_t* out, int n, int P, int K, int regs, hipStream_t s) {
  const int ppr = 16 / lut_kp(K), tiles = gradient_mfma_tiles(P, K, regs);
  const long total = (long)tiles * n * ppr;
  hipLaunchKernelGGL(tip_code_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, codes, out, n, P, ppr,
                     regs > 0 ? regs : kRegs, tiles);
}
const char* gradient_walk_lut_kernel_name() { return "gradient_walk_lut_kernel"; }
const char* gradient_walk_lut_fused_kernel_name() { return "gradient_walk_lut_fused_kernel"; }

}  // namespace miphylo
