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
// Scope: engines whose tip vectors are the five of SitePattern (one-hot / all ones:
// src/site_pattern.cpp:117-131 -- everything the reference produces), three or four rate
// categories (one category group), stored vectors in LDS, no analytic substitution gradient;
// everything else keeps the second generation.
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

constexpr int R = kLlR;                  // registers (16 columns each) per vector
constexpr unsigned kTwCol = 32;          // LDS bytes per (macro, column): six words, padded
constexpr unsigned kTStride = 4 * kTwCol;  // per macro: four pattern columns per register
constexpr unsigned kPos = 1280u, kVisit = 6u * kPos;
constexpr unsigned kTipCat = 320u, kTipRow = 80u;
// where a visit issues its scalar loads (slots of the next visit, shape three ahead): at its
// top, right behind the operand requests and the LDS read of the tip words, or at its end
// (W3_SCALARS: 0 at the visit's end, 1 at its top, 2 in its middle -- behind the visit's last
// LDS read, so that no LDS wait of the visit waits for them and they have the rest of the
// visit to arrive: scalar loads return out of order, any wait for LDS data with one of them
// in flight is a wait for everything)
#ifndef W3_SCALARS
#define W3_SCALARS 0
#endif
constexpr int kScalars = W3_SCALARS;
// W3_STORE_S: the post-order walk overwrites a stored node's vector L, once its parent's visit
// has consumed it, with the product S = P L it has just formed; the pre-order walk then reads
// S instead of recomputing it, and takes the edge's derivative as sum qs . Q S instead of
// sum (P^T qs) . Q L -- the same number because P and Q commute (what the 20-state kernels
// do).  36 matrix instructions of a DS1 tile job less and one dependent stage
// off the visit's chain, for 36 more LDS stores; results then differ from the second
// generation's in the last bits (P Q against Q P).
#ifndef W3_STORE_S
#define W3_STORE_S 0
#endif
constexpr bool kStoreS = W3_STORE_S != 0;
// when a visit reads the stored vectors of its node and of its stored children from LDS: first
// thing at its top, BEFORE the operand wait and the next visit's requests (their LDS latency
// then passes under those), or where the visit's arithmetic needs them
#ifndef W3_EARLY_LDS
#define W3_EARLY_LDS 0
#endif
constexpr bool kEarlyLds = W3_EARLY_LDS != 0;
// W3_SCHED_REGS: see load_shape / load_slots
#ifndef W3_SCHED_REGS
#define W3_SCHED_REGS 0
#endif

// operands of one child of a visit (see fetch_child)
template <bool PRE>
struct Ops {
  double x[PRE ? 14 : 7];
};

// Hand-off word of the one-launch small call (round 5), one per tree: the set-up waves of a
// tree add kReadyQuarter each when their share of the tree's operand records (wave 0 also the
// macro list and the model instance) has been written through to memory, wave 0 adds the macro
// count in the low byte; the walk waves of the tree wait for kSetupQuarters of them.
constexpr int kSetupQuarters = 4, kReadyQuarter = 1 << 8;
// polls before a walk wave gives up (0.2 us apart: a fifth of a second): a set-up wave that
// never ran is an error (status kFusedTimeout), not a hang
constexpr int kReadySpins = 1 << 20;

// block: this wave's number among the walk waves of the launch.  FUSED: `ready` is the
// hand-off word array (above), the launch's first workgroups are set-up waves.
template <bool RESCALE, bool FUSED>
__device__ __forceinline__ void walk_lut_body(const LikArgs& a, double* wlds, const int block,
                                              const int32_t* ready) {
  static_assert(R >= 1 && R <= 4, "a tip word holds one byte per register");
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
    // Wait for this tree's set-up waves.  Everything they hand over (macro list, operand
    // records, model instance) lies in cache lines of its own per tree (macro_stride, kVisit,
    // alignas(128) DevModel) that no wave of this launch reads before the word says so, and was
    // stored write-through (sc1) and waited for (vmcnt(0)) before the word was added to: the
    // first touch of such a line after the poll misses every cache of this CU and XCD (they
    // were invalidated when the kernel started) and is served with the stored bytes.
    int v = 0, spins = 0;
    for (;;) {
      v = __builtin_amdgcn_readfirstlane(
          __hip_atomic_load(ready + (size_t)t * kReadyStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if ((v >> 8) >= kSetupQuarters || ++spins >= kReadySpins) break;
      __builtin_amdgcn_s_sleep(8);
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
  int node_j[2] = {-1, -1};
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int j = lane + 64 * u;
    if (j < jmax) node_j[u] = mwv[(j / 6) * 16 + 1 + (j % 6)];
  }
  const int M = FUSED ? M_ready : __builtin_amdgcn_readfirstlane(a.macro_count[t]);
  if (M <= 0) {
    if (FUSED && M < 0 && lane == 0) set_status(a.status, kFusedTimeout, t);
    return;
  }
  constexpr int ppr = 4, TP = ppr * R;
  struct __attribute__((packed)) Bytes12 {  // (4 R bytes: the pair's codes of this tile)
    uint32_t d[R];
  };
  const int tile_start = tile * TP;
  const bool whole = tile_start + TP <= a.P;
  Bytes12 bytes_now[2] = {};
  if (whole) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int j = lane + 64 * u, node = node_j[u];
      if (j < jmax && (unsigned)node < (unsigned)n)
        bytes_now[u] = *reinterpret_cast<const Bytes12*>(a.tip_codes + (size_t)node * a.P + tile_start);
    }
  }
  const int cat = b, catc = cat < K ? cat : K - 1;
  // this lane's constant offsets into a position's record: internal (16-byte {f, tr} slot of
  // (category, lo, hi)) and tip (row hi of the category's table; the state's offset is added)
  const unsigned lane_moff = 16u * (unsigned)(catc * 16 + lo * 4 + hi);
  const unsigned lane_tip = (unsigned)catc * kTipCat + (unsigned)hi * kTipRow;
  const char* __restrict__ mm_g = reinterpret_cast<const char*>(a.mmats) + (size_t)gi * Mmax * kVisit;
  const int col = lo;  // this lane's pattern column; register r adds r * ppr
  const double pi_l = model->pi[hi];
  const double cw_l = cat < K ? model->cat_weight[cat] : 0.0;
  const double rate_l = model->cat_rate[catc], drate_l = model->cat_drate[catc];
  const double AQ = model->Q[lo * 4 + hi];  // A operand for Q L (same in every block)

  struct Tw {
    uint32_t w[6];
  };
  struct Slots {  // scalars (s_load_dwordx8)
    int q, c[2], g[4], dst;
  };
#if W3_SCHED_REGS
  // (experiment, round 5: the schedule words of the whole tree in nine vector registers --
  // lane m holds macro m's shape and its eight slot words -- read with v_readlane when a visit
  // needs them: no scalar-memory instruction inside the walk, so a wait for LDS data is no
  // longer a wait for scalar loads that return out of order)
  int sched_w[9];
  {
    const int ml = lane < Mmax ? lane : 0;
    sched_w[0] = mwv[ml * 16];
    const int4 lo4 = *reinterpret_cast<const int4*>(mwv + ml * 16 + 8);
    const int4 hi4 = *reinterpret_cast<const int4*>(mwv + ml * 16 + 12);
    sched_w[1] = lo4.x; sched_w[2] = lo4.y; sched_w[3] = lo4.z; sched_w[4] = lo4.w;
    sched_w[5] = hi4.x; sched_w[6] = hi4.y; sched_w[7] = hi4.z; sched_w[8] = hi4.w;
  }
  auto load_shape = [&](int m) { return __builtin_amdgcn_readlane(sched_w[0], m); };
  auto load_slots = [&](int m) {
    return Slots{__builtin_amdgcn_readlane(sched_w[1], m),
                 {__builtin_amdgcn_readlane(sched_w[2], m), __builtin_amdgcn_readlane(sched_w[3], m)},
                 {__builtin_amdgcn_readlane(sched_w[4], m), __builtin_amdgcn_readlane(sched_w[5], m),
                  __builtin_amdgcn_readlane(sched_w[6], m), __builtin_amdgcn_readlane(sched_w[7], m)},
                 __builtin_amdgcn_readlane(sched_w[8], m)};
  };
#else
  auto load_shape = [&](int m) { return mw[m * 16]; };
  auto load_slots = [&](int m) {
    const cint_ptr p = mw + m * 16 + 8;
    return Slots{p[0], {p[1], p[2]}, {p[3], p[4], p[5], p[6]}, p[7]};
  };
#endif
  // LDS: [macro][column][8 words: tip codes of positions 0..5, one byte per register r] --
  // re-used, macro by macro, for that macro's edge sums once its tip words were consumed --
  // | vectors [slot][r][lane] | RESCALE: exponents
  char* const lds0 = reinterpret_cast<char*>(wlds);
  char* const plv = lds0 + (unsigned)Mmax * kTStride;
  int16_t* exps = reinterpret_cast<int16_t*>(plv + (size_t)max_stored(n) * R * kTile * 8);
  const unsigned tw_lane = (unsigned)col * kTwCol;
  auto fetch_tw = [&](int m) {  // the six tip words of visit m (LDS)
    Tw t;
    const char* twp = lds0 + ((unsigned)m * kTStride + tw_lane);
    const uint4 w4 = *reinterpret_cast<const uint4*>(twp);
    const uint2 w2 = *reinterpret_cast<const uint2*>(twp + 16);
    t.w[0] = w4.x;
    t.w[1] = w4.y;
    t.w[2] = w4.z;
    t.w[3] = w4.w;
    t.w[4] = w2.x;
    t.w[5] = w2.y;
    return t;
  };

  // ---- operands of one child (J = 0, 1) of a visit, requested a visit ahead ----
  // W doubles per internal position ({f} post-order, {f, tr} pre-order), 3 W per tip position
  // ({P} / {P, P Q} per register).  One register group per child, laid out by its kind:
  //   tip:      x[0 .. 3W)                      stored: x[0 .. W)
  //   unstored: x[0 .. W), first grandchild at x[W ..), second at x[4W ..) (tip: 3W, else W)
  // (destinations are compile-time indices into the group: the groups live in registers)
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
  auto load_tip = [&](auto pre_tag, const char* at, uint32_t word, auto& o, auto off_tag) {
    constexpr bool PRE = decltype(pre_tag)::value;
    constexpr int OFF = decltype(off_tag)::value;
#pragma unroll
    for (int r = 0; r < R; r++) {
#ifdef W3_ABL_COALESCED_TIPS  // (timing experiment, wrong results: every lane its row's entry 0)
      unsigned voff = lane_tip + ((word >> (8 * r)) & 0x0u);
#else
      unsigned voff = lane_tip + ((word >> (8 * r)) & 0xffu);
#endif
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
  auto fetch_child = [&](auto pre_tag, auto jtag, int sh, const Tw& tw, const char* sb, const char* sb4,
                         auto& o) {
    constexpr bool PRE = decltype(pre_tag)::value;
    constexpr int J = decltype(jtag)::value;
    constexpr int W = PRE ? 2 : 1;
    using O0 = std::integral_constant<int, 0>;
    using OA = std::integral_constant<int, W>;
    using OB = std::integral_constant<int, 4 * W>;
    const int kind = (sh >> (2 * J)) & 3;
    if (kind == 0) {
      load_tip(pre_tag, sb + J * kPos, tw.w[J], o, O0{});
      return;
    }
    load_internal(pre_tag, sb + J * kPos, o, O0{});
    if (kind != 2) return;
    // positions 2, 3 (J = 0) lie below the 4095-byte immediate, 4, 5 (J = 1) beyond it
    const char* ga = J == 0 ? sb + 2 * kPos : sb4;
    const char* gb = J == 0 ? sb + 3 * kPos : sb4 + kPos;
    if (sh & (1 << (10 + 2 * J))) load_tip(pre_tag, ga, tw.w[2 + 2 * J], o, OA{});
    else load_internal(pre_tag, ga, o, OA{});
    if (sh & (1 << (11 + 2 * J))) load_tip(pre_tag, gb, tw.w[3 + 2 * J], o, OB{});
    else load_internal(pre_tag, gb, o, OB{});
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
    fetch_child(pre_tag, J0{}, sh, tw, sb, sb4, o0);
    fetch_child(pre_tag, J1{}, sh, tw, sb, sb4, o1);
#ifdef W3_STAMPS
    stamp_fetch += __builtin_amdgcn_s_memtime() - f0;
    stamp_visits++;
#endif
  };

  // W3_SPLIT_FETCH: child 0's operands are requested at the top of the visit before, child 1's
  // from that visit's middle (behind its LDS reads): two shorter bursts of loads instead of one
  // (a wave stalls in the issue of a load while the CU's address path is full: DESIGN.md 4.1)
#ifndef W3_SPLIT_FETCH
#define W3_SPLIT_FETCH 0
#endif
  constexpr bool kSplitFetch = W3_SPLIT_FETCH != 0;
  auto fetch_j = [&](auto pre_tag, auto jtag, int m, int sh, const Tw& tw, auto& o) {
    const char* sb = mm_g + (size_t)((unsigned)m * kVisit);
    unsigned off4 = 4 * kPos;
    asm volatile("" : "+s"(off4));
    const char* sb4 = sb + off4;
    fetch_child(pre_tag, jtag, sh, tw, sb, sb4, o);
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
      char* dst = lds0 + (unsigned)m * kTStride + (unsigned)pos * 4u;
      for (int q = 0; q < TP; q++) {
        const int pp = tile_start + q < a.P ? tile_start + q : a.P - 1;
        dst[(q & (ppr - 1)) * kTwCol + (q >> 2)] = (char)src[pp];
      }
    };
    if (whole) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int j = lane + 64 * u, node = node_j[u];
        if (j < jmax && (unsigned)node < (unsigned)n) {
          const Bytes12 w = bytes_now[u];
          const int m = j / 6, pos = j - m * 6;
          char* dst = lds0 + (unsigned)m * kTStride + (unsigned)pos * 4u;
#pragma unroll
          for (int c = 0; c < 4; c++) {
            uint32_t word = 0;  // byte r: the code of column c in register r
#pragma unroll
            for (int r = 0; r < R; r++) word |= ((w.d[r] >> (8 * c)) & 0xffu) << (8 * r);
            *reinterpret_cast<uint32_t*>(dst + c * kTwCol) = word;
          }
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

  struct V {
    double v[R];
  };
  const unsigned lane8 = 8u * lane;
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
  auto tip_pq = [&](const Ops<true>& o, auto off_tag) {
    constexpr int OFF = decltype(off_tag)::value;
    V y;
#pragma unroll
    for (int r = 0; r < R; r++) y.v[r] = o.x[OFF + 2 * r + 1];
    return y;
  };

  double qroot[R];  // root pre-order vector: pi * category weight * w_p / site likelihood
  int esum[R];      // RESCALE: exponents removed so far, per pattern
#pragma unroll
  for (int r = 0; r < R; r++) esum[r] = 0;

  // What a visit knows of one child: S = P_c L_c (the message to the parent / the sibling),
  // and for an internal child its vector L (stored: from its slot; unstored: Ap o Bp, the
  // products of its two children, whose vectors xa / xb are read only where they are stored
  // nodes).
  struct Child {
    V S, L, xa, xb, Ap, Bp;
  };
  struct Early {  // a visit's early LDS reads: q of its node (pre-order), its stored children
    V q, l0, l1;
  };
  auto early_loads = [&](bool with_q, int sh, const Slots& sl, Early& ea) {
    if (with_q) ea.q = load_slot(sl.q);
    if ((sh & 3) == 1) ea.l0 = load_slot(sl.c[0]);
    if (((sh >> 2) & 3) == 1) ea.l1 = load_slot(sl.c[1]);
  };
  auto child_S = [&](auto pre_tag, auto jtag, int sh, const auto& o, const Slots& sl, Child& c,
                     const Early& ea) {
    constexpr bool PRE = decltype(pre_tag)::value;
    constexpr int J = decltype(jtag)::value;
    constexpr int W = PRE ? 2 : 1;
    using O0 = std::integral_constant<int, 0>;
    using OA = std::integral_constant<int, W>;
    using OB = std::integral_constant<int, 4 * W>;
    const int kind = (sh >> (2 * J)) & 3;
    if (kind == 0) {
      c.S = tip_p(pre_tag, o, O0{});
      return;
    }
    if (kind == 1) {
      c.L = kEarlyLds ? (J == 0 ? ea.l0 : ea.l1) : load_slot(sl.c[J]);
      if (kStoreS && PRE) {  // (the slot holds S since the post-order walk)
        c.S = c.L;
        return;
      }
    } else {
      if (sh & (1 << (10 + 2 * J))) {
        c.Ap = tip_p(pre_tag, o, OA{});
      } else {
        c.xa = load_slot(sl.g[2 * J]);
        if (kStoreS && PRE) {
          c.Ap = c.xa;
        } else {
          c.Ap = mm(o.x[W], c.xa);
          if (kStoreS) store_slot(sl.g[2 * J], c.Ap);
        }
      }
      if (sh & (1 << (11 + 2 * J))) {
        c.Bp = tip_p(pre_tag, o, OB{});
      } else {
        c.xb = load_slot(sl.g[2 * J + 1]);
        if (kStoreS && PRE) {
          c.Bp = c.xb;
        } else {
          c.Bp = mm(o.x[4 * W], c.xb);
          if (kStoreS) store_slot(sl.g[2 * J + 1], c.Bp);
        }
      }
      c.L = mul(c.Ap, c.Bp);
    }
    c.S = mm(o.x[0], c.L);
    if (kStoreS && !PRE && kind == 1) store_slot(sl.c[J], c.S);
  };

  // ================= post-order over the stored nodes, then the root (site likelihood) ====
  auto post_visit = [&](auto root_tag, int sh, const Slots& sl, const Ops<false>& o0, const Ops<false>& o1,
                        int tile_for_ll, const Early& ea, auto&& mid) {
    constexpr bool ROOT = decltype(root_tag)::value;
    Child c0, c1;
    child_S(Post{}, J0{}, sh, o0, sl, c0, ea);
    child_S(Post{}, J1{}, sh, o1, sl, c1, ea);
    mid();  // (the visit's LDS reads are behind it)
    V Lv = mul(c0.S, c1.S);
    if (!ROOT) {
      if (RESCALE) {
#pragma unroll
        for (int r = 0; r < R; r++) {
          const double colsum = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, Lv.v[r], 0.0, 0, 0, 0);
          int ex = colsum > 0.0 ? __builtin_amdgcn_frexp_exp(colsum) : -4096;
          ex = max(ex, __shfl_xor(ex, 4, 64));
          ex = max(ex, __shfl_xor(ex, 8, 64));
          ex = ex == -4096 ? 0 : ex;
          Lv.v[r] = ldexp(Lv.v[r], -ex);
          esum[r] += ex;
          exps[(unsigned)sl.q * (unsigned)TP + (unsigned)(r * ppr + col)] = (int16_t)ex;
        }
      }
      store_slot(sl.q, Lv);
    } else {
      // root: site likelihood per pattern, log-likelihood partial, derivative weights
      double sitev[R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        double v = cw_l * pi_l * Lv.v[r];
        v = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, v, 0.0, 0, 0, 0);  // states
        v = row_ror_add<8>(v);                                           // categories
        v = row_ror_add<4>(v);
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
      if (hi < R && b == 0 && pv < a.P)
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
    // (LDS).  The visit ends with the scalar loads: slots of m + 1, shape of m + 3.
    Ops<false> a0, a1, b0, b1;
    int s0 = sh_a, s1 = sh_b, s2 = sh_c;
    Slots la = sl_a, lb;
    Tw tw = fetch_tw(0);
    fetch(Post{}, 0, s0, tw, a0, a1);
    tw = fetch_tw(min(1, M1));
    for (int m = 0; m < M1; m += 2) {
      // ---- visit m (set A) ----
      Early ea;
      if (kEarlyLds) early_loads(false, s0, la, ea);
      settle(a0, a1);
      const Tw tw_a = tw;
      if (kSplitFetch) fetch_j(Post{}, J0{}, min(m + 1, M1), s1, tw_a, b0);
      else fetch(Post{}, min(m + 1, M1), s1, tw, b0, b1);
      tw = fetch_tw(min(m + 2, M1));
      int s3;
      auto req = [&]() {
        lb = load_slots(min(m + 1, M1));
        s3 = load_shape(min(m + 3, M1));
      };
      if (kScalars == 1) req();
      post_visit(Inner{}, s0, la, a0, a1, 0, ea, [&]() {
        if (kScalars == 2) req();
        if (kSplitFetch) fetch_j(Post{}, J1{}, min(m + 1, M1), s1, tw_a, b1);
      });
      if (kScalars == 0) req();
      s0 = s1;
      s1 = s2;
      s2 = s3;
      if (m + 1 < M1) {
        // ---- visit m + 1 (set B) ----
        Early eb;
        if (kEarlyLds) early_loads(false, s0, lb, eb);
        settle(b0, b1);
        const Tw tw_b = tw;
        if (kSplitFetch) fetch_j(Post{}, J0{}, min(m + 2, M1), s1, tw_b, a0);
        else fetch(Post{}, min(m + 2, M1), s1, tw, a0, a1);
        tw = fetch_tw(min(m + 3, M1));
        int s4;
        auto req = [&]() {
          la = load_slots(min(m + 2, M1));
          s4 = load_shape(min(m + 4, M1));
        };
        if (kScalars == 1) req();
        post_visit(Inner{}, s0, lb, b0, b1, 0, eb, [&]() {
          if (kScalars == 2) req();
          if (kSplitFetch) fetch_j(Post{}, J1{}, min(m + 2, M1), s1, tw_b, a1);
        });
        if (kScalars == 0) req();
        s0 = s1;
        s1 = s2;
        s2 = s4;
      } else {  // M1 odd: the root's operands arrived in set B
        a0 = b0;
        a1 = b1;
        la = lb;
      }
    }
    {
      Early ea;
      if (kEarlyLds) early_loads(false, s0, la, ea);
      settle(a0, a1);
      post_visit(Root{}, s0, la, a0, a1, tile, ea, [] {});
    }
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
  auto child_edges = [&](auto jtag, int sh, const Ops<true>& o, const Slots& sl, const Child& c, const V& qs,
                         int m, V& nout) {
    constexpr int J = decltype(jtag)::value;
    const int kind = (sh >> (2 * J)) & 3;
    if (kind == 0) {
      nout = mul(qs, tip_pq(o, std::integral_constant<int, 0>{}));
      return;
    }
    V qc;
    if (kStoreS && kind == 1) {  // sum qs . Q S = sum (P^T qs) . Q L
      qc = mm(o.x[1], qs);
      nout = mul(qs, mm(AQ, c.S));
    } else {
      nout = inner_edge(o.x[1], qs, c.L, qc);
    }
    if (kind == 1) {
      store_slot(sl.c[J], qc);
      return;
    }
    V na, nb, qa, qb;
    const V qsa = mul(qc, c.Bp), qsb = mul(qc, c.Ap);
    if (sh & (1 << (10 + 2 * J))) {
      na = mul(qsa, tip_pq(o, std::integral_constant<int, 2>{}));
    } else {
      if (kStoreS) {
        qa = mm(o.x[3], qsa);
        na = mul(qsa, mm(AQ, c.Ap));
      } else {
        na = inner_edge(o.x[3], qsa, c.xa, qa);
      }
      store_slot(sl.g[2 * J], qa);
    }
    if (sh & (1 << (11 + 2 * J))) {
      nb = mul(qsb, tip_pq(o, std::integral_constant<int, 8>{}));
    } else {
      if (kStoreS) {
        qb = mm(o.x[9], qsb);
        nb = mul(qsb, mm(AQ, c.Bp));
      } else {
        nb = inner_edge(o.x[9], qsb, c.xb, qb);
      }
      store_slot(sl.g[2 * J + 1], qb);
    }
    edge_sums(na, nb, m, 2 + 2 * J);
  };
  auto pre_visit = [&](auto root_tag, int sh, const Slots& sl, const Ops<true>& o0, const Ops<true>& o1, int m,
                       const Early& ea, auto&& mid) {
    constexpr bool ROOT = decltype(root_tag)::value;
    V qv;
    if (ROOT) {
#pragma unroll
      for (int r = 0; r < R; r++) qv.v[r] = qroot[r];
    } else {
      qv = kEarlyLds ? ea.q : load_slot(sl.q);
      if (RESCALE) {
#pragma unroll
        for (int r = 0; r < R; r++)
          qv.v[r] = ldexp(qv.v[r], -(int)exps[(unsigned)sl.q * (unsigned)TP + (unsigned)(r * ppr + col)]);
      }
    }
    Child c0, c1;
    child_S(Pre{}, J0{}, sh, o0, sl, c0, ea);
    child_S(Pre{}, J1{}, sh, o1, sl, c1, ea);
    mid();  // (the visit's LDS reads are behind it)
    V n0, n1;
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
    Ops<true> a0, a1, b0, b1;
    int s0 = load_shape(M1), s1 = load_shape(dn(M1 - 1)), s2 = load_shape(dn(M1 - 2));
    Slots la = load_slots(M1), lb;
    Tw tw = fetch_tw(M1);
    fetch(Pre{}, M1, s0, tw, a0, a1);
    tw = fetch_tw(dn(M1 - 1));
    {  // ---- visit M1 (the root, set A) ----
      Early ea;
      if (kEarlyLds) early_loads(false, s0, la, ea);
      settle(a0, a1);
      fetch(Pre{}, dn(M1 - 1), s1, tw, b0, b1);
      tw = fetch_tw(dn(M1 - 2));
      int s3;
      auto req = [&]() {
        lb = load_slots(dn(M1 - 1));
        s3 = load_shape(dn(M1 - 3));
      };
      if (kScalars == 1) req();
      pre_visit(Root{}, s0, la, a0, a1, M1, ea, [&]() {
        if (kScalars == 2) req();
      });
      if (kScalars == 0) req();
      s0 = s1;
      s1 = s2;
      s2 = s3;
    }
    for (int m = M1 - 1; m >= 0; m -= 2) {
      // ---- visit m (set B) ----
      Early eb;
      if (kEarlyLds) early_loads(true, s0, lb, eb);
      settle(b0, b1);
      const Tw tw_b = tw;
      if (kSplitFetch) fetch_j(Pre{}, J0{}, dn(m - 1), s1, tw_b, a0);
      else fetch(Pre{}, dn(m - 1), s1, tw, a0, a1);
      tw = fetch_tw(dn(m - 2));
      int s3;
      auto req = [&]() {
        la = load_slots(dn(m - 1));
        s3 = load_shape(dn(m - 3));
      };
      if (kScalars == 1) req();
      pre_visit(Inner{}, s0, lb, b0, b1, m, eb, [&]() {
        if (kScalars == 2) req();
        if (kSplitFetch) fetch_j(Pre{}, J1{}, dn(m - 1), s1, tw_b, a1);
      });
      if (kScalars == 0) req();
      s0 = s1;
      s1 = s2;
      s2 = s3;
      if (m >= 1) {
        // ---- visit m - 1 (set A) ----
        Early ea;
        if (kEarlyLds) early_loads(true, s0, la, ea);
        settle(a0, a1);
        const Tw tw_a = tw;
        if (kSplitFetch) fetch_j(Pre{}, J0{}, dn(m - 2), s1, tw_a, b0);
        else fetch(Pre{}, dn(m - 2), s1, tw, b0, b1);
        tw = fetch_tw(dn(m - 3));
        int s4;
        auto req = [&]() {
          lb = load_slots(dn(m - 2));
          s4 = load_shape(dn(m - 4));
        };
        if (kScalars == 1) req();
        pre_visit(Inner{}, s0, la, a0, a1, m - 1, ea, [&]() {
          if (kScalars == 2) req();
          if (kSplitFetch) fetch_j(Pre{}, J1{}, dn(m - 2), s1, tw_a, b1);
        });
        if (kScalars == 0) req();
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

template <bool RESCALE>
// (R = 2 -- 8 patterns per wave, 14 KB of LDS -- is built with three waves per SIMD:
// `make EXTRA_LLVM=-DMI_LLR=2`, an experiment of round 4)
__global__ __launch_bounds__(kTile, R == 2 ? 3 : 2) void gradient_walk_lut_kernel(LikArgs a) {
  extern __shared__ double wlds[];
  walk_lut_body<RESCALE, false>(a, wlds, blockIdx.x, nullptr);
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
// nothing.  Should a walk wave ever wait in vain, it gives up after kReadySpins polls with
// the sticky status kFusedTimeout -- the call fails with a message instead of hanging.
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
  const int t = bid / kSetupQuarters, q = bid - t * kSetupQuarters;
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
  if (lane == 0)
    __hip_atomic_fetch_add(f.ready + (size_t)t * kReadyStride, kReadyQuarter + (q == 0 ? M : 0), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

template <bool RESCALE>
__global__ __launch_bounds__(kTile, R == 2 ? 3 : 2) void gradient_walk_lut_fused_kernel(LikArgs a, FusedSetupArgs f) {
  extern __shared__ double wlds[];
  if ((int)blockIdx.x < f.setup_blocks) {
    fused_setup_role(f, blockIdx.x, reinterpret_cast<char*>(wlds));
    return;
  }
  walk_lut_body<RESCALE, true>(a, wlds, (int)blockIdx.x - f.setup_blocks, f.ready);
#ifdef W3_SETUP_STAMPS
  if (threadIdx.x == 0 && (((int)blockIdx.x - f.setup_blocks) % 1499) == 0)
    printf("walk block %d: out %lld\n", (int)blockIdx.x - f.setup_blocks, (long long)__builtin_amdgcn_s_memrealtime());
#endif
}

}  // namespace

size_t gradient_walk_lut_mats_bytes_per_eval(int n) { return (size_t)max_macros(n) * kVisit; }

// trees whose arrays fit one register per lane (64 nodes), a quarter of the (node, category)
// pairs per set-up wave
bool gradient_walk_lut_fused_applies(int n, int K) {
  return gradient_walk_lut_applies(K) && 2 * n - 1 <= 64 && ((2 * n - 2) * K + 3) / 4 <= 64;
}

void launch_gradient_walk_lut_fused(const LikArgs& a_in, const FusedSetupArgs& f_in, int count, bool rescale,
                                    hipStream_t s) {
  if (count <= 0) return;
  LikArgs a = a_in;
  FusedSetupArgs f = f_in;
  a.kp = 4;
  a.cat_groups = 1;
  a.walk_evals = count;
  f.setup_blocks = kSetupQuarters * count;
  const int gtiles = gradient_mfma_tiles(a.P, a.K);
  const dim3 grid((unsigned)((size_t)count * gtiles + f.setup_blocks));
  const size_t lds = std::max<size_t>(gradient_walk_lds_bytes(a.n, a.K, rescale, false), fused_setup_lds(a.n, a.K));
  if (rescale) {
    allow_large_lds(reinterpret_cast<const void*>(gradient_walk_lut_fused_kernel<true>), lds);
    hipLaunchKernelGGL(gradient_walk_lut_fused_kernel<true>, grid, dim3(kTile), lds, s, a, f);
  } else {
    allow_large_lds(reinterpret_cast<const void*>(gradient_walk_lut_fused_kernel<false>), lds);
    hipLaunchKernelGGL(gradient_walk_lut_fused_kernel<false>, grid, dim3(kTile), lds, s, a, f);
  }
}

void launch_transition_lut(const TransitionMacroArgs& a, hipStream_t s) {
  if (a.count <= 0) return;
  const int per_eval = (a.N - 1) * a.K;
  const dim3 grid(a.count, (per_eval + kTlBlock - 1) / kTlBlock);
  hipLaunchKernelGGL(transition_lut_kernel, grid, dim3(kTlBlock), sizeof(int) * (size_t)(a.N - 1), s, a);
}

// the third generation takes calls the second would run with its stored vectors in LDS, one
// category group of three or four categories
// (R = 1 -- four patterns per wave -- is not worth the look-up walk's per-visit cost: the
// engine's tile-width choice never pairs them)
bool gradient_walk_lut_applies(int K) { return (K == 3 || K == 4) && R >= 2; }

void launch_gradient_walk_lut(const LikArgs& a_in, int count, bool rescale, hipStream_t s) {
  if (count <= 0) return;
  LikArgs a = a_in;
  a.kp = 4;
  a.cat_groups = 1;
  a.walk_evals = count;
  const int gtiles = gradient_mfma_tiles(a.P, a.K);
  const dim3 grid((unsigned)((size_t)count * gtiles));
  const size_t lds = gradient_walk_lds_bytes(a.n, a.K, rescale, false);
  if (rescale) {
    allow_large_lds(reinterpret_cast<const void*>(gradient_walk_lut_kernel<true>), lds);
    hipLaunchKernelGGL(gradient_walk_lut_kernel<true>, grid, dim3(kTile), lds, s, a);
  } else {
    allow_large_lds(reinterpret_cast<const void*>(gradient_walk_lut_kernel<false>), lds);
    hipLaunchKernelGGL(gradient_walk_lut_kernel<false>, grid, dim3(kTile), lds, s, a);
  }
}
const char* gradient_walk_lut_kernel_name() { return "gradient_walk_lut_kernel"; }
const char* gradient_walk_lut_fused_kernel_name() { return "gradient_walk_lut_fused_kernel"; }

}  // namespace miphylo
