// The arena variant's slot assignment for one tree by one workgroup (macro_slots_wg_kernel,
// kernels_gradient.hip -- and, since round 6, the tail of tree_setup_wg_kernel, kernels_setup.hip:
// the same workgroup-per-tree mapping, one launch and one trip of the macro list through memory
// less per arena call).  See kernels_gradient.hip for the algorithm.
#pragma once
#include <hip/hip_runtime.h>

#include "mi_phylo_device_utils.h"
#include "mi_phylo_kernels.h"

namespace miphylo {
namespace {
using namespace dev;

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

// slot_scratch: the workgroup's dynamic LDS, macro_slots_wg_lds_bytes(n) of it; M: macro_count[t]
// (wave-uniform); every thread of the workgroup calls it (barriers inside).
inline size_t macro_slots_wg_lds_bytes(int n) {
  return sizeof(int32_t) * ((size_t)max_macros(n) * (16 + 2 + 2 + 1 + 1) + (size_t)max_stored(n));
}
__device__ __forceinline__ void macro_slots_wg_body(int32_t* slot_scratch, const MacroEntry* macros_in,
                                                    MacroEntry* macros_out, const int M, const int n, const int t,
                                                    int32_t* need, const int sure, int32_t* status) {
  __shared__ int more[3], used_s, placed_s, wave_tot[16];
  const int tid = threadIdx.x, nthreads = blockDim.x;
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

}  // namespace
}  // namespace miphylo
