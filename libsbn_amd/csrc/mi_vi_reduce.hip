// Caller-side reductions of one variational-inference step, on the device
// (vip/burrito.py:143-166: sum of the per-tree log-likelihoods; vip/branch_model.py:125-132:
// per-tree branch gradients scatter-added by split index).
//
// No float atomics and a FIXED order of every sum: the (tree, node) entries are sorted by their
// index with a STABLE radix sort (rocPRIM, keys = the index, values = the entry number), so
// the entries of one index stay in (tree, node) order; the first entry of each run then adds
// its run front to back -- exactly the order of a host loop over (t, v), the order np.add.at
// uses.  Work is O(entries), whatever the number of indices (round 3's kernel had the thread
// that owns an index scan ALL entries: 17 workgroups x 53 000 entries, 2.85 ms for 1000 DS1
// trees and 4096 indices; this form: see profiles/).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>  // before rocprim: its texture iterator calls memset from host code

#include <rocprim/rocprim.hpp>

#include "mi_phylo_kernels.h"

namespace miphylo {
namespace {

// entry e = (t, v): key = its index, or index_count for "not a parameter" (sorted to the end)
__global__ __launch_bounds__(256) void vi_keys_kernel(ViReduceArgs a, uint32_t* keys,
                                                      uint32_t* entries) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  if (blockIdx.x == gridDim.x - 1) {  // the last workgroup: the two scalar sums
    for (int which = 0; which < 2; which++) {
      const double* x = which ? a.site : a.ll;
      double s = 0;
      if (x)
        for (int t = tid; t < a.T; t += 256) s += (a.tree_weights ? a.tree_weights[t] : 1.0) * x[t];
      red[tid] = s;
      __syncthreads();
      for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
      }
      if (tid == 0) a.out_sums[which] = red[0];
      __syncthreads();
    }
    return;
  }
  const long total = (long)a.T * a.N;
  const long stride = (long)(gridDim.x - 1) * 256;
  for (long e = (long)blockIdx.x * 256 + tid; e < total; e += stride) {
    const int32_t k = a.branch_index[e];
    keys[e] = k < 0 || k >= a.index_count ? (uint32_t)a.index_count : (uint32_t)k;
    entries[e] = (uint32_t)e;
  }
  // indices that no entry carries keep a zero
  for (long k = (long)blockIdx.x * 256 + tid; k < a.index_count; k += stride)
    a.out_index_gradient[k] = 0.0;
}

// sorted position p: the head of a run (first entry of its index) adds the run in order
// -- ONE lane per run, entry after entry: that IS the definition of the result (the sum in
// (tree, node) order, what numpy's add.at and the reference's loops produce bit for bit), and
// it makes the longest run the kernel's time: with I indices spread evenly a run is T N / I
// entries (13 for the bench's 53 000 entries over 4 096 indices), but an index map that sends
// most branches to ONE index degenerates to a single lane adding T N numbers (~2 ns each:
// 0.1 ms per 53 000).  A chunked sum would be faster and differently rounded; callers with
// such maps can reduce per tree first.  (Entry numbers are 32-bit: the entry points refuse
// T N >= 2^32.)
__global__ __launch_bounds__(256) void vi_runs_kernel(ViReduceArgs a, const uint32_t* keys,
                                                      const uint32_t* entries) {
  const long total = (long)a.T * a.N;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= total) return;
  const uint32_t k = keys[p];
  if (k >= (uint32_t)a.index_count || (p > 0 && keys[p - 1] == k)) return;
  double sum = 0;
  for (long q = p; q < total && keys[q] == k; q++) {
    const uint32_t e = entries[q];
    sum += a.branch[e] * (a.tree_weights ? a.tree_weights[e / (uint32_t)a.N] : 1.0);
  }
  a.out_index_gradient[k] = sum;
}

int key_bits(int index_count) {  // keys are 0 .. index_count
  int bits = 1;
  while (bits < 32 && (1u << bits) <= (uint32_t)index_count) bits++;
  return bits;
}

}  // namespace

size_t vi_reduce_workspace_bytes(long entries, int index_count) {
  size_t need = 0;
  uint32_t* none = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, need, none, none, none, none, (size_t)std::max(entries, 1L),
                                  0, (unsigned)key_bits(index_count));
  // [keys | entries | sorted keys | sorted entries | rocPRIM's temporary storage]
  return 4 * sizeof(uint32_t) * (size_t)std::max(entries, 1L) + ((need + 255) & ~(size_t)255);
}

int launch_vi_reduce(const ViReduceArgs& a, void* workspace, size_t workspace_bytes, hipStream_t s) {
  const long total = (long)a.T * a.N;
  const unsigned blocks = (unsigned)std::min<long>((std::max<long>(total, a.index_count) + 255) / 256, 4096);
  uint32_t* keys = static_cast<uint32_t*>(workspace);
  uint32_t* entries = keys + std::max(total, 1L);
  uint32_t* keys2 = entries + std::max(total, 1L);
  uint32_t* entries2 = keys2 + std::max(total, 1L);
  void* tmp = entries2 + std::max(total, 1L);
  size_t tmp_bytes = workspace_bytes - 4 * sizeof(uint32_t) * (size_t)std::max(total, 1L);
  hipLaunchKernelGGL(vi_keys_kernel, dim3(std::max(blocks, 1u) + 1), dim3(256), 0, s, a, keys, entries);
  if (total <= 0) return 0;
  if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, entries, entries2, (size_t)total, 0,
                                (unsigned)key_bits(a.index_count), s) != hipSuccess)
    return 1;
  hipLaunchKernelGGL(vi_runs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, keys2,
                     entries2);
  return 0;
}

}  // namespace miphylo
