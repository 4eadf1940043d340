// Site-pattern compression on the device.
//
// Replaces SitePattern::Compress (reference src/site_pattern.cpp:77-115): the columns of
// an alignment are reduced to the distinct ones with their multiplicities, in the
// iteration order of the reference's std::unordered_map<vector<int>, double,
// IntVectorHasher> -- integer work, bit-exact target.
//
// That order is a function of (a) the hash of each distinct column and (b) the order in
// which distinct columns first appear: a column seen again only increments a counter.
// So the O(taxa x sites) part runs on the GPU --
//   1. hash every column (the reference's 32-bit hash, and a 64-bit one for grouping),
//   2. stable radix sort of (64-bit hash, site), group equal hashes, VERIFY that grouped
//      columns really are equal (a 64-bit collision is retried with another seed; there is
//      no CPU path in this library),
//   3. per group: first site (its smallest, thanks to the stable sort) and size,
//   4. groups ordered by first site,
// and the host replays P insertions (P = distinct columns) into a real
// std::unordered_map with a hasher that returns the stored reference hash.  Same
// container, same hashes, same insertion sequence => the same iteration order.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>  // before rocprim: its texture iterator calls memset from host code
#include <string>

#include <rocprim/rocprim.hpp>

#include <unordered_map>
#include <vector>

#include "../../include/mi_phylo.h"

namespace miphylo {
int fail(const std::string& message);  // mi_phylo_engine.cpp
}

namespace {

using miphylo::fail;

#define SP_TRY(expr)                                                                  \
  do {                                                                                \
    hipError_t err_ = (expr);                                                         \
    if (err_ != hipSuccess)                                                           \
      return fail(std::string(#expr) + ": " + hipGetErrorString(err_));               \
  } while (0)

// site_pattern.cpp:67-75 (boost hash_combine style, int arithmetic, int result)
__host__ __device__ inline int32_t ref_hash_step(int32_t h, int32_t c) {
  const uint32_t add = (uint32_t)c + 0x9e3779b9u + ((uint32_t)h << 6) + (uint32_t)(h >> 2);
  return (int32_t)((uint32_t)h ^ add);
}

__global__ void hash_columns_kernel(int n, long L, uint64_t seed, const int8_t* __restrict__ codes,
                                    int32_t* __restrict__ ref_hash, uint64_t* __restrict__ key,
                                    uint32_t* __restrict__ site) {
  const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= L) return;
  int32_t h = codes[s];
  uint64_t k = (0xcbf29ce484222325ull + seed) ^ (uint64_t)(uint8_t)codes[s];
  k *= 0x100000001b3ull;
  // The hashes are sequential in the taxon index but the loads are not: fetch eight rows
  // at a time (consecutive sites of a row: coalesced) so that the memory latency is paid
  // once per batch, not once per symbol.
  int t = 1;
  for (; t + 8 <= n; t += 8) {
    int8_t c[8];
#pragma unroll
    for (int u = 0; u < 8; u++) c[u] = codes[(size_t)(t + u) * L + s];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      h = ref_hash_step(h, c[u]);
      k = (k ^ (uint64_t)(uint8_t)c[u]) * 0x100000001b3ull;
      k ^= k >> 29;
    }
  }
  for (; t < n; t++) {
    const int c = codes[(size_t)t * L + s];
    h = ref_hash_step(h, c);
    k = (k ^ (uint64_t)(uint8_t)c) * 0x100000001b3ull;
    k ^= k >> 29;
  }
  k ^= k >> 32;
  k *= 0xd6e8feb86659fd93ull;
  k ^= k >> 32;
  ref_hash[s] = h;
  key[s] = k;
  site[s] = (uint32_t)s;
}

__global__ void mark_heads_kernel(int n, long L, const int8_t* __restrict__ codes,
                                  const uint64_t* __restrict__ key_sorted,
                                  const uint32_t* __restrict__ site_sorted,
                                  uint32_t* __restrict__ head, int32_t* __restrict__ collision) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const bool is_head = i == 0 || key_sorted[i] != key_sorted[i - 1];
  head[i] = is_head ? 1u : 0u;
  if (!is_head) {
    // same 64-bit hash as the previous element of the run: the columns must be equal
    const uint32_t a = site_sorted[i], b = site_sorted[i - 1];
    bool same = true;
    for (int t = 0; t < n; t++) same = same && codes[(size_t)t * L + a] == codes[(size_t)t * L + b];
    if (!same) atomicExch(collision, 1);
  }
}

__global__ void group_leaders_kernel(long L, const uint32_t* __restrict__ head,
                                     const uint32_t* __restrict__ group_end_incl,
                                     const uint32_t* __restrict__ site_sorted,
                                     uint32_t* __restrict__ leader, uint32_t* __restrict__ start) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L || !head[i]) return;
  const uint32_t g = group_end_incl[i] - 1;  // inclusive scan of the head flags
  leader[g] = site_sorted[i];
  start[g] = (uint32_t)i;
}

__global__ void gather_groups_kernel(uint32_t P, long L, const uint32_t* __restrict__ order_gid,
                                     const uint32_t* __restrict__ leader,
                                     const uint32_t* __restrict__ start,
                                     const int32_t* __restrict__ ref_hash,
                                     uint32_t* __restrict__ out_site, uint32_t* __restrict__ out_count,
                                     int32_t* __restrict__ out_hash) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= P) return;
  const uint32_t g = order_gid[j];
  const uint32_t first = leader[g];
  const uint32_t end = g + 1 < P ? start[g + 1] : (uint32_t)L;
  out_site[j] = first;
  out_count[j] = end - start[g];
  out_hash[j] = ref_hash[first];
}

// patterns[t][j] = codes[t][first_site[j]] for the distinct columns in their final order
__global__ void gather_patterns_kernel(int n, long L, uint32_t P, const int8_t* __restrict__ codes,
                                       const uint32_t* __restrict__ first_site,
                                       int32_t* __restrict__ patterns) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (j < P) patterns[(size_t)t * P + j] = codes[(size_t)t * L + first_site[j]];
}

__global__ void iota_kernel(uint32_t n, uint32_t* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}

struct StoredHash {
  const int32_t* hashes;
  size_t operator()(int j) const { return (size_t)hashes[j]; }  // int -> size_t as the reference
};

// The iteration order of the reference's map, from the distinct columns in
// first-appearance order and their reference hashes.
std::vector<int> replay_unordered_map_order(const std::vector<int32_t>& hashes) {
  std::unordered_map<int, double, StoredHash> seen(0, StoredHash{hashes.data()});
  for (int j = 0; j < (int)hashes.size(); j++) seen.insert({j, 1.});
  std::vector<int> order;
  order.reserve(hashes.size());
  for (const auto& kv : seen) order.push_back(kv.first);
  return order;
}

struct DevMem {
  void* p = nullptr;
  ~DevMem() {
    if (p) (void)hipFree(p);
  }
  template <typename T>
  T* as() { return static_cast<T*>(p); }
};

}  // namespace

// keep != nullptr: the device-resident form -- the gathered pattern matrix and the weights stay
// on the device and their pointers are handed out (out_patterns / out_weights unused)
struct KeepOnDevice {
  int32_t* patterns = nullptr;
  double* weights = nullptr;
};
static int32_t compress_impl(int32_t device, int32_t taxon_count, int64_t site_count,
                             const int8_t* codes, int32_t* out_pattern_count,
                             int32_t* out_patterns, double* out_weights,
                             double* out_hash_kernel_ms, KeepOnDevice* keep) {
  if (taxon_count <= 0 || site_count <= 0) return fail("empty alignment");
  if (!codes || !out_pattern_count || (!keep && (!out_patterns || !out_weights)))
    return fail("null argument");
  if (site_count > 0x7fffffffLL) return fail("more than 2^31 - 1 sites");
  const int n = taxon_count;
  const long L = (long)site_count;
  if (out_hash_kernel_ms) *out_hash_kernel_ms = 0.0;

  int device_count = 0;
  if (hipGetDeviceCount(&device_count) != hipSuccess || device_count == 0)
    return fail("no HIP device: site-pattern compression on the device needs a GPU "
                "(libmi_phylo_host.so has the CPU implementation)");
  SP_TRY(hipSetDevice(device < 0 ? 0 : device));

  DevMem d_codes, d_ref, d_key, d_key2, d_site, d_site2, d_head, d_scan, d_leader, d_start, d_gid,
      d_gid2, d_leader2, d_osite, d_ocount, d_ohash, d_flag, d_tmp;
  SP_TRY(hipMalloc(&d_codes.p, (size_t)n * L));
  SP_TRY(hipMalloc(&d_ref.p, sizeof(int32_t) * L));
  SP_TRY(hipMalloc(&d_key.p, sizeof(uint64_t) * L));
  SP_TRY(hipMalloc(&d_key2.p, sizeof(uint64_t) * L));
  SP_TRY(hipMalloc(&d_site.p, sizeof(uint32_t) * L));
  SP_TRY(hipMalloc(&d_site2.p, sizeof(uint32_t) * L));
  SP_TRY(hipMalloc(&d_head.p, sizeof(uint32_t) * L));
  SP_TRY(hipMalloc(&d_scan.p, sizeof(uint32_t) * L));
  SP_TRY(hipMalloc(&d_leader.p, sizeof(uint32_t) * L));
  SP_TRY(hipMalloc(&d_start.p, sizeof(uint32_t) * L));
  SP_TRY(hipMalloc(&d_gid.p, sizeof(uint32_t) * L));
  SP_TRY(hipMalloc(&d_gid2.p, sizeof(uint32_t) * L));
  SP_TRY(hipMalloc(&d_leader2.p, sizeof(uint32_t) * L));
  SP_TRY(hipMalloc(&d_osite.p, sizeof(uint32_t) * L));
  SP_TRY(hipMalloc(&d_ocount.p, sizeof(uint32_t) * L));
  SP_TRY(hipMalloc(&d_ohash.p, sizeof(int32_t) * L));
  SP_TRY(hipMalloc(&d_flag.p, sizeof(int32_t)));
  SP_TRY(hipMemcpy(d_codes.p, codes, (size_t)n * L, hipMemcpyHostToDevice));
  const unsigned blocks = (unsigned)((L + 255) / 256);
  // temporary storage of the rocPRIM calls (queried once)
  size_t tmp_bytes = 0, need = 0;
  SP_TRY(rocprim::radix_sort_pairs(nullptr, need, d_key.as<uint64_t>(), d_key2.as<uint64_t>(),
                                   d_site.as<uint32_t>(), d_site2.as<uint32_t>(), (size_t)L));
  tmp_bytes = need;
  SP_TRY(rocprim::inclusive_scan(nullptr, need, d_head.as<uint32_t>(), d_scan.as<uint32_t>(),
                                 (size_t)L, rocprim::plus<uint32_t>()));
  tmp_bytes = std::max(tmp_bytes, need);
  SP_TRY(rocprim::radix_sort_pairs(nullptr, need, d_leader.as<uint32_t>(),
                                   d_leader2.as<uint32_t>(), d_gid.as<uint32_t>(),
                                   d_gid2.as<uint32_t>(), (size_t)L));
  tmp_bytes = std::max(tmp_bytes, need);
  SP_TRY(hipMalloc(&d_tmp.p, tmp_bytes));

  uint32_t P = 0;
  hipEvent_t ev0, ev1;
  SP_TRY(hipEventCreate(&ev0));
  SP_TRY(hipEventCreate(&ev1));
  bool grouped = false;
  for (uint64_t seed = 0; seed < 4 && !grouped; seed++) {
    SP_TRY(hipMemset(d_flag.p, 0, sizeof(int32_t)));
    SP_TRY(hipEventRecord(ev0, nullptr));
    hipLaunchKernelGGL(hash_columns_kernel, dim3(blocks), dim3(256), 0, nullptr, n, L,
                       seed * 0x9e3779b97f4a7c15ull, d_codes.as<int8_t>(), d_ref.as<int32_t>(),
                       d_key.as<uint64_t>(), d_site.as<uint32_t>());
    SP_TRY(hipEventRecord(ev1, nullptr));
    // stable sort by the 64-bit hash, run heads, verification
    need = tmp_bytes;
    SP_TRY(rocprim::radix_sort_pairs(d_tmp.p, need, d_key.as<uint64_t>(), d_key2.as<uint64_t>(),
                                     d_site.as<uint32_t>(), d_site2.as<uint32_t>(), (size_t)L));
    hipLaunchKernelGGL(mark_heads_kernel, dim3(blocks), dim3(256), 0, nullptr, n, L,
                       d_codes.as<int8_t>(), d_key2.as<uint64_t>(), d_site2.as<uint32_t>(),
                       d_head.as<uint32_t>(), d_flag.as<int32_t>());
    need = tmp_bytes;
    SP_TRY(rocprim::inclusive_scan(d_tmp.p, need, d_head.as<uint32_t>(), d_scan.as<uint32_t>(),
                                   (size_t)L, rocprim::plus<uint32_t>()));
    int32_t collision = 0;
    SP_TRY(hipMemcpy(&P, d_scan.as<uint32_t>() + (L - 1), sizeof P, hipMemcpyDeviceToHost));
    SP_TRY(hipMemcpy(&collision, d_flag.p, sizeof collision, hipMemcpyDeviceToHost));
    grouped = collision == 0;  // two different columns with one 64-bit hash: other seed
  }
  float hash_ms = 0;
  SP_TRY(hipEventElapsedTime(&hash_ms, ev0, ev1));
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  if (out_hash_kernel_ms) *out_hash_kernel_ms = hash_ms;
  if (!grouped) return fail("site patterns: the grouping hash collided under every seed");

  hipLaunchKernelGGL(group_leaders_kernel, dim3(blocks), dim3(256), 0, nullptr, L,
                     d_head.as<uint32_t>(), d_scan.as<uint32_t>(), d_site2.as<uint32_t>(),
                     d_leader.as<uint32_t>(), d_start.as<uint32_t>());
  const unsigned pblocks = (P + 255) / 256;
  hipLaunchKernelGGL(iota_kernel, dim3(pblocks), dim3(256), 0, nullptr, P, d_gid.as<uint32_t>());
  need = tmp_bytes;
  SP_TRY(rocprim::radix_sort_pairs(d_tmp.p, need, d_leader.as<uint32_t>(),
                                   d_leader2.as<uint32_t>(), d_gid.as<uint32_t>(),
                                   d_gid2.as<uint32_t>(), (size_t)P));
  hipLaunchKernelGGL(gather_groups_kernel, dim3(pblocks), dim3(256), 0, nullptr, P, L,
                     d_gid2.as<uint32_t>(), d_leader.as<uint32_t>(), d_start.as<uint32_t>(),
                     d_ref.as<int32_t>(), d_osite.as<uint32_t>(), d_ocount.as<uint32_t>(),
                     d_ohash.as<int32_t>());
  std::vector<uint32_t> first(P), count(P);
  std::vector<int32_t> hashes(P);
  SP_TRY(hipMemcpy(first.data(), d_osite.p, sizeof(uint32_t) * P, hipMemcpyDeviceToHost));
  SP_TRY(hipMemcpy(count.data(), d_ocount.p, sizeof(uint32_t) * P, hipMemcpyDeviceToHost));
  SP_TRY(hipMemcpy(hashes.data(), d_ohash.p, sizeof(int32_t) * P, hipMemcpyDeviceToHost));
  SP_TRY(hipGetLastError());

  const std::vector<int> order = replay_unordered_map_order(hashes);
  *out_pattern_count = (int32_t)P;
  std::vector<uint32_t> final_site(P);
  std::vector<double> weights_host(P);
  for (uint32_t j = 0; j < P; j++) {
    final_site[j] = first[order[j]];
    weights_host[j] = (double)count[order[j]];
  }
  if (!keep) std::copy(weights_host.begin(), weights_host.end(), out_weights);
  // the pattern matrix is gathered on the device and comes back in one copy
  DevMem d_pat;
  SP_TRY(hipMalloc(&d_pat.p, sizeof(int32_t) * (size_t)n * P));
  SP_TRY(hipMemcpy(d_osite.p, final_site.data(), sizeof(uint32_t) * P, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(gather_patterns_kernel, dim3(pblocks, n), dim3(256), 0, nullptr, n, L, P,
                     d_codes.as<int8_t>(), d_osite.as<uint32_t>(), d_pat.as<int32_t>());
  if (keep) {
    DevMem d_w;
    SP_TRY(hipMalloc(&d_w.p, sizeof(double) * P));
    SP_TRY(hipMemcpy(d_w.p, weights_host.data(), sizeof(double) * P, hipMemcpyHostToDevice));
    SP_TRY(hipDeviceSynchronize());
    keep->patterns = d_pat.as<int32_t>();
    keep->weights = d_w.as<double>();
    d_pat.p = d_w.p = nullptr;  // (ownership passes to the caller: mi_device_free)
    return 0;
  }
  SP_TRY(hipMemcpy(out_patterns, d_pat.p, sizeof(int32_t) * (size_t)n * P, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int32_t mi_site_pattern_compress(int32_t device, int32_t taxon_count, int64_t site_count,
                                            const int8_t* codes, int32_t* out_pattern_count,
                                            int32_t* out_patterns, double* out_weights,
                                            double* out_hash_kernel_ms) {
  return compress_impl(device, taxon_count, site_count, codes, out_pattern_count, out_patterns,
                       out_weights, out_hash_kernel_ms, nullptr);
}

extern "C" int32_t mi_site_pattern_compress_device(int32_t device, int32_t taxon_count,
                                                   int64_t site_count, const int8_t* codes,
                                                   int32_t* out_pattern_count,
                                                   int32_t** out_device_patterns,
                                                   double** out_device_weights,
                                                   double* out_hash_kernel_ms) {
  if (!out_device_patterns || !out_device_weights) return fail("null argument");
  KeepOnDevice keep;
  if (compress_impl(device, taxon_count, site_count, codes, out_pattern_count, nullptr, nullptr,
                    out_hash_kernel_ms, &keep))
    return 1;
  *out_device_patterns = keep.patterns;
  *out_device_weights = keep.weights;
  return 0;
}

extern "C" void mi_device_free(void* device_pointer) {
  if (device_pointer) (void)hipFree(device_pointer);
}
