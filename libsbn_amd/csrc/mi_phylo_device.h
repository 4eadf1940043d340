// Device-side data layout shared by the kernels and the host engine.
// Everything here lives in HBM for the duration of one engine call; see
// DESIGN.md "Data layout in HBM".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace miphylo {

constexpr int kStates = 4;       // the reference is DNA-only (substitution_model.cpp:6-15)
constexpr int kMaxCategories = 64;  // (the reference parses any weibull+K, site_model.cpp:15-24)
constexpr int kTile = 64;        // site patterns per wavefront
#ifndef MI_LLR
#define MI_LLR 3
#endif
constexpr int kLlR = MI_LLR;     // matrix-core gradient kernel: registers (16 columns each) per node
constexpr int kFdModels = 17;    // base model + 2*(3 frequency + 5 rate) perturbed models

// One substitution+site model instance (what FatBeagle::SetParameters pushes to
// BEAGLE per tree: fat_beagle.cpp:273-300).
// (128-byte aligned, hence a multiple of 128 bytes: no cache line holds two instances -- see
// macro_stride)
struct alignas(128) DevModel {
  double pi[kStates];
  double Q[kStates * kStates];     // row-major
  double V[kStates * kStates];     // eigenvectors
  double Vinv[kStates * kStates];  // inverse eigenvectors
  double lambda[kStates];
  double cat_rate[kMaxCategories];
  double cat_weight[kMaxCategories];
  double cat_drate[kMaxCategories];  // d rate_k / d shape
};

// One entry of a tree's evaluation schedule: an internal node with its two
// children (node ids as in node.cpp:341-357) and the LDS slots the on-chip
// log-likelihood kernel keeps the partial-likelihood vectors in.
struct SchedEntry {
  int32_t node;
  int32_t child0;
  int32_t child1;
  int32_t slots;  // dst | slot(child0) << 8 | slot(child1) << 16
};

// One entry of the on-chip gradient kernels' schedule: a node whose pre-order
// vector is available (the root, or a "stored" node with an LDS slot) together
// with its two children and, for children whose vectors are NOT stored, their
// children.  kind: 0 = tip, 1 = stored internal node (has a slot), 2 = unstored
// internal node (both of its children are tips or stored nodes).
// Two 32-byte halves: the first is what the matrix-core kernel needs a macro AHEAD
// (node ids for the matrix / tip fetches, `shape` for its scalar control flow), the
// second (LDS slots) only during the macro itself.
struct MacroEntry {
  int32_t shape;     // kind[0] | kind[1] << 2 | root << 4 | (tip flags of child0, child1,
                     // grand0..3) << 8   (unused grand entries are node 0, flagged as tips)
  int32_t child[2];  // ids
  int32_t grand[4];  // children of child 0 (a0, b0) and of child 1 (a1, b1) when unstored
  int32_t node;      // id of the node whose q is known
  int32_t qslot;     // its LDS slot, -1 for the root (q = frequencies)
  int32_t cslot[2];  // slot of a stored child (else 0)
  int32_t gslot[4];  // slots of the grandchildren (0 for tips)
  int32_t pad;       // arena variant (macro_slots_kernel): index of the node's post-order vector in the arena
};
constexpr int kMacroPositions = 6;  // edges a macro can own: child0, child1, grand0..3

__host__ __device__ inline int macro_shape(int kind0, int kind1, bool root, const int32_t* child,
                                            const int32_t* grand, int n) {
  int s = kind0 | (kind1 << 2) | (root ? 16 : 0);
  for (int j = 0; j < 2; j++) s |= (child[j] < n ? 1 : 0) << (8 + j);
  for (int j = 0; j < 4; j++) s |= (grand[j] < n ? 1 : 0) << (10 + j);
  return s;
}

inline __host__ __device__ int max_macros(int n) { return (n - 2) / 2 + 1; }
// Stride of a tree's macro list in HBM: whole 128-byte lines per tree (an even number of the
// 64-byte entries), so that no cache line holds macros of two trees -- the one-launch small
// call hands each tree's list from its set-up wave to its walk waves INSIDE a kernel, and a
// line first touched after the hand-off cannot be stale (kernels_walk3.hip).
inline __host__ __device__ int macro_stride(int n) { return (max_macros(n) + 1) & ~1; }
inline __host__ __device__ int max_stored(int n) { return (n - 2) / 2 > 1 ? (n - 2) / 2 : 1; }

enum StatusCode : int32_t {
  kOk = 0,
  kBadParentIds = 1,
  kNotBifurcating = 2,
  kNotTrifurcatingRoot = 3,
  kGtrFrequencies = 4,
  kGtrRates = 5,
  kBadRateCount = 6,
  kTooManySlots = 7,
  kFusedTimeout = 8,
};

}  // namespace miphylo
