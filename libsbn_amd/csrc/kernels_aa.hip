// 20-state (amino-acid) path: Felsenstein pruning, the pre-order sweep and the edge
// derivatives for alignments whose partial-likelihood vectors (PLVs) do not fit on chip
// (BASELINE.json configs[4]: 512 taxa x 50 000 patterns x 4 categories, 32 MB per PLV).
//
// Restates, for s = 20, what the reference asks BEAGLE to do in
//   FatBeagle::LogLikelihoodInternals   src/fat_beagle.cpp:50-70
//   FatBeagle::BranchGradientInternals  src/fat_beagle.cpp:119-175
// (update transition matrices, post-order partials with rescaling, root log-likelihood,
// pre-order partials, edge derivatives).  The reference itself is DNA-only
// (src/substitution_model.cpp:6-15); parity is against the s-generic CPU oracle.
//
// Mapping (gfx950, wave64).  One wave = one (evaluation, rate category, block of
// kAa{Post,Pre}Tiles x 16 site patterns); it walks the whole tree for its block and never
// talks to another wave.  A 16-pattern tile of one category is five registers:
//     reg t, lane l  =  L[state 4t + (l >> 4)][pattern l & 15]
// which is at once
//   * the B operand of k-step t of  v_mfma_f64_16x16x4_f64  (B[k][j] at lane 16k + j) and of
//     v_mfma_f64_4x4x4_4b_f64 (B_b[k][j'] at lane 16k + 4b + j'),
//   * the C/D layout of both (16x16x4: row (l>>4) + 4r, col l&15; 4x4x4: row l>>4), so the
//     result of P x L -- rows 0..15 from five 16x16x4 steps, rows 16..19 from five 4x4x4
//     steps: 20 = 16 + 4, no padding -- is again a tile, with no cross-lane movement,
//   * 512 contiguous bytes in HBM: the arena layout is
//     [evaluation][node][category][tile][state][16 patterns] (pattern-major inside a tile),
//     so every load/store instruction of a wave moves one contiguous 512-byte line group.
// Measured (tools/fp64_peak_probe.hip, profiles/r02_fp64_peak_probe.txt): this 16+4 mix
// sustains 69-71 TFLOP/s device-wide at >= 2 waves per SIMD, the same as 4x4x4 alone, with
// 2.5x fewer instructions; 16x16x4 alone (rows padded to 32) reaches 34-49.
// Transition matrices are pre-packed by aa_transition_kernel as A operands (kAaPack: five
// registers of 64 lanes + five blocks of 16 values per matrix), tips are compact states: a tip child's product is a gather of a COLUMN of P.
// Rescaling is by exact powers of two per (pattern, category) column at every internal
// node (the column sum's exponent, obtained in all four lanes of the column by one more
// 4x4x4 product with a ones matrix).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "mi_phylo_device_utils.h"
#include "mi_phylo_kernels.h"

namespace miphylo {

namespace {
using namespace dev;

typedef double double4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

// Which matrix entry a pack's element `idx` holds (kAaPack, mi_phylo_kernels.h): elements
// 0..319 are register t = idx >> 6 of lane l = idx & 63 for the 16x16x4 steps -- A[i = l & 15]
// [k = l >> 4] = entry (row l & 15, column 4 t + (l >> 4)); elements 320..399 are the 4x4x4
// steps' 16 values per step t -- value (g, i) at 320 + 16 t + 4 g + i = entry (row 16 + i,
// column 4 t + g), which lane l of that instruction reads with g = l >> 4, i = l & 3.
struct AaPackEntry {
  int row, col;
  bool used;
};
__device__ __forceinline__ AaPackEntry aa_pack_entry(int idx) {
  if (idx < kAaPackRows16) {
    const int t = idx >> 6, l = idx & 63;
    return {l & 15, 4 * t + (l >> 4), true};
  }
  const int e = idx - kAaPackRows16, t = e >> 4, g = (e >> 2) & 3, i = e & 3;
  return {16 + i, 4 * t + g, e < 80};
}
// the ten A-operand registers of this lane from a pack (global memory or LDS)
__device__ __forceinline__ void pack_operands(const double* __restrict__ pack, int lane, double (&A)[10]) {
  const int tail = kAaPackRows16 + 4 * (lane >> 4) + (lane & 3);
#pragma unroll
  for (int r = 0; r < 5; r++) A[r] = pack[r * 64 + lane];
#pragma unroll
  for (int t = 0; t < 5; t++) A[5 + t] = pack[tail + 16 * t];
}

__device__ __forceinline__ double wave_sum_aa(double v) {  // every lane gets the sum
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ------------------------------------------------------------------------
// Model set-up: Q from (exchangeabilities, frequencies) by the reference's GTR recipe
// (substitution_model.cpp:39-80) and its symmetric eigendecomposition (cyclic Jacobi,
// one wave; once per engine).
// ------------------------------------------------------------------------
__global__ __launch_bounds__(64) void aa_model_setup_kernel(const double* exch,
                                                             const double* freqs, AaModel* m,
                                                             int32_t* status) {
  __shared__ double A[kAa * kAa], U[kAa * kAa], Q[kAa * kAa], pi[kAa], sq[kAa], ev[kAa];
  __shared__ double rot[2];
  __shared__ int done;
  const int lane = threadIdx.x;
  if (lane < kAa) {
    pi[lane] = freqs[lane];
    sq[lane] = sqrt(freqs[lane]);
  }
  __syncthreads();
  if (lane == 0) {
    double fsum = 0;
    for (int i = 0; i < kAa; i++) fsum += pi[i];
    if (fabs(fsum - 1.) >= 0.001) set_status(status, kGtrFrequencies, 0);
    int ri = 0;
    for (int i = 0; i < kAa; i++)
      for (int j = i + 1; j < kAa; j++) {
        const double r = exch[ri++];
        Q[i * kAa + j] = r * pi[j];
        Q[j * kAa + i] = r * pi[i];
      }
    double total = 0;
    for (int i = 0; i < kAa; i++) {
      double row = 0;
      for (int j = 0; j < kAa; j++)
        if (i != j) row += Q[i * kAa + j];
      Q[i * kAa + i] = -row;
      total += row * pi[i];
    }
    for (int i = 0; i < kAa * kAa; i++) Q[i] /= total;
  }
  __syncthreads();
  for (int idx = lane; idx < kAa * kAa; idx += 64) {
    const int i = idx / kAa, j = idx % kAa;
    // S = Pi^1/2 Q Pi^-1/2; the lower triangle is authoritative
    const int a = i >= j ? i : j, b = i >= j ? j : i;
    A[idx] = sq[a] * Q[a * kAa + b] * (1.0 / sq[b]);
    U[idx] = i == j ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int sweep = 0; sweep < 100; sweep++) {
    if (lane == 0) {
      double off = 0, diag = 0;
      for (int i = 0; i < kAa; i++)
        for (int j = 0; j < kAa; j++) {
          const double x = A[i * kAa + j];
          if (i != j) off += x * x;
          else diag += x * x;
        }
      done = (off <= 1e-45 * diag || off == 0.) ? 1 : 0;
    }
    __syncthreads();
    if (done) break;
    for (int p = 0; p < kAa - 1; p++)
      for (int q = p + 1; q < kAa; q++) {
        if (lane == 0) {
          const double apq = A[p * kAa + q];
          double c = 1.0, sn = 0.0;
          if (apq != 0.) {
            const double theta = (A[q * kAa + q] - A[p * kAa + p]) / (2. * apq);
            const double t = (theta >= 0 ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
            c = 1. / sqrt(t * t + 1.);
            sn = t * c;
          }
          rot[0] = c;
          rot[1] = sn;
        }
        __syncthreads();
        const double c = rot[0], sn = rot[1];
        if (lane < kAa) {
          const int k = lane;
          const double akp = A[k * kAa + p], akq = A[k * kAa + q];
          A[k * kAa + p] = c * akp - sn * akq;
          A[k * kAa + q] = sn * akp + c * akq;
        }
        __syncthreads();
        if (lane < kAa) {
          const int k = lane;
          const double apk = A[p * kAa + k], aqk = A[q * kAa + k];
          A[p * kAa + k] = c * apk - sn * aqk;
          A[q * kAa + k] = sn * apk + c * aqk;
          const double ukp = U[k * kAa + p], ukq = U[k * kAa + q];
          U[k * kAa + p] = c * ukp - sn * ukq;
          U[k * kAa + q] = sn * ukp + c * ukq;
        }
        __syncthreads();
      }
  }
  if (lane == 0) {
    for (int i = 0; i < kAa; i++) ev[i] = A[i * kAa + i];
    for (int i = 0; i < kAa - 1; i++) {  // ascending, permuting the columns of U
      int mi = i;
      for (int j = i + 1; j < kAa; j++)
        if (ev[j] < ev[mi]) mi = j;
      if (mi != i) {
        const double t = ev[i]; ev[i] = ev[mi]; ev[mi] = t;
        for (int k = 0; k < kAa; k++) {
          const double u = U[k * kAa + i]; U[k * kAa + i] = U[k * kAa + mi]; U[k * kAa + mi] = u;
        }
      }
    }
  }
  __syncthreads();
  if (lane < kAa) {
    m->pi[lane] = pi[lane];
    m->lambda[lane] = ev[lane];
  }
  for (int idx = lane; idx < kAa * kAa; idx += 64) {
    const int i = idx / kAa, j = idx % kAa;
    m->Q[idx] = Q[idx];
    m->V[idx] = (1.0 / sq[i]) * U[idx];
    m->Vinv[idx] = U[j * kAa + i] * sq[j];
  }
  for (int idx = lane; idx < kAaPack; idx += 64) {  // Q in the A-operand layout (below)
    const AaPackEntry pe = aa_pack_entry(idx);
    m->Qpack[idx] = pe.used ? Q[pe.row * kAa + pe.col] : 0.0;
  }
}

// ------------------------------------------------------------------------
// The same, wave-parallel (round 6): the kernel above applies its 190
// rotations per sweep one after the other, the rotation's cosine and sine computed by ONE lane
// between two barriers, and its stopping bound (off <= 1e-45 diag on squared norms) is
// unreachable in double precision -- it always runs its 100 sweeps: 10.1 ms for ONE 20 x 20
// eigensystem at every 20-state engine creation.  Here
//   * a sweep is 19 steps of TEN disjoint rotations (round-robin pairing: player 19 stays, the
//     other 19 rotate -- every pair (p, q) exactly once per sweep); ten lanes compute the
//     cosines / sines, then all 64 lanes apply the ten column rotations to A and U (200 + 200
//     element pairs), then the ten row rotations to A: disjoint rotations commute, so the step
//     is J^T A J with J their product;
//   * the off-diagonal / diagonal norms are summed by the wave, and the loop stops at the
//     rounding level (off <= 1e-30 diag) or when a sweep no longer reduces the off-diagonal
//     norm (at most 30 sweeps; 7-9 for WAG-like matrices);
//   * Q is built by all lanes (each entry and each row sum by the same expression in the same
//     order as above: bit-identical Q), the eigenvalues are ranked in parallel.
// Another rotation order: eigenvalues and P(t) agree with the kernel above to ~1e-15, not
// bit for bit (eigenvectors may come out with the other sign; V and V^-1 change together).
// MI_PHYLO_AA_JACOBI=seq selects the kernel above (tests/test_aa_gpu.py compares the two).
// ------------------------------------------------------------------------
__global__ __launch_bounds__(64) void aa_model_setup_wave_kernel(const double* exch,
                                                                  const double* freqs, AaModel* m,
                                                                  int32_t* status) {
  __shared__ double A[kAa * kAa], U[kAa * kAa], U2[kAa * kAa], Q[kAa * kAa], pi[kAa], sq[kAa], ev[kAa], rowsum[kAa];
  __shared__ double rc[10], rs[10];
  __shared__ int rank_of[kAa];
  const int lane = threadIdx.x;
  if (lane < kAa) {
    pi[lane] = freqs[lane];
    sq[lane] = sqrt(freqs[lane]);
  }
  __syncthreads();
  if (lane == 0) {
    double fsum = 0;
    for (int i = 0; i < kAa; i++) fsum += pi[i];
    if (fabs(fsum - 1.) >= 0.001) set_status(status, kGtrFrequencies, 0);
  }
  for (int idx = lane; idx < kAa * kAa; idx += 64) {
    const int i = idx / kAa, j = idx % kAa;
    if (i != j) {
      const int a = i < j ? i : j, b = i < j ? j : i;
      const double r = exch[a * (2 * kAa - a - 1) / 2 + (b - a - 1)];
      Q[idx] = r * pi[j];
    }
  }
  __syncthreads();
  if (lane < kAa) {
    double row = 0;
    for (int j = 0; j < kAa; j++)
      if (lane != j) row += Q[lane * kAa + j];
    Q[lane * kAa + lane] = -row;
    rowsum[lane] = row;
  }
  __syncthreads();
  double total = 0;
  for (int i = 0; i < kAa; i++) total += rowsum[i] * pi[i];
  for (int idx = lane; idx < kAa * kAa; idx += 64) Q[idx] /= total;
  __syncthreads();
  for (int idx = lane; idx < kAa * kAa; idx += 64) {
    const int i = idx / kAa, j = idx % kAa;
    // S = Pi^1/2 Q Pi^-1/2; the lower triangle is authoritative
    const int a = i >= j ? i : j, b = i >= j ? j : i;
    A[idx] = sq[a] * Q[a * kAa + b] * (1.0 / sq[b]);
    U[idx] = i == j ? 1.0 : 0.0;
  }
  __syncthreads();
  double prev_off = 1e300;
  for (int sweep = 0; sweep < 30; sweep++) {
    double off = 0, diag = 0;
    for (int idx = lane; idx < kAa * kAa; idx += 64) {
      const double x = A[idx];
      if (idx / kAa != idx % kAa) off += x * x;
      else diag += x * x;
    }
    off = wave_sum_aa(off);
    diag = wave_sum_aa(diag);
    if (off <= 1e-30 * diag || off == 0. || off >= prev_off) break;  // (wave-uniform: every lane holds the sums)
    prev_off = off;
    for (int step = 0; step < kAa - 1; step++) {
      // this lane's pair of the step (lanes 0..9 compute it; every lane needs one for the updates)
      auto pair_of = [&](int i, int& p, int& q) {
        const int x = i == 0 ? kAa - 1 : (step + i) % (kAa - 1);
        const int y = i == 0 ? step : (step - i + (kAa - 1)) % (kAa - 1);
        p = x < y ? x : y;
        q = x < y ? y : x;
      };
      if (lane < 10) {
        int p, q;
        pair_of(lane, p, q);
        const double apq = A[p * kAa + q];
        double c = 1.0, sn = 0.0;
        if (apq != 0.) {
          const double theta = (A[q * kAa + q] - A[p * kAa + p]) / (2. * apq);
          const double t = (theta >= 0 ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
          c = 1. / sqrt(t * t + 1.);
          sn = t * c;
        }
        rc[lane] = c;
        rs[lane] = sn;
      }
      __syncthreads();
      for (int idx = lane; idx < 10 * kAa; idx += 64) {  // columns p, q of A and U, row k
        const int i = idx / kAa, k = idx - i * kAa;
        int p, q;
        pair_of(i, p, q);
        const double c = rc[i], sn = rs[i];
        const double akp = A[k * kAa + p], akq = A[k * kAa + q];
        A[k * kAa + p] = c * akp - sn * akq;
        A[k * kAa + q] = sn * akp + c * akq;
        const double ukp = U[k * kAa + p], ukq = U[k * kAa + q];
        U[k * kAa + p] = c * ukp - sn * ukq;
        U[k * kAa + q] = sn * ukp + c * ukq;
      }
      __syncthreads();
      for (int idx = lane; idx < 10 * kAa; idx += 64) {  // rows p, q of A, column k
        const int i = idx / kAa, k = idx - i * kAa;
        int p, q;
        pair_of(i, p, q);
        const double c = rc[i], sn = rs[i];
        const double apk = A[p * kAa + k], aqk = A[q * kAa + k];
        A[p * kAa + k] = c * apk - sn * aqk;
        A[q * kAa + k] = sn * apk + c * aqk;
      }
      __syncthreads();
    }
  }
  // ascending eigenvalues: rank = how many are smaller (ties by index); columns of U permuted
  if (lane < kAa) {
    const double x = A[lane * kAa + lane];
    int r = 0;
    for (int j = 0; j < kAa; j++) {
      const double y = A[j * kAa + j];
      r += (y < x || (y == x && j < lane)) ? 1 : 0;
    }
    rank_of[lane] = r;
    ev[r] = x;
  }
  __syncthreads();
  for (int idx = lane; idx < kAa * kAa; idx += 64) {
    const int k = idx / kAa, j = idx % kAa;
    U2[k * kAa + rank_of[j]] = U[idx];
  }
  __syncthreads();
  if (lane < kAa) {
    m->pi[lane] = pi[lane];
    m->lambda[lane] = ev[lane];
  }
  for (int idx = lane; idx < kAa * kAa; idx += 64) {
    const int i = idx / kAa, j = idx % kAa;
    m->Q[idx] = Q[idx];
    m->V[idx] = (1.0 / sq[i]) * U2[idx];
    m->Vinv[idx] = U2[j * kAa + i] * sq[j];
  }
  for (int idx = lane; idx < kAaPack; idx += 64) {  // Q in the A-operand layout (below)
    const AaPackEntry pe = aa_pack_entry(idx);
    m->Qpack[idx] = pe.used ? Q[pe.row * kAa + pe.col] : 0.0;
  }
}

// ------------------------------------------------------------------------
// Transition matrices (beagleUpdateTransitionMatrices, fat_beagle.cpp:304-314), one
// workgroup per (edge, category, evaluation): P = I + V expm1(L r t) V^-1 (the stable form
// of V exp(L r t) V^-1, see kernels_setup.hip), negative entries clamped to 0 as BEAGLE
// does; for a gradient call also P^T (internal edges) and the columns of P Q (tip edges; an
// internal edge's derivative is taken as Q (P L) with the model's one Q operand: P and Q
// commute, so no P Q per edge).  Internal edges are written as matrix-core
// A operands:
//   registers 0..4 (16x16x4, k-step t):  lane l holds M[l & 15][4t + (l >> 4)]
//   registers 5..9 (4x4x4,   k-step t):  lane l holds M[16 + (l & 3)][4t + (l >> 4)]
// tip edges as per-state columns (state 20 = gap: P 1 = 1, (P Q) 1 = 0).
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aa_transition_kernel(AaTransitionArgs a) {
  // (round 6: the eigenvectors -- and, for a tip edge of a gradient call, Q -- are staged in LDS:
  // read from global memory inside the 20-term sums they were 8 000 dependent-latency loads per
  // workgroup, 36 us per call for one 512-taxon tree -- 6 % of a rank's 6 250-pattern share of
  // BASELINE configs[4].  Same products in the same order.)
  __shared__ double ex[kAa], Pm[kAa * kAa], PQm[kAa * kAa], Vs[kAa * kAa], Vis[kAa * kAa];
  const int edge = blockIdx.x, k = blockIdx.y, el = blockIdx.z;
  const int tree = a.eval_offset + el;
  const int tid = threadIdx.x;
  const AaModel& m = *a.model;
  const double rt = a.models[tree].cat_rate[k] * a.bl_eff[(size_t)tree * a.N + edge];
  if (tid < kAa) ex[tid] = expm1(m.lambda[tid] * rt);
  for (int idx = tid; idx < kAa * kAa; idx += 256) {
    Vs[idx] = m.V[idx];
    Vis[idx] = m.Vinv[idx];
  }
  __syncthreads();
  for (int idx = tid; idx < kAa * kAa; idx += 256) {
    const int i = idx / kAa, j = idx % kAa;
    double sum = i == j ? 1.0 : 0.0;
    for (int x = 0; x < kAa; x++) sum += Vs[i * kAa + x] * ex[x] * Vis[x * kAa + j];
    Pm[idx] = sum > 0 ? sum : 0;
  }
  __syncthreads();
  if (a.gradient && edge < a.n) {  // (tip edges only: columns of P Q; internal edges use Q (P L))
    for (int idx = tid; idx < kAa * kAa; idx += 256) Vs[idx] = m.Q[idx];  // (V is no longer needed)
    __syncthreads();
    for (int idx = tid; idx < kAa * kAa; idx += 256) {
      const int i = idx / kAa, j = idx % kAa;
      double sum = 0;
      for (int x = 0; x < kAa; x++) sum += Pm[i * kAa + x] * Vs[x * kAa + j];
      PQm[idx] = sum;
    }
    __syncthreads();
  }
  if (edge < a.n) {
    const size_t base = (((size_t)el * a.n + edge) * a.K + k) * kAaTipTable;
    for (int idx = tid; idx < kAaTipTable; idx += 256) {
      const int x = idx / kAa, i = idx % kAa;
      a.tipP[base + idx] = x < kAa ? Pm[i * kAa + x] : 1.0;
      if (a.gradient) a.tipPQ[base + idx] = x < kAa ? PQm[i * kAa + x] : 0.0;
    }
  } else {
    const size_t base = (((size_t)el * (a.n - 1) + (edge - a.n)) * a.K + k) * kAaPack;
    for (int idx = tid; idx < kAaPack; idx += 256) {
      const AaPackEntry pe = aa_pack_entry(idx);
      a.matP[base + idx] = pe.used ? Pm[pe.row * kAa + pe.col] : 0.0;
      if (a.gradient) a.matPT[base + idx] = pe.used ? Pm[pe.col * kAa + pe.row] : 0.0;
    }
  }
}

// ------------------------------------------------------------------------
// Walk helpers
// ------------------------------------------------------------------------
__device__ __forceinline__ int sgpr(int x) { return __builtin_amdgcn_readfirstlane(x); }
// a pointer that is the same in every lane, told to the compiler (scalar registers: loads and
// stores through it take the scalar-base + lane-offset addressing form, no 64-bit vector adds)
template <class T>
__device__ __forceinline__ T* sgpr_ptr(T* p) {
  const uint64_t x = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
  typedef T __attribute__((address_space(1))) * global_ptr;  // (keeps it a GLOBAL pointer: no flat loads)
  return (T*)(global_ptr)(((uint64_t)hi << 32) | lo);
}

// S = Mat x L for M tiles of 16 patterns, the matrix given as its ten A-operand registers:
// per tile five 16x16x4 steps (rows 0..15) and five 4x4x4 steps (rows 16..19)
template <int M>
__device__ __forceinline__ void mat_apply(const double (&A)[10], const double (&L)[M][5],
                                          double (&S)[M][5]) {
#pragma unroll
  for (int u = 0; u < M; u++) {
    double4v c = {0, 0, 0, 0};
    double d = 0;
#pragma unroll
    for (int t = 0; t < 5; t++) {
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t], L[u][t], c, 0, 0, 0);
#ifdef AA_MFMA_NOPS  // (experiment, DESIGN 4.6 finding 2: the wave stays silent while its MFMA runs)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop %0" ::"n"(AA_MFMA_NOPS));
      __builtin_amdgcn_sched_barrier(0);
#endif
      d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[5 + t], L[u][t], d, 0, 0, 0);
#ifdef AA_MFMA_NOPS
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 2");
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    S[u][0] = c.x; S[u][1] = c.y; S[u][2] = c.z; S[u][3] = c.w;
    S[u][4] = d;
  }
}

__device__ __forceinline__ void load_pack(const double* __restrict__ pack, int lane,
                                          double (&A)[10]) {
  pack_operands(pack, lane, A);
}

// compact states of a tip for this lane's pattern column of each tile (20 = gap / padding)
// (rows of the tip-state matrix are padded with gaps to whole tiles: no bounds test)
template <int M>
__device__ __forceinline__ void load_tip_states(const int8_t* __restrict__ tips, int p0, int lane,
                                                int (&x)[M]) {
#pragma unroll
  for (int u = 0; u < M; u++) x[u] = (tips + p0)[(unsigned)(u * 16 + (lane & 15))];
}
// product of a tip child: column `state` of the matrix (tables [21][20])
template <int M>
__device__ __forceinline__ void tip_gather(const double* __restrict__ table, const int (&x)[M],
                                           int lane, double (&S)[M][5]) {
#pragma unroll
  for (int u = 0; u < M; u++) {
    const double* col = table + x[u] * kAa + (lane >> 4);
#pragma unroll
    for (int t = 0; t < 5; t++) S[u][t] = col[4 * t];
  }
}

// A tile in memory (arena, LDS rings): 320 doubles = registers (0, 1) of the 64 lanes
// interleaved (lane l: doubles 2l, 2l + 1), registers (2, 3) likewise from double 128, register
// 4 from double 256 -- two 16-byte and one 8-byte access per lane and tile instead of five
// 8-byte ones (round 5: a vector-memory instruction of these walks costs 40-50 clocks of issue
// whatever its width; the stamps of aa_pre_wg_kernel).  Every access goes through the three
// functions below.
template <int M>
__device__ __forceinline__ void load_tiles(const double* __restrict__ src, int lane,
                                           double (&L)[M][5]) {
#pragma unroll
  for (int u = 0; u < M; u++) {
    const double* p = src + u * kAaTileDoubles;
    const double2v a = *reinterpret_cast<const double2v*>(p + 2 * lane);
    const double2v b = *reinterpret_cast<const double2v*>(p + 128 + 2 * lane);
    L[u][0] = a.x;
    L[u][1] = a.y;
    L[u][2] = b.x;
    L[u][3] = b.y;
    L[u][4] = p[256 + lane];
  }
}
// Sum of one double per lane over the wave without LDS round trips: a product with a ones
// matrix adds the four 16-lane rows (v_mfma_f64_4x4x4: D[i][j] = sum_k B[k][j], k = lane >>
// 4), four row rotations add the 16 lanes of a row.  Every lane ends with the total.  (The
// ds_bpermute butterfly this replaces is six dependent LDS-crossbar round trips: two such
// sums per pre-order visit were ~15 % of the visit's latency.)
template <int SHIFT>
__device__ __forceinline__ double aa_row_ror_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int slo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + SHIFT, 0xf, 0xf, true);
  const int shi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + SHIFT, 0xf, 0xf, true);
  return v + __hiloint2double(shi, slo);
}
__device__ __forceinline__ double wave_sum_mfma(double v) {
  v = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, v, 0.0, 0, 0, 0);
  v = aa_row_ror_add<8>(v);
  v = aa_row_ror_add<4>(v);
  v = aa_row_ror_add<2>(v);
  v = aa_row_ror_add<1>(v);
  return v;
}

template <int M>
__device__ __forceinline__ void store_tiles(double* __restrict__ dst, int lane,
                                            const double (&L)[M][5]) {
#pragma unroll
  for (int u = 0; u < M; u++) {
    double* p = dst + u * kAaTileDoubles;
    *reinterpret_cast<double2v*>(p + 2 * lane) = double2v{L[u][0], L[u][1]};
    *reinterpret_cast<double2v*>(p + 128 + 2 * lane) = double2v{L[u][2], L[u][3]};
    p[256 + lane] = L[u][4];
  }
}

// The same for a tile in an LDS ring, through pointers that SAY they are LDS pointers: a ring
// read and an arena read under the two arms of one branch are otherwise merged into ONE load
// through a generic pointer -- a flat instruction, which reaches LDS by way of the vector
// memory path (round 5: the pre-order kernel's ring reads were such loads).
typedef __attribute__((address_space(3))) double* lds_f64_ptr;
typedef __attribute__((address_space(3))) const double* lds_cf64_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;
template <int M>
__device__ __forceinline__ void load_tiles_lds(const double* src_generic, int lane, double (&L)[M][5]) {
  const lds_cf64_ptr src = (lds_cf64_ptr)(lds_void_ptr)const_cast<double*>(src_generic);
#pragma unroll
  for (int u = 0; u < M; u++) {
    const lds_cf64_ptr p = src + u * kAaTileDoubles;
    typedef __attribute__((address_space(3))) const double2v* lds_pair_ptr;
    const double2v a = *(lds_pair_ptr)(p + 2 * lane);
    const double2v b = *(lds_pair_ptr)(p + 128 + 2 * lane);
    L[u][0] = a.x;
    L[u][1] = a.y;
    L[u][2] = b.x;
    L[u][3] = b.y;
    L[u][4] = p[256 + lane];
  }
}
template <int M>
__device__ __forceinline__ void store_tiles_lds(double* dst_generic, int lane, const double (&L)[M][5]) {
  const lds_f64_ptr dst = (lds_f64_ptr)(lds_void_ptr)dst_generic;
#pragma unroll
  for (int u = 0; u < M; u++) {
    const lds_f64_ptr p = dst + u * kAaTileDoubles;
    typedef __attribute__((address_space(3))) double2v* lds_pair_ptr;
    *(lds_pair_ptr)(p + 2 * lane) = double2v{L[u][0], L[u][1]};
    *(lds_pair_ptr)(p + 128 + 2 * lane) = double2v{L[u][2], L[u][3]};
    p[256 + lane] = L[u][4];
  }
}

// Workgroups are dealt round-robin over the 8 XCDs by linear id.  The walk kernels give each
// XCD a CONTIGUOUS range of the (evaluation x category)-major work list, so that an XCD's
// L2 holds the transition matrices of one or two (evaluation, category) units instead of all.
struct AaUnit {
  int ec, blk;
  bool valid;
};
__device__ __forceinline__ AaUnit aa_unit(int blocks, int units) {
  const long W = (long)blocks * units;
  const long per = (W + 7) / 8;
  const long id = blockIdx.x;
  const long w = (id & 7) * per + (id >> 3);
  AaUnit r;
  r.valid = (id >> 3) < per && w < W;
  r.ec = (int)(w / blocks);
  r.blk = (int)(w - (long)r.ec * blocks);
  return r;
}

// The schedule of a tree in LDS, a window of kSchedWindow entries at a time (a wave looks
// two visits ahead); one wave per workgroup, so the window is private to the wave.
constexpr int kSchedWindow = 128;  // (2 KB: with the ring's 21 KB four log-likelihood workgroups fit a CU)
struct SchedWindow {
  const SchedEntry* sched;  // the tree's n-1 entries in HBM
  SchedEntry* lds;
  int count, base, lane;
  // forward walk: entries [base, base + window)
  __device__ __forceinline__ void fill_from(int first) {
    base = first;
    __syncthreads();
    for (int i = lane; i < kSchedWindow && base + i < count; i += 64) lds[i] = sched[base + i];
    __syncthreads();
  }
  // backward walk: entries (last - window, last]
  __device__ __forceinline__ void fill_upto(int last) {
    base = last - kSchedWindow + 1;
    if (base < 0) base = 0;
    __syncthreads();
    for (int i = lane; i < kSchedWindow && base + i < count; i += 64) lds[i] = sched[base + i];
    __syncthreads();
  }
  __device__ __forceinline__ SchedEntry at(int i) const { return lds[i - base]; }
};

// ------------------------------------------------------------------------
// Post-order (beagleUpdatePartials with rescaling + beagleCalculateRootLogLikelihoods'
// per-pattern part; fat_beagle.cpp:60-68,139-141).  The schedule is tree_setup's
// Sethi-Ullman post-order: the node visited just before a node is one of its children
// (unless both are tips), and that child's vector is taken from the registers the previous
// visit left it in.  GRAD: every internal vector is also kept in the arena, by node, for the
// pre-order kernel; otherwise only vectors that are not consumed from registers are
// written, into the schedule's slot (<= floor(log2 n) + 1 per evaluation).
//
// Stores are deferred: gfx9 counts loads and stores in ONE in-order counter (vmcnt), so a
// wave that waits for the loads of visit i+1 also waits for the acknowledgement of every
// store issued before them -- measured: the loads of the walk are hidden by the other waves,
// the stores were not (removing them took 4 ms off a 19 ms gradient, their bandwidth time).
// The result of visit i therefore stays in registers and is stored in visit i+1 AFTER that
// visit's products have consumed (waited for) their loads: the stores then have a whole
// visit to drain before the next wait.
// Latency: a visit is ~0.3 us of matrix-core work behind ~1.5 us of memory latency, hidden by
// occupancy (4-5 waves per SIMD), so the kernel is written for few registers and for ONE
// memory round trip per visit: the schedule sits in LDS, and the tip states that address
// the table columns of visit i+1 are requested during visit i.  (A register double buffer
// of the operands of visit i+1 was tried: hipcc turns the counted waits of such a pipeline
// into vmcnt(0) wherever loads sit in divergent-count branches, and the lost occupancy
// cost more than the prefetch gained: 4.5 -> 5.0 ms.)
// ------------------------------------------------------------------------
template <int M, bool GRAD>
__global__ __launch_bounds__(64) void aa_post_kernel(AaWalkArgs a) {
  __shared__ SchedEntry sched_lds[kSchedWindow];
  const int blocks = a.tiles / M;
  const AaUnit un = aa_unit(blocks, a.evals * a.K);
  if (!un.valid) return;
  const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
  const int el = un.ec / a.K, cat = un.ec - el * a.K, blk = un.blk;
  const int tree = a.eval_offset + el;
  const int n = a.n, K = a.K;
  const int p0 = blk * M * 16;
  const size_t tiles = a.tiles, tip_stride = tiles * 16;
  const int nodes = GRAD ? n - 1 : a.slots;
  double* arena = a.arena + (((size_t)el * nodes * K + cat) * tiles + (size_t)blk * M) * kAaTileDoubles;
  const size_t arena_stride = (size_t)K * tiles * kAaTileDoubles;  // per node / slot
  int32_t* exp_cum = a.exp_cum + ((size_t)el * nodes * K + cat) * tiles * 16 + p0;
  int32_t* exp_loc = GRAD ? a.exp_loc + ((size_t)el * (n - 1) * K + cat) * tiles * 16 + p0 : nullptr;
  const size_t exp_stride = (size_t)K * tiles * 16;
  const double* matP = a.matP + ((size_t)el * (n - 1) * K + cat) * kAaPack;
  const double* tipP = a.tipP + ((size_t)el * n * K + cat) * kAaTipTable;
  const int count = n - 1;
  SchedWindow win{a.sched + (size_t)tree * count, sched_lds, count, 0, lane};
  win.fill_from(0);

  // stage 0 of a visit: tip states of its tip children (they address the table columns)
  auto stage0 = [&](int c0, int c1, int (&x)[2][M]) {
    if (c0 < n) load_tip_states<M>(a.tip_states + (size_t)c0 * tip_stride, p0, lane, x[0]);
    if (c1 < n) load_tip_states<M>(a.tip_states + (size_t)c1 * tip_stride, p0, lane, x[1]);
  };
  int xc[2][M], xn[2][M];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int u = 0; u < M; u++) xc[c][u] = xn[c][u] = kAa;
  // the current schedule entry lives in scalar registers; the next one is read (once) from
  // the LDS window while this one computes
  int v, ch0, ch1, slots;
  {
    const SchedEntry e0 = win.at(0);
    v = sgpr(e0.node);
    ch0 = sgpr(e0.child0);
    ch1 = sgpr(e0.child1);
    slots = sgpr(e0.slots);
  }
  stage0(ch0, ch1, xc);

  double R[M][5];
  int E[M];
#pragma unroll
  for (int u = 0; u < M; u++) {
    E[u] = 0;
#pragma unroll
    for (int t = 0; t < 5; t++) R[u][t] = 0;
  }
  int prev = -1, prev_slots = 0;
  int eloc[M];
#pragma unroll
  for (int u = 0; u < M; u++) eloc[u] = 0;
  for (int i = 0; i < count; i++) {
    int nv = -1, next_c0 = -1, next_c1 = -1, nslots = 0;
    if (i + 1 < count) {
      if (i + 1 >= win.base + kSchedWindow) win.fill_from(i + 1);
      const SchedEntry s1 = win.at(i + 1);
      nv = sgpr(s1.node);
      next_c0 = sgpr(s1.child0);
      next_c1 = sgpr(s1.child1);
      nslots = sgpr(s1.slots);
      stage0(next_c0, next_c1, xn);
    }
    double S[2][M][5];
    int Ec[2][M];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int ch = c ? ch1 : ch0;
      if (ch < n) {
        tip_gather<M>(tipP + (size_t)ch * K * kAaTipTable, xc[c], lane, S[c]);
#pragma unroll
        for (int u = 0; u < M; u++) Ec[c][u] = 0;
      } else {
        double A[10], L[M][5];
        load_pack(matP + (size_t)(ch - n) * K * kAaPack, lane, A);
        if (ch == prev) {
#pragma unroll
          for (int u = 0; u < M; u++) {
            Ec[c][u] = E[u];
#pragma unroll
            for (int t = 0; t < 5; t++) L[u][t] = R[u][t];
          }
        } else {
          const int idx = GRAD ? ch - n : ((slots >> (8 + 8 * c)) & 0xff);
          load_tiles<M>(arena + idx * arena_stride, lane, L);
#pragma unroll
          for (int u = 0; u < M; u++) Ec[c][u] = exp_cum[idx * exp_stride + u * 16 + j];
        }
        mat_apply<M>(A, L, S[c]);
      }
    }
    // (the product first: it consumes -- waits for -- everything this visit loaded)
    double Tn[M][5];
#pragma unroll
    for (int u = 0; u < M; u++)
#pragma unroll
      for (int t = 0; t < 5; t++) Tn[u][t] = S[0][u][t] * S[1][u][t];
    // the deferred stores of the previous visit (its vector is still in R)
    if (prev >= 0) {
      if (GRAD) {
        // what the pre-order pass needs of a node is P L (its vector at the parent's end of
        // the edge): for the child this visit took from registers that product exists right
        // here, so IT is stored, and the pre-order pass skips the product; a node consumed
        // later is stored as L (its parent's visits, here and there, form P L themselves)
        double* dstp = arena + (size_t)(prev - n) * arena_stride;
        if (prev == ch0) store_tiles<M>(dstp, lane, S[0]);
        else if (prev == ch1) store_tiles<M>(dstp, lane, S[1]);
        else store_tiles<M>(dstp, lane, R);
        if (g == 0) {
#pragma unroll
          for (int u = 0; u < M; u++) {
            exp_loc[(size_t)(prev - n) * exp_stride + u * 16 + j] = eloc[u];
            exp_cum[(size_t)(prev - n) * exp_stride + u * 16 + j] = E[u];
          }
        }
      } else if (prev != ch0 && prev != ch1) {
        const int dst = prev_slots & 0xff;
        store_tiles<M>(arena + dst * arena_stride, lane, R);
        if (g == 0) {
#pragma unroll
          for (int u = 0; u < M; u++) exp_cum[dst * exp_stride + u * 16 + j] = E[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < M; u++) {
      // column sum over the 20 states: this lane's five rows with vector adds, then ONE product
      // with a ones matrix for the four row groups (only the exponent of the sum is used)
      const double part = ((Tn[u][0] + Tn[u][1]) + (Tn[u][2] + Tn[u][3])) + Tn[u][4];
      const double cs = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, part, 0.0, 0, 0, 0);
      // exact power-of-two rescaling by the exponent of the column sum (all four lanes of
      // a column hold the same sum)
      const int e = cs > 0.0 ? __builtin_amdgcn_frexp_exp(cs) - 1 : 0;
#pragma unroll
      for (int t = 0; t < 5; t++) R[u][t] = ldexp(Tn[u][t], -e);
      eloc[u] = e;
      E[u] = Ec[0][u] + Ec[1][u] + e;
    }
    prev_slots = slots;
    prev = v;
    v = nv;
    ch0 = next_c0;
    ch1 = next_c1;
    slots = nslots;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int u = 0; u < M; u++) xc[c][u] = xn[c][u];
  }
  if (GRAD && g == 0) {  // (the root's vector itself is not needed)
#pragma unroll
    for (int u = 0; u < M; u++) exp_loc[(size_t)(prev - n) * exp_stride + u * 16 + j] = eloc[u];
  }
  // root: sum_i pi_i L_root[i] per pattern (scaled) and its exponent
  double pi[5];
#pragma unroll
  for (int t = 0; t < 5; t++) pi[t] = a.model->pi[4 * t + g];
  const size_t rbase = ((size_t)el * K + cat) * tiles * 16 + p0;
#pragma unroll
  for (int u = 0; u < M; u++) {
    double s = 0;
#pragma unroll
    for (int t = 0; t < 5; t++) s += pi[t] * R[u][t];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0) {
      a.root_val[rbase + u * 16 + j] = s;
      a.root_exp[rbase + u * 16 + j] = E[u];
    }
  }
}

// ------------------------------------------------------------------------
// Post-order, workgroup form: four waves = four consecutive pattern blocks of ONE
// (evaluation, category) walk the tree in lock-step (one barrier per visit), and what they
// all need -- the children's matrices as A operands (5 KB each) or, for a tip child, its
// table of matrix columns (3.4 KB) -- is fetched from L2 ONCE per workgroup into LDS (the
// "LDS-staged 20x20 state tiles" of the north star) while the previous visit computes:
// global -> registers at the top of visit i, registers -> LDS at its end, barrier, visit i+1
// reads A operands (ds_read_b64, lane-linear) and tip columns (LDS gather) from there.
// Measured motivation: the per-visit time of the wave-per-block form grows with the number
// of waves per CU (4.7 k -> 9 k cycles from 2 to 8 waves) while neither HBM nor the matrix
// cores are busy -- the operand traffic L2 -> CU (packs + column gathers, about 25 KB per
// visit and wave) is the shared resource; this form divides it by four.
// ------------------------------------------------------------------------
// waves per workgroup: four (two or more workgroups per CU overlap each other's barriers;
// eight were measured: the same for log-likelihoods, 7 % slower for gradient calls)
constexpr int kPostWaves = 4, kPostThreads = 64 * kPostWaves;

// LDS-DMA (global_load_lds_dwordx4): one wave-instruction moves 1 KB global -> LDS with no
// register destination; the LDS destination is M0 (wave-uniform byte address) + 16 B x lane.
// Issued from inline assembly on purpose: hipcc then neither tracks it in its own vmcnt
// bookkeeping (which, with stores in flight, degrades every wait to vmcnt(0)) nor fences the
// barrier with vmcnt(0).  The kernels wait for it themselves: ONE `s_waitcnt vmcnt(0)` per
// visit, placed just before the visit's stores are issued -- everything older (the DMA from
// the top of the visit, the previous visit's stores) has had most of a visit to finish -- so
// the stores themselves stay in flight across the barrier.
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(3))) int8_t* lds_i8_ptr;
// All of them address memory as (wave-uniform base in scalar registers) + (lane offset in
// one vector register) + immediate: no 64-bit vector address arithmetic.
__device__ __forceinline__ void dma_1k(const double* src, uint32_t lane16, double* dst_lds) {
  // (M0 is a reserved register: hipcc only writes it immediately before an instruction that
  // reads it, never keeps a value in it, so setting it here clobbers nothing)
  const uint32_t m0 = (uint32_t)(uintptr_t)(lds_ptr)dst_lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(src), "s"(m0) : "memory");
}
// A run of COUNT (<= 4) consecutive 1 KB pieces by ONE wave: one scalar base, one M0 value,
// the pieces as immediate offsets -- the immediate moves the global source AND the LDS
// destination (tools/lds_dma_probe.hip).  Round 3: the staging of a visit dealt its pieces
// round-robin to the waves with a branch, a 64-bit scalar add and an M0 write per piece --
// most of the ~185 scalar instructions a pre-order visit executed.
template <int COUNT>
__device__ __forceinline__ void dma_run(const double* src, uint32_t lane16, double* dst_lds) {
  static_assert(COUNT >= 1 && COUNT <= 4, "immediate offsets reach 4095 bytes");
  const uint32_t m0 = (uint32_t)(uintptr_t)(lds_ptr)dst_lds;
  if (COUNT == 1)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(src), "s"(m0) : "memory");
  else if (COUNT == 2)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(lane16), "s"(src), "s"(m0) : "memory");
  else if (COUNT == 3)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(lane16), "s"(src), "s"(m0) : "memory");
  else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(lane16), "s"(src), "s"(m0) : "memory");
}
// a whole operand table of PIECES (4: tip column table, 5: matrix pack) by one wave
template <int PIECES>
__device__ __forceinline__ void dma_whole(const double* src, uint32_t lane16, double* dst_lds) {
  dma_run<4>(src, lane16, dst_lds);
  if (PIECES == 5) dma_run<1>(src + 4 * 128, lane16, dst_lds + 4 * 128);
}
// Piece k of a table (128 doubles) is issued by wave (first + k) % waves of the workgroup.
__device__ __forceinline__ void dma_table(const double* __restrict__ src, double* dst_lds,
                                          int pieces, int first, int wave, int waves, int lane) {
  const uint32_t lane16 = (uint32_t)lane * 16;
  for (int k = 0; k < pieces; k++)
    if ((first + k) % waves == wave) dma_1k(src + k * 128, lane16, dst_lds + k * 128);
}
// Stores the compiler does not see (same reason: a store it knows to be in flight turns its
// next wait for any load into vmcnt(0), i.e. into a wait for the store's acknowledgement).
// Same-wave program order to the same address is kept by the memory pipeline; the kernels'
// one vmcnt(0) per visit retires them a visit later.
template <int M>
__device__ __forceinline__ void store_tiles_async(double* dst, int lane, const double (&L)[M][5]) {
  const uint32_t lane8 = (uint32_t)lane * 8, lane16 = (uint32_t)lane * 16;
#pragma unroll
  for (int u = 0; u < M; u++) {
    const double* p = dst + u * kAaTileDoubles;
    const double2v a = {L[u][0], L[u][1]}, b = {L[u][2], L[u][3]};
    asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(lane16), "v"(a), "s"(p) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:1024" ::"v"(lane16), "v"(b), "s"(p) : "memory");
    asm volatile("global_store_dwordx2 %0, %1, %2 offset:2048" ::"v"(lane8), "v"(L[u][4]), "s"(p) : "memory");
  }
}
// (base: wave-uniform; index: this lane's element)
__device__ __forceinline__ void store_async(int32_t* base, int index, int v) {
  asm volatile("global_store_dword %0, %1, %2" ::"v"((uint32_t)index * 4), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void store_async(double* base, int index, double v) {
  asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"((uint32_t)index * 8), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void wait_all_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// workgroup barrier that orders LDS traffic only (no vmcnt: stores stay in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int kPreWaves = 4, kPreThreads = 64 * kPreWaves;
template <int M, bool GRAD>
__global__ __launch_bounds__(64 * kPostWaves, M == 4 ? 3 : 4) void aa_post_wg_kernel(AaWalkArgs a) {
  __shared__ SchedEntry sched_lds[kSchedWindow];
  __shared__ double ops_lds[2][2][kAaPack];  // [buffer][child][512]: pack (400), or tip table (420)
  __shared__ int8_t tips_lds[2][2][kPostWaves * M * 16];  // [buffer][child]: a tip child's states
  // Log-likelihood form (round 5): the TOP of the wave's stack of kept
  // vectors lives in LDS.  A post-order walk keeps vectors in stack order -- the schedule's slot
  // number of a kept vector IS its stack position (tree set-up hands out the lowest free slot,
  // and the two children of a visit are the two topmost entries) -- and most kept vectors are
  // consumed a few visits later: of the ~171 vectors a random 512-taxon tree keeps (the rest
  // pass from a visit to the next in registers), 90 % never have more than one other vector
  // pushed on top of them, 97 % never more than two (the stack is never deeper than 5 there).
  // Until round 5 every one of them went to the HBM arena and back: 42 + 42 GB of the
  // kernel's ~100 GB per eight 512 x 50 000 x 4 trees.  Now the slots [lo, lo + S) of the wave
  // are a ring in LDS (S = a.ring_slots entries; vector and exponents); a push that finds the ring full
  // spills the ring's OLDEST entry to its arena slot first (lo++), a pop below lo reads the
  // arena as before.  Same loads, products and stores in the same order: bit-identical.
  extern __shared__ __align__(16) double ring_lds[];  // [wave][S][M tiles x (320 doubles + 16 ints)]
  const int blocks = a.tiles / M;
  const int wgs = (blocks + kPostWaves - 1) / kPostWaves;  // workgroups per (evaluation, category)
  const AaUnit un = aa_unit(wgs, a.evals * a.K);
  if (!un.valid) return;
  const int tid = threadIdx.x, wave = sgpr(tid >> 6), lane = tid & 63, g = lane >> 4, j = lane & 15;
  const int el = un.ec / a.K, cat = un.ec - el * a.K;
  const int blk_raw = un.blk * kPostWaves + wave;
  const bool active = blk_raw < blocks;         // (a padding wave joins the staging and barriers only)
  const int blk = active ? blk_raw : blocks - 1;
  const int tree = a.eval_offset + el;
  const int n = a.n, K = a.K;
  const int p0 = sgpr(blk * M * 16);
  constexpr int kTipBytes = kPostWaves * M * 16;  // tip states of the workgroup's patterns
  const size_t p0_wg = (size_t)un.blk * kTipBytes;
  const size_t tiles = a.tiles, tip_stride = tiles * 16;
  const int nodes = GRAD ? n - 1 : a.slots;
  double* arena = sgpr_ptr(a.arena + (((size_t)el * nodes * K + cat) * tiles + (size_t)blk * M) * kAaTileDoubles);
  const size_t arena_stride = (size_t)K * tiles * kAaTileDoubles;
  int32_t* exp_cum = sgpr_ptr(a.exp_cum + ((size_t)el * nodes * K + cat) * tiles * 16 + p0);
  int32_t* exp_loc = GRAD ? sgpr_ptr(a.exp_loc + ((size_t)el * (n - 1) * K + cat) * tiles * 16 + p0) : nullptr;
  const size_t exp_stride = (size_t)K * tiles * 16;
  const double* matP = a.matP + ((size_t)el * (n - 1) * K + cat) * kAaPack;
  const double* tipP = a.tipP + ((size_t)el * n * K + cat) * kAaTipTable;
  const int count = n - 1;
  SchedWindow win{a.sched + (size_t)tree * count, sched_lds, count, 0, tid};
  // (the window is filled by the whole workgroup: 256 threads, 256 entries)
  auto fill = [&](int first) {
    win.base = first;
    __syncthreads();
    if (tid < kSchedWindow && first + tid < count) sched_lds[tid] = win.sched[first + tid];
    __syncthreads();
  };
  fill(0);

  // staging of one visit's shared operands into LDS buffer `buf` by LDS-DMA: a child's pack
  // (4 pieces of 1 KB) or, for a tip, its column table (3 360 B: 4 pieces, the tail of the
  // last one is the next table's head -- the arrays are padded by one piece)
  const int wave_s = sgpr(wave);
  auto stage = [&](int c0, int c1, int buf) {
    // two operand tables per visit, each split between two waves: wave 2c takes the first
    // pieces of child c's table, wave 2c + 1 the rest (runs with immediate offsets: dma_run)
    static_assert(kPostWaves == 4, "operand staging deals two children's tables to two waves each");
    const int c = wave_s >> 1, second = wave_s & 1;
    const int ch = c ? c1 : c0;
    const uint32_t lane16 = (uint32_t)lane * 16;
    if (ch < n) {
      const double* src = tipP + (size_t)ch * K * kAaTipTable + (second ? 2 * 128 : 0);
      dma_run<2>(src, lane16, ops_lds[buf][c] + (second ? 2 * 128 : 0));
      // the tip's states for the workgroup's patterns (kTipBytes of them): ONE 4-byte LDS-DMA by
      // the table's second wave instead of M byte loads by every wave (round 5: a vector-memory
      // instruction costs these walks 50-150 clocks of issue whatever it moves)
      if (second && lane < kTipBytes / 4) {
        const int8_t* ts = sgpr_ptr(a.tip_states + (size_t)ch * tip_stride + p0_wg);
        const uint32_t m0 = (uint32_t)(uintptr_t)(lds_ptr)(tips_lds[buf][c]);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"((uint32_t)lane * 4), "s"(ts), "s"(m0) : "memory");
      }
    } else {
      const double* src = matP + (size_t)(ch - n) * K * kAaPack + (second ? 2 * 128 : 0);
      dma_run<2>(src, lane16, ops_lds[buf][c] + (second ? 2 * 128 : 0));
    }
  };
  // the states of a tip child, this lane's M patterns: from the bytes stage() put behind the
  // child's table (read first thing in the visit, used after the next visit's requests)
  auto tip_states_of = [&](int buf, int c, int (&x)[M]) {
    // (an explicit LDS pointer: left generic, the byte reads become flat loads in some builds)
    const lds_i8_ptr ts = (lds_i8_ptr)(lds_ptr)(tips_lds[buf][c]) + wave * (M * 16) + j;
#pragma unroll
    for (int u = 0; u < M; u++) x[u] = ts[u * 16];
  };
  int xc[2][M];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int u = 0; u < M; u++) xc[c][u] = kAa;
  int v, ch0, ch1, slots;
  {
    const SchedEntry e0 = win.at(0);
    v = sgpr(e0.node);
    ch0 = sgpr(e0.child0);
    ch1 = sgpr(e0.child1);
    slots = sgpr(e0.slots);
  }
  stage(ch0, ch1, 0);
  wait_all_vm();
  lds_barrier();

  double R[M][5];
  int E[M];
#pragma unroll
  for (int u = 0; u < M; u++) {
    E[u] = 0;
#pragma unroll
    for (int t = 0; t < 5; t++) R[u][t] = 0;
  }
  int prev = -1, prev_slots = 0;
  int eloc[M];
#pragma unroll
  for (int u = 0; u < M; u++) eloc[u] = 0;
  // the ring of this wave; lo: first slot that lives in it (slots below are in the arena)
  constexpr int kRingEntry = M * (kAaTileDoubles + 8);  // doubles per entry (16 ints = 8 doubles per tile)
  const int RS = sgpr(a.ring_slots);
  double* const ring = ring_lds + (size_t)sgpr(wave) * RS * kRingEntry;
  int lo = RS > 0 ? 0 : 0x7fffffff;
  // (ring sizes are powers of two: launch_aa_post)
  auto ring_entry = [&](int slot) { return ring + (slot & (RS - 1)) * kRingEntry; };
#ifdef AA_STAMPS  // (timing experiment: where a wave's visit goes, in shader clocks)
  unsigned long long stamp_acc[5] = {0, 0, 0, 0, 0};
  unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#define AA_STAMP(k)                                                  \
  {                                                                  \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
    stamp_acc[k] += now_ - stamp_t;                                  \
    stamp_t = now_;                                                  \
  }
#else
#define AA_STAMP(k)
#endif
  // (the schedule entry of visit i + 1 is read from the LDS window during visit i - 1: the
  // top of a visit then goes straight from the barrier to its requests, without an LDS round
  // trip in front of them)
  SchedEntry ahead = count > 1 ? win.at(1) : SchedEntry{-1, -1, -1, 0};
  for (int i = 0; i < count; i++) {
    const int buf = i & 1;
    if (ch0 < n) tip_states_of(buf, 0, xc[0]);
    if (ch1 < n) tip_states_of(buf, 1, xc[1]);
    int nv = -1, next_c0 = -1, next_c1 = -1, nslots = 0;
    if (i + 1 < count) {
      nv = sgpr(ahead.node);
      next_c0 = sgpr(ahead.child0);
      next_c1 = sgpr(ahead.child1);
      nslots = sgpr(ahead.slots);
      stage(next_c0, next_c1, buf ^ 1);  // the next visit's shared operands: global -> LDS
      if (i + 2 < count) {
        if (i + 2 >= win.base + kSchedWindow) fill(i + 2);
        ahead = win.at(i + 2);
      }
    }
    AA_STAMP(0);  // requests for the next visit issued
    if constexpr (GRAD) {
      // Gradient form: every internal vector goes to the arena, by node, for the pre-order
      // kernel (P L of the child taken from registers, L of a node consumed later: see
      // aa_post_kernel), and every node keeps its own exponent.
      double S[2][M][5];
      int Ec[2][M];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int ch = c ? ch1 : ch0;
        const double* shared = ops_lds[buf][c];
        if (ch < n) {
#pragma unroll
          for (int u = 0; u < M; u++) {
            Ec[c][u] = 0;
            const double* col = shared + xc[c][u] * kAa + g;
#pragma unroll
            for (int t = 0; t < 5; t++) S[c][u][t] = col[4 * t];
          }
        } else {
          double A[10], L[M][5];
          if (ch == prev) {
#pragma unroll
            for (int u = 0; u < M; u++) {
              Ec[c][u] = E[u];
#pragma unroll
              for (int t = 0; t < 5; t++) L[u][t] = R[u][t];
            }
          } else {
            // (round 5: a vector that was left on the stack is ALSO in the LDS ring while it is
            // near the top -- the arena copy is for the pre-order kernel, this read need not
            // fetch it back: 44 of the launch's 200 GB)
            const int idx = ch - n, slot = (slots >> (8 + 8 * c)) & 0xff;
            if (slot >= lo) {
              const double* src = ring_entry(slot);
              load_tiles_lds<M>(src, lane, L);
              const int* ex = reinterpret_cast<const int*>(src + M * kAaTileDoubles);
#pragma unroll
              for (int u = 0; u < M; u++) Ec[c][u] = ex[u * 16 + j];
            } else {
              load_tiles<M>(arena + idx * arena_stride, lane, L);
#pragma unroll
              for (int u = 0; u < M; u++) Ec[c][u] = exp_cum[idx * exp_stride + u * 16 + j];
            }
          }
          pack_operands(shared, lane, A);
          mat_apply<M>(A, L, S[c]);
        }
      }
      AA_STAMP(1);
      // the deferred stores of the previous visit (its vector is still in R); before them the
      // visit's one wait for the DMA of the next visit's operands (the children's exponents are
      // consumed here so that no later use makes the compiler wait after the stores)
      int Es[M];
#pragma unroll
      for (int u = 0; u < M; u++) {
        Es[u] = Ec[0][u] + Ec[1][u];
        asm volatile("" : "+v"(Es[u]));
      }
      wait_all_vm();
      AA_STAMP(2);
      if (prev >= 0 && active) {
        double* dstp = arena + (size_t)(prev - n) * arena_stride;
        if (prev == ch0) store_tiles_async<M>(dstp, lane, S[0]);
        else if (prev == ch1) store_tiles_async<M>(dstp, lane, S[1]);
        else {
          store_tiles_async<M>(dstp, lane, R);
          if (RS > 0) {  // left on the stack: into the ring as well (an entry pushed out is simply dropped)
            const int dst = prev_slots & 0xff;
            if (dst < lo) lo = dst;
            if (dst - lo >= RS) lo = lo + 1;
            double* rp = ring_entry(dst);
            store_tiles_lds<M>(rp, lane, R);
            if (g == 0) {
              int* ex = reinterpret_cast<int*>(rp + M * kAaTileDoubles);
#pragma unroll
              for (int u = 0; u < M; u++) ex[u * 16 + j] = E[u];
            }
          }
        }
        if (g == 0) {
          // (the cumulative exponent is read back only for a node that was left on the stack)
          const bool stacked = prev != ch0 && prev != ch1;
#pragma unroll
          for (int u = 0; u < M; u++) {
            store_async(exp_loc + (size_t)(prev - n) * exp_stride, u * 16 + j, eloc[u]);
            if (stacked) store_async(exp_cum + (size_t)(prev - n) * exp_stride, u * 16 + j, E[u]);
          }
        }
      }
      // (the product after the stores, which may read S: fewer registers)
#pragma unroll
      for (int u = 0; u < M; u++) {
        double Tn[5];
#pragma unroll
        for (int t = 0; t < 5; t++) Tn[t] = S[0][u][t] * S[1][u][t];
        // column sum over the 20 states: this lane's five rows with vector adds, then ONE product
        // with a ones matrix for the four row groups (only the exponent of the sum is used)
        const double part = ((Tn[0] + Tn[1]) + (Tn[2] + Tn[3])) + Tn[4];
        const double cs = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, part, 0.0, 0, 0, 0);
        const int e = cs > 0.0 ? __builtin_amdgcn_frexp_exp(cs) - 1 : 0;
#pragma unroll
        for (int t = 0; t < 5; t++) R[u][t] = ldexp(Tn[t], -e);
        eloc[u] = e;
        E[u] = Es[u] + e;
      }
    } else {
      // Log-likelihood form (round 5).  In a post-order over the internal nodes the node visited
      // just before v is a child of v unless both children of v are tips (the last node a
      // depth-first walk leaves before v is the root of the subtree it finished last).  A visit
      // is therefore one of two kinds, told apart by a scalar compare:
      //   (T) two tips: the previous visit's vector, still in R, is not consumed here and is
      //       PUSHED first (to the ring, or to its arena slot); the new vector is the product of
      //       two table columns, written over it, and is not rescaled (far from underflow; powers
      //       of two are exact, so the result is bit-identical -- a third of the nodes of a
      //       random tree);
      //   (P) one child is the previous visit's node and is multiplied where it lies, in R; the
      //       other child is a tip or the top of the stack (ring, else arena).
      // The element-wise product of the two children's vectors commutes bit for bit, so the
      // code is written for "the previous node" and "the other child" instead of child 0 and
      // child 1: no source of a vector has to be merged into common registers (the merged form
      // copied R for every visit: 12 64-bit moves, on the one pipe the FP64 MFMAs also occupy).
      const bool two_tips = ch0 < n && ch1 < n;
      if (two_tips) {
        if (prev >= 0 && active) {
          const int dst = prev_slots & 0xff;
          if (RS > 0) {
            if (dst < lo) lo = dst;  // (the stack has unwound below the ring: it starts anew here)
            if (dst - lo >= RS) {
              // the ring is full: its oldest entry goes to its arena slot
              const double* old = ring_entry(lo);
              double Lo[M][5];
              load_tiles_lds<M>(old, lane, Lo);
              const int* ex = reinterpret_cast<const int*>(old + M * kAaTileDoubles);
              int eo[M];
#pragma unroll
              for (int u = 0; u < M; u++) eo[u] = ex[u * 16 + j];
              store_tiles_async<M>(sgpr_ptr(arena + lo * arena_stride), lane, Lo);
              if (g == 0) {
                int32_t* eb = sgpr_ptr(exp_cum + lo * exp_stride);
#pragma unroll
                for (int u = 0; u < M; u++) store_async(eb, u * 16 + j, eo[u]);
              }
              lo = lo + 1;
            }
            double* dstp = ring_entry(dst);
            store_tiles_lds<M>(dstp, lane, R);
            if (g == 0) {
              int* ex = reinterpret_cast<int*>(dstp + M * kAaTileDoubles);
#pragma unroll
              for (int u = 0; u < M; u++) ex[u * 16 + j] = E[u];
            }
          }
        }
        double Tn[M][5];
#pragma unroll
        for (int u = 0; u < M; u++) {
          const double* col0 = ops_lds[buf][0] + xc[0][u] * kAa + g;
          const double* col1 = ops_lds[buf][1] + xc[1][u] * kAa + g;
#pragma unroll
          for (int t = 0; t < 5; t++) Tn[u][t] = col0[4 * t] * col1[4 * t];
        }
        if (RS == 0) {
          // no ring (small launches): the push is a store to the arena, issued AFTER the visit's
          // wait so that it has the whole next visit to be acknowledged
          wait_all_vm();
          if (prev >= 0 && active) {
            const int dst = prev_slots & 0xff;
            store_tiles_async<M>(arena + dst * arena_stride, lane, R);
            if (g == 0) {
#pragma unroll
              for (int u = 0; u < M; u++) store_async(exp_cum + dst * exp_stride, u * 16 + j, E[u]);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < M; u++) {
#pragma unroll
          for (int t = 0; t < 5; t++) R[u][t] = Tn[u][t];
          E[u] = 0;
        }
      } else {
        const int cb = ch1 == prev ? 0 : 1;  // the child that is NOT the previous visit's node
        const int chb = cb ? ch1 : ch0;
        const double* ops_a = ops_lds[buf][cb ^ 1];
        const double* ops_b = ops_lds[buf][cb];
        double SA[M][5], SB[M][5];
        int Es[M];
        {
          double A[10];
          pack_operands(ops_a, lane, A);
          mat_apply<M>(A, R, SA);
        }
        if (chb < n) {
#pragma unroll
          for (int u = 0; u < M; u++) {
            const int x = cb ? xc[1][u] : xc[0][u];
            const double* col = ops_b + x * kAa + g;
#pragma unroll
            for (int t = 0; t < 5; t++) SB[u][t] = col[4 * t];
            Es[u] = E[u];
          }
        } else {
          const int idx = (slots >> (8 + 8 * cb)) & 0xff;
          double A[10], L[M][5];
          int Eb[M];
          if (idx >= lo) {  // in the ring
            const double* src = ring_entry(idx);
            load_tiles_lds<M>(src, lane, L);
            const int* ex = reinterpret_cast<const int*>(src + M * kAaTileDoubles);
#pragma unroll
            for (int u = 0; u < M; u++) Eb[u] = ex[u * 16 + j];
          } else {
            load_tiles<M>(arena + idx * arena_stride, lane, L);
#pragma unroll
            for (int u = 0; u < M; u++) Eb[u] = exp_cum[idx * exp_stride + u * 16 + j];
            // (the compiler's wait for these loads belongs in THIS branch: placed after the merge
            // with the ring source it would also wait -- vmcnt counts in order -- for the next
            // visit's operands, requested above, on every visit)
#pragma unroll
            for (int u = 0; u < M; u++) {
              asm volatile("" : "+v"(Eb[u]));
#pragma unroll
              for (int t = 0; t < 5; t++) asm volatile("" : "+v"(L[u][t]));
            }
          }
          pack_operands(ops_b, lane, A);
          mat_apply<M>(A, L, SB);
#pragma unroll
          for (int u = 0; u < M; u++) Es[u] = E[u] + Eb[u];
        }
#pragma unroll
        for (int u = 0; u < M; u++) {
          double Tn[5];
#pragma unroll
          for (int t = 0; t < 5; t++) Tn[t] = SA[u][t] * SB[u][t];
          // column sum over the 20 states: this lane's five rows with vector adds, then ONE
          // product with a ones matrix for the four row groups (only the exponent of the sum is used)
          const double part = ((Tn[0] + Tn[1]) + (Tn[2] + Tn[3])) + Tn[4];
          const double cs = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, part, 0.0, 0, 0, 0);
          const int e = cs > 0.0 ? __builtin_amdgcn_frexp_exp(cs) - 1 : 0;
#pragma unroll
          for (int t = 0; t < 5; t++) R[u][t] = ldexp(Tn[t], -e);
          E[u] = Es[u] + e;
        }
      }
#ifdef AA_STAMPS
#pragma unroll
      for (int u = 0; u < M; u++)
#pragma unroll
        for (int t = 0; t < 5; t++) asm volatile("" : "+v"(R[u][t]));
#endif
      AA_STAMP(1);  // the visit's work: push or products, rescaling
      // the visit's one wait, for the next visit's operands
      if (!(two_tips && RS == 0)) wait_all_vm();
      AA_STAMP(2);  // waited for the next visit's operands
    }
    prev_slots = slots;
#ifdef AA_STAMPS
#pragma unroll
    for (int u = 0; u < M; u++)
#pragma unroll
      for (int t = 0; t < 5; t++) asm volatile("" : "+v"(R[u][t]));
#endif
    AA_STAMP(3);  // pushes (ring or arena) and rescaling
    // the barrier publishes the other buffer (its DMA was waited for above) and tells
    // everyone this visit's buffer has been read
    lds_barrier();
    AA_STAMP(4);  // barrier
    prev = v;
    v = nv;
    ch0 = next_c0;
    ch1 = next_c1;
    slots = nslots;
  }
#ifdef AA_STAMPS
  if (!GRAD && lane == 0 && (blockIdx.x % 4001) == 7)
    printf("aa_post_wg block %d wave %d visits %d: issue %llu compute %llu wait %llu store+rescale %llu barrier %llu (clocks per visit)\n",
           (int)blockIdx.x, wave, count, stamp_acc[0] / count, stamp_acc[1] / count, stamp_acc[2] / count,
           stamp_acc[3] / count, stamp_acc[4] / count);
#endif
  if (!active) return;
  if (GRAD && g == 0) {  // (the root's vector itself is not needed)
#pragma unroll
    for (int u = 0; u < M; u++) exp_loc[(size_t)(prev - n) * exp_stride + u * 16 + j] = eloc[u];
  }
  // root: sum_i pi_i L_root[i] per pattern (scaled) and its exponent
  double pi[5];
#pragma unroll
  for (int t = 0; t < 5; t++) pi[t] = a.model->pi[4 * t + g];
  const size_t rbase = ((size_t)el * K + cat) * tiles * 16 + p0;
#pragma unroll
  for (int u = 0; u < M; u++) {
    double s = 0;
#pragma unroll
    for (int t = 0; t < 5; t++) s += pi[t] * R[u][t];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0) {
      a.root_val[rbase + u * 16 + j] = s;
      a.root_exp[rbase + u * 16 + j] = E[u];
    }
  }
}

// ------------------------------------------------------------------------
// Root: the categories meet (site likelihood), log-likelihood partial sums, and the weight
// the root's pre-order vector carries: w_p cw_k 2^(E_k - Emax) / site_p  (the pre-order
// recursion is linear, so every edge derivative is then a plain sum over patterns).
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aa_root_kernel(AaWalkArgs a) {
  __shared__ double red[256];
  const int el = blockIdx.y, tree = a.eval_offset + el;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const size_t tp = (size_t)a.tiles * 16;
  const DevModel& m = a.models[tree];
  double ll = 0;
  if (p < (int)tp) {
    const size_t base = (size_t)el * a.K * tp + p;
    if (p < a.P) {
      int emax = INT_MIN;
      for (int k = 0; k < a.K; k++) emax = max(emax, a.root_exp[base + k * tp]);
      double site = 0;
      for (int k = 0; k < a.K; k++)
        site += m.cat_weight[k] * ldexp(a.root_val[base + k * tp], a.root_exp[base + k * tp] - emax);
      const double w = a.weights[p];
      ll = w * (log(site) + emax * 0.6931471805599453094);
      if (a.root_scale)
        for (int k = 0; k < a.K; k++)
          a.root_scale[base + k * tp] =
              w * m.cat_weight[k] * ldexp(1.0, a.root_exp[base + k * tp] - emax) / site;
    } else if (a.root_scale) {
      for (int k = 0; k < a.K; k++) a.root_scale[base + k * tp] = 0.0;
    }
  }
  red[threadIdx.x] = ll;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.ll_part[(size_t)tree * a.ll_stride + blockIdx.x] = red[0];
}

// ------------------------------------------------------------------------
// Pre-order + edge derivatives (beagleUpdatePrePartials, beagleCalculateEdgeDerivatives;
// fat_beagle.cpp:144-166), walking the same schedule backwards: a parent before its
// children, and the node visited next is one of the current node's children, whose
// pre-order vector stays in registers.  With L_s = L 2^-E (E: powers of two removed in the
// subtree) and q_s = q 2^E, q_s o L_s is scale free and
//     q_child_s = P_child^T (q_s o P_sib L_sib_s) 2^-e_node
// (e_node: the power removed AT the node), so only e_node is needed here.  The derivative of
// edge x (child of v, sibling y) is  sum (q_v o P_y L_y) . (P_x Q) L_x : for a tip x both
// products are table look-ups.  A child's pre-order vector overwrites its post-order
// vector in the arena (dead once the parent has been visited).
//
// As in the post-order kernel: the schedule in LDS, tip states one visit ahead, everything a
// visit reads from the arena requested at its top, few registers (the derivative products
// are consumed tile by tile) for occupancy.
// ------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(64) void aa_pre_kernel(AaWalkArgs a) {
  __shared__ SchedEntry sched_lds[kSchedWindow];
  const int blocks = a.tiles / M;
  const AaUnit un = aa_unit(blocks, a.evals * a.K);
  if (!un.valid) return;
  const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
  const int el = un.ec / a.K, cat = un.ec - el * a.K, blk = un.blk;
  const int tree = a.eval_offset + el;
  const int n = a.n, N = a.N, K = a.K;
  const int p0 = sgpr(blk * M * 16);
  const size_t tiles = a.tiles, tip_stride = tiles * 16;
  double* arena = a.arena + (((size_t)el * (n - 1) * K + cat) * tiles + (size_t)blk * M) * kAaTileDoubles;
  const size_t arena_stride = (size_t)K * tiles * kAaTileDoubles;
  const int32_t* exp_loc = a.exp_loc + ((size_t)el * (n - 1) * K + cat) * tiles * 16 + p0;
  const size_t exp_stride = (size_t)K * tiles * 16;
  const size_t mbase = ((size_t)el * (n - 1) * K + cat) * kAaPack;
  const double* matP = a.matP + mbase;
  const double* matPT = a.matPT + mbase;
  const double* tipP = a.tipP + ((size_t)el * n * K + cat) * kAaTipTable;
  const double* tipPQ = a.tipPQ + ((size_t)el * n * K + cat) * kAaTipTable;
  double* gp = a.g_part + (((size_t)el * K + cat) * blocks + blk) * N;
  const int count = n - 1;
  SchedWindow win{a.sched + (size_t)tree * count, sched_lds, count, 0, lane};
  win.fill_upto(count - 1);

  auto stage0 = [&](int c0, int c1, int (&x)[2][M]) {
    if (c0 < n) load_tip_states<M>(a.tip_states + (size_t)c0 * tip_stride, p0, lane, x[0]);
    if (c1 < n) load_tip_states<M>(a.tip_states + (size_t)c1 * tip_stride, p0, lane, x[1]);
  };
  int xc[2][M], xn[2][M];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int u = 0; u < M; u++) xc[c][u] = xn[c][u] = kAa;
  int v, ch[2];
  {
    const SchedEntry e0 = win.at(count - 1);
    v = sgpr(e0.node);
    ch[0] = sgpr(e0.child0);
    ch[1] = sgpr(e0.child1);
  }
  stage0(ch[0], ch[1], xc);

  // the node's pre-order vector: computed by the parent's visit and kept here when this
  // visit follows it immediately, else read back from the arena
  double q[M][5];
#pragma unroll
  for (int u = 0; u < M; u++)
#pragma unroll
    for (int t = 0; t < 5; t++) q[u][t] = 0;
  int kept = -1;
  for (int i = count - 1; i >= 0; i--) {
    int next = -1, nc0 = -1, nc1 = -1;
    if (i > 0) {
      if (i - 1 < win.base) win.fill_upto(i - 1);
      const SchedEntry s1 = win.at(i - 1);
      next = sgpr(s1.node);
      nc0 = sgpr(s1.child0);
      nc1 = sgpr(s1.child1);
      stage0(nc0, nc1, xn);
    }
    if (i == count - 1) {
      const size_t rbase = ((size_t)el * K + cat) * tiles * 16 + p0;
#pragma unroll
      for (int u = 0; u < M; u++) {
        const double rs = a.root_scale[rbase + u * 16 + j];
#pragma unroll
        for (int t = 0; t < 5; t++) q[u][t] = a.model->pi[4 * t + g] * rs;
      }
    } else if (kept != v) {
      load_tiles<M>(arena + (size_t)(v - n) * arena_stride, lane, q);
    }
    // the children's post-order vectors (internal children), requested together
    double L[2][M][5];
#pragma unroll
    for (int c = 0; c < 2; c++)
      if (ch[c] >= n) load_tiles<M>(arena + (size_t)(ch[c] - n) * arena_stride, lane, L[c]);
#pragma unroll
    for (int u = 0; u < M; u++) {
      const int ev = exp_loc[(size_t)(v - n) * exp_stride + u * 16 + j];
#pragma unroll
      for (int t = 0; t < 5; t++) q[u][t] = ldexp(q[u][t], -ev);
    }
    // S[c] = P_c L_c (tip: column of P)
    double S[2][M][5];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      if (ch[c] < n) {
        tip_gather<M>(tipP + (size_t)ch[c] * K * kAaTipTable, xc[c], lane, S[c]);
      } else if (ch[c] == next) {
        // (the child the post-order pass took from registers: the arena holds P L already)
#pragma unroll
        for (int u = 0; u < M; u++)
#pragma unroll
          for (int t = 0; t < 5; t++) S[c][u][t] = L[c][u][t];
      } else {
        double A[10];
        load_pack(matP + (size_t)(ch[c] - n) * K * kAaPack, lane, A);
        mat_apply<M>(A, L[c], S[c]);
      }
    }
    // X_c = sum u_c . Q S_c with u_c = q o S[sibling] (q carries 2^-e); tip: the column of
    // P Q; the derivative products consumed tile by tile
    double X[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      double x = 0;
      if (ch[c] < n) {
        double D[M][5];
        tip_gather<M>(tipPQ + (size_t)ch[c] * K * kAaTipTable, xc[c], lane, D);
#pragma unroll
        for (int u = 0; u < M; u++)
#pragma unroll
          for (int t = 0; t < 5; t++) x += (q[u][t] * S[1 - c][u][t]) * D[u][t];
      } else {
        double A[10];
        load_pack(a.model->Qpack, lane, A);
#pragma unroll
        for (int u = 0; u < M; u++) {
          double S1[1][5], D1[1][5];
#pragma unroll
          for (int t = 0; t < 5; t++) S1[0][t] = S[c][u][t];
          mat_apply<1>(A, S1, D1);
#pragma unroll
          for (int t = 0; t < 5; t++) x += (q[u][t] * S[1 - c][u][t]) * D1[0][t];
        }
      }
      X[c] = wave_sum_mfma(x);
    }
    // u_c, kept in S[sibling]
#pragma unroll
    for (int u = 0; u < M; u++)
#pragma unroll
      for (int t = 0; t < 5; t++) {
        const double u0 = q[u][t] * S[1][u][t], u1 = q[u][t] * S[0][u][t];
        S[1][u][t] = u0;
        S[0][u][t] = u1;
      }
    if (lane == 0) {
      gp[ch[0]] = X[0];
      gp[ch[1]] = X[1];
    }
    // q_c = P_c^T u_c for internal children: into the arena, or kept for the next visit
    int keep_next = -1;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      if (ch[c] < n || ch[c] == next) continue;
      double A[10], qc[M][5];
      load_pack(matPT + (size_t)(ch[c] - n) * K * kAaPack, lane, A);
      mat_apply<M>(A, S[1 - c], qc);
      store_tiles<M>(arena + (size_t)(ch[c] - n) * arena_stride, lane, qc);
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      if (ch[c] < n || ch[c] != next) continue;
      double A[10];
      load_pack(matPT + (size_t)(ch[c] - n) * K * kAaPack, lane, A);
      mat_apply<M>(A, S[1 - c], q);
      keep_next = ch[c];
    }
    kept = keep_next;
    v = next;
    ch[0] = nc0;
    ch[1] = nc1;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int u = 0; u < M; u++) xc[c][u] = xn[c][u];
  }
}

// ------------------------------------------------------------------------
// Pre-order + derivatives, workgroup form (see aa_post_wg_kernel): the three matrices of
// an internal child (P, P^T: 8 KB) or the two column tables of a tip child are staged
// in LDS once per workgroup of four pattern blocks, one visit ahead.
//
// Round 5: the top of the stack of pending pre-order vectors in LDS.  A visit hands the
// pre-order vector of ONE child to the next visit in registers (the child the walk descends
// into); the other internal child's vector waits until the walk comes back for it -- until
// round 5 in the arena, by node: written once, read once.  The walk is the post-order
// schedule backwards, so these pending vectors are a stack with the SAME positions the
// schedule's slot numbers give the post-order vectors (a push here is a pop there): the
// child's slot in its parent's entry on the push, the node's own slot on the pop.  Slots
// [lo, lo + S) of a wave are a ring in LDS (S = a.pre_ring_slots, a power of two; 20 KB per
// workgroup and entry: one fits beside the operand buffers at the two workgroups per CU the
// registers allow); a push that finds the ring full spills its oldest entry to the arena, a
// pop below lo reads the arena as before.  Same values in the same registers: bit-identical.
// Of the ~171 vectors a random 512-taxon tree parks, 67 % are taken back before another is
// parked on top: one entry keeps them on the CU.
// ------------------------------------------------------------------------
#ifndef AA_PRE_MIN_WGS
#define AA_PRE_MIN_WGS 2
#endif
constexpr int kPreOps = 2 * kAaPack;  // doubles per child in LDS: P | P^T  (tip: tipP | tipPQ)
template <int M>
__global__ __launch_bounds__(64 * kPreWaves, AA_PRE_MIN_WGS) void aa_pre_wg_kernel(AaWalkArgs a) {
  extern __shared__ __align__(16) double pre_lds[];  // [2][2][kPreOps] doubles, Q, the schedule window, the ring
  double (*ops_lds)[2][kPreOps] = reinterpret_cast<double (*)[2][kPreOps]>(pre_lds);
  double* q_lds = pre_lds + 2 * 2 * kPreOps;
  SchedEntry* sched_lds = reinterpret_cast<SchedEntry*>(q_lds + kAaPack);
  double* ring_lds = reinterpret_cast<double*>(sched_lds + kSchedWindow);  // [wave][S][M tiles x 320]
  __shared__ int8_t tips_lds[2][2][kPreWaves * M * 16];  // [buffer][child]: a tip child's states
  const int blocks = a.tiles / M;
  const int wgs = (blocks + kPreWaves - 1) / kPreWaves;
  const AaUnit un = aa_unit(wgs, a.evals * a.K);
  if (!un.valid) return;
  const int tid = threadIdx.x, wave = sgpr(tid >> 6), lane = tid & 63, g = lane >> 4, j = lane & 15;
  const int el = un.ec / a.K, cat = un.ec - el * a.K;
  const int blk_raw = un.blk * kPreWaves + wave;
  const bool active = blk_raw < blocks;
  const int blk = active ? blk_raw : blocks - 1;
  const int tree = a.eval_offset + el;
  const int n = a.n, N = a.N, K = a.K;
  const int p0 = sgpr(blk * M * 16);
  constexpr int kTipBytes = kPreWaves * M * 16;  // tip states of the workgroup's patterns
  const size_t p0_wg = (size_t)un.blk * kTipBytes;
  const size_t tiles = a.tiles, tip_stride = tiles * 16;
  double* arena = sgpr_ptr(a.arena + (((size_t)el * (n - 1) * K + cat) * tiles + (size_t)blk * M) * kAaTileDoubles);
  const size_t arena_stride = (size_t)K * tiles * kAaTileDoubles;
  const int32_t* exp_loc = sgpr_ptr(a.exp_loc + ((size_t)el * (n - 1) * K + cat) * tiles * 16 + p0);
  const size_t exp_stride = (size_t)K * tiles * 16;
  const size_t mbase = ((size_t)el * (n - 1) * K + cat) * kAaPack;
  const double* matP = a.matP + mbase;
  const double* matPT = a.matPT + mbase;
  const double* tipP = a.tipP + ((size_t)el * n * K + cat) * kAaTipTable;
  const double* tipPQ = a.tipPQ + ((size_t)el * n * K + cat) * kAaTipTable;
  double* gp = sgpr_ptr(a.g_part + (((size_t)el * K + cat) * blocks + blk) * N);
  const int count = n - 1;
  for (int idx = tid; idx < kAaPack; idx += kPreThreads) q_lds[idx] = a.model->Qpack[idx];
  SchedWindow win{a.sched + (size_t)tree * count, sched_lds, count, 0, tid};
  auto fill_upto = [&](int last) {
    win.base = last - kSchedWindow + 1;
    if (win.base < 0) win.base = 0;
    __syncthreads();
    if (tid < kSchedWindow && win.base + tid < count) sched_lds[tid] = win.sched[win.base + tid];
    __syncthreads();
  };
  fill_upto(count - 1);

  // staging by LDS-DMA into buffer `buf`: internal child P | P^T (2 x 4 pieces), tip child
  // its two column tables (2 x 4 pieces) at offsets 0 and kAaPack
  const int wave_s = sgpr(wave);
  const uint32_t lane16 = (uint32_t)lane * 16;
  // four operand tables per visit (child 0 and child 1: P | P^T, or the two column tables of
  // a tip), ONE per wave: wave 2c takes child c's first table, wave 2c + 1 its second
  auto stage = [&](int c0, int c1, int buf) {
    static_assert(kPreWaves == 4, "operand staging deals one table to each of four waves");
    const int c = wave_s >> 1, second = wave_s & 1;
    const int ch = c ? c1 : c0;
    double* dst = ops_lds[buf][c] + (second ? kAaPack : 0);
    if (ch < n) {
      const double* src = (second ? tipPQ : tipP) + (size_t)ch * K * kAaTipTable;
      dma_whole<4>(src, lane16, dst);
      // (the tip's states for the workgroup's patterns: see aa_post_wg_kernel)
      if (second && lane < kTipBytes / 4) {
        const int8_t* ts = sgpr_ptr(a.tip_states + (size_t)ch * tip_stride + p0_wg);
        const uint32_t m0 = (uint32_t)(uintptr_t)(lds_ptr)(tips_lds[buf][c]);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"((uint32_t)lane * 4), "s"(ts), "s"(m0) : "memory");
      }
    } else {
      const double* src = (second ? matPT : matP) + (size_t)(ch - n) * K * kAaPack;
      dma_whole<4>(src, lane16, dst);
    }
  };
  auto lds_pack = [&](const double* base, double (&A)[10]) { pack_operands(base, lane, A); };
  auto lds_cols = [&](const double* table, const int (&x)[M], double (&S)[M][5]) {
#pragma unroll
    for (int u = 0; u < M; u++) {
      const double* col = table + x[u] * kAa + g;
#pragma unroll
      for (int t = 0; t < 5; t++) S[u][t] = col[4 * t];
    }
  };
  auto tip_states_of = [&](int buf, int c, int (&x)[M]) {
    // (an explicit LDS pointer: left generic, the byte reads become flat loads in some builds)
    const lds_i8_ptr ts = (lds_i8_ptr)(lds_ptr)(tips_lds[buf][c]) + wave * (M * 16) + j;
#pragma unroll
    for (int u = 0; u < M; u++) x[u] = ts[u * 16];
  };
  int xc[2][M];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int u = 0; u < M; u++) xc[c][u] = kAa;
  int v, ch[2], vslots;
  {
    const SchedEntry e0 = win.at(count - 1);
    v = sgpr(e0.node);
    ch[0] = sgpr(e0.child0);
    ch[1] = sgpr(e0.child1);
    vslots = sgpr(e0.slots);
  }
  // the ring of this wave: entries by slot & (RS - 1), the node each entry belongs to (its
  // arena address, should it be spilled), lo: the first slot that lives in it
  const int RS = sgpr(a.pre_ring_slots);
  constexpr int kPreRingEntry = M * kAaTileDoubles;
  double* const ring = ring_lds + (size_t)wave * RS * kPreRingEntry;
  int lo = RS > 0 ? 0 : 0x7fffffff;
  int ring_node[2] = {-1, -1};
  auto ring_entry = [&](int slot) { return ring + (slot & (RS - 1)) * kPreRingEntry; };
  // Everything a visit reads from global memory -- its children's post-order vectors, its own
  // pre-order vector unless the previous visit hands it over in registers, its exponents, the
  // tip states, the LDS operands -- is requested during the visit BEFORE it, each into the
  // registers that visit has just finished with, and retired by that visit's one vmcnt(0): a
  // visit's own arithmetic never waits for memory.  (Round 5: the node's own vector and
  // exponents are requested at the TOP of the visit before, the vector into registers of its
  // own -- until then they were the last thing requested before the wait, which every visit
  // then paid in full: 1 400-1 500 of a visit's 9 000 clocks in the stamps.  The registers are
  // there: the operand buffers in LDS allow two waves per SIMD whatever the kernel uses.)
  double q[M][5], qn[M][5], L[2][M][5];
  int ev[M];
#pragma unroll
  for (int u = 0; u < M; u++) {
    ev[u] = 0;
#pragma unroll
    for (int t = 0; t < 5; t++) q[u][t] = qn[u][t] = L[0][u][t] = L[1][u][t] = 0;
  }
  auto fetch_children = [&](int node, int c0, int c1) {
    if (c0 >= n) load_tiles<M>(arena + (size_t)(c0 - n) * arena_stride, lane, L[0]);
    if (c1 >= n) load_tiles<M>(arena + (size_t)(c1 - n) * arena_stride, lane, L[1]);
  };
  // (own_slot: the node's stack position; a parked vector that is still in the ring is read
  // from LDS -- this is called in a visit without pushes: one that parks a vector descends into
  // the other child, whose vector it keeps in registers)
  auto fetch_node = [&](int node, bool load_q, int own_slot) {
    if (load_q) {
      if (own_slot >= lo) load_tiles_lds<M>(ring_entry(own_slot), lane, qn);
      else load_tiles<M>(arena + (size_t)(node - n) * arena_stride, lane, qn);
    }
#pragma unroll
    for (int u = 0; u < M; u++) ev[u] = exp_loc[(size_t)(node - n) * exp_stride + u * 16 + j];
  };
  stage(ch[0], ch[1], (count - 1) & 1);
  fetch_children(v, ch[0], ch[1]);
  fetch_node(v, false, 0);
  {
    const size_t rbase = ((size_t)el * K + cat) * tiles * 16 + p0;
#pragma unroll
    for (int u = 0; u < M; u++) {
      const double rs = a.root_scale[rbase + u * 16 + j];
#pragma unroll
      for (int t = 0; t < 5; t++) q[u][t] = a.model->pi[4 * t + g] * rs;
    }
  }
  wait_all_vm();
  lds_barrier();

#ifdef AA_STAMPS  // (timing experiment: where a wave's visit goes, in shader clocks)
  unsigned long long pstamp_acc[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long pstamp_t = __builtin_amdgcn_s_memtime();
#define AA_PSTAMP(k)                                                 \
  {                                                                  \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
    pstamp_acc[k] += now_ - pstamp_t;                                \
    pstamp_t = now_;                                                 \
  }
#define AA_PIN(arr)                                                  \
  _Pragma("unroll") for (int u_ = 0; u_ < M; u_++) _Pragma("unroll") for (int t_ = 0; t_ < 5; t_++) asm volatile("" : "+v"(arr[u_][t_]));
#else
#define AA_PSTAMP(k)
#define AA_PIN(arr)
#endif
  SchedEntry ahead = count > 1 ? win.at(count - 2) : SchedEntry{-1, -1, -1, 0};
  for (int i = count - 1; i >= 0; i--) {
    const int buf = i & 1;
    if (ch[0] < n) tip_states_of(buf, 0, xc[0]);
    if (ch[1] < n) tip_states_of(buf, 1, xc[1]);
    int next = -1, nc0 = -1, nc1 = -1, nslots = 0;
    if (i > 0) {
      // (the entry of visit i - 1 was read from the LDS window during visit i + 1)
      next = sgpr(ahead.node);
      nc0 = sgpr(ahead.child0);
      nc1 = sgpr(ahead.child1);
      nslots = sgpr(ahead.slots);
      stage(nc0, nc1, buf ^ 1);
      if (i > 1) {
        if (i - 2 < win.base) fill_upto(i - 2);
        ahead = win.at(i - 2);
      }
    }
    AA_PSTAMP(0);  // schedule entry, operand DMA of the next visit issued
#pragma unroll
    for (int u = 0; u < M; u++)
#pragma unroll
      for (int t = 0; t < 5; t++) q[u][t] = ldexp(q[u][t], -ev[u]);
    // ev is done with: the next node's exponents, and its vector unless this visit computes it
    const bool keep = i > 0 && (next == ch[0] || next == ch[1]);
#pragma unroll
    for (int u = 0; u < M; u++) asm volatile("" : "+v"(ev[u]));
    if (i > 0) fetch_node(next, !keep, nslots & 0xff);
    // S[c] = P_c L_c (tip: column of P)
    double S[2][M][5];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const double* shared = ops_lds[buf][c];
      if (ch[c] < n) {
        lds_cols(shared, xc[c], S[c]);
      } else if (ch[c] == next) {
        // (the child the post-order pass took from registers: the arena holds P L already)
#pragma unroll
        for (int u = 0; u < M; u++)
#pragma unroll
          for (int t = 0; t < 5; t++) S[c][u][t] = L[c][u][t];
      } else {
        double A[10];
        lds_pack(shared, A);
        mat_apply<M>(A, L[c], S[c]);
      }
    }
    // L is done with: the next visit's vectors go into it (and its tip states into xn); the
    // compiler may not move these loads up: they follow a memory barrier
    AA_PIN(S[0]);
    AA_PIN(S[1]);
    AA_PSTAMP(1);  // S_c = P_c L_c
    asm volatile("" ::: "memory");
    if (i > 0) fetch_children(next, nc0, nc1);
    AA_PSTAMP(2);  // the next visit's vectors requested
    // X_c = sum (q o S[sibling]) . Q S_c (tip: the column of P Q), tile by tile
    double X[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      double x = 0;
      if (ch[c] < n) {
        double D[M][5];
        lds_cols(ops_lds[buf][c] + kAaPack, xc[c], D);
#pragma unroll
        for (int u = 0; u < M; u++)
#pragma unroll
          for (int t = 0; t < 5; t++) x += (q[u][t] * S[1 - c][u][t]) * D[u][t];
      } else {
        double A[10];
        lds_pack(q_lds, A);
#pragma unroll
        for (int u = 0; u < M; u++) {
          double S1[1][5], D1[1][5];
#pragma unroll
          for (int t = 0; t < 5; t++) S1[0][t] = S[c][u][t];
          mat_apply<1>(A, S1, D1);
#pragma unroll
          for (int t = 0; t < 5; t++) x += (q[u][t] * S[1 - c][u][t]) * D1[0][t];
        }
      }
      X[c] = wave_sum_mfma(x);
    }
#pragma unroll
    for (int u = 0; u < M; u++)
#pragma unroll
      for (int t = 0; t < 5; t++) {
        const double u0 = q[u][t] * S[1][u][t], u1 = q[u][t] * S[0][u][t];
        S[1][u][t] = u0;
        S[0][u][t] = u1;
      }
    AA_PIN(S[0]);
    AA_PIN(S[1]);
#ifdef AA_STAMPS
    asm volatile("" : "+v"(X[0]), "+v"(X[1]));
#endif
    AA_PSTAMP(3);  // edge derivatives (Q S, two wave sums), messages u_c
    // The visit's one wait: the DMA and the loads for the next visit.  What they loaded is
    // consumed HERE as far as the compiler is concerned, so that it puts its own waits before
    // the stores below and none after them (the stores stay in flight).
    wait_all_vm();
#pragma unroll
    for (int u = 0; u < M; u++) {
      asm volatile("" : "+v"(ev[u]));
#pragma unroll
      for (int t = 0; t < 5; t++) {
        asm volatile("" : "+v"(L[0][u][t]));
        asm volatile("" : "+v"(L[1][u][t]));
        asm volatile("" : "+v"(qn[u][t]));
      }
    }
    if (i > 0 && !keep) {
#pragma unroll
      for (int u = 0; u < M; u++)
#pragma unroll
        for (int t = 0; t < 5; t++) q[u][t] = qn[u][t];
    }
    AA_PSTAMP(4);  // the visit's wait
    // (both edge sums with ONE store instruction: lane 0 child 0's, lane 1 child 1's)
    if (lane < 2 && active) store_async(gp, lane ? ch[1] : ch[0], lane ? X[1] : X[0]);
    // q_c = P_c^T u_c for internal children: into the arena, or handed to the next visit
#pragma unroll
    for (int c = 0; c < 2; c++) {
      if (ch[c] < n || ch[c] == next) continue;
      double A[10], qc[M][5];
      lds_pack(ops_lds[buf][c] + kAaPack, A);
      mat_apply<M>(A, S[1 - c], qc);
      if (RS > 0) {
        // parked in the ring, at the child's stack position
        const int dst = (vslots >> (8 + 8 * c)) & 0xff;
        if (dst < lo) lo = dst;  // (the stack has unwound below the ring: it starts anew here)
        if (dst - lo >= RS) {
          // the ring is full: its oldest entry goes to its node's place in the arena
          double old[M][5];
          load_tiles_lds<M>(ring_entry(lo), lane, old);
          const int old_node = (lo & (RS - 1)) ? ring_node[1] : ring_node[0];
          if (active) store_tiles_async<M>(sgpr_ptr(arena + (size_t)(old_node - n) * arena_stride), lane, old);
          lo = lo + 1;
        }
        store_tiles_lds<M>(ring_entry(dst), lane, qc);
        if (dst & (RS - 1)) ring_node[1] = ch[c];
        else ring_node[0] = ch[c];
      } else if (active) {
        store_tiles_async<M>(arena + (size_t)(ch[c] - n) * arena_stride, lane, qc);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      if (ch[c] < n || ch[c] != next) continue;
      double A[10];
      lds_pack(ops_lds[buf][c] + kAaPack, A);
      mat_apply<M>(A, S[1 - c], q);
    }
    AA_PIN(q);
    AA_PSTAMP(5);  // q_c = P_c^T u_c, parked or handed over
    lds_barrier();
    AA_PSTAMP(6);  // barrier
    v = next;
    ch[0] = nc0;
    ch[1] = nc1;
    vslots = nslots;
  }
#ifdef AA_STAMPS
  if (lane == 0 && (blockIdx.x % 4001) == 7)
    printf("aa_pre_wg block %d wave %d visits %d: issue %llu S %llu fetch %llu derivatives %llu wait %llu q_c %llu barrier %llu (clocks per visit)\n",
           (int)blockIdx.x, wave, count, pstamp_acc[0] / count, pstamp_acc[1] / count, pstamp_acc[2] / count,
           pstamp_acc[3] / count, pstamp_acc[4] / count, pstamp_acc[5] / count, pstamp_acc[6] / count);
#endif
}

// ------------------------------------------------------------------------
// Reduction: log-likelihood per evaluation; branch gradient sum_k r_k X_k and site-model
// numerator sum_k (d r_k / d shape) X_k per edge, blocks summed in order (deterministic).
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aa_reduce_kernel(AaWalkArgs a, int ll_blocks) {
  // grid (1 + ceil(N / 64), evaluations): workgroup 0 sums the log-likelihood partials, the
  // others take 64 edges each, four waves splitting the pattern blocks (fixed order)
  __shared__ double red[256], red2[256];
  const int el = blockIdx.y, tree = a.eval_offset + el;
  const int N = a.N, tid = threadIdx.x;
  if (blockIdx.x == 0) {
    double s = 0;
    for (int i = tid; i < ll_blocks; i += 256) s += a.ll_part[(size_t)tree * a.ll_stride + i];
    red[tid] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    if (tid == 0) a.ll_sum[tree] = red[0];
    return;
  }
  const int edge = (blockIdx.x - 1) * 64 + (tid & 63), part = tid >> 6;
  const DevModel& m = a.models[tree];
  const int blocks = a.tiles / kAaPreTiles;
  const int b0 = (int)((long)blocks * part / 4), b1 = (int)((long)blocks * (part + 1) / 4);
  double br = 0, si = 0;
  if (edge < N - 1) {
    for (int k = 0; k < a.K; k++) {
      const double* gp = a.g_part + ((size_t)el * a.K + k) * blocks * N + edge;
      double x0 = 0, x1 = 0, x2 = 0, x3 = 0;
      int b = b0;
      for (; b + 4 <= b1; b += 4) {
        x0 += gp[(size_t)b * N];
        x1 += gp[(size_t)(b + 1) * N];
        x2 += gp[(size_t)(b + 2) * N];
        x3 += gp[(size_t)(b + 3) * N];
      }
      for (; b < b1; b++) x0 += gp[(size_t)b * N];
      const double xs = (x0 + x1) + (x2 + x3);
      br += m.cat_rate[k] * xs;
      si += m.cat_drate[k] * xs;
    }
  }
  red[tid] = br;
  red2[tid] = si;
  __syncthreads();
  if (part == 0 && edge < N) {
    const int l = tid & 63;
    a.g_sum[((size_t)tree * 2 + 0) * N + edge] = (red[l] + red[64 + l]) + (red[128 + l] + red[192 + l]);
    a.g_sum[((size_t)tree * 2 + 1) * N + edge] = (red2[l] + red2[64 + l]) + (red2[128 + l] + red2[192 + l]);
  }
}

}  // namespace

// ------------------------------------------------------------------------
// Launch wrappers
// ------------------------------------------------------------------------
void launch_aa_model_setup(const double* exch, const double* freqs, AaModel* model,
                           int32_t* status, hipStream_t s) {
  const bool seq = getenv("MI_PHYLO_AA_JACOBI") && std::string(getenv("MI_PHYLO_AA_JACOBI")) == "seq";  // (read per engine)
  if (seq) hipLaunchKernelGGL(aa_model_setup_kernel, dim3(1), dim3(64), 0, s, exch, freqs, model, status);
  else hipLaunchKernelGGL(aa_model_setup_wave_kernel, dim3(1), dim3(64), 0, s, exch, freqs, model, status);
}
void launch_aa_transition(const AaTransitionArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(aa_transition_kernel, dim3(a.N - 1, a.K, a.evals), dim3(256), 0, s, a);
}
int aa_tiles(int P) {
  const int per = 4;  // every tile count a wave may take (2, 4) divides it
  const int tiles = (P + kAaTile - 1) / kAaTile;
  return (tiles + per - 1) / per * per;
}
int aa_ll_blocks(int P) { return (aa_tiles(P) * kAaTile + 255) / 256; }
static unsigned aa_grid(int blocks, int units) {
  const long W = (long)blocks * units;
  return (unsigned)(8 * ((W + 7) / 8));
}
// MI_PHYLO_AA_LDS_PAD=<bytes>: extra dynamic LDS per wave, to study the walk kernels at lower
// occupancy (results are unaffected)
static size_t aa_lds_pad() {
  static const size_t pad = getenv("MI_PHYLO_AA_LDS_PAD") ? strtoul(getenv("MI_PHYLO_AA_LDS_PAD"), nullptr, 10) : 0;
  return pad;
}
// tiles a wave of the post-order kernel takes: two (measured, workgroup form, log-likelihoods
// of 8 trees: 17.1-17.3 ms against 17.7-17.8 with four at three waves per SIMD; gradients 20.8
// against 24.5 in round 3) -- or ONE when the launch would otherwise leave the chip short of
// work (round 6): a rank's share of BASELINE configs[4] under 8-way pattern
// sharding is 6 250 patterns of one tree = 196 two-tile workgroups on 256 CUs, each of its waves
// alone on a SIMD and the visit's dependent chain all there is to hide latency with; one tile
// per wave is twice the workgroups with half the matrix instructions per visit each.
// aa_post_tiles_threshold: workgroups (at two tiles) below which a launch takes one tile.
// MI_PHYLO_AA_POST_TILES=1|2|4 overrides (workgroup form only: the wave form has 2 and 4).
static int aa_post_tiles(const AaWalkArgs& a) {
  static const int forced = getenv("MI_PHYLO_AA_POST_TILES") ? atoi(getenv("MI_PHYLO_AA_POST_TILES")) : 0;
  if (forced == 4 || forced == 2 || forced == 1) return forced;
  static const long threshold =
      getenv("MI_PHYLO_AA_POST_ONE_TILE_BELOW") ? atol(getenv("MI_PHYLO_AA_POST_ONE_TILE_BELOW")) : 2L * device_compute_units();
  const long wgs2 = (long)((a.tiles / 2 + 4 - 1) / 4) * a.evals * a.K;
  return wgs2 < threshold ? 1 : 2;
}
// MI_PHYLO_AA_POST=wave selects the wave-per-block form of the post-order kernel
static bool aa_post_wg() {
  static const bool wg = !(getenv("MI_PHYLO_AA_POST") && std::string(getenv("MI_PHYLO_AA_POST")) == "wave");
  return wg;
}
// Entries of the LDS ring a log-likelihood wave keeps the top of its vector stack in
// (aa_post_wg_kernel); MI_PHYLO_AA_RING=0|1|2|4 overrides (0: every kept vector through the
// arena, the form until round 4).  An entry is 5 KB per wave, 21 KB per workgroup; since the
// operand packs are four pieces instead of five and the schedule window 2 KB, ONE entry fits
// beside them at the four workgroups per CU the registers allow (39.9 KB each), so one entry
// is the default for every launch: less than a third of the arena's HBM traffic (67 % of the
// kept vectors never have another pushed on top of them) and 2-3 % less time; two entries
// cost a workgroup per CU and lose.
static int aa_ring_slots(size_t workgroups) {
  static const int forced = getenv("MI_PHYLO_AA_RING") ? atoi(getenv("MI_PHYLO_AA_RING")) : -1;
  if (forced == 0 || forced == 1 || forced == 2 || forced == 4) return forced;  // (powers of two)
  // (round 6: a launch that leaves the chip short of work -- fewer than two workgroups per CU: a
  // rank's pattern block of one tree -- is bound by its visits' latencies, not by workgroups per
  // CU: four entries keep 97 % of the kept vectors out of the arena; 512 x 6 250 x 4, one tree:
  // 0.581 / 0.566 / 0.564 ms with 1 / 2 / 4 entries)
  return workgroups < 2 * (size_t)device_compute_units() ? 4 : 1;
}
// (what launch_aa_post / launch_aa_pre will choose: for the engine's description of a call)
int aa_post_tiles_per_wave(const AaWalkArgs& a) { return aa_post_wg() ? aa_post_tiles(a) : std::max(2, aa_post_tiles(a)); }
int aa_post_ring_entries(const AaWalkArgs& a) {
  if (!aa_post_wg()) return 0;
  const int m = aa_post_tiles(a), blocks = a.tiles / m;
  int entries = aa_ring_slots(aa_grid((blocks + kPostWaves - 1) / kPostWaves, a.evals * a.K));
  // (a forced size must still fit a CU beside the 24.5 KB of operand buffers and schedule)
  while (entries > 0 && sizeof(double) * (size_t)kPostWaves * entries * m * (kAaTileDoubles + 8) + aa_lds_pad() >
                            160 * 1024 - 25 * 1024)
    entries >>= 1;
  return entries;
}
template <int M, bool GRAD>
static void launch_aa_post_wg(const AaWalkArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  // (static LDS of the kernel: operand buffers, schedule window, tip bytes -- about 25 KB; the
  // opt-in covers static + dynamic.  Both forms use the ring since round 5: ADVICE r5)
  allow_large_lds(reinterpret_cast<const void*>(aa_post_wg_kernel<M, GRAD>), lds + 32 * 1024);
  hipLaunchKernelGGL((aa_post_wg_kernel<M, GRAD>), grid, block, lds, s, a);
}
void launch_aa_post(const AaWalkArgs& a_in, hipStream_t s) {
  AaWalkArgs a = a_in;
  int m = aa_post_tiles(a);
  if (aa_post_wg()) {
    const int blocks = a.tiles / m;
    const dim3 grid(aa_grid((blocks + kPostWaves - 1) / kPostWaves, a.evals * a.K)), block(kPostThreads);
    a.ring_slots = aa_post_ring_entries(a);
    const size_t ring = sizeof(double) * (size_t)kPostWaves * a.ring_slots * m * (kAaTileDoubles + 8);
    const size_t lds = ring + aa_lds_pad();
    if (m == 4) {
      if (a.gradient) launch_aa_post_wg<4, true>(a, grid, block, lds, s);
      else launch_aa_post_wg<4, false>(a, grid, block, lds, s);
    } else if (m == 1) {
      if (a.gradient) launch_aa_post_wg<1, true>(a, grid, block, lds, s);
      else launch_aa_post_wg<1, false>(a, grid, block, lds, s);
    } else {
      if (a.gradient) launch_aa_post_wg<2, true>(a, grid, block, lds, s);
      else launch_aa_post_wg<2, false>(a, grid, block, lds, s);
    }
    return;
  }
  if (m == 1) m = 2;  // (the wave-per-block form has two and four tiles)
  const dim3 grid(aa_grid(a.tiles / m, a.evals * a.K));
  if (m == 4) {
    if (a.gradient)
      hipLaunchKernelGGL((aa_post_kernel<4, true>), grid, dim3(64), aa_lds_pad(), s, a);
    else
      hipLaunchKernelGGL((aa_post_kernel<4, false>), grid, dim3(64), aa_lds_pad(), s, a);
    return;
  }
  if (a.gradient)
    hipLaunchKernelGGL((aa_post_kernel<2, true>), grid, dim3(64), aa_lds_pad(), s, a);
  else
    hipLaunchKernelGGL((aa_post_kernel<2, false>), grid, dim3(64), aa_lds_pad(), s, a);
}
void launch_aa_root(const AaWalkArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(aa_root_kernel, dim3(aa_ll_blocks(a.P), a.evals), dim3(256), 0, s, a);
}
static bool aa_pre_wg() {
  static const bool wg = !(getenv("MI_PHYLO_AA_PRE") && std::string(getenv("MI_PHYLO_AA_PRE")) == "wave");
  return wg;
}
// entries of the LDS ring of parked pre-order vectors (aa_pre_wg_kernel): one -- 20 KB per
// workgroup beside 49 KB of operand buffers, two workgroups per CU by registers either way;
// MI_PHYLO_AA_PRE_RING=0|1|2 overrides (2 costs a workgroup per CU)
int aa_pre_ring_entries() {
  static const int forced = getenv("MI_PHYLO_AA_PRE_RING") ? atoi(getenv("MI_PHYLO_AA_PRE_RING")) : -1;
  if (!aa_pre_wg()) return 0;
  return forced == 0 || forced == 1 || forced == 2 ? forced : 1;
}
void launch_aa_pre(const AaWalkArgs& a, hipStream_t s) {
  if (aa_pre_wg()) {
    const int blocks = a.tiles / kAaPreTiles;
    const dim3 grid(aa_grid((blocks + kPreWaves - 1) / kPreWaves, a.evals * a.K)), block(kPreThreads);
    AaWalkArgs b = a;
    b.pre_ring_slots = aa_pre_ring_entries();
    const size_t lds = sizeof(double) * (2 * 2 * kPreOps + kAaPack) + sizeof(SchedEntry) * kSchedWindow +
                       sizeof(double) * (size_t)kPreWaves * b.pre_ring_slots * kAaPreTiles * kAaTileDoubles + aa_lds_pad();
    allow_large_lds(reinterpret_cast<const void*>(aa_pre_wg_kernel<kAaPreTiles>), lds);
    hipLaunchKernelGGL((aa_pre_wg_kernel<kAaPreTiles>), grid, block, lds, s, b);
    return;
  }
  const dim3 grid(aa_grid(a.tiles / kAaPreTiles, a.evals * a.K));
  hipLaunchKernelGGL((aa_pre_kernel<kAaPreTiles>), grid, dim3(64), aa_lds_pad(), s, a);
}
void launch_aa_reduce(const AaWalkArgs& a, hipStream_t s) {
  const int gx = a.gradient ? 1 + (a.N + 63) / 64 : 1;
  hipLaunchKernelGGL(aa_reduce_kernel, dim3(gx, a.evals), dim3(256), 0, s, a, aa_ll_blocks(a.P));
}
// (the names rocprofv3 prints: the workgroup forms unless MI_PHYLO_AA_POST / _PRE = wave)
const char* aa_post_kernel_name() { return aa_post_wg() ? "aa_post_wg_kernel" : "aa_post_kernel"; }
const char* aa_pre_kernel_name() { return aa_pre_wg() ? "aa_pre_wg_kernel" : "aa_pre_kernel"; }

}  // namespace miphylo
