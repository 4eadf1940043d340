// Device-side helpers shared by the kernel translation units of libmi_phylo.so
// (4-state vectors, wave reductions, the XCD-aware workgroup mapping, error status) and
// the host-side large-LDS opt-in.  Everything here is inline: each .hip file gets its own
// copy of the device code.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>

#include "mi_phylo_kernels.h"

namespace miphylo {
namespace dev {

struct D4 {
  double x0, x1, x2, x3;
};

// Read-only data written by an EARLIER kernel (transition matrices, schedules,
// models) is addressed through the constant address space: loads with a
// wave-uniform address then become scalar loads (s_load_*) into SGPRs no matter
// what else the kernel stores.
typedef const double __attribute__((address_space(4))) * cdouble_ptr;
typedef const int __attribute__((address_space(4))) * cint_ptr;
__device__ __forceinline__ cdouble_ptr as_const(const double* p) {
  return (cdouble_ptr)(uintptr_t)p;
}
__device__ __forceinline__ cint_ptr as_const(const int* p) { return (cint_ptr)(uintptr_t)p; }

__device__ __forceinline__ D4 mul4(D4 a, D4 b) {
  return {a.x0 * b.x0, a.x1 * b.x1, a.x2 * b.x2, a.x3 * b.x3};
}

// a_i = sum_j M[i][j] L_j   (M row-major, wave-uniform)
template <typename MP>
__device__ __forceinline__ D4 matvec(MP M, D4 L) {
  D4 a;
  a.x0 = M[0] * L.x0 + M[1] * L.x1 + M[2] * L.x2 + M[3] * L.x3;
  a.x1 = M[4] * L.x0 + M[5] * L.x1 + M[6] * L.x2 + M[7] * L.x3;
  a.x2 = M[8] * L.x0 + M[9] * L.x1 + M[10] * L.x2 + M[11] * L.x3;
  a.x3 = M[12] * L.x0 + M[13] * L.x1 + M[14] * L.x2 + M[15] * L.x3;
  return a;
}

// q_j = sum_i M[i][j] u_i
template <typename MP>
__device__ __forceinline__ D4 matTvec(MP M, D4 u) {
  D4 q;
  q.x0 = M[0] * u.x0 + M[4] * u.x1 + M[8] * u.x2 + M[12] * u.x3;
  q.x1 = M[1] * u.x0 + M[5] * u.x1 + M[9] * u.x2 + M[13] * u.x3;
  q.x2 = M[2] * u.x0 + M[6] * u.x1 + M[10] * u.x2 + M[14] * u.x3;
  q.x3 = M[3] * u.x0 + M[7] * u.x1 + M[11] * u.x2 + M[15] * u.x3;
  return q;
}

__device__ __forceinline__ double dot4(D4 a, D4 b) {
  return a.x0 * b.x0 + a.x1 * b.x1 + a.x2 * b.x2 + a.x3 * b.x3;
}

// Compact tip state -> partial vector: one-hot, or all ones for a gap
// (site_pattern.cpp:117-131; BEAGLE treats compact states >= s the same way).
__device__ __forceinline__ D4 tip_vector(int st) {
  return {(st == 0 || st > 3) ? 1.0 : 0.0, (st == 1 || st > 3) ? 1.0 : 0.0,
          (st == 2 || st > 3) ? 1.0 : 0.0, (st == 3 || st > 3) ? 1.0 : 0.0};
}

// P * tip_vector(st) without arithmetic: column st of P, or 1 (rows of P sum to 1).
// Written as a chain of selects so that it compiles to v_cndmask, never to branches.
__device__ __forceinline__ double select_state(int st, double m0, double m1, double m2,
                                               double m3, double other) {
  double r = other;
  r = st == 3 ? m3 : r;
  r = st == 2 ? m2 : r;
  r = st == 1 ? m1 : r;
  r = st == 0 ? m0 : r;
  return r;
}
template <typename MP>
__device__ __forceinline__ D4 tip_column(MP M, int st) {
  return {select_state(st, M[0], M[1], M[2], M[3], 1.0),
          select_state(st, M[4], M[5], M[6], M[7], 1.0),
          select_state(st, M[8], M[9], M[10], M[11], 1.0),
          select_state(st, M[12], M[13], M[14], M[15], 1.0)};
}

__device__ __forceinline__ D4 load4(const double* __restrict__ ptr) {
  const double2 lo = *reinterpret_cast<const double2*>(ptr);
  const double2 hi = *reinterpret_cast<const double2*>(ptr + 2);
  return {lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ void store4(double* ptr, D4 v) {
  *reinterpret_cast<double2*>(ptr) = double2{v.x0, v.x1};
  *reinterpret_cast<double2*>(ptr + 2) = double2{v.x2, v.x3};
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Power-of-two rescaling: exact, so rescaled and unscaled results agree to the
// last bit wherever the unscaled ones are representable.
__device__ __forceinline__ int max_exponent(double m) { return m > 0.0 ? ilogb(m) : 0; }
__device__ __forceinline__ D4 scale4(D4 a, int e) {
  return {ldexp(a.x0, e), ldexp(a.x1, e), ldexp(a.x2, e), ldexp(a.x3, e)};
}
__device__ __forceinline__ double max4(D4 a) {
  return fmax(fmax(a.x0, a.x1), fmax(a.x2, a.x3));
}

// Workgroups are dealt to the 8 XCDs (each with a private 4 MiB L2) round-robin by
// linear id: ids b and b+8 share an XCD.  The walk kernels launch (tiles x evaluations)
// workgroups; this bijection hands each XCD whole evaluations, so that an evaluation's
// transition matrices and schedule are fetched into ONE L2 instead of all eight.
// Placement is a speed matter only.
struct TileEval {
  int tile, eval;
};
// (id: linear workgroup id, a multiple of 8 apart from the XCD it runs on; tiles x count
// workgroups)
__device__ __forceinline__ TileEval xcd_map(int id, int tiles, int count) {
  const int full = count & ~7;  // evaluations in complete groups of 8
  TileEval te;
  if (id < full * tiles) {
    const int s = id >> 3;  // s-th workgroup of its XCD
    te.eval = (s / tiles) * 8 + (id & 7);
    te.tile = s % tiles;
  } else {
    te.eval = id / tiles;
    te.tile = id - te.eval * tiles;
  }
  return te;
}
__device__ __forceinline__ TileEval xcd_tile_eval() {
  return xcd_map(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y);
}

// Divided difference of exp(lambda tau) for the analytic substitution gradient:
// Phi[row][col] = tau e^{l_col tau} expm1(x) / x, x = (l_row - l_col) tau (stable for close and
// equal eigenvalues).  Explicitly rounded operations: both transition kernels (node order,
// macro order) must produce the same bits, whatever the compiler would contract around them.
__device__ __forceinline__ double phi_divided_difference(double lam_row, double lam_col, double tau) {
  const double x = __dmul_rn(__dsub_rn(lam_row, lam_col), tau);
  const double r = fabs(x) < 1e-5
                       ? __dadd_rn(__dadd_rn(1.0, __dmul_rn(0.5, x)), __dmul_rn(__dmul_rn(x, x), 1.0 / 6.0))
                       : __ddiv_rn(expm1(x), x);
  return __dmul_rn(__dmul_rn(tau, exp(__dmul_rn(lam_col, tau))), r);
}

__device__ __forceinline__ void set_status(int32_t* status, int code, int tree) {
  if (atomicCAS(status, 0, code) == 0) status[1] = tree;
}

}  // namespace dev

// Kernels that want more than 64 KiB of dynamic LDS have to opt in, per function AND per
// device (a process may drive several GPUs): remembered per (function, device).
inline void allow_large_lds(const void* func, size_t lds) {
  if (lds <= 64 * 1024) return;
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> configured;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = configured[{func, dev}];
  if (lds > have) {
    (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    have = lds;
  }
}

}  // namespace miphylo
