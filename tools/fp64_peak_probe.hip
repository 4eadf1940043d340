// Saturating FP64 probe for MI355X (gfx950): device-wide WALL-CLOCK TFLOP/s of
//   v_fma_f64, v_mfma_f64_4x4x4_4b_f64, v_mfma_f64_16x16x4_f64 and the mixed
//   5 x 16x16x4 + 5 x 4x4x4 sequence the 20-state kernels issue per (child, tile),
// with 1, 2, 4, 8 waves per SIMD, independent accumulators, random (non-trivial) operands.
// Reconciles profiles/r01_mfma_f64_probe.txt (per-wave s_memtime cycles, which said
// 28-32 MAC/clk/SIMD at two waves per SIMD) with the 78.6 TFLOP/s data-sheet figure.
// Also reports the shader clock held under load (s_memtime / s_memrealtime).
//
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_peak_probe.hip -o /tmp/fp64_peak_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double double4v __attribute__((ext_vector_type(4)));

struct Stamp {
  long long cyc, real;
};

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(double* out, Stamp* stamps, int iters) {
  const int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-3 + blockIdx.x * 1e-6, b = 0.5 + l * 1e-4;
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  double sink = 0;
  if constexpr (KIND == 0) {  // vector FMA, 8 independent chains
    double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
    const double m = 1.0 + 1e-9 * l;
    for (int i = 0; i < iters; i++) {
      c0 = fma(m, c0, b); c1 = fma(m, c1, b); c2 = fma(m, c2, b); c3 = fma(m, c3, b);
      c4 = fma(m, c4, b); c5 = fma(m, c5, b); c6 = fma(m, c6, b); c7 = fma(m, c7, b);
    }
    sink = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  } else if constexpr (KIND == 1) {  // 4x4x4, four blocks: 256 MAC per instruction
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    for (int i = 0; i < iters; i++) {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
      c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
      c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
    }
    sink = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  } else if constexpr (KIND == 2) {  // 16x16x4: 1024 MAC per instruction
    double4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    sink = c0.x + c1.y + c2.z + c3.w;
  } else if constexpr (KIND == 4) {
    // do f64 matrix and f64 vector instructions share a datapath?  Even waves issue only
    // 4x4x4 MFMAs, odd waves only v_fma_f64 (one of each per SIMD at 2 waves/SIMD).
    if ((threadIdx.x >> 6) & 1) {
      double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
      const double m = 1.0 + 1e-9 * l;
      for (int i = 0; i < iters; i++) {
        c0 = fma(m, c0, b); c1 = fma(m, c1, b); c2 = fma(m, c2, b); c3 = fma(m, c3, b);
        c4 = fma(m, c4, b); c5 = fma(m, c5, b); c6 = fma(m, c6, b); c7 = fma(m, c7, b);
      }
      sink = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    } else {
      double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
      for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
      }
      sink = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    }
  } else if constexpr (KIND == 5) {
    // same question for f64 MFMA beside 32-bit integer / f32 vector work
    if ((threadIdx.x >> 6) & 1) {
      float c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
      const float m = 1.0f + 1e-6f * l, bb = (float)b;
      for (int i = 0; i < iters; i++) {
        c0 = fmaf(m, c0, bb); c1 = fmaf(m, c1, bb); c2 = fmaf(m, c2, bb); c3 = fmaf(m, c3, bb);
        c4 = fmaf(m, c4, bb); c5 = fmaf(m, c5, bb); c6 = fmaf(m, c6, bb); c7 = fmaf(m, c7, bb);
      }
      sink = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    } else {
      double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
      for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
      }
      sink = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    }
  } else if constexpr (KIND == 6) {
    // 16x16x4 with ONE accumulator chain (dependent issue) -- latency of the instruction
    double4v c0 = {0, 0, 0, 0};
    for (int i = 0; i < iters; i++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    }
    sink = c0.x;
  } else if constexpr (KIND == 7) {
    // 16x16x4 alternating with 4x4x4, ONE chain each (what a one-tile-per-wave walk issues)
    double4v c0 = {0, 0, 0, 0};
    double d0 = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int t = 0; t < 4; t++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
      }
    }
    sink = c0.x + d0;
  } else if constexpr (KIND == 8) {
    // even waves only 16x16x4, odd waves only 4x4x4: the arbiter interleaves the two kinds
    if ((threadIdx.x >> 6) & 1) {
      double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
      for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
      }
      sink = c0 + c1 + c2 + c3;
    } else {
      double4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      }
      sink = c0.x + c1.y + c2.z + c3.w;
    }
  } else if constexpr (KIND == 9 || KIND == 10) {
    // round 3: does the INTEGER bookkeeping of the walk kernels co-issue
    // with the f64 matrix pipe?  Even waves issue only 4x4x4 MFMAs; odd waves only
    // v_mad_u32_u24 / v_bfe_u32 (KIND 9) or only scalar ALU instructions (KIND 10).
    if ((threadIdx.x >> 6) & 1) {
      if constexpr (KIND == 9) {
        unsigned c0 = l, c1 = l + 1, c2 = l + 2, c3 = l + 3, c4 = l + 4, c5 = l + 5, c6 = l + 6, c7 = l + 7;
        const unsigned m = 3 + (l & 1);
        for (int i = 0; i < iters; i++) {
          c0 = __umul24(c0, m) + c1; c1 = __builtin_amdgcn_ubfe(c1, 1u, 23u) + c0;
          c2 = __umul24(c2, m) + c3; c3 = __builtin_amdgcn_ubfe(c3, 1u, 23u) + c2;
          c4 = __umul24(c4, m) + c5; c5 = __builtin_amdgcn_ubfe(c5, 1u, 23u) + c4;
          c6 = __umul24(c6, m) + c7; c7 = __builtin_amdgcn_ubfe(c7, 1u, 23u) + c6;
        }
        sink = (double)(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7);
      } else {
        unsigned s0 = __builtin_amdgcn_readfirstlane(blockIdx.x), s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
        for (int i = 0; i < iters; i++) {
          asm volatile("s_mul_i32 %0, %0, 3\n s_add_i32 %1, %1, %0\n s_lshr_b32 %2, %2, 1\n"
                       "s_add_i32 %3, %3, %2\n s_mul_i32 %0, %0, 5\n s_add_i32 %1, %1, %0\n"
                       "s_xor_b32 %2, %2, %1\n s_add_i32 %3, %3, %2"
                       : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                       :
                       : "scc");
        }
        sink = (double)(s0 + s1 + s2 + s3);
      }
    } else {
      double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
      for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
      }
      sink = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    }
  } else if constexpr (KIND == 11 || KIND == 12 || KIND == 13) {
    // the same question inside ONE wave (what a walk kernel at two waves per SIMD mostly
    // is): every MFMA followed by NV independent vector instructions -- 32-bit integer
    // (KIND 11: 2 per MFMA, KIND 12: 4 per MFMA) or f64 multiplies (KIND 13: 2 per MFMA).
    // If they hide in the 16 cycles the matrix instruction occupies its pipe, the time
    // equals the pure-MFMA row's.
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    unsigned u0 = l, u1 = l + 1, u2 = l + 2, u3 = l + 3;
    double f0 = 1.0 + 1e-9 * l, f1 = f0, f2 = f0, f3 = f0;
    const double fm = 1.0 + 1e-12 * l;
    const unsigned m = 3 + (l & 1);
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int t = 0; t < 2; t++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        if constexpr (KIND == 13) { f0 *= fm; f1 *= fm; } else { u0 = __umul24(u0, m) + u1; u1 = __builtin_amdgcn_ubfe(u1, 1u, 23u) + u0; }
        if constexpr (KIND == 12) { u2 = __umul24(u2, m) + u3; u3 = __builtin_amdgcn_ubfe(u3, 1u, 23u) + u2; }
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        if constexpr (KIND == 13) { f2 *= fm; f3 *= fm; } else { u2 = __umul24(u2, m) + u3; u3 = __builtin_amdgcn_ubfe(u3, 1u, 23u) + u2; }
        if constexpr (KIND == 12) { u0 = __umul24(u0, m) + u1; u1 = __builtin_amdgcn_ubfe(u1, 1u, 23u) + u0; }
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        if constexpr (KIND == 13) { f0 *= fm; f1 *= fm; } else { u0 = __umul24(u0, m) + u1; u1 = __builtin_amdgcn_ubfe(u1, 1u, 23u) + u0; }
        if constexpr (KIND == 12) { u2 = __umul24(u2, m) + u3; u3 = __builtin_amdgcn_ubfe(u3, 1u, 23u) + u2; }
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        if constexpr (KIND == 13) { f2 *= fm; f3 *= fm; } else { u2 = __umul24(u2, m) + u3; u3 = __builtin_amdgcn_ubfe(u3, 1u, 23u) + u2; }
        if constexpr (KIND == 12) { u0 = __umul24(u0, m) + u1; u1 = __builtin_amdgcn_ubfe(u1, 1u, 23u) + u0; }
      }
    }
    sink = c0 + c1 + c2 + c3 + (double)(u0 + u1 + u2 + u3) + f0 + f1 + f2 + f3;
  } else {  // the 20-state inner sequence: 5 k-steps of (16 rows) + 5 of (4 rows), two tiles
    double4v c0 = {0, 0, 0, 0}, c1 = c0;
    double d0 = 0, d1 = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int t = 0; t < 5; t++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
      }
    }
    sink = c0.x + c1.y + d0 + d1;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sink;
  if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp{t1 - t0, r1 - r0};
}

// C/D, A, B lane maps of v_mfma_f64_16x16x4_f64: one-hot A at (lane la), one-hot B at lane lb
__global__ void layout16(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  double4v c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l], B[l], c, 0, 0, 0);
  D[l * 4 + 0] = c.x; D[l * 4 + 1] = c.y; D[l * 4 + 2] = c.z; D[l * 4 + 3] = c.w;
}

template <int KIND>
static void run(const char* name, double mac_per_instr, int instr_per_iter, double* out,
                Stamp* stamps, int iters) {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wps : {1, 2, 4, 8}) {
    // wps waves per SIMD = wps blocks of 256 threads per CU, all co-resident (few registers)
    const int blocks = cus * wps;
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> st(blocks);
    hipMemcpy(st.data(), stamps, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (const Stamp& s : st) { cyc += s.cyc; real += s.real; }
    cyc /= blocks; real /= blocks;
    const double instr = (double)blocks * 4 * iters * instr_per_iter;  // per wave -> total
    const double tflops = 2.0 * instr * mac_per_instr / (ms * 1e-3) / 1e12;
    const double ghz = cyc / (real / 100e6) / 1e9;  // s_memrealtime ticks at 100 MHz
    printf("%-22s %d waves/SIMD: %8.3f ms  %7.2f TFLOP/s wall  | per wave %.2f cyc/instr, "
           "%.1f MAC/clk/SIMD, clock %.2f GHz\n",
           name, wps, ms, tflops, cyc / ((double)iters * instr_per_iter),
           mac_per_instr * wps / (cyc / ((double)iters * instr_per_iter)), ghz);
  }
}

int main() {
  double *dA, *dB, *dD, *out;
  Stamp* stamps;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMalloc(&out, sizeof(double) * 256 * 8 * 256);
  hipMalloc(&stamps, sizeof(Stamp) * 256 * 8);
  // layout of 16x16x4: expected (guide): A[i][k] at lane 16k+i, B[k][j] at lane 16k+j,
  // D[row][col]: col = lane&15, row = (lane>>4) + 4*reg
  std::vector<double> A(64), B(64), D(256);
  int bad = 0;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      for (int i = 0; i < 64; i++) { A[i] = i == la; B[i] = i == lb; }
      hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(layout16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
      hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
      const int i = la & 15, ka = la >> 4, j = lb & 15, kb = lb >> 4;
      for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
          const int row = (l >> 4) + 4 * r, col = l & 15;
          const double want = (ka == kb && row == i && col == j) ? 1.0 : 0.0;
          if (D[l * 4 + r] != want) bad++;
        }
    }
  printf("v_mfma_f64_16x16x4 layout (A[i][k]@16k+i, B[k][j]@16k+j, D[(l>>4)+4r][l&15]): %s\n",
         bad ? "MISMATCH" : "confirmed");
  const int iters = 40000;
  run<0>("v_fma_f64", 64, 8, out, stamps, iters);
  run<1>("mfma_f64_4x4x4_4b", 256, 8, out, stamps, iters);
  run<2>("mfma_f64_16x16x4", 1024, 4, out, stamps, iters / 2);
  // mixed: 10 x 16x16x4 + 10 x 4x4x4 per iteration = 12800 MAC over 20 instructions
  run<3>("20-state mix (5+5)x2", 640, 20, out, stamps, iters / 8);
  // co-issue: half the waves MFMA f64 (256 MAC/instr), half vector (64 MAC/instr): the printed
  // TFLOP/s uses the average 160 MAC per instruction; compare the TIME with the pure runs above
  // at equal iterations (pure MFMA at 1 wave/SIMD + pure FMA at 1 wave/SIMD side by side)
  run<6>("16x16x4, one chain", 1024, 4, out, stamps, iters / 4);
  run<7>("16x16x4+4x4x4, one chain ea", 640, 8, out, stamps, iters / 4);
  run<8>("16x16x4 waves || 4x4x4 waves", 640, 4, out, stamps, iters / 2);
  run<4>("mfma_f64 || v_fma_f64", 160, 8, out, stamps, iters);
  run<5>("mfma_f64 || v_fma_f32", 160, 8, out, stamps, iters);
  // round 3: integer VALU / scalar ALU beside the matrix pipe (TFLOP/s column: MFMA flops of
  // the even waves only, i.e. 128 MAC per instruction averaged over all waves; compare the
  // TIME with mfma_f64_4x4x4_4b at half the waves per SIMD)
  run<9>("mfma_f64 || int VALU waves", 128, 8, out, stamps, iters);
  run<10>("mfma_f64 || SALU waves", 128, 8, out, stamps, iters);
  // one wave: 8 MFMA + 2 (or 4) vector instructions each; MFMA flops only (256 MAC x 8 per iteration)
  run<11>("1 wave: mfma + 2 int VALU", 256, 8, out, stamps, iters);
  run<12>("1 wave: mfma + 4 int VALU", 256, 8, out, stamps, iters);
  run<13>("1 wave: mfma + 2 v_mul_f64", 256, 8, out, stamps, iters);
  return 0;
}
