#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4v __attribute__((ext_vector_type(4)));

__global__ void layout_probe(const double* A, const double* B, double* D) {
  // one wave: lane l supplies a = A[l], b = B[l]
  const int l = threadIdx.x;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
  D[l] = d;
}

__global__ void rate_probe_4x4(double* out, int iters) {
  const int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-3, b = 0.5 + l * 1e-4;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  long t0 = clock64();
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
  long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0) / (8.0 * iters);
}

__global__ void rate_probe_16x16(double* out, int iters) {
  const int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-3, b = 0.5 + l * 1e-4;
  double4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0.x + c1.y + c2.z + c3.w;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0) / (4.0 * iters);
}

__global__ void rate_probe_fma(double* out, int iters) {
  const int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-9, b = 1e-9;
  double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
  long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    c0 = fma(a, c0, b); c1 = fma(a, c1, b); c2 = fma(a, c2, b); c3 = fma(a, c3, b);
    c4 = fma(a, c4, b); c5 = fma(a, c5, b); c6 = fma(a, c6, b); c7 = fma(a, c7, b);
  }
  long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0) / (8.0 * iters);
}

int main() {
  double *dA, *dB, *dD, *dout;
  hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dD, 64 * 8);
  hipMalloc(&dout, ((1 << 20) + 8) * 8);
  // layout: for every (lane_a) set A one-hot at lane_a and B all distinct primes-ish; read D
  std::vector<double> A(64), B(64), D(64);
  // experiment 1: A[l] = 1 only at lane la, B[l] = 1 only at lane lb -> which D lane becomes 1?
  printf("A-lane B-lane -> D-lane (nonzero outputs)\n");
  for (int la = 0; la < 64; la += 1) {
    for (int lb = 0; lb < 64; lb += 1) {
      for (int i = 0; i < 64; i++) { A[i] = (i == la); B[i] = (i == lb); }
      hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(layout_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
      hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
      for (int i = 0; i < 64; i++)
        if (D[i] != 0 && la < 20 && (lb < 20)) printf("  a%2d b%2d -> d%2d\n", la, lb, i);
    }
  }
  // summary of structure: for each D lane, list (la, lb) pairs contributing
  // rates: one wave per SIMD, 4 waves per block, many blocks
  const int iters = 20000;
  for (int waves = 1; waves <= 2; waves++) {
    hipLaunchKernelGGL(rate_probe_4x4, dim3(256), dim3(256 * waves), 0, 0, dout, iters);
    double cyc; hipMemcpy(&cyc, dout + (1 << 20), 8, hipMemcpyDeviceToHost);
    printf("mfma_f64_4x4x4_4b : %.2f cycles/instr/wave (%d waves per SIMD) -> %.1f MAC/clk/SIMD\n", cyc, waves, 256.0 / cyc * waves);
    hipLaunchKernelGGL(rate_probe_16x16, dim3(256), dim3(256 * waves), 0, 0, dout, iters);
    hipMemcpy(&cyc, dout + (1 << 20), 8, hipMemcpyDeviceToHost);
    printf("mfma_f64_16x16x4  : %.2f cycles/instr/wave (%d waves per SIMD) -> %.1f MAC/clk/SIMD\n", cyc, waves, 1024.0 / cyc * waves);
    hipLaunchKernelGGL(rate_probe_fma, dim3(256), dim3(256 * waves), 0, 0, dout, iters);
    hipMemcpy(&cyc, dout + (1 << 20), 8, hipMemcpyDeviceToHost);
    printf("v_fma_f64         : %.2f cycles/instr/wave (%d waves per SIMD) -> %.1f MAC/clk/SIMD\n", cyc, waves, 64.0 / cyc * waves);
  }
  return 0;
}
