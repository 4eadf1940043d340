// Micro-benchmark (gfx950), third form: v_mfma_f64_4x4x4 chains (the nucleotide walk's matrix
// instruction).  Wave A: N dependent chains interleaved, optionally with own VALU filler between
// the instructions; wave B on the same SIMD: a v_add_u32 loop.  Prints clocks per MFMA for A
// alone and beside B, and B's clocks per instruction WHILE A runs (B loops until A is done).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/coissue3.hip -o tools/micro/coissue3
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 4000;

__device__ __forceinline__ unsigned long long clk() {
  unsigned long long t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

#define MF(c) "v_mfma_f64_4x4x4_4b_f64 " c ", %6, %7, " c "\n"
#define AD "v_add_u32 %4, %4, %5\n"
#define AD2 AD AD
#define AD4 AD2 AD2

template <int V>
__device__ __forceinline__ void a_role(double* sink, int lane) {
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  double a = lane, b = 1.0 / (lane + 1);
  unsigned w0 = lane, w1 = 17;
  for (int i = 0; i < kIters; i++) {
    // 8 MFMAs per iteration
    if (V == 1) asm volatile(MF("%0") MF("%0") MF("%0") MF("%0") MF("%0") MF("%0") MF("%0") MF("%0") : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    if (V == 2) asm volatile(MF("%0") MF("%1") MF("%0") MF("%1") MF("%0") MF("%1") MF("%0") MF("%1") : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    if (V == 4) asm volatile(MF("%0") MF("%1") MF("%2") MF("%3") MF("%0") MF("%1") MF("%2") MF("%3") : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    // one chain with own VALU between dependent MFMAs: 1, 2, 4 adds
    if (V == 11) asm volatile(MF("%0") AD MF("%0") AD MF("%0") AD MF("%0") AD MF("%0") AD MF("%0") AD MF("%0") AD MF("%0") AD : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    if (V == 12) asm volatile(MF("%0") AD2 MF("%0") AD2 MF("%0") AD2 MF("%0") AD2 MF("%0") AD2 MF("%0") AD2 MF("%0") AD2 MF("%0") AD2 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    if (V == 14) asm volatile(MF("%0") AD4 MF("%0") AD4 MF("%0") AD4 MF("%0") AD4 MF("%0") AD4 MF("%0") AD4 MF("%0") AD4 MF("%0") AD4 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    // one chain with s_nop 1 / s_nop 3 between (the wave does not ask for issue)
    if (V == 21) asm volatile(MF("%0") "s_nop 1\n" MF("%0") "s_nop 1\n" MF("%0") "s_nop 1\n" MF("%0") "s_nop 1\n" MF("%0") "s_nop 1\n" MF("%0") "s_nop 1\n" MF("%0") "s_nop 1\n" MF("%0") "s_nop 1\n" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    if (V == 23) asm volatile(MF("%0") "s_nop 3\n" MF("%0") "s_nop 3\n" MF("%0") "s_nop 3\n" MF("%0") "s_nop 3\n" MF("%0") "s_nop 3\n" MF("%0") "s_nop 3\n" MF("%0") "s_nop 3\n" MF("%0") "s_nop 3\n" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
  }
  sink[lane] = c0 + c1 + c2 + c3 + w0 + w1;
}

// B: v_add_u32 groups of 8 until the flag in LDS says A is done; counts its instructions
__device__ __forceinline__ unsigned b_role(double* sink, int lane, volatile int* done) {
  unsigned u0 = lane, u1 = lane * 3, u2 = lane * 5, u3 = lane * 7, n = 0;
  while (true) {
    asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                 "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                 "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                 "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
    n += 16;
    if ((n & 255) == 0 && *done) break;
  }
  sink[lane] = u0 + u1 + u2 + u3;
  return n;
}

template <int V>
__global__ __launch_bounds__(512) void k(double* sink, unsigned long long* out, int withB) {
  __shared__ int done[8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < 8) done[threadIdx.x] = 0;
  __syncthreads();
  if (wave < 4) {
    const unsigned long long t0 = clk();
    a_role<V>(sink + wave * 64, lane);
    const unsigned long long t1 = clk();
    if (lane == 0) done[wave] = 1;  // B wave (wave + 4) shares this wave's SIMD
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
  } else if (withB) {
    const unsigned long long t0 = clk();
    const unsigned n = b_role(sink + wave * 64, lane, done + (wave - 4));
    const unsigned long long t1 = clk();
    if (lane == 0 && blockIdx.x == 0) { out[wave] = t1 - t0; out[8 + wave] = n; }
  }
}

template <int V>
void run(const char* name, double* sink, unsigned long long* out) {
  unsigned long long h[2][16];
  for (int withB = 0; withB < 2; withB++) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL((k<V>), dim3(256), dim3(512), 0, 0, sink, out, withB);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h[withB], out, sizeof(h[0]), hipMemcpyDeviceToHost);
  }
  const double per = 8.0 * kIters;
  printf("%-26s A alone %6.1f clocks/MFMA | beside B: A %6.1f clocks/MFMA, B %6.1f v_add per A-MFMA (%5.1f clocks per v_add)\n", name,
         h[0][0] / per, h[1][0] / per, h[1][12] / per, (double)h[1][4] / (double)h[1][12]);
}

int main() {
  double* sink;
  unsigned long long* out;
  (void)hipMalloc(&sink, 1 << 20);
  (void)hipMalloc(&out, 256);
  run<1>("1 chain", sink, out);
  run<2>("2 chains", sink, out);
  run<4>("4 chains", sink, out);
  run<11>("1 chain + 1 own v_add", sink, out);
  run<12>("1 chain + 2 own v_add", sink, out);
  run<14>("1 chain + 4 own v_add", sink, out);
  run<21>("1 chain + s_nop 1", sink, out);
  run<23>("1 chain + s_nop 3", sink, out);
  return 0;
}
