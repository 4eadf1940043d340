// Micro-benchmark (gfx950), second form: a wave that issues one v_mfma_f64_16x16x4 followed by
// OWN filler (k x v_add_u32, or s_nop), beside B waves on the same SIMD that run v_add_u32 /
// s_mul_i32 / ds_read_b64 loops.  Prints, per configuration, the clocks per MFMA of wave 0 and
// the clocks per instruction of a B wave, alone and together.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/coissue2.hip -o tools/micro/coissue2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int kIters = 4000;

__device__ __forceinline__ unsigned long long clk() {
  unsigned long long t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

#define ADD2 "v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %2\n"
#define ADD4 ADD2 ADD2
#define MF(c) "v_mfma_f64_16x16x4_f64 " c ", %4, %5, " c "\n"

// FILL: 0 none, 1..4: 4/8/12/16 own v_add_u32 after each MFMA, 5: s_nop 15, 6: s_nop 7, 7: two chains no filler,
// 8: 4 own v_mul_f64 ... (kept small)
template <int FILL>
__device__ __forceinline__ void mfma_role(double* sink, int lane) {
  double4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  double a = lane, b = 1.0 / (lane + 1);
  unsigned w0 = lane, w1 = 17;
  for (int i = 0; i < kIters; i++) {
    if (FILL == 0) asm volatile(MF("%0") MF("%0") : "+v"(c0), "+v"(c1), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    if (FILL == 1) asm volatile(MF("%0") ADD4 MF("%0") ADD4 : "+v"(c0), "+v"(c1), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    if (FILL == 2) asm volatile(MF("%0") ADD4 ADD4 MF("%0") ADD4 ADD4 : "+v"(c0), "+v"(c1), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    if (FILL == 3) asm volatile(MF("%0") ADD4 ADD4 ADD4 MF("%0") ADD4 ADD4 ADD4 : "+v"(c0), "+v"(c1), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    if (FILL == 4) asm volatile(MF("%0") ADD4 ADD4 ADD4 ADD4 MF("%0") ADD4 ADD4 ADD4 ADD4 : "+v"(c0), "+v"(c1), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    if (FILL == 5) asm volatile(MF("%0") "s_nop 15\n" MF("%0") "s_nop 15\n" : "+v"(c0), "+v"(c1), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    if (FILL == 6) asm volatile(MF("%0") "s_nop 7\n" MF("%0") "s_nop 7\n" : "+v"(c0), "+v"(c1), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
    if (FILL == 7) asm volatile(MF("%0") MF("%1") : "+v"(c0), "+v"(c1), "+v"(w0), "+v"(w1) : "v"(a), "v"(b));
  }
  sink[lane] = c0.x + c1.y + w0 + w1;
}

template <int KIND>
__device__ __forceinline__ void other_role(double* sink, int lane) {
  unsigned u0 = lane, u1 = lane * 3, u2 = lane * 5, u3 = lane * 7;
  double x0 = lane, x1 = 1, x2 = 2, x3 = 3;
  for (int i = 0; i < kIters; i++) {
    if (KIND == 1)
      asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                   "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
    if (KIND == 2) {
      int s0 = i, s1 = i + 1;
      asm volatile("s_mul_i32 %0, %0, %1\n s_mul_i32 %1, %1, %0\n s_mul_i32 %0, %0, %1\n s_mul_i32 %1, %1, %0\n"
                   "s_mul_i32 %0, %0, %1\n s_mul_i32 %1, %1, %0\n s_mul_i32 %0, %0, %1\n s_mul_i32 %1, %1, %0\n"
                   : "+s"(s0), "+s"(s1));
      u0 += s0;
    }
    if (KIND == 3) {
      unsigned ad = lane * 8;
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n"
                   "ds_read_b64 %0, %4 offset:2048\n ds_read_b64 %1, %4 offset:2560\n ds_read_b64 %2, %4 offset:3072\n ds_read_b64 %3, %4 offset:3584\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(ad) : "memory");
    }
  }
  sink[lane] = x0 + x1 + x2 + x3 + u0 + u1 + u2 + u3;
}

// waves 0-3: MFMA role (one per SIMD); waves 4 .. 4 + 4 nb - 1: B role (nb per SIMD)
template <int FILL, int KIND>
__global__ __launch_bounds__(1024) void k(double* sink, unsigned long long* clocks, int mode, int nb) {
  __shared__ double lds[16][512];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < 512; i += 64) lds[wave][i] = i;
  __syncthreads();
  const unsigned long long t0 = clk();
  if (wave < 4) {
    if (mode & 1) mfma_role<FILL>(sink + wave * 64, lane);
  } else if (wave < 4 + 4 * nb) {
    if (mode & 2) other_role<KIND>(sink + wave * 64, lane);
  }
  const unsigned long long t1 = clk();
  if (lane == 0 && blockIdx.x == 0) clocks[wave] = t1 - t0;
}

template <int FILL, int KIND>
void run(const char* name, double* sink, unsigned long long* clocks, int nb) {
  unsigned long long h[3][16];
  for (int mode = 1; mode <= 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL((k<FILL, KIND>), dim3(256), dim3(1024), 0, 0, sink, clocks, mode, nb);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h[mode - 1], clocks, sizeof(h[0]), hipMemcpyDeviceToHost);
  }
  printf("%-28s B waves/SIMD %d | alone: mfma %6.1f  B %5.1f | together: mfma %6.1f  B %5.1f\n", name, nb,
         h[0][0] / (2.0 * kIters), h[1][4] / (8.0 * kIters), h[2][0] / (2.0 * kIters), h[2][4] / (8.0 * kIters));
}

int main() {
  double* sink;
  unsigned long long* clocks;
  (void)hipMalloc(&sink, 1 << 20);
  (void)hipMalloc(&clocks, 128);
  printf("clocks per MFMA (wave 0) and per B instruction (wave 4)\n");
  run<0, 1>("1 chain | v_add", sink, clocks, 1);
  run<7, 1>("2 chains | v_add", sink, clocks, 1);
  run<1, 1>("mfma+4 own add | v_add", sink, clocks, 1);
  run<2, 1>("mfma+8 own add | v_add", sink, clocks, 1);
  run<3, 1>("mfma+12 own add | v_add", sink, clocks, 1);
  run<4, 1>("mfma+16 own add | v_add", sink, clocks, 1);
  run<5, 1>("mfma+s_nop15 | v_add", sink, clocks, 1);
  run<6, 1>("mfma+s_nop7 | v_add", sink, clocks, 1);
  run<0, 1>("1 chain | v_add", sink, clocks, 3);
  run<5, 1>("mfma+s_nop15 | v_add", sink, clocks, 3);
  run<0, 2>("1 chain | s_mul", sink, clocks, 1);
  run<5, 2>("mfma+s_nop15 | s_mul", sink, clocks, 1);
  run<3, 2>("mfma+12 own add | s_mul", sink, clocks, 1);
  run<0, 3>("1 chain | ds_read", sink, clocks, 1);
  run<5, 3>("mfma+s_nop15 | ds_read", sink, clocks, 1);
  run<3, 3>("mfma+12 own add | ds_read", sink, clocks, 1);
  return 0;
}
