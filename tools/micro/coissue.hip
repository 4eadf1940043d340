// Micro-benchmark (gfx950): what a SIMD can issue beside another wave's FP64 MFMAs.
// One workgroup of 8 waves per CU; waves 0-3 (one per SIMD) run a chain of
// v_mfma_f64_16x16x4, waves 4-7 (the same four SIMDs) run a loop of one instruction kind.
// Prints shader clocks per instruction for each role alone and together.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/coissue.hip -o tools/micro/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kIters = 2000;

template <int KIND>
__device__ __forceinline__ void other_loop(double* sink, int lane, double* lds, const double* uni) {
  double x0 = lane * 1.0001, x1 = lane + 2.0, x2 = 3.0 + lane, x3 = 4.0 - lane;
  unsigned u0 = lane, u1 = lane * 3, u2 = lane * 5, u3 = lane * 7;
  for (int i = 0; i < kIters; i++) {
    if (KIND == 1) {  // 8 x v_add_u32 (independent pairs)
      asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                   "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
    } else if (KIND == 2) {  // 8 x v_mul_f64
      asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0\n"
                   "v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    } else if (KIND == 3) {  // 8 x v_mov_b64
      asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n"
                   "v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    } else if (KIND == 4) {  // 8 x v_ldexp_f64
      asm volatile("v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %4\n v_ldexp_f64 %2, %2, %4\n v_ldexp_f64 %3, %3, %4\n"
                   "v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %4\n v_ldexp_f64 %2, %2, %4\n v_ldexp_f64 %3, %3, %4\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(u0 & 1));
    } else if (KIND == 5) {  // 8 x ds_read_b64 (+ one wait)
      unsigned a = lane * 8;
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n"
                   "ds_read_b64 %0, %4 offset:2048\n ds_read_b64 %1, %4 offset:2560\n ds_read_b64 %2, %4 offset:3072\n ds_read_b64 %3, %4 offset:3584\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "memory");
    } else if (KIND == 6) {  // 8 x s_mul_i32
      int s0 = i, s1 = i + 1;
      asm volatile("s_mul_i32 %0, %0, %1\n s_mul_i32 %1, %1, %0\n s_mul_i32 %0, %0, %1\n s_mul_i32 %1, %1, %0\n"
                   "s_mul_i32 %0, %0, %1\n s_mul_i32 %1, %1, %0\n s_mul_i32 %0, %0, %1\n s_mul_i32 %1, %1, %0\n"
                   : "+s"(s0), "+s"(s1));
      u0 += s0;
    } else if (KIND == 7) {  // 8 x v_mad_u64_u32
      unsigned long long m0 = u0, m1 = u1;
      asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n v_mad_u64_u32 %1, s[10:11], %2, %3, %1\n v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n v_mad_u64_u32 %1, s[10:11], %2, %3, %1\n"
                   "v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n v_mad_u64_u32 %1, s[10:11], %2, %3, %1\n v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n v_mad_u64_u32 %1, s[10:11], %2, %3, %1\n"
                   : "+v"(m0), "+v"(m1) : "v"(u2), "v"(u3) : "s10", "s11");
      u0 += (unsigned)m0 + (unsigned)m1;
    } else if (KIND == 8) {  // 8 x global_load_dwordx2 (same line: L1 hits) + wait
      asm volatile("global_load_dwordx2 %0, %4, %5\n global_load_dwordx2 %1, %4, %5 offset:512\n global_load_dwordx2 %2, %4, %5 offset:1024\n global_load_dwordx2 %3, %4, %5 offset:1536\n"
                   "global_load_dwordx2 %0, %4, %5 offset:2048\n global_load_dwordx2 %1, %4, %5 offset:2560\n global_load_dwordx2 %2, %4, %5 offset:3072\n global_load_dwordx2 %3, %4, %5 offset:3584\n"
                   "s_waitcnt vmcnt(0)\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"((unsigned)lane * 8), "s"(uni) : "memory");
    } else if (KIND == 9) {  // 8 x v_mfma_f64_4x4x4 (two chains)
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %0\n v_mfma_f64_4x4x4_4b_f64 %1, %2, %3, %1\n v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %0\n v_mfma_f64_4x4x4_4b_f64 %1, %2, %3, %1\n"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %0\n v_mfma_f64_4x4x4_4b_f64 %1, %2, %3, %1\n v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %0\n v_mfma_f64_4x4x4_4b_f64 %1, %2, %3, %1\n"
                   : "+v"(x0), "+v"(x1) : "v"(x2), "v"(x3));
    } else if (KIND == 10) {  // 8 x ds_write_b64
      unsigned a = lane * 8;
      asm volatile("ds_write_b64 %4, %0\n ds_write_b64 %4, %1 offset:512\n ds_write_b64 %4, %2 offset:1024\n ds_write_b64 %4, %3 offset:1536\n"
                   "ds_write_b64 %4, %0 offset:2048\n ds_write_b64 %4, %1 offset:2560\n ds_write_b64 %4, %2 offset:3072\n ds_write_b64 %4, %3 offset:3584\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "memory");
    }
  }
  sink[lane] = x0 + x1 + x2 + x3 + u0 + u1 + u2 + u3;
}

// MODE bit 0: the MFMA role runs; bit 1: the other role runs
template <int KIND, int MV>
__global__ __launch_bounds__(512) void coissue_kernel(double* sink, unsigned long long* clocks, int mode) {
  __shared__ double lds[8][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < 1024; i += 64) lds[wave][i] = i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (mode & 1) {
      double4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
      double a = lane, b = 1.0 / (lane + 1);
      for (int i = 0; i < kIters; i++) {
        if (MV == 0) {
          // 8 MFMAs, two dependent chains (as the walk issues them)
          asm volatile("v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n v_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n"
                       "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n v_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n"
                       "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n v_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n"
                       "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n v_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n"
                       : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
        } else if (MV == 1) {  // one chain
          asm volatile("v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n"
                       "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n"
                       "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n"
                       "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n"
                       : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
        } else if (MV == 2) {  // each MFMA followed by 60 clocks of s_nop
#define MF_NOP(c) "v_mfma_f64_16x16x4_f64 " c ", %2, %3, " c "\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 11\n"
          asm volatile(MF_NOP("%0") MF_NOP("%1") MF_NOP("%0") MF_NOP("%1") MF_NOP("%0") MF_NOP("%1") MF_NOP("%0") MF_NOP("%1")
                       : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
        } else if (MV == 3) {  // each MFMA followed by 32 clocks of s_nop
#define MF_NOP2(c) "v_mfma_f64_16x16x4_f64 " c ", %2, %3, " c "\n s_nop 15\n s_nop 15\n"
          asm volatile(MF_NOP2("%0") MF_NOP2("%1") MF_NOP2("%0") MF_NOP2("%1") MF_NOP2("%0") MF_NOP2("%1") MF_NOP2("%0") MF_NOP2("%1")
                       : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
        } else if (MV == 4) {  // each MFMA followed by s_sleep 1
#define MF_SLP(c) "v_mfma_f64_16x16x4_f64 " c ", %2, %3, " c "\n s_sleep 1\n"
          asm volatile(MF_SLP("%0") MF_SLP("%1") MF_SLP("%0") MF_SLP("%1") MF_SLP("%0") MF_SLP("%1") MF_SLP("%0") MF_SLP("%1")
                       : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
        }
      }
      sink[1024 + threadIdx.x] = c0.x + c1.y;
    }
  } else if (mode & 2) {
    other_loop<KIND>(sink + 2048 + (wave - 4) * 64, lane, lds[wave], sink + 8192);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0 && blockIdx.x == 0) clocks[wave] = t1 - t0;
}

template <int KIND, int MV = 0>
void run(const char* name, double* sink, unsigned long long* clocks, int blocks) {
  unsigned long long h[3][8];
  for (int mode = 1; mode <= 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL((coissue_kernel<KIND, MV>), dim3(blocks), dim3(512), 0, 0, sink, clocks, mode);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h[mode - 1], clocks, sizeof(h[0]), hipMemcpyDeviceToHost);
  }
  const double per = 8.0 * kIters;
  printf("%-16s mfma alone %6.1f  other alone %6.1f | together: mfma %6.1f  other %6.1f   (clocks per instruction, wave 0 / wave 4)\n",
         name, h[0][0] / per, h[1][4] / per, h[2][0] / per, h[2][4] / per);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  double* sink;
  unsigned long long* clocks;
  (void)hipMalloc(&sink, 1 << 20);
  (void)hipMemset(sink, 0, 1 << 20);
  (void)hipMalloc(&clocks, 64);
  printf("blocks %d, 8 waves each (4 MFMA f64 16x16x4 waves + 4 others, one of each per SIMD)\n", blocks);
  run<1>("v_add_u32", sink, clocks, blocks);
  run<2>("v_mul_f64", sink, clocks, blocks);
  run<3>("v_mov_b64", sink, clocks, blocks);
  run<4>("v_ldexp_f64", sink, clocks, blocks);
  run<5>("ds_read_b64", sink, clocks, blocks);
  run<10>("ds_write_b64", sink, clocks, blocks);
  run<6>("s_mul_i32", sink, clocks, blocks);
  run<7>("v_mad_u64_u32", sink, clocks, blocks);
  run<8>("global_load_x2", sink, clocks, blocks);
  run<9>("v_mfma_f64_4x4x4", sink, clocks, blocks);
  printf("MFMA wave variants (other: v_add_u32 / ds_read_b64 / s_mul_i32)\n");
  run<1, 1>("1chain v_add", sink, clocks, blocks);
  run<1, 2>("nop60 v_add", sink, clocks, blocks);
  run<5, 2>("nop60 ds_read", sink, clocks, blocks);
  run<6, 2>("nop60 s_mul", sink, clocks, blocks);
  run<2, 2>("nop60 v_mul_f64", sink, clocks, blocks);
  run<8, 2>("nop60 gload", sink, clocks, blocks);
  run<1, 3>("nop32 v_add", sink, clocks, blocks);
  run<1, 4>("sleep1 v_add", sink, clocks, blocks);
  return 0;
}
