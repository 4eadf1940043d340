// Does the immediate offset of global_load_lds_dwordx4 move BOTH the global source and the
// LDS destination?  (If so, the pieces of one operand table need one scalar base and one M0
// value, not one pair per 1 KB piece.)   hipcc --offload-arch=gfx950 -O2 tools/lds_dma_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ void probe(const double* src, double* out) {
  __shared__ double buf[512];  // 4 KB
  const uint32_t lane16 = threadIdx.x * 16;
  for (int i = threadIdx.x; i < 512; i += 64) buf[i] = -1.0;
  __syncthreads();
  const uint32_t m0 = (uint32_t)(uintptr_t)(lds_ptr)buf;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
               "s_waitcnt vmcnt(0)" ::"v"(lane16), "s"(src), "s"(m0)
               : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = buf[i];
}

int main() {
  std::vector<double> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = i;
  double *d, *o;
  hipMalloc(&d, 8192);
  hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
  std::vector<double> r(512);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  // expectation if the offset applies to both sides: buf[i] == i for i < 384
  int both = 1, src_only = 1;
  for (int i = 0; i < 384; i++)
    if (r[i] != i) both = 0;
  // offset on the source only: every instruction lands at buf[0..127]: last one wins
  for (int i = 0; i < 128; i++)
    if (r[i] != 256 + i) src_only = 0;
  printf("offset moves source and LDS destination: %s; source only: %s\n", both ? "YES" : "no",
         src_only ? "YES" : "no");
  printf("buf[0]=%g buf[127]=%g buf[128]=%g buf[255]=%g buf[256]=%g buf[383]=%g buf[384]=%g\n", r[0],
         r[127], r[128], r[255], r[256], r[383], r[384]);
  return 0;
}
