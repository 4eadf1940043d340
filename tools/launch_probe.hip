#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(double* out, int work) {
  extern __shared__ double lds[];
  double v = threadIdx.x;
  for (int i = 0; i < work; i++) v = v * 1.0000001 + 1e-9;
  lds[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) out[0] = lds[0];
}
int main() {
  double* d; hipMalloc(&d, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int ldss[] = {1024, 12 * 1024, 24 * 1024, 52 * 1024};
  int blocks[] = {64, 256};
  for (int lds : ldss) for (int bs : blocks) for (int work : {0, 1000}) {
    const int total_threads = 30000 * 64;
    dim3 grid(total_threads / bs / 1000 + 1, 1000);
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      for (int i = 0; i < 10; i++) hipLaunchKernelGGL(empty_kernel, grid, dim3(bs), lds, 0, d, work);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("lds=%5d block=%3d work=%4d grid=(%d,1000): %.3f ms per launch (%.1f ns per workgroup)\n", lds, bs, work, grid.x, ms / 10, ms / 10 * 1e6 / (grid.x * 1000.0));
  }
  return 0;
}
