// Test helper: runs the engine's tree-setup launcher on a file of parent-id vectors and
// dumps what it produced (status, macro counts, log-likelihood schedule, gradient
// schedule), so that the register-array kernel (N <= 64) can be compared bit for bit with
// the general LDS kernel (MI_PHYLO_TREE_SETUP=lds) and the workgroup-per-tree kernel of large
// trees (default above 256 nodes, MI_PHYLO_TREE_SETUP=wg below).
// Usage: prog n T in.bin out.bin [rooted [fold]]   (fold = 1: the arena's slot assignment is asked
// of the set-up launch itself -- tree_setup_wg_kernel does it, the others leave it to launch_macro_slots)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../../libsbn_amd/csrc/mi_phylo_kernels.h"
using namespace miphylo;
int main(int argc, char** argv) {
  const int n = atoi(argv[1]), T = atoi(argv[2]);
  const int N = 2 * n - 1;
  const int rooted = argc > 5 ? atoi(argv[5]) : 0;
  FILE* f = fopen(argv[3], "rb");
  std::vector<int32_t> pid((size_t)T * (2 * n - 3 + rooted));
  fread(pid.data(), 4, pid.size(), f); fclose(f);
  std::vector<double> bl((size_t)T * (2 * n - 2 + rooted), 0.1);
  int32_t *d_pid, *d_scratch, *d_mc, *d_status; double *d_bl, *d_ble; SchedEntry* d_sched; MacroEntry* d_mac;
  hipMalloc(&d_pid, pid.size() * 4); hipMalloc(&d_bl, bl.size() * 8); hipMalloc(&d_scratch, (size_t)T * 13 * N * 4);
  hipMalloc(&d_mc, T * 4); hipMalloc(&d_status, 8); hipMalloc(&d_ble, (size_t)T * N * 8);
  hipMalloc(&d_sched, sizeof(SchedEntry) * (size_t)T * (n - 1)); hipMalloc(&d_mac, sizeof(MacroEntry) * (size_t)T * macro_stride(n));
  hipMemcpy(d_pid, pid.data(), pid.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_bl, bl.data(), bl.size() * 8, hipMemcpyHostToDevice);
  hipMemset(d_status, 0, 8); hipMemset(d_mac, 0, sizeof(MacroEntry) * (size_t)T * macro_stride(n));
  TreeSetupArgs a{}; a.n = n; a.T = T; a.rooted = rooted; a.parent_ids = d_pid; a.bl = d_bl; a.rates = nullptr; a.scratch = d_scratch;
  a.sched = d_sched; a.macros = d_mac; a.macro_count = d_mc; a.bl_eff = d_ble; a.status = d_status; a.max_slots = 32; a.need_slots = 1;
  const int fold = argc > 6 ? atoi(argv[6]) : 0;
  MacroEntry* d_mac2; int32_t* d_need;
  hipMalloc(&d_mac2, sizeof(MacroEntry) * (size_t)T * macro_stride(n)); hipMalloc(&d_need, T * 4);
  hipMemset(d_mac2, 0, sizeof(MacroEntry) * (size_t)T * macro_stride(n)); hipMemset(d_need, 0, T * 4);
  if (fold) { a.arena_macros = d_mac2; a.slot_need = d_need; }
  const bool folded = launch_tree_setup(a, nullptr); hipDeviceSynchronize();
  std::vector<SchedEntry> sc((size_t)T * (n - 1)); std::vector<MacroEntry> mac((size_t)T * macro_stride(n)); std::vector<int32_t> mc(T); int32_t st[2];
  hipMemcpy(sc.data(), d_sched, sc.size() * sizeof(SchedEntry), hipMemcpyDeviceToHost);
  hipMemcpy(mac.data(), d_mac, mac.size() * sizeof(MacroEntry), hipMemcpyDeviceToHost);
  hipMemcpy(mc.data(), d_mc, T * 4, hipMemcpyDeviceToHost); hipMemcpy(st, d_status, 8, hipMemcpyDeviceToHost);
  // the arena variant's macro order / LDS slots / arena indices (macro_slots kernels)
  std::vector<MacroEntry> mac2((size_t)T * macro_stride(n)); std::vector<int32_t> need(T);
  if (st[0] == 0) {
    if (!folded) launch_macro_slots(d_mac, d_mac2, d_mc, n, T, d_need, d_status, nullptr);
    hipDeviceSynchronize();
    hipMemcpy(mac2.data(), d_mac2, mac2.size() * sizeof(MacroEntry), hipMemcpyDeviceToHost);
    hipMemcpy(need.data(), d_need, T * 4, hipMemcpyDeviceToHost);
  }
  FILE* o = fopen(argv[4], "wb");
  fwrite(st, 4, 2, o); fwrite(mc.data(), 4, T, o); fwrite(sc.data(), sizeof(SchedEntry), sc.size(), o); fwrite(mac.data(), sizeof(MacroEntry), mac.size(), o);
  fwrite(need.data(), 4, T, o); fwrite(mac2.data(), sizeof(MacroEntry), mac2.size(), o); fclose(o);
  printf("status %d %d, macro_count[0]=%d folded=%d\n", st[0], st[1], mc[0], (int)folded);
  return 0;
}
