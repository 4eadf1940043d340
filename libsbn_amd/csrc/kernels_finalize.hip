// Tile reduction, analytic substitution-gradient finish, per-tree finalize.
// (gfx950 / CDNA4, wave64; see DESIGN.md for the mapping and what bounds each kernel.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "mi_phylo_device_utils.h"
#include "mi_phylo_kernels.h"

namespace miphylo {

namespace {
using namespace dev;

// ------------------------------------------------------------------------
// Tile reduction (one workgroup per evaluation): four waves split the tiles
// (wave w takes tiles w, w+4, ...), combine through LDS in a fixed order.
// ------------------------------------------------------------------------
// keep (reduce_finalize_kernel): the sums stay in LDS for the finalize step -- [2][N] by node
// id, then the log-likelihood -- instead of going through g_sum / ll_sum in HBM and back.
__device__ __forceinline__ void reduce_tiles_body(const ReduceArgs& a, const int b, double* red_lds /* [4][W] */,
                                                  double* keep = nullptr) {
  __shared__ double llw[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int N2 = 2 * a.N;
  const int W = a.g_width ? a.g_width : N2;  // doubles per (evaluation, tile)
  const int t = b < a.T ? b : b - a.T;  // gradient evaluations: [0,T) main, [T,2T) site pass
  // Everything that does not depend on anything is requested FIRST (round 5): the macro
  // count, the log-likelihood terms, and the schedule words that say which node a column of
  // positional sums belongs to -- their round trips then pass under the tile loop's instead of
  // standing in front of it and behind it (two of the ~five this kernel's time is made of on a
  // small batch).
  const bool positional = b < a.Eg && a.g_width;
  const int mcount = positional ? a.macro_count[t] : 0;
  double llp = 0;
  for (int i = threadIdx.x; i < a.ll_used.of(b); i += 256) llp += a.ll_part[(size_t)b * a.ll_tiles + i];
  const MacroEntry* mac = a.macros + (size_t)t * macro_stride(a.n);
  int my_shape = 0, my_node = 0;  // of column threadIdx.x (the first of this thread's columns)
  if (positional && (int)threadIdx.x < W - a.extra) {
    const int m = threadIdx.x / (kMacroPositions * 2), pos = (threadIdx.x % (kMacroPositions * 2)) >> 1;
    const int32_t* mw = reinterpret_cast<const int32_t*>(mac + m);
    my_shape = mw[0];
    my_node = mw[1 + pos];  // child[0..1], grand[0..3] follow the shape word
  }
  if (b < a.Eg) {
    const double* src = a.g_part + (size_t)b * a.g_tiles * W;
    const int tail = a.g_width ? W - a.extra : W;  // plain sums after the positional part
    // which columns: without trailing plain sums (every call but the analytic substitution
    // gradient's) all W of them -- columns past the tree's last macro hold nothing anyone
    // wrote, are summed for nothing and dropped below --, so that the loads do not wait for
    // the macro count; with trailing sums the positional ones, then those
    const int used_cols = a.extra ? mcount * kMacroPositions * 2 : W;
    // Four columns per lane side by side and the tile loop unrolled: 32 loads in flight per
    // lane (round 5; a column at a time with 16 in flight, the 156 columns of a DS1 tree were
    // six memory round trips one after the other -- most of this kernel's time on a small
    // batch).  Per (wave, column) the additions are the same in the same order.
    constexpr int VU = 4;
    const int total = used_cols + a.extra;
    for (int v0 = lane; v0 < total; v0 += 64 * VU) {
      int col[VU];
      bool on[VU];
      double s0[VU], s1[VU];
#pragma unroll
      for (int j = 0; j < VU; j++) {
        const int v = v0 + 64 * j;
        on[j] = v < total;
        col[j] = !on[j] ? 0 : (v < used_cols ? v : tail + (v - used_cols));
        s0[j] = s1[j] = 0;
      }
      int i = wv;
#pragma unroll 4
      for (; i + 4 < a.g_tiles; i += 8) {
        // (no condition around a load: lanes past the last column read column 0 and drop
        // the sum -- a branch per load would put a wait behind each)
#pragma unroll
        for (int j = 0; j < VU; j++) {
          s0[j] += src[(size_t)i * W + col[j]];
          s1[j] += src[(size_t)(i + 4) * W + col[j]];
        }
      }
      if (i < a.g_tiles) {
#pragma unroll
        for (int j = 0; j < VU; j++) s0[j] += src[(size_t)i * W + col[j]];
      }
#pragma unroll
      for (int j = 0; j < VU; j++)
        if (on[j]) red_lds[wv * W + col[j]] = s0[j] + s1[j];
    }
  }
  llp = wave_sum(llp);
  if (lane == 0) llw[wv] = llp;
  int used = W;
  if (positional) used = mcount * kMacroPositions * 2;
  __syncthreads();
  if (threadIdx.x == 0) (keep ? keep[N2] : a.ll_sum[b]) = (llw[0] + llw[1]) + (llw[2] + llw[3]);
  if (b >= a.Eg) return;
  double* out = keep ? keep : a.g_sum + (size_t)b * N2;
  if (!a.g_width) {
    for (int v = threadIdx.x; v < N2; v += 256)
      out[v] = (red_lds[v] + red_lds[W + v]) + (red_lds[2 * W + v] + red_lds[3 * W + v]);
    return;
  }
  // positional: entry (m, pos, q) belongs to the edge above child/grandchild `pos` of macro m
  if (a.extra)
    for (int v = threadIdx.x; v < a.extra; v += 256) {
      const int c = W - a.extra + v;
      a.x_sum[(size_t)b * a.extra + v] =
          (red_lds[c] + red_lds[W + c]) + (red_lds[2 * W + c] + red_lds[3 * W + c]);
    }
  if (threadIdx.x < 2) out[threadIdx.x * a.N + a.N - 1] = 0.0;  // the root has no edge
  for (int v = threadIdx.x; v < used; v += 256) {
    const int m = v / (kMacroPositions * 2), r = v - m * (kMacroPositions * 2);
    const int pos = r >> 1, q = r & 1;
    int shape = my_shape, node = my_node;
    if (v >= 256) {  // (larger trees: this thread's further columns)
      const MacroEntry& me = mac[m];
      shape = me.shape;
      node = pos < 2 ? me.child[pos] : me.grand[pos - 2];
    }
    const bool exists = pos < 2 || ((shape >> (2 * ((pos - 2) >> 1))) & 3) == 2;
    if (!exists) continue;
    out[q * a.N + node] =
        (red_lds[v] + red_lds[W + v]) + (red_lds[2 * W + v] + red_lds[3 * W + v]);
  }
}

__global__ __launch_bounds__(256) void reduce_tiles_kernel(ReduceArgs a) {
  extern __shared__ double red_lds[];
  reduce_tiles_body(a, blockIdx.x, red_lds);
}

// ------------------------------------------------------------------------
// Analytic substitution-model gradient, last step (one thread per tree).
// In: H^T per category block (64 doubles, lane order of gradient_mfma_kernel) and the
// root term d logL / d pi (4 doubles).  GTR as built by model_setup_kernel
// (substitution_model.cpp:17-80): Qt_ab = rho_ab pi_b, mu = sum_a pi_a sum_{b != a} Qt_ab,
// Q = Qt / mu.  Out: derivatives w.r.t. the stick-breaking coordinates of the rates (5)
// and of the frequencies (3), the quantities the reference obtains by finite
// differences (fat_beagle.cpp:400-465).
// ------------------------------------------------------------------------
__device__ void stick_breaking_chain(int K, const double* x, const double* g, double* out) {
  // x = stick_breaking(y): x_k = s_k z_k, s_k = prod_{j<k} (1 - z_j), x_{K-1} = s_{K-1};
  // dz_k/dy_k = z_k (1 - z_k)  =>  dL/dy_k = g_k x_k (1 - z_k) - z_k sum_{m>k} g_m x_m
  double tail[8];
  double acc = 0;
  for (int m = K - 1; m >= 0; m--) {
    tail[m] = acc;  // sum_{m' > m} g_m' x_m'
    acc += g[m] * x[m];
  }
  double used = 0;
  for (int k = 0; k < K - 1; k++) {
    const double z = x[k] / (1.0 - used);
    out[k] = g[k] * x[k] * (1.0 - z) - z * tail[k];
    used += x[k];
  }
}

__global__ void subst_gradient_kernel(SubstGradArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.T) return;
  const DevModel& m = a.models[t];
  const double* x = a.x_sum + (size_t)t * kSubstExtra;
  const double* row = a.params + (size_t)t * a.param_count;
  double H[16];  // H[i][j] = sum over blocks of H^T[j][i]
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double sum = 0;
      for (int b = 0; b < 4; b++) sum += x[16 * j + 4 * b + i];
      H[i * 4 + j] = sum;
    }
  // D = V^-T H V^T : D[a][b] = sum_ij Vinv[i][a] H[i][j] V[b][j]
  double HV[16], D[16];
  for (int i = 0; i < 4; i++)
    for (int b = 0; b < 4; b++) {
      double sum = 0;
      for (int j = 0; j < 4; j++) sum += H[i * 4 + j] * m.V[b * 4 + j];
      HV[i * 4 + b] = sum;
    }
  double S = 0;  // <D, Q>
  for (int c = 0; c < 4; c++)
    for (int b = 0; b < 4; b++) {
      double sum = 0;
      for (int i = 0; i < 4; i++) sum += m.Vinv[i * 4 + c] * HV[i * 4 + b];
      D[c * 4 + b] = sum;
      S += sum * m.Q[c * 4 + b];
    }
  double rates[6], pi[4];
  for (int i = 0; i < 6; i++) rates[i] = row[a.rates_off + i];
  for (int i = 0; i < 4; i++) pi[i] = row[a.freqs_off + i];
  double mu = 0;
  {
    int ri = 0;
    for (int i = 0; i < 4; i++)
      for (int k = i + 1; k < 4; k++) {
        const double r = rates[ri++];
        mu += pi[i] * r * pi[k] + pi[k] * r * pi[i];
      }
  }
  // d logL / d Qt_ab (a != b, diagonal follows) = (D_ab - D_aa - S pi_a) / mu
  auto dQt = [&](int c, int b) { return (D[c * 4 + b] - D[c * 4 + c] - S * pi[c]) / mu; };
  double g_rate[6], g_pi[4];
  for (int c = 0; c < 4; c++) g_pi[c] = x[64 + c] + S * m.Q[c * 4 + c];  // root term, explicit pi in mu
  {
    int ri = 0;
    for (int i = 0; i < 4; i++)
      for (int k = i + 1; k < 4; k++) {
        const double r = rates[ri];
        g_rate[ri] = pi[k] * dQt(i, k) + pi[i] * dQt(k, i);
        g_pi[k] += r * dQt(i, k);  // Qt_ik = r pi_k
        g_pi[i] += r * dQt(k, i);  // Qt_ki = r pi_i
        ri++;
      }
  }
  double* out = a.out_subst + (size_t)t * 8;
  stick_breaking_chain(6, rates, g_rate, out);
  stick_breaking_chain(4, pi, g_pi, out + 5);
}

// ------------------------------------------------------------------------
// Finalize (one thread per tree): sum tile partials in a fixed order
// (deterministic), assemble PhyloGradient, rooted chain rule.
// ------------------------------------------------------------------------
__device__ double sum_tiles(const double* part, int tiles) {
  double s = 0;
  for (int i = 0; i < tiles; i++) s += part[i];
  return s;
}

__device__ double node_partial(int v, int n, const double* h, const double* ratios,
                               const double* bound) {
  return (h[v] - bound[v]) / ratios[v - n];
}
__device__ double epoch_addition(int v, int c, int n, const double* h, const double* ratios,
                                 const double* bound, const double* acc) {
  if (c < n) return 0.0;
  if (bound[v] == bound[c]) return acc[c - n] * ratios[c - n] / ratios[v - n];
  return acc[c - n] * ratios[c - n] / (h[v] - bound[c]) * node_partial(v, n, h, ratios, bound);
}

// rooted_gradient_transforms.cpp:78-130 for one input vector gh -> out (+ root entry)
__device__ void ratio_transform(int n, const int32_t* c0, const int32_t* c1, const double* h,
                                const double* ratios, const double* bound, const double* gh,
                                double* mult, double* out) {
  const int N = 2 * n - 1, root = N - 1;
  for (int i = 0; i < n - 1; i++) out[i] = 0;
  for (int v = n; v < root; v++) {
    out[v - n] += node_partial(v, n, h, ratios, bound) * gh[v - n];
    out[v - n] += epoch_addition(v, c0[v - n], n, h, ratios, bound, out);
    out[v - n] += epoch_addition(v, c1[v - n], n, h, ratios, bound, out);
  }
  mult[root - n] = 1.0;
  for (int v = root; v >= n; v--) {
    const int a = c0[v - n], b = c1[v - n];
    if (a >= n) mult[a - n] = ratios[a - n] * mult[v - n];
    if (b >= n) mult[b - n] = ratios[b - n] * mult[v - n];
  }
  double sum = 0;
  for (int i = 0; i < n - 1; i++) sum += gh[i] * mult[i];
  out[root - n] = sum;
}

// The two O(n) recurrences of the rooted chain rule (rooted_gradient_transforms.cpp:47-64 bottom-up,
// :102-130 top-down) for a tree of at most 64 NB internal nodes, in the REGISTERS of one wave:
// lane l holds nodes l, l + 64, ...; a step fetches what it needs from the owning lanes with
// v_readlane (a node id is wave-uniform) and writes its result into the owning lane -- no LDS
// round trip on the dependent chain (one fluA tree: 58 k -> 13 k cycles; the chain of LDS reads and
// writes of a single lane was 26 of the kernel's 35 microseconds).  Same operations in the same
// order as the single-lane loops below (explicitly rounded): bit-identical.
template <int NB>
struct WaveArrayD {
  double r[NB];
  __device__ __forceinline__ double rd(int idx) const {  // idx: wave-uniform
    double x = r[0];
#pragma unroll
    for (int nb = 1; nb < NB; nb++) x = (idx >> 6) == nb ? r[nb] : x;
    const int l = idx & 63;
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                            __builtin_amdgcn_readlane(__double2loint(x), l));
  }
  __device__ __forceinline__ void wr(int idx, double v, int lane) {
#pragma unroll
    for (int nb = 0; nb < NB; nb++) r[nb] = (lane + 64 * nb == idx) ? v : r[nb];
  }
};
template <int NB>
struct WaveArrayI {
  int r[NB];
  __device__ __forceinline__ int rd(int idx) const {
    int x = r[0];
#pragma unroll
    for (int nb = 1; nb < NB; nb++) x = (idx >> 6) == nb ? r[nb] : x;
    return __builtin_amdgcn_readlane(x, idx & 63);
  }
};
template <int NB>
__device__ __forceinline__ void rooted_recurrences_wave(int n, int lane, const int32_t* c0, const int32_t* c1,
                                                        const double* Pv, const double* hg, const double* aux,
                                                        const double* E0, const double* E1,
                                                        const double* ratios, double* outA, double* outB,
                                                        double* mult) {
  WaveArrayD<NB> pa, pb, e0, e1, ra, oa, ob, mu;
  WaveArrayI<NB> k0, k1;  // internal children (index - n), -1 for a leaf
#pragma unroll
  for (int nb = 0; nb < NB; nb++) {
    const int i = lane + 64 * nb;
    const bool in = i < n - 1;
    const int a0 = in ? c0[i] : 0, a1 = in ? c1[i] : 0;
    k0.r[nb] = in && a0 >= n ? a0 - n : -1;
    k1.r[nb] = in && a1 >= n ? a1 - n : -1;
    pa.r[nb] = in ? __dmul_rn(Pv[i], hg[i]) : 0.0;
    pb.r[nb] = in ? __dmul_rn(Pv[i], aux[i]) : 0.0;
    e0.r[nb] = in ? E0[i] : 0.0;
    e1.r[nb] = in ? E1[i] : 0.0;
    ra.r[nb] = in ? ratios[i] : 0.0;
    oa.r[nb] = ob.r[nb] = mu.r[nb] = 0.0;
  }
  // (a step broadcasts only the children's values; every lane then evaluates its node's
  // formula with its own coefficients and the owner of node i keeps the result.
  // Nodes are taken 64 at a time -- block b lives in register b of each array and its
  // children in registers <= b: no selection among registers for the first 64 nodes.)
  auto rd_upto = [&](const WaveArrayD<NB>& w, int b, int idx) {  // idx < 64 (b + 1), uniform
    double x = w.r[0];
#pragma unroll
    for (int nb = 1; nb < NB; nb++)
      if (nb <= b) x = (idx >> 6) == nb ? w.r[nb] : x;
    const int l = idx & 63;
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                            __builtin_amdgcn_readlane(__double2loint(x), l));
  };
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const int end = min(64 * (b + 1), n - 2);
    for (int i = 64 * b; i < end; i++) {
      const int l = i & 63;
      // (a leaf child, index -1, contributes NOTHING: its term is selected away on the
      // wave-uniform index, never multiplied by zero -- node 0's entries may be Inf / NaN for a
      // degenerate tree (a zero ratio, a height on its bound) and 0 x Inf would poison every
      // node with a tip child, where the single-lane loops and the reference stay finite)
      const int c0 = __builtin_amdgcn_readlane(k0.r[b], l);
      const int c1 = __builtin_amdgcn_readlane(k1.r[b], l);
      const int a0 = max(c0, 0), a1 = max(c1, 0);
      const double xa0 = rd_upto(oa, b, a0), xb0 = rd_upto(ob, b, a0);
      const double xa1 = rd_upto(oa, b, a1), xb1 = rd_upto(ob, b, a1);
      const double ta = c0 >= 0 ? fma(e0.r[b], xa0, pa.r[b]) : pa.r[b];
      const double tb = c0 >= 0 ? fma(e0.r[b], xb0, pb.r[b]) : pb.r[b];
      const double va = c1 >= 0 ? fma(e1.r[b], xa1, ta) : ta;
      const double vb = c1 >= 0 ? fma(e1.r[b], xb1, tb) : tb;
      const bool mine = lane == l;
      oa.r[b] = mine ? va : oa.r[b];
      ob.r[b] = mine ? vb : ob.r[b];
    }
  }
  mu.wr(n - 2, 1.0, lane);
#pragma unroll
  for (int b = NB - 1; b >= 0; b--) {
    const int top = min(64 * (b + 1), n - 1) - 1;
    for (int i = top; i >= 64 * b; i--) {
      const int l = i & 63;
      const int a0 = __builtin_amdgcn_readlane(k0.r[b], l);  // (-1: no lane owns it)
      const int a1 = __builtin_amdgcn_readlane(k1.r[b], l);
      const double m = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mu.r[b]), l),
                                        __builtin_amdgcn_readlane(__double2loint(mu.r[b]), l));
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        if (nb > b) continue;
        const int node = lane + 64 * nb;
        mu.r[nb] = (node == a0 || node == a1) ? __dmul_rn(ra.r[nb], m) : mu.r[nb];
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; nb++) {
    const int i = lane + 64 * nb;
    if (i < n - 1) {
      outA[i] = oa.r[nb];
      outB[i] = ob.r[nb];
      mult[i] = mu.r[nb];
    }
  }
}

// keep: see reduce_tiles_body (the reduced sums in LDS instead of a.ll_part / a.g_part)
// (LDS, ROOTED: LDS and ROOTED as compile-time constants -- with either left a run-time
// value the working-set pointers are "LDS or global" selects, and every access through them a
// flat instruction on the dependent chains of this kernel: 156 of them until round 5)
template <bool LDS, bool ROOTED>
__device__ __forceinline__ void finalize_body_t(const FinalizeArgs& a, const int t, double* fin_lds,
                                                const double* keep) {
  // One wave per tree: lanes run over nodes / tiles for the reductions, lane 0
  // walks the O(n) recurrences of the rooted chain rule.  Working set (6n doubles, for
  // rooted trees also the tree's heights, bounds, ratios, rates and the ratio gradient
  // being built: each access of those recurrences is on a dependent chain, and a global
  // load there costs ten LDS reads) in LDS unless the tree is too large.
  const int lane = threadIdx.x;
  const int n = a.n, N = a.N, T = a.T;
  double* base = LDS ? fin_lds : a.scratch + (size_t)t * 6 * n;
  int32_t* c0 = reinterpret_cast<int32_t*>(base);
  int32_t* c1 = c0 + n;
  double* work = base + n;  // 5n doubles
  const SchedEntry* sched = a.sched + (size_t)t * (n - 1);
  if (ROOTED)  // (only the rooted chain rule and log-det-Jacobian look at the tree)
    for (int i = lane; i < n - 1; i += 64) {
      const SchedEntry se = sched[i];
      c0[se.node - n] = se.child0;
      c1[se.node - n] = se.child1;
    }
  const double* h = a.node_heights ? a.node_heights + (size_t)t * N : nullptr;
  const double* bd = a.node_bounds ? a.node_bounds + (size_t)t * N : nullptr;
  const double* ratios = a.height_ratios ? a.height_ratios + (size_t)t * (n - 1) : nullptr;
  const double* rates = a.rates ? a.rates + (size_t)t * (N - 1) : nullptr;
  double* outr_stage = nullptr;
  if (ROOTED && LDS) {
    double* stage = fin_lds + 6 * n;  // h[N] | bd[N] | rates[N] | ratios[n] | out[n]
    for (int v = lane; v < N; v += 64) {
      if (h) stage[v] = h[v];
      if (bd) stage[N + v] = bd[v];
      if (rates && v < N - 1) stage[2 * N + v] = rates[v];
      if (ratios && v < n - 1) stage[3 * N + v] = ratios[v];
    }
    if (h) h = stage;
    if (bd) bd = stage + N;
    if (rates) rates = stage + 2 * N;
    if (ratios) ratios = stage + 3 * N;
    outr_stage = stage + 3 * N + n;
  }
  __syncthreads();
  __shared__ double sh_ll, sh_jac;
  {
    // log-det-Jacobian of the height-ratio transform (fat_beagle.cpp:82-94): the sum over the
    // internal non-root nodes c of log(h_parent(c) - bound_c).  Every lane takes the internal
    // nodes lane, lane + 64, ... and their internal children, the wave sums up in a fixed
    // order.  (Until round 3 lane 0 walked the tree in the reference's traversal order with
    // a logarithm per step: 20 of the 44 microseconds of a one-tree fluA call.  The order of
    // the additions differs from the reference's in the last bits only.)
    double jac = 0.0;
    if (ROOTED && (a.with_jacobian || (a.gradient && a.gtr))) {
      for (int i = lane; i < n - 1; i += 64) {
        const int v = n + i, a0 = c0[i], a1 = c1[i];
        if (a0 >= n) jac += log(h[v] - bd[a0]);
        if (a1 >= n) jac += log(h[v] - bd[a1]);
      }
      jac = wave_sum(jac);
    }
    if (lane == 0) {
      sh_ll = keep ? keep[2 * N] : sum_tiles(a.ll_part + (size_t)t * a.ll_tiles, a.ll_used.of(t));
      sh_jac = jac;
    }
  }
  __syncthreads();
  const double ll = sh_ll, jac = sh_jac;
  if (!a.gradient) {
    if (lane == 0) a.out_ll[t] = a.with_jacobian ? ll + jac : ll;
    return;
  }
  if (lane == 0) a.out_ll[t] = ll;
  const double* ble = a.bl_eff + (size_t)t * N;
  // branch gradient of the main evaluation: tile partials summed in tile order
  double* bg = work;  // N doubles (N < 2n)
  for (int v = lane; v < N; v += 64) {
    double sum = 0;
    if (keep) {
      sum += keep[v];  // (0 + x, as the loop below forms it)
    } else {
      for (int i = 0; i < a.g_tiles; i++)
        sum += a.g_part[(((size_t)t * a.g_tiles + i) * 2) * N + v];
    }
    bg[v] = sum;
  }
  if (a.out_site && (a.site_fused || a.site_separate)) {
    // DiscreteSiteModelGradient fat_beagle.cpp:389-398
    const size_t gi = a.site_separate ? (size_t)T + t : (size_t)t;
    double r = 0;
    for (int v = lane; v < N - 1; v += 64) {
      double sum = 0;
      if (keep) {
        sum += keep[N + v];
      } else {
        for (int i = 0; i < a.g_tiles; i++)
          sum += a.g_part[((gi * a.g_tiles + i) * 2 + 1) * N + v];
      }
      r += sum * ble[v];
    }
    r = wave_sum(r);
    if (lane == 0) a.out_site[t] = r;
  }
  if (a.gtr && a.out_subst && lane < 8) {
    // fat_beagle.cpp:431,455-464: rates (5) then frequencies (3)
    const int coord = lane < 5 ? 3 + lane : lane - 5;
    const size_t ep = (size_t)T + (size_t)t * 16 + 2 * coord;
    double lp = sum_tiles(a.ll_part + ep * a.ll_tiles, a.ll_used.of((long)ep));
    double lm = sum_tiles(a.ll_part + (ep + 1) * a.ll_tiles, a.ll_used.of((long)ep + 1));
    if (ROOTED) {
      lp += jac;
      lm += jac;
    }
    a.out_subst[(size_t)t * 8 + lane] = (lp - lm) / (2. * 1.e-6);
  }
  __syncthreads();
  if (!ROOTED) {
    double* ob = a.out_branch + (size_t)t * N;
    // fixed node = second child of the root (fat_beagle.cpp:499); root entry is 0
    for (int v = lane; v < N; v += 64) ob[v] = v < N - 2 ? bg[v] : 0.0;
    return;
  }
  // ---- rooted: clock + ratios/root-height gradients ----
  const double* tb = a.bl_raw + (size_t)t * N;
  double* oc = a.out_clock + (size_t)t * (N - 1);
  const int rc = a.rate_counts[t];
  if (rc == 1) {
    // ClockGradient fat_beagle.cpp:367-387 (strict): sum_i g_i * t_i
    double acc = 0;
    for (int v = lane; v < N - 1; v += 64) acc += bg[v] * tb[v];
    acc = wave_sum(acc);
    for (int v = lane; v < N - 1; v += 64) oc[v] = v == 0 ? acc : 0.0;
  } else if (rc == N - 1) {
    for (int v = lane; v < N - 1; v += 64) oc[v] = bg[v] * tb[v];
  } else {
    if (lane == 0) set_status(a.status, kBadRateCount, t);
    for (int v = lane; v < N - 1; v += 64) oc[v] = 0;
  }
  double* out_global = a.out_ratios + (size_t)t * (n - 1);
  if (outr_stage) {
    // Working set in LDS: everything that is not a recurrence is done by all lanes (height
    // gradient, the per-node coefficients of the ratio transform, the root sums), and the
    // two ratio transforms (height gradient, log-Jacobian) share ONE bottom-up and one
    // top-down chain of multiply-adds (rooted_gradient_transforms.cpp:17-170).
    const int root = N - 1;
    double* rw = outr_stage + n;  // 8n doubles
    double *hg = rw, *aux = rw + n, *Pv = rw + 2 * n, *E0 = rw + 3 * n, *E1 = rw + 4 * n;
    double *outA = rw + 5 * n, *outB = rw + 6 * n, *mult = rw + 7 * n;
    for (int i = lane; i < n - 1; i += 64) {
      const int v = n + i, a0 = c0[i], a1 = c1[i];
      // HeightGradient :17-37
      double x = v != root ? -bg[v] * rates[v] : 0.0;
      x += bg[a0] * rates[a0];
      x += bg[a1] * rates[a1];
      hg[i] = x;
      aux[i] = i < n - 2 ? 1.0 / (h[v] - bd[v]) : 0.0;  // d log|J| / d height
      // out_v = partial_v gh_v + sum over internal children c of out_c * epoch(v, c) :47-64
      const double partial = v != root ? (h[v] - bd[v]) / ratios[i] : 0.0;
      auto epoch = [&](int c) {
        if (c < n || v == root) return 0.0;
        if (bd[v] == bd[c]) return ratios[c - n] / ratios[i];
        return ratios[c - n] / (h[v] - bd[c]) * partial;
      };
      Pv[i] = partial;
      E0[i] = epoch(a0);
      E1[i] = epoch(a1);
    }
    __syncthreads();
    if (n - 1 <= 64) {
      rooted_recurrences_wave<1>(n, lane, c0, c1, Pv, hg, aux, E0, E1, ratios, outA, outB, mult);
    } else if (n - 1 <= 128) {
      rooted_recurrences_wave<2>(n, lane, c0, c1, Pv, hg, aux, E0, E1, ratios, outA, outB, mult);
    } else if (n - 1 <= 256) {
      rooted_recurrences_wave<4>(n, lane, c0, c1, Pv, hg, aux, E0, E1, ratios, outA, outB, mult);
    } else if (lane == 0) {
      for (int i = 0; i < n - 2; i++) {
        // a leaf child contributes nothing -- and must not be READ either: LDS is not
        // cleared between workgroups, and 0 * (NaN residue) would poison every ancestor
        const bool i0 = c0[i] >= n, i1 = c1[i] >= n;
        double va = __dmul_rn(Pv[i], hg[i]), vb = __dmul_rn(Pv[i], aux[i]);
        if (i0) {
          va = fma(E0[i], outA[c0[i] - n], va);
          vb = fma(E0[i], outB[c0[i] - n], vb);
        }
        if (i1) {
          va = fma(E1[i], outA[c1[i] - n], va);
          vb = fma(E1[i], outB[c1[i] - n], vb);
        }
        outA[i] = va;
        outB[i] = vb;
      }
      mult[root - n] = 1.0;  // :102-130
      for (int v = root; v >= n; v--) {
        const int a0 = c0[v - n], a1 = c1[v - n];
        const double m = mult[v - n];
        if (a0 >= n) mult[a0 - n] = __dmul_rn(ratios[a0 - n], m);
        if (a1 >= n) mult[a1 - n] = __dmul_rn(ratios[a1 - n], m);
      }
    }
    __syncthreads();
    double ra = 0, rb = 0;
    for (int i = lane; i < n - 1; i += 64) {
      ra += hg[i] * mult[i];
      rb += aux[i] * mult[i];
    }
    ra = wave_sum(ra);
    rb = wave_sum(rb);
    for (int i = lane; i < n - 2; i += 64) out_global[i] = outA[i] + (outB[i] - 1.0 / ratios[i]);
    if (lane == 0) out_global[n - 2] = ra + rb;
    return;
  }
  double* outr = out_global;
  if (lane == 0) {
  double* hg = work + 2 * n;    // n-1
  double* aux = work + 3 * n;   // n-1 (log_time)
  double* jacg = work + 4 * n;  // n-1
  // HeightGradient rooted_gradient_transforms.cpp:17-37
  for (int v = N - 1; v >= n; v--) {
    double x = v != N - 1 ? -bg[v] * rates[v] : 0.0;
    x += bg[c0[v - n]] * rates[c0[v - n]];
    x += bg[c1[v - n]] * rates[c1[v - n]];
    hg[v - n] = x;
  }
  double* mult = work;  // bg is dead from here on
  ratio_transform(n, c0, c1, h, ratios, bd, hg, mult, outr);
  for (int i = 0; i < n - 1; i++) aux[i] = 0;
  for (int i = 0; i < n - 2; i++) aux[i] = 1.0 / (h[n + i] - bd[n + i]);
  ratio_transform(n, c0, c1, h, ratios, bd, aux, mult, jacg);
  for (int i = 0; i < n - 2; i++) outr[i] += jacg[i] - 1.0 / ratios[i];
  outr[n - 2] += jacg[n - 2];
  }
}
__device__ __forceinline__ void finalize_body(const FinalizeArgs& a, const int t, double* fin_lds,
                                              const double* keep = nullptr) {
  if (a.use_lds) {
    if (a.rooted) finalize_body_t<true, true>(a, t, fin_lds, keep);
    else finalize_body_t<true, false>(a, t, fin_lds, keep);
  } else {
    if (a.rooted) finalize_body_t<false, true>(a, t, fin_lds, keep);
    else finalize_body_t<false, false>(a, t, fin_lds, keep);
  }
}

__global__ __launch_bounds__(64) void finalize_kernel(FinalizeArgs a) {
  extern __shared__ double fin_lds[];
  finalize_body(a, blockIdx.x, fin_lds);
}

// Both in one launch, a workgroup per tree, for calls with ONE evaluation per tree (JC69-type
// models, the analytic GTR gradient; rooted or not): the four waves sum the tiles, then wave 0
// alone goes on to the finalize step with what the workgroup has just written (one dispatch
// less on the latency path of small batches: ~4 us of a 125-tree step).
__global__ __launch_bounds__(256) void reduce_finalize_kernel(ReduceArgs ra, FinalizeArgs fa) {
  extern __shared__ double rf_lds[];
  // (the one-launch call's hand-off word of this tree: every walk wave that polled it is done)
  if (threadIdx.x == 0 && fa.clear_ready) fa.clear_ready[(size_t)blockIdx.x * kReadyStride] = 0;
  // (the reduced sums stay in LDS, behind whatever either step needs of it)
  double* keep = rf_lds + ra.keep_offset;
  reduce_tiles_body(ra, blockIdx.x, rf_lds, keep);
  __threadfence_block();
  __syncthreads();
  if (threadIdx.x >= 64) return;  // (ended waves do not take part in later barriers)
  finalize_body(fa, blockIdx.x, rf_lds, keep);
}

}  // namespace

// ------------------------------------------------------------------------
// Launch wrappers
// ------------------------------------------------------------------------
void launch_subst_gradient(const SubstGradArgs& a, hipStream_t s) {
  if (a.T <= 0) return;
  hipLaunchKernelGGL(subst_gradient_kernel, dim3((a.T + 63) / 64), dim3(64), 0, s, a);
}
bool reduce_tiles_fits(int N) {
  return sizeof(double) * 4 * (size_t)(3 * N + 12 + kSubstExtra) <= 64 * 1024;
}
void launch_reduce_tiles(const ReduceArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  const size_t W = a.g_width ? a.g_width : 2 * (size_t)a.N;
  hipLaunchKernelGGL(reduce_tiles_kernel, dim3(a.E), dim3(256), sizeof(double) * 4 * W, s, a);
}
static size_t finalize_lds_bytes(const FinalizeArgs& a) {
  return sizeof(double) * (a.rooted ? 22 * (size_t)a.n : 6 * (size_t)a.n);
}
void launch_reduce_finalize(const ReduceArgs& ra_in, const FinalizeArgs& fa_in, hipStream_t s) {
  FinalizeArgs fa = fa_in;
  ReduceArgs ra = ra_in;
  const size_t W = ra.g_width ? ra.g_width : 2 * (size_t)ra.N;
  const size_t fin = finalize_lds_bytes(fa);
  fa.use_lds = fin <= 48 * 1024;
  const size_t work = std::max(sizeof(double) * 4 * W, fa.use_lds ? fin : 0);
  ra.keep_offset = (int)(work / sizeof(double));
  const size_t lds = work + sizeof(double) * (2 * (size_t)ra.N + 1);
  allow_large_lds(reinterpret_cast<const void*>(reduce_finalize_kernel), lds);
  hipLaunchKernelGGL(reduce_finalize_kernel, dim3(fa.T), dim3(256), lds, s, ra, fa);
}
void launch_finalize(const FinalizeArgs& a_in, hipStream_t s) {
  FinalizeArgs a = a_in;
  // 6n of working set (+ for a rooted tree its staged state, 3N + 2n, and 8n of
  // coefficients and partial results)
  const size_t lds = sizeof(double) * (a.rooted ? 22 * (size_t)a.n : 6 * (size_t)a.n);
  a.use_lds = lds <= 48 * 1024;
  hipLaunchKernelGGL(finalize_kernel, dim3(a.T), dim3(64), a.use_lds ? lds : 0, s, a);
}

}  // namespace miphylo
