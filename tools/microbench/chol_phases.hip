// Phase timestamps (s_memrealtime, 100 MHz) of the serial chain of the blocked Cholesky (dense.hip): where the
// ~30 us of a 64-column step go.  Runs dense_cholesky on a random SPD matrix and prints the phase deltas of the
// LAST k_chol_step launch of the critical workgroup (block 0), plus event-timed totals.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form \
//     -Iinclude -Isfm_amd/csrc tools/microbench/chol_phases.hip sfm_amd/csrc/ctx.hip -o /tmp/chol_phases && /tmp/chol_phases
#define SFM_DENSE_PHASE_TIMING 1
#include "../../sfm_amd/csrc/dense.hip"
#include <vector>
#include <cmath>
#include <random>

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2000;
  std::vector<double> A((size_t)(n + 1) * n);
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nd;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
    const double v = (i == j) ? 50.0 + std::abs(nd(rng)) : 0.3 * nd(rng) / (1.0 + std::abs(i - j) * 0.05);
    A[(size_t)i * n + j] = v; A[(size_t)j * n + i] = v;
  }
  for (int j = 0; j < n; ++j) A[(size_t)n * n + j] = nd(rng);
  sfm_handle h; if (sfm_create(0, &h) != 0) { printf("no device\n"); return 1; }
  double *dA, *dA0, *base;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dA0, A.size() * 8);
  hipMemcpy(dA0, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMalloc(&base, (size_t)dense_ws_doubles(n) * 8);
  DenseWs w; dense_ws_carve(base, n, &w);
  hipMemset(w.flag, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0, tot = 0;
  const int reps = 10;
  for (int r = 0; r < reps + 2; ++r) {
    hipMemcpy(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    dense_cholesky(h, dA, n, n + 1, w);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2) tot += ms;
  }
  printf("dense_cholesky n=%d: %.1f us\n", n, tot / reps * 1e3);
  unsigned long long t[32];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(g_phase_t), sizeof(t));
  const char* names[16] = {"kernel entry", "D staged", "trsm done", "product done", "", "", "", "", "crit start", "chol32 #1 (wave 0)",
                           "P1 done (incl. inverse)", "P2 + L11 stores", "P3", "P4 done", "final stores issued", ""};
  // the last launch with trailing columns is not the last launch overall (that one only solves the bordered row):
  // slots 0..2 come from the final launch, 3..14 from the one before; print raw deltas within each group
  for (int k : {1, 2}) printf("  %-28s +%6.2f us\n", names[k], (double)(t[k] - t[k - 1]) * 0.01);
  printf("  (previous launch) product -> crit start  %6.2f us\n", (double)(t[8] - t[3]) * 0.01);
  for (int k = 9; k <= 14; ++k) printf("  %-28s +%6.2f us\n", names[k], (double)(t[k] - t[k == 10 ? 8 : k - 1]) * 0.01);
  int fail = 0; hipMemcpy(&fail, w.flag, 4, hipMemcpyDeviceToHost);
  printf("fail flag %d\n", fail);
  return 0;
}
