// Micro-benchmark of the in-wave 32x32 factor chain (wave_chol32 / wave_inv32_follow of dense.hip).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form \
//     -Iinclude -Isfm_amd/csrc tools/microbench/chol32_bench.hip -o /tmp/chol32_bench && /tmp/chol32_bench
#include "../../sfm_amd/csrc/dense.hip"
#include "chol_variants.h"
#include <vector>
#include <cmath>

template <int MODE>   // 0: factor + follower, 1: factor only; 3 / 4: the same with the lane = row factor (wave_chol32_rows)
__global__ __launch_bounds__(128) void k_bench(const double* __restrict__ Ain, double* __restrict__ out, int reps) {
  __shared__ double sC[1024], srd[32], sLi[32 * LDL];
  __shared__ int s_ready;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  double keep = 0.0;
  for (int r = 0; r < reps; ++r) {
    if (tid == 0) s_ready = 0;
    __syncthreads();
    if (w == 0 && MODE >= 3) {
      double a[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) a[q] = (q <= i) ? Ain[i * 32 + q] : 0.0;
      wave_chol32_rows(a, lane, sC, srd, &s_ready);
      keep += a[31];
    } else if (w == 0) {
      double a[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) { const int q = 2 * t + h; a[t] = (q <= i) ? Ain[i * 32 + q] : 0.0; }
      wave_chol32(a, lane, sC, srd, &s_ready);
      keep += a[15];
    } else if (MODE == 0 || MODE == 3) {
      wave_inv32_follow(sC, srd, &s_ready, lane, sLi, LDL);
    }
    __syncthreads();
  }
  if (w == 0) {
    out[lane] = keep + sC[lane] + ((MODE == 0 || MODE == 3) ? sLi[lane] : 0.0);
    for (int e2 = lane; e2 < 1024; e2 += 64) { out[64 + e2] = sC[e2]; out[64 + 1024 + e2] = sLi[(e2 >> 5) * LDL + (e2 & 31)]; }
  }
}

int main() {
  const int n = 32;
  std::vector<double> A(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? 40.0 : 0.0) + 1.0 / (1.0 + std::abs(i - j));
  double *dA, *dout;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dout, (64 + 2048) * 8);
  hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 2000;
  auto run = [&](const char* name, auto kern) {
    hipLaunchKernelGGL(kern, dim3(1), dim3(128), 0, 0, dA, dout, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(1), dim3(128), 0, 0, dA, dout, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f us per 32x32 block\n", name, ms * 1e3 / reps);
  };
  std::vector<double> Lr(n * n, 0.0);
  for (int j = 0; j < n; ++j) {
    double sdiag = A[j * n + j];
    for (int k = 0; k < j; ++k) sdiag -= Lr[j * n + k] * Lr[j * n + k];
    Lr[j * n + j] = std::sqrt(sdiag);
    for (int i = j + 1; i < n; ++i) { double t = A[i * n + j]; for (int k = 0; k < j; ++k) t -= Lr[i * n + k] * Lr[j * n + k]; Lr[i * n + j] = t / Lr[j * n + j]; }
  }
  auto check = [&](const char* name) {
    std::vector<double> o(64 + 2048);
    hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
    double eL = 0, eI = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) eL = std::fmax(eL, std::fabs(o[64 + j * 32 + i] - Lr[i * n + j]));
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
      double sum = 0; for (int k = j; k <= i; ++k) sum += o[64 + 1024 + i * 32 + k] * Lr[k * n + j];
      eI = std::fmax(eI, std::fabs(sum - (i == j ? 1.0 : 0.0)));
    }
    printf("   %s: max |L - Lref| = %.3e, max |Li L - I| = %.3e\n", name, eL, eI);
  };
  run("factor + pipelined inverse", k_bench<0>); check("split layout");
  run("factor only", k_bench<1>);
  run("rows: factor + inverse", k_bench<3>); check("lane = row");
  run("rows: factor only", k_bench<4>);
  return 0;
}
