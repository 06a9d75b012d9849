// Micro-benchmark + accuracy check of the in-wave 64x64 factor (wave_chol64) and its pipelined inverse
// (wave_inv64_follow) of dense.hip.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form \
//     -Iinclude -Isfm_amd/csrc tools/microbench/chol64_bench.hip -o /tmp/chol64_bench && /tmp/chol64_bench
#include "../../sfm_amd/csrc/dense.hip"
#include "chol_variants.h"
#include <vector>
#include <cmath>
#include <random>

template <int MODE>   // 0: factor + follower, 1: factor only, 2: follower only (never finishes a column: timing is meaningless, build check only)
__global__ __launch_bounds__(256) void k_bench(const double* __restrict__ Ain, double* __restrict__ Lout, double* __restrict__ Liout,
                                               int reps) {
  __shared__ double sC[64 * 64], srd[64], sLi[64 * LDI];
  __shared__ int s_ready;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  double keep = 0.0;
  for (int r = 0; r < reps; ++r) {
    if (tid == 0) s_ready = 0;
    __syncthreads();
    if (w == 0 && MODE != 2) {
      double a[64];
#pragma unroll
      for (int q = 0; q < 64; ++q) a[q] = (q <= lane) ? Ain[lane * 64 + q] : 0.0;
      wave_chol64(a, lane, sC, srd, &s_ready);
      keep += a[63];
    } else if (w == 1 && MODE != 1) {
      wave_inv64_follow(sC, srd, &s_ready, lane, sLi, LDI);
    }
    __syncthreads();
  }
  if (w == 0) {
    for (int e = lane; e < 64 * 64; e += 64) { Lout[e] = sC[e]; if (MODE == 0) Liout[e] = sLi[(e >> 6) * LDI + (e & 63)]; }
    Lout[64 * 64] = keep;
  }
}

int main() {
  const int n = 64;
  std::vector<double> A(n * n), L(n * n, 0.0);
  std::mt19937_64 rng(3); std::normal_distribution<double> nd;
  std::vector<double> R(n * n);
  for (auto& v : R) v = nd(rng);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    double s = (i == j) ? 8.0 : 0.0;
    for (int k = 0; k < n; ++k) s += R[i * n + k] * R[j * n + k];
    A[i * n + j] = s;
  }
  for (int j = 0; j < n; ++j) {          // reference Cholesky (row-major lower)
    double s = A[j * n + j];
    for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
    L[j * n + j] = std::sqrt(s);
    for (int i = j + 1; i < n; ++i) {
      double t = A[i * n + j];
      for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / L[j * n + j];
    }
  }
  double *dA, *dL, *dLi;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dL, (n * n + 8) * 8); hipMalloc(&dLi, n * n * 8);
  hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 2000;
  auto run = [&](const char* name, auto kern) {
    hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, 0, dA, dL, dLi, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, 0, dA, dL, dLi, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f us per 64x64 block\n", name, ms * 1e3 / reps);
  };
  run("factor only", k_bench<1>);
  run("factor + pipelined inverse", k_bench<0>);
  std::vector<double> gL(n * n), gLi(n * n);
  hipMemcpy(gL.data(), dL, n * n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(gLi.data(), dLi, n * n * 8, hipMemcpyDeviceToHost);
  double errL = 0, maxL = 0, errI = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
    errL = std::fmax(errL, std::fabs(gL[j * 64 + i] - L[i * n + j]));      // sC is column-major
    maxL = std::fmax(maxL, std::fabs(L[i * n + j]));
  }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {                 // Li * L = I
    double s = 0; for (int k = 0; k < n; ++k) s += ((k <= i) ? gLi[i * n + k] : 0.0) * ((j <= k) ? L[k * n + j] : 0.0);
    errI = std::fmax(errI, std::fabs(s - (i == j ? 1.0 : 0.0)));
  }
  printf("max |L - Lref| / max|L| = %.3e   max |Li L - I| = %.3e\n", errL / maxL, errI);
  return 0;
}
