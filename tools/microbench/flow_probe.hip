// Feasibility probe for a persistent "chain" workgroup on one stream exchanging flags with per-step grid kernels on
// another stream (the protocol a persistent Cholesky chain would use).  Prints the time per step.
// hipcc -O3 --offload-arch=gfx950 tools/microbench/flow_probe.hip -o /tmp/flow_probe && /tmp/flow_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ bool wait_ge(int* f, int v, int* abort_flag) {
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > 5000000ull) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }  // 50 ms
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
  }
  return true;
}

__global__ void k_chain(int* flags, int steps, int work_cycles) {   // flags[0] = published by chain, flags[1] = abort, flags[2+i] = step i tiles done
  for (int i = 0; i < steps; ++i) {
    if (threadIdx.x == 0 && i >= 1) wait_ge(&flags[2 + i - 1], 2, &flags[1]);
    __syncthreads();
    const unsigned long long t0 = clock64();
    while (clock64() - t0 < (unsigned long long)work_cycles) {}
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&flags[0], i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // D_{i+1} (D_0 came from the first-block kernel)
  }
}
__global__ void k_rest(int* flags, int step, int work_cycles) {
  if (threadIdx.x == 0) wait_ge(&flags[0], step + 1, &flags[1]);
  __syncthreads();
  const unsigned long long t0 = clock64();
  while (clock64() - t0 < (unsigned long long)work_cycles) {}
  __syncthreads();
  if (threadIdx.x == 0 && (blockIdx.x == 1 || blockIdx.x == 2)) {
    __threadfence();
    __hip_atomic_fetch_add(&flags[2 + step], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  const int steps = 32;
  int* flags; hipMalloc(&flags, 4096);
  hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStreamCreateWithPriority(&b, hipStreamNonBlocking, lo);
  hipEvent_t e0, e1, t0, t1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&t0); hipEventCreate(&t1);
  for (int rep = 0; rep < 4; ++rep) {
    for (int grid : {256, 528}) {
      hipMemsetAsync(flags, 0, 4096, a);
      hipMemsetD32Async((hipDeviceptr_t)flags, 1, 1, a);     // D_0 is available before anything starts
      hipEventRecord(t0, a);
      hipEventRecord(e0, a); hipStreamWaitEvent(b, e0, 0);
      hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), 0, a, flags, steps, 36000);        // ~15 us of chain work per step
      for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(k_rest, dim3(grid), dim3(256), 102400, b, flags, i, 19000);   // ~8 us tiles
      hipEventRecord(e1, b); hipStreamWaitEvent(a, e1, 0);
      hipEventRecord(t1, a); hipEventSynchronize(t1);
      float ms; hipEventElapsedTime(&ms, t0, t1);
      int hf[4]; hipMemcpy(hf, flags, 16, hipMemcpyDeviceToHost);
      printf("grid %3d: %.1f us per step (chain work 15 us), published %d, abort %d\n", grid, ms * 1e3 / steps, hf[0], hf[1]);
    }
  }
  return 0;
}
