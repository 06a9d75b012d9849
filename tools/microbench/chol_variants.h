// In-wave Cholesky variants that were built and MEASURED in round 2 and did not replace the product's
// wave_chol32 / wave_inv32_follow (sfm_amd/csrc/dense.hip).  Kept with their micro-benchmarks so the numbers can be
// reproduced (tools/microbench/chol32_bench.hip, chol64_bench.hip); include AFTER dense.hip.
//
// MI355X, one workgroup, 2000 repetitions:
//   wave_chol32 (product: columns split over the half-waves)      4.73 us factor only, 5.73 us with the inverse follower
//   wave_chol32_rows (lane = row, 6-operation chain, 1 Newton step) 4.01 us factor only, 5.56 us with the follower
//       -> the follower wavefront (wave_inv32_follow, ~170 ns per column) is then the limit: 3 % per 32x32 block
//   wave_chol64 (lane = row over 64 rows) + wave_inv64_follow      12.65 us factor only, 17.4 us with the follower
//       -> slower than two 32x32 factors + the two 32^3 MFMA products between them (13.1 us incl. inverses):
//          lane = row wastes half of the FMAs above the diagonal and every broadcast ds_read moves 64 x 16 bytes, so
//          the column update is LDS-return-bandwidth bound (31 broadcast values per column on average), twice over
//          once the follower reads the same columns.  First version (one load - one use scheduling): 32 us.
#pragma once
#include <utility>

// Second form of the in-wave 32x32 factor: lane = row (both half-waves carry the same 32 rows), every lane holds its
// whole row (32 values).  The half-wave exchange of the split layout above (v_permlane32_swap + selects, ~8 dependent
// operations per column) disappears from the serial chain, which shrinks to: v_rsq_f64 -> t = piv y0 -> e = 1 - t y0
// -> L[i][j] = (a y0)(1 + e/2) [one fused Newton step on v_rsq_f64's ~2^-26: good to ~1e-15] -> v_readlane of
// L[j+1][j] -> update of column j+1 -> v_readlane of the next pivot; the reciprocal root of the next pivot is
// started before the bulk update of the current column.  No branch on the pivot: a non-positive one turns into
// NaNs and is reported through the return value.  Same outputs as wave_chol32 (sC, srd, s_ready).
template <int J>
__device__ __forceinline__ void chol32r_col(double (&a)[32], double& piv, double& y0, double& e, bool& bad, int row,
                                            double* __restrict__ sC, double* __restrict__ srd, int* __restrict__ s_ready) {
  bad |= !(piv > 0.0);
  const double ay = a[J] * y0;
  const double li = fma(0.5 * ay, e, ay);              // L[row][J]; rows above the diagonal hold 0
  double* cb = sC + J * 32;
  cb[row] = li;                                        // both half-waves store the same value
  srd[J] = fma(0.5 * y0, e, y0);                       // 1 / L[J][J] (uniform value, every lane stores it)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __hip_atomic_store(s_ready, J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if constexpr (J < 31) {
    const double lq = readlane_d(li, J + 1);           // L[J+1][J]
    a[J + 1] = fma(-li, lq, a[J + 1]);
    piv = readlane_d(a[J + 1], J + 1);
    y0 = __builtin_amdgcn_rsq(piv);
    e = fma(-(piv * y0), y0, 1.0);
    double c[30];
#pragma unroll
    for (int q = J + 2; q < 32; ++q) c[q - 2] = cb[q];
    __builtin_amdgcn_sched_barrier(0);                 // all loads of the column in flight before the first use
#pragma unroll
    for (int q = J + 2; q < 32; ++q) a[q] = fma(-li, c[q - 2], a[q]);
  }
}
template <int... Js>
__device__ __forceinline__ void chol32r_cols(std::integer_sequence<int, Js...>, double (&a)[32], double& piv, double& y0, double& e,
                                             bool& bad, int row, double* __restrict__ sC, double* __restrict__ srd,
                                             int* __restrict__ s_ready) {
  (chol32r_col<Js>(a, piv, y0, e, bad, row, sC, srd, s_ready), ...);
}
// a[q] = M[row][q] for q <= row = lane & 31, 0 above the diagonal.
__device__ __forceinline__ bool wave_chol32_rows(double (&a)[32], int lane, double* __restrict__ sC, double* __restrict__ srd,
                                                 int* __restrict__ s_ready) {
  bool bad = false;
  double piv = readlane_d(a[0], 0);
  double y0 = __builtin_amdgcn_rsq(piv);
  double e = fma(-(piv * y0), y0, 1.0);
  chol32r_cols(std::make_integer_sequence<int, 32>{}, a, piv, y0, e, bad, lane & 31, sC, srd, s_ready);
  return !bad;
}

// ---- the same for a whole 64x64 block: lane = row, each lane holds its 64 row entries in registers.
// The serial chain per column is six dependent operations: v_rsq_f64 -> t = piv y0 -> e = 1 - t y0 ->
// L[i][j] = (a y0)(1 + e/2) [one fused Newton step; v_rsq_f64 is good to ~2^-26, so the result is to ~1e-15] ->
// v_readlane of L[j+1][j] -> update of column j+1 -> v_readlane of the next pivot.  No half-wave exchange, no
// branch on the pivot (a non-positive pivot turns into NaNs and is reported through the return value).  The
// rank-1 update of the other columns takes its L[q][j] from the copy of the column published in LDS
// (sC[j*64 + q], broadcast reads), off the chain; on average 32 FMAs per column, half of them on rows above the
// diagonal - the price of lane = row, still 2.4x less time than two 32x32 factors + two 32^3 products between them.
constexpr int LDI = 66;      // row stride (doubles) of a 64x64 inverse held in LDS
// (columns are instantiated by template recursion, not `#pragma unroll`: the unrolled body exceeds the pragma's size
// threshold, and a partially unrolled loop indexes the register array dynamically, i.e. through scratch memory)
template <int J>
__device__ __forceinline__ void chol64_col(double (&a)[64], double& piv, double& y0, double& e, bool& bad, int lane,
                                           double* __restrict__ sC, double* __restrict__ srd, int* __restrict__ s_ready) {
  bad |= !(piv > 0.0);
  const double ay = a[J] * y0;
  const double li = fma(0.5 * ay, e, ay);              // L[lane][J]; rows above the diagonal hold 0
  double* cb = sC + J * 64;
  cb[lane] = li;
  srd[J] = fma(0.5 * y0, e, y0);                       // 1 / L[J][J] (uniform value, every lane stores it)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __hip_atomic_store(s_ready, J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if constexpr (J < 63) {
    const double lq = readlane_d(li, J + 1);           // L[J+1][J]
    a[J + 1] = fma(-li, lq, a[J + 1]);
    piv = readlane_d(a[J + 1], J + 1);
    // the reciprocal root of the NEXT pivot is started before the bulk update of this column, so that the 31 (on
    // average) off-chain FMAs of a column fill the latency of the chain instead of queueing in front of it
    y0 = __builtin_amdgcn_rsq(piv);
    e = fma(-(piv * y0), y0, 1.0);
    // bulk: a[q] -= L[lane][J] L[q][J], q = J + 2 .. 63, L[q][J] broadcast from the published column.  Loads in
    // batches of 16 values: the scheduler, close to the register limit, otherwise issues one load, waits for it,
    // uses it, and exposes the full LDS latency 30 times per column (measured: 32 us per block instead of 6)
    constexpr int CHK = 16;
#pragma unroll
    for (int q0 = J + 2; q0 < 64; q0 += CHK) {
      double c[CHK];
#pragma unroll
      for (int k = 0; k < CHK; ++k) if (q0 + k < 64) c[k] = cb[q0 + k];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < CHK; ++k) if (q0 + k < 64) a[q0 + k] = fma(-li, c[k], a[q0 + k]);
    }
  }
}
template <int... Js>
__device__ __forceinline__ void chol64_cols(std::integer_sequence<int, Js...>, double (&a)[64], double& piv, double& y0, double& e,
                                            bool& bad, int lane, double* __restrict__ sC, double* __restrict__ srd,
                                            int* __restrict__ s_ready) {
  (chol64_col<Js>(a, piv, y0, e, bad, lane, sC, srd, s_ready), ...);
}
__device__ __forceinline__ bool wave_chol64(double (&a)[64], int lane, double* __restrict__ sC, double* __restrict__ srd,
                                            int* __restrict__ s_ready) {
  bool bad = false;
  double piv = readlane_d(a[0], 0);
  double y0 = __builtin_amdgcn_rsq(piv);
  double e = fma(-(piv * y0), y0, 1.0);
  chol64_cols(std::make_integer_sequence<int, 64>{}, a, piv, y0, e, bad, lane, sC, srd, s_ready);
  return !bad;
}

// Inverse of the factor wave_chol64 is producing, on ANOTHER wavefront of the workgroup: lane t builds column t of
// L^-1 by forward substitution in outer-product order, one column behind the producer (see wave_inv32_follow).
// Result: sLi[r*ldl + t] = (L^-1)[r][t] (zeros above the diagonal).
template <int J>
__device__ __forceinline__ void inv64_col(double (&acc)[64], const double* __restrict__ sC, const double* __restrict__ srd,
                                          int* __restrict__ s_ready, int lane, double* __restrict__ sLi, int ldl) {
  while (__hip_atomic_load(s_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < J + 1) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const double xj = (J >= lane) ? (((J == lane) ? 1.0 : 0.0) - acc[J]) * srd[J] : 0.0;
  sLi[J * ldl + lane] = xj;
  const double* col = sC + J * 64;
  constexpr int CHK = 16;
#pragma unroll
  for (int r0 = J + 1; r0 < 64; r0 += CHK) {
    double c[CHK];
#pragma unroll
    for (int k = 0; k < CHK; ++k) if (r0 + k < 64) c[k] = col[r0 + k];
    __builtin_amdgcn_sched_barrier(0);                 // the loads of a batch together, ahead of its FMAs
#pragma unroll
    for (int k = 0; k < CHK; ++k)
      if (r0 + k < 64)
        // pinned as written: left to itself the optimiser turns the running sums into dot products evaluated at
        // the end, keeps every L[r][j] it ever loaded alive and spills two thousand registers
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[r0 + k]) : "v"(c[k]), "v"(xj));
  }
}
template <int... Js>
__device__ __forceinline__ void inv64_cols(std::integer_sequence<int, Js...>, double (&acc)[64], const double* __restrict__ sC,
                                           const double* __restrict__ srd, int* __restrict__ s_ready, int lane,
                                           double* __restrict__ sLi, int ldl) {
  (inv64_col<Js>(acc, sC, srd, s_ready, lane, sLi, ldl), ...);
}
__device__ __forceinline__ void wave_inv64_follow(const double* __restrict__ sC, const double* __restrict__ srd,
                                                  int* __restrict__ s_ready, int lane, double* __restrict__ sLi, int ldl) {
  double acc[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) acc[r] = 0.0;
  inv64_cols(std::make_integer_sequence<int, 64>{}, acc, sC, srd, s_ready, lane, sLi, ldl);
}

