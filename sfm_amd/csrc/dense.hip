// Dense SPD solver for the reduced camera system (n = n_cams * cam_dim, a few thousand): bordered
// lower Cholesky + triangular solves, fp64, gfx950.
//
// The factorisation is latency-bound (n/64 dependent steps), so the design minimises the work on that chain and
// the number of kernel boundaries (phase timings: tools/microbench/chol_phases.hip):
//   wave_chol32    : 32x32 diagonal block in ONE wavefront (lane = row, columns split over the two
//                    half-waves, finished column broadcast through LDS, 1/sqrt by v_rsq_f64 + two Newton
//                    steps instead of sqrt + divide)
//   wave_inv32_follow : its inverse on a SECOND wavefront of the same workgroup, in outer-product order one
//                    column behind the factor (producer/consumer through LDS): +0.6 us on the chain
//   crit64_lite    : a 64x64 diagonal block = two of those + two 32^3 MFMA products; only the two 32x32
//                    inverses are formed on the chain
//   k_chol_step    : ONE launch per 64-column step: every trailing tile re-solves its own panel rows against
//                    the step data [Li11 0; L21 Li22] (trsm_rows16, MFMA), applies the rank-64 update, and the
//                    workgroup of tile (0,0) factors the next diagonal block right away (LOOK-AHEAD).  The
//                    factor is written to a separate matrix Lm; the panel columns of A stay read-only.
//   k_syrk_lower   : systems from n = 4096 work in 256-column strips: the steps of a strip only update the strip's
//                    columns, this kernel applies the strip's rank-256 update to the rest (4x the flops per byte)
//   k_inv64_fix / k_inv_merge : explicit inverses of the 128x128 diagonal blocks of L (both layouts), so that
//   k_trsv_flow    : a whole triangular solve is one launch of n/128 workgroups handing their 128 unknowns
//                    on through the output vector itself (k_trsv_step: one launch per block, for n > 16384).
#include "dense.h"
#include <cstdlib>

typedef double v4d __attribute__((ext_vector_type(4)));

// Phase timestamps of the serial chain (tools/microbench/chol_phases.hip defines SFM_DENSE_PHASE_TIMING); no-ops
// in the library build.
#ifdef SFM_DENSE_PHASE_TIMING
__device__ unsigned long long g_phase_t[32];
#define PHASE_T(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_phase_t[k] = wall_clock64(); } while (0)
#else
#define PHASE_T(k) do { } while (0)
#endif

__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

int64_t dense_ws_doubles(int n) {
  const int64_t nb = (n + 127) / 128;
  return 2 * 64 * 64 + 2 * nb * 128 * 128 + nb * 2 * 64 * 64 + 32 + align_up((int64_t)(n + 1) * n, 32) + (int64_t)n * n;
}
int64_t dense_ws_lm_offset(int n) {
  const int64_t nb = (n + 127) / 128;
  return 2 * 64 * 64 + 2 * nb * 128 * 128 + nb * 2 * 64 * 64 + 32;
}
void dense_ws_carve(double* base, int n, DenseWs* w) {
  const int64_t nb = (n + 127) / 128;
  double* p = base;
  w->Ld = p; p += 2 * 64 * 64;
  w->Dinv = p; p += nb * 128 * 128;
  w->DinvT = p; p += nb * 128 * 128;
  w->inv64 = p; p += nb * 2 * 64 * 64;
  w->flag = (int*)p; p += 32;
  w->Lm = p; p += align_up((int64_t)(n + 1) * n, 32);      // = base + dense_ws_lm_offset(n)
  w->LmT = p;
}

// ------------------------------------------------------------------------------------ Cholesky
// Value of the half-wave `hj` broadcast to both half-waves (lane i and lane i + 32 get lane (i + 32 hj)'s v):
// v_permlane32_swap of a register with itself leaves {lower half twice, upper half twice}.
__device__ __forceinline__ double half_bcast(double v, int hj) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto rlo = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto rhi = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)(hj ? rhi[1] : rhi[0]), (int)(hj ? rlo[1] : rlo[0]));
}

// In-wave factorisation of a 32x32 SPD block.  Lane l = (row i = l & 31, half h = l >> 5) holds the 16
// entries M[i][2t + h] of its row in a[t]: the two half-waves split the columns, so a rank-1 update costs
// <= 16 FMAs per lane.  The serial chain pivot -> 1/sqrt -> column -> next pivot stays in registers
// (v_permlane32_swap shares L[i][j] between the half-waves, v_readlane fetches L[j+1][j] for the next
// pivot's column); the other columns take their L[q][j] from the copy of the column published in LDS,
// sC[j*32 + i] = L[i][j] (same wavefront: LDS is in order), off the chain.  s_ready <- j + 1 after each
// column lets a SECOND wavefront build L^-1 one column behind (wave_inv32_follow).
// Returns false on a non-positive pivot.
__device__ __forceinline__ bool wave_chol32(double (&a)[16], int lane, double* __restrict__ sC,
                                            double* __restrict__ srd, int* __restrict__ s_ready) {
  const int i = lane & 31, h = lane >> 5;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int hj = j & 1, tj = j >> 1;
    double piv = readlane_d(a[tj], j + 32 * hj);
    if (!(piv > 0.0)) { bad = true; piv = 1.0; }
    const double rinv = rsqrt_nr(piv);
    const double li = half_bcast(a[tj] * rinv, hj);            // L[i][j] in both half-waves
    a[tj] = (h == hj) ? li : a[tj];
    double* cb = sC + j * 32;
    cb[i] = li;                                                // both halves store the same value
    srd[j] = rinv;                                             // uniform value, every lane stores it
    // Flag = relaxed workgroup-scope atomic (a plain store is deleted / its load hoisted out of the spin loop
    // by hipcc, a `volatile` one becomes flat_store sc0 sc1 with a full drain on the pivot chain).  The
    // wavefront-scope fence only pins the compiler's order; the LDS unit executes a wave's operations in order.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __hip_atomic_store(s_ready, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (j < 31) {                                              // next pivot's column first, from registers
      const int q = j + 1, hq = q & 1, tq = q >> 1;
      const double lq = readlane_d(li, q);                     // L[q][j]
      a[tq] = (h == hq) ? a[tq] - li * lq : a[tq];
    }
    if (hj == 0) {
      // j = 2 tj: column j + 1 (half 1, a[tj]) was the fast path; both halves continue at t = tj + 1
#pragma unroll
      for (int t = tj + 1; t < 16; ++t) a[t] -= li * cb[2 * t + h];
    } else {
      // j = 2 tj + 1: column j + 1 = 2 (tj + 1) (half 0) was the fast path; half 1 still owes column 2 tj + 3
      if (tj + 1 < 16) {
        const double upd = li * cb[2 * (tj + 1) + 1];
        a[tj + 1] = (h == 1) ? a[tj + 1] - upd : a[tj + 1];
      }
#pragma unroll
      for (int t = tj + 2; t < 16; ++t) a[t] -= li * cb[2 * t + h];
    }
  }
  return !bad;
}

// Inverse of the factor being produced by wave_chol32 in ANOTHER wavefront of the same workgroup: lane t
// (both half-waves alike) builds column t of L^-1 by forward substitution in OUTER-PRODUCT order: when column j
// of L is published, x[j] = (e_t[j] - acc[j]) / L[j][j] is final and acc[r] += L[r][j] x[j] (r > j) are
// independent FMAs, so the consumer is one multiply behind the producer's last column instead of a 31-long
// dependent chain.  The producer never waits for the consumer, so the spin cannot deadlock.
// Result: sLi[r*ldl + t] = (L^-1)[r][t].
__device__ __forceinline__ void wave_inv32_follow(const double* __restrict__ sC, const double* __restrict__ srd,
                                                  int* __restrict__ s_ready, int lane, double* __restrict__ sLi, int ldl) {
  // lane (t, h): column t of the inverse, rows r = 2u + h of the running sums acc (the two half-waves split the
  // rows, v_permlane32_swap hands x[j] from the half that owns row j to the other one)
  const int t = lane & 31, h = lane >> 5;
  double acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.0;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int hj = j & 1, uj = j >> 1;
    while (__hip_atomic_load(s_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < j + 1) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double mine = (j >= t) ? (((j == t) ? 1.0 : 0.0) - acc[uj]) * srd[j] : 0.0;     // valid in half hj
    const double xj = half_bcast(mine, hj);
    sLi[j * ldl + t] = xj;
    const double* col = sC + j * 32 + h;
    if (hj == 0) acc[uj] += (h == 1) ? col[2 * uj] * xj : 0.0;       // row j + 1 = 2 uj + 1 belongs to half 1
#pragma unroll
    for (int u = uj + 1; u < 16; ++u) acc[u] += col[2 * u] * xj;
  }
}

// ---- two-level factorisation + inversion of a 64x64 SPD tile held in LDS (all 256 threads of a workgroup)
// sM [64][LDM] row-major, lower part valid (identity outside the valid nb x nb corner).  On return
//   L  = [L11 0; L21 L22]:  L11 column-major in ... written to A by the caller-provided store lambda
// The tile is processed as 2x2 blocks of 32: wavefront 0 runs the serial in-wave factor (wave_chol32),
// wavefront 1 builds the inverse one column behind (wave_inv32_follow), all four waves do the 32^3 products
// on v_mfma_f64_16x16x4_f64 (one 16x16 tile each).  LDS carve (doubles), given base pointer `w`:
//   sM 64*66 | sC 1024 | sLi11 32*34 | sLi22 32*34 | sT 32*34 | srd 64 | flag (2 ints)
constexpr int LDM = 66, LDL = 34;
constexpr int CRIT64_DOUBLES = 64 * LDM + 1024 + 3 * 32 * LDL + 64 + 2;   // 8578 doubles; the update kernel stages 2 * 64 * LDM = 8448 in the same array
static_assert(CRIT64_DOUBLES + 6 >= 2 * 64 * LDM, "staging must fit the shared array");

struct Crit64 {
  double *sM, *sC, *sLi11, *sLi22, *sT, *srd;
  int* flag;
};
__device__ __forceinline__ Crit64 crit64_carve(double* base) {
  Crit64 c;
  c.sM = base; base += 64 * LDM;
  c.sC = base; base += 1024;
  c.sLi11 = base; base += 32 * LDL;
  c.sLi22 = base; base += 32 * LDL;
  c.sT = base; base += 32 * LDL;
  c.srd = base; base += 64;
  c.flag = (int*)base;
  return c;
}

// one 16x16 output tile (tr, tc) of a 32x32x32 product; operands in LDS.
//   A(r, k) = pa[r * lda + k];   B(k, c) = TRANSB ? pb[c * ldb + k] : pb[k * ldb + c]
template <bool TRANSB>
__device__ __forceinline__ v4d mm32_tile(const double* __restrict__ pa, int lda, const double* __restrict__ pb, int ldb,
                                         int tr, int tc, int lane) {
  const int r16 = lane & 15, kq = lane >> 4;
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k0 = 0; k0 < 32; k0 += 4) {
    const double a = pa[(tr * 16 + r16) * lda + k0 + kq];
    const double b = TRANSB ? pb[(tc * 16 + r16) * ldb + k0 + kq] : pb[(k0 + kq) * ldb + tc * 16 + r16];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// Factor the tile in c.sM and invert its two 32x32 diagonal blocks:  [A11 .; A21 A22] = [L11 0; L21 L22] [..]^T.
// store_L(r, c, v) receives every lower-triangular entry of L (64x64 coordinates), store_Li the entries of
// L11^-1 (r, c < 32) and L22^-1 (r, c >= 32), store_L21 the raw block L21 (r >= 32, c < 32).  The full 64x64
// inverse is NOT formed here: the trailing rows are solved in two 32-wide stages (trsm_rows16), which keeps the
// two products L21 L11^-1 and L22^-1 (...) off the serial chain; k_inv64_fix builds them later for the solves.
template <class FL, class FI, class F21>
__device__ __forceinline__ void crit64_lite(const Crit64& c, int tid, FL store_L, FI store_Li, F21 store_L21,
                                            int* __restrict__ fail) {
  const int lane = tid & 63, w = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int tr = w >> 1, tc = w & 1;
  const int kq = lane >> 4, r16 = lane & 15;
  if (tid == 0) { c.flag[0] = 0; c.flag[1] = 0; }
  __syncthreads();
  PHASE_T(8);
  if (w < 2) __builtin_amdgcn_s_setprio(3);     // the serial chain: ahead of the bulk tiles sharing these SIMDs
  // ---- P1: A11 = L11 L11^T (wave 0), Li11 = L11^-1 (wave 1, one column behind)
  if (w == 0) {
    double a[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { const int q = 2 * t + h; a[t] = (q <= i) ? c.sM[i * LDM + q] : 0.0; }
    if (!wave_chol32(a, lane, c.sC, c.srd, &c.flag[0]) && lane == 0) *fail = 1;
    PHASE_T(9);
  } else if (w == 1) {
    wave_inv32_follow(c.sC, c.srd, &c.flag[0], lane, c.sLi11, LDL);
  }
  __syncthreads();
  PHASE_T(10);
  // ---- P2: L21 = A21 Li11^T
  {
    const v4d acc = mm32_tile<true>(c.sM + 32 * LDM, LDM, c.sLi11, LDL, tr, tc, lane);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) c.sM[(32 + tr * 16 + kq + 4 * q) * LDM + tc * 16 + r16] = acc[q];
  }
  // L11 / Li11 leave LDS now: sC is reused by the second factor
  for (int e = tid; e < 32 * 32; e += 256) {
    const int r = e >> 5, cc = e & 31;
    if (cc <= r) { store_L(r, cc, c.sC[cc * 32 + r]); store_Li(r, cc, c.sLi11[r * LDL + cc]); }
  }
  __syncthreads();
  PHASE_T(11);
  // ---- P3: A22 -= L21 L21^T
  {
    const v4d u = mm32_tile<true>(c.sM + 32 * LDM, LDM, c.sM + 32 * LDM, LDM, tr, tc, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) c.sM[(32 + tr * 16 + kq + 4 * q) * LDM + 32 + tc * 16 + r16] -= u[q];
  }
  __syncthreads();
  PHASE_T(12);
  // ---- P4: A22 = L22 L22^T, Li22
  if (w == 0) {
    double a[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { const int q = 2 * t + h; a[t] = (q <= i) ? c.sM[(32 + i) * LDM + 32 + q] : 0.0; }
    if (!wave_chol32(a, lane, c.sC, c.srd + 32, &c.flag[1]) && lane == 0) *fail = 1;
  } else if (w == 1) {
    wave_inv32_follow(c.sC, c.srd + 32, &c.flag[1], lane, c.sLi22, LDL);
  }
  __syncthreads();
  PHASE_T(13);
  for (int e = tid; e < 32 * 32; e += 256) {
    const int r = e >> 5, cc = e & 31;
    const double l21 = c.sM[(32 + r) * LDM + cc];
    store_L(32 + r, cc, l21);
    store_L21(32 + r, cc, l21);
    if (cc <= r) { store_L(32 + r, 32 + cc, c.sC[cc * 32 + r]); store_Li(32 + r, 32 + cc, c.sLi22[r * LDL + cc]); }
  }
  PHASE_T(14);
}

// Step data of one diagonal block as the step kernel reads it: D [64][64] row-major =
// [Li11 0; L21 Li22] (32x32 quadrants; identity-padded when the block is short).
// Factor the first 64x64 diagonal block: L -> Lm, step data -> D, diagonal 32-block inverses -> inv64.
__device__ __forceinline__ void chol_diag_block(double* smem, const double* __restrict__ A, double* __restrict__ Lm, int n,
                                                int j, double* __restrict__ D, double* __restrict__ inv64,
                                                int* __restrict__ fail) {
  const Crit64 c = crit64_carve(smem);
  const int tid = threadIdx.x;
  const int nb = (n - j) < 64 ? (n - j) : 64;
  const double* Ab = A + (size_t)j * n + j;
  double* Lb = Lm + (size_t)j * n + j;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, cc = e & 63;
    c.sM[r * LDM + cc] = (r < nb && cc < nb) ? ((cc <= r) ? Ab[(size_t)r * n + cc] : 0.0) : ((r == cc) ? 1.0 : 0.0);
  }
  __syncthreads();
  crit64_lite(c, tid,
              [&](int r, int cc, double v) { if (r < nb && cc < nb) Lb[(size_t)r * n + cc] = v; },
              [&](int r, int cc, double v) { D[r * 64 + cc] = v; inv64[r * 64 + cc] = v; },
              [&](int r, int cc, double v) { D[r * 64 + cc] = v; }, fail);
}
// Factor the 64x64 diagonal block at (j, j) on its own (the first block; in the two-level scheme the first block
// of every 256-column strip): L -> Lm, step data -> D, diagonal 32-block inverses -> inv64 (block j / 64).
__global__ __launch_bounds__(256) void k_chol_diag(const double* __restrict__ A, double* __restrict__ Lm, int n, int j,
                                                   double* __restrict__ D, double* __restrict__ inv64,
                                                   int* __restrict__ fail) {
  __shared__ double smem[CRIT64_DOUBLES + 6];     // static: with `extern __shared__` hipcc needs 256 + 68 registers here
  chol_diag_block(smem, A, Lm, n, j, D, inv64, fail);
}

struct TrsmIn {
  double a1[8];        // A1 as MFMA A operand: row r16, k = 4 s + kq
  double a2c[2][4];    // A2 in the result layout: row kq + 4 i, column 32 + 16 t + r16
};
// the global loads of trsm_rows16, separate so that they are in flight together with the step-data staging
__device__ __forceinline__ void trsm_load(const double* A, int n, int nrows, int j0, int nb, int row, int lane, TrsmIn& in) {
  const int r16 = lane & 15, kq = lane >> 4;
  const int arow = row + r16;
  const bool aok = arow < nrows;
  const double* ap = A + (size_t)arow * n + j0;
#pragma unroll
  for (int s4 = 0; s4 < 8; ++s4) in.a1[s4] = (aok && (4 * s4 + kq) < nb) ? ap[4 * s4 + kq] : 0.0;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gr = row + kq + 4 * i, col = 32 + 16 * t + r16;
      in.a2c[t][i] = (gr < nrows && col < nb) ? A[(size_t)gr * n + j0 + col] : 0.0;
    }
}
__device__ __forceinline__ void trsm_rows16(const TrsmIn& in, const double* __restrict__ sD, double* __restrict__ dst,
                                            int lane, v4d (&x)[4]) {
  const int r16 = lane & 15, kq = lane >> 4;
  const double (&a1)[8] = in.a1;
  const double (&a2c)[2][4] = in.a2c;
#pragma unroll
  for (int t = 0; t < 4; ++t) x[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  // stage 1: Li11 is lower triangular, column tile t only has k <= 16 t + 15
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const double* bp = sD + (t * 16 + r16) * LDM + kq;
#pragma unroll
    for (int s4 = 0; s4 < 4 * (t + 1); ++s4) x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s4], bp[4 * s4], x[t], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[(kq + 4 * i) * LDM + t * 16 + r16] = x[t][i];
  }
  __syncthreads();
  // stage 2a: A2' = A2 - X1 L21^T
  {
    double xa[8];
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) xa[s4] = dst[r16 * LDM + 4 * s4 + kq];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      v4d u = {0.0, 0.0, 0.0, 0.0};
      const double* bp = sD + (32 + t * 16 + r16) * LDM + kq;
#pragma unroll
      for (int s4 = 0; s4 < 8; ++s4) u = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s4], bp[4 * s4], u, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[(kq + 4 * i) * LDM + 32 + t * 16 + r16] = a2c[t][i] - u[i];
    }
  }
  __syncthreads();
  // stage 2b: X2 = A2' Li22^T
  {
    double xa[8];
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) xa[s4] = dst[r16 * LDM + 32 + 4 * s4 + kq];
    __syncthreads();                              // every lane has its operands before the tile is overwritten
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const double* bp = sD + (32 + t * 16 + r16) * LDM + 32 + kq;
#pragma unroll
      for (int s4 = 0; s4 < 4 * (t + 1); ++s4)
        x[2 + t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s4], bp[4 * s4], x[2 + t], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[(kq + 4 * i) * LDM + 32 + t * 16 + r16] = x[2 + t][i];
    }
  }
  __syncthreads();
}

// One step of the blocked factorisation = ONE kernel: for the 64-column panel at j0 (diagonal block already
// factored, its step data in D) every lower-triangular 64x64 tile (ti, tj) of the trailing matrix
//   1. solves its own panel rows  X_I = A[I, j0:j1] L_jj^-T  and X_J (trsm_rows16; recomputing them per tile
//      costs a few microseconds of MFMA time and removes the separate panel kernel from the serial chain),
//   2. tiles of the first tile column (tj == 0) write X_I to the factor Lm,
//   3. A[I, J] -= X_I X_J^T  (rank-64 update from LDS, row stride 66 doubles = conflict-free ds_read_b64),
//   4. LOOK-AHEAD: the workgroup of tile (0,0) then factors the next diagonal block in LDS (crit64_lite) and
//      publishes its step data to Dn for the next launch, so the serial chain never waits for a kernel of its own.
// A's panel columns are only read here (never overwritten), which is what makes step 1 race-free.
constexpr int CHOL_STEP_SMEM = CRIT64_DOUBLES + 6 + 64 * LDM;     // 102.5 KB static (gfx950 allows up to 160 KB)
constexpr int CHOL_STRIP_MIN_N = 4096;       // systems at least this large use the two-level (strip + rank-256 update) scheme
// tile b of the step at panel j0 (see k_chol_step); all 256 threads of the workgroup, smem = CHOL_STEP_SMEM doubles
// col_end < n (two-level scheme): only the trailing columns [j1, col_end) of the current 256-column strip are
// updated (tiles enumerated column by column); the rest waits for the strip's rank-256 update (k_syrk_lower).
__device__ __forceinline__ void chol_tile(double* smem, int b, bool stage_d, double* A, double* __restrict__ Lm,
                                          double* __restrict__ LmT, int n, int nrows, int j0, int col_end, const double* D,
                                          double* Dn, double* inv64_next, int* __restrict__ fail) {
  double* sI = smem;                 // [64][LDM]
  double* sJ = smem + 64 * LDM;      // [64][LDM]   (the factor's buffers later reuse sI/sJ)
  double* sD = smem + CRIT64_DOUBLES + 6;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nb = (n - j0) < 64 ? (n - j0) : 64;
  const int j1 = j0 + nb;
  const int rem_r = nrows - j1, rem_c = col_end - j1;
  int ti, tj;
  if (rem_c <= 0) {                     // no columns to update: one tile per 64 rows, panel solve only
    ti = b; tj = 0;
  } else if (col_end >= n) {
    // 1-D order over the lower-triangular tiles only: b -> (ti, tj), tj <= ti, b = ti (ti + 1) / 2 + tj
    ti = (int)((sqrtf(8.0f * (float)b + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
    while (ti * (ti + 1) / 2 > b) --ti;
    tj = b - ti * (ti + 1) / 2;
  } else {
    // strip: column tj holds the tiles ti = tj .. T - 1, columns one after the other (at most 4 of them)
    const int T = (rem_r + 63) / 64;
    tj = 0;
    int rest = b;
    while (rest >= T - tj) { rest -= T - tj; ++tj; }
    ti = tj + rest;
  }
  const int I0 = ti * 64, J0 = tj * 64;
  const int col16 = lane & 15, rq = lane >> 4;
  // prefetch the C tile this lane updates (rows 16 w + (lane>>4) + 4 i, cols 16 t + (lane&15))
  double cv[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gr = I0 + w * 16 + rq + 4 * i, gcol = J0 + t * 16 + col16;
      cv[t][i] = (gr < rem_r && gcol < rem_c && gcol <= gr) ? A[(size_t)(j1 + gr) * n + j1 + gcol] : 0.0;
    }
  PHASE_T(0);
  TrsmIn inI, inJ;
  trsm_load(A, n, nrows, j0, nb, j1 + I0 + w * 16, lane, inI);
  if (ti != tj && rem_c > 0) trsm_load(A, n, nrows, j0, nb, j1 + J0 + w * 16, lane, inJ);
  if (stage_d)
    for (int e = tid; e < 64 * 64; e += 256) sD[(e >> 6) * LDM + (e & 63)] = D[e];
  __syncthreads();
  PHASE_T(1);
  v4d xi[4], xj[4];
  trsm_rows16(inI, sD, sI + w * 16 * LDM, lane, xi);
  if (tj == 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gr = j1 + I0 + w * 16 + rq + 4 * i, col = t * 16 + col16;
        if (gr < nrows && col < nb) Lm[(size_t)gr * n + j0 + col] = xi[t][i];
      }
  }
  const bool has_tile10 = rem_c > 0 && rem_r > 64;         // tile (1,0) exists in this launch
  if (tj == 0 && !(ti == 0 && has_tile10)) {
    // transposed copy for the forward solve, from the LDS tile so that the stores run along rows of LmT.  The
    // rows of tile (0,0) - the workgroup that carries the serial chain - are written by tile (1,0), which has
    // the same rows as its X_J
    for (int e = tid; e < 64 * 64; e += 256) {
      const int col = e >> 6, r = e & 63;
      const int gr = j1 + I0 + r;
      if (gr < n && col < nb) LmT[(size_t)(j0 + col) * n + gr] = sI[r * LDM + col];
    }
  }
  PHASE_T(2);
  if (rem_c <= 0) return;                          // only the bordered row was left: nothing to update
  const double* sJr = sI;
  if (ti != tj) {
    trsm_rows16(inJ, sD, sJ + w * 16 * LDM, lane, xj);
    sJr = sJ;
    if (ti == 1 && tj == 0) {
      for (int e = tid; e < 64 * 64; e += 256) {
        const int col = e >> 6, r = e & 63;
        const int gr = j1 + r;
        if (gr < n && col < nb) LmT[(size_t)(j0 + col) * n + gr] = sJ[r * LDM + col];
      }
    }
  }
  v4d acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int k0 = 0; k0 < 64; k0 += 4) {
    const double a = sI[(w * 16 + col16) * LDM + k0 + rq];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double b = sJr[(t * 16 + col16) * LDM + k0 + rq];
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
  PHASE_T(3);
  const bool crit = (ti == 0 && tj == 0);
  const int nbn = rem_c < 64 ? rem_c : 64;        // size of the next diagonal block (rem_c >= 1 here)
  if (crit) __syncthreads();                       // the tiles are about to become the factor's buffers
  const Crit64 c = crit64_carve(smem);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = w * 16 + rq + 4 * i, lc = t * 16 + col16;      // position inside the tile
      const int gr = I0 + lr, gcol = J0 + lc;
      const double v = cv[t][i] - acc[t][i];
      const bool inside = gr < rem_r && gcol < rem_c && gcol <= gr;
      // the next diagonal block (top-left nbn x nbn of tile (0,0)) goes to LDS for the factorisation; a short
      // last block leaves the bordered right-hand-side row (lr >= nbn) on the normal path
      if (crit && lr < nbn && lc < nbn) { if (lc <= lr) c.sM[lr * LDM + lc] = v; }
      else if (inside) A[(size_t)(j1 + gr) * n + j1 + gcol] = v;
    }
  if (!crit) return;
  // identity outside the valid corner (Dn / inv64_next: every launch writes the same positions - the lower
  // triangles of the two 32x32 inverses and L21 - and dense_cholesky cleared the rest once)
  if (nbn < 64) {
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, cc = e & 63;
      if (r >= nbn || cc >= nbn) c.sM[r * LDM + cc] = (r == cc) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  crit64_lite(c, tid,
              [&](int r, int cc, double v) { if (r < nbn && cc < nbn) Lm[(size_t)(j1 + r) * n + j1 + cc] = v; },
              [&](int r, int cc, double v) { Dn[r * 64 + cc] = v; inv64_next[r * 64 + cc] = v; },
              [&](int r, int cc, double v) { Dn[r * 64 + cc] = v; }, fail);
}

__global__ __launch_bounds__(256) void k_chol_step(double* A, double* __restrict__ Lm, double* __restrict__ LmT, int n,
                                                   int nrows, int j0, int col_end, const double* D, double* Dn,
                                                   double* inv64_next, int* __restrict__ fail) {
  __shared__ double smem[CHOL_STEP_SMEM];
  chol_tile(smem, (int)blockIdx.x, true, A, Lm, LmT, n, nrows, j0, col_end, D, Dn, inv64_next, fail);
}

// Rank-K update of the trailing matrix after a 256-column strip (two-level scheme, large systems):
//   C[r][c] -= sum_k X[r][k] X[c][k],  r < R rows, c < Cn columns, c <= r,  X = the strip's columns of the factor.
// 128x128 tiles, 512 threads = 4x2 wavefronts of 32x64 (2x4 accumulators of v_mfma_f64_16x16x4_f64), K in chunks
// of 32 staged through LDS (row stride 34 doubles = conflict-free operand reads); the next chunk's global loads
// are in flight while the current one is multiplied.  1-D grid over the lower-triangular tiles.
template <int NW>   // wavefronts per workgroup: 4 (each 64x64 of the tile) or 8 (each 32x64: half the registers, twice the occupancy)
__device__ __forceinline__ void syrk_lower_body(double* __restrict__ C, int ldc, const double* __restrict__ X, int ldx,
                                                int R, int Cn, int K) {
  constexpr int KC = 32, LDK = 34;
  constexpr int NT = NW * 64;                 // threads
  constexpr int RT = NW == 8 ? 2 : 4;         // 16-row MFMA tiles per wavefront
  constexpr int QN = 128 * KC / 2 / NT;       // double2 loads per thread and operand per chunk: 8 (256 thr) or 4 (512 thr)
  __shared__ double sA[128 * LDK], sB[128 * LDK];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  int ti = (int)((sqrtf(8.0f * (float)blockIdx.x + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= (int)blockIdx.x) ++ti;
  while (ti * (ti + 1) / 2 > (int)blockIdx.x) --ti;
  const int tj = (int)blockIdx.x - ti * (ti + 1) / 2;
  const int I0 = ti * 128, J0 = tj * 128;
  const bool diag = ti == tj;
  constexpr int TPR = NT / 128;                           // threads per staged row
  const int lr = tid / TPR, lc = (tid % TPR) * (KC / TPR); // this thread stages KC / TPR doubles of row lr
  double2 ra[QN], rb[QN];
  const bool vec2 = ((ldx & 1) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);     // 16-byte loads need even strides
  // rows past the end read row 0 and are zeroed when the registers go to LDS - a select right after the load would
  // make the compiler wait for the data at once and lose the overlap with the MFMAs of the current chunk
  const int ga = I0 + lr, gb = J0 + lr;
  const bool oka = ga < R, okb = !diag && gb < Cn;
  const double* rowa = X + (size_t)(oka ? ga : 0) * ldx + lc;
  const double* rowb = X + (size_t)(okb ? gb : 0) * ldx + lc;
  auto load_chunk = [&](int k0) {
    if (vec2 && k0 + KC <= K) {
#pragma unroll
      for (int q = 0; q < QN; ++q) { ra[q] = *(const double2*)(rowa + k0 + 2 * q); rb[q] = *(const double2*)(rowb + k0 + 2 * q); }
    } else {
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        const int kk = k0 + lc + 2 * q;
        ra[q].x = kk < K ? rowa[k0 + 2 * q] : 0.0;      ra[q].y = kk + 1 < K ? rowa[k0 + 2 * q + 1] : 0.0;
        rb[q].x = kk < K ? rowb[k0 + 2 * q] : 0.0;      rb[q].y = kk + 1 < K ? rowb[k0 + 2 * q + 1] : 0.0;
      }
    }
  };
  v4d acc[RT][4];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
  const double* sBr = diag ? sA : sB;
  const int r16 = lane & 15, kq = lane >> 4;
  const int row0 = wr * (RT * 16);                        // first tile row of this wavefront
  load_chunk(0);
  for (int k0 = 0; k0 < K; k0 += KC) {
    __syncthreads();                                     // the previous chunk has been consumed
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      *(double2*)(sA + lr * LDK + lc + 2 * q) = oka ? ra[q] : make_double2(0.0, 0.0);
      if (!diag) *(double2*)(sB + lr * LDK + lc + 2 * q) = okb ? rb[q] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    if (k0 + KC < K) load_chunk(k0 + KC);
#pragma unroll
    for (int ks = 0; ks < KC; ks += 4) {
      double a[RT], b[4];
#pragma unroll
      for (int t = 0; t < RT; ++t) a[t] = sA[(row0 + t * 16 + r16) * LDK + ks + kq];
#pragma unroll
      for (int t = 0; t < 4; ++t) b[t] = sBr[(wc * 64 + t * 16 + r16) * LDK + ks + kq];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt], b[ct], acc[rt][ct], 0, 0, 0);
    }
  }
  // read-modify-write of C in batches of 16 values: all loads of a batch are issued before its first store (a
  // load after a store through the same pointer would otherwise wait for it - 64 serialised round trips)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    double cold[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gr = I0 + row0 + rt * 16 + kq + 4 * i, gc = J0 + wc * 64 + ct * 16 + r16;
        cold[ct][i] = (gr < R && gc < Cn && gc <= gr) ? C[(size_t)gr * ldc + gc] : 0.0;
      }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gr = I0 + row0 + rt * 16 + kq + 4 * i, gc = J0 + wc * 64 + ct * 16 + r16;
        if (gr < R && gc < Cn && gc <= gr) C[(size_t)gr * ldc + gc] = cold[ct][i] - acc[rt][ct][i];
      }
  }
}

// 8 wavefronts of 32x64 each (142 registers: measured 5 % faster than 4 wavefronts of 64x64 at 248; forcing 128
// registers for a second resident workgroup spills and loses it again)
__global__ __launch_bounds__(512) void k_syrk_lower(double* __restrict__ C, int ldc, const double* __restrict__ X, int ldx,
                                                    int R, int Cn, int K) {
  syrk_lower_body<8>(C, ldc, X, ldx, R, Cn, K);
}

// Off-diagonal quadrant of each 64x64 diagonal-block inverse, for the triangular solves:
// inv64[b] holds [Li11 0; 0 Li22] after the factorisation; Li21 = -Li22 (L21 Li11).  One workgroup per block.
__global__ __launch_bounds__(256) void k_inv64_fix(const double* __restrict__ Lm, int n, double* __restrict__ inv64) {
  __shared__ double sA[32 * LDL], sB[32 * LDL], sC2[32 * LDL], sT[32 * LDL];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r0 = b * 64;
  double* M = inv64 + (size_t)b * 64 * 64;
  for (int e = tid; e < 32 * 32; e += 256) {
    const int r = e >> 5, cc = e & 31;
    sA[r * LDL + cc] = (r0 + 32 + r < n) ? Lm[(size_t)(r0 + 32 + r) * n + r0 + cc] : 0.0;     // L21
    sB[r * LDL + cc] = M[r * 64 + cc];                                                       // Li11
    sC2[r * LDL + cc] = M[(32 + r) * 64 + 32 + cc];                                          // Li22
  }
  __syncthreads();
  const int tr = w >> 1, tc = w & 1, kq = lane >> 4, r16 = lane & 15;
  const v4d t = mm32_tile<false>(sA, LDL, sB, LDL, tr, tc, lane);
#pragma unroll
  for (int q = 0; q < 4; ++q) sT[(tr * 16 + kq + 4 * q) * LDL + tc * 16 + r16] = t[q];
  __syncthreads();
  const v4d m = mm32_tile<false>(sC2, LDL, sT, LDL, tr, tc, lane);
#pragma unroll
  for (int q = 0; q < 4; ++q) M[(32 + tr * 16 + kq + 4 * q) * 64 + tc * 16 + r16] = -m[q];
}

// ------------------------------------------------------------------------------------ diagonal-block inverses
// inv64[b] = inverse of the 64x64 lower-triangular diagonal block b of L (identity-padded at the tail).
// One thread per column: forward substitution of e_t, L and the result in LDS.
__global__ __launch_bounds__(64) void k_inv64(const double* __restrict__ L, int n, double* __restrict__ inv64) {
  __shared__ double sLd[64 * 65];
  const int b = blockIdx.x, t = threadIdx.x, r0 = b * 64;
  for (int i = t; i < 64 * 64; i += 64) {
    const int r = i >> 6, c = i & 63;
    sLd[r * 65 + c] = (r0 + r < n && c <= r) ? L[(size_t)(r0 + r) * n + r0 + c] : ((r == c) ? 1.0 : 0.0);
  }
  __syncthreads();
  double x[64];
  double* out = inv64 + (size_t)b * 64 * 64;
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    double s = (r == t) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < r; ++c) s -= sLd[r * 65 + c] * x[c];
    x[r] = (r >= t) ? s / sLd[r * 65 + r] : 0.0;
    out[r * 64 + t] = x[r];                      // coalesced over t
  }
}

__global__ void k_set_identity64(double* __restrict__ m) {
  for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) m[e] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0;
}

// 128-block = [A 0; B C]:  inverse = [Ai 0; -Ci B Ai, Ci].  One workgroup per block; both 64x64x64 products
// on v_mfma_f64_16x16x4_f64 (wave w owns rows 16 w .. 16 w + 15), T = B Ai kept in LDS between them.
__global__ __launch_bounds__(256) void k_inv_merge(const double* __restrict__ L, int n, const double* __restrict__ inv64,
                                                   double* __restrict__ Dinv, double* __restrict__ DinvT) {
  constexpr int LDT = 66;
  __shared__ double sT[64 * LDT];
  const int b = blockIdx.x, r0 = b * 128, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const double* Ai = inv64 + (size_t)(2 * b) * 64 * 64;
  const double* Ci = inv64 + (size_t)(2 * b + 1) * 64 * 64;
  double* D = Dinv + (size_t)b * 128 * 128;
  double* Dt = DinvT + (size_t)b * 128 * 128;
  const int r16 = lane & 15, kq = lane >> 4;
  // T = B Ai : A operand = B[row][k] (rows r0+64.., cols r0..r0+63 of L), B operand = Ai[k][col]
  v4d acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  const int brow = r0 + 64 + w * 16 + r16;
  const double* bptr = L + (size_t)brow * n + r0;
  const bool bok = brow < n;
  for (int k0 = 0; k0 < 64; k0 += 4) {
    const double a = bok ? bptr[k0 + kq] : 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double bb = Ai[(k0 + kq) * 64 + t * 16 + r16];
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) sT[(w * 16 + kq + 4 * i) * LDT + t * 16 + r16] = acc[t][i];
  __syncthreads();
  // M = -Ci T : A operand = Ci[row][k], B operand = T[k][col]
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < 64; k0 += 4) {
    const double a = Ci[(w * 16 + r16) * 64 + k0 + kq];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double bb = sT[(k0 + kq) * LDT + t * 16 + r16];
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 64 + w * 16 + kq + 4 * i, c = t * 16 + r16;
      const double v = -acc[t][i];
      D[r * 128 + c] = v;
      Dt[c * 128 + r] = v;
    }
  // the two diagonal quadrants and the zero quadrant
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    const double va = Ai[e], vc = Ci[e];
    D[r * 128 + c] = va;               Dt[c * 128 + r] = va;
    D[(64 + r) * 128 + 64 + c] = vc;   Dt[(64 + c) * 128 + 64 + r] = vc;
    D[r * 128 + 64 + c] = 0.0;         Dt[(64 + c) * 128 + r] = 0.0;
  }
}

// ------------------------------------------------------------------------------------ triangular solves
// One 128-row block per launch.  transpose = 0: L y = b top-down; 1: L^T x = b bottom-up.
// Every workgroup computes x_blk = Dinv(^T) rhs_blk itself, then updates its own later/earlier rows of b.
__global__ __launch_bounds__(256) void k_trsv_step(const double* __restrict__ L, int n, const double* __restrict__ Dinv,
                                                   const double* __restrict__ DinvT, double* __restrict__ b,
                                                   double* __restrict__ xout, int blk, int transpose) {
  __shared__ double srhs[128], sx[128], shalf[128];
  const int tid = threadIdx.x, r0 = blk * 128;
  const int nb = (n - r0) < 128 ? (n - r0) : 128;
  if (tid < 128) srhs[tid] = (tid < nb) ? b[r0 + tid] : 0.0;
  __syncthreads();
  {
    // forward: y[r] = sum_c Dinv[r][c] rhs[c] read as DinvT[c][r];  transpose: x[r] = sum_c Dinv[c][r] rhs[c]
    const double* M = (transpose ? Dinv : DinvT) + (size_t)blk * 128 * 128;
    const int r = tid & 127, h = tid >> 7;
    double s = 0.0;
#pragma unroll 8
    for (int c = h * 64; c < h * 64 + 64; ++c) s += M[c * 128 + r] * srhs[c];
    if (h == 1) shalf[r] = s;
    __syncthreads();
    if (h == 0) {
      s += shalf[r];
      sx[r] = s;
      if (blockIdx.x == 0 && r < nb) xout[r0 + r] = s;
    }
    __syncthreads();
  }
  if (!transpose) {
    const int lane = tid & 63, w = tid >> 6;
    const int base = r0 + 128 + blockIdx.x * 32;
    for (int q = w; q < 32; q += 4) {
      const int i = base + q;
      if (i >= n) break;
      const double* lrow = L + (size_t)i * n + r0;
      double t = lrow[lane] * sx[lane] + lrow[64 + lane] * sx[64 + lane];   // rows i > r0 + 127: full 128 columns exist
      t = wave_sum_d(t);
      if (lane == 0) b[i] -= t;
    }
  } else {
    const int i = blockIdx.x * 256 + tid;
    if (i >= r0) return;
    double t = 0.0;
#pragma unroll 8
    for (int c = 0; c < nb; ++c) t += L[(size_t)(r0 + c) * n + i] * sx[c];
    b[i] -= t;
  }
}

// Whole triangular solve in ONE launch (n <= 128 * TRSV_FLOW_MAX_BLOCKS): workgroup g owns the 128 unknowns of
// block g, keeps Dinv_g(^T) in LDS, and consumes the other blocks' solutions as they are published.  There is
// no flag: xout is pre-filled with an all-ones bit pattern (a NaN no arithmetic produces), producers store their
// solution with agent-scope atomic stores and each consumer thread polls ITS element with agent-scope atomic
// loads, so a hand-off is one L2 round trip.  After its last dependency a block only has one 128x128
// product (from registers) + the Dinv product (from LDS) left: the chain is n/128 hand-offs, not n/128 launches.
// Every working workgroup must be resident at once: one per CU (134 KB of LDS).  Placement: workgroups go
// round-robin over the 8 XCDs, so worker p sits at blockIdx 8 p + xcd(p) with 32 consecutive workers per XCD
// (32 CUs each): hand-offs between neighbours stay inside one L2.  Progress: block g only waits for blocks that
// never wait for g, the first block waits for none, and the idle workgroups of the grid exit at once.
constexpr int TRSV_FLOW_MAX_BLOCKS = 128;
__device__ __forceinline__ bool trsv_pending(double v) { return __double_as_longlong(v) == -1LL; }
template <bool TRANSPOSE>
__global__ __launch_bounds__(256) void k_trsv_flow(const double* __restrict__ L, const double* __restrict__ LT, int n,
                                                   const double* __restrict__ Dinv, const double* __restrict__ DinvT,
                                                   const double* __restrict__ b, double* __restrict__ xout,
                                                   int* __restrict__ fail) {
  __shared__ double sM[128 * 128];
  __shared__ double sacc[128], sx[128], shalf[128];
  const int tid = threadIdx.x;
  const int nblk = (n + 127) / 128;
  const int p = (int)blockIdx.x >> 3;                                          // position in the dependency order
  if (((int)blockIdx.x & 7) != ((p >> 5) & 7)) return;
  const int g = TRANSPOSE ? nblk - 1 - p : p;
  const int r0 = g * 128;
  const int nbg = (n - r0) < 128 ? (n - r0) : 128;
  {
    // forward: y[r] = sum_c Dinv[r][c] acc[c] read as DinvT[c][r];  transpose: x[r] = sum_c Dinv[c][r] acc[c]
    const double* M = (TRANSPOSE ? Dinv : DinvT) + (size_t)g * 128 * 128;
    for (int e = tid; e < 128 * 128; e += 256) sM[e] = M[e];
  }
  if (tid < 128) sacc[tid] = (tid < nbg) ? b[r0 + tid] : 0.0;
  const int i = tid & 127, h = tid >> 7;
  double part = 0.0;                       // this thread's half of row/column i's sum, over all blocks
  __syncthreads();
  const int first = TRANSPOSE ? nblk - 1 : 0, step = TRANSPOSE ? -1 : 1;
  // The 128x128 tile of L that couples block `blk` to this block does not depend on any flag: it is loaded into
  // registers BEFORE waiting for x_blk, so after the hand-off only LDS reads and FMAs remain on the chain.
  //   TRANSPOSE: lt[c] = L[rb + 64 h + c][r0 + i]      (column i of this block: coalesced over i)
  //   forward  : lt[c] = L[r0 + i][rb + 64 h + c] read from the transposed copy LT[rb + 64 h + c][r0 + i] the
  //              factorisation leaves behind (LT == nullptr: from L itself, 512 contiguous bytes per thread)
  double lt[64];
  auto load_tile = [&](int blk) {
    const int rb = blk * 128;
    const int nbb = (n - rb) < 128 ? (n - rb) : 128;
    if (TRANSPOSE) {
      const double* lp = L + (size_t)(rb + h * 64) * n + r0 + i;
#pragma unroll
      for (int c = 0; c < 64; ++c) lt[c] = (i < nbg && h * 64 + c < nbb) ? lp[(size_t)c * n] : 0.0;
    } else {
      const double* lp = LT + (size_t)(rb + h * 64) * n + r0 + i;             // earlier blocks are full
#pragma unroll
      for (int c = 0; c < 64; ++c) lt[c] = (i < nbg) ? lp[(size_t)c * n] : 0.0;
    }
  };
  if (first != g) load_tile(first);
  for (int blk = first; blk != g; blk += step) {
    const int rb = blk * 128;
    const int nbb = (n - rb) < 128 ? (n - rb) : 128;
    __syncthreads();                                 // everyone is done with the previous sx
    if (tid < 128) {
      double v = 0.0;
      if (tid < nbb) {
        // bounded: a producer that never publishes (it cannot, by construction) must not hang the device -
        // after ~4 s the element is taken as it is (an all-ones NaN), the failure flag is raised (value 2) and the
        // caller gets SFM_ERR_NUMERIC / a raised SfmError instead of a silently non-finite step
        int spins = 0;
        do { v = __hip_atomic_load(&xout[rb + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        while (trsv_pending(v) && ++spins < (1 << 22));
        if (trsv_pending(v)) *fail = 2;
      }
      sx[tid] = v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 64; ++c) part += lt[c] * sx[h * 64 + c];
    if (blk + step != g) load_tile(blk + step);     // in flight while the next flag is awaited
  }
  __syncthreads();
  if (h == 1) shalf[i] = part;
  __syncthreads();
  if (h == 0) sacc[i] -= part + shalf[i];
  __syncthreads();
  double s = 0.0;
#pragma unroll 8
  for (int c = h * 64; c < h * 64 + 64; ++c) s += sM[c * 128 + i] * sacc[c];
  if (h == 1) shalf[i] = s;
  __syncthreads();
  if (h == 0 && i < nbg) __hip_atomic_store(&xout[r0 + i], s + shalf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------ host
int dense_cholesky(sfm_ctx* h, double* A, int n, int nrows, const DenseWs& w) {
  // step data and block inverses: the kernels only write their structural non-zeros
  if (hipMemsetAsync(w.Ld, 0, (size_t)2 * 64 * 64 * sizeof(double), h->stream) != hipSuccess ||
      hipMemsetAsync(w.inv64, 0, (size_t)((n + 127) / 128) * 2 * 64 * 64 * sizeof(double), h->stream) != hipSuccess)
    return sfm_fail(h, SFM_ERR_HIP, "dense_cholesky", "memset");
  // Two-level scheme from CHOL_STRIP_MIN_N on: inside a 256-column strip the 64-column steps only touch the
  // strip's own columns, the rest of the trailing matrix gets ONE rank-256 update per strip (k_syrk_lower) - 4x the
  // flops per byte streamed from HBM.  Small systems are a pure latency chain and keep the one-level scheme.
  const char* strip_env = getenv("SFM_CHOL_STRIP_MIN_N");       // test knob
  const int strip_min_n = strip_env ? atoi(strip_env) : CHOL_STRIP_MIN_N;
  const int strip = n >= strip_min_n ? 256 : n;
  int step = 0;
  for (int jb = 0; jb < n; jb += strip) {
    const int je = (jb + strip) < n ? (jb + strip) : n;            // columns [jb, je) form this strip
    hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, h->stream, A, w.Lm, n, jb, w.Ld + (size_t)(step & 1) * 64 * 64,
                       w.inv64 + (size_t)(jb / 64) * 64 * 64, w.flag);
    for (int j0 = jb; j0 < je; j0 += 64, ++step) {
      const int nb = (n - j0) < 64 ? (n - j0) : 64;
      const int j1 = j0 + nb;
      const int below = nrows - j1;
      if (below <= 0) break;
      const unsigned T = cdiv(below, 64);
      double* D = w.Ld + (size_t)(step & 1) * 64 * 64;
      double* Dn = w.Ld + (size_t)((step + 1) & 1) * 64 * 64;
      const int tcols = (int)cdiv(je - j1 > 0 ? je - j1 : 0, 64);  // tile columns still inside the strip
      unsigned grid;
      if (tcols == 0) grid = T;                                    // panel solve only (strip end / bordered row)
      else if (je >= n) grid = T * (T + 1) / 2;                    // last (or only) strip: the whole trailing triangle
      else { grid = 0; for (int c = 0; c < tcols; ++c) grid += T - c; }
      hipLaunchKernelGGL(k_chol_step, dim3(grid), dim3(256), 0, h->stream, A, w.Lm, w.LmT, n, nrows, j0, je, D, Dn,
                         w.inv64 + (size_t)(j1 / 64) * 64 * 64, w.flag);
    }
    if (je < n) {
      const int R = nrows - je, Cn = n - je;
      const unsigned T2 = cdiv(R, 128);
      hipLaunchKernelGGL(k_syrk_lower, dim3(T2 * (T2 + 1) / 2), dim3(512), 0, h->stream, A + (size_t)je * n + je, n,
                         w.Lm + (size_t)je * n + jb, n, R, Cn, je - jb);
    }
  }
  // 64x64 and then 128x128 diagonal-block inverses for the triangular solves
  const unsigned nb64 = cdiv(n, 64), nb128 = cdiv(n, 128);
  hipLaunchKernelGGL(k_inv64_fix, dim3(nb64), dim3(256), 0, h->stream, w.Lm, n, w.inv64);
  if ((nb64 & 1u) != 0)             // odd number of 64-blocks: the partner of the last one is an identity block
    hipLaunchKernelGGL(k_set_identity64, dim3(1), dim3(256), 0, h->stream, w.inv64 + (size_t)nb64 * 64 * 64);
  hipLaunchKernelGGL(k_inv_merge, dim3(nb128), dim3(256), 0, h->stream, w.Lm, n, w.inv64, w.Dinv, w.DinvT);
  SFM_LAUNCH_CHECK(h, "dense_cholesky");
  return SFM_OK;
}

int dense_trsv(sfm_ctx* h, int n, const DenseWs& w, double* b, double* xout, int transpose) {
  const double* L = w.Lm;
  const int nblk = (n + 127) / 128;
  const char* flow = getenv("SFM_TRSV_FLOW");       // "0": one launch per block (the path taken for n > 16384)
  if (nblk <= TRSV_FLOW_MAX_BLOCKS && !(flow && flow[0] == '0')) {
    if (hipMemsetAsync(xout, 0xFF, (size_t)n * sizeof(double), h->stream) != hipSuccess)     // "not published yet"
      return sfm_fail(h, SFM_ERR_HIP, "dense_trsv", "memset");
    if (transpose)
      hipLaunchKernelGGL(k_trsv_flow<true>, dim3(8 * nblk), dim3(256), 0, h->stream, L, w.LmT, n, w.Dinv, w.DinvT, b, xout, w.flag);
    else
      hipLaunchKernelGGL(k_trsv_flow<false>, dim3(8 * nblk), dim3(256), 0, h->stream, L, w.LmT, n, w.Dinv, w.DinvT, b, xout, w.flag);
    SFM_LAUNCH_CHECK(h, "dense_trsv");
    return SFM_OK;
  }
  if (!transpose) {
    for (int blk = 0; blk < nblk; ++blk) {
      const int after = n - (blk * 128 + 128);
      const unsigned g = after > 0 ? cdiv(after, 32) : 1;
      hipLaunchKernelGGL(k_trsv_step, dim3(g), dim3(256), 0, h->stream, L, n, w.Dinv, w.DinvT, b, xout, blk, 0);
    }
  } else {
    for (int blk = nblk - 1; blk >= 0; --blk) {
      const int before = blk * 128;
      const unsigned g = before > 0 ? cdiv(before, 256) : 1;
      hipLaunchKernelGGL(k_trsv_step, dim3(g), dim3(256), 0, h->stream, L, n, w.Dinv, w.DinvT, b, xout, blk, 1);
    }
  }
  SFM_LAUNCH_CHECK(h, "dense_trsv");
  return SFM_OK;
}

// ------------------------------------------------------------------------------------ exported helpers (tests)
extern "C" int sfm_dense_cholesky(sfm_handle h, double* a, int32_t n, int32_t* fail_flag) {
  if (!h || !a || n < 1 || !fail_flag) return SFM_ERR_ARG;
  double* base = nullptr;
  SFM_HIP(h, hipMalloc(&base, (size_t)dense_ws_doubles(n) * sizeof(double)));
  DenseWs w; dense_ws_carve(base, n, &w);
  w.flag = fail_flag;
  SFM_HIP(h, hipMemsetAsync(fail_flag, 0, sizeof(int), h->stream));
  SFM_HIP(h, hipMemsetAsync(w.Lm, 0, (size_t)n * n * sizeof(double), h->stream));
  int rc = dense_cholesky(h, a, n, n, w);
  SFM_HIP(h, hipMemcpyAsync(a, w.Lm, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  SFM_HIP(h, hipStreamSynchronize(h->stream));
  SFM_HIP(h, hipFree(base));
  return rc;
}

// test helper only: LT[c][r] = L[r][c] (the factorisation writes this copy itself)
__global__ __launch_bounds__(256) void k_transpose_copy(const double* __restrict__ L, int n, double* __restrict__ LT) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)n * n) return;
  const int r = (int)(e / n), c = (int)(e - (int64_t)r * n);
  LT[(size_t)c * n + r] = L[e];
}

extern "C" int sfm_dense_trsv(sfm_handle h, const double* l, int32_t n, double* b, int transpose) {
  if (!h || !l || !b || n < 1) return SFM_ERR_ARG;
  double* base = nullptr;
  SFM_HIP(h, hipMalloc(&base, ((size_t)dense_ws_doubles(n) + n) * sizeof(double)));
  DenseWs w; dense_ws_carve(base, n, &w);          // w.flag points into the workspace
  SFM_HIP(h, hipMemsetAsync(w.flag, 0, sizeof(int), h->stream));
  double* xout = base + dense_ws_doubles(n);
  SFM_HIP(h, hipMemcpyAsync(w.Lm, l, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  hipLaunchKernelGGL(k_transpose_copy, dim3(cdiv((int64_t)n * n, 256)), dim3(256), 0, h->stream, l, n, w.LmT);
  const unsigned nb128 = cdiv(n, 128);
  hipLaunchKernelGGL(k_inv64, dim3(2 * nb128), dim3(64), 0, h->stream, l, n, w.inv64);
  hipLaunchKernelGGL(k_inv_merge, dim3(nb128), dim3(256), 0, h->stream, l, n, w.inv64, w.Dinv, w.DinvT);
  int rc = dense_trsv(h, n, w, b, xout, transpose ? 1 : 0);
  if (rc == SFM_OK) {
    SFM_HIP(h, hipMemcpyAsync(b, xout, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    SFM_HIP(h, hipStreamSynchronize(h->stream));
  }
  SFM_HIP(h, hipFree(base));
  return rc;
}
