// Dense SPD solver for the reduced camera system (n = n_cams * cam_dim, a few thousand): bordered
// lower Cholesky + triangular solves, fp64, gfx950.
//
// The factorisation is latency-bound (n/32 dependent panel steps), so the design minimises the work
// on that chain and the number of kernel boundaries:
//   wave_chol32    : 32x32 diagonal block in ONE wavefront (lane = row, columns split over the two
//                    half-waves, finished column broadcast through LDS, 1/sqrt by v_rsq_f64 + two Newton
//                    steps instead of sqrt + divide)
//   wave_inv32_follow : its inverse on a SECOND wavefront of the same workgroup, one column behind the
//                    factor (producer/consumer through LDS), so it adds almost nothing to the chain
//   k_chol_panel   : rows below the block as a 32-deep MFMA product with L_jj^-1
//   k_chol_update  : rank-32 update of the trailing lower tiles on v_mfma_f64_16x16x4_f64, panel rows
//                    staged in LDS (row stride 34 doubles = conflict-free ds_read_b64); LOOK-AHEAD: the
//                    workgroup of tile (0,0) factors the next diagonal block right after updating it, so
//                    the serial chain has no kernel of its own                -> 2 launches per step
//   k_inv64 / k_inv_merge : explicit inverses of the 128x128 diagonal blocks of L (both layouts), so
//   k_trsv_step    : one launch per 128-row block: every workgroup recomputes x_blk = Dinv * rhs_blk
//                    (coalesced 128-long dot products) and updates its own rows of the right-hand side.
#include "dense.h"

typedef double v4d __attribute__((ext_vector_type(4)));

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
  return align_up((int64_t)(n + 1) * 64, 32) + 64 * 64 + 32 + 2 * nb * 128 * 128 + nb * 2 * 64 * 64 + nb * 64 * 64 + 32;
}
void dense_ws_carve(double* base, int n, DenseWs* w) {
  const int64_t nb = (n + 127) / 128;
  double* p = base;
  w->panel = p; p += align_up((int64_t)(n + 1) * 64, 32);
  w->Ld = p; p += 64 * 64;
  w->rd = p; p += 32;
  w->Dinv = p; p += nb * 128 * 128;
  w->DinvT = p; p += nb * 128 * 128;
  w->inv64 = p; p += nb * 2 * 64 * 64;
  w->tmp = p; p += nb * 64 * 64;
  w->flag = (int*)p;
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
// (both half-waves alike) forward-substitutes column t; row r only needs columns <= r of L, so this runs one
// column behind the factorisation instead of after it.  The producer never waits for the consumer, so the
// spin cannot deadlock.  Result: sLi[r*33 + t] = (L^-1)[r][t].
__device__ __forceinline__ void wave_inv32_follow(const double* __restrict__ sC, const double* __restrict__ srd,
                                                  int* __restrict__ s_ready, int lane, double* __restrict__ sLi, int ldl) {
  const int t = lane & 31;
  double x[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    while (__hip_atomic_load(s_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < r + 1) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double s = (r == t) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < r; ++c) s -= sC[c * 32 + r] * x[c];
    x[r] = (r >= t) ? s * srd[r] : 0.0;
    sLi[r * ldl + t] = x[r];
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

// Factor + invert the tile in c.sM.  store_L(r, c, v) / store_Li(r, c, v) receive every lower-triangular
// entry of L and L^-1 (64x64 coordinates); *ok is cleared on a non-positive pivot.  nb = valid size.
template <class FL, class FI>
__device__ __forceinline__ void crit64_run(const Crit64& c, int tid, FL store_L, FI store_Li, int* __restrict__ fail) {
  const int lane = tid & 63, w = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int tr = w >> 1, tc = w & 1;
  const int kq = lane >> 4, r16 = lane & 15;
  if (tid == 0) { c.flag[0] = 0; c.flag[1] = 0; }
  __syncthreads();
  if (w < 2) __builtin_amdgcn_s_setprio(3);     // the serial chain: ahead of the bulk tiles sharing these SIMDs
  // ---- P1: A11 = L11 L11^T (wave 0), Li11 = L11^-1 (wave 1, one column behind)
  if (w == 0) {
    double a[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { const int q = 2 * t + h; a[t] = (q <= i) ? c.sM[i * LDM + q] : 0.0; }
    if (!wave_chol32(a, lane, c.sC, c.srd, &c.flag[0]) && lane == 0) *fail = 1;
  } else if (w == 1) {
    wave_inv32_follow(c.sC, c.srd, &c.flag[0], lane, c.sLi11, LDL);
  }
  __syncthreads();
  // ---- P2: L21 = A21 Li11^T
  {
    const v4d acc = mm32_tile<true>(c.sM + 32 * LDM, LDM, c.sLi11, LDL, tr, tc, lane);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) c.sM[(32 + tr * 16 + kq + 4 * q) * LDM + tc * 16 + r16] = acc[q];
  }
  __syncthreads();
  // ---- P2b: T = L21 Li11 ;  P3: A22 -= L21 L21^T
  {
    const v4d t = mm32_tile<false>(c.sM + 32 * LDM, LDM, c.sLi11, LDL, tr, tc, lane);
    const v4d u = mm32_tile<true>(c.sM + 32 * LDM, LDM, c.sM + 32 * LDM, LDM, tr, tc, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = tr * 16 + kq + 4 * q, cc = tc * 16 + r16;
      c.sT[r * LDL + cc] = t[q];
      c.sM[(32 + r) * LDM + 32 + cc] -= u[q];
    }
  }
  // L11 / Li11 leave LDS now: sC is reused by the second factor, sLi11 later receives Li21
  for (int e = tid; e < 32 * 32; e += 256) {
    const int r = e >> 5, cc = e & 31;
    if (cc <= r) { store_L(r, cc, c.sC[cc * 32 + r]); store_Li(r, cc, c.sLi11[r * LDL + cc]); }
  }
  __syncthreads();
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
  // ---- P5: Li21 = -Li22 T
  {
    const v4d acc = mm32_tile<false>(c.sLi22, LDL, c.sT, LDL, tr, tc, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) store_Li(32 + tr * 16 + kq + 4 * q, tc * 16 + r16, -acc[q]);
  }
  for (int e = tid; e < 32 * 32; e += 256) {
    const int r = e >> 5, cc = e & 31;
    store_L(32 + r, cc, c.sM[(32 + r) * LDM + cc]);                          // L21 (full block)
    if (cc <= r) { store_L(32 + r, 32 + cc, c.sC[cc * 32 + r]); store_Li(32 + r, 32 + cc, c.sLi22[r * LDL + cc]); }
  }
}

// Factor the first 64x64 diagonal block in place; its inverse goes to Li (64x64 row-major, zeros above the
// diagonal) for the panel kernel and to inv64 for the triangular solves.
__global__ __launch_bounds__(256) void k_chol_diag(double* __restrict__ A, int n, double* __restrict__ Li,
                                                   double* __restrict__ inv64, int* __restrict__ fail) {
  __shared__ double smem[CRIT64_DOUBLES + 6];     // static: with `extern __shared__` hipcc needs 256 + 68 registers here
  const Crit64 c = crit64_carve(smem);
  const int tid = threadIdx.x;
  const int nb = n < 64 ? n : 64;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, cc = e & 63;
    c.sM[r * LDM + cc] = (r < nb && cc < nb) ? ((cc <= r) ? A[(size_t)r * n + cc] : 0.0) : ((r == cc) ? 1.0 : 0.0);
    Li[e] = 0.0;
    inv64[e] = 0.0;
  }
  __syncthreads();
  crit64_run(c, tid,
             [&](int r, int cc, double v) { if (r < nb && cc < nb) A[(size_t)r * n + cc] = v; },
             [&](int r, int cc, double v) { Li[r * 64 + cc] = v; inv64[r * 64 + cc] = v; }, fail);
}

// Rows below the diagonal block:  X = A_panel L_jj^-T  as a 64-deep product with the explicit inverse
// Li = L_jj^-1 (from the look-ahead workgroup):  X[r][c] = sum_{k<=c} A[r][k] Li[c][k]  on
// v_mfma_f64_16x16x4_f64, wave w = 16 rows, four 16-column tiles.  64 rows per workgroup.
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ A, int n, int nrows, int j0,
                                                    double* __restrict__ panel, const double* __restrict__ Li) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nb = (n - j0) < 64 ? (n - j0) : 64;
  const int j1 = j0 + nb;
  const int r16 = lane & 15, kq = lane >> 4;
  const int rbase = j1 + blockIdx.x * 64 + w * 16;     // first global row of this wave
  if (rbase >= nrows) return;
  const int arow = rbase + r16;
  const bool aok = arow < nrows;
  const double* ap = A + (size_t)arow * n + j0;
  // the 16 A-operand values of this lane (k = 4 s + kq), issued together
  double av[16];
#pragma unroll
  for (int s4 = 0; s4 < 16; ++s4) av[s4] = (aok && (4 * s4 + kq) < nb) ? ap[4 * s4 + kq] : 0.0;
  v4d acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  // Li is lower triangular: column tile t (columns 16 t ..) only has k <= 16 t + 15; B operand read from L2
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const double* bp = Li + (size_t)(t * 16 + r16) * 64 + kq;
#pragma unroll
    for (int s4 = 0; s4 < 4 * (t + 1); ++s4)
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bp[4 * s4], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rbase + kq + 4 * i, col = t * 16 + r16;
      if (row < nrows) {
        const double v = (col < nb) ? acc[t][i] : 0.0;
        if (col < nb) A[(size_t)row * n + j0 + col] = v;
        panel[(size_t)(row - j1) * 64 + col] = v;
      }
    }
}

// Rank-64 update of the trailing lower 64x64 tiles on v_mfma_f64_16x16x4_f64 (panel rows staged in LDS in two
// 32-deep halves, row stride 34 doubles = conflict-free ds_read_b64).  Look-ahead: the workgroup of tile
// (0,0) then factors AND inverts the next 64x64 diagonal block in LDS (crit64_run) and publishes L^-1 for the
// next panel kernel, so the serial factorisation chain never waits for a kernel of its own.
__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ A, int n, int nrows, int j1,
                                                     const double* __restrict__ panel, double* __restrict__ Li,
                                                     double* __restrict__ inv64_next, int* __restrict__ fail) {
  __shared__ double smem[CRIT64_DOUBLES + 6];     // 68.7 KB static (gfx950 allows up to 160 KB): 2 workgroups per CU
  double* sI = smem;                 // [64][LDM]
  double* sJ = smem + 64 * LDM;      // [64][LDM]   (the factor's buffers later reuse this space)
  // 1-D grid over the lower-triangular tiles only: block b -> (ti, tj), tj <= ti, b = ti (ti + 1) / 2 + tj
  const int tid = threadIdx.x;
  int ti = (int)((sqrtf(8.0f * (float)blockIdx.x + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= (int)blockIdx.x) ++ti;
  while (ti * (ti + 1) / 2 > (int)blockIdx.x) --ti;
  const int tj = (int)blockIdx.x - ti * (ti + 1) / 2;
  const int lane = tid & 63, w = tid >> 6;
  const int rem_r = nrows - j1, rem_c = n - j1;
  const int I0 = ti * 64, J0 = tj * 64;
  // prefetch the C tile this lane updates (rows 16 w + (lane>>4) + 4 i, cols 16 t + (lane&15))
  const int col16 = lane & 15, rq = lane >> 4;
  double cv[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gr = I0 + w * 16 + rq + 4 * i, gcol = J0 + t * 16 + col16;
      cv[t][i] = (gr < rem_r && gcol < rem_c && gcol <= gr) ? A[(size_t)(j1 + gr) * n + j1 + gcol] : 0.0;
    }
  v4d acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int i = tid; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    sI[r * LDM + c] = (I0 + r < rem_r) ? panel[(size_t)(I0 + r) * 64 + c] : 0.0;
    sJ[r * LDM + c] = (J0 + r < rem_c) ? panel[(size_t)(J0 + r) * 64 + c] : 0.0;
  }
  __syncthreads();
#pragma unroll 4
  for (int k0 = 0; k0 < 64; k0 += 4) {
    const double a = sI[(w * 16 + col16) * LDM + k0 + rq];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double b = sJ[(t * 16 + col16) * LDM + k0 + rq];
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
  const bool crit = (ti == 0 && tj == 0);
  const int nbn = rem_c < 64 ? rem_c : 64;        // size of the next diagonal block (rem_c >= 1 here)
  if (crit) __syncthreads();                       // staging is about to become the factor's tile buffer
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
  // identity outside the valid corner, zero the strict upper part the factor reads as "<= i" only
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, cc = e & 63;
    if (r >= nbn || cc >= nbn) c.sM[r * LDM + cc] = (r == cc) ? 1.0 : 0.0;
    Li[e] = 0.0;
    inv64_next[e] = 0.0;
  }
  __syncthreads();
  crit64_run(c, tid,
             [&](int r, int cc, double v) { if (r < nbn && cc < nbn) A[(size_t)(j1 + r) * n + j1 + cc] = v; },
             [&](int r, int cc, double v) { Li[r * 64 + cc] = v; inv64_next[r * 64 + cc] = v; }, fail);
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

// ------------------------------------------------------------------------------------ host
int dense_cholesky(sfm_ctx* h, double* A, int n, int nrows, const DenseWs& w) {
  hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, h->stream, A, n, w.Ld, w.inv64, w.flag);
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int nb = (n - j0) < 64 ? (n - j0) : 64;
    const int j1 = j0 + nb;
    const int below = nrows - j1;
    if (below <= 0) break;
    hipLaunchKernelGGL(k_chol_panel, dim3(cdiv(below, 64)), dim3(256), 0, h->stream, A, n, nrows, j0, w.panel, w.Ld);
    if (j1 < n) {         // trailing columns exist: update them; tile (0,0) factors + inverts the next diagonal block
      const unsigned T = cdiv(below, 64);
      hipLaunchKernelGGL(k_chol_update, dim3(T * (T + 1) / 2), dim3(256), 0, h->stream, A, n, nrows, j1, w.panel, w.Ld,
                         w.inv64 + (size_t)(j1 / 64) * 64 * 64, w.flag);
    }
  }
  // 128x128 diagonal-block inverses for the triangular solves from the 64x64 ones the factorisation left behind
  const unsigned nb128 = cdiv(n, 128);
  if ((cdiv(n, 64) & 1u) != 0)      // odd number of 64-blocks: the partner of the last one is an identity block
    hipLaunchKernelGGL(k_set_identity64, dim3(1), dim3(256), 0, h->stream, w.inv64 + (size_t)cdiv(n, 64) * 64 * 64);
  hipLaunchKernelGGL(k_inv_merge, dim3(nb128), dim3(256), 0, h->stream, A, n, w.inv64, w.Dinv, w.DinvT);
  SFM_LAUNCH_CHECK(h, "dense_cholesky");
  return SFM_OK;
}

int dense_trsv(sfm_ctx* h, const double* L, int n, const DenseWs& w, double* b, double* xout, int transpose) {
  const int nblk = (n + 127) / 128;
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
  int rc = dense_cholesky(h, a, n, n, w);
  SFM_HIP(h, hipStreamSynchronize(h->stream));
  SFM_HIP(h, hipFree(base));
  return rc;
}

extern "C" int sfm_dense_trsv(sfm_handle h, const double* l, int32_t n, double* b, int transpose) {
  if (!h || !l || !b || n < 1) return SFM_ERR_ARG;
  double* base = nullptr;
  SFM_HIP(h, hipMalloc(&base, ((size_t)dense_ws_doubles(n) + n) * sizeof(double)));
  DenseWs w; dense_ws_carve(base, n, &w);
  double* xout = base + dense_ws_doubles(n);
  const unsigned nb128 = cdiv(n, 128);
  hipLaunchKernelGGL(k_inv64, dim3(2 * nb128), dim3(64), 0, h->stream, l, n, w.inv64);
  hipLaunchKernelGGL(k_inv_merge, dim3(nb128), dim3(256), 0, h->stream, l, n, w.inv64, w.Dinv, w.DinvT);
  int rc = dense_trsv(h, l, n, w, b, xout, transpose ? 1 : 0);
  if (rc == SFM_OK) {
    SFM_HIP(h, hipMemcpyAsync(b, xout, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    SFM_HIP(h, hipStreamSynchronize(h->stream));
  }
  SFM_HIP(h, hipFree(base));
  return rc;
}
