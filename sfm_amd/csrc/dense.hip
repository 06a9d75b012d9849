// Dense SPD solver for the reduced camera system (n = n_cams * cam_dim, a few thousand): bordered
// lower Cholesky + triangular solves, fp64, gfx950.
//
// The factorisation is latency-bound (n/32 dependent panel steps), so the design minimises the work
// on that chain and the number of kernel boundaries:
//   k_chol_panel   : EVERY workgroup factors the 32x32 diagonal block itself in one wavefront
//                    (lane = row, 32 columns in registers, the finished column broadcast through LDS,
//                    1/sqrt by v_rsq_f64 + two Newton steps instead of sqrt + divide) and then solves
//                    its 256 rows of the panel by forward substitution from LDS.  The factored block
//                    is parked in `Ld` (other workgroups still read the unfactored one from A).
//   k_chol_update  : rank-32 update of the trailing lower tiles on v_mfma_f64_16x16x4_f64, panel rows
//                    staged in LDS (row stride 34 doubles = conflict-free ds_read_b64); copies `Ld`
//                    into A.                                                 -> 2 launches per step
//   k_inv64 / k_inv_merge_* : explicit inverses of the 128x128 diagonal blocks of L (both layouts), so
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
  return align_up((int64_t)(n + 1) * 32, 32) + 32 * 32 + 32 + 2 * nb * 128 * 128 + nb * 2 * 64 * 64 + nb * 64 * 64 + 32;
}
void dense_ws_carve(double* base, int n, DenseWs* w) {
  const int64_t nb = (n + 127) / 128;
  double* p = base;
  w->panel = p; p += align_up((int64_t)(n + 1) * 32, 32);
  w->Ld = p; p += 32 * 32;
  w->rd = p; p += 32;
  w->Dinv = p; p += nb * 128 * 128;
  w->DinvT = p; p += nb * 128 * 128;
  w->inv64 = p; p += nb * 2 * 64 * 64;
  w->tmp = p; p += nb * 64 * 64;
  w->flag = (int*)p;
}

// ------------------------------------------------------------------------------------ Cholesky
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ A, int n, int nrows, int j0,
                                                    double* __restrict__ panel, double* __restrict__ Ld,
                                                    int* __restrict__ fail) {
  __shared__ double sL[32 * 33];
  __shared__ double scol[2][32];
  __shared__ double srd[32];
  const int tid = threadIdx.x;
  const int nb = (n - j0) < 32 ? (n - j0) : 32;
  if (tid < 64) {
    const int i = tid & 31;            // both half-waves run the same rows; only lanes < 32 publish
    double a[32];
#pragma unroll
    for (int q = 0; q < 32; ++q)
      a[q] = (i < nb && q < nb && q <= i) ? A[(size_t)(j0 + i) * n + j0 + q] : ((q == i) ? 1.0 : 0.0);
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      double piv = readlane_d(a[j], j);
      if (!(piv > 0.0)) { bad = true; piv = 1.0; }
      const double rinv = rsqrt_nr(piv);
      a[j] *= rinv;                                    // lane j: piv * rinv = sqrt(piv); rows above j: unused
      srd[j] = rinv;                 // every lane stores the same value (uniform) -
      scol[j & 1][i] = a[j];         // and lanes i, i + 32 hold identical rows: no predication needed
      // same wavefront, LDS is in order: the reads below see the column just written
#pragma unroll
      for (int q = j + 1; q < 32; ++q) a[q] -= a[j] * scol[j & 1][q];
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) sL[i * 33 + q] = (q <= i) ? a[q] : 0.0;
    if (blockIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < 32; ++q) Ld[i * 32 + q] = (q <= i) ? a[q] : 0.0;
    }
    if (bad && blockIdx.x == 0 && tid == 0) *fail = 1;
  }
  __syncthreads();
  const int j1 = j0 + nb;
  const int i = j1 + blockIdx.x * 256 + tid;   // global row below the diagonal block
  if (i >= nrows) return;
  double x[32];
  double* arow = A + (size_t)i * n + j0;
#pragma unroll
  for (int q = 0; q < 32; ++q) x[q] = (q < nb) ? arow[q] : 0.0;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const double xq = x[q] * srd[q];             // rows >= nb of the block are identity rows (rd = 1)
    x[q] = xq;
#pragma unroll
    for (int c = q + 1; c < 32; ++c) x[c] -= xq * sL[c * 33 + q];
  }
  double* prow = panel + (size_t)(i - j1) * 32;
  // x[q] is exactly 0 for q >= nb.  The store into A is steered by an address select, not a branch
  // (32 predicated branches here made hipcc spill 600+ registers).
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    double* dst = (q < nb) ? (arow + q) : (prow + q);
    *dst = x[q];
    prow[q] = x[q];
  }
}

__global__ void k_chol_store_diag(double* __restrict__ A, int n, int j0, int nb, const double* __restrict__ Ld) {
  for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) {
    const int r = i >> 5, c = i & 31;
    if (r < nb && c <= r) A[(size_t)(j0 + r) * n + j0 + c] = Ld[i];
  }
}

__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ A, int n, int nrows, int j1,
                                                     const double* __restrict__ panel, int j0, int nb,
                                                     const double* __restrict__ Ld) {
  constexpr int LDP = 34;
  __shared__ double sI[64 * LDP], sJ[64 * LDP];
  const int ti = blockIdx.y, tj = blockIdx.x, tid = threadIdx.x;
  if (ti == 0 && tj == 0) {
    for (int i = tid; i < 32 * 32; i += 256) {
      const int r = i >> 5, c = i & 31;
      if (r < nb && c <= r) A[(size_t)(j0 + r) * n + j0 + c] = Ld[i];
    }
  }
  if (tj > ti) return;
  const int lane = tid & 63, w = tid >> 6;
  const int rem_r = nrows - j1, rem_c = n - j1;
  const int I0 = ti * 64, J0 = tj * 64;
  for (int i = tid; i < 64 * 32; i += 256) {
    const int r = i >> 5, c = i & 31;
    sI[r * LDP + c] = (I0 + r < rem_r) ? panel[(size_t)(I0 + r) * 32 + c] : 0.0;
    sJ[r * LDP + c] = (J0 + r < rem_c) ? panel[(size_t)(J0 + r) * 32 + c] : 0.0;
  }
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
  __syncthreads();
  v4d acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k0 = 0; k0 < 32; k0 += 4) {
    const double a = sI[(w * 16 + col16) * LDP + k0 + rq];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double b = sJ[(t * 16 + col16) * LDP + k0 + rq];
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gr = I0 + w * 16 + rq + 4 * i, gcol = J0 + t * 16 + col16;
      if (gr < rem_r && gcol < rem_c && gcol <= gr) A[(size_t)(j1 + gr) * n + j1 + gcol] = cv[t][i] - acc[t][i];
    }
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
// 128-block = [A 0; B C]:  inverse = [Ai 0; -Ci B Ai, Ci].   T = B Ai
__global__ __launch_bounds__(256) void k_inv_merge_T(const double* __restrict__ L, int n, const double* __restrict__ inv64,
                                                     double* __restrict__ tmp) {
  const int b = blockIdx.x, r0 = b * 128;
  const double* Ai = inv64 + (size_t)(2 * b) * 64 * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    double s = 0.0;
    if (r0 + 64 + r < n) {
      const double* brow = L + (size_t)(r0 + 64 + r) * n + r0;
      for (int k = c; k < 64; ++k) s += brow[k] * Ai[k * 64 + c];      // Ai lower: k >= c
    }
    tmp[(size_t)b * 64 * 64 + e] = s;
  }
}
__global__ __launch_bounds__(256) void k_inv_merge_M(int n, const double* __restrict__ inv64, const double* __restrict__ tmp,
                                                     double* __restrict__ Dinv, double* __restrict__ DinvT) {
  const int b = blockIdx.x;
  const double* Ai = inv64 + (size_t)(2 * b) * 64 * 64;
  const double* Ci = inv64 + (size_t)(2 * b + 1) * 64 * 64;
  const double* T = tmp + (size_t)b * 64 * 64;
  double* D = Dinv + (size_t)b * 128 * 128;
  double* Dt = DinvT + (size_t)b * 128 * 128;
  for (int e = threadIdx.x; e < 128 * 128; e += 256) {
    const int r = e >> 7, c = e & 127;
    double v = 0.0;
    if (r < 64) { if (c < 64) v = Ai[r * 64 + c]; }
    else if (c >= 64) v = Ci[(r - 64) * 64 + (c - 64)];
    else {
      const int rr = r - 64;
      double s = 0.0;
      for (int k = 0; k <= rr; ++k) s += Ci[rr * 64 + k] * T[k * 64 + c];   // Ci lower: k <= rr
      v = -s;
    }
    D[r * 128 + c] = v;
    Dt[c * 128 + r] = v;
  }
  (void)n;
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
  for (int j0 = 0; j0 < n; j0 += 32) {
    const int nb = (n - j0) < 32 ? (n - j0) : 32;
    const int j1 = j0 + nb;
    const int below = nrows - j1;
    const unsigned g = below > 0 ? cdiv(below, 256) : 1;
    hipLaunchKernelGGL(k_chol_panel, dim3(g), dim3(256), 0, h->stream, A, n, nrows, j0, w.panel, w.Ld, w.flag);
    if (below > 0) {
      const unsigned T = cdiv(below, 64);
      hipLaunchKernelGGL(k_chol_update, dim3(T, T), dim3(256), 0, h->stream, A, n, nrows, j1, w.panel, j0, nb, w.Ld);
    } else {
      hipLaunchKernelGGL(k_chol_store_diag, dim3(1), dim3(256), 0, h->stream, A, n, j0, nb, w.Ld);
    }
  }
  // diagonal-block inverses for the triangular solves
  const unsigned nb64 = cdiv(n, 64), nb128 = cdiv(n, 128);
  hipLaunchKernelGGL(k_inv64, dim3(2 * nb128), dim3(64), 0, h->stream, A, n, w.inv64);
  hipLaunchKernelGGL(k_inv_merge_T, dim3(nb128), dim3(256), 0, h->stream, A, n, w.inv64, w.tmp);
  hipLaunchKernelGGL(k_inv_merge_M, dim3(nb128), dim3(256), 0, h->stream, n, w.inv64, w.tmp, w.Dinv, w.DinvT);
  (void)nb64;
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
  hipLaunchKernelGGL(k_inv_merge_T, dim3(nb128), dim3(256), 0, h->stream, l, n, w.inv64, w.tmp);
  hipLaunchKernelGGL(k_inv_merge_M, dim3(nb128), dim3(256), 0, h->stream, n, w.inv64, w.tmp, w.Dinv, w.DinvT);
  int rc = dense_trsv(h, l, n, w, b, xout, transpose ? 1 : 0);
  if (rc == SFM_OK) {
    SFM_HIP(h, hipMemcpyAsync(b, xout, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    SFM_HIP(h, hipStreamSynchronize(h->stream));
  }
  SFM_HIP(h, hipFree(base));
  return rc;
}
