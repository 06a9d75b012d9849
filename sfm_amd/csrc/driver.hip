// Driver-side rows either side of the hot path, gfx950 only:
//   sfm_assoc_radius     track point <-> correspondence association   sfm_reconstruction.py:209-218
//   sfm_triangulate2     two-view DLT + 4 px reprojection gate        sfm_reconstruction.py:287-307
//   sfm_epipolar_errors  symmetric epipolar distance + inlier mask    find_matches.py:160-174
// All three are HBM-trivial maps over small records; what matters is that one launch covers every image
// pair of a driver step (segments) and that the arithmetic is the reference's, operation for operation:
// no FMA contraction anywhere in this file (NumPy / OpenCV's generic x86 code do not fuse).
#include "common.h"
#include <cfloat>

#pragma clang fp contract(off)

namespace {

// largest s in [0, n_seg) with ptr[s] <= i (ptr ascending, ptr[0] = 0, i < ptr[n_seg]); skips empty segments
__device__ __forceinline__ int seg_of(const int64_t* __restrict__ ptr, int n_seg, int64_t i) {
  int lo = 0, hi = n_seg;                 // invariant: ptr[lo] <= i < ptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// exclusive scan of one int per thread over a 256-thread block; total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* s_w, int& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  __syncthreads();
  if (lane == 63) s_w[w] = incl;
  __syncthreads();
  int off = 0;
  total = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { if (k < w) off += s_w[k]; total += s_w[k]; }
  return off + incl - v;
}

// ------------------------------------------------------------------------------------------ association
// One thread per (track row, column split).  The correspondence index is wave-uniform, so the compiler
// keeps the correspondence in SGPRs (scalar loads, no LDS, no barrier) and a pair test is 5 fp64 VALU
// instructions: dx, dy, dx*dx, fma -> d^2 (prefilter only, 1e-9 slack), compare.  Only candidates redo the
// reference's arithmetic: `sqrt(dx*dx + dy*dy) < radius`, unfused, as np.linalg.norm(..., axis=2) <
// MATCHING_THRESHOLD evaluates it.  Blocks walk the segments their 256 rows touch (one, except at segment
// boundaries); blockIdx.y takes the y-th slice of each segment's correspondences so that a few hundred
// row blocks still fill 256 CUs.  Pass 1 counts per (row, split), pass 2 writes at
// blk_off + scan(row totals) + prefix over splits: np.where order, deterministic.
template <bool FILL>
__global__ __launch_bounds__(256) void k_assoc(const double2* __restrict__ tp, const int64_t* __restrict__ t_ptr,
                                               const double2* __restrict__ cp, const int64_t* __restrict__ m_ptr,
                                               int n_seg, int64_t T, int nsplit, double radius, double r2_hi,
                                               int* __restrict__ cnt_rs, int* __restrict__ blk_cnt,
                                               const int* __restrict__ blk_off, int* __restrict__ out_row,
                                               int* __restrict__ out_col, int64_t capacity) {
  __shared__ int s_w[4];
  const int tid = threadIdx.x, split = blockIdx.y;
  const int64_t row0 = (int64_t)blockIdx.x * 256;
  const int64_t i = row0 + tid;
  const bool active = i < T;
  const int64_t last = (row0 + 255 < T ? row0 + 255 : T - 1);
  const int seg_first = seg_of(t_ptr, n_seg, row0);
  const int seg_last = seg_of(t_ptr, n_seg, last);
  const int my_seg = active ? seg_of(t_ptr, n_seg, i) : -1;
  double2 t = make_double2(0.0, 0.0);
  if (active) t = tp[i];
  int64_t off = 0;
  if (FILL) {
    int row_total = 0, before = 0;
    if (active)
      for (int y = 0; y < nsplit; ++y) {
        const int c = cnt_rs[i * nsplit + y];
        row_total += c;
        if (y < split) before += c;
      }
    int total;
    off = (int64_t)blk_off[blockIdx.x] + block_excl_scan(row_total, s_w, total) + before;
  }
  int cnt = 0;
  for (int s = seg_first; s <= seg_last; ++s) {
    const int64_t m0 = m_ptr[s], m1 = m_ptr[s + 1];
    const int64_t chunk = (m1 - m0 + nsplit - 1) / nsplit;
    const int64_t b = m0 + chunk * split;
    const int64_t e = (b + chunk < m1 ? b + chunk : m1);
    if (s != my_seg) continue;
    // 8 correspondences per trip: their loads are issued together (s_load_dwordx16 x2), the prefilter keeps
    // only min d^2, and the exact test reruns the 8 only when some lane has a candidate (rare)
    int64_t m = b;
    for (; m + 8 <= e; m += 8) {
      double2 c[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k] = cp[m + k];
      double dmin = r2_hi;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double dx = t.x - c[k].x, dy = t.y - c[k].y;
        dmin = fmin(dmin, __builtin_fma(dx, dx, dy * dy));
      }
      if (dmin < r2_hi) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const double dx = t.x - c[k].x, dy = t.y - c[k].y;
          if (sqrt(dx * dx + dy * dy) < radius) {
            if (FILL) {
              if (off < capacity) { out_row[off] = (int)i; out_col[off] = (int)(m + k); }
              ++off;
            } else {
              ++cnt;
            }
          }
        }
      }
    }
    for (; m < e; ++m) {
      const double2 c = cp[m];
      const double dx = t.x - c.x, dy = t.y - c.y;
      if (sqrt(dx * dx + dy * dy) < radius) {
        if (FILL) {
          if (off < capacity) { out_row[off] = (int)i; out_col[off] = (int)m; }
          ++off;
        } else {
          ++cnt;
        }
      }
    }
  }
  if (!FILL) {
    if (active) cnt_rs[i * nsplit + split] = cnt;
    int total;
    (void)block_excl_scan(cnt, s_w, total);
    if (tid == 0 && total) atomicAdd(&blk_cnt[blockIdx.x], total);
  }
}

// exclusive scan of n ints by one 256-thread block (n = number of 256-row blocks, a few thousand at most)
__global__ __launch_bounds__(256) void k_scan_blocks(int n, const int* __restrict__ in, int* __restrict__ out,
                                                     int64_t* __restrict__ total_out) {
  __shared__ long long s_sum[256];
  const int tid = threadIdx.x;
  const int per = (n + 255) / 256;
  const int b = tid * per, e = (b + per < n ? b + per : n);
  long long local = 0;
  for (int k = b; k < e; ++k) local += in[k];
  s_sum[tid] = local;
  __syncthreads();
  if (tid == 0) {
    long long run = 0;
    for (int k = 0; k < 256; ++k) { const long long v = s_sum[k]; s_sum[k] = run; run += v; }
    *total_out = run;
  }
  __syncthreads();
  long long run = s_sum[tid];
  for (int k = b; k < e; ++k) { out[k] = (int)run; run += in[k]; }
}

// ---------------------------------------------------------------------------------------- triangulation
// One thread per candidate track.  cv2.triangulatePoints (opencv-python 4.11.0) builds, per point, the
// 4x4 matrix with rows x*P[2]-P[0], y*P[2]-P[1] for both views and takes the right singular vector of the
// smallest singular value from its one-sided Jacobi SVD; the same Hestenes iteration runs here on the
// columns of A held in registers (eps = 10*DBL_EPSILON, <= 30 sweeps).
__global__ __launch_bounds__(256) void k_triangulate2(const double* __restrict__ proj, const int* __restrict__ cam0,
                                                      const int* __restrict__ cam1, const double2* __restrict__ x0,
                                                      const double2* __restrict__ x1, int64_t n, double max_err,
                                                      double* __restrict__ X, int* __restrict__ valid,
                                                      double* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double P0[12], P1[12];
  {
    const double* a = proj + 12 * (int64_t)cam0[i];
    const double* b = proj + 12 * (int64_t)cam1[i];
#pragma unroll
    for (int k = 0; k < 12; ++k) { P0[k] = a[k]; P1[k] = b[k]; }
  }
  const double2 p0 = x0[i], p1 = x1[i];
  double U[4][4], V[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    U[0][k] = p0.x * P0[8 + k] - P0[k];
    U[1][k] = p0.y * P0[8 + k] - P0[4 + k];
    U[2][k] = p1.x * P1[8 + k] - P1[k];
    U[3][k] = p1.y * P1[8 + k] - P1[4 + k];
#pragma unroll
    for (int r = 0; r < 4; ++r) V[r][k] = (r == k) ? 1.0 : 0.0;
  }
  const double eps = 10.0 * DBL_EPSILON;
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool changed = false;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double a = 0.0, b = 0.0, g = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) { a += U[r][p] * U[r][p]; b += U[r][q] * U[r][q]; g += U[r][p] * U[r][q]; }
        if (fabs(g) > eps * sqrt(a * b)) {
          changed = true;
          const double zeta = (b - a) / (2.0 * g);
          const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double up = U[r][p], uq = U[r][q];
            U[r][p] = c * up - s * uq; U[r][q] = s * up + c * uq;
            const double vp = V[r][p], vq = V[r][q];
            V[r][p] = c * vp - s * vq; V[r][q] = s * vp + c * vq;
          }
        }
      }
    }
    if (!changed) break;
  }
  double best = 0.0;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double nk = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) nk += U[r][k] * U[r][k];
    if (k == 0 || nk < best) {
      best = nk;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = V[r][k];
    }
  }
  const double Xx = v[0] / v[3], Xy = v[1] / v[3], Xz = v[2] / v[3];
  X[3 * i] = Xx; X[3 * i + 1] = Xy; X[3 * i + 2] = Xz;
  // reprojection gate, sfm_reconstruction.py:298-305
  double e[2];
#pragma unroll
  for (int view = 0; view < 2; ++view) {
    const double* P = view ? P1 : P0;
    const double2 uv = view ? p1 : p0;
    const double hx = P[0] * Xx + P[1] * Xy + P[2] * Xz + P[3];
    const double hy = P[4] * Xx + P[5] * Xy + P[6] * Xz + P[7];
    const double hw = P[8] * Xx + P[9] * Xy + P[10] * Xz + P[11];
    const double du = hx / hw - uv.x, dv = hy / hw - uv.y;
    e[view] = sqrt(du * du + dv * dv);
  }
  valid[i] = (!(e[0] > max_err) && !(e[1] > max_err)) ? 1 : 0;
  if (err) { err[2 * i] = e[0]; err[2 * i + 1] = e[1]; }
}

// ------------------------------------------------------------------------------- epipolar verification
// hipcc's float `/` and sqrtf are correctly rounded by default (-fhip-fp32-correctly-rounded-divide-sqrt); the
// bit-exact tests against NumPy on the 148 shipped pairs and on random data hold it to that
__device__ __forceinline__ float div_rn_f32(float a, float b) { return a / b; }
__device__ __forceinline__ float sqrt_rn_f32(float a) { return __builtin_sqrtf(a); }

// line = M x with M = F (lines in image 2 of points in image 1) or F^T; cv2.computeCorrespondEpilines
__device__ __forceinline__ void epiline(const double* __restrict__ f, bool transpose, float2 pt, float& la,
                                        float& lb, float& lc) {
  const double x = (double)pt.x, y = (double)pt.y;
  double a, b, c;
  if (transpose) {
    a = f[0] * x + f[3] * y + f[6]; b = f[1] * x + f[4] * y + f[7]; c = f[2] * x + f[5] * y + f[8];
  } else {
    a = f[0] * x + f[1] * y + f[2]; b = f[3] * x + f[4] * y + f[5]; c = f[6] * x + f[7] * y + f[8];
  }
  double nu = a * a + b * b;
  nu = (nu != 0.0) ? 1.0 / sqrt(nu) : 1.0;
  la = (float)(a * nu); lb = (float)(b * nu); lc = (float)(c * nu);
}

__global__ __launch_bounds__(256) void k_epipolar(const double* __restrict__ F, const int64_t* __restrict__ seg_ptr,
                                                  int n_seg, const float2* __restrict__ pts1,
                                                  const float2* __restrict__ pts2, int64_t n, float threshold,
                                                  float* __restrict__ err, uint8_t* __restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double* f = F + 9 * (int64_t)seg_of(seg_ptr, n_seg, i);
  const float2 a = pts1[i], b = pts2[i];
  float l1a, l1b, l1c, l2a, l2b, l2c;
  epiline(f, true, b, l1a, l1b, l1c);     // lines1 = epilines of pts2 in image 1 (find_matches.py:160)
  epiline(f, false, a, l2a, l2b, l2c);    // lines2 = epilines of pts1 in image 2 (:162)
  const float e1 = div_rn_f32(fabsf((a.x * l1a + a.y * l1b) + l1c), sqrt_rn_f32(l1a * l1a + l1b * l1b));
  const float e2 = div_rn_f32(fabsf((b.x * l2a + b.y * l2b) + l2c), sqrt_rn_f32(l2a * l2a + l2b * l2b));
  const float sym = (e1 + e2) / 2.0f;
  err[i] = sym;
  mask[i] = (sym < threshold) ? 1 : 0;
}

}  // namespace

// ================================================================================================ C ABI
static int assoc_nsplit(int64_t n_rows) {
  // enough (row block, split) workgroups for ~8 waves per SIMD on 256 CUs
  const int64_t nblk = (n_rows + 255) / 256;
  int64_t ns = (2048 + nblk - 1) / nblk;
  return (int)(ns < 1 ? 1 : ns > 32 ? 32 : ns);
}

extern "C" int sfm_assoc_workspace_bytes(int64_t n_rows, int64_t* bytes_host) {
  if (!bytes_host || n_rows < 0) return SFM_ERR_ARG;
  const int64_t nblk = (n_rows + 255) / 256;
  *bytes_host = align_up(n_rows * assoc_nsplit(n_rows) * 4, 256) + 2 * align_up(nblk * 4, 256) + 256;
  return SFM_OK;
}

extern "C" int sfm_assoc_radius(sfm_handle h, const double* track_xy, const int64_t* t_ptr, const double* corr_xy,
                                const int64_t* m_ptr, int32_t n_seg, int64_t n_rows, double radius,
                                int32_t* out_row, int32_t* out_col, int64_t capacity, int64_t* n_pairs,
                                void* workspace, int64_t workspace_bytes) {
  if (!h) return SFM_ERR_ARG;
  if (!n_pairs || n_rows < 0 || n_seg < 0 || capacity < 0 || n_rows > 0x3fffffffLL)
    return sfm_fail(h, SFM_ERR_ARG, "sfm_assoc_radius", "bad argument");
  if (n_rows == 0 || n_seg == 0) {
    SFM_HIP(h, hipMemsetAsync(n_pairs, 0, sizeof(int64_t), h->stream));
    return SFM_OK;
  }
  if (!track_xy || !t_ptr || !corr_xy || !m_ptr || !workspace || (capacity > 0 && (!out_row || !out_col)))
    return sfm_fail(h, SFM_ERR_ARG, "sfm_assoc_radius", "null pointer");
  int64_t need = 0;
  sfm_assoc_workspace_bytes(n_rows, &need);
  if (workspace_bytes < need) return sfm_fail(h, SFM_ERR_WORKSPACE, "sfm_assoc_radius", "workspace too small");
  const int nblk = (int)cdiv(n_rows, 256);
  const int nsplit = assoc_nsplit(n_rows);
  char* ws = (char*)workspace;
  int* cnt_rs = (int*)ws;                  ws += align_up(n_rows * nsplit * 4, 256);
  int* blk_cnt = (int*)ws;                 ws += align_up((int64_t)nblk * 4, 256);
  int* blk_off = (int*)ws;
  const double r2_hi = radius * radius * (1.0 + 1e-9) + DBL_MIN;
  SFM_HIP(h, hipMemsetAsync(blk_cnt, 0, (size_t)nblk * 4, h->stream));
  hipLaunchKernelGGL((k_assoc<false>), dim3(nblk, nsplit), dim3(256), 0, h->stream, (const double2*)track_xy, t_ptr,
                     (const double2*)corr_xy, m_ptr, n_seg, n_rows, nsplit, radius, r2_hi, cnt_rs, blk_cnt,
                     (const int*)nullptr, (int*)nullptr, (int*)nullptr, (int64_t)0);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, h->stream, nblk, blk_cnt, blk_off, n_pairs);
  if (capacity > 0)
    hipLaunchKernelGGL((k_assoc<true>), dim3(nblk, nsplit), dim3(256), 0, h->stream, (const double2*)track_xy, t_ptr,
                       (const double2*)corr_xy, m_ptr, n_seg, n_rows, nsplit, radius, r2_hi, cnt_rs, blk_cnt,
                       (const int*)blk_off, out_row, out_col, capacity);
  SFM_LAUNCH_CHECK(h, "sfm_assoc_radius");
  return SFM_OK;
}

extern "C" int sfm_triangulate2(sfm_handle h, const double* proj, int32_t n_cams, const int32_t* cam0,
                                const int32_t* cam1, const double* x0, const double* x1, int64_t n, double max_err,
                                double* X, int32_t* valid, double* err) {
  if (!h) return SFM_ERR_ARG;
  if (n < 0 || n_cams < 0) return sfm_fail(h, SFM_ERR_ARG, "sfm_triangulate2", "bad argument");
  if (n == 0) return SFM_OK;
  if (!proj || !cam0 || !cam1 || !x0 || !x1 || !X || !valid || n_cams < 1)
    return sfm_fail(h, SFM_ERR_ARG, "sfm_triangulate2", "null pointer");
  hipLaunchKernelGGL(k_triangulate2, dim3(cdiv(n, 256)), dim3(256), 0, h->stream, proj, cam0, cam1,
                     (const double2*)x0, (const double2*)x1, n, max_err, X, valid, err);
  SFM_LAUNCH_CHECK(h, "sfm_triangulate2");
  return SFM_OK;
}

extern "C" int sfm_epipolar_errors(sfm_handle h, const double* F, const int64_t* seg_ptr, int32_t n_seg,
                                   const float* pts1, const float* pts2, int64_t n, float threshold,
                                   float* err, uint8_t* mask) {
  if (!h) return SFM_ERR_ARG;
  if (n < 0 || n_seg < 0) return sfm_fail(h, SFM_ERR_ARG, "sfm_epipolar_errors", "bad argument");
  if (n == 0) return SFM_OK;
  if (!F || !seg_ptr || !pts1 || !pts2 || !err || !mask || n_seg < 1)
    return sfm_fail(h, SFM_ERR_ARG, "sfm_epipolar_errors", "null pointer");
  hipLaunchKernelGGL(k_epipolar, dim3(cdiv(n, 256)), dim3(256), 0, h->stream, F, seg_ptr, n_seg,
                     (const float2*)pts1, (const float2*)pts2, n, threshold, err, mask);
  SFM_LAUNCH_CHECK(h, "sfm_epipolar_errors");
  return SFM_OK;
}
