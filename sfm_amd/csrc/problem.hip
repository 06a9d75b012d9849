// Construction of a bundle-adjustment problem on the device (gfx950): from the three arrays the reference packs
// (camera index, point index, pixel per observation; /root/reference/utils/sfm_reconstruction.py:409-451) to the
// index structure the kernels of ba.hip walk:
//   pt_ptr            track (observation range) of every point            - boundaries of the point-major order
//   cam_ptr, cam_obs  observations grouped by camera, ascending            - stable sort by camera id
//   blk_ptr, pair_k, pair_k2   for every upper-triangular camera pair (c <= c2) the observation pairs (k, k2) that
//                     share a track: every (k, k2) with cam(k) <= cam(k2), ordered by (block, k, k2)
//                                                                          - enumerate per observation, stable sort by block id
//   item_*            the pair list of a block cut into work items of <= 256 pairs (one wavefront each)
//   xcd_*             work items grouped by block row mod 8 (one XCD serves one group: L2 reuse of G)
//   cch_*             the camera lists cut into chunks of <= 256 observations (one workgroup each)
// sfm_amd/structure.py is the host-side mirror (NumPy); tests/test_ba_gpu.py checks the arrays bit for bit.
// The two global stable sorts and the prefix sums are rocPRIM device primitives (set-up plumbing, once per
// problem); everything per-iteration is hand-written in ba.hip / dense.hip.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ba_internal.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#define ITEM_PAIRS 256
#define CHUNK_OBS 256

namespace {

// ---- validation + track boundaries.  err: 1 = index out of range, 2 = not point-major.
__global__ __launch_bounds__(256) void k_tracks(int64_t N, int P, int C, const int* __restrict__ cam_idx,
                                                const int* __restrict__ pt_idx, int* __restrict__ pt_ptr,
                                                int* __restrict__ cam_cnt, int* __restrict__ err) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= N) return;
  const int c = cam_idx[k], j = pt_idx[k];
  const int prev = k > 0 ? pt_idx[k - 1] : -1;
  const bool ok = c >= 0 && c < C && j >= 0 && j < P && prev >= -1 && prev < P;
  if (!ok) { atomicOr(err, 1); return; }
  if (prev > j) { atomicOr(err, 2); return; }
  atomicAdd(&cam_cnt[c], 1);                                  // integer atomics: the counts are order-independent
  for (int q = prev + 1; q <= j; ++q) pt_ptr[q] = (int)k;     // points prev+1 .. j start at observation k (empty tracks included)
  if (k == N - 1)
    for (int q = j + 1; q <= P; ++q) pt_ptr[q] = (int)N;
}

// exclusive prefix sum of a short array (n <= a few thousand) by one thread; out has n + 1 entries
__global__ void k_scan_small(const int* __restrict__ in, int n, int* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int run = 0;
  for (int i = 0; i < n; ++i) { out[i] = run; run += in[i]; }
  out[n] = run;
}

__global__ __launch_bounds__(256) void k_iota_keys(int64_t N, const int* __restrict__ cam_idx, unsigned* __restrict__ keys,
                                                   unsigned* __restrict__ vals) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= N) return;
  keys[k] = (unsigned)cam_idx[k];
  vals[k] = (unsigned)k;
}

// number of pairs (k, k2) observation k starts: the k2 of its track with cam(k2) >= cam(k)
__global__ __launch_bounds__(256) void k_pair_count(int64_t N, const int* __restrict__ cam_idx, const int* __restrict__ pt_idx,
                                                    const int* __restrict__ pt_ptr, int* __restrict__ cnt,
                                                    int* __restrict__ dup) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= N) return;
  const int j = pt_idx[k], c = cam_idx[k];
  int n = 0, same = 0;
  for (int k2 = pt_ptr[j]; k2 < pt_ptr[j + 1]; ++k2) { n += (cam_idx[k2] >= c) ? 1 : 0; same += (cam_idx[k2] == c) ? 1 : 0; }
  cnt[k] = n;
  // a camera twice on one track (the reference's dict-keyed tracks cannot produce it, the C API can): the diagonal
  // block (c, c) then also holds the cross pairs (k_a, k_b), (k_b, k_a) - see sfm_ba_prob::has_dup
  if (same > 1) atomicOr(dup, 1);
}

__global__ __launch_bounds__(256) void k_pair_fill(int64_t N, int C, const int* __restrict__ cam_idx,
                                                   const int* __restrict__ pt_idx, const int* __restrict__ pt_ptr,
                                                   const long long* __restrict__ off, unsigned* __restrict__ keys,
                                                   unsigned long long* __restrict__ vals) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= N) return;
  const int j = pt_idx[k];
  const long long c = cam_idx[k];
  long long o = off[k];
  for (int k2 = pt_ptr[j]; k2 < pt_ptr[j + 1]; ++k2) {
    const long long c2 = cam_idx[k2];
    if (c2 < c) continue;
    keys[o] = (unsigned)(c * C - c * (c - 1) / 2 + (c2 - c));        // block (c <= c2)
    vals[o] = ((unsigned long long)(unsigned)k << 32) | (unsigned)k2;
    ++o;
  }
}

// sorted keys -> CSR pointer (empty keys included), and the packed values back to two int32 arrays
__global__ __launch_bounds__(256) void k_pairs_finish(int64_t n_pairs, int64_t n_blk, const unsigned* __restrict__ keys,
                                                      const unsigned long long* __restrict__ vals, int* __restrict__ blk_ptr,
                                                      int* __restrict__ pair_k, int* __restrict__ pair_k2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pairs) return;
  const unsigned long long v = vals[i];
  pair_k[i] = (int)(v >> 32);
  pair_k2[i] = (int)(v & 0xFFFFFFFFull);
  const long long key = keys[i];
  const long long prev = i > 0 ? (long long)keys[i - 1] : -1;
  for (long long q = prev + 1; q <= key; ++q) blk_ptr[q] = (int)i;
  if (i == n_pairs - 1)
    for (long long q = key + 1; q <= n_blk; ++q) blk_ptr[q] = (int)n_pairs;
}

__global__ __launch_bounds__(256) void k_piece_count(int64_t n, const int* __restrict__ ptr, int size, int* __restrict__ npiece) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  npiece[b] = (ptr[b + 1] - ptr[b] + size - 1) / size;
}

__global__ __launch_bounds__(256) void k_piece_fill(int64_t n, const int* __restrict__ ptr, const int* __restrict__ piece_ptr,
                                                    int size, int* __restrict__ beg, int* __restrict__ end) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  const int lo = ptr[b], hi = ptr[b + 1];
  int o = piece_ptr[b];
  for (int s = lo; s < hi; s += size, ++o) { beg[o] = s; end[o] = (s + size) < hi ? (s + size) : hi; }
}

// items of block row r = item_ptr[rowstart(r + 1)] - item_ptr[rowstart(r)]
// how far from the diagonal the blocks with work lie: stat[0] += number of non-empty blocks (c, c2 > c), stat[1] += sum of c2 - c
// over them (integer sums: the order of the atomics does not matter).  One thread per block row.
__global__ void k_band_stat(int C, const int* __restrict__ item_ptr, unsigned long long* __restrict__ stat) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= C) return;
  const long long a = (long long)r * C - (long long)r * (r - 1) / 2;
  unsigned long long n = 0, d = 0;
  for (int c2 = r + 1; c2 < C; ++c2)
    if (item_ptr[a + (c2 - r) + 1] > item_ptr[a + (c2 - r)]) { ++n; d += (unsigned long long)(c2 - r); }
  if (n) { atomicAdd(stat, n); atomicAdd(stat + 1, d); }
}
__global__ void k_row_items(int C, const int* __restrict__ item_ptr, int* __restrict__ row_first, int* __restrict__ row_cnt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= C) return;
  const long long a = (long long)r * C - (long long)r * (r - 1) / 2;
  const long long b = (long long)(r + 1) * C - (long long)(r + 1) * r / 2;
  row_first[r] = item_ptr[a];
  row_cnt[r] = item_ptr[b] - item_ptr[a];
}

// xcd_items[row_base[r] + i] = row_first[r] + i: the items of a row are consecutive ids
__global__ __launch_bounds__(256) void k_xcd_fill(int C, const int* __restrict__ row_first, const int* __restrict__ row_cnt,
                                                  const int* __restrict__ row_base, int* __restrict__ xcd_items) {
  const int r = blockIdx.y;
  if (r >= C) return;
  const int n = row_cnt[r], f = row_first[r], o = row_base[r];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) xcd_items[o + i] = f + i;
}

int bits_for(uint64_t n_values) {       // bits needed to hold values 0 .. n_values - 1
  int b = 1;
  while (b < 32 && (1ull << b) < n_values) ++b;
  return b;
}

struct DevBuf {      // scoped device allocation for the construction's temporaries
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <class T> T* as() { return (T*)p; }
};

}  // namespace

#define PB_HIP(call)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) { sfm_ba_destroy_problem(p); return sfm_fail(h, SFM_ERR_HIP, #call, hipGetErrorString(e_)); } \
  } while (0)

__global__ void k_cam_pt(int64_t N, const int* __restrict__ cam_obs, const int* __restrict__ pt_idx, int* __restrict__ cam_pt) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < N) cam_pt[q] = pt_idx[cam_obs[q]];
}
__global__ void k_pair_uv(int64_t N, const int* __restrict__ cam_obs, const double2* __restrict__ uv_in, double2* __restrict__ uv_out) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < N) uv_out[cam_obs[q]] = uv_in[q];
}

extern "C" void sfm_ba_destroy_problem(sfm_ba_problem p) {
  if (!p) return;
  void* owned[] = {p->cam_idx, p->pt_idx, p->uv, p->pt_ptr, p->cam_ptr, p->cam_obs, p->cam_pt, p->blk_ptr, p->pair_k, p->pair_k2,
                   p->item_ptr, p->item_beg, p->item_end, p->xcd_ptr, p->xcd_items, p->cch_ptr, p->cch_beg, p->cch_end};
  for (void* q : owned)
    if (q) (void)hipFree(q);
  if (p->owns_workspace && p->workspace) (void)hipFree(p->workspace);
  if (p->host_sc) (void)hipHostFree(p->host_sc);
  delete p;
}

extern "C" int sfm_ba_create_problem(sfm_handle h, const sfm_ba_desc* d, sfm_ba_problem* out) {
  if (!h) return SFM_ERR_ARG;
  if (!d || !out) return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_create_problem", "null argument");
  if (d->n_cams < 1 || d->n_pts < 1 || d->n_obs < 1 || (d->cam_dim != 6 && d->cam_dim != 10))
    return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_create_problem", "bad problem sizes / cam_dim");
  if (!d->cam_idx || !d->pt_idx || !d->uv) return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_create_problem", "null index / pixel array");
  if ((int64_t)d->n_cams * d->cam_dim > 32000) return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_create_problem", "reduced system too large");
  if (d->n_obs >= (1ll << 31)) return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_create_problem", "too many observations for int32 indices");
  if (d->precision != SFM_BA_FP64 && d->precision != SFM_BA_MIXED)
    return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_create_problem", "unknown precision");
  if (d->camera_solver < SFM_CAMERA_SOLVER_AUTO || d->camera_solver > SFM_CAMERA_SOLVER_CG)
    return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_create_problem", "unknown camera_solver");
  if (d->uv_pairing != SFM_UV_AS_GIVEN && d->uv_pairing != SFM_UV_REFERENCE_PAIRING)
    return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_create_problem", "unknown uv_pairing");
  const int C = d->n_cams, P = d->n_pts;
  const int64_t N = d->n_obs;
  const int64_t n_blk = (int64_t)C * (C + 1) / 2;
  hipStream_t st = h->stream;

  sfm_ba_prob* p = new sfm_ba_prob();
  memset(p, 0, sizeof(*p));
  p->h = h;
  p->n_cams = C; p->n_pts = P; p->cam_dim = d->cam_dim; p->apply_reg = d->apply_reg; p->precision = d->precision;
  p->camera_solver = d->camera_solver;
  p->einv_alpha = -1.0; p->st_alpha = -1.0; p->s_valid = 1;
  // pinned host mirror of the scalars (every kernel that writes one writes it here too: sfm_ba_read_scalars only waits)
  if (hipHostMalloc((void**)&p->host_sc, SFM_HSC_WORDS * sizeof(double), hipHostMallocDefault) != hipSuccess) {
    p->host_sc = nullptr; sfm_ba_destroy_problem(p);
    return sfm_fail(h, SFM_ERR_HIP, "sfm_ba_create_problem", "pinned host memory for the scalars");
  }
  memset(p->host_sc, 0, SFM_HSC_WORDS * sizeof(double));
  p->n_obs = N;
  p->fx0 = d->fx0; p->fy0 = d->fy0; p->cx0 = d->cx0; p->cy0 = d->cy0;
  p->width = d->width; p->height = d->height; p->reg_weight = d->reg_weight;

  // ---- the caller's arrays (host or device pointers) become the problem's own
  PB_HIP(hipMalloc((void**)&p->cam_idx, N * 4));
  PB_HIP(hipMalloc((void**)&p->pt_idx, N * 4));
  PB_HIP(hipMalloc((void**)&p->uv, N * 16));
  PB_HIP(hipMemcpyAsync(p->cam_idx, d->cam_idx, N * 4, hipMemcpyDefault, st));
  PB_HIP(hipMemcpyAsync(p->pt_idx, d->pt_idx, N * 4, hipMemcpyDefault, st));
  PB_HIP(hipMemcpyAsync(p->uv, d->uv, N * 16, hipMemcpyDefault, st));

  // ---- tracks, camera counts, validation
  PB_HIP(hipMalloc((void**)&p->pt_ptr, ((int64_t)P + 1) * 4));
  PB_HIP(hipMalloc((void**)&p->cam_ptr, ((int64_t)C + 1) * 4));
  DevBuf small;                       // cam_cnt [C] | err [1] | row_first [C] | row_cnt [C] | row_base [C]
  PB_HIP(small.alloc(((int64_t)4 * C + 8) * 4));
  int* cam_cnt = small.as<int>();
  int* err = cam_cnt + C;
  int* row_first = err + 1;
  int* row_cnt = row_first + C;
  int* row_base = row_cnt + C;
  PB_HIP(hipMemsetAsync(cam_cnt, 0, ((int64_t)C + 1) * 4, st));
  PB_HIP(hipMemsetAsync(p->pt_ptr, 0, ((int64_t)P + 1) * 4, st));
  hipLaunchKernelGGL(k_tracks, dim3(cdiv(N, 256)), dim3(256), 0, st, N, P, C, p->cam_idx, p->pt_idx, p->pt_ptr, cam_cnt, err);
  hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1), 0, st, cam_cnt, C, p->cam_ptr);
  {
    // nothing below may run on unchecked indices (an out-of-range point id would send the pair kernels out of bounds)
    int herr = 0;
    PB_HIP(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, st));
    PB_HIP(hipStreamSynchronize(st));
    if (herr) {
      sfm_ba_destroy_problem(p);
      return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_create_problem",
                      (herr & 1) ? "camera / point index out of range" : "observations must be point-major (pt_idx non-decreasing)");
    }
  }

  // ---- cam_obs: stable sort of the observation ids by camera
  PB_HIP(hipMalloc((void**)&p->cam_obs, N * 4));
  DevBuf ck_in, ck_out, cv_in, tmp;
  PB_HIP(ck_in.alloc(N * 4)); PB_HIP(ck_out.alloc(N * 4)); PB_HIP(cv_in.alloc(N * 4));
  hipLaunchKernelGGL(k_iota_keys, dim3(cdiv(N, 256)), dim3(256), 0, st, N, p->cam_idx, ck_in.as<unsigned>(), cv_in.as<unsigned>());
  size_t tmp_bytes = 0;
  PB_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ck_in.as<unsigned>(), ck_out.as<unsigned>(), cv_in.as<unsigned>(),
                                   (unsigned*)p->cam_obs, (size_t)N, 0, (unsigned)bits_for((uint64_t)C), st));
  PB_HIP(tmp.alloc(tmp_bytes));
  PB_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, ck_in.as<unsigned>(), ck_out.as<unsigned>(), cv_in.as<unsigned>(),
                                   (unsigned*)p->cam_obs, (size_t)N, 0, (unsigned)bits_for((uint64_t)C), st));

  PB_HIP(hipMalloc((void**)&p->cam_pt, N * 4));
  hipLaunchKernelGGL(k_cam_pt, dim3(cdiv(N, 256)), dim3(256), 0, st, N, p->cam_obs, p->pt_idx, p->cam_pt);

  // ---- the reference's residual pairing (sfm_reconstruction.py:480-486): the q-th observation of the camera-sorted list
  // meets the q-th point-major pixel - a scatter through cam_obs (the host's np.argsort + scatter took 55 ms at 1M observations)
  if (d->uv_pairing == SFM_UV_REFERENCE_PAIRING) {
    DevBuf uv_in;
    PB_HIP(uv_in.alloc(N * 16));
    PB_HIP(hipMemcpyAsync(uv_in.p, p->uv, N * 16, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_pair_uv, dim3(cdiv(N, 256)), dim3(256), 0, st, N, p->cam_obs, (const double2*)uv_in.p, (double2*)p->uv);
    PB_HIP(hipStreamSynchronize(st));            // uv_in is released at the end of this scope
  }

  // ---- pairs: count per observation, prefix sum (64-bit), fill, stable sort by block id
  DevBuf cnt, off;
  PB_HIP(cnt.alloc((N + 1) * 4)); PB_HIP(off.alloc((N + 1) * 8));
  PB_HIP(hipMemsetAsync(cnt.p, 0, (N + 1) * 4, st));
  PB_HIP(hipMemsetAsync(err, 0, 4, st));         // (checked and found 0 above) re-used as the duplicate flag
  hipLaunchKernelGGL(k_pair_count, dim3(cdiv(N, 256)), dim3(256), 0, st, N, p->cam_idx, p->pt_idx, p->pt_ptr, cnt.as<int>(), err);
  {
    size_t sb = 0;
    PB_HIP(rocprim::exclusive_scan(nullptr, sb, cnt.as<int>(), off.as<long long>(), 0ll, (size_t)N + 1,
                                   rocprim::plus<long long>(), st));
    DevBuf stmp; PB_HIP(stmp.alloc(sb));
    PB_HIP(rocprim::exclusive_scan(stmp.p, sb, cnt.as<int>(), off.as<long long>(), 0ll, (size_t)N + 1,
                                   rocprim::plus<long long>(), st));
    long long total = 0;
    int hdup = 0;
    PB_HIP(hipMemcpyAsync(&total, off.as<long long>() + N, 8, hipMemcpyDeviceToHost, st));
    PB_HIP(hipMemcpyAsync(&hdup, err, 4, hipMemcpyDeviceToHost, st));
    PB_HIP(hipStreamSynchronize(st));       // stmp is released after the scan has run
    p->has_dup = hdup ? 1 : 0;
    if (total >= (1ll << 31)) { sfm_ba_destroy_problem(p); return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_create_problem", "too many camera pairs for int32 indices"); }
    p->n_pairs = total;
  }
  const int64_t NP = p->n_pairs;
  PB_HIP(hipMalloc((void**)&p->blk_ptr, (n_blk + 1) * 4));
  PB_HIP(hipMalloc((void**)&p->pair_k, NP * 4));
  PB_HIP(hipMalloc((void**)&p->pair_k2, NP * 4));
  {
    DevBuf pk_in, pk_out, pv_in, pv_out, ptmp;
    PB_HIP(pk_in.alloc(NP * 4)); PB_HIP(pk_out.alloc(NP * 4)); PB_HIP(pv_in.alloc(NP * 8)); PB_HIP(pv_out.alloc(NP * 8));
    hipLaunchKernelGGL(k_pair_fill, dim3(cdiv(N, 256)), dim3(256), 0, st, N, C, p->cam_idx, p->pt_idx, p->pt_ptr,
                       off.as<long long>(), pk_in.as<unsigned>(), pv_in.as<unsigned long long>());
    size_t sb = 0;
    const unsigned bits = (unsigned)bits_for((uint64_t)n_blk);
    PB_HIP(rocprim::radix_sort_pairs(nullptr, sb, pk_in.as<unsigned>(), pk_out.as<unsigned>(), pv_in.as<unsigned long long>(),
                                     pv_out.as<unsigned long long>(), (size_t)NP, 0, bits, st));
    PB_HIP(ptmp.alloc(sb));
    PB_HIP(rocprim::radix_sort_pairs(ptmp.p, sb, pk_in.as<unsigned>(), pk_out.as<unsigned>(), pv_in.as<unsigned long long>(),
                                     pv_out.as<unsigned long long>(), (size_t)NP, 0, bits, st));
    PB_HIP(hipMemsetAsync(p->blk_ptr, 0, (n_blk + 1) * 4, st));
    hipLaunchKernelGGL(k_pairs_finish, dim3(cdiv(NP, 256)), dim3(256), 0, st, NP, n_blk, pk_out.as<unsigned>(),
                       pv_out.as<unsigned long long>(), p->blk_ptr, p->pair_k, p->pair_k2);
    PB_HIP(hipStreamSynchronize(st));       // the sort buffers go out of scope here
  }

  // ---- work items (<= 256 pairs of one block)
  PB_HIP(hipMalloc((void**)&p->item_ptr, (n_blk + 1) * 4));
  {
    DevBuf npiece, stmp;
    PB_HIP(npiece.alloc((n_blk + 1) * 4));
    PB_HIP(hipMemsetAsync(npiece.p, 0, (n_blk + 1) * 4, st));
    hipLaunchKernelGGL(k_piece_count, dim3(cdiv(n_blk, 256)), dim3(256), 0, st, n_blk, p->blk_ptr, ITEM_PAIRS, npiece.as<int>());
    size_t sb = 0;
    PB_HIP(rocprim::exclusive_scan(nullptr, sb, npiece.as<int>(), p->item_ptr, 0, (size_t)n_blk + 1, rocprim::plus<int>(), st));
    PB_HIP(stmp.alloc(sb));
    PB_HIP(rocprim::exclusive_scan(stmp.p, sb, npiece.as<int>(), p->item_ptr, 0, (size_t)n_blk + 1, rocprim::plus<int>(), st));
    hipLaunchKernelGGL(k_row_items, dim3(cdiv(C, 64)), dim3(64), 0, st, C, p->item_ptr, row_first, row_cnt);
    PB_HIP(hipStreamSynchronize(st));
  }
  std::vector<int> h_cam_ptr(C + 1), h_row_cnt(C);
  int n_items = 0;
  PB_HIP(hipMemcpy(&n_items, p->item_ptr + n_blk, 4, hipMemcpyDeviceToHost));
  PB_HIP(hipMemcpy(h_cam_ptr.data(), p->cam_ptr, ((size_t)C + 1) * 4, hipMemcpyDeviceToHost));
  PB_HIP(hipMemcpy(h_row_cnt.data(), row_cnt, (size_t)C * 4, hipMemcpyDeviceToHost));
  p->n_items = n_items;
  PB_HIP(hipMalloc((void**)&p->item_beg, ((int64_t)n_items + 1) * 4));
  PB_HIP(hipMalloc((void**)&p->item_end, ((int64_t)n_items + 1) * 4));
  PB_HIP(hipMalloc((void**)&p->xcd_items, ((int64_t)n_items + 1) * 4));
  PB_HIP(hipMalloc((void**)&p->xcd_ptr, 9 * 4));
  hipLaunchKernelGGL(k_piece_fill, dim3(cdiv(n_blk, 256)), dim3(256), 0, st, n_blk, p->blk_ptr, p->item_ptr, ITEM_PAIRS,
                     p->item_beg, p->item_end);
  // items grouped by block row: 8 groups, one per XCD (workgroup b of the Schur kernel serves group b % 8); rows (and the
  // items inside a row) stay in ascending order inside a group: C numbers, on the host.  Default: rows dealt back and forth over the groups.
  // SFM_XCD_GROUP=contig: contiguous row ranges balanced by item count - neighbouring block rows share their partner
  // cameras when the camera numbering is spatially coherent (a capture order), so one XCD's L2 sees them together.
  {
    // Which of the two: a locality statistic of the problem decides (SFM_XCD_GROUP=contig / mod8 overrides it).  With a capture
    // order for a camera numbering the blocks with work lie nearer the diagonal - their mean distance from it is below the C / 3 of
    // blocks spread evenly over the triangle - and contiguous row groups then pay (Schur gather 329 -> 320 us on the spatially
    // coherent bench scene: mean distance 0.61 x C / 3, 3,522 of 19,900 blocks non-empty); on evenly spread blocks (the BASELINE
    // scene: 1.0 x) they do nothing.  The line is drawn at 0.8.
    const char* ge = getenv("SFM_XCD_GROUP");
    bool contig = ge && ge[0] == 'c';
    if (!ge && C >= 16) {
      unsigned long long* d_stat = nullptr;
      unsigned long long h_stat[2] = {0, 0};
      PB_HIP(hipMalloc((void**)&d_stat, 16));
      PB_HIP(hipMemsetAsync(d_stat, 0, 16, st));
      hipLaunchKernelGGL(k_band_stat, dim3(cdiv(C, 64)), dim3(64), 0, st, C, p->item_ptr, d_stat);
      PB_HIP(hipMemcpyAsync(h_stat, d_stat, 16, hipMemcpyDeviceToHost, st));
      PB_HIP(hipStreamSynchronize(st));
      (void)hipFree(d_stat);
      if (h_stat[0] > 0) contig = (double)h_stat[1] / (double)h_stat[0] < 0.8 * ((double)C / 3.0);
    }
    std::vector<int> grp(C);
    if (contig) {
      long long total = 0, run = 0;
      for (int r = 0; r < C; ++r) total += h_row_cnt[r];
      for (int r = 0; r < C; ++r) {
        const long long mid = run + h_row_cnt[r] / 2;                  // the group the middle of the row falls into
        int g = total > 0 ? (int)((mid * 8) / total) : 0;
        grp[r] = g > 7 ? 7 : g;
        run += h_row_cnt[r];
      }
    } else {
      // dealt back and forth (rows 0..7 -> groups 0..7, rows 8..15 -> groups 7..0, ...): row r holds C - r blocks, so plain
      // r mod 8 gives group 0 5 % more pairs than group 7 at 200 cameras (wave stamps: the XCDs finished 12 us apart)
      for (int r = 0; r < C; ++r) grp[r] = (r & 8) ? 7 - (r & 7) : (r & 7);
    }
    int xcd_ptr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < C; ++r) xcd_ptr[grp[r] + 1] += h_row_cnt[r];
    int64_t mx = 0;
    for (int g = 0; g < 8; ++g) { mx = xcd_ptr[g + 1] > mx ? xcd_ptr[g + 1] : mx; xcd_ptr[g + 1] += xcd_ptr[g]; }
    p->xcd_max_items = mx;
    std::vector<int> h_row_base(C);
    int run[8];
    for (int g = 0; g < 8; ++g) run[g] = xcd_ptr[g];
    for (int r = 0; r < C; ++r) { h_row_base[r] = run[grp[r]]; run[grp[r]] += h_row_cnt[r]; }
    PB_HIP(hipMemcpyAsync(p->xcd_ptr, xcd_ptr, 9 * 4, hipMemcpyHostToDevice, st));
    PB_HIP(hipMemcpyAsync(row_base, h_row_base.data(), (size_t)C * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_xcd_fill, dim3(4, C), dim3(256), 0, st, C, row_first, row_cnt, row_base, p->xcd_items);
    PB_HIP(hipStreamSynchronize(st));       // the host vectors above are pageable: the copies must have left them
  }

  // ---- camera chunks (<= 256 observations of one camera): C + 1 numbers, on the host
  {
    std::vector<int> cch_ptr(C + 1, 0);
    for (int c = 0; c < C; ++c) cch_ptr[c + 1] = cch_ptr[c] + (h_cam_ptr[c + 1] - h_cam_ptr[c] + CHUNK_OBS - 1) / CHUNK_OBS;
    const int n_ch = cch_ptr[C];
    p->n_cchunks = n_ch;
    std::vector<int> beg(n_ch + 1), end(n_ch + 1);
    for (int c = 0, o = 0; c < C; ++c)
      for (int s = h_cam_ptr[c]; s < h_cam_ptr[c + 1]; s += CHUNK_OBS, ++o) {
        beg[o] = s;
        end[o] = (s + CHUNK_OBS) < h_cam_ptr[c + 1] ? (s + CHUNK_OBS) : h_cam_ptr[c + 1];
      }
    PB_HIP(hipMalloc((void**)&p->cch_ptr, ((int64_t)C + 1) * 4));
    PB_HIP(hipMalloc((void**)&p->cch_beg, ((int64_t)n_ch + 1) * 4));
    PB_HIP(hipMalloc((void**)&p->cch_end, ((int64_t)n_ch + 1) * 4));
    PB_HIP(hipMemcpy(p->cch_ptr, cch_ptr.data(), ((size_t)C + 1) * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(p->cch_beg, beg.data(), ((size_t)n_ch + 1) * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(p->cch_end, end.data(), ((size_t)n_ch + 1) * 4, hipMemcpyHostToDevice));
  }
  {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sfm_ba_destroy_problem(p); return sfm_fail(h, SFM_ERR_HIP, "sfm_ba_create_problem", hipGetErrorString(e)); }
  }
  p->L = ba_layout(C, P, N, d->cam_dim, p->n_items, p->n_cchunks, p->precision);
  *out = p;
  return SFM_OK;
}

extern "C" int sfm_ba_solver_stats(sfm_ba_problem p, int64_t* cg_iters, int64_t* cg_fallbacks) {
  if (!p || !cg_iters || !cg_fallbacks) return SFM_ERR_ARG;
  *cg_iters = p->cg_iters; *cg_fallbacks = p->cg_fallbacks;
  return SFM_OK;
}

extern "C" int sfm_ba_pcg_stats(sfm_ba_problem p, int64_t* fallbacks, double* worst_relres) {
  if (!p || !fallbacks || !worst_relres) return SFM_ERR_ARG;
  *fallbacks = p->pcg_fallbacks; *worst_relres = p->pcg_worst_relres;
  return SFM_OK;
}

extern "C" int sfm_ba_get_structure(sfm_ba_problem p, sfm_ba_structure* o) {
  if (!p || !o) return SFM_ERR_ARG;
  o->n_obs = p->n_obs; o->n_pairs = p->n_pairs; o->n_items = p->n_items; o->n_cchunks = p->n_cchunks;
  o->xcd_max_items = p->xcd_max_items;
  o->pt_ptr = p->pt_ptr; o->cam_ptr = p->cam_ptr; o->cam_obs = p->cam_obs; o->blk_ptr = p->blk_ptr;
  o->pair_k = p->pair_k; o->pair_k2 = p->pair_k2;
  o->item_ptr = p->item_ptr; o->item_beg = p->item_beg; o->item_end = p->item_end;
  o->xcd_ptr = p->xcd_ptr; o->xcd_items = p->xcd_items;
  o->cch_ptr = p->cch_ptr; o->cch_beg = p->cch_beg; o->cch_end = p->cch_end;
  return SFM_OK;
}

extern "C" int sfm_ba_bind_workspace(sfm_handle h, sfm_ba_problem p, void* workspace, int64_t workspace_bytes) {
  if (!h) return SFM_ERR_ARG;
  if (!p) return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_bind_workspace", "null problem");
  const int64_t need = p->L.total * 8;
  if (p->owns_workspace && p->workspace) { (void)hipFree(p->workspace); p->workspace = nullptr; p->owns_workspace = 0; }
  if (!workspace) {
    SFM_HIP(h, hipMalloc(&p->workspace, (size_t)need));
    SFM_HIP(h, hipMemsetAsync(p->workspace, 0, (size_t)need, h->stream));
    p->owns_workspace = 1;
    p->workspace_bytes = need;
    return SFM_OK;
  }
  if (workspace_bytes < need) return sfm_fail(h, SFM_ERR_WORKSPACE, "sfm_ba_bind_workspace", "workspace too small");
  p->workspace = workspace;
  p->workspace_bytes = workspace_bytes;
  // A caller-owned workspace may have served another problem (or another handle) before.  The persistent camera CG validates
  // what it reads from its mailbox by tags alone, so the mailbox must not hold a previous owner's granules: cleared here, with
  // the CG status words (the salts are process-wide and never repeat either - cgs_persist_launch).
  SFM_HIP(h, hipMemsetAsync((double*)workspace + p->L.cg_mail, 0, (size_t)(4 * (int64_t)p->n_cams * p->cam_dim) * sizeof(double), h->stream));
  SFM_HIP(h, hipMemsetAsync((double*)workspace + p->L.cg_scal, 0, 16 * sizeof(double), h->stream));
  return SFM_OK;
}
