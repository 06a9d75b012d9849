/* sfm_amd.h - C-ABI of libsfm_amd.so (MI355X / gfx950 hot path of Sovik-Ghosh/SFM).
 *
 * The reference has no FFI layer: its hot path is two Python methods,
 *   ImageMatcher.match_features          /root/reference/utils/find_matches.py:141-155
 *   StructureFromMotion.bundle_adjust    /root/reference/utils/sfm_reconstruction.py:401-549
 * (plus compute_reconstruction_stats :582-631, a by-product of the residual kernel).
 * This header is what a ctypes binding for those two methods binds instead of
 * cv2.BFMatcher.knnMatch (find_matches.py:144-147) and scipy.optimize.least_squares
 * (sfm_reconstruction.py:506-514).  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: every data pointer is a DEVICE pointer (HBM resident, caller-allocated,
 * caller-owned) unless its name ends in _host.  All work is enqueued on the handle's
 * HIP stream (sfm_set_stream; default = the null stream); functions return without
 * synchronising unless stated.  Return value 0 = OK, <0 = error (sfm_last_error).
 * One handle per (process, device); a handle is not thread-safe.
 */
#ifndef SFM_AMD_H
#define SFM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sfm_ctx* sfm_handle;

enum { SFM_OK = 0, SFM_ERR_ARG = -1, SFM_ERR_HIP = -2, SFM_ERR_WORKSPACE = -3, SFM_ERR_NUMERIC = -4 };

int         sfm_create(int device, sfm_handle* out);
void        sfm_destroy(sfm_handle h);
const char* sfm_last_error(sfm_handle h);
int         sfm_set_stream(sfm_handle h, void* hip_stream);
int         sfm_synchronize(sfm_handle h);
const char* sfm_version(void);
/* Stream-ordered copy of device memory the library owns (e.g. sfm_ba_get_structure) to the host; synchronises. */
int         sfm_copy_to_host(sfm_handle h, void* dst_host, const void* src_device, int64_t bytes);

/* The persistent conjugate-gradient kernel of the camera solve (SFM_CAMERA_SOLVER_CG, n <= 2048) needs its n / 8 workgroups
 * co-resident.  A launch whose workgroups give up waiting for each other (bounded spins; e.g. the CUs are shared with another
 * process) is abandoned, the solve falls back to one launch per iteration and the handle stops using the persistent kernel.
 * sfm_cgs_persist_enable switches it back on (or off); sfm_cgs_persist_enabled reports the state. */
int sfm_cgs_persist_enable(sfm_handle h, int enabled);
int sfm_cgs_persist_enabled(sfm_handle h);

/* Per-kernel device timing with HIP events recorded on the handle's stream (what bench.py's
 * roofline numbers are computed from).  Off by default.  sfm_profile_read synchronises the stream,
 * returns the accumulated milliseconds and launch count of one slot and resets it. */
enum { SFM_PROF_LIN_OBS = 0,   /* k_lin_obs: residual + Jacobian + Huber scaling, one launch per linearisation */
       SFM_PROF_LIN_REST = 1,  /* per-point / per-camera block sums of a linearisation */
       SFM_PROF_BUILD_G = 2,   /* point factors + G = W L^-T */
       SFM_PROF_SCHUR = 3,     /* reduced camera system S, r */
       SFM_PROF_CHOL = 4,      /* camera solve for the step: CG on the scaled system, or the bordered Cholesky of S */
       SFM_PROF_TRSV = 5,      /* second system (q term) by CG, or the triangular solves with the factor */
       SFM_PROF_BACKSUB = 6,   /* point back-substitution and the q pieces */
       SFM_PROF_STEP = 7,      /* trial step: x + s, predicted reduction sums, cost(x + s) */
       SFM_PROF_KNN = 8,       /* matcher distance + top-2 kernel */
       SFM_PROF_SCHUR_ITEMS = 9, /* k_schur_items alone (inside SFM_PROF_SCHUR) */
       SFM_PROF_COUNT = 10 };
int sfm_set_profiling(sfm_handle h, int enabled);
int sfm_profile_read(sfm_handle h, int slot, double* total_ms_host, int64_t* count_host);

/* ------------------------------------------------------------------ matcher
 * Replaces cv2.BFMatcher(norm).knnMatch(desc1, desc2, k=2) + the ratio loop
 * (find_matches.py:144-153).  Distances are float32 as OpenCV's DMatch.distance.
 */
enum {
  SFM_METRIC_L2_U8   = 0,  /* uint8 [n,dim], dim % 32 == 0 (SIFT: 128): exact integer d^2 on i8 MFMA */
  SFM_METRIC_L2_F32  = 1,  /* float32 [n,dim]: sequential float32 sum of (a-b)^2 (general floats)   */
  SFM_METRIC_HAMMING = 2   /* uint8 [n,dim] bit strings, dim 16 / 32 (ORB) / 64 bytes: popcount distance;
                              128 / 256 bits run as exact uint8 L2 over unpacked bits on the i8 MFMA path       */
};

int sfm_match_workspace_bytes(int metric, int64_t nq, int64_t nt, int dim, int64_t* bytes_host);

/* Per query row: two nearest train rows (idx1/d1 nearest), ties -> lower train index.
 * d1/d2: L2 = sqrtf(d^2), Hamming = bit count.  Needs nt >= 2. */
int sfm_match_knn2(sfm_handle h, int metric, const void* q, int64_t nq, const void* t, int64_t nt,
                   int dim, int32_t* idx1, int32_t* idx2, float* d1, float* d2,
                   void* workspace, int64_t workspace_bytes);

/* Lowe ratio test `(double)d1 < ratio * (double)d2` (find_matches.py:152) and compaction in
 * query order.  query_idx/train_idx/dist have room for nq entries; *n_matches is a device int64. */
int sfm_match_ratio(sfm_handle h, int64_t nq, const int32_t* idx1, const float* d1, const float* d2,
                    double ratio, int32_t* query_idx, int32_t* train_idx, float* dist,
                    int64_t* n_matches, void* workspace, int64_t workspace_bytes);

/* Batched form: every image pair of a preprocessing step in ONE call.  The reference calls match_features once per
 * pair in a serial loop (find_matches.py:329-350, call at :272) on sets of a few hundred to a few thousand
 * descriptors, where a launch per pair is all overhead.  Segment s (= one pair) matches the query rows
 * [q_beg[s], q_end[s]) of q against the train rows [t_beg[s], t_end[s]) of t (q and t may be the same array holding
 * the descriptors of all images back to back).  Output row out_ptr[s] + i belongs to query row q_beg[s] + i;
 * out_ptr = exclusive prefix sum of the query counts, n_out = out_ptr[n_seg].  idx1 / idx2 are train indices
 * RELATIVE to t_beg[s] (what DMatch.trainIdx is for that pair).  Results per segment are bit-identical to a
 * sfm_match_knn2 call on that pair.  The four segment arrays are HOST pointers (the caller knows its image sizes);
 * out_ptr_device (optional, [n_seg+1] device int64) receives out_ptr for sfm_match_ratio_batched. */
int sfm_match_batched_workspace_bytes(int metric, int32_t n_seg, const int64_t* q_beg_host, const int64_t* q_end_host,
                                      const int64_t* t_beg_host, const int64_t* t_end_host, int64_t nq_rows,
                                      int64_t nt_rows, int64_t* n_out_host, int64_t* bytes_host);
int sfm_match_knn2_batched(sfm_handle h, int metric, const void* q, int64_t nq_rows, const void* t, int64_t nt_rows,
                           int dim, int32_t n_seg, const int64_t* q_beg_host, const int64_t* q_end_host,
                           const int64_t* t_beg_host, const int64_t* t_end_host, int32_t* idx1, int32_t* idx2,
                           float* d1, float* d2, int64_t* out_ptr_device, void* workspace, int64_t workspace_bytes);
/* Ratio test + compaction over all n_out rows of a batch: the matches of segment s are entries
 * [seg_match_ptr[s], seg_match_ptr[s+1]) (device int64 [n_seg+1]) of query_idx / train_idx / dist, query indices
 * relative to the segment (DMatch.queryIdx), in query order.  workspace: ceil(n_out / 256) * 8 + 64 bytes. */
int sfm_match_ratio_batched(sfm_handle h, int64_t n_out, int32_t n_seg, const int64_t* out_ptr_device, const int32_t* idx1,
                            const float* d1, const float* d2, double ratio, int32_t* query_idx, int32_t* train_idx,
                            float* dist, int64_t* seg_match_ptr_device, void* workspace, int64_t workspace_bytes);

/* float32 descriptors whose every value is an integer in [0,255] (what SIFT emits) -> uint8 copy;
 * *all_integral (device int32) is 0 if any value is not such an integer. */
int sfm_match_f32_to_u8(sfm_handle h, const float* src, int64_t n_elems, uint8_t* dst, int32_t* all_integral);

/* ------------------------------------------------------------------ bundle adjustment
 * Replaces what scipy.optimize.least_squares does for bundle_adjust: evaluation of the
 * closure `objective` (sfm_reconstruction.py:472-501), its Jacobian, the Huber scaling
 * (scipy _lsq/common.py:720-731), the damped step (H + alpha I) p = -g of the exact
 * trust-region solver (scipy _lsq/common.py:57-168), done block-sparse with a Schur complement,
 * and the trust-region loop itself (scipy _lsq/trf.py:401-560): sfm_ba_run_trf is the single call
 * that stands where `optimize.least_squares(...)` stands at sfm_reconstruction.py:506-514.
 * The stages are exported too: a multi-rank host (points sharded over GPUs, cameras replicated)
 * all-reduces the regions named `reduce_*` between them, either itself (sfm_amd/ba.py) or through
 * the sfm_reduce_fn hook of the trust-region loop.
 *
 * Parameter vector x = [cams (n_cams*cam_dim) | pts (n_pts*3)] float64, camera block
 * [rvec(3), t(3), fx, fy, cx, cy] for cam_dim 10 (reference, :416-427) or [rvec, t] for 6.
 * Observations are in the reference's point-major order (:430-435): pt_idx non-decreasing.
 */

/* What bundle_adjust packs before it calls SciPy (sfm_reconstruction.py:409-451) - nothing kernel-specific.
 * cam_idx / pt_idx / uv may be host or device pointers; sfm_ba_create_problem copies them. */
enum { SFM_CAMERA_SOLVER_AUTO = 0,       /* CG on the block-scaled system (n = n_cams * cam_dim even), the factorisation as its fallback (default) */
       SFM_CAMERA_SOLVER_CHOLESKY = 1,   /* bordered dense Cholesky + triangular solves */
       SFM_CAMERA_SOLVER_CG = 2 };       /* conjugate gradients on the block-scaled system, relative residual 1e-13 (~25 iterations
                                            at 200 cameras, ~40 at 1000): ONE persistent launch per system for n <= 2048 (the
                                            matrix rows in registers, the product all-gathered between workgroups through
                                            self-validating 8-byte granules); beyond, three launches per iteration that stream
                                            the 128 x 128 tiles of the LOWER triangle only (a tile serves both products it takes
                                            part in; partial sums added in fixed order); falls back to the factorisation when it
                                            does not converge (160 / 400 iterations) or meets non-positive curvature */
enum { SFM_BA_FP64 = 0,    /* every intermediate in float64 (default; the reference's arithmetic) */
       SFM_BA_MIXED = 1 }; /* Jacobian rows (and scaled residuals) stored in float32; every sum, W L^-T, S and the solve in float64 */
enum { SFM_UV_AS_GIVEN = 0,           /* observation k is compared with uv[k] */
       SFM_UV_REFERENCE_PAIRING = 1 }; /* the reference's own residual (sfm_reconstruction.py:480-486): projections are stacked camera
                                          by camera, `points2D` stays point-major, so the q-th observation in stable camera-sorted
                                          order meets uv[q].  Applied on the device from the camera-sorted list the problem builds
                                          anyway.  The pairing is a permutation of ALL observations: a sharded host applies it
                                          before it shards (sfm_amd.reconstruction.reference_pairing) and passes SFM_UV_AS_GIVEN */
typedef struct {
  int32_t n_cams, n_pts, cam_dim, apply_reg;   /* apply_reg: add the 4 regulariser rows per camera (:489-499); rank 0 only */
  int64_t n_obs;
  const int32_t* cam_idx;    /* [n_obs] */
  const int32_t* pt_idx;     /* [n_obs] non-decreasing */
  const double*  uv;         /* [n_obs*2] pixel each observation is compared with */
  double fx0, fy0, cx0, cy0; /* pre-BA self.K (:492-497); intrinsics of every camera when cam_dim == 6 */
  double width, height, reg_weight;
  int32_t precision;         /* SFM_BA_FP64 | SFM_BA_MIXED */
  int32_t camera_solver;     /* how sfm_ba_schur_solve solves the formed n x n camera system: SFM_CAMERA_SOLVER_* */
  int32_t uv_pairing;        /* SFM_UV_AS_GIVEN | SFM_UV_REFERENCE_PAIRING */
  int32_t reserved;          /* 0 */
} sfm_ba_desc;

typedef struct sfm_ba_prob* sfm_ba_problem;    /* opaque; owns its index structure (device memory) */

/* Multi-rank hook (points sharded over GPUs, cameras replicated): all-reduce `count` doubles at device pointer
 * `data` (inside the bound workspace) in place across the ranks: op 0 = SUM, 1 = MAX.  Return 0 on success.
 * NULL = single rank. */
typedef int (*sfm_reduce_fn)(void* user, void* data, int64_t count, int op);

/* ---- collectives inside the library (multi-rank: points sharded over GPUs, cameras replicated - SURVEY.md section 8e).
 * RCCL (= NCCL's API over xGMI) is dlopen'ed on first use, so nothing here is needed on one GPU.  One communicator per
 * handle; sfm_comm_allreduce enqueues ncclAllReduce (float64, in place, op 0 = SUM / 1 = MAX) on the handle's stream: no
 * host synchronisation.  sfm_comm_reduce_hook IS an sfm_reduce_fn: pass it as `reduce` with the handle as `reduce_user`
 * and every exchange of the trust-region loop / of sfm_ba_solve_pcg runs on the stream between the stages.
 * Bootstrap like any NCCL program: rank 0 calls sfm_comm_unique_id and ships the 128 bytes to the other ranks by whatever
 * the host has (MPI, a file, torch.distributed), then every rank calls sfm_comm_init_rank (collective).  A host that
 * already owns an ncclComm_t for the handle's device hands it over with sfm_comm_adopt (not destroyed by the library). */
enum { SFM_COMM_ID_BYTES = 128 };
int sfm_comm_unique_id(sfm_handle h, void* id_host);
int sfm_comm_init_rank(sfm_handle h, const void* id_host, int32_t n_ranks, int32_t rank);
int sfm_comm_adopt(sfm_handle h, void* nccl_comm, int32_t n_ranks, int32_t rank);
int sfm_comm_destroy(sfm_handle h);
int sfm_comm_info(sfm_handle h, int32_t* n_ranks_host, int32_t* rank_host);      /* 0 ranks: no communicator */
int sfm_comm_allreduce(sfm_handle h, double* data, int64_t count, int op);
int sfm_comm_reduce_hook(void* handle_as_user, void* data, int64_t count, int op);

/* Validates the indices and builds, ON THE DEVICE, everything the kernels need besides the arrays above:
 * per-point / per-camera observation lists, the camera-pair lists of the Schur complement and their split
 * into work items (sfm_ba_structure shows them).  Synchronises the stream (sizes are data-dependent).
 * SFM_ERR_ARG for out-of-range or non point-major indices. */
int  sfm_ba_create_problem(sfm_handle h, const sfm_ba_desc* desc, sfm_ba_problem* out);
void sfm_ba_destroy_problem(sfm_ba_problem p);

/* The index structure as built (device pointers, int32), for inspection and tests; sfm_amd/structure.py is
 * its host-side mirror and produces bit-identical arrays. */
typedef struct {
  int64_t n_obs, n_pairs, n_items, n_cchunks, xcd_max_items;
  const int32_t* pt_ptr;     /* [n_pts+1]  obs range of each point (track) */
  const int32_t* cam_ptr;    /* [n_cams+1] ranges into cam_obs */
  const int32_t* cam_obs;    /* [n_obs] observation ids grouped by camera, ascending inside a camera */
  const int32_t* blk_ptr;    /* [n_cams*(n_cams+1)/2 + 1] ranges into pair_k/pair_k2, block (c<=c2) at c*n_cams - c*(c-1)/2 + (c2-c) */
  const int32_t* pair_k;     /* [n_pairs] observation of camera c  on a shared track */
  const int32_t* pair_k2;    /* [n_pairs] observation of camera c2 on the same track */
  const int32_t* item_ptr;   /* [n_blocks+1] work items per block: each <= 256 consecutive pairs of ONE block */
  const int32_t* item_beg;   /* [n_items] ranges into pair_k / pair_k2 */
  const int32_t* item_end;   /* [n_items] */
  const int32_t* xcd_ptr;    /* [9]  item ids of block rows c = x (mod 8): xcd_items[xcd_ptr[x] .. xcd_ptr[x+1]) */
  const int32_t* xcd_items;  /* [n_items] (workgroup b of the Schur kernel serves group b % 8: XCD-local L2 reuse of G) */
  const int32_t* cch_ptr;    /* [n_cams+1] chunks per camera: each <= 256 consecutive entries of cam_obs of ONE camera */
  const int32_t* cch_beg;    /* [n_cchunks] ranges into cam_obs */
  const int32_t* cch_end;    /* [n_cchunks] */
} sfm_ba_structure;
int sfm_ba_get_structure(sfm_ba_problem p, sfm_ba_structure* out_host);

/* Byte offsets into the workspace of the regions the host reads or all-reduces. */
typedef struct {
  int64_t total_bytes;
  int64_t rec_off, rec_stride;      /* per observation: Jc~ [2][cam_dim] (robust-scaled), doubles; stride in bytes */
  int64_t recB_off;                 /* per observation: Jp~ [2][3], f~ [2] (8 doubles) */
  int64_t B_off, gc_off;            /* [n_cams][cam_dim][cam_dim], [n_cams][cam_dim]   (this rank's partial sums) */
  int64_t Cp_off, gp_off;           /* [n_pts][6] (xx,xy,xz,yy,yz,zz), [n_pts][3] */
  int64_t reduce_lin_off, reduce_lin_count;     /* doubles: [gc copy (n) | cost | ||gp||^2 | diag(B) (n)], n = n_cams*cam_dim  SUM */
  int64_t gmax_off;                              /* 2 doubles: max |gp|, max diag(C_j)                       MAX */
  int64_t reduce_S_off, reduce_S_count;         /* doubles: [S (n x n, n = n_cams*cam_dim) | r (n)]          SUM */
  int64_t reduce_Sp_off, reduce_Sp_count;       /* doubles: lower triangle of S by rows, then r: n(n+1)/2 + n - what
                                                   sfm_ba_pack_system fills and sfm_ba_unpack_system reads (the factorisation
                                                   only reads the lower triangle, so ranks exchange half the bytes) SUM */
  int64_t reduce_q_off, reduce_q_count;         /* doubles: [rhs2 (n) | ||p_pts||^2 | p_pts^T C_a^-1 p_pts]  SUM */
  int64_t reduce_step_off, reduce_step_count;   /* doubles: [||J~ s||^2 | f~^T J~ s | cost(x+s) | ||s_pts||^2 | ||x_pts+s_pts||^2 ] SUM */
  int64_t pc_off, pp_off;           /* camera / point part of p = -(H + alpha I)^-1 g */
  int64_t scalars_off;              /* 16 doubles, see SFM_SC_* */
  int64_t G_off;                    /* [n_obs][G stride]: [3][cam_dim] doubles per observation, padded to 32 doubles (256 B = two whole
                                       128-byte lines) for cam_dim 10, 18 for cam_dim 6 */
  int64_t cg_Ap_off, cg_M_off;      /* sfm_ba_solve_pcg: S p [n] and the block-Jacobi blocks [n_cams][cam_dim][cam_dim] (this rank's partial sums) SUM */
} sfm_ba_layout;

enum { SFM_SC_COST = 0, SFM_SC_GNORM2 = 1, SFM_SC_GINF = 2, SFM_SC_PNORM2 = 3, SFM_SC_PQ = 4,
       SFM_SC_JS2 = 5, SFM_SC_GTS = 6, SFM_SC_COST_NEW = 7, SFM_SC_SNORM2 = 8, SFM_SC_XNEW_NORM2 = 9,
       SFM_SC_CHOL_FAIL = 10 /* 0 ok, 1 not positive definite, 2 triangular solve stalled, 3 step not finite */,
       SFM_SC_HDIAG = 11 /* max diag(H) */, SFM_SC_COUNT = 16 };

/* Layout of the workspace of a problem.  The workspace (total_bytes, device memory) is the caller's: bind it
 * before the first stage; workspace == NULL makes the library allocate (and own) one.  rec / recB hold
 * float32 values when the problem was created with SFM_BA_MIXED (rec_stride is in bytes); G is always float64. */
int sfm_ba_get_layout(sfm_ba_problem p, sfm_ba_layout* out_host);
int sfm_ba_bind_workspace(sfm_handle h, sfm_ba_problem p, void* workspace, int64_t workspace_bytes);

/* cost(x) = 1/2 sum rho(f_i^2) (Huber, per scalar) -> partial into reduce_step[2] (this rank's observations). */
int sfm_ba_cost(sfm_handle h, sfm_ba_problem p, const double* x);
/* per-observation reprojection error ||proj - uv||_2.  shared_k != 0: ONE shared K = (fx0,fy0,cx0,cy0)
 * for every camera (compute_reconstruction_stats, :582-631); shared_k == 0: each camera's own
 * intrinsics when cam_dim == 10 (the reprojection rows of `objective`, :478-486). */
int sfm_ba_reproj_errors(sfm_handle h, sfm_ba_problem p, const double* x, int shared_k, double* err_out);

/* sum over the problem's observations of ||proj - uv||^2 at x, to the host (synchronises): the reprojection part of the
 * ||objective(x)||_2 that bundle_adjust logs before and after the solve (:522-524). */
int sfm_ba_residual_norm2(sfm_handle h, sfm_ba_problem p, const double* x, int shared_k, double* out_host);

/* The same without a problem object (nothing but the three packed arrays is needed): what
 * compute_reconstruction_stats (:582-631) computes per observation.  All pointers are device pointers; indices
 * must be in range (the caller's responsibility: there is no structure pass here). */
int sfm_reproj_errors(sfm_handle h, int32_t n_cams, int32_t cam_dim, int64_t n_obs, const int32_t* cam_idx,
                      const int32_t* pt_idx, const double* uv, const double* x, double fx, double fy, double cx,
                      double cy, int shared_k, double* err_out);

/* Linearise at x: records, B, gc, Cp, gp, cost -> reduce_lin region (+ gmax). */
int sfm_ba_linearize(sfm_handle h, sfm_ba_problem p, const double* x);
/* After the host has (all-)reduced reduce_lin and gmax: scalars COST, GNORM2, GINF, HDIAG. */
int sfm_ba_finish_linearize(sfm_handle h, sfm_ba_problem p);

/* Damped solve in three stages around two reductions. */
int sfm_ba_schur_build(sfm_handle h, sfm_ba_problem p, double alpha);          /* -> reduce_S (partial) */
/* Multi-rank only: reduce_S (lower triangle + r) -> reduce_Sp before the all-reduce, and back after it. */
int sfm_ba_pack_system(sfm_handle h, sfm_ba_problem p);
int sfm_ba_unpack_system(sfm_handle h, sfm_ba_problem p);
int sfm_ba_schur_solve(sfm_handle h, sfm_ba_problem p, double alpha, int want_q); /* chol, p_c, p_p; -> reduce_q (partial) */
int sfm_ba_finish_solve(sfm_handle h, sfm_ba_problem p, int want_q);              /* scalars PNORM2 (and PQ) */

/* The same damped solve WITHOUT forming or factoring S: conjugate gradients on the implicit Schur complement
 * S v = (B + alpha I) v - W (C + alpha I)^-1 W^T v (two passes over G per product), preconditioned with the exact
 * d x d diagonal blocks of S.  One call = schur_build + schur_solve + finish_solve: afterwards pc / pp hold the step
 * and the scalars PNORM2 (and PQ when want_q) are set; SFM_SC_CHOL_FAIL != 0 when S or a block is not positive
 * definite.  A system that has not reached rtol after max_iter iterations (measured: near convergence of the outer loop,
 * alpha ~ 1e-3, S nearly singular along the gauge directions, block-Jacobi PCG stalls at 1e-2 .. 1e-4) is never accepted:
 * the damped solve is then redone by the formed-S route (schur_build / schur_solve / finish_solve, reductions through the
 * same hook) and counted (sfm_ba_pcg_stats) - an inexact p would silently steer the alpha iteration.  Stops when ||r|| <= rtol ||r_0|| (checked every 8 iterations) or after max_iter iterations per system
 * (want_q solves two).  Multi-rank: `reduce` sums, per call, the right-hand side and the diagonal blocks once and
 * ONE vector of n doubles per iteration - against n(n+1)/2 + n doubles and a replicated factorisation on the dense
 * route; meant for many cameras (BASELINE config 5) and for scaling over GPUs.  iters_host: CG iterations spent. */
int sfm_ba_solve_pcg(sfm_handle h, sfm_ba_problem p, double alpha, int want_q, double rtol, int32_t max_iter,
                     sfm_reduce_fn reduce, void* reduce_user, int32_t* iters_host);

/* s = scale * p;  x_new = x + s;  partial sums for the predicted reduction and cost(x_new) -> reduce_step. */
int sfm_ba_step(sfm_handle h, sfm_ba_problem p, const double* x, double scale, double* x_new);
int sfm_ba_finish_step(sfm_handle h, sfm_ba_problem p, const double* x, double scale, const double* x_new);

/* Copy the SFM_SC_COUNT scalars to the host.  Waits for the FINISHING call of the stage enqueued last (sfm_ba_finish_linearize /
 * _finish_solve / _finish_step, sfm_ba_solve_pcg, the trust-region loop's own stages): its kernel publishes a ticket behind the
 * scalars in a pinned page and this call spins on it, so it returns as soon as the scalars are there - everything enqueued BEFORE
 * that kernel has completed by then (stream order), but the call is not a full stream synchronisation: work enqueued after a
 * finishing call is not waited for (SFM_POLL_SCALARS=0 restores the synchronisation).  This is also where a damped solve COMPLETES: the
 * persistent conjugate-gradient launch of the second camera system (sfm_ba_finish_solve with want_q) is not waited for
 * there - its verdict arrives with this synchronisation, and if it did not converge the q term is redone from the
 * factorisation here (no exchange between ranks is involved).  Read the scalars through this call, not from the workspace. */
int sfm_ba_read_scalars(sfm_handle h, sfm_ba_problem p, double* out_host);
/* Tell the library that this problem is ONE RANK'S SHARD of a multi-rank solve (points sharded, cameras replicated: SURVEY.md
 * section 8e; the reference has no counterpart - its solve is single-process, /root/reference/utils/sfm_reconstruction.py:506-514).
 * Every rank must then take the same route through the replicated camera solve, because the routes sum in different orders:
 * a persistent-CG launch that had to be abandoned on one rank is launched again instead of being replaced by the
 * launch-per-iteration route on that rank alone, and if it cannot run the solve fails with SFM_ERR_HIP (set SFM_CGS_PERSIST=0
 * on all ranks).  Implied by a non-null reduce hook in sfm_ba_trf_begin / sfm_ba_run_trf / sfm_ba_solve_pcg; a host that
 * drives the stages itself and reduces between them calls this once after sfm_ba_create_problem. */
int sfm_ba_set_sharded(sfm_handle h, sfm_ba_problem p, int sharded);
/* SFM_CAMERA_SOLVER_CG bookkeeping since the problem was created: CG iterations spent, solves that fell back. */
int sfm_ba_solver_stats(sfm_ba_problem p, int64_t* cg_iters_host, int64_t* cg_fallbacks_host);

/* sfm_ba_solve_pcg bookkeeping since the problem was created: damped solves that were handed to the formed-S route
 * because a system ran out of iterations above rtol, and the worst relative residual ||r|| / ||r_0|| PCG had reached there. */
int sfm_ba_pcg_stats(sfm_ba_problem p, int64_t* fallbacks_host, double* worst_relres_host);

/* ---- the trust-region loop (scipy _lsq/trf.py:401-560 trf_no_bounds + common.py:57-168,222-248,705-717),
 * control flow on the host, every data-parallel stage above on the device.  Same state machine as
 * sfm_amd/trf.py (the host-language mirror the multi-rank CPU tests drive). */
enum { SFM_SOLVER_DENSE = 0, SFM_SOLVER_PCG = 1 };
typedef struct {
  double ftol, xtol, gtol;        /* the reference passes ftol = xtol = 1e-4 (:512-513); SciPy's default gtol = 1e-8 */
  int32_t max_nfev;               /* 100 (:511) */
  int32_t max_outer;              /* < 0: no limit; otherwise stop after this many outer iterations (fixed schedules) */
  int32_t check_tolerances;       /* 0 turns the gtol / ftol / xtol tests off (throughput runs) */
  int32_t solver;                 /* SFM_SOLVER_DENSE: Schur complement + dense Cholesky; SFM_SOLVER_PCG: sfm_ba_solve_pcg */
  double  pcg_rtol;               /* SFM_SOLVER_PCG: relative residual (<= 0: 1e-13) */
  int32_t pcg_max_iter;           /* SFM_SOLVER_PCG: iterations per system before the formed-S fallback (<= 0: 400) */
  int32_t reserved;
} sfm_trf_options;
typedef struct {
  double cost, optimality;        /* 1/2 sum rho(f^2) and ||g||_inf at the returned x */
  int32_t nfev, njev, status;     /* SciPy's counters and termination status (0: max_nfev, 1 gtol, 2 ftol, 3 xtol, 4 both) */
  int32_t n_solves, n_outer;
  int32_t cg_iters;               /* SFM_SOLVER_PCG: conjugate-gradient iterations over all damped solves */
} sfm_trf_result;
typedef struct sfm_trf_state_s* sfm_trf_state;
/* x: [n_cams*cam_dim + 3*n_pts] device doubles, start point in, current iterate out (after every sfm_ba_trf_outer).
 * x_norm_pts_only_local: nothing to set - with a reduce hook the point part of ||x|| is summed over the ranks. */
int  sfm_ba_trf_begin(sfm_handle h, sfm_ba_problem p, double* x, const sfm_trf_options* opt,
                      sfm_reduce_fn reduce, void* reduce_user, sfm_trf_state* out);
/* One outer iteration: the trial steps of the current linearisation up to the accepted one, then the next
 * linearisation.  *more = 0 once the loop has ended (status set, max_nfev or max_outer reached). */
int  sfm_ba_trf_outer(sfm_trf_state st, int* more);
int  sfm_ba_trf_result(sfm_trf_state st, sfm_trf_result* out);
/* (alpha, Delta, ||step||, accepted) of every trial so far, 4 doubles each; returns the number of trials. */
int  sfm_ba_trf_trace(sfm_trf_state st, double* out_host, int32_t capacity_trials);
void sfm_ba_trf_end(sfm_trf_state st);
/* begin + outer until done + result + end. SFM_ERR_NUMERIC when a damped system is not positive definite or a
 * step is not finite (x then holds the last accepted iterate). */
int  sfm_ba_run_trf(sfm_handle h, sfm_ba_problem p, double* x, const sfm_trf_options* opt,
                    sfm_reduce_fn reduce, void* reduce_user, sfm_trf_result* out);

/* Dense SPD helpers used by the solve, exported for tests: in-place lower Cholesky of a [n][n]
 * row-major matrix and solves with the factor.  fail_flag: device int32, set when a pivot <= 0. */
int sfm_dense_cholesky(sfm_handle h, double* a, int32_t n, int32_t* fail_flag);
int sfm_dense_trsv(sfm_handle h, const double* l, int32_t n, double* b, int transpose);

/* ------------------------------------------------------------------ driver-side rows either side of the path
 * (SURVEY.md section 8f).  All three are batched over "segments" (one segment = one image pair), so the
 * per-pair Python loops of the reference become one launch.  seg pointers are DEVICE int64 arrays.
 */

/* Replaces the dense [T,M,2] broadcast + np.where of find_2d3d_matches
 * (/root/reference/utils/sfm_reconstruction.py:209-218): for every segment s, all pairs
 * (track row i in [t_ptr[s], t_ptr[s+1]), correspondence m in [m_ptr[s], m_ptr[s+1])) with
 * sqrt(dx*dx + dy*dy) < radius in float64, emitted segment-major, row-major (np.where order).
 * track_xy [T][2], corr_xy [M][2] float64 (float32 pixels widened exactly, as NumPy's broadcast does).
 * out_row / out_col: global row / correspondence indices, room for `capacity` pairs; pairs beyond the
 * capacity are counted but not written.  *n_pairs: device int64 total.  workspace from
 * sfm_assoc_workspace_bytes. */
int sfm_assoc_workspace_bytes(int64_t n_rows, int64_t* bytes_host);
int sfm_assoc_radius(sfm_handle h, const double* track_xy, const int64_t* t_ptr, const double* corr_xy,
                     const int64_t* m_ptr, int32_t n_seg, int64_t n_rows, double radius,
                     int32_t* out_row, int32_t* out_col, int64_t capacity, int64_t* n_pairs,
                     void* workspace, int64_t workspace_bytes);

/* Replaces the per-track cv2.triangulatePoints call + reprojection gate of triangulate_point
 * (sfm_reconstruction.py:287-307) inside add_new_matches' loop (:381-385): two-view DLT (null vector of
 * the 4x4 system x*P[2]-P[0], y*P[2]-P[1], by one-sided Jacobi SVD in float64), X = v[:3]/v[3], and
 * valid[i] = 0 when either view reprojects further than max_err px (comparison `err > max_err`, so NaN
 * passes exactly as in the reference).  proj [n_cams][12] row-major 3x4 K[R|t]; cam0/cam1 [n] index it;
 * x0/x1 [n][2] float64 pixels.  X [n][3]; err [n][2] (optional, may be NULL). */
int sfm_triangulate2(sfm_handle h, const double* proj, int32_t n_cams, const int32_t* cam0, const int32_t* cam1,
                     const double* x0, const double* x1, int64_t n, double max_err,
                     double* X, int32_t* valid, double* err);

/* Replaces the per-match arithmetic of geometric_verification
 * (/root/reference/utils/find_matches.py:160-174): epilines as cv2.computeCorrespondEpilines forms them
 * (float64 accumulate, a^2+b^2 = 1, stored float32), the float32 symmetric epipolar distance and
 * mask = err < threshold.  F [n_seg][9] float64; seg_ptr [n_seg+1]; pts1/pts2 [n][2] float32. */
int sfm_epipolar_errors(sfm_handle h, const double* F, const int64_t* seg_ptr, int32_t n_seg,
                        const float* pts1, const float* pts2, int64_t n, float threshold,
                        float* err, uint8_t* mask);

#ifdef __cplusplus
}
#endif
#endif
