// Internal representation of a bundle-adjustment problem (include/sfm_amd.h: sfm_ba_problem is a pointer to this).
#pragma once
#include "dense.h"

struct Lay {   // workspace offsets in doubles (regions holding float32 in mixed precision are sized in doubles too)
  int64_t recA, recB, campre, campre2, B, gc, Cp, gp, Linv, e, v, tmp3, G, eobs, red_lin, gmax, red_S, red_q,
      red_step, pc, pp, y, tvec, scalars, part_obs, part_pt, part_x, cost_reg, regrec, dense, sch_part, cch_part, cbl_part,
      cg_r, cg_z, cg_p, cg_Ap, cg_M, cg_Minv, cg_scal, total;
  int64_t nblk_obs, nblk_pt;
};

struct sfm_ba_prob {
  sfm_ctx* h;
  int32_t n_cams, n_pts, cam_dim, apply_reg, precision;
  int64_t n_obs;
  double fx0, fy0, cx0, cy0, width, height, reg_weight;
  // owned device arrays (one allocation each; freed by sfm_ba_destroy_problem)
  int32_t *cam_idx, *pt_idx;
  double* uv;
  int32_t *pt_ptr, *cam_ptr, *cam_obs, *blk_ptr, *pair_k, *pair_k2;
  int64_t n_pairs;
  int32_t *item_ptr, *item_beg, *item_end;
  int64_t n_items;
  int32_t *xcd_ptr, *xcd_items;
  int64_t xcd_max_items;
  int32_t *cch_ptr, *cch_beg, *cch_end;
  int64_t n_cchunks;
  // workspace: the caller's (sfm_ba_bind_workspace) or owned
  void* workspace;
  int64_t workspace_bytes;
  int owns_workspace;
  Lay L;
  // camera-system solver (sfm_ba_desc.camera_solver) and the state of the current damped solve between
  // sfm_ba_schur_solve and sfm_ba_finish_solve: cg_state 1 = p_c came from CG on the scaled system (still in the
  // factor's buffer), 0 = from the factorisation
  int camera_solver, cg_state, cg_iters, cg_fallbacks;
  double cg_alpha;
};

Lay ba_layout(int64_t C, int64_t P, int64_t N, int64_t D, int64_t n_items, int64_t n_cchunks, int precision);

// growable per-handle device scratch (stream-ordered use only)
void* sfm_scratch(sfm_ctx* h, size_t bytes);

// ||x||^2 in two stages around the multi-rank reduction: the point part of this rank -> reduce_step[4], then
// camera part + reduced point part -> scalar SFM_SC_XNEW_NORM2 (ba.hip; used by the trust-region loop's start)
int ba_xnorm_partial(sfm_ctx* h, sfm_ba_problem p, const double* x);
int ba_xnorm_finish(sfm_ctx* h, sfm_ba_problem p, const double* x);
