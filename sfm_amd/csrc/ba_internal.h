// Internal representation of a bundle-adjustment problem (include/sfm_amd.h: sfm_ba_problem is a pointer to this).
#pragma once
#include "dense.h"

struct Lay {   // workspace offsets in doubles (regions holding float32 in mixed precision are sized in doubles too)
  int64_t recA, recB, campre, campre2, B, gc, Cp, gp, Linv, e, v, tmp3, G, eobs, red_lin, gmax, red_S, red_q,
      red_step, pc, pp, y, tvec, scalars, part_obs, part_pt, part_x, cost_reg, regrec, dense, sch_part, cch_part, cbl_part,
      cg_r, cg_z, cg_p, cg_Ap, cg_M, cg_Minv, cg_scal, cg_mail, cg_warm, total;
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
  int32_t* cam_pt;           // [n_obs] point of cam_obs[i]: the camera-wise passes fetch it beside the observation id, not behind it

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
  int cg2_pending;           // the persistent CG of the second system is in flight: sfm_ba_read_scalars looks at its verdict
  double* host_sc;           // pinned host mirror of the SFM_SC_* scalars (written by the kernels that write them; sfm_ba_read_scalars),
                             // followed by the 8 status words of THIS problem's pending second-system CG (a slot per problem: two
                             // problems on one handle may both have a verdict in flight between finish_solve and read_scalars)
  // sfm_ba_read_scalars waits for a TICKET instead of for the stream: the kernel that finishes a stage (k_finish_linearize /
  // _solve / _solve_pcg / _step, k_xnorm_finish, the epilogue of the persistent CG's second system) writes the scalars into the
  // pinned mirror and, behind a system-scope fence, the ticket it was launched with into word SFM_HSC_SEQ of the same page; the
  // host spins on that word.  look_seq: the last ticket handed out; look_pending: a ticket is out and has not been waited for.
  double look_seq;
  int look_pending;
  int sharded;               // this problem is one rank's shard of a multi-rank solve (sfm_ba_set_sharded, or a reduce hook given to
                             // the loop / sfm_ba_solve_pcg): a route whose choice could differ between ranks is never switched locally
  int cg_scal_clean;         // sfm_ba_schur_build has just cleared the CG status words (sfm_ba_schur_solve then skips its memset)
  double einv_alpha;         // >= 0: k_schur_assemble left the diagonal blocks' factors E_c, E_c^-1 for this alpha (unsharded problems)
  double st_alpha;           // > 0: sfm_ba_schur_build left the SCALED system S~ for this alpha in the factor's buffer (tile-streaming route)
  int s_valid;               // S itself (red_S) holds the current system; 0: only S~ was formed - schur_materialise_S forms S from the items
  // 1: some camera appears more than once on a track.  The diagonal Schur blocks then hold cross pairs besides the
  // self-pairs, so k_schur_items must not take its fused diagonal path (one gather for both operands, right-hand side in
  // accumulator column D): every item runs the general path and the right-hand side comes from the camera-wise pass.
  int has_dup;
  // sfm_ba_solve_pcg: damped solves handed to the formed-S route because a system ran out of iterations above rtol, and
  // the worst relative residual PCG had reached in such a system
  int64_t pcg_fallbacks;
  double pcg_worst_relres;
  double cg_alpha;
  // Camera CG, SFM_CAMERA_SOLVER_AUTO: what this problem has taught about where the iteration budget runs out (alpha relative
  // to max diag H): the largest alpha at which CG ran out of iterations (forgotten by 20 % per linearisation), and the last two
  // converged step systems at different alpha, whose iteration counts give the local exponent of iterations ~ alpha^-s
  // (measured: s ~ 0.4 on the spatially coherent scene, ~ 0.2 on the random one, falling towards alpha -> 0).  A damped solve at
  // or below 4 x the failure bound, or for which that power law - with 0.85 s - predicts more than 1.25 x the budget, goes to
  // the factorisation at once instead of burning the whole budget first (sfm_ba_schur_solve).
  double cgp_fail_rel, cgp_ok_rel[2];
  int cgp_ok_its[2];
  int cg_its_sys1;           // iterations of the step system of the current damped solve (launch-per-iteration routes)
  // warm start of the camera CG (ba.hip, sfm_ba_schur_solve): p_c / q_c of the previous damped solve of THIS linearisation
  int warm_pc_ok, warm_qc_ok;
  double warm_alpha;
};

// layout of the pinned page host_sc: the SFM_SC_* scalars, the 8 status words of the pending second-system CG, the ticket
constexpr int SFM_HSC_CG2 = SFM_SC_COUNT, SFM_HSC_SEQ = SFM_SC_COUNT + 8, SFM_HSC_WORDS = SFM_SC_COUNT + 16;

// the ticket the finishing kernel of a stage is launched with (sfm_ba_read_scalars)
static inline double next_ticket(sfm_ba_prob* p) { p->look_seq += 1.0; p->look_pending = 1; return p->look_seq; }

Lay ba_layout(int64_t C, int64_t P, int64_t N, int64_t D, int64_t n_items, int64_t n_cchunks, int precision);

// growable per-handle device scratch (stream-ordered use only)
void* sfm_scratch(sfm_ctx* h, size_t bytes);

// ||x||^2 in two stages around the multi-rank reduction: the point part of this rank -> reduce_step[4], then
// camera part + reduced point part -> scalar SFM_SC_XNEW_NORM2 (ba.hip; used by the trust-region loop's start)
int ba_xnorm_partial(sfm_ctx* h, sfm_ba_problem p, const double* x);
int ba_xnorm_finish(sfm_ctx* h, sfm_ba_problem p, const double* x);
