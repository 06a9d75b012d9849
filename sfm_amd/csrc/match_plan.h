// Host-side planning of the matcher launches: how a (query set, train set) pair - or a batch of image pairs - is cut into
// workgroup-sized pieces.  Plain C++ (no HIP): included by match.hip, and compiled on its own with the address and
// undefined-behaviour sanitizers by tests/test_host_logic.py (which defines SFM_MATCH_PLAN_STANDALONE).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <vector>

#ifdef SFM_MATCH_PLAN_STANDALONE
enum { SFM_METRIC_L2_U8 = 0, SFM_METRIC_L2_F32 = 1, SFM_METRIC_HAMMING = 2 };      // as in include/sfm_amd.h
#endif

// Batched (segmented) matching: one launch covers every image pair of a preprocessing step
// (find_matches.py:329-350 calls match_features once per pair, serially).  A segment = one pair = a range of query
// rows against a range of train rows; the host cuts the segments into workgroup-sized pieces (plan_segments) and
// every workgroup of the distance kernels reads its piece from this record.  wg == nullptr = one segment covering
// the whole arrays, pieces computed from blockIdx.
struct MatchWG {
  int64_t q_first, q_end;     // query rows of this workgroup: [q_first, min(q_first + rows per workgroup, q_end))
  int64_t t_first, t_end;     // train rows of this workgroup's split
  int64_t t_seg;              // first train row of the segment: train indices are reported relative to it
  int64_t out_first;          // output row of q_first
  int32_t split, pad;
};
static inline int64_t plan_align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

static inline int pick_nsplit(int64_t nt, int64_t n_qblocks) {
  // Splits of the train set buy parallelism (workgroups = query blocks x splits; the splits of one query block run side
  // by side and share their candidate threshold, so they do not loosen the filter).  The distance kernel holds two
  // workgroups per CU: take the split count (<= 8, splits of at least 512 rows) whose LAST round of 512 workgroups is
  // fullest - at 50k x 50k, 196 query blocks: 5 splits = 1.9 rounds against 8 = 3.06, 10 % of the launch.
  // SFM_MATCH_NSPLIT overrides (tuning).
  const char* env = getenv("SFM_MATCH_NSPLIT");
  int best = 1;
  if (env && *env) {
    best = atoi(env);
    if (best > 8) best = 8;
    if (best < 1) best = 1;
  } else {
    double best_cost = 1e30;
    for (int ns = 1; ns <= 8; ++ns) {
      const int64_t wgs = n_qblocks * ns, rounds = (wgs + 511) / 512;
      const double cost = (double)rounds / ns;             // time ~ rounds x rows per split
      if (cost < best_cost * 0.98) { best_cost = cost; best = ns; }
    }
  }
  while (best > 1 && nt / best < 512) --best;
  return best;
}

// queries one workgroup of the distance kernel takes
static inline int64_t match_qpw(int metric, int dim, int64_t nq, bool batched) {
  // k_knn2_u8 with four query blocks per wave (512 queries per workgroup: half the train bytes through LDS per pair)
  // once there are enough queries to fill the chip that way; single segment, dim 128.  SFM_MATCH_QB = 2 / 4 overrides.
  const char* qb_env = getenv("SFM_MATCH_QB");
  // measured crossover (round 3, 20 calls each, us per call, 4 blocks / 2 blocks per wave): 12,288 queries 82 / 74,
  // 16,384: 98 / 89, 24,576: 166 / 146 (square sets; against 50,000 train rows 267 / 260), 32,768: 223 / 228 (301 / 317),
  // 40,960: 312 / 332, 50,000: 418 / 451 - two blocks (three waves per SIMD) win up to ~28k queries
  const bool qb4 = !batched && metric == SFM_METRIC_L2_U8 && dim == 128 && (qb_env ? qb_env[0] == '4' : nq >= 28672);
  return qb4 ? 512 : 256;
}

// rows of train data per split, and how many splits
static inline void match_tiling(int metric, int64_t nq, int64_t qpw, int64_t nt, int* nsplit, int64_t* rps) {
  int ns = pick_nsplit(nt, (nq + qpw - 1) / qpw);
  int64_t r = (nt + ns - 1) / ns;
  if (r < 1) r = 1;                                        // an empty train set: one (empty) piece, not a division by zero
  if (metric == SFM_METRIC_L2_U8) r = plan_align_up(r, 128);
  *nsplit = nt > 0 ? (int)((nt + r - 1) / r) : 1;
  *rps = r;
}

static inline void plan_segments(int metric, int dim, int32_t n_seg, const int64_t* q_beg, const int64_t* q_end, const int64_t* t_beg,
                          const int64_t* t_end, std::vector<MatchWG>* wgs, std::vector<int64_t>* out_ptr) {
  out_ptr->assign((size_t)n_seg + 1, 0);
  int64_t all_queries = 0;                           // parallelism comes from all segments together
  for (int s = 0; s < n_seg; ++s) all_queries += (q_end[s] - q_beg[s] + 255) / 256 * 256;
  for (int s = 0; s < n_seg; ++s) {
    const int64_t nq = q_end[s] - q_beg[s], nt = t_end[s] - t_beg[s];
    (*out_ptr)[s + 1] = (*out_ptr)[s] + nq;
    if (nq <= 0) continue;
    int nsplit; int64_t rps;
    match_tiling(metric, all_queries, 256, nt, &nsplit, &rps);
    for (int64_t qb = 0; qb < nq; qb += match_qpw(metric, dim, nq, true))
      for (int sp = 0; sp < nsplit; ++sp) {          // consecutive workgroups = consecutive splits: one XCD per split as in the single-pair launch
        MatchWG r;
        r.q_first = q_beg[s] + qb; r.q_end = q_end[s];
        r.t_first = t_beg[s] + sp * rps; r.t_end = (r.t_first + rps) < t_end[s] ? (r.t_first + rps) : t_end[s];
        r.t_seg = t_beg[s]; r.out_first = (*out_ptr)[s] + qb; r.split = sp; r.pad = 0;
        wgs->push_back(r);
      }
  }
}
