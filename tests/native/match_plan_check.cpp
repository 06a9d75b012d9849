// Host-only check of sfm_amd/csrc/match_plan.h, built with -fsanitize=address,undefined by tests/test_host_logic.py.
// Random segment tables -> plan_segments: every (query row, train row) of every segment must be covered by exactly one
// piece, pieces stay inside their segment, splits are whole 128-row chunks except the last, output rows are the
// segment's query rows in order; pick_nsplit respects its limits.  Prints "ok <pieces>" or a diagnostic and exits 1.
#define SFM_MATCH_PLAN_STANDALONE 1
#include "match_plan.h"
#include <cstdio>
#include <random>

static int fail(const char* what, long a, long b) { std::printf("FAIL %s %ld %ld\n", what, a, b); return 1; }

int main(int argc, char** argv) {
  std::mt19937_64 rng(argc > 1 ? std::atoll(argv[1]) : 1);
  long total_pieces = 0;
  for (int round = 0; round < 200; ++round) {
    const int n_seg = 1 + (int)(rng() % 40);
    const int metric = (int)(rng() % 3);
    const int dim = metric == SFM_METRIC_HAMMING ? 32 : (int[]){32, 64, 128}[rng() % 3];
    std::vector<int64_t> img_ptr(1, 0);
    const int n_img = 2 + (int)(rng() % 12);
    for (int i = 0; i < n_img; ++i) {
      const int kind = (int)(rng() % 5);
      const int64_t sz = kind == 0 ? 0 : kind == 1 ? 2 + (int64_t)(rng() % 40) : kind == 2 ? 200 + (int64_t)(rng() % 3000)
                                                   : kind == 3 ? 2048 + (int64_t)(rng() % 9000) : 511 + (int64_t)(rng() % 3);
      img_ptr.push_back(img_ptr.back() + sz);
    }
    std::vector<int64_t> qb(n_seg), qe(n_seg), tb(n_seg), te(n_seg);
    for (int s = 0; s < n_seg; ++s) {
      const int i = (int)(rng() % n_img), j = (int)(rng() % n_img);
      qb[s] = img_ptr[i]; qe[s] = img_ptr[i + 1]; tb[s] = img_ptr[j]; te[s] = img_ptr[j + 1];
    }
    std::vector<MatchWG> wgs; std::vector<int64_t> out_ptr;
    plan_segments(metric, dim, n_seg, qb.data(), qe.data(), tb.data(), te.data(), &wgs, &out_ptr);
    total_pieces += (long)wgs.size();
    size_t k = 0;
    for (int s = 0; s < n_seg; ++s) {
      const int64_t nq = qe[s] - qb[s], nt = te[s] - tb[s];
      if (out_ptr[s + 1] - out_ptr[s] != nq) return fail("out_ptr", s, (long)nq);
      if (nq <= 0) continue;
      const int64_t qpw = match_qpw(metric, dim, nq, true);
      for (int64_t q0 = 0; q0 < nq; q0 += qpw) {
        int64_t covered = 0; int split = 0;
        while (k < wgs.size() && wgs[k].q_first == qb[s] + q0 && wgs[k].t_seg == tb[s] && wgs[k].out_first == out_ptr[s] + q0) {
          const MatchWG& r = wgs[k];
          if (r.q_end != qe[s] || r.split != split) return fail("piece header", s, split);
          if (r.t_first != tb[s] + covered) return fail("split start", (long)r.t_first, (long)(tb[s] + covered));
          if (r.t_end > te[s] || r.t_end < r.t_first) return fail("split end", (long)r.t_end, (long)te[s]);
          if (r.t_end != te[s] && metric == SFM_METRIC_L2_U8 && (r.t_end - r.t_first) % 128 != 0) return fail("chunk multiple", s, split);
          covered += r.t_end - r.t_first; ++split; ++k;
          if (split > 8) return fail("more than 8 splits", s, split);
          if (covered == nt) break;
        }
        if (covered != nt && nt > 0) return fail("train rows covered", (long)covered, (long)nt);
        if (nt <= 0 && split != 1) return fail("empty train set pieces", s, split);
      }
    }
    if (k != wgs.size()) return fail("stray pieces", (long)k, (long)wgs.size());
  }
  for (int64_t nt : {2L, 511L, 512L, 1024L, 4095L, 50000L, 1000000L})
    for (int64_t nqb : {1L, 7L, 98L, 196L, 4000L}) {
      const int ns = pick_nsplit(nt, nqb);
      if (ns < 1 || ns > 8 || (ns > 1 && nt / ns < 512)) return fail("pick_nsplit", (long)nt, ns);
    }
  std::printf("ok %ld\n", total_pieces);
  return 0;
}
