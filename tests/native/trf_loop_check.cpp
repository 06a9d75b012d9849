// Host-only run of sfm_amd/csrc/trf_loop.h (the trust-region state machine of sfm_ba_trf_*) on a small dense problem,
// built with -fsanitize=address,undefined by tests/test_host_logic.py and compared there with sfm_amd/trf.py driving the
// same problem in NumPy.  Problem: f_i(x) = x0 exp(-x1 t_i) + x2 exp(-x3 t_i) + x4 - y_i, 16 samples, cost 1/2 |f|^2.
// argv: ftol xtol max_nfev x0..x4.  Prints: nfev njev status cost x0..x4 n_trials.
#include "trf_loop.h"
#include <cstdio>
#include <cstdlib>

namespace {
constexpr int M = 16, N = 5;

struct Dense {
  double x[N], xn[N], f[M], J[M][N], g[N], H[N][N], p[N];
  double y[M], t[M];
  void residuals(const double* v, double* out) const {
    for (int i = 0; i < M; ++i) out[i] = v[0] * std::exp(-v[1] * t[i]) + v[2] * std::exp(-v[3] * t[i]) + v[4] - y[i];
  }
  int linearize(double* cost, double* g_norm, double* g_inf, double* hdiag) {
    residuals(x, f);
    for (int i = 0; i < M; ++i) {
      const double e1 = std::exp(-x[1] * t[i]), e3 = std::exp(-x[3] * t[i]);
      J[i][0] = e1; J[i][1] = -x[0] * t[i] * e1; J[i][2] = e3; J[i][3] = -x[2] * t[i] * e3; J[i][4] = 1.0;
    }
    double c = 0; for (int i = 0; i < M; ++i) c += f[i] * f[i];
    *cost = 0.5 * c;
    double g2 = 0, gi = 0, hd = 0;
    for (int a = 0; a < N; ++a) {
      g[a] = 0; for (int i = 0; i < M; ++i) g[a] += J[i][a] * f[i];
      g2 += g[a] * g[a]; gi = std::fmax(gi, std::fabs(g[a]));
      for (int b = 0; b < N; ++b) { H[a][b] = 0; for (int i = 0; i < M; ++i) H[a][b] += J[i][a] * J[i][b]; }
      hd = std::fmax(hd, H[a][a]);
    }
    *g_norm = std::sqrt(g2); *g_inf = gi; *hdiag = hd;
    return 0;
  }
  static bool chol_solve(const double A[N][N], const double* b, double* out) {      // A = L L^T, out = A^-1 b
    double L[N][N] = {};
    for (int i = 0; i < N; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = A[i][j];
        for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
        if (i == j) { if (!(s > 0)) return false; L[i][i] = std::sqrt(s); } else L[i][j] = s / L[j][j];
      }
    double z[N];
    for (int i = 0; i < N; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i][k] * z[k]; z[i] = s / L[i][i]; }
    for (int i = N - 1; i >= 0; --i) { double s = z[i]; for (int k = i + 1; k < N; ++k) s -= L[k][i] * out[k]; out[i] = s / L[i][i]; }
    return true;
  }
  int solve(double alpha, int want_q, double* p_norm, double* pq) {
    double A[N][N], mg[N], q[N];
    for (int a = 0; a < N; ++a) { mg[a] = -g[a]; for (int b = 0; b < N; ++b) A[a][b] = H[a][b] + (a == b ? alpha : 0.0); }
    if (!chol_solve(A, mg, p)) return 7;
    double p2 = 0; for (int a = 0; a < N; ++a) p2 += p[a] * p[a];
    *p_norm = std::sqrt(p2); *pq = 0;
    if (want_q) { if (!chol_solve(A, p, q)) return 7; for (int a = 0; a < N; ++a) *pq += p[a] * q[a]; }
    return 0;
  }
  int step(double scale, double* js2, double* gts, double* cost_new, double* step_norm, double* xnew_norm) {
    double s[N], fn[M];
    double s2 = 0, x2 = 0, gs = 0;
    for (int a = 0; a < N; ++a) { s[a] = scale * p[a]; xn[a] = x[a] + s[a]; s2 += s[a] * s[a]; x2 += xn[a] * xn[a]; gs += g[a] * s[a]; }
    double j2 = 0;
    for (int i = 0; i < M; ++i) { double v = 0; for (int a = 0; a < N; ++a) v += J[i][a] * s[a]; j2 += v * v; }
    residuals(xn, fn);
    double c = 0; for (int i = 0; i < M; ++i) c += fn[i] * fn[i];
    *js2 = j2; *gts = gs; *cost_new = 0.5 * c; *step_norm = std::sqrt(s2); *xnew_norm = std::sqrt(x2);
    return 0;
  }
  int x_norm(double* out) { double s = 0; for (int a = 0; a < N; ++a) s += x[a] * x[a]; *out = std::sqrt(s); return 0; }
  int accept() { for (int a = 0; a < N; ++a) x[a] = xn[a]; return 0; }
};
}  // namespace

int main(int argc, char** argv) {
  if (argc != 4 + N) { std::printf("usage: ftol xtol max_nfev x0..x4\n"); return 2; }
  Dense be;
  const double truth[N] = {2.0, 0.7, 1.5, 0.1, 0.3};
  for (int i = 0; i < M; ++i) {
    be.t[i] = 0.25 * i;
    be.y[i] = truth[0] * std::exp(-truth[1] * be.t[i]) + truth[2] * std::exp(-truth[3] * be.t[i]) + truth[4] + 0.01 * std::sin(3.7 * i);
  }
  for (int a = 0; a < N; ++a) be.x[a] = std::atof(argv[4 + a]);
  trf_core::Options o = {std::atof(argv[1]), std::atof(argv[2]), 1e-8, std::atoi(argv[3]), -1, 1};
  trf_core::State s;
  int rc = trf_core::begin(be, s), more = 1;
  while (!rc && more) rc = trf_core::outer(be, s, o, &more);
  if (rc) { std::printf("rc %d\n", rc); return 1; }
  std::printf("%d %d %d %.17g", s.nfev, s.njev, s.status < 0 ? 0 : s.status, s.cost);
  for (int a = 0; a < N; ++a) std::printf(" %.17g", be.x[a]);
  std::printf(" %zu\n", s.trace.size() / 4);
  return 0;
}
