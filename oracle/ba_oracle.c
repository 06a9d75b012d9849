/* CPU oracle in plain C (TEST INFRASTRUCTURE / cpu_baseline - not product code).
 *
 * Scalable restatement of the same algorithm as oracle/ba_oracle.py (which is pinned against the
 * reference's own runs, tests/golden/): residual + analytic Jacobian of the reference's `objective`
 * (/root/reference/utils/sfm_reconstruction.py:453-501), Huber row scaling (scipy _lsq/common.py:720-731),
 * block normal equations, damped Schur-complement solve, and SciPy's trust-region loop
 * (scipy _lsq/trf.py:401-560, common.py:57-168,222-248,705-717).  Also the matcher restatement
 * (/root/reference/utils/find_matches.py:141-155) for uint8 descriptors.
 * OpenMP over cameras / points / queries; fp64.  Built by oracle/build_c.py into oracle/_build/.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPS_D 2.220446049250313e-16
#define SQRT_EPS_D 1.4901161193847656e-08
#define ALPHA_FLOOR_REL 1e-13

typedef struct {
  int C, P, d, apply_reg;
  int64_t N;
  const int32_t *cam_idx, *pt_idx;
  const double* uv;
  double fx0, fy0, cx0, cy0, width, height, w;
  /* derived */
  int32_t *pt_ptr, *cam_ptr, *cam_obs;
  /* linearisation */
  double *R, *dR;              /* per camera 9, 27 */
  double *Jc, *Jp, *ft;        /* [N][2][d], [N][2][3], [N][2] robust scaled */
  double *B, *gc, *Cp, *gp;    /* [C][d][d], [C][d], [P][6], [P][3] */
  double *regJ, *regf;         /* [C][16], [C][4] */
  double cost, gnorm, ginf, hdiag;
  /* solve */
  double *Minv, *e, *G, *S, *r, *pc, *pp, *v, *y;
  int n;
} bao;

/* Sensitivity probe (tools/diag/oracle_rounding_sensitivity.py): -DBAO_REVERSE_SUMS walks every camera's observation list
   backwards in the sums B_c, g_c and in the rows of S - the same arithmetic in another, equally legitimate summation order. */
#ifdef BAO_REVERSE_SUMS
#define BAO_ORDER(i, lo, hi) ((lo) + (hi) - 1 - (i))
#else
#define BAO_ORDER(i, lo, hi) (i)
#endif

static void rod(const double* r, double* R, double* dR) {
  double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2], a, b, a1, b1;
  if (th2 < 1e-4) {
    double z = th2;
    a = 1.0 - z / 6.0 + z * z / 120.0; b = 0.5 - z / 24.0 + z * z / 720.0;
    a1 = -1.0 / 3.0 + z / 30.0 - z * z / 840.0; b1 = -1.0 / 12.0 + z / 180.0 - z * z / 6720.0;
  } else {
    double t = sqrt(th2), s = sin(t), c = cos(t);
    a = s / t; b = (1.0 - c) / th2; a1 = (t * c - s) / (th2 * t); b1 = (t * s - 2.0 * (1.0 - c)) / (th2 * th2);
  }
  double S[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0}, S2[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) S2[i * 3 + j] = S[i * 3] * S[j] + S[i * 3 + 1] * S[3 + j] + S[i * 3 + 2] * S[6 + j];
  for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * S[i] + b * S2[i];
  if (!dR) return;
  for (int i = 0; i < 3; ++i) {
    double e[3] = {0, 0, 0}; e[i] = 1.0;
    double E[9] = {0, -e[2], e[1], e[2], 0, -e[0], -e[1], e[0], 0}, ES[9], SE[9];
    for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) {
      ES[p * 3 + q] = E[p * 3] * S[q] + E[p * 3 + 1] * S[3 + q] + E[p * 3 + 2] * S[6 + q];
      SE[p * 3 + q] = S[p * 3] * E[q] + S[p * 3 + 1] * E[3 + q] + S[p * 3 + 2] * E[6 + q];
    }
    for (int q = 0; q < 9; ++q) dR[i * 9 + q] = a * E[q] + b * (ES[q] + SE[q]) + (a1 * r[i]) * S[q] + (b1 * r[i]) * S2[q];
  }
}

static double huber_row(double f, double* scale, double* ft) {
  double z = f * f;
  if (z <= 1.0) { *scale = 1.0; *ft = f; return z; }
  double sz = sqrt(z);
  *scale = SQRT_EPS_D; *ft = f * (1.0 / sz) / SQRT_EPS_D;
  return 2.0 * sz - 1.0;
}
static double huber_rho0(double f) { double z = f * f; return z <= 1.0 ? z : 2.0 * sqrt(z) - 1.0; }

static void intr(const bao* b, const double* cam, double* k) {
  if (b->d == 10) { k[0] = cam[6]; k[1] = cam[7]; k[2] = cam[8]; k[3] = cam[9]; }
  else { k[0] = b->fx0; k[1] = b->fy0; k[2] = b->cx0; k[3] = b->cy0; }
}

bao* bao_create(int C, int P, int d, int64_t N, const int32_t* cam_idx, const int32_t* pt_idx, const double* uv,
                const double* K0, double width, double height, double w, int apply_reg) {
  bao* b = (bao*)calloc(1, sizeof(bao));
  b->C = C; b->P = P; b->d = d; b->N = N; b->cam_idx = cam_idx; b->pt_idx = pt_idx; b->uv = uv;
  b->fx0 = K0[0]; b->fy0 = K0[1]; b->cx0 = K0[2]; b->cy0 = K0[3]; b->width = width; b->height = height; b->w = w;
  b->apply_reg = apply_reg && d == 10;
  b->n = C * d;
  b->pt_ptr = (int32_t*)calloc(P + 1, 4); b->cam_ptr = (int32_t*)calloc(C + 1, 4); b->cam_obs = (int32_t*)malloc(N * 4);
  for (int64_t k = 0; k < N; ++k) { b->pt_ptr[pt_idx[k] + 1]++; b->cam_ptr[cam_idx[k] + 1]++; }
  for (int j = 0; j < P; ++j) b->pt_ptr[j + 1] += b->pt_ptr[j];
  for (int c = 0; c < C; ++c) b->cam_ptr[c + 1] += b->cam_ptr[c];
  int32_t* fill = (int32_t*)calloc(C, 4);
  for (int64_t k = 0; k < N; ++k) { int c = cam_idx[k]; b->cam_obs[b->cam_ptr[c] + fill[c]++] = (int32_t)k; }
  free(fill);
  size_t n = b->n;
  b->R = malloc(C * 9 * 8); b->dR = malloc(C * 27 * 8);
  b->Jc = malloc(N * 2 * d * 8); b->Jp = malloc(N * 6 * 8); b->ft = malloc(N * 2 * 8);
  b->B = malloc((size_t)C * d * d * 8); b->gc = malloc(n * 8); b->Cp = malloc((size_t)P * 6 * 8); b->gp = malloc((size_t)P * 3 * 8);
  b->regJ = calloc(C * 16, 8); b->regf = calloc(C * 4, 8);
  b->Minv = malloc((size_t)P * 6 * 8); b->e = malloc((size_t)P * 3 * 8); b->G = malloc(N * 3 * d * 8);
  b->S = malloc(n * n * 8); b->r = malloc(n * 8); b->pc = malloc(n * 8); b->pp = malloc((size_t)P * 3 * 8);
  b->v = malloc((size_t)P * 3 * 8); b->y = malloc(n * 8);
  return b;
}

void bao_destroy(bao* b) {
  if (!b) return;
  free(b->pt_ptr); free(b->cam_ptr); free(b->cam_obs); free(b->R); free(b->dR); free(b->Jc); free(b->Jp); free(b->ft);
  free(b->B); free(b->gc); free(b->Cp); free(b->gp); free(b->regJ); free(b->regf); free(b->Minv); free(b->e); free(b->G);
  free(b->S); free(b->r); free(b->pc); free(b->pp); free(b->v); free(b->y); free(b);
}

/* cost(x) = 1/2 sum rho (reprojection rows + regulariser rows) */
double bao_cost(bao* b, const double* x) {
  const int C = b->C, d = b->d;
  const double* pts = x + (size_t)C * d;
  double* R = malloc(C * 9 * 8);
  for (int c = 0; c < C; ++c) rod(x + (size_t)c * d, R + c * 9, NULL);
  double cost = 0.0;
#pragma omp parallel for reduction(+ : cost) schedule(static)
  for (int64_t k = 0; k < b->N; ++k) {
    const int c = b->cam_idx[k];
    const double *cam = x + (size_t)c * d, *Rc = R + c * 9, *X = pts + 3 * (size_t)b->pt_idx[k];
    double kk[4]; intr(b, cam, kk);
    double Y0 = Rc[0] * X[0] + Rc[1] * X[1] + Rc[2] * X[2] + cam[3], Y1 = Rc[3] * X[0] + Rc[4] * X[1] + Rc[5] * X[2] + cam[4],
           Y2 = Rc[6] * X[0] + Rc[7] * X[1] + Rc[8] * X[2] + cam[5];
    double iz = 1.0 / Y2;
    cost += 0.5 * (huber_rho0(kk[0] * (Y0 * iz) + kk[2] - b->uv[2 * k]) + huber_rho0(kk[1] * (Y1 * iz) + kk[3] - b->uv[2 * k + 1]));
  }
  if (b->apply_reg)
    for (int c = 0; c < C; ++c) {
      const double* p = x + (size_t)c * 10;
      cost += 0.5 * (huber_rho0((p[6] - b->fx0) / b->fx0 * b->w) + huber_rho0((p[7] - p[6]) / p[6] * b->w) +
                     huber_rho0((p[8] - b->cx0) / b->width * b->w) + huber_rho0((p[9] - b->cy0) / b->height * b->w));
    }
  free(R);
  return cost;
}

void bao_linearize(bao* b, const double* x) {
  const int C = b->C, P = b->P, d = b->d;
  const double* pts = x + (size_t)C * d;
  for (int c = 0; c < C; ++c) rod(x + (size_t)c * d, b->R + c * 9, b->dR + c * 27);
  double cost = 0.0;
#pragma omp parallel for reduction(+ : cost) schedule(static)
  for (int64_t k = 0; k < b->N; ++k) {
    const int c = b->cam_idx[k];
    const double *cam = x + (size_t)c * d, *R = b->R + c * 9, *dR = b->dR + c * 27, *X = pts + 3 * (size_t)b->pt_idx[k];
    double kk[4]; intr(b, cam, kk);
    const double Y0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + cam[3], Y1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + cam[4],
                 Y2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + cam[5];
    const double iz = 1.0 / Y2, xn = Y0 * iz, yn = Y1 * iz;
    const double f0 = kk[0] * xn + kk[2] - b->uv[2 * k], f1 = kk[1] * yn + kk[3] - b->uv[2 * k + 1];
    double s0, s1, ft0, ft1;
    cost += 0.5 * (huber_row(f0, &s0, &ft0) + huber_row(f1, &s1, &ft1));
    const double p00 = s0 * kk[0] * iz, p02 = -s0 * kk[0] * xn * iz, p11 = s1 * kk[1] * iz, p12 = -s1 * kk[1] * yn * iz;
    double* jc = b->Jc + (size_t)k * 2 * d; double* jp = b->Jp + (size_t)k * 6;
    for (int i = 0; i < 3; ++i) {
      const double* D = dR + i * 9;
      const double d0 = D[0] * X[0] + D[1] * X[1] + D[2] * X[2], d1 = D[3] * X[0] + D[4] * X[1] + D[5] * X[2], d2 = D[6] * X[0] + D[7] * X[1] + D[8] * X[2];
      jc[i] = p00 * d0 + p02 * d2; jc[d + i] = p11 * d1 + p12 * d2;
    }
    jc[3] = p00; jc[4] = 0; jc[5] = p02; jc[d + 3] = 0; jc[d + 4] = p11; jc[d + 5] = p12;
    if (d == 10) { jc[6] = s0 * xn; jc[7] = 0; jc[8] = s0; jc[9] = 0; jc[16] = 0; jc[17] = s1 * yn; jc[18] = 0; jc[19] = s1; }
    for (int q = 0; q < 3; ++q) { jp[q] = p00 * R[q] + p02 * R[6 + q]; jp[3 + q] = p11 * R[3 + q] + p12 * R[6 + q]; }
    b->ft[2 * k] = ft0; b->ft[2 * k + 1] = ft1;
  }
#pragma omp parallel for schedule(static)
  for (int j = 0; j < P; ++j) {
    double c[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int k = b->pt_ptr[j]; k < b->pt_ptr[j + 1]; ++k) {
      const double* a = b->Jp + (size_t)k * 6; const double f0 = b->ft[2 * k], f1 = b->ft[2 * k + 1];
      c[0] += a[0] * a[0] + a[3] * a[3]; c[1] += a[0] * a[1] + a[3] * a[4]; c[2] += a[0] * a[2] + a[3] * a[5];
      c[3] += a[1] * a[1] + a[4] * a[4]; c[4] += a[1] * a[2] + a[4] * a[5]; c[5] += a[2] * a[2] + a[5] * a[5];
      g[0] += a[0] * f0 + a[3] * f1; g[1] += a[1] * f0 + a[4] * f1; g[2] += a[2] * f0 + a[5] * f1;
    }
    memcpy(b->Cp + (size_t)j * 6, c, 48); memcpy(b->gp + (size_t)j * 3, g, 24);
  }
  double creg = 0.0;
#pragma omp parallel for reduction(+ : creg) schedule(dynamic, 1)
  for (int c = 0; c < C; ++c) {
    double* Bc = b->B + (size_t)c * d * d; double* g = b->gc + (size_t)c * d;
    memset(Bc, 0, (size_t)d * d * 8); memset(g, 0, d * 8);
    for (int i0 = b->cam_ptr[c]; i0 < b->cam_ptr[c + 1]; ++i0) {
      const int i = BAO_ORDER(i0, b->cam_ptr[c], b->cam_ptr[c + 1]);
      const int k = b->cam_obs[i];
      const double* j = b->Jc + (size_t)k * 2 * d; const double f0 = b->ft[2 * k], f1 = b->ft[2 * k + 1];
      for (int a = 0; a < d; ++a) {
        g[a] += j[a] * f0 + j[d + a] * f1;
        for (int q = 0; q < d; ++q) Bc[a * d + q] += j[a] * j[q] + j[d + a] * j[d + q];
      }
    }
    if (b->apply_reg) {
      const double* p = x + (size_t)c * 10;
      double f[4] = {(p[6] - b->fx0) / b->fx0 * b->w, (p[7] - p[6]) / p[6] * b->w, (p[8] - b->cx0) / b->width * b->w, (p[9] - b->cy0) / b->height * b->w};
      double J[16]; memset(J, 0, sizeof J);
      J[0] = b->w / b->fx0; J[4] = -b->w * p[7] / (p[6] * p[6]); J[5] = b->w / p[6]; J[10] = b->w / b->width; J[15] = b->w / b->height;
      double ft[4];
      for (int r = 0; r < 4; ++r) { double sc; creg += 0.5 * huber_row(f[r], &sc, &ft[r]); for (int q = 0; q < 4; ++q) J[r * 4 + q] *= sc; }
      for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 4; ++r) g[6 + i] += J[r * 4 + i] * ft[r];
        for (int q = 0; q < 4; ++q) { double h = 0; for (int r = 0; r < 4; ++r) h += J[r * 4 + i] * J[r * 4 + q]; Bc[(6 + i) * 10 + 6 + q] += h; }
      }
      memcpy(b->regJ + c * 16, J, 128); memcpy(b->regf + c * 4, ft, 32);
    }
  }
  b->cost = cost + creg;
  double g2 = 0, gi = 0, hd = 0;
  for (int i = 0; i < b->n; ++i) { g2 += b->gc[i] * b->gc[i]; gi = fmax(gi, fabs(b->gc[i])); }
  for (size_t i = 0; i < (size_t)P * 3; ++i) { g2 += b->gp[i] * b->gp[i]; gi = fmax(gi, fabs(b->gp[i])); }
  for (int c = 0; c < C; ++c) for (int a = 0; a < d; ++a) hd = fmax(hd, b->B[(size_t)c * d * d + a * d + a]);
  for (int j = 0; j < P; ++j) hd = fmax(hd, fmax(b->Cp[(size_t)j * 6], fmax(b->Cp[(size_t)j * 6 + 3], b->Cp[(size_t)j * 6 + 5])));
  b->gnorm = sqrt(g2); b->ginf = gi; b->hdiag = hd;
}

/* in-place lower Cholesky, row-major n x n, blocked + OpenMP; returns 0 on success */
static int chol(double* A, int n) {
  const int NB = 64;
  for (int j0 = 0; j0 < n; j0 += NB) {
    const int nb = (n - j0) < NB ? (n - j0) : NB, j1 = j0 + nb;
    for (int j = j0; j < j1; ++j) {
      double s = A[(size_t)j * n + j];
      for (int k = j0; k < j; ++k) s -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
      if (!(s > 0.0)) return 1;
      const double l = sqrt(s); A[(size_t)j * n + j] = l;
      for (int i = j + 1; i < j1; ++i) {
        double t = A[(size_t)i * n + j];
        for (int k = j0; k < j; ++k) t -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        A[(size_t)i * n + j] = t / l;
      }
    }
#pragma omp parallel for schedule(static)
    for (int i = j1; i < n; ++i) {
      double* ai = A + (size_t)i * n;
      for (int j = j0; j < j1; ++j) {
        double t = ai[j]; const double* aj = A + (size_t)j * n;
        for (int k = j0; k < j; ++k) t -= ai[k] * aj[k];
        ai[j] = t / aj[j];
      }
    }
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = j1; i < n; ++i) {
      double* ai = A + (size_t)i * n;
      for (int q = j1; q <= i; ++q) {
        const double* aq = A + (size_t)q * n; double t = 0.0;
        for (int k = j0; k < j1; ++k) t += ai[k] * aq[k];
        ai[q] -= t;
      }
    }
  }
  return 0;
}
static void fwd(const double* L, int n, double* b) {
  for (int i = 0; i < n; ++i) { double t = b[i]; const double* l = L + (size_t)i * n; for (int k = 0; k < i; ++k) t -= l[k] * b[k]; b[i] = t / l[i]; }
}
static void bwd(const double* L, int n, double* b) {
  for (int i = n - 1; i >= 0; --i) { double t = b[i]; for (int k = i + 1; k < n; ++k) t -= L[(size_t)k * n + i] * b[k]; b[i] = t / L[(size_t)i * n + i]; }
}

/* p = -(H + alpha I)^-1 g by point elimination; optionally p^T (H + alpha I)^-1 p.  returns 0 / 1 (not PD) */
int bao_solve(bao* b, double alpha, int want_q, double* pnorm, double* pq) {
  const int C = b->C, P = b->P, d = b->d, n = b->n;
#pragma omp parallel for schedule(static)
  for (int j = 0; j < P; ++j) {
    const double* c = b->Cp + (size_t)j * 6;
    const double a00 = c[0] + alpha, a10 = c[1], a20 = c[2], a11 = c[3] + alpha, a21 = c[4], a22 = c[5] + alpha;
    const double l00 = sqrt(a00), l10 = a10 / l00, l20 = a20 / l00, l11 = sqrt(a11 - l10 * l10), l21 = (a21 - l20 * l10) / l11,
                 l22 = sqrt(a22 - l20 * l20 - l21 * l21);
    const double m00 = 1.0 / l00, m11 = 1.0 / l11, m22 = 1.0 / l22, m10 = -l10 * m00 * m11, m21 = -l21 * m11 * m22,
                 m20 = -(l20 * m00 + l21 * m10) * m22;
    double* M = b->Minv + (size_t)j * 6; M[0] = m00; M[1] = m10; M[2] = m11; M[3] = m20; M[4] = m21; M[5] = m22;
    const double* g = b->gp + (size_t)j * 3; double* e = b->e + (size_t)j * 3;
    e[0] = m00 * g[0]; e[1] = m10 * g[0] + m11 * g[1]; e[2] = m20 * g[0] + m21 * g[1] + m22 * g[2];
    for (int k = b->pt_ptr[j]; k < b->pt_ptr[j + 1]; ++k) {
      const double *jc = b->Jc + (size_t)k * 2 * d, *jp = b->Jp + (size_t)k * 6; double* G = b->G + (size_t)k * 3 * d;
      const double v00 = jp[0] * m00, v10 = jp[3] * m00, v01 = jp[0] * m10 + jp[1] * m11, v11 = jp[3] * m10 + jp[4] * m11,
                   v02 = jp[0] * m20 + jp[1] * m21 + jp[2] * m22, v12 = jp[3] * m20 + jp[4] * m21 + jp[5] * m22;
      for (int a = 0; a < d; ++a) { G[a] = jc[a] * v00 + jc[d + a] * v10; G[d + a] = jc[a] * v01 + jc[d + a] * v11; G[2 * d + a] = jc[a] * v02 + jc[d + a] * v12; }
    }
  }
  /* S row-block of camera c: B_c + alpha I - sum_{k in c} G_k sum_{k' on track} G_k'^T ; r_c = g_c - sum G_k e_j */
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < C; ++c) {
    double* Sr = b->S + (size_t)c * d * n;
    memset(Sr, 0, (size_t)d * n * 8);
    for (int a = 0; a < d; ++a) { for (int q = 0; q < d; ++q) Sr[(size_t)a * n + c * d + q] = b->B[(size_t)c * d * d + a * d + q]; Sr[(size_t)a * n + c * d + a] += alpha; }
    double* r = b->r + (size_t)c * d; memcpy(r, b->gc + (size_t)c * d, d * 8);
    for (int i0 = b->cam_ptr[c]; i0 < b->cam_ptr[c + 1]; ++i0) {
      const int i = BAO_ORDER(i0, b->cam_ptr[c], b->cam_ptr[c + 1]);
      const int k = b->cam_obs[i], j = b->pt_idx[k];
      const double *G = b->G + (size_t)k * 3 * d, *e = b->e + (size_t)j * 3;
      for (int a = 0; a < d; ++a) r[a] -= G[a] * e[0] + G[d + a] * e[1] + G[2 * d + a] * e[2];
      for (int k2 = b->pt_ptr[j]; k2 < b->pt_ptr[j + 1]; ++k2) {
        const double* G2 = b->G + (size_t)k2 * 3 * d; const int c2 = b->cam_idx[k2];
        for (int a = 0; a < d; ++a) {
          double* row = Sr + (size_t)a * n + c2 * d; const double g0 = G[a], g1 = G[d + a], g2 = G[2 * d + a];
          for (int q = 0; q < d; ++q) row[q] -= g0 * G2[q] + g1 * G2[d + q] + g2 * G2[2 * d + q];
        }
      }
    }
  }
  if (chol(b->S, n)) return 1;
  for (int i = 0; i < n; ++i) b->pc[i] = -b->r[i];
  fwd(b->S, n, b->pc); bwd(b->S, n, b->pc);
  double pp2 = 0.0, v2 = 0.0;
#pragma omp parallel for reduction(+ : pp2, v2) schedule(static)
  for (int j = 0; j < P; ++j) {
    const double* e = b->e + (size_t)j * 3; double u0 = e[0], u1 = e[1], u2 = e[2];
    for (int k = b->pt_ptr[j]; k < b->pt_ptr[j + 1]; ++k) {
      const double *G = b->G + (size_t)k * 3 * d, *p = b->pc + (size_t)b->cam_idx[k] * d;
      for (int a = 0; a < d; ++a) { u0 += G[a] * p[a]; u1 += G[d + a] * p[a]; u2 += G[2 * d + a] * p[a]; }
    }
    const double* M = b->Minv + (size_t)j * 6;
    const double q0 = -(M[0] * u0 + M[1] * u1 + M[3] * u2), q1 = -(M[2] * u1 + M[4] * u2), q2 = -(M[5] * u2);
    double* pp = b->pp + (size_t)j * 3; pp[0] = q0; pp[1] = q1; pp[2] = q2;
    double* v = b->v + (size_t)j * 3; v[0] = M[0] * q0; v[1] = M[1] * q0 + M[2] * q1; v[2] = M[3] * q0 + M[4] * q1 + M[5] * q2;
    pp2 += q0 * q0 + q1 * q1 + q2 * q2; v2 += v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  }
  double pc2 = 0.0; for (int i = 0; i < n; ++i) pc2 += b->pc[i] * b->pc[i];
  *pnorm = sqrt(pc2 + pp2);
  *pq = 0.0;
  if (want_q) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < C; ++c) {
      double* y = b->y + (size_t)c * d; memcpy(y, b->pc + (size_t)c * d, d * 8);
      for (int i = b->cam_ptr[c]; i < b->cam_ptr[c + 1]; ++i) {
        const int k = b->cam_obs[i]; const double *G = b->G + (size_t)k * 3 * d, *v = b->v + (size_t)b->pt_idx[k] * 3;
        for (int a = 0; a < d; ++a) y[a] -= G[a] * v[0] + G[d + a] * v[1] + G[2 * d + a] * v[2];
      }
    }
    fwd(b->S, n, b->y);
    double y2 = 0.0; for (int i = 0; i < n; ++i) y2 += b->y[i] * b->y[i];
    *pq = v2 + y2;
  }
  return 0;
}

/* s = scale p; x_new = x + s; out = [ ||J~ s||^2, g^T s, cost(x_new), ||s||, ||x_new|| ] */
void bao_step(bao* b, const double* x, double scale, double* x_new, double* out) {
  const int C = b->C, P = b->P, d = b->d, n = b->n;
  double s2 = 0.0, x2 = 0.0;
  for (int i = 0; i < n; ++i) { const double s = scale * b->pc[i]; x_new[i] = x[i] + s; s2 += s * s; x2 += x_new[i] * x_new[i]; }
  for (size_t i = 0; i < (size_t)P * 3; ++i) { const double s = scale * b->pp[i]; x_new[n + i] = x[n + i] + s; s2 += s * s; x2 += x_new[n + i] * x_new[n + i]; }
  double js2 = 0.0, gts = 0.0;
#pragma omp parallel for reduction(+ : js2, gts) schedule(static)
  for (int64_t k = 0; k < b->N; ++k) {
    const double *jc = b->Jc + (size_t)k * 2 * d, *jp = b->Jp + (size_t)k * 6, *c = b->pc + (size_t)b->cam_idx[k] * d, *q = b->pp + (size_t)b->pt_idx[k] * 3;
    for (int row = 0; row < 2; ++row) {
      double t = jp[row * 3] * q[0] + jp[row * 3 + 1] * q[1] + jp[row * 3 + 2] * q[2];
      for (int a = 0; a < d; ++a) t += jc[row * d + a] * c[a];
      t *= scale; js2 += t * t; gts += b->ft[2 * k + row] * t;
    }
  }
  if (b->apply_reg)
    for (int c = 0; c < C; ++c) for (int r = 0; r < 4; ++r) {
      const double *J = b->regJ + c * 16 + r * 4, *s = b->pc + (size_t)c * 10 + 6;
      const double t = scale * (J[0] * s[0] + J[1] * s[1] + J[2] * s[2] + J[3] * s[3]);
      js2 += t * t; gts += b->regf[c * 4 + r] * t;
    }
  out[0] = js2; out[1] = gts; out[2] = bao_cost(b, x_new); out[3] = sqrt(s2); out[4] = sqrt(x2);
}

/* SciPy trf_no_bounds + More'.  x is updated in place.  res = [cost, nfev, njev, status, n_solves, optimality] */
int bao_trf(bao* b, double* x, double ftol, double xtol, double gtol, int max_nfev, int max_outer, int check_tol, double* res) {
  const size_t nv = (size_t)b->n + (size_t)b->P * 3;
  double* x_new = malloc(nv * 8);
  bao_linearize(b, x);
  double cost = b->cost; int nfev = 1, njev = 1, status = -1, iteration = 0, n_solves = 0;
  double xn = 0.0; for (size_t i = 0; i < nv; ++i) xn += x[i] * x[i]; xn = sqrt(xn);
  double Delta = xn > 0 ? xn : 1.0, alpha = 0.0;
  for (;;) {
    if (check_tol && b->ginf < gtol) status = 1;
    if (status >= 0 || nfev == max_nfev) break;
    if (max_outer >= 0 && iteration >= max_outer) break;
    double actual = -1.0, cost_new = cost, xnew_norm = xn;
    const double floor_a = ALPHA_FLOOR_REL * b->hdiag;
    while (actual <= 0 && nfev < max_nfev) {
      double au = b->gnorm / Delta, al = 0.0, pn = 0.0, pq = 0.0;
      if (alpha == 0.0) alpha = fmax(0.001 * au, sqrt(al * au));
      int interior = 0;
      for (int it = 0; it < 10; ++it) {
        if (alpha < al || alpha > au) alpha = fmax(0.001 * au, sqrt(al * au));
        const int on_floor = alpha <= floor_a;
        if (on_floor) alpha = floor_a;
        if (bao_solve(b, alpha, 1, &pn, &pq)) { free(x_new); return -1; }
        n_solves++;
        const double phi = pn - Delta;
        if (on_floor && phi < 0) { interior = 1; break; }
        const double phip = -pq / pn;
        if (phi < 0) au = alpha;
        const double ratio = phi / phip;
        al = fmax(al, alpha - ratio);
        alpha -= (phi + Delta) * ratio / Delta;
        if (fabs(phi) < 0.01 * Delta) break;
      }
      if (!interior) {
        alpha = fmax(fmax(alpha, floor_a), 1e-300);
        if (bao_solve(b, alpha, 0, &pn, &pq)) { free(x_new); return -1; }
        n_solves++;
      }
      double o[5];
      bao_step(b, x, Delta / pn, x_new, o);
      const double predicted = -(0.5 * o[0] + o[1]);
      cost_new = o[2]; nfev++;
      const double step_norm = o[3]; xnew_norm = o[4];
      if (!isfinite(cost_new)) { Delta = 0.25 * step_norm; continue; }
      actual = cost - cost_new;
      double ratio;
      if (predicted > 0) ratio = actual / predicted; else if (predicted == 0 && actual == 0) ratio = 1; else ratio = 0;
      double Delta_new = Delta;
      if (ratio < 0.25) Delta_new = 0.25 * step_norm; else if (ratio > 0.75 && step_norm > 0.95 * Delta) Delta_new = 2.0 * Delta;
      if (check_tol) {
        const int f_ok = (actual < ftol * cost) && ratio > 0.25, x_ok = step_norm < xtol * (xtol + xn);
        if (f_ok && x_ok) status = 4; else if (f_ok) status = 2; else if (x_ok) status = 3;
        if (status >= 0) break;
      }
      alpha *= Delta / Delta_new; Delta = Delta_new;
    }
    if (actual > 0) { memcpy(x, x_new, nv * 8); xn = xnew_norm; cost = cost_new; bao_linearize(b, x); njev++; }
    iteration++;
  }
  if (status < 0) status = 0;
  res[0] = cost; res[1] = nfev; res[2] = njev; res[3] = status; res[4] = n_solves; res[5] = b->ginf;
  free(x_new);
  return 0;
}

void bao_get(bao* b, double* out4) { out4[0] = b->cost; out4[1] = b->gnorm; out4[2] = b->ginf; out4[3] = b->hdiag; }
void bao_get_step(bao* b, double* pc, double* pp) { memcpy(pc, b->pc, (size_t)b->n * 8); memcpy(pp, b->pp, (size_t)b->P * 24); }
void bao_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int bao_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---------------------------------------------------------------- matcher: uint8 L2 kNN(2), ties -> lower index */
void mo_knn2_u8(const uint8_t* q, int64_t nq, const uint8_t* t, int64_t nt, int dim, int32_t* idx1, int32_t* idx2, float* d1, float* d2) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nq; ++i) {
    const uint8_t* a = q + i * dim;
    float b1 = INFINITY, b2 = INFINITY; int32_t i1 = -1, i2 = -1;
    for (int64_t j = 0; j < nt; ++j) {
      const uint8_t* c = t + j * dim; int32_t s = 0;
      for (int k = 0; k < dim; ++k) { const int32_t df = (int32_t)a[k] - (int32_t)c[k]; s += df * df; }
      const float dist = sqrtf((float)s);          /* compared after sqrtf, as the float32 DMatch.distance */
      if (dist < b1) { b2 = b1; i2 = i1; b1 = dist; i1 = (int32_t)j; }
      else if (dist < b2) { b2 = dist; i2 = (int32_t)j; }
    }
    idx1[i] = i1; idx2[i] = i2; d1[i] = b1; d2[i] = b2;
  }
}
