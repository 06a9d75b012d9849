// Dense SPD solver of the reduced camera system (dense.hip).
#pragma once
#include "common.h"

struct DenseWs {
  double* panel;   // [(n+1)][64] current panel
  double* Ld;      // [64][64] inverse of the diagonal block of the current step
  double* rd;      // unused spare
  double* Dinv;    // [ceil(n/128)][128][128] inverses of the diagonal blocks of L, row-major
  double* DinvT;   // same, transposed
  double* inv64;   // [2*ceil(n/128)][64][64]
  double* tmp;     // [ceil(n/128)][64][64]
  int* flag;       // set to 1 when a pivot is not positive
};

int64_t dense_ws_doubles(int n);
void dense_ws_carve(double* base, int n, DenseWs* out);
// In-place lower Cholesky of the leading n x n of A ([nrows][n] row-major, nrows = n or n + 1: a last
// row is carried along as a right-hand side and leaves as L^-1 rhs); then the block inverses for trsv.
int dense_cholesky(sfm_ctx* h, double* A, int n, int nrows, const DenseWs& w);
// Solve L y = b (transpose 0) or L^T x = b (transpose 1): b is destroyed, the solution goes to xout.
int dense_trsv(sfm_ctx* h, const double* L, int n, const DenseWs& w, double* b, double* xout, int transpose);
