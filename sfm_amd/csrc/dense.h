// Dense SPD solver of the reduced camera system (dense.hip).
#pragma once
#include "common.h"

struct DenseWs {
  double* Ld;      // 2 x [64][64] step data of the current / next diagonal block: [Li11 0; L21 Li22] (32x32 quadrants)
  double* Dinv;    // [ceil(n/128)][128][128] inverses of the diagonal blocks of L, row-major
  double* DinvT;   // same, transposed
  double* inv64;   // [2*ceil(n/128)][64][64] inverses of the 64x64 diagonal blocks
  int* flag;       // set to 1 when a pivot is not positive
  double* Lm;      // [(n+1)][n] the factor (lower part; row n = L^-1 rhs when the input carried a bordered row)
  double* LmT;     // [n][n] transposed copy of the sub-diagonal blocks of the factor (the forward solve reads columns)
};

int64_t dense_ws_doubles(int n);
int64_t dense_ws_lm_offset(int n);   // offset (doubles) of DenseWs::Lm inside the workspace
void dense_ws_carve(double* base, int n, DenseWs* out);
// Lower Cholesky of the leading n x n of A ([nrows][n] row-major, nrows = n or n + 1: a last row is carried
// along as a right-hand side and leaves as L^-1 rhs).  The factor goes to w.Lm (A's trailing part is
// consumed as scratch); then the block inverses for trsv.
int dense_cholesky(sfm_ctx* h, double* A, int n, int nrows, const DenseWs& w);
// Solve L y = b (transpose 0) or L^T x = b (transpose 1) with L = w.Lm: b is destroyed, the solution goes to xout.
int dense_trsv(sfm_ctx* h, int n, const DenseWs& w, double* b, double* xout, int transpose);
