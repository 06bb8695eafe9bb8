// Small fp64 device helpers shared by the pose kernels: symmetric 3x3 Jacobi eigen-solver,
// Kabsch rotation from a 3x3 cross-covariance, wave / block reductions.
#pragma once
#include "common.h"

namespace eyoc {

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
  return v;  // valid in lane 0
}

// eigen-decomposition of a symmetric 3x3 matrix (cyclic Jacobi, fp64): A = V diag(w) V^T,
// eigenvalues sorted descending, V column-major in v[col][row]
__device__ inline void jacobi_eig3(double a[3][3], double w[3], double v[3][3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    if (off <= 1e-30 + 1e-17 * diag) break;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const double apq = a[p][q];
      if (fabs(apq) < 1e-300) continue;
      const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
      for (int k = 0; k < 3; ++k) {  // A <- A J
        const double akp = a[k][p], akq = a[k][q];
        a[k][p] = c * akp - s * akq;
        a[k][q] = s * akp + c * akq;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {  // A <- J^T A
        const double apk = a[p][k], aqk = a[q][k];
        a[p][k] = c * apk - s * aqk;
        a[q][k] = s * apk + c * aqk;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {  // V <- V J   (v[col][row])
        const double vkp = v[p][k], vkq = v[q][k];
        v[p][k] = c * vkp - s * vkq;
        v[q][k] = s * vkp + c * vkq;
      }
    }
  }
  w[0] = a[0][0]; w[1] = a[1][1]; w[2] = a[2][2];
  // sort descending (3 elements)
#define EYOC_SWAP_EIG(i, j)                                                         \
  if (w[i] < w[j]) {                                                                \
    double tw = w[i]; w[i] = w[j]; w[j] = tw;                                       \
    for (int k = 0; k < 3; ++k) { double tv = v[i][k]; v[i][k] = v[j][k]; v[j][k] = tv; } \
  }
  EYOC_SWAP_EIG(0, 1) EYOC_SWAP_EIG(0, 2) EYOC_SWAP_EIG(1, 2)
#undef EYOC_SWAP_EIG
}

// Rotation of the Kabsch problem from H = sum w (a - ca)(b - cb)^T  (H[i][j], i over a, j over b):
// with H = U S V^T the answer is R = V diag(1,1,det(V U^T)) U^T  (scripts/SC2_PCR/common.py:36-41).
// V comes from the eigen-decomposition of H^T H, u_i = H v_i / s_i; u_3 = u_1 x u_2 makes det(U) = +1,
// which yields the same R as any valid SVD (the sign of the third pair cancels in the formula).
__device__ inline void kabsch_rotation(const double H[3][3], double R[3][3]) {
  double hth[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) hth[i][j] = H[0][i] * H[0][j] + H[1][i] * H[1][j] + H[2][i] * H[2][j];
  double w[3], v[3][3];
  jacobi_eig3(hth, w, v);
  if (!(w[0] > 1e-300)) {  // H == 0 (no weight at all): identity, like an SVD of the zero matrix
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[i][j] = i == j ? 1.0 : 0.0;
    return;
  }
  double u[3][3];  // u[col][row]
  // u1 = H v1 / |H v1|
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) u[c][r] = H[r][0] * v[c][0] + H[r][1] * v[c][1] + H[r][2] * v[c][2];
  double n0 = sqrt(u[0][0] * u[0][0] + u[0][1] * u[0][1] + u[0][2] * u[0][2]);
#pragma unroll
  for (int r = 0; r < 3; ++r) u[0][r] /= n0;
  // u2: orthogonalise against u1; if H has rank 1 pick any unit vector orthogonal to u1
  double d01 = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
#pragma unroll
  for (int r = 0; r < 3; ++r) u[1][r] -= d01 * u[0][r];
  double n1 = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
  if (n1 > 1e-14 * n0) {
#pragma unroll
    for (int r = 0; r < 3; ++r) u[1][r] /= n1;
  } else {
    const int m = fabs(u[0][0]) <= fabs(u[0][1]) ? (fabs(u[0][0]) <= fabs(u[0][2]) ? 0 : 2)
                                                 : (fabs(u[0][1]) <= fabs(u[0][2]) ? 1 : 2);
    double e[3] = {0, 0, 0};
    e[m] = 1.0;
    const double d = u[0][m];
#pragma unroll
    for (int r = 0; r < 3; ++r) u[1][r] = e[r] - d * u[0][r];
    n1 = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) u[1][r] /= n1;
  }
  u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
  u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
  u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
  // make V right-handed bookkeeping: d = det(V) (det(U) = +1)
  const double detv = v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1]) - v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0]) +
                      v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
  const double d = detv >= 0 ? 1.0 : -1.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i][j] = v[0][i] * u[0][j] + v[1][i] * u[1][j] + d * v[2][i] * u[2][j];
}

__device__ inline void write_T(float* T, const double R[3][3], const double t[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    T[4 * i + 0] = (float)R[i][0]; T[4 * i + 1] = (float)R[i][1]; T[4 * i + 2] = (float)R[i][2];
    T[4 * i + 3] = (float)t[i];
  }
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

}  // namespace eyoc
