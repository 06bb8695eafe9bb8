// Pose solvers: batched weighted Kabsch (scripts/SC2_PCR/common.py:7-45 of the reference) and the
// IRLS small-angle solver (util/transform_estimation.py:89-116).  Both are reductions over a few
// thousand points followed by a tiny dense solve, so they run as one workgroup per problem with
// fp64 accumulation; the 3x3 SVD / 6x6 solve happens in lane 0 (no CPU hop, unlike the reference's
// torch.svd(H.cpu())).
#include "pose_math.h"

using namespace eyoc;

namespace {

constexpr int KB_THREADS = 256;

template <int NV>
__device__ inline void block_sum(double (&v)[NV], double* lds /*[waves][NV]*/, int nwaves) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = wave_sum(v[i]);
    if (lane == 0) lds[wave * NV + i] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 0; i < NV; ++i) {
      double s = 0;
      for (int w = 0; w < nwaves; ++w) s += lds[w * NV + i];
      lds[i] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = lds[i];
  __syncthreads();
}

// one workgroup per batch element
__global__ __launch_bounds__(KB_THREADS) void kabsch_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                            const float* __restrict__ W, int n, float* __restrict__ T) {
  __shared__ double red[(KB_THREADS / 64) * 9];
  const int b = blockIdx.x;
  const float* a = A + (size_t)b * n * 3;
  const float* bb = B + (size_t)b * n * 3;
  const float* w = W ? W + (size_t)b * n : nullptr;
  // pass 1: weighted centroids (denominator + 1e-6 as in the reference)
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += KB_THREADS) {
    const double wi = w ? (double)w[i] : 1.0;
    s[0] += wi;
    s[1] += wi * a[3 * i]; s[2] += wi * a[3 * i + 1]; s[3] += wi * a[3 * i + 2];
    s[4] += wi * bb[3 * i]; s[5] += wi * bb[3 * i + 1]; s[6] += wi * bb[3 * i + 2];
  }
  block_sum<7>(s, red, KB_THREADS / 64);
  const double den = s[0] + 1e-6;
  const double ca[3] = {s[1] / den, s[2] / den, s[3] / den};
  const double cb[3] = {s[4] / den, s[5] / den, s[6] / den};
  // pass 2: H = sum w (a - ca)(b - cb)^T
  double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += KB_THREADS) {
    const double wi = w ? (double)w[i] : 1.0;
    const double ax = a[3 * i] - ca[0], ay = a[3 * i + 1] - ca[1], az = a[3 * i + 2] - ca[2];
    const double bx = (bb[3 * i] - cb[0]) * wi, by = (bb[3 * i + 1] - cb[1]) * wi, bz = (bb[3 * i + 2] - cb[2]) * wi;
    h[0] += ax * bx; h[1] += ax * by; h[2] += ax * bz;
    h[3] += ay * bx; h[4] += ay * by; h[5] += ay * bz;
    h[6] += az * bx; h[7] += az * by; h[8] += az * bz;
  }
  block_sum<9>(h, red, KB_THREADS / 64);
  if (threadIdx.x == 0) {
    double H[3][3] = {{h[0], h[1], h[2]}, {h[3], h[4], h[5]}, {h[6], h[7], h[8]}};
    double R[3][3], t[3];
    kabsch_rotation(H, R);
    for (int i = 0; i < 3; ++i) t[i] = cb[i] - (R[i][0] * ca[0] + R[i][1] * ca[1] + R[i][2] * ca[2]);
    write_T(T + 16 * (size_t)b, R, t);
  }
}

// small-n variant: one wave per batch element, 4 problems per workgroup
__global__ __launch_bounds__(256) void kabsch_wave_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          const float* __restrict__ W, int bs, int n,
                                                          float* __restrict__ T) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= bs) return;
  const float* a = A + (size_t)b * n * 3;
  const float* bb = B + (size_t)b * n * 3;
  const float* w = W ? W + (size_t)b * n : nullptr;
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = lane; i < n; i += 64) {
    const double wi = w ? (double)w[i] : 1.0;
    s[0] += wi;
    s[1] += wi * a[3 * i]; s[2] += wi * a[3 * i + 1]; s[3] += wi * a[3 * i + 2];
    s[4] += wi * bb[3 * i]; s[5] += wi * bb[3 * i + 1]; s[6] += wi * bb[3 * i + 2];
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) s[i] = __shfl(wave_sum(s[i]), 0, 64);
  const double den = s[0] + 1e-6;
  const double ca[3] = {s[1] / den, s[2] / den, s[3] / den};
  const double cb[3] = {s[4] / den, s[5] / den, s[6] / den};
  double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = lane; i < n; i += 64) {
    const double wi = w ? (double)w[i] : 1.0;
    const double ax = a[3 * i] - ca[0], ay = a[3 * i + 1] - ca[1], az = a[3 * i + 2] - ca[2];
    const double bx = (bb[3 * i] - cb[0]) * wi, by = (bb[3 * i + 1] - cb[1]) * wi, bz = (bb[3 * i + 2] - cb[2]) * wi;
    h[0] += ax * bx; h[1] += ax * by; h[2] += ax * bz;
    h[3] += ay * bx; h[4] += ay * by; h[5] += ay * bz;
    h[6] += az * bx; h[7] += az * by; h[8] += az * bz;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) h[i] = wave_sum(h[i]);
  if (lane == 0) {
    double H[3][3] = {{h[0], h[1], h[2]}, {h[3], h[4], h[5]}, {h[6], h[7], h[8]}};
    double R[3][3], t[3];
    kabsch_rotation(H, R);
    for (int i = 0; i < 3; ++i) t[i] = cb[i] - (R[i][0] * ca[0] + R[i][1] * ca[1] + R[i][2] * ca[2]);
    write_T(T + 16 * (size_t)b, R, t);
  }
}

// ------------------------------------------------------------------------------------------------
// IRLS (util/transform_estimation.py:89-116): one workgroup, all iterations inside the kernel.
// Unknown x = (rx, ry, rz, tx, ty, tz); per point the three residual rows are
//   [0, z, -y, 1, 0, 0], [-z, 0, x, 0, 1, 0], [y, -x, 0, 0, 0, 1]  (each scaled by the weight),
// so the 6x6 normal matrix and right-hand side are 16 weighted moments of the current points.
// The accumulated transform is applied to the ORIGINAL points every iteration.
// ------------------------------------------------------------------------------------------------
constexpr int IRLS_THREADS = 1024;

__device__ inline bool solve6(double M[6][7]) {  // Gauss-Jordan with partial pivoting, solution in column 6
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    double best = fabs(M[c][c]);
    for (int r = c + 1; r < 6; ++r)
      if (fabs(M[r][c]) > best) { best = fabs(M[r][c]); piv = r; }
    if (!(best > 0)) return false;
    if (piv != c)
      for (int k = 0; k < 7; ++k) { double t = M[c][k]; M[c][k] = M[piv][k]; M[piv][k] = t; }
    const double inv = 1.0 / M[c][c];
    for (int k = c; k < 7; ++k) M[c][k] *= inv;
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      const double f = M[r][c];
      if (f != 0)
        for (int k = c; k < 7; ++k) M[r][k] -= f * M[c][k];
    }
  }
  return true;
}

__global__ __launch_bounds__(IRLS_THREADS) void irls_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                            const float* __restrict__ w0, int n, int iters,
                                                            float* __restrict__ Tout) {
  __shared__ double red[(IRLS_THREADS / 64) * 16];
  __shared__ double Tsh[12];  // current accumulated [R | t], row-major 3x4
  if (threadIdx.x < 12) Tsh[threadIdx.x] = (threadIdx.x % 5 == 0) ? 1.0 : 0.0;  // entries 0,5,10 = identity
  __syncthreads();
  // the reference computes the weights of iteration i at the END of iteration i-1, i.e. with the
  // value `par` had there; `par` itself halves at the top of iterations 5, 10, 15
  double par = 1.0, par_w = 1.0;
  for (int it = 0; it < iters; ++it) {
    par_w = par;
    if (it > 0 && it % 5 == 0) par *= 0.5;
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = Tsh[i];
    double s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0;
    for (int i = threadIdx.x; i < n; i += IRLS_THREADS) {
      const double ox = p0[3 * i], oy = p0[3 * i + 1], oz = p0[3 * i + 2];
      const double x = T[0] * ox + T[1] * oy + T[2] * oz + T[3];
      const double y = T[4] * ox + T[5] * oy + T[6] * oz + T[7];
      const double z = T[8] * ox + T[9] * oy + T[10] * oz + T[11];
      const double rx = p1[3 * i] - x, ry = p1[3 * i + 1] - y, rz = p1[3 * i + 2] - z;
      double w;
      if (it == 0) w = w0 ? (double)w0[i] : 1.0;
      else w = par_w / (sqrt(rx * rx + ry * ry + rz * rz) + par_w);
      const double w2 = w * w;
      s[0] += w2 * x * x; s[1] += w2 * y * y; s[2] += w2 * z * z;
      s[3] += w2 * x * y; s[4] += w2 * x * z; s[5] += w2 * y * z;
      s[6] += w2 * x; s[7] += w2 * y; s[8] += w2 * z; s[9] += w2;
      s[10] += w2 * (y * rz - z * ry); s[11] += w2 * (z * rx - x * rz); s[12] += w2 * (x * ry - y * rx);
      s[13] += w2 * rx; s[14] += w2 * ry; s[15] += w2 * rz;
    }
    block_sum<16>(s, red, IRLS_THREADS / 64);
    if (threadIdx.x == 0) {
      const double xx = s[0], yy = s[1], zz = s[2], xy = s[3], xz = s[4], yz = s[5];
      const double sx = s[6], sy = s[7], sz = s[8], s1 = s[9];
      double M[6][7] = {
          {yy + zz, -xy, -xz, 0, -sz, sy, s[10]},
          {-xy, xx + zz, -yz, sz, 0, -sx, s[11]},
          {-xz, -yz, xx + yy, -sy, sx, 0, s[12]},
          {0, sz, -sy, s1, 0, 0, s[13]},
          {-sz, 0, sx, 0, s1, 0, s[14]},
          {sy, -sx, 0, 0, 0, s1, s[15]}};
      double q[6] = {0, 0, 0, 0, 0, 0};
      if (solve6(M))
        for (int i = 0; i < 6; ++i) q[i] = M[i][6];
      else
        for (int i = 0; i < 6; ++i) q[i] = __builtin_nan("");  // singular system: NaN pose, like inverse() of a singular matrix
      const double cx = cos(q[0]), sxx = sin(q[0]), cy = cos(q[1]), syy = sin(q[1]), cz = cos(q[2]), szz = sin(q[2]);
      // R = Rz(rz) Ry(ry) Rx(rx)   (util/transform_estimation.py:41-45)
      const double Ri[3][3] = {{cz * cy, cz * syy * sxx - szz * cx, cz * syy * cx + szz * sxx},
                               {szz * cy, szz * syy * sxx + cz * cx, szz * syy * cx - cz * sxx},
                               {-syy, cy * sxx, cy * cx}};
      double Tn[12];
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 4; ++j) Tn[4 * i + j] = Ri[i][0] * T[j] + Ri[i][1] * T[4 + j] + Ri[i][2] * T[8 + j];
        Tn[4 * i + 3] += q[3 + i];
      }
      for (int i = 0; i < 12; ++i) Tsh[i] = Tn[i];
    }
    __syncthreads();
  }
  if (threadIdx.x < 16) {
    const int i = threadIdx.x;
    Tout[i] = i < 12 ? (float)Tsh[i] : (i == 15 ? 1.0f : 0.0f);
  }
}

}  // namespace

extern "C" {

int eyoc_kabsch_batched(eyoc_ctx* ctx, const float* A_dev, const float* B_dev, const float* w_dev, int bs, int n,
                        float* T_dev, void* stream) {
  EYOC_REQUIRE(ctx && A_dev && B_dev && T_dev, EYOC_ERR_INVALID, "eyoc_kabsch_batched: NULL argument");
  EYOC_REQUIRE(bs >= 0 && n >= 1, EYOC_ERR_INVALID, "eyoc_kabsch_batched: bs %d n %d", bs, n);
  if (bs == 0) return EYOC_OK;
  hipStream_t st = (hipStream_t)stream;
  if (n <= 256)
    hipLaunchKernelGGL(kabsch_wave_kernel, dim3(cdiv(bs, 4)), dim3(256), 0, st, A_dev, B_dev, w_dev, bs, n, T_dev);
  else
    hipLaunchKernelGGL(kabsch_kernel, dim3(bs), dim3(KB_THREADS), 0, st, A_dev, B_dev, w_dev, n, T_dev);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

int eyoc_irls_quad(eyoc_ctx* ctx, const float* p0_dev, const float* p1_dev, const float* w_dev, int n, int iters,
                   float* T_dev, void* stream) {
  EYOC_REQUIRE(ctx && p0_dev && p1_dev && T_dev, EYOC_ERR_INVALID, "eyoc_irls_quad: NULL argument");
  EYOC_REQUIRE(n >= 1 && iters >= 0, EYOC_ERR_INVALID, "eyoc_irls_quad: n %d iters %d", n, iters);
  hipLaunchKernelGGL(irls_kernel, dim3(1), dim3(IRLS_THREADS), 0, (hipStream_t)stream, p0_dev, p1_dev, w_dev, n, iters,
                     T_dev);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

}  // extern "C"
