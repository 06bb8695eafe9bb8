// Feature-matching RANSAC on the GPU (replaces Open3D's
// registration_ransac_based_on_feature_matching at scripts/test_kitti.py:169-177).
//
// Three phases, all on the device:
//   1. generate: one lane per hypothesis h - counter-hash sampler (bit-identical to
//      oracle/ransac.py), edge-length checker, 4-point Kabsch in fp64, distance checker.  Survivors
//      (typically well under a few percent) append h to a compact list with one atomic.
//   2. score: one WAVE per survivor re-derives its transform and sweeps all correspondences with
//      coalesced loads (a lane-per-hypothesis loop over 5000 points would idle 63 lanes whenever one
//      hypothesis survives) - inlier count and squared error are wave-reduced.
//   3. select: single workgroup arg-max with the total order (more inliers, lower RMSE, lower h).
#include "pose_math.h"

using namespace eyoc;

namespace {

// two sample indices per 64-bit word of the counter hash (low half, high half), each mapped to [0, n) by
// (u * n) >> 32 - one v_mul_hi_u32 - so a hypothesis costs two splitmix64 finalisers, not four
__device__ inline void sample_pair(unsigned long long base, unsigned long long ctr, unsigned int n, unsigned int& i0,
                                   unsigned int& i1) {
  unsigned long long x = base + ctr;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x = x ^ (x >> 31);
  i0 = __umulhi((unsigned int)x, n);
  i1 = __umulhi((unsigned int)(x >> 32), n);
}

// correspondence i as two 16-byte records (source point, its matched target point): the hypothesis generator
// fetches a sampled point with ONE 128-bit load instead of three scattered dwords
__global__ void k_gather_targets(const float* __restrict__ src, const float* __restrict__ tgt,
                                 const long long* __restrict__ corr, int n, float4* __restrict__ src4,
                                 float4* __restrict__ tc4) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long j = corr[i];
  src4[i] = make_float4(src[3 * i], src[3 * i + 1], src[3 * i + 2], 0.f);
  tc4[i] = make_float4(tgt[3 * j], tgt[3 * j + 1], tgt[3 * j + 2], 0.f);
}

// transform of hypothesis h, or false if a checker rejects it
__device__ inline bool hypothesis(const float4* __restrict__ src, const float4* __restrict__ tc, unsigned int n,
                                  unsigned long long base, unsigned int h, double edge_sim, double max_dist,
                                  double R[3][3], double t[3]) {
  double s[4][3], q[4][3];
  unsigned int idx[4];
  sample_pair(base, 2ull * h, n, idx[0], idx[1]);
  sample_pair(base, 2ull * h + 1, n, idx[2], idx[3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 a = src[idx[j]], b = tc[idx[j]];
    s[j][0] = a.x; s[j][1] = a.y; s[j][2] = a.z;
    q[j][0] = b.x; q[j][1] = b.y; q[j][2] = b.z;
  }
  // edge-length checker on squared lengths (no square roots): ds < e*dt  <=>  ds^2 < e^2 dt^2
  const double e2 = edge_sim * edge_sim;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a + 1; b < 4; ++b) {
      const double ds2 = (s[a][0] - s[b][0]) * (s[a][0] - s[b][0]) + (s[a][1] - s[b][1]) * (s[a][1] - s[b][1]) +
                         (s[a][2] - s[b][2]) * (s[a][2] - s[b][2]);
      const double dt2 = (q[a][0] - q[b][0]) * (q[a][0] - q[b][0]) + (q[a][1] - q[b][1]) * (q[a][1] - q[b][1]) +
                         (q[a][2] - q[b][2]) * (q[a][2] - q[b][2]);
      if (ds2 < dt2 * e2 || dt2 < ds2 * e2) return false;
    }
  double cs[3], cq[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    cs[d] = 0.25 * (s[0][d] + s[1][d] + s[2][d] + s[3][d]);
    cq[d] = 0.25 * (q[0][d] + q[1][d] + q[2][d] + q[3][d]);
  }
  double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) H[a][b] += (s[j][a] - cs[a]) * (q[j][b] - cq[b]);
  kabsch_rotation(H, R);
#pragma unroll
  for (int d = 0; d < 3; ++d) t[d] = cq[d] - (R[d][0] * cs[0] + R[d][1] * cs[1] + R[d][2] * cs[2]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double e2 = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double r = R[d][0] * s[j][0] + R[d][1] * s[j][1] + R[d][2] * s[j][2] + t[d] - q[j][d];
      e2 += r * r;
    }
    if (sqrt(e2) > max_dist) return false;
  }
  return true;
}

__global__ __launch_bounds__(256) void k_generate(const float4* __restrict__ src, const float4* __restrict__ tc, int n,
                                                  unsigned long long base, int H, float edge_sim, float max_dist,
                                                  int* __restrict__ n_surv, int* __restrict__ surv) {
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h >= H) return;
  double R[3][3], t[3];
  if (hypothesis(src, tc, (unsigned)n, base, (unsigned)h, (double)edge_sim, (double)max_dist, R, t))
    surv[atomicAdd(n_surv, 1)] = h;
}

// key: (inliers << 32) | ~bits(rmse_f32): larger is better; ties on the key are broken by lower h
__global__ __launch_bounds__(256) void k_score(const float4* __restrict__ src, const float4* __restrict__ tc, int n,
                                               unsigned long long base, float edge_sim, float max_dist,
                                               const int* __restrict__ n_surv, const int* __restrict__ surv,
                                               unsigned long long* __restrict__ keys) {
  const int lane = threadIdx.x & 63;
  const int ns = *n_surv;
  for (int sidx = blockIdx.x * 4 + (threadIdx.x >> 6); sidx < ns; sidx += gridDim.x * 4) {
    const int h = surv[sidx];
    double R[3][3], t[3];
    hypothesis(src, tc, (unsigned)n, base, (unsigned)h, (double)edge_sim, (double)max_dist, R, t);
    int cnt = 0;
    double err2 = 0;
    for (int i = lane; i < n; i += 64) {
      const float4 a = src[i], b = tc[i];
      const double x = a.x, y = a.y, z = a.z;
      const double dx = R[0][0] * x + R[0][1] * y + R[0][2] * z + t[0] - b.x;
      const double dy = R[1][0] * x + R[1][1] * y + R[1][2] * z + t[1] - b.y;
      const double dz = R[2][0] * x + R[2][1] * y + R[2][2] * z + t[2] - b.z;
      const double d2 = dx * dx + dy * dy + dz * dz;
      if (sqrt(d2) < (double)max_dist) { ++cnt; err2 += d2; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    err2 = wave_sum(err2);
    if (lane == 0) {
      const float rmse = cnt > 0 ? (float)sqrt(err2 / cnt) : __builtin_inff();
      keys[sidx] = ((unsigned long long)(unsigned)cnt << 32) | (unsigned long long)(~__float_as_uint(rmse));
    }
  }
}

__global__ __launch_bounds__(1024) void k_select(const float4* __restrict__ src, const float4* __restrict__ tc, int n,
                                                 unsigned long long base, float edge_sim, float max_dist,
                                                 const int* __restrict__ n_surv, const int* __restrict__ surv,
                                                 const unsigned long long* __restrict__ keys,
                                                 eyoc_ransac_result* __restrict__ out) {
  __shared__ unsigned long long bk[16];
  __shared__ int bh[16];
  const int ns = *n_surv;
  unsigned long long best_k = 0;
  int best_h = 0x7FFFFFFF;
  for (int i = threadIdx.x; i < ns; i += 1024) {
    const unsigned long long k = keys[i];
    const int h = surv[i];
    if (k > best_k || (k == best_k && h < best_h)) { best_k = k; best_h = h; }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long ok = __shfl_down(best_k, d, 64);
    const int oh = __shfl_down(best_h, d, 64);
    if (ok > best_k || (ok == best_k && oh < best_h)) { best_k = ok; best_h = oh; }
  }
  if ((threadIdx.x & 63) == 0) { bk[threadIdx.x >> 6] = best_k; bh[threadIdx.x >> 6] = best_h; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (bk[w] > best_k || (bk[w] == best_k && bh[w] < best_h)) { best_k = bk[w]; best_h = bh[w]; }
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, t[3] = {0, 0, 0};
    out->survivors = ns;
    if (ns > 0 && best_h != 0x7FFFFFFF) {
      hypothesis(src, tc, (unsigned)n, base, (unsigned)best_h, (double)edge_sim, (double)max_dist, R, t);
      out->inliers = (int)(best_k >> 32);
      out->best_hypothesis = best_h;
      out->inlier_rmse = __uint_as_float(~(unsigned)(best_k & 0xFFFFFFFFull));
    } else {  // nothing survived: identity, fitness 0 (Open3D returns its default-constructed result)
      out->inliers = 0;
      out->best_hypothesis = -1;
      out->inlier_rmse = 0.0f;
    }
    write_T(out->T, R, t);
  }
}

}  // namespace

extern "C" int eyoc_ransac(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int64_t* corr_tgt_dev, int n,
                           const eyoc_ransac_params* p, eyoc_ransac_result* result_dev, void* stream) {
  EYOC_REQUIRE(ctx && src_dev && tgt_dev && corr_tgt_dev && p && result_dev, EYOC_ERR_INVALID, "eyoc_ransac: NULL argument");
  EYOC_REQUIRE(n >= 4, EYOC_ERR_INVALID, "eyoc_ransac: need at least 4 correspondences, got %d", n);
  EYOC_REQUIRE(p->max_iteration >= 1, EYOC_ERR_INVALID, "eyoc_ransac: max_iteration %d", p->max_iteration);
  hipStream_t st = (hipStream_t)stream;
  const int H = p->max_iteration;
  // scratch: [counter 256 B][src4 n float4][tc4 n float4][surv H i32][keys H u64]
  const size_t off_src = 256, off_tc = align_up(off_src + (size_t)n * 16), off_surv = align_up(off_tc + (size_t)n * 16),
               off_keys = align_up(off_surv + (size_t)H * 4);
  int rc = ctx->ensure_scratch(off_keys + (size_t)H * 8);
  if (rc) return rc;
  char* sc = (char*)ctx->scratch;
  int* n_surv = (int*)sc;
  float4* src4 = (float4*)(sc + off_src);
  float4* tc = (float4*)(sc + off_tc);
  int* surv = (int*)(sc + off_surv);
  unsigned long long* keys = (unsigned long long*)(sc + off_keys);
  const unsigned long long base = (unsigned long long)p->seed * 0x9E3779B97F4A7C15ull;
  EYOC_CHECK_HIP(hipMemsetAsync(n_surv, 0, 256, st));
  hipLaunchKernelGGL(k_gather_targets, dim3(cdiv(n, 256)), dim3(256), 0, st, src_dev, tgt_dev, (const long long*)corr_tgt_dev, n,
                     src4, tc);
  hipLaunchKernelGGL(k_generate, dim3(cdiv(H, 256)), dim3(256), 0, st, src4, tc, n, base, H, p->edge_similarity,
                     p->max_distance, n_surv, surv);
  hipLaunchKernelGGL(k_score, dim3(2048), dim3(256), 0, st, src4, tc, n, base, p->edge_similarity, p->max_distance,
                     n_surv, surv, keys);
  hipLaunchKernelGGL(k_select, dim3(1), dim3(1024), 0, st, src4, tc, n, base, p->edge_similarity, p->max_distance,
                     n_surv, surv, keys, result_dev);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}
