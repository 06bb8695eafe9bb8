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

// Correspondence records of all pairs of a batch: 6 floats (source point, matched target point).
constexpr int CHUNK = 8;   // pairs per launch
struct PairArgs {          // a chunk of pairs; segment bounds travel as kernel arguments (no H2D copy)
  int s0[CHUNK], n[CHUNK], t0[CHUNK];   // first source row, correspondences, first target row of each pair
  int pair0;               // index of the chunk's first pair (seed offset, result slot)
  float* rec;              // [total, 6]
  unsigned int seed;       // pair b uses seed + b
  int H;
  float edge_sim, max_dist;
  int* n_surv;             // [chunk] survivor counters (CNT_STRIDE ints apart)
  int* surv;               // [chunk][H]
  unsigned long long* keys;  // [chunk][H]
};

__global__ void k_gather_targets(const float* __restrict__ src, const float* __restrict__ tgt,
                                 const long long* __restrict__ corr, PairArgs a) {
  const int c = blockIdx.y;
  const int s0 = a.s0[c], n = a.n[c];
  float* rec = a.rec;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long j = a.t0[c] + corr[s0 + i];
  float* r = rec + (size_t)(s0 + i) * 6;
  r[0] = src[3 * (size_t)(s0 + i)]; r[1] = src[3 * (size_t)(s0 + i) + 1]; r[2] = src[3 * (size_t)(s0 + i) + 2];
  r[3] = tgt[3 * j]; r[4] = tgt[3 * j + 1]; r[5] = tgt[3 * j + 2];
}

// transform of hypothesis h, or false if a checker rejects it.  `rec` may point to LDS or global memory.
__device__ inline bool hypothesis(const float* __restrict__ rec, unsigned int n, unsigned long long base, unsigned int h,
                                  double edge_sim, double max_dist, double R[3][3], double t[3]) {
  double s[4][3], q[4][3];
  unsigned int idx[4];
  sample_pair(base, 2ull * h, n, idx[0], idx[1]);
  sample_pair(base, 2ull * h + 1, n, idx[2], idx[3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float2* r = reinterpret_cast<const float2*>(rec + (size_t)idx[j] * 6);
    const float2 a = r[0], b = r[1], c = r[2];
    s[j][0] = a.x; s[j][1] = a.y; s[j][2] = b.x;
    q[j][0] = b.y; q[j][1] = c.x; q[j][2] = c.y;
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

constexpr int CNT_STRIDE = 64;
constexpr int GEN_THREADS = 1024;
constexpr int GEN_BLOCKS_MIN = 64, GEN_BLOCKS_TOTAL = 512;   // workgroups per pair: enough to fill the chip even for one pair
constexpr int LDS_RECORDS = 6400;                    // 6400 * 24 B = 150 KB of the 160 KB LDS

__device__ inline unsigned long long pair_base(unsigned int seed, int b) {
  return (unsigned long long)(seed + (unsigned)b) * 0x9E3779B97F4A7C15ull;
}

// One workgroup stages its pair's records in LDS (sampling is 8 random reads per hypothesis: L2 round trips
// dominated the first version of this kernel) and walks a strided slice of the hypotheses.
template <bool IN_LDS>
__global__ __launch_bounds__(GEN_THREADS) void k_generate(PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lrec[];
  const int c = blockIdx.y, b = a.pair0 + c;
  const int s0 = a.s0[c], n = a.n[c];
  const float* rec = a.rec + (size_t)s0 * 6;
  if (IN_LDS) {
    for (int i = threadIdx.x; i < n * 6 / 2; i += GEN_THREADS)
      reinterpret_cast<float2*>(lrec)[i] = reinterpret_cast<const float2*>(rec)[i];
    __syncthreads();
    rec = lrec;
  }
  const unsigned long long base = pair_base(a.seed, b);
  int* cnt = a.n_surv + c * CNT_STRIDE;
  int* surv = a.surv + (size_t)c * a.H;
  for (int h = blockIdx.x * GEN_THREADS + threadIdx.x; h < a.H; h += gridDim.x * GEN_THREADS) {
    double R[3][3], t[3];
    if (hypothesis(rec, (unsigned)n, base, (unsigned)h, (double)a.edge_sim, (double)a.max_dist, R, t))
      surv[atomicAdd(cnt, 1)] = h;
  }
}

// key: (inliers << 32) | ~bits(rmse_f32): larger is better; ties on the key are broken by lower h
__global__ __launch_bounds__(256) void k_score(PairArgs a) {
  const int c = blockIdx.y, b = a.pair0 + c;
  const int s0 = a.s0[c], n = a.n[c];
  const float* rec = a.rec + (size_t)s0 * 6;
  const unsigned long long base = pair_base(a.seed, b);
  const int lane = threadIdx.x & 63;
  const int ns = a.n_surv[c * CNT_STRIDE];
  const int* surv = a.surv + (size_t)c * a.H;
  unsigned long long* keys = a.keys + (size_t)c * a.H;
  for (int sidx = blockIdx.x * 4 + (threadIdx.x >> 6); sidx < ns; sidx += gridDim.x * 4) {
    const int h = surv[sidx];
    double R[3][3], t[3];
    hypothesis(rec, (unsigned)n, base, (unsigned)h, (double)a.edge_sim, (double)a.max_dist, R, t);
    int cnt = 0;
    double err2 = 0;
    for (int i = lane; i < n; i += 64) {
      const float2* r = reinterpret_cast<const float2*>(rec + (size_t)i * 6);
      const float2 p0 = r[0], p1 = r[1], p2 = r[2];
      const double x = p0.x, y = p0.y, z = p1.x;
      const double dx = R[0][0] * x + R[0][1] * y + R[0][2] * z + t[0] - p1.y;
      const double dy = R[1][0] * x + R[1][1] * y + R[1][2] * z + t[1] - p2.x;
      const double dz = R[2][0] * x + R[2][1] * y + R[2][2] * z + t[2] - p2.y;
      const double d2 = dx * dx + dy * dy + dz * dz;
      if (sqrt(d2) < (double)a.max_dist) { ++cnt; err2 += d2; }
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

__global__ __launch_bounds__(1024) void k_select(PairArgs a, eyoc_ransac_result* __restrict__ results) {
  __shared__ unsigned long long bk[16];
  __shared__ int bh[16];
  const int c = blockIdx.x, b = a.pair0 + c;
  const int s0 = a.s0[c], n = a.n[c];
  const float* rec = a.rec + (size_t)s0 * 6;
  const int ns = a.n_surv[c * CNT_STRIDE];
  const int* surv = a.surv + (size_t)c * a.H;
  const unsigned long long* keys = a.keys + (size_t)c * a.H;
  eyoc_ransac_result* out = results + b;
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
      hypothesis(rec, (unsigned)n, pair_base(a.seed, b), (unsigned)best_h, (double)a.edge_sim, (double)a.max_dist, R, t);
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

extern "C" int eyoc_ransac_batched(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int64_t* corr_tgt_dev,
                                   const int32_t* seg_src_host, const int32_t* seg_tgt_host, int n_pairs,
                                   const eyoc_ransac_params* p, eyoc_ransac_result* results_dev, void* stream) {
  EYOC_REQUIRE(ctx && src_dev && tgt_dev && corr_tgt_dev && seg_src_host && seg_tgt_host && p && results_dev, EYOC_ERR_INVALID,
               "eyoc_ransac_batched: NULL argument");
  EYOC_REQUIRE(n_pairs >= 1, EYOC_ERR_INVALID, "eyoc_ransac_batched: n_pairs %d", n_pairs);
  EYOC_REQUIRE(p->max_iteration >= 1, EYOC_ERR_INVALID, "eyoc_ransac: max_iteration %d", p->max_iteration);
  int max_n = 0;
  for (int b = 0; b < n_pairs; ++b) {
    const int n = seg_src_host[b + 1] - seg_src_host[b];
    EYOC_REQUIRE(n >= 4, EYOC_ERR_INVALID, "eyoc_ransac: need at least 4 correspondences, got %d (pair %d)", n, b);
    max_n = n > max_n ? n : max_n;
  }
  hipStream_t st = (hipStream_t)stream;
  const int H = p->max_iteration;
  const int total = seg_src_host[n_pairs];
  // survivor lists are sized for the worst case (every hypothesis survives): run the pairs in chunks so the
  // scratch stays bounded (12 bytes per hypothesis and pair of the chunk)
  const int chunk = n_pairs < CHUNK ? n_pairs : CHUNK;
  const size_t off_cnt = 0, off_rec = align_up((size_t)chunk * CNT_STRIDE * 4), off_surv = align_up(off_rec + (size_t)total * 24);
  const size_t off_keys = align_up(off_surv + (size_t)chunk * H * 4);
  int rc = ctx->ensure_scratch(off_keys + (size_t)chunk * H * 8);
  if (rc) return rc;
  char* sc = (char*)ctx->scratch;
  PairArgs a;
  a.rec = (float*)(sc + off_rec); a.seed = p->seed; a.H = H; a.edge_sim = p->edge_similarity; a.max_dist = p->max_distance;
  a.n_surv = (int*)(sc + off_cnt); a.surv = (int*)(sc + off_surv); a.keys = (unsigned long long*)(sc + off_keys);
  const bool in_lds = max_n <= LDS_RECORDS;
  const size_t lds_bytes = in_lds ? (size_t)max_n * 24 : 0;
  if (in_lds) {
    static bool attr_set = false;
    if (!attr_set) {
      EYOC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_generate<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         LDS_RECORDS * 24));
      attr_set = true;
    }
  }
  for (int p0 = 0; p0 < n_pairs; p0 += chunk) {
    const int nc = n_pairs - p0 < chunk ? n_pairs - p0 : chunk;
    a.pair0 = p0;
    int chunk_max = 0;
    for (int c = 0; c < CHUNK; ++c) {
      const int b = p0 + (c < nc ? c : 0);
      a.s0[c] = seg_src_host[b]; a.n[c] = seg_src_host[b + 1] - seg_src_host[b]; a.t0[c] = seg_tgt_host[b];
      chunk_max = a.n[c] > chunk_max ? a.n[c] : chunk_max;
    }
    EYOC_CHECK_HIP(hipMemsetAsync(a.n_surv, 0, (size_t)nc * CNT_STRIDE * 4, st));
    hipLaunchKernelGGL(k_gather_targets, dim3(cdiv(chunk_max, 256), nc), dim3(256), 0, st, src_dev, tgt_dev,
                       (const long long*)corr_tgt_dev, a);
    const int gen_blocks = GEN_BLOCKS_TOTAL / nc > GEN_BLOCKS_MIN ? GEN_BLOCKS_TOTAL / nc : GEN_BLOCKS_MIN;
    if (in_lds) hipLaunchKernelGGL(k_generate<true>, dim3(gen_blocks, nc), dim3(GEN_THREADS), lds_bytes, st, a);
    else hipLaunchKernelGGL(k_generate<false>, dim3(gen_blocks, nc), dim3(GEN_THREADS), 0, st, a);
    hipLaunchKernelGGL(k_score, dim3(256, nc), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_select, dim3(nc), dim3(1024), 0, st, a, results_dev);
  }
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

extern "C" int eyoc_ransac(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int64_t* corr_tgt_dev, int n,
                           const eyoc_ransac_params* p, eyoc_ransac_result* result_dev, void* stream) {
  EYOC_REQUIRE(n >= 4, EYOC_ERR_INVALID, "eyoc_ransac: need at least 4 correspondences, got %d", n);
  const int32_t seg_src[2] = {0, n}, seg_tgt[2] = {0, 0x7FFFFFFF};
  return eyoc_ransac_batched(ctx, src_dev, tgt_dev, corr_tgt_dev, seg_src, seg_tgt, 1, p, result_dev, stream);
}
