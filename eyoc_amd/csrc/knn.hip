// Brute-force 1-nearest-neighbour in feature space (lib/eval.py:18-48, lib/metrics.py:22-29 of the
// reference).  One lane owns one query row (its C features live in VGPRs); target rows stream
// through LDS in 256-row tiles and are read back as wave-wide broadcasts, so the kernel is pure
// VALU work: 3 ops per (query, target, channel).  The four waves of a block walk different
// quarters of each tile and grid.y splits the target range; partial minima are merged with one
// 64-bit atomicMin on (distance bits << 32 | index), which also implements "ties go to the lowest
// index" independently of how the work was split.
//
// Arithmetic contract (bit-exact with oracle/matching.py): d = a - b; s = d * d; acc = acc + s in
// fp32, channels in order, no FMA contraction.
#include "common.h"

namespace {

constexpr int MAX_SEG = 64;
struct SegArgs {
  int a[MAX_SEG + 1];
  int b[MAX_SEG + 1];
};

template <int C> constexpr int tile_rows() { return C <= 32 ? 256 : 8192 / C; }  // 32 KB of LDS

template <int C>
__global__ __launch_bounds__(256) void knn1_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                   SegArgs seg, int split_len, int dist_type,
                                                   unsigned long long* __restrict__ best) {
#pragma clang fp contract(off)
  constexpr int TILE = tile_rows<C>();
  __shared__ float tile[TILE * C];
  const int s = blockIdx.z;
  const int a0 = seg.a[s], na = seg.a[s + 1] - a0;
  const int b0 = seg.b[s], nb = seg.b[s + 1] - b0;
  const int q0 = blockIdx.x * 64;
  if (q0 >= na) return;
  const int t_begin = blockIdx.y * split_len;
  const int t_end = min(nb, t_begin + split_len);
  if (t_begin >= t_end) return;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = q0 + lane;
  const bool q_ok = q < na;
  float a[C];
  {
    const float4* src = reinterpret_cast<const float4*>(A + (size_t)(a0 + (q_ok ? q : 0)) * C);
#pragma unroll
    for (int i = 0; i < C / 4; ++i) {
      float4 v = src[i];
      a[4 * i] = v.x; a[4 * i + 1] = v.y; a[4 * i + 2] = v.z; a[4 * i + 3] = v.w;
    }
  }
  float best_d = __builtin_inff();
  int best_j = 0x7FFFFFFF;
  bool any = false;

  for (int t0 = t_begin; t0 < t_end; t0 += TILE) {
    const int cnt = min(TILE, t_end - t0);
    __syncthreads();
    {
      const float4* src = reinterpret_cast<const float4*>(B + (size_t)(b0 + t0) * C);
      float4* dst = reinterpret_cast<float4*>(tile);
      for (int i = threadIdx.x; i < cnt * (C / 4); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int j_lo = wave * (TILE / 4);
    const int j_hi = min(cnt, j_lo + TILE / 4);
    for (int j = j_lo; j < j_hi; ++j) {
      const float4* row = reinterpret_cast<const float4*>(tile + j * C);
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < C / 4; ++i) {
        float4 b = row[i];
        float d;
        d = a[4 * i] - b.x;     acc = acc + d * d;
        d = a[4 * i + 1] - b.y; acc = acc + d * d;
        d = a[4 * i + 2] - b.z; acc = acc + d * d;
        d = a[4 * i + 3] - b.w; acc = acc + d * d;
      }
      if (dist_type == 1) acc = sqrtf(acc + 1e-7f);
      if (acc < best_d) {
        best_d = acc;
        best_j = t0 + j;
        any = true;
      }
    }
  }
  if (q_ok && any) {
    unsigned long long packed = ((unsigned long long)__float_as_uint(best_d) << 32) | (unsigned)best_j;
    atomicMin(&best[a0 + q], packed);
  }
}

__global__ void knn1_unpack(const unsigned long long* __restrict__ best, int n, long long* __restrict__ idx,
                            float* __restrict__ dist) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long p = best[i];
  if (p == ~0ull) {  // every candidate compared false (NaN features) or the target set was empty
    if (idx) idx[i] = 0;
    if (dist) dist[i] = __builtin_nanf("");
    return;
  }
  if (idx) idx[i] = (long long)(unsigned)(p & 0xFFFFFFFFull);
  if (dist) dist[i] = __uint_as_float((unsigned)(p >> 32));
}

// dense distance matrix (lib/metrics.py:22-29) with the same arithmetic contract as knn1_kernel
__global__ void pdist_kernel(const float* __restrict__ A, int n, const float* __restrict__ B, int m, int c,
                             int dist_type, float* __restrict__ out) {
#pragma clang fp contract(off)
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)n * m) return;
  const int i = (int)(idx / m), j = (int)(idx % m);
  const float* a = A + (size_t)i * c;
  const float* b = B + (size_t)j * c;
  float acc = 0.0f;
  for (int k = 0; k < c; ++k) {
    const float d = a[k] - b[k];
    acc = acc + d * d;
  }
  if (dist_type == 1) acc = sqrtf(acc + 1e-7f);
  out[idx] = acc;
}

template <int C>
void launch_knn(const float* A, const float* B, const SegArgs& seg, int nseg, int max_na, int max_nb, int dist_type,
                unsigned long long* best, hipStream_t st) {
  constexpr int TILE = tile_rows<C>();
  int qtiles = eyoc::cdiv(max_na, 64);
  int total = qtiles * nseg;
  int max_splits = eyoc::cdiv(max_nb, TILE);
  int nsplit = 2048 / (total > 0 ? total : 1);
  if (nsplit < 1) nsplit = 1;
  if (nsplit > max_splits) nsplit = max_splits;
  int split_len = eyoc::cdiv(eyoc::cdiv(max_nb, nsplit), TILE) * TILE;
  nsplit = eyoc::cdiv(max_nb, split_len);
  dim3 grid(qtiles, nsplit, nseg);
  hipLaunchKernelGGL(knn1_kernel<C>, grid, dim3(256), 0, st, A, B, seg, split_len, dist_type, best);
}

}  // namespace

extern "C" int eyoc_knn1(eyoc_ctx* ctx, const float* A_dev, const float* B_dev, int c, const int32_t* seg_a,
                         const int32_t* seg_b, int nseg, int dist_type, int64_t* idx_dev, float* dist_dev,
                         void* stream) {
  EYOC_REQUIRE(ctx && A_dev && B_dev && seg_a && seg_b, EYOC_ERR_INVALID, "eyoc_knn1: NULL argument");
  EYOC_REQUIRE(nseg >= 1 && nseg <= MAX_SEG, EYOC_ERR_INVALID, "eyoc_knn1: nseg %d not in [1,%d]", nseg, MAX_SEG);
  EYOC_REQUIRE(dist_type == 0 || dist_type == 1, EYOC_ERR_INVALID, "eyoc_knn1: dist_type %d", dist_type);
  EYOC_REQUIRE(c == 16 || c == 32 || c == 64 || c == 128, EYOC_ERR_INVALID,
               "eyoc_knn1: feature dimension %d not supported (16/32/64/128)", c);
  hipStream_t st = (hipStream_t)stream;
  SegArgs seg;
  int max_na = 0, max_nb = 0;
  for (int s = 0; s <= nseg; ++s) { seg.a[s] = seg_a[s]; seg.b[s] = seg_b[s]; }
  for (int s = 0; s < nseg; ++s) {
    int na = seg_a[s + 1] - seg_a[s], nb = seg_b[s + 1] - seg_b[s];
    EYOC_REQUIRE(na >= 0 && nb >= 0, EYOC_ERR_INVALID, "eyoc_knn1: negative segment length");
    max_na = na > max_na ? na : max_na;
    max_nb = nb > max_nb ? nb : max_nb;
  }
  const int n_total = seg_a[nseg] - seg_a[0];
  if (n_total == 0) return EYOC_OK;
  EYOC_REQUIRE(seg_a[0] == 0, EYOC_ERR_INVALID, "eyoc_knn1: seg_a[0] must be 0");
  int rc = ctx->ensure_scratch((size_t)n_total * sizeof(unsigned long long));
  if (rc) return rc;
  unsigned long long* best = (unsigned long long*)ctx->scratch;
  EYOC_CHECK_HIP(hipMemsetAsync(best, 0xFF, (size_t)n_total * sizeof(unsigned long long), st));
  if (max_nb > 0) {
    switch (c) {
      case 16: launch_knn<16>(A_dev, B_dev, seg, nseg, max_na, max_nb, dist_type, best, st); break;
      case 32: launch_knn<32>(A_dev, B_dev, seg, nseg, max_na, max_nb, dist_type, best, st); break;
      case 64: launch_knn<64>(A_dev, B_dev, seg, nseg, max_na, max_nb, dist_type, best, st); break;
      default: launch_knn<128>(A_dev, B_dev, seg, nseg, max_na, max_nb, dist_type, best, st); break;
    }
  }
  hipLaunchKernelGGL(knn1_unpack, dim3(eyoc::cdiv(n_total, 256)), dim3(256), 0, st, best, n_total,
                     (long long*)idx_dev, dist_dev);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

extern "C" int eyoc_pdist(eyoc_ctx* ctx, const float* A_dev, int n, const float* B_dev, int m, int c, int dist_type,
                          float* out_dev, void* stream) {
  EYOC_REQUIRE(ctx && A_dev && B_dev && out_dev, EYOC_ERR_INVALID, "eyoc_pdist: NULL argument");
  EYOC_REQUIRE(n >= 0 && m >= 0 && c >= 1, EYOC_ERR_INVALID, "eyoc_pdist: bad shape %d x %d x %d", n, m, c);
  EYOC_REQUIRE(dist_type == 0 || dist_type == 1, EYOC_ERR_INVALID, "eyoc_pdist: dist_type %d", dist_type);
  const long long total = (long long)n * m;
  if (total == 0) return EYOC_OK;
  hipLaunchKernelGGL(pdist_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A_dev, n,
                     B_dev, m, c, dist_type, out_dev);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}
