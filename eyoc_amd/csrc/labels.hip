// Label-generation kernels: the EYOC-specific matching filters of lib/trainer.py:993-1151
// (Lowe ratio weights on the two nearest feature neighbours, top-k by weight, spherical filter) and the
// pose-consistency filter of lib/trainer.py:1199-1218.  The nearest neighbours themselves come from
// eyoc_knn2 / eyoc_knn1 (knn.hip).
#include "common.h"

using namespace eyoc;

namespace {

// lib/trainer.py:993-1010 + :1066-1070 in the reference's own order of fp32 operations:
//   cosine = 1 - 0.5 * dist;  x = clamp(1 - cosine, min = 1e-9);  ratio = x0 / x1;  weight = 1 - ratio.
// The sort key orders floats descending (radix sort ascending on the key), ties keep the input order.
__global__ void k_lowe_weight(const float* __restrict__ d1, const float* __restrict__ d2, int n, float* __restrict__ w,
                              unsigned int* __restrict__ key, int* __restrict__ row) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float c1 = 1.0f - 0.5f * d1[i], c2 = 1.0f - 0.5f * d2[i];
  const float x1 = fmaxf(1.0f - c1, 1e-9f), x2 = fmaxf(1.0f - c2, 1e-9f);
  const float wt = 1.0f - x1 / x2;
  w[i] = wt;
  unsigned int u = __float_as_uint(wt);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending float order
  key[i] = ~u;                                        // descending
  row[i] = i;
}

__global__ void k_take_topk(const int* __restrict__ sorted_row, const float* __restrict__ w, int k, long long* __restrict__ idx_out,
                            float* __restrict__ w_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const int r = sorted_row[i];
  idx_out[i] = r;
  if (w_out) w_out[i] = w[r];
}

__device__ inline float norm3(float x, float y, float z) {
#pragma clang fp contract(off)
  return sqrtf((x * x + y * y) + z * z);
}

// mode 0 (lib/trainer.py:1107-1110): keep pair i iff |P0[i0]| > radius and |P1[i1]| > radius.
// mode 1 (lib/trainer.py:1203-1206): keep iff |R P0[i0] + t - P1[i1]| < radius, T row-major 4x4.
// mode 2 (lib/trainer.py:1118-1149, "Similarity"): d0 = |P0[i0]|, d1 = |P1[i1]|; look the pair up in the
//         distance-similarity table at [min(int(|d0 - d1| / g1), xlim - 1)][min(int(min(d0, d1) / g0), ylim - 1)] and keep it
//         iff the entry exceeds the threshold (fp64 table and compare, like the reference's float64 tensor).
// One workgroup, order-preserving compaction (m is a few thousand).
struct SimTable { const double* t; int xlim, ylim; float g0, g1; double thresh; };

__global__ __launch_bounds__(1024) void k_pair_filter(int mode, const float* __restrict__ P0, const float* __restrict__ P1,
                                                     const long long* __restrict__ i0, const long long* __restrict__ i1, int m,
                                                     const float* __restrict__ T, float radius, long long* __restrict__ out,
                                                     int* __restrict__ n_out, SimTable sim) {
#pragma clang fp contract(off)
  __shared__ int wave_cnt[16];
  __shared__ int base_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (int start = 0; start < m; start += 1024) {
    const int i = start + threadIdx.x;
    bool keep = false;
    long long a = 0, b = 0;
    if (i < m) {
      a = i0[i]; b = i1[i];
      const float* p = P0 + 3 * a;
      const float* q = P1 + 3 * b;
      if (mode == 0) {
        keep = norm3(p[0], p[1], p[2]) > radius && norm3(q[0], q[1], q[2]) > radius;
      } else if (mode == 2) {
        const float d0 = norm3(p[0], p[1], p[2]), d1 = norm3(q[0], q[1], q[2]);
        long long c0 = (long long)(fminf(d0, d1) / sim.g0), c1 = (long long)(fabsf(d0 - d1) / sim.g1);   // .long(): towards zero
        c0 = c0 < 0 ? 0 : (c0 >= sim.ylim ? sim.ylim - 1 : c0);
        c1 = c1 < 0 ? 0 : (c1 >= sim.xlim ? sim.xlim - 1 : c1);
        keep = sim.t[c1 * sim.ylim + c0] > sim.thresh;
      } else {
        const float x = ((T[0] * p[0] + T[1] * p[1]) + T[2] * p[2]) + T[3];
        const float y = ((T[4] * p[0] + T[5] * p[1]) + T[6] * p[2]) + T[7];
        const float z = ((T[8] * p[0] + T[9] * p[1]) + T[10] * p[2]) + T[11];
        keep = norm3(x - q[0], y - q[1], z - q[2]) < radius;
      }
    }
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = __popcll(mask);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (keep) {
      const int pos = off + __popcll(mask & ((1ull << lane) - 1ull));
      out[2 * (size_t)pos] = a;
      out[2 * (size_t)pos + 1] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wave_cnt[w];
      base_s += tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = base_s;
}

}  // namespace

extern "C" {

// weights of n queries from their two nearest squared feature distances, and the k best (largest weight first,
// ties in query order): idx_out int64 [k] query indices, w_out f32 [k] or NULL.  k <= n.
int eyoc_lowe_topk(eyoc_ctx* ctx, const float* d1_dev, const float* d2_dev, int n, int k, int64_t* idx_out_dev,
                   float* w_out_dev, void* stream) {
  EYOC_REQUIRE(ctx && d1_dev && d2_dev && idx_out_dev, EYOC_ERR_INVALID, "eyoc_lowe_topk: NULL argument");
  EYOC_REQUIRE(n >= 0 && k >= 0 && k <= n, EYOC_ERR_INVALID, "eyoc_lowe_topk: k %d not in [0, n = %d]", k, n);
  if (k == 0) return EYOC_OK;
  hipStream_t st = (hipStream_t)stream;
  const size_t tmp = align_up(sort_rows_tmp_bytes(n, 32));
  const size_t off_w = 0, off_k0 = align_up((size_t)n * 4), off_k1 = off_k0 + align_up((size_t)n * 4);
  const size_t off_r0 = off_k1 + align_up((size_t)n * 4), off_r1 = off_r0 + align_up((size_t)n * 4), off_tmp = off_r1 + align_up((size_t)n * 4);
  int rc = ctx->ensure_scratch(off_tmp + tmp, st);
  if (rc) return rc;
  char* sc = (char*)ctx->scratch;
  float* w = (float*)(sc + off_w);
  unsigned int* k0 = (unsigned int*)(sc + off_k0);
  unsigned int* k1 = (unsigned int*)(sc + off_k1);
  int* r0 = (int*)(sc + off_r0);
  int* r1 = (int*)(sc + off_r1);
  hipLaunchKernelGGL(k_lowe_weight, dim3(cdiv(n, 256)), dim3(256), 0, st, d1_dev, d2_dev, n, w, k0, r0);
  rc = sort_rows_by_key(sc + off_tmp, tmp, k0, k1, r0, r1, n, 32, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_take_topk, dim3(cdiv(k, 256)), dim3(256), 0, st, r1, w, k, (long long*)idx_out_dev, w_out_dev);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

int eyoc_pair_filter(eyoc_ctx* ctx, int mode, const float* P0_dev, const float* P1_dev, const int64_t* idx0_dev,
                     const int64_t* idx1_dev, int m, const float* T_dev, float radius, int64_t* pairs_out_dev, int32_t* n_out_dev,
                     void* stream) {
  EYOC_REQUIRE(ctx && P0_dev && P1_dev && idx0_dev && idx1_dev && pairs_out_dev && n_out_dev, EYOC_ERR_INVALID,
               "eyoc_pair_filter: NULL argument");
  EYOC_REQUIRE(mode == 0 || (mode == 1 && T_dev), EYOC_ERR_INVALID, "eyoc_pair_filter: mode %d (1 needs a pose)", mode);
  EYOC_REQUIRE(m >= 0, EYOC_ERR_INVALID, "eyoc_pair_filter: m %d", m);
  hipLaunchKernelGGL(k_pair_filter, dim3(1), dim3(1024), 0, (hipStream_t)stream, mode, P0_dev, P1_dev, (const long long*)idx0_dev,
                     (const long long*)idx1_dev, m, T_dev, radius, (long long*)pairs_out_dev, n_out_dev, SimTable{});
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

int eyoc_pair_filter_similarity(eyoc_ctx* ctx, const float* P0_dev, const float* P1_dev, const int64_t* idx0_dev,
                                const int64_t* idx1_dev, int m, const double* table_dev, int xlim, int ylim, float grid0,
                                float grid1, double thresh, int64_t* pairs_out_dev, int32_t* n_out_dev, void* stream) {
  EYOC_REQUIRE(ctx && P0_dev && P1_dev && idx0_dev && idx1_dev && table_dev && pairs_out_dev && n_out_dev, EYOC_ERR_INVALID,
               "eyoc_pair_filter_similarity: NULL argument");
  EYOC_REQUIRE(m >= 0 && xlim >= 1 && ylim >= 1 && grid0 > 0.f && grid1 > 0.f, EYOC_ERR_INVALID,
               "eyoc_pair_filter_similarity: m %d table %d x %d grid %g %g", m, xlim, ylim, grid0, grid1);
  SimTable sim{table_dev, xlim, ylim, grid0, grid1, thresh};
  hipLaunchKernelGGL(k_pair_filter, dim3(1), dim3(1024), 0, (hipStream_t)stream, 2, P0_dev, P1_dev, (const long long*)idx0_dev,
                     (const long long*)idx1_dev, m, (const float*)nullptr, 0.0f, (long long*)pairs_out_dev, n_out_dev, sim);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

}  // extern "C"
