// Training-mode support kernels (lib/trainer.py:1655-1676 back-propagates through the whole network):
//   * batch normalisation with BATCH statistics (MinkowskiBatchNorm = nn.BatchNorm1d over the rows of a sparse tensor,
//     model/common.py:4-6): per-channel mean / biased variance, y = (x - mean) / sqrt(var + eps) * gamma + beta (+ ReLU),
//     and its backward dx = gamma / sigma * (dy - mean(dy) - xhat * mean(dy * xhat)), dgamma = sum dy * xhat, dbeta = sum dy;
//   * the first convolution's window gather: G[row][k] = feature of the voxel at window offset k (or 0), so that the
//     C_in = 1 convolution and its weight gradient are one plain [N, K] x [K, C_out] product each.
// All of it is HBM-streaming work: coalesced row-major reads (consecutive lanes = consecutive channels), fp64 partial sums
// per workgroup written to a scratch array and added up in a FIXED order (bit-reproducible statistics), float4 element-wise passes.
#include "common.h"

using namespace eyoc;

namespace {

constexpr int BN_BLOCKS = 1024;      // partial-sum workgroups (4 per CU)

// partial[b][0..c) = sum of a, partial[b][c..2c) = sum of b over the block's rows; a / b chosen by MODE:
//   MODE 0: a = x, b = x^2                       (forward statistics)
//   MODE 1: a = dy', b = dy' * xhat              (backward sums; dy' = dy where y > 0 when a ReLU followed, xhat from mean / invstd)
template <int MODE>
__global__ __launch_bounds__(256) void k_bn_partial(const float* __restrict__ x, int ld_x, const float* __restrict__ y, int ld_y,
                                                    const float* __restrict__ dy, int ld_dy, int n, int c, const float* __restrict__ mean_var,
                                                    float eps, double* __restrict__ partial) {
  __shared__ double sa[256], sb[256];
  const int ch = threadIdx.x % c, sub = threadIdx.x / c, nsub = 256 / c;       // c divides 256
  const int rows_per_block = (n + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
  double a = 0.0, b = 0.0;
  float mean = 0.f, invstd = 0.f;
  if (MODE == 1) { mean = mean_var[ch]; invstd = 1.0f / sqrtf(mean_var[c + ch] + eps); }
  for (int r = r0 + sub; r < r1; r += nsub) {
    const float xv = x[(size_t)r * ld_x + ch];
    if (MODE == 0) {
      a += (double)xv;
      b += (double)xv * (double)xv;
    } else {
      float g = dy[(size_t)r * ld_dy + ch];
      if (y && !(y[(size_t)r * ld_y + ch] > 0.0f)) g = 0.0f;
      a += (double)g;
      b += (double)g * (double)((xv - mean) * invstd);
    }
  }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  if (sub == 0) {
    for (int s = 1; s < nsub; ++s) { a += sa[s * c + ch]; b += sb[s * c + ch]; }    // fixed order
    partial[(size_t)blockIdx.x * 2 * c + ch] = a;
    partial[(size_t)blockIdx.x * 2 * c + c + ch] = b;
  }
}

// MODE 0: out[ch] = mean, out[c + ch] = biased variance.  MODE 1: out[ch] = sum a (dbeta), out[c + ch] = sum b (dgamma).
// One workgroup per TWO channels: thread t adds the partial sums of column (t & 3) = (channel, a / b) over the blocks t >> 2,
// t >> 2 + 64, ... in ascending order, then the 64 per-thread sums meet in a fixed tree - bit-reproducible whatever the timing.
// (Round 5: one thread per channel walked all <= 1024 partial blocks alone, 49 us of dependent loads per call and 42 calls per
// training iteration - 2 of its 12 ms.)
// MODE 0 with `running` != NULL also moves the running statistics like nn.BatchNorm1d: running = (1 - m) running + m batch, the
// variance unbiased (n / (n - 1)) - one launch instead of five element-wise ones per norm.
template <int MODE>
__global__ __launch_bounds__(256) void k_bn_final(const double* __restrict__ partial, int blocks, int n, int c, float* __restrict__ out0,
                                                  float* __restrict__ out1, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                  float momentum) {
  __shared__ double red[256];
  const int col = threadIdx.x & 3, sub = threadIdx.x >> 2;
  const int ch = blockIdx.x * 2 + (col >> 1), ab = col & 1;
  double s = 0.0;
  if (ch < c)
    for (int k = sub; k < blocks; k += 64) s += partial[(size_t)k * 2 * c + ab * c + ch];
  red[threadIdx.x] = s;
  __syncthreads();
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    if (sub < w) red[threadIdx.x] += red[threadIdx.x + 4 * w];
    __syncthreads();
  }
  if (threadIdx.x < 4 && ab == 0 && ch < c) {            // threads 0 and 2: one per channel
    const double a = red[threadIdx.x], b = red[threadIdx.x + 1];
    if (MODE == 0) {
      const double m = a / n;
      double v = b / n - m * m;
      if (v < 0.0) v = 0.0;
      out0[ch] = (float)m;
      out1[ch] = (float)v;
      if (run_mean) {
        const float mf = (float)m, vf = (float)v * ((float)n / (float)(n > 1 ? n - 1 : 1));
        run_mean[ch] = run_mean[ch] * (1.0f - momentum) + momentum * mf;
        run_var[ch] = run_var[ch] * (1.0f - momentum) + momentum * vf;
      }
    } else {
      out0[ch] = (float)a;
      out1[ch] = (float)b;
    }
  }
}

__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, int ld_x, int n, int c, const float* __restrict__ mean_var,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int relu,
                                                  float* __restrict__ y, int ld_y) {
  const int c4 = c / 4;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * c4) return;
  const int r = (int)(i / c4), q = (int)(i % c4) * 4;
  const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ld_x + q);
  const float4 m = *reinterpret_cast<const float4*>(mean_var + q), va = *reinterpret_cast<const float4*>(mean_var + c + q);
  const float4 g = *reinterpret_cast<const float4*>(gamma + q), b = *reinterpret_cast<const float4*>(beta + q);
  float4 o;
  o.x = (v.x - m.x) * (1.0f / sqrtf(va.x + eps)) * g.x + b.x;
  o.y = (v.y - m.y) * (1.0f / sqrtf(va.y + eps)) * g.y + b.y;
  o.z = (v.z - m.z) * (1.0f / sqrtf(va.z + eps)) * g.z + b.z;
  o.w = (v.w - m.w) * (1.0f / sqrtf(va.w + eps)) * g.w + b.w;
  if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
  *reinterpret_cast<float4*>(y + (size_t)r * ld_y + q) = o;
}

// dx = gamma * invstd * (dy' - sum_dy / n - xhat * sum_dy_xhat / n)
__global__ __launch_bounds__(256) void k_bn_backward_apply(const float* __restrict__ x, int ld_x, const float* __restrict__ y, int ld_y,
                                                           const float* __restrict__ dy, int ld_dy, int n, int c,
                                                           const float* __restrict__ mean_var, const float* __restrict__ gamma, float eps,
                                                           const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                           float* __restrict__ dx, int ld_dx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * c) return;
  const int r = (int)(i / c), ch = (int)(i % c);
  const float invstd = 1.0f / sqrtf(mean_var[c + ch] + eps);
  const float xhat = (x[(size_t)r * ld_x + ch] - mean_var[ch]) * invstd;
  float g = dy[(size_t)r * ld_dy + ch];
  if (y && !(y[(size_t)r * ld_y + ch] > 0.0f)) g = 0.0f;
  const float inv_n = 1.0f / (float)n;
  dx[(size_t)r * ld_dx + ch] = gamma[ch] * invstd * (g - dbeta[ch] * inv_n - xhat * dgamma[ch] * inv_n);
}

// G[row][k * cin + ci] = feats[neighbour(row, k)][ci] or 0; window offsets enumerate x fastest (like every rulebook)
__global__ __launch_bounds__(256) void k_gather_window(const int32_t* __restrict__ coords, int n, HashTable table, int ks, const float* __restrict__ feats,
                                                       int cin, float* __restrict__ out) {
  const int K = ks * ks * ks, r = ks / 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * K) return;
  const int row = (int)(i / K), k = (int)(i % K);
  const int4 c = reinterpret_cast<const int4*>(coords)[row];
  const int dx = k % ks - r, dy = (k / ks) % ks - r, dz = k / (ks * ks) - r;
  const int idx = hash_lookup(table, pack_key(c.x, c.y + dx, c.z + dy, c.w + dz));
  float* dst = out + (size_t)i * cin;
  for (int ci = 0; ci < cin; ++ci) dst[ci] = idx >= 0 ? feats[(size_t)idx * cin + ci] : 0.0f;
}

bool bn_shape_ok(int n, int c) { return n >= 1 && c >= 4 && c <= 256 && 256 % c == 0; }
int bn_blocks(int n) { return n < BN_BLOCKS * 64 ? (n + 63) / 64 : BN_BLOCKS; }

}  // namespace

extern "C" {

size_t eyoc_bn_workspace_bytes(int n, int c) { return bn_shape_ok(n, c) ? align_up((size_t)bn_blocks(n) * 2 * c * sizeof(double)) : 0; }

static int bn_train_forward(eyoc_ctx* ctx, const float* x_dev, int n, int c, int ld_x, const float* gamma_dev, const float* beta_dev, float eps,
                            int relu, float* y_dev, int ld_y, float* mean_var_dev, float* run_mean_dev, float* run_var_dev, float momentum,
                            void* ws_dev, size_t ws_bytes, void* stream) {
  EYOC_REQUIRE(ctx && x_dev && gamma_dev && beta_dev && y_dev && mean_var_dev && ws_dev, EYOC_ERR_INVALID, "eyoc_bn_train_forward: NULL argument");
  EYOC_REQUIRE(bn_shape_ok(n, c) && ld_x % 4 == 0 && ld_y % 4 == 0 && ld_x >= c && ld_y >= c, EYOC_ERR_INVALID,
               "eyoc_bn_train_forward: n %d, c %d (a divisor of 256, >= 4), leading dimensions %d / %d (multiples of 4)", n, c, ld_x, ld_y);
  EYOC_REQUIRE(ws_bytes >= eyoc_bn_workspace_bytes(n, c), EYOC_ERR_WORKSPACE, "eyoc_bn_train_forward: workspace %zu < %zu bytes", ws_bytes,
               eyoc_bn_workspace_bytes(n, c));
  hipStream_t st = (hipStream_t)stream;
  const int nb = bn_blocks(n);
  hipLaunchKernelGGL(k_bn_partial<0>, dim3(nb), dim3(256), 0, st, x_dev, ld_x, (const float*)nullptr, 0, (const float*)nullptr, 0, n, c,
                     (const float*)nullptr, eps, (double*)ws_dev);
  hipLaunchKernelGGL(k_bn_final<0>, dim3(cdiv(c, 2)), dim3(256), 0, st, (const double*)ws_dev, nb, n, c, mean_var_dev, mean_var_dev + c,
                     run_mean_dev, run_var_dev, momentum);
  hipLaunchKernelGGL(k_bn_apply, dim3(cdiv((long long)n * (c / 4), 256)), dim3(256), 0, st, x_dev, ld_x, n, c, mean_var_dev, gamma_dev, beta_dev, eps,
                     relu, y_dev, ld_y);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

int eyoc_bn_train_forward(eyoc_ctx* ctx, const float* x_dev, int n, int c, int ld_x, const float* gamma_dev, const float* beta_dev, float eps,
                          int relu, float* y_dev, int ld_y, float* mean_var_dev, void* ws_dev, size_t ws_bytes, void* stream) {
  return bn_train_forward(ctx, x_dev, n, c, ld_x, gamma_dev, beta_dev, eps, relu, y_dev, ld_y, mean_var_dev, nullptr, nullptr, 0.0f, ws_dev, ws_bytes,
                          stream);
}

int eyoc_bn_train_forward_running(eyoc_ctx* ctx, const float* x_dev, int n, int c, int ld_x, const float* gamma_dev, const float* beta_dev,
                                  float eps, int relu, float* y_dev, int ld_y, float* mean_var_dev, float* running_mean_dev,
                                  float* running_var_dev, float momentum, void* ws_dev, size_t ws_bytes, void* stream) {
  EYOC_REQUIRE(running_mean_dev && running_var_dev, EYOC_ERR_INVALID, "eyoc_bn_train_forward_running: NULL running statistics");
  return bn_train_forward(ctx, x_dev, n, c, ld_x, gamma_dev, beta_dev, eps, relu, y_dev, ld_y, mean_var_dev, running_mean_dev, running_var_dev,
                          momentum, ws_dev, ws_bytes, stream);
}

int eyoc_bn_train_backward(eyoc_ctx* ctx, const float* x_dev, int ld_x, const float* y_dev, int ld_y, const float* dy_dev, int ld_dy, int n, int c,
                           const float* gamma_dev, const float* mean_var_dev, float eps, float* dx_dev, int ld_dx, float* dgamma_dev,
                           float* dbeta_dev, void* ws_dev, size_t ws_bytes, void* stream) {
  EYOC_REQUIRE(ctx && x_dev && dy_dev && gamma_dev && mean_var_dev && dx_dev && dgamma_dev && dbeta_dev && ws_dev, EYOC_ERR_INVALID,
               "eyoc_bn_train_backward: NULL argument");
  EYOC_REQUIRE(bn_shape_ok(n, c) && ld_x >= c && ld_dy >= c && ld_dx >= c && (!y_dev || ld_y >= c), EYOC_ERR_INVALID,
               "eyoc_bn_train_backward: n %d, c %d (a divisor of 256, >= 4)", n, c);
  EYOC_REQUIRE(ws_bytes >= eyoc_bn_workspace_bytes(n, c), EYOC_ERR_WORKSPACE, "eyoc_bn_train_backward: workspace %zu < %zu bytes", ws_bytes,
               eyoc_bn_workspace_bytes(n, c));
  hipStream_t st = (hipStream_t)stream;
  const int nb = bn_blocks(n);
  hipLaunchKernelGGL(k_bn_partial<1>, dim3(nb), dim3(256), 0, st, x_dev, ld_x, y_dev, ld_y, dy_dev, ld_dy, n, c, mean_var_dev, eps, (double*)ws_dev);
  hipLaunchKernelGGL(k_bn_final<1>, dim3(cdiv(c, 2)), dim3(256), 0, st, (const double*)ws_dev, nb, n, c, dbeta_dev, dgamma_dev, (float*)nullptr,
                     (float*)nullptr, 0.0f);
  hipLaunchKernelGGL(k_bn_backward_apply, dim3(cdiv((long long)n * c, 256)), dim3(256), 0, st, x_dev, ld_x, y_dev, ld_y, dy_dev, ld_dy, n, c,
                     mean_var_dev, gamma_dev, eps, dbeta_dev, dgamma_dev, dx_dev, ld_dx);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

static int gather_window_rows(eyoc_ctx* ctx, eyoc_maps* maps, int ks, const float* feats_dev, int cin, float* out_dev, void* stream, const char* who) {
  EYOC_REQUIRE(ctx && maps && feats_dev && out_dev, EYOC_ERR_INVALID, "%s: NULL argument", who);
  EYOC_REQUIRE((ks == 1 || ks == 3 || ks == 5 || ks == 7) && cin >= 1 && cin <= 64, EYOC_ERR_INVALID, "%s: ks %d, cin %d", who, ks, cin);
  hipStream_t st = (hipStream_t)stream;
  int rc = maps_build_table0(maps, st);
  if (rc) return rc;
  const int n = maps->rows[0];
  if (n == 0) return EYOC_OK;
  const long long total = (long long)n * ks * ks * ks;
  hipLaunchKernelGGL(k_gather_window, dim3(cdiv(total, 256)), dim3(256), 0, st, maps->coords[0], n, maps->table[0], ks, feats_dev, cin, out_dev);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

// feats and the result in the CALLER's rows: maps whose internal order differs (eyoc_maps_row_order != NULL) are refused - a caller that
// holds Z-ordered maps and caller-ordered features would otherwise get silently permuted rows
int eyoc_maps_gather_window(eyoc_ctx* ctx, eyoc_maps* maps, int ks, const float* feats_dev, int cin, float* out_dev, void* stream) {
  EYOC_REQUIRE(!maps || !maps->row_perm, EYOC_ERR_INVALID,
               "eyoc_maps_gather_window: the maps are in Z-order internally - permute the features with eyoc_maps_row_order and call "
               "eyoc_maps_gather_window_internal, or build the maps in the caller's order");
  return gather_window_rows(ctx, maps, ks, feats_dev, cin, out_dev, stream, "eyoc_maps_gather_window");
}

// feats and the result in the maps' INTERNAL rows (row i = the caller's row eyoc_maps_row_order()[i]; the caller's rows when that is NULL)
int eyoc_maps_gather_window_internal(eyoc_ctx* ctx, eyoc_maps* maps, int ks, const float* feats_dev, int cin, float* out_dev, void* stream) {
  return gather_window_rows(ctx, maps, ks, feats_dev, cin, out_dev, stream, "eyoc_maps_gather_window_internal");
}

}  // extern "C"
