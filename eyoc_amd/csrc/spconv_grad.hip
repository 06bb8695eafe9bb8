// Backward of the sparse convolution (the reference back-propagates through MinkowskiEngine at
// lib/trainer.py:1667 `loss.backward()`).
//
//   out[o] = sum_k in[nbr[k][o]] W[k]
//
// * gradient w.r.t. the input: dIn[i] = sum_k dOut[o] W[k]^T over the pairs (i -> o, k) - itself a sparse convolution,
//   over the TRANSPOSED rulebook with transposed kernels.  The transposed rulebooks already exist: a stride-1 table is
//   its own transpose under k -> K-1-k (mirrored offset), and the strided (down) and transposed (up) tables of a level
//   boundary are each other's transposes offset by offset.  So grad-input runs the forward kernels of spconv.hip /
//   spconv_wave.hip on weights packed by eyoc_spconv_pack_weights_transposed; no new device code.
// * gradient w.r.t. the kernel: dW[k] = sum over the pairs of offset k of in[i]^T dOut[o] - a [C_in x P_k] x [P_k x C_out]
//   product per offset whose two operands are both gathered.  k_grad_weight below: a wave walks a range of output rows
//   of one offset, ballots the valid ones into a small LDS list, and feeds four pairs at a time to
//   v_mfma_f32_16x16x4_f32 with A[i][k] = in[pair k][channel i], B[k][j] = dOut[pair k][channel j]; its 64 x 64 tile of
//   dW[k] stays in registers.  Waves and row blocks are reduced in a fixed order (deterministic results).
#include "spconv.h"

using namespace eyoc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GW_WAVES = 4;

// TI / TJ: 16-channel tiles of C_in / C_out handled by one wave (1, 2 or 4)
template <int TI, int TJ>
__global__ __launch_bounds__(GW_WAVES * 64) void k_grad_weight(const int32_t* __restrict__ nbr, int n_out, const float* __restrict__ X,
                                                              int ld_x, const float* __restrict__ dY, int ld_dy, int cin, int cout,
                                                              int rows_per_block, float* __restrict__ partial) {
  __shared__ int list[GW_WAVES][64][2];
  __shared__ float red[TI * TJ * 4 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, i16 = lane & 15;
  const int k = blockIdx.y;
  const int n_cj = cout / (16 * TJ);
  const int ci0 = (blockIdx.z / n_cj) * 16 * TI, co0 = (blockIdx.z % n_cj) * 16 * TJ;
  const int r_begin = blockIdx.x * rows_per_block, r_end = min(n_out, r_begin + rows_per_block);
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int t = 0; t < TI; ++t)
#pragma unroll
    for (int u = 0; u < TJ; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int base = r_begin + wave * 64; base < r_end; base += GW_WAVES * 64) {
    const int o = base + lane;
    int idx = -1;
    if (o < r_end) idx = nbr ? nbr[(size_t)k * n_out + o] : o;   // nbr == NULL: identity map (1x1 convolution)
    const unsigned long long m = __ballot(idx >= 0);
    const int P = __popcll(m);
    if (idx >= 0) {
      const int r = __popcll(m & ((1ull << lane) - 1ull));
      list[wave][r][0] = idx; list[wave][r][1] = o;
    }
    for (int p0 = 0; p0 < P; p0 += 4) {          // four pairs per MFMA step (k-slot g = pair p0 + g)
      const bool ok = p0 + g < P;
      const int pi = ok ? list[wave][p0 + g][0] : 0, po = ok ? list[wave][p0 + g][1] : 0;
      float a[TI], b[TJ];
#pragma unroll
      for (int t = 0; t < TI; ++t) a[t] = ok ? X[(size_t)pi * ld_x + ci0 + 16 * t + i16] : 0.f;
#pragma unroll
      for (int u = 0; u < TJ; ++u) b[u] = ok ? dY[(size_t)po * ld_dy + co0 + 16 * u + i16] : 0.f;
#pragma unroll
      for (int t = 0; t < TI; ++t)
#pragma unroll
        for (int u = 0; u < TJ; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[u], acc[t][u], 0, 0, 0);
    }
  }
  // waves of the block added in order (wave 0 stores, 1..3 add in turn), then one partial tile per (row block, k, tile)
  for (int w = 0; w < GW_WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < TI; ++t)
#pragma unroll
        for (int u = 0; u < TJ; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = &red[((t * TJ + u) * 4 + r) * 64 + lane];
            *d = w == 0 ? acc[t][u][r] : *d + acc[t][u][r];
          }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < TI * TJ * 4 * 64; e += GW_WAVES * 64) {
    const float s = red[e];
    const int ln = e & 63, r = (e >> 6) & 3, tu = e >> 8, t = tu / TJ, u = tu % TJ;
    const int ci = ci0 + 16 * t + 4 * (ln >> 4) + r, co = co0 + 16 * u + (ln & 15);   // D[4g + r][j]
    partial[(((size_t)blockIdx.x * gridDim.y + k) * cin + ci) * cout + co] = s;
  }
}

__global__ void k_reduce_partials(const float* __restrict__ partial, int n_blocks, size_t per_block, float* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= per_block) return;
  float s = partial[e];
  for (int b = 1; b < n_blocks; ++b) s += partial[(size_t)b * per_block + e];
  out[e] = s;
}

// row blocks of a gradient launch: 1024 rows each (4096 until round 5: a level-0 layer of two 11 k-voxel clouds was 6 x 27 workgroups of
// 16 sequential steps on 256 CUs - 147 us), fewer where 27 offsets x the channel tiles already give the chip a few thousand workgroups
int grad_weight_blocks(int n_out, int K, int cin, int cout) {
  const int ti = cin % 64 == 0 ? 4 : (cin % 32 == 0 ? 2 : 1), tj = cout % 64 == 0 ? 4 : (cout % 32 == 0 ? 2 : 1);
  const long long per = (long long)K * (cin / (16 * ti)) * (cout / (16 * tj));
  int nb = cdiv(n_out, 1024);
  if (nb > 64) nb = 64;
  while (nb > 1 && nb * per > 4096) nb >>= 1;
  return nb < 1 ? 1 : nb;
}

}  // namespace

extern "C" {

// packed[...] of eyoc_spconv_pack_weights for the convolution V with V[k] = W[mirror ? K-1-k : k]^T ([K, cout, cin]):
// running eyoc_spconv with it over the transposed rulebook gives the gradient w.r.t. the input.
int eyoc_spconv_pack_weights_transposed(const float* w, int K, int cin, int cout, int mirror, float* packed) {
  EYOC_REQUIRE(w && packed, EYOC_ERR_INVALID, "pack_weights_transposed: NULL argument");
  std::vector<float> v((size_t)K * cin * cout);
  for (int k = 0; k < K; ++k) {
    const float* src = w + (size_t)(mirror ? K - 1 - k : k) * cin * cout;
    float* dst = v.data() + (size_t)k * cin * cout;
    for (int i = 0; i < cin; ++i)
      for (int j = 0; j < cout; ++j) dst[(size_t)j * cin + i] = src[(size_t)i * cout + j];
  }
  return eyoc_spconv_pack_weights(v.data(), nullptr, K, cout, cin, packed);
}

size_t eyoc_spconv_grad_weight_workspace_bytes(int K, int n_out, int cin, int cout) {
  if (K < 1 || n_out < 0 || cin < 1 || cout < 1) return 0;
  return align_up((size_t)grad_weight_blocks(n_out, K, cin, cout) * K * cin * cout * sizeof(float)) + 256;
}

// dW[k][ci][co] = sum over o with nbr[k][o] >= 0 of in[nbr[k][o]][ci] * dout[o][co]   (plain [K, cin, cout] layout)
int eyoc_spconv_grad_weight(eyoc_ctx* ctx, const int32_t* nbr_dev, int K, int n_out, const float* in_dev, int ld_in, int cin,
                            const float* dout_dev, int ld_dout, int cout, float* dw_dev, void* ws, size_t ws_bytes, void* stream) {
  EYOC_REQUIRE(ctx && in_dev && dout_dev && dw_dev && ws, EYOC_ERR_INVALID, "eyoc_spconv_grad_weight: NULL argument");
  EYOC_REQUIRE(nbr_dev || K == 1, EYOC_ERR_INVALID, "eyoc_spconv_grad_weight: identity map needs K == 1");
  EYOC_REQUIRE(K >= 1 && n_out >= 0 && cin % 16 == 0 && cout % 16 == 0 && cin >= 16 && cout >= 16, EYOC_ERR_INVALID,
               "eyoc_spconv_grad_weight: K %d n_out %d C_in %d C_out %d (channels must be multiples of 16)", K, n_out, cin, cout);
  EYOC_REQUIRE(((uintptr_t)ws & 255) == 0 && ws_bytes >= eyoc_spconv_grad_weight_workspace_bytes(K, n_out, cin, cout),
               EYOC_ERR_WORKSPACE, "eyoc_spconv_grad_weight: workspace %zu < required %zu bytes (256-byte aligned)", ws_bytes,
               eyoc_spconv_grad_weight_workspace_bytes(K, n_out, cin, cout));
  hipStream_t st = (hipStream_t)stream;
  const int nb = grad_weight_blocks(n_out, K, cin, cout);
  const int rows_per_block = cdiv(cdiv(n_out, nb), 64) * 64;
  float* partial = (float*)ws;
  const int ti = cin % 64 == 0 ? 4 : (cin % 32 == 0 ? 2 : 1), tj = cout % 64 == 0 ? 4 : (cout % 32 == 0 ? 2 : 1);
  const dim3 grid(nb, K, (cin / (16 * ti)) * (cout / (16 * tj))), block(GW_WAVES * 64);
#define EYOC_GW(TI_, TJ_) \
  hipLaunchKernelGGL((k_grad_weight<TI_, TJ_>), grid, block, 0, st, nbr_dev, n_out, in_dev, ld_in, dout_dev, ld_dout, cin, cout, rows_per_block, partial)
  if (ti == 4 && tj == 4) EYOC_GW(4, 4);
  else if (ti == 4 && tj == 2) EYOC_GW(4, 2);
  else if (ti == 2 && tj == 4) EYOC_GW(2, 4);
  else if (ti == 2 && tj == 2) EYOC_GW(2, 2);
  else if (ti == 4) EYOC_GW(4, 1);
  else if (tj == 4) EYOC_GW(1, 4);
  else if (ti == 2) EYOC_GW(2, 1);
  else if (tj == 2) EYOC_GW(1, 2);
  else EYOC_GW(1, 1);
#undef EYOC_GW
  const size_t per_block = (size_t)K * cin * cout;
  hipLaunchKernelGGL(k_reduce_partials, dim3(cdiv((long long)per_block, 256)), dim3(256), 0, st, partial, nb, per_block, dw_dev);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

}  // extern "C"
