// The network's 1x1 tail in ONE kernel (gfx950, SPLIT16 arithmetic): model/resunet.py:183-191
//
//     out = conv1_tr(cat1)        1x1, 96 -> 64, no norm          \
//     out = relu(out)                                              |   [N, 96] SPLIT16 rows in, [N, 32] fp32 rows out,
//     out = final(out)            1x1, 64 -> 32, bias              |   nothing in between touches memory
//     out = out / |out|_2         (normalize_feature)             /
//
// As two launches the [N, 64] intermediate went out to HBM and came back: ~2 GB of the 3.9 GB the two layers moved on the
// 3.8 M-row bench batch.  Here a wave takes 16 rows at a time: 36 fp16 MFMAs (4 channel tiles x 3 input blocks x the three
// split16 terms) give the 64 intermediate channels in registers; bias / ReLU / hi-lo split happen there; 12 more MFMAs give
// the 32 output channels; the row norm is two cross-lane adds.
//
// The trick that avoids any data movement between the two products: the K index of an MFMA is just a summation index, so
// its order is free as long as both operands agree.  After the first product lane (g, j) holds channels 16 t + 4 g + r
// (t = 0..3, r = 0..3) of row j.  The second product declares exactly those, in that order, to be the 16 K-elements lane
// group g supplies (two K blocks of 8: t = 0,1 and t = 2,3), and loads the matching rows of `final`'s packed weights for
// its A operand - two 8-byte pieces per fragment out of the standard packing (eyoc_spconv_pack_weights_split16), read once
// per wave.  All weights of both layers stay in registers (128 VGPRs, one wave per SIMD); waves are persistent and stream rows
// four chunks ahead: the kernel is bound by HBM (384 B in, 128 B out per row).
#include <algorithm>

#include "spconv.h"

using namespace eyoc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TailArgs {
  const float* in;     // [n, ld_in] SPLIT16 rows; columns [0, 96)
  int ld_in, n;
  const float* w1;     // conv1_tr, split16 packing (K = 1, 96 -> 64)
  const float* s1;     // its out_scale scalar
  const float* b1;     // its shift [64]
  const float* w2;     // final, split16 packing (K = 1, 64 -> 32)
  const float* s2;
  const float* b2;     // [32]
  float* out;          // [n, ld_out] fp32
  int ld_out;
  const int32_t* out_perm;   // output row of internal row o (NULL: o)
  unsigned int* range;       // SPLIT16 range guard words
  int relu1, l2norm;
};

constexpr int C1 = 96, CM = 64, C2 = 32;

__global__ __launch_bounds__(256, 1) void tail_fused_kernel(TailArgs a) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, j = lane & 15;
  const int n_chunks = (a.n + 15) >> 4;
  const int wave_id = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)), n_waves = (int)(gridDim.x * (blockDim.x >> 6));

  // ---- weights, once.  conv1_tr: CT = 64 (NT = 4 tiles), CC = 32 (JQ = 2, ncc = 3): fragment ((cc * 4 + nt) * 2 + p), 1 KB each.
  half8_t W1[4][3][2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        W1[nt][cc][p] = *reinterpret_cast<const half8_t*>(reinterpret_cast<const char*>(a.w1) + ((cc * 4 + nt) * 2 + p) * 1024 + lane * 16);
  // final: CT = 32 (NT = 2), CC = 64 (JQ = 4, ncc = 1): fragment (nt * 4 + 2 kb + p).  K block kb, element e of lane group g is
  // intermediate channel 32 kb + 16 (e >> 2) + 4 g + (e & 3): in the standard packing that is lane group 2 (e >> 2) + (g >> 1),
  // elements 4 (g & 1) + (e & 3) - the two 8-byte halves below.
  half8_t W2[2][2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const char* f = reinterpret_cast<const char*>(a.w2) + (nt * 4 + 2 * kb + p) * 1024 + (((g >> 1) * 16 + j) * 16) + 8 * (g & 1);
        const uint2 lo = *reinterpret_cast<const uint2*>(f), hi = *reinterpret_cast<const uint2*>(f + 512);
        W2[nt][kb][p] = __builtin_bit_cast(half8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
      }
  const float os1 = *a.s1, os2 = *a.s2;
  float4 b1[4], b2[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) b1[t] = *reinterpret_cast<const float4*>(a.b1 + 16 * t + 4 * g);
#pragma unroll
  for (int t = 0; t < 2; ++t) b2[t] = *reinterpret_cast<const float4*>(a.b2 + 16 * t + 4 * g);
  const bool poisoned = split16_poisoned(a.range);
  const float qnan = __builtin_nanf("");
  float mx = 0.f;

  // ---- rows: lane (g, j) supplies, per 32-channel input block, the 16 bytes of hi halves (and of lo halves) of channels
  // 8 g .. 8 g + 7 of row 16 chunk + j - 64 contiguous bytes per row and instruction; the next chunk's loads fly during this one
  auto load_x = [&](int chunk, half8_t (&X)[3][2]) {
    int o = chunk * 16 + j;
    if (o >= a.n) o = a.n - 1;                                         // ragged tail: a valid row, never stored
    const char* row = reinterpret_cast<const char*>(a.in + (size_t)o * a.ld_in) + g * 16;
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      X[cc][0] = *reinterpret_cast<const half8_t*>(row + cc * 128);
      X[cc][1] = *reinterpret_cast<const half8_t*>(row + cc * 128 + SPLIT16_LO);
    }
  };
  auto compute = [&](int chunk, const half8_t (&X)[3][2]) {
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W1[t][cc][term == 2 ? 1 : 0], X[cc][term == 1 ? 1 : 0], acc[t], 0, 0, 0);
    // bias, ReLU, hi / lo split: the lane's 16 values ARE its K elements of the second product (kb = t >> 1, e = 4 (t & 1) + r)
    half8_t Y[2][2];
    float cmx = 0.f;                                                   // largest |intermediate| of this lane's quarter of row j
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 v = make_float4(acc[t][0] * os1 + b1[t].x, acc[t][1] * os1 + b1[t].y, acc[t][2] * os1 + b1[t].z, acc[t][3] * os1 + b1[t].w);
      if (a.relu1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      split16_track(cmx, v);
      uint2 h, l;
      split16_encode4(v, h, l);
      const half4_t h4 = __builtin_bit_cast(half4_t, h), l4 = __builtin_bit_cast(half4_t, l);
#pragma unroll
      for (int r = 0; r < 4; ++r) { Y[t >> 1][0][4 * (t & 1) + r] = h4[r]; Y[t >> 1][1][4 * (t & 1) + r] = l4[r]; }
    }
    f32x4 o2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) o2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          o2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W2[t][kb][term == 2 ? 1 : 0], Y[kb][term == 1 ? 1 : 0], o2[t], 0, 0, 0);
    float4 v[2];
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      v[t] = make_float4(o2[t][0] * os2 + b2[t].x, o2[t][1] * os2 + b2[t].y, o2[t][2] * os2 + b2[t].z, o2[t][3] * os2 + b2[t].w);
      ss += v[t].x * v[t].x + v[t].y * v[t].y + v[t].z * v[t].z + v[t].w * v[t].w;
    }
    if (a.l2norm) {                                                    // the row's 32 channels: this lane and lanes j + 16, 32, 48
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      const float inv = 1.0f / sqrtf(ss);                              // no epsilon: a zero row gives NaN (0 * inf), like the reference
#pragma unroll
      for (int t = 0; t < 2; ++t) { v[t].x *= inv; v[t].y *= inv; v[t].z *= inv; v[t].w *= inv; }
    }
    // range guard: an overflow upstream poisons every row of this forward; one in THIS kernel's intermediate (it never leaves
    // the registers, so no later layer could answer for it) poisons the row it happened in
    mx = split16_merge(mx, cmx);
    cmx = split16_merge(cmx, __shfl_xor(cmx, 16, 64));
    cmx = split16_merge(cmx, __shfl_xor(cmx, 32, 64));
    const bool bad = poisoned || split16_over(cmx);
    const int o = chunk * 16 + j;
    if (o < a.n) {
      const size_t oo = a.out_perm ? (size_t)a.out_perm[o] : (size_t)o;
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<float4*>(a.out + oo * a.ld_out + 16 * t + 4 * g) = bad ? make_float4(qnan, qnan, qnan, qnan) : v[t];
    }
  };

  // one wave per SIMD (the weights of both layers take 128 of its registers): four chunks of rows in flight per wave keep
  // ~100 KB of loads outstanding per CU
  constexpr int DEPTH = 4;
  half8_t X[DEPTH][3][2];
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
    if (wave_id + d * n_waves < n_chunks) load_x(wave_id + d * n_waves, X[d]);
  for (int base = wave_id; base < n_chunks; base += DEPTH * n_waves) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int chunk = base + d * n_waves;
      if (chunk >= n_chunks) break;
      const int ahead = chunk + (DEPTH - 1) * n_waves;
      if (ahead < n_chunks) load_x(ahead, X[(d + DEPTH - 1) % DEPTH]);
      compute(chunk, X[d]);
    }
  }
  split16_report(a.range, mx);
}

}  // namespace

namespace eyoc {

// conv1_tr (1x1, 96 -> 64, optional ReLU) + final (1x1, 64 -> 32, bias folded into b2) + optional row normalisation; every
// pointer as the unfused layers would get it (split16 packings of the blob, their out_scale scalars and shifts)
int launch_tail_fused(const float* in, int ld_in, int n, const float* w1, const float* s1, const float* b1, int relu1, const float* w2,
                      const float* s2, const float* b2, int l2norm, float* out, int ld_out, const int32_t* out_perm, unsigned int* range,
                      hipStream_t st) {
  EYOC_REQUIRE(in && w1 && s1 && b1 && w2 && s2 && b2 && out && ld_in % 32 == 0 && ld_in >= C1 && ld_out >= C2, EYOC_ERR_INVALID,
               "tail_fused: bad argument");
  if (n <= 0) return EYOC_OK;
  TailArgs a{in, ld_in, n, w1, s1, b1, w2, s2, b2, out, ld_out, out_perm, range, relu1, l2norm};
  const int n_chunks = cdiv(n, 16);
  const int blocks = std::min(cdiv(n_chunks, 4), 256);                  // persistent: one 4-wave workgroup per CU, one wave per SIMD
  hipLaunchKernelGGL(tail_fused_kernel, dim3(blocks), dim3(256), 0, st, a);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

bool tail_fusable(int cin1, int cmid, int cout) { return cin1 == C1 && cmid == CM && cout == C2; }

}  // namespace eyoc
