// Row gather of the feature matrix (the `F[inds]` of scripts/test_kitti.py:30-35,159-160 and
// Matcher.match_pair's `src_features[:, src_sel_ind, :]`, scripts/SC2_PCR/SC2_PCR.py:291-294), optionally blended
// with a per-row descriptor and re-normalised.  The blend is what the synthetic benchmark uses to give the
// matcher signal (random-init weights carry none): out = normalise(F[sel] + beta * G).
#include "common.h"

using namespace eyoc;

namespace {

// c / 4 lanes per row (c in {4 .. 256}, a power of two >= 4): one float4 per lane, coalesced row segments
template <int L>
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ F, int ld, const long long* __restrict__ sel,
                                                     int n, const float* __restrict__ G, float beta, float* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int r = t / L, q = t % L;
  if (r >= n) return;   // L divides 64, so a row never straddles two waves and the shuffles below see whole rows
  float4 v = *reinterpret_cast<const float4*>(F + (size_t)sel[r] * ld + q * 4);
  if (G) {
    const float4 g = *reinterpret_cast<const float4*>(G + (size_t)r * (L * 4) + q * 4);
    v.x += beta * g.x; v.y += beta * g.y; v.z += beta * g.z; v.w += beta * g.w;
    float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int d = 1; d < L; d <<= 1) s += __shfl_xor(s, d, 64);
    const float nrm = sqrtf(s);
    v.x /= nrm; v.y /= nrm; v.z /= nrm; v.w /= nrm;
  }
  *reinterpret_cast<float4*>(out + (size_t)r * (L * 4) + q * 4) = v;
}

}  // namespace

extern "C" int eyoc_gather_rows(eyoc_ctx* ctx, const float* F_dev, int ld, int c, const int64_t* sel_dev, int n,
                                const float* G_dev, float beta, float* out_dev, void* stream) {
  EYOC_REQUIRE(ctx && F_dev && sel_dev && out_dev, EYOC_ERR_INVALID, "eyoc_gather_rows: NULL argument");
  EYOC_REQUIRE(n >= 0 && ld >= c && ld % 4 == 0, EYOC_ERR_INVALID, "eyoc_gather_rows: n %d ld %d c %d", n, ld, c);
  if (n == 0) return EYOC_OK;
  hipStream_t st = (hipStream_t)stream;
  const long long* sel = (const long long*)sel_dev;
#define EYOC_GATHER_CASE(C_)                                                                                          \
  case C_:                                                                                                            \
    hipLaunchKernelGGL((k_gather_rows<C_ / 4>), dim3(cdiv((long long)n * (C_ / 4), 256)), dim3(256), 0, st, F_dev, ld, \
                       sel, n, G_dev, beta, out_dev);                                                                 \
    break;
  switch (c) {
    EYOC_GATHER_CASE(4) EYOC_GATHER_CASE(8) EYOC_GATHER_CASE(16) EYOC_GATHER_CASE(32) EYOC_GATHER_CASE(64)
    EYOC_GATHER_CASE(128) EYOC_GATHER_CASE(256)
    default:
      set_error("eyoc_gather_rows: c %d not a power of two in [4,256]", c);
      return EYOC_ERR_INVALID;
  }
#undef EYOC_GATHER_CASE
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}
