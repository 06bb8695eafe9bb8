// SC2-PCR registration on the GPU: Matcher.SC2_PCR of the reference
// (scripts/SC2_PCR/SC2_PCR.py:307-384 with pick_seeds :33-59, cal_leading_eigenvector :170-196,
// cal_seed_trans :61-168, post_refinement :238-278 and rigid_transform_3d, common.py:7-45), bs == 1.
//
// The reference materialises ~8 dense N x N fp32 tensors (256 MB each at N = 8000) and a
// [0.2N, N] x [N, N] GEMM.  Here nothing N x N is ever stored in fp32:
//   * the first-order compatibility SC[i][j] = max(0, 1 - (|s_i-s_j| - |t_i-t_j|)^2 / d^2) is
//     recomputed from the 96 KB of coordinates inside every power-iteration sweep;
//   * the two hard masks (< d, < d/2) are kept as bit matrices (N x N/64 words, 8 MB at N = 8000);
//     the second-order measure of a seed row is popcount(tight[seed] & tight[j]) * hard[seed][j] -
//     exact, since the reference's GEMM operands are 0/1;
//   * per seed: top-k1 by that count (histogram threshold + ordered pick, ties to the lower index),
//     then one wave does the whole local stage (30 x 30 and 20 x 20 problems, power iteration,
//     weighted Kabsch in fp64) and counts the inliers of its hypothesis;
//   * the 20-round refinement is one workgroup looping on the device.
#include "pose_math.h"

using namespace eyoc;

namespace {

constexpr int MAX_N = 16384;
constexpr int K1_MAX = 32, K2_MAX = 32;

struct Sc2Ctl {          // device-side control block
  int converged;         // power iteration reached allclose
  int iters;             // sweeps actually applied
  int best_seed;
  float best_fitness;
  float norm;            // ||M v|| of the current sweep
  int dense;             // the compatibility graph has more edges than the CSR arrays hold: dense sweeps instead
  int pad[2];
};

// Squared length with the roundings PINNED (round 6): one rounded product, two fused multiply-adds.  Written as `dx * dx + dy * dy +
// dz * dz` the compiler chose per call site - fused chains in one kernel, a packed multiply and one fma in another - so two kernels
// could disagree in the last bit about the same pair (a hard-mask bit whose CSR value then came out 0), and the A/B forms of a kernel
// (eyoc_sc2pcr_select_kernels) could not be compared bit for bit.  The reference's torch kernels promise no particular order either.
__device__ inline float sq_len(float dx, float dy, float dz) { return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)); }
__device__ inline float cross_len(float sx, float sy, float sz, float tx, float ty, float tz, float sjx, float sjy,
                                  float sjz, float tjx, float tjy, float tjz) {
  return fabsf(sqrtf(sq_len(sx - sjx, sy - sjy, sz - sjz)) - sqrtf(sq_len(tx - tjx, ty - tjy, tz - tjz)));
}
// first-order compatibility of a cross length (SC2_PCR.py:317-319), roundings pinned likewise
__device__ inline float sc_value(float c, float inv_d2) { return fmaxf(__builtin_fmaf(-(c * c), inv_d2, 1.0f), 0.0f); }

// ---- round 6: the square roots of the O(N^2) sweeps.  sqrtf is correctly rounded (clang's default for HIP: v_sqrt_f32 plus a
// +-1 ulp fix-up from two fma residuals, ~14 VALU slots), hence monotone:  { x >= 0 : sqrtf(x) < R } = { x < T }  for
// T = min { x : sqrtf(x) >= R }.  T is found from fl(R R) by stepping ulps (a handful of sqrtf per thread, once per kernel); the
// sweeps then compare the squared length with T - the same decisions, bit for bit, without a square root per pair.
// R <= 0: never true (T = 0); R = inf: true for every finite x (T = inf); R NaN: never true (T = NaN).
__device__ inline float sqrt_lt_threshold(float R) {
  if (!(R > 0.0f)) return R != R ? R : 0.0f;
  if (R > 1.9e19f) return __builtin_inff();                    // sqrtf of every finite x is below 1.85e19 < R: x < inf says the same (x = inf: false both ways)
  float t = R * R;                                             // 0 for R < 1e-23, inf for R > 1.84e19: the loops walk on from there (inf - 1 ulp = FLT_MAX, FLT_MAX + 1 ulp = inf)
  while (t > 0.0f && sqrtf(t) >= R) t = __uint_as_float(__float_as_uint(t) - 1u);
  while (sqrtf(t) < R) t = __uint_as_float(__float_as_uint(t) + 1u);
  return t;
}
// The cross length |sqrt(a) - sqrt(b)| against a threshold: v_sqrt_f32 alone (1 ulp) decides all but the residuals within `band` of
// it.  With sa', sb' within 2^-23 of the rounded roots and one rounding in each difference, |c' - c| <= 1.5 * 2^-22 max(sa, sb); the
// band is 2^-20 max(sa', sb') + 1e-18 (v_sqrt_f32 flushes denormal inputs: roots below 1.1e-19 read 0).  A wave with an undecided
// lane takes the exact expression for all of them (about one 64 x 64 tile row in 10^4).
struct CrossFast { float c, band; };
__device__ inline CrossFast cross_len_fast(float sx, float sy, float sz, float tx, float ty, float tz, float sjx, float sjy,
                                           float sjz, float tjx, float tjy, float tjz) {
  const float sa = __builtin_amdgcn_sqrtf(sq_len(sx - sjx, sy - sjy, sz - sjz)), sb = __builtin_amdgcn_sqrtf(sq_len(tx - tjx, ty - tjy, tz - tjz));
  CrossFast r;
  r.c = fabsf(sa - sb);
  r.band = fmaf(fmaxf(sa, sb), 0x1p-20f, 1e-18f);
  return r;
}

// ---- v_new = y / (||y|| + 1e-6); converged = allclose(v_new, v_old); one workgroup
__device__ __forceinline__ void d_sc_normalize(const float* __restrict__ y, float* __restrict__ v, int n, Sc2Ctl* __restrict__ ctl) {
  if (ctl->converged) return;
  __shared__ int bad[16];
  __shared__ double wsq[16];
  double sq = 0;   // ||y||^2 in fp64, fixed order (thread-strided partial sums, wave trees, waves ascending)
  for (int i = threadIdx.x; i < n; i += 1024) sq += (double)y[i] * (double)y[i];
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) wsq[threadIdx.x >> 6] = sq;
  __syncthreads();
  double tot = 0;
  for (int w = 0; w < 16; ++w) tot += wsq[w];
  const float nrm = (float)sqrt(tot) + 1e-6f;
  int nb = 0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float nv = y[i] / nrm, ov = v[i];
    if (!(fabsf(nv - ov) <= 1e-8f + 1e-5f * fabsf(ov))) nb = 1;   // torch.allclose defaults
    v[i] = nv;
  }
  nb = __any(nb);
  if ((threadIdx.x & 63) == 0) bad[threadIdx.x >> 6] = nb;
  __syncthreads();
  if (threadIdx.x == 0) {
    int any_bad = 0;
    for (int w = 0; w < 16; ++w) any_bad |= bad[w];
    ctl->iters += 1;
    ctl->norm = nrm;
    if (!any_bad) ctl->converged = 1;
  }
}

// ---- bit matrices: hard[i][w] bit b = cross(i, 64 w + b) < d ; tight = < d/2.   One wave per (row, 64 columns).
__device__ __forceinline__ void d_masks(const float* __restrict__ src, const float* __restrict__ tgt, int n,
                                               int words, float d, unsigned long long* __restrict__ hard,
                                               unsigned long long* __restrict__ tight, int legacy) {
  // one wave per 64 x 64 tile: the lane's column stays in registers, the 64 rows come through LDS broadcasts, and
  // lane r keeps the two ballots of row r, so a tile costs 12 loads per lane instead of 12 per cross length
  __shared__ float rows[4][64 * 6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long tile = (long long)blockIdx.x * 4 + wave;
  const long long n_tiles = (long long)words * words;
  if (tile >= n_tiles) return;   // wave-uniform; no barrier below
  const int ti = (int)(tile / words), w = (int)(tile % words);
  // Round 5: the cross length is symmetric, so only the tiles on and above the diagonal are computed (each cross length costs ~28
  // VALU issue slots, two square roots among them); the wave of tile (ti, w) also writes tile (w, ti) - the 64 x 64 bit transpose
  // of what its lanes hold, six exchange steps per matrix
  if (ti > w) return;
  const int j = w * 64 + lane, i_mine = ti * 64 + lane;
  float* R = rows[wave];
  {
    const int ii = i_mine < n ? i_mine : 0;
    R[lane * 6 + 0] = src[3 * ii]; R[lane * 6 + 1] = src[3 * ii + 1]; R[lane * 6 + 2] = src[3 * ii + 2];
    R[lane * 6 + 3] = tgt[3 * ii]; R[lane * 6 + 4] = tgt[3 * ii + 1]; R[lane * 6 + 5] = tgt[3 * ii + 2];
  }
  const int jj = j < n ? j : 0;
  const float sjx = src[3 * jj], sjy = src[3 * jj + 1], sjz = src[3 * jj + 2];
  const float tjx = tgt[3 * jj], tjy = tgt[3 * jj + 1], tjz = tgt[3 * jj + 2];
  unsigned long long my_h = 0, my_t = 0;
  const int rows_here = min(64, n - ti * 64);
  const float dh = 0.5f * d;
  const bool fast_ok = !legacy && d > 0.0f && d < 1e30f;           // (a NaN / huge / non-positive d: the plain expression)
  for (int r = 0; r < rows_here; ++r) {
    unsigned long long hm, tm;
    bool exact = !fast_ok;
    if (fast_ok) {
      const CrossFast f = cross_len_fast(R[r * 6], R[r * 6 + 1], R[r * 6 + 2], R[r * 6 + 3], R[r * 6 + 4], R[r * 6 + 5], sjx, sjy, sjz,
                                         tjx, tjy, tjz);
      const float e1 = f.c - d, e2 = f.c - dh;                     // NaN / inf coordinates: every comparison below is false -> undecided
      const bool in1 = e1 < -f.band, in2 = e2 < -f.band;
      const bool und = !(in1 || e1 > f.band) || !(in2 || e2 > f.band);
      hm = __ballot(j < n && in1);
      tm = __ballot(j < n && in2);
      exact = __any(und);
    }
    if (exact) {                                                   // wave-uniform
      const float c = cross_len(R[r * 6], R[r * 6 + 1], R[r * 6 + 2], R[r * 6 + 3], R[r * 6 + 4], R[r * 6 + 5], sjx, sjy, sjz, tjx,
                                tjy, tjz);
      hm = __ballot(j < n && c < d);
      tm = __ballot(j < n && c < dh);
    }
    if (lane == r) { my_h = hm; my_t = tm; }
  }
  if (lane < rows_here) {
    hard[(size_t)i_mine * words + w] = my_h;
    tight[(size_t)i_mine * words + w] = my_t;
  }
  if (ti == w) return;
  // transpose: lane l holds row l (bit c = column c); afterwards lane c holds column c (bit l = row l).  Step s exchanges, between
  // lanes l and l ^ s, the off-diagonal s x s blocks of every 2s x 2s block
  auto step = [&](unsigned long long x, int sft, unsigned long long lo_mask) {
    const unsigned int ylo = (unsigned int)__shfl_xor((int)(unsigned int)x, sft, 64);
    const unsigned int yhi = (unsigned int)__shfl_xor((int)(unsigned int)(x >> 32), sft, 64);
    const unsigned long long y = ((unsigned long long)yhi << 32) | ylo;
    // a lane with bit `sft` clear keeps its low-position blocks and takes the partner's low-position blocks into its high positions
    return (lane & sft) ? ((x & ~lo_mask) | ((y & ~lo_mask) >> sft)) : ((x & lo_mask) | ((y & lo_mask) << sft));
  };
  auto transpose64 = [&](unsigned long long x) {
    x = step(x, 32, 0x00000000FFFFFFFFull);
    x = step(x, 16, 0x0000FFFF0000FFFFull);
    x = step(x, 8, 0x00FF00FF00FF00FFull);
    x = step(x, 4, 0x0F0F0F0F0F0F0F0Full);
    x = step(x, 2, 0x3333333333333333ull);
    return step(x, 1, 0x5555555555555555ull);
  };
  const unsigned long long th = transpose64(my_h), tt = transpose64(my_t);
  if (j < n) {                                   // row j = w * 64 + lane of the mirrored tile; its columns ti * 64 ..
    hard[(size_t)j * words + ti] = th;
    tight[(size_t)j * words + ti] = tt;
  }
}

// ---- CSR form of the two compatibility graphs.  The hard graph (c < d) is exactly the support of the first-order
// matrix SC = max(0, 1 - c^2/d^2): a few percent of the N^2 entries on real correspondences, so the 20 power sweeps
// run as sparse mat-vecs over (column, value) lists instead of recomputing N^2 cross lengths each (k_sc_matvec was
// 45 % of SC2-PCR).  The tight graph's column lists serve the second-order counts of k_seed_topk.
// One workgroup per pair: exclusive scan of the row lengths; more edges than the arrays hold -> ctl->dense.
// row lengths: one wave per row pops the row's words (coalesced)
__device__ __forceinline__ void d_csr_count(const unsigned long long* __restrict__ hard, int n, int words, int* __restrict__ cnt_h) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  int c = 0;
  for (int w = lane; w < words; w += 64) c += __popcll(hard[(size_t)i * words + w]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
  if (lane == 0) cnt_h[i] = c;
}
__device__ __forceinline__ void d_csr_scan(int* __restrict__ cnt_h, int n, long long cap, Sc2Ctl* __restrict__ ctl) {
  __shared__ long long tot[16];
  __shared__ long long carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int i0 = 0; i0 <= n; i0 += 1024) {      // position n receives the total
    const int i = i0 + threadIdx.x;
    const int vh = i < n ? cnt_h[i] : 0;
    long long ih = vh;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const long long oh = __shfl_up(ih, d, 64);
      if (lane >= d) ih += oh;
    }
    if (lane == 63) tot[wave] = ih;
    __syncthreads();
    long long bh = carry;
    for (int w = 0; w < wave; ++w) bh += tot[w];
    if (i <= n) {
      const long long eh = bh + ih - vh;
      cnt_h[i] = (int)(eh < 0x7FFFFFFF ? eh : 0x7FFFFFFF);
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = bh + ih;
    __syncthreads();
  }
  if (threadIdx.x == 0 && carry > cap) ctl->dense = 1;
}

// one wave per row: the set bits of the hard mask row become an ascending column list + the SC values.
// Round 6: the row is compacted first.  Up to round 5 the wave walked the row word by word and evaluated the cross length under the
// word's bits as the execution mask - `words` (125 at n = 8000) trips through ~70 VALU slots with a few percent of the lanes
// alive, 1.04 ms per 16-pair step, the largest kernel of the back-end.  Now lane l takes word w0 + l of a 64-word chunk, a wave scan
// of the popcounts gives every word its place, the lanes write their words' columns into a wave-private LDS list (a trip per set bit
// of the fullest word), and the list is then worked off 64 entries at a time with every lane busy; columns and values leave as
// contiguous runs.  Same expression per entry, same order: the arrays are what they were, bit for bit.
constexpr int CSR_LIST = 64 * 64;       // columns of a 64-word chunk
__device__ __forceinline__ void d_csr_fill(const float* __restrict__ src, const float* __restrict__ tgt, int n, int words,
                                           float inv_d2, const unsigned long long* __restrict__ hard,
                                           const int* __restrict__ ptr_h, unsigned short* __restrict__ col_h,
                                           float* __restrict__ val_h, const Sc2Ctl* __restrict__ ctl, int legacy) {
  if (ctl->dense) return;
  __shared__ unsigned short list[4][CSR_LIST];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;                                              // wave-uniform; no workgroup barrier below
  const float sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
  const float tx = tgt[3 * i], ty = tgt[3 * i + 1], tz = tgt[3 * i + 2];
  int oh = ptr_h[i];
  if (legacy) {                                                    // the round-5 walk (eyoc_sc2pcr_select_kernels bit 0)
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int w = 0; w < words; ++w) {
      const unsigned long long hm = hard[(size_t)i * words + w];
      const int j = w * 64 + lane;
      if ((hm >> lane) & 1ull) {
        const float c = cross_len(sx, sy, sz, tx, ty, tz, src[3 * j], src[3 * j + 1], src[3 * j + 2], tgt[3 * j], tgt[3 * j + 1],
                                  tgt[3 * j + 2]);
        const int pos = oh + __popcll(hm & lt);
        col_h[pos] = (unsigned short)j;
        val_h[pos] = sc_value(c, inv_d2);   // the expression of the dense sweep
      }
      oh += __popcll(hm);
    }
    return;
  }
  unsigned short* L = list[wave];
  for (int w0 = 0; w0 < words; w0 += 64) {
    const int w = w0 + lane;
    unsigned long long m = w < words ? hard[(size_t)i * words + w] : 0ull;
    const int pc = __popcll(m);
    int incl = pc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    const int total = __shfl(incl, 63, 64);
    int p = incl - pc;
    while (m) {                                                    // columns of word w in ascending order
      L[p++] = (unsigned short)(w * 64 + __builtin_ctzll(m));
      m &= m - 1ull;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // wave-private list: writes before the reads below
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < total; k += 64) {
      const int j = L[k];
      const float c = cross_len(sx, sy, sz, tx, ty, tz, src[3 * j], src[3 * j + 1], src[3 * j + 2], tgt[3 * j], tgt[3 * j + 1],
                                tgt[3 * j + 2]);
      col_h[oh + k] = (unsigned short)j;
      val_h[oh + k] = sc_value(c, inv_d2);   // the expression of the dense sweep
    }
    oh += total;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // ... and the reads before the next chunk's writes
    __builtin_amdgcn_wave_barrier();
  }
}

// y = SC x (one power sweep): one wave per row, lanes stride the row's entries, fixed-order wave reduction.  Over the CSR lists - or,
// for a pair whose graph overflowed them (ctl->dense: more than N^2 / 4 edges, e.g. all-inlier inputs), over the whole row with the
// compatibilities recomputed from the coordinates.  (Up to round 5 the dense fallback was two kernels of its own - a column-split
// mat-vec and its reduction - launched behind every sparse sweep and leaving at once for the pairs that had their lists: 40 launches
// and 0.26 ms per 16-pair step of nothing.)
constexpr int SPMV_ROWS = 2, SPMV_AHEAD = 4;       // rows per wave x entries per lane and row in flight
__device__ __forceinline__ void d_sc_spmv(const float* __restrict__ src, const float* __restrict__ tgt, float inv_d2,
                                          const int* __restrict__ ptr_h, const unsigned short* __restrict__ col_h,
                                          const float* __restrict__ val_h, int n, const float* __restrict__ x,
                                          float* __restrict__ y, const Sc2Ctl* __restrict__ ctl) {
  if (ctl->converged) return;
  const int lane = threadIdx.x & 63;
  const int i0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * SPMV_ROWS;
  if (i0 >= n) return;
  if (ctl->dense) {
#pragma unroll 1
    for (int r = 0; r < SPMV_ROWS && i0 + r < n; ++r) {
      const int i = i0 + r;
      const float sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
      const float tx = tgt[3 * i], ty = tgt[3 * i + 1], tz = tgt[3 * i + 2];
      float a = 0.0f;
      for (int j = lane; j < n; j += 64) {
        const float c = cross_len(sx, sy, sz, tx, ty, tz, src[3 * j], src[3 * j + 1], src[3 * j + 2], tgt[3 * j], tgt[3 * j + 1],
                                  tgt[3 * j + 2]);
        a = __builtin_fmaf(sc_value(c, inv_d2), x[j], a);
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) a += __shfl_down(a, d, 64);
      if (lane == 0) y[i] = a;
    }
    return;
  }
  // Round 6: SPMV_ROWS rows per wave, SPMV_AHEAD entries per lane and row in flight, and NO branch around a load.  A wave used to own one
  // row and walk it in a rolled loop: row bounds -> (column, value) -> gathered x -> six-step reduction, every arrow a memory round trip,
  // ~5 us of life for 80 entries - and the chip holds 8192 waves, so a sweep of 16 pairs (128 k rows) took 70 us at 2.3 TB/s whatever
  // the memory system could do (a single pair: 5.9 us, four: 15.8 - linear in the rows).  Conditional loads (`if (k < e) acc += val[k] * x[col[k]]`,
  // several rows or entries unrolled) did not help: each conditional region waits for its own column before its gather.  Here the
  // indices are clamped into the row instead, all loads of a step are issued back to back, and only the fma is predicated: 43.8 -> 35.0 us
  // per launch with two rows x four entries (37.3 with 4 x 2, 36.1 with 1 x 4, 38.3 with 1 x 8, 45 with 8 x 1).  (x staged in LDS on top - the gathers touch up to 64 lines per instruction - measured 39.6 us with 32 rows per
  // workgroup and 84 us with 256: the copies and the lost parallelism cost what the LDS gathers save.)  Every
  // row's sum is formed in the same order as before (k ascending per lane, one fma each, the same tree).
  int k[SPMV_ROWS], b[SPMV_ROWS], e[SPMV_ROWS];
  float acc[SPMV_ROWS];
#pragma unroll
  for (int r = 0; r < SPMV_ROWS; ++r) {
    const int i = min(i0 + r, n - 1);
    b[r] = ptr_h[i];
    e[r] = i0 + r < n ? ptr_h[i + 1] : b[r];                // rows past the end: empty
    k[r] = b[r] + lane;
    acc[r] = 0.0f;
  }
  bool more = true;
  while (more) {                                            // wave-uniform
    int c[SPMV_ROWS][SPMV_AHEAD];
    float v[SPMV_ROWS][SPMV_AHEAD], xv[SPMV_ROWS][SPMV_AHEAD];
#pragma unroll
    for (int r = 0; r < SPMV_ROWS; ++r)
#pragma unroll
      for (int u = 0; u < SPMV_AHEAD; ++u) {
        const int kc = max(min(k[r] + 64 * u, e[r] - 1), 0);   // inside the row (an empty row reads a neighbour's entry; nothing is added)
        c[r][u] = col_h[kc];
        v[r][u] = val_h[kc];
      }
#pragma unroll
    for (int r = 0; r < SPMV_ROWS; ++r)
#pragma unroll
      for (int u = 0; u < SPMV_AHEAD; ++u) xv[r][u] = x[c[r][u]];
    bool m = false;
#pragma unroll
    for (int r = 0; r < SPMV_ROWS; ++r) {
#pragma unroll
      for (int u = 0; u < SPMV_AHEAD; ++u) {
        const float f = __builtin_fmaf(v[r][u], xv[r][u], acc[r]);
        acc[r] = k[r] + 64 * u < e[r] ? f : acc[r];
      }
      k[r] += 64 * SPMV_AHEAD;
      m |= k[r] < e[r];
    }
    more = __any(m);
  }
#pragma unroll
  for (int r = 0; r < SPMV_ROWS; ++r) {
    float a = acc[r];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_down(a, d, 64);
    if (lane == 0 && i0 + r < n) y[i0 + r] = a;
  }
}

// ---- non-maximum suppression in source space: score = conf if no j within R has a larger conf, else 0
// grid.y splits the columns; `dom` (zero-initialised) collects "some column dominates row i" with atomicOr.
__device__ __forceinline__ void d_nms(const float* __restrict__ src, const float* __restrict__ conf, int n, float R,
                                             int col_chunk, int* __restrict__ dom, int legacy) {
  __shared__ float ls[1024 * 3], lc[1024];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool ok = i < n;
  const int ii = ok ? i : 0;
  const float sx = src[3 * ii], sy = src[3 * ii + 1], sz = src[3 * ii + 2], ci = conf[ii];
  bool dominated = false;
  const float T = sqrt_lt_threshold(R);                 // sqrtf(x) < R  <=>  x < T
  const int c_begin = blockIdx.y * col_chunk, c_end = min(n, c_begin + col_chunk);
  for (int j0 = c_begin; j0 < c_end; j0 += 1024) {
    const int cnt = min(1024, c_end - j0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * 3; t += 256) ls[t] = src[3 * j0 + t];
    for (int t = threadIdx.x; t < cnt; t += 256) lc[t] = conf[j0 + t];
    __syncthreads();
    if (legacy) {
      for (int j = 0; j < cnt; ++j) {
        const float dx = sx - ls[3 * j], dy = sy - ls[3 * j + 1], dz = sz - ls[3 * j + 2];
        dominated |= (lc[j] > ci) && (sqrtf(sq_len(dx, dy, dz)) < R);
      }
    } else {
      for (int j = 0; j < cnt; ++j) {
        const float dx = sx - ls[3 * j], dy = sy - ls[3 * j + 1], dz = sz - ls[3 * j + 2];
        dominated |= (lc[j] > ci) & (sq_len(dx, dy, dz) < T);
      }
    }
  }
  if (ok && dominated) atomicOr(&dom[i], 1);
}

__device__ __forceinline__ void d_nms_score(const float* __restrict__ conf, const int* __restrict__ dom, int n, float* __restrict__ score) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) score[i] = dom[i] ? 0.0f : conf[i];
}

// ---- stable descending rank of score; the first n_seed ranks are the seeds
// (grid.y splits the columns; the partial ranks are integers, so atomicAdd keeps the result exact)
__device__ __forceinline__ void d_rank(const float* __restrict__ score, int n, int col_chunk, int* __restrict__ rank_out) {
  __shared__ float lc[1024];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool ok = i < n;
  const float si = score[ok ? i : 0];
  int rank = 0;
  const int c_begin = blockIdx.y * col_chunk, c_end = min(n, c_begin + col_chunk);
  for (int j0 = c_begin; j0 < c_end; j0 += 1024) {
    const int cnt = min(1024, c_end - j0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) lc[t] = score[j0 + t];
    __syncthreads();
    for (int j = 0; j < cnt; ++j) rank += (lc[j] > si) || (lc[j] == si && (j0 + j) < i);
  }
  if (ok && rank) atomicAdd(&rank_out[i], rank);
}

__device__ __forceinline__ void d_seeds(const int* __restrict__ rank, int n, int n_seed, int* __restrict__ seeds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && rank[i] < n_seed) seeds[rank[i]] = i;
}

// ---- per seed: SC2 row = popcount(tight[seed] & tight[j]) * hard[seed][j]; stable top-k1 of it
// (value descending, index ascending among equals).  One workgroup per seed; the row (n uint16
// counts) lives in LDS.  Selection is exact and deterministic.  Round 5: a SHORT LIST first - the k1-th largest of 64
// group maxima (thread t scans entries t, t + 256, ...; four threads form a group) is a lower bound L of the k1-th
// largest value, so the entries >= max(L, 1) (a few dozen) hold the whole answer: each counts the list entries in front of
// it (value descending, index ascending = one comparison of packed keys) and stores itself at that position; a row with
// fewer than k1 non-zero counts is completed with its lowest-index zeros.  Rows whose list does not fit (more than 1024
// entries at or above the bound: many equal values) take the histogram path the kernel used for every seed before -
// coarse then fine histogram give the k1-th largest value v*, everything above v* is taken, and of the values equal to
// v* the lowest indices through an ordered scan.  (The histogram's LDS atomics - 2400 per inlier seed, clustered on a few
// buckets - and thread 0's walk down the 1024 buckets were 2.6 of the kernel's 4.3 ms per 16-pair step.)
// ``cnt_row`` != NULL: the seed sits in a DENSE block (d_seed_dense below) and its row of counts is already in memory.
__device__ __forceinline__ void d_seed_topk(const unsigned long long* __restrict__ hard,
                                                   const unsigned long long* __restrict__ tight, int n, int words,
                                                   const int* __restrict__ seeds, int k1, int* __restrict__ knn1,
                                                   const unsigned short* __restrict__ cnt_row, int list_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  unsigned short* row = reinterpret_cast<unsigned short*>(dyn);                                            // [n]
  unsigned long long* srow = reinterpret_cast<unsigned long long*>(dyn + ((size_t)n * 2 + 15) / 16 * 16);  // [words]
  unsigned short* cand = reinterpret_cast<unsigned short*>(dyn + ((size_t)n * 2 + 15) / 16 * 16 + (size_t)words * 8);  // [<= n]
  __shared__ int cand_n;
  __shared__ int hist[1024];
  __shared__ int fine[16];
  __shared__ int thr_bucket, thr_value, n_above, need_eq;
  __shared__ int above_idx[K1_MAX], above_val[K1_MAX], above_n;
  __shared__ int eq_idx[K1_MAX];
  __shared__ int wave_cnt[4];
  const int s = blockIdx.x;
  const int seed = seeds[s];
  const unsigned long long* ts = tight + (size_t)seed * words;
  const unsigned long long* hs = hard + (size_t)seed * words;
  for (int w = threadIdx.x; w < words; w += 256) srow[w] = ts[w];
  for (int t = threadIdx.x; t < 1024; t += 256) hist[t] = 0;
  if (threadIdx.x < 16) fine[threadIdx.x] = 0;
  if (threadIdx.x == 0) above_n = 0;
  __syncthreads();
  // The hard row is sparse (a few percent): compact its set bits into an LDS list, then ONE WAVE per candidate j
  // ANDs the two 1 KB tight rows with coalesced loads (a lane-per-j loop read them 8 bytes at a time, every lane a
  // different row, and idled on the unset bits).
  if (cnt_row) {                                       // workgroup-uniform
    // 16 bytes per lane and load, all of a thread's loads in flight (round 6; the row was copied two bytes at a time in a loop the
    // compiler kept rolled: 31 dependent round trips to memory per seed).  A count row is words * 64 entries = a multiple of 128 bytes
    // at a 256-byte aligned base, and `row` is rounded up to 16 bytes, so the last vector stays inside both.
    const uint4* __restrict__ src4 = reinterpret_cast<const uint4*>(cnt_row);
    uint4* dst4 = reinterpret_cast<uint4*>(row);
    const int n8 = (n + 7) / 8;
#pragma unroll 4
    for (int q = threadIdx.x; q < n8; q += 256) dst4[q] = src4[q];
  } else {
  for (int j = threadIdx.x; j < n; j += 256) row[j] = 0;
  if (threadIdx.x == 0) cand_n = 0;
  __syncthreads();
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = wave; w < words; w += 4) {
      const unsigned long long bits = hs[w];
      const bool set = (bits >> lane) & 1ull;          // columns >= n are never set (k_masks)
      int base = 0;
      if (lane == 0 && bits) base = atomicAdd(&cand_n, __popcll(bits));
      base = __shfl(base, 0, 64);
      if (set) cand[base + __popcll(bits & ((1ull << lane) - 1ull))] = (unsigned short)(w * 64 + lane);
    }
    __syncthreads();
    const int nc = cand_n;
    // four candidates per wave in flight: the loop is bound by the latency of the row loads, not by the popcounts
    for (int ci = wave * 4; ci < nc; ci += 16) {
      int c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = cand[min(ci + u, nc - 1)];
        const unsigned long long* tj = tight + (size_t)j * words;
        int acc = 0;
        for (int w = lane; w < words; w += 64) acc += __popcll(srow[w] & tj[w]);
        c[u] = acc;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c[u] += __shfl_down(c[u], d, 64);
      }
      if (lane == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ci + u < nc) row[cand[ci + u]] = (unsigned short)c[u];
      }
    }
  }
  }
  __syncthreads();
  {   // ---- short list
    constexpr int CAP = 1024;
    __shared__ unsigned int keys[CAP];
    __shared__ unsigned short tmax[256];
    __shared__ int keys_n, bound;
    const int t = threadIdx.x, ln = t & 63;
    unsigned int mx = 0;
    for (int j = t; j < n; j += 256) mx = max(mx, (unsigned int)row[j]);
    tmax[t] = (unsigned short)mx;
    if (t == 0) keys_n = 0;
    __syncthreads();
    if (t < 64) {
      const unsigned int g = max(max((unsigned int)tmax[4 * t], (unsigned int)tmax[4 * t + 1]), max((unsigned int)tmax[4 * t + 2], (unsigned int)tmax[4 * t + 3]));
      int r = 0;
      for (int u = 0; u < 64; ++u) {
        const unsigned int o = (unsigned int)__shfl((int)g, u, 64);
        r += (o > g) | ((o == g) & (u < ln));
      }
      if (r == k1 - 1) bound = (int)g;                 // k1 <= K1_MAX = 32 <= 64 groups: exactly one lane
    }
    __syncthreads();
    const unsigned int lc = (unsigned int)max(bound, 1);
    for (int j = t; j < n; j += 256) {
      const unsigned int c = row[j];
      if (c >= lc) {
        const int p = atomicAdd(&keys_n, 1);
        if (p < CAP) keys[p] = (c << 14) | (unsigned int)(16383 - j);     // j < MAX_N = 2^14, c <= n: larger key = earlier in the order
      }
    }
    __syncthreads();
    const int m = keys_n;
    if (m <= min(CAP, list_cap)) {                       // workgroup-uniform (list_cap: eyoc_sc2pcr_set_shortlist_cap, 0 = always the histogram)
      for (int a = t; a < m; a += 256) {
        const unsigned int key = keys[a];
        int r = 0;
        for (int q = 0; q < m; ++q) r += keys[q] > key;
        if (r < k1) knn1[(size_t)s * k1 + r] = 16383 - (int)(key & 16383u);
      }
      if (m < k1 && t < 64) {                            // fewer than k1 non-zero counts: the lowest-index zeros complete the list
        const int need = k1 - m;
        int base = 0;
        for (int j0 = 0; j0 < n && base < need; j0 += 64) {
          const int j = j0 + ln;
          const bool f = j < n && row[j] == 0;
          const unsigned long long bm = __ballot(f);
          const int pos = base + __popcll(bm & ((1ull << ln) - 1ull));
          if (f && pos < need) knn1[(size_t)s * k1 + m + pos] = j;
          base += __popcll(bm);
        }
      }
      return;
    }
  }
  // ---- histogram path (the list overflowed)
  for (int j = threadIdx.x; j < n; j += 256) atomicAdd(&hist[row[j] >> 4], 1);
  __syncthreads();
  {   // bucket of the k1-th largest value.  (Round 5: thread 0 used to walk the 1024 buckets down from the top - ~30 us of
      // dependent LDS reads per seed when most counts are small, 0.75 ms per 16-pair step.)  Thread t owns buckets 4 t .. 4 t + 3;
      // a suffix scan over the threads gives every thread the number of entries above its buckets, and exactly one
      // (thread, bucket) has  above < k1 <= above + own.
    const int t = threadIdx.x, ln = t & 63, wv = t >> 6;
    const int h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
    const int mine = h0 + h1 + h2 + h3;
    int suf = mine;                                   // inclusive suffix sum over the lanes of the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_down(suf, d, 64);
      if (ln + d < 64) suf += o;
    }
    if (ln == 0) wave_cnt[wv] = suf;
    __syncthreads();
    for (int w = wv + 1; w < 4; ++w) suf += wave_cnt[w];
    int cum = suf - mine;                             // entries in buckets above 4 t + 3
    if (cum < k1 && suf >= k1) {                      // the threshold bucket is one of mine (t == 0: bucket 0 takes what is left)
      int b = 3;
      const int hh[4] = {h0, h1, h2, h3};
      for (; b > 0; --b) {
        if (cum + hh[b] >= k1) break;
        cum += hh[b];
      }
      thr_bucket = 4 * t + b;
      n_above = cum;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += 256)
    if ((row[j] >> 4) == thr_bucket) atomicAdd(&fine[row[j] & 15], 1);
  __syncthreads();
  if (threadIdx.x == 0) {   // exact k1-th largest value
    int cum = n_above, f = 15;
    for (; f > 0; --f) {
      if (cum + fine[f] >= k1) break;
      cum += fine[f];
    }
    thr_value = thr_bucket * 16 + f;
    n_above = cum;                       // values strictly greater than v*
    need_eq = min(k1 - cum, fine[f]);    // how many v* entries complete the list (n >= k1 guarantees enough)
  }
  __syncthreads();
  const int vstar = thr_value;
  // strictly-greater entries: at most k1 - 1 of them, order fixed afterwards
  for (int j = threadIdx.x; j < n; j += 256)
    if (row[j] > vstar) {
      const int p = atomicAdd(&above_n, 1);
      if (p < K1_MAX) { above_idx[p] = j; above_val[p] = row[j]; }
    }
  // equal entries: lowest indices first.  Wave w owns the w-th quarter of the row and walks it 64 entries at a time (ballots give
  // the order inside a step; consecutive lanes read consecutive entries - a thread per contiguous range of 32 entries had every
  // wave access 16-way bank-conflicted)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qlen = (n + 3) / 4, q_lo = wave * qlen, q_hi = min(n, q_lo + qlen);
  int cnt_eq = 0;
  for (int j0 = q_lo; j0 < q_hi; j0 += 64) {
    const int j = j0 + lane;
    cnt_eq += __popcll(__ballot(j < q_hi && row[j] == vstar));
  }
  if (lane == 0) wave_cnt[wave] = cnt_eq;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wave_cnt[w];
  for (int j0 = q_lo; j0 < q_hi && base < need_eq; j0 += 64) {      // wave-uniform
    const int j = j0 + lane;
    const bool f = j < q_hi && row[j] == vstar;
    const unsigned long long m = __ballot(f);
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (f && pos < need_eq) eq_idx[pos] = j;
    base += __popcll(m);
  }
  __syncthreads();
  {   // order: value descending, index ascending - every entry counts the entries in front of it (<= 31 of them)
    const int na = min(above_n, K1_MAX);
    if ((int)threadIdx.x < na) {
      const int v = above_val[threadIdx.x], ix = above_idx[threadIdx.x];
      int r = 0;
      for (int q = 0; q < na; ++q) r += (above_val[q] > v) | ((above_val[q] == v) & (above_idx[q] < ix));
      if (r < k1) knn1[(size_t)s * k1 + r] = ix;
    }
    const int ne = min(need_eq, k1 - min(na, k1));
    if ((int)threadIdx.x < ne) knn1[(size_t)s * k1 + na + threadIdx.x] = eq_idx[threadIdx.x];
    for (int m = na + max(ne, 0) + (int)threadIdx.x; m < k1; m += 256) knn1[(size_t)s * k1 + m] = seed;   // unreachable for n >= k1
  }
}

// ---- dense seed blocks (round 5).  The seeds are the correspondences ranked by the leading eigenvector, so at a real inlier
// ratio the first few hundred seeds are inliers, each compatible with ALL other inliers: 2400 candidates x 1 KB of tight-row
// reads per seed at n = 8000, 30 % inliers - d_seed_topk's one-wave-per-(seed, candidate) loop spent 4 of the 18 ms of a
// 16-pair step on them, bound by the latency of the row loads and its cross-lane reductions.  A block of 64 consecutive seeds
// whose hard rows hold >= 2 n candidates together takes this path (measured on the bench's nuScenes-shaped pairs: x = 1-2 is the
// minimum of the two kernels' sum, 2.14 ms against 2.35 at x = 6 and 2.95 without dense blocks) instead: LANE = SEED, the seeds' tight rows live in
// REGISTERS (wave k of the workgroup keeps words [k W, (k+1) W) of all 64 rows), a candidate's row arrives through the scalar
// cache (wave-uniform address), and a (seed, candidate) count costs 4 VALU instructions per word with no LDS read, no
// reduction and no memory traffic beyond 1 KB per (block, candidate).  The four partial counts meet in LDS; wave k then
// stores 16 candidates x 64 seeds as 32-byte pieces of the seeds' count rows.  Only candidates some seed of the block is
// hard-compatible with are visited (the union of the 64 hard words).  d_seed_topk reads a dense block's rows instead of
// computing them; the counts are integers, so the result is the same whichever path produced them.
constexpr int DENSE_SPLITS = 16;     // workgroups per (pair, seed block): each takes 1/16 of the candidate words

__device__ __forceinline__ void d_seed_blocks(const int* __restrict__ ptr_h, const int* __restrict__ seeds, int n, int n_seed,
                                              const Sc2Ctl* __restrict__ ctl, unsigned char* __restrict__ blk_dense, int dense_x) {
  const int lane = threadIdx.x & 63;
  const int si = blockIdx.x * 64 + lane;
  long long deg = 0;
  if (si < n_seed) {
    const int s = seeds[si];
    deg = ctl->dense ? n : (long long)(ptr_h[s + 1] - ptr_h[s]);     // (the CSR offsets are clamped when the graph overflowed them)
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) deg += __shfl_down(deg, d, 64);
  if (lane == 0) blk_dense[blockIdx.x] = (dense_x >= 0 && deg >= (long long)dense_x * n) ? 1 : 0;
}

template <int W>
__device__ __forceinline__ void d_seed_dense(const unsigned long long* __restrict__ hard,
                                             const unsigned long long* __restrict__ tight, int n, int words,
                                             const int* __restrict__ seeds, int n_seed, unsigned short* __restrict__ cnt) {
  __shared__ unsigned short part[4][64][64];          // [wave][candidate bit][seed]: conflict-free on both sides
  __shared__ unsigned long long hws[64];
  __shared__ unsigned long long uni;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int si = blockIdx.x * 64 + lane;
  const bool valid = si < n_seed;
  const int seed = seeds[valid ? si : blockIdx.x * 64];
  const int w0 = wave * W;
  const int wn = max(0, min(W, words - w0));            // this wave's words of a row (wave-uniform)
  unsigned long long S[W];
#pragma unroll
  for (int t = 0; t < W; ++t) S[t] = t < wn ? tight[(size_t)seed * words + w0 + t] : 0ull;
  const int per = (words + (int)gridDim.y - 1) / (int)gridDim.y;
  const int jw0 = (int)blockIdx.y * per, jw1 = min(words, jw0 + per);
  const size_t pitch = (size_t)words * 64;
  for (int jw = jw0; jw < jw1; ++jw) {
    if (wave == 0) {
      unsigned long long hw = valid ? hard[(size_t)seed * words + jw] : 0ull;
      hws[lane] = hw;
      unsigned int lo = (unsigned int)hw, hi = (unsigned int)(hw >> 32);
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { lo |= __shfl_xor(lo, d, 64); hi |= __shfl_xor(hi, d, 64); }
      if (lane == 0) uni = ((unsigned long long)hi << 32) | lo;
    }
    __syncthreads();
    const unsigned long long u_all = uni;
    unsigned int ulo = __builtin_amdgcn_readfirstlane((unsigned int)u_all), uhi = __builtin_amdgcn_readfirstlane((unsigned int)(u_all >> 32));
    unsigned long long u = ((unsigned long long)uhi << 32) | ulo;
    while (u) {                                          // wave-uniform: the candidates some seed of the block needs
      const int b = __builtin_ctzll(u);
      u &= u - 1;
      const unsigned long long* __restrict__ tj = tight + (size_t)(jw * 64 + b) * words + w0;   // uniform -> scalar loads
      // every word unconditionally: the S words past the row's end are zero, so what the candidate's pointer reads there (the next
      // row, or the workspace arrays behind the matrix) does not count - and without a guard per word the compiler merges the
      // uniform loads into s_load_dwordx16 (with the guards it issued one s_load_dwordx2 + s_waitcnt per word: 0.33 of the VALU rate)
      unsigned int a0 = 0, a1 = 0, a2 = 0, a3 = 0;          // four chains: v_bcnt accumulates into its own result
#pragma unroll
      for (int t = 0; t < W; t += 2) {
        const unsigned long long x = S[t] & tj[t], y = S[t + 1] & tj[t + 1];
        a0 += __popc((unsigned int)x);
        a1 += __popc((unsigned int)(x >> 32));
        a2 += __popc((unsigned int)y);
        a3 += __popc((unsigned int)(y >> 32));
      }
      part[wave][b][lane] = (unsigned short)((a0 + a1) + (a2 + a3));
    }
    __syncthreads();
    {   // wave k: candidates 16 k .. 16 k + 15 of this word, 64 seeds: sum of the four partial counts, masked by the seed's hard bit
      const unsigned long long hw = hws[lane];
      unsigned int pk[8];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int b = 16 * wave + e;
        unsigned int c = 0;
        if ((u_all >> b) & 1ull) {                       // uniform
          c = (unsigned int)part[0][b][lane] + part[1][b][lane] + part[2][b][lane] + part[3][b][lane];
          c = ((hw >> b) & 1ull) ? c : 0u;
        }
        if (e & 1) pk[e >> 1] |= c << 16; else pk[e >> 1] = c;
      }
      if (valid) {
        uint4* dst = reinterpret_cast<uint4*>(cnt + (size_t)si * pitch + (size_t)jw * 64 + 16 * wave);
        dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      }
    }
    __syncthreads();
  }
}

// rotation and translation of a seed's weighted Kabsch problem (scripts/SC2_PCR/common.py:7-45); the translation's roundings pinned so that
// the wave-per-seed kernel and the lane-per-seed kernel give the same bits
__device__ inline void seed_pose(const double ca[3], const double cb[3], const double H[3][3], double R[3][3], double t[3]) {
  kabsch_rotation(H, R);
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = cb[i] - __builtin_fma(R[i][2], ca[2], __builtin_fma(R[i][1], ca[1], R[i][0] * ca[0]));
}

// lane = seed: the 3 x 3 problems of 64 seeds side by side (d_seed_solve left centroids and cross-covariance in seed_h)
__device__ __forceinline__ void d_seed_kabsch(const double* __restrict__ seed_h, int n_seed, float* __restrict__ Ts,
                                              float* __restrict__ fitness) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seed) return;
  const double* o = seed_h + 16 * (size_t)s;
  const double ca[3] = {o[0], o[1], o[2]}, cb[3] = {o[3], o[4], o[5]};
  const double H[3][3] = {{o[6], o[7], o[8]}, {o[9], o[10], o[11]}, {o[12], o[13], o[14]}};
  double R[3][3], t[3];
  seed_pose(ca, cb, H, R, t);
  write_T(Ts + 16 * (size_t)s, R, t);
  fitness[s] = 0.0f;                                     // d_seed_fitness adds its point ranges' counts (integers: exact in any order)
}

// fitness of the hypotheses over ALL correspondences (fp32 like the reference's einsum path).  A wave keeps 256 correspondences in
// registers (four per lane) and walks a range of the seeds, whose transforms arrive through the scalar cache (three s_load_dwordx4
// per seed) - 19 VALU instructions per seed and 64 points, no vector load in the loop, the count of a (wave, seed) by ballots.  The
// shares of a seed's point blocks meet by atomicAdd on the float count: integers below 2^24, exact in any order.  (First version:
// lane = seed with the POINTS through the scalar cache - 48 scalar address computations and loads per 8 points, SGPRs spilled
// to VGPR lanes: 0.29 ms per 16-pair step against 0.13 for this one.)
constexpr int FIT_PTS = 4;          // correspondences per lane
constexpr int FIT_SPLIT = 16;       // seed ranges (grid.y)
template <bool LEGACY>      // (a template, not a run-time select inside the loop: the compiler evaluated both forms of the test for every residual)
__device__ __forceinline__ void d_seed_fitness(const float* __restrict__ src, const float* __restrict__ tgt, int n, int n_seed,
                                               const float* __restrict__ Ts, float inlier_thr, float* __restrict__ fitness) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int base = ((int)blockIdx.x * 4 + wave) * (64 * FIT_PTS);
  if (base >= n) return;                                   // wave-uniform
  float x[FIT_PTS], y[FIT_PTS], z[FIT_PTS], qx[FIT_PTS], qy[FIT_PTS], qz[FIT_PTS];
#pragma unroll
  for (int u = 0; u < FIT_PTS; ++u) {
    const int j = base + u * 64 + lane;
    if (j < n) {
      x[u] = src[3 * j]; y[u] = src[3 * j + 1]; z[u] = src[3 * j + 2];
      qx[u] = tgt[3 * j]; qy[u] = tgt[3 * j + 1]; qz[u] = tgt[3 * j + 2];
    } else {   // past the end: a correspondence no transform brings within reach (squared residual ~3e36: no overflow, never counted)
      x[u] = y[u] = z[u] = 0.0f; qx[u] = qy[u] = qz[u] = 1e18f;
    }
  }
  const int per = (n_seed + (int)gridDim.y - 1) / (int)gridDim.y;
  const int s0 = (int)blockIdx.y * per, s1 = min(n_seed, s0 + per);
  const float T_in = LEGACY ? 0.0f : sqrt_lt_threshold(inlier_thr);      // sqrtf(x) < inlier_thr  <=>  x < T_in
  if (s0 >= s1) return;
  float Tn[12];                                            // the next seed's transform, requested one seed ahead (an L2 round trip of the scalar cache otherwise sits in front of every seed's 76 instructions)
#pragma unroll
  for (int i = 0; i < 12; ++i) Tn[i] = Ts[16 * (size_t)s0 + i];
  for (int s = s0; s < s1; ++s) {                          // wave-uniform
    // = (float) of the fp64 pose, what the one-kernel form used
    const float r00 = Tn[0], r01 = Tn[1], r02 = Tn[2], t0 = Tn[3], r10 = Tn[4], r11 = Tn[5], r12 = Tn[6], t1 = Tn[7], r20 = Tn[8], r21 = Tn[9],
                r22 = Tn[10], t2 = Tn[11];
    {
      const float* __restrict__ T = Ts + 16 * (size_t)min(s + 1, s1 - 1);
#pragma unroll
      for (int i = 0; i < 12; ++i) Tn[i] = T[i];
    }
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < FIT_PTS; ++u) {
      const float dx = (__builtin_fmaf(r02, z[u], __builtin_fmaf(r01, y[u], r00 * x[u])) + t0) - qx[u];       // the expressions of d_seed_solve
      const float dy = (__builtin_fmaf(r12, z[u], __builtin_fmaf(r11, y[u], r10 * x[u])) + t1) - qy[u];
      const float dz = (__builtin_fmaf(r22, z[u], __builtin_fmaf(r21, y[u], r20 * x[u])) + t2) - qz[u];
      const float d2 = sq_len(dx, dy, dz);
      cnt += __popcll(__ballot(LEGACY ? sqrtf(d2) < inlier_thr : d2 < T_in));
    }
    if (lane == 0 && cnt) atomicAdd(&fitness[s], (float)cnt);
  }
}

// ---- per seed (one wave): local consensus, power iteration, weighted Kabsch, inlier count
__device__ __forceinline__ void d_seed_solve(const float* __restrict__ src, const float* __restrict__ tgt, int n,
                                                    int n_seed, const int* __restrict__ knn1, int k1, int k2, float d,
                                                    int max_iter, float inlier_thr, float* __restrict__ Ts,
                                                    float* __restrict__ fitness, int legacy, double* __restrict__ seed_h) {
  __shared__ volatile int inv[4][K1_MAX];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int s = blockIdx.x * 4 + wv;
  if (s >= n_seed) return;
  // stage 1: the k1 neighbours, one per lane
  const bool in1 = lane < k1;
  const int idx = knn1[(size_t)s * k1 + (in1 ? lane : 0)];
  float px = src[3 * idx], py = src[3 * idx + 1], pz = src[3 * idx + 2];
  float qx = tgt[3 * idx], qy = tgt[3 * idx + 1], qz = tgt[3 * idx + 2];
  unsigned int rowmask = 0;   // bit a: compatible(a, lane) under d
  for (int a = 0; a < k1; ++a) {
    const float c = cross_len(px, py, pz, qx, qy, qz, __shfl(px, a, 64), __shfl(py, a, 64), __shfl(pz, a, 64),
                              __shfl(qx, a, 64), __shfl(qy, a, 64), __shfl(qz, a, 64));
    if (c < d) rowmask |= 1u << a;
  }
  // local second-order score: first row of the hard matrix times the matrix
  const unsigned int row0 = (unsigned int)__shfl((int)rowmask, 0, 64);
  const int L = in1 ? __popc(row0 & rowmask) : -1;
  int rank = 0;
  for (int c = 0; c < k1; ++c) {
    const int Lc = __shfl(L, c, 64);
    rank += (Lc > L) || (Lc == L && c < lane);
  }
  if (in1 && rank < k2) inv[wv][rank] = lane;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // wave-private LDS slot: order the write before the read
  __builtin_amdgcn_wave_barrier();
  // stage 2: the k2 survivors, re-ordered by rank, one per lane
  const bool in2 = lane < k2;
  const int from = inv[wv][in2 ? lane : 0];
  px = __shfl(px, from, 64); py = __shfl(py, from, 64); pz = __shfl(pz, from, 64);
  qx = __shfl(qx, from, 64); qy = __shfl(qy, from, 64); qz = __shfl(qz, from, 64);
  float S[K2_MAX];   // soft compatibility row of this lane, zero diagonal
  const float inv_d2 = 1.0f / (d * d);
#pragma unroll
  for (int b = 0; b < K2_MAX; ++b) {
    float v = 0.0f;
    if (b < k2) {
      const float c = cross_len(px, py, pz, qx, qy, qz, __shfl(px, b, 64), __shfl(py, b, 64), __shfl(pz, b, 64),
                                __shfl(qx, b, 64), __shfl(qy, b, 64), __shfl(qz, b, 64));
      v = (b == lane) ? 0.0f : sc_value(c, inv_d2);
    }
    S[b] = v;
  }
  float v = in2 ? 1.0f : 0.0f;
  for (int it = 0; it < max_iter; ++it) {
    float y = 0.0f;
#pragma unroll
    for (int b = 0; b < K2_MAX; ++b) y += S[b] * __shfl(v, b, 64);
    if (!in2) y = 0.0f;
    float ss = y * y;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float nv = y / (sqrtf(ss) + 1e-6f);
    const bool bad = in2 && !(fabsf(nv - v) <= 1e-8f + 1e-5f * fabsf(v));
    v = nv;
    if (!__any(bad)) break;
  }
  float wsum = in2 ? v : 0.0f;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) wsum += __shfl_xor(wsum, o, 64);
  const double w = in2 ? (double)(v / (wsum + 1e-6f)) : 0.0;
  // weighted Kabsch (scripts/SC2_PCR/common.py:7-45) over the k2 points
  double sm[7] = {w, w * px, w * py, w * pz, w * qx, w * qy, w * qz};
#pragma unroll
  for (int i = 0; i < 7; ++i) sm[i] = __shfl(wave_sum(sm[i]), 0, 64);
  const double den = sm[0] + 1e-6;
  const double ca[3] = {sm[1] / den, sm[2] / den, sm[3] / den}, cb[3] = {sm[4] / den, sm[5] / den, sm[6] / den};
  const double ax = px - ca[0], ay = py - ca[1], az = pz - ca[2];
  const double bx = (qx - cb[0]) * w, by = (qy - cb[1]) * w, bz = (qz - cb[2]) * w;
  double h[9] = {ax * bx, ax * by, ax * bz, ay * bx, ay * by, ay * bz, az * bx, az * by, az * bz};
#pragma unroll
  for (int i = 0; i < 9; ++i) h[i] = __shfl(wave_sum(h[i]), 0, 64);
  if (seed_h) {
    // round 6: the wave's part ends here - centroids and cross-covariance go to memory, k_seed_kabsch solves 64 seeds per wave
    // (lane = seed) and k_seed_fitness counts their inliers.  The fp64 Jacobi solver behind kabsch_rotation is ~3 500 instructions at
    // half rate that every lane of this wave would execute on the same numbers: ~45 % of the kernel's time for 1 / 64 of its lanes
    if (lane == 0) {
      double* o = seed_h + 16 * (size_t)s;
#pragma unroll
      for (int i = 0; i < 3; ++i) { o[i] = ca[i]; o[3 + i] = cb[i]; }
#pragma unroll
      for (int i = 0; i < 9; ++i) o[6 + i] = h[i];
    }
    return;
  }
  double H[3][3] = {{h[0], h[1], h[2]}, {h[3], h[4], h[5]}, {h[6], h[7], h[8]}};
  double R[3][3], t[3];
  seed_pose(ca, cb, H, R, t);
  if (lane == 0) write_T(Ts + 16 * (size_t)s, R, t);
  // fitness of this hypothesis over ALL correspondences (fp32 like the reference's einsum path)
  const float r00 = (float)R[0][0], r01 = (float)R[0][1], r02 = (float)R[0][2], r10 = (float)R[1][0], r11 = (float)R[1][1],
              r12 = (float)R[1][2], r20 = (float)R[2][0], r21 = (float)R[2][1], r22 = (float)R[2][2];
  const float t0 = (float)t[0], t1 = (float)t[1], t2 = (float)t[2];
  int cnt = 0;
  const float T_in = legacy ? 0.0f : sqrt_lt_threshold(inlier_thr);      // sqrtf(x) < inlier_thr  <=>  x < T_in
  for (int j = lane; j < n; j += 64) {
    const float x = src[3 * j], yv = src[3 * j + 1], z = src[3 * j + 2];
    const float dx = (__builtin_fmaf(r02, z, __builtin_fmaf(r01, yv, r00 * x)) + t0) - tgt[3 * j];       // roundings pinned (see sq_len)
    const float dy = (__builtin_fmaf(r12, z, __builtin_fmaf(r11, yv, r10 * x)) + t1) - tgt[3 * j + 1];
    const float dz = (__builtin_fmaf(r22, z, __builtin_fmaf(r21, yv, r20 * x)) + t2) - tgt[3 * j + 2];
    const float d2 = sq_len(dx, dy, dz);
    cnt += legacy ? sqrtf(d2) < inlier_thr : d2 < T_in;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) fitness[s] = (float)cnt;
}

// ---- first arg-max of the seed-wise fitness, then <= it_num refinement rounds (post_refinement)
__device__ __forceinline__ void d_refine(const float* __restrict__ src, const float* __restrict__ tgt, int n,
                                                 const float* __restrict__ Ts, const float* __restrict__ fitness,
                                                 int n_seed, float refine_thr, int it_num, float* __restrict__ Tout,
                                                 Sc2Ctl* __restrict__ ctl) {
  __shared__ double red[16 * 16];
  __shared__ float bf[16];
  __shared__ int bi[16];
  __shared__ double Tsh[12];
  // arg-max (first maximum)
  float bestf = -1.0f;
  int besti = 0x7FFFFFFF;
  for (int s = threadIdx.x; s < n_seed; s += 1024) {
    const float f = fitness[s];
    if (f > bestf || (f == bestf && s < besti)) { bestf = f; besti = s; }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float of = __shfl_down(bestf, o, 64);
    const int oi = __shfl_down(besti, o, 64);
    if (of > bestf || (of == bestf && oi < besti)) { bestf = of; besti = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bf[threadIdx.x >> 6] = bestf; bi[threadIdx.x >> 6] = besti; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (bf[w] > bestf || (bf[w] == bestf && bi[w] < besti)) { bestf = bf[w]; besti = bi[w]; }
    if (besti == 0x7FFFFFFF) besti = 0;
    ctl->best_seed = besti;
    ctl->best_fitness = bestf;
    for (int i = 0; i < 12; ++i) Tsh[i] = n_seed > 0 ? (double)Ts[16 * (size_t)besti + i] : ((i % 5 == 0) ? 1.0 : 0.0);
  }
  __syncthreads();
  int prev = 0;
  for (int it = 0; it < it_num; ++it) {
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = (float)Tsh[i];
    // pass 1: residuals in fp32 (as the reference), inlier count, weighted centroid sums in fp64
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = threadIdx.x; j < n; j += 1024) {
      const float x = src[3 * j], y = src[3 * j + 1], z = src[3 * j + 2];
      const float dx = (__builtin_fmaf(T[2], z, __builtin_fmaf(T[1], y, T[0] * x)) + T[3]) - tgt[3 * j];     // roundings pinned: both passes see the same inliers
      const float dy = (__builtin_fmaf(T[6], z, __builtin_fmaf(T[5], y, T[4] * x)) + T[7]) - tgt[3 * j + 1];
      const float dz = (__builtin_fmaf(T[10], z, __builtin_fmaf(T[9], y, T[8] * x)) + T[11]) - tgt[3 * j + 2];
      const float dist = sqrtf(sq_len(dx, dy, dz));
      if (dist < refine_thr) {
        const float r = dist / refine_thr;
        const double w = (double)(1.0f / (1.0f + r * r));
        s[0] += w; s[1] += w * x; s[2] += w * y; s[3] += w * z;
        s[4] += w * tgt[3 * j]; s[5] += w * tgt[3 * j + 1]; s[6] += w * tgt[3 * j + 2];
        s[7] += 1.0;
      }
    }
    {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
      for (int i = 0; i < 8; ++i) { double v = wave_sum(s[i]); if (lane == 0) red[wave * 16 + i] = v; }
      __syncthreads();
      if (threadIdx.x < 8) { double v = 0; for (int w = 0; w < 16; ++w) v += red[w * 16 + threadIdx.x]; red[threadIdx.x] = v; }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] = red[i];
      __syncthreads();
    }
    const int n_inl = (int)(s[7] + 0.5);
    if (n_inl == prev) break;             // abs(inlier_num - previous) < 1  (block-uniform)
    prev = n_inl;
    const double den = s[0] + 1e-6;
    const double ca[3] = {s[1] / den, s[2] / den, s[3] / den}, cb[3] = {s[4] / den, s[5] / den, s[6] / den};
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = threadIdx.x; j < n; j += 1024) {
      const float x = src[3 * j], y = src[3 * j + 1], z = src[3 * j + 2];
      const float dx = (__builtin_fmaf(T[2], z, __builtin_fmaf(T[1], y, T[0] * x)) + T[3]) - tgt[3 * j];     // roundings pinned: both passes see the same inliers
      const float dy = (__builtin_fmaf(T[6], z, __builtin_fmaf(T[5], y, T[4] * x)) + T[7]) - tgt[3 * j + 1];
      const float dz = (__builtin_fmaf(T[10], z, __builtin_fmaf(T[9], y, T[8] * x)) + T[11]) - tgt[3 * j + 2];
      const float dist = sqrtf(sq_len(dx, dy, dz));
      if (dist < refine_thr) {
        const float r = dist / refine_thr;
        const double w = (double)(1.0f / (1.0f + r * r));
        const double ax = x - ca[0], ay = y - ca[1], az = z - ca[2];
        const double bx = (tgt[3 * j] - cb[0]) * w, by = (tgt[3 * j + 1] - cb[1]) * w, bz = (tgt[3 * j + 2] - cb[2]) * w;
        h[0] += ax * bx; h[1] += ax * by; h[2] += ax * bz;
        h[3] += ay * bx; h[4] += ay * by; h[5] += ay * bz;
        h[6] += az * bx; h[7] += az * by; h[8] += az * bz;
      }
    }
    {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
      for (int i = 0; i < 9; ++i) { double v = wave_sum(h[i]); if (lane == 0) red[wave * 16 + i] = v; }
      __syncthreads();
      if (threadIdx.x == 0) {
        double H[3][3];
        for (int i = 0; i < 9; ++i) { double v = 0; for (int w = 0; w < 16; ++w) v += red[w * 16 + i]; H[i / 3][i % 3] = v; }
        double R[3][3];
        kabsch_rotation(H, R);
        for (int i = 0; i < 3; ++i) {
          Tsh[4 * i] = R[i][0]; Tsh[4 * i + 1] = R[i][1]; Tsh[4 * i + 2] = R[i][2];
          Tsh[4 * i + 3] = cb[i] - (R[i][0] * ca[0] + R[i][1] * ca[1] + R[i][2] * ca[2]);
        }
      }
      __syncthreads();
    }
  }
  if (threadIdx.x < 16) {
    const int i = threadIdx.x;
    Tout[i] = i < 12 ? (float)Tsh[i] : (i == 15 ? 1.0f : 0.0f);
  }
}

__device__ __forceinline__ void d_fill(float* p, int n, float v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---- batching: a launch handles up to SC2_CHUNK independent pairs, pair = blockIdx.z.  Every per-pair argument of
// the kernels above travels in the kernel-argument block (no descriptor upload, nothing to keep alive); grids are
// sized for the largest pair of the chunk and the surplus blocks of smaller pairs leave at once.  A pair's arithmetic
// is untouched, so batched results are bit-identical to single-pair calls.
constexpr int SC2_CHUNK = 16;
struct Sc2Pair {
  const float* src; const float* tgt;
  float* T_out; float* fitness;
  Sc2Ctl* ctl; float* v; float* y; float* score; int* seeds;
  unsigned long long* hard; unsigned long long* tight;
  int* knn; float* Ts; int* dom; int* rank;
  int* ptr_h; unsigned short* col_h; float* val_h;   // CSR of the hard graph (support of the first-order matrix)
  long long csr_cap;
  unsigned short* cnt;        // [n_seed][words * 64] second-order counts of the seeds of dense blocks (d_seed_dense)
  unsigned char* blk_dense;   // [ceil(n_seed / 64)]
  double* seed_h;             // [n_seed + 1][16]: centroids (3 + 3) and cross-covariance (9) of every seed's local problem
  int n, words, n_seed, k1, k2, n_part, col_chunk, num_iterations;
  float d, inlier_thr, nms_radius, refine_thr;
  int list_cap, dense_x;      // per-ctx diagnostics (eyoc_sc2pcr_set_shortlist_cap / _set_dense_threshold)
  int legacy;                 // eyoc_sc2pcr_select_kernels: bit 0 the round-5 CSR fill, bit 1 masks without the v_sqrt pre-test, bit 2 sqrtf in the NMS / fitness sweeps, bit 3 Kabsch + fitness inside k_seed_solve
};
struct Sc2Batch { Sc2Pair p[SC2_CHUNK]; };
static_assert(sizeof(Sc2Batch) <= 4000, "the batch descriptor travels as a kernel argument (4 KB limit)");

__global__ __launch_bounds__(256) void k_init(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < q.n) { q.v[i] = 1.0f; q.dom[i] = 0; q.rank[i] = 0; }
  if (i < (int)(sizeof(Sc2Ctl) / 4)) reinterpret_cast<int*>(q.ctl)[i] = 0;
}
__global__ __launch_bounds__(256) void k_sc_spmv(Sc2Batch B, int it) {
  const Sc2Pair& q = B.p[blockIdx.z];
  if (it >= q.num_iterations) return;
  d_sc_spmv(q.src, q.tgt, 1.0f / (q.d * q.d), q.ptr_h, q.col_h, q.val_h, q.n, q.v, q.y, q.ctl);
}
__global__ __launch_bounds__(1024) void k_csr_scan(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  d_csr_scan(q.ptr_h, q.n, q.csr_cap, q.ctl);
}
__global__ __launch_bounds__(256) void k_csr_count(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  d_csr_count(q.hard, q.n, q.words, q.ptr_h);
}
__global__ __launch_bounds__(256) void k_csr_fill(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  d_csr_fill(q.src, q.tgt, q.n, q.words, 1.0f / (q.d * q.d), q.hard, q.ptr_h, q.col_h, q.val_h, q.ctl, q.legacy & 1);
}
__global__ __launch_bounds__(1024) void k_sc_normalize(Sc2Batch B, int it) {
  const Sc2Pair& q = B.p[blockIdx.z];
  if (it >= q.num_iterations) return;
  d_sc_normalize(q.y, q.v, q.n, q.ctl);
}
__global__ __launch_bounds__(256) void k_nms(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  if ((int)blockIdx.x * 256 >= q.n || (int)blockIdx.y >= q.n_part) return;
  d_nms(q.src, q.v, q.n, q.nms_radius, q.col_chunk, q.dom, q.legacy & 4);
}
__global__ __launch_bounds__(256) void k_nms_score(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  d_nms_score(q.v, q.dom, q.n, q.score);
}
__global__ __launch_bounds__(256) void k_rank(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  if ((int)blockIdx.x * 256 >= q.n || (int)blockIdx.y >= q.n_part) return;
  d_rank(q.score, q.n, q.col_chunk, q.rank);
}
__global__ __launch_bounds__(256) void k_seeds(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  d_seeds(q.rank, q.n, q.n_seed, q.seeds);
}
__global__ __launch_bounds__(256) void k_masks(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  d_masks(q.src, q.tgt, q.n, q.words, q.d, q.hard, q.tight, q.legacy & 2);
}
__global__ __launch_bounds__(64) void k_seed_blocks(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  if ((int)blockIdx.x * 64 >= q.n_seed) return;
  d_seed_blocks(q.ptr_h, q.seeds, q.n, q.n_seed, q.ctl, q.blk_dense, q.dense_x);
}
template <int W>
__global__ __launch_bounds__(256) void k_seed_dense(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  if ((int)blockIdx.x * 64 >= q.n_seed || !q.blk_dense[blockIdx.x] || q.words > 4 * W || (W == 64 && q.words <= 128)) return;
  d_seed_dense<W>(q.hard, q.tight, q.n, q.words, q.seeds, q.n_seed, q.cnt);
}
// Two launches (round 6): the seeds of dense blocks find their row of counts in memory and need no candidate list - 17 KB of dynamic LDS
// instead of 33 at n = 8000, six workgroups per CU instead of three for a kernel that mostly waits (row copy, four passes over the
// row with a barrier each); the other seeds take the full layout.  Each launch leaves the other kind's seeds at once.
template <bool DENSE_ROWS>
__global__ __launch_bounds__(256) void k_seed_topk(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  if ((int)blockIdx.x >= q.n_seed) return;
  if ((q.blk_dense[blockIdx.x >> 6] != 0) != DENSE_ROWS) return;
  const unsigned short* row = DENSE_ROWS ? q.cnt + (size_t)blockIdx.x * q.words * 64 : nullptr;
  d_seed_topk(q.hard, q.tight, q.n, q.words, q.seeds, q.k1, q.knn, row, q.list_cap);
}
__global__ __launch_bounds__(256) void k_seed_solve(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  d_seed_solve(q.src, q.tgt, q.n, q.n_seed, q.knn, q.k1, q.k2, q.d, q.num_iterations, q.inlier_thr, q.Ts, q.fitness, q.legacy & 4,
               (q.legacy & 8) ? nullptr : q.seed_h);
}
__global__ __launch_bounds__(64) void k_seed_kabsch(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  d_seed_kabsch(q.seed_h, q.n_seed, q.Ts, q.fitness);
}
__global__ __launch_bounds__(256) void k_seed_fitness(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  if ((int)blockIdx.x * (256 * FIT_PTS) >= q.n) return;
  if (q.legacy & 4) d_seed_fitness<true>(q.src, q.tgt, q.n, q.n_seed, q.Ts, q.inlier_thr, q.fitness);
  else d_seed_fitness<false>(q.src, q.tgt, q.n, q.n_seed, q.Ts, q.inlier_thr, q.fitness);
}
__global__ __launch_bounds__(1024) void k_refine(Sc2Batch B) {
  const Sc2Pair& q = B.p[blockIdx.z];
  d_refine(q.src, q.tgt, q.n, q.Ts, q.fitness, q.n_seed, q.refine_thr, 20, q.T_out, q.ctl);
}

struct Plan {
  int n, words, n_seed, k1, k2;
  int n_part, col_chunk;   // column ranges of the lane-per-row sweeps (NMS, rank)
  size_t off_ctl, off_v, off_y, off_score, off_seeds, off_hard, off_tight, off_knn, off_Ts, off_int, total;
  size_t off_ptr_h, off_col_h, off_val_h, off_cnt, off_blk, off_seed_h;
  long long csr_cap;   // entries each CSR list can hold: a quarter of the N^2 pairs (denser graphs sweep densely)
};

Plan make_plan(int n, const eyoc_sc2pcr_params* p) {
  Plan pl;
  pl.n = n;
  pl.words = (n + 63) / 64;
  pl.n_seed = (int)((double)n * (double)p->ratio);   // int(num_corr * self.ratio)
  pl.k1 = p->k1;
  pl.k2 = p->k2;
  if (pl.k1 > n) { pl.k1 = 4; pl.k2 = 4; }           // SC2_PCR.py:76-78
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes); return r; };
  pl.off_ctl = take(sizeof(Sc2Ctl));
  pl.off_v = take((size_t)n * 4);
  pl.off_y = take((size_t)n * 4);
  pl.off_score = take((size_t)n * 4);
  pl.off_seeds = take((size_t)(pl.n_seed + 1) * 4);
  pl.off_hard = take((size_t)n * pl.words * 8);
  pl.off_tight = take((size_t)n * pl.words * 8);
  pl.off_knn = take((size_t)(pl.n_seed + 1) * K1_MAX * 4);
  pl.off_Ts = take((size_t)(pl.n_seed + 1) * 16 * 4);
  // enough workgroups for the whole chip: rows / 256 x column ranges >= ~1024, ranges of at least 64 columns
  const int row_blocks = (n + 255) / 256;
  int parts = (1024 + row_blocks - 1) / row_blocks;
  if (parts > (n + 63) / 64) parts = (n + 63) / 64;
  if (parts < 1) parts = 1;
  pl.col_chunk = (n + parts - 1) / parts;
  pl.n_part = (n + pl.col_chunk - 1) / pl.col_chunk;
  pl.off_int = take((size_t)2 * n * 4);          // NMS domination flags, ranks
  pl.csr_cap = (long long)n * n / 4 + 64;
  pl.off_ptr_h = take((size_t)(n + 1) * 4);
  pl.off_col_h = take((size_t)pl.csr_cap * 2);
  pl.off_val_h = take((size_t)pl.csr_cap * 4);
  pl.off_cnt = take((size_t)(pl.n_seed + 1) * pl.words * 64 * 2);
  pl.off_blk = take((size_t)(pl.n_seed + 64) / 64 + 1);
  pl.off_seed_h = take((size_t)(pl.n_seed + 1) * 16 * 8);
  pl.total = o + 256;
  return pl;
}

}  // namespace

extern "C" {

// Diagnostics, per ctx (tests run both paths of each and compare): the seed top-k's short list holds at most `cap` entries (0 = every
// seed takes the histogram path, default and maximum 1024); a block of 64 seeds is "dense" when its hard rows hold >= x n candidates
// (default 0 since round 6 - measured 3.84 ms per 16-pair step against 3.94 at x = 1 and 3.96 at x = 2; a negative x switches the dense-block kernel off).  Both return the previous value; neither changes any result.
int eyoc_sc2pcr_set_shortlist_cap(eyoc_ctx* ctx, int cap) {
  if (!ctx) return -1;
  const int prev = ctx->sc2_list_cap;
  if (cap >= 0) ctx->sc2_list_cap = cap > 1024 ? 1024 : cap;
  return prev;
}
int eyoc_sc2pcr_set_dense_threshold(eyoc_ctx* ctx, int x) {
  if (!ctx) return -1;
  const int prev = ctx->sc2_dense_x;
  ctx->sc2_dense_x = x;
  return prev;
}

int eyoc_sc2pcr_select_kernels(eyoc_ctx* ctx, int legacy_bits) {
  if (!ctx) return -1;
  const int prev = ctx->sc2_legacy;
  if (legacy_bits >= 0 && legacy_bits <= 15) ctx->sc2_legacy = legacy_bits;
  return prev;
}

size_t eyoc_sc2pcr_workspace_bytes(int n, const eyoc_sc2pcr_params* params) {
  if (!params || n < 1 || n > MAX_N) return 0;
  return make_plan(n, params).total;
}

// one chunk of <= SC2_CHUNK pairs: ~70 launches whatever the number of pairs
static int sc2pcr_chunk(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int32_t* seg_host, int n_pairs,
                        const eyoc_sc2pcr_params* params, float* T_dev, float* fitness_dev, int fitness_stride, char* ws,
                        size_t slice, hipStream_t st) {
  Sc2Batch B;
  int n_max = 0, part_max = 0, seed_max = 0, words_max = 0, it_max = 0;
  for (int c = 0; c < SC2_CHUNK; ++c) {
    const int b = c < n_pairs ? c : 0;   // unused slots repeat pair 0 (never launched: grid.z = n_pairs)
    const int s0 = seg_host[b], n = seg_host[b + 1] - s0;
    const eyoc_sc2pcr_params* p = &params[b];
    const Plan pl = make_plan(n, p);
    char* w = ws + slice * (size_t)b;
    Sc2Pair& q = B.p[c];
    q.src = src_dev + 3 * (size_t)s0; q.tgt = tgt_dev + 3 * (size_t)s0;
    q.T_out = T_dev + 16 * (size_t)b; q.fitness = fitness_dev + (size_t)b * fitness_stride;
    q.ctl = (Sc2Ctl*)(w + pl.off_ctl); q.v = (float*)(w + pl.off_v); q.y = (float*)(w + pl.off_y);
    q.score = (float*)(w + pl.off_score); q.seeds = (int*)(w + pl.off_seeds);
    q.hard = (unsigned long long*)(w + pl.off_hard); q.tight = (unsigned long long*)(w + pl.off_tight);
    q.knn = (int*)(w + pl.off_knn); q.Ts = (float*)(w + pl.off_Ts);
    q.dom = (int*)(w + pl.off_int); q.rank = q.dom + n;
    q.ptr_h = (int*)(w + pl.off_ptr_h); q.col_h = (unsigned short*)(w + pl.off_col_h);
    q.val_h = (float*)(w + pl.off_val_h); q.csr_cap = pl.csr_cap;
    q.cnt = (unsigned short*)(w + pl.off_cnt); q.blk_dense = (unsigned char*)(w + pl.off_blk);
    q.seed_h = (double*)(w + pl.off_seed_h);
    q.n = n; q.words = pl.words; q.n_seed = pl.n_seed; q.k1 = pl.k1; q.k2 = pl.k2; q.n_part = pl.n_part;
    q.col_chunk = pl.col_chunk; q.num_iterations = p->num_iterations;
    q.d = p->d_thre; q.inlier_thr = p->inlier_threshold; q.nms_radius = p->nms_radius;
    // the reference refines with 0.10 m for its 3DMatch setting and 1.2 m otherwise (SC2_PCR.py:254-257)
    q.refine_thr = p->inlier_threshold == 0.10f ? 0.10f : 1.2f;
    q.list_cap = ctx->sc2_list_cap; q.dense_x = ctx->sc2_dense_x; q.legacy = ctx->sc2_legacy;
    if (c < n_pairs) {
      n_max = n > n_max ? n : n_max; part_max = pl.n_part > part_max ? pl.n_part : part_max;
      seed_max = pl.n_seed > seed_max ? pl.n_seed : seed_max; words_max = pl.words > words_max ? pl.words : words_max;
      it_max = p->num_iterations > it_max ? p->num_iterations : it_max;
    }
  }
  const unsigned Z = (unsigned)n_pairs;
  const int rb = cdiv(n_max, 256);
  // leading eigenvector of the first-order compatibility matrix (power iteration from all-ones); a pair whose own
  // num_iterations is below the chunk's maximum sits the surplus sweeps out
  hipLaunchKernelGGL(k_init, dim3(rb, 1, Z), dim3(256), 0, st, B);
  // the two compatibility graphs as bit matrices, then as CSR lists (or ctl->dense when they do not fit)
  hipLaunchKernelGGL(k_masks, dim3(cdiv((long long)words_max * words_max, 4), 1, Z), dim3(256), 0, st, B);
  hipLaunchKernelGGL(k_csr_count, dim3(cdiv(n_max, 4), 1, Z), dim3(256), 0, st, B);
  hipLaunchKernelGGL(k_csr_scan, dim3(1, 1, Z), dim3(1024), 0, st, B);
  hipLaunchKernelGGL(k_csr_fill, dim3(cdiv(n_max, 4), 1, Z), dim3(256), 0, st, B);
  // leading eigenvector of the first-order compatibility matrix (power iteration from all-ones); a pair whose own
  // num_iterations is below the chunk's maximum sits the surplus sweeps out.  A pair whose graph overflowed the CSR arrays sweeps densely
  for (int it = 0; it < it_max; ++it) {
    hipLaunchKernelGGL(k_sc_spmv, dim3(cdiv(n_max, 4 * SPMV_ROWS), 1, Z), dim3(256), 0, st, B, it);
    hipLaunchKernelGGL(k_sc_normalize, dim3(1, 1, Z), dim3(1024), 0, st, B, it);
  }
  // seeds: NMS on the eigenvector in source space, stable top-n_seed
  hipLaunchKernelGGL(k_nms, dim3(rb, part_max, Z), dim3(256), 0, st, B);
  hipLaunchKernelGGL(k_nms_score, dim3(rb, 1, Z), dim3(256), 0, st, B);
  hipLaunchKernelGGL(k_rank, dim3(rb, part_max, Z), dim3(256), 0, st, B);
  hipLaunchKernelGGL(k_seeds, dim3(rb, 1, Z), dim3(256), 0, st, B);
  // second-order measure per seed, two-stage consensus, hypotheses
  const size_t dyn = ((size_t)n_max * 2 + 15) / 16 * 16 + (size_t)words_max * 8 + (size_t)words_max * 64 * 2;   // row, seed row, candidates
  if (dyn > 48 * 1024 && !ctx->sc2_attr_set) {   // beyond the default dynamic-LDS allowance (n > ~12000)
    EYOC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_seed_topk<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    ctx->sc2_attr_set = true;
  }
  hipLaunchKernelGGL(k_seed_blocks, dim3(cdiv(seed_max, 64), 1, Z), dim3(64), 0, st, B);
  if (words_max <= 128) hipLaunchKernelGGL(k_seed_dense<32>, dim3(cdiv(seed_max, 64), DENSE_SPLITS, Z), dim3(256), 0, st, B);
  else {      // pairs of a chunk may fall on either side of 128 words: each kernel takes the pairs of its size class
    hipLaunchKernelGGL(k_seed_dense<32>, dim3(cdiv(seed_max, 64), DENSE_SPLITS, Z), dim3(256), 0, st, B);
    hipLaunchKernelGGL(k_seed_dense<64>, dim3(cdiv(seed_max, 64), DENSE_SPLITS, Z), dim3(256), 0, st, B);
  }
  const size_t dyn_rows = ((size_t)n_max * 2 + 15) / 16 * 16 + (size_t)words_max * 8;      // row, seed row (MAX_N: 34 KB)
  hipLaunchKernelGGL(k_seed_topk<true>, dim3(seed_max, 1, Z), dim3(256), dyn_rows, st, B);
  hipLaunchKernelGGL(k_seed_topk<false>, dim3(seed_max, 1, Z), dim3(256), dyn, st, B);
  hipLaunchKernelGGL(k_seed_solve, dim3(cdiv(seed_max, 4), 1, Z), dim3(256), 0, st, B);
  if (!(ctx->sc2_legacy & 8)) {       // round 6: the 3 x 3 solves and the inlier counts lane-per-seed (bit 3 set: all in k_seed_solve, the round-5 form)
    hipLaunchKernelGGL(k_seed_kabsch, dim3(cdiv(seed_max, 64), 1, Z), dim3(64), 0, st, B);
    hipLaunchKernelGGL(k_seed_fitness, dim3(cdiv(n_max, 256 * FIT_PTS), FIT_SPLIT, Z), dim3(256), 0, st, B);
  }
  hipLaunchKernelGGL(k_refine, dim3(1, 1, Z), dim3(1024), 0, st, B);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

static int sc2pcr_check(int n, const eyoc_sc2pcr_params* p, const float* fitness_dev) {
  EYOC_REQUIRE(n >= 8 && n <= MAX_N, EYOC_ERR_INVALID, "eyoc_sc2pcr: n %d not in [8, %d]", n, MAX_N);
  EYOC_REQUIRE(n <= p->max_points, EYOC_ERR_INVALID, "eyoc_sc2pcr: n %d exceeds max_points %d (truncate first)", n, p->max_points);
  const Plan pl = make_plan(n, p);
  EYOC_REQUIRE(pl.k1 >= 1 && pl.k1 <= K1_MAX && pl.k2 >= 1 && pl.k2 <= pl.k1 && pl.k2 <= K2_MAX, EYOC_ERR_INVALID,
               "eyoc_sc2pcr: need 1 <= k2 <= k1 <= %d (got k1 %d k2 %d)", K1_MAX, pl.k1, pl.k2);
  EYOC_REQUIRE(pl.n_seed >= 1 && fitness_dev, EYOC_ERR_INVALID, "eyoc_sc2pcr: ratio %g gives no seeds for n %d", p->ratio, n);
  return EYOC_OK;
}

int eyoc_sc2pcr(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, int n, const eyoc_sc2pcr_params* p, float* T_dev,
                float* fitness_dev, void* ws, size_t ws_bytes, void* stream) {
  EYOC_REQUIRE(ctx && src_dev && tgt_dev && p && T_dev && ws, EYOC_ERR_INVALID, "eyoc_sc2pcr: NULL argument");
  int rc = sc2pcr_check(n, p, fitness_dev);
  if (rc) return rc;
  const Plan pl = make_plan(n, p);
  EYOC_REQUIRE(ws_bytes >= pl.total && ((uintptr_t)ws & 255) == 0, EYOC_ERR_WORKSPACE,
               "eyoc_sc2pcr: workspace %zu < required %zu bytes (256-byte aligned)", ws_bytes, pl.total);
  const int32_t seg[2] = {0, n};
  return sc2pcr_chunk(ctx, src_dev, tgt_dev, seg, 1, p, T_dev, fitness_dev, pl.n_seed, (char*)ws, 0, (hipStream_t)stream);
}

// A batch of independent pairs (the loop of lib/trainer.py:1157-1166, which the reference leaves sequential with a
// "ToDo: ... batched and parallelized"): SC2_CHUNK pairs per launch (blockIdx.z = pair), ~70 launches per chunk instead
// of ~70 per pair.  Results are bit-identical to eyoc_sc2pcr per pair.
//   src/tgt: pairs back to back, pair b = rows [seg[b], seg[b+1]) (HOST array of n_pairs + 1 ints);
//   params[b]: per pair (the seed count int(ratio * n) depends on n);  T_dev f32 [n_pairs,16];
//   fitness_dev f32 [n_pairs, fitness_stride] (row b holds the n_seed_b seedwise values).
size_t eyoc_sc2pcr_batched_workspace_bytes(int max_n, const eyoc_sc2pcr_params* params) {
  return eyoc_sc2pcr_batched_workspace_bytes_n(max_n, SC2_CHUNK, params);
}

// ... for a call of n_pairs pairs: min(n_pairs, 16) slices (a single pair of 8000 points needs 0.11 GB, not 1.8)
size_t eyoc_sc2pcr_batched_workspace_bytes_n(int max_n, int n_pairs, const eyoc_sc2pcr_params* params) {
  const size_t one = eyoc_sc2pcr_workspace_bytes(max_n, params);
  const int slices = n_pairs < 1 ? 1 : n_pairs < SC2_CHUNK ? n_pairs : SC2_CHUNK;
  return one ? align_up(one) * slices : 0;
}

int eyoc_sc2pcr_batched(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int32_t* seg_host, int n_pairs,
                        const eyoc_sc2pcr_params* params, float* T_dev, float* fitness_dev, int fitness_stride, void* ws,
                        size_t ws_bytes, void* stream) {
  EYOC_REQUIRE(ctx && src_dev && tgt_dev && seg_host && params && T_dev && fitness_dev && ws, EYOC_ERR_INVALID,
               "eyoc_sc2pcr_batched: NULL argument");
  EYOC_REQUIRE(n_pairs >= 1, EYOC_ERR_INVALID, "eyoc_sc2pcr_batched: n_pairs %d", n_pairs);
  size_t slice = 0;
  for (int b = 0; b < n_pairs; ++b) {
    const int n = seg_host[b + 1] - seg_host[b];
    int rc = sc2pcr_check(n, &params[b], fitness_dev);
    if (rc) return rc;
    const size_t need = make_plan(n, &params[b]).total;
    EYOC_REQUIRE((int)(params[b].ratio * n) <= fitness_stride, EYOC_ERR_INVALID,
                 "eyoc_sc2pcr_batched: fitness_stride %d too small for pair %d", fitness_stride, b);
    slice = need > slice ? need : slice;
  }
  slice = align_up(slice);
  const size_t slices = n_pairs < SC2_CHUNK ? n_pairs : SC2_CHUNK;
  EYOC_REQUIRE(ws_bytes >= slice * slices && ((uintptr_t)ws & 255) == 0, EYOC_ERR_WORKSPACE,
               "eyoc_sc2pcr_batched: workspace %zu < required %zu bytes (256-byte aligned)", ws_bytes, slice * slices);
  for (int b0 = 0; b0 < n_pairs; b0 += SC2_CHUNK) {   // chunks run back to back on `stream` and reuse the workspace slices
    const int nc = n_pairs - b0 < SC2_CHUNK ? n_pairs - b0 : SC2_CHUNK;
    int rc = sc2pcr_chunk(ctx, src_dev, tgt_dev, seg_host + b0, nc, params + b0, T_dev + 16 * (size_t)b0,
                          fitness_dev + (size_t)b0 * fitness_stride, fitness_stride, (char*)ws, slice, (hipStream_t)stream);
    if (rc) return rc;
  }
  return EYOC_OK;
}

}  // extern "C"
