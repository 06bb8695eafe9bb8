// Brute-force 1-nearest-neighbour in feature space (lib/eval.py:18-48, lib/metrics.py:22-29 of the
// reference).  One lane owns one query row (its C features live in VGPRs); the targets are transposed once
// (channel-major) and read through the scalar cache as SGPR operands of packed-fp32 ops, so the kernel is pure
// VALU work: 3 packed ops per (query, 2 targets, channel).  The four waves of a block walk interleaved
// groups of targets and grid.y splits the target range; partial minima are merged with one
// 64-bit atomicMin on (distance bits << 32 | index), which also implements "ties go to the lowest
// index" independently of how the work was split.
//
// Arithmetic contract (bit-exact with oracle/matching.py): d = a - b; s = d * d; acc = acc + s in
// fp32, channels in order, no FMA contraction.
#include "common.h"

namespace {

constexpr int MAX_SEG = 128;
struct SegArgs {
  int a[MAX_SEG + 1];
  int b[MAX_SEG + 1];
  int ld[MAX_SEG];              // targets of the segment rounded up to a whole group (row length of its transposed block)
  long long bt_off[MAX_SEG];    // float offset of the segment's transposed block
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int TGROUP = 8;      // targets per inner iteration (four packed pairs)
constexpr int SPLIT_ALIGN = 32;  // split boundaries: every wave of a block starts on a whole group

// targets of segment s, channel-major and padded to a multiple of TGROUP: Bt[bt_off[s] + c * ld[s] + j].
// Padding columns are zero and never compared.
__global__ void knn_transpose_targets(const float* __restrict__ B, SegArgs seg, int C, float* __restrict__ Bt) {
  const int s = blockIdx.z;
  const int b0 = seg.b[s], nb = seg.b[s + 1] - b0, ld = seg.ld[s];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (c4, j): consecutive threads = consecutive targets
  const int c4n = C / 4;
  if (i >= ld * c4n) return;
  const int j = i % ld, c4 = i / ld;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (j < nb) v = *reinterpret_cast<const float4*>(B + (size_t)(b0 + j) * C + 4 * c4);
  float* dst = Bt + seg.bt_off[s] + (size_t)(4 * c4) * ld + j;
  dst[0] = v.x; dst[ld] = v.y; dst[2 * (size_t)ld] = v.z; dst[3 * (size_t)ld] = v.w;
}

// Lane = query (C features in VGPRs); the targets are wave-uniform, so they come through the SCALAR cache
// (s_load of eight consecutive targets of one channel) and feed the packed-fp32 ops as SGPR pairs: no LDS
// traffic at all (the LDS-broadcast version of this kernel was bound by ds_read bandwidth, not by the VALU).
// Two targets share one v_pk_* instruction; each (query, target) still sees the scalar contract.
// DOT: the value minimised is -<a, b> (one packed FMA per channel and target pair) instead of the squared distance:
// the arg-max of the inner product of util/transform_estimation.py:131-133 without its [N0, N1] matrix.
// MODE 2 (dist_type 2, "GemmL2"): the quantity minimised is the reference's sqrt(2 - 2 S + 1e-6) of
// scripts/SC2_PCR/SC2_PCR.py:296-298 with S the same FMA chain as DOT, every later step rounded separately in fp32; a NaN
// (S > 1 + 5e-7: un-normalised descriptors) is smaller than every number and the first one wins, like torch.argmin.
// TOP2: also track the second smallest distance (Lowe's ratio test of the label generator,
// lib/trainer.py:1060-1072); the target range is then NOT split over blocks (a 64-bit atomicMin cannot merge a
// runner-up) and the four waves of the block merge their partial results through LDS.
template <int C, bool TOP2, int MODE>
__global__ __launch_bounds__(256) void knn1_kernel(const float* __restrict__ A, const float* __restrict__ Bt,
                                                   SegArgs seg, int split_len, int dist_type,
                                                   unsigned long long* __restrict__ best, float* __restrict__ second,
                                                   const int* __restrict__ only = nullptr, const int* __restrict__ only_cnt = nullptr) {
#pragma clang fp contract(off)
  constexpr bool DOT = MODE != 0;
  const int s = blockIdx.z;
  const int a0 = seg.a[s], na = seg.a[s + 1] - a0;
  const int nb = seg.b[s + 1] - seg.b[s], ld = seg.ld[s];
  const int t_begin = blockIdx.y * split_len;
  const int t_end = min(nb, t_begin + split_len);
  if (t_begin >= t_end) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // query tiles: one per block - or, in the second pass of the MFMA pre-filter, a few blocks per segment that walk the
  // dense list of the rows it could not decide (any number of them, usually a handful)
  for (int q0 = blockIdx.x * 64;; q0 += gridDim.x * 64) {
  int q = q0 + lane;
  bool q_ok = q < na;
  if (only) {
    const int cnt = only_cnt[s];
    if (q0 >= cnt) return;                                           // the whole block: its four waves share the queries
    q_ok = q0 + lane < cnt;
    q = q_ok ? only[a0 + q0 + lane] : 0;
  } else if (q0 >= na) {
    return;
  }
  f32x2 a[C];   // {a_c, a_c}: the query's feature in both halves of a packed operand
  {
    const float4* src = reinterpret_cast<const float4*>(A + (size_t)(a0 + (q_ok ? q : 0)) * C);
#pragma unroll
    for (int i = 0; i < C / 4; ++i) {
      const float4 v = src[i];
      a[4 * i] = f32x2{v.x, v.x}; a[4 * i + 1] = f32x2{v.y, v.y}; a[4 * i + 2] = f32x2{v.z, v.z}; a[4 * i + 3] = f32x2{v.w, v.w};
    }
  }
  float best_d = __builtin_inff(), second_d = __builtin_inff();
  int best_j = 0x7FFFFFFF;
  bool any = false;
  // the four waves of the block take interleaved groups of TGROUP targets
  const float* bt = Bt + seg.bt_off[s];
  for (int t = t_begin + wave * TGROUP; t < t_end; t += 4 * TGROUP) {
    f32x2 acc[TGROUP / 2];
#pragma unroll
    for (int k = 0; k < TGROUP / 2; ++k) acc[k] = f32x2{0.0f, 0.0f};
    const float* col = bt + t;   // wave-uniform address
    // Channels in batches of CB (eight SGPRs per channel).  Scalar loads return out of order, so a wait is always
    // "all of them": the next batch is therefore issued right AFTER the wait for the current one (behind the first
    // packed op that needs it) and its latency hides behind the rest of the current batch.
    constexpr int CB = C >= 8 ? 4 : C / 2;
    auto load_batch = [&](int cb, f32x2 (&b)[CB][TGROUP / 2]) {
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const float4 lo = *reinterpret_cast<const float4*>(col + (size_t)(cb + c) * ld);
        const float4 hi = *reinterpret_cast<const float4*>(col + (size_t)(cb + c) * ld + 4);
        b[c][0] = f32x2{lo.x, lo.y}; b[c][1] = f32x2{lo.z, lo.w}; b[c][2] = f32x2{hi.x, hi.y}; b[c][3] = f32x2{hi.z, hi.w};
      }
    };
    // the four target pairs side by side, so that dependent packed ops are never back to back (a wait state each)
    auto channel = [&](int c, const f32x2 (&b)[TGROUP / 2]) {
      if (DOT) {
#pragma unroll
        for (int k = 0; k < TGROUP / 2; ++k) acc[k] = __builtin_elementwise_fma(a[c], -b[k], acc[k]);
        return;
      }
      f32x2 d[TGROUP / 2];
#pragma unroll
      for (int k = 0; k < TGROUP / 2; ++k) d[k] = a[c] - b[k];
#pragma unroll
      for (int k = 0; k < TGROUP / 2; ++k) d[k] = d[k] * d[k];
#pragma unroll
      for (int k = 0; k < TGROUP / 2; ++k) acc[k] = acc[k] + d[k];
    };
    f32x2 b0[CB][TGROUP / 2], b1[CB][TGROUP / 2];
    load_batch(0, b0);
#pragma unroll
    for (int cb = 0; cb < C; cb += 2 * CB) {
      __builtin_amdgcn_sched_barrier(0);
      channel(cb, b0[0]);                               // forces the wait for batch b0
      __builtin_amdgcn_sched_barrier(0);
      load_batch(cb + CB, b1);                          // C is a multiple of 2 CB
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 1; c < CB; ++c) channel(cb + c, b0[c]);
      __builtin_amdgcn_sched_barrier(0);
      channel(cb + CB, b1[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (cb + 2 * CB < C) load_batch(cb + 2 * CB, b0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 1; c < CB; ++c) channel(cb + CB + c, b1[c]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < TGROUP; ++e) {
      // branch-free on purpose: with branches here LLVM sinks the arithmetic of targets 2..7 below the first
      // compare and keeps every scalar-loaded operand alive until then (SGPR spills)
      float v = (e & 1) ? acc[e / 2].y : acc[e / 2].x;
      v = dist_type == 1 ? sqrtf(v + 1e-7f) : v;
      if (MODE == 2) {          // acc = -S exactly (the negated chain rounds symmetrically)
        const float w = (2.0f - 2.0f * (-v)) + 1e-6f;
        const float d = sqrtf(w);
        v = d != d ? -1.0f : d;                                        // NaN first: below every distance
      }
      const bool valid = t + e < t_end;
      const bool better = valid & (v < best_d);
      if (TOP2) {   // the displaced best, or a value between the two, becomes the runner-up (NaNs never enter)
        const float cand = better ? best_d : v;
        second_d = (valid & (cand < second_d)) ? cand : second_d;
      }
      best_d = better ? v : best_d;
      best_j = better ? t + e : best_j;
      any = any | better;
    }
  }
  if (TOP2) {
    __shared__ float sd1[4][64], sd2[4][64];
    __shared__ int sj[4][64];
    sd1[wave][lane] = best_d; sd2[wave][lane] = second_d; sj[wave][lane] = any ? best_j : 0x7FFFFFFF;
    __syncthreads();
    if (wave == 0 && q_ok) {
      float d1 = sd1[0][lane], d2 = sd2[0][lane];
      int j1 = sj[0][lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float e1 = sd1[w][lane], e2 = sd2[w][lane];
        const int k1 = sj[w][lane];
        const bool take = (e1 < d1) | ((e1 == d1) & (k1 < j1));   // smaller distance, ties to the lower index
        const float lose = take ? d1 : e1;                          // the loser of the two bests is a runner-up candidate
        d1 = take ? e1 : d1;
        j1 = take ? k1 : j1;
        float r = d2 < e2 ? d2 : e2;
        d2 = lose < r ? lose : r;
      }
      if (j1 != 0x7FFFFFFF) best[a0 + q] = ((unsigned long long)__float_as_uint(d1) << 32) | (unsigned)j1;
      second[a0 + q] = d2;
    }
    return;
  }
  if (q_ok && any) {
    unsigned int bits = __float_as_uint(best_d);
    if (DOT) bits = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);   // negative values too: order-preserving key
    unsigned long long packed = ((unsigned long long)bits << 32) | (unsigned)best_j;
    atomicMin(&best[a0 + q], packed);
  }
  if (!only) return;
  }
}

__global__ void knn1_unpack(const unsigned long long* __restrict__ best, int n, long long* __restrict__ idx,
                            float* __restrict__ dist, int dot) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long p = best[i];
  if (p == ~0ull) {  // every candidate compared false (NaN features) or the target set was empty
    if (idx) idx[i] = 0;
    if (dist) dist[i] = __builtin_nanf("");
    return;
  }
  if (idx) idx[i] = (long long)(unsigned)(p & 0xFFFFFFFFull);
  unsigned int bits = (unsigned)(p >> 32);
  if (dot) bits = (bits & 0x80000000u) ? (bits & 0x7FFFFFFFu) : ~bits;   // undo the key, then negate: the inner product
  if (dist) {
    const float v = __uint_as_float(bits);
    dist[i] = dot == 1 ? -v : dot == 2 ? (v < 0.0f ? __builtin_nanf("") : v) : v;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// MFMA pre-filter for the plain index query (SquareL2, no distances asked for, enough rows to fill the chip).
//
// The contract above costs 3 VALU ops per (query, target, channel) - 3.3 ms for the bench's 64 x 5000 x 5000 x 32.  The
// arg-min, however, is almost always decided by a far cheaper score: s(i, j) = |b_j|^2 - 2 <a_i, b_j> with the inner
// product on the fp16 matrix pipe at fp32 accuracy - three v_mfma_f32_16x16x32_f16 on hi / lo-split operands, ONE
// instruction spanning all 32 channels (rows are scaled by a power of two first so that both halves of the split are
// accurate whatever the magnitude of the features; the first version used nine v_mfma_f32_16x16x4_f32: 288 instead
// of 48 matrix cycles per 16 x 16 tile, 1.25 ms instead of 0.45 for the bench's query).  With
// |s(i, j) - (d(i, j) - |a_i|^2)| <= e for every j - d the contract's fp32 value - the contract's arg-min j* satisfies
// s(j*) <= min_j s + 2 e.  So every wave tracks, per query, the smallest score with its index AND the second smallest
// score: if the runner-up is farther than 2 e, the index is final (exact ties included: they would both be within
// 2 e); otherwise the row is flagged and the exact kernel above recomputes it (waves without a flagged row exit at
// once).  e = 128 u (|a_i| + max_j |b_j|)^2, u = 2^-24, covers the rounding of both evaluations with a wide margin
// (the split products err by <= 2^-20 |a||b|, the norm and the contract's own 96 roundings by <= 64 u (|a| + |b|)^2).  Rows with non-finite scores are flagged too.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
struct SegMfma { long long bm_off[MAX_SEG]; };
constexpr int MF_ROWS = 64;                    // query rows per wave (4 MFMA row tiles)

// power of two that lifts a row's largest |element| into [128, 256): both halves of the fp16 split of the scaled row are
// then as accurate as they can be (a row of all zeros, or with a non-finite element, keeps scale 1)
__device__ inline float split_scale(float maxabs) {
  if (!(maxabs > 0.0f) || !(maxabs < __builtin_inff())) return 1.0f;
  int e;
  (void)frexpf(maxabs, &e);                                          // maxabs = m 2^e, m in [0.5, 1)
  return ldexpf(1.0f, 8 - e);
}

// B operand of v_mfma_f32_16x16x32_f16 per tile of 16 targets, 3 x 64 float4: [0] lane (g, j): the fp16 hi halves of
// channels 8 g .. 8 g + 7 of target 16 t + j, scaled by the target's split_scale; [1] the lo halves; [2] (|b_j|^2,
// 1 / scale_j, 0, 0) (padding targets: norm +inf, everything else 0)
template <int C>
__global__ void knn_pack_targets_mfma(const float* __restrict__ B, SegArgs seg, SegMfma sm, float* __restrict__ Bm,
                                      int* __restrict__ bmax_bits, int zero_norm) {
  static_assert(C == 32, "one 16x16x32 MFMA spans the 32 channels");
  const int s = blockIdx.z;
  const int b0 = seg.b[s], nb = seg.b[s + 1] - b0;
  const int n16 = (nb + 15) / 16;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;        // (tile, lane)
  if (idx >= n16 * 64) return;
  const int t = idx >> 6, l = idx & 63, j = 16 * t + (l & 15), g = l >> 4;
  float4* dst = reinterpret_cast<float4*>(Bm + sm.bm_off[s]) + (size_t)t * 3 * 64 + l;
  const bool ok = j < nb;
  const float* row = B + (size_t)(b0 + (ok ? j : 0)) * C;
  float n2 = 0.f, mx = 0.f;
  float mine[8];
#pragma unroll
  for (int q = 0; q < C / 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(row + 4 * q);
    n2 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    if ((q >> 1) == g) { mine[4 * (q & 1)] = v.x; mine[4 * (q & 1) + 1] = v.y; mine[4 * (q & 1) + 2] = v.z; mine[4 * (q & 1) + 3] = v.w; }
  }
  if (!(n2 == n2)) mx = __builtin_nanf("");                            // a NaN element: fmaxf would have dropped it
  const float sc = split_scale(mx);
  half8_t hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = ok ? mine[i] * sc : 0.0f;
    const _Float16 h = (_Float16)x;
    hi[i] = h;
    lo[i] = (_Float16)(x - (float)h);
  }
  dst[0] = __builtin_bit_cast(float4, hi);
  dst[64] = __builtin_bit_cast(float4, lo);
  dst[128] = make_float4(ok ? (zero_norm ? 0.0f : n2) : __builtin_inff(), ok ? 1.0f / sc : 0.0f, 0.f, 0.f);   // zero_norm: score -2 <a, b>
  // the segment's largest |b|^2: one atomic per wave (= per tile of 16 targets), not per target - 5000 atomics on one word
  // per segment were most of this kernel's time
  int nb2 = (ok && n2 == n2) ? __float_as_int(n2) : 0;                // non-negative floats order like ints
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) nb2 = max(nb2, __shfl_xor(nb2, d, 64));
  if (l == 0 && nb2 > 0) atomicMax(bmax_bits + s, nb2);
}

__global__ void knn_row_norms(const float* __restrict__ A, int n, int C, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float n2 = 0.f;
  if ((C & 3) == 0) {                                                  // 16-byte loads, the same sum in the same order
    const float4* row = reinterpret_cast<const float4*>(A + (size_t)i * C);
    for (int q = 0; q < C / 4; ++q) {
      const float4 v = row[q];
      n2 += v.x * v.x; n2 += v.y * v.y; n2 += v.z * v.z; n2 += v.w * v.w;
    }
  } else {
    for (int c = 0; c < C; ++c) n2 += A[(size_t)i * C + c] * A[(size_t)i * C + c];
  }
  out[i] = n2;
}

// Second pass of the pre-filter for the plain SquareL2 query, lane = TARGET: the handful of rows the MFMA scores could not
// decide are few (0.35 % on the bench) but every one needs all targets of its segment.  With lane = query (knn1_kernel) a wave
// has 17 of its 64 lanes busy and walks its targets through dependent scalar loads (152 us, plus the 50 us transposition of
// the targets that only this pass needed).  Here a workgroup keeps 256 targets in registers (one row per lane), stages the
// listed queries in LDS (256 at a time) and evaluates the contract - d = a - b; acc = acc + d * d over the channels in order, no
// contraction - for every (listed query, its target); the wave's smallest (distance bits, index) goes to the same 64-bit
// atomicMin.  Same values, same tie rule (lowest index), NaN / inf candidates never enter.
template <int C>
__global__ __launch_bounds__(256) void knn1_list_kernel(const float* __restrict__ A, const float* __restrict__ B, SegArgs seg,
                                                        unsigned long long* __restrict__ best, const int* __restrict__ only,
                                                        const int* __restrict__ only_cnt) {
#pragma clang fp contract(off)
  constexpr int QB = 256;
  __shared__ __attribute__((aligned(16))) float qs[QB][C];
  __shared__ int qrow[QB];
  const int s = blockIdx.z;
  const int cnt = only_cnt[s];
  const int a0 = seg.a[s], b0 = seg.b[s], nb = seg.b[s + 1] - b0;
  if (cnt <= 0 || (int)blockIdx.x * 256 >= nb) return;                 // workgroup-uniform
  const int lane = threadIdx.x & 63;
  const int j = (int)blockIdx.x * 256 + (int)threadIdx.x;              // this lane's target
  float b[C];
  {
    const float4* src = reinterpret_cast<const float4*>(B + (size_t)(b0 + (j < nb ? j : nb - 1)) * C);
#pragma unroll
    for (int i = 0; i < C / 4; ++i) { const float4 v = src[i]; b[4 * i] = v.x; b[4 * i + 1] = v.y; b[4 * i + 2] = v.z; b[4 * i + 3] = v.w; }
  }
  for (int q0 = 0; q0 < cnt; q0 += QB) {
    const int nq = min(QB, cnt - q0);
    __syncthreads();
    for (int e = threadIdx.x; e < nq * (C / 4); e += 256) {
      const int qi = e / (C / 4), part = e % (C / 4);
      const int row = only[a0 + q0 + qi];
      if (part == 0) qrow[qi] = row;
      reinterpret_cast<float4*>(qs[qi])[part] = reinterpret_cast<const float4*>(A + (size_t)(a0 + row) * C)[part];
    }
    __syncthreads();
    for (int i = 0; i < nq; ++i) {
      float acc = 0.0f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float d = qs[i][c] - b[c];
        d = d * d;
        acc = acc + d;
      }
      unsigned long long p = (j < nb && acc < __builtin_inff()) ? (((unsigned long long)__float_as_uint(acc) << 32) | (unsigned)j) : ~0ull;
#pragma unroll
      for (int dlt = 32; dlt >= 1; dlt >>= 1) {
        const unsigned int lo = __shfl_xor((unsigned int)p, dlt, 64), hi = __shfl_xor((unsigned int)(p >> 32), dlt, 64);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        p = o < p ? o : p;
      }
      if (lane == 0 && p != ~0ull) atomicMin(&best[a0 + qrow[i]], p);
    }
  }
}

template <int C, int MODE>
__global__ __launch_bounds__(256) void knn_mfma_kernel(const float* __restrict__ A, const float* __restrict__ Bm, SegArgs seg,
                                                       SegMfma sm, const float* __restrict__ anorm,
                                                       const int* __restrict__ bmax_bits,
                                                       unsigned long long* __restrict__ best, int* __restrict__ flist,
                                                       int* __restrict__ fcnt) {
  static_assert(C == 32, "one 16x16x32 MFMA spans the 32 channels");
  constexpr int RT = MF_ROWS / 16;
  const int s = blockIdx.z;
  const int a0 = seg.a[s], na = seg.a[s + 1] - a0;
  const int nb = seg.b[s + 1] - seg.b[s];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int q0 = (blockIdx.x * 4 + wave) * MF_ROWS;
  if (q0 >= na || nb <= 0) return;                                  // wave-uniform; no barrier in this kernel
  const int jl = lane & 15, g = lane >> 4;
  // A fragments: row tile r: channels 8 g .. 8 g + 7 of query q0 + 16 r + jl, scaled per row, split into fp16 hi / lo;
  // mrow[r][e] = -2 / scale of query 16 r + 4 g + e (the rows this lane holds in the MFMA result)
  half8_t ah[RT], al[RT];
  float mrow[RT][4];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int i = q0 + 16 * r + jl;
    const float* row = A + (size_t)(a0 + (i < na ? i : na - 1)) * C + 8 * g;
    const float4 v0 = *reinterpret_cast<const float4*>(row), v1 = *reinterpret_cast<const float4*>(row + 4);
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    float mx = 0.f, sum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { mx = fmaxf(mx, fabsf(x[k])); sum += x[k]; }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (!(sum == sum)) mx = __builtin_nanf("");                        // a NaN anywhere in the row
    const float sc = split_scale(mx);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float y = x[k] * sc;
      const _Float16 h = (_Float16)y;
      ah[r][k] = h;
      al[r][k] = (_Float16)(y - (float)h);
    }
    const float inv = -2.0f / sc;                                      // of row jl of the tile
#pragma unroll
    for (int e = 0; e < 4; ++e) mrow[r][e] = __shfl(inv, 4 * g + e, 64);
  }
  // per lane: queries 16 r + 4 g + e (e = 0..3) against the targets j = jl (mod 16)
  float m1[RT][4], m2[RT][4];
  int j1[RT][4];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int e = 0; e < 4; ++e) { m1[r][e] = __builtin_inff(); m2[r][e] = __builtin_inff(); j1[r][e] = 0x7FFFFFFF; }
  const float4* bm = reinterpret_cast<const float4*>(Bm + sm.bm_off[s]) + lane;
  const int n16 = (nb + 15) / 16;
  float4 bf[2][3];
#pragma unroll
  for (int q = 0; q < 3; ++q) bf[0][q] = bm[q * 64];
  for (int t = 0; t < n16; t += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int tt = t + half;
      if (tt >= n16) break;                                          // wave-uniform
      {   // unconditional (the last tile re-loads itself): a load inside a branch is waited for on the spot
        const int tn = tt + 1 < n16 ? tt + 1 : tt;
#pragma unroll
        for (int q = 0; q < 3; ++q) bf[half ^ 1][q] = bm[((size_t)tn * 3 + q) * 64];
      }
      const int jt = 16 * tt + jl;
      const half8_t bh = __builtin_bit_cast(half8_t, bf[half][0]), bl = __builtin_bit_cast(half8_t, bf[half][1]);
      const float bn = bf[half][2].x, binv = bf[half][2].y;          // of target jt
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bh, acc, 0, 0, 0);   // D[query 4 g + e][target jl]
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[r], bh, acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaf(acc[e] * binv, mrow[r][e], bn);         // |b|^2 - 2 <a, b>
          const bool better = v < m1[r][e];
          const float lose = better ? m1[r][e] : v;                  // NaN scores never enter
          m2[r][e] = lose < m2[r][e] ? lose : m2[r][e];
          m1[r][e] = better ? v : m1[r][e];
          j1[r][e] = better ? jt : j1[r][e];
        }
      }
    }
  }
  // merge the 16 lanes of a group (same queries, interleaved targets): smaller score, ties to the lower index; the
  // loser's best and both runner-ups compete for the runner-up
  const float bmax = sqrtf(__int_as_float(bmax_bits[s]));
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a1 = m1[r][e], a2 = m2[r][e];
      int k1 = j1[r][e];
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const float o1 = __shfl_xor(a1, d, 64), o2 = __shfl_xor(a2, d, 64);
        const int ok1 = __shfl_xor(k1, d, 64);
        const bool take = (o1 < a1) | ((o1 == a1) & (ok1 < k1));
        const float lose = take ? a1 : o1;
        a1 = take ? o1 : a1;
        k1 = take ? ok1 : k1;
        const float rr = a2 < o2 ? a2 : o2;
        a2 = lose < rr ? lose : rr;
      }
      const int i = q0 + 16 * r + 4 * g + e;
      if (jl == 0 && i < na) {
        const float an = sqrtf(anorm[a0 + i]);
        const float err = 128.0f * 5.9604645e-8f * (an + bmax) * (an + bmax);
        // decided: a finite best whose runner-up is out of reach.  (a2 - a1 <= 2 err, NaN / inf anywhere: not decided)
        bool decided = (a1 < __builtin_inff()) & (a1 > -__builtin_inff()) & (a2 - a1 > 2.0f * err) & (k1 != 0x7FFFFFFF);
        if (MODE == 2) {
          // score = -2 S.  sqrt(2 - 2 S + 1e-6) in fp32 can round two different S to one distance (then the LOWER
          // index wins, not the larger S): S1 > S2 + 3 * 2^-22 (1 + |S|) guarantees d1 < d2 strictly, so the runner-up
          // must also be farther than cm = 2^-19 (1 + |a| |b|max) in score units; and no S may exceed 1 (a NaN
          // distance beats everything): every S_j < 1 is implied by a1 > -2 + err
          const float cm = 1.9073486e-6f * (1.0f + an * bmax);
          decided = decided & (a2 - a1 > 2.0f * err + cm) & (a1 > -2.0f + err);
        }
        best[a0 + i] = decided ? (unsigned long long)(unsigned)k1 : ~0ull;
        if (!decided) flist[a0 + atomicAdd(fcnt + s, 1)] = i;            // the segment's undecided rows, densely (any order)
      }
    }
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
void launch_knn(const float* A, const float* Bt, const SegArgs& seg, int nseg, int max_na, int max_nb, int dist_type,
                unsigned long long* best, float* second, bool dot, hipStream_t st, const int* only = nullptr,
                const int* only_cnt = nullptr) {
  int qtiles = only ? 2 : eyoc::cdiv(max_na, 64);   // second pass of the pre-filter: two blocks per (segment, split) walk the list
  int total = qtiles * nseg;
  int max_splits = eyoc::cdiv(max_nb, SPLIT_ALIGN);
  int nsplit = 4096 / (total > 0 ? total : 1);
  if (nsplit < 1 || second) nsplit = 1;
  if (nsplit > max_splits) nsplit = max_splits;
  int split_len = eyoc::cdiv(eyoc::cdiv(max_nb, nsplit), SPLIT_ALIGN) * SPLIT_ALIGN;
  nsplit = eyoc::cdiv(max_nb, split_len);
  dim3 grid(qtiles, nsplit, nseg);
  if (second) hipLaunchKernelGGL((knn1_kernel<C, true, 0>), grid, dim3(256), 0, st, A, Bt, seg, split_len, dist_type, best, second, (const int*)nullptr, (const int*)nullptr);
  else if (dot) hipLaunchKernelGGL((knn1_kernel<C, false, 1>), grid, dim3(256), 0, st, A, Bt, seg, split_len, 0, best, second, (const int*)nullptr, (const int*)nullptr);
  else if (dist_type == 2) hipLaunchKernelGGL((knn1_kernel<C, false, 2>), grid, dim3(256), 0, st, A, Bt, seg, split_len, 0, best, second, only, only_cnt);
  else hipLaunchKernelGGL((knn1_kernel<C, false, 0>), grid, dim3(256), 0, st, A, Bt, seg, split_len, dist_type, best, second, only, only_cnt);
}

}  // namespace

extern "C" int eyoc_knn_prefilter(eyoc_ctx* ctx, int mode) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.knn_prefilter;
  if (mode >= 0 && mode <= 2) ctx->knobs.knn_prefilter = mode;
  return prev;
}

static int knn_run(eyoc_ctx* ctx, const float* A_dev, const float* B_dev, int c, const int32_t* seg_a,
                   const int32_t* seg_b, int nseg, int dist_type, int64_t* idx_dev, float* dist_dev, float* second_dev,
                   void* stream, bool dot = false) {
  EYOC_REQUIRE(ctx && A_dev && B_dev && seg_a && seg_b, EYOC_ERR_INVALID, "eyoc_knn1: NULL argument");
  EYOC_REQUIRE(nseg >= 1 && nseg <= MAX_SEG, EYOC_ERR_INVALID, "eyoc_knn1: nseg %d not in [1,%d]", nseg, MAX_SEG);
  EYOC_REQUIRE(dist_type >= 0 && dist_type <= 2, EYOC_ERR_INVALID, "eyoc_knn1: dist_type %d", dist_type);
  EYOC_REQUIRE(!(dist_type == 2 && (dot || second_dev)), EYOC_ERR_INVALID, "eyoc_knn1: dist_type 2 is a 1-NN mode");
  EYOC_REQUIRE(c == 4 || c == 16 || c == 32 || c == 64 || c == 128, EYOC_ERR_INVALID,
               "eyoc_knn1: feature dimension %d not supported (4/16/32/64/128)", c);
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
  long long bt_floats = 0;
  int max_ld = 0;
  for (int s = 0; s < nseg; ++s) {
    const int nb = seg_b[s + 1] - seg_b[s];
    seg.ld[s] = (nb + TGROUP - 1) / TGROUP * TGROUP;
    seg.bt_off[s] = bt_floats;
    bt_floats += (long long)seg.ld[s] * c;
    max_ld = seg.ld[s] > max_ld ? seg.ld[s] : max_ld;
  }
  // MFMA pre-filter (see knn_mfma_kernel): plain index queries that fill the chip
  const int prefilter_env = ctx->knobs.knn_prefilter;
  long long waves = 0, bm_floats = 0;
  SegMfma sm;
  for (int s = 0; s < nseg; ++s) {
    waves += eyoc::cdiv(seg_a[s + 1] - seg_a[s], MF_ROWS);
    sm.bm_off[s] = bm_floats;
    bm_floats += (long long)eyoc::cdiv(seg_b[s + 1] - seg_b[s], 16) * 3 * 64 * 4;   // 3 x 64 float4 per tile of 16 targets
  }
  const bool prefilter = prefilter_env != 0 && !dot && !second_dev && !dist_dev && dist_type != 1 && c == 32 && max_nb > 0 &&
                         (waves >= 512 || prefilter_env == 2);   // (round 5: a single pair - 79 waves, each walking all 5000 targets - takes 0.15 ms here against 0.10 in the exact kernel)
  const size_t off_bt = eyoc::align_up((size_t)n_total * sizeof(unsigned long long));
  const size_t off_bm = eyoc::align_up(off_bt + (size_t)bt_floats * sizeof(float) + 64);
  const size_t off_an = eyoc::align_up(off_bm + (prefilter ? (size_t)bm_floats * sizeof(float) : 0));
  const size_t off_fl = eyoc::align_up(off_an + (prefilter ? (size_t)n_total * sizeof(float) : 0));
  const size_t off_bx = eyoc::align_up(off_fl + (prefilter ? (size_t)n_total * sizeof(int) : 0));
  int rc = ctx->ensure_scratch(off_bx + 2 * MAX_SEG * sizeof(int) + 64, st);
  if (rc) return rc;
  unsigned long long* best = (unsigned long long*)ctx->scratch;
  float* Bt = (float*)((char*)ctx->scratch + off_bt);
  EYOC_CHECK_HIP(hipMemsetAsync(best, 0xFF, (size_t)n_total * sizeof(unsigned long long), st));
  const int *only = nullptr, *only_cnt = nullptr;
  if (prefilter) {
    float* Bm = (float*)((char*)ctx->scratch + off_bm);
    float* anorm = (float*)((char*)ctx->scratch + off_an);
    int* flags = (int*)((char*)ctx->scratch + off_fl);
    int* bmax = (int*)((char*)ctx->scratch + off_bx);
    int* fcnt = bmax + MAX_SEG;
    EYOC_CHECK_HIP(hipMemsetAsync(bmax, 0, 2 * MAX_SEG * sizeof(int), st));
    hipLaunchKernelGGL(knn_pack_targets_mfma<32>, dim3(eyoc::cdiv((long long)eyoc::cdiv(max_nb, 16) * 64, 256), 1, nseg), dim3(256), 0, st,
                       B_dev, seg, sm, Bm, bmax, dist_type == 2 ? 1 : 0);
    hipLaunchKernelGGL(knn_row_norms, dim3(eyoc::cdiv(n_total, 256)), dim3(256), 0, st, A_dev, n_total, c, anorm);
    if (dist_type == 2)
      hipLaunchKernelGGL((knn_mfma_kernel<32, 2>), dim3(eyoc::cdiv(max_na, 4 * MF_ROWS), 1, nseg), dim3(256), 0, st, A_dev, Bm, seg,
                         sm, anorm, bmax, best, flags, fcnt);
    else
      hipLaunchKernelGGL((knn_mfma_kernel<32, 0>), dim3(eyoc::cdiv(max_na, 4 * MF_ROWS), 1, nseg), dim3(256), 0, st, A_dev, Bm, seg,
                         sm, anorm, bmax, best, flags, fcnt);
    only = flags;
    only_cnt = fcnt;
  }
  if (prefilter && dist_type == 0 && max_nb > 0) {
    // the undecided rows, lane = target (no transposed copy of the targets needed)
    hipLaunchKernelGGL(knn1_list_kernel<32>, dim3(eyoc::cdiv(max_nb, 256), 1, nseg), dim3(256), 0, st, A_dev, B_dev, seg, best, only, only_cnt);
  } else if (max_nb > 0) {
    hipLaunchKernelGGL(knn_transpose_targets, dim3(eyoc::cdiv((long long)max_ld * (c / 4), 256), 1, nseg), dim3(256), 0, st, B_dev,
                       seg, c, Bt);
    switch (c) {
      case 4: launch_knn<4>(A_dev, Bt, seg, nseg, max_na, max_nb, dist_type, best, second_dev, dot, st); break;
      case 16: launch_knn<16>(A_dev, Bt, seg, nseg, max_na, max_nb, dist_type, best, second_dev, dot, st); break;
      case 32: launch_knn<32>(A_dev, Bt, seg, nseg, max_na, max_nb, dist_type, best, second_dev, dot, st, only, only_cnt); break;
      case 64: launch_knn<64>(A_dev, Bt, seg, nseg, max_na, max_nb, dist_type, best, second_dev, dot, st); break;
      default: launch_knn<128>(A_dev, Bt, seg, nseg, max_na, max_nb, dist_type, best, second_dev, dot, st); break;
    }
  }
  hipLaunchKernelGGL(knn1_unpack, dim3(eyoc::cdiv(n_total, 256)), dim3(256), 0, st, best, n_total,
                     (long long*)idx_dev, dist_dev, dot ? 1 : dist_type == 2 ? 2 : 0);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

extern "C" int eyoc_knn1(eyoc_ctx* ctx, const float* A_dev, const float* B_dev, int c, const int32_t* seg_a,
                         const int32_t* seg_b, int nseg, int dist_type, int64_t* idx_dev, float* dist_dev,
                         void* stream) {
  return knn_run(ctx, A_dev, B_dev, c, seg_a, seg_b, nseg, dist_type, idx_dev, dist_dev, nullptr, stream);
}

extern "C" int eyoc_dotmax(eyoc_ctx* ctx, const float* A_dev, const float* B_dev, int c, const int32_t* seg_a,
                           const int32_t* seg_b, int nseg, int64_t* idx_dev, float* weight_dev, void* stream) {
  return knn_run(ctx, A_dev, B_dev, c, seg_a, seg_b, nseg, 0, idx_dev, weight_dev, nullptr, stream, true);
}

extern "C" int eyoc_knn2(eyoc_ctx* ctx, const float* A_dev, const float* B_dev, int c, const int32_t* seg_a,
                         const int32_t* seg_b, int nseg, int64_t* idx_dev, float* d1_dev, float* d2_dev, void* stream) {
  EYOC_REQUIRE(d2_dev != nullptr, EYOC_ERR_INVALID, "eyoc_knn2: d2_dev is NULL");
  return knn_run(ctx, A_dev, B_dev, c, seg_a, seg_b, nseg, 0, idx_dev, d1_dev, d2_dev, stream);
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
