// Feature-matching RANSAC on the GPU (replaces Open3D's
// registration_ransac_based_on_feature_matching at scripts/test_kitti.py:169-177).
//
// Three phases, all on the device:
//   1. generate: one lane per hypothesis h - counter-hash sampler (bit-identical to
//      oracle/ransac.py), edge-length checker, 4-point Kabsch in fp64, distance checker.  Survivors
//      (typically well under a few percent) append h to a compact list with one atomic.
//      The survivor's transform (12 doubles) is stored next to its index.
//   2. count: a wave takes 16 survivors at a time and sweeps all correspondences, 4 per lane held in registers
//      as doubles; the 16 transforms are wave-uniform and come through the scalar cache as SGPR operands of
//      v_fma_f64 (17 VALU instructions per (hypothesis, correspondence), no loads or conversions in the
//      inner loop).  With a trained network's inlier ratio p about p^4 of the 4M hypotheses survive (32 000 per
//      pair at p = 0.3) and this sweep IS the cost of RANSAC: the first version (one wave per survivor,
//      transform re-derived per wave, sqrt per residual) spent 8.2 ms per 16 pairs there.
//      The largest count of each pair is kept with an atomicMax.
//   3. rmse: only the survivors that reach the largest count (almost always one) get their inlier RMSE.
//   4. select: single workgroup arg-max with the total order (more inliers, lower RMSE, lower h).
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
constexpr int CHUNK = 64;  // pairs per launch (the per-pair kernels - bucket, rmse, select - are one workgroup per pair: 8 pairs per launch left them at 8 workgroups on 256 CUs, eight times per 64-pair batch)
struct PairArgs {          // a chunk of pairs; segment bounds travel as kernel arguments (no H2D copy)
  int s0[CHUNK], n[CHUNK], t0[CHUNK];   // first source row, correspondences, first target row of each pair
  int pair0;               // index of the chunk's first pair (seed offset, result slot)
  float* rec;              // [total, 6]
  unsigned int seed;       // pair b uses seed + b
  int H;
  float edge_sim, max_dist;
  int* n_surv;             // [chunk] survivor counters (CNT_STRIDE ints apart); [+1] = largest inlier count
  int* surv;               // [chunk][H]
  int* cnts;               // [chunk][H] inlier count of every survivor
  unsigned int* rmse;      // [chunk][H] fp32 bits of the inlier RMSE (only written for survivors at the largest count)
  double* xf;              // [chunk][cap_t][12] transforms (R row-major, t) of the first cap_t survivors of a pair
  int cap_t;
  float* rec_sorted;       // [total, 6] the records of every pair bucketed by their residual under the pair's reference transform
  float* rr_sorted;        // [total, 4] fp32( that residual vector ), same order (what the packed fp32 sweep of k_count adds the deltas to)
  int* bucket_end;         // [chunk][NBUCKET + 1] records in buckets 0 .. b (prefix lengths); [NBUCKET] = n
  double* pmax;            // [chunk] largest |source point| of the pair
  int count_bound;         // k_count: survivors that cannot reach the pair's best count any more stop counting (eyoc_ransac_select_pruning 2)
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

// the four sampled correspondences of hypothesis h and the edge-length checker on them
__device__ inline bool sample_and_check_edges(const float* __restrict__ rec, unsigned int n, unsigned long long base,
                                              unsigned int h, double edge_sim, double s[4][3], double q[4][3]) {
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
  return true;
}

// the first of the six edge checks alone (samples 0 and 1: one hash, two records) - the same expressions as above, so a
// hypothesis it rejects is one sample_and_check_edges rejects
__device__ inline bool first_edge_ok(const float* __restrict__ rec, unsigned int n, unsigned long long base, unsigned int h,
                                     double edge_sim) {
  unsigned int i0, i1;
  sample_pair(base, 2ull * h, n, i0, i1);
  const float2* ra = reinterpret_cast<const float2*>(rec + (size_t)i0 * 6);
  const float2* rb = reinterpret_cast<const float2*>(rec + (size_t)i1 * 6);
  const float2 a0 = ra[0], a1 = ra[1], a2 = ra[2], b0 = rb[0], b1 = rb[1], b2 = rb[2];
  const double sa[3] = {a0.x, a0.y, a1.x}, qa[3] = {a1.y, a2.x, a2.y}, sb[3] = {b0.x, b0.y, b1.x}, qb[3] = {b1.y, b2.x, b2.y};
  const double e2 = edge_sim * edge_sim;
  const double ds2 = (sa[0] - sb[0]) * (sa[0] - sb[0]) + (sa[1] - sb[1]) * (sa[1] - sb[1]) + (sa[2] - sb[2]) * (sa[2] - sb[2]);
  const double dt2 = (qa[0] - qb[0]) * (qa[0] - qb[0]) + (qa[1] - qb[1]) * (qa[1] - qb[1]) + (qa[2] - qb[2]) * (qa[2] - qb[2]);
  return !(ds2 < dt2 * e2 || dt2 < ds2 * e2);
}

// 4-point Kabsch and the distance checker
__device__ inline bool fit_and_check_distance(const double s[4][3], const double q[4][3], double max_dist, double R[3][3],
                                              double t[3]) {
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

// transform of hypothesis h, or false if a checker rejects it.  `rec` may point to LDS or global memory.
__device__ inline bool hypothesis(const float* __restrict__ rec, unsigned int n, unsigned long long base, unsigned int h,
                                  double edge_sim, double max_dist, double R[3][3], double t[3]) {
  double s[4][3], q[4][3];
  if (!sample_and_check_edges(rec, n, base, h, edge_sim, s, q)) return false;
  return fit_and_check_distance(s, q, max_dist, R, t);
}

constexpr int CNT_STRIDE = 64;
constexpr int GEN_THREADS = 1024;
constexpr int GEN_BLOCKS_MIN = 8, GEN_BLOCKS_TOTAL = 256;   // workgroups per pair: 256 over the pairs of a launch - one per CU (512 = two rounds of workgroups, each copying the records again: a single pair took 104 us instead of 59) - (every workgroup copies its pair's records to LDS first: 64 per pair at 64 pairs per launch was 4096 copies of 120 KB), at least 8
constexpr int LDS_RECORDS = 6016;                    // 6016 * 24 B = 141 KB of the 160 KB LDS (+ 16 KB of queues)

__device__ inline unsigned long long pair_base(unsigned int seed, int b) {
  return (unsigned long long)(seed + (unsigned)b) * 0x9E3779B97F4A7C15ull;
}

// One workgroup stages its pair's records in LDS (sampling is 8 random reads per hypothesis: L2 round trips
// dominated the first version of this kernel) and walks a strided slice of the hypotheses.
// The edge-length checker rejects ~99 % of the hypotheses even at a 30 % inlier ratio, but in lock step a wave pays
// for the 4-point Kabsch (a Jacobi eigen-solver) whenever ONE of its 64 lanes gets through - about half of all waves
// at that ratio (k_generate 0.2 -> 1.0 ms per 8 pairs).  So the wave queues the hypotheses that pass the edge check
// in LDS and only fits them 64 at a time, all lanes busy.
template <bool IN_LDS>
__global__ __launch_bounds__(GEN_THREADS) void k_generate(PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lrec[];
  __shared__ int queue[GEN_THREADS / 64][128], queue1[GEN_THREADS / 64][128];
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
  int* ncand = a.n_surv + c * CNT_STRIDE + 2;              // the pair's candidate counter ([0] survivors, [1] largest count)
  int* cand = a.cnts + (size_t)c * a.H;                    // candidates wait in the pair's count array (k_count writes it after k_fit)
  const int lane = threadIdx.x & 63;
  int* wq = queue[threadIdx.x >> 6];
  int queued = 0;   // wave-uniform
  const double edge_sim = (double)a.edge_sim;
  // Round 5: the hypotheses that pass all six edge checks (~1 %) are handed to k_fit - 64 at a time, one atomic per wave and
  // one coalesced store - instead of being fitted here: the 4-point Kabsch (a Jacobi eigen-solver in fp64) needs more registers
  // than the 128 a 1024-thread workgroup leaves a lane (400 bytes of scratch per lane, also paid by the sampling loop around it)
  auto flush = [&](int h) {   // h < 0: idle lane
    const unsigned long long m = __ballot(h >= 0);
    int pos = 0;
    if (lane == 0) pos = atomicAdd(ncand, __popcll(m));
    pos = __shfl(pos, 0, 64);
    if (h >= 0) cand[pos + __popcll(m & ((1ull << lane) - 1ull))] = h;
  };
  // Two queues per wave: every hypothesis takes the FIRST edge check only (one hash, two records: most fail it); the ones that
  // pass are queued and take the full six-edge check 64 at a time; the ones that pass that are queued for the fit.
  // (No nested lambdas here: a closure that captures another one ends up in scratch memory, and the records with it behind
  // generic pointers - flat loads instead of LDS reads, 5 x slower.)
  int* wq1 = queue1[threadIdx.x >> 6];
  int queued1 = 0;
  const int stride = gridDim.x * GEN_THREADS;
  const int rounds = (a.H + stride - 1) / stride;
  for (int r = 0; r <= rounds; ++r) {                       // the extra round drains the first queue
    if (r < rounds) {
      const int h = r * stride + blockIdx.x * GEN_THREADS + threadIdx.x;
      const bool pass1 = h < a.H && first_edge_ok(rec, (unsigned)n, base, (unsigned)h, edge_sim);
      const unsigned long long m1 = __ballot(pass1);
      if (pass1) wq1[queued1 + __popcll(m1 & ((1ull << lane) - 1ull))] = h;
      queued1 += __popcll(m1);
    }
    const bool drain = r == rounds;
    if (queued1 >= 64 || (drain && queued1 > 0)) {          // wave-uniform
      int h2;
      if (drain) { h2 = lane < queued1 ? wq1[lane] : -1; queued1 = 0; }
      else { queued1 -= 64; h2 = wq1[queued1 + lane]; }
      bool pass = false;
      if (h2 >= 0) {
        double s[4][3], q[4][3];
        pass = sample_and_check_edges(rec, (unsigned)n, base, (unsigned)h2, edge_sim, s, q);
      }
      const unsigned long long m = __ballot(pass);
      if (pass) wq[queued + __popcll(m & ((1ull << lane) - 1ull))] = h2;
      queued += __popcll(m);
      if (queued >= 64) {
        queued -= 64;
        flush(wq[queued + lane]);
      }
    }
  }
  flush(lane < queued ? wq[lane] : -1);
}

// 4-point fit + distance checker of the queued candidates: one lane per candidate, 256-thread workgroups (the whole register
// file: no scratch), records from global memory (a pair's 120 KB stay in L2).  Survivors append their hypothesis number and
// transform exactly as the fused kernel did; which slot a survivor gets depends on atomics, the results do not.
__global__ __launch_bounds__(256) void k_fit(PairArgs a) {
  const int c = blockIdx.y;
  const float* __restrict__ rec = a.rec + (size_t)a.s0[c] * 6;
  const int n = a.n[c];
  const unsigned long long base = pair_base(a.seed, a.pair0 + c);
  int* cnt = a.n_surv + c * CNT_STRIDE;
  const int ncand = cnt[2];
  const int* __restrict__ cand = a.cnts + (size_t)c * a.H;
  int* surv = a.surv + (size_t)c * a.H;
  const double edge_sim = (double)a.edge_sim, max_dist = (double)a.max_dist;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < ncand; i += gridDim.x * 256) {
    const int h = cand[i];
    double s[4][3], q[4][3], R[3][3], t[3];
    sample_and_check_edges(rec, (unsigned)n, base, (unsigned)h, edge_sim, s, q);   // re-draws the sample (cheaper than queueing 24 doubles)
    if (!fit_and_check_distance(s, q, max_dist, R, t)) continue;
    const int slot = atomicAdd(cnt, 1);
    surv[slot] = h;
    if (slot < a.cap_t) {
      double* x = a.xf + ((size_t)c * a.cap_t + slot) * 12;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        x[3 * k] = R[k][0]; x[3 * k + 1] = R[k][1]; x[3 * k + 2] = R[k][2];
        x[9 + k] = t[k];
      }
    }
  }
}

// squared form of Open3D's `dist < max_correspondence_distance` (no fp64 square root per residual)
__device__ inline double thr2_of(float max_dist) { return (double)max_dist * (double)max_dist; }

// ---- reference pruning of the count.  Survivors of a pair are near-duplicates of one another (at an inlier ratio p
// nearly all of them are all-inlier samples), and a correspondence that is a far outlier under ONE of them is an outlier
// under all that are close to it: with T_ref = the pair's first survivor, d_s(i) >= d_ref(i) - delta_s,
// delta_s = |R_s - R_ref|_F max_i |p_i| + |t_s - t_ref|.  k_bucket sorts the pair's records into NBUCKET buckets of
// width max_distance by d_ref (the last one open-ended); survivor s then only evaluates the records of buckets
// 0 .. floor((max_distance + delta_s) / width) + 1 - every other record is provably farther than max_distance.  The
// counts are exactly those of the full sweep (tests/test_gpu_pose.py: bit-exact against the oracle's); at p = 0.3 a
// survivor evaluates ~1600 of 5000 correspondences.  Which survivor holds slot 0 depends on atomics; only the amount of
// skipped work depends on it.
constexpr int NBUCKET = 64;

__global__ __launch_bounds__(256) void k_bucket(PairArgs a) {
  __shared__ int hist[NBUCKET], start[NBUCKET];
  __shared__ double red[4];
  const int c = blockIdx.x;
  const int s0 = a.s0[c], n = a.n[c];
  const float* __restrict__ rec = a.rec + (size_t)s0 * 6;
  float* __restrict__ out = a.rec_sorted + (size_t)s0 * 6;
  int* bend = a.bucket_end + c * (NBUCKET + 1);
  const int ns = a.n_surv[c * CNT_STRIDE];
  if (threadIdx.x < NBUCKET) hist[threadIdx.x] = 0;
  __syncthreads();
  const bool have_ref = ns > 0;                                        // block-uniform
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  if (have_ref) {
    const double* x = a.xf + (size_t)c * a.cap_t * 12;
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = x[i];
    t[0] = x[9]; t[1] = x[10]; t[2] = x[11];
  }
  const double inv_w = 1.0 / (double)a.max_dist;
  double pm = 0.0;
  // pass 1: bucket of every record (kept in registers: at most 32 records per thread, n <= 8192)
  int bkt[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int i = u * 256 + (int)threadIdx.x;
    bkt[u] = -1;
    if (i < n) {
      const float2* r = reinterpret_cast<const float2*>(rec + (size_t)i * 6);
      const float2 p0 = r[0], p1 = r[1], p2 = r[2];
      const double x = p0.x, y = p0.y, z = p1.x;
      const double dx = fma(R[0], x, fma(R[1], y, fma(R[2], z, t[0] - (double)p1.y)));
      const double dy = fma(R[3], x, fma(R[4], y, fma(R[5], z, t[1] - (double)p2.x)));
      const double dz = fma(R[6], x, fma(R[7], y, fma(R[8], z, t[2] - (double)p2.y)));
      const double d = sqrt(fma(dx, dx, fma(dy, dy, dz * dz)));
      const double q = d * inv_w;
      int b = have_ref ? (q < (double)(NBUCKET - 1) ? (int)q : NBUCKET - 1) : 0;   // NaN -> last bucket
      if (!(q == q)) b = NBUCKET - 1;
      bkt[u] = b;
      atomicAdd(&hist[b], 1);
      pm = fmax(pm, sqrt(fma(x, x, fma(y, y, z * z))));
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) pm = fmax(pm, __shfl_xor(pm, d, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pm;
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int b = 0; b < NBUCKET; ++b) { start[b] = acc; acc += hist[b]; bend[b] = acc; }
    bend[NBUCKET] = n;
    a.pmax[c] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  }
  __syncthreads();
  // pass 2: scatter (the order inside a bucket is arbitrary - the counts do not depend on it)
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int i = u * 256 + (int)threadIdx.x;
    if (bkt[u] >= 0) {
      const int pos = atomicAdd(&start[bkt[u]], 1);
      const float2* r = reinterpret_cast<const float2*>(rec + (size_t)i * 6);
      float2* w = reinterpret_cast<float2*>(out + (size_t)pos * 6);
      const float2 p0 = r[0], p1 = r[1], p2 = r[2];
      w[0] = p0; w[1] = p1; w[2] = p2;
      const double x = p0.x, y = p0.y, z = p1.x;                          // the residual under the reference transform once more (pass 1's expression)
      const double dx = fma(R[0], x, fma(R[1], y, fma(R[2], z, t[0] - (double)p1.y)));
      const double dy = fma(R[3], x, fma(R[4], y, fma(R[5], z, t[1] - (double)p2.x)));
      const double dz = fma(R[6], x, fma(R[7], y, fma(R[8], z, t[2] - (double)p2.y)));
      reinterpret_cast<float4*>(a.rr_sorted)[(size_t)s0 + pos] = make_float4((float)dx, (float)dy, (float)dz, 0.f);
    }
  }
}

// ---- count in fp64 (records that are not bucketed: more than 8192 correspondences, or pruning switched off).  A wave takes a group of G <= 64 survivors and sweeps the (bucketed) correspondences in blocks of 64 * PPL
// held in registers as doubles.  For one survivor of the group its transform arrives in SGPRs (scalar loads: the address
// is wave-uniform), the residual test of the block is 16 VALU instructions per correspondence, and the inlier count
// of the block is s_bcnt1 of the compare masks - scalar work.  Lane s of one VGPR accumulates survivor s's count, so
// the group's counts are written with one coalesced store and nothing is ever reduced across lanes.  A survivor skips
// the blocks past its prefix (reference pruning above).
constexpr int PPL = 4;

__global__ __launch_bounds__(256) void k_count_fp64(PairArgs a, const double* __restrict__ xf_all, int pruned) {
  const int c = blockIdx.y;
  const int s0 = a.s0[c], n = a.n[c];
  const float* __restrict__ rec = (pruned ? a.rec_sorted : a.rec) + (size_t)s0 * 6;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ns_all = a.n_surv[c * CNT_STRIDE];
  const int ns = ns_all < a.cap_t ? ns_all : a.cap_t;
  const double thr2 = thr2_of(a.max_dist);
  const double* __restrict__ xf = xf_all + (size_t)c * a.cap_t * 12;
  const int* __restrict__ bend = a.bucket_end + c * (NBUCKET + 1);
  int* cnts = a.cnts + (size_t)c * a.H;
  int G = 64;   // fewer survivors: smaller groups, so that a pair still gives every wave of its grid slice a group
  while (G > 8 && (ns + G - 1) / G < (int)gridDim.x * 4) G >>= 1;
  const int n_groups = (ns + G - 1) / G;
  int best = 0;
  for (int grp = blockIdx.x * 4 + wave; grp < n_groups; grp += gridDim.x * 4) {
    const int sbase = grp * G;
    const int gs = min(G, ns - sbase);
    // lane s: how many (bucketed) records survivor s has to look at
    int hi = 0;
    if (lane < gs) {
      hi = n;
      if (pruned) {
        const double* __restrict__ T = xf + (size_t)(sbase + lane) * 12;
        double fr = 0.0, dt = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) { const double e = T[i] - xf[i]; fr = fma(e, e, fr); }
#pragma unroll
        for (int i = 9; i < 12; ++i) { const double e = T[i] - xf[i]; dt = fma(e, e, dt); }
        const double delta = (sqrt(fr) * a.pmax[c] + sqrt(dt)) * (1.0 + 1e-9) + 1e-9;
        const double q = ((double)a.max_dist + delta) / (double)a.max_dist;
        const int b = q < (double)(NBUCKET - 2) ? (int)q + 1 : NBUCKET;   // one bucket of margin; NaN / huge: everything
        hi = bend[b < NBUCKET ? b : NBUCKET];
      }
    }
    int hi_max = hi;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) hi_max = max(hi_max, __shfl_xor(hi_max, d, 64));
    hi_max = __builtin_amdgcn_readfirstlane(hi_max);
    int mycnt = 0;
    for (int i0 = 0; i0 < hi_max; i0 += 64 * PPL) {
      double x[PPL], y[PPL], z[PPL], qx[PPL], qy[PPL], qz[PPL];
#pragma unroll
      for (int u = 0; u < PPL; ++u) {
        const int i = i0 + u * 64 + lane;
        if (i < n) {
          const float2* r = reinterpret_cast<const float2*>(rec + (size_t)i * 6);
          const float2 p0 = r[0], p1 = r[1], p2 = r[2];
          x[u] = p0.x; y[u] = p0.y; z[u] = p1.x; qx[u] = p1.y; qy[u] = p2.x; qz[u] = p2.y;
        } else {   // past the end: a correspondence no transform brings within reach
          x[u] = y[u] = z[u] = 0.0; qx[u] = qy[u] = qz[u] = 1e18;
        }
      }
#pragma unroll 1
      for (int s = 0; s < gs; ++s) {
        if (i0 >= __builtin_amdgcn_readlane(hi, s)) continue;          // wave-uniform: this survivor's prefix ends before the block
        const double* __restrict__ T = xf + (size_t)(sbase + s) * 12;   // wave-uniform
        const double r00 = T[0], r01 = T[1], r02 = T[2], r10 = T[3], r11 = T[4], r12 = T[5], r20 = T[6], r21 = T[7], r22 = T[8];
        const double t0 = T[9], t1 = T[10], t2 = T[11];
        int csum = 0;
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
          const double dx = fma(r00, x[u], fma(r01, y[u], fma(r02, z[u], t0 - qx[u])));
          const double dy = fma(r10, x[u], fma(r11, y[u], fma(r12, z[u], t1 - qy[u])));
          const double dz = fma(r20, x[u], fma(r21, y[u], fma(r22, z[u], t2 - qz[u])));
          const double d2 = fma(dx, dx, fma(dy, dy, dz * dz));
          csum += __popcll(__ballot(d2 < thr2));
        }
        mycnt += lane == s ? csum : 0;
      }
    }
    if (lane < gs) cnts[sbase + lane] = mycnt;
    best = mycnt > best ? mycnt : best;   // lanes >= gs hold 0
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d, 64));
  if (lane == 0 && best > 0) atomicMax(a.n_surv + c * CNT_STRIDE + 1, best);
}

// ---- count on bucketed records (the production path).  Same organisation as k_count_fp64 - a wave per group of G survivors,
// blocks of records in registers, the survivor's numbers through scalar loads, lane s accumulating survivor s's count - but the
// residual test runs in PACKED fp32 (two residuals per v_pk_fma_f32: 16 packed instructions + 4 compares per two residuals
// against 16 v_*_f64 per residual; round 4's sweep sat at 0.84-0.91 of the fp64 issue rate, profiles/r5_valu.json) and stays exact:
//
//   R p + t - q = r_ref + dR p + dt,     r_ref = R_ref p + t_ref - q (k_bucket: fp64, rounded to fp32 once),  dR = R - R_ref, dt = t - t_ref
//
// With the pair's survivors near-duplicates of the reference, every term on the right is small (|r_ref| <= a few max_distance
// inside the pruned prefix, |dR| |p| and |dt| a fraction of a metre), so fp32 rounding moves a component by ~1e-6 m where the
// plain fp32 expression (terms of ~100 m cancelling) moves it by ~1e-3 m - inside the inlier residuals' own spread: the first
// version of this kernel did that, sent most blocks to the fp64 recount and was slower than fp64 throughout.  Bound used:
// every intermediate of a component is at most A = |r_ref|_max + |dt|_inf + 3 |dR|_max |p|_max (maxima over the block's records);
// rounding r_ref, dt and dR to fp32 plus the four roundings of the chain put the fp32 component within 8 u A (u = 2^-24) of the
// real-number value; the fp64 values involved (the oracle's own chain, r_ref, dR, dt) are within ~2^-50 A_full of theirs,
// A_full = the magnitudes before cancellation (2^-40 A_full is added): delta = 10 u A + 2^-40 A_full.  For |r| <= thr + 2 delta
// the squared norms then differ by at most 2 sqrt(3) delta |r| + 3 delta^2 + 3 u d2; `band` is 1.5 x that.  fp32 d2 below
// thr2 - band: the fp64 d2 is below thr2; above thr2 + band: it is not; in between (about one residual in 10^5) the lane
// evaluates the oracle's fp64 expression on the original record.  A survivor whose band is not small against thr2 (far from the
// reference, huge coordinates, non-finite numbers) sweeps the block in fp64.  The counts are the oracle's, bit for bit
// (tests/test_gpu_pose.py, tests/test_gpu_ransac_scale.py, tests/test_gpu_round2.py).
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int PKP = 4;                       // packed pairs of records per lane
constexpr int KC_RECORDS = 64 * 2 * PKP;     // records of a block

// the oracle's test of record i under transform T (12 doubles)
__device__ inline bool inlier_fp64(const float* __restrict__ rec, int i, const double* __restrict__ T, double thr2) {
  const float2* r = reinterpret_cast<const float2*>(rec + (size_t)i * 6);
  const float2 p0 = r[0], p1 = r[1], p2 = r[2];
  const double x = p0.x, y = p0.y, z = p1.x;
  const double dx = fma(T[0], x, fma(T[1], y, fma(T[2], z, T[9] - (double)p1.y)));
  const double dy = fma(T[3], x, fma(T[4], y, fma(T[5], z, T[10] - (double)p2.x)));
  const double dz = fma(T[6], x, fma(T[7], y, fma(T[8], z, T[11] - (double)p2.y)));
  return fma(dx, dx, fma(dy, dy, dz * dz)) < thr2;
}

// five waves per SIMD (96 VGPRs; the spills this costs are in the block prologue and the fp64 recount, not in the sweep): 1.92 -> 1.72 ms
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_count(PairArgs a, const double* __restrict__ xf_all) {
  __shared__ float4 deltas[4][64][3];    // per wave: fp32( T_s - T_ref ) of its group's survivors - dR row-major, dt
  const int c = blockIdx.y;
  const int s0 = a.s0[c], n = a.n[c];
  const float* __restrict__ rec = a.rec_sorted + (size_t)s0 * 6;
  const float4* __restrict__ rr = reinterpret_cast<const float4*>(a.rr_sorted) + s0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ns_all = a.n_surv[c * CNT_STRIDE];
  const int ns = ns_all < a.cap_t ? ns_all : a.cap_t;
  const double thr = (double)a.max_dist, thr2 = thr2_of(a.max_dist);
  const double* __restrict__ xf = xf_all + (size_t)c * a.cap_t * 12;
  const int* __restrict__ bend = a.bucket_end + c * (NBUCKET + 1);
  int* bestp = a.n_surv + c * CNT_STRIDE + 1;                             // the pair's largest count so far
  // who publishes partial maxima: at most ~64 workgroups per pair.  Every wave of a launch of ONE pair (4096 workgroups) doing so put
  // 50 k atomics on one word - a single pair's call went from 1.58 to 1.99 ms; any subset keeps the bound valid (a published value is a
  // count some survivor really has), and strong survivors are in every group
  int pub_every = 1;
  while ((int)gridDim.x / pub_every > 64) pub_every <<= 1;
  const bool publisher = (blockIdx.x & (pub_every - 1)) == 0;
  int* cnts = a.cnts + (size_t)c * a.H;
  int G = 64;   // fewer survivors: smaller groups, so that a pair still gives every wave of its grid slice a group
  while (G > 8 && (ns + G - 1) / G < (int)gridDim.x * 4) G >>= 1;
  const int n_groups = (ns + G - 1) / G;
  int best = 0;
  for (int grp = blockIdx.x * 4 + wave; grp < n_groups; grp += gridDim.x * 4) {
    const int sbase = grp * G;
    const int gs = min(G, ns - sbase);
    // lane s: how many bucketed records survivor s has to look at, and the magnitudes its error band is made of
    int hi = 0;
    double m_dt = 0.0, m_dr = 0.0, m_t = 0.0, m_r = 1.0;   // |dt|_inf, |dR|_max, |t|_inf and |R|_max over {survivor, reference}
    if (lane < gs) {
      const double* __restrict__ T = xf + (size_t)(sbase + lane) * 12;
      double fr = 0.0, dt = 0.0;
      bool finite = true;
      float ef[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const double v = T[i], r = xf[i], e = v - r;
        ef[i] = (float)e;
        finite = finite && fabs(v) < 1e300 && fabs(r) < 1e300;             // false for a NaN as well
        if (i < 9) { fr = fma(e, e, fr); m_dr = fmax(m_dr, fabs(e)); m_r = fmax(m_r, fmax(fabs(v), fabs(r))); }
        else { dt = fma(e, e, dt); m_dt = fmax(m_dt, fabs(e)); m_t = fmax(m_t, fmax(fabs(v), fabs(r))); }
      }
      if (!finite) m_t = 1e300;                                            // no band: this survivor counts in fp64
#pragma unroll
      for (int k = 0; k < 3; ++k) deltas[wave][lane][k] = make_float4(ef[4 * k], ef[4 * k + 1], ef[4 * k + 2], ef[4 * k + 3]);
      const double delta = (sqrt(fr) * a.pmax[c] + sqrt(dt)) * (1.0 + 1e-9) + 1e-9;
      const double q = ((double)a.max_dist + delta) / (double)a.max_dist;
      const int b = q < (double)(NBUCKET - 2) ? (int)q + 1 : NBUCKET;   // one bucket of margin; NaN / huge: everything
      hi = bend[b < NBUCKET ? b : NBUCKET];
    }
    int hi_max = hi;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) hi_max = max(hi_max, __shfl_xor(hi_max, d, 64));
    hi_max = __builtin_amdgcn_readfirstlane(hi_max);
    int mycnt = 0;
    int known_next = a.count_bound ? __hip_atomic_load(bestp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    for (int i0 = 0; i0 < hi_max; i0 += KC_RECORDS) {
      // Round 6: only the arg-max matters downstream (k_rmse and k_select look at the survivors AT the largest count), so a survivor
      // stops counting once count so far + records of its prefix still ahead < the largest count any survivor of the pair has reached
      // so far (partial counts included: each is a lower bound of a final count, so the bound never cuts a survivor that ends at the
      // maximum, ties included).  Its entry in `cnts` stays below the maximum - which is all anyone asks of it.  The buckets come in
      // order of the reference residual, i.e. inliers first: a weak survivor falls behind within its first blocks.
      // (the largest count is read one block ahead: the load's round trip to L2 then sits behind the sweep of a block instead of in
      // front of it - with the 8-survivor groups of a small batch that wait was as long as the sweep; a stale value is a smaller one: safe)
      const int known = __builtin_amdgcn_readfirstlane(known_next);
      if (a.count_bound) known_next = __hip_atomic_load(bestp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int ahead = hi - i0;                                         // lane s: records of survivor s's prefix not looked at yet (<= 0: none)
      const unsigned long long work = __ballot(lane < gs && ahead > 0 && mycnt + ahead >= known);
      if (!work) break;                                                  // wave-uniform: nobody of the group needs the remaining blocks
      f32x2 x[PKP], y[PKP], z[PKP], rx[PKP], ry[PKP], rz[PKP];
      float mp = 0.f, mr = 0.f;            // largest |source coordinate| and |reference residual component| of the block's records
#pragma unroll
      for (int u = 0; u < PKP; ++u) {
        float v[2][6];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = i0 + (2 * u + h) * 64 + lane;
          if (i < n) {
            const float2* r = reinterpret_cast<const float2*>(rec + (size_t)i * 6);
            const float2 p0 = r[0], p1 = r[1];
            const float4 e = rr[i];
            v[h][0] = p0.x; v[h][1] = p0.y; v[h][2] = p1.x; v[h][3] = e.x; v[h][4] = e.y; v[h][5] = e.z;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              const float m = fabsf(v[h][k]) <= 3e38f ? fabsf(v[h][k]) : 3e38f;   // NaN / inf -> no band
              if (k < 3) mp = fmaxf(mp, m); else mr = fmaxf(mr, m);
            }
          } else {   // past the end: a correspondence nothing brings within reach (d2 = 3e36 in fp32: no overflow)
            v[h][0] = v[h][1] = v[h][2] = 0.f; v[h][3] = v[h][4] = v[h][5] = 1e18f;
          }
        }
        x[u] = f32x2{v[0][0], v[1][0]}; y[u] = f32x2{v[0][1], v[1][1]}; z[u] = f32x2{v[0][2], v[1][2]};
        rx[u] = f32x2{v[0][3], v[1][3]}; ry[u] = f32x2{v[0][4], v[1][4]}; rz[u] = f32x2{v[0][5], v[1][5]};
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { mp = fmaxf(mp, __shfl_xor(mp, d, 64)); mr = fmaxf(mr, __shfl_xor(mr, d, 64)); }
      // lane s: the two fp32 thresholds of (survivor s, this block); flo < 0 = "no usable band: count the block in fp64"
      float flo = -1.f, fhi = -1.f;
      {
        const double A = (double)mr + m_dt + 3.0 * m_dr * (double)mp;
        const double A_full = 4.0 * (m_t + (double)mr) + 8.0 * m_r * (double)mp;
        const double delta = 10.0 * 0x1p-24 * A + 0x1p-40 * A_full;
        const double band = 1.5 * (4.0 * delta * (thr + 2.0 * delta) + 3.0 * delta * delta + 8.0 * 0x1p-24 * thr2);
        if (band < 0.25 * thr2 && A_full < 1e15 && thr2 < 1e30 && thr2 > 1e-30) {
          flo = __double2float_rd(thr2 - band);
          fhi = __double2float_ru(thr2 + band);
        }
      }
      int blkcnt = 0;                                                    // lane s: survivor s's count in this block (v_writelane)
#pragma unroll 1
      for (int s = 0; s < gs; ++s) {
        if (!((work >> s) & 1ull)) continue;                            // wave-uniform: this survivor's prefix ends before the block, or it cannot reach the maximum any more
        const float lo_s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, flo), s));
        const float hi_s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fhi), s));
        int csum = 0;
        unsigned long long amb[PKP][2];
        unsigned long long amb_any = 0;
        if (lo_s >= 0.f) {
          // the survivor's deltas: one LDS address for the whole wave (a broadcast read).  They came through scalar loads from a
          // global fp32 copy at first: 3 KB per wave and block sweep, ~16 waves behind one 16 KB scalar cache - every load missed,
          // and the halved arithmetic bought nothing (2.08 against 2.17 ms)
          const float4 D0 = deltas[wave][s][0], D1 = deltas[wave][s][1], D2 = deltas[wave][s][2];
          const float r00 = D0.x, r01 = D0.y, r02 = D0.z, r10 = D0.w, r11 = D1.x, r12 = D1.y, r20 = D1.z, r21 = D1.w, r22 = D2.x;
          const float t0 = D2.y, t1 = D2.z, t2 = D2.w;
#pragma unroll
          for (int u = 0; u < PKP; ++u) {
            const f32x2 dx = __builtin_elementwise_fma(f32x2{r00, r00}, x[u], __builtin_elementwise_fma(f32x2{r01, r01}, y[u],
                             __builtin_elementwise_fma(f32x2{r02, r02}, z[u], f32x2{t0, t0} + rx[u])));
            const f32x2 dy = __builtin_elementwise_fma(f32x2{r10, r10}, x[u], __builtin_elementwise_fma(f32x2{r11, r11}, y[u],
                             __builtin_elementwise_fma(f32x2{r12, r12}, z[u], f32x2{t1, t1} + ry[u])));
            const f32x2 dz = __builtin_elementwise_fma(f32x2{r20, r20}, x[u], __builtin_elementwise_fma(f32x2{r21, r21}, y[u],
                             __builtin_elementwise_fma(f32x2{r22, r22}, z[u], f32x2{t2, t2} + rz[u])));
            const f32x2 d2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
            const unsigned long long sure0 = __ballot(d2.x < lo_s), sure1 = __ballot(d2.y < lo_s);
            amb[u][0] = __ballot(d2.x < hi_s) ^ sure0;                  // "sure" is a subset of "maybe" (lo_s < hi_s)
            amb[u][1] = __ballot(d2.y < hi_s) ^ sure1;
            csum += __popcll(sure0) + __popcll(sure1);
            amb_any |= amb[u][0] | amb[u][1];
          }
        } else {
#pragma unroll
          for (int u = 0; u < PKP; ++u) amb[u][0] = amb[u][1] = ~0ull;
          amb_any = ~0ull;
        }
        if (amb_any) {                                                   // wave-uniform, rare: the undecided residuals in fp64
          const double* __restrict__ T = xf + (size_t)(sbase + s) * 12;
#pragma unroll
          for (int u = 0; u < PKP; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              if (!amb[u][h]) continue;
              const int i = i0 + (2 * u + h) * 64 + lane;
              const bool in = ((amb[u][h] >> lane) & 1) && i < n && inlier_fp64(rec, i, T, thr2);
              csum += __popcll(__ballot(in));
            }
        }
        // csum and s are wave-uniform (SGPRs); gfx9 allows one SGPR per VOP3, so the lane select travels in m0 (saved and restored:
        // the compiler does not track m0 through inline assembly)
        int m0_keep;
        asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                     : "+v"(blkcnt), "=&s"(m0_keep) : "s"(csum), "s"(s));
      }
      mycnt += blkcnt;
      if (a.count_bound && publisher) {                                   // the group's best count so far, for everyone's bound
        int wm = mycnt;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) wm = max(wm, __shfl_xor(wm, d, 64));
        if (lane == 0 && wm > known && wm > __hip_atomic_load(bestp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(bestp, wm);
      }
    }
    if (lane < gs) cnts[sbase + lane] = mycnt;
    best = mycnt > best ? mycnt : best;   // lanes >= gs hold 0
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d, 64));
  if (lane == 0 && best > 0) atomicMax(a.n_surv + c * CNT_STRIDE + 1, best);
}

// residual sweep of one hypothesis by one wave: inlier count and sum of squared inlier residuals (lane 0)
__device__ inline void sweep(const float* __restrict__ rec, int n, const double R[3][3], const double t[3], double thr2,
                             int lane, int& cnt_out, double& err2_out) {
  int cnt = 0;
  double err2 = 0;
  for (int i = lane; i < n; i += 64) {
    const float2* r = reinterpret_cast<const float2*>(rec + (size_t)i * 6);
    const float2 p0 = r[0], p1 = r[1], p2 = r[2];
    const double x = p0.x, y = p0.y, z = p1.x;
    const double dx = fma(R[0][0], x, fma(R[0][1], y, fma(R[0][2], z, t[0] - (double)p1.y)));
    const double dy = fma(R[1][0], x, fma(R[1][1], y, fma(R[1][2], z, t[1] - (double)p2.x)));
    const double dz = fma(R[2][0], x, fma(R[2][1], y, fma(R[2][2], z, t[2] - (double)p2.y)));
    const double d2 = fma(dx, dx, fma(dy, dy, dz * dz));   // the same expression as k_count: the counts must agree
    if (d2 < thr2) { ++cnt; err2 += d2; }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_down(cnt, d, 64);
  cnt_out = cnt;
  err2_out = wave_sum(err2);
}

// survivors beyond the transform store (more than cap_t per pair): one wave per survivor re-derives the transform
__global__ __launch_bounds__(256) void k_count_overflow(PairArgs a) {
  const int c = blockIdx.y, b = a.pair0 + c;
  const int s0 = a.s0[c], n = a.n[c];
  const float* rec = a.rec + (size_t)s0 * 6;
  const unsigned long long base = pair_base(a.seed, b);
  const int lane = threadIdx.x & 63;
  const int ns = a.n_surv[c * CNT_STRIDE];
  const int* surv = a.surv + (size_t)c * a.H;
  int* cnts = a.cnts + (size_t)c * a.H;
  const double thr2 = thr2_of(a.max_dist);
  int best = 0;
  for (int sidx = a.cap_t + blockIdx.x * 4 + (threadIdx.x >> 6); sidx < ns; sidx += gridDim.x * 4) {
    double R[3][3], t[3], err2;
    int cnt;
    hypothesis(rec, (unsigned)n, base, (unsigned)surv[sidx], (double)a.edge_sim, (double)a.max_dist, R, t);
    sweep(rec, n, R, t, thr2, lane, cnt, err2);
    cnt = __shfl(cnt, 0, 64);
    if (lane == 0) cnts[sidx] = cnt;
    best = cnt > best ? cnt : best;
  }
  if (lane == 0 && best > 0) atomicMax(a.n_surv + c * CNT_STRIDE + 1, best);
}

// inlier RMSE (fp32 bits) of the survivors that reach the largest count
__global__ __launch_bounds__(256) void k_rmse(PairArgs a) {
  const int c = blockIdx.y, b = a.pair0 + c;
  const int s0 = a.s0[c], n = a.n[c];
  const float* rec = a.rec + (size_t)s0 * 6;
  const unsigned long long base = pair_base(a.seed, b);
  const int lane = threadIdx.x & 63;
  const int ns = a.n_surv[c * CNT_STRIDE], best = a.n_surv[c * CNT_STRIDE + 1];
  const int* surv = a.surv + (size_t)c * a.H;
  const int* cnts = a.cnts + (size_t)c * a.H;
  unsigned int* rmse = a.rmse + (size_t)c * a.H;
  const double thr2 = thr2_of(a.max_dist);
  // every wave scans 64 counts at a time (coalesced) and sweeps only the survivors at the largest count
  for (int s64 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; s64 < ns; s64 += gridDim.x * 4 * 64) {
    unsigned long long cand = __ballot(s64 + lane < ns && cnts[s64 + lane] == best);
    while (cand) {   // wave-uniform
      const int sidx = s64 + __builtin_ctzll(cand);
      cand &= cand - 1;
      double R[3][3], t[3], err2;
      int cnt;
      if (sidx < a.cap_t) {               // the stored transform, so that this sweep sees what k_count saw
        const double* x = a.xf + ((size_t)c * a.cap_t + sidx) * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i) { R[i][0] = x[3 * i]; R[i][1] = x[3 * i + 1]; R[i][2] = x[3 * i + 2]; t[i] = x[9 + i]; }
      } else {
        hypothesis(rec, (unsigned)n, base, (unsigned)surv[sidx], (double)a.edge_sim, (double)a.max_dist, R, t);
      }
      sweep(rec, n, R, t, thr2, lane, cnt, err2);
      if (lane == 0) rmse[sidx] = __float_as_uint(cnt > 0 ? (float)sqrt(err2 / cnt) : __builtin_inff());
    }
  }
}

// key: (inliers << 32) | ~bits(rmse_f32): larger is better; ties on the key are broken by lower h
__global__ __launch_bounds__(1024) void k_select(PairArgs a, eyoc_ransac_result* __restrict__ results) {
  __shared__ unsigned long long bk[16];
  __shared__ int bh[16];
  const int c = blockIdx.x, b = a.pair0 + c;
  const int s0 = a.s0[c], n = a.n[c];
  const float* rec = a.rec + (size_t)s0 * 6;
  const int ns = a.n_surv[c * CNT_STRIDE], best = a.n_surv[c * CNT_STRIDE + 1];
  const int* surv = a.surv + (size_t)c * a.H;
  const int* cnts = a.cnts + (size_t)c * a.H;
  const unsigned int* rmse = a.rmse + (size_t)c * a.H;
  eyoc_ransac_result* out = results + b;
  unsigned long long best_k = 0;
  int best_h = 0x7FFFFFFF;
  for (int i = threadIdx.x; i < ns; i += 1024) {
    if (cnts[i] != best) continue;
    const unsigned long long k = ((unsigned long long)(unsigned)best << 32) | (unsigned long long)(~rmse[i]);
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

namespace {

// scratch layout of one launch chunk (`chunk` pairs of a batch holding `total` correspondences, H hypotheses per pair):
// survivor lists are sized for the worst case (every hypothesis survives) - 12 bytes per hypothesis and pair - and
// transforms are stored for the first cap_t survivors of a pair (96 B each; 2^20 = the survivors of an inlier ratio of 0.7);
// the rest - only ever reached by degenerate inputs - are re-derived by k_count_overflow
struct RansacLayout {
  size_t off_cnt, off_rec, off_surv, off_cnts, off_rmse, off_xf, off_rs, off_rr, off_be, off_pm, bytes;
};
// eyoc_ransac_transform_store / eyoc_ransac_select_pruning: test switches of the ctx (eyoc_ctx::Knobs).  Every entry point reads each
// ONCE (a snapshot that its layout, its chunk size and its kernels all use)
inline int ransac_cap_t(int H, int store) { return H < store ? H : store; }
RansacLayout ransac_layout(int chunk, int total, int H, int store) {
  RansacLayout l;
  const int cap_t = ransac_cap_t(H, store);
  l.off_cnt = 0;
  l.off_rec = align_up((size_t)chunk * CNT_STRIDE * 4);
  l.off_surv = align_up(l.off_rec + (size_t)total * 24);
  l.off_cnts = align_up(l.off_surv + (size_t)chunk * H * 4);
  l.off_rmse = align_up(l.off_cnts + (size_t)chunk * H * 4);
  l.off_xf = align_up(l.off_rmse + (size_t)chunk * H * 4);
  l.off_rs = align_up(l.off_xf + (size_t)chunk * cap_t * 96);
  l.off_rr = align_up(l.off_rs + (size_t)total * 24);
  l.off_be = align_up(l.off_rr + (size_t)total * 16);
  l.off_pm = align_up(l.off_be + (size_t)chunk * (NBUCKET + 1) * 4);
  l.bytes = align_up(l.off_pm + (size_t)chunk * 8);
  return l;
}
// the largest chunk (n_pairs capped at CHUNK, then halved) whose layout stays within `budget` bytes; never below one pair
int ransac_pick_chunk(int n_pairs, int total, int H, size_t budget, int store) {
  int chunk = n_pairs < CHUNK ? n_pairs : CHUNK;
  while (chunk > 1 && ransac_layout(chunk, total, H, store).bytes > budget) chunk >>= 1;
  return chunk;
}

int ransac_run(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int64_t* corr_tgt_dev, const int32_t* seg_src_host,
               const int32_t* seg_tgt_host, int n_pairs, const eyoc_ransac_params* p, eyoc_ransac_result* results_dev, char* sc,
               int chunk, int max_n, int store, hipStream_t st) {
  const int H = p->max_iteration;
  const int total = seg_src_host[n_pairs];
  const int cap_t = ransac_cap_t(H, store);
  const RansacLayout l = ransac_layout(chunk, total, H, store);
  PairArgs a;
  a.rec = (float*)(sc + l.off_rec); a.seed = p->seed; a.H = H; a.edge_sim = p->edge_similarity; a.max_dist = p->max_distance;
  a.n_surv = (int*)(sc + l.off_cnt); a.surv = (int*)(sc + l.off_surv); a.cnts = (int*)(sc + l.off_cnts);
  a.rmse = (unsigned int*)(sc + l.off_rmse); a.xf = (double*)(sc + l.off_xf); a.cap_t = cap_t;
  a.rec_sorted = (float*)(sc + l.off_rs); a.rr_sorted = (float*)(sc + l.off_rr); a.bucket_end = (int*)(sc + l.off_be); a.pmax = (double*)(sc + l.off_pm);
  const int pruned = ctx->knobs.ransac_prune && max_n <= 8192 ? 1 : 0;
  const int want_bound = ctx->knobs.ransac_prune >= 2 ? 1 : 0;
  const bool in_lds = max_n <= LDS_RECORDS;
  const size_t lds_bytes = in_lds ? (size_t)max_n * 24 : 0;
  if (in_lds) {
    if (!ctx->ransac_attr_set) {
      EYOC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_generate<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         LDS_RECORDS * 24));
      ctx->ransac_attr_set = true;
    }
  }
  for (int p0 = 0; p0 < n_pairs; p0 += chunk) {
    const int nc = n_pairs - p0 < chunk ? n_pairs - p0 : chunk;
    a.pair0 = p0;
    // the count bound polls ONE word per pair from every wave and block: 100 k agent-scope loads per launch whatever the number of pairs
    // (the groups shrink as the pairs get fewer), and loads of one address are served one after the other - spread over 64 words they
    // cost nothing that shows, on the 8 words of an 8-pair chunk they cost 0.55 ms (k_count 0.36 -> 0.93 ms): small chunks count like round 5
    a.count_bound = want_bound && nc >= 32 ? 1 : 0;                       // (measured: 16 pairs 0.87 -> 1.03 ms with the bound, 32 pairs 1.50 -> 1.32, 64 pairs 3.1 -> 2.5)
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
    hipLaunchKernelGGL(k_fit, dim3(2048 / nc > 8 ? 2048 / nc : 8, nc), dim3(256), 0, st, a);
    if (pruned) hipLaunchKernelGGL(k_bucket, dim3(nc), dim3(256), 0, st, a);
    constexpr int KC_BLOCKS = 4096;   // workgroups of the count over the pairs of a launch (2048: 1.6 rounds of the 1280 the chip holds - measured 3.5 vs 3.3 ms in round 4; with the round-6 count bound 1.22 vs 1.19 ms, 8192: 1.59)
    if (pruned) hipLaunchKernelGGL(k_count, dim3(KC_BLOCKS / nc, nc), dim3(256), 0, st, a, (const double*)a.xf);
    else hipLaunchKernelGGL(k_count_fp64, dim3(KC_BLOCKS / nc, nc), dim3(256), 0, st, a, (const double*)a.xf, pruned);
    if (H > cap_t) hipLaunchKernelGGL(k_count_overflow, dim3(256, nc), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_rmse, dim3(64, nc), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_select, dim3(nc), dim3(1024), 0, st, a, results_dev);
  }
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

int ransac_validate(const float* src_dev, const float* tgt_dev, const int64_t* corr_tgt_dev, const int32_t* seg_src_host,
                    const int32_t* seg_tgt_host, int n_pairs, const eyoc_ransac_params* p, eyoc_ransac_result* results_dev, int* max_n) {
  EYOC_REQUIRE(src_dev && tgt_dev && corr_tgt_dev && seg_src_host && seg_tgt_host && p && results_dev, EYOC_ERR_INVALID,
               "eyoc_ransac_batched: NULL argument");
  EYOC_REQUIRE(n_pairs >= 1, EYOC_ERR_INVALID, "eyoc_ransac_batched: n_pairs %d", n_pairs);
  EYOC_REQUIRE(p->max_iteration >= 1, EYOC_ERR_INVALID, "eyoc_ransac: max_iteration %d", p->max_iteration);
  *max_n = 0;
  for (int b = 0; b < n_pairs; ++b) {
    const int n = seg_src_host[b + 1] - seg_src_host[b];
    EYOC_REQUIRE(n >= 4, EYOC_ERR_INVALID, "eyoc_ransac: need at least 4 correspondences, got %d (pair %d)", n, b);
    *max_n = n > *max_n ? n : *max_n;
  }
  return EYOC_OK;
}

}  // namespace

extern "C" int eyoc_ransac_transform_store(eyoc_ctx* ctx, int survivors) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.ransac_store;
  if (survivors >= 1) ctx->knobs.ransac_store = survivors;
  return prev;
}

extern "C" int eyoc_ransac_select_pruning(eyoc_ctx* ctx, int on) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.ransac_prune;
  if (on >= 0 && on <= 2) ctx->knobs.ransac_prune = on;
  return prev;
}

extern "C" size_t eyoc_ransac_workspace_bytes(const eyoc_ctx* ctx, int n_pairs, int total_corr, int max_iteration, size_t budget_bytes) {
  if (n_pairs < 1 || total_corr < 0 || max_iteration < 1) return 0;
  const int store = knobs_of(ctx).ransac_store;
  const int chunk = ransac_pick_chunk(n_pairs, total_corr, max_iteration, budget_bytes ? budget_bytes : ~(size_t)0, store);
  return ransac_layout(chunk, total_corr, max_iteration, store).bytes;
}

extern "C" int eyoc_ransac_batched_ws(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int64_t* corr_tgt_dev,
                                      const int32_t* seg_src_host, const int32_t* seg_tgt_host, int n_pairs,
                                      const eyoc_ransac_params* p, eyoc_ransac_result* results_dev, void* workspace_dev,
                                      size_t workspace_bytes, void* stream) {
  EYOC_REQUIRE(ctx && workspace_dev, EYOC_ERR_INVALID, "eyoc_ransac_batched_ws: NULL argument");
  EYOC_REQUIRE(((uintptr_t)workspace_dev & 255) == 0, EYOC_ERR_INVALID, "eyoc_ransac_batched_ws: workspace must be 256-byte aligned");
  int max_n = 0;
  int rc = ransac_validate(src_dev, tgt_dev, corr_tgt_dev, seg_src_host, seg_tgt_host, n_pairs, p, results_dev, &max_n);
  if (rc) return rc;
  const int total = seg_src_host[n_pairs];
  const int store = ctx->knobs.ransac_store;
  const int chunk = ransac_pick_chunk(n_pairs, total, p->max_iteration, workspace_bytes, store);
  const size_t need = ransac_layout(chunk, total, p->max_iteration, store).bytes;
  EYOC_REQUIRE(need <= workspace_bytes, EYOC_ERR_WORKSPACE,
               "eyoc_ransac_batched_ws: workspace %zu < %zu bytes (one pair per launch; eyoc_ransac_workspace_bytes)", workspace_bytes, need);
  return ransac_run(ctx, src_dev, tgt_dev, corr_tgt_dev, seg_src_host, seg_tgt_host, n_pairs, p, results_dev, (char*)workspace_dev, chunk,
                    max_n, store, (hipStream_t)stream);
}

// the same with scratch the context owns (grow-only): the chunk is sized from the device's FREE memory - at most a quarter
// of what is free plus what the context already holds, and never more than 16 GB - and halved again when the allocation
// fails all the same (another process on the GPU); a one-pair chunk (144 MB at 4 M hypotheses) is the floor
extern "C" int eyoc_ransac_batched(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int64_t* corr_tgt_dev,
                                   const int32_t* seg_src_host, const int32_t* seg_tgt_host, int n_pairs,
                                   const eyoc_ransac_params* p, eyoc_ransac_result* results_dev, void* stream) {
  EYOC_REQUIRE(ctx, EYOC_ERR_INVALID, "eyoc_ransac_batched: NULL argument");
  int max_n = 0;
  int rc = ransac_validate(src_dev, tgt_dev, corr_tgt_dev, seg_src_host, seg_tgt_host, n_pairs, p, results_dev, &max_n);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int total = seg_src_host[n_pairs];
  const int store = ctx->knobs.ransac_store;
  // the scratch already there is free to use: the device is only asked how much memory it has left (a driver round trip) when the
  // largest chunk does not fit into it - i.e. on the first call of a batch shape, not on the hot path
  int chunk = ransac_pick_chunk(n_pairs, total, p->max_iteration, ctx->scratch_bytes, store);
  const int want = n_pairs < CHUNK ? n_pairs : CHUNK;
  if (chunk < want || ransac_layout(chunk, total, p->max_iteration, store).bytes > ctx->scratch_bytes) {
    size_t free_b = 0, total_b = 0;
    EYOC_CHECK_HIP(hipMemGetInfo(&free_b, &total_b));
    size_t budget = (free_b + ctx->scratch_bytes) / 4;
    if (budget > ((size_t)16 << 30)) budget = (size_t)16 << 30;
    if (budget < ctx->scratch_bytes) budget = ctx->scratch_bytes;
    chunk = ransac_pick_chunk(n_pairs, total, p->max_iteration, budget, store);
  }
  for (;;) {
    rc = ctx->ensure_scratch(ransac_layout(chunk, total, p->max_iteration, store).bytes, st);
    if (rc == EYOC_OK) break;
    if (chunk == 1) return rc;
    (void)hipGetLastError();                                             // the failed hipMalloc's sticky error
    chunk >>= 1;
  }
  return ransac_run(ctx, src_dev, tgt_dev, corr_tgt_dev, seg_src_host, seg_tgt_host, n_pairs, p, results_dev, (char*)ctx->scratch, chunk,
                    max_n, store, st);
}

extern "C" int eyoc_ransac(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int64_t* corr_tgt_dev, int n,
                           const eyoc_ransac_params* p, eyoc_ransac_result* result_dev, void* stream) {
  EYOC_REQUIRE(n >= 4, EYOC_ERR_INVALID, "eyoc_ransac: need at least 4 correspondences, got %d", n);
  const int32_t seg_src[2] = {0, n}, seg_tgt[2] = {0, 0x7FFFFFFF};
  return eyoc_ransac_batched(ctx, src_dev, tgt_dev, corr_tgt_dev, seg_src, seg_tgt, 1, p, result_dev, stream);
}
