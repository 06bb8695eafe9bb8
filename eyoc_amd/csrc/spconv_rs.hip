// Sparse convolution, row-stationary (gfx950, SPLIT16 arithmetic only).
//
// The wave-private kernel (spconv_wave.hip) compacts the (input row, output row) pairs of every kernel offset so
// that the matrix work is proportional to the true pair count, and pays for it around the MFMAs: pair lists, per-unit
// record reads, and a 128-bit LDS read-add-write per (chunk, 16-channel tile) because the pair -> output-row mapping
// changes from offset to offset.  With fp32 MFMAs that was the right trade - the matrix pipe was the bound.  With
// SPLIT16 (three fp16 MFMAs of 16 cycles per product block) the matrix pipe is ~16 % busy (profiles/r2_mfma_counters.csv)
// and the kernel is bound by everything else.  So this kernel spends matrix work to delete the rest:
//
//   * a wave owns 64 output rows x 64 (or 32) output channels and keeps the WHOLE accumulator tile in registers
//     (4 row chunks x 4 channel tiles x 4 VGPRs) for all 27 offsets: MFMA column j of chunk c is ALWAYS output row
//     16 c + j, so nothing is ever scattered - no LDS, no pair lists, no compaction, no barrier;
//   * a row that has no neighbour at the current offset contributes a zero operand: its gather goes through a buffer
//     resource with an out-of-range offset, which the hardware answers with zeros without touching memory;
//   * offsets where none of the 64 rows has a neighbour are skipped (wave-uniform), so are 16-row chunks; with the
//     pattern-sorted tiling orders a transposed / strided convolution touches 3-8 of its 27 offsets per tile;
//   * a stage = (offset, 32-channel block): 8 gather loads per lane and 48 MFMAs; the gather of stage s + 1 and the
//     rulebook entries of the offset after that are in flight while stage s multiplies;
//   * every wave does exactly the same work per stage, whatever its rows look like - so the NW = 4 waves of a workgroup
//     (four consecutive 64-row tiles) can march in lock step and SHARE the weights: the 8 KB weight slice of a stage is
//     copied global -> LDS once per workgroup (two 16-byte pieces per thread, two stages ahead, three-slot ring, one
//     barrier per stage) and every wave reads its MFMA operand fragments from LDS.  Measured reason: with per-wave
//     weight loads both this kernel and the wave-private one sat at ~50 B/clk/CU of vector-memory traffic (the L1 /
//     texture path peaks at 64), 3/4 of it weights - 432 KB of weight fragments against 151 KB of gathered rows per
//     64-row tile of a 64 -> 64 layer.
//
// On a stride-1 layer ~2.9x as many MFMAs are issued as pairs exist (9 of 27 neighbours occupied) - matrix time the
// SPLIT16 pipe has to spare.  Summation order per output element: offsets ascending, channels ascending - fixed, so
// results are bit-reproducible; they differ from the wave-private kernel's only in rounding (same products, other order).
#include "spconv.h"

using namespace eyoc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned int OOB = 0xFFFFFFF0u;   // beyond every buffer's num_records: the load returns zeros

// NTW: 16-channel tiles per wave, 4 (64 output channels) or 2 (32).  CC: C_in slice of the packed weight layout (spconv_cc)
constexpr int NW = 4;            // waves per workgroup = consecutive 64-row tiles that share their weight fragments
constexpr int RING = 3;          // weight slices in LDS: stage s multiplies, s + 1 is being written, s + 2 is in flight

template <int NTW, int CC>
__global__ __launch_bounds__(NW * 64, 2) void spconv_rs_kernel(SpconvArgs a) {
  constexpr int CTW = NTW * 16, NC = 4;
  constexpr int SLICE_BYTES = NTW * 2 * 64 * 16;               // weight fragments of one stage: [t][hi/lo][lane] x 16 B
  constexpr int PIECES = SLICE_BYTES / 16 / (NW * 64);         // 16-byte pieces each thread copies per stage (2 or 1)
  __shared__ __attribute__((aligned(16))) unsigned char wring[RING][SLICE_BYTES];
  __shared__ unsigned int wg_mask;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane >> 4, j = lane & 15;
  const int n_cg = a.cout / CTW;
  // heavy tiles first (the pattern-sorted orders end with the rows that have the most neighbours), XCD round robin
  const int wg = (int)gridDim.x - 1 - (int)blockIdx.x;
  const int rgw = wg / n_cg, cg = wg - rgw * n_cg;
  const int row0 = (rgw * NW + wave) * 64;                     // may lie beyond n_out: the wave then multiplies zeros (no early exit: barriers)
  const int ct0 = cg * CTW;
  const int CT = a.cout >= 128 ? 128 : a.cout;                 // column slice of the packed layout (spconv_ct)
  const int n_slices = a.cout / CT, slice = ct0 / CT, nt0 = (ct0 - slice * CT) / 16;
  constexpr int JQ = CC / 16;
  const int ncc = a.cin / CC;
  const int nqb = a.cin / 32;                                  // 32-channel blocks
  const int K = a.K;

  // this lane's output row of every chunk (every 16-lane group g holds the same four rows)
  int grow[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int r = row0 + 16 * c + j;
    grow[c] = r < a.n_out ? (a.perm ? a.perm[r] : r) : -1;
  }
  auto load_idx = [&](int k, int (&idx)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (a.nbr) idx[c] = (k < K && grow[c] >= 0) ? a.nbr[(size_t)k * a.n_out + grow[c]] : -1;
      else idx[c] = (k == 0) ? grow[c] : -1;                   // identity map (1x1 convolution)
    }
  };
  // occupied offsets of this tile (bit k).  The four 16-lane groups hold the same rows, so group g checks the offsets
  // k = g, g + 4, ...: 7 x 4 rulebook loads per lane, all issued before the first use - one memory latency for the whole
  // sweep - and the sweep warms the cache lines the main loop reads again
  unsigned int kmask = 0;
  {
    int v[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      int idx[NC];
      load_idx(4 * r + g, idx);
      v[r] = (idx[0] >= 0 || idx[1] >= 0 || idx[2] >= 0 || idx[3] >= 0) ? 1 : 0;
    }
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const unsigned long long m = __ballot(v[r] != 0);
#pragma unroll
      for (int gg = 0; gg < 4; ++gg)
        if ((m >> (16 * gg)) & 0xFFFFull) kmask |= 1u << (4 * r + gg);
    }
  }
  kmask = __builtin_amdgcn_readfirstlane(kmask) & (K >= 32 ? 0xFFFFFFFFu : ((1u << K) - 1u));
  // the workgroup walks the UNION of its waves' occupied offsets (a wave without neighbours at one of them multiplies zeros)
  if (threadIdx.x == 0) wg_mask = 0;
  __syncthreads();
  if (lane == 0 && kmask) atomicOr(&wg_mask, kmask);
  __syncthreads();
  kmask = wg_mask;

  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, K * a.cin * a.cout * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, 0xFFFFFFE0u, 0x00020000);
  const int tile4 = CC * CT / 4;                               // float4s of one packed (k, slice, cc) weight tile
  const unsigned int ld_bytes = (unsigned)a.ld_in * 4u;

  f32x4 acc[NC][NTW];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // gathered operands of the stage being multiplied / the stage in flight
  float4 XA[NC][2], XB[NC][2];
  auto gather = [&](int qb, const int (&idx)[NC], float4 (&X)[NC][2]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      // channels 32 qb + 8 g .. +7 of the row: 16 B of hi halves in the first 64 bytes of the block's line, 16 B of lo
      // halves in the second - the four lanes of a row read 64 contiguous bytes per instruction
      const unsigned int off = idx[c] >= 0 ? (unsigned)idx[c] * ld_bytes + (unsigned)(qb * 128 + g * 16) : OOB;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, idx[c] >= 0 ? off + 64u * p : OOB, 0, 0);
        X[c][p] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
    }
  };
  // this thread's pieces of the weight slice of stage (k, qb): piece pi = fragment f = pi / 64 (t = f / 2, hi / lo = f % 2),
  // lane pi % 64 - global -> registers now, registers -> LDS one stage later
  auto fetch_w = [&](int k, int qb, float4 (&P)[PIECES]) {
    const int cc = (qb * 32) / CC, qp = ((qb * 32) % CC) / 32;
    const int wbase = __builtin_amdgcn_readfirstlane((((k * n_slices + slice) * ncc + cc) * tile4 + (nt0 * JQ + 2 * qp) * 64) * 16);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int pi = (int)threadIdx.x + i * NW * 64, f = pi >> 6;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (pi & 63) * 16 + ((f >> 1) * JQ + (f & 1)) * 1024, wbase, 0);
      P[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto store_w = [&](int slot, const float4 (&P)[PIECES]) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i)
      *reinterpret_cast<float4*>(wring[slot] + ((int)threadIdx.x + i * NW * 64) * 16) = P[i];
  };
  auto multiply = [&](int slot, const float4 (&X)[NC][2]) {
    float4 W[NTW][2];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) W[t][p] = *reinterpret_cast<const float4*>(wring[slot] + ((t * 2 + p) * 64 + lane) * 16);
    // term outermost: the three MFMAs of one accumulator (hi*hi, hi*lo, lo*hi) are NC * NTW issues apart, so none
    // waits for its predecessor's result
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const half8_t xv = __builtin_bit_cast(half8_t, X[c][term == 1 ? 1 : 0]);
#pragma unroll
        for (int t = 0; t < NTW; ++t)
          acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, W[t][term == 2 ? 1 : 0]), xv, acc[c][t], 0, 0, 0);
      }
  };

  // ---- the stage stream: (occupied offset k ascending, block qb ascending); every cursor below is wave-uniform
  if (kmask) {
    const int n_stages = __popc(kmask) * nqb;
    // three cursors over the offsets: the gather runs one stage ahead, the weight fetch two
    unsigned int rest_x = kmask, rest_w = kmask;
    int kx = __builtin_ctz(rest_x), qx = 0;          // stage whose gather is issued next
    rest_x &= rest_x - 1;
    int kw = __builtin_ctz(rest_w), qw = 0;          // stage whose weights are fetched next
    rest_w &= rest_w - 1;
    auto advance = [&](int& k, int& q, unsigned int& rest) {
      if (++q == nqb) { q = 0; k = rest ? __builtin_ctz(rest) : -1; rest &= rest - 1; }
    };
    int idx_cur[NC], idx_nxt[NC];
    load_idx(kx, idx_cur);
    int k_ahead = rest_x ? __builtin_ctz(rest_x) : -1;   // offset after kx: its rulebook entries travel one offset ahead
    if (k_ahead >= 0) load_idx(k_ahead, idx_nxt);
    float4 PA[PIECES], PB[PIECES];
    // prologue: weights of stage 0 -> ring[0], stage 1 -> registers, gather of stage 0
    fetch_w(kw, qw, PA); advance(kw, qw, rest_w);
    if (kw >= 0) { fetch_w(kw, qw, PB); advance(kw, qw, rest_w); }
    gather(qx, idx_cur, XA);
    {
      const int kprev = kx;
      advance(kx, qx, rest_x);
      if (kx != kprev && kx >= 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) idx_cur[c] = idx_nxt[c];
        k_ahead = rest_x ? __builtin_ctz(rest_x) : -1;
        if (k_ahead >= 0) load_idx(k_ahead, idx_nxt);
      }
    }
    store_w(0, PA);
    __syncthreads();
    // stage s: X_c holds its gather, ring[s % 3] its weights; P_s1 holds the weight pieces of stage s + 1
    auto step = [&](int s, float4 (&X_c)[NC][2], float4 (&X_n)[NC][2], float4 (&P_s1)[PIECES], float4 (&P_s2)[PIECES]) {
      if (s + 2 < n_stages) { fetch_w(kw, qw, P_s2); advance(kw, qw, rest_w); }      // global -> registers, two stages ahead
      if (s + 1 < n_stages) {
        gather(qx, idx_cur, X_n);                                                      // gather of stage s + 1
        const int kprev = kx;
        advance(kx, qx, rest_x);
        if (kx != kprev && kx >= 0) {                                                  // stage s + 2 opens a new offset
#pragma unroll
          for (int c = 0; c < NC; ++c) idx_cur[c] = idx_nxt[c];
          k_ahead = rest_x ? __builtin_ctz(rest_x) : -1;
          if (k_ahead >= 0) load_idx(k_ahead, idx_nxt);
        }
        store_w((s + 1) % RING, P_s1);                                                 // registers -> LDS, one stage ahead
      }
      multiply(s % RING, X_c);
      lds_barrier();     // ring[(s + 1) % 3] complete for everyone; ring[s % 3] free again from stage s + 2's store on.
                         // NOT __syncthreads(): that drains vmcnt and would wait for the gather and the weight pieces
                         // issued above, which are meant to land during the NEXT stage
    };
    for (int s = 0; s < n_stages; s += 2) {
      step(s, XA, XB, PB, PA);
      if (s + 1 < n_stages) step(s + 1, XB, XA, PA, PB);
    }
  }

  // ---- epilogue straight from the registers: lane (g, j) holds channels 16 t + 4 g .. +3 of row 16 c + j
  const float os = a.out_scale ? *a.out_scale : 1.0f;
  float mx = 0.f;
  const bool poisoned = !a.out_split && split16_poisoned(a.range);     // the network's fp32 output after an overflow upstream
  const float qnan = __builtin_nanf("");
  float4 b4[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
    b4[t] = a.bias ? *reinterpret_cast<const float4*>(a.bias + ct0 + 16 * t + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int o = grow[c];
    float4 v[NTW];
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      v[t] = make_float4(acc[c][t][0] * os + b4[t].x, acc[c][t][1] * os + b4[t].y, acc[c][t][2] * os + b4[t].z,
                         acc[c][t][3] * os + b4[t].w);
      ss += v[t].x * v[t].x + v[t].y * v[t].y + v[t].z * v[t].z + v[t].w * v[t].w;
    }
    if (a.l2norm) {                                            // the row's channels live in this lane and in lanes j + 16, 32, 48
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      const float nrm = sqrtf(ss);
#pragma unroll
      for (int t = 0; t < NTW; ++t) { v[t].x /= nrm; v[t].y /= nrm; v[t].z /= nrm; v[t].w /= nrm; }
    }
    if (o < 0) continue;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int ch = ct0 + 16 * t + 4 * g;
      if (!a.l2norm) {
        if (a.res) {
          const float4 q = split16_load4(a.res + (size_t)o * a.ld_res, ch);
          v[t].x += q.x; v[t].y += q.y; v[t].z += q.z; v[t].w += q.w;
        }
        if (a.relu) {
          v[t].x = fmaxf(v[t].x, 0.f); v[t].y = fmaxf(v[t].y, 0.f); v[t].z = fmaxf(v[t].z, 0.f); v[t].w = fmaxf(v[t].w, 0.f);
        }
      }
      const size_t oo = a.out_perm ? (size_t)a.out_perm[o] : (size_t)o;
      if (a.out_split) { split16_track(mx, v[t]); split16_store4(a.out + oo * a.ld_out, ch, v[t]); }
      else *reinterpret_cast<float4*>(a.out + oo * a.ld_out + ch) = poisoned ? make_float4(qnan, qnan, qnan, qnan) : v[t];
    }
  }
  if (a.out_split) split16_report(a.range, mx);
}

}  // namespace

namespace eyoc {

// the gather goes through one buffer resource (32-bit byte offsets): the input tensor must end below 4 GB
bool spconv_rs_fits(const SpconvArgs& a) {
  const long long rows = a.nbr ? (a.n_in > 0 ? a.n_in : (1ll << 24)) : a.n_out;
  return rows * a.ld_in * 4 < 0xFFFFFFE0ll;
}

// SPLIT16 layers only (a.math == 1).  l2norm needs the whole row in one wave: C_out <= 64.
int launch_spconv_rs(const SpconvArgs& a, hipStream_t st) {
  EYOC_REQUIRE(a.math == 1, EYOC_ERR_INVALID, "spconv_rs: split16 arithmetic only");
  EYOC_REQUIRE(!a.l2norm || a.cout <= 64, EYOC_ERR_INVALID, "spconv_rs: l2norm needs C_out <= 64");
  EYOC_REQUIRE((long long)a.K * a.cin * a.cout * 4 < 0x7FFFFFFFll, EYOC_ERR_INVALID, "spconv_rs: weight tensor too large");
  EYOC_REQUIRE(spconv_rs_fits(a), EYOC_ERR_INVALID, "spconv_rs: input tensor beyond the 4 GB a buffer resource addresses");
  const int ctw = a.cout >= 64 ? 64 : 32;
  const bool wide = spconv_cc(a.cin, a.cout) == 64;
  const long long wgs = (long long)cdiv(a.n_out, 64 * NW) * (a.cout / ctw);
  const dim3 grid((unsigned)wgs), block(NW * 64);
  if (ctw == 64) {
    if (wide) hipLaunchKernelGGL((spconv_rs_kernel<4, 64>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((spconv_rs_kernel<4, 32>), grid, block, 0, st, a);
  } else {
    if (wide) hipLaunchKernelGGL((spconv_rs_kernel<2, 64>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((spconv_rs_kernel<2, 32>), grid, block, 0, st, a);
  }
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

}  // namespace eyoc
