// Transposed 3^3 / stride-2 sparse convolution with a tile-local input stage (gfx950, SPLIT16 arithmetic, Z-ordered rows).
//
// A fine output row of the transposed convolution has 1, 2, 4 or 8 coarse neighbours - which of the 27 offsets can be
// occupied is decided by its parity class (on which axes it sits between two coarse voxels): 2.5 pairs per row on the
// bench geometry.  Walking all 27 offsets densely like the stride-1 staged kernel (spconv_st.hip) would multiply ten
// zeros for every product; the gathering kernels in pattern-sorted order (spconv_rs.hip) avoid that but fetch every
// (row, offset) pair from L2 / HBM - 10 GB for the 3.8 M-row 1 -> 0 layer whose distinct inputs are 0.8 GB, 1.6 ms at
// the HBM limit.  This kernel does both things right:
//
//   * tile = 256 consecutive (Z-ordered) output rows; their ~100-200 distinct coarse input rows are staged in LDS like in
//     spconv_st.hip - as many 32-channel blocks at once as the 640 row slots hold (all four of a 128-channel layer for most
//     tiles: one global -> LDS round trip per tile instead of four);
//   * INSIDE the tile the rows are sorted by their occupancy mask (k_local_rulebook_up: a 256-key bitonic sort), so a
//     16-row MFMA group holds rows of one or two patterns; the rulebook carries the union mask of every group and the
//     wave only multiplies the (group, offset) blocks whose mask bit is set (~5 of 27 instead of 27);
//     (the 27 offsets split into 8 disjoint sets by parity class, so class-sorted slots give every wave - 64 slots, 32
//     output channels - about 7 offsets to walk);
//   * the offset loop is dynamic (scalar find-first-bit over the wave's union mask), weights and rulebook entries of
//     the next occupied offset are prefetched while the current one multiplies; outputs go back to their rows through
//     the tile's slot -> row list.
#include <cstdlib>

#include "spconv.h"

using namespace eyoc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NW = 4, TILE = 256;                            // builder: one thread per row
constexpr int NWK = 8;                                      // kernel: 8 waves = 4 slot quarters x 2 output-channel halves
constexpr int XROWS = 640, UMAX = XROWS - 1, X_BYTES = XROWS * 128, NIT = XROWS / (8 * NWK);   // 80 KB: 2 workgroups = 16 waves per CU
// rulebook of one tile: int n_unique (-1: more than UMAX), pad[3]; int U[640]; int row[256] (output row of slot s, -1:
// none); unsigned gmask[16] (union of the occupancy masks of slots 16 g ..); uint2 loc[27][4][16]: entry (k, w, j) packs
// the LDS slots v = 8 l + (l & 7) of the neighbours at offset k of tile slots 64 w + 16 c + j, c = 0..3 (l = n_unique: none -
// the zero row behind the tile's rows in every block's region of the stage)
constexpr int OFF_U = 16, OFF_ROW = OFF_U + XROWS * 4, OFF_GM = OFF_ROW + TILE * 4, OFF_LOC = OFF_GM + 16 * 4;
constexpr int UP_LR_BYTES = OFF_LOC + 27 * 4 * 16 * 8;      // 17488
constexpr int HSLOTS = 4096;                                // >= 2 x (256 rows x 8 neighbours)

__global__ __launch_bounds__(256) void k_local_rulebook_up(const int32_t* __restrict__ nbr, int K, int n_out,
                                                           unsigned char* __restrict__ out, int* __restrict__ overflow) {
  __shared__ int hk[HSLOTS];
  __shared__ unsigned short hid[HSLOTS];
  __shared__ unsigned short ids_row[27][TILE];               // LDS slot of (offset, local row), UMAX = none
  __shared__ unsigned long long key[TILE];
  __shared__ int wave_cnt[NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x;
  const int r = (int)threadIdx.x;
  const int row = tile * TILE + r;
  for (int i = threadIdx.x; i < HSLOTS; i += 256) hk[i] = -1;
  __syncthreads();
  unsigned short slot[27];
  unsigned int mask = 0;
  int idxs[27];                                                      // all 27 loads first: behind the CAS loops each would wait alone
#pragma unroll
  for (int k = 0; k < 27; ++k) idxs[k] = (k < K && row < n_out) ? nbr[(size_t)k * n_out + row] : -1;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int idx = idxs[k];
    unsigned int s = 0xFFFFu;
    if (idx >= 0) {
      mask |= 1u << k;
      s = ((unsigned)idx * 2654435761u) >> 20;
      while (true) {
        const int prev = atomicCAS(&hk[s], -1, idx);
        if (prev == -1 || prev == idx) break;
        s = (s + 1) & (HSLOTS - 1);
      }
    }
    slot[k] = (unsigned short)s;
  }
  __syncthreads();
  // number the occupied hash slots in slot order (deterministic)
  constexpr int PER_WAVE = HSLOTS / NW;
  int cnt = 0;
  for (int i0 = 0; i0 < PER_WAVE; i0 += 64) cnt += __popcll(__ballot(hk[wave * PER_WAVE + i0 + lane] >= 0));
  if (lane == 0) wave_cnt[wave] = cnt;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < NW; ++w) { if (w < wave) base += wave_cnt[w]; total += wave_cnt[w]; }
  unsigned char* lr = out + (size_t)tile * UP_LR_BYTES;
  int* U = reinterpret_cast<int*>(lr + OFF_U);
  for (int i0 = 0; i0 < PER_WAVE; i0 += 64) {
    const int s = wave * PER_WAVE + i0 + lane;
    const int k_ = hk[s];
    const unsigned long long m = __ballot(k_ >= 0);
    const int id = base + __popcll(m & ((1ull << lane) - 1ull));
    if (k_ >= 0) {
      hid[s] = (unsigned short)id;
      if (id < UMAX) U[id] = k_;
    }
    base += __popcll(m);
  }
  if (threadIdx.x == 0) {
    reinterpret_cast<int*>(lr)[0] = total <= UMAX ? total : -1;
    if (total > UMAX) atomicAdd(overflow, 1);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int id = slot[k] != 0xFFFFu ? (int)hid[slot[k]] : total;     // no neighbour: the zero row right behind the tile's rows
    const int l = id < UMAX ? id : UMAX;
    ids_row[k][r] = (unsigned short)(l * 8 + (l & 7));
  }
  // sort the tile's rows by (parity class, occupancy mask): the 27 offsets split into 8 disjoint sets, one per class
  // (offset (dx, dy, dz) is only ever occupied for rows that sit between two coarse voxels on exactly the axes with
  // d != 0), so class-contiguous slots make every offset touch one short run of slots.  Rows past the end: mask 0, they
  // sort first and carry row -1.
  unsigned int cls = 0;
  if (mask) {
    const int k0 = __builtin_ctz(mask);
    cls = (k0 % 3 != 1 ? 1u : 0u) | ((k0 / 3) % 3 != 1 ? 2u : 0u) | (k0 / 9 != 1 ? 4u : 0u);
  }
  key[r] = ((unsigned long long)cls << 40) | ((unsigned long long)mask << 8) | (unsigned)r;
  __syncthreads();
  for (int kk = 2; kk <= TILE; kk <<= 1)
    for (int jj = kk >> 1; jj > 0; jj >>= 1) {
      const int p = r ^ jj;
      if (p > r) {
        const unsigned long long a = key[r], b = key[p];
        const bool up = (r & kk) == 0;
        if ((a > b) == up) { key[r] = b; key[p] = a; }
      }
      __syncthreads();
    }
  // thread s now owns slot s
  const unsigned long long mine = key[r];
  const int src = (int)(mine & 255u);
  const unsigned int smask = (unsigned int)(mine >> 8) & 0x7FFFFFFu;
  const int grow = tile * TILE + src;
  reinterpret_cast<int*>(lr + OFF_ROW)[r] = grow < n_out ? grow : -1;
  unsigned int gm = smask;
  gm |= __shfl_xor(gm, 1, 64); gm |= __shfl_xor(gm, 2, 64); gm |= __shfl_xor(gm, 4, 64); gm |= __shfl_xor(gm, 8, 64);
  if ((r & 15) == 0) reinterpret_cast<unsigned int*>(lr + OFF_GM)[r >> 4] = gm;
  unsigned short* loc = reinterpret_cast<unsigned short*>(lr + OFF_LOC);
  const int h = r >> 6, c = (r >> 4) & 3, j = r & 15;
#pragma unroll
  for (int k = 0; k < 27; ++k) loc[(((size_t)(k * 4 + h) * 16) + j) * 4 + c] = ids_row[k][src];
}

#ifdef EYOC_UP_TRACE
// diagnostics (scripts/trace_up.py): per workgroup of the 128 -> 64 layer {start, stage issued, stage landed, loops done, values ready, end, HW_ID}
constexpr int UP_TRACE_WGS = 16384, UP_TRACE_N = 8;
__device__ unsigned long long g_up_trace[UP_TRACE_WGS * UP_TRACE_N];
#define UP_STAMP(i) do { if (a.cin == 128 && a.cout == 64 && threadIdx.x == 0 && blockIdx.x < UP_TRACE_WGS) g_up_trace[blockIdx.x * UP_TRACE_N + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define UP_STAMP(i) do {} while (0)
#endif

template <int CC>
__global__ __launch_bounds__(NWK * 64, 4) void spconv_up_kernel(SpconvArgs a, const unsigned char* __restrict__ local, int n_tiles) {
  constexpr int NTW = 2, CTG = 64, NG = 4;                             // per wave: 64 slots (4 groups) x 32 output channels
  extern __shared__ __attribute__((aligned(16))) unsigned char xs[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane >> 4, j = lane & 15;
  const int n_cg = a.cout / CTG;
  const int xcd = (int)blockIdx.x & 7, per = 8 / n_cg;
  const int cg = xcd % n_cg, tile = ((int)blockIdx.x >> 3) * per + xcd / n_cg;
  if (tile >= n_tiles) return;
  const int sq = wave >> 1;                                            // slot quarter of the tile
  const int ct0 = cg * CTG + (wave & 1) * 32;
  const int CT = a.cout >= 128 ? 128 : a.cout;
  const int n_slices = a.cout / CT, slice = ct0 / CT, nt0 = (ct0 - slice * CT) / 16;
  constexpr int JQ = CC / 16;
  const int ncc = a.cin / CC;
  const int nqb = a.cin / 32;
  constexpr int K = 27;

  UP_STAMP(0);
  // L2 warm-up of the record (header, row list, slot entries: 17 KB, written by the builder long ago) of the tile that will
  // most likely take this workgroup's slot next - 512 workgroups further on, same XCD: a tile spends a third of its life getting
  // its header, its row list and its rows (scripts/trace_up.py), the first two as dependent HBM round trips
  int warm = 0;
  {
    const int tn = tile + 64 * per;
    if (tn < n_tiles && (int)threadIdx.x * 128 < UP_LR_BYTES)
      warm = *reinterpret_cast<const int*>(local + (size_t)tn * UP_LR_BYTES + threadIdx.x * 128);
  }
  const unsigned char* lr = local + (size_t)tile * UP_LR_BYTES;
  const int* __restrict__ U = reinterpret_cast<const int*>(lr + OFF_U);
  // the tile's row numbers: requested FIRST - before the header is waited for (the list has XROWS entries whatever n_unique
  // says) - and kept in registers over the 32-channel blocks: between two rounds the stage is then a barrier and the DMA issue only
  int Ureg[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) Ureg[it] = U[(it * NWK + wave) * 8 + (lane >> 3)];
  const int n_u = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(lr)[0]);
  const int* __restrict__ rowp = reinterpret_cast<const int*>(lr + OFF_ROW) + sq * 64;
  const unsigned int* __restrict__ gmp = reinterpret_cast<const unsigned int*>(lr + OFF_GM) + sq * 4;
  const uint2* __restrict__ locp = reinterpret_cast<const uint2*>(lr + OFF_LOC) + sq * 16 + j;     // entry k at [k * 64]
  unsigned int gm[NG], wmask = 0;
#pragma unroll
  for (int c = 0; c < NG; ++c) { gm[c] = __builtin_amdgcn_readfirstlane(gmp[c]); wmask |= gm[c]; }
  // The stage holds SEVERAL 32-channel blocks of the tile's rows at once when they fit (a tile has ~150 distinct input rows, the
  // stage 640 row slots): region b = rows of block b + one zero row (what "no neighbour" entries point at), padded to the swizzle
  // period.  A 128-channel layer then pays one or two global -> LDS round trips per tile instead of four - the offset loop of one
  // block (4-9 offsets) is as short as such a round trip, and with two workgroups per CU nothing else hides it.
  const int srows = (n_u + 8) & ~7;
  const int nb_max = min(XROWS / srows, nqb);
  UP_STAMP(7);
  if ((int)threadIdx.x < 8 * nb_max)
    *reinterpret_cast<float4*>(xs + (((int)threadIdx.x >> 3) * srows + n_u) * 128 + (threadIdx.x & 7) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, K * a.cin * a.cout * 4, 0x00020000);
  const int tile4 = CC * CT / 4;
  f32x4 acc[NG][NTW];
#pragma unroll
  for (int c = 0; c < NG; ++c)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // MFMA row 4 q + r of channel tile t is made channel 8 q + 4 t + r of the wave's 32 (instead of 16 t + 4 q + r), so that a
  // lane's two accumulator tuples are 8 CONSECUTIVE channels = 16-byte epilogue accesses: lane (g, m) of tile t takes its
  // weight row from the packed fragment of tile ch >> 4 at lane position 16 g + (ch & 15), ch = 8 (m >> 2) + 4 t + (m & 3)
  int lane_off[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int ch = 8 * (j >> 2) + 4 * t + (j & 3);
    lane_off[t] = (ch >> 4) * JQ * 1024 + (g * 16 + (ch & 15)) * 16;
  }
  auto load_w = [&](int k, int qb, float4 (&W)[NTW][2]) {
    const int cc = (qb * 32) / CC, qp = ((qb * 32) % CC) / 32;
    const int wbase = __builtin_amdgcn_readfirstlane((((k * n_slices + slice) * ncc + cc) * tile4 + (nt0 * JQ + 2 * qp) * 64) * 16);
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_off[t] + p * 1024, wbase, 0);
        W[t][p] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
  };
  auto stage = [&](int qb0, int nb) {
    for (int b = 0; b < nb; ++b)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int l0 = (it * NWK + wave) * 8;
        if (l0 < n_u) {                                              // wave-uniform
          const int l = l0 + (lane >> 3);
          const float* src = a.in + (size_t)Ureg[it] * a.ld_in + (qb0 + b) * 32 + (((lane & 7) ^ (l & 7)) << 2);
          if (l < n_u)
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(xs + (b * srows + l0) * 128), 16, 0, 0);
        }
      }
  };
  const unsigned int gh = (unsigned)g << 4, gl = (unsigned)(g ^ 4) << 4;
  // one occupied offset: operands of all 8 groups from LDS (a group without neighbours reads the zero row), products
  // only for the groups whose mask has the offset
  auto compute = [&](int k, const uint2 L, const float4 (&W)[NTW][2], int b) {
    float4 X[NG][2];
    const unsigned int w4[2] = {L.x, L.y};
    const unsigned char* xb = xs + b * srows * 128;                    // region of the block (a multiple of the swizzle period)
#pragma unroll
    for (int c = 0; c < NG; ++c) {
      const unsigned int ad = ((w4[c >> 1] >> (16 * (c & 1))) & 0xFFFFu) << 4;
      X[c][0] = *reinterpret_cast<const float4*>(xb + (ad ^ gh));
      X[c][1] = *reinterpret_cast<const float4*>(xb + (ad ^ gl));
    }
#pragma unroll
    for (int c = 0; c < NG; ++c) {
      if (!((gm[c] >> k) & 1u)) continue;                              // scalar branch
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        const half8_t xv = __builtin_bit_cast(half8_t, X[c][term == 1 ? 1 : 0]);
#pragma unroll
        for (int t = 0; t < NTW; ++t)
          acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, W[t][term == 2 ? 1 : 0]), xv, acc[c][t], 0, 0, 0);
      }
    }
  };

  float4 WA[NTW][2], WB[NTW][2];
  uint2 LA = make_uint2(0, 0), LB = make_uint2(0, 0);
  const int n_off = __builtin_popcount(wmask);                         // wave-uniform: the offsets any of the wave's groups has
  for (int qb0 = 0; qb0 < nqb; qb0 += nb_max) {
    const int nb = min(nb_max, nqb - qb0);
    if (qb0) __syncthreads();                                          // every wave is done with the previous round's rows
    stage(qb0, nb);
    if (qb0 == 0) UP_STAMP(1);
    unsigned int rest = wmask;
    int kc = rest ? __builtin_ctz(rest) : 0, bc = 0;                   // the (offset, block of the round) being multiplied
    load_w(kc, qb0, WA);                                               // unconditional: a load inside a branch is waited for on the spot
    LA = locp[kc * 64];
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    if (qb0 == 0) UP_STAMP(2);
    // the walk over (block, occupied offset): weights and rulebook entries of the next step are in flight while this one multiplies
    auto next = [&](int& k, int& bb) {
      rest &= rest - 1;
      if (!rest) { rest = wmask; bb = bb + 1 < nb ? bb + 1 : bb; }     // past the last step: re-loads the last block's first offset
      k = __builtin_ctz(rest);
    };
    const int steps = nb * n_off;
    for (int i = 0; i < steps; i += 2) {
      int kn = kc, bn = bc;
      next(kn, bn);
      load_w(kn, qb0 + bn, WB);
      LB = locp[kn * 64];
      compute(kc, LA, WA, bc);
      kc = kn; bc = bn;
      if (i + 1 >= steps) break;
      next(kn, bn);
      load_w(kn, qb0 + bn, WA);
      LA = locp[kn * 64];
      compute(kc, LB, WB, bc);
      kc = kn; bc = bn;
    }
  }

  UP_STAMP(3);
  asm volatile("" :: "v"(warm));
  // ---- epilogue: lane (g, j) holds channels 8 g .. 8 g + 7 (tuples t = 0, 1) of tile slot 64 sq + 16 c + j
  const float os = a.out_scale ? *a.out_scale : 1.0f;
  const int ch = ct0 + 8 * g;
  float4 b4[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
    b4[t] = a.bias ? *reinterpret_cast<const float4*>(a.bias + ch + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
  float mx = 0.f;
  // Two passes: (1) the four output rows (one round trip, not one per group), the residual rows, the values of all four groups;
  // (2) the stores, back to back.  In one pass the compiler re-used a group's store registers for the next group and waited for
  // the stores (vmcnt counts them) before every group: four dependent round trips at the end of every tile.
  int orow[NG];
#pragma unroll
  for (int c = 0; c < NG; ++c) orow[c] = rowp[16 * c + j];
  uint4 rh[NG], rl[NG];
  if (a.res) {
#pragma unroll
    for (int c = 0; c < NG; ++c)
      if (orow[c] >= 0) {
        const char* rp = reinterpret_cast<const char*>(a.res + (size_t)orow[c] * a.ld_res) + split16_off4(ch);
        rh[c] = *reinterpret_cast<const uint4*>(rp);
        rl[c] = *reinterpret_cast<const uint4*>(rp + SPLIT16_LO);
      }
  }
  float4 v[NG][NTW];
#pragma unroll
  for (int c = 0; c < NG; ++c) {
#pragma unroll
    for (int t = 0; t < NTW; ++t)
      v[c][t] = make_float4(acc[c][t][0] * os + b4[t].x, acc[c][t][1] * os + b4[t].y, acc[c][t][2] * os + b4[t].z, acc[c][t][3] * os + b4[t].w);
    if (a.res && orow[c] >= 0) {
      const float4 q0 = split16_decode4(make_uint2(rh[c].x, rh[c].y), make_uint2(rl[c].x, rl[c].y));
      const float4 q1 = split16_decode4(make_uint2(rh[c].z, rh[c].w), make_uint2(rl[c].z, rl[c].w));
      v[c][0].x += q0.x; v[c][0].y += q0.y; v[c][0].z += q0.z; v[c][0].w += q0.w;
      v[c][1].x += q1.x; v[c][1].y += q1.y; v[c][1].z += q1.z; v[c][1].w += q1.w;
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      if (a.relu) { v[c][t].x = fmaxf(v[c][t].x, 0.f); v[c][t].y = fmaxf(v[c][t].y, 0.f); v[c][t].z = fmaxf(v[c][t].z, 0.f); v[c][t].w = fmaxf(v[c][t].w, 0.f); }
      if (orow[c] >= 0) split16_track(mx, v[c][t]);
    }
  }
  UP_STAMP(4);
#pragma unroll
  for (int c = 0; c < NG; ++c) {
    const int o = orow[c];
    if (o < 0) continue;
    if (a.out_split) {
      uint2 h0, l0, h1, l1;
      split16_encode4(v[c][0], h0, l0);
      split16_encode4(v[c][1], h1, l1);
      char* op = reinterpret_cast<char*>(a.out + (size_t)o * a.ld_out) + split16_off4(ch);
      *reinterpret_cast<uint4*>(op) = make_uint4(h0.x, h0.y, h1.x, h1.y);
      *reinterpret_cast<uint4*>(op + SPLIT16_LO) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    } else {
      float* op = a.out + (size_t)o * a.ld_out + ch;
      *reinterpret_cast<float4*>(op) = v[c][0];
      *reinterpret_cast<float4*>(op + 4) = v[c][1];
    }
  }
  if (a.out_split) split16_report(a.range, mx);
#ifdef EYOC_UP_TRACE
  UP_STAMP(5);
  if (a.cin == 128 && a.cout == 64 && threadIdx.x == 0 && blockIdx.x < UP_TRACE_WGS) {
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_up_trace[blockIdx.x * UP_TRACE_N + 6] = ((unsigned long long)(xcc & 0xF) << 32) | hw;
  }
#endif
}

}  // namespace

#ifdef EYOC_UP_TRACE
extern "C" int eyoc_debug_up_trace(unsigned long long* host, size_t n) {
  if (n > (size_t)UP_TRACE_WGS * UP_TRACE_N) n = (size_t)UP_TRACE_WGS * UP_TRACE_N;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_up_trace), n * 8) != hipSuccess) return -1;
  unsigned long long* z = (unsigned long long*)calloc(n, 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_up_trace), z, n * 8);
  free(z);
  return 0;
}
#endif

namespace eyoc {

size_t local_rulebook_up_bytes(int n_out) { return (size_t)cdiv(n_out, TILE) * UP_LR_BYTES; }

// per-tile rulebooks of a transposed table; *overflow_dev (zeroed by the caller) counts tiles with more than 639 distinct
// input rows (the kernel must not be used for the table then)
int build_local_rulebook_up(const int32_t* nbr_dev, int K, int n_out, unsigned char* out_dev, int* overflow_dev, hipStream_t st) {
  if (n_out <= 0) return EYOC_OK;
  hipLaunchKernelGGL(k_local_rulebook_up, dim3(cdiv(n_out, TILE)), dim3(256), 0, st, nbr_dev, K, n_out, out_dev, overflow_dev);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

int launch_spconv_up(const SpconvArgs& a, const unsigned char* local_dev, hipStream_t st) {
  EYOC_REQUIRE(a.math == 1 && local_dev && !a.l2norm && a.K == 27 && a.cout % 64 == 0 && a.cin % 32 == 0, EYOC_ERR_INVALID,
               "spconv_up: unsupported layer (%d -> %d channels)", a.cin, a.cout);
  const bool wide = spconv_cc(a.cin, a.cout) == 64;
  const int n_cg = a.cout / 64;
  EYOC_REQUIRE(n_cg >= 1 && n_cg <= 8 && 8 % n_cg == 0, EYOC_ERR_INVALID, "spconv_up: %d output channels", a.cout);
  const int n_tiles = cdiv(a.n_out, TILE);
  const dim3 grid((unsigned)(cdiv(n_tiles, 8 / n_cg) * 8)), block(NWK * 64);
  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remembered in the caller's context (set on every launch without one)
  bool* attr_done = a.ctx ? a.ctx->up_attr_set : nullptr;
  if (wide) {
    if (!attr_done || !attr_done[0]) {
      EYOC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spconv_up_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, X_BYTES));
      if (attr_done) attr_done[0] = true;
    }
    hipLaunchKernelGGL(spconv_up_kernel<64>, grid, block, (size_t)X_BYTES, st, a, local_dev, n_tiles);
  } else {
    if (!attr_done || !attr_done[1]) {
      EYOC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spconv_up_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, X_BYTES));
      if (attr_done) attr_done[1] = true;
    }
    hipLaunchKernelGGL(spconv_up_kernel<32>, grid, block, (size_t)X_BYTES, st, a, local_dev, n_tiles);
  }
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

}  // namespace eyoc
