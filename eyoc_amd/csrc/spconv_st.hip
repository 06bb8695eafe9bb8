// Sparse convolution with a tile-local input stage (gfx950, SPLIT16 arithmetic, stride-1 rulebooks).
//
// A stride-1 layer reads every input row ~10 times (once per occupied offset of each output row it neighbours); the
// gathering kernels (spconv_wave.hip, spconv_rs.hip) fetch each of those reads from L2 / HBM.  This kernel fetches every
// input row a tile needs ONCE per 32-channel block:
//
//   * the rows of a level are stored in Morton order (eyoc_maps_build sorts them), so 256 consecutive output rows are a
//     compact blob of voxels whose 27-neighbourhoods overlap: ~2500 (row, offset) pairs but only ~380..500 distinct
//     input rows (at most ~720);
//   * a "local rulebook" per 256-row tile, built once per stride-1 table (k_local_rulebook, shared by every layer and
//     every channel block that uses the table): the list U of distinct input rows and, per (offset, output row), the
//     LDS slot of that row (slot 639 = a row of zeros = no neighbour);
//   * per (tile, 32-channel block) the workgroup's 4 waves copy the U rows' 128-byte lines global -> LDS (80 KB, 16-byte
//     pieces XOR-swizzled by the row so that random rows spread over the banks), then every wave runs all 27 offsets of
//     its 64 output rows from LDS: register accumulators, zero operands for missing neighbours, weight fragments
//     straight from L2 one stage ahead.
//
// Two workgroups (2 x 80 KB of LDS, 8 waves) per CU, two waves per SIMD: while one workgroup waits for its stage (or
// loads its rulebook, or stores its outputs) the other one owns the matrix pipe, and inside the offset loop the two
// waves of a SIMD fill each other's LDS / L2 latencies.  (The first version of this kernel staged per WAVE, 32 KB each,
// one wave per SIMD: every stage, prologue and epilogue was exposed - 2.3 ms on the 3.8M-row 64 -> 64 layer, MFMA
// pipe 43% busy; scripts/bench_staged.py + the EYOC_ST_ABL builds have the breakdown.)
#include <atomic>
#include <cstdlib>

#include "spconv.h"
#include "derive.h"

using namespace eyoc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NW = 4;                                      // waves per workgroup, 64 output rows each
constexpr int TILE = NW * 64;                              // output rows per workgroup
constexpr int XROWS = 640;                                 // LDS stage: rows of one 32-channel block (128 bytes each)
constexpr int UMAX = XROWS - 1;                            // distinct input rows staged per pass; slot UMAX holds zeros
constexpr int NPASS = 2;                                   // a tile with more distinct rows (<= 1278) takes a second pass
constexpr int UCAP = 1280;
constexpr int X_BYTES = XROWS * 128;                       // 80 KB: two workgroups per CU
constexpr int LO_REGION = XROWS * 64;                      // the stage: XROWS x 64 B of hi halves, then XROWS x 64 B of lo halves
constexpr int NIT = XROWS / (16 * NW);                     // staging steps per wave and block (16 rows = one hi + one lo instruction each)
// LDS byte address of the hi pieces of stage slot l: four 16-byte pieces (8 fp16 hi halves each) per row, piece p stored at
// position p ^ s(l), s(l) = (l >> 2) & 3, so that rows 4 apart do not collide; the lane holding operand piece g reads at
// slot_addr(l) ^ (g << 4), and the lo piece LO_REGION bytes further (one XOR per gathered row, the lo read through the
// instruction's offset field).  Fits 16 bits: this IS the rulebook's entry.
__host__ __device__ constexpr int slot_addr(int l) { return l * 64 + ((l >> 2) & 3) * 16; }

// ---- local rulebook of one 256-row tile (LR_BYTES bytes): int n_unique (-1: more than NPASS * UMAX), pad[3];
// int U[UCAP]; uint2 loc[NPASS][27][64]: entry (pass, k, 16 w + j) packs, as four 16-bit values slot_addr(l), the
// LDS slots l of the neighbours at offset k of output rows 64 w + 16 c + j, c = 0..3 (slot UMAX = no neighbour, or a
// neighbour staged in the other pass).
// Then, per pass, 27 (+1 pad) 16-bit occupancy masks: bit 4 w + c of mask (pass, k) = some row of rows 64 w + 16 c .. +15 has a
// neighbour at offset k that is staged in this pass (the hand-scheduled loop skips the MFMAs of a chunk whose bit is clear).
// Last, 256 bytes: the tile's ROW MAP.  The record describes 256 tile SLOTS; slot 64 w + 16 c + j holds local row
// rowmap[(16 w + j) * 4 + c] of the tile (a permutation of 0 .. 255).  With row grouping on (round 4, the default) the builder
// sorts the tile's rows by their neighbour pattern - 6 bits "any neighbour in the z+1 / z-1 / y+1 / y-1 / x+1 / x-1 layer", then the
// 27-bit occupancy mask, z layers first - so that the 16 rows of an MFMA chunk miss the same offsets: non-empty (chunk, offset)
// blocks 0.81 -> 0.68 at level 0, 0.93 -> 0.72 at levels 1 and 2, 0.66 -> 0.58 at level 3 on the bench geometry (the loop skips
// the rest).  Only the ORDER of a tile's rows inside the workgroup changes: the stage, the products and their summation order per
// output element are the same, results are bit-identical to the ungrouped records.
constexpr int MASK_OFF = 16 + UCAP * 4 + NPASS * 27 * 64 * 8;   // 32784
constexpr int MASK_PASS_BYTES = 56;
constexpr int RM_OFF = MASK_OFF + NPASS * MASK_PASS_BYTES;      // 32896
constexpr int INV_OFF = RM_OFF + TILE;                          // 33152: the inverse map, inv[local row] = slot (conv1_bf_kernel finds a parent's entries with it)
constexpr int LR_BYTES = INV_OFF + TILE;                        // 33408
static_assert(INV_OFF == ST_INV_OFF, "spconv.h mirrors this layout");
static_assert(TILE == ST_TILE && UMAX == ST_UMAX && UCAP == ST_UCAP && NPASS == ST_NPASS && LR_BYTES == ST_LR_BYTES, "spconv.h mirrors this layout");
static_assert(LR_BYTES % 128 == 0, "records start on cache lines");
// LDS hash of a tile's distinct input rows: 2048 slots.  A usable tile has at most NPASS * UMAX = 1278 of them (62 % load; 370-620
// typical); a tile with more than the table holds (rows in no spatial order) gives up after a full round of probing and counts as
// overflowed.  Clearing and numbering the table is a fixed cost per tile: 8192 slots (room for all 256 * 27 possible rows) took 278 us
// at level 0, 4096 ~190 (rounds 3-5), 2048 - 28 KB of LDS instead of 40, five workgroups per CU - is what round 6 runs.
#ifndef EYOC_LR_HSLOTS_LOG2
#define EYOC_LR_HSLOTS_LOG2 11
#endif
constexpr int HSLOTS_LOG2 = EYOC_LR_HSLOTS_LOG2, HSLOTS_T = 1 << HSLOTS_LOG2;

// NWB = waves of the builder = 64-row quarters of the tile it describes: 4 (256-row tiles: the stride-1 tables) or 2 (128-row tiles:
// the strided tables, round 6 - a 128-row coarse tile reads 370-620 distinct fine rows, a 256-row one 700-1250).  Same record layout
// either way (a 128-row tile leaves the entries of quarters 2 and 3 unwritten: nobody reads them).
template <bool DERIVE, int NWB>
__global__ __launch_bounds__(NWB * 64) void k_local_rulebook(const int32_t* __restrict__ nbr, int K, int n_out, unsigned char* __restrict__ out,
                                                             int* __restrict__ overflow, int group, DeriveSrc d) {
  constexpr int TR = NWB * 64;                                       // rows per tile
  constexpr int HS = HSLOTS_T, HSHIFT = 32 - HSLOTS_LOG2;
  __shared__ int hk[HS];
  __shared__ unsigned short hid[HS];
  __shared__ unsigned short srow[27][TR];                            // hash slot of (offset, local row), 0xFFFF = no neighbour
  __shared__ __attribute__((aligned(16))) unsigned long long key[TR];
  __shared__ int wave_cnt[NWB];
  __shared__ int too_many;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x;
  const int row = tile * TR + (int)threadIdx.x;
  for (int i = threadIdx.x; i < HS; i += TR) hk[i] = -1;
  if (threadIdx.x == 0) too_many = 0;
  __syncthreads();
  // 1. insert every valid entry (linear probing; duplicates meet their own key); the slot found stays in a register
  unsigned short slot[27];
  int idxs[27];                                                      // all 27 loads first: behind the CAS loops each would wait alone
  if constexpr (DERIVE) {                                            // no table in memory: the row's window through the octree (derive.h)
#pragma unroll
    for (int k = 0; k < 27; ++k) idxs[k] = -1;
    if (row < n_out) {
      const int4 c = reinterpret_cast<const int4*>(d.coords)[row];
      const int b[3] = {(c.y >> d.sh) & 1, (c.z >> d.sh) & 1, (c.w >> d.sh) & 1};
      int blk[8];
      derive_blocks(b, d.parent[row], d.s1c, d.nc, blk);
      derive_window(b, blk, d.children, idxs);
      if (d.up8) {
        int bl[8];
        derive_up_blocks(b, blk, bl);
        const int cls = b[0] | (b[1] << 1) | (b[2] << 2);
#pragma unroll
        for (int m = 0; m < 8; ++m) d.up8[(size_t)m * n_out + row] = (m & ~cls) == 0 ? bl[m] : -1;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 27; ++k) idxs[k] = (k < K && row < n_out) ? nbr[(size_t)k * n_out + row] : -1;
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int idx = idxs[k];
    unsigned int s = 0xFFFFu;
    if (idx >= 0) {
      s = ((unsigned)idx * 2654435761u) >> HSHIFT;
      int probes = 0;
      while (true) {
        const int prev = atomicCAS(&hk[s], -1, idx);
        if (prev == -1 || prev == idx) break;
        s = (s + 1) & (HS - 1);
        if (++probes >= HS) { too_many = 1; s = 0xFFFFu; break; }   // the table is full: more distinct rows than it has slots
      }
    }
    slot[k] = (unsigned short)s;
  }
  // the tile's row order: sort key = (6 layer bits, occupancy mask with the z layers first, local row); ungrouped: the row itself
  {
    unsigned int mask = 0;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      srow[k][threadIdx.x] = slot[k];
      if (slot[k] != 0xFFFFu) mask |= 1u << k;
    }
    // offsets enumerate x fastest: k = (dx + 1) + 3 (dy + 1) + 9 (dz + 1)
    const unsigned int zp = (mask >> 18) & 0x1FFu, zm = mask & 0x1FFu, z0 = (mask >> 9) & 0x1FFu;
    const unsigned int yp = mask & 0x70381C0u ? 1u : 0u;             // dy = +1: bits 6..8 of every z layer
    const unsigned int ym = mask & 0x01C0E07u ? 1u : 0u;             // dy = -1: bits 0..2
    const unsigned int xp = mask & 0x4924924u ? 1u : 0u;             // dx = +1: bits 2, 5, 8, ...
    const unsigned int xm = mask & 0x1249249u ? 1u : 0u;             // dx = -1: bits 0, 3, 6, ...
    unsigned int cb = (zp ? 32u : 0u) | (zm ? 16u : 0u) | (yp << 3) | (ym << 2) | (xp << 1) | xm;
    cb ^= cb >> 1; cb ^= cb >> 2; cb ^= cb >> 4;                     // rank in the reflected Gray sequence: neighbouring buckets differ in one layer bit
    const unsigned long long coarse = cb;
    const unsigned long long pattern = ((unsigned long long)zp << 18) | ((unsigned long long)zm << 9) | z0;
    key[threadIdx.x] = group ? (coarse << 35) | (pattern << 8) | threadIdx.x : (unsigned long long)threadIdx.x;
  }
  __syncthreads();
  if (too_many) {                                                      // workgroup-uniform
    if (threadIdx.x == 0) {
      reinterpret_cast<int*>(out + (size_t)tile * LR_BYTES)[0] = -1;
      atomicAdd(overflow, 1);
    }
    return;
  }
  // 2. number the occupied slots in slot order (deterministic): every wave owns a quarter of the table
  constexpr int PER_WAVE = HS / NWB;
  int cnt = 0;
  for (int i0 = 0; i0 < PER_WAVE; i0 += 64) cnt += __popcll(__ballot(hk[wave * PER_WAVE + i0 + lane] >= 0));
  if (lane == 0) wave_cnt[wave] = cnt;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < NWB; ++w) { if (w < wave) base += wave_cnt[w]; total += wave_cnt[w]; }
  unsigned char* lr = out + (size_t)tile * LR_BYTES;
  int* U = reinterpret_cast<int*>(lr + 16);
  if (total <= UMAX) {                                                 // one pass: any numbering gives the same sums - slot order
    for (int i0 = 0; i0 < PER_WAVE; i0 += 64) {
      const int s = wave * PER_WAVE + i0 + lane;
      const int key = hk[s];
      const unsigned long long m = __ballot(key >= 0);
      const int id = base + __popcll(m & ((1ull << lane) - 1ull));
      if (key >= 0) {
        hid[s] = (unsigned short)id;
        U[id] = key;
      }
      base += __popcll(m);
    }
  } else {                                                              // two passes: a numbering that does not depend on the order the CAS loops ran in (canonical_slot_id, spconv.h)
    // (hid holds the slot-order positions first and the canonical numbers after - a second 8 KB array would cost the kernel its
    // fourth workgroup per CU)
    for (int i0 = 0; i0 < PER_WAVE; i0 += 64) {
      const int s = wave * PER_WAVE + i0 + lane;
      const unsigned long long m = __ballot(hk[s] >= 0);
      if (hk[s] >= 0) hid[s] = (unsigned short)(base + __popcll(m & ((1ull << lane) - 1ull)));   // occupied slots in front of s
      base += __popcll(m);
    }
    __syncthreads();
    unsigned short ids[PER_WAVE / 64];
#pragma unroll
    for (int q = 0; q < PER_WAVE / 64; ++q) {
      const int s = wave * PER_WAVE + q * 64 + lane;
      ids[q] = hk[s] >= 0 ? (unsigned short)canonical_slot_id<HS>(hk, hid, s) : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER_WAVE / 64; ++q) {
      const int s = wave * PER_WAVE + q * 64 + lane;
      const int key = hk[s];
      if (key >= 0) {
        hid[s] = ids[q];
        if (ids[q] < NPASS * UMAX) U[ids[q]] = key;
      }
    }
  }
  if (threadIdx.x == 0) {
    reinterpret_cast<int*>(lr)[0] = total <= NPASS * UMAX ? total : -1;
    if (total > NPASS * UMAX) atomicAdd(overflow, 1);
  }
  __syncthreads();
  if (group) {
    // the tile's keys in ascending order (workgroup-uniform branch).  Rank by counting: the keys are distinct (the local row is in the low
    // bits), so a key's position is the number of smaller keys - TR broadcast reads per thread (two keys per 16-byte read) and ONE
    // barrier, against the 36 barrier-separated compare-exchange rounds of the bitonic network this replaces (round 6)
    const unsigned long long mine = key[threadIdx.x];
    int rank = 0;
    const ulonglong2* kp = reinterpret_cast<const ulonglong2*>(key);
#pragma unroll 8
    for (int i = 0; i < TR / 2; ++i) {
      const ulonglong2 kk = kp[i];
      rank += (kk.x < mine) + (kk.y < mine);
    }
    __syncthreads();
    key[rank] = mine;
    __syncthreads();
  }
  // 3. LDS slot of every (offset, tile slot): thread s owns slot s = 64 w + 16 c + j, which holds local row src; the slot's
  // entries go to 16-bit lane c of entry (k, 16 w + j)
  unsigned short* loc = reinterpret_cast<unsigned short*>(lr + 16 + UCAP * 4);
  unsigned char* msk = lr + MASK_OFF;
  __shared__ unsigned char nib[NPASS][27][4];
  const int r = (int)threadIdx.x, w = r >> 6, c = (r >> 4) & 3, j = r & 15;
  // sorted 16-row chunk i goes to slot chunk (w, c) = (i % 4, i / 4): dealt round-robin over the tile's four 64-slot quarters, so
  // that every wave gets its share of the sparse and of the dense patterns (in sorted order the first quarter's waves would run
  // out of non-empty blocks long before the last one's and wait at the tile's barriers)
  // (snake order - 0 1 2 3 3 2 1 0 ... - so that the two row halves of a workgroup, quarters {0, 1} and {2, 3}, carry the same load)
  const int src = (int)(key[group ? (c * NWB + ((c & 1) ? NWB - 1 - w : w)) * 16 + j : r] & 255u);
  lr[RM_OFF + (w * 16 + j) * 4 + c] = (unsigned char)src;
  lr[INV_OFF + src] = (unsigned char)r;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const unsigned int hs = srow[k][src];
    const int id = hs != 0xFFFFu ? (int)hid[hs] : -1;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      if (p > 0 && total <= p * UMAX) continue;                         // nobody reads the entries of a pass that does not happen
      const int l = (id >= p * UMAX && id < (p + 1) * UMAX) ? id - p * UMAX : UMAX;
      loc[(((size_t)(p * 27 + k) * 64) + w * 16 + j) * 4 + c] = (unsigned short)slot_addr(l);
      const unsigned long long b = __ballot(l != UMAX);                  // 16 lanes per chunk
      if (lane == 0)
        nib[p][k][w] = (unsigned char)(((b & 0xFFFFull) != 0) | (((b >> 16) & 0xFFFFull) != 0) << 1 | (((b >> 32) & 0xFFFFull) != 0) << 2 |
                                       (((b >> 48) & 0xFFFFull) != 0) << 3);
    }
  }
  __syncthreads();
  if (threadIdx.x < NPASS * 28) {
    const int p = (int)threadIdx.x / 28, k = (int)threadIdx.x % 28;
    unsigned short m = 0;
    if (k < 27 && !(p > 0 && total <= p * UMAX)) {
      m = (unsigned short)(nib[p][k][0] | nib[p][k][1] << 4);
      if (NWB == 4) m |= (unsigned short)(nib[p][k][2] << 8 | nib[p][k][3] << 12);
    }
    reinterpret_cast<unsigned short*>(msk)[p * 28 + k] = m;
  }
}

// The workgroup's 256 rows x CTG output channels are split over its 4 waves as 4 row quarters x CTG channels (NH = 1:
// 64 rows x NTW * 16 channels per wave) or as 2 row halves x 2 channel halves (NH = 2: 128 rows x 32 channels per wave -
// half the weight bytes through the L1, twice the operand reads from LDS).
template <int NTW, int CC, int NH>
__global__ __launch_bounds__(NW * 64, 2) void spconv_st_kernel(SpconvArgs a, const unsigned char* __restrict__ local, int n_tiles) {
  constexpr int CTW = NTW * 16, NC = 4;
  constexpr int CTG = CTW * NH;                                        // output channels per workgroup
  // offsets the weight fragments run ahead (measured: the 32-channel layers' short offsets want two - 0.43 -> 0.39 ms on
  // the 1.95 M-row 32 -> 32 layer - the others lose 8 % to the third register set)
  constexpr int WD = (NTW * NH <= 2) ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char xs[];   // the workgroup's stage: XROWS x 128 bytes
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane >> 4, j = lane & 15;
  // workgroup -> (tile, channel group): consecutive workgroups go to consecutive XCDs, so the channel group is taken
  // from the XCD number - each XCD's L2 then holds the weights of one channel group only (a 256 -> 256 layer has 7 MB of
  // split weights, an L2 has 4 MB)
  const int n_cg = a.cout / CTG;                                       // 1, 2, 4 or 8
  const int xcd = (int)blockIdx.x & 7, per = 8 / n_cg;
  const int cg = xcd % n_cg, tile = ((int)blockIdx.x >> 3) * per + xcd / n_cg;
  if (tile >= n_tiles) return;
  const int w0 = NH == 1 ? wave : 2 * (wave >> 1);                     // first 64-row quarter of this wave
  const int ct0 = cg * CTG + (NH == 2 ? (wave & 1) * CTW : 0);
  const int CT = a.cout >= 128 ? 128 : a.cout;
  const int n_slices = a.cout / CT, slice = ct0 / CT, nt0 = (ct0 - slice * CT) / 16;
  constexpr int JQ = CC / 16;
  const int ncc = a.cin / CC;
  const int nqb = a.cin / 32;
  constexpr int K = 27;

  const unsigned char* lr = local + (size_t)tile * LR_BYTES;
  const int n_u = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(lr)[0]);
  const int* __restrict__ U = reinterpret_cast<const int*>(lr + 16);
  const uint2* __restrict__ locp = reinterpret_cast<const uint2*>(lr + 16 + UCAP * 4) + w0 * 16 + j;
  const int n_pass = n_u > UMAX ? 2 : 1;                               // workgroup-uniform; the second pass is rare
  // zero row (slot UMAX), written once: no stage ever touches it
  if (threadIdx.x < 8)
    *reinterpret_cast<float4*>(xs + (threadIdx.x >> 2) * LO_REGION + UMAX * 64 + (threadIdx.x & 3) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, K * a.cin * a.cout * 4, 0x00020000);
  const int tile4 = CC * CT / 4;

  f32x4 acc[NH][NC][NTW];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[h][c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this wave's MFMA weight fragments of stage (k, qb), straight from L2 / L1 into operand registers, one stage ahead
  const int lane_off = lane * 16;
  auto load_w = [&](int k, int qb, float4 (&W)[NTW][2]) {
    const int cc = (qb * 32) / CC, qp = ((qb * 32) % CC) / 32;
    const int wbase = __builtin_amdgcn_readfirstlane((((k * n_slices + slice) * ncc + cc) * tile4 + (nt0 * JQ + 2 * qp) * 64) * 16);
#if defined(EYOC_ST_ABL) && (EYOC_ST_ABL & 4)
    for (int t = 0; t < NTW; ++t) { asm volatile("" : "+v"(W[t][0].x), "+v"(W[t][1].x)); }
    return;
#endif
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_off + (t * JQ + p) * 1024, wbase, 0);
        W[t][p] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
  };
  // stage block qb of the tile's distinct input rows: 4 lanes per row and half (64 bytes of hi halves / of lo halves), 16 rows per
  // instruction, wave w takes the 16-row groups 4 it + w; global -> LDS directly (global_load_lds_dwordx4: lane i lands at
  // base + 16 i, so the piece swizzle is applied on the SOURCE side), all loads of a stage in flight at once
  // (the row numbers are re-read from the rulebook for every block: 20 registers held across the offset loop cost more
  // than one L2 round trip per block, which the other workgroup of the CU covers)
  int n_up = 0;                                                      // distinct rows of the current pass
  auto stage = [&](int pass, int qb) {
#if defined(EYOC_ST_ABL) && (EYOC_ST_ABL & 2)
    return;
#endif
    int Ureg[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int l = (it * NW + wave) * 16 + (lane >> 2);
      asm volatile("" : "+v"(l));                                      // opaque: or the loads are hoisted out of the block loop (and spill)
      Ureg[it] = l < n_up ? U[pass * UMAX + l] : 0;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int l0 = (it * NW + wave) * 16;
      if (l0 < n_up) {                                               // wave-uniform
        const int l = l0 + (lane >> 2);
        const float* src = a.in + (size_t)Ureg[it] * a.ld_in + qb * 32 + (((lane & 3) ^ ((l >> 2) & 3)) << 2);
        if (l < n_up) {
          __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(xs + l0 * 64), 16, 0, 0);
          __builtin_amdgcn_global_load_lds(src + 16, (__attribute__((address_space(3))) void*)(xs + LO_REGION + l0 * 64), 16, 0, 0);
        }
      }
    }
  };
  // operands of one offset: lane (g, j) reads pieces g (hi halves of channels 8 g ..) and 4 + g (lo halves) of the four
  // rows its entry names
  const unsigned int gh = (unsigned)g << 4;
  auto read_x = [&](const uint2 L, float4 (&X)[NC][2]) {
#if defined(EYOC_ST_ABL) && (EYOC_ST_ABL & 8)
    for (int c = 0; c < NC; ++c) { X[c][0].x = __uint_as_float(L.x); asm volatile("" : "+v"(X[c][0].y), "+v"(X[c][1].x)); }
    return;
#endif
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const unsigned int ad = (((c < 2 ? L.x : L.y) >> (16 * (c & 1))) & 0xFFFFu) ^ gh;
      X[c][0] = *reinterpret_cast<const float4*>(xs + ad);
      X[c][1] = *reinterpret_cast<const float4*>(xs + ad + LO_REGION);
    }
  };
  auto multiply = [&](int h, const float4 (&X)[NC][2], const float4 (&W)[NTW][2]) {
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const half8_t xv = __builtin_bit_cast(half8_t, X[c][term == 1 ? 1 : 0]);
#pragma unroll
        for (int t = 0; t < NTW; ++t)
          acc[h][c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, W[t][term == 2 ? 1 : 0]), xv, acc[h][c][t], 0, 0, 0);
      }
  };

  // ---- per (pass, block): stage, then 27 offsets of NH half-steps (64 rows each).  While a half-step multiplies, the
  // operands of the next one are read from LDS into the other X set; weight fragments and rulebook entries run two
  // offsets ahead (three register sets each).
  float4 XA[NC][2], XB[NC][2], W[3][NTW][2];
  uint2 L[3][NH];
  bool first = true;
  for (int pass = 0; pass < n_pass; ++pass) {
    n_up = min(n_u - pass * UMAX, UMAX);
    const uint2* lp = locp + pass * 27 * 64;
    for (int qb = 0; qb < nqb; ++qb) {
      if (!first) __syncthreads();                                     // every wave is done with the previous block's rows
      first = false;
      stage(pass, qb);
      load_w(0, qb, W[0]);
      if (WD == 2) load_w(1, qb, W[1]);
#pragma unroll
      for (int h = 0; h < NH; ++h) { L[0][h] = lp[h * 16]; L[1][h] = lp[64 + h * 16]; }
      __builtin_amdgcn_s_waitcnt(0x0070);                              // vmcnt(0): this wave's part of the stage has landed
      __syncthreads();
      read_x(L[0][0], XA);
#pragma unroll
      for (int hs = 0; hs < 27 * NH; ++hs) {
        const int k = hs / NH, h = hs % NH;                            // compile-time after unrolling
        if (h == 0 && k + 2 < 27) {
#pragma unroll
          for (int hh = 0; hh < NH; ++hh) L[(k + 2) % 3][hh] = lp[(k + 2) * 64 + hh * 16];
        }
        if (h == 0 && k + WD < 27) load_w(k + WD, qb, W[(k + WD) % 3]);   // weights run WD offsets ahead
        const int hn = hs + 1, kn = hn / NH, hhn = hn % NH;
        if (hs & 1) {
          if (hn < 27 * NH) read_x(L[kn % 3][hhn], XA);
          multiply(h, XB, W[k % 3]);
        } else {
          if (hn < 27 * NH) read_x(L[kn % 3][hhn], XB);
          multiply(h, XA, W[k % 3]);
        }
#if defined(EYOC_ST_ABL) && (EYOC_ST_ABL & 256)
        // explicit interleave: one LDS operand read per three MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);             // the prefetch loads first
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
#endif
#if !(defined(EYOC_ST_ABL) && (EYOC_ST_ABL & 128))
        __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting later offsets' loads up here (register budget: 256)
#endif
      }
    }
  }

  // ---- epilogue straight from the registers: lane (g, j) holds channels 16 t + 4 g .. +3 of row 16 c + j
  const float os = a.out_scale ? *a.out_scale : 1.0f;
  float4 b4[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
    b4[t] = a.bias ? *reinterpret_cast<const float4*>(a.bias + ct0 + 16 * t + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  float mx = 0.f;
#pragma unroll
  for (int hc = 0; hc < NH * NC; ++hc) {
    const int h = hc / NC, c = hc % NC;
    const int o = tile * TILE + (int)lr[RM_OFF + ((w0 + h) * 16 + j) * 4 + c];       // the slot's row (record's row map)
    if (o >= a.n_out) continue;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int ch = ct0 + 16 * t + 4 * g;
      float4 v = make_float4(acc[h][c][t][0] * os + b4[t].x, acc[h][c][t][1] * os + b4[t].y, acc[h][c][t][2] * os + b4[t].z,
                             acc[h][c][t][3] * os + b4[t].w);
      if (a.res) {
        const float4 q = split16_load4(a.res + (size_t)o * a.ld_res, ch);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      split16_track(mx, v);
      if (a.out_split) split16_store4(a.out + (size_t)o * a.ld_out, ch, v);
      else *reinterpret_cast<float4*>(a.out + (size_t)o * a.ld_out + ch) = v;
    }
  }
  if (a.out_split) split16_report(a.range, mx);
}


// ---------------------------------------------------------------------------------------------------------------------
// The same kernel with the offset loop as hand-scheduled assembly (spconv_st_loop.inc, generated by gen_st_loop.py: register
// map, double-buffered operand reads interleaved with the MFMAs, exact wait counts, scalar branches around the MFMAs of
// empty (16-row chunk, offset) blocks).  NTW = 2 output-channel tiles per wave; NH = 2: 128 rows x 32 channels per wave
// (2 row halves x 2 channel halves per workgroup, >= 64 output channels), NH = 1: 64 rows x 32 channels (32-channel layers).
// The stage lives in STATIC shared memory (80 KB; gfx950 allows 160 KB per workgroup), so the LDS base is 0 by
// construction and no per-device function attribute is needed.
#ifdef EYOC_ST_TRACE
// diagnostics (scripts/trace_staged.py): per workgroup {start, header read, [blob in, barrier open, blob out] x 2, end, HW_ID}
constexpr int TRACE_WGS = 16384, TRACE_N = 16;           // [10..13]: when blocks 0 / 1 reached offsets 9 and 18
__device__ unsigned long long g_st_trace[TRACE_WGS * TRACE_N];
#define ST_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < TRACE_WGS) g_st_trace[blockIdx.x * TRACE_N + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ST_STAMP(i) do {} while (0)
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));

#include "spconv_st_loop.inc"
#if defined(EYOC_ST_ABLATIONS) || defined(EYOC_ST_TRACE)
#include "spconv_st_loop_abl.inc"
#endif

// ksplit > 1 (small inputs, NH = 1 only; round 5): blockIdx.y = which share of the tile's (pass, 32-channel block) list this
// workgroup multiplies.  A single pair's level-3 layer is 72 workgroups, each walking 8 blocks x 27 offsets one after the other -
// 104 us for 8 GFLOP; split eight ways it is 576 workgroups of one block each.  Two launches: phase 0 stores every share's raw
// accumulators, phase 1 (one workgroup per (tile, channel group), no stage, no loop) adds the shares in share order and runs the
// epilogue.  (One launch with an arrival counter and the last workgroup summing was built first: its agent-scope fences - an L2
// write-back and invalidate per workgroup on gfx950 - made a single pair 1 ms SLOWER.)
// TR = 128 (round 6, the strided tables; NH = 1, 4 waves): the tile is 128 output rows x 64 channels - waves = 2 row quarters x 2 channel
// halves of 64 rows x 32 channels, records from k_local_rulebook<.., 2> (same layout; quarters 2 and 3 do not exist).
// TAILF = 1 (round 6; NH = 2, 256-row tiles, a 64-channel layer): the network's 1x1 tail (spconv_tail.hip: conv1_tr -> ReLU -> final -> row
// normalisation) runs in this kernel's epilogue on the tile's rows - see the epilogue.
template <int CC, int NH, int SKIP, int NWV = NW, int TR = TILE, int TAILF = 0>
__global__ __launch_bounds__(NWV * 64, NWV / 2) void spconv_st_asm_kernel(SpconvArgs a, const unsigned char* __restrict__ local, int n_tiles, int ksplit, int phase) {
  constexpr int NTW = 2, CTW = NTW * 16, NC = 4;
  constexpr bool T128 = TR == 128;
  static_assert(TAILF == 0 || (NH == 2 && TR == TILE && NWV == NW && SKIP == 1), "the fused tail rides on the 256-row NH = 2 kernel");
  constexpr int CTG = T128 ? (NH == 2 ? 4 * CTW : 2 * CTW) : CTW * NH * (NWV / NW);   // NWV = 8: 4 row quarters x 2 channel halves, NH = 1
  constexpr int NITV = XROWS / (16 * NWV);
  static_assert(NWV == NW || (NWV == 2 * NW && NH == 1), "8 waves: 64 rows x 32 channels each");
  // 128-row tiles: four waves of 64 rows x 32 channels (2 row quarters x 2 channel halves: 64 channels per workgroup) or, NH = 2, of
  // 128 rows x 32 channels (4 channel quarters: 128 channels per workgroup - half the stages and tiles of a >= 128-channel layer)
  static_assert(TR == TILE || (T128 && NWV == NW), "128-row tiles: four waves");
  // (EYOC_ST_ABLATIONS 15-17: a padded stage = ONE workgroup per CU, to time a workgroup that has its CU to itself)
  __shared__ __attribute__((aligned(128))) unsigned char xs[X_BYTES + (SKIP >= 15 && SKIP <= 17 ? 8192 : 0)];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane >> 4, j = lane & 15;
  const int n_cg = a.cout / CTG;
  const int xcd = (int)blockIdx.x & 7, per = 8 / n_cg;
  // workgroup -> (tile, channel group).  Consecutive workgroups go to consecutive XCDs (each with its own L2).  Default: the channel
  // group from the XCD number, so that an L2 holds one group's weights only (a 256 -> 256 layer has 7 MB of split weights, an L2 4 MB) -
  // the groups of a tile then stage the same rows from HBM once per XCD.  cg_local (round 6; layers whose whole packed kernel fits an L2
  // beside the stream: 128 -> 128 = 1.8 MB): the groups of a tile are workgroups b and b + 8 - the same XCD, back to back - and the
  // second one's stage hits the L2 the first one filled
  const int cg = a.cg_local ? ((int)blockIdx.x >> 3) % n_cg : xcd % n_cg;
  const int tile = a.cg_local ? (((int)blockIdx.x >> 3) / n_cg) * 8 + xcd : ((int)blockIdx.x >> 3) * per + xcd / n_cg;
  if (tile >= n_tiles) return;
  const int w0 = T128 ? (NH == 2 ? 0 : (wave >> 1)) : NH == 1 ? (wave & 3) : 2 * (wave >> 1);
  const int ct0 = cg * CTG + (T128 && NH == 2 ? wave * CTW : T128 || NH == 2 ? (wave & 1) * CTW : (wave >> 2) * CTW);
  const int CT = a.cout >= 128 ? 128 : a.cout;
  const int n_slices = a.cout / CT, slice = ct0 / CT, nt0 = (ct0 - slice * CT) / 16;
  constexpr int JQ = CC / 16;
  const int ncc = a.cin / CC;
  const int nqb = a.cin / 32;
  constexpr int K = 27;

  ST_STAMP(0);
  const unsigned char* lr = local + (size_t)tile * LR_BYTES;
  const int* __restrict__ U = reinterpret_cast<const int*>(lr + 16);
  if (threadIdx.x < 8)
    *reinterpret_cast<float4*>(xs + (threadIdx.x >> 2) * LO_REGION + UMAX * 64 + (threadIdx.x & 3) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

  // raw buffer resource of the packed weights (stride 0, byte-granular bounds, the flags make_buffer_rsrc takes elsewhere)
  const unsigned long long wbits = (unsigned long long)(size_t)a.w;
  u32x4 wr;
  wr[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wbits);
  wr[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(wbits >> 32) & 0xFFFFu));
  wr[2] = (unsigned)__builtin_amdgcn_readfirstlane(K * a.cin * a.cout * 4);
  wr[3] = 0x00020000u;
  const int tile4 = CC * CT / 4;
  const unsigned int kstride = (unsigned)(n_slices * ncc * tile4 * 16);
  const unsigned int w1off = JQ * 1024;                                // byte distance of the second channel tile's fragments
  // the loop addresses the stage from LDS byte 0 (the kernel's only shared allocation)
  if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)xs != 0u) __builtin_trap();

  f32x16 A0, A1, A2, A3;
#pragma unroll
  for (int i = 0; i < 16; ++i) { A0[i] = 0.f; A1[i] = 0.f; A2[i] = 0.f; A3[i] = 0.f; }

  // the row numbers of a pass are loaded ONCE, before anything depends on the tile's header (the list has UCAP entries whatever
  // n_unique says; entries past it are never staged), and stay in 10 registers over the pass's 32-channel blocks: between two
  // blocks the stage is then only a barrier and the DMA issue, not a dependent load -> DMA chain (the trace of
  // scripts/trace_staged.py had 15 % of a workgroup's life between its blocks and 10 % before the first one)
  int n_up = 0;
  int Ureg[NITV];
  auto load_rows = [&](int pass) {
#pragma unroll
    for (int it = 0; it < NITV; ++it) Ureg[it] = U[pass * UMAX + (it * NWV + wave) * 16 + (lane >> 2)];
  };
  auto stage = [&](int pass, int qb) {
    if constexpr (SKIP == 6) return;                                   // EYOC_ST_ABLATIONS: no stage
#pragma unroll
    for (int it = 0; it < NITV; ++it) {
      const int l0 = (it * NWV + wave) * 16;
      if (l0 < n_up) {
        const int l = l0 + (lane >> 2);
        const float* src = a.in + (size_t)Ureg[it] * a.ld_in + qb * 32 + (((lane & 3) ^ ((l >> 2) & 3)) << 2);
        if (l < n_up) {
          __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(xs + l0 * 64), 16, 0, 0);
          __builtin_amdgcn_global_load_lds(src + 16, (__attribute__((address_space(3))) void*)(xs + LO_REGION + l0 * 64), 16, 0, 0);
        }
      }
    }
  };

  if (!phase) load_rows(0);
  // pull the tile's rulebook entries (13.8 KB, last touched by the builder: HBM by now) into L2 while the stage is in flight: the
  // loop requests them only two offsets (~2000 cycles) ahead, less than an HBM round trip under load - one dword per 128-byte line
  int warm = 0;
  if (!phase && threadIdx.x < 27 * 64 * 8 / 128) warm = *reinterpret_cast<const int*>(lr + 16 + UCAP * 4 + threadIdx.x * 128);
  const int n_u = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(lr)[0]);
  const int n_pass = n_u > UMAX ? 2 : 1;
  // this workgroup's share [i0, i1) of the (pass, block) list (everything without a split)
  const int ks = __builtin_amdgcn_readfirstlane((int)blockIdx.y);
  const int i0 = ks * (n_pass * nqb) / ksplit, i1 = phase ? 0 : (ks + 1) * (n_pass * nqb) / ksplit;     // phase 1: no block at all
  bool first = true;
  for (int pass = 0; pass < n_pass; ++pass) {
    const int q_lo = max(i0 - pass * nqb, 0), q_hi = min(i1 - pass * nqb, nqb);
    if (q_lo >= q_hi) continue;                                          // workgroup-uniform
    n_up = min(n_u - pass * UMAX, UMAX);
    if (pass > 0) load_rows(pass);
    // this wave's occupancy masks of the pass: 14 dwords through the scalar cache, shifted so that bit (k & 1) * 16 + 4 h + c
    // of dword k / 2 is chunk c of the wave's row half h
    const unsigned int* mp = reinterpret_cast<const unsigned int*>(lr + MASK_OFF + pass * MASK_PASS_BYTES);
    u32x8 M0, M1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      M0[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)mp[i]) >> (w0 * 4);
      M1[i] = i < 6 ? (unsigned)__builtin_amdgcn_readfirstlane((int)mp[8 + i]) >> (w0 * 4) : 0u;
    }
    // rulebook entries of (pass, k, h): 8 bytes at lb + lv + (k * 64 + h * 16) * 8
    const unsigned char* lb = lr + 16 + UCAP * 4 + ((size_t)pass * 27 * 64 + w0 * 16) * 8;
#ifdef EYOC_ST_QBREV                                                     // diagnostics (timing only): the 32-channel blocks in descending order
    for (int qbr = q_lo; qbr < q_hi; ++qbr) {
      const int qb = q_hi - 1 - (qbr - q_lo);
#else
    for (int qb = q_lo; qb < q_hi; ++qb) {
#endif
      if (!first) __syncthreads();                                     // every wave is done with the previous block's rows
#ifdef EYOC_ST_DELAY_PRE                                                 // diagnostics: an idle gap in front of the tile's first STAGE
      if (first) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); for (int z_ = 0; z_ < EYOC_ST_DELAY_PRE; ++z_) __builtin_amdgcn_s_sleep(127); }
#endif
      first = false;
      stage(pass, qb);
      const int cc = (qb * 32) / CC, qp = ((qb * 32) % CC) / 32;
      const unsigned int ws0 = (unsigned)__builtin_amdgcn_readfirstlane(((slice * ncc + cc) * tile4 + (nt0 * JQ + 2 * qp) * 64) * 16);
      unsigned int so;
#ifdef EYOC_ST_DELAY                                                     // diagnostics: an idle gap in front of the tile's first offset loop
      if (pass == 0 && qb == q_lo) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); for (int z_ = 0; z_ < EYOC_ST_DELAY; ++z_) __builtin_amdgcn_s_sleep(127); }
#endif
#ifdef EYOC_ST_TRACE
      unsigned long long tb = 0, tb1 = 0, tb2 = 0;
      const int blk_i = pass * nqb + qb;
      if (blk_i < 2) ST_STAMP(2 + 3 * blk_i);
      if (blk_i == 0) ST_STAMP(1);
#endif
#define EYOC_ST_OPERANDS                                                                                                            \
  [wr] "s"(wr), [ws0] "s"(ws0), [ks] "s"(kstride), [lb] "s"(lb), [w1] "s"(w1off), "{s[36:43]}"(M0), "{s[44:51]}"(M1)
#define EYOC_ST_ASM_NH2(TEXT)                                                                                                       \
  asm volatile(TEXT : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), "+{v[96:111]}"(A2), "+{v[112:127]}"(A3), [so] "=&s"(so)                   \
               : EYOC_ST_OPERANDS : "memory", "scc", EYOC_ST_LOOP_CLOBBERS)
#ifdef EYOC_ST_TRACE
      if constexpr (NH == 2 && SKIP == 1) {
        asm volatile(EYOC_ST_LOOP_NH2_TRACE : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), "+{v[96:111]}"(A2), "+{v[112:127]}"(A3), [so] "=&s"(so), [tb] "=&s"(tb), [tb1] "=&s"(tb1), [tb2] "=&s"(tb2)
                     : EYOC_ST_OPERANDS : "memory", "scc", EYOC_ST_LOOP_CLOBBERS);
        if (blk_i < 2) {
          if (threadIdx.x == 0 && blockIdx.x < TRACE_WGS) {
            g_st_trace[blockIdx.x * TRACE_N + 3 + 3 * blk_i] = tb;
            g_st_trace[blockIdx.x * TRACE_N + 10 + 2 * blk_i] = tb1;
            g_st_trace[blockIdx.x * TRACE_N + 11 + 2 * blk_i] = tb2;
          }
          ST_STAMP(4 + 3 * blk_i);
        }
      } else
#endif
      if constexpr (NH == 2 && (SKIP == 1 || SKIP == 6 || SKIP == 7 || SKIP == 15)) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2);
      else if constexpr (NH == 2 && SKIP == 0) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_NOSKIP);
#ifdef EYOC_ST_ABLATIONS       // timing-only
      else if constexpr (NH == 2 && SKIP == 2) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_LAZY);   // (not timing-only: operand reads for non-empty blocks only - bit-identical, level) builds of the loop (results are garbage): no weight loads / no operand reads / no address VALU
      else if constexpr (NH == 2 && SKIP == 3) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_NOW);
      else if constexpr (NH == 2 && SKIP == 4) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_NOX);
      else if constexpr (NH == 2 && SKIP == 5) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_NOV);
      else if constexpr (NH == 2 && SKIP == 8) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_NOM);
      else if constexpr (NH == 2 && SKIP == 9) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_NOMW);
      else if constexpr (NH == 2 && SKIP == 10) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_NOL);
      else if constexpr (NH == 2 && SKIP == 11) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_NOWL);
      else if constexpr (NH == 2 && SKIP == 12) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_NOMWL);
      else if constexpr (NH == 2 && SKIP == 13) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_EMPTY);
      else if constexpr (NH == 2 && SKIP == 14)
        asm volatile(EYOC_ST_LOOP_NH2_W2L3 : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), "+{v[96:111]}"(A2), "+{v[112:127]}"(A3), [so] "=&s"(so)
                     : EYOC_ST_OPERANDS : "memory", "scc", EYOC_ST_LOOP_CLOBBERS_LOW);
      else if constexpr (NH == 2 && SKIP == 16) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_NOWL);
      else if constexpr (NH == 2 && SKIP == 17) EYOC_ST_ASM_NH2(EYOC_ST_LOOP_NH2_EMPTY);
#endif
#ifdef EYOC_ST_ABLATIONS
      else if constexpr (NH == 1 && SKIP == 13)
        asm volatile(EYOC_ST_LOOP_NH1_EMPTY : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), [so] "=&s"(so) : EYOC_ST_OPERANDS : "memory", "scc", EYOC_ST_LOOP_CLOBBERS);
      else if constexpr (NH == 1 && SKIP == 9)
        asm volatile(EYOC_ST_LOOP_NH1_NOMW : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), [so] "=&s"(so) : EYOC_ST_OPERANDS : "memory", "scc", EYOC_ST_LOOP_CLOBBERS);
      else if constexpr (NH == 1 && SKIP == 10)
        asm volatile(EYOC_ST_LOOP_NH1_NOMX : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), [so] "=&s"(so) : EYOC_ST_OPERANDS : "memory", "scc", EYOC_ST_LOOP_CLOBBERS);
      else if constexpr (NH == 1 && SKIP == 11)
        asm volatile(EYOC_ST_LOOP_NH1_NOML : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), [so] "=&s"(so) : EYOC_ST_OPERANDS : "memory", "scc", EYOC_ST_LOOP_CLOBBERS);
      else if constexpr (NH == 1 && SKIP == 8)
        asm volatile(EYOC_ST_LOOP_NH1_NOM : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), [so] "=&s"(so) : EYOC_ST_OPERANDS : "memory", "scc", EYOC_ST_LOOP_CLOBBERS);
#endif
      else if constexpr (SKIP == 19)       // NH = 1, operand reads for non-empty blocks only
        asm volatile(EYOC_ST_LOOP_NH1_LAZY : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), [so] "=&s"(so)
                     : EYOC_ST_OPERANDS : "memory", "scc", EYOC_ST_LOOP_CLOBBERS);
      else if constexpr (SKIP == 20)       // ... with the deeper prefetch as well
        asm volatile(EYOC_ST_LOOP_NH1_DEEP_LAZY : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), [so] "=&s"(so)
                     : EYOC_ST_OPERANDS : "memory", "scc", EYOC_ST_LOOP_CLOBBERS_NH1);
      else if constexpr (SKIP == 18)       // NH = 1 with weights three offsets ahead and rulebook entries five (round 6)
        asm volatile(EYOC_ST_LOOP_NH1_DEEP : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), [so] "=&s"(so)
                     : EYOC_ST_OPERANDS : "memory", "scc", EYOC_ST_LOOP_CLOBBERS_NH1);
      else
        asm volatile(EYOC_ST_LOOP_NH1 : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), [so] "=&s"(so)
                     : EYOC_ST_OPERANDS : "memory", "scc", EYOC_ST_LOOP_CLOBBERS);
#undef EYOC_ST_ASM_NH2
#undef EYOC_ST_OPERANDS
    }
  }

  asm volatile("" :: "v"(warm));
  if constexpr (NH == 1 && NWV == NW && !T128) {
    if (ksplit > 1) {                                                    // launch-uniform
      constexpr int NR4 = 8;                                             // 32 accumulator registers per thread = 8 float4
      float4* P = reinterpret_cast<float4*>(a.ks_part) + (size_t)((tile * n_cg + cg) * ksplit) * NR4 * 256;
      if (phase == 0) {
        float4* mine = P + (size_t)ks * NR4 * 256 + threadIdx.x;
#pragma unroll
        for (int r = 0; r < NR4; ++r) {
          const f32x16& A = r < 4 ? A0 : A1;
          mine[r * 256] = make_float4(A[(r & 3) * 4], A[(r & 3) * 4 + 1], A[(r & 3) * 4 + 2], A[(r & 3) * 4 + 3]);
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) { A0[i] = 0.f; A1[i] = 0.f; }
      for (int k = 0; k < ksplit; ++k) {                                 // share order
        const float4* src = P + (size_t)k * NR4 * 256 + threadIdx.x;
#pragma unroll
        for (int r = 0; r < NR4; ++r) {
          const float4 v = src[r * 256];
          f32x16& A = r < 4 ? A0 : A1;
          A[(r & 3) * 4] += v.x; A[(r & 3) * 4 + 1] += v.y; A[(r & 3) * 4 + 2] += v.z; A[(r & 3) * 4 + 3] += v.w;
        }
      }
    }
  }
  // ---- epilogue straight from the registers.  The loop loads the weight rows permuted (gen_st_loop.py) so that lane (g, j)
  // holds, for row 64 h + 16 c + j, channels 8 g .. 8 g + 3 in tuple t = 0 and 8 g + 4 .. 8 g + 7 in tuple t = 1: 8 consecutive
  // channels = one 16-byte access per half of a SPLIT16 row (half as many memory instructions as 4-channel tuples)
  const float os = a.out_scale ? *a.out_scale : 1.0f;
  const int ch = ct0 + 8 * g;
  float4 b4[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
    b4[t] = a.bias ? *reinterpret_cast<const float4*>(a.bias + ch + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
  float mx = 0.f;
  // Three passes over the wave's NH * NC row groups - all residual loads, all values, all stores - instead of one: written as one
  // loop the compiler re-used a group's store registers for the next group and put `s_waitcnt vmcnt(0)` (which counts stores as
  // well) in front of every group, i.e. a store round trip AND a residual round trip per group, eight times in a row at the end of
  // every tile.  After the loop the 128 registers of the assembly blob are free: everything is in flight at once.
  constexpr int NG = NH * NC;
  // output row of the lane's slot 64 (w0 + h) + 16 c + j: through the record's row map (a permutation of the tile's rows)
  int orow[NG];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const unsigned int rm = *reinterpret_cast<const unsigned int*>(lr + RM_OFF + ((w0 + h) * 16 + j) * 4);
#pragma unroll
    for (int c = 0; c < NC; ++c) orow[h * NC + c] = tile * TR + (int)((rm >> (8 * c)) & 255u);
  }
  uint4 rh[NG], rl[NG];
  if (a.res) {
#pragma unroll
    for (int hc = 0; hc < NG; ++hc) {
      const int o = orow[hc];
      if (o < a.n_out) {
        const char* rp = reinterpret_cast<const char*>(a.res + (size_t)o * a.ld_res) + split16_off4(ch);
        rh[hc] = *reinterpret_cast<const uint4*>(rp);
        rl[hc] = *reinterpret_cast<const uint4*>(rp + SPLIT16_LO);
      }
    }
  }
  float4 v[NG][NTW];
#pragma unroll
  for (int hc = 0; hc < NG; ++hc) {
    const int o = orow[hc];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int ai = (hc * NTW + t) * 4;                                // register ACC(h, c, t) - 64 of the generator's map
      const f32x16& A = ai < 16 ? A0 : ai < 32 ? A1 : ai < 48 ? A2 : A3;
      v[hc][t] = make_float4(A[ai % 16 + 0] * os + b4[t].x, A[ai % 16 + 1] * os + b4[t].y, A[ai % 16 + 2] * os + b4[t].z,
                             A[ai % 16 + 3] * os + b4[t].w);
    }
    if (a.res && o < a.n_out) {
      const float4 q0 = split16_decode4(make_uint2(rh[hc].x, rh[hc].y), make_uint2(rl[hc].x, rl[hc].y));
      const float4 q1 = split16_decode4(make_uint2(rh[hc].z, rh[hc].w), make_uint2(rl[hc].z, rl[hc].w));
      v[hc][0].x += q0.x; v[hc][0].y += q0.y; v[hc][0].z += q0.z; v[hc][0].w += q0.w;
      v[hc][1].x += q1.x; v[hc][1].y += q1.y; v[hc][1].z += q1.z; v[hc][1].w += q1.w;
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      if (a.relu) { v[hc][t].x = fmaxf(v[hc][t].x, 0.f); v[hc][t].y = fmaxf(v[hc][t].y, 0.f); v[hc][t].z = fmaxf(v[hc][t].z, 0.f); v[hc][t].w = fmaxf(v[hc][t].w, 0.f); }
      if (o < a.n_out) split16_track(mx, v[hc][t]);
    }
  }
  if constexpr (TAILF != 0) {
    // ---- the network's 1x1 tail on this tile's rows (model/resunet.py:183-191; the arithmetic of spconv_tail.hip, bit for bit):
    //   out = normalise(final(relu(conv1_tr([this layer's 64 channels | 32 skip channels]))))
    // Lane (g, j) holds, for each of its wave's 8 row groups, channels ct0 + 8 g .. + 7 of row j - encoded, that IS the MFMA operand
    // piece of input block cc = (wave & 1) of conv1_tr (16 bytes of hi halves, 16 of lo halves of 8 channels).  The other 32 decoder
    // channels of the same rows are in the partner wave (wave ^ 1): the two swap halves through the (now idle) stage - wave 2 q takes
    // the row groups of h = 0, wave 2 q + 1 those of h = 1, each 4 chunks of 16 rows with all 96 input channels - and the skip
    // channels come from memory.  The layer's 64-channel output is never written: 512 bytes per row less HBM traffic than the
    // separate tail kernel (256 written, 256 read back), 1.96 GB on the 3.8 M-row bench batch.
    const TailFuse& tf = a.tail;
    const bool odd = (wave & 1) != 0;
    uint4 Xh[NG], Xl[NG];
#pragma unroll
    for (int hc = 0; hc < NG; ++hc) {
      uint2 h0, l0, h1, l1;
      split16_encode4(v[hc][0], h0, l0);
      split16_encode4(v[hc][1], h1, l1);
      Xh[hc] = make_uint4(h0.x, h0.y, h1.x, h1.y);
      Xl[hc] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
    auto sel4 = [](bool c, const uint4 x, const uint4 y) { return make_uint4(c ? x.x : y.x, c ? x.y : y.y, c ? x.z : y.z, c ? x.w : y.w); };
    constexpr int NK = NG / 2;                                           // my 4 chunks: row groups odd * 4 + k
    int myrow[NK];
    uint4 Oh[NK], Ol[NK], Sh[NK], Sl[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      myrow[k] = odd ? orow[NK + k] : orow[k];
      Oh[k] = sel4(odd, Xh[NK + k], Xh[k]);
      Ol[k] = sel4(odd, Xl[NK + k], Xl[k]);
      const int o = myrow[k] < a.n_out ? myrow[k] : a.n_out - 1;         // ragged tile: a valid row, never stored
      const char* sp = reinterpret_cast<const char*>(tf.skip + (size_t)o * tf.ld_skip) + g * 16;
      Sh[k] = *reinterpret_cast<const uint4*>(sp);
      Sl[k] = *reinterpret_cast<const uint4*>(sp + SPLIT16_LO);
    }
    // the tail's weights start their way from L2 before the exchange: the first input block's fragments of conv1_tr, all of final's
    // (index algebra in spconv_tail.hip), shifts and scales - behind the barriers each would be a round trip of its own
    auto load_w1 = [&](int cc, half8_t (&W)[4][2]) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          W[nt][p] = *reinterpret_cast<const half8_t*>(reinterpret_cast<const char*>(tf.w1) + ((cc * 4 + nt) * 2 + p) * 1024 + lane * 16);
    };
    half8_t Wc[2][4][2];
    load_w1(0, Wc[0]);
    half8_t W2[2][2][2];
    auto load_w2 = [&]() {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const char* f = reinterpret_cast<const char*>(tf.w2) + (nt * 4 + 2 * kb + p) * 1024 + (((g >> 1) * 16 + j) * 16) + 8 * (g & 1);
            const uint2 lo = *reinterpret_cast<const uint2*>(f), hi = *reinterpret_cast<const uint2*>(f + 512);
            W2[nt][kb][p] = __builtin_bit_cast(half8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
          }
    };
    const float os1 = *tf.s1, os2 = *tf.s2;
    float4 b1[4], b2[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) b1[t] = *reinterpret_cast<const float4*>(tf.b1 + 16 * t + 4 * g);
#pragma unroll
    for (int t = 0; t < 2; ++t) b2[t] = *reinterpret_cast<const float4*>(tf.b2 + 16 * t + 4 * g);
    lds_barrier();                                                       // every wave has left its offset loop: the stage is free
    unsigned char* xw = xs + (size_t)(wave * NK) * 2048 + lane * 16;
#pragma unroll
    for (int k = 0; k < NK; ++k) {                                       // the partner's chunks go out
      *reinterpret_cast<uint4*>(xw + k * 2048) = sel4(odd, Xh[k], Xh[NK + k]);
      *reinterpret_cast<uint4*>(xw + k * 2048 + 1024) = sel4(odd, Xl[k], Xl[NK + k]);
    }
    lds_barrier();
    const unsigned char* xr = xs + (size_t)((wave ^ 1) * NK) * 2048 + lane * 16;
    uint4 Ph[NK], Pl[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      Ph[k] = *reinterpret_cast<const uint4*>(xr + k * 2048);
      Pl[k] = *reinterpret_cast<const uint4*>(xr + k * 2048 + 1024);
    }
    // conv1_tr: input blocks in the order 0, 1, 2 (decoder channels 0..31, 32..63, skip), the three split16 terms inside - the
    // summation order of tail_fused_kernel.  One block's weight fragments at a time (32 registers), four chunks of accumulators.
    f32x4 acc1[NK][4];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc1[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      if (cc < 2) load_w1(cc + 1, Wc[(cc + 1) & 1]);                     // the next block's fragments fly during this block's products
      else load_w2();                                                    // ... and final's during the last block's
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const bool mine = cc == (odd ? 1 : 0);
        const uint4 xh = cc == 2 ? Sh[k] : sel4(mine, Oh[k], Ph[k]), xl = cc == 2 ? Sl[k] : sel4(mine, Ol[k], Pl[k]);
        const half8_t X0 = __builtin_bit_cast(half8_t, xh), X1 = __builtin_bit_cast(half8_t, xl);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc1[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc[cc & 1][t][term == 2 ? 1 : 0], term == 1 ? X1 : X0, acc1[k][t], 0, 0, 0);
      }
    }
    // final: the lane's 16 intermediate values are its K elements of the second product
    // range guard: an overflow in a layer upstream (all finished: word 0) poisons every row; one in THIS layer's 64-channel output -
    // which other workgroups may still be computing - raises word 4, and the launcher's k_tail_poison answers for it after the
    // kernel; one in the tail's own intermediate poisons the row it happened in (as in tail_fused_kernel) and raises word 5.  Neither
    // touches word 0 while the kernel runs (a later workgroup would take it for an upstream overflow): k_tail_poison folds them in
    const bool poisoned = split16_poisoned(a.range) || split16_over(mx);
    const float qnan = __builtin_nanf("");
    float mx2 = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      half8_t Y[2][2];
      float cmx = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 u = make_float4(acc1[k][t][0] * os1 + b1[t].x, acc1[k][t][1] * os1 + b1[t].y, acc1[k][t][2] * os1 + b1[t].z, acc1[k][t][3] * os1 + b1[t].w);
        if (tf.relu1) { u.x = fmaxf(u.x, 0.f); u.y = fmaxf(u.y, 0.f); u.z = fmaxf(u.z, 0.f); u.w = fmaxf(u.w, 0.f); }
        split16_track(cmx, u);
        uint2 h, l;
        split16_encode4(u, h, l);
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
      float4 w[2];
      float ss = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        w[t] = make_float4(o2[t][0] * os2 + b2[t].x, o2[t][1] * os2 + b2[t].y, o2[t][2] * os2 + b2[t].z, o2[t][3] * os2 + b2[t].w);
        ss += w[t].x * w[t].x + w[t].y * w[t].y + w[t].z * w[t].z + w[t].w * w[t].w;
      }
      if (tf.l2norm) {                                                   // the row's 32 channels: this lane and lanes j + 16, 32, 48
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float inv = 1.0f / sqrtf(ss);                              // no epsilon: a zero row gives NaN (0 * inf), like the reference
#pragma unroll
        for (int t = 0; t < 2; ++t) { w[t].x *= inv; w[t].y *= inv; w[t].z *= inv; w[t].w *= inv; }
      }
      if (myrow[k] < a.n_out) mx2 = split16_merge(mx2, cmx);
      cmx = split16_merge(cmx, __shfl_xor(cmx, 16, 64));
      cmx = split16_merge(cmx, __shfl_xor(cmx, 32, 64));
      const bool bad = poisoned || split16_over(cmx);
      if (myrow[k] < a.n_out) {
        const size_t oo = tf.out_perm ? (size_t)tf.out_perm[myrow[k]] : (size_t)myrow[k];
#pragma unroll
        for (int t = 0; t < 2; ++t)
          *reinterpret_cast<float4*>(tf.out + oo * tf.ld_out + 16 * t + 4 * g) = bad ? make_float4(qnan, qnan, qnan, qnan) : w[t];
      }
    }
    if (a.range) {
      if (split16_over(mx)) { atomicOr(a.range + 4, 1u); atomicOr(a.range + 3, 1u); }
      if (split16_over(mx2)) { atomicOr(a.range + 5, 1u); atomicOr(a.range + 3, 1u); }
      if (__builtin_nontemporal_load(a.range + 2)) atomicMax(a.range + 1, __float_as_uint(split16_merge(mx, mx2)) & 0x7FFFFFFFu);   // probe mode only
    }
    return;
  }
#pragma unroll
  for (int hc = 0; hc < NG; ++hc) {
    const int o = orow[hc];
    if (o >= a.n_out) continue;
    if (SKIP == 7 && os != 12345.f) continue;                          // EYOC_ST_ABLATIONS: no epilogue (the accumulators stay live)
    if (a.out_split) {
      uint2 h0, l0, h1, l1;
      split16_encode4(v[hc][0], h0, l0);
      split16_encode4(v[hc][1], h1, l1);
      char* op = reinterpret_cast<char*>(a.out + (size_t)o * a.ld_out) + split16_off4(ch);
      *reinterpret_cast<uint4*>(op) = make_uint4(h0.x, h0.y, h1.x, h1.y);
      *reinterpret_cast<uint4*>(op + SPLIT16_LO) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    } else {
      float* op = a.out + (size_t)o * a.ld_out + ch;
      *reinterpret_cast<float4*>(op) = v[hc][0];
      *reinterpret_cast<float4*>(op + 4) = v[hc][1];
    }
  }
  if (a.out_split) split16_report(a.range, mx);
#ifdef EYOC_ST_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ST_STAMP(8);
  if (threadIdx.x == 0 && blockIdx.x < TRACE_WGS) {
    unsigned int hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_st_trace[blockIdx.x * TRACE_N + 9] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
}

}  // namespace

#ifdef EYOC_ST_TRACE
extern "C" int eyoc_debug_st_trace(unsigned long long* host, size_t n) {
  if (n > (size_t)TRACE_WGS * TRACE_N) n = (size_t)TRACE_WGS * TRACE_N;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_st_trace), n * 8) != hipSuccess) return -1;
  unsigned long long* z = (unsigned long long*)calloc(n, 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_st_trace), z, n * 8);
  free(z);
  return 0;
}
#endif

namespace eyoc {

size_t local_rulebook_bytes(int n_out) { return (size_t)cdiv(n_out, TILE) * LR_BYTES; }

// builds the per-tile local rulebooks of a stride-1 table; *overflow_dev (zeroed by the caller) counts tiles with more
// than NPASS * UMAX distinct input rows (the staged kernel must not be used for the table then)
// group: 1 = the tile's rows sorted by neighbour pattern (eyoc_ctx::Knobs::st_group, the default), 0 = in their own order
int build_local_rulebook(const int32_t* nbr_dev, int K, int n_out, unsigned char* out_dev, int* overflow_dev, hipStream_t st, int group) {
  if (n_out <= 0) return EYOC_OK;
  hipLaunchKernelGGL((k_local_rulebook<false, 4>), dim3(cdiv(n_out, TILE)), dim3(256), 0, st, nbr_dev, K, n_out, out_dev, overflow_dev, group, DeriveSrc());
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

int build_local_rulebook_derived(const DeriveSrc& src, int n_out, unsigned char* out_dev, int* overflow_dev, hipStream_t st, int group) {
  if (n_out <= 0) return EYOC_OK;
  EYOC_REQUIRE(src.coords && src.parent && src.children && src.s1c && src.nc > 0, EYOC_ERR_INVALID, "build_local_rulebook_derived: incomplete octree");
  hipLaunchKernelGGL((k_local_rulebook<true, 4>), dim3(cdiv(n_out, TILE)), dim3(256), 0, st, (const int32_t*)nullptr, 27, n_out, out_dev, overflow_dev, group, src);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

// which staged kernel runs: 0 = the C++ offset loop (spconv_st_kernel), 1 = the assembly loop (default), 2 = the assembly loop
// without the empty-block branches (diagnostics); -DEYOC_ST_ABLATIONS builds add 3 = operand reads only for non-empty blocks
// (round 5: level) and 13 ... = timing-only ablations
#ifdef EYOC_ST_ABLATIONS
constexpr int ST_VARIANTS = 28;
#else
// 64-row waves (NH = 1: 32-channel layers, 128-row strided tiles of 64-channel layers): 1 (default, round 6) weights three offsets ahead
// and rulebook entries five - an offset of 4 chunks lasts ~300 cycles, "two ahead" is less than an L2 round trip: 0.56 -> 0.54, 0.61 -> 0.58,
// 0.52 -> 0.51 ms on the bench's three such layers; 4 = the round-5 loop (two ahead); 5 / 6 = operand reads for non-empty blocks only,
// without / with the deeper prefetch: level (the NH = 1 loops are not bound by LDS reads either)
constexpr int ST_VARIANTS = 7;      // 4 = variant 1 with the deep-prefetch loop in the NH = 1 kernels (32-channel layers, 128-row strided tiles)
#endif
int st_variants() { return ST_VARIANTS; }

// the 1x1 tail can ride in the epilogue of a layer that launch_spconv_st runs as 256 rows x 64 channels per workgroup on the default
// assembly loop (NH = 2): a 64-channel stride-1 layer of a batch (small inputs take 32-channel workgroups)
bool spconv_st_can_fuse_tail(const SpconvArgs& a) {
  if (a.math != 1 || !a.local || a.l2norm || a.K != 27 || a.cout != 64 || a.cin % 32 != 0 || !a.out_split || a.out_perm) return false;
  const eyoc_ctx::Knobs& kn = knobs_of(a.ctx);
  const int v = kn.st_variant;
  if (!(v == 1 || v == 4 || v == 5 || v == 6)) return false;           // the NH = 2 loop of these variants is the default one
  const int n_tiles = cdiv(a.n_out, TILE);
  return !((long long)n_tiles * (a.cout / 64) < kn.st_split_below);
}

// after a kernel with a fused tail: an overflow in that layer's own 64-channel output (range word 4; workgroups that finished before
// it was seen wrote clean rows) turns the whole output into NaN rows, as the separate tail kernel's `poisoned` did
__global__ __launch_bounds__(256) void k_tail_poison(unsigned int* __restrict__ range, float* __restrict__ out, int n, int ld, int c) {
  const unsigned int stored = __builtin_nontemporal_load(range + 4), inter = __builtin_nontemporal_load(range + 5);
  if (blockIdx.x == 0 && threadIdx.x == 0 && (stored | inter)) atomicOr(range, 1u);   // the forward's own flag (word 0), now that no workgroup reads it any more
  if (stored == 0u) return;
  const float qnan = __builtin_nanf("");
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)n * c; i += (long long)gridDim.x * 256)
    out[(i / c) * ld + i % c] = qnan;
}

// stride-1 SPLIT16 layers whose table has a local rulebook (rows in natural = Morton order, no tiling permutation)
int launch_spconv_st(const SpconvArgs& a_in, const unsigned char* local_dev, hipStream_t st) {
  const SpconvArgs& a = a_in;
  EYOC_REQUIRE(a.math == 1 && local_dev && !a.l2norm && a.K == 27 && !a.perm, EYOC_ERR_INVALID, "spconv_st: unsupported layer");
  const int n_tiles = cdiv(a.n_out, TILE);
  // output channels per workgroup: 64 (waves of 128 rows x 32 channels) when that gives the 512 workgroup slots of the chip a few
  // rounds of work; small problems (a batch of 8 pairs: 70 tiles at the coarsest level) take 32 (waves of 64 rows x 32 channels,
  // twice the workgroups, each half as long) - below ~2 rounds a layer lasts as long as ONE workgroup does
  // (measured: single pair 2.08 -> 1.84 ms with the threshold at 1024 workgroups; a batch of 8 pairs does not care)
  const eyoc_ctx::Knobs& kn = knobs_of(a.ctx);
  const int split_below = kn.st_split_below;
  const int variant0 = kn.st_variant;
  const bool small = variant0 != 0 && a.cout >= 64 && a.cout <= 256 && (long long)n_tiles * (a.cout / 64) < split_below;
  const int ctg = a.cout >= 64 && !small ? 64 : 32;                  // output channels per workgroup
  const bool wide = spconv_cc(a.cin, a.cout) == 64;
  const int n_cg = a.cout / ctg;
  EYOC_REQUIRE(a.cout % ctg == 0 && n_cg >= 1 && n_cg <= 8 && 8 % n_cg == 0 && a.cin % 32 == 0, EYOC_ERR_INVALID,
               "spconv_st: %d -> %d channels", a.cin, a.cout);
  SpconvArgs a_loc = a_in;
  // (see the kernel's workgroup -> tile mapping) the packed split16 kernel of the layer against half an L2
  a_loc.cg_local = (kn.st_cg_local && n_cg > 1 && variant0 != 0 && (size_t)27 * a.cin * a.cout * 4 <= (size_t)2 << 20) ? 1 : 0;
  const SpconvArgs& a2 = a_loc;
  dim3 grid((unsigned)(a_loc.cg_local ? cdiv(n_tiles, 8) * 8 * n_cg : cdiv(n_tiles, 8 / n_cg) * 8)), block(NW * 64);
  const int variant = kn.st_variant;
  // small inputs (32-channel workgroups that do not fill the chip's 512 slots): the 32-channel input blocks of a tile split over
  // ksplit workgroups, as many as keep workgroups x splits within the scratch (and never more than there are blocks)
  int ksplit = 1;
  if (variant == 1 && ctg == 32 && a.ks_part && kn.st_ksplit && a.cin >= 128) {   // (two blocks: the second launch costs what the split saves)
    const int wgs = (int)grid.x, nqb = a.cin / 32;
    ksplit = KS_MAX_SLOTS / (wgs > 0 ? wgs : 1);
    if (ksplit > nqb) ksplit = nqb;
    if (ksplit < 4 || (long long)n_tiles * n_cg > KS_MAX_SLOTS) ksplit = 1;
    grid.y = (unsigned)ksplit;
  }
  if (variant == 0) {
    // the dynamic-LDS attribute is per (function, device): set on every launch of this diagnostics path (no process-wide cache)
#define EYOC_ST(NTW_, CC_, NH_)                                                                                             \
  do {                                                                                                                      \
    EYOC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spconv_st_kernel<NTW_, CC_, NH_>),                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, X_BYTES));                               \
    hipLaunchKernelGGL((spconv_st_kernel<NTW_, CC_, NH_>), grid, block, (size_t)X_BYTES, st, a, local_dev, n_tiles);        \
  } while (0)
    if (ctg == 64) { if (wide) EYOC_ST(2, 64, 2); else EYOC_ST(2, 32, 2); }
    else { if (wide) EYOC_ST(2, 64, 1); else EYOC_ST(2, 32, 1); }
#undef EYOC_ST
  } else {
#define EYOC_STA(CC_, NH_, SK_)                                                                                          \
  do {                                                                                                                  \
    hipLaunchKernelGGL((spconv_st_asm_kernel<CC_, NH_, SK_>), grid, block, 0, st, a2, local_dev, n_tiles, ksplit, 0);    \
    if (ksplit > 1)                                                                                                     \
      hipLaunchKernelGGL((spconv_st_asm_kernel<CC_, NH_, SK_>), dim3(grid.x), block, 0, st, a2, local_dev, n_tiles, ksplit, 1); \
  } while (0)
    if (ctg == 64) {
      if (variant == 2) { if (wide) EYOC_STA(64, 2, 0); else EYOC_STA(32, 2, 0); }
#ifdef EYOC_ST_ABLATIONS
      else if (variant == 3) { if (wide) EYOC_STA(64, 2, 2); else EYOC_STA(32, 2, 2); }
      else if (variant == 13) EYOC_STA(64, 2, 3);
      else if (variant == 14) EYOC_STA(64, 2, 4);
      else if (variant == 15) EYOC_STA(64, 2, 5);
      else if (variant == 16) EYOC_STA(64, 2, 6);
      else if (variant == 17) EYOC_STA(64, 2, 7);
      else if (variant == 18) EYOC_STA(64, 2, 8);
      else if (variant == 19) EYOC_STA(64, 2, 9);
      else if (variant == 20) EYOC_STA(64, 2, 10);
      else if (variant == 21) EYOC_STA(64, 2, 11);
      else if (variant == 22) EYOC_STA(64, 2, 12);
      else if (variant == 23) EYOC_STA(64, 2, 13);
      else if (variant == 24) EYOC_STA(64, 2, 14);
      else if (variant == 25) EYOC_STA(64, 2, 15);
      else if (variant == 26) EYOC_STA(64, 2, 16);
      else if (variant == 27) EYOC_STA(64, 2, 17);
#endif
      else if (a.tail.out) {                                             // the 1x1 tail in the epilogue (spconv_st_can_fuse_tail said yes)
        EYOC_REQUIRE(spconv_st_can_fuse_tail(a) && a.range, EYOC_ERR_INVALID, "spconv_st: this layer cannot carry the fused tail");
        if (wide) hipLaunchKernelGGL((spconv_st_asm_kernel<64, 2, 1, NW, TILE, 1>), grid, block, 0, st, a2, local_dev, n_tiles, 1, 0);
        else hipLaunchKernelGGL((spconv_st_asm_kernel<32, 2, 1, NW, TILE, 1>), grid, block, 0, st, a2, local_dev, n_tiles, 1, 0);
        hipLaunchKernelGGL(k_tail_poison, dim3(256), dim3(256), 0, st, a.range, a.tail.out, a.n_out, a.tail.ld_out, 32);
      }
      else { if (wide) EYOC_STA(64, 2, 1); else EYOC_STA(32, 2, 1); }
    } else if (variant == 4) { if (wide) EYOC_STA(64, 1, 1); else EYOC_STA(32, 1, 1); }
    else if (variant == 5) { if (wide) EYOC_STA(64, 1, 19); else EYOC_STA(32, 1, 19); }
    else if (variant == 6) { if (wide) EYOC_STA(64, 1, 20); else EYOC_STA(32, 1, 20); }
    else { if (wide) EYOC_STA(64, 1, 18); else EYOC_STA(32, 1, 18); }
#undef EYOC_STA
  }
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

// strided (2ts <- ts) SPLIT16 layers whose table has 128-row tile records (build_local_rulebook128): 128 output rows x 64 channels per
// workgroup, the NH = 1 assembly loop
int launch_spconv_st128(const SpconvArgs& a, const unsigned char* local_dev, hipStream_t st) {
  EYOC_REQUIRE(a.math == 1 && local_dev && !a.l2norm && a.K == 27 && !a.perm, EYOC_ERR_INVALID, "spconv_st128: unsupported layer");
  const eyoc_ctx::Knobs& kn128 = knobs_of(a.ctx);
  // >= 128 output channels: workgroups of 128 rows x 128 channels (waves of 128 rows x 32 channels, the NH = 2 loop) - the tile's rows
  // are staged once per 128 channels and every weight fragment serves 8 chunks
  const bool wide_wg = a.cout % 128 == 0 && kn128.st128_wide != 0;
  const int n_tiles = cdiv(a.n_out, 128), n_cg = a.cout / (wide_wg ? 128 : 64);
  EYOC_REQUIRE(a.cout % 64 == 0 && n_cg >= 1 && n_cg <= 8 && 8 % n_cg == 0 && a.cin % 32 == 0, EYOC_ERR_INVALID, "spconv_st128: %d -> %d channels",
               a.cin, a.cout);
  const dim3 grid((unsigned)(cdiv(n_tiles, 8 / n_cg) * 8)), block(NW * 64);
  if (wide_wg) {
    if (spconv_cc(a.cin, a.cout) == 64) hipLaunchKernelGGL((spconv_st_asm_kernel<64, 2, 1, NW, 128>), grid, block, 0, st, a, local_dev, n_tiles, 1, 0);
    else hipLaunchKernelGGL((spconv_st_asm_kernel<32, 2, 1, NW, 128>), grid, block, 0, st, a, local_dev, n_tiles, 1, 0);
    EYOC_CHECK_HIP(hipGetLastError());
    return EYOC_OK;
  }
  const int nh1 = knobs_of(a.ctx).st_variant;   // 4 the round-5 loop (two offsets ahead), 5 lazy operand reads, 6 lazy + deep; else deep prefetch
  const bool deep = nh1 != 4;
#ifdef EYOC_ST_ABLATIONS       // timing-only (results are garbage): 16 no stage, 17 no epilogue, 23 nothing in the loop, 18 no MFMAs
  const int v_ = knobs_of(a.ctx).st_variant;
#define EYOC_ST128_ABL(SK_)                                                                                                         \
  do {                                                                                                                              \
    if (spconv_cc(a.cin, a.cout) == 64) hipLaunchKernelGGL((spconv_st_asm_kernel<64, 1, SK_, NW, 128>), grid, block, 0, st, a, local_dev, n_tiles, 1, 0); \
    else hipLaunchKernelGGL((spconv_st_asm_kernel<32, 1, SK_, NW, 128>), grid, block, 0, st, a, local_dev, n_tiles, 1, 0);          \
    EYOC_CHECK_HIP(hipGetLastError());                                                                                              \
    return EYOC_OK;                                                                                                                 \
  } while (0)
  if (v_ == 16) EYOC_ST128_ABL(6);
  if (v_ == 17) EYOC_ST128_ABL(7);
  if (v_ == 23) EYOC_ST128_ABL(13);
  if (v_ == 18) EYOC_ST128_ABL(8);
  if (v_ == 19) EYOC_ST128_ABL(9);
  if (v_ == 20) EYOC_ST128_ABL(10);
  if (v_ == 21) EYOC_ST128_ABL(11);
#undef EYOC_ST128_ABL
#endif
  if (nh1 == 5 || nh1 == 6) {
    if (spconv_cc(a.cin, a.cout) == 64) {
      if (nh1 == 5) hipLaunchKernelGGL((spconv_st_asm_kernel<64, 1, 19, NW, 128>), grid, block, 0, st, a, local_dev, n_tiles, 1, 0);
      else hipLaunchKernelGGL((spconv_st_asm_kernel<64, 1, 20, NW, 128>), grid, block, 0, st, a, local_dev, n_tiles, 1, 0);
    } else {
      if (nh1 == 5) hipLaunchKernelGGL((spconv_st_asm_kernel<32, 1, 19, NW, 128>), grid, block, 0, st, a, local_dev, n_tiles, 1, 0);
      else hipLaunchKernelGGL((spconv_st_asm_kernel<32, 1, 20, NW, 128>), grid, block, 0, st, a, local_dev, n_tiles, 1, 0);
    }
    EYOC_CHECK_HIP(hipGetLastError());
    return EYOC_OK;
  }
  if (spconv_cc(a.cin, a.cout) == 64) {
    if (deep) hipLaunchKernelGGL((spconv_st_asm_kernel<64, 1, 18, NW, 128>), grid, block, 0, st, a, local_dev, n_tiles, 1, 0);
    else hipLaunchKernelGGL((spconv_st_asm_kernel<64, 1, 1, NW, 128>), grid, block, 0, st, a, local_dev, n_tiles, 1, 0);
  } else {
    if (deep) hipLaunchKernelGGL((spconv_st_asm_kernel<32, 1, 18, NW, 128>), grid, block, 0, st, a, local_dev, n_tiles, 1, 0);
    else hipLaunchKernelGGL((spconv_st_asm_kernel<32, 1, 1, NW, 128>), grid, block, 0, st, a, local_dev, n_tiles, 1, 0);
  }
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

size_t local_rulebook128_bytes(int n_out) { return (size_t)cdiv(n_out, 128) * LR_BYTES; }

int build_local_rulebook128(const int32_t* nbr_dev, int K, int n_out, unsigned char* out_dev, int* overflow_dev, hipStream_t st, int group) {
  if (n_out <= 0) return EYOC_OK;
  hipLaunchKernelGGL((k_local_rulebook<false, 2>), dim3(cdiv(n_out, 128)), dim3(128), 0, st, nbr_dev, K, n_out, out_dev, overflow_dev, group, DeriveSrc());
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

}  // namespace eyoc
