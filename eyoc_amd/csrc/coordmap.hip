// Coordinate maps and rulebooks: the device-side replacement of MinkowskiEngine's coordinate
// manager for the ResUNet path (ME.SparseTensor at scripts/test_kitti.py:143-148; the kernel maps
// behind model/resunet.py:31-116 and model/residual_block.py:23-33).
//
//  * keys: (batch,x,y,z) packed into 64 bits, open addressing (linear probing, load <= 0.5) in HBM;
//    insertion is atomicCAS on the key + atomicMin on the value, so the value of a slot is the
//    smallest row index that mapped to it - this makes "first occurrence" deterministic.
//  * level l+1 = unique floor(c / 2ts) * 2ts of level l: insert coarse keys, flag the rows that won
//    their slot, exclusive-scan the flags, compact -> coarse rows ordered by first occurrence.
//  * every fine row records its coarse parent and every coarse row its (up to 8) children, so the
//    levels form an octree.  Only the COARSEST level's 27-neighbour table is found by hash probes;
//    every finer table (stride-1, strided, transposed) is derived top-down from the table one level
//    up: a neighbour of a voxel lives in one of 8 coarse blocks around its parent, and is that
//    block's child in a known slot.  Per fine row that is 8 table reads + 8 child vectors instead of
//    27 + 27 + 27 hash probes (3/4 of which would be misses that walk a probe chain).
//  * rulebooks are output-stationary tables nbr[27][n_out] (k-major so that the 64 rows of a wave
//    read/write one contiguous segment per offset).
#include <cstdlib>

#include "spconv.h"
#include "derive.h"

using namespace eyoc;

namespace {

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_TILE = SCAN_ITEMS * SCAN_BLOCK;  // 2048 flags per block

__device__ inline void coarse_coord(const int32_t* c, int ts2, int& b, int& x, int& y, int& z) {
  const int m = ~(ts2 - 1);  // ts2 is a power of two: floor to a multiple of ts2, also for negatives
  b = c[0]; x = c[1] & m; y = c[2] & m; z = c[3] & m;
}

// ts2 == 1: keys of the rows themselves.  err[0] counts out-of-range coordinates.
__global__ void k_insert(const int32_t* __restrict__ coords, int n, int ts2, HashTable t, int* __restrict__ slot_out,
                         int* __restrict__ err) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int b, x, y, z;
  coarse_coord(coords + 4 * (size_t)i, ts2, b, x, y, z);
  constexpr int LIM = COORD_BIAS - 16;  // margin: neighbour probes reach +-8 voxels at tensor stride 8
  if (b < 0 || b >= 1024 || x < -LIM || x >= LIM || y < -LIM || y >= LIM || z < -LIM || z >= LIM) {
    if (err) atomicAdd(&err[0], 1);
    if (slot_out) slot_out[i] = 0;
    return;
  }
  const unsigned long long key = pack_key(b, x, y, z);
  unsigned int s = hash_key(key) & t.mask;
  while (true) {
    unsigned long long prev = atomicCAS(&t.keys[s], KEY_EMPTY, key);
    if (prev == KEY_EMPTY || prev == key) break;
    s = (s + 1) & t.mask;
  }
  atomicMin(&t.vals[s], i);
  if (slot_out) slot_out[i] = (int)s;
}

// flag[i] = 1 iff row i is the first row of its slot; for ts2 == 1 any 0 flag is a duplicate row
__global__ void k_flag(const int* __restrict__ slot, const int* __restrict__ vals, int n, int* __restrict__ flag,
                       int* __restrict__ dup) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int f = vals[slot[i]] == i;
  flag[i] = f;
  if (dup && !f) atomicAdd(dup, 1);
}

__device__ inline int block_exclusive_scan(int v, int* total) {
  __shared__ int wave_sum[SCAN_BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
    int s = wave_sum[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_partials(const int* __restrict__ flag, int n, int* __restrict__ partial) {
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) s += (base + j < n) ? flag[base + j] : 0;
  int tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// single block: exclusive scan of partial[0..nb) in place, total -> *total
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_top(int* __restrict__ partial, int nb, int* __restrict__ total) {
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += SCAN_BLOCK) {
    int i = b0 + threadIdx.x;
    int v = i < nb ? partial[i] : 0;
    int tot;
    int ex = block_exclusive_scan(v, &tot);
    if (i < nb) partial[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

// compaction: first rows write their coarse coordinate at pos and re-label their slot with it
__global__ __launch_bounds__(SCAN_BLOCK) void k_compact(const int* __restrict__ flag, const int* __restrict__ partial,
                                                        const int* __restrict__ slot, const int32_t* __restrict__ coords,
                                                        int n, int ts2, int32_t* __restrict__ coords_out,
                                                        int* __restrict__ vals, int32_t* __restrict__ sel_out) {
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int f[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    f[j] = (base + j < n) ? flag[base + j] : 0;
    s += f[j];
  }
  int tot;
  int pos = partial[blockIdx.x] + block_exclusive_scan(s, &tot);
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    if (f[j]) {
      const int i = base + j;
      int b, x, y, z;
      coarse_coord(coords + 4 * (size_t)i, ts2, b, x, y, z);
      int4 c = make_int4(b, x, y, z);
      reinterpret_cast<int4*>(coords_out)[pos] = c;
      vals[slot[i]] = pos;
      if (sel_out) sel_out[pos] = i;
      ++pos;
    }
  }
}

// ---- Z-ordered levels without a hash table: the rows of level l-1 are sorted by Morton key, so the rows of one coarse
// voxel are adjacent and a coarse voxel's first occurrence is where the coarse coordinate changes.  (The hash insert of
// the 3.8 M level-0 rows of the 64-pair batch cost 0.5 ms; it stays for the coarsest level, whose table the 27-probe
// neighbour search needs, and for batches in the caller's order.)
__global__ void k_flag_sorted(const int32_t* __restrict__ coords, int n, int ts2, int* __restrict__ flag, int* __restrict__ err,
                              int* __restrict__ dup) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int b, x, y, z;
  coarse_coord(coords + 4 * (size_t)i, ts2, b, x, y, z);
  constexpr int LIM = COORD_BIAS - 16;
  if (err && (b < 0 || b >= 1024 || x < -LIM || x >= LIM || y < -LIM || y >= LIM || z < -LIM || z >= LIM)) atomicAdd(err, 1);
  int f = 1;
  if (i > 0) {
    int pb, px, py, pz;
    coarse_coord(coords + 4 * (size_t)(i - 1), ts2, pb, px, py, pz);
    f = (pb != b || px != x || py != y || pz != z) ? 1 : 0;
    if (dup) {   // level 0 -> 1: two equal rows are a duplicate coordinate
      const int4 c = reinterpret_cast<const int4*>(coords)[i], q = reinterpret_cast<const int4*>(coords)[i - 1];
      if (c.x == q.x && c.y == q.y && c.z == q.z && c.w == q.w) atomicAdd(dup, 1);
    }
  }
  flag[i] = f;
}

// Z-ordered rows: the row counts of ALL coarser levels from the sorted level-0 rows in one pass (a level's voxels are runs of
// adjacent rows at every level), so that eyoc_maps_build reads them with ONE synchronisation instead of one per level
// counts[0]: rows that do not fit the speculated key width (+-2^kbits, batch < 2^bbits); counts[EYOC_MAX_LEVELS + 2 / + 3]: the
// largest |coordinate| / batch index (the next build's speculation)
__global__ __launch_bounds__(256) void k_count_levels(const int32_t* __restrict__ coords, int n, int kbits, int bbits,
                                                      int* __restrict__ counts /* [EYOC_MAX_LEVELS + 4] */) {
  // grid-stride: a few thousand atomics on the counters in all (one per level and workgroup), not one per wave of 64 rows -
  // 180 k atomics on three words cost 2 ms on the 3.8 M-row batch
  __shared__ int part[EYOC_MAX_LEVELS + 2];
  if (threadIdx.x < EYOC_MAX_LEVELS + 2) part[threadIdx.x] = 0;
  __syncthreads();
  int cnt[EYOC_MAX_LEVELS + 2] = {};
  int amax = 0, bmax = 0;
  constexpr int LIM = COORD_BIAS - 16;
  const int klim = 1 << kbits;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const int4 q = i > 0 ? reinterpret_cast<const int4*>(coords)[i - 1] : c;
    if (c.x < 0 || (c.x >> bbits) != 0 || c.y < -klim || c.y >= klim || c.z < -klim || c.z >= klim || c.w < -klim || c.w >= klim) ++cnt[0];
    amax = max(amax, max(max(c.y < 0 ? ~c.y : c.y, c.z < 0 ? ~c.z : c.z), c.w < 0 ? ~c.w : c.w));   // v fits iff -2^k <= v < 2^k iff (v < 0 ? ~v : v) < 2^k
    bmax = max(bmax, c.x);
    // validation here too: the build must not run its table kernels on keys outside the supported range
    if (c.x < 0 || c.x >= 1024 || c.y < -LIM || c.y >= LIM || c.z < -LIM || c.z >= LIM || c.w < -LIM || c.w >= LIM) ++cnt[EYOC_MAX_LEVELS];
    if (i > 0 && c.x == q.x && c.y == q.y && c.z == q.z && c.w == q.w) ++cnt[EYOC_MAX_LEVELS + 1];
#pragma unroll
    for (int l = 1; l < EYOC_MAX_LEVELS; ++l) {
      const int m = ~((1 << l) - 1);
      if (i == 0 || c.x != q.x || (c.y & m) != (q.y & m) || (c.z & m) != (q.z & m) || (c.w & m) != (q.w & m)) ++cnt[l];
    }
  }
#pragma unroll
  for (int l = 0; l < EYOC_MAX_LEVELS + 2; ++l) {
    int v = cnt[l];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&part[l], v);
  }
  __shared__ int pmax[2];
  if (threadIdx.x < 2) pmax[threadIdx.x] = 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { amax = max(amax, __shfl_xor(amax, d, 64)); bmax = max(bmax, __shfl_xor(bmax, d, 64)); }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { atomicMax(&pmax[0], amax); atomicMax(&pmax[1], bmax); }
  __syncthreads();
  if (threadIdx.x < EYOC_MAX_LEVELS + 2 && part[threadIdx.x]) atomicAdd(counts + threadIdx.x, part[threadIdx.x]);
  if (threadIdx.x < 2 && pmax[threadIdx.x] > 0) atomicMax(counts + EYOC_MAX_LEVELS + 2 + threadIdx.x, pmax[threadIdx.x]);
}

// compaction + octree links in one pass: first rows write their coarse coordinate, every row learns its parent (the
// number of first rows up to and including it, minus one) and enters itself as that parent's child
// One lane per row, 64 consecutive rows per wave step (coalesced coordinate reads and parent writes; the first version gave
// every thread 8 consecutive rows - every load instruction of a wave then touched 64 different lines): the position of a row's
// coarse voxel is the block's scanned base + the flags before it, counted with ballots.
__global__ __launch_bounds__(SCAN_BLOCK) void k_compact_sorted(const int* __restrict__ flag, const int* __restrict__ partial,
                                                               const int32_t* __restrict__ coords, int n, int ts2, int sh,
                                                               int32_t* __restrict__ coords_out, int32_t* __restrict__ parent,
                                                               int32_t* __restrict__ children) {
  constexpr int NWV = SCAN_BLOCK / 64, PER_WAVE = SCAN_TILE / NWV, STEPS = PER_WAVE / 64;
  __shared__ int wave_tot[NWV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = blockIdx.x * SCAN_TILE + wave * PER_WAVE;
  int f[STEPS];
  int tot = 0;
#pragma unroll
  for (int j = 0; j < STEPS; ++j) {
    const int i = base + j * 64 + lane;
    f[j] = i < n ? flag[i] : 0;
    tot += __popcll(__ballot(f[j] != 0));
  }
  if (lane == 0) wave_tot[wave] = tot;
  __syncthreads();
  int pos0 = partial[blockIdx.x];
  for (int w = 0; w < wave; ++w) pos0 += wave_tot[w];
#pragma unroll
  for (int j = 0; j < STEPS; ++j) {
    const int i = base + j * 64 + lane;
    const unsigned long long m = __ballot(f[j] != 0);
    const int pos = pos0 + __popcll(m & ((2ull << lane) - 1ull));      // flags up to and including this row
    pos0 += __popcll(m);
    if (i >= n) continue;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    if (f[j]) {
      int b, x, y, z;
      coarse_coord(coords + 4 * (size_t)i, ts2, b, x, y, z);
      reinterpret_cast<int4*>(coords_out)[pos - 1] = make_int4(b, x, y, z);
    }
    const int cs = ((c.y >> sh) & 1) | (((c.z >> sh) & 1) << 1) | (((c.w >> sh) & 1) << 2);
    parent[i] = pos - 1;
    children[(size_t)(pos - 1) * 8 + cs] = i;
  }
}

// ---- Z-ordered rows, round 6: ALL coarser levels from the sorted level-0 rows in two launches.  A level-l voxel is a run of adjacent
// level-0 rows (k_count_levels counts them that way), and a level-l boundary is also a boundary of every finer level, so one pass over
// the level-0 rows knows, for every row, its voxel's index at each level: the number of level-l boundaries up to and including it,
// minus one.  k_levels_count leaves the three boundary counts of every 2048-row tile; k_levels_fill adds up the tiles in front of
// its own, and its rows write - at the level where they open a voxel - that voxel's coordinate, its parent link and its entry in the
// parent's child list.  The same arrays, bit for bit, as k_flag_sorted -> k_scan_partials -> k_scan_top -> k_compact_sorted per level
// (rounds 3-5: twelve short dependent launches - a single pair spent 110 us of its 1.55 ms in their launch gaps).
static_assert(EYOC_MAX_LEVELS == 4, "k_levels_*: three coarser levels");
constexpr int LV_BLOCK = 256, LV_STEPS = 8, LV_TILE = LV_BLOCK * LV_STEPS;     // wave w of a block: rows [512 w, 512 w + 512) of the tile, 64 per step
__device__ inline void level_flags(const int4 c, const int4 q, bool first, bool& f1, bool& f2, bool& f3) {
  const unsigned int d = (unsigned int)((c.y ^ q.y) | (c.z ^ q.z) | (c.w ^ q.w));
  const bool nb = first || c.x != q.x;
  f1 = nb || (d >> 1) != 0u; f2 = nb || (d >> 2) != 0u; f3 = nb || (d >> 3) != 0u;
}
__global__ __launch_bounds__(LV_BLOCK) void k_levels_count(const int32_t* __restrict__ coords, int n, int* __restrict__ part /* [tiles][4] */,
                                                           int* __restrict__ counters) {
  __shared__ int tot[LV_BLOCK / 64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = blockIdx.x * LV_TILE + wave * (LV_TILE / (LV_BLOCK / 64));
  int c1 = 0, c2 = 0, c3 = 0, bad = 0, dup = 0;
  constexpr int LIM = COORD_BIAS - 16;
#pragma unroll
  for (int j = 0; j < LV_STEPS; ++j) {
    const int i = base + j * 64 + lane;
    bool f1 = false, f2 = false, f3 = false, e = false, d = false;
    if (i < n) {
      const int4 c = reinterpret_cast<const int4*>(coords)[i];
      const int4 q = i > 0 ? reinterpret_cast<const int4*>(coords)[i - 1] : c;
      level_flags(c, q, i == 0, f1, f2, f3);
      e = c.x < 0 || c.x >= 1024 || c.y < -LIM || c.y >= LIM || c.z < -LIM || c.z >= LIM || c.w < -LIM || c.w >= LIM;   // (k_flag_sorted's check at level 1)
      d = i > 0 && c.x == q.x && c.y == q.y && c.z == q.z && c.w == q.w;
    }
    c1 += __popcll(__ballot(f1)); c2 += __popcll(__ballot(f2)); c3 += __popcll(__ballot(f3));
    bad += __popcll(__ballot(e)); dup += __popcll(__ballot(d));
  }
  if (lane == 0) {
    tot[wave][0] = c1; tot[wave][1] = c2; tot[wave][2] = c3;
    if (bad) atomicAdd(counters, bad);
    if (dup) atomicAdd(counters + 1, dup);
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    int v = 0;
    for (int w = 0; w < LV_BLOCK / 64; ++w) v += tot[w][threadIdx.x];
    part[4 * blockIdx.x + threadIdx.x] = v;
  }
}
struct LevelOut {
  int32_t* coords[3];     // level 1..3 coordinates
  int32_t* parent[3];     // parent[l]: level l -> l + 1
  int32_t* children[3];   // children[l]: level l + 1 -> its up to 8 rows of level l
};
__global__ __launch_bounds__(LV_BLOCK) void k_levels_fill(const int32_t* __restrict__ coords, int n, const int* __restrict__ part,
                                                          LevelOut o, int* __restrict__ totals /* counters + 3: rows of level 1..3 */) {
  constexpr int NWV = LV_BLOCK / 64;
  __shared__ int red[NWV][3];
  __shared__ int tot[NWV][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // boundaries in the tiles in front of this one
  int b1 = 0, b2 = 0, b3 = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += LV_BLOCK) { b1 += part[4 * t]; b2 += part[4 * t + 1]; b3 += part[4 * t + 2]; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { b1 += __shfl_xor(b1, d, 64); b2 += __shfl_xor(b2, d, 64); b3 += __shfl_xor(b3, d, 64); }
  if (lane == 0) { red[wave][0] = b1; red[wave][1] = b2; red[wave][2] = b3; }
  // this wave's rows: the three boundary masks of each step
  const int base = blockIdx.x * LV_TILE + wave * (LV_TILE / NWV);
  unsigned long long m1[LV_STEPS], m2[LV_STEPS], m3[LV_STEPS];
  int4 cs[LV_STEPS];
  int c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
  for (int j = 0; j < LV_STEPS; ++j) {
    const int i = base + j * 64 + lane;
    bool f1 = false, f2 = false, f3 = false;
    cs[j] = make_int4(0, 0, 0, 0);
    if (i < n) {
      cs[j] = reinterpret_cast<const int4*>(coords)[i];
      const int4 q = i > 0 ? reinterpret_cast<const int4*>(coords)[i - 1] : cs[j];
      level_flags(cs[j], q, i == 0, f1, f2, f3);
    }
    m1[j] = __ballot(f1); m2[j] = __ballot(f2); m3[j] = __ballot(f3);
    c1 += __popcll(m1[j]); c2 += __popcll(m2[j]); c3 += __popcll(m3[j]);
  }
  if (lane == 0) { tot[wave][0] = c1; tot[wave][1] = c2; tot[wave][2] = c3; }
  __syncthreads();
  int p1 = 0, p2 = 0, p3 = 0;
  for (int w = 0; w < NWV; ++w) { p1 += red[w][0]; p2 += red[w][1]; p3 += red[w][2]; }
  if (blockIdx.x + 1 == gridDim.x && threadIdx.x == 0) {            // the scans' own totals (eyoc_maps_build compares them with k_count_levels')
    int t1 = p1, t2 = p2, t3 = p3;
    for (int w = 0; w < NWV; ++w) { t1 += tot[w][0]; t2 += tot[w][1]; t3 += tot[w][2]; }
    totals[0] = t1; totals[1] = t2; totals[2] = t3;
  }
  for (int w = 0; w < wave; ++w) { p1 += tot[w][0]; p2 += tot[w][1]; p3 += tot[w][2]; }
  const unsigned long long upto = (2ull << lane) - 1ull;             // lanes 0 .. lane
#pragma unroll
  for (int j = 0; j < LV_STEPS; ++j) {
    const int i = base + j * 64 + lane;
    const int i1 = p1 + __popcll(m1[j] & upto) - 1, i2 = p2 + __popcll(m2[j] & upto) - 1, i3 = p3 + __popcll(m3[j] & upto) - 1;
    p1 += __popcll(m1[j]); p2 += __popcll(m2[j]); p3 += __popcll(m3[j]);
    if (i >= n) continue;
    const int4 c = cs[j];
    o.parent[0][i] = i1;
    o.children[0][(size_t)i1 * 8 + ((c.y & 1) | ((c.z & 1) << 1) | ((c.w & 1) << 2))] = i;
    if ((m1[j] >> lane) & 1ull) {
      reinterpret_cast<int4*>(o.coords[0])[i1] = make_int4(c.x, c.y & ~1, c.z & ~1, c.w & ~1);
      o.parent[1][i1] = i2;
      o.children[1][(size_t)i2 * 8 + (((c.y >> 1) & 1) | (((c.z >> 1) & 1) << 1) | (((c.w >> 1) & 1) << 2))] = i1;
    }
    if ((m2[j] >> lane) & 1ull) {
      reinterpret_cast<int4*>(o.coords[1])[i2] = make_int4(c.x, c.y & ~3, c.z & ~3, c.w & ~3);
      o.parent[2][i2] = i3;
      o.children[2][(size_t)i3 * 8 + (((c.y >> 2) & 1) | (((c.z >> 2) & 1) << 1) | (((c.w >> 2) & 1) << 2))] = i2;
    }
    if ((m3[j] >> lane) & 1ull) reinterpret_cast<int4*>(o.coords[2])[i3] = make_int4(c.x, c.y & ~7, c.z & ~7, c.w & ~7);
  }
}

// nbr[k][o] = row of table_in at c_out[o] + sign * off_k * step
__global__ void k_neighbours(const int32_t* __restrict__ coords_out, int n_out, HashTable tin, int step, int sign,
                             int32_t* __restrict__ nbr) {
  int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  const int4 c = reinterpret_cast<const int4*>(coords_out)[o];
  const int d = sign * step;
#pragma unroll 1
  for (int k = 0; k < 27; ++k) {
    const int dx = (k % 3 - 1) * d, dy = ((k / 3) % 3 - 1) * d, dz = (k / 9 - 1) * d;
    nbr[(size_t)k * n_out + o] = hash_lookup(tin, pack_key(c.x, c.y + dx, c.z + dy, c.w + dz));
  }
}

// sort key of a row's stride-1 neighbour pattern (see eyoc_maps::perm_s1); offsets enumerate x fastest, so
// k / 9 is the z layer and (k / 3) % 3 the y row.  bits 0-1: does the z layer below / above hold any neighbour;
// bits 2-3 (only when key_bits == 4): the y rows before / behind inside the own layer.  Coarse on purpose: finer
// keys (up to the full 27-bit pattern) pack the chunks better but scatter a tile's rows over the cloud, and the
// lost L2 locality of the gather costs more than the saved matrix work (measured on MI355X).
// Z-ordered levels sort their tiling orders inside WINDOWS of 2^wshift consecutive rows (window number above the
// pattern key): a window's rows and their neighbours stay L2-resident while its pattern runs are walked, instead of
// every run sweeping the whole level (the transposed convolution 1 -> 0 read 11 GB that way, 2.2x its gather bytes).
constexpr int KEY_PATTERN_BITS = 11, KEY_WINDOW_BITS = 8;
__device__ inline unsigned int window_bits(int row, int wshift) {
  return wshift < 0 ? 0u : (unsigned)min(row >> wshift, (1 << KEY_WINDOW_BITS) - 1) << KEY_PATTERN_BITS;
}
__global__ void k_pattern_key(const int32_t* __restrict__ nbr, int n, int key_bits, unsigned int* __restrict__ key,
                              int* __restrict__ row, unsigned int key_tag, int wshift) {
  int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  unsigned int mask = 0;
#pragma unroll 1
  for (int k = 0; k < 27; ++k) mask |= (nbr[(size_t)k * n + o] >= 0 ? 1u : 0u) << k;
  const unsigned int lo = mask & 0x1FFu, mid = (mask >> 9) & 0x1FFu, hi = mask >> 18;
  unsigned int kv = (lo ? 1u : 0u) | (hi ? 2u : 0u);
  if (key_bits == 4) kv |= ((mid & 7u) ? 4u : 0u) | ((mid >> 6) ? 8u : 0u);
  key[o] = key_tag | window_bits(o, wshift) | kv;
  row[o] = o;
}

// sort key of a coarse row for the strided convolution: which of its own 8 children exist
__global__ void k_children_key(const int32_t* __restrict__ children, int nc, unsigned int* __restrict__ key,
                               int* __restrict__ row, unsigned int key_tag, int wshift) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nc) return;
  const int4 lo = reinterpret_cast<const int4*>(children)[2 * (size_t)v], hi = reinterpret_cast<const int4*>(children)[2 * (size_t)v + 1];
  const unsigned int m = (lo.x >= 0 ? 1u : 0u) | (lo.y >= 0 ? 2u : 0u) | (lo.z >= 0 ? 4u : 0u) | (lo.w >= 0 ? 8u : 0u) |
                         (hi.x >= 0 ? 16u : 0u) | (hi.y >= 0 ? 32u : 0u) | (hi.z >= 0 ? 64u : 0u) | (hi.w >= 0 ? 128u : 0u);
  key[v] = key_tag | window_bits(v, wshift) | m;
  row[v] = v;
}

// Z-order key of a row: batch outermost, then the bits of x, y, z (biased to be non-negative) interleaved
__device__ inline unsigned long long spread18(unsigned int v) {
  unsigned long long x = v & 0x3FFFFu;
  x = (x | (x << 32)) & 0x001F00000000FFFFull;
  x = (x | (x << 16)) & 0x001F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}
// (coordinates biased by 2^kbits, kbits + 1 bits per axis, the batch index above them: 3 (kbits + 1) + bbits significant bits.  A row
// outside +-2^kbits gets a meaningless key - k_count_levels reports it and the build sorts again with kbits = 17.)
__global__ void k_morton_key(const int32_t* __restrict__ coords, int n, int kbits, unsigned long long* __restrict__ key, int* __restrict__ row) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(coords)[i];
  const int bias = 1 << kbits;
  const unsigned int m = (2u << kbits) - 1u;
  key[i] = ((unsigned long long)(unsigned)c.x << (3 * (kbits + 1))) | spread18((unsigned)(c.y + bias) & m) | (spread18((unsigned)(c.z + bias) & m) << 1) |
           (spread18((unsigned)(c.w + bias) & m) << 2);
  row[i] = i;
}
__global__ void k_gather_coords(const int32_t* __restrict__ coords, const int* __restrict__ perm, int n, int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) reinterpret_cast<int4*>(out)[i] = reinterpret_cast<const int4*>(coords)[perm[i]];
}

__global__ void k_iota(int32_t* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}

// voxel coordinates of raw points: floor(x / voxel) in fp32 (IEEE division, like torch / numpy on fp32 input)
__global__ void k_quantize(const float* __restrict__ xyz, int n, int stride, float voxel, int batch,
                           int32_t* __restrict__ coords) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = xyz + (size_t)i * stride;
  int4 c = make_int4(batch, (int)floorf(p[0] / voxel), (int)floorf(p[1] / voxel), (int)floorf(p[2] / voxel));
  reinterpret_cast<int4*>(coords)[i] = c;
}

// octree links for the rows of level l (tensor stride 1 << sh): slot[] / vals still describe where each
// fine row landed in the coarse table, and k_compact has re-labelled vals with coarse row indices
// Two rows landing in the same child slot of the same block have identical coordinates: `dup` (level 0 only) counts
// them - the duplicate check of the input needs no hash table of its own.
__global__ void k_children(const int* __restrict__ slot, const int* __restrict__ vals, const int32_t* __restrict__ coords,
                           int n, int sh, int32_t* __restrict__ parent, int32_t* __restrict__ children, int* __restrict__ dup) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(coords)[i];
  const int p = vals[slot[i]];
  const int cs = ((c.y >> sh) & 1) | (((c.z >> sh) & 1) << 1) | (((c.w >> sh) & 1) << 2);
  parent[i] = p;
  if (dup) {
    if (atomicCAS(&children[(size_t)p * 8 + cs], -1, i) != -1) atomicAdd(dup, 1);
  } else {
    children[(size_t)p * 8 + cs] = i;
  }
}

// stride-1 table and transposed (2ts -> ts) table of level l from the stride-1 table of level l+1 (derive.h has the geometry).
// S1: write the stride-1 table.  UP: 0 = no transposed table, 1 = the full [27][n] one (+ the tiling-order keys when asked for),
// 2 = the compact [8][n] one (derive.h: up8) - all a class-major record builder reads (spconv_upc.hip), 32 instead of 108 bytes per row.
template <bool S1, int UP>
__global__ void k_derive_fine(const int32_t* __restrict__ coords, int n, int sh, const int32_t* __restrict__ parent,
                              const int32_t* __restrict__ children, const int32_t* __restrict__ s1c, int nc,
                              int32_t* __restrict__ s1, int32_t* __restrict__ up, unsigned int* __restrict__ up_key,
                              int* __restrict__ up_row, unsigned int key_tag, int wshift) {
  int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  const int4 c = reinterpret_cast<const int4*>(coords)[o];
  const int b[3] = {(c.y >> sh) & 1, (c.z >> sh) & 1, (c.w >> sh) & 1};
  const int p = parent[o];
  int blk[8];   // a bit per axis: 0 = own coarse block, 1 = the neighbouring block on the side this voxel leans to
  derive_blocks(b, p, s1c, nc, blk);
  if constexpr (S1) {
    int v[27];
    derive_window(b, blk, children, v);
#pragma unroll
    for (int k = 0; k < 27; ++k) s1[(size_t)k * n + o] = v[k];
  }
  if constexpr (UP != 0) {
    int bl[8];
    derive_up_blocks(b, blk, bl);
    const int cls = b[0] | (b[1] << 1) | (b[2] << 2);
    if constexpr (UP == 1) {
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const int off[3] = {k % 3 - 1, (k / 3) % 3 - 1, k / 9 - 1};
        // transposed map: c_u - off*ts must be a coarse coordinate: even position needs off == 0,
        // odd position needs off == +1 (the parent) or off == -1 (the next block)
        bool up_ok = true;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) up_ok = up_ok && (b[ax] ? off[ax] != 0 : off[ax] == 0);
        up[(size_t)k * n + o] = up_ok ? bl[up8_slot_of_offset(k)] : -1;
      }
      if (up_key) {
        // pattern of the transposed map: parity class (which axes sit on an odd position) and which of the coarse
        // blocks that class can reach exist -> 11-bit sort key (UP_KEY_BITS)
        unsigned int present = 0;
#pragma unroll
        for (int a = 0; a < 8; ++a)
          if ((a & cls) == a && blk[a] >= 0) present |= 1u << a;
        up_key[o] = key_tag | window_bits(o, wshift) | ((unsigned)cls << 8) | present;
        up_row[o] = o;
      }
    } else {
#pragma unroll
      for (int m = 0; m < 8; ++m) up[(size_t)m * n + o] = (m & ~cls) == 0 ? bl[m] : -1;
    }
  }
}

// strided (ts -> 2ts) table: for coarse row v the fine row at c_v + off*ts:
//   off -1 -> coarse block at -1, child bit 1;  off 0 / +1 -> own block, bit 0 / 1.
__global__ void k_derive_down(int nc, const int32_t* __restrict__ children, const int32_t* __restrict__ s1c,
                              int32_t* __restrict__ down) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nc) return;
  int blk[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int kc = (1 - (a & 1)) + 3 * (1 - ((a >> 1) & 1)) + 9 * (1 - ((a >> 2) & 1));
    blk[a] = a == 0 ? v : s1c[(size_t)kc * nc + v];
  }
  // the 8 child records whole (32 bytes each) instead of 27 four-byte loads out of them; block and child slot of an offset are
  // compile-time constants here
  int rec[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    int4 lo = make_int4(-1, -1, -1, -1), hi = lo;
    if (blk[a] >= 0) {
      lo = reinterpret_cast<const int4*>(children)[2 * (size_t)blk[a]];
      hi = reinterpret_cast<const int4*>(children)[2 * (size_t)blk[a] + 1];
    }
    rec[a][0] = lo.x; rec[a][1] = lo.y; rec[a][2] = lo.z; rec[a][3] = lo.w; rec[a][4] = hi.x; rec[a][5] = hi.y; rec[a][6] = hi.z; rec[a][7] = hi.w;
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int off[3] = {k % 3 - 1, (k / 3) % 3 - 1, k / 9 - 1};
    int a = 0, cs = 0;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      a |= (off[ax] < 0 ? 1 : 0) << ax;
      cs |= (off[ax] != 0 ? 1 : 0) << ax;
    }
    down[(size_t)k * nc + v] = rec[a][cs];
  }
}

__global__ void k_count_valid(const int32_t* __restrict__ t, long long n, unsigned long long* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned int c = 0;
  for (; i < n; i += stride) c += t[i] >= 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

__global__ void k_count_region(const int32_t* __restrict__ coords, int n, HashTable t, int ks,
                               unsigned long long* __restrict__ out) {
  int o = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int c = 0;
  if (o < n) {
    const int4 p = reinterpret_cast<const int4*>(coords)[o];
    const int r = ks / 2;
    for (int dz = -r; dz <= r; ++dz)
      for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) c += hash_lookup(t, pack_key(p.x, p.y + dx, p.z + dy, p.w + dz)) >= 0;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

constexpr int UP_KEY_BITS = KEY_PATTERN_BITS + KEY_WINDOW_BITS;
// Z-ordered levels: window of the tiling orders, log2 rows (eyoc_ctx::Knobs::maps_window_shift).  Measured on the 64-pair bench:
// 2^17-2^18 rows (2^12, 2^14 lose - too many short pattern runs; no windows: +23 % on the 1 -> 0 transposed convolution)
// Z-order (and with it the staged split16 kernels, model.hip) from 8192 rows: measured faster than the fp32 kernels on
// the caller's row order at every size tried - maps + forward of a 15 k-row half cloud 2.30 -> 1.78 ms, one 31 k-row
// cloud 2.45 -> 2.02, a pair 2.86 -> 2.17 (scripts/bench_small_forward.py).  Smaller inputs keep the caller's order.
constexpr int ZORDER_MIN_ROWS = 8192;

unsigned int table_capacity(int n) {
  unsigned int cap = 1024;
  while (cap < 2u * (unsigned)n) cap <<= 1;
  return cap;
}

}  // namespace

extern "C" {

size_t eyoc_maps_workspace_bytes(int n_rows) {
  if (n_rows < 0) return 0;
  const size_t n = (size_t)n_rows;
  const size_t cap = table_capacity(n_rows);
  size_t b = 0;
  b += EYOC_MAX_LEVELS * (align_up(cap * 8) + align_up(cap * 4));   // hash tables
  b += EYOC_MAX_LEVELS * align_up(n * 16);                           // coordinates
  b += 10 * align_up(n * 27 * 4);                                    // 4 s1 + 3 down + 3 up tables
  b += 3 * align_up(n * 8 * 4);                                      // compact transposed tables (lazy tables: derive.h up8)
  b += 3 * align_up(n * 4) + align_up((n / SCAN_TILE + 2) * 4);      // slot, flag, partial sums
  b += 3 * (align_up(n * 4) + align_up(n * 32));                     // parent / children links
  b += 4 * align_up(4 * n * 4) + align_up(sort_rows_tmp_bytes(4 * n_rows, UP_KEY_BITS + 4));   // tiling orders: keys, rows, result (<= 2 segments per level, levels sum to < 2 n)
  b += align_up(n * 8) * 2 + align_up(n * 4) * 2 + align_up(sort_rows64_tmp_bytes(n_rows));   // Z-order: keys in / out, rows in, permutation
  // local rulebooks of the stride-1 tables (levels sum to < 2 n rows; every level rounds up to a whole tile)
  b += 2 * align_up(local_rulebook_bytes(n_rows)) + EYOC_MAX_LEVELS * (align_up(local_rulebook_bytes(1)) + 256);   // stride-1 tables
  b += 2 * align_up(local_rulebook_up_bytes(n_rows)) + EYOC_MAX_LEVELS * (align_up(local_rulebook_up_bytes(1)) + 256);
  b += 2 * (align_up(upc_kept_bytes(n_rows)) + align_up(upc_scratch_bytes(n_rows))) + EYOC_MAX_LEVELS * (align_up(upc_kept_bytes(1)) + align_up(upc_scratch_bytes(1)) + 512);   // class-major transposed records + their builder's scratch
  b += 4 * (align_up(local_rulebook128_bytes(n_rows)) + 256);        // strided tables (3) and the coarsest stride-1 table in 128-row tiles (a level has at most n rows)
  b += 4096;                                                         // counters
  return b + 96 * 256;
}

int eyoc_maps_build(eyoc_ctx* ctx, const int32_t* coords_dev, int n, void* ws, size_t ws_bytes, void* stream,
                    eyoc_maps** out) {
  return eyoc_maps_build_ordered(ctx, coords_dev, n, ws, ws_bytes, stream, 0, out);   // the caller's row order: what the accessors promise
}

int eyoc_maps_build_ordered(eyoc_ctx* ctx, const int32_t* coords_dev, int n, void* ws, size_t ws_bytes, void* stream,
                            int order, eyoc_maps** out) {
  EYOC_REQUIRE(ctx && out && ws, EYOC_ERR_INVALID, "eyoc_maps_build: NULL argument");
  EYOC_REQUIRE(order >= -1 && order <= 1, EYOC_ERR_INVALID, "eyoc_maps_build_ordered: order %d", order);
  EYOC_REQUIRE(n > 0 && coords_dev, EYOC_ERR_INVALID, "eyoc_maps_build: empty coordinate set (n=%d)", n);
  EYOC_REQUIRE(((uintptr_t)ws & 255) == 0, EYOC_ERR_INVALID, "eyoc_maps_build: workspace must be 256-byte aligned");
  EYOC_REQUIRE(ws_bytes >= eyoc_maps_workspace_bytes(n), EYOC_ERR_WORKSPACE,
               "eyoc_maps_build: workspace %zu < required %zu bytes", ws_bytes, eyoc_maps_workspace_bytes(n));
  hipStream_t st = (hipStream_t)stream;
  Carver cv(ws, ws_bytes);
  eyoc_maps* m = new eyoc_maps();
  m->n_levels = EYOC_MAX_LEVELS;
  int* counters = cv.take<int>(64);       // [0] range errors, [1] duplicates, [2+l] coarse totals
  int* slot = cv.take<int>(n);
  int* flag = cv.take<int>(n);
  int* partial = cv.take<int>(n / SCAN_TILE + 2);
  int* host = (int*)ctx->pinned;
#define FAIL_HIP(expr)                                                                       \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
      delete m;                                                                              \
      return EYOC_ERR_HIP;                                                                   \
    }                                                                                        \
  } while (0)

  FAIL_HIP(hipMemsetAsync(counters, 0, 64 * sizeof(int), st));
  // ---- level 0: the caller's rows, in the caller's order
  m->rows[0] = n;
  m->coords[0] = cv.take<int32_t>((size_t)n * 4);
  const eyoc_ctx::Knobs& kn = ctx->knobs;
  const int order_mode = kn.maps_internal_order >= 0 ? kn.maps_internal_order : order;   // the ctx's switch (tests) beats the call's wish
  const bool zorder = order_mode > 0 || (order_mode < 0 && n >= ZORDER_MIN_ROWS);
  int pre_rows[EYOC_MAX_LEVELS] = {n, 0, 0, 0};                          // Z-order: known before the levels are built
  if (zorder) {
    unsigned long long* zk_in = cv.take<unsigned long long>(n);
    unsigned long long* zk_out = cv.take<unsigned long long>(n);
    int* zr_in = cv.take<int>(n);
    m->row_perm = cv.take<int32_t>(n);
    const size_t zb = sort_rows64_tmp_bytes(n);
    void* ztmp = cv.take<char>(zb);
    // The key width is speculated from the ctx's previous build (KITTI-shaped clouds: 10 + 10 + 10 + 7 bits = five 8-bit radix
    // passes instead of eight); a batch that does not fit sorts once more with the full width.
    int kbits = ctx->zorder_kbits, bbits = ctx->zorder_bbits;
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (attempt) FAIL_HIP(hipMemsetAsync(counters + 16, 0, (EYOC_MAX_LEVELS + 4) * sizeof(int), st));
      hipLaunchKernelGGL(k_morton_key, dim3(cdiv(n, 256)), dim3(256), 0, st, coords_dev, n, kbits, zk_in, zr_in);
      if (int rc = sort_rows_by_key64(ztmp, zb, zk_in, zk_out, zr_in, m->row_perm, n, 3 * (kbits + 1) + bbits, st)) { delete m; return rc; }
      hipLaunchKernelGGL(k_gather_coords, dim3(cdiv(n, 256)), dim3(256), 0, st, coords_dev, m->row_perm, n, m->coords[0]);
      // every level's row count - and the validation - now, with one read-back
      hipLaunchKernelGGL(k_count_levels, dim3(cdiv(n, 256) < 1024 ? cdiv(n, 256) : 1024), dim3(256), 0, st, m->coords[0], n, kbits, bbits,
                         counters + 16);
      FAIL_HIP(hipMemcpyAsync(host + 16, counters + 16, (EYOC_MAX_LEVELS + 4) * sizeof(int), hipMemcpyDeviceToHost, st));
      FAIL_HIP(hipStreamSynchronize(st));
      if (host[16] == 0) break;                                         // every row fitted the speculated width
      kbits = 17; bbits = 10;
    }
    {
      // next build: one bit of head room over what this one needed
      int kb = 1, bb = 1;
      while (kb < 17 && (1 << kb) <= host[16 + EYOC_MAX_LEVELS + 2]) ++kb;
      while (bb < 10 && (1 << bb) <= host[16 + EYOC_MAX_LEVELS + 3]) ++bb;
      ctx->zorder_kbits = kb + 1 < 17 ? kb + 1 : 17;
      ctx->zorder_bbits = bb + 1 < 10 ? bb + 1 : 10;
    }
    if (host[16 + EYOC_MAX_LEVELS] != 0) {
      set_error("eyoc_maps_build: %d coordinate rows outside the supported key range (|c| < 2^17 - 16, 0 <= batch < 1024)",
                host[16 + EYOC_MAX_LEVELS]);
      delete m;
      return EYOC_ERR_RANGE;
    }
    if (host[16 + EYOC_MAX_LEVELS + 1] != 0) {
      set_error("eyoc_maps_build: %d duplicate coordinate rows (a sparse tensor needs unique coordinates)", host[16 + EYOC_MAX_LEVELS + 1]);
      delete m;
      return EYOC_ERR_DUPLICATE;
    }
    for (int l = 1; l < EYOC_MAX_LEVELS; ++l) pre_rows[l] = host[16 + l];
  } else {
    FAIL_HIP(hipMemcpyAsync(m->coords[0], coords_dev, (size_t)n * 16, hipMemcpyDeviceToDevice, st));
  }
  // Level 0 needs no hash table: validation rides on the level-1 build (range check in its k_insert, duplicate
  // rows = two rows in one child slot in k_children); the first convolution walks the octree (spconv.hip).  The
  // table's memory stays reserved so that maps_build_table0 can build it for the hash-probing fallback.
  {
    const unsigned int cap = table_capacity(n);
    m->table[0].keys = cv.take<unsigned long long>(cap);
    m->table[0].vals = cv.take<int>(cap);
    m->table[0].mask = cap - 1;
    m->table0_built = false;
  }
  // Z-ordered rows: every level's row count is known by now, so the links' and the coarsest table's fills go out up front - between the
  // short dependent kernels of a level each costs a launch gap (a single pair: five fills, ~8 us apiece on the critical path)
  unsigned int top_cap = 0;
  if (zorder) {
    for (int l = 1; l < EYOC_MAX_LEVELS; ++l) m->children[l - 1] = cv.take<int32_t>((size_t)pre_rows[l] * 8);
    HashTable& tt = m->table[EYOC_MAX_LEVELS - 1];
    top_cap = table_capacity(pre_rows[EYOC_MAX_LEVELS - 2]);
    tt.keys = cv.take<unsigned long long>(top_cap);
    tt.vals = cv.take<int>(top_cap);
    tt.mask = top_cap - 1;
    // the three link arrays and the table's keys are carved back to back and all start as 0xFF bytes: ONE fill (the alignment gaps
    // between them belong to nobody) instead of four - a fill is a ~5 us launch on the build's critical path
    FAIL_HIP(hipMemsetAsync(m->children[0], 0xFF, (size_t)((char*)(tt.keys + top_cap) - (char*)m->children[0]), st));
    FAIL_HIP(hipMemsetAsync(tt.vals, 0x7F, (size_t)top_cap * 4, st));
  }
  const bool fused_levels = zorder && kn.maps_fused_levels;
  if (fused_levels) {
    // all three coarser levels in two launches (k_levels_count / k_levels_fill); carved in the order the level loop below carves them
    LevelOut lo;
    for (int l = 1; l < EYOC_MAX_LEVELS; ++l) {
      m->rows[l] = pre_rows[l];
      m->coords[l] = cv.take<int32_t>((size_t)m->rows[l] * 4);
      m->parent[l - 1] = cv.take<int32_t>((size_t)m->rows[l - 1]);
      lo.coords[l - 1] = m->coords[l]; lo.parent[l - 1] = m->parent[l - 1]; lo.children[l - 1] = m->children[l - 1];
    }
    const int tiles = cdiv(n, LV_TILE);
    int* part = flag;                                                  // [tiles][4] <= n ints: the flag array is free on this path
    hipLaunchKernelGGL(k_levels_count, dim3(tiles), dim3(LV_BLOCK), 0, st, m->coords[0], n, part, counters);
    hipLaunchKernelGGL(k_levels_fill, dim3(tiles), dim3(LV_BLOCK), 0, st, m->coords[0], n, part, lo, counters + 3);
    const int lt = EYOC_MAX_LEVELS - 1;
    hipLaunchKernelGGL(k_insert, dim3(cdiv(m->rows[lt], 256)), dim3(256), 0, st, m->coords[lt], m->rows[lt], 1 << lt, m->table[lt], (int*)nullptr, counters);
  }
  for (int l = 1; l < EYOC_MAX_LEVELS && !fused_levels; ++l) {
    // table of THIS level is built from the previous level's rows
    const int n_src = m->rows[l - 1];
    const int32_t* src = m->coords[l - 1];
    const int ts2 = 1 << l;
    const int nb = cdiv(n_src, SCAN_TILE);
    const bool sorted_level = zorder;                                  // adjacent rows (see k_flag_sorted); only the coarsest level gets a table, below
    HashTable& t = m->table[l];
    if (sorted_level) {
      hipLaunchKernelGGL(k_flag_sorted, dim3(cdiv(n_src, 256)), dim3(256), 0, st, src, n_src, ts2, flag,
                         l == 1 ? counters : (int*)nullptr, l == 1 ? counters + 1 : (int*)nullptr);
    } else {
      const unsigned int cap = table_capacity(n_src);
      t.keys = cv.take<unsigned long long>(cap);
      t.vals = cv.take<int>(cap);
      t.mask = cap - 1;
      FAIL_HIP(hipMemsetAsync(t.keys, 0xFF, (size_t)cap * 8, st));
      FAIL_HIP(hipMemsetAsync(t.vals, 0x7F, (size_t)cap * 4, st));
      hipLaunchKernelGGL(k_insert, dim3(cdiv(n_src, 256)), dim3(256), 0, st, src, n_src, ts2, t, slot, counters);
      hipLaunchKernelGGL(k_flag, dim3(cdiv(n_src, 256)), dim3(256), 0, st, slot, t.vals, n_src, flag, (int*)nullptr);
    }
    hipLaunchKernelGGL(k_scan_partials, dim3(nb), dim3(SCAN_BLOCK), 0, st, flag, n_src, partial);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_BLOCK), 0, st, partial, nb, counters + 2 + l);
    if (zorder) {
      m->rows[l] = pre_rows[l];
    } else {
    FAIL_HIP(hipMemcpyAsync(host, counters + 2 + l, sizeof(int), hipMemcpyDeviceToHost, st));
    if (l == 1) FAIL_HIP(hipMemcpyAsync(host + 1, counters, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    FAIL_HIP(hipStreamSynchronize(st));
    if (l == 1 && host[1] != 0) {
      set_error("eyoc_maps_build: %d coordinate rows outside the supported key range (|c| < 2^17 - 16, 0 <= batch < 1024)", host[1]);
      delete m;
      return EYOC_ERR_RANGE;
    }
    if (l == 1 && sorted_level && host[2] != 0) {
      set_error("eyoc_maps_build: %d duplicate coordinate rows (a sparse tensor needs unique coordinates)", host[2]);
      delete m;
      return EYOC_ERR_DUPLICATE;
    }
    m->rows[l] = host[0];
    }
    m->coords[l] = cv.take<int32_t>((size_t)m->rows[l] * 4);
    m->parent[l - 1] = cv.take<int32_t>((size_t)n_src);
    if (!zorder) {
      m->children[l - 1] = cv.take<int32_t>((size_t)m->rows[l] * 8);
      FAIL_HIP(hipMemsetAsync(m->children[l - 1], 0xFF, (size_t)m->rows[l] * 32, st));
    }
    if (sorted_level) {
      hipLaunchKernelGGL(k_compact_sorted, dim3(nb), dim3(SCAN_BLOCK), 0, st, flag, partial, src, n_src, ts2, l - 1,
                         m->coords[l], m->parent[l - 1], m->children[l - 1]);
      if (l + 1 == EYOC_MAX_LEVELS) {
        // the coarsest level's table (k_neighbours probes it): its UNIQUE rows are inserted - one uncontended CAS per voxel
        // instead of one per finer row, four neighbouring lanes fighting over every slot (118 -> ~20 us on the bench batch)
        // (table allocated and filled up front, above)
        hipLaunchKernelGGL(k_insert, dim3(cdiv(m->rows[l], 256)), dim3(256), 0, st, m->coords[l], m->rows[l], ts2, t, (int*)nullptr, counters);
      }
      continue;
    }
    hipLaunchKernelGGL(k_compact, dim3(nb), dim3(SCAN_BLOCK), 0, st, flag, partial, slot, src, n_src, ts2,
                       m->coords[l], t.vals, (int32_t*)nullptr);
    // octree links fine (l-1) <-> coarse (l); same stream, so k_compact's re-labelling is visible
    hipLaunchKernelGGL(k_children, dim3(cdiv(n_src, 256)), dim3(256), 0, st, slot, t.vals, src, n_src, l - 1,
                       m->parent[l - 1], m->children[l - 1], l == 1 ? counters + 1 : (int*)nullptr);
    if (l == 1) {
      FAIL_HIP(hipMemcpyAsync(host, counters + 1, sizeof(int), hipMemcpyDeviceToHost, st));
      FAIL_HIP(hipStreamSynchronize(st));
      if (host[0] != 0) {
        set_error("eyoc_maps_build: %d duplicate coordinate rows (a sparse tensor needs unique coordinates)", host[0]);
        delete m;
        return EYOC_ERR_DUPLICATE;
      }
    }
  }
  // ---- rulebooks: hash probes at the coarsest level only, everything else derived top-down
  // transposed tables on Z-ordered maps: class-major tiles (spconv_upc.hip) for batches, Morton tiles (spconv_up.hip) below
  // UPC_MIN_ROWS voxels - the partition's five small launches per level cost a single pair (60 k voxels) more than its kernel saves
  // (eyoc_spconv_upc_min_rows, default 2^17)
  const bool use_upc = kn.up_kernel == 2 && n >= kn.upc_min_rows;
  const bool use_up = kn.up_kernel == 1 || (kn.up_kernel == 2 && !use_upc);
  // lazy tables (common.h eyoc_maps): the level-0 stride-1 table and the transposed tables are read by their record builders only
  const bool lazy = zorder && use_upc && kn.maps_lazy_tables != 0;
  for (int l = 0; l < EYOC_MAX_LEVELS; ++l) {
    m->nbr_s1[l] = cv.take<int32_t>((size_t)27 * m->rows[l]);
    if (l + 1 < EYOC_MAX_LEVELS) {
      m->nbr_down[l] = cv.take<int32_t>((size_t)27 * m->rows[l + 1]);
      m->nbr_up[l] = cv.take<int32_t>((size_t)27 * m->rows[l]);
      if (lazy) {
        m->up8[l] = cv.take<int32_t>((size_t)8 * m->rows[l]);
        m->up_ready[l] = false;
      }
    }
  }
  if (lazy) m->s1_ready[0] = false;
  {
    const int top = EYOC_MAX_LEVELS - 1;
    hipLaunchKernelGGL(k_neighbours, dim3(cdiv(m->rows[top], 256)), dim3(256), 0, st, m->coords[top], m->rows[top],
                       m->table[top], 1 << top, 1, m->nbr_s1[top]);
  }
  // ---- tiling orders (perm_up / perm_s1): every ordered table contributes a segment of (key, row) pairs tagged with
  // its segment number in the high key bits, and ONE stable radix sort orders all segments at once (seven separate
  // rocPRIM sorts cost 38 small launches, 0.7 ms per 64-cloud batch).  The orders only serve the wave-private
  // convolution kernel, which takes over above ~4000 row tiles: small levels (single-pair latency path) skip them.
  const bool s1_order = kn.maps_s1_order != 0;
  int seg_up[EYOC_MAX_LEVELS], seg_s1[EYOC_MAX_LEVELS], seg_dn[EYOC_MAX_LEVELS], seg_base[3 * EYOC_MAX_LEVELS], n_seg = 0;
  size_t total = 0;
  for (int l = 0; l < EYOC_MAX_LEVELS; ++l) {
    seg_up[l] = seg_s1[l] = seg_dn[l] = -1;
    // Z-ordered maps tile the strided convolutions in natural order: a tile's 64 coarse rows read their (adjacent)
    // children, which beats the pattern order's fuller chunks (2.38 -> 2.25 ms for the three layers;
    // eyoc_maps_select_orders(-1, 1) restores the sort)
    const bool dn_order = kn.maps_down_order == 1;
    if (l + 1 < EYOC_MAX_LEVELS && m->rows[l + 1] >= kn.maps_order_min_rows && (dn_order || !zorder)) {   // outputs of the strided conv l -> l+1
      seg_dn[l] = n_seg; seg_base[n_seg++] = (int)total; total += m->rows[l + 1];
    }
    if (m->rows[l] < kn.maps_order_min_rows) continue;
    // the transposed tables' order serves the gathering kernels only: Z-ordered maps with the staged transposed kernel (which sorts
    // inside its tiles) skip it - and with it the whole radix sort, nothing else being ordered there (0.3 ms per 128-cloud batch)
    if (l + 1 < EYOC_MAX_LEVELS && !(zorder && (use_up || use_upc))) { seg_up[l] = n_seg; seg_base[n_seg++] = (int)total; total += m->rows[l]; }
    if (s1_order && !zorder) { seg_s1[l] = n_seg; seg_base[n_seg++] = (int)total; total += m->rows[l]; }
  }
  constexpr int TAG_SHIFT = UP_KEY_BITS, TAG_BITS = 4;   // pattern keys < 2^11, window number, segment tag above (at most 10 segments)
  const int wshift = zorder ? kn.maps_window_shift : -1;
  unsigned int* key_in = cv.take<unsigned int>(total);
  unsigned int* key_out = cv.take<unsigned int>(total);
  int* row_in = cv.take<int>(total);
  int32_t* perm_all = cv.take<int32_t>(total);
  const size_t sort_bytes = sort_rows_tmp_bytes((int)total, TAG_SHIFT + TAG_BITS);
  void* sort_tmp = cv.take<char>(sort_bytes);
  for (int l = EYOC_MAX_LEVELS - 2; l >= 0; --l) {
    const int nl = m->rows[l], nc = m->rows[l + 1];
    const int su = seg_up[l];
    if (lazy) {
      // levels >= 1: the stride-1 table (the next finer level derives from it) + the compact transposed table; level 0: nothing here -
      // its tile-record builder derives the windows itself and writes up8[0] on the way (build_local_rulebook_derived, below)
      if (l > 0)
        hipLaunchKernelGGL((k_derive_fine<true, 2>), dim3(cdiv(nl, 256)), dim3(256), 0, st, m->coords[l], nl, l, m->parent[l], m->children[l],
                           m->nbr_s1[l + 1], nc, m->nbr_s1[l], m->up8[l], (unsigned int*)nullptr, (int*)nullptr, 0u, wshift);
    } else {
      hipLaunchKernelGGL((k_derive_fine<true, 1>), dim3(cdiv(nl, 256)), dim3(256), 0, st, m->coords[l], nl, l, m->parent[l],
                         m->children[l], m->nbr_s1[l + 1], nc, m->nbr_s1[l], m->nbr_up[l],
                         su >= 0 ? key_in + seg_base[su] : (unsigned int*)nullptr, su >= 0 ? row_in + seg_base[su] : (int*)nullptr,
                         (unsigned int)(su >= 0 ? su : 0) << TAG_SHIFT, wshift);
    }
    hipLaunchKernelGGL(k_derive_down, dim3(cdiv(nc, 256)), dim3(256), 0, st, nc, m->children[l], m->nbr_s1[l + 1],
                       m->nbr_down[l]);
  }
  for (int l = 0; l + 1 < EYOC_MAX_LEVELS; ++l) {
    const int sd = seg_dn[l];
    if (sd < 0) continue;
    hipLaunchKernelGGL(k_children_key, dim3(cdiv(m->rows[l + 1], 256)), dim3(256), 0, st, m->children[l], m->rows[l + 1],
                       key_in + seg_base[sd], row_in + seg_base[sd], (unsigned int)sd << TAG_SHIFT, wshift);
  }
  for (int l = 0; l < EYOC_MAX_LEVELS; ++l) {
    const int ss = seg_s1[l];
    if (ss < 0) continue;
    const int bits = l == 0 ? 2 : 4;   // level 0 feeds the 32-channel, bandwidth-bound layers: keep more locality
    hipLaunchKernelGGL(k_pattern_key, dim3(cdiv(m->rows[l], 256)), dim3(256), 0, st, m->nbr_s1[l], m->rows[l], bits,
                       key_in + seg_base[ss], row_in + seg_base[ss], (unsigned int)ss << TAG_SHIFT, wshift);
  }
  if (total > 0) {
    if (int rc = sort_rows_by_key(sort_tmp, sort_bytes, key_in, key_out, row_in, perm_all, (int)total, TAG_SHIFT + TAG_BITS, st)) {
      delete m;
      return rc;
    }
    for (int l = 0; l < EYOC_MAX_LEVELS; ++l) {
      if (seg_up[l] >= 0) m->perm_up[l] = perm_all + seg_base[seg_up[l]];
      if (seg_s1[l] >= 0) m->perm_s1[l] = perm_all + seg_base[seg_s1[l]];
      if (seg_dn[l] >= 0) m->perm_down[l] = perm_all + seg_base[seg_dn[l]];
    }
  }
  // ---- local rulebooks of the stride-1 tables (tile-local input stage of the sparse convolution): only for Z-ordered rows
  if (zorder) {
    // counters [8] stride-1, [9] transposed, [10 + l] strided table l, [13] wide level-3 tiles: zero since the fill of all 64 words at the top (nothing in between writes them)
    for (int l = 0; l < EYOC_MAX_LEVELS; ++l) {
      m->local_s1[l] = cv.take<unsigned char>(local_rulebook_bytes(m->rows[l]));
      if (lazy && l == 0) {
        DeriveSrc src;
        src.coords = m->coords[0]; src.parent = m->parent[0]; src.children = m->children[0]; src.s1c = m->nbr_s1[1];
        src.nc = m->rows[1]; src.sh = 0; src.up8 = m->up8[0];
        if (int rc = build_local_rulebook_derived(src, m->rows[0], m->local_s1[0], counters + 8, st, kn.st_group)) { delete m; return rc; }
      } else if (int rc = build_local_rulebook(m->nbr_s1[l], 27, m->rows[l], m->local_s1[l], counters + 8, st, kn.st_group)) { delete m; return rc; }
      if (l + 1 < EYOC_MAX_LEVELS && use_up) {   // the transposed table whose outputs are this level's rows
        m->local_up[l] = cv.take<unsigned char>(local_rulebook_up_bytes(m->rows[l]));
        if (int rc = build_local_rulebook_up(m->nbr_up[l], 27, m->rows[l], m->local_up[l], counters + 9, st)) { delete m; return rc; }
      }
      if (l + 1 < EYOC_MAX_LEVELS && use_upc) {   // ... in class-major order (spconv_upc.hip)
        m->local_upc[l] = cv.take<unsigned char>(upc_kept_bytes(m->rows[l]));
        unsigned char* scratch = cv.take<unsigned char>(upc_scratch_bytes(m->rows[l]));
        if (m->local_upc[l]) {
          if (int rc = build_upc(lazy ? m->up8[l] : m->nbr_up[l], m->coords[l], 1 << l, m->rows[l], m->local_upc[l], scratch, st, lazy)) { delete m; return rc; }
        } else if (int rc = maps_ensure_table(m, EYOC_MAP_UP, l, st)) { delete m; return rc; }
      }
    }
    // strided tables (outputs = the rows of level l + 1) in 128-row tiles: batches only - a single pair's 97 level-1 tiles do not fill the chip
    const bool use_down = kn.down_kernel == 1 && n >= kn.upc_min_rows;
    // the coarsest level's stride-1 table in 128-row tiles too: its layers have 256 channels, and a workgroup of 128 rows x 128 channels
    // stages a tile's rows once per 128 output channels instead of once per 64 (block4: 0.89 -> 0.78 ms per layer on the bench batch; the
    // 128-channel layers of level 2 measured level - 0.84 -> 0.83 - and keep the 256-row tiles)
    for (int l = EYOC_MAX_LEVELS - 1; l < EYOC_MAX_LEVELS && use_down && kn.s1_wide; ++l) {
      m->local_s1w[l] = cv.take<unsigned char>(local_rulebook128_bytes(m->rows[l]));
      if (m->local_s1w[l])
        if (int rc = build_local_rulebook128(m->nbr_s1[l], 27, m->rows[l], m->local_s1w[l], counters + 13, st, kn.st_group)) { delete m; return rc; }
    }
    for (int l = 0; l + 1 < EYOC_MAX_LEVELS && use_down; ++l) {
      m->local_down[l] = cv.take<unsigned char>(local_rulebook128_bytes(m->rows[l + 1]));
      if (m->local_down[l])
        if (int rc = build_local_rulebook128(m->nbr_down[l], 27, m->rows[l + 1], m->local_down[l], counters + 10 + l, st, kn.st_group)) { delete m; return rc; }
    }
    for (int l = 0; l < EYOC_MAX_LEVELS; ++l) {
      host[16 + l] = 0;
      if (m->local_upc[l]) FAIL_HIP(hipMemcpyAsync(host + 16 + l, upc_overflow_ptr(m->local_upc[l]), sizeof(int), hipMemcpyDeviceToHost, st));
    }
    FAIL_HIP(hipMemcpyAsync(host, counters + 8, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
    FAIL_HIP(hipMemcpyAsync(host + 32, counters, 8 * sizeof(int), hipMemcpyDeviceToHost, st));   // errors + the scans' own totals
    FAIL_HIP(hipStreamSynchronize(st));
    if (host[32] != 0) {
      set_error("eyoc_maps_build: %d coordinate rows outside the supported key range (|c| < 2^17 - 16, 0 <= batch < 1024)", host[32]);
      delete m;
      return EYOC_ERR_RANGE;
    }
    if (host[33] != 0) {
      set_error("eyoc_maps_build: %d duplicate coordinate rows (a sparse tensor needs unique coordinates)", host[33]);
      delete m;
      return EYOC_ERR_DUPLICATE;
    }
    for (int l = 1; l < EYOC_MAX_LEVELS; ++l)
      if (host[32 + 2 + l] != m->rows[l]) {
        set_error("eyoc_maps_build: internal error - level %d has %d rows by its scan, %d by the up-front count", l, host[32 + 2 + l], m->rows[l]);
        delete m;
        return EYOC_ERR_INVALID;
      }
    if (host[0] != 0) {   // a tile with more distinct input rows than two passes stage (does not happen for Z-ordered rows): no staged kernel
      for (int l = 0; l < EYOC_MAX_LEVELS; ++l) m->local_s1[l] = nullptr;
      if (int rc = maps_ensure_table(m, EYOC_MAP_S1, 0, st)) { delete m; return rc; }   // the gathering kernels read the table
    }
    if (host[1] != 0)   // ... more than 639 distinct coarse rows under a 256-row tile
      for (int l = 0; l < EYOC_MAX_LEVELS; ++l) m->local_up[l] = nullptr;
    for (int l = 0; l + 1 < EYOC_MAX_LEVELS; ++l)   // a 128-row coarse tile with more than 1278 distinct fine rows: that strided table stays on the gathering kernel
      if (host[2 + l] != 0) m->local_down[l] = nullptr;
    if (host[5] != 0)
      for (int l = 0; l < EYOC_MAX_LEVELS; ++l) m->local_s1w[l] = nullptr;
    for (int l = 0; l < EYOC_MAX_LEVELS; ++l)   // a class tile with more than 1278 distinct coarse rows: that table stays on the gathering kernels
      if (host[16 + l] != 0) {
        m->local_upc[l] = nullptr;
        if (int rc = maps_ensure_table(m, EYOC_MAP_UP, l, st)) { delete m; return rc; }
      }
  }
  FAIL_HIP(hipGetLastError());
#undef FAIL_HIP
  if (!cv.ok()) {
    set_error("eyoc_maps_build: internal workspace accounting error (%zu > %zu)", cv.off, cv.cap);
    delete m;
    return EYOC_ERR_WORKSPACE;
  }
  *out = m;
  return EYOC_OK;
}

// ---- voxeliser: first point of every occupied voxel, in input order
size_t eyoc_voxelize_workspace_bytes(int n_points) {
  if (n_points < 0) return 0;
  const size_t n = (size_t)n_points, cap = table_capacity(n_points);
  return align_up(cap * 8) + align_up(cap * 4) + align_up(n * 16) + 3 * align_up(n * 4) +
         align_up((n / SCAN_TILE + 2) * 4) + 4096 + 16 * 256;
}

int eyoc_voxelize(eyoc_ctx* ctx, const float* xyz_dev, int n, int stride, float voxel_size, int batch_index,
                  int32_t* sel_dev, int32_t* coords_dev, int* n_out, void* ws, size_t ws_bytes, void* stream) {
  EYOC_REQUIRE(ctx && xyz_dev && sel_dev && coords_dev && n_out && ws, EYOC_ERR_INVALID, "eyoc_voxelize: NULL argument");
  EYOC_REQUIRE(n > 0 && stride >= 3 && voxel_size > 0.0f, EYOC_ERR_INVALID, "eyoc_voxelize: n %d stride %d voxel %g", n,
               stride, voxel_size);
  EYOC_REQUIRE(batch_index >= 0 && batch_index < 1024, EYOC_ERR_RANGE, "eyoc_voxelize: batch index %d", batch_index);
  EYOC_REQUIRE(((uintptr_t)ws & 255) == 0 && ws_bytes >= eyoc_voxelize_workspace_bytes(n), EYOC_ERR_WORKSPACE,
               "eyoc_voxelize: workspace %zu < required %zu bytes (256-byte aligned)", ws_bytes,
               eyoc_voxelize_workspace_bytes(n));
  hipStream_t st = (hipStream_t)stream;
  Carver cv(ws, ws_bytes);
  int* counters = cv.take<int>(64);
  int32_t* raw = cv.take<int32_t>((size_t)n * 4);
  int* slot = cv.take<int>(n);
  int* flag = cv.take<int>(n);
  int* partial = cv.take<int>(n / SCAN_TILE + 2);
  HashTable t;
  const unsigned int cap = table_capacity(n);
  t.keys = cv.take<unsigned long long>(cap);
  t.vals = cv.take<int>(cap);
  t.mask = cap - 1;
  EYOC_CHECK_HIP(hipMemsetAsync(counters, 0, 64 * sizeof(int), st));
  EYOC_CHECK_HIP(hipMemsetAsync(t.keys, 0xFF, (size_t)cap * 8, st));
  EYOC_CHECK_HIP(hipMemsetAsync(t.vals, 0x7F, (size_t)cap * 4, st));
  hipLaunchKernelGGL(k_quantize, dim3(cdiv(n, 256)), dim3(256), 0, st, xyz_dev, n, stride, voxel_size, batch_index, raw);
  hipLaunchKernelGGL(k_insert, dim3(cdiv(n, 256)), dim3(256), 0, st, raw, n, 1, t, slot, counters);
  hipLaunchKernelGGL(k_flag, dim3(cdiv(n, 256)), dim3(256), 0, st, slot, t.vals, n, flag, (int*)nullptr);
  const int nb = cdiv(n, SCAN_TILE);
  hipLaunchKernelGGL(k_scan_partials, dim3(nb), dim3(SCAN_BLOCK), 0, st, flag, n, partial);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_BLOCK), 0, st, partial, nb, counters + 2);
  hipLaunchKernelGGL(k_compact, dim3(nb), dim3(SCAN_BLOCK), 0, st, flag, partial, slot, raw, n, 1, coords_dev, t.vals,
                     sel_dev);
  int* host = (int*)ctx->pinned;
  EYOC_CHECK_HIP(hipMemcpyAsync(host, counters, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  EYOC_CHECK_HIP(hipStreamSynchronize(st));
  EYOC_REQUIRE(host[0] == 0, EYOC_ERR_RANGE, "eyoc_voxelize: %d points fall outside the key range (|c| < 2^17 - 16)", host[0]);
  *n_out = host[2];
  return EYOC_OK;
}

int eyoc_maps_free(eyoc_maps* maps) {
  delete maps;
  return EYOC_OK;
}

int eyoc_maps_rows(const eyoc_maps* maps, int level) {
  if (!maps || level < 0 || level >= maps->n_levels) return -1;
  return maps->rows[level];
}

const int32_t* eyoc_maps_coords(const eyoc_maps* maps, int level) {
  if (!maps || level < 0 || level >= maps->n_levels) return nullptr;
  return maps->coords[level];
}

const int32_t* eyoc_maps_table(const eyoc_maps* maps, int kind, int level) {
  if (!maps || level < 0 || level >= maps->n_levels) return nullptr;
  // a table the build skipped (lazy tables) is filled now and waited for: the pointer is valid for any stream.  On the NULL stream -
  // eyoc_maps_build synchronised its own stream before it returned, so everything the fill reads is complete, and the stream the
  // maps were built on may be gone by now (a torch side stream)
  if ((kind == EYOC_MAP_S1 && !maps->s1_ready[level]) || (kind == EYOC_MAP_UP && level + 1 < maps->n_levels && !maps->up_ready[level])) {
    eyoc_maps* mm = const_cast<eyoc_maps*>(maps);
    if (maps_ensure_table(mm, kind, level, (hipStream_t)nullptr) != EYOC_OK) return nullptr;
    if (hipStreamSynchronize((hipStream_t)nullptr) != hipSuccess) return nullptr;
  }
  switch (kind) {
    case EYOC_MAP_S1: return maps->nbr_s1[level];
    case EYOC_MAP_DOWN: return level + 1 < maps->n_levels ? maps->nbr_down[level] : nullptr;
    case EYOC_MAP_UP: return level + 1 < maps->n_levels ? maps->nbr_up[level] : nullptr;
    default: return nullptr;
  }
}

int eyoc_maps_copy_coords(const eyoc_maps* maps, int level, int32_t* out_dev, void* stream) {
  const int32_t* src = eyoc_maps_coords(maps, level);
  EYOC_REQUIRE(src && out_dev, EYOC_ERR_INVALID, "eyoc_maps_copy_coords: bad level %d or NULL output", level);
  EYOC_CHECK_HIP(hipMemcpyAsync(out_dev, src, (size_t)maps->rows[level] * 16, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return EYOC_OK;
}

int eyoc_maps_copy_table(const eyoc_maps* maps, int kind, int level, int32_t* out_dev, void* stream) {
  const int32_t* src = eyoc_maps_table(maps, kind, level);
  EYOC_REQUIRE(src && out_dev, EYOC_ERR_INVALID, "eyoc_maps_copy_table: bad kind %d / level %d or NULL output", kind, level);
  const int n_out = kind == EYOC_MAP_DOWN ? maps->rows[level + 1] : maps->rows[level];
  EYOC_CHECK_HIP(hipMemcpyAsync(out_dev, src, (size_t)n_out * 27 * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return EYOC_OK;
}

}  // extern "C"

int eyoc::maps_ensure_table(eyoc_maps* m, int kind, int level, hipStream_t st) {
  if (!m || level < 0 || level + 1 >= m->n_levels) return EYOC_OK;      // the coarsest level's table is always built
  const bool s1 = kind == EYOC_MAP_S1 && !m->s1_ready[level], up = kind == EYOC_MAP_UP && !m->up_ready[level];
  if (!s1 && !up) return EYOC_OK;
  const int nl = m->rows[level], nc = m->rows[level + 1];
  if (nl > 0) {
    if (s1)
      hipLaunchKernelGGL((k_derive_fine<true, 0>), dim3(cdiv(nl, 256)), dim3(256), 0, st, m->coords[level], nl, level, m->parent[level],
                         m->children[level], m->nbr_s1[level + 1], nc, m->nbr_s1[level], (int32_t*)nullptr, (unsigned int*)nullptr,
                         (int*)nullptr, 0u, -1);
    else
      hipLaunchKernelGGL((k_derive_fine<false, 1>), dim3(cdiv(nl, 256)), dim3(256), 0, st, m->coords[level], nl, level, m->parent[level],
                         m->children[level], m->nbr_s1[level + 1], nc, (int32_t*)nullptr, m->nbr_up[level], (unsigned int*)nullptr,
                         (int*)nullptr, 0u, -1);
    EYOC_CHECK_HIP(hipGetLastError());
  }
  (s1 ? m->s1_ready : m->up_ready)[level] = true;
  return EYOC_OK;
}

// level-0 hash table on demand (only the hash-probing first-convolution fallback reads it)
int eyoc::maps_build_table0(eyoc_maps* m, hipStream_t st) {
  if (m->table0_built) return EYOC_OK;
  HashTable& t = m->table[0];
  const size_t cap = (size_t)t.mask + 1;
  EYOC_CHECK_HIP(hipMemsetAsync(t.keys, 0xFF, cap * 8, st));
  EYOC_CHECK_HIP(hipMemsetAsync(t.vals, 0x7F, cap * 4, st));
  hipLaunchKernelGGL(k_insert, dim3(cdiv(m->rows[0], 256)), dim3(256), 0, st, m->coords[0], m->rows[0], 1, t, (int*)nullptr,
                     (int*)nullptr /* no range errors: the rows passed eyoc_maps_build */);
  EYOC_CHECK_HIP(hipGetLastError());
  m->table0_built = true;
  return EYOC_OK;
}

extern "C" {

int eyoc_maps_internal_order(eyoc_ctx* ctx, int mode) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.maps_internal_order;
  if (mode >= -1 && mode <= 1) ctx->knobs.maps_internal_order = mode;
  return prev + 2;
}

const int32_t* eyoc_maps_row_order(const eyoc_maps* maps) { return maps ? maps->row_perm : nullptr; }

int eyoc_spconv_select_down_kernel(eyoc_ctx* ctx, int mode) {
  EYOC_REQUIRE(ctx, EYOC_ERR_INVALID, "eyoc_spconv_select_down_kernel: NULL ctx");
  const int prev = ctx->knobs.down_kernel;
  if (mode == 0 || mode == 1) ctx->knobs.down_kernel = mode;           // anything else: a query ...
  if (mode == 2 || mode == 3) ctx->knobs.st128_wide = mode - 2;        // ... or (diagnostics) 2 / 3: 64- / 128-channel workgroups for >= 128-channel layers
  if (mode == 4 || mode == 5) ctx->knobs.s1_wide = mode - 4;           // 4 / 5: stride-1 layers with >= 128 channels on 256-row x 64-channel / 128-row x 128-channel workgroups
  return prev;
}

int eyoc_maps_lazy_tables(eyoc_ctx* ctx, int on) {
  EYOC_REQUIRE(ctx, EYOC_ERR_INVALID, "eyoc_maps_lazy_tables: NULL ctx");
  const int prev = ctx->knobs.maps_lazy_tables;
  if (on == 0 || on == 1) ctx->knobs.maps_lazy_tables = on;            // anything else: a query
  return prev;
}

int eyoc_maps_fused_levels(eyoc_ctx* ctx, int on) {
  EYOC_REQUIRE(ctx, EYOC_ERR_INVALID, "eyoc_maps_fused_levels: NULL ctx");
  const int prev = ctx->knobs.maps_fused_levels;
  if (on == 0 || on == 1) ctx->knobs.maps_fused_levels = on;          // anything else: a query
  return prev;
}

int eyoc_maps_select_orders(eyoc_ctx* ctx, int s1, int down) {
  if (!ctx) return -1;
  eyoc_ctx::Knobs& kn = ctx->knobs;
  const int prev = kn.maps_s1_order | kn.maps_down_order << 1;
  if (s1 == 0 || s1 == 1) kn.maps_s1_order = s1;
  if (down == 0 || down == 1) kn.maps_down_order = down;
  return prev;
}

int eyoc_maps_order_window_shift(eyoc_ctx* ctx, int shift) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.maps_window_shift;
  if (shift >= 0) ctx->knobs.maps_window_shift = shift;
  return prev;
}

int eyoc_maps_order_min_rows(eyoc_ctx* ctx, int min_rows) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.maps_order_min_rows;
  if (min_rows >= 0) ctx->knobs.maps_order_min_rows = min_rows;
  return prev;
}

int eyoc_maps_copy_up_order(const eyoc_maps* maps, int level, int32_t* out_dev, void* stream) {
  EYOC_REQUIRE(maps && out_dev && level >= 0 && level + 1 < maps->n_levels, EYOC_ERR_INVALID,
               "eyoc_maps_copy_up_order: bad level %d or NULL argument", level);
  const int n = maps->rows[level];
  if (maps->perm_up[level]) {
    EYOC_CHECK_HIP(hipMemcpyAsync(out_dev, maps->perm_up[level], (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  } else {   // level too small to be ordered: the convolutions tile it in natural order
    hipLaunchKernelGGL(k_iota, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out_dev, n);
    EYOC_CHECK_HIP(hipGetLastError());
  }
  return EYOC_OK;
}

int eyoc_maps_copy_row_order(const eyoc_maps* maps, int32_t* out_dev, void* stream) {
  EYOC_REQUIRE(maps && out_dev, EYOC_ERR_INVALID, "eyoc_maps_copy_row_order: NULL argument");
  const int n = maps->rows[0];
  if (maps->row_perm) {
    EYOC_CHECK_HIP(hipMemcpyAsync(out_dev, maps->row_perm, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  } else {   // the caller's order was kept
    hipLaunchKernelGGL(k_iota, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out_dev, n);
    EYOC_CHECK_HIP(hipGetLastError());
  }
  return EYOC_OK;
}

int eyoc_maps_info(eyoc_ctx* ctx, const eyoc_maps* m, int conv1_ks, void* stream, eyoc_maps_info_t* info) {
  EYOC_REQUIRE(ctx && m && info, EYOC_ERR_INVALID, "eyoc_maps_info: NULL argument");
  hipStream_t st = (hipStream_t)stream;
  int rc = ctx->ensure_scratch(64 * sizeof(unsigned long long), st);
  if (rc) return rc;
  unsigned long long* cnt = (unsigned long long*)ctx->scratch;
  EYOC_CHECK_HIP(hipMemsetAsync(cnt, 0, 64 * sizeof(unsigned long long), st));
  memset(info, 0, sizeof(*info));
  info->n_levels = m->n_levels;
  auto count = [&](const int32_t* t, long long n, int slot) {
    if (!t || n == 0) return;
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_count_valid, dim3(blocks), dim3(256), 0, st, t, n, cnt + slot);
  };
  for (int l = 0; l < m->n_levels; ++l) {
    info->rows[l] = m->rows[l];
    if ((rc = maps_ensure_table(const_cast<eyoc_maps*>(m), EYOC_MAP_S1, l, st))) return rc;
    if (l + 1 < m->n_levels && (rc = maps_ensure_table(const_cast<eyoc_maps*>(m), EYOC_MAP_UP, l, st))) return rc;
    count(m->nbr_s1[l], 27ll * m->rows[l], l);
    if (l + 1 < m->n_levels) {
      count(m->nbr_down[l], 27ll * m->rows[l + 1], 8 + l);
      count(m->nbr_up[l], 27ll * m->rows[l], 16 + l);
    }
  }
  if (conv1_ks > 0) {   // the one statistic that probes the level-0 hash table: build it now if nobody has
    rc = maps_build_table0(const_cast<eyoc_maps*>(m), st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_count_region, dim3(cdiv(m->rows[0], 256)), dim3(256), 0, st, m->coords[0], m->rows[0],
                       m->table[0], conv1_ks, cnt + 24);
  }
  unsigned long long* host = (unsigned long long*)ctx->pinned;
  EYOC_CHECK_HIP(hipMemcpyAsync(host, cnt, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  EYOC_CHECK_HIP(hipStreamSynchronize(st));
  for (int l = 0; l < m->n_levels; ++l) {
    info->pairs_s1[l] = (int64_t)host[l];
    info->pairs_down[l] = (int64_t)host[8 + l];
    info->pairs_up[l] = (int64_t)host[16 + l];
  }
  info->pairs_conv1 = (int64_t)host[24];
  return EYOC_OK;
}

}  // extern "C"
