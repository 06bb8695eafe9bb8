// Transposed convolution (kernel 3^3, stride 2) in CLASS-MAJOR order (gfx950, SPLIT16 arithmetic) - round 4.
//
// A fine output row reaches coarse input rows only through the offsets of ITS parity class: an axis on which the row's
// coordinate is an even multiple of the fine stride admits offset 0 only, an odd one the offsets -1 and +1.  The 27 offsets
// fall into 8 disjoint classes of 1, 2, 2, 4, 2, 4, 4, 8 offsets (1 + 3*2 + 3*4 + 8 = 27), and a row of class b reads through
// 2^|b| of them - 3.4 on average, of which 2.5 exist.
//
// spconv_up.hip tiles the fine rows in their own (Morton) order: the 256 rows of a tile are of all 8 classes, every wave
// walks (nearly) all 27 offsets for 1-2 chunks of 16 rows each, and the kernel is bound by its weight stream through L1 / L2
// (27 x 8 KB x C_in / 32 per tile, ~900 KB for 64 KB of output).  Here the rows are first sorted by
// class (a stable 8-way partition: Morton order inside a class), a tile is 256 rows of ONE class, and the workgroup runs the
// stride-1 staged kernel's generated assembly loop (gen_st_loop.py, blobs UPC1 / 2 / 4 / 8) over that class's offsets only:
//   * the weight fragments of an offset serve all 16 chunks of the tile (weight bytes per tile / 6 on average);
//   * no offset is walked in vain (the 27-offset loop on a transposed table spends 2/3 of its time on loop overhead:
//     scripts/exp_up_staged.py), and inside a tile the rows are grouped by which of their parents exist, so that the loop's
//     empty-block branches remove most of the products with missing parents (2.7-3.0 offsets per chunk against 2.5 useful);
//   * the price: the rows of a class tile (256; 192 for the class with 8 offsets) spread over 8 x the volume, ~420 distinct coarse rows instead of ~145 (two stage
//     passes for 4-7 % of the tiles); the tiles of the 8 classes that cover the same stretch of the Morton curve run back to
//     back on ONE XCD (tile_order), so that their common coarse rows are fetched from HBM once and from that XCD's L2 after.
// Output rows are written through a per-slot row index (the tile's rows are not contiguous in the output).
#include <atomic>

#include "spconv.h"
#include "derive.h"

using namespace eyoc;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NW = 4;
constexpr int TILE = NW * 64;
constexpr int XROWS = 640;                                 // the stage of spconv_st.hip: 640 rows x 128 bytes of one 32-channel block
constexpr int UMAX = XROWS - 1;
constexpr int NPASS = 2;
constexpr int UCAP = NPASS * XROWS;
constexpr int X_BYTES = XROWS * 128;
constexpr int LO_REGION = XROWS * 64;
constexpr int KC = 8;                                      // offset slots of a record (a class has 1, 2, 4 or 8)
__host__ __device__ constexpr int slot_addr(int l) { return l * 64 + ((l >> 2) & 3) * 16; }

// ---- record of one class tile (UPC_LR bytes): int n_unique (-1: more than NPASS * UMAX distinct coarse rows), int class,
// pad[2]; int U[UCAP]; uint2 loc[NPASS][KC][64] (entry (pass, i, 16 w + j): the LDS slot addresses of the parents at the class's
// i-th offset of tile slots 64 w + 16 c + j, c = 0..3, as four 16-bit values; slot UMAX = no parent / other pass);
// unsigned short mask[NPASS][KC] (bit 4 w + c: some row of slot chunk (w, c) has a parent at offset i staged in this pass);
// int orow[64][4]: output row of slot 64 w + 16 c + j at [(16 w + j)][c], -1 = a padding slot (a class-7 tile holds 192 rows; the
// last tile of a class what is left).
constexpr int U_OFF = 16;
constexpr int LOC_OFF = U_OFF + UCAP * 4;                          // 5136
constexpr int MASK_OFF = LOC_OFF + NPASS * KC * 64 * 8;            // 13328
constexpr int OROW_OFF = MASK_OFF + NPASS * KC * 2;                // 13360
constexpr int UPC_LR = (OROW_OFF + TILE * 4 + 127) / 128 * 128;    // 14464
static_assert(OROW_OFF % 16 == 0 && LOC_OFF % 16 == 0, "16-byte accesses");

// offsets enumerate x fastest: k = (dx + 1) + 3 (dy + 1) + 9 (dz + 1); class bit a = "d_a != 0"
__host__ __device__ constexpr int class_of(int k) { return (k % 3 != 1 ? 1 : 0) | ((k / 3) % 3 != 1 ? 2 : 0) | (k / 9 != 1 ? 4 : 0); }
struct ClassTable {
  int order[27];   // the offsets, class by class, ascending inside a class
  int start[9];
};
constexpr ClassTable make_classes() {
  ClassTable t{};
  int n = 0;
  for (int b = 0; b < 8; ++b) {
    t.start[b] = n;
    for (int k = 0; k < 27; ++k)
      if (class_of(k) == b) t.order[n++] = k;
  }
  t.start[8] = n;
  return t;
}
__constant__ ClassTable c_classes = make_classes();
// Rows per tile, by class.  The 256 rows of a class-7 tile (8 parents each) reach 650 distinct coarse rows at the median - more
// than one stage pass holds (639) - and a second pass repeats most of the tile's products (the parents of a chunk are spread over
// both passes): 192 rows (12 of the 16 chunks; ~520 distinct rows) keep 9 of 10 such tiles in one pass.
// (eyoc_spconv_upc_tile_rows sets the rows per tile of the classes with 0 / 1 / 2 / 3 odd axes - for measurements)
constexpr int MIN_TILE_ROWS = 128;
__device__ int g_tile_rows[8] = {256, 256, 256, 256, 256, 256, 256, 192};
__device__ inline int tile_rows(int b) { return g_tile_rows[b]; }

// ---- per-level header (device): what the class partition came out as
struct UpcHeader {
  int n_tiles;            // sum over classes of ceil(count / 256)
  int tile_start[9];      // first tile of class b (class-major numbering)
  int count[8];
  int overflow;           // tiles with more than NPASS * UMAX distinct coarse rows
};
constexpr int HDR_BYTES = 256;

// 1. class of every fine row and the class counts of every 1024-row block.  With the level's coordinates at hand the class is
// the parity of (c / stride) per axis (an even axis admits offset 0 only, an odd one -1 and +1); without, that of the row's first
// valid offset (each offset belongs to one class, and a row's own coarse cell always exists) - 27 table entries instead of 16 bytes.
constexpr int PBLK = 1024;                                             // rows per partition block (4 per thread)
__global__ __launch_bounds__(256) void k_upc_class(const int32_t* __restrict__ nbr, const int32_t* __restrict__ coords, int stride, int n,
                                                   unsigned char* __restrict__ cls, int* __restrict__ blk_cnt) {
  __shared__ int cnt[NW][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < PBLK / 256; ++i) {
    const int row = blockIdx.x * PBLK + i * 256 + (int)threadIdx.x;
    int c = 8;                                                        // 8 = no row
    if (row < n) {
      if (coords) {
        const int4 q = reinterpret_cast<const int4*>(coords)[row];   // (batch, x, y, z), multiples of the level's stride
        c = ((q.y / stride) & 1) | (((q.z / stride) & 1) << 1) | (((q.w / stride) & 1) << 2);
      } else {
        c = 0;
#pragma unroll
        for (int k = 26; k >= 0; --k)
          if (nbr[(size_t)k * n + row] >= 0) c = class_of(k);
      }
      cls[row] = (unsigned char)c;
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) mine[b] += __popcll(__ballot(c == b));
  }
  if (lane == 0)
#pragma unroll
    for (int b = 0; b < 8; ++b) cnt[wave][b] = mine[b];
  __syncthreads();
  if (threadIdx.x < 8) blk_cnt[blockIdx.x * 8 + threadIdx.x] = cnt[0][threadIdx.x] + cnt[1][threadIdx.x] + cnt[2][threadIdx.x] + cnt[3][threadIdx.x];
}

// 2. one workgroup: exclusive scan of the block counts per class (in place: blk_cnt becomes the block's first position inside
// its class), the class sizes, the tile numbering
__global__ __launch_bounds__(1024) void k_upc_scan(int* __restrict__ blk_cnt, int nblk, UpcHeader* __restrict__ hdr) {
  __shared__ int part[1024][8];
  const int t = (int)threadIdx.x;
  const int per = (nblk + 1023) / 1024, b0 = t * per, b1 = min(nblk, b0 + per);
  int s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = b0; b < b1; ++b) {
    const int4 lo = reinterpret_cast<const int4*>(blk_cnt)[b * 2], hi = reinterpret_cast<const int4*>(blk_cnt)[b * 2 + 1];
    s[0] += lo.x; s[1] += lo.y; s[2] += lo.z; s[3] += lo.w; s[4] += hi.x; s[5] += hi.y; s[6] += hi.z; s[7] += hi.w;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) part[t][c] = s[c];
  __syncthreads();
  // Hillis-Steele over the 1024 partial sums, 8 classes at once
  for (int d = 1; d < 1024; d <<= 1) {
    int v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = t >= d ? part[t - d][c] : 0;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c) part[t][c] += v[c];
    __syncthreads();
  }
  int run[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) run[c] = part[t][c] - s[c];             // exclusive
  for (int b = b0; b < b1; ++b) {
    const int4 lo = reinterpret_cast<const int4*>(blk_cnt)[b * 2], hi = reinterpret_cast<const int4*>(blk_cnt)[b * 2 + 1];
    reinterpret_cast<int4*>(blk_cnt)[b * 2] = make_int4(run[0], run[1], run[2], run[3]);
    reinterpret_cast<int4*>(blk_cnt)[b * 2 + 1] = make_int4(run[4], run[5], run[6], run[7]);
    run[0] += lo.x; run[1] += lo.y; run[2] += lo.z; run[3] += lo.w; run[4] += hi.x; run[5] += hi.y; run[6] += hi.z; run[7] += hi.w;
  }
  if (t == 0) {
    int tiles = 0;
    for (int c = 0; c < 8; ++c) {
      const int total = part[1023][c];
      hdr->count[c] = total;
      hdr->tile_start[c] = tiles;
      tiles += (total + tile_rows(c) - 1) / tile_rows(c);
    }
    hdr->tile_start[8] = tiles;
    hdr->n_tiles = tiles;
    hdr->overflow = 0;
  }
}

// 3. the stable partition: sorted[tile_start[class] * 256 + position inside the class] = row (the array is pre-set to -1: the
// padding slots of a class's last tile)
__global__ __launch_bounds__(256) void k_upc_scatter(const unsigned char* __restrict__ cls, int n, const int* __restrict__ blk_base,
                                                     const UpcHeader* __restrict__ hdr, int* __restrict__ sorted) {
  __shared__ int cnt[PBLK / 256][NW][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int c[PBLK / 256], rank[PBLK / 256];
#pragma unroll
  for (int i = 0; i < PBLK / 256; ++i) {
    const int row = blockIdx.x * PBLK + i * 256 + (int)threadIdx.x;
    c[i] = row < n ? (int)cls[row] : 8;
    rank[i] = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot(c[i] == b);
      if (c[i] == b) rank[i] = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) cnt[i][wave][b] = __popcll(m);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PBLK / 256; ++i) {
    if (c[i] >= 8) continue;
    int r = rank[i];
    for (int q = 0; q < i * NW + wave; ++q) r += cnt[q / NW][q % NW][c[i]];       // the rows in front: earlier quarters, earlier waves
    const int pos = blk_base[blockIdx.x * 8 + c[i]] + r, R = tile_rows(c[i]);                 // position inside the class
    sorted[(size_t)(hdr->tile_start[c[i]] + pos / R) * TILE + pos % R] = blockIdx.x * PBLK + i * 256 + (int)threadIdx.x;
  }
}

// 4. launch order of the tiles: tile ti of class b (of T_b) sits at (ti + 1/2) / T_b of the Morton curve; the tiles are
// numbered by that position (ties: class), so that consecutive entries are the 8 classes' tiles over the same stretch
__global__ __launch_bounds__(256) void k_upc_order(const UpcHeader* __restrict__ hdr, int* __restrict__ tile_order) {
  const int t = blockIdx.x * 256 + (int)threadIdx.x;
  if (t >= hdr->n_tiles) return;
  int b = 0;
  while (t >= hdr->tile_start[b + 1]) ++b;
  const long long ti = t - hdr->tile_start[b], Tb = hdr->tile_start[b + 1] - hdr->tile_start[b];
  int rank = 0;
  for (int o = 0; o < 8; ++o) {
    const long long To = hdr->tile_start[o + 1] - hdr->tile_start[o];
    if (To == 0) continue;
    if (o == b) { rank += (int)ti; continue; }
    // tiles ti' of class o with (2 ti' + 1) Tb < (2 ti + 1) To  (<= when o < b)
    const long long A = (2 * ti + 1) * To;
    const long long q = o < b ? A / Tb : (A + Tb - 1) / Tb - 1;       // largest admissible value of 2 ti' + 1
    long long cnt = q >= 1 ? (q - 1) / 2 + 1 : 0;
    if (cnt > To) cnt = To;
    rank += (int)cnt;
  }
  tile_order[rank] = t;
}

// 5. the records
constexpr int HSLOTS = 4096;                                          // LDS hash of a tile's distinct coarse rows (at most 256 * 8)
// compact: `nbr` is the [8][n] table of derive.h (up8: slot = bit per axis "offset -1") instead of the full [27][n] one
__global__ __launch_bounds__(256) void k_upc_records(const int32_t* __restrict__ nbr, int n, const int* __restrict__ sorted,
                                                     UpcHeader* __restrict__ hdr, unsigned char* __restrict__ out, int compact) {
  __shared__ int hk[HSLOTS];
  __shared__ unsigned short hid[HSLOTS], hpos[HSLOTS];
  __shared__ unsigned short srow[KC][TILE];
  __shared__ __attribute__((aligned(16))) unsigned int key[TILE];
  __shared__ int wave_cnt[NW];
  __shared__ unsigned char nib[NPASS][KC][NW];
  const int tile = blockIdx.x;
  if (tile >= hdr->n_tiles) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int b = 0;
  while (tile >= hdr->tile_start[b + 1]) ++b;
  const int k0 = c_classes.start[b], nk = c_classes.start[b + 1] - k0;
  const int row = sorted[(size_t)tile * TILE + threadIdx.x];
  for (int i = threadIdx.x; i < HSLOTS; i += 256) hk[i] = -1;
  __syncthreads();
  int idxs[KC];
#pragma unroll
  for (int i = 0; i < KC; ++i) {
    const int k = c_classes.order[k0 + (i < nk ? i : 0)];
    idxs[i] = (i < nk && row >= 0) ? nbr[(size_t)(compact ? up8_slot_of_offset(k) : k) * n + row] : -1;
  }
  unsigned int pattern = 0;
#pragma unroll
  for (int i = 0; i < KC; ++i) {
    const int idx = idxs[i];
    unsigned int s = 0xFFFFu;
    if (idx >= 0) {
      pattern |= 1u << i;
      s = ((unsigned)idx * 2654435761u) >> 20;
      while (true) {                                                   // at most 2048 distinct keys in 4096 slots: always ends
        const int prev = atomicCAS(&hk[s], -1, idx);
        if (prev == -1 || prev == idx) break;
        s = (s + 1) & (HSLOTS - 1);
      }
    }
    srow[i][threadIdx.x] = (unsigned short)s;
  }
  // rows with the same existing parents next to each other (padding rows - no parent at all - first)
  key[threadIdx.x] = (pattern << 8) | threadIdx.x;
  __syncthreads();
  constexpr int PER_WAVE = HSLOTS / NW;
  int cnt = 0;
  for (int i0 = 0; i0 < PER_WAVE; i0 += 64) cnt += __popcll(__ballot(hk[wave * PER_WAVE + i0 + lane] >= 0));
  if (lane == 0) wave_cnt[wave] = cnt;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < NW; ++w) { if (w < wave) base += wave_cnt[w]; total += wave_cnt[w]; }
  unsigned char* lr = out + (size_t)tile * UPC_LR;
  int* U = reinterpret_cast<int*>(lr + U_OFF);
  if (total <= UMAX) {                                                 // one pass: any numbering gives the same sums - slot order
    for (int i0 = 0; i0 < PER_WAVE; i0 += 64) {
      const int s = wave * PER_WAVE + i0 + lane;
      const int kv = hk[s];
      const unsigned long long m = __ballot(kv >= 0);
      const int id = base + __popcll(m & ((1ull << lane) - 1ull));
      if (kv >= 0) {
        hid[s] = (unsigned short)id;
        U[id] = kv;
      }
      base += __popcll(m);
    }
  } else {                                                              // several passes: a numbering that does not depend on the order the CAS loops ran in (canonical_slot_id, spconv.h)
    for (int i0 = 0; i0 < PER_WAVE; i0 += 64) {
      const int s = wave * PER_WAVE + i0 + lane;
      const unsigned long long m = __ballot(hk[s] >= 0);
      if (hk[s] >= 0) hpos[s] = (unsigned short)(base + __popcll(m & ((1ull << lane) - 1ull)));   // occupied slots in front of s
      base += __popcll(m);
    }
    __syncthreads();
    for (int i0 = 0; i0 < PER_WAVE; i0 += 64) {
      const int s = wave * PER_WAVE + i0 + lane;
      const int kv = hk[s];
      if (kv >= 0) {
        const int id = canonical_slot_id<HSLOTS>(hk, hpos, s);
        hid[s] = (unsigned short)id;
        if (id < NPASS * UMAX) U[id] = kv;
      }
    }
  }
  if (threadIdx.x == 0) {
    reinterpret_cast<int*>(lr)[0] = total <= NPASS * UMAX ? total : -1;
    reinterpret_cast<int*>(lr)[1] = b;
    if (total > NPASS * UMAX) atomicAdd(&hdr->overflow, 1);
  }
  {                                                                    // the keys in ascending order: rank by counting (distinct keys; see k_local_rulebook)
    const unsigned int mine = key[threadIdx.x];
    int rank = 0;
    const uint4* kp = reinterpret_cast<const uint4*>(key);
#pragma unroll 8
    for (int i = 0; i < TILE / 4; ++i) {
      const uint4 kk = kp[i];
      rank += (kk.x < mine) + (kk.y < mine) + (kk.z < mine) + (kk.w < mine);
    }
    __syncthreads();
    key[rank] = mine;
    __syncthreads();
  }
  // slot s = 64 w + 16 c + j takes the row at sorted position (c * 4 + snake(w)) * 16 + j: the sorted 16-row chunks are dealt
  // over the four 64-slot quarters in snake order (as in k_local_rulebook), so that the waves carry the same load
  const int r = (int)threadIdx.x, w = r >> 6, c = (r >> 4) & 3, j = r & 15;
  const int src = (int)(key[(c * 4 + ((c & 1) ? 3 - w : w)) * 16 + j] & 255u);
  const int orow = sorted[(size_t)tile * TILE + src];
  reinterpret_cast<int*>(lr + OROW_OFF)[(w * 16 + j) * 4 + c] = orow;
  unsigned short* loc = reinterpret_cast<unsigned short*>(lr + LOC_OFF);
#pragma unroll
  for (int i = 0; i < KC; ++i) {
    if (i >= nk) break;
    const unsigned int hs = srow[i][src];
    const int id = hs != 0xFFFFu ? (int)hid[hs] : -1;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      if (p > 0 && total <= p * UMAX) continue;
      const int l = (id >= p * UMAX && id < (p + 1) * UMAX) ? id - p * UMAX : UMAX;
      loc[(((size_t)(p * KC + i) * 64) + w * 16 + j) * 4 + c] = (unsigned short)slot_addr(l);
      const unsigned long long bl = __ballot(l != UMAX);
      if (lane == 0)
        nib[p][i][w] = (unsigned char)(((bl & 0xFFFFull) != 0) | (((bl >> 16) & 0xFFFFull) != 0) << 1 | (((bl >> 32) & 0xFFFFull) != 0) << 2 |
                                       (((bl >> 48) & 0xFFFFull) != 0) << 3);
    }
  }
  __syncthreads();
  if (threadIdx.x < NPASS * KC) {
    const int p = (int)threadIdx.x / KC, i = (int)threadIdx.x % KC;
    unsigned short m = 0;
    if (i < nk && !(p > 0 && total <= p * UMAX)) m = (unsigned short)(nib[p][i][0] | nib[p][i][1] << 4 | nib[p][i][2] << 8 | nib[p][i][3] << 12);
    reinterpret_cast<unsigned short*>(lr + MASK_OFF)[p * KC + i] = m;
  }
}

#include "spconv_st_loop.inc"
// timing-only builds (results are garbage): -DEYOC_UPC_ABL=1 no MFMAs, 2 no weight loads, 3 nothing in the loop, 4 no stage DMA,
// 5 no output stores (make ../lib/libeyoc_hip_upcabl<N>.so)
#ifdef EYOC_UPC_ABL
#include "spconv_st_loop_abl.inc"
#if EYOC_UPC_ABL == 1
#define EYOC_UPC_BLOB EYOC_ST_LOOP_UPC_NOM
#elif EYOC_UPC_ABL == 2
#define EYOC_UPC_BLOB EYOC_ST_LOOP_UPC_NOW
#elif EYOC_UPC_ABL == 3
#define EYOC_UPC_BLOB EYOC_ST_LOOP_UPC_EMPTY
#else
#define EYOC_UPC_BLOB EYOC_ST_LOOP_UPC
#endif
#else
#define EYOC_UPC_ABL 0
#define EYOC_UPC_BLOB EYOC_ST_LOOP_UPC
#endif

// One workgroup = one class tile x 64 output channels: 4 waves = 2 row halves x 2 channel halves (128 rows x 32 channels of
// register accumulators per wave), two workgroups per CU - the shape of spconv_st_asm_kernel<CC, 2, 1>.
template <int CC>
__global__ __launch_bounds__(NW * 64, 2) void spconv_upc_kernel(SpconvArgs a, const unsigned char* __restrict__ ws, int max_tiles) {
  constexpr int NTW = 2, CTW = NTW * 16, NC = 4, NH = 2, CTG = 64;
  constexpr int NITV = XROWS / (16 * NW);
  __shared__ __attribute__((aligned(128))) unsigned char xs[X_BYTES];
  const UpcHeader* hdr = reinterpret_cast<const UpcHeader*>(ws);
  const int n_tiles = __builtin_amdgcn_readfirstlane(hdr->n_tiles);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane >> 4, j = lane & 15;
  const int n_cg = a.cout / CTG;
  // XCD x (workgroups x, x + 8, ...) takes the x-th eighth of the launch order, a tile's channel groups back to back
  const int xcd = (int)blockIdx.x & 7, seq = (int)blockIdx.x >> 3;
  const int per_xcd = (n_tiles + 7) >> 3;
  const int e = xcd * per_xcd + seq / n_cg, cg = seq % n_cg;
  if (seq / n_cg >= per_xcd || e >= n_tiles) return;
  const int* tile_order = reinterpret_cast<const int*>(ws + HDR_BYTES);
  const int tile = __builtin_amdgcn_readfirstlane(tile_order[e]);
  const unsigned char* lr = ws + HDR_BYTES + (((size_t)max_tiles * 4 + 255) & ~(size_t)255) + (size_t)tile * UPC_LR;
  const int w0 = 2 * (wave >> 1);
  const int ct0 = cg * CTG + (wave & 1) * CTW;
  const int CT = a.cout >= 128 ? 128 : a.cout;
  const int n_slices = a.cout / CT, slice = ct0 / CT, nt0 = (ct0 - slice * CT) / 16;
  constexpr int JQ = CC / 16;
  const int ncc = a.cin / CC;
  const int nqb = a.cin / 32;

  const int* __restrict__ U = reinterpret_cast<const int*>(lr + U_OFF);
  if (threadIdx.x < 8)
    *reinterpret_cast<float4*>(xs + (threadIdx.x >> 2) * LO_REGION + UMAX * 64 + (threadIdx.x & 3) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned long long wbits = (unsigned long long)(size_t)a.w;
  u32x4 wr;
  wr[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wbits);
  wr[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(wbits >> 32) & 0xFFFFu));
  wr[2] = (unsigned)__builtin_amdgcn_readfirstlane(27 * a.cin * a.cout * 4);
  wr[3] = 0x00020000u;
  const int tile4 = CC * CT / 4;
  const unsigned int kstride = (unsigned)(n_slices * ncc * tile4 * 16);
  const unsigned int w1off = JQ * 1024;
  if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)xs != 0u) __builtin_trap();

  f32x16 A0, A1, A2, A3;
#pragma unroll
  for (int i = 0; i < 16; ++i) { A0[i] = 0.f; A1[i] = 0.f; A2[i] = 0.f; A3[i] = 0.f; }

  int n_up = 0;
  int Ureg[NITV];
  auto load_rows = [&](int pass) {
#pragma unroll
    for (int it = 0; it < NITV; ++it) Ureg[it] = U[pass * UMAX + (it * NW + wave) * 16 + (lane >> 2)];
  };
  auto stage = [&](int qb) {
    if (EYOC_UPC_ABL == 4) return;
#pragma unroll
    for (int it = 0; it < NITV; ++it) {
      const int l0 = (it * NW + wave) * 16;
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

  load_rows(0);
  int warm = 0;                                                        // the rulebook entries' lines into L2 while the stage is in flight
  if (threadIdx.x < KC * 64 * 8 / 128) warm = *reinterpret_cast<const int*>(lr + LOC_OFF + threadIdx.x * 128);
  const int n_u = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(lr)[0]);
  const int cls = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(lr)[1]);
  const int k0 = c_classes.start[cls], nk = c_classes.start[cls + 1] - k0;
  // byte offsets of the class's offsets in the packed weights
  u32x8 KO;
#pragma unroll
  for (int i = 0; i < 8; ++i) KO[i] = (unsigned)__builtin_amdgcn_readfirstlane(c_classes.order[k0 + (i < nk ? i : 0)]) * kstride;
  const int n_pass = n_u > UMAX ? 2 : 1;
  bool first = true;
  for (int pass = 0; pass < n_pass; ++pass) {
    n_up = min(n_u - pass * UMAX, UMAX);
    if (pass > 0) load_rows(pass);
    // the wave's occupancy masks: bit (i & 1) * 16 + 4 h + c of dword i / 2 = chunk c of the wave's row half h at offset i
    const unsigned int* mp = reinterpret_cast<const unsigned int*>(lr + MASK_OFF + pass * KC * 2);
    u32x8 M0;
#pragma unroll
    for (int i = 0; i < 8; ++i) M0[i] = i < 4 ? (unsigned)__builtin_amdgcn_readfirstlane((int)mp[i]) >> (w0 * 4) : 0u;
    const unsigned char* lb = lr + LOC_OFF + ((size_t)pass * KC * 64 + w0 * 16) * 8;
    for (int qb = 0; qb < nqb; ++qb) {
      if (!first) __syncthreads();
      first = false;
      stage(qb);
      const int cc = (qb * 32) / CC, qp = ((qb * 32) % CC) / 32;
      const unsigned int ws0 = (unsigned)__builtin_amdgcn_readfirstlane(((slice * ncc + cc) * tile4 + (nt0 * JQ + 2 * qp) * 64) * 16);
      unsigned int so;
      asm volatile(EYOC_UPC_BLOB : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), "+{v[96:111]}"(A2), "+{v[112:127]}"(A3), [so] "=&s"(so)
                   : [wr] "s"(wr), [ws0] "s"(ws0), [lb] "s"(lb), [w1] "s"(w1off), [nk] "s"(nk), "{s[36:43]}"(M0), "{s[52:59]}"(KO)
                   : "memory", "scc", EYOC_ST_LOOP_CLOBBERS);
    }
  }
  asm volatile("" :: "v"(warm));

  // ---- epilogue (that of spconv_st_asm_kernel; rows through the record's output-row index)
  const float os = a.out_scale ? *a.out_scale : 1.0f;
  const int ch = ct0 + 8 * g;
  float4 b4[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
    b4[t] = a.bias ? *reinterpret_cast<const float4*>(a.bias + ch + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
  float mx = 0.f;
  constexpr int NG = NH * NC;
  int orow[NG];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const int4 rm = *reinterpret_cast<const int4*>(lr + OROW_OFF + ((w0 + h) * 16 + j) * 16);
    orow[h * NC + 0] = rm.x; orow[h * NC + 1] = rm.y; orow[h * NC + 2] = rm.z; orow[h * NC + 3] = rm.w;
  }
  float4 v[NG][NTW];
#pragma unroll
  for (int hc = 0; hc < NG; ++hc) {
    const int o = orow[hc];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int ai = (hc * NTW + t) * 4;
      const f32x16& A = ai < 16 ? A0 : ai < 32 ? A1 : ai < 48 ? A2 : A3;
      v[hc][t] = make_float4(A[ai % 16 + 0] * os + b4[t].x, A[ai % 16 + 1] * os + b4[t].y, A[ai % 16 + 2] * os + b4[t].z,
                             A[ai % 16 + 3] * os + b4[t].w);
      if (a.relu) { v[hc][t].x = fmaxf(v[hc][t].x, 0.f); v[hc][t].y = fmaxf(v[hc][t].y, 0.f); v[hc][t].z = fmaxf(v[hc][t].z, 0.f); v[hc][t].w = fmaxf(v[hc][t].w, 0.f); }
      if (o >= 0) split16_track(mx, v[hc][t]);
    }
  }
#pragma unroll
  for (int hc = 0; hc < NG; ++hc) {
    const int o = orow[hc];
    if (o < 0) continue;
    if (EYOC_UPC_ABL == 5 && os != 12345.f) continue;
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
}

}  // namespace

namespace eyoc {

static inline int upc_max_tiles(int n_out) { return cdiv(n_out, MIN_TILE_ROWS) + 8; }
static inline size_t upc_records_off(int n_out) { return (size_t)HDR_BYTES + (((size_t)upc_max_tiles(n_out) * 4 + 255) & ~(size_t)255); }

// workspace of one transposed table with n_out fine rows: header, tile order, records - kept while the maps live - followed
// by the builder's scratch (class bytes, block counts, the partitioned row list), which the records make redundant
size_t upc_kept_bytes(int n_out) { return n_out <= 0 ? 0 : upc_records_off(n_out) + (size_t)upc_max_tiles(n_out) * UPC_LR; }
size_t upc_scratch_bytes(int n_out) {
  if (n_out <= 0) return 0;
  const size_t nblk = (size_t)cdiv(n_out, PBLK);
  return (((size_t)n_out + 255) & ~(size_t)255) + nblk * 8 * 4 + 256 + (size_t)upc_max_tiles(n_out) * TILE * 4;
}

int build_upc(const int32_t* nbr_dev, const int32_t* coords_dev, int stride, int n_out, unsigned char* ws, unsigned char* scratch, hipStream_t st,
              bool compact) {
  EYOC_REQUIRE(!compact || coords_dev, EYOC_ERR_INVALID, "build_upc: the compact table needs the level's coordinates for the classes");
  if (n_out <= 0) return EYOC_OK;
  const int nblk = cdiv(n_out, PBLK), max_tiles = upc_max_tiles(n_out);
  unsigned char* cls = scratch;
  int* blk = reinterpret_cast<int*>(scratch + (((size_t)n_out + 255) & ~(size_t)255));
  int* sorted = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(blk) + (size_t)nblk * 8 * 4 + 256);
  UpcHeader* hdr = reinterpret_cast<UpcHeader*>(ws);
  int* order = reinterpret_cast<int*>(ws + HDR_BYTES);
  EYOC_CHECK_HIP(hipMemsetAsync(sorted, 0xFF, (size_t)max_tiles * TILE * 4, st));
  hipLaunchKernelGGL(k_upc_class, dim3(nblk), dim3(256), 0, st, nbr_dev, coords_dev, stride, n_out, cls, blk);
  hipLaunchKernelGGL(k_upc_scan, dim3(1), dim3(1024), 0, st, blk, nblk, hdr);
  hipLaunchKernelGGL(k_upc_scatter, dim3(nblk), dim3(256), 0, st, cls, n_out, blk, hdr, sorted);
  hipLaunchKernelGGL(k_upc_order, dim3(cdiv(max_tiles, 256)), dim3(256), 0, st, hdr, order);
  hipLaunchKernelGGL(k_upc_records, dim3(max_tiles), dim3(256), 0, st, nbr_dev, n_out, sorted, hdr, ws + upc_records_off(n_out), compact ? 1 : 0);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

// device address of the int that counts tiles whose distinct coarse rows exceed two stage passes (the kernel must not run then)
int upc_set_tile_rows(int odd_axes, int rows) {
  if (odd_axes < 0 || odd_axes > 3 || rows < MIN_TILE_ROWS || rows > TILE || rows % 16) return EYOC_ERR_INVALID;
  int h[8];
  EYOC_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile_rows), sizeof(h)));
  for (int b = 0; b < 8; ++b)
    if (__builtin_popcount(b) == odd_axes) h[b] = rows;
  EYOC_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_rows), h, sizeof(h)));
  return EYOC_OK;
}

const int* upc_overflow_ptr(const unsigned char* ws) { return &reinterpret_cast<const UpcHeader*>(ws)->overflow; }

int launch_spconv_upc(const SpconvArgs& a, const unsigned char* ws, hipStream_t st) {
  EYOC_REQUIRE(a.math == 1 && ws && !a.l2norm && a.K == 27 && !a.res && !a.out_perm, EYOC_ERR_INVALID, "spconv_upc: unsupported layer");
  EYOC_REQUIRE(a.cout % 64 == 0 && a.cout <= 512 && a.cin % 32 == 0, EYOC_ERR_INVALID, "spconv_upc: %d -> %d channels", a.cin, a.cout);
  if (a.n_out <= 0) return EYOC_OK;
  const int max_tiles = upc_max_tiles(a.n_out), n_cg = a.cout / 64;
  const dim3 grid((unsigned)(cdiv(max_tiles, 8) * n_cg * 8)), block(NW * 64);
  if (spconv_cc(a.cin, a.cout) == 64) hipLaunchKernelGGL((spconv_upc_kernel<64>), grid, block, 0, st, a, ws, max_tiles);
  else hipLaunchKernelGGL((spconv_upc_kernel<32>), grid, block, 0, st, a, ws, max_tiles);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

}  // namespace eyoc
