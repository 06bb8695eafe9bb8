// Internal interface of the sparse-convolution kernels (spconv.hip) used by model.hip.
#pragma once
#include "common.h"

namespace eyoc {

inline int spconv_ct(int cout) { return cout >= 128 ? 128 : cout; }  // output channels per block
// input channels per staged item: 64 whenever C_in allows it (half as many barriers), else 32
inline int spconv_cc(int cin, int /*cout*/) { return cin % 64 == 0 ? 64 : 32; }

// The network's 1x1 tail fused behind a 64-channel staged layer (spconv_st.hip, TAILF): conv1_tr (1x1, [that layer's 64 channels | 32
// skip channels] -> 64, optional ReLU) + final (64 -> 32, bias folded into b2) + optional row normalisation; pointers as
// launch_tail_fused takes them.  out == NULL: no fused tail.
struct TailFuse {
  const float* skip = nullptr;   // SPLIT16 rows holding the 32 skip channels in their first block (the cat buffer at column 64)
  int ld_skip = 0;
  const float *w1 = nullptr, *s1 = nullptr, *b1 = nullptr, *w2 = nullptr, *s2 = nullptr, *b2 = nullptr;
  int relu1 = 0, l2norm = 0;
  float* out = nullptr;          // [n_out, ld_out] fp32: the network's output
  int ld_out = 0;
  const int32_t* out_perm = nullptr;
};

struct SpconvArgs {
  const int32_t* nbr;   // [K][n_out] or NULL (identity, K == 1)
  int K, n_out;
  int n_in = 0;         // rows of the input tensor when known (every entry of nbr is < n_in); must be < 2^24
  const float* in;      // rows of ld_in floats; the layer reads columns [0, cin)
  int ld_in, cin;
  const float* w;       // packed weights (eyoc_spconv_pack_weights)
  int cout;
  const float* bias;    // [cout] or NULL
  const float* res;     // residual rows (ld_res floats) or NULL
  int ld_res;
  int relu;
  int l2norm;           // divide every output row by its 2-norm (needs cout <= 128); no epsilon
  float* out;           // rows of ld_out floats; the layer writes columns [0, cout)
  int ld_out;
  // ---- arithmetic / storage format (wave-private kernel only; SPLIT16 is documented in spconv_wave.hip)
  // math 0: fp32 rows, v_mfma_f32_16x16x4_f32.  math 1: `in` (and `res`) are SPLIT16 rows, `w` is packed by
  // eyoc_spconv_pack_weights_split16, three v_mfma_f32_16x16x32_f16 per product block (hi*hi + hi*lo + lo*hi).
  int math = 0;
  int out_split = 0;               // write SPLIT16 rows (internal activations) instead of fp32 rows
  const int32_t* out_perm = nullptr;      // output row o is written to row out_perm[o] of `out` (the network output in the caller's order); res is not permuted
  const unsigned char* local = nullptr;   // per-tile local rulebooks of `nbr` (build_local_rulebook / build_local_rulebook128) or NULL: enables the staged kernel
  const unsigned char* local128 = nullptr;   // ... of `nbr` in 128-row tiles (build_local_rulebook128): a stride-1 layer with >= 128 output channels then runs as 128 rows x 128 channels per workgroup
  const unsigned char* local_down = nullptr;   // ... of a strided table in 128-row tiles (build_local_rulebook128): enables launch_spconv_st128
  const unsigned char* local_up = nullptr;   // ... of a transposed table (build_local_rulebook_up): enables spconv_up.hip
  const unsigned char* local_upc = nullptr;  // ... in class-major order (build_upc): enables spconv_upc.hip
  const float* out_scale = nullptr;  // device scalar multiplied into the accumulated sums (undoes the weight pre-scale); NULL = 1
  // optional: a permutation of the output rows; tiles take rows in this order (any order gives the same result,
  // a good one makes the rows of a tile share their occupied offsets).  NULL = natural order.
  const int32_t* perm = nullptr;
  // wave-private kernel only: the first `small_rows` rows (in tiling order; a multiple of 64) are cut into 32-row
  // tiles.  They run last (heavy tiles first), so the kernel's tail is made of short-lived waves.  0 = none.
  int small_rows = 0;
  // SPLIT16 range guard (see split16_guard below): device words {overflow flag, max |x| bits, probe switch} or NULL
  unsigned int* range = nullptr;
  eyoc_ctx* ctx = nullptr;         // the caller's context (per-device launch state, e.g. function attributes already set) or NULL
  // staged stride-1 kernel on SMALL inputs (spconv_st.hip, round 5): scratch for splitting a tile's 32-channel input blocks over
  // several workgroups - partial accumulators [KS_MAX_SLOTS][32 floats x 256 threads].  NULL: no split.
  float* ks_part = nullptr;
  // fp32 workgroup-tiled kernel (spconv.hip), small inputs: `offset_split` > 1 workgroups share a row tile, each walking a contiguous
  // range of the K offsets and storing its raw sums in `offset_part` [offset_split][n_out][cout]; a second launch adds the shares in
  // order and applies the epilogue.  Set by launch_spconv itself when the caller allows it (`allow_offset_split`: the summation order
  // then depends on the problem size - training layers, where no other kernel has to give the same bits)
  int allow_offset_split = 0, offset_split = 1;
  float* offset_part = nullptr;
  int cg_local = 0;                // staged kernels: 1 = the channel groups of a tile run back to back on ONE XCD (set by the launcher, see launch_spconv_st)
  TailFuse tail;                   // staged stride-1 kernel only (spconv_st_can_fuse_tail): this layer's output goes straight into the 1x1 tail
};
constexpr int KS_MAX_SLOTS = 1024;                        // (workgroups x splits) a split launch may use: 32 KB of partial sums each
constexpr size_t KS_PART_BYTES = (size_t)KS_MAX_SLOTS * 256 * 32 * 4;

// ---- SPLIT16 row format: every block of 32 channels takes 128 bytes (one cache line), the 32 fp16 "hi" halves (x rounded
// to fp16) followed by the 32 fp16 "lo" halves (x - hi rounded to fp16): the same 4 bytes per channel as fp32, 22
// significant bits (hi + lo reproduces x to 2^-22 relative, or 2^-25 absolute below 2^-3: fp16 subnormals are honoured by
// the fp16 MFMA on gfx950, scripts/micro/mfma_f16_denorm.hip).  Channel c of a row lives at byte (c / 32) * 128 +
// (c % 32) * 2 (hi) and + 64 (lo), so leading dimensions and column offsets stay what they are for fp32 rows as long as
// they are multiples of 32 channels.  The 64-byte halves matter: the four lanes that gather one row's MFMA operand read
// 64 CONTIGUOUS bytes per instruction (hi halves of 32 channels, then the lo halves).  An earlier layout interleaved hi
// and lo per 8 channels; each gather instruction then touched both 64-byte sectors of the line, and the HBM-side read
// traffic of the row-stationary kernel was 1.5x the algorithmic gather bytes (FETCH_SIZE, calibrated with
// scripts/micro/fetch_calib.hip).
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

__device__ inline void split16_encode4(const float4 v, uint2& hi, uint2& lo) {
  half4_t h, l;
  h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
  l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]);
  l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
  hi = __builtin_bit_cast(uint2, h);
  lo = __builtin_bit_cast(uint2, l);
}
__device__ inline float4 split16_decode4(const uint2 hi, const uint2 lo) {
  const half4_t h = __builtin_bit_cast(half4_t, hi), l = __builtin_bit_cast(half4_t, lo);
  return make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]);
}
// byte offset of the hi halves of channels [c, c + 4) (c % 4 == 0) inside a SPLIT16 row; the lo halves sit 64 bytes on
__host__ __device__ inline int split16_off4(int c) { return (c >> 5) * 128 + (c & 31) * 2; }
constexpr int SPLIT16_LO = 64;   // byte distance from a channel's hi half to its lo half
// store / load 4 consecutive channels of a SPLIT16 row that starts at `row` (fp32-typed pointer, 4 bytes per channel)
__device__ inline void split16_store4(float* row, int c, const float4 v) {
  uint2 hi, lo;
  split16_encode4(v, hi, lo);
  char* p = reinterpret_cast<char*>(row) + split16_off4(c);
  *reinterpret_cast<uint2*>(p) = hi;
  *reinterpret_cast<uint2*>(p + SPLIT16_LO) = lo;
}
__device__ inline float4 split16_load4(const float* row, int c) {
  const char* p = reinterpret_cast<const char*>(row) + split16_off4(c);
  return split16_decode4(*reinterpret_cast<const uint2*>(p), *reinterpret_cast<const uint2*>(p + SPLIT16_LO));
}

// ---- SPLIT16 range guard.  The hi half of an activation at or above 65520 is inf (and the lo half x - inf): every
// epilogue that WRITES SPLIT16 rows tracks the largest magnitude it stores (one v_max per value) and raises words 0 and 3
// of `range` once a value reaches SPLIT16_LIMIT - before anything became inf.  Word 0 belongs to ONE forward (cleared, in
// stream order, when a split16 forward starts): the fp32-writing last layer of that forward answers with NaN rows
// (split16_poisoned).  Word 3 is sticky until eyoc_model_range_check reads it (EYOC_ERR_RANGE), so a caller that pipelines
// several forwards before it checks learns that one of them overflowed - and only the forwards that overflowed carry NaN
// rows.  Word 2 != 0 (the debug probe, eyoc_model_set_probe) also keeps the running maximum of |x| over all stored
// activations in word 1.
// The running maximum is kept as the BIT PATTERN of |x| compared as an unsigned integer (carried in a float register): finite
// magnitudes order like their bits, and inf / NaN - a weight or an input feature beyond fp16, an overflowed product - sort
// ABOVE every finite value, so they trip the guard too (v_max_f32 would drop a NaN and report a clean layer).
constexpr float SPLIT16_LIMIT = 6.0e4f;
__device__ inline float split16_merge(float a, float b) {              // both are magnitudes (sign bit clear) or NaN patterns
  const unsigned int x = __float_as_uint(a) & 0x7FFFFFFFu, y = __float_as_uint(b) & 0x7FFFFFFFu;
  return __uint_as_float(x > y ? x : y);
}
__device__ inline bool split16_over(float mx) { return (__float_as_uint(mx) & 0x7FFFFFFFu) >= __float_as_uint(SPLIT16_LIMIT); }
__device__ inline void split16_track(float& mx, const float4 v) {
  mx = split16_merge(split16_merge(split16_merge(mx, v.x), split16_merge(v.y, v.z)), v.w);
}
__device__ inline void split16_report(unsigned int* range, float mx) {
  if (!range) return;
  if (split16_over(mx)) { atomicOr(range, 1u); atomicOr(range + 3, 1u); }   // rare
  if (__builtin_nontemporal_load(range + 2)) atomicMax(range + 1, __float_as_uint(mx) & 0x7FFFFFFFu);   // probe mode only
}
__device__ inline bool split16_poisoned(const unsigned int* range) { return range && __builtin_nontemporal_load(range) != 0u; }

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits for
// every global load the wave has in flight - which turns a software pipeline whose loads are meant to land a stage
// later (gathers, weight pieces) into a blocking one: each stage then pays a full memory latency at its barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Tile builders hash a tile's distinct input rows into an LDS table (linear probing, atomicCAS) and number the occupied slots
// in slot order.  Which key of a probe cluster landed in which of its slots depends on the order the CAS loops ran in; the SET
// of occupied slots and the keys of a cluster (a maximal circular run of occupied slots) do not.  canonical_slot_id gives the
// r-th smallest key of a cluster the number of the cluster's r-th slot: the numbering - and with it WHICH PASS stages a row, i.e.
// the order in which a multi-pass tile's products are summed - is the same in every run (bit-reproducible outputs).
// hk: the table (-1 = empty), hpos[s]: occupied slots in front of slot s (valid for occupied s), s: an occupied slot.
template <int HS>
__device__ inline int canonical_slot_id(const int* hk, const unsigned short* hpos, int s) {
  const int kv = hk[s];
  int start = s, len = 1;
  while (len < HS && hk[(start - 1) & (HS - 1)] >= 0) { start = (start - 1) & (HS - 1); ++len; }
  int rank = 0;
  for (int q = start; q != s; q = (q + 1) & (HS - 1)) rank += hk[q] < kv;
  for (int q = (s + 1) & (HS - 1); hk[q] >= 0 && q != start; q = (q + 1) & (HS - 1)) rank += hk[q] < kv;
  return hpos[(start + rank) & (HS - 1)];
}

int launch_spconv(const SpconvArgs& a, hipStream_t st);
int launch_permute_rows(const float* in, const int32_t* perm, int n, int c, float* out, hipStream_t st);
int launch_spconv_st(const SpconvArgs& a, const unsigned char* local_dev, hipStream_t st);   // tile-local input stage (spconv_st.hip)
int launch_spconv_up(const SpconvArgs& a, const unsigned char* local_dev, hipStream_t st);   // transposed 3^3 / stride 2 (spconv_up.hip)
size_t local_rulebook_up_bytes(int n_out);
bool spconv_st_can_fuse_tail(const SpconvArgs& a);   // spconv_st.hip: would launch_spconv_st run the 256-row NH = 2 assembly kernel on this (64-channel) layer
int spconv_record_path(const SpconvArgs& a);   // spconv.hip: 1 / 2 / 3 = a tile-record kernel takes the layer (a.nbr is not read), 0 = a gathering kernel
int build_local_rulebook_up(const int32_t* nbr_dev, int K, int n_out, unsigned char* out_dev, int* overflow_dev, hipStream_t st);
// class-major transposed kernel (spconv_upc.hip): `ws` = upc_kept_bytes of header + tile order + records, built by build_upc
// (which also needs upc_scratch_bytes of scratch it does not keep)
size_t upc_kept_bytes(int n_out);
size_t upc_scratch_bytes(int n_out);
// coords_dev: the fine level's coordinates ([n_out][4], multiples of `stride`) or NULL (the class is then read off the table)
int build_upc(const int32_t* nbr_dev, const int32_t* coords_dev, int stride, int n_out, unsigned char* ws, unsigned char* scratch, hipStream_t st, bool compact = false);
const int* upc_overflow_ptr(const unsigned char* ws);
int upc_set_tile_rows(int odd_axes, int rows);   // rows per tile (128 .. 256, multiple of 16) of the classes with that many odd axes; this device
int launch_spconv_upc(const SpconvArgs& a, const unsigned char* ws, hipStream_t st);
int st_variants();   // spconv_st.hip: how many staged-kernel variants this build has (eyoc_spconv_select_st_kernel's range)
size_t local_rulebook_bytes(int n_out);
// group: 1 = the rows of a tile sorted by neighbour pattern (fewer non-empty MFMA blocks), 0 = in their own order (what
// conv1_bf_kernel needs: it finds a parent's entries by its local row); eyoc_maps_build passes its ctx's eyoc_spconv_st_group_rows switch
int build_local_rulebook(const int32_t* nbr_dev, int K, int n_out, unsigned char* out_dev, int* overflow_dev, hipStream_t st, int group = 1);
// The same records WITHOUT a stride-1 table in memory (round 6): every thread derives its row's 27 neighbours from the octree links
// and the coarser level's table (derive.h) inside the builder - the finest level's [27][n] table (413 MB on the 128-cloud bench batch)
// is then neither written nor read back.  up8 (optional): the compact transposed table [8][n_out] of the level, written on the way.
struct DeriveSrc {
  const int32_t* coords = nullptr;     // [n_out, 4] level coordinates
  const int32_t* parent = nullptr;     // [n_out] row of the coarser level
  const int32_t* children = nullptr;   // [nc, 8] rows of this level
  const int32_t* s1c = nullptr;        // [27][nc] stride-1 table of the coarser level
  int nc = 0, sh = 0;                  // coarser level's rows; log2 of this level's stride
  int32_t* up8 = nullptr;
};
// 128-row tiles (strided tables; the record layout of the 256-row tiles, quarters 0 and 1 only)
size_t local_rulebook128_bytes(int n_out);
int build_local_rulebook128(const int32_t* nbr_dev, int K, int n_out, unsigned char* out_dev, int* overflow_dev, hipStream_t st, int group = 1);
int launch_spconv_st128(const SpconvArgs& a, const unsigned char* local_dev, hipStream_t st);
int build_local_rulebook_derived(const DeriveSrc& src, int n_out, unsigned char* out_dev, int* overflow_dev, hipStream_t st, int group = 1);
int launch_spconv_rs(const SpconvArgs& a, hipStream_t st);     // row-stationary, SPLIT16 only (spconv_rs.hip)
// the network's 1x1 tail in one kernel (spconv_tail.hip): conv1_tr (96 -> 64, ReLU) -> final (64 -> 32, bias) -> row normalisation
bool tail_fusable(int cin1, int cmid, int cout);
int launch_tail_fused(const float* in, int ld_in, int n, const float* w1, const float* s1, const float* b1, int relu1, const float* w2,
                      const float* s2, const float* b2, int l2norm, float* out, int ld_out, const int32_t* out_perm, unsigned int* range,
                      hipStream_t st);
bool spconv_rs_fits(const SpconvArgs& a);
int launch_spconv_wave(const SpconvArgs& a, hipStream_t st);   // wave-private tiling (spconv_wave.hip)

// first convolution: K = ks^3 offsets probed straight from the level-0 hash table (C_in is tiny)
struct Conv1Args {
  const int32_t* coords;  // [n,4]
  int n;
  HashTable table;
  int ks;
  const float* in;        // [n, cin]
  int cin;
  const float* w;         // [K][cin][cout] with the BN scale folded in
  const float* bias;      // [cout]
  int cout;               // 32 <= cout <= 128, multiple of 32
  const float* wscale = nullptr;   // MFMA kernels: device {2^sh, 2^-sh}, the power of two their fp16 weight halves are lifted by (NULL: 2^8)
  float* out;             // rows of ld_out floats
  int ld_out;
  int out_split = 0;      // write SPLIT16 rows (see above) instead of fp32 rows
  unsigned int* range = nullptr;      // SPLIT16 range guard words (split16_guard) or NULL
  const int32_t* in_perm = nullptr;   // input row i is read from row in_perm[i] of `in` (the network input in the caller's order)
  // octree links (level 0 <-> 1) and the level-1 stride-1 table: with them a 3^3 / 5^3 window is read
  // from the 27 coarse blocks around the parent without any hash probe; NULL -> probe the hash table
  const int32_t* parent;    // [n]
  const int32_t* children;  // [nc][8]
  const int32_t* s1c;       // [27][nc]
  int nc;
  // tile-local rulebooks of the level-1 stride-1 table (build_local_rulebook; Z-ordered maps) or NULL: with them the first
  // convolution stages the child features of a 256-parent tile's neighbourhood in LDS once (conv1_bf_kernel)
  const unsigned char* local1 = nullptr;
  const eyoc_ctx* ctx = nullptr;      // whose switches decide the kernel (eyoc_spconv_select_conv1_kernel); NULL: the defaults
};
// layout of a local rulebook record (spconv_st.hip owns it; conv1_bf_kernel in spconv.hip reads the row list and the entries)
constexpr int ST_TILE = 256, ST_UMAX = 639, ST_UCAP = 1280, ST_NPASS = 2, ST_LOC_OFF = 16 + ST_UCAP * 4, ST_INV_OFF = 33152, ST_LR_BYTES = 33408;
int launch_conv1(const Conv1Args& a, hipStream_t st);
bool conv1_walks_octree(const Conv1Args& a);   // false: launch_conv1 will probe a.table (it must be built)

}  // namespace eyoc
