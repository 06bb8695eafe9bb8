// Shared internals of libeyoc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/eyoc_hip.h"

namespace eyoc {

void set_error(const char* fmt, ...);

#define EYOC_CHECK_HIP(expr)                                                               \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      eyoc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                      __LINE__);                                                           \
      return EYOC_ERR_HIP;                                                                 \
    }                                                                                      \
  } while (0)

#define EYOC_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      eyoc::set_error(__VA_ARGS__);    \
      return (code);                   \
    }                                  \
  } while (0)

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// carve typed arrays out of a caller-owned workspace
struct Carver {
  char* base;
  size_t off = 0, cap;
  Carver(void* p, size_t bytes) : base((char*)p), cap(bytes) {}
  template <typename T>
  T* take(size_t count) {
    size_t o = align_up(off);
    off = o + count * sizeof(T);
    return (T*)(base ? base + o : nullptr);
  }
  bool ok() const { return off <= cap; }
};

}  // namespace eyoc

struct eyoc_ctx {
  int device = 0;
  // grow-only scratch owned by the ctx (knn packed minima, ransac survivor lists, reductions)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // pinned host staging for small read-backs
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  // grow-only device scratch shared by kNN / label kernels / RANSAC / maps_info.  ONE user at a time: a call on another
  // stream first waits (event) for everything the previous user's stream was given, so two torch streams cannot race on it
  int ensure_scratch(size_t bytes, hipStream_t st);
  hipStream_t scratch_owner = nullptr;
  bool scratch_owned = false;
  hipEvent_t scratch_ev = nullptr;
  // side streams for batches of small independent problems (eyoc_sc2pcr_batched): forked from and joined to the
  // caller's stream with events, created on first use
  static constexpr int POOL = 8;
  hipStream_t pool[POOL] = {};
  hipEvent_t pool_done[POOL] = {};
  hipEvent_t pool_fork = nullptr;
  bool pool_ready = false;
  int ensure_pool();
  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remembered per ctx, not per process
  bool ransac_attr_set = false, sc2_attr_set = false;
  bool up_attr_set[2] = {false, false};   // spconv_up_kernel<64> / <32>
  // Z-order sort of eyoc_maps_build: how many bits the Morton key needs is SPECULATED from the previous build of this ctx
  // (coordinates within +-2^zorder_kbits, batch index below 2^zorder_bbits); a build whose rows do not fit redoes its sort with
  // the full 18 + 10 bits.  The permutation is the same either way (the bias is order-preserving); only the number of radix passes differs.
  int zorder_kbits = 17, zorder_bbits = 10;
  // SC2-PCR diagnostics (eyoc_sc2pcr_set_shortlist_cap / eyoc_sc2pcr_set_dense_threshold): per ctx, not per process
  int sc2_list_cap = 1024, sc2_dense_x = 0;
  int sc2_legacy = 0;      // eyoc_sc2pcr_select_kernels (sc2pcr.hip): bits select the round-5 forms of three kernels, for A/B tests
  // kernel-selection and tiling switches (tests, diagnostics, bench flags).  Per ctx since round 5: they were file-scope statics, so two
  // models in one process - or a test that forgot to restore one - shared kernel selection.  Every entry point reads them from the
  // ctx it was given; the setters (eyoc_spconv_select_*, eyoc_maps_*, eyoc_knn_prefilter, eyoc_ransac_*, eyoc_model_fuse_tail) take the ctx
  struct Knobs {
    int maps_order_min_rows = 65536;   // eyoc_maps_order_min_rows
    int maps_window_shift = 18;        // eyoc_maps_order_window_shift: window of the tiling orders, log2 rows (measured: 2^17-2^18; 2^12, 2^14 lose)
    int maps_s1_order = 1, maps_down_order = 0;   // eyoc_maps_select_orders: tiling orders of the stride-1 / strided tables
    int maps_internal_order = -1;      // eyoc_maps_internal_order: -1 automatic (Z-order from 8192 rows), 0 caller's order, 1 Z-order
    int maps_lazy_tables = 1, maps_fused_levels = 1;          // eyoc_maps_lazy_tables: big Z-ordered batches skip the tables only their record builders read (coordmap.hip)
    int knn_prefilter = 1;             // eyoc_knn_prefilter
    int fuse_tail = 2;                 // eyoc_model_fuse_tail: 2 the 1x1 tail in the epilogue of the last staged layer where that layer allows it (spconv_st.hip TAILF), 1 in one kernel of its own (spconv_tail.hip), 0 two launches
    int spconv_kernel = -1;            // eyoc_spconv_select_kernel: -1 automatic, 0 workgroup-tiled, 1 wave-private
    int split16_kernel = 1;            // eyoc_spconv_select_split16_kernel: 1 per layer, 0 always the wave-private kernel, 2 always the row-stationary one
    int up_kernel = 2;                 // eyoc_spconv_select_up_kernel: 0 gathering kernels, 1 spconv_up.hip (Morton tiles), 2 spconv_upc.hip (class-major tiles)
    int s1_wide = 1;                   // (diagnostics, eyoc_spconv_select_down_kernel(4 / 5)): stride-1 layers with >= 128 channels on 128-row x 128-channel workgroups
    int st128_wide = 1;                // (diagnostics, eyoc_spconv_select_down_kernel(2 / 3)): 128-row strided tiles of >= 128-channel layers as 128-channel workgroups
    int down_kernel = 1;               // eyoc_spconv_select_down_kernel: 1 staged on 128-row tiles (Z-ordered maps of >= upc_min_rows rows), 0 the gathering kernel
    int upc_min_rows = 1 << 17;        // eyoc_spconv_upc_min_rows: maps with fewer level-0 rows keep spconv_up.hip under mode 2
    int conv1_kernel = 1;              // eyoc_spconv_select_conv1_kernel: 1 conv1_bf_kernel (block feature vectors), 0 conv1_mfma_kernel, 2 the exact-fp32 octree walker
    int st_group = 1;                  // eyoc_spconv_st_group_rows: the staged kernel's tiles sorted by neighbour pattern
    int st_variant = 1;                // eyoc_spconv_select_st_kernel
    int st_split_below = 1024;         // eyoc_spconv_st_split_below
    int st_cg_local = 1;               // (diagnostics, eyoc_spconv_st_ksplit(2 / 3) = off / on): channel groups of a tile on one XCD when the layer's weights fit an L2 beside the stream
    int st_ksplit = 1;                 // eyoc_spconv_st_ksplit
    int ransac_store = 1 << 20;        // eyoc_ransac_transform_store
    int ransac_prune = 2;              // eyoc_ransac_select_pruning: 0 full sweeps, 1 reference pruning, 2 + survivors stop counting once they cannot reach the largest count (round 6)
  } knobs;
};
// the switches of a call: the ctx's, or the defaults where an internal launcher was handed no ctx
inline const eyoc_ctx::Knobs& knobs_of(const eyoc_ctx* ctx) {
  static const eyoc_ctx::Knobs defaults;
  return ctx ? ctx->knobs : defaults;
}

// ---------------------------------------------------------------------------------------------
// coordinate keys and the open-addressing hash shared by coordmap.hip and spconv.hip (conv1)
// ---------------------------------------------------------------------------------------------
namespace eyoc {

constexpr unsigned long long KEY_EMPTY = 0xFFFFFFFFFFFFFFFFull;
constexpr int COORD_BIAS = 1 << 17;   // |coordinate| < 2^17
constexpr int VAL_UNSET = 0x7F7F7F7F;  // hipMemset(0x7F)

struct HashTable {
  unsigned long long* keys;
  int* vals;
  unsigned int mask;  // capacity - 1 (capacity is a power of two)
};

__host__ __device__ inline unsigned long long pack_key(int b, int x, int y, int z) {
  return ((unsigned long long)(unsigned)b << 54) |
         ((unsigned long long)((unsigned)(x + COORD_BIAS) & 0x3FFFFu) << 36) |
         ((unsigned long long)((unsigned)(y + COORD_BIAS) & 0x3FFFFu) << 18) |
         ((unsigned long long)((unsigned)(z + COORD_BIAS) & 0x3FFFFu));
}

__host__ __device__ inline unsigned int hash_key(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (unsigned int)k;
}

__device__ inline int hash_lookup(const HashTable& t, unsigned long long key) {
  unsigned int s = hash_key(key) & t.mask;
  for (unsigned int probes = 0; probes <= t.mask; ++probes) {   // bounded: a table is never full (load <= 1/2)
    unsigned long long k = t.keys[s];
    if (k == key) return t.vals[s];
    if (k == KEY_EMPTY) return -1;
    s = (s + 1) & t.mask;
  }
  return -1;
}

}  // namespace eyoc

struct eyoc_maps {
  int n_levels = 0;
  int rows[EYOC_MAX_LEVELS] = {0, 0, 0, 0};
  int32_t* coords[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};  // [rows,4]
  eyoc::HashTable table[EYOC_MAX_LEVELS];
  int32_t* nbr_s1[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};    // [27][rows[l]]
  int32_t* nbr_down[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};  // [l]: [27][rows[l+1]]
  int32_t* nbr_up[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};    // [l]: [27][rows[l]]
  // octree links between consecutive levels: parent[l][row of level l] = row of level l+1;
  // children[l][row of level l+1][8] = rows of level l (slot = x-bit | y-bit << 1 | z-bit << 2) or -1
  int32_t* parent[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};
  int32_t* children[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};
  // perm_up[l]: the rows of level l ordered by their transposed-map pattern (which coarse blocks exist around
  // them), stable.  A transposed convolution that tiles its output rows in THIS order finds 2-3 occupied
  // offsets per tile instead of ~24 (spconv_wave.hip); the result does not depend on the order.
  int32_t* perm_up[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};
  // perm_s1[l]: same idea for the stride-1 convolutions of level l (rows grouped by a coarse key of their
  // neighbour pattern); NULL = natural order
  int32_t* perm_s1[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};
  // perm_down[l]: rows of level l+1 (outputs of the strided convolution l -> l+1) grouped by which of their own 8
  // children exist (those are 8 of the 27 offsets of the strided map)
  int32_t* perm_down[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};
  // Internal row order.  For large batches level 0 is stored in Z-order (Morton order of (batch, x, y, z)) instead of
  // the caller's order: 64 consecutive rows are then a compact blob of voxels, which is what the tile-local input stage
  // of the sparse convolution needs (spconv_st.hip).  row_perm[i] = caller's row of internal row i (NULL: identity).
  // Only the network input (read through row_perm by the first convolution) and output (scattered through it by the
  // last) are in the caller's order; coarser levels follow from level 0 as before (first occurrence = Z-order too).
  int32_t* row_perm = nullptr;
  // per-tile local rulebooks of the stride-1 tables (spconv_st.hip), built when the rows are in Z-order; NULL otherwise
  unsigned char* local_s1[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};
  unsigned char* local_up[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};   // transposed tables (outputs at level l)
  unsigned char* local_s1w[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};   // stride-1 tables in 128-row tiles as well (levels whose layers have >= 128 channels: 128-row x 128-channel workgroups)
  unsigned char* local_down[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};  // strided tables (outputs at level l + 1) in 128-row tiles (spconv_st.hip launch_spconv_st128)
  unsigned char* local_upc[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};  // ... partitioned by parity class (spconv_upc.hip: header, tile order, records)
  bool table0_built = false;   // table[0] has its memory reserved but is only filled on demand (maps_build_table0)
  // Lazy tables (round 6; Z-ordered batches with class-major transposed records, eyoc_ctx::Knobs::maps_lazy_tables): the finest level's
  // stride-1 table and every transposed table are read by their tile-record builders only, so the build derives the records straight
  // from the octree (spconv_st.hip build_local_rulebook_derived) / from the compact [8][n] table up8[l] (derive.h) and leaves the
  // [27][n] tables UNWRITTEN (their memory stays reserved).  eyoc::maps_ensure_table fills one on first use: the accessors
  // (eyoc_maps_table, _copy_table, _info), a forward whose layer falls back to a gathering kernel, a build whose records overflowed.
  bool s1_ready[EYOC_MAX_LEVELS] = {true, true, true, true};
  bool up_ready[EYOC_MAX_LEVELS] = {true, true, true, true};
  int32_t* up8[EYOC_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};
};

namespace eyoc {
int maps_build_table0(eyoc_maps* maps, hipStream_t st);
// fills a lazily skipped [27][n] table (kind: EYOC_MAP_S1 / EYOC_MAP_UP; anything else is always there) on `st`; no-op when ready
int maps_ensure_table(eyoc_maps* maps, int kind, int level, hipStream_t st);
size_t sort_rows64_tmp_bytes(int n);
int sort_rows_by_key64(void* tmp, size_t tmp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out, const int* vals_in,
                       int* vals_out, int n, int bits, hipStream_t st);
size_t sort_rows_tmp_bytes(int n, int bits);
int sort_rows_by_key(void* tmp, size_t tmp_bytes, const unsigned int* keys_in, unsigned int* keys_out, const int* vals_in,
                     int* vals_out, int n, int bits, hipStream_t st);
}  // namespace eyoc
