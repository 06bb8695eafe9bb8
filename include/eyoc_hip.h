/*
 * eyoc_hip.h - C ABI of libeyoc_hip.so, the MI355X (gfx950) implementation of EYOC's registration
 * hot path.  The reference (liuQuan98/EYOC) has no FFI of its own for this path: its Python calls
 * land in MinkowskiEngine / PyTorch / Open3D.  Every entry point below names the reference call
 * site it stands in for (paths relative to the reference repository root).
 *
 * Conventions
 *   - every function returns 0 on success and a negative eyoc_status on failure;
 *     eyoc_last_error() returns a thread-local message for the last failure on this thread;
 *   - "dev" pointers are device (HBM) pointers owned by the caller; the library never frees them;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *     unless stated (eyoc_maps_build does: it must learn the data-dependent level sizes);
 *   - one eyoc_ctx per (process, device); a ctx is not re-entrant (one stream at a time).
 *   - every tensor that crosses this boundary is fp32 (the reference's dtype) and fp32 accumulates every sum; the
 *     sparse-convolution PRODUCTS inside eyoc_model_forward run either as fp32 MFMAs or - "split16", automatic for
 *     batches of >= 8192 rows - as three fp16 MFMAs on hi/lo-split operands (22-bit significands, range-guarded:
 *     eyoc_model_set_math / eyoc_model_range_check); indices are int32 on the device side and int64 where the
 *     reference hands int64 to its callers.
 */
#ifndef EYOC_HIP_H
#define EYOC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct eyoc_ctx eyoc_ctx;
typedef struct eyoc_maps eyoc_maps;
typedef struct eyoc_model eyoc_model;

typedef enum {
  EYOC_OK = 0,
  EYOC_ERR_INVALID = -1,     /* bad argument / unsupported shape            */
  EYOC_ERR_HIP = -2,         /* a HIP runtime call failed                   */
  EYOC_ERR_WORKSPACE = -3,   /* caller-provided workspace too small         */
  EYOC_ERR_DUPLICATE = -4,   /* duplicate coordinates in a sparse tensor    */
  EYOC_ERR_RANGE = -5        /* coordinate / batch index out of key range; split16 activation out of fp16 range */
} eyoc_status;

#define EYOC_MAX_LEVELS 4
/* 111 (round 6): eyoc_model_desc ends with `expanded` (ResUNetExpanded / ResUNetExpBN2C run through eyoc_model_forward too).
 * 110 (round 6): the kernel-selection setters and eyoc_ransac_workspace_bytes take the ctx first (round 5), eyoc_maps_gather_window
 * refuses Z-ordered maps again and eyoc_maps_gather_window_internal exists, eyoc_model_workspace_bytes depends on the maps' size class */
#define EYOC_VERSION 111

/* ------------------------------------------------------------------------------------------------
 * context
 * --------------------------------------------------------------------------------------------- */
int eyoc_version(void);
const char* eyoc_last_error(void);
int eyoc_create(int device, eyoc_ctx** out);
int eyoc_destroy(eyoc_ctx* ctx);
/* The kernel-selection and tiling switches declared further down (eyoc_maps_order_*, eyoc_maps_select_orders, eyoc_maps_internal_order,
 * eyoc_spconv_select_*, eyoc_spconv_st_*, eyoc_spconv_upc_min_rows, eyoc_model_fuse_tail, eyoc_knn_prefilter, eyoc_ransac_select_pruning,
 * eyoc_ransac_transform_store) are state of the ctx they are given: every entry point reads them from ITS ctx (file-scope statics until
 * round 4).  They exist for parity tests and profiling; production code never calls them.  Each returns the previous value (-1 for a
 * NULL ctx) and only queries when the argument is out of range. */

/* ------------------------------------------------------------------------------------------------
 * coordinate maps + rulebooks
 *   replaces: ME.SparseTensor(features, coordinates=) and the coordinate manager it creates
 *             (scripts/test_kitti.py:143-148, util/transform_estimation.py:128-131), and the kernel
 *             maps MinkowskiConvolution / MinkowskiConvolutionTranspose build on first use
 *             (model/resunet.py:31-116, model/residual_block.py:23-33).
 *   coords: int32 [n,4] = (batch, x, y, z), unique rows; |x|,|y|,|z| < 2^17, 0 <= batch < 1024.
 *   Builds the 4 levels (tensor stride 1,2,4,8) and, per level, output-stationary neighbour tables
 *   nbr[27][n_out] (int32 input row or -1): stride-1 (EYOC_MAP_S1), strided ts->2ts
 *   (EYOC_MAP_DOWN, indexed by the fine level) and transposed 2ts->ts (EYOC_MAP_UP, indexed by the
 *   fine level).  Kernel offsets enumerate x fastest.  Row order of level 0 is the input order;
 *   coarser levels are ordered by first occurrence.
 *   The workspace (eyoc_maps_workspace_bytes(n) bytes, 256-byte aligned) is owned by the caller and
 *   must outlive the maps object.  This call synchronises `stream` (3 small read-backs).
 * --------------------------------------------------------------------------------------------- */
typedef enum { EYOC_MAP_S1 = 0, EYOC_MAP_DOWN = 1, EYOC_MAP_UP = 2 } eyoc_map_kind;

typedef struct {
  int32_t n_levels;
  int32_t rows[EYOC_MAX_LEVELS];        /* N1, N2, N4, N8                                   */
  int64_t pairs_s1[EYOC_MAX_LEVELS];    /* valid entries of each stride-1 table             */
  int64_t pairs_down[EYOC_MAX_LEVELS];  /* [l] = table level l -> l+1 (l < n_levels-1)      */
  int64_t pairs_up[EYOC_MAX_LEVELS];    /* [l] = table level l+1 -> l                       */
  int64_t pairs_conv1;                  /* valid (row, offset) pairs of the first conv      */
} eyoc_maps_info_t;

size_t eyoc_maps_workspace_bytes(int n_rows);
int eyoc_maps_build(eyoc_ctx* ctx, const int32_t* coords_dev, int n_rows, void* workspace_dev,
                    size_t workspace_bytes, void* stream, eyoc_maps** out);
/* eyoc_maps_build keeps the CALLER's row order (the accessors below return level coordinates and tables in the caller's
 * rows).  The same with the internal row order chosen by the caller: -1 automatic (Z-order from 8192 rows: what
 * eyoc_model_forward is fastest on), 0 the caller's order (the level coordinates and tables the accessors below return
 * are then in the caller's rows), 1 Z-order.  eyoc_maps_internal_order(ctx, 0 / 1) overrides it for every build of that ctx. */
int eyoc_maps_build_ordered(eyoc_ctx* ctx, const int32_t* coords_dev, int n_rows, void* workspace_dev,
                            size_t workspace_bytes, void* stream, int order, eyoc_maps** out);
int eyoc_maps_free(eyoc_maps* maps);
/* Row order the transposed convolutions tile their outputs in: the rows of `level` (a fine level, 0 <= level <
 * n_levels-1) sorted, stably, by the pattern of coarse blocks their transposed map reaches.  Purely a
 * scheduling aid (tiles whose rows share their occupied kernel offsets) - results do not depend on it.
 * out_dev: int32 [rows[level]]. */
int eyoc_maps_copy_up_order(const eyoc_maps* maps, int level, int32_t* out_dev, void* stream);
/* Levels with fewer rows than this keep the natural order (the sorts only pay off for the large-batch kernel;
 * default 65536).  Per ctx; min_rows < 0 only queries.  Returns the previous value.  For tests / profiling. */
int eyoc_maps_order_min_rows(eyoc_ctx* ctx, int min_rows);
/* Z-ordered maps sort their tiling orders inside windows of 2^shift consecutive rows (default 18; the window's rows and
 * their neighbours stay cache-resident while its pattern runs are walked).  Per ctx; shift < 0 only queries.
 * Returns the previous value.  For tests / profiling. */
int eyoc_maps_order_window_shift(eyoc_ctx* ctx, int shift);
/* Which tables get a pattern-sorted tiling order at all: s1 = the stride-1 tables (default 1), down = the strided tables of
 * Z-ordered maps (default 0: natural order wins there).  0 / 1 set, anything else leaves the switch alone.  Returns the
 * previous state (s1 | down << 1).  Per ctx, read when maps are built; for tests / profiling. */
int eyoc_maps_select_orders(eyoc_ctx* ctx, int s1, int down);
/* Internal row order.  From 8192 rows on (mode -1, the default) the maps store level 0 in Z-order (Morton order of
 * (batch, x, y, z)) instead of the caller's order, so that 64 consecutive rows are a compact blob of voxels - what
 * the tile-local input stage of the sparse convolution needs.  eyoc_maps_coords / _table then describe the INTERNAL
 * rows; eyoc_maps_row_order returns the device array perm[i] = caller's row of internal row i (NULL: the caller's order
 * was kept).  eyoc_model_forward reads its input and writes its output in the caller's order either way.
 * eyoc_maps_internal_order(mode): -1 automatic, 0 always the caller's order, 1 always Z-order; returns the previous
 * mode + 2; per ctx, for tests. */
int eyoc_maps_internal_order(eyoc_ctx* ctx, int mode);
/* Lazy tables (round 6, default on): a Z-ordered build of >= eyoc_spconv_upc_min_rows rows derives the tile records of the finest
 * level's stride-1 table and of the transposed tables straight from the octree links and leaves those [27][n] tables unwritten (their
 * only readers on the hot path were the record builders: 0.8 GB per 128-cloud batch).  They are filled on first use - by
 * eyoc_maps_table / _copy_table / _info, by a forward whose layer runs a gathering kernel, by a build whose records overflowed - so
 * nothing a caller can observe changes.  eyoc_maps_lazy_tables(ctx, 0) builds every table eagerly again; returns the previous setting. */
int eyoc_maps_lazy_tables(eyoc_ctx* ctx, int on);
/* EYOC_VERSION >= 111.  Z-ordered builds make the three coarser levels (coordinates, parent and child links) in two launches over the
 * sorted level-0 rows (default, round 6) instead of four short dependent launches per level; the arrays are the same bit for bit.
 * eyoc_maps_fused_levels(ctx, 0) brings the per-level kernels back (tests compare); returns the previous setting. */
int eyoc_maps_fused_levels(eyoc_ctx* ctx, int on);
/* Strided convolutions of a split16 forward on Z-ordered maps of >= eyoc_spconv_upc_min_rows rows: 1 (default, round 6) the staged kernel
 * on 128-row output tiles (tile records built by eyoc_maps_build), 0 the gathering kernel; returns the previous setting. */
int eyoc_spconv_select_down_kernel(eyoc_ctx* ctx, int mode);
const int32_t* eyoc_maps_row_order(const eyoc_maps* maps);
/* stream-ordered copy of the same array (the identity when the caller's order was kept); out_dev: int32 [rows[0]] */
int eyoc_maps_copy_row_order(const eyoc_maps* maps, int32_t* out_dev, void* stream);
int eyoc_maps_rows(const eyoc_maps* maps, int level);
/* device pointers into the workspace; valid while the maps object lives */
const int32_t* eyoc_maps_coords(const eyoc_maps* maps, int level);             /* [rows,4]        */
const int32_t* eyoc_maps_table(const eyoc_maps* maps, int kind, int level);    /* [27][n_out]; a lazily skipped table is filled first (NULL stream, waited for) */
/* stream-ordered device-to-device copies of the same arrays into caller-owned buffers */
int eyoc_maps_copy_coords(const eyoc_maps* maps, int level, int32_t* out_dev, void* stream);
int eyoc_maps_copy_table(const eyoc_maps* maps, int kind, int level, int32_t* out_dev, void* stream);
/* counts valid pairs (one reduction per table + a sync); conv1_kernel_size 0 skips pairs_conv1 */
int eyoc_maps_info(eyoc_ctx* ctx, const eyoc_maps* maps, int conv1_kernel_size, void* stream,
                   eyoc_maps_info_t* info);

/* ------------------------------------------------------------------------------------------------
 * voxeliser (the step immediately before the path; SURVEY.md 8f "next" row 1)
 *   replaces: ME.utils.sparse_quantize(xyz / voxel_size, return_index=True) followed by
 *             floor(xyz[sel] / voxel_size).int()   (lib/data_loaders.py:940-943,969-979; util/misc.py:80-84)
 *   xyz f32 rows of `stride` floats (3 = xyz, 4 = KITTI .bin xyzr) -> sel int32 [n_out] = index of the
 *   first point of every occupied voxel, ascending; coords int32 [n_out,4] = (batch_index, floor(p / v)).
 *   Outputs need room for n rows.  Synchronises `stream` once (to return n_out).
 * --------------------------------------------------------------------------------------------- */
size_t eyoc_voxelize_workspace_bytes(int n_points);
int eyoc_voxelize(eyoc_ctx* ctx, const float* xyz_dev, int n_points, int stride, float voxel_size, int batch_index,
                  int32_t* sel_dev, int32_t* coords_dev, int* n_out, void* workspace_dev, size_t workspace_bytes,
                  void* stream);

/* ------------------------------------------------------------------------------------------------
 * one sparse convolution layer (unit tests, profiling)
 *   replaces: one MinkowskiConvolution / MinkowskiConvolutionTranspose forward with the batch norm
 *             that follows it folded in, plus the residual add / ReLU / concat write around it
 *             (model/residual_block.py:37-53, model/resunet.py:142-186).
 *   out[o, :] = act( sum_k in[nbr[k][o], :] @ W[k] + bias (+ res[o, :]) ),  nbr == NULL: K must be
 *   1 and the map is the identity (1x1 convolution).  cin % 32 == 0, cout in {32,64,128,256}; every entry of
 *   nbr must be < 2^24 (the model forward checks its level sizes; ~550 clouds of 30k voxels in one batch).
 *   Weights must be in the packed layout produced by eyoc_spconv_pack_weights (host side).
 * --------------------------------------------------------------------------------------------- */
size_t eyoc_spconv_packed_floats(int K, int cin, int cout);
int eyoc_spconv_pack_weights(const float* w_host /*[K,cin,cout]*/, const float* scale_host /*[cout]|NULL*/,
                             int K, int cin, int cout, float* packed_host);
int eyoc_spconv(eyoc_ctx* ctx, const int32_t* nbr_dev, int K, int n_out, const float* in_dev,
                int ld_in, int cin, const float* wpacked_dev, int cout, const float* bias_dev,
                const float* res_dev, int ld_res, int relu, float* out_dev, int ld_out, void* stream);
/* The same layer on the fp16 matrix pipe at fp32 accuracy ("split16", eyoc_amd/csrc/spconv_wave.hip): math = 1 expects
 * `in_dev` / `res_dev` rows in the SPLIT16 format (eyoc_split16_encode: per 32 channels 64 B of fp16 hi halves + 64 B
 * of fp16 lo halves, x = hi + lo to 2^-22 - the same 4 bytes per channel, so leading dimensions are unchanged; column
 * offsets and widths must be multiples of 32 channels) and
 * weights packed by eyoc_spconv_pack_weights_split16, which also returns the scalar the kernel multiplies its sums
 * with (upload it and pass it as out_scale_dev).  out_split = 1 writes SPLIT16 rows, 0 fp32 rows.  math = 0 is
 * eyoc_spconv.  n_in = rows of the input tensor (0 = unknown: every entry of nbr is assumed < 2^24).  Two kernels
 * implement split16 layers - wave-private (spconv_wave.hip, pair compaction + LDS accumulators) and row-stationary
 * (spconv_rs.hip, register accumulators + zero operands, weights shared through LDS); the launcher picks per layer,
 * eyoc_spconv_select_split16_kernel forces one (0 wave-private, 2 row-stationary, 1 automatic; returns the previous
 * mode; per ctx, for tests and profiling). */
int eyoc_spconv_select_split16_kernel(eyoc_ctx* ctx, int mode);
/* Staged kernel for the transposed 3^3 / stride-2 convolutions on Z-ordered maps (spconv_up.hip: tile rows sorted by
 * parity class, only occupied (16-row group, offset) blocks multiplied): 1 on (default: as fast as the row-stationary
 * kernel in windowed pattern order, 18 GB less HBM traffic per 128-cloud forward), 0 off (gathering kernels), 2 (default since
 * round 4) = the class-major kernel below for batches and this one for small inputs; other values only query.  Returns the
 * previous state.  Per ctx, read when maps are built; for tests and profiling. */
int eyoc_spconv_select_up_kernel(eyoc_ctx* ctx, int on);
/* The same layers in CLASS-MAJOR order (spconv_upc.hip; eyoc_spconv_select_up_kernel(2)): the fine rows are partitioned by parity
 * class (8 classes of 1-8 offsets), a tile is 256 rows of one class and runs the staged kernel's assembly loop over that class's
 * offsets only.  Standalone entry points (the model uses the same kernels through its maps): workspace size for a table with
 * n_out fine rows; build (ws_dev 256-byte aligned; info_host = NULL or 19 ints {tiles, first tile of class 0..8, rows of class
 * 0..7, tiles with more distinct coarse rows than two stage passes hold - the kernel must not be used then}, which
 * synchronises the stream); the layer (split16 rows in, split16 or fp32 rows out, no residual).
 *   replaces: ME.MinkowskiConvolutionTranspose(kernel_size=3, stride=2) of model/resunet.py:83-116 */
/* eyoc_spconv_select_up_kernel(2) (the default) uses the class-major kernel for maps with at least this many level-0 rows (default
 * 2^17; the partition's extra launches cost a single 60 k-voxel pair more than the kernel saves) and spconv_up.hip below; a
 * negative argument only queries; returns the previous value.  Per ctx, read when maps are built. */
int eyoc_spconv_upc_min_rows(eyoc_ctx* ctx, int rows);
/* Rows per tile of the classes with `odd_axes` (0..3) odd axes: 128..256, a multiple of 16 (default 256, and 192 for the
 * 8-offset class, whose 256-row tiles would need two stage passes).  Current device; read when records are built; for
 * measurements (workspace sizes assume >= 128). */
int eyoc_spconv_upc_tile_rows(int odd_axes, int rows);
size_t eyoc_spconv_upc_bytes(int n_out);
int eyoc_spconv_upc_build(eyoc_ctx* ctx, const int32_t* nbr_dev, int n_out, void* ws_dev, int32_t* info_host, void* stream);
int eyoc_spconv_upc(eyoc_ctx* ctx, const int32_t* nbr_dev, const void* ws_dev, int n_out, int n_in, const float* in_dev, int ld_in,
                    int cin, const float* wpacked_dev, int cout, const float* bias_dev, int relu, float* out_dev, int ld_out,
                    int out_split, const float* out_scale_dev, void* stream);
/* First convolution (C_in = 1, 32 output channels) of split16 forwards on Z-ordered maps: 1 (default) = conv1_bf_kernel - the block
 * feature vectors (8 child features per level-1 row) of a 256-parent tile's neighbourhood staged in LDS through the level-1 tile
 * rulebook, the tile's fine rows grouped by parity class, a K = 27 product over blocks; 0 = conv1_mfma_kernel, which probes
 * the octree per fine row; 2 = the exact-fp32 octree walker (conv1_tree_kernel) even in front of split16 consumers.  Other values only query.  Returns the previous state; per ctx, for tests and profiling. */
int eyoc_spconv_select_conv1_kernel(eyoc_ctx* ctx, int on);
/* Stride-1 (3^3) split16 layers with a tile-local input stage (spconv_st.hip): per 256-row tile the distinct input rows are
 * copied to LDS once per 32-channel block and all 27 offsets run from there.  Needs the table's per-tile "local
 * rulebooks" (built once per table; *overflow_dev counts 256-row tiles with more than 1278 distinct input rows - the staged
 * kernel must not be used when it is non-zero; rows in Morton order never overflow).  eyoc_model_forward builds and
 * uses them itself; these entry points exist for tests and profiling.
 * eyoc_spconv_select_st_kernel picks the implementation of the offset loop: 1 (default) = hand-scheduled gfx950 assembly
 * with scalar branches around the MFMAs of empty (16-row chunk, offset) blocks; 2 = the same without the branches;
 * 0 = the compiler-scheduled C++ loop.  Any other value only queries.  Returns the previous variant; per ctx, for tests and profiling. */
int eyoc_spconv_select_st_kernel(eyoc_ctx* ctx, int variant);
/* A layer with fewer than `workgroups` 64-output-channel workgroups (default 1024 = two rounds of the chip's 512 slots) runs
 * in 32-channel workgroups instead (twice as many, each half as long: single pairs and small batches).  0 = never (tests force
 * the wide kernels onto small clouds with it); negative only queries.  Returns the previous threshold; per ctx. */
int eyoc_spconv_st_split_below(eyoc_ctx* ctx, int workgroups);
/* Row grouping inside the 256-row tile records (default on): the builder sorts a tile's rows by their neighbour pattern so that the 16
 * rows of an MFMA chunk miss the same offsets - the staged loop skips (chunk, offset) blocks without a neighbour, and 0.81-0.93 of
 * them are non-empty in row order, 0.68-0.72 grouped.  Results are bit-identical either way (only the order of a tile's rows inside
 * its workgroup changes).  1 / 0 set, anything else only queries; returns the previous state; per ctx, read when records are built. */
int eyoc_spconv_st_group_rows(eyoc_ctx* ctx, int on);
/* Small inputs (a single pair: the level-3 layer is 72 workgroups walking 8 input blocks x 27 offsets each): eyoc_model_forward lets
 * the staged kernel split a tile's 32-channel input blocks over several workgroups (partial sums in the forward's workspace, added in
 * share order by a second launch: bit-reproducible).  1 on (default) / 0 off, anything else only queries; returns the
 * previous state; per ctx, for tests and profiling.  Results differ from the unsplit kernel by fp32 rounding (another
 * summation order). */
int eyoc_spconv_st_ksplit(eyoc_ctx* ctx, int on);
size_t eyoc_spconv_local_rulebook_bytes(int n_out);
int eyoc_spconv_build_local_rulebook(eyoc_ctx* ctx, const int32_t* nbr_dev, int K, int n_out, void* out_dev,
                                     int32_t* overflow_dev, void* stream);
int eyoc_spconv_staged(eyoc_ctx* ctx, const int32_t* nbr_dev, const void* local_dev, int n_out, int n_in, const float* in_dev,
                       int ld_in, int cin, const float* wpacked_dev, int cout, const float* bias_dev, const float* res_dev,
                       int ld_res, int relu, float* out_dev, int ld_out, int out_split, const float* out_scale_dev, void* stream);
int eyoc_spconv_pack_weights_split16(const float* w_host, const float* scale_host, int K, int cin, int cout,
                                     float* packed_host, float* out_scale_host);
int eyoc_spconv_ex(eyoc_ctx* ctx, const int32_t* nbr_dev, int K, int n_out, int n_in, const float* in_dev, int ld_in, int cin,
                   const float* wpacked_dev, int cout, const float* bias_dev, const float* res_dev, int ld_res, int relu,
                   float* out_dev, int ld_out, int math, int out_split, const float* out_scale_dev, void* stream);
int eyoc_split16_encode(eyoc_ctx* ctx, const float* in_dev, int n, int c, int ld_in, float* out_dev, int ld_out, void* stream);
int eyoc_split16_decode(eyoc_ctx* ctx, const float* in_dev, int n, int c, int ld_in, float* out_dev, int ld_out, void* stream);
/* The bare operator out[o] = sum_k in[nbr[k][o]] W[k] for the training path (forward and input gradient of autograd.sparse_conv;
 * lib/trainer.py:1655-1676 runs them through MinkowskiEngine): no epilogue, fp32 MFMAs, and for small inputs the launcher may split
 * the K offsets over 2-4 workgroups per row tile (scratch of the ctx; shares added in a fixed order) - the summation order then depends
 * on n_out, which eyoc_spconv never lets happen.  Same argument meaning as eyoc_spconv.  The split's shares live in the ctx's ONE
 * grow-only scratch, which kNN / label kernels / RANSAC of the same ctx use too: a call on another stream first waits (event) for
 * the previous user's stream, i.e. offset-split training layers and matching on a second stream of the same ctx serialise there -
 * give concurrent phases their own eyoc_ctx. */
int eyoc_spconv_sum(eyoc_ctx* ctx, const int32_t* nbr_dev, int K, int n_out, const float* in_dev, int ld_in, int cin,
                    const float* wpacked_dev, int cout, float* out_dev, int ld_out, void* stream);
/* Backward of one layer (the reference back-propagates through MinkowskiEngine, lib/trainer.py:1667).
 *   grad-input: dIn[i] = sum_k dOut[o] W[k]^T over the pairs (i -> o, k) is a sparse convolution over the TRANSPOSED
 *     rulebook - for a stride-1 table the same table with mirrored offsets (mirror = 1), for the strided (EYOC_MAP_DOWN)
 *     table the EYOC_MAP_UP table of the same level and vice versa (mirror = 0) - so it runs eyoc_spconv on weights
 *     packed by eyoc_spconv_pack_weights_transposed (C_in and C_out swap roles: the call has cin = C_out, cout = C_in).
 *   grad-weight: dW[k][ci][co] = sum over o with nbr[k][o] >= 0 of in[nbr[k][o]][ci] * dout[o][co], plain [K, cin, cout]
 *     layout, fp32 MFMA, deterministic (fixed reduction order).  C_in, C_out: multiples of 16; nbr == NULL: identity map. */
int eyoc_spconv_pack_weights_transposed(const float* w_host /*[K,cin,cout]*/, int K, int cin, int cout, int mirror,
                                        float* packed_host);
size_t eyoc_spconv_grad_weight_workspace_bytes(int K, int n_out, int cin, int cout);
int eyoc_spconv_grad_weight(eyoc_ctx* ctx, const int32_t* nbr_dev, int K, int n_out, const float* in_dev, int ld_in, int cin,
                            const float* dout_dev, int ld_dout, int cout, float* dw_dev, void* workspace_dev,
                            size_t workspace_bytes, void* stream);
/* Two decompositions implement the operator (workgroup-tiled: spconv.hip, wave-private: spconv_wave.hip);
 * by default the launcher picks by problem size.  mode -1 = automatic (default), 0 = workgroup-tiled,
 * 1 = wave-private.  Per ctx; meant for parity tests and profiling.  Returns the previous mode. */
int eyoc_spconv_select_kernel(eyoc_ctx* ctx, int mode);

/* ------------------------------------------------------------------------------------------------
 * ResUNet2 family (ResUNetBN2C in production)
 *   replaces: Model(in_channels, out_channels, bn_momentum=, conv1_kernel_size=, normalize_feature=)
 *             + load_state_dict + eval + __call__   (model/resunet.py:18-193,
 *             scripts/test_kitti.py:83-93,143-150).
 *   Layers are matched by MinkowskiEngine state_dict names: "conv1", "norm1", "block1.conv1",
 *   "block1.norm1", ... "conv1_tr", "final".  Batch norms are folded into the preceding
 *   convolution (eval mode, eps = bn_eps).
 *   The packed weights live in a caller-owned device blob of eyoc_model_blob_floats() floats so it
 *   can be broadcast between ranks: rank 0 passes `layers`, the others pass layers == NULL after
 *   receiving the blob.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t in_channels, out_channels, conv1_kernel_size, normalize_feature;
  int32_t channels[5];     /* CHANNELS    = [-, 32, 64, 128, 256] for BN2C (index 0 unused) */
  int32_t tr_channels[5];  /* TR_CHANNELS = [-, 64, 64, 64, 128]  for BN2C                   */
  float bn_eps;            /* 1e-5 */
  int32_t expanded;        /* EYOC_VERSION >= 111.  1 = ResUNetExpanded (model/resunet.py:254-484): every stage runs
                              block<i> -> norm<i>_2 -> block<i>_2; the layer list then also names "norm<i>_2" (a batch norm that
                              stands alone: one elementwise layer of the plan) and "block<i>_2.conv1" ... for i in 1..4, 4_tr..2_tr */
} eyoc_model_desc;

typedef struct {
  const char* name;        /* "conv1", "block2.conv1", "final", "norm1", "block2.norm2", ... */
  const float* kernel;     /* conv: [K,cin,cout] (K==1 may be [cin,cout]); NULL for norms    */
  int32_t K, cin, cout;
  const float* bias;       /* conv bias [cout] or NULL (only "final")                        */
  const float* bn_weight;  /* norm entries: gamma, beta, running_mean, running_var [c]       */
  const float* bn_bias;
  const float* bn_mean;
  const float* bn_var;
} eyoc_layer_params;

size_t eyoc_model_blob_floats(const eyoc_model_desc* desc);
/* Host-only packing (touches no device): folds the batch norms and writes the blob eyoc_model_create would upload into
 * blob_host (>= eyoc_model_blob_floats() floats).  Lets the broadcasting rank build - and a CPU test verify - the exact
 * bytes that travel over RCCL; eyoc_model_create(layers != NULL) = this + one hipMemcpy. */
int eyoc_model_pack_host(const eyoc_model_desc* desc, const eyoc_layer_params* layers, int n_layers,
                         float* blob_host, size_t blob_floats);
int eyoc_model_create(eyoc_ctx* ctx, const eyoc_model_desc* desc, const eyoc_layer_params* layers,
                      int n_layers, float* blob_dev, size_t blob_floats, eyoc_model** out);
int eyoc_model_destroy(eyoc_model* model);
/* split16 forwards run the network's 1x1 tail (conv1_tr -> ReLU -> final + bias -> row normalisation, model/resunet.py:183-191)
 * as ONE kernel whose 64-channel intermediate never leaves the registers (spconv_tail.hip; BN2C's 96 -> 64 -> 32 widths): mode 1;
 * 2 (default since round 6) = in the epilogue of the last staged stride-1 layer (block2_tr.conv2 of a batch: its 64 output channels
 * go straight into the tail's first product and never reach memory; bit-identical to mode 1), falling back to mode 1 where that
 * layer does not run the 256-row staged kernel; 0 = two launches; other values only query.  Returns the previous state; per ctx. */
int eyoc_model_fuse_tail(eyoc_ctx* ctx, int on);
size_t eyoc_model_workspace_bytes(const eyoc_model* model, const eyoc_maps* maps);
/* feats_dev f32 [N1, in_channels] -> out_dev f32 [N1, out_channels], rows in input order */
int eyoc_model_forward(eyoc_ctx* ctx, const eyoc_model* model, const eyoc_maps* maps,
                       const float* feats_dev, float* out_dev, void* workspace_dev,
                       size_t workspace_bytes, void* stream);
/* per-layer algorithmic work of the last forward geometry (SURVEY.md 8d formulas); arrays of
 * eyoc_model_num_layers() entries, any may be NULL */
int eyoc_model_num_layers(const eyoc_model* model);
int eyoc_model_layer_work(eyoc_ctx* ctx, const eyoc_model* model, const eyoc_maps* maps, void* stream,
                          const char** names, int64_t* pairs, double* flops, double* gather_bytes,
                          double* compulsory_bytes);
/* Arithmetic of the sparse convolutions inside eyoc_model_forward: -1 automatic (default: split16 once the batch
 * has >= 8192 level-0 rows - else fp32), 0 fp32 MFMA, 1 split16 (three fp16 MFMAs per product on
 * hi/lo-split operands: 22-bit significands, fp32 accumulation; activations must stay below 6e4 - guarded, see
 * eyoc_model_range_check).  Returns the
 * previous mode + 2, or a negative status.  eyoc_model_last_math: what the last forward used (0 / 1). */
int eyoc_model_set_math(eyoc_model* model, int mode);
int eyoc_model_last_math(const eyoc_model* model);
/* Range guard of the split16 arithmetic.  Every kernel that stores split16 activations tracks the largest magnitude it
 * writes; once one reaches 6e4 (before any fp16 half became inf) the forward's own device flag is raised - its fp32 output
 * is all-NaN, never plausible-looking garbage (an overflow inside the fused 1x1 tail, whose intermediate never leaves the
 * registers, NaNs the rows it happened in) - together with a sticky flag.  The per-forward flag is cleared, in stream order,
 * when the next split16 forward starts, so forwards enqueued behind an overflowing one are judged on their own.
 * eyoc_model_range_check synchronises `stream`, returns EYOC_ERR_RANGE if ANY forward since the last check overflowed (and
 * clears the sticky flag) else EYOC_OK: a caller that pipelines k forwards and then checks must, on EYOC_ERR_RANGE, treat all
 * k outputs as suspect (the ones that overflowed are the ones holding NaN rows); *max_abs (may be NULL) = largest |activation| stored since eyoc_model_set_probe(model, 1)
 * switched the debug probe on (-1 when the probe is off).  fp32 forwards never raise it. */
int eyoc_model_range_check(eyoc_model* model, void* stream, float* max_abs);
/* The guard's four device words {this forward's overflow flag, max |activation| bits (probe), probe switch, sticky flag} copied
 * to `words_host` (16 bytes of pinned host memory) in stream order, WITHOUT synchronising: enqueued right behind a forward it
 * captures that forward's own verdict (word 0), which a caller that pipelines steps reads once its own event behind the copy has
 * fired - eyoc_model_range_check would queue its read behind everything enqueued since.  Nothing is cleared. */
int eyoc_model_range_snapshot(eyoc_model* model, uint32_t* words_host, void* stream);
/* Every forward records `hip_event` (a hipEvent_t the caller owns; NULL switches it off) on its stream in front of layer `layer`
 * (0-based in launch order, negative counts from the end: -1 = in front of the last layer; == number of layers: behind it).  A
 * caller that pipelines batches lets a side stream wait for it - bench.py starts the next batch's map build there, beside the
 * last layers of the forward instead of behind it. */
int eyoc_model_set_progress_event(eyoc_model* model, int layer, void* hip_event);
int eyoc_model_set_probe(eyoc_model* model, int on);
/* when `on`, eyoc_model_forward brackets every layer with hipEvents on `stream` and
 * eyoc_model_layer_ms returns the per-layer durations of the last forward (synchronises) */
int eyoc_model_set_timing(eyoc_model* model, int on);
int eyoc_model_layer_ms(eyoc_model* model, float* ms /*[num_layers]*/);
/* Two event sets: forwards record into, and eyoc_model_layer_ms reads from, the selected one (0 or 1) - so that a caller
 * can enqueue step k + 1 before it reads step k's durations, and the GPU never waits for the host between steps. */
int eyoc_model_timing_slot(eyoc_model* model, int slot);

/* MFMA pre-filter of eyoc_knn1's plain index query (dist_type 0 or 2, idx only, C = 32): an fp32-MFMA score decides every row
 * whose runner-up is out of rounding reach, the exact kernel recomputes the rest - the indices are identical either way.
 * mode 0: never, 1 (default): when the query fills the chip (>= 512 waves of 64 rows), 2: always; < 0 only queries.
 * Returns the previous mode.  Per ctx; for tests and profiling. */
int eyoc_knn_prefilter(eyoc_ctx* ctx, int mode);
/* ------------------------------------------------------------------------------------------------
 * feature matching
 *   replaces: lib.eval.find_nn_gpu + lib.metrics.pdist (lib/eval.py:18-48, lib/metrics.py:22-29)
 *             and the nearest-neighbour step of Matcher.match_pair (scripts/SC2_PCR/SC2_PCR.py:296-298).
 *   For every row of A the index of the nearest row of B.  dist_type 0: squared L2 (difference
 *   form, fp32, channels left to right, no FMA - bit-exact with oracle/matching.py); 1: L2 =
 *   sqrt(d2 + 1e-7); 2: the GEMM form of Matcher.match_pair, sqrt(2 - 2 S + 1e-6) with S = <A_i, B_j> accumulated by
 *   fp32 FMAs over the channels in order and every later step rounded in fp32 - NOT an L2 distance unless the rows
 *   have unit norm; a NaN distance (S > 1 + 5e-7) is below every number and the first NaN wins, as in torch.argmin.
 *   Ties go to the lowest index.  Segmented form: nseg independent problems,
 *   rows [seg_a[s], seg_a[s+1]) of A against rows [seg_b[s], seg_b[s+1]) of B; indices are local to
 *   the B segment.  seg arrays are HOST arrays; nseg <= 128.  c in {4, 16, 32, 64, 128}.
 * --------------------------------------------------------------------------------------------- */
int eyoc_knn1(eyoc_ctx* ctx, const float* A_dev, const float* B_dev, int c, const int32_t* seg_a,
              const int32_t* seg_b, int nseg, int dist_type, int64_t* idx_dev, float* dist_dev,
              void* stream);

/* Row gather: out[i,:] = F[sel[i], 0:c] (F rows are `ld` floats apart) - the `F[inds]` of
 * scripts/test_kitti.py:30-35,159-160 and of Matcher.match_pair (scripts/SC2_PCR/SC2_PCR.py:291-294).
 * With G_dev != NULL (f32 [n,c]) the gathered row is blended and re-normalised,
 * out[i,:] = (F[sel[i],:] + beta * G[i,:]) / |.|_2 : the synthetic benchmark's descriptor mode (random-init
 * weights carry no geometric signal; G plants it at a stated inlier ratio).  c: power of two in [4,256]. */
int eyoc_gather_rows(eyoc_ctx* ctx, const float* F_dev, int ld, int c, const int64_t* sel_dev, int n,
                     const float* G_dev, float beta, float* out_dev, void* stream);

/* Arg-max of the inner product (replaces ``corr = F0.mm(F1.t()); weight, inds = corr.max(dim=1)`` of
 * util/transform_estimation.py:131-133 without materialising the [N0,N1] matrix - 3.6 GB at 30k voxels):
 * idx int64 [n] (local to the B segment, ties to the lowest index), weight f32 [n] = max_j <A_i, B_j> accumulated
 * with fp32 FMAs over the channels in order. */
int eyoc_dotmax(eyoc_ctx* ctx, const float* A_dev, const float* B_dev, int c, const int32_t* seg_a,
                const int32_t* seg_b, int nseg, int64_t* idx_dev, float* weight_dev, void* stream);

/* Two nearest neighbours for Lowe's ratio test (replaces pytorch3d.ops.knn_points(..., K=2) at
 * lib/trainer.py:1060-1061, squared L2 like knn_points): idx int64 [n] (nearest, local to the B segment), d1 / d2
 * f32 [n] smallest and second smallest squared distance (+inf when the segment has < 2 rows).  Same arithmetic
 * and tie rule as eyoc_knn1; c may also be 4 (xyz padded with a zero column) for the 3-D nearest neighbour
 * of lib/trainer.py:1195. */
int eyoc_knn2(eyoc_ctx* ctx, const float* A_dev, const float* B_dev, int c, const int32_t* seg_a,
              const int32_t* seg_b, int nseg, int64_t* idx_dev, float* d1_dev, float* d2_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * label generation
 *   replaces: calculate_ratio_test + get_topk_matches (lib/trainer.py:993-1017, 1066-1084): weight of every
 *   query from its two nearest squared feature distances (cosine = 1 - d/2; x = clamp(1 - cosine, 1e-9);
 *   weight = 1 - x0/x1, fp32 in that order) and the k queries of largest weight, largest first (torch.topk
 *   order; ties in query order).  idx_out int64 [k], w_out f32 [k] or NULL; k <= n.
 *   eyoc_pair_filter: order-preserving filter of index pairs (idx0[i], idx1[i]):
 *     mode 0 (spherical filter, lib/trainer.py:1107-1110): keep iff |P0[idx0]| > radius and |P1[idx1]| > radius;
 *     mode 1 (pose consistency, lib/trainer.py:1203-1206): keep iff |R P0[idx0] + t - P1[idx1]| < radius,
 *            T_dev f32 [16] row-major.  pairs_out int64 [m,2] (first *n_out rows valid), n_out int32 on the device.
 * --------------------------------------------------------------------------------------------- */
int eyoc_lowe_topk(eyoc_ctx* ctx, const float* d1_dev, const float* d2_dev, int n, int k, int64_t* idx_out_dev,
                   float* w_out_dev, void* stream);
int eyoc_pair_filter(eyoc_ctx* ctx, int mode, const float* P0_dev, const float* P1_dev, const int64_t* idx0_dev,
                     const int64_t* idx1_dev, int m, const float* T_dev, float radius, int64_t* pairs_out_dev,
                     int32_t* n_out_dev, void* stream);

/* The "Similarity" spatial filter of match_and_filter_corr (lib/trainer.py:1118-1149): with d0 = |P0[idx0]|, d1 = |P1[idx1]|
 * a pair is kept iff table[min(int(|d0 - d1| / grid1), xlim - 1)][min(int(min(d0, d1) / grid0), ylim - 1)] > thresh.
 * table_dev: fp64 [xlim, ylim] row-major (one frame-distance slice of config/dist_sim_plot/<dataset>_distSimPlot.npz;
 * the reference uses grid0 = 5 and grid1 in {1, 1.5, 2, 2.5} by frame index).  Output like eyoc_pair_filter. */
int eyoc_pair_filter_similarity(eyoc_ctx* ctx, const float* P0_dev, const float* P1_dev, const int64_t* idx0_dev,
                                const int64_t* idx1_dev, int m, const double* table_dev, int xlim, int ylim, float grid0,
                                float grid1, double thresh, int64_t* pairs_out_dev, int32_t* n_out_dev, void* stream);

/* replaces: lib.metrics.pdist (lib/metrics.py:22-29): dense out f32 [n,m]; same arithmetic as eyoc_knn1 */
int eyoc_pdist(eyoc_ctx* ctx, const float* A_dev, int n, const float* B_dev, int m, int c, int dist_type,
               float* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pose solvers
 * --------------------------------------------------------------------------------------------- */
/* replaces: rigid_transform_3d (scripts/SC2_PCR/common.py:7-45).  A,B f32 [bs,n,3], w f32 [bs,n] or
 * NULL -> T f32 [bs,4,4] with B ~ R A + t.  Weighted centroids use the reference's 1e-6 epsilon. */
int eyoc_kabsch_batched(eyoc_ctx* ctx, const float* A_dev, const float* B_dev, const float* w_dev,
                        int bs, int n, float* T_dev, void* stream);
/* replaces: est_quad_linear_robust (util/transform_estimation.py:89-116).  p0,p1 f32 [n,3],
 * w f32 [n] or NULL -> T f32 [4,4]. */
int eyoc_irls_quad(eyoc_ctx* ctx, const float* p0_dev, const float* p1_dev, const float* w_dev, int n,
                   int iters, float* T_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * training-mode support (lib/trainer.py:1655-1676 back-propagates through the network in train mode)
 * --------------------------------------------------------------------------------------------- */
/* replaces: MinkowskiBatchNorm in training mode (model/common.py:4-6 = nn.BatchNorm1d over the rows): per-channel batch mean
 * and BIASED variance of x f32 [n, c] (rows ld_x floats apart) -> mean_var_dev f32 [2 c] = {mean, variance};
 * y = (x - mean) / sqrt(var + eps) * gamma + beta, then ReLU if `relu`.  c must divide 256 (32 ... 256 in this model family).
 * The statistics are summed in fp64 in a fixed order (bit-reproducible).  The running-statistics update (momentum, unbiased
 * variance) is host glue on the returned 2 c floats.  Workspace: eyoc_bn_workspace_bytes(n, c), caller-owned. */
size_t eyoc_bn_workspace_bytes(int n, int c);
int eyoc_bn_train_forward(eyoc_ctx* ctx, const float* x_dev, int n, int c, int ld_x, const float* gamma_dev, const float* beta_dev,
                          float eps, int relu, float* y_dev, int ld_y, float* mean_var_dev, void* workspace_dev, size_t workspace_bytes,
                          void* stream);
/* ... and the running statistics moved in the same call, like nn.BatchNorm1d in training mode: running_mean = (1 - momentum)
 * running_mean + momentum * mean, running_var likewise with the UNBIASED batch variance (n / (n - 1)).  (num_batches_tracked and a
 * momentum of None - the cumulative average - stay with the caller.) */
int eyoc_bn_train_forward_running(eyoc_ctx* ctx, const float* x_dev, int n, int c, int ld_x, const float* gamma_dev, const float* beta_dev,
                                  float eps, int relu, float* y_dev, int ld_y, float* mean_var_dev, float* running_mean_dev,
                                  float* running_var_dev, float momentum, void* workspace_dev, size_t workspace_bytes, void* stream);
/* Its backward: dy' = dy where y > 0 (y_dev = the forward's output when a ReLU was fused, else NULL);
 * dbeta = sum dy', dgamma = sum dy' xhat, dx = gamma / sigma (dy' - dbeta / n - xhat dgamma / n). */
int eyoc_bn_train_backward(eyoc_ctx* ctx, const float* x_dev, int ld_x, const float* y_dev, int ld_y, const float* dy_dev, int ld_dy, int n,
                           int c, const float* gamma_dev, const float* mean_var_dev, float eps, float* dx_dev, int ld_dx,
                           float* dgamma_dev, float* dbeta_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
/* The first convolution's window as a dense matrix (model/resunet.py:31-38, C_in = 1 in production, K = ks^3):
 * out f32 [n, ks^3 * cin], out[row][k * cin + c] = feats[voxel at window offset k of row][c] or 0 (offsets x fastest, like every
 * rulebook) - the convolution and its weight gradient are then plain [n, K cin] x [K cin, C_out] products.  Builds the level-0 hash
 * table on first use.  eyoc_maps_gather_window: feats and out in the CALLER's rows - maps that are Z-ordered internally
 * (eyoc_maps_row_order != NULL) are refused with EYOC_ERR_INVALID; eyoc_maps_gather_window_internal (EYOC_VERSION >= 110): feats and
 * out in the maps' INTERNAL rows (row i = the caller's row eyoc_maps_row_order()[i]). */
int eyoc_maps_gather_window(eyoc_ctx* ctx, eyoc_maps* maps, int ks, const float* feats_dev, int cin, float* out_dev, void* stream);
int eyoc_maps_gather_window_internal(eyoc_ctx* ctx, eyoc_maps* maps, int ks, const float* feats_dev, int cin, float* out_dev, void* stream);

/* replaces: o3d.pipelines.registration.registration_ransac_based_on_feature_matching(..., 4,
 * [EdgeLength(0.9), Distance(d)], RANSACConvergenceCriteria(4000000, 10000))
 * (scripts/test_kitti.py:169-177) given the feature correspondences.  Hypothesis h samples
 * correspondences with the counter hash documented in oracle/ransac.py. */
typedef struct {
  float max_distance;        /* config.voxel_size * 1.0  */
  float edge_similarity;     /* 0.9                      */
  int32_t max_iteration;     /* 4000000                  */
  uint32_t seed;
} eyoc_ransac_params;

typedef struct {
  float T[16];               /* row-major 4x4            */
  int32_t inliers;
  int32_t best_hypothesis;   /* -1 when nothing survived */
  int32_t survivors;
  float inlier_rmse;
} eyoc_ransac_result;

/* src f32 [n,3], tgt f32 [m,3], corr_tgt int64 [n] (target index for source point i);
 * result_dev points to one eyoc_ransac_result in device memory. */
int eyoc_ransac(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int64_t* corr_tgt_dev,
                int n, const eyoc_ransac_params* params, eyoc_ransac_result* result_dev, void* stream);
/* The same for a batch of independent pairs in a few launches (the per-pair loop of scripts/test_kitti.py:130-225
 * collapsed; SURVEY 8f "batched registration").  Pair b owns source rows / correspondences
 * [seg_src[b], seg_src[b+1]) and target rows starting at seg_tgt[b]; corr_tgt holds target indices LOCAL to the
 * pair's target segment; pair b samples with seed params->seed + b, so results[b] is bit-identical to
 * eyoc_ransac on that pair with that seed.  seg_* are HOST arrays of n_pairs + 1 ints.
 * Up to 64 pairs go through one set of launches (a "chunk"); the scratch holds 12 bytes per hypothesis plus 96 bytes per
 * stored survivor transform (at most 2^20) for every pair of a chunk - 144 MB per pair at 4 000 000 hypotheses.  The results
 * do not depend on the chunk size.
 *
 * eyoc_ransac_batched_ws: CALLER-OWNED scratch (256-byte aligned device memory; nothing is allocated inside the call).
 * eyoc_ransac_workspace_bytes(ctx, n_pairs, total_corr = seg_src[n_pairs], max_iteration, budget) = the bytes of the largest
 * chunk (n_pairs capped at 64, then halved) that stays within `budget` bytes (0 = no limit), never less than a one-pair
 * chunk; the call derives its chunk from workspace_bytes the same way and returns EYOC_ERR_WORKSPACE if not even one pair
 * fits.
 * eyoc_ransac_batched (and eyoc_ransac): the same on the ctx's grow-only scratch, for callers without an allocator - the
 * chunk is sized so that the scratch takes at most a quarter of the device's free memory (hipMemGetInfo) and at most
 * 16 GB, and is halved again (down to one pair) if the allocation fails all the same. */
size_t eyoc_ransac_workspace_bytes(const eyoc_ctx* ctx, int n_pairs, int total_corr, int max_iteration, size_t budget_bytes);
int eyoc_ransac_batched_ws(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int64_t* corr_tgt_dev,
                           const int32_t* seg_src_host, const int32_t* seg_tgt_host, int n_pairs,
                           const eyoc_ransac_params* params, eyoc_ransac_result* results_dev, void* workspace_dev,
                           size_t workspace_bytes, void* stream);
int eyoc_ransac_batched(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int64_t* corr_tgt_dev,
                        const int32_t* seg_src_host, const int32_t* seg_tgt_host, int n_pairs,
                        const eyoc_ransac_params* params, eyoc_ransac_result* results_dev, void* stream);
/* Reference pruning of the scorer (k_bucket: a survivor only evaluates the correspondences that can be inliers given its
 * distance to the pair's first survivor; counts are exactly those of the full sweep): 0 off, 1 on, 2 (default since EYOC_VERSION 111)
 * on + a survivor stops counting once its count so far plus the records still ahead of it is below the largest count any survivor of
 * the pair has reached - the result (arg-max by count, RMSE, hypothesis number) is the same, only the counts of survivors that
 * cannot win stay partial; it is applied to launch chunks of >= 32 pairs (every wave polls one word per pair: on fewer words the
 * polling costs more than the skipped work).  Other values only query.  Returns the previous state.  Per ctx; for tests / profiling. */
int eyoc_ransac_select_pruning(eyoc_ctx* ctx, int on);
/* How many survivor transforms per pair are stored for the scorer (default 2^20 = 96 MB per pair; survivors beyond it are
 * re-derived from their hypothesis number by k_count_overflow - same counts, more work).  survivors >= 1 sets, anything else
 * only queries; returns the previous value.  Per ctx; tests set it tiny to drive every survivor through the overflow path. */
int eyoc_ransac_transform_store(eyoc_ctx* ctx, int survivors);

/* replaces: Matcher.SC2_PCR (scripts/SC2_PCR/SC2_PCR.py:307-384) for bs == 1.
 * src,tgt f32 [n,3] matched correspondences -> T f32 [4,4], seedwise_fitness f32 [int(ratio*n)]. */
typedef struct {
  float inlier_threshold, d_thre, ratio, nms_radius;
  int32_t num_iterations, max_points, k1, k2;
} eyoc_sc2pcr_params;
size_t eyoc_sc2pcr_workspace_bytes(int n, const eyoc_sc2pcr_params* params);
int eyoc_sc2pcr(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, int n,
                const eyoc_sc2pcr_params* params, float* T_dev, float* fitness_dev,
                void* workspace_dev, size_t workspace_bytes, void* stream);
/* A batch of independent pairs (the per-pair loop of lib/trainer.py:1157-1166; SURVEY 8f "batched registration"):
 * pair b = rows [seg[b], seg[b+1]) of src / tgt (seg: HOST array of n_pairs + 1 ints), params[b] per pair,
 * T_dev f32 [n_pairs,16], fitness_dev f32 [n_pairs, fitness_stride].  Pairs run concurrently on internal side
 * streams, forked from / joined to `stream`; results are bit-identical to eyoc_sc2pcr on each pair. */
size_t eyoc_sc2pcr_batched_workspace_bytes(int max_n, const eyoc_sc2pcr_params* params);   /* for any n_pairs (16 slices) */
/* ... for a call of exactly n_pairs pairs: min(n_pairs, 16) slices of the largest pair's workspace */
size_t eyoc_sc2pcr_batched_workspace_bytes_n(int max_n, int n_pairs, const eyoc_sc2pcr_params* params);
int eyoc_sc2pcr_batched(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, const int32_t* seg_host,
                        int n_pairs, const eyoc_sc2pcr_params* params, float* T_dev, float* fitness_dev,
                        int fitness_stride, void* workspace_dev, size_t workspace_bytes, void* stream);
/* Diagnostics of the per-seed stage (second-order counts + top-k1, sc2pcr.hip), per ctx; neither changes any result - tests run
 * both sides of each and compare bit for bit.  Both return the previous value.
 * _set_shortlist_cap: the top-k1 short list holds at most `cap` <= 1024 entries (default 1024; 0 = every seed takes the histogram
 * selection); _set_dense_threshold: a block of 64 consecutive seeds counts as dense - lane = seed kernel, rows in registers - when
 * its hard rows hold >= x * n candidates together (default 0 = every block, since round 6: with the top-k of dense blocks at 0.1 us per seed
 * the one-wave-per-candidate path of sparse blocks costs more than their share of the dense kernel; 2 in round 5; negative = that kernel off). */
int eyoc_sc2pcr_set_shortlist_cap(eyoc_ctx* ctx, int cap);
int eyoc_sc2pcr_set_dense_threshold(eyoc_ctx* ctx, int x);
/* EYOC_VERSION >= 111.  Round-6 forms of three back-end kernels against their round-5 forms (bit-identical results, tests compare):
 * bit 0 = CSR fill walks the row word by word (default: compacted rows), bit 1 = the mask kernel evaluates both square roots of every
 * cross length exactly (default: v_sqrt_f32 pre-test, exact only for undecided lanes), bit 2 = sqrtf(x) < r in the NMS and the
 * seed-fitness sweeps (default: x < T(r), the exact threshold of the correctly rounded sqrtf), bit 3 = every seed's 3 x 3 Kabsch
 * solve and inlier count inside its wave's kernel (default: lane-per-seed kernels behind it).  0 = default; returns the previous bits. */
int eyoc_sc2pcr_select_kernels(eyoc_ctx* ctx, int legacy_bits);

#ifdef __cplusplus
}
#endif
#endif /* EYOC_HIP_H */
