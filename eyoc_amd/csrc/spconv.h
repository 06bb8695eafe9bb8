// Internal interface of the sparse-convolution kernels (spconv.hip) used by model.hip.
#pragma once
#include "common.h"

namespace eyoc {

inline int spconv_ct(int cout) { return cout >= 128 ? 128 : cout; }  // output channels per block
// input channels per staged item: 64 whenever C_in allows it (half as many barriers), else 32
inline int spconv_cc(int cin, int /*cout*/) { return cin % 64 == 0 ? 64 : 32; }

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
  // optional: a permutation of the output rows; tiles take rows in this order (any order gives the same result,
  // a good one makes the rows of a tile share their occupied offsets).  NULL = natural order.
  const int32_t* perm = nullptr;
  // wave-private kernel only: the first `small_rows` rows (in tiling order; a multiple of 64) are cut into 32-row
  // tiles.  They run last (heavy tiles first), so the kernel's tail is made of short-lived waves.  0 = none.
  int small_rows = 0;
};

int launch_spconv(const SpconvArgs& a, hipStream_t st);
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
  float* out;             // rows of ld_out floats
  int ld_out;
  // octree links (level 0 <-> 1) and the level-1 stride-1 table: with them a 3^3 / 5^3 window is read
  // from the 27 coarse blocks around the parent without any hash probe; NULL -> probe the hash table
  const int32_t* parent;    // [n]
  const int32_t* children;  // [nc][8]
  const int32_t* s1c;       // [27][nc]
  int nc;
};
int launch_conv1(const Conv1Args& a, hipStream_t st);
bool conv1_walks_octree(const Conv1Args& a);   // false: launch_conv1 will probe a.table (it must be built)

}  // namespace eyoc
