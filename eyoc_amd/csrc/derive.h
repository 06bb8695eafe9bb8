// Neighbour derivation through the octree links (device code shared by coordmap.hip's k_derive_fine and the fused tile-record
// builder of spconv_st.hip).  A fine row o of level l with parity bits b (per axis: is its coordinate an odd multiple of the level's
// stride) and parent p reaches, per axis, position b + off of the 4-wide strip [own block bit 0, own block bit 1, next block bit 0,
// next block bit 1] rotated so that -1 lands in the block on the side the voxel leans away from:
//   off -1 -> (b ? own block, bit 0 : block at -1, bit 1);  off 0 -> own block, bit b;  off +1 -> (b ? block at +1, bit 0 : own block, bit 1).
// So the 27 stride-1 neighbours are a 3 x 3 x 3 window of the 4 x 4 x 4 cube of children of 8 coarse blocks: the own one and the
// neighbours on the side the voxel leans to, read from the COARSE stride-1 table (7 entries) and the children records (8 x 32 bytes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace eyoc {

// blk[a], a = bit per axis: 0 = own coarse block, 1 = the neighbouring block on the side this voxel leans to (-1: no such voxel)
__device__ __forceinline__ void derive_blocks(const int (&b)[3], int p, const int32_t* __restrict__ s1c, int nc, int (&blk)[8]) {
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int ox = (a & 1) ? (b[0] ? 1 : -1) : 0, oy = (a & 2) ? (b[1] ? 1 : -1) : 0, oz = (a & 4) ? (b[2] ? 1 : -1) : 0;
    const int kc = (ox + 1) + 3 * (oy + 1) + 9 * (oz + 1);
    blk[a] = a == 0 ? p : s1c[(size_t)kc * nc + p];
  }
}

// v[k] = row of the neighbour at offset k (x fastest) or -1.  The 8 child RECORDS are loaded whole (32 bytes each: 16 wide loads
// instead of 27 four-byte loads that each pull a 32-byte sector) and the window is cut out with per-axis selects; nothing is indexed
// by a run-time value.
__device__ __forceinline__ void derive_window(const int (&b)[3], const int (&blk)[8], const int32_t* __restrict__ children, int (&v)[27]) {
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
  // cube position u per axis: 0 / 1 = own block, child bit 0 / 1; 2 / 3 = the neighbouring block, child bit 0 / 1.
  // Window position k (offset k - 1) of an axis with parity bit p: p = 0 -> u = 3, 0, 1 (the block at -1 ends with its bit-1
  // child); p = 1 -> u = 0, 1, 2 (the block at +1 starts with its bit-0 child).
  int wx[4][4][3];                                                     // [uz][uy][kx]
#pragma unroll
  for (int uz = 0; uz < 4; ++uz)
#pragma unroll
    for (int uy = 0; uy < 4; ++uy) {
      int in[4];
#pragma unroll
      for (int ux = 0; ux < 4; ++ux) in[ux] = rec[(ux >> 1) | ((uy >> 1) << 1) | ((uz >> 1) << 2)][(ux & 1) | ((uy & 1) << 1) | ((uz & 1) << 2)];
      wx[uz][uy][0] = b[0] ? in[0] : in[3];
      wx[uz][uy][1] = b[0] ? in[1] : in[0];
      wx[uz][uy][2] = b[0] ? in[2] : in[1];
    }
  int wy[4][3][3];                                                     // [uz][ky][kx]
#pragma unroll
  for (int uz = 0; uz < 4; ++uz)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      wy[uz][0][kx] = b[1] ? wx[uz][0][kx] : wx[uz][3][kx];
      wy[uz][1][kx] = b[1] ? wx[uz][1][kx] : wx[uz][0][kx];
      wy[uz][2][kx] = b[1] ? wx[uz][2][kx] : wx[uz][1][kx];
    }
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int kx = k % 3, ky = (k / 3) % 3, kz = k / 9;
    v[k] = kz == 0 ? (b[2] ? wy[0][ky][kx] : wy[3][ky][kx]) : kz == 1 ? (b[2] ? wy[1][ky][kx] : wy[0][ky][kx]) : (b[2] ? wy[2][ky][kx] : wy[1][ky][kx]);
  }
}

// blocks of the transposed map: per axis the own block (offset 0 on an even position, +1 on an odd one) or, for offset -1 on an odd
// position, the neighbouring one: bl[m] = blk[m & parity class], m = bit per axis "offset -1"
__device__ __forceinline__ void derive_up_blocks(const int (&b)[3], const int (&blk)[8], int (&bl)[8]) {
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    int sx[2][2];
#pragma unroll
    for (int az = 0; az < 2; ++az)
#pragma unroll
      for (int ay = 0; ay < 2; ++ay) sx[az][ay] = ((m & 1) && b[0]) ? blk[1 | (ay << 1) | (az << 2)] : blk[(ay << 1) | (az << 2)];
    const int sy0 = ((m & 2) && b[1]) ? sx[0][1] : sx[0][0], sy1 = ((m & 2) && b[1]) ? sx[1][1] : sx[1][0];
    bl[m] = ((m & 4) && b[2]) ? sy1 : sy0;
  }
}

// Compact transposed table up8[m][o] (m = bit per axis "offset -1"): the coarse row a fine row o of parity class cls reads through
// the offset with off_axis = (cls bit ? (m bit ? -1 : +1) : 0), or -1 - defined for m a subset of cls only (-1 elsewhere).  The full
// [27][n] table has the same entries at k = sum_axis (off_axis + 1) 3^axis and -1 everywhere else.
__host__ __device__ constexpr int up8_slot_of_offset(int k) { return ((k % 3) == 0 ? 1 : 0) | (((k / 3) % 3) == 0 ? 2 : 0) | ((k / 9) == 0 ? 4 : 0); }

}  // namespace eyoc
