// Sparse convolution with a tile-local input stage (gfx950, SPLIT16 arithmetic, stride-1 rulebooks).
//
// The split16 kernels are bound by their gather: a stride-1 layer reads every input row ~9 times (once per occupied
// offset of each output row it neighbours), the re-reads are 10^4..10^5 cycles apart, and nothing survives that long
// in a 32 KB L1 through which the CU streams its other waves' gathers - HBM-side traffic equals the algorithmic gather
// bytes, 3.9x the compulsory bytes.  This kernel gathers every input row a tile needs ONCE:
//
//   * the rows of a level are stored in Morton order (eyoc_maps_build sorts them), so 64 consecutive output rows are
//     a compact blob of voxels whose 27-neighbourhoods overlap: ~680 (row, offset) pairs but only ~140 distinct
//     input rows (5x re-use; at most ~250);
//   * a "local rulebook" per tile, built once per stride-1 table (k_local_rulebook, shared by every layer and every
//     32-channel block that uses the table): the list U of distinct input rows (<= 255) and, per (offset, output row),
//     the index into U (255 = no neighbour);
//   * per (tile, 32-channel block) the wave copies its U rows' 128-byte lines global -> LDS (32 KB per wave, 16-byte
//     pieces XOR-swizzled by the row so that 16 random rows spread over the banks), then runs all 27 offsets from LDS:
//     register accumulators, zero operands for missing neighbours (LDS row 255 is zero) and weights shared through an
//     LDS ring by the 4 waves of the workgroup, exactly like the row-stationary kernel (spconv_rs.hip).
//
// One workgroup (4 waves, 152 KB of LDS) per CU, one wave per SIMD: the inner loop has no global loads at all (operands
// and weights from LDS, local indices in 27 registers), so it does not need co-resident waves to hide memory latency.
#include <cstdlib>

#include "spconv.h"

using namespace eyoc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NW = 4;
constexpr int UMAX = 255;                                  // distinct input rows staged per pass (row 255 = zeros)
constexpr int NPASS = 2;                                   // a tile with more distinct rows (<= 510) takes a second pass
constexpr int X_BYTES = 256 * 128;                         // per-wave stage: 256 rows x one 32-channel block

// ---- local rulebook of one 64-row tile (LR_BYTES bytes): int n_unique (-1: more than NPASS * UMAX), pad[3]; int U[512];
// unsigned loc[NPASS][27][16] = per pass the four chunk rows' local indices of (offset k, column j) packed as bytes
// (chunk c in bits 8c..; 255 = no neighbour, or a neighbour staged in the other pass)
constexpr int LR_BYTES = 16 + 512 * 4 + NPASS * 27 * 16 * 4;   // 5520
constexpr int HSLOTS = 2048;                               // > 64 * 27 possible distinct rows: probing always terminates

__global__ __launch_bounds__(256) void k_local_rulebook(const int32_t* __restrict__ nbr, int K, int n_out, unsigned char* __restrict__ out,
                                                        int* __restrict__ overflow) {
  __shared__ int keys[4][HSLOTS];
  __shared__ unsigned short ids[4][HSLOTS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + wave;
  const int row0 = tile * 64;
  if (row0 >= n_out) return;   // wave-uniform; no barrier in this kernel
  int* hk = keys[wave];
  unsigned short* hid = ids[wave];
  for (int i = lane; i < HSLOTS; i += 64) hk[i] = -1;
  const int row = row0 + lane;
  // 1. insert every valid entry (linear probing; duplicates meet their own key)
  for (int k = 0; k < K; ++k) {
    const int idx = row < n_out ? nbr[(size_t)k * n_out + row] : -1;
    if (idx >= 0) {
      unsigned int s = ((unsigned)idx * 2654435761u) >> 21;
      while (true) {
        const int prev = atomicCAS(&hk[s], -1, idx);
        if (prev == -1 || prev == idx) break;
        s = (s + 1) & (HSLOTS - 1);
      }
    }
  }
  // 2. number the occupied slots in slot order (deterministic)
  int base = 0;
  unsigned char* lr = out + (size_t)tile * LR_BYTES;
  int* U = reinterpret_cast<int*>(lr + 16);
  for (int i0 = 0; i0 < HSLOTS; i0 += 64) {
    const int key = hk[i0 + lane];
    const unsigned long long m = __ballot(key >= 0);
    const int id = base + __popcll(m & ((1ull << lane) - 1ull));
    if (key >= 0) {
      hid[i0 + lane] = (unsigned short)id;
      if (id < NPASS * UMAX) U[id] = key;
    }
    base += __popcll(m);
  }
  if (lane == 0) {
    reinterpret_cast<int*>(lr)[0] = base <= NPASS * UMAX ? base : -1;
    if (base > NPASS * UMAX) atomicAdd(overflow, 1);
  }
  // 3. local index of every (offset, row); lanes j = lane & 15 of chunk c = lane >> 4 pack into one word per (k, j)
  unsigned int* loc = reinterpret_cast<unsigned int*>(lr + 16 + 512 * 4);
  for (int k = 0; k < 27; ++k) {
    const int idx = (k < K && row < n_out) ? nbr[(size_t)k * n_out + row] : -1;
    int id = -1;
    if (idx >= 0) {
      unsigned int s = ((unsigned)idx * 2654435761u) >> 21;
      while (hk[s] != idx) s = (s + 1) & (HSLOTS - 1);
      id = hid[s];
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const unsigned int l = (id >= p * UMAX && id < (p + 1) * UMAX) ? (unsigned)(id - p * UMAX) : 255u;
      // gather the four chunks' bytes of column j into lane j: lane (c, j) holds byte c
      unsigned int w = l << (8 * (lane >> 4));
      w |= __shfl_xor(w, 16, 64);
      w |= __shfl_xor(w, 32, 64);
      if (lane < 16) loc[(p * 27 + k) * 16 + lane] = w;
    }
  }
}

// byte offset of 16-byte piece p (0..3 hi halves of channels 8p.., 4..7 lo halves) of staged row l
__device__ __forceinline__ int xs_off(int l, int p) { return l * 128 + ((p ^ (l & 7)) << 4); }

template <int NTW, int CC>
__global__ __launch_bounds__(NW * 64, 1) void spconv_st_kernel(SpconvArgs a, const unsigned char* __restrict__ local) {
  constexpr int CTW = NTW * 16, NC = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned char* xs = smem + wave * X_BYTES;                          // this wave's stage
  const int g = lane >> 4, j = lane & 15;
  const int n_cg = a.cout / CTW;
  const int wg = (int)blockIdx.x;
  const int rgw = wg / n_cg, cg = wg - rgw * n_cg;
  const int tile = rgw * NW + wave;
  const int row0 = tile * 64;
  const bool live = row0 < a.n_out;                                    // wave-uniform; dead waves keep the barriers company
  const int ct0 = cg * CTW;
  const int CT = a.cout >= 128 ? 128 : a.cout;
  const int n_slices = a.cout / CT, slice = ct0 / CT, nt0 = (ct0 - slice * CT) / 16;
  constexpr int JQ = CC / 16;
  const int ncc = a.cin / CC;
  const int nqb = a.cin / 32;
  constexpr int K = 27;

  const unsigned char* lr = local + (size_t)(live ? tile : 0) * LR_BYTES;
  const int n_u = live ? __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(lr)[0]) : 0;
  const int* __restrict__ U = reinterpret_cast<const int*>(lr + 16);
  const unsigned int* __restrict__ locp = reinterpret_cast<const unsigned int*>(lr + 16 + 512 * 4);
  const int n_pass = n_u > UMAX ? 2 : 1;                               // wave-uniform; the second pass is rare (a tile across a Z-curve jump)
  unsigned int loc4[27];
  // zero row (local index 255), written once: no stage ever touches it
  if (lane < 8) *reinterpret_cast<float4*>(xs + 255 * 128 + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, K * a.cin * a.cout * 4, 0x00020000);
  const int tile4 = CC * CT / 4;

  f32x4 acc[NC][NTW];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this wave's MFMA weight fragments of stage (k, qb), straight from L2 / L1 into operand registers, one stage ahead.
  // (No LDS ring and no barrier here, unlike spconv_rs.hip: with one wave per SIMD a barrier per stage leaves the
  // matrix pipe idle while the slowest wave arrives, and 4 waves per CU ask the L1 for 8 KB per ~800-cycle stage.)
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
  // stage block qb of this tile's distinct input rows: 8 lanes per row (one 128-byte line), 8 rows per instruction,
  // global -> LDS directly (global_load_lds_dwordx4: lane i lands at base + 16 i, so the XOR swizzle of the pieces is
  // applied on the SOURCE side), all loads of a stage in flight at once - one memory latency per (tile, block)
  int Ureg[32];                                                      // this lane's rows of the pass: U[pass * 255 + 8 it + lane / 8]
  int n_up = 0;                                                      // distinct rows of the current pass
  auto begin_pass = [&](int pass) {
    n_up = min(n_u - pass * UMAX, UMAX);
#pragma unroll
    for (int it = 0; it < 32; ++it) {
      const int l = it * 8 + (lane >> 3);
      Ureg[it] = l < n_up ? U[pass * UMAX + l] : 0;
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) loc4[k] = live ? locp[(pass * 27 + k) * 16 + j] : 0xFFFFFFFFu;
  };
  auto stage = [&](int qb) {
#if defined(EYOC_ST_ABL) && (EYOC_ST_ABL & 2)
    return;
#endif
#pragma unroll
    for (int it = 0; it < 32; ++it) {
      if (it * 8 < n_up) {                                           // wave-uniform
        const int l = it * 8 + (lane >> 3);
        const float* src = a.in + (size_t)Ureg[it] * a.ld_in + qb * 32 + (((lane & 7) ^ (l & 7)) << 2);
        if (l < n_up)
          __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(xs + it * 1024), 16, 0, 0);
      }
    }
  };
  auto read_x = [&](unsigned int w, float4 (&X)[NC][2]) {
    // the LDS addresses only depend on the local indices, which do not change from block to block: left alone, the
    // compiler hoists all 27 x 8 address computations out of the block loop and spills (512 VGPRs); this keeps them here
    asm volatile("" : "+v"(w));
#if defined(EYOC_ST_ABL) && (EYOC_ST_ABL & 8)
    for (int c = 0; c < NC; ++c) { X[c][0].x = __uint_as_float(w); asm volatile("" : "+v"(X[c][0].y), "+v"(X[c][1].x)); }
    return;
#endif
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#if defined(EYOC_ST_ABL) && (EYOC_ST_ABL & 1)
      const int l = 255 + 0 * (int)w;
#else
      const int l = (int)((w >> (8 * c)) & 255u);
#endif
      X[c][0] = *reinterpret_cast<const float4*>(xs + xs_off(l, g));
      X[c][1] = *reinterpret_cast<const float4*>(xs + xs_off(l, 4 + g));
    }
  };
  auto multiply = [&](const float4 (&X)[NC][2], const float4 (&W)[NTW][2]) {
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

  // ---- stage stream: s = 27 qb + k (this kernel serves 3^3 stride-1 tables only, K == 27; the offset loop is unrolled so
  // that the local indices stay in registers).  While stage s multiplies, the inputs of stage s + 1 are read from the
  // LDS stage and its weight fragments from L2 into the other operand set.  The waves of a workgroup never wait for
  // each other.
  float4 XA[NC][2], XB[NC][2], WA[NTW][2], WB[NTW][2];
  for (int pass = 0; pass < n_pass; ++pass) {
    begin_pass(pass);
    load_w(0, 0, WA);
    stage(0);
    __builtin_amdgcn_s_waitcnt(0x0070);                              // vmcnt(0): the stage has landed
    for (int qb = 0; qb < nqb; ++qb) {
      read_x(loc4[0], XA);
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const int kn = k < 26 ? k + 1 : 0, qn = k < 26 ? qb : qb + 1;
        if (k & 1) {
          if (qn < nqb) load_w(kn, qn, WA);
          if (k < 26) read_x(loc4[kn], XA);
          multiply(XB, WB);
        } else {
          if (qn < nqb) load_w(kn, qn, WB);
          if (k < 26) read_x(loc4[kn], XB);
          multiply(XA, WA);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting later stages' loads up here (it spills at 512 VGPRs)
      }
      // offset 26 multiplied set A and fetched the next block's first weights into set B: hand them over, refill the stage
      if (qb + 1 < nqb) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) { WA[t][0] = WB[t][0]; WA[t][1] = WB[t][1]; }
        stage(qb + 1);
        __builtin_amdgcn_s_waitcnt(0x0070);
      }
    }
  }

  // ---- epilogue straight from the registers: lane (g, j) holds channels 16 t + 4 g .. +3 of row 16 c + j
  if (!live) return;
  const float os = a.out_scale ? *a.out_scale : 1.0f;
  float4 b4[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
    b4[t] = a.bias ? *reinterpret_cast<const float4*>(a.bias + ct0 + 16 * t + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int o = row0 + 16 * c + j;
    if (o >= a.n_out) continue;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int ch = ct0 + 16 * t + 4 * g;
      float4 v = make_float4(acc[c][t][0] * os + b4[t].x, acc[c][t][1] * os + b4[t].y, acc[c][t][2] * os + b4[t].z,
                             acc[c][t][3] * os + b4[t].w);
      if (a.res) {
        const float4 q = split16_load4(a.res + (size_t)o * a.ld_res, ch);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (a.out_split) split16_store4(a.out + (size_t)o * a.ld_out, ch, v);
      else *reinterpret_cast<float4*>(a.out + (size_t)o * a.ld_out + ch) = v;
    }
  }
}

}  // namespace

namespace eyoc {

size_t local_rulebook_bytes(int n_out) { return (size_t)cdiv(n_out, 64) * LR_BYTES; }

// builds the per-tile local rulebooks of a stride-1 table; *overflow_dev (zeroed by the caller) counts tiles with more
// than 255 distinct input rows (the staged kernel must not be used for the table then)
int build_local_rulebook(const int32_t* nbr_dev, int K, int n_out, unsigned char* out_dev, int* overflow_dev, hipStream_t st) {
  if (n_out <= 0) return EYOC_OK;
  hipLaunchKernelGGL(k_local_rulebook, dim3(cdiv(cdiv(n_out, 64), 4)), dim3(256), 0, st, nbr_dev, K, n_out, out_dev, overflow_dev);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

// stride-1 SPLIT16 layers whose table has a local rulebook (rows in natural = Morton order, no tiling permutation)
int launch_spconv_st(const SpconvArgs& a, const unsigned char* local_dev, hipStream_t st) {
  EYOC_REQUIRE(a.math == 1 && local_dev && !a.l2norm && a.K == 27 && !a.perm, EYOC_ERR_INVALID, "spconv_st: unsupported layer");
  const int ctw = a.cout >= 64 ? 64 : 32;
  const bool wide = spconv_cc(a.cin, a.cout) == 64;
  const long long wgs = (long long)cdiv(a.n_out, 64 * NW) * (a.cout / ctw);
  const dim3 grid((unsigned)wgs), block(NW * 64);
  const size_t lds = (size_t)NW * X_BYTES;
  static bool attr_done[4] = {false, false, false, false};
#define EYOC_ST(NTW_, CC_, I_)                                                                                              \
  do {                                                                                                                      \
    if (!attr_done[I_]) {                                                                                                   \
      EYOC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spconv_st_kernel<NTW_, CC_>),                        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                          \
      attr_done[I_] = true;                                                                                                 \
    }                                                                                                                       \
    hipLaunchKernelGGL((spconv_st_kernel<NTW_, CC_>), grid, block, lds, st, a, local_dev);                                  \
  } while (0)
  if (ctw == 64) { if (wide) EYOC_ST(4, 64, 0); else EYOC_ST(4, 32, 1); }
  else { if (wide) EYOC_ST(2, 64, 2); else EYOC_ST(2, 32, 3); }
#undef EYOC_ST
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

}  // namespace eyoc
