// Sparse convolution, wave-private tiling (gfx950).
//
// Same operator and the same packed-weight layout as spconv.hip, different decomposition: every WAVE owns
// BMW consecutive output rows x CTW output channels and runs with no workgroup barrier at all.
//   * the wave walks the kernel offsets k; for each it ballots which of its rows have a neighbour and
//     compacts (input row << 8 | local output row) into a small LDS list (rulebook loads run two
//     offsets ahead);
//   * an item = (k, <= NCMAX chunks of 16 pairs).  The gathered input rows go STRAIGHT from global memory
//     into MFMA operand registers: lane (g = lane >> 4, j = lane & 15) reads the 16 bytes of pair j's row
//     that hold channels 16q + 4g .. +3, which is exactly operand element [k-slot g][column j] of
//     v_mfma_f32_16x16x4_f32 for the four steps e = 0..3 of channel block q (the weights are packed on the
//     host with the same k-slot <-> channel permutation, spconv.hip: eyoc_spconv_pack_weights).  No LDS
//     staging of operands, no barriers, and one wave amortises its bookkeeping over NC x CTW/16 x CC/4
//     MFMAs (128 .. 256) instead of ~23 in the workgroup-tiled kernel;
//   * the weights are the other MFMA operand (read from L2/L1 in fragment order), so a lane's four
//     accumulator registers are four consecutive output channels of one output row: one 128-bit LDS
//     read-add-write per (chunk, 16-channel tile) into the wave's XOR-swizzled accumulator rows;
//   * padding pairs gather input row 0 and add into a trash accumulator row (no masks on the hot path);
//   * epilogue from LDS: + bias (+ residual) -> ReLU -> (row L2 normalisation) -> coalesced float4 stores.
// Summation order per output element is fixed (k ascending, channels ascending inside the MFMA chain), so
// results are bit-reproducible from run to run.
//
// MATH = 1 (SPLIT16): the same decomposition on the fp16 matrix pipe.  fp32 MFMA runs at the VALU rate (157 TFLOP/s),
// fp16 MFMA 8x (16x16x32) to 16x faster, and 1e-4 parity rules out plain fp16 / bf16 operands - but not exact
// splitting: every operand is kept as hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 significant bits, spconv.h),
// and a product block is three MFMAs, W_hi X_hi + W_hi X_lo + W_lo X_hi, accumulated in fp32 (the dropped lo*lo term
// is 2^-22 relative).  Activations are STORED in that format by the producing layer's epilogue (4 bytes per channel
// like fp32, so gather bytes are unchanged) and weights are split on the host, so the hot loop has no conversions:
// lane (g, j) reads the two 16-byte pieces of pair j's row that hold channels 32q + 8g .. +7 (hi, lo) = operand
// element [k-slots 8g .. 8g+7][column j] of v_mfma_f32_16x16x32_f16.  48 MFMAs of 32 cycles per unit instead of 128.
// Errors against an fp64 oracle are those of the fp32 path (tests/test_gpu_split16.py).
#include <cstdlib>
#include <type_traits>
#ifndef EYOC_WPB
#define EYOC_WPB 1
#endif
#ifndef EYOC_REV
#define EYOC_REV 1
#endif
#ifndef EYOC_XG
#define EYOC_XG 32
#endif
#ifndef EYOC_ABL
#define EYOC_ABL 0   // ablation builds (diagnostics only)
#endif

#include "spconv.h"

using namespace eyoc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build (-DEYOC_TRACE, scripts/trace_spconv.py): a few waves stamp the cycle counter at phase boundaries.
#ifdef EYOC_TRACE
constexpr int TRACE_STAMPS = 512, TRACE_WAVES = 16;
__device__ unsigned long long g_wtrace[TRACE_WAVES * TRACE_STAMPS];
#define TR_DECL const int tr_w = (tile >= 6000 && tile < 6000 + TRACE_WAVES) ? tile - 6000 : -1; int tr_n = 0
#define TR() do { if (tr_w >= 0 && lane == 0 && tr_n < TRACE_STAMPS) g_wtrace[tr_w * TRACE_STAMPS + tr_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define TR_DECL
#define TR()
#endif

template <int CTW, int BMW, int CC, int NCMAX>
struct WCfg {
  static constexpr int NTW = CTW / 16;          // 16-channel tiles per wave
  static constexpr int JQ = CC / 16;            // 16-channel blocks per C_in slice
  static constexpr int C4N = CTW / 4;           // float4 columns per accumulator row
  static constexpr int RND = BMW / 64;
  static constexpr int LIST = BMW + 16 * NCMAX; // compacted pairs of one offset, padded to whole items
  static constexpr int ACC_BYTES = (BMW + 1) * CTW * 4;   // + 1: trash row for padding pairs
  static constexpr int WAVE_BYTES = ACC_BYTES + 2 * LIST * 4;
  // One wave per workgroup: the waves are independent, and a two-wave workgroup keeps its slot until the slower
  // wave is done (tile costs vary +-30 %): measured -1.7 % forward time for 1 vs 2.
  static constexpr int WPB = EYOC_WPB;
  static_assert(BMW == 64 || BMW == 128, "rows per wave");
  static_assert(WAVE_BYTES % 16 == 0 && WPB * WAVE_BYTES <= 64 * 1024, "LDS budget");
};

template <int CTW, int BMW, int CC, int NCMAX, int OCC, int MATH>
__global__ __launch_bounds__((WCfg<CTW, BMW, CC, NCMAX>::WPB * 64), OCC) void spconv_wave_kernel(SpconvArgs a) {
  using C = WCfg<CTW, BMW, CC, NCMAX>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::WPB * C::WAVE_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n_cg = a.cout / CTW;
  // XCD-aware tile order: consecutive workgroup ids land on different XCDs (8 of them, each with its own L2).
  // Within every run of 8 * XG workgroups, XCD x takes XG CONSECUTIVE tile blocks, so neighbouring row tiles (whose
  // gathers overlap) share an L2; the runs themselves stay round-robin because the work per tile drifts along the
  // pattern-sorted row order (one contiguous range per XCD was 30 % slower: the XCDs finish at different times).
  // Measured effect of XG in {1 (plain round robin), 2, 8, 32}: within +-1 % - the gathers are not L2-capacity bound.
  constexpr int XG = EYOC_XG;
  const int bid = (int)blockIdx.x, run = bid / (8 * XG), in_run = bid % (8 * XG);
  int tile = (run * 8 * XG + (in_run & 7) * XG + (in_run >> 3)) * C::WPB + wave;
  // heavy tiles first: the pattern-sorted row orders put the rows with the most neighbours last, and a kernel's
  // tail is as long as its last tiles (measured -2 % forward time)
  if (EYOC_REV) tile = (int)gridDim.x * C::WPB - 1 - tile;
  const int rg = tile / n_cg, cg = tile - rg * n_cg;
  const int n_small = a.small_rows / (BMW / 2);                   // half-height tiles at the low end of the order
  if (rg < 0) return;
  const int row0 = rg < n_small ? rg * (BMW / 2) : a.small_rows + (rg - n_small) * BMW;
  if (row0 >= a.n_out) return;   // wave-uniform; there is no barrier anywhere in this kernel
  const int rows_here = min(rg < n_small ? BMW / 2 : BMW, a.n_out - row0);
  TR_DECL;
  TR();
  float* acc = reinterpret_cast<float*>(smem + wave * C::WAVE_BYTES);
  unsigned int* list = reinterpret_cast<unsigned int*>(smem + wave * C::WAVE_BYTES + C::ACC_BYTES);
  auto acc_off = [](int row, int c4) { return row * CTW + ((c4 ^ row) & (C::C4N - 1)) * 4; };

  const int ct0 = cg * CTW;
  const int CT = a.cout >= 128 ? 128 : a.cout;   // = spconv_ct(cout): the packed layout's column slice
  const int n_slices = a.cout / CT, slice = ct0 / CT, nt0 = (ct0 - slice * CT) / 16;
  const int ncc = a.cin / CC;
  const int K = a.K;
  const int j16 = lane & 15, g = lane >> 4;

  for (int i = lane; i < (BMW + 1) * C::C4N; i += 64) reinterpret_cast<float4*>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  // global output row of this lane's tile row(s): the tile takes rows in a.perm order when given
  int grow0 = lane < rows_here ? row0 + lane : -1, grow1 = -1;
  if constexpr (C::RND == 2) grow1 = 64 + lane < rows_here ? row0 + 64 + lane : -1;
  if (a.perm) {
    if (grow0 >= 0) grow0 = a.perm[grow0];
    if (grow1 >= 0) grow1 = a.perm[grow1];
  }
  auto load_idx = [&](int k, int r) -> int {
    const int grow = r == 0 ? grow0 : grow1;
#if EYOC_ABL >= 6
    return (k < K && grow >= 0) ? grow : -1;   // dense synthetic rulebook, no memory
#endif
    if (a.nbr) return (k < K && grow >= 0) ? a.nbr[(size_t)k * a.n_out + grow] : -1;
    return (k == 0 && grow >= 0) ? grow : -1;   // identity map (1x1 convolution)
  };
  auto compact = [&](int i0, int i1, int slot) -> int {
    unsigned int* L = list + slot * C::LIST;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long m0 = __ballot(i0 >= 0);
    if (i0 >= 0) L[__popcll(m0 & lt)] = ((unsigned)i0 << 8) | (unsigned)lane;
    int P = __popcll(m0);
    if constexpr (C::RND == 2) {
      const unsigned long long m1 = __ballot(i1 >= 0);
      if (i1 >= 0) L[P + __popcll(m1 & lt)] = ((unsigned)i1 << 8) | (unsigned)(64 + lane);
      P += __popcll(m1);
    }
    P = __builtin_amdgcn_readfirstlane(P);
    const int padded = (P + 16 * NCMAX - 1) / (16 * NCMAX) * (16 * NCMAX);
    if (lane < padded - P) L[P + lane] = (unsigned)BMW;   // gathers input row 0, adds into the trash row
    return P;
  };

  const int tile4 = CC * CT / 4;   // float4s of one packed (k, slice, cc) weight tile

  // ---- the unit stream.  A unit = (offset k, item of <= NCMAX chunks, C_in slice cc).  While unit u runs its
  // MFMAs, the operands of unit u+1 are already in flight: vmcnt retires in order, so everything unit u
  // still needs (the second half of its weights) is issued BEFORE the long-latency gather of unit u+1.
  constexpr int QH = 1;           // weight blocks [0,QH) are prefetched a unit ahead, [QH,JQ) at the start of the unit
  int kq = 0;                     // next offset to compact
  int ia0 = load_idx(0, 0), ia1 = C::RND == 2 ? load_idx(0, 1) : -1;
  int ib0 = load_idx(1, 0), ib1 = C::RND == 2 ? load_idx(1, 1) : -1;
  // compacts offsets into `slot` until a non-empty one is found; returns its pair count (0: none left)
  auto produce = [&](int slot, int& k_out) -> int {
    int P = 0;
    k_out = K;
    while (kq < K) {
      P = compact(ia0, ia1, slot);
      const int kt = kq;
      ia0 = ib0; ia1 = ib1;
      ib0 = load_idx(kq + 2, 0);
      if constexpr (C::RND == 2) ib1 = load_idx(kq + 2, 1);
      ++kq;
      if (P > 0) { k_out = kt; break; }
    }
    return P;
  };
  // weight fragments come through buffer loads: descriptor + wave-uniform tile offset in SGPRs, the lane's
  // 16 bytes in ONE VGPR for the whole kernel, the fragment index in the immediate - no 64-bit VALU math
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, K * a.cin * a.cout * 4, 0x00020000);
  const int lane_off = lane * 16;
  auto wptr = [&](int k, int cc) -> int {   // byte offset of this wave's fragments of tile (k, slice, cc)
    return __builtin_amdgcn_readfirstlane((((k * n_slices + slice) * ncc + cc) * tile4 + nt0 * C::JQ * 64) * 16);
  };
  auto ldw = [&](int wt, int t, int q) -> float4 {
    constexpr int FPI = 4;   // fragments reachable through the 12-bit immediate
    const int f = t * C::JQ + q;
#if EYOC_ABL == 2 || EYOC_ABL >= 5
    return make_float4((float)lane, 1.f, (float)wt, 3.f);
#endif
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_off + (f % FPI) * 1024, wt + (f / FPI) * FPI * 1024, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  };

  TR();
  int k_c = K, k_n = K;
  const int P_first = produce(0, k_c);
  if (P_first > 0) {
    int P_n = produce(1, k_n);
    int slot_c = 0, c0 = 0, cc = 0, nch_c = (P_first + 15) >> 4;
    // two operand sets that swap roles every unit (no register rotation): gather pointers, output rows,
    // gathered fragments and weight fragments of the unit in flight / the unit being multiplied
    const float* gpA[NCMAX]; const float* gpB[NCMAX];
    int orowA[NCMAX], orowB[NCMAX];
    float4 GA[C::JQ][NCMAX], GB[C::JQ][NCMAX], WA[C::JQ][C::NTW], WB[C::JQ][C::NTW];
    f32x4 accr[NCMAX][C::NTW];

    // orow[] holds, per chunk, the BYTE offset of this lane's float4 column of tile 0 inside its accumulator row:
    // row * CTW * 4 + ((g ^ row) & (C4N - 1)) * 16.  Tile t sits at that offset XOR t * 64 (the swizzle of acc_off,
    // with the row-only part hoisted out of the per-tile work).
    auto read_records = [&](int slot, int cbase, const float* (&gp)[NCMAX], int (&orow)[NCMAX]) {
      const unsigned int* L = list + slot * C::LIST + cbase * 16 + j16;
#pragma unroll
      for (int c = 0; c < NCMAX; ++c) {
        const unsigned rec = L[c * 16];
        gp[c] = a.in + (size_t)(rec >> 8) * a.ld_in + g * 4;
        const int row = (int)(rec & 255u);
        orow[c] = row * (CTW * 4) + (((g ^ row) & (C::C4N - 1)) << 4);
      }
    };
    auto load_gather = [&](const float* (&gp)[NCMAX], int cc_, float4 (&G)[C::JQ][NCMAX]) {
#pragma unroll
      for (int q = 0; q < C::JQ; ++q)
#pragma unroll
        for (int c = 0; c < NCMAX; ++c) {
#if EYOC_ABL == 1 || EYOC_ABL >= 5
          G[q][c] = make_float4((float)lane, 1.f, 2.f, (float)cc_);
#else
          // fp32: channels 16 q + 4 g .. +3.  SPLIT16: fragment q = 2 q' + p holds the hi (p = 0) / lo (p = 1) halves of
          // channels 32 q' + 8 g .. +7: 16 bytes at g * 16 inside the first / second 64 bytes of block q' (spconv.h)
          G[q][c] = *reinterpret_cast<const float4*>(gp[c] + cc_ * CC + (MATH ? (q >> 1) * 32 + (q & 1) * 16 : q * 16));
#endif
        }
    };
    auto load_w_head = [&](int wt, float4 (&W)[C::JQ][C::NTW]) {
#pragma unroll
      for (int q = 0; q < QH; ++q)
#pragma unroll
        for (int t = 0; t < C::NTW; ++t) W[q][t] = ldw(wt, t, q);
    };
    auto load_w_tail = [&](int wt, float4 (&W)[C::JQ][C::NTW]) {
#pragma unroll
      for (int q = QH; q < C::JQ; ++q)
#pragma unroll
        for (int t = 0; t < C::NTW; ++t) W[q][t] = ldw(wt, t, q);
    };

    // MFMAs and accumulator flush of the current unit, specialised on its chunk count
    auto compute = [&](auto nc_tag, bool first_cc, bool last_cc, const float4 (&G)[C::JQ][NCMAX],
                       const float4 (&W)[C::JQ][C::NTW], const int (&orow)[NCMAX]) {
      constexpr int NC = decltype(nc_tag)::value;
      if constexpr (MATH == 1) {
        // three fp16 MFMAs per (chunk, tile, 32-channel block): W_hi X_hi, W_hi X_lo, W_lo X_hi.  The first one of an
        // item's first C_in slice takes a literal zero as its C operand (no accumulator clearing); that choice is made
        // ONCE per unit - a test inside the loops costs a scalar branch per MFMA and fences the schedule
        if (first_cc) {
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int t = 0; t < C::NTW; ++t)
              accr[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, W[0][t]), __builtin_bit_cast(half8_t, G[0][c]),
                                                                  f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        } else {
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int t = 0; t < C::NTW; ++t)
              accr[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, W[0][t]), __builtin_bit_cast(half8_t, G[0][c]),
                                                                  accr[c][t], 0, 0, 0);
        }
#pragma unroll
        for (int qq = 0; qq < C::JQ / 2; ++qq)
#pragma unroll
          for (int term = 0; term < 3; ++term) {
            if (qq == 0 && term == 0) continue;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const half8_t xv = __builtin_bit_cast(half8_t, G[2 * qq + (term == 1 ? 1 : 0)][c]);
#pragma unroll
              for (int t = 0; t < C::NTW; ++t) {
                const half8_t wv = __builtin_bit_cast(half8_t, W[2 * qq + (term == 2 ? 1 : 0)][t]);
                accr[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xv, accr[c][t], 0, 0, 0);
              }
            }
          }
      } else {
      // the first MFMA of an item's first C_in slice takes a literal zero as its C operand: no accumulator clearing
      {
        if (first_cc) {
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int t = 0; t < C::NTW; ++t)
              accr[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[0][t].x, G[0][c].x, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        } else {
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int t = 0; t < C::NTW; ++t)
              accr[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[0][t].x, G[0][c].x, accr[c][t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < C::JQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (q == 0 && e == 0) continue;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const float av = e == 0 ? G[q][c].x : e == 1 ? G[q][c].y : e == 2 ? G[q][c].z : G[q][c].w;
#pragma unroll
            for (int t = 0; t < C::NTW; ++t) {
              const float bv = e == 0 ? W[q][t].x : e == 1 ? W[q][t].y : e == 2 ? W[q][t].z : W[q][t].w;
              accr[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, accr[c][t], 0, 0, 0);
            }
          }
        }
      }
      if (last_cc && EYOC_ABL != 3) {
        // D[i = 4 g + reg][j] = (output channel i of the tile, pair j): one 128-bit read-add-write per chunk and tile
        float4 old[NC][C::NTW];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int t = 0; t < C::NTW; ++t)
            old[c][t] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(acc) + (orow[c] ^ (t << 6)));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int t = 0; t < C::NTW; ++t) {
            float4 v = old[c][t];
            v.x += accr[c][t][0]; v.y += accr[c][t][1]; v.z += accr[c][t][2]; v.w += accr[c][t][3];
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(acc) + (orow[c] ^ (t << 6))) = v;
          }
      }
    };

    // one unit: (cur) operand set is multiplied while the (nxt) set is filled; returns false after the last unit
    auto step = [&](const float* (&gp_c)[NCMAX], int (&orow_c)[NCMAX], float4 (&G_c)[C::JQ][NCMAX], float4 (&W_c)[C::JQ][C::NTW],
                    const float* (&gp_n)[NCMAX], int (&orow_n)[NCMAX], float4 (&G_n)[C::JQ][NCMAX], float4 (&W_n)[C::JQ][C::NTW]) -> bool {
      TR();
      int cc_n = cc + 1, c0_n = c0, k_x = k_c, slot_x = slot_c, nch_x = nch_c;
      bool have = true;
      if (cc_n == ncc) {
        cc_n = 0;
        c0_n = c0 + NCMAX;
        if (c0_n >= nch_c) {
          if (P_n > 0) {
            k_x = k_n; slot_x = slot_c ^ 1; nch_x = (P_n + 15) >> 4; c0_n = 0;
            P_n = produce(slot_c, k_n);   // this item's records are in registers: its list slot is free
          } else {
            have = false; cc_n = cc; c0_n = c0;   // last unit: re-read our own operands (keeps the loads unconditional)
          }
        }
      }
      load_w_tail(wptr(k_c, cc), W_c);                 // (A) the rest of this unit's weights
      read_records(slot_x, c0_n, gp_n, orow_n);        // (B) gather of the next unit
      load_gather(gp_n, cc_n, G_n);
      load_w_head(wptr(k_x, cc_n), W_n);               // (C) first weights of the next unit
      TR();
      const int nc = min(NCMAX, nch_c - c0);
      const bool first_cc = cc == 0, last_cc = cc == ncc - 1;
      if constexpr (NCMAX >= 4) {
        if (nc == 4) compute(std::integral_constant<int, 4>{}, first_cc, last_cc, G_c, W_c, orow_c);
        if (nc == 3) compute(std::integral_constant<int, 3>{}, first_cc, last_cc, G_c, W_c, orow_c);
      }
      if (nc == 2) compute(std::integral_constant<int, 2>{}, first_cc, last_cc, G_c, W_c, orow_c);
      if (nc == 1) compute(std::integral_constant<int, 1>{}, first_cc, last_cc, G_c, W_c, orow_c);
      TR();
      k_c = k_x; slot_c = slot_x; nch_c = nch_x; c0 = c0_n; cc = cc_n;
      return have;
    };

    read_records(slot_c, 0, gpA, orowA);
    load_gather(gpA, 0, GA);
    load_w_head(wptr(k_c, 0), WA);
    while (true) {
      if (!step(gpA, orowA, GA, WA, gpB, orowB, GB, WB)) break;
      if (!step(gpB, orowB, GB, WB, gpA, orowA, GA, WA)) break;
    }
  }

  TR();
  // ---- epilogue
  list[lane] = (unsigned)grow0;   // the pair lists are dead: park the global row of every tile row there
  if constexpr (C::RND == 2) list[64 + lane] = (unsigned)grow1;
  constexpr int RPI = 64 / C::C4N;   // rows per store instruction
  const int er = lane / C::C4N, ec4 = lane % C::C4N;
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + ct0 + ec4 * 4);
  const float os = a.out_scale ? *a.out_scale : 1.0f;   // undoes the weight pre-scale of the SPLIT16 packing (a power of two)
  float mx = 0.f;
  const bool poisoned = MATH && !a.out_split && split16_poisoned(a.range);   // the network's fp32 output after an overflow upstream
  const float qnan = __builtin_nanf("");
  for (int r = er; r < rows_here; r += RPI) {
    const size_t o = (size_t)list[r];
    float4 v = *reinterpret_cast<const float4*>(acc + acc_off(r, ec4));
    v.x = v.x * os + b4.x; v.y = v.y * os + b4.y; v.z = v.z * os + b4.z; v.w = v.w * os + b4.w;
    if (a.l2norm) {
      float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
      for (int d = 1; d < C::C4N; d <<= 1) s += __shfl_xor(s, d, 64);
      const float nrm = sqrtf(s);
      v.x /= nrm; v.y /= nrm; v.z /= nrm; v.w /= nrm;
    } else {
      if (a.res) {
        const float4 q = MATH ? split16_load4(a.res + o * a.ld_res, ct0 + ec4 * 4)
                              : *reinterpret_cast<const float4*>(a.res + o * a.ld_res + ct0 + ec4 * 4);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      if (a.relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
    }
    const size_t oo = a.out_perm ? (size_t)a.out_perm[o] : o;
    if (a.out_split) { split16_track(mx, v); split16_store4(a.out + oo * a.ld_out, ct0 + ec4 * 4, v); }
    else *reinterpret_cast<float4*>(a.out + oo * a.ld_out + ct0 + ec4 * 4) = poisoned ? make_float4(qnan, qnan, qnan, qnan) : v;
  }
  if (MATH && a.out_split) split16_report(a.range, mx);
  TR();
}

template <int CTW, int BMW, int CC, int NCMAX, int OCC = 2>
void launch_wave_cfg(const SpconvArgs& a, hipStream_t st) {
  using C = WCfg<CTW, BMW, CC, NCMAX>;
  const long long tiles = (long long)(a.small_rows / (BMW / 2) + cdiv(a.n_out - a.small_rows, BMW)) * (a.cout / CTW);
  const int blocks = (cdiv(tiles, C::WPB) + 8 * EYOC_XG - 1) / (8 * EYOC_XG) * (8 * EYOC_XG);   // whole runs (surplus waves exit at once)
  if (a.math == 1) hipLaunchKernelGGL((spconv_wave_kernel<CTW, BMW, CC, NCMAX, OCC, 1>), dim3(blocks), dim3(C::WPB * 64), 0, st, a);
  else hipLaunchKernelGGL((spconv_wave_kernel<CTW, BMW, CC, NCMAX, OCC, 0>), dim3(blocks), dim3(C::WPB * 64), 0, st, a);
}

}  // namespace

namespace eyoc {

int launch_spconv_wave(const SpconvArgs& a_in, hipStream_t st) {
  SpconvArgs a = a_in;
  // few waves per SIMD slot (deep, narrow levels): cut the last-running quarter of the rows into half-height tiles
  // (measured: -5 % on the 256-channel level-3 layers of the bench, neutral from ~3 waves per slot upwards)
  const long long waves = (long long)cdiv(a.n_out, 64) * (a.cout >= 64 ? a.cout / 64 : 1);
  a.small_rows = waves < 6000 ? a.n_out / 4 / 64 * 64 : 0;
  const bool wide = spconv_cc(a.cin, a.cout) == 64;
  if (a.cout == 32) {
    wide ? launch_wave_cfg<32, 64, 64, 2>(a, st) : launch_wave_cfg<32, 64, 32, 2>(a, st);
  } else {
    // C_in >= 128 (two or more 64-channel slices per item): 128-row tiles at ONE wave per SIMD - up to four chunks
    // per unit (256 MFMAs), the whole 512-register file for one wave (no spills), the item's bookkeeping shared by
    // its slices.  Measured +10..20 % on the 128/256-channel layers, -5 % on the 64-channel ones (which keep 64-row
    // tiles at two waves per SIMD).
    const long long waves128 = (long long)cdiv(a.n_out, 128) * (a.cout / 64);
    if (wide && a.cin >= 128 && waves128 >= 512) {
      a.small_rows = waves128 < 3000 ? a.n_out / 4 / 128 * 128 : 0;
      launch_wave_cfg<64, 128, 64, 4, 1>(a, st);
    } else if (wide) {      // (also the 128-channel-input layers of a single pair: 72 waves of 128 rows on 1024 SIMDs last twice as long as 144 of 64)
      launch_wave_cfg<64, 64, 64, 2>(a, st);
    } else {
      launch_wave_cfg<64, 64, 32, 2>(a, st);
    }
  }
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

}  // namespace eyoc

#ifdef EYOC_TRACE
extern "C" int eyoc_debug_trace_wave(unsigned long long* out_host, size_t count) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_wtrace), count * sizeof(unsigned long long));
}
#endif
