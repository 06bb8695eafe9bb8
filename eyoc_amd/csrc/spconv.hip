// Sparse convolution on gfx950: gather -> fp32 MFMA -> LDS accumulate, output stationary.
//
// Replaces one MinkowskiConvolution / MinkowskiConvolutionTranspose (+ folded batch norm, residual
// add, ReLU, concat placement) of the reference's ResUNet (model/resunet.py:142-186,
// model/residual_block.py:37-53).
//
// Work decomposition (one workgroup = BM consecutive output rows x CT output channels):
//   1. rulebook compaction: for each kernel offset k the wave ballots which of the BM rows have a
//      neighbour and writes the compacted (input row, local output row) pairs to LDS.  On LiDAR
//      clouds only ~9 of 27 offsets are occupied per voxel and the occupied ones differ from row to
//      row, so a dense [BM x 27] sweep would spend 3x the MFMA time on zeros; compaction makes the
//      matrix-core work proportional to the true number of pairs.
//   2. for every non-empty (k, 32-channel slice of C_in): the W[k] slice (pre-packed on the host in
//      exact B-fragment order, so the copy is linear and fragment reads are conflict-free
//      ds_read_b128) is double-buffered through LDS; each wave takes 16-pair chunks, gathers their
//      input rows straight from global memory into A fragments (every lane reads 32 contiguous
//      bytes of one row) and runs v_mfma_f32_16x16x4_f32 (exact fp32).
//   3. after the last C_in slice of an offset the 16 x CT products are added into the workgroup's
//      LDS accumulator at their output rows - within one offset every output row occurs at most
//      once, so no atomics are needed and the summation order (k ascending) is deterministic.
//   4. epilogue from LDS: + bias (folded BN shift) (+ residual) -> ReLU -> (row L2 normalisation)
//      -> coalesced float4 row stores at the layer's column offset inside a concat buffer.
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "spconv.h"

using namespace eyoc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KMAX = 27;

// Diagnostic build (-DEYOC_TRACE, scripts/trace_spconv.py): a few workgroups stamp s_memtime at their
// phase boundaries.  Expands to nothing in the production build.
#ifdef EYOC_TRACE
constexpr int TRACE_STAMPS = 512, TRACE_WAVES = 8, TRACE_BLOCKS = 8;
__device__ unsigned long long g_trace[TRACE_BLOCKS * TRACE_WAVES * TRACE_STAMPS];
#define TR_DECL const int tr_blk = (blockIdx.x >= 300 && blockIdx.x < 300 + TRACE_BLOCKS && blockIdx.y == 0) ? (int)blockIdx.x - 300 : -1; int tr_n = 0
#define TR() do { if (tr_blk >= 0 && (threadIdx.x & 63) == 0 && tr_n < TRACE_STAMPS) \
    g_trace[(tr_blk * TRACE_WAVES + (threadIdx.x >> 6)) * TRACE_STAMPS + tr_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define TR_DECL
#define TR()
#endif

template <int CT, int BM, int NW, int CC>
struct Cfg {
  static constexpr int THREADS = NW * 64;
  static constexpr int NTILES = CT / 16;                       // 16-column tiles per block
  static constexpr int NWN = NW < NTILES ? NW : NTILES;        // waves along the output channels
  static constexpr int NWM = NW / NWN;                         // waves along the pair chunks
  static constexpr int NTW = NTILES / NWN;                     // column tiles per wave
  // pairs staged per item (<= 4 chunks).  Staging the whole 128-row tile per offset (27 items instead of
  // ~38) was measured 5 % SLOWER on MI355X: the time goes with the pairs, not with the item count.
  static constexpr int SUB = BM < 64 ? BM : 64;
  static constexpr int SCH = SUB / 16;                         // chunks per item
  static constexpr int MAXCW = SCH / NWM;                      // chunks per wave per item
  static constexpr int GCH = (SCH + NW - 1) / NW;              // chunks each wave gathers per item
  static constexpr int JQ = CC / 16;                           // 4-channel groups per lane group
  static constexpr int P = CC / 4;                             // 16-byte pieces per staged row
  static constexpr int RPI = 64 / P;                           // rows covered by one gather instruction
  static constexpr int NI = 16 / RPI;                          // gather instructions per chunk
  static constexpr int TILE_FLOATS = CC * CT;                  // one packed weight tile
  static constexpr int ACC_LD = CT;                            // accumulator row; float4 columns XOR-swizzled by the row
  static constexpr int C4N = CT / 4;                           // float4 columns per accumulator row
  static constexpr int MAX_ITEMS = KMAX * (BM / SUB);
  static constexpr int OFF_PAIR_IN = 0;                                      // u32 [KMAX][BM]: input row << 8 | local output row
  static constexpr int OFF_CNT = OFF_PAIR_IN + KMAX * BM * 4;                // int [KMAX] counts + [1] n_items
  static constexpr int OFF_ITEMS = OFF_CNT + (KMAX + 1) * 4;                 // u32 [MAX_ITEMS + 2]: k | sub << 8 | pairs << 16
  static constexpr int OFF_A = (OFF_ITEMS + (MAX_ITEMS + 2) * 4 + 15) / 16 * 16;  // float [SUB][CC] (XOR-swizzled pieces)
  static constexpr int OFF_ACC = OFF_A + SUB * CC * 4;                       // float [BM][ACC_LD]
  static constexpr int OFF_NORM = OFF_ACC + BM * ACC_LD * 4;                 // float [BM]
  static constexpr int LDS_BYTES = OFF_NORM + BM * 4;
  static_assert(NWN * NWM == NW && NTW * NWN == NTILES && MAXCW >= 1 && MAXCW * NWM == SCH, "wave grid must tile the block");
  static_assert(BM % SUB == 0 && BM <= 256, "BM must be a multiple of SUB; the local output row is packed in 8 bits");
  static_assert(LDS_BYTES <= 80 * 1024, "two workgroups must fit one CU");
};

// One workgroup = BM output rows x CT output channels.  Waves split the OUTPUT CHANNELS (and, when
// CT is narrow, the pair chunks): every wave then has the same MFMA count for every offset, however
// few pairs the offset has, so nobody idles at the per-item barriers.  Per item (offset k, <= 64
// compacted pairs, one C_in slice):
//   - the input rows of the pairs are gathered cooperatively with full-row coalesced loads
//     (consecutive lanes read consecutive 16 B of one row) into registers one item ahead, then into
//     an LDS tile whose 16-byte pieces are XOR-swizzled by the row so that the MFMA A-fragment reads
//     (ds_read_b128, lane = (row, channel group)) are bank-conflict free without padding;
//   - each wave reads ITS slice of W[k] (host-packed in B-fragment order) straight from L2 into
//     registers, one item ahead - weights never go through LDS;
//   - v_mfma_f32_16x16x4_f32 over the chunk x column-tile grid, independent accumulators back to back;
//   - after the last C_in slice the products are added to the LDS accumulator rows of their outputs.
template <int CT, int BM, int NW, int CC>
__global__ __launch_bounds__(NW * 64, (NW >= 8 ? 4 : 2)) void spconv_kernel(SpconvArgs a) {
  using C = Cfg<CT, BM, NW, CC>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
  unsigned int* pairs = reinterpret_cast<unsigned int*>(smem + C::OFF_PAIR_IN);
  int* cnt = reinterpret_cast<int*>(smem + C::OFF_CNT);
  int* n_items_p = cnt + KMAX;
  unsigned int* items = reinterpret_cast<unsigned int*>(smem + C::OFF_ITEMS);
  float* atile = reinterpret_cast<float*>(smem + C::OFF_A);
  float* acc = reinterpret_cast<float*>(smem + C::OFF_ACC);
  float* rnorm = reinterpret_cast<float*>(smem + C::OFF_NORM);
  // accumulator element (row, float4 column): columns are XOR-swizzled by the row so that the 128-bit
  // read-add-writes of lanes holding different rows spread over the banks without padding
  auto acc_off = [](int row, int c4) { return row * C::ACC_LD + ((c4 ^ row) & (C::C4N - 1)) * 4; };

  TR_DECL;
  TR();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * BM;
  const int slice = blockIdx.y, n_slices = gridDim.y;
  const int ct0 = slice * CT;
  const int ncc = a.cin / CC;
  const int rows_here = min(BM, a.n_out - row0);

  // ---- phase 0: zero the accumulator, compact the rulebook of this row tile
  for (int i = tid; i < BM * C::ACC_LD; i += C::THREADS) acc[i] = 0.0f;
  if (a.nbr) {
    // all rulebook loads of this wave are issued back to back (one memory latency, not one per offset)
    constexpr int KPW = (KMAX + NW - 1) / NW, RND = (BM + 63) / 64;
    int idxv[KPW][RND];
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
      const int k = wave + NW * kk;
#pragma unroll
      for (int r = 0; r < RND; ++r) {
        const int row = r * 64 + lane;
        idxv[kk][r] = (k < a.K && row < rows_here) ? a.nbr[(size_t)k * a.n_out + row0 + row] : -1;
      }
    }
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
      const int k = wave + NW * kk;
      if (k < a.K) {
        int base = 0;
#pragma unroll
        for (int r = 0; r < RND; ++r) {
          const int idx = idxv[kk][r];
          const bool valid = idx >= 0;
          const unsigned long long m = __ballot(valid);
          if (valid) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            pairs[k * BM + pos] = ((unsigned)idx << 8) | (unsigned)(r * 64 + lane);
          }
          base += __popcll(m);
        }
        if (lane == 0) cnt[k] = base;
      }
    }
  } else {  // identity map (1x1 convolution)
    for (int r = tid; r < BM; r += C::THREADS) {
      pairs[r] = ((unsigned)(row0 + r) << 8) | (unsigned)r;
    }
    if (tid == 0) cnt[0] = rows_here;
  }
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    // (offset split: workgroup z of gridDim.z walks the offsets [K z / Z, K (z + 1) / Z) only)
    const int k_lo = a.K * (int)blockIdx.z / (int)gridDim.z, k_hi = a.K * ((int)blockIdx.z + 1) / (int)gridDim.z;
    for (int k = k_lo; k < k_hi; ++k)
      for (int s0 = 0; s0 < cnt[k]; s0 += C::SUB)
        items[n++] = (unsigned)k | ((unsigned)(s0 / C::SUB) << 8) | ((unsigned)min(C::SUB, cnt[k] - s0) << 16);
    items[n] = items[n + 1] = n ? items[n - 1] : 0u;   // padding: the pipeline decodes two items ahead
    *n_items_p = n;
  }
  __syncthreads();
  // Everything that steers control flow below is forced into SGPRs with readfirstlane: values read
  // from LDS are "divergent" to the compiler, and a divergent branch around an MFMA makes it copy
  // every accumulator at each region boundary (thousands of v_mov per item).
  const int n_items = __builtin_amdgcn_readfirstlane(*n_items_p);
  TR();

  if (n_items > 0) {  // block-uniform
    const int r16 = lane & 15, g = lane >> 4;
    const int grow = lane / C::P, gpiece = lane % C::P;   // gather role: row within an instruction, 16 B piece
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wn_s = wave_s % C::NWN, wm_s = wave_s / C::NWN;
    const float4* wbase = reinterpret_cast<const float4*>(a.w);

    float4 stage[C::GCH][C::NI];
    float4 bnxt[C::NTW][C::JQ];
    f32x4 accreg[C::MAXCW][C::NTW];

    struct Item { int k, base, n_here, nch; };
    auto unpack = [&](unsigned rec) {   // rec must already be wave-uniform (readfirstlane'd)
      Item d;
      d.k = rec & 255u;
      d.base = ((rec >> 8) & 255u) * C::SUB;
      d.n_here = rec >> 16;
      d.nch = (d.n_here + 15) >> 4;
      return d;
    };
    // ---- pipeline pieces, ordered by hand inside one iteration so that every LDS / global latency
    // sits behind MFMAs instead of in front of them (only two waves share a SIMD)
    // (1) LDS reads of the NEXT item's gather indices
    auto read_gather_rows = [&](const Item& d, int (&rows)[C::GCH][C::NI]) {
#pragma unroll
      for (int j = 0; j < C::GCH; ++j)
#pragma unroll
        for (int i = 0; i < C::NI; ++i) {
          const int pl = (wave_s + NW * j) * 16 + i * C::RPI + grow;      // pair index inside the item
          rows[j][i] = pl < d.n_here ? (int)(pairs[d.k * BM + d.base + pl] >> 8) : -1;
        }
    };
    // (2) global loads of the NEXT item: this wave's weight slice and its share of the gathered rows
    auto issue_loads = [&](const Item& d, int cc, const int (&rows)[C::GCH][C::NI]) {
      const float4* wt = wbase + ((size_t)(d.k * n_slices + slice) * ncc + cc) * (C::TILE_FLOATS / 4) +
                         (size_t)(wn_s * C::NTW) * C::JQ * 64 + lane;
#pragma unroll
      for (int t = 0; t < C::NTW; ++t)
#pragma unroll
        for (int q = 0; q < C::JQ; ++q) bnxt[t][q] = wt[(t * C::JQ + q) * 64];
#pragma unroll
      for (int j = 0; j < C::GCH; ++j)
#pragma unroll
        for (int i = 0; i < C::NI; ++i) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rows[j][i] >= 0)
            v = *reinterpret_cast<const float4*>(a.in + (size_t)rows[j][i] * a.ld_in + cc * CC + gpiece * 4);
          stage[j][i] = v;
        }
    };
    // (3) stage -> LDS tile; piece index XOR (row & 15) keeps the fragment reads conflict free
    auto commit = [&](const Item& d) {   // only the chunks the item really has
#pragma unroll
      for (int j = 0; j < C::GCH; ++j) {
        const int chunk = wave_s + NW * j;
        if (chunk < d.nch) {
#pragma unroll
          for (int i = 0; i < C::NI; ++i) {
            const int row = chunk * 16 + i * C::RPI + grow;
            *reinterpret_cast<float4*>(atile + row * CC + ((gpiece ^ (row & 15)) % C::P) * 4) = stage[j][i];
          }
        }
      }
    };

    // one iteration for a wave that owns NC chunks of the current item: straight-line code
    auto body = [&](auto nc_tag, const Item& cur, int cc, const Item& nxt, int cc_n,
                    const float4 (&bcur)[C::NTW][C::JQ]) {
      constexpr int NC = decltype(nc_tag)::value;
      float4 af[NC > 0 ? NC : 1];
      auto load_af = [&](int q) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int row = (wm_s + C::NWM * c) * 16 + r16;
          af[c] = *reinterpret_cast<const float4*>(atile + row * CC + (((q * 4 + g) ^ r16) % C::P) * 4);
        }
      };
      auto mfma_q = [&](int q) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const float av = e == 0 ? af[c].x : e == 1 ? af[c].y : e == 2 ? af[c].z : af[c].w;
#pragma unroll
            for (int t = 0; t < C::NTW; ++t) {
              const float bv = e == 0 ? bcur[t][q].x : e == 1 ? bcur[t][q].y : e == 2 ? bcur[t][q].z : bcur[t][q].w;
              accreg[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, accreg[c][t], 0, 0, 0);
            }
          }
      };
      const bool do_flush = cc == ncc - 1;
      if (cc == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int t = 0; t < C::NTW; ++t) accreg[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      load_af(0);
      int rows[C::GCH][C::NI];
      read_gather_rows(nxt, rows);
      int orow[NC > 0 ? NC : 1];   // local output row of this lane's pair in each chunk (-1: padding)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int pl = (wm_s + C::NWM * c) * 16 + r16;
        orow[c] = pl < cur.n_here ? (int)(pairs[cur.k * BM + cur.base + pl] & 255u) : -1;
      }
      mfma_q(0);
      issue_loads(nxt, cc_n, rows);
#pragma unroll
      for (int q = 1; q < C::JQ; ++q) {
        load_af(q);
        mfma_q(q);
      }
      if (do_flush) {
        // The weights are the MFMA A operand and the gathered rows the B operand, so
        // D[i = 4*(lane>>4) + reg][j = lane&15] is (output channel i, pair j): every lane holds four
        // CONSECUTIVE channels of one output row -> one 128-bit LDS read-add-write per chunk and tile.
        // All reads are issued before any write (the rows of one item are distinct, no aliasing).
        float4 old[NC > 0 ? NC : 1][C::NTW];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int t = 0; t < C::NTW; ++t)
            if (orow[c] >= 0)
              old[c][t] = *reinterpret_cast<const float4*>(acc + acc_off(orow[c], (wn_s * C::NTW + t) * 4 + g));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int t = 0; t < C::NTW; ++t)
            if (orow[c] >= 0) {
              float4 v = old[c][t];
              v.x += accreg[c][t][0]; v.y += accreg[c][t][1]; v.z += accreg[c][t][2]; v.w += accreg[c][t][3];
              *reinterpret_cast<float4*>(acc + acc_off(orow[c], (wn_s * C::NTW + t) * 4 + g)) = v;
            }
      }
    };

    int ii = 0, cc = 0;
    Item cur = unpack(__builtin_amdgcn_readfirstlane((int)items[0]));
    {
      int rows[C::GCH][C::NI];
      read_gather_rows(cur, rows);
      issue_loads(cur, 0, rows);
    }
    commit(cur);
    __syncthreads();

    while (true) {
      TR();
      float4 bcur[C::NTW][C::JQ];
#pragma unroll
      for (int t = 0; t < C::NTW; ++t)
#pragma unroll
        for (int q = 0; q < C::JQ; ++q) bcur[t][q] = bnxt[t][q];
      // next (item, cc); the last iteration re-reads its own operands (harmless, keeps staging in VGPRs)
      int ii_n = ii, cc_n = cc + 1;
      if (cc_n == ncc) { cc_n = 0; ii_n = ii + 1; }
      const bool last = ii_n >= n_items;
      if (last) { ii_n = ii; cc_n = cc; }
      const Item nxt = unpack(__builtin_amdgcn_readfirstlane((int)items[ii_n]));

      // chunks of this item owned by this wave's row group (chunk = wm + NWM * c)
      const int my_nc = (cur.nch - wm_s + C::NWM - 1) / C::NWM;
      if constexpr (C::MAXCW >= 4) {
        if (my_nc >= 4) body(std::integral_constant<int, 4>{}, cur, cc, nxt, cc_n, bcur);
        else if (my_nc == 3) body(std::integral_constant<int, 3>{}, cur, cc, nxt, cc_n, bcur);
      }
      if constexpr (C::MAXCW >= 2) {
        if (my_nc == 2) body(std::integral_constant<int, 2>{}, cur, cc, nxt, cc_n, bcur);
      }
      if (my_nc == 1) body(std::integral_constant<int, 1>{}, cur, cc, nxt, cc_n, bcur);
      if (my_nc <= 0) body(std::integral_constant<int, 0>{}, cur, cc, nxt, cc_n, bcur);

      TR();
      __syncthreads();   // everyone is done reading the staged tile
      TR();
      if (last) break;
      commit(nxt);
      TR();
      __syncthreads();
      ii = ii_n; cc = cc_n; cur = nxt;
    }
  }  // n_items > 0
  __syncthreads();
  TR();

  // ---- epilogue
  constexpr int C4 = CT / 4;
  if (gridDim.z > 1) {      // offset split: the raw sums of this share; k_offset_reduce adds the shares and applies the epilogue
    float* part = a.offset_part + (size_t)blockIdx.z * a.n_out * a.cout;
    for (int i = tid; i < BM * C4; i += C::THREADS) {
      const int r = i / C4, c4 = i % C4;
      if (r >= rows_here) continue;
      *reinterpret_cast<float4*>(part + (size_t)(row0 + r) * a.cout + ct0 + c4 * 4) = *reinterpret_cast<const float4*>(acc + acc_off(r, c4));
    }
    return;
  }
  if (a.l2norm) {
    for (int i = tid; i < BM * C4; i += C::THREADS) {
      const int r = i / C4, c4 = i % C4;
      float4* p = reinterpret_cast<float4*>(acc + acc_off(r, c4));
      float4 v = *p;
      if (a.bias) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias + ct0 + c4 * 4);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      *p = v;
    }
    __syncthreads();
    for (int r = tid; r < BM; r += C::THREADS) {
      float s = 0.0f;
      for (int c = 0; c < CT; ++c) {
        const float v = acc[acc_off(r, c >> 2) + (c & 3)];
        s += v * v;
      }
      rnorm[r] = sqrtf(s);
    }
    __syncthreads();
  }
  for (int i = tid; i < BM * C4; i += C::THREADS) {
    const int r = i / C4, c4 = i % C4;
    if (r >= rows_here) continue;
    const size_t o = (size_t)(row0 + r);
    float4 v = *reinterpret_cast<const float4*>(acc + acc_off(r, c4));
    if (a.l2norm) {
      const float nrm = rnorm[r];
      v.x /= nrm; v.y /= nrm; v.z /= nrm; v.w /= nrm;
    } else {
      if (a.bias) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias + ct0 + c4 * 4);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if (a.res) {
        const float4 q = *reinterpret_cast<const float4*>(a.res + o * a.ld_res + ct0 + c4 * 4);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      if (a.relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
    }
    *reinterpret_cast<float4*>(a.out + (a.out_perm ? (size_t)a.out_perm[o] : o) * a.ld_out + ct0 + c4 * 4) = v;
  }
  TR();
}

// out = sum of the offset shares (in share order: deterministic) + bias + residual, ReLU
__global__ __launch_bounds__(256) void k_offset_reduce(SpconvArgs a) {
  const int c4n = a.cout / 4;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)a.n_out * c4n) return;
  const size_t o = (size_t)(i / c4n);
  const int c = (int)(i % c4n) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < a.offset_split; ++z) {
    const float4 q = *reinterpret_cast<const float4*>(a.offset_part + ((size_t)z * a.n_out + o) * a.cout + c);
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
  }
  if (a.bias) {
    const float4 b = *reinterpret_cast<const float4*>(a.bias + c);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (a.res) {
    const float4 q = *reinterpret_cast<const float4*>(a.res + o * a.ld_res + c);
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
  }
  if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  *reinterpret_cast<float4*>(a.out + (a.out_perm ? (size_t)a.out_perm[o] : o) * a.ld_out + c) = v;
}

template <int CT, int BM, int NW, int CC>
void launch_cfg(const SpconvArgs& a, hipStream_t st) {
  dim3 grid(cdiv(a.n_out, BM), a.cout / CT, a.offset_split > 1 ? a.offset_split : 1);
  hipLaunchKernelGGL((spconv_kernel<CT, BM, NW, CC>), grid, dim3(NW * 64), 0, st, a);
  if (a.offset_split > 1)
    hipLaunchKernelGGL(k_offset_reduce, dim3(cdiv((long long)a.n_out * (a.cout / 4), 256)), dim3(256), 0, st, a);
}

template <int CT, int CC>
void launch_ct(const SpconvArgs& a, hipStream_t st) {
  // Row tile: as large as still gives the 256 CUs a few workgroups each.  Narrow layers (CT <= 64)
  // run 8 waves (4 column groups x 2 chunk groups): measured +3..6 % over 4 waves on MI355X.
  const long long slices = a.cout / CT;
  if constexpr (CT <= 64) {
    if ((long long)cdiv(a.n_out, 128) * slices >= 1024) {
      launch_cfg<CT, 128, 8, CC>(a, st);
    } else if ((long long)cdiv(a.n_out, 64) * slices >= 512) {
      launch_cfg<CT, 64, 8, CC>(a, st);
    } else {
      launch_cfg<CT, 32, 2, CC>(a, st);
    }
  } else {
    if ((long long)cdiv(a.n_out, 64) * slices >= 512) {
      launch_cfg<CT, 64, 4, CC>(a, st);
    } else {
      launch_cfg<CT, 32, 4, CC>(a, st);   // 8 column tiles: keep 4 waves (2 tiles each), 2 chunks per item
    }
  }
}

// ------------------------------------------------------------------------------------------------
// first convolution (C_in = 1 in production): one lane per output row probes the ks^3 offsets in
// the level-0 hash table and accumulates f * W[k] from an LDS copy of the (BN-folded) kernel.
// ------------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void conv1_kernel(Conv1Args a) {
  constexpr int W_FLOATS = 8192;  // 32 KB of LDS for a slab of offsets
  __shared__ __attribute__((aligned(16))) float ws[W_FLOATS];
  const int o = blockIdx.x * 256 + threadIdx.x;
  const bool ok = o < a.n;
  int4 c = make_int4(0, 0, 0, 0);
  if (ok) c = reinterpret_cast<const int4*>(a.coords)[o];
  float acc[COUT];
#pragma unroll
  for (int i = 0; i < COUT; ++i) acc[i] = 0.0f;
  const int K = a.ks * a.ks * a.ks, r = a.ks / 2;
  const int per_k = a.cin * COUT;
  const int kslab = W_FLOATS / per_k;
  for (int k0 = 0; k0 < K; k0 += kslab) {
    const int kn = min(kslab, K - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < kn * per_k; i += 256) ws[i] = a.w[(size_t)k0 * per_k + i];
    __syncthreads();
    for (int kk = 0; kk < kn; ++kk) {
      const int k = k0 + kk;
      const int dx = k % a.ks - r, dy = (k / a.ks) % a.ks - r, dz = k / (a.ks * a.ks) - r;
      int idx = -1;
      if (ok) idx = hash_lookup(a.table, pack_key(c.x, c.y + dx, c.z + dy, c.w + dz));
      if (__ballot(idx >= 0) == 0ull) continue;
      for (int ci = 0; ci < a.cin; ++ci) {
        const float f = idx >= 0 ? a.in[(size_t)(a.in_perm ? a.in_perm[idx] : idx) * a.cin + ci] : 0.0f;
        const float* wk = ws + (kk * a.cin + ci) * COUT;
#pragma unroll
        for (int i = 0; i < COUT; ++i) acc[i] = fmaf(f, wk[i], acc[i]);
      }
    }
  }
  if (ok) {
    float* dst = a.out + (size_t)o * a.ld_out;
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < COUT; i += 4) {
      float4 v;
      v.x = acc[i] + a.bias[i]; v.y = acc[i + 1] + a.bias[i + 1];
      v.z = acc[i + 2] + a.bias[i + 2]; v.w = acc[i + 3] + a.bias[i + 3];
      split16_track(mx, v);
      if (a.out_split) split16_store4(dst, i, v);
      else *reinterpret_cast<float4*>(dst + i) = v;
    }
    if (a.out_split) split16_report(a.range, mx);
  }
}

// Same convolution without hash probes: the ks^3 window (ks = 3 or 5) of a voxel lies inside the 27
// level-1 blocks around its parent, whose rows come from the level-1 stride-1 table and whose
// occupants come from the 8-entry child vectors (one 32-byte read per block).  ~1 KB of table reads
// per output row instead of 125 probes of a 64-bit-key hash table.
template <int COUT>
__global__ __launch_bounds__(256) void conv1_tree_kernel(Conv1Args a) {
  constexpr int LDW = COUT + 4;   // padded weight rows: lanes hold different offsets k, keep them on different banks
  extern __shared__ float wsm[];
  const int K = a.ks * a.ks * a.ks, r = a.ks / 2;
  for (int i = threadIdx.x; i < K * a.cin * COUT; i += 256) wsm[(i / COUT) * LDW + i % COUT] = a.w[i];
  __syncthreads();
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= a.n) return;
  const int4 c = reinterpret_cast<const int4*>(a.coords)[o];
  const int bx = c.y & 1, by = c.z & 1, bz = c.w & 1;
  const int p = a.parent[o];
  float acc[COUT];
#pragma unroll
  for (int i = 0; i < COUT; ++i) acc[i] = 0.0f;
#pragma unroll 1
  for (int kc = 0; kc < 27; ++kc) {
    const int ox = kc % 3 - 1, oy = (kc / 3) % 3 - 1, oz = kc / 9 - 1;
    const int B = kc == 13 ? p : a.s1c[(size_t)kc * a.nc + p];
    if (B < 0) continue;
    const int4 lo = reinterpret_cast<const int4*>(a.children)[2 * (size_t)B];
    const int4 hi = reinterpret_cast<const int4*>(a.children)[2 * (size_t)B + 1];
    const int ch[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int cs = 0; cs < 8; ++cs) {
      const int idx = ch[cs];
      const int dx = 2 * ox + (cs & 1) - bx, dy = 2 * oy + ((cs >> 1) & 1) - by, dz = 2 * oz + (cs >> 2) - bz;
      if (idx < 0 || dx < -r || dx > r || dy < -r || dy > r || dz < -r || dz > r) continue;
      const int k = (dx + r) + a.ks * (dy + r) + a.ks * a.ks * (dz + r);
      for (int ci = 0; ci < a.cin; ++ci) {
        const float f = a.in[(size_t)(a.in_perm ? a.in_perm[idx] : idx) * a.cin + ci];
        const float4* wk = reinterpret_cast<const float4*>(wsm + (k * a.cin + ci) * LDW);
#pragma unroll
        for (int i = 0; i < COUT / 4; ++i) {
          const float4 w4 = wk[i];
          acc[4 * i] = fmaf(f, w4.x, acc[4 * i]); acc[4 * i + 1] = fmaf(f, w4.y, acc[4 * i + 1]);
          acc[4 * i + 2] = fmaf(f, w4.z, acc[4 * i + 2]); acc[4 * i + 3] = fmaf(f, w4.w, acc[4 * i + 3]);
        }
      }
    }
  }
  float* dst = a.out + (size_t)o * a.ld_out;
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < COUT; i += 4) {
    float4 v;
    v.x = acc[i] + a.bias[i]; v.y = acc[i + 1] + a.bias[i + 1];
    v.z = acc[i + 2] + a.bias[i + 2]; v.w = acc[i + 3] + a.bias[i + 3];
    split16_track(mx, v);
    if (a.out_split) split16_store4(dst, i, v);
    else *reinterpret_cast<float4*>(dst + i) = v;
  }
  if (a.out_split) split16_report(a.range, mx);
}

// The same first convolution (C_in = 1, 5^3 or 3^3 window, 32 output channels) as MFMA work.  conv1_tree_kernel walks the
// 216 (coarse block, child) slots of a row in lock step - a wave executes every slot although ~12 % of its lanes hold a
// neighbour there - and pays 32 FMAs with lane-varying weight rows per slot: 1.65 ms on the 3.8 M-row batch.  Here a wave
// takes 16 output rows, its four 16-lane groups walk a quarter of the coarse blocks each (a neighbour's feature goes to
// column k = its window position of the row's line in an LDS tile, as fp16 hi / lo halves), and the products are ONE
// [16 x 128] x [128 x 32] split16 GEMM per tile: 4 k-steps x 3 terms x 2 channel tiles = 24 MFMAs, weights resident in
// registers (pre-scaled by 2^8 so that their lo halves stay normal).
__global__ __launch_bounds__(256, 4) void conv1_mfma_kernel(Conv1Args a) {
  constexpr int COUT = 32, KP = 128, KPS = KP + 8, NT = COUT / 16, NQ = KP / 32;
  // row stride KP + 8 halves = 272 B = 68 dwords: the 16 rows of a tile that write the SAME window position (one ds_write_b16, lanes of
  // one parity class) land in 16 different banks - at 256 B they all hit one (a 16-way conflict on every write: measured 1.1 -> ? ms)
  __shared__ __attribute__((aligned(16))) _Float16 xt[4][2][16][KPS];   // [wave][hi / lo][row][window position]
  __shared__ __attribute__((aligned(8))) signed char ktab[8][216];                                // window position of slot (block kc, child cs) for a row of parity class cls, -1: outside
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int K = a.ks * a.ks * a.ks, r = a.ks / 2;
  for (int e = threadIdx.x; e < 8 * 216; e += 256) {
    const int cls = e / 216, sl = e % 216, kc = sl >> 3, cs = sl & 7;
    const int dx = 2 * (kc % 3 - 1) + (cs & 1) - (cls & 1), dy = 2 * ((kc / 3) % 3 - 1) + ((cs >> 1) & 1) - ((cls >> 1) & 1),
              dz = 2 * (kc / 9 - 1) + (cs >> 2) - (cls >> 2);
    const bool in = dx >= -r && dx <= r && dy >= -r && dy <= r && dz >= -r && dz <= r;
    ktab[cls][sl] = (signed char)(in ? (dx + r) + a.ks * (dy + r) + a.ks * a.ks * (dz + r) : -1);
  }
  __syncthreads();
  // weight fragments (A operand): lane (g, j): W[k = 32 q + 8 g + i][cout = 16 t + j] * 2^sh, split (2^sh lifts the layer's
  // largest weight into [256, 512): the lo halves stay normal and no hi half leaves the fp16 range; 2^8 without a.wscale)
  const float wsc = a.wscale ? a.wscale[0] : 256.0f, iwsc = a.wscale ? a.wscale[1] : 1.0f / 256.0f;
  half8_t wh[NQ][NT], wl[NQ][NT];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 32 * q + 8 * g + i;
        const float w = k < K ? a.w[(size_t)k * COUT + 16 * t + j] * wsc : 0.0f;
        const _Float16 h = (_Float16)w;
        wh[q][t][i] = h;
        wl[q][t][i] = (_Float16)(w - (float)h);
      }
  _Float16 (*xh)[KPS] = xt[wave][0];
  _Float16 (*xl)[KPS] = xt[wave][1];
  const int n_tiles = (a.n + 15) / 16;
  // persistent waves: the weight fragments above are loaded once per wave, not once per 16 rows
  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
  const int o = tile * 16 + j;                                      // this lane's row (the 4 groups share it)
  // clear the tile: 16 rows x 256 B x 2 = 8 KB per wave
  {
    float4* z = reinterpret_cast<float4*>(&xt[wave][0][0][0]);
#pragma unroll
    for (int i = 0; i < (2 * 16 * KPS * 2 / 16 + 63) / 64; ++i)
      if (i * 64 + lane < 2 * 16 * KPS * 2 / 16) z[i * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (o < a.n) {
    const int4 c = reinterpret_cast<const int4*>(a.coords)[o];
    const signed char* kt = ktab[(c.y & 1) | ((c.z & 1) << 1) | ((c.w & 1) << 2)];
    const int p = a.parent[o];
    // per half of the group's (up to 7) coarse blocks: three rounds of independent loads - the blocks, their child
    // vectors, the features of ALL their children back to back (absent / outside slots read row 0; a load inside a
    // branch would wait for its data before the next slot starts) - then the LDS writes
    const float* __restrict__ fin = a.in;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      constexpr int NB = 4;
      int Bk[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int kc = g + 4 * (half * NB + i);
        Bk[i] = kc >= 27 ? -1 : kc == 13 ? p : a.s1c[(size_t)kc * a.nc + p];
      }
      int ch[NB][8];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        int4 lo = make_int4(-1, -1, -1, -1), hi = make_int4(-1, -1, -1, -1);
        if (Bk[i] >= 0) {
          lo = reinterpret_cast<const int4*>(a.children)[2 * (size_t)Bk[i]];
          hi = reinterpret_cast<const int4*>(a.children)[2 * (size_t)Bk[i] + 1];
        }
        ch[i][0] = lo.x; ch[i][1] = lo.y; ch[i][2] = lo.z; ch[i][3] = lo.w;
        ch[i][4] = hi.x; ch[i][5] = hi.y; ch[i][6] = hi.z; ch[i][7] = hi.w;
      }
      int kk[NB][8];
      float fv[NB][8];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int kc = g + 4 * (half * NB + i);
        const unsigned long long k8 = kc < 27 ? *reinterpret_cast<const unsigned long long*>(kt + kc * 8) : ~0ull;   // 8 positions
#pragma unroll
        for (int cs = 0; cs < 8; ++cs) {
          const int idx = ch[i][cs];
          const int kpos = (int)(signed char)(k8 >> (8 * cs));
          const bool ok = idx >= 0 && kpos >= 0;
          kk[i][cs] = ok ? kpos : -1;
          fv[i][cs] = fin[ok ? idx : 0];
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int cs = 0; cs < 8; ++cs)
          if (kk[i][cs] >= 0) {
            const float f = fv[i][cs];
            const _Float16 h = (_Float16)f;
            xh[j][kk[i][cs]] = h;
            xl[j][kk[i][cs]] = (_Float16)(f - (float)h);
          }
    }
  }
  // the wave's LDS operations execute in order; the wait makes the writes of the other lanes visible to the reads below
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const half8_t vh = *reinterpret_cast<const half8_t*>(&xh[j][32 * q + 8 * g]);   // B operand: X[k = 32 q + 8 g + i][row j]
    const half8_t vl = *reinterpret_cast<const half8_t*>(&xl[j][32 * q + 8 * g]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q][t], vh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q][t], vl, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[q][t], vh, acc[t], 0, 0, 0);
    }
  }
  if (o < a.n) {
    // lane (g, j) holds channels 16 t + 4 g .. +3 of row o
    float* dst = a.out + (size_t)o * a.ld_out;
    float mx = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ch0 = 16 * t + 4 * g;
      const float4 b = *reinterpret_cast<const float4*>(a.bias + ch0);
      const float4 v = make_float4(acc[t][0] * iwsc + b.x, acc[t][1] * iwsc + b.y,
                                   acc[t][2] * iwsc + b.z, acc[t][3] * iwsc + b.w);
      split16_track(mx, v);
      if (a.out_split) split16_store4(dst, ch0, v);
      else *reinterpret_cast<float4*>(dst + ch0) = v;
    }
    if (a.out_split) split16_report(a.range, mx);
  }
  __builtin_amdgcn_wave_barrier();                                   // the next tile's clear stays behind this tile's operand reads
  }
}

// fp32 rows <-> SPLIT16 rows (tests, and callers that feed eyoc_spconv_ex directly)
__global__ void k_split16_encode(const float* __restrict__ in, int n, int c, int ld_in, float* __restrict__ out, int ld_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = t / (c / 4), q = t % (c / 4);
  if (r >= n) return;
  split16_store4(out + (size_t)r * ld_out, q * 4, *reinterpret_cast<const float4*>(in + (size_t)r * ld_in + q * 4));
}
__global__ void k_split16_decode(const float* __restrict__ in, int n, int c, int ld_in, float* __restrict__ out, int ld_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = t / (c / 4), q = t % (c / 4);
  if (r >= n) return;
  *reinterpret_cast<float4*>(out + (size_t)r * ld_out + q * 4) = split16_load4(in + (size_t)r * ld_in, q * 4);
}

}  // namespace

namespace eyoc {

// which kernels run is decided per ctx (eyoc_ctx::Knobs, common.h): automatic by default, forced by the eyoc_spconv_select_* setters.
// The staged kernel for the transposed convolutions is on by default on Z-ordered maps: level with the row-stationary kernel in
// windowed pattern order in time (0.68 / 1.06 / 1.44 vs 0.63 / 1.02 / 1.57 ms on the bench's three layers, + 0.25 ms of rulebooks in
// the map build; the step is the same within noise) and 18 GB less HBM traffic per forward

// Which tile-record kernel takes a split16 layer: 1 the staged stride-1 kernel (spconv_st.hip), 2 class-major transposed tiles
// (spconv_upc.hip), 3 Morton transposed tiles (spconv_up.hip), 4 the staged kernel on 128-row tiles of a strided table, 5 ... of a stride-1 table (>= 128 output channels), 0 none - a gathering kernel, which reads a.nbr.  ONE predicate for the
// launcher below and for eyoc_model_forward, which fills a lazily skipped table only when a layer really reads it (common.h eyoc_maps).
int spconv_record_path(const SpconvArgs& a) {
  if (a.math != 1 || a.K != 27 || a.l2norm) return 0;
  const int use_rs = knobs_of(a.ctx).split16_kernel;
  if (use_rs == 0) return 0;
  // stride-1 table with local rulebooks: staged kernel (its workgroups cover 64 - or 32 - output channels each and
  // 1, 2, 4 or 8 of them share a tile; other widths stay on the gathering kernels)
  const int st_ctg = a.cout >= 64 ? 64 : 32;
  const bool st_ok = a.cout % st_ctg == 0 && a.cout / st_ctg <= 8 && 8 % (a.cout / st_ctg) == 0 && a.cin % 32 == 0;
  if (a.local128 && a.cout % 128 == 0 && a.cout / 128 <= 8 && 8 % (a.cout / 128) == 0 && a.cin % 32 == 0) return 5;
  if (a.local && st_ok) return 1;
  if (a.local_down && a.cout % 64 == 0 && a.cout / 64 <= 8 && 8 % (a.cout / 64) == 0 && a.cin % 32 == 0) return 4;
  if (a.local_upc && !a.res && !a.out_perm && a.cout % 64 == 0) return 2;
  if (a.local_up && a.cout % 64 == 0) return 3;
  return 0;
}

int launch_spconv(const SpconvArgs& a, hipStream_t st) {
  EYOC_REQUIRE(a.in && a.w && a.out, EYOC_ERR_INVALID, "spconv: NULL tensor");
  EYOC_REQUIRE(a.n_out >= 0, EYOC_ERR_INVALID, "spconv: n_out %d", a.n_out);
  EYOC_REQUIRE(a.cin > 0 && a.cin % 32 == 0, EYOC_ERR_INVALID, "spconv: C_in %d must be a multiple of 32", a.cin);
  EYOC_REQUIRE(a.cout == 32 || a.cout == 64 || a.cout == 128 || a.cout == 256, EYOC_ERR_INVALID,
               "spconv: C_out %d not in {32,64,128,256}", a.cout);
  EYOC_REQUIRE(a.K >= 1 && a.K <= KMAX, EYOC_ERR_INVALID, "spconv: K %d not in [1,%d]", a.K, KMAX);
  EYOC_REQUIRE(a.nbr || a.K == 1, EYOC_ERR_INVALID, "spconv: identity map needs K == 1");
  EYOC_REQUIRE(a.ld_in % 4 == 0 && a.ld_out % 4 == 0 && (!a.res || a.ld_res % 4 == 0), EYOC_ERR_INVALID,
               "spconv: leading dimensions must be multiples of 4 floats");
  EYOC_REQUIRE((((uintptr_t)a.in | (uintptr_t)a.out | (uintptr_t)a.w | (uintptr_t)a.res | (uintptr_t)a.bias) & 15) == 0,
               EYOC_ERR_INVALID, "spconv: pointers must be 16-byte aligned");
  EYOC_REQUIRE(!a.l2norm || a.cout <= 128, EYOC_ERR_INVALID, "spconv: l2norm needs C_out <= 128");
  // the compacted pair records carry the input row in 24 bits (spconv.hip / spconv_wave.hip: row << 8 | local row)
  EYOC_REQUIRE((a.nbr ? a.n_in : a.n_out) < (1 << 24), EYOC_ERR_INVALID,
               "spconv: %d input rows exceed the 2^24 rows one launch can address", a.nbr ? a.n_in : a.n_out);
  if (a.n_out == 0) return EYOC_OK;
  // Two decompositions: the wave-private kernel (spconv_wave.hip) wins once its 64-row tiles give every SIMD
  // a few waves' worth of work (measured cross-over ~4000 tiles on MI355X); below that the workgroup-tiled
  // kernel here balances better.  eyoc_spconv_select_kernel forces one of them.
  const eyoc_ctx::Knobs& kn = knobs_of(a.ctx);
  const int force = kn.spconv_kernel;
  const long long wave_tiles = (long long)cdiv(a.n_out, 64) * (a.cout >= 64 ? a.cout / 64 : 1);
  // row normalisation needs the whole output row in one tile: the wave-private kernel's tiles are at most 64
  // channels wide, so a normalised 128-channel layer always takes the workgroup-tiled kernel (CT = 128)
  const bool wave_ok = !(a.l2norm && a.cout > 64);
  if (a.math != 0 || a.out_split) {   // SPLIT16 rows exist in the wave-private kernel only
    EYOC_REQUIRE(wave_ok, EYOC_ERR_INVALID, "spconv: a normalised %d-channel layer has no split16 kernel", a.cout);
    EYOC_REQUIRE(a.math == 0 || (a.cin % 32 == 0 && a.ld_in % 32 == 0), EYOC_ERR_INVALID, "spconv: split16 rows come in blocks of 32 channels");
    EYOC_REQUIRE(!a.out_split || a.ld_out % 32 == 0, EYOC_ERR_INVALID, "spconv: split16 output rows come in blocks of 32 channels");
    // split16 layers: the row-stationary kernel (spconv_rs.hip) unless eyoc_spconv_select_split16_kernel(0) asks for the wave-private one
    const int use_rs = kn.split16_kernel;
    // measured per layer of the 64-pair bench: the row-stationary kernel wins on the stride-1, transposed and 1x1 layers
    // with C_in >= 64 (-5 .. -15 %); the 32-channel layers and the strided convolutions (few, scattered pairs per
    // output row: 2.9x more zero MFMAs buy nothing there) stay on the wave-private kernel
    // stride-1 table with local rulebooks: staged kernel (its workgroups cover 64 - or 32 - output channels each and
    // 1, 2, 4 or 8 of them share a tile; other widths stay on the gathering kernels)
    switch (spconv_record_path(a)) {
      case 1: { SpconvArgs b = a; b.perm = nullptr; return launch_spconv_st(b, a.local, st); }
      case 2: { SpconvArgs b = a; b.perm = nullptr; return launch_spconv_upc(b, a.local_upc, st); }     // transposed table, class-major tiles
      case 3: { SpconvArgs b = a; b.perm = nullptr; return launch_spconv_up(b, a.local_up, st); }       // transposed table with tile rulebooks
      case 4: { SpconvArgs b = a; b.perm = nullptr; return launch_spconv_st128(b, a.local_down, st); }  // strided table, 128-row tiles
      case 5: { SpconvArgs b = a; b.perm = nullptr; return launch_spconv_st128(b, a.local128, st); }    // stride-1 table, 128-row x 128-channel workgroups
      default: break;
    }
    const bool rs_layer = a.cin >= 64 && !(a.n_in > a.n_out);
    if (a.math == 1 && spconv_rs_fits(a) && (use_rs == 2 || (use_rs == 1 && rs_layer))) return launch_spconv_rs(a, st);
    return launch_spconv_wave(a, st);
  }
  if (wave_ok && (force > 0 || (force < 0 && wave_tiles >= 4096))) return launch_spconv_wave(a, st);
  const bool wide = spconv_cc(a.cin, a.cout) == 64;
  // Small inputs leave most of the chip idle while every workgroup walks its 27 offsets x C_in slices one after the other, each item
  // a weight fetch from L2 (a level-2 / level-3 layer of two 11 k-voxel clouds: 88-176 workgroups of 32 rows, 54-108 items of ~3 us:
  // 180 us for 10 MFLOP).  Where the caller allows it the offsets are split over 2-4 workgroups per row tile (scratch of the ctx).
  SpconvArgs b = a;
  if (a.allow_offset_split && a.ctx && a.K == 27 && !a.l2norm && a.math == 0 && !a.out_split) {
    const long long wgs = (long long)cdiv(a.n_out, 32) * (a.cout / spconv_ct(a.cout));
    // (up to 2048 workgroups: 20 KB of LDS and 2-4 waves each - eight fit a CU, and what bounds a workgroup is its chain of items, not the CU)
    const int z = wgs * 4 <= 2048 ? 4 : wgs * 3 <= 2048 ? 3 : wgs * 2 <= 2048 ? 2 : 1;
    if (z > 1) {
      const size_t bytes = (size_t)z * a.n_out * a.cout * sizeof(float);
      if (int rc = a.ctx->ensure_scratch(bytes, st)) return rc;
      b.offset_split = z;
      b.offset_part = (float*)a.ctx->scratch;
    }
  }
  switch (spconv_ct(a.cout)) {
    case 32: wide ? launch_ct<32, 64>(b, st) : launch_ct<32, 32>(b, st); break;
    case 64: wide ? launch_ct<64, 64>(b, st) : launch_ct<64, 32>(b, st); break;
    default: wide ? launch_ct<128, 64>(b, st) : launch_ct<128, 32>(b, st); break;
  }
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

bool conv1_walks_octree(const Conv1Args& a) {
  const size_t tree_lds = (size_t)a.ks * a.ks * a.ks * a.cin * (a.cout + 4) * sizeof(float);
  return a.parent && a.children && a.s1c && (a.ks == 3 || a.ks == 5) && tree_lds <= 64 * 1024;
}

// out[i, :] = in[perm[i], :] for narrow rows (the network input, C_in = 1 in production)
__global__ void k_permute_rows(const float* __restrict__ in, const int32_t* __restrict__ perm, int n, int c, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n * c) return;
  const int r = (int)(i / c), j = (int)(i - (long long)r * c);
  out[i] = in[(size_t)perm[r] * c + j];
}
int launch_permute_rows(const float* in, const int32_t* perm, int n, int c, float* out, hipStream_t st) {
  if (n <= 0) return EYOC_OK;
  hipLaunchKernelGGL(k_permute_rows, dim3((unsigned)cdiv((long long)n * c, 256)), dim3(256), 0, st, in, perm, n, c, out);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

// ---- first convolution as a K = 27 sparse convolution over level-1 BLOCK FEATURE VECTORS (round 5; Z-ordered maps, C_in = 1,
// 32 output channels, split16 downstream).  conv1_mfma_kernel above probes, for EVERY fine row, the 27 coarse blocks around its
// parent and their 8 children each (~60 dependent loads per lane and 16-row tile) although the ~2.4 children of a parent share
// all of them and neighbouring parents most.  Round 3's conv1_st_kernel staged the child features of a 256-parent level-1 tile's
// neighbourhood in LDS once (the tile record the staged stride-1 kernel of level 1 uses) but still assembled, per fine row, a
// [128]-entry line of window positions in LDS (zero the tile, then one conditional 2-byte write per child in the window) before
// it could multiply: 0.8 ms per 128 clouds for 0.5 GB of compulsory traffic, bound by that assembly and its dependent loads.
// Here nothing is assembled (0.49 ms; the old kernel is gone).  The 8 child features of a level-1 block - one 16-byte
// line of fp16 hi halves, one of lo halves - ARE an MFMA operand piece: for the fine row in column j, k-slot group g of k-step t
// is the block vector of neighbour block 4 t + g of the row's parent, read straight from the staged vectors (the slot comes
// from the level-1 tile record, as in the staged stride-1 kernel).  The weights are what depends on the row: which of the 216
// (block, child) pairs lies where in the row's 5^3 window is a function of the row's PARITY CLASS (its child slot in its
// parent), so the tile's fine rows are grouped by class - the class is the child slot, the rows of class q are `children[p][q]`
// of the tile's 256 parents: eight ballot compactions, no sort - and a wave multiplies all 16-row chunks of one class against
// that class's [216 x 32] weight matrix, gathered once per (tile, class) from an LDS copy of the [125 x 32] kernel into 112
// registers.  42 MFMAs per 16 rows instead of 24 (the matrix pipe idles in this layer either way), no operand tile, no
// per-row dependent global loads: the tile's rulebook entries are staged in LDS with the block vectors.
// Same products as conv1_mfma_kernel, summed in another order (blocks, not window positions): equal to fp32 rounding
// (tests/test_gpu_split16.py).  Timing-only ablations (make ../lib/libeyoc_hip_c1abl<N>.so): the class-weight gathers 0.07 ms, the
// LDS copy of the kernel 0.04 ms of the 0.49; the rest is the tile's dependent stage rounds and its stores.
#ifndef CONV1_BF_ABL
#define CONV1_BF_ABL 0
#endif
__global__ __launch_bounds__(256, 2) void conv1_bf_kernel(Conv1Args a) {
  constexpr int COUT = 32, NT = COUT / 16, NS7 = 7;
  constexpr int NSLOT = ST_NPASS * ST_UMAX + 1, ZSLOT = NSLOT - 1;
  __shared__ __attribute__((aligned(16))) _Float16 cfh[NSLOT][8], cfl[NSLOT][8];     // block vectors of the staged level-1 rows
  __shared__ __attribute__((aligned(16))) _Float16 w5h[128][COUT], w5l[128][COUT];   // the kernel, scaled and split; row 127 = zeros
  __shared__ __attribute__((aligned(16))) unsigned short ent[27 * 64 * 4];           // pass-0 rulebook entries of the tile
  __shared__ unsigned short lrow[8][ST_TILE];                                        // fine rows of class q (minus the tile's first fine row)
  __shared__ unsigned char lpar[8][ST_TILE];                                         // ... and their parents (local index)
  __shared__ signed char ktab[8][216];
  __shared__ unsigned char inv[ST_TILE];
  __shared__ int ccnt[8][4], ctot[8];
  __shared__ int f0s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int K = a.ks * a.ks * a.ks, r = a.ks / 2;
  const int tile = blockIdx.x;
  const unsigned char* lr = a.local1 + (size_t)tile * ST_LR_BYTES;
  inv[threadIdx.x] = lr[ST_INV_OFF + threadIdx.x];
  for (int e = threadIdx.x; e < 8 * 216; e += 256) {
    const int cls = e / 216, kc = (e % 216) / 8, cs = e % 8;
    const int dx = 2 * (kc % 3 - 1) + (cs & 1) - (cls & 1), dy = 2 * ((kc / 3) % 3 - 1) + ((cs >> 1) & 1) - ((cls >> 1) & 1),
              dz = 2 * (kc / 9 - 1) + (cs >> 2) - (cls >> 2);
    const bool in = dx >= -r && dx <= r && dy >= -r && dy <= r && dz >= -r && dz <= r;
    ktab[cls][kc * 8 + cs] = (signed char)(in ? (dx + r) + a.ks * (dy + r) + a.ks * a.ks * (dz + r) : 127);
  }
  {   // the kernel: W * 2^sh split into fp16 halves (2^sh lifts the layer's largest weight into [256, 512))
    const float wsc = a.wscale ? a.wscale[0] : 256.0f;
    for (int e = threadIdx.x; e < 128 * COUT; e += 256) {
      const int k = e / COUT;
#if CONV1_BF_ABL == 1
      const float w = 1.0f * wsc;
#else
      const float w = k < K ? a.w[e] * wsc : 0.0f;
#endif
      const _Float16 h = (_Float16)w;
      (&w5h[0][0])[e] = h;
      (&w5l[0][0])[e] = (_Float16)(w - (float)h);
    }
  }
  {   // the tile's pass-0 entries: 13.5 KB, 16 bytes per load
    const uint4* src = reinterpret_cast<const uint4*>(lr + ST_LOC_OFF);
    uint4* dst = reinterpret_cast<uint4*>(ent);
    for (int e = threadIdx.x; e < 27 * 64 * 4 * 2 / 16; e += 256) dst[e] = src[e];
  }
  // ---- stage: block vectors of the tile's distinct level-1 rows (three rounds of independent loads: row numbers, child links, child features)
  const int n_u = *reinterpret_cast<const int*>(lr);
  const int* __restrict__ U = reinterpret_cast<const int*>(lr + 16);
  const float* __restrict__ fin = a.in;
  {
    constexpr int NS = (NSLOT + 255) / 256;
    int u[NS];
    int4 c0[NS], c1[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int s = (int)threadIdx.x + 256 * i;
      u[i] = s < n_u ? U[s] : -1;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      c0[i] = c1[i] = make_int4(-1, -1, -1, -1);
      if (u[i] >= 0) {
        c0[i] = reinterpret_cast<const int4*>(a.children)[2 * (size_t)u[i]];
        c1[i] = reinterpret_cast<const int4*>(a.children)[2 * (size_t)u[i] + 1];
      }
    }
    float f[NS][8];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int v[8] = {c0[i].x, c0[i].y, c0[i].z, c0[i].w, c1[i].x, c1[i].y, c1[i].z, c1[i].w};
#pragma unroll
      for (int q = 0; q < 8; ++q) f[i][q] = (256 * i < n_u && v[q] >= 0) ? fin[v[q]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int s = (int)threadIdx.x + 256 * i;
      if (s < n_u || s == ZSLOT) {
        half8_t h8, l8;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float x = s < n_u ? f[i][q] : 0.f;
          h8[q] = (_Float16)x;
          l8[q] = (_Float16)(x - (float)h8[q]);
        }
        *reinterpret_cast<half8_t*>(&cfh[s][0]) = h8;
        *reinterpret_cast<half8_t*>(&cfl[s][0]) = l8;
      }
    }
  }
  // ---- the tile's fine rows by class: thread p holds parent p's child links
  {
    const int prow = tile * ST_TILE + (int)threadIdx.x;
    int v[8];
    if (prow < a.nc) {
      const int4 c0 = reinterpret_cast<const int4*>(a.children)[2 * (size_t)prow], c1 = reinterpret_cast<const int4*>(a.children)[2 * (size_t)prow + 1];
      v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = -1;
    }
    if (threadIdx.x == 0) {                              // the tile's first fine row: the first child of its first parent (Z-order)
      int m = a.n;
#pragma unroll
      for (int q = 0; q < 8; ++q) if (v[q] >= 0 && v[q] < m) m = v[q];
      f0s = m;
    }
    unsigned long long bm[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      bm[q] = __ballot(v[q] >= 0);
      if (lane == 0) ccnt[q][wave] = __popcll(bm[q]);
    }
    __syncthreads();
    const int f0 = f0s;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      int base = 0;
      for (int w = 0; w < wave; ++w) base += ccnt[q][w];
      if (v[q] >= 0) {
        const int pos = base + __popcll(bm[q] & ((1ull << lane) - 1ull));
        lrow[q][pos] = (unsigned short)(v[q] - f0);
        lpar[q][pos] = (unsigned char)threadIdx.x;
      }
      if (threadIdx.x == 0) ctot[q] = ccnt[q][0] + ccnt[q][1] + ccnt[q][2] + ccnt[q][3];
    }
  }
  __syncthreads();
  const int f0 = f0s;
  const int n_pass = n_u > ST_UMAX ? 2 : 1;
  const unsigned short* __restrict__ loc = reinterpret_cast<const unsigned short*>(lr + ST_LOC_OFF);
  const float iwsc = a.wscale ? a.wscale[1] : 1.0f / 256.0f;
  float4 bias4[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bias4[t] = *reinterpret_cast<const float4*>(a.bias + 16 * t + 4 * g);
  float mx = 0.f;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  for (int qi = 0; qi < 2; ++qi) {
    const int q = wave + 4 * qi;                         // this wave's class
    const int nq = ctot[q];
    if (nq == 0) continue;                               // wave-uniform
    // the class's weights: lane (g, j) of k-step t holds W[window position of (block 4 t + g, child i)][channel 16 tt + j], i = 0..7
    half8_t wh[NS7][NT], wl[NS7][NT];
#pragma unroll
    for (int t = 0; t < NS7; ++t) {
      const int kc = 4 * t + g;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#if CONV1_BF_ABL == 2
        const int kp = (kc + i) & 127;
#elif CONV1_BF_ABL == 3
        const int kp = i;
#else
        const int kp = kc < 27 ? (int)ktab[q][kc * 8 + i] : 127;
#endif
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          wh[t][tt][i] = w5h[kp][16 * tt + j];
          wl[t][tt][i] = w5l[kp][16 * tt + j];
        }
      }
    }
#if CONV1_BF_ABL == 4
    for (int c0 = 0; c0 < min(nq, 16); c0 += 16) {
#else
    for (int c0 = 0; c0 < nq; c0 += 16) {
#endif
      const int idx = c0 + j;
      const bool ok = idx < nq;
      const int pl = lpar[q][ok ? idx : 0];
      const int o = f0 + (int)lrow[q][ok ? idx : 0];
      const int sl = inv[pl];
      const int ew = (sl >> 6) * 16 + (sl & 15), ec = (sl >> 4) & 3;
      f32x4 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NS7; ++t) {
        const int kc = 4 * t + g;
        int slot = ZSLOT;
        if (kc < 27) {
          const int e0 = ent[(kc * 64 + ew) * 4 + ec] >> 6;                            // entry = 64 l + swizzle
          if (e0 != ST_UMAX) slot = e0;
          else if (n_pass > 1) {                                                        // rare: the neighbour sits in the second pass
            const int e1 = loc[((size_t)(27 + kc) * 64 + ew) * 4 + ec] >> 6;
            if (e1 != ST_UMAX) slot = ST_UMAX + e1;
          }
        }
        const half8_t vh = *reinterpret_cast<const half8_t*>(&cfh[slot][0]);
        const half8_t vl = *reinterpret_cast<const half8_t*>(&cfl[slot][0]);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t][tt], vh, acc[tt], 0, 0, 0);
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t][tt], vl, acc[tt], 0, 0, 0);
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t][tt], vh, acc[tt], 0, 0, 0);
        }
      }
      if (ok) {
        float* dst = a.out + (size_t)o * a.ld_out;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int ch0 = 16 * t + 4 * g;
          const float4 v = make_float4(acc[t][0] * iwsc + bias4[t].x, acc[t][1] * iwsc + bias4[t].y,
                                       acc[t][2] * iwsc + bias4[t].z, acc[t][3] * iwsc + bias4[t].w);
          split16_track(mx, v);
          if (a.out_split) split16_store4(dst, ch0, v);
          else *reinterpret_cast<float4*>(dst + ch0) = v;
        }
      }
    }
  }
  if (a.out_split) split16_report(a.range, mx);
}

int launch_conv1(const Conv1Args& a, hipStream_t st) {
  EYOC_REQUIRE(a.ks == 1 || a.ks == 3 || a.ks == 5 || a.ks == 7, EYOC_ERR_INVALID, "conv1: kernel size %d", a.ks);
  EYOC_REQUIRE(a.cin >= 1 && a.cin * a.cout <= 8192, EYOC_ERR_INVALID, "conv1: C_in %d x C_out %d too large", a.cin, a.cout);
  EYOC_REQUIRE(a.ld_out % 4 == 0, EYOC_ERR_INVALID, "conv1: ld_out must be a multiple of 4");
  if (a.n == 0) return EYOC_OK;
  dim3 grid(cdiv(a.n, 256));
  const size_t tree_lds = (size_t)a.ks * a.ks * a.ks * a.cin * (a.cout + 4) * sizeof(float);
  if (conv1_walks_octree(a)) {
    // C_in = 1, 32 output channels, split16 activations downstream (the large-batch path): the MFMA formulation.  Its
    // products carry 22-bit significands like every split16 layer; fp32 consumers keep the exact-fp32 walker
    if (knobs_of(a.ctx).conv1_kernel != 2 && a.cin == 1 && a.cout == 32 && a.out_split && a.ks * a.ks * a.ks < 128 && !a.in_perm) {
      if (a.local1 && knobs_of(a.ctx).conv1_kernel == 1) {                               // Z-ordered maps: block vectors staged per 256-parent tile
        hipLaunchKernelGGL(conv1_bf_kernel, dim3(cdiv(a.nc, ST_TILE)), dim3(256), 0, st, a);
        EYOC_CHECK_HIP(hipGetLastError());
        return EYOC_OK;
      }
      const int wgs = cdiv(a.n, 64);
      hipLaunchKernelGGL(conv1_mfma_kernel, dim3(wgs < 2560 ? wgs : 2560), dim3(256), 0, st, a);   // 10 workgroups per CU, persistent
      EYOC_CHECK_HIP(hipGetLastError());
      return EYOC_OK;
    }
    switch (a.cout) {
      case 32: hipLaunchKernelGGL(conv1_tree_kernel<32>, grid, dim3(256), tree_lds, st, a); break;
      case 64: hipLaunchKernelGGL(conv1_tree_kernel<64>, grid, dim3(256), tree_lds, st, a); break;
      case 128: hipLaunchKernelGGL(conv1_tree_kernel<128>, grid, dim3(256), tree_lds, st, a); break;
      default:
        set_error("conv1: C_out %d not in {32,64,128}", a.cout);
        return EYOC_ERR_INVALID;
    }
    EYOC_CHECK_HIP(hipGetLastError());
    return EYOC_OK;
  }
  switch (a.cout) {
    case 32: hipLaunchKernelGGL(conv1_kernel<32>, grid, dim3(256), 0, st, a); break;
    case 64: hipLaunchKernelGGL(conv1_kernel<64>, grid, dim3(256), 0, st, a); break;
    case 128: hipLaunchKernelGGL(conv1_kernel<128>, grid, dim3(256), 0, st, a); break;
    default:
      set_error("conv1: C_out %d not in {32,64,128}", a.cout);
      return EYOC_ERR_INVALID;
  }
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

}  // namespace eyoc

#ifdef EYOC_TRACE
extern "C" int eyoc_debug_trace(unsigned long long* out_host, size_t count) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_trace), count * sizeof(unsigned long long));
}
#endif

extern "C" {

size_t eyoc_spconv_packed_floats(int K, int cin, int cout) { return (size_t)K * cin * cout; }

int eyoc_spconv_select_kernel(eyoc_ctx* ctx, int mode) {
  if (!ctx) return -2;
  const int prev = ctx->knobs.spconv_kernel;
  ctx->knobs.spconv_kernel = mode;
  return prev;
}

// packed[k][slice][cc][nt][jq][lane][e] = W[k][cc*CC + (jq*4 + (lane>>4))*4 + e][slice*CT + nt*16 + (lane&15)] * scale[col]
// with CT = spconv_ct(cout), CC = spconv_cc(cin, cout): for one (k, slice, cc) the [nt][jq][lane] float4s are
// exactly the B fragments of v_mfma_f32_16x16x4_f32 (B[k = lane>>4][n = lane&15]) in the order a wave consumes them.
int eyoc_spconv_pack_weights(const float* w, const float* scale, int K, int cin, int cout, float* packed) {
  EYOC_REQUIRE(w && packed, EYOC_ERR_INVALID, "pack_weights: NULL argument");
  EYOC_REQUIRE(cin > 0 && cin % 32 == 0 && (cout == 32 || cout == 64 || cout == 128 || cout == 256), EYOC_ERR_INVALID,
               "pack_weights: unsupported shape C_in %d C_out %d", cin, cout);
  const int CT = spconv_ct(cout), CC = spconv_cc(cin, cout);
  const int n_slices = cout / CT, ncc = cin / CC, NT = CT / 16, JQ = CC / 16;
  size_t q = 0;
  for (int k = 0; k < K; ++k)
    for (int s = 0; s < n_slices; ++s)
      for (int cc = 0; cc < ncc; ++cc)
        for (int nt = 0; nt < NT; ++nt)
          for (int jq = 0; jq < JQ; ++jq)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 4; ++e) {
                const int ci = cc * CC + (jq * 4 + (lane >> 4)) * 4 + e;
                const int co = s * CT + nt * 16 + (lane & 15);
                const float v = w[((size_t)k * cin + ci) * cout + co];
                packed[q++] = scale ? v * scale[co] : v;
              }
  return EYOC_OK;
}

// SPLIT16 packing: the same [k][slice][cc][nt][jq][lane] grid of 16-byte fragments as the fp32 layout, but fragment
// jq = 2 q + p holds 8 fp16 values: the hi (p = 0) or lo (p = 1) halves of
//   W[k][cc*CC + q*32 + (lane>>4)*8 + e][slice*CT + nt*16 + (lane&15)] * scale[col] * 2^sh,   e = 0..7
// i.e. the A operand of v_mfma_f32_16x16x32_f16 (A[i = lane&15][k-slot (lane>>4)*8 + e]).  2^sh lifts the largest
// |weight| of the layer into [256, 512) so that the lo halves stay clear of the fp16 subnormals; the kernel multiplies
// its sums by *out_scale = 2^-sh.
int eyoc_spconv_pack_weights_split16(const float* w, const float* scale, int K, int cin, int cout, float* packed,
                                     float* out_scale) {
  EYOC_REQUIRE(w && packed && out_scale, EYOC_ERR_INVALID, "pack_weights_split16: NULL argument");
  EYOC_REQUIRE(cin > 0 && cin % 32 == 0 && (cout == 32 || cout == 64 || cout == 128 || cout == 256), EYOC_ERR_INVALID,
               "pack_weights_split16: unsupported shape C_in %d C_out %d", cin, cout);
  const int CT = spconv_ct(cout), CC = spconv_cc(cin, cout);
  const int n_slices = cout / CT, ncc = cin / CC, NT = CT / 16, JQ = CC / 16;
  float wmax = 0.f;
  for (size_t i = 0; i < (size_t)K * cin * cout; ++i) {
    const float v = fabsf(scale ? w[i] * scale[i % cout] : w[i]);
    if (v > wmax) wmax = v;
  }
  int sh = 0;
  if (wmax > 0.f && std::isfinite(wmax)) {
    int e;
    frexpf(wmax, &e);        // wmax = m * 2^e, m in [0.5, 1)
    sh = 9 - e;              // wmax * 2^sh in [256, 512)
    if (sh > 24) sh = 24;
    if (sh < -6) sh = -6;
  }
  const float up = ldexpf(1.0f, sh);
  *out_scale = ldexpf(1.0f, -sh);
  _Float16* out = reinterpret_cast<_Float16*>(packed);
  size_t q8 = 0;
  for (int k = 0; k < K; ++k)
    for (int s = 0; s < n_slices; ++s)
      for (int cc = 0; cc < ncc; ++cc)
        for (int nt = 0; nt < NT; ++nt)
          for (int jq = 0; jq < JQ; ++jq)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 8; ++e) {
                const int ci = cc * CC + (jq >> 1) * 32 + (lane >> 4) * 8 + e;
                const int co = s * CT + nt * 16 + (lane & 15);
                float v = w[((size_t)k * cin + ci) * cout + co];
                if (scale) v *= scale[co];
                v *= up;
                const _Float16 h = (_Float16)v;
                out[q8++] = (jq & 1) ? (_Float16)(v - (float)h) : h;
              }
  return EYOC_OK;
}

size_t eyoc_spconv_local_rulebook_bytes(int n_out) { return n_out < 0 ? 0 : local_rulebook_bytes(n_out) + 256; }

int eyoc_spconv_build_local_rulebook(eyoc_ctx* ctx, const int32_t* nbr_dev, int K, int n_out, void* out_dev, int32_t* overflow_dev,
                                     void* stream) {
  EYOC_REQUIRE(ctx && nbr_dev && out_dev && overflow_dev && K >= 1 && K <= 27 && n_out >= 0, EYOC_ERR_INVALID,
               "eyoc_spconv_build_local_rulebook: bad argument");
  EYOC_CHECK_HIP(hipMemsetAsync(overflow_dev, 0, 4, (hipStream_t)stream));
  return build_local_rulebook(nbr_dev, K, n_out, (unsigned char*)out_dev, overflow_dev, (hipStream_t)stream, ctx->knobs.st_group);
}

int eyoc_spconv_staged(eyoc_ctx* ctx, const int32_t* nbr_dev, const void* local_dev, int n_out, int n_in, const float* in_dev, int ld_in,
                       int cin, const float* wpacked_dev, int cout, const float* bias_dev, const float* res_dev, int ld_res, int relu,
                       float* out_dev, int ld_out, int out_split, const float* out_scale_dev, void* stream) {
  EYOC_REQUIRE(ctx && nbr_dev && local_dev, EYOC_ERR_INVALID, "eyoc_spconv_staged: NULL argument");
  SpconvArgs a;
  a.nbr = nbr_dev; a.K = 27; a.n_out = n_out; a.n_in = n_in; a.in = in_dev; a.ld_in = ld_in; a.cin = cin; a.w = wpacked_dev;
  a.cout = cout; a.bias = bias_dev; a.res = res_dev; a.ld_res = ld_res; a.relu = relu; a.l2norm = 0;
  a.out = out_dev; a.ld_out = ld_out; a.math = 1; a.out_split = out_split; a.out_scale = out_scale_dev;
  a.local = (const unsigned char*)local_dev; a.ctx = ctx;
  return launch_spconv(a, (hipStream_t)stream);
}

// the setters below return the previous value (-1: NULL ctx) and leave the switch alone for an out-of-range argument, so
// f(ctx, -1) reads it
int eyoc_spconv_st_group_rows(eyoc_ctx* ctx, int on) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.st_group;
  if (on == 0 || on == 1) ctx->knobs.st_group = on;
  return prev;
}
int eyoc_spconv_st_ksplit(eyoc_ctx* ctx, int on) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.st_ksplit;
  if (on == 0 || on == 1) ctx->knobs.st_ksplit = on;
  if (on == 2 || on == 3) ctx->knobs.st_cg_local = on - 2;             // (diagnostics) the channel groups of a tile on one XCD: off / on
  return prev;
}

int eyoc_spconv_select_up_kernel(eyoc_ctx* ctx, int on) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.up_kernel;
  if (on >= 0 && on <= 2) ctx->knobs.up_kernel = on;
  return prev;
}

int eyoc_spconv_upc_min_rows(eyoc_ctx* ctx, int rows) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.upc_min_rows;
  if (rows >= 0) ctx->knobs.upc_min_rows = rows;
  return prev;
}

int eyoc_spconv_upc_tile_rows(int odd_axes, int rows) { return eyoc::upc_set_tile_rows(odd_axes, rows); }

size_t eyoc_spconv_upc_bytes(int n_out) { return n_out < 0 ? 0 : eyoc::upc_kept_bytes(n_out) + eyoc::upc_scratch_bytes(n_out) + 512; }

int eyoc_spconv_upc_build(eyoc_ctx* ctx, const int32_t* nbr_dev, int n_out, void* ws_dev, int32_t* info_host, void* stream) {
  EYOC_REQUIRE(ctx && nbr_dev && ws_dev && n_out >= 0 && ((uintptr_t)ws_dev & 255) == 0, EYOC_ERR_INVALID, "eyoc_spconv_upc_build: bad argument");
  unsigned char* ws = (unsigned char*)ws_dev;
  if (int rc = eyoc::build_upc(nbr_dev, nullptr, 1, n_out, ws, ws + ((eyoc::upc_kept_bytes(n_out) + 255) & ~(size_t)255), (hipStream_t)stream)) return rc;
  if (info_host) {   // {n_tiles, tile_start[9], count[8], overflow}: one synchronising copy (tests, diagnostics)
    EYOC_CHECK_HIP(hipMemcpyAsync(info_host, ws, 19 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    EYOC_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  }
  return EYOC_OK;
}

int eyoc_spconv_upc(eyoc_ctx* ctx, const int32_t* nbr_dev, const void* ws_dev, int n_out, int n_in, const float* in_dev, int ld_in, int cin,
                    const float* wpacked_dev, int cout, const float* bias_dev, int relu, float* out_dev, int ld_out, int out_split,
                    const float* out_scale_dev, void* stream) {
  EYOC_REQUIRE(ctx && nbr_dev && ws_dev, EYOC_ERR_INVALID, "eyoc_spconv_upc: NULL argument");
  SpconvArgs a;
  a.nbr = nbr_dev; a.K = 27; a.n_out = n_out; a.n_in = n_in; a.in = in_dev; a.ld_in = ld_in; a.cin = cin; a.w = wpacked_dev;
  a.cout = cout; a.bias = bias_dev; a.res = nullptr; a.ld_res = 0; a.relu = relu; a.l2norm = 0;
  a.out = out_dev; a.ld_out = ld_out; a.math = 1; a.out_split = out_split; a.out_scale = out_scale_dev;
  a.local_upc = (const unsigned char*)ws_dev; a.ctx = ctx;
  return launch_spconv(a, (hipStream_t)stream);
}

int eyoc_spconv_select_conv1_kernel(eyoc_ctx* ctx, int on) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.conv1_kernel;
  if (on >= 0 && on <= 2) ctx->knobs.conv1_kernel = on;
  return prev;
}

int eyoc_spconv_select_st_kernel(eyoc_ctx* ctx, int variant) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.st_variant;
  if (variant >= 0 && variant < eyoc::st_variants()) ctx->knobs.st_variant = variant;
  return prev;
}

int eyoc_spconv_st_split_below(eyoc_ctx* ctx, int workgroups) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.st_split_below;
  if (workgroups >= 0) ctx->knobs.st_split_below = workgroups;
  return prev;
}

int eyoc_spconv_select_split16_kernel(eyoc_ctx* ctx, int mode) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.split16_kernel;
  if (mode >= 0 && mode <= 2) ctx->knobs.split16_kernel = mode;
  return prev;
}

int eyoc_spconv_ex(eyoc_ctx* ctx, const int32_t* nbr_dev, int K, int n_out, int n_in, const float* in_dev, int ld_in, int cin,
                   const float* wpacked_dev, int cout, const float* bias_dev, const float* res_dev, int ld_res, int relu,
                   float* out_dev, int ld_out, int math, int out_split, const float* out_scale_dev, void* stream) {
  EYOC_REQUIRE(ctx, EYOC_ERR_INVALID, "eyoc_spconv_ex: NULL ctx");
  EYOC_REQUIRE(math == 0 || math == 1, EYOC_ERR_INVALID, "eyoc_spconv_ex: math %d", math);
  SpconvArgs a;
  a.nbr = nbr_dev; a.K = K; a.n_out = n_out; a.n_in = n_in; a.in = in_dev; a.ld_in = ld_in; a.cin = cin; a.w = wpacked_dev;
  a.cout = cout; a.bias = bias_dev; a.res = res_dev; a.ld_res = ld_res; a.relu = relu; a.l2norm = 0;
  a.out = out_dev; a.ld_out = ld_out; a.math = math; a.out_split = out_split; a.out_scale = out_scale_dev; a.ctx = ctx;
  return launch_spconv(a, (hipStream_t)stream);
}

int eyoc_split16_encode(eyoc_ctx* ctx, const float* in_dev, int n, int c, int ld_in, float* out_dev, int ld_out, void* stream) {
  EYOC_REQUIRE(ctx && in_dev && out_dev && n >= 0 && c > 0 && c % 32 == 0 && ld_in % 4 == 0 && ld_out % 32 == 0, EYOC_ERR_INVALID,
               "eyoc_split16_encode: bad argument (c %d must be a multiple of 32)", c);
  if (n) hipLaunchKernelGGL(k_split16_encode, dim3(cdiv((long long)n * (c / 4), 256)), dim3(256), 0, (hipStream_t)stream, in_dev, n, c, ld_in, out_dev, ld_out);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

int eyoc_split16_decode(eyoc_ctx* ctx, const float* in_dev, int n, int c, int ld_in, float* out_dev, int ld_out, void* stream) {
  EYOC_REQUIRE(ctx && in_dev && out_dev && n >= 0 && c > 0 && c % 32 == 0 && ld_in % 32 == 0 && ld_out % 4 == 0, EYOC_ERR_INVALID,
               "eyoc_split16_decode: bad argument (c %d must be a multiple of 32)", c);
  if (n) hipLaunchKernelGGL(k_split16_decode, dim3(cdiv((long long)n * (c / 4), 256)), dim3(256), 0, (hipStream_t)stream, in_dev, n, c, ld_in, out_dev, ld_out);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

int eyoc_spconv(eyoc_ctx* ctx, const int32_t* nbr_dev, int K, int n_out, const float* in_dev, int ld_in, int cin,
                const float* wpacked_dev, int cout, const float* bias_dev, const float* res_dev, int ld_res, int relu,
                float* out_dev, int ld_out, void* stream) {
  EYOC_REQUIRE(ctx, EYOC_ERR_INVALID, "eyoc_spconv: NULL ctx");
  SpconvArgs a;
  a.nbr = nbr_dev; a.K = K; a.n_out = n_out; a.in = in_dev; a.ld_in = ld_in; a.cin = cin; a.w = wpacked_dev;
  a.cout = cout; a.bias = bias_dev; a.res = res_dev; a.ld_res = ld_res; a.relu = relu; a.l2norm = 0;
  a.out = out_dev; a.ld_out = ld_out; a.ctx = ctx;
  return launch_spconv(a, (hipStream_t)stream);
}

// The bare operator for the training path (autograd.sparse_conv: forward and input gradient): no epilogue, and the launcher may split
// the offsets of a small input over several workgroups per row tile - the summation order, and with it the last bits of the result,
// then depend on the problem size (deterministic for a given input; eyoc_spconv keeps ONE order for every size).
int eyoc_spconv_sum(eyoc_ctx* ctx, const int32_t* nbr_dev, int K, int n_out, const float* in_dev, int ld_in, int cin,
                    const float* wpacked_dev, int cout, float* out_dev, int ld_out, void* stream) {
  EYOC_REQUIRE(ctx, EYOC_ERR_INVALID, "eyoc_spconv_sum: NULL ctx");
  SpconvArgs a;
  a.nbr = nbr_dev; a.K = K; a.n_out = n_out; a.in = in_dev; a.ld_in = ld_in; a.cin = cin; a.w = wpacked_dev;
  a.cout = cout; a.bias = nullptr; a.res = nullptr; a.ld_res = 0; a.relu = 0; a.l2norm = 0;
  a.out = out_dev; a.ld_out = ld_out; a.ctx = ctx; a.allow_offset_split = 1;
  return launch_spconv(a, (hipStream_t)stream);
}

}  // extern "C"
