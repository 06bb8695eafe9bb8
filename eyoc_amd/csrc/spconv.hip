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
#include "spconv.h"

using namespace eyoc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KMAX = 27;

template <int CT, int BM, int NW, int CC>
struct Cfg {
  static constexpr int NT = CT / 16;               // 16-column tiles per block
  static constexpr int JQ = CC / 16;               // ds_read_b128 B-fragment groups per column tile
  static constexpr int APL = CC / 4;               // A floats per lane per 16-pair chunk
  static constexpr int MAXCH = BM / 16 / NW;       // 16-pair chunks per wave per offset (worst case)
  static constexpr int THREADS = NW * 64;
  static constexpr int TILE_FLOATS = CC * CT;      // one packed weight tile
  static constexpr int WPT = TILE_FLOATS / 4 / THREADS;  // float4 per thread per tile
  static constexpr int ACC_LD = CT + 4;            // padded accumulator row (keeps float4 alignment)
  static constexpr int OFF_PAIR_IN = 0;                                   // int [KMAX][BM]
  static constexpr int OFF_PAIR_OUT = OFF_PAIR_IN + KMAX * BM * 4;        // u8  [KMAX][BM]
  static constexpr int OFF_CNT = OFF_PAIR_OUT + KMAX * BM;                // int [KMAX + 1 + KMAX]
  static constexpr int OFF_W = (OFF_CNT + (2 * KMAX + 8) * 4 + 15) / 16 * 16;  // float [2][TILE_FLOATS]
  // two weight buffers (one barrier per iteration) unless that would push the block past half the
  // CU's 160 KB of LDS and cost the second resident workgroup; then one buffer and two barriers
  static constexpr bool DB = OFF_W + 2 * TILE_FLOATS * 4 + BM * ACC_LD * 4 + BM * 4 <= 80 * 1024;
  static constexpr int OFF_ACC = OFF_W + (DB ? 2 : 1) * TILE_FLOATS * 4;  // float [BM][ACC_LD]
  static constexpr int OFF_NORM = OFF_ACC + BM * ACC_LD * 4;              // float [BM]
  static constexpr int LDS_BYTES = OFF_NORM + BM * 4;
  static_assert((WPT == 1 || WPT == 2 || WPT == 3 || WPT == 4 || WPT == 8) && WPT * 4 * THREADS == TILE_FLOATS, "weight tile must divide evenly");
  static_assert(MAXCH >= 1, "BM too small for the wave count");
};

template <int CT, int BM, int NW, int CC>
__global__ __launch_bounds__(NW * 64, 2) void spconv_kernel(SpconvArgs a) {
  using C = Cfg<CT, BM, NW, CC>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
  int* pair_in = reinterpret_cast<int*>(smem + C::OFF_PAIR_IN);
  unsigned char* pair_out = smem + C::OFF_PAIR_OUT;
  int* cnt = reinterpret_cast<int*>(smem + C::OFF_CNT);   // [KMAX] counts, [KMAX] active list, [1] n_act
  int* act = cnt + KMAX;
  int* n_act_p = act + KMAX;
  float* wt = reinterpret_cast<float*>(smem + C::OFF_W);
  float* acc = reinterpret_cast<float*>(smem + C::OFF_ACC);
  float* rnorm = reinterpret_cast<float*>(smem + C::OFF_NORM);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * BM;
  const int slice = blockIdx.y, n_slices = gridDim.y;
  const int ct0 = slice * CT;
  const int ncc = a.cin / CC;
  const int rows_here = min(BM, a.n_out - row0);

  // ---- phase 0: zero the accumulator, compact the rulebook of this row tile
  for (int i = tid; i < BM * C::ACC_LD; i += C::THREADS) acc[i] = 0.0f;
  if (a.nbr) {
    for (int k = wave; k < a.K; k += NW) {
      const int32_t* col = a.nbr + (size_t)k * a.n_out + row0;
      int base = 0;
      for (int r0 = 0; r0 < BM; r0 += 64) {
        const int r = r0 + lane;
        const int idx = (r < rows_here) ? col[r] : -1;
        const bool valid = idx >= 0;
        const unsigned long long m = __ballot(valid);
        if (valid) {
          const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
          pair_in[k * BM + pos] = idx;
          pair_out[k * BM + pos] = (unsigned char)r;
        }
        base += __popcll(m);
      }
      if (lane == 0) cnt[k] = base;
    }
  } else {  // identity map (1x1 convolution)
    for (int r = tid; r < BM; r += C::THREADS) {
      pair_in[r] = row0 + r;
      pair_out[r] = (unsigned char)r;
    }
    if (tid == 0) cnt[0] = rows_here;
  }
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    for (int k = 0; k < a.K; ++k)
      if (cnt[k] > 0) act[n++] = k;
    *n_act_p = n;
  }
  __syncthreads();
  const int n_iter = *n_act_p * ncc;

  const float4* wbase = reinterpret_cast<const float4*>(a.w);
  auto tile_ptr = [&](int it) {
    const int k = act[it / ncc], cc = it % ncc;
    return wbase + ((size_t)(k * n_slices + slice) * ncc + cc) * (C::TILE_FLOATS / 4);
  };

  f32x4 accreg[C::MAXCH][C::NT];
  const int r16 = lane & 15, g = lane >> 4;

  // A fragments of iteration `it`: lane (r16, g) holds channels [cc*CC + g*APL, +APL) of pair chunk*16 + r16
  auto gather_a = [&](int it, float (&dst)[C::MAXCH][C::APL]) {
    const int k = act[it / ncc], cc = it % ncc;
    const int count = cnt[k];
    const int nch = (count + 15) >> 4;
#pragma unroll
    for (int c = 0; c < C::MAXCH; ++c) {
      const int p = (wave + NW * c) * 16 + r16;
      if (wave + NW * c < nch && p < count) {
        const int in_row = pair_in[k * BM + p];
        const float4* src = reinterpret_cast<const float4*>(a.in + (size_t)in_row * a.ld_in + cc * CC + g * C::APL);
#pragma unroll
        for (int q = 0; q < C::APL / 4; ++q) {
          const float4 v = src[q];
          dst[c][4 * q] = v.x; dst[c][4 * q + 1] = v.y; dst[c][4 * q + 2] = v.z; dst[c][4 * q + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < C::APL; ++e) dst[c][e] = 0.0f;
      }
    }
  };

  if (n_iter > 0) {   // block-uniform; keeps every staging array's definition and uses in one region
  // weight-tile staging registers: named scalars, not an array (an indexed array here sometimes
  // stays an alloca -> scratch, which serialises the prefetch behind s_waitcnt + scratch_store)
  float4 w0, w1, w2, w3, w4, w5, w6, w7;
  auto load_w = [&](const float4* src) {
    w0 = src[tid];
    if constexpr (C::WPT > 1) w1 = src[tid + C::THREADS];
    if constexpr (C::WPT > 2) w2 = src[tid + 2 * C::THREADS];
    if constexpr (C::WPT > 3) w3 = src[tid + 3 * C::THREADS];
    if constexpr (C::WPT > 4) {
      w4 = src[tid + 4 * C::THREADS]; w5 = src[tid + 5 * C::THREADS];
      w6 = src[tid + 6 * C::THREADS]; w7 = src[tid + 7 * C::THREADS];
    }
  };
  auto store_w = [&](float4* dst) {
    dst[tid] = w0;
    if constexpr (C::WPT > 1) dst[tid + C::THREADS] = w1;
    if constexpr (C::WPT > 2) dst[tid + 2 * C::THREADS] = w2;
    if constexpr (C::WPT > 3) dst[tid + 3 * C::THREADS] = w3;
    if constexpr (C::WPT > 4) {
      dst[tid + 4 * C::THREADS] = w4; dst[tid + 5 * C::THREADS] = w5;
      dst[tid + 6 * C::THREADS] = w6; dst[tid + 7 * C::THREADS] = w7;
    }
  };
  load_w(tile_ptr(0));
  store_w(reinterpret_cast<float4*>(wt));
  float a_nxt[C::MAXCH][C::APL];
  gather_a(0, a_nxt);
  __syncthreads();

  for (int it = 0; it < n_iter; ++it) {
    const int k = act[it / ncc], cc = it % ncc, buf = C::DB ? (it & 1) : 0;
    float av[C::MAXCH][C::APL];
#pragma unroll
    for (int c = 0; c < C::MAXCH; ++c)
#pragma unroll
      for (int e = 0; e < C::APL; ++e) av[c][e] = a_nxt[c][e];
    {  // software pipeline: next weight tile -> registers, next A fragments -> registers; both land
       // while this iteration's MFMAs run (the last iteration re-reads its own operands: an
       // unconditional load/store pair keeps the staging registers out of scratch)
      const int nx = min(it + 1, n_iter - 1);
      load_w(tile_ptr(nx));
      gather_a(nx, a_nxt);
    }
    const int count = cnt[k];
    const int nch = (count + 15) >> 4;
    if (wave < nch) {  // this wave owns at least one chunk of this offset
      if (cc == 0) {
#pragma unroll
        for (int c = 0; c < C::MAXCH; ++c)
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt) accreg[c][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const float4* wl = reinterpret_cast<const float4*>(wt + buf * C::TILE_FLOATS);
#pragma unroll
      for (int jq = 0; jq < C::JQ; ++jq) {
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) {
          const float4 b = wl[(nt * C::JQ + jq) * 64 + lane];
#pragma unroll
          for (int c = 0; c < C::MAXCH; ++c) {
            if (wave + NW * c < nch) {
              accreg[c][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][jq * 4 + 0], b.x, accreg[c][nt], 0, 0, 0);
              accreg[c][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][jq * 4 + 1], b.y, accreg[c][nt], 0, 0, 0);
              accreg[c][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][jq * 4 + 2], b.z, accreg[c][nt], 0, 0, 0);
              accreg[c][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][jq * 4 + 3], b.w, accreg[c][nt], 0, 0, 0);
            }
          }
        }
      }
      if (cc == ncc - 1) {  // D[row = 4*(lane>>4) + reg][col = lane&15] -> accumulator rows
#pragma unroll
        for (int c = 0; c < C::MAXCH; ++c) {
          const int chunk = wave + NW * c;
          if (chunk < nch) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const int p = chunk * 16 + g * 4 + reg;
              if (p < count) {
                float* dst = acc + (int)pair_out[k * BM + p] * C::ACC_LD + r16;
#pragma unroll
                for (int nt = 0; nt < C::NT; ++nt) dst[nt * 16] += accreg[c][nt][reg];
              }
            }
          }
        }
      }
    }
    if (!C::DB) __syncthreads();   // everyone is done reading the single weight buffer
    store_w(reinterpret_cast<float4*>(wt + (C::DB ? (buf ^ 1) : 0) * C::TILE_FLOATS));
    __syncthreads();
  }
  }  // n_iter > 0

  // ---- epilogue
  constexpr int C4 = CT / 4;
  if (a.l2norm) {
    for (int i = tid; i < BM * C4; i += C::THREADS) {
      const int r = i / C4, c4 = i % C4;
      float4* p = reinterpret_cast<float4*>(acc + r * C::ACC_LD + c4 * 4);
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
        const float v = acc[r * C::ACC_LD + c];
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
    float4 v = *reinterpret_cast<const float4*>(acc + r * C::ACC_LD + c4 * 4);
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
    *reinterpret_cast<float4*>(a.out + o * a.ld_out + ct0 + c4 * 4) = v;
  }
}

template <int CT, int BM, int NW, int CC>
void launch_cfg(const SpconvArgs& a, hipStream_t st) {
  dim3 grid(cdiv(a.n_out, BM), a.cout / CT);
  hipLaunchKernelGGL((spconv_kernel<CT, BM, NW, CC>), grid, dim3(NW * 64), 0, st, a);
}

template <int CT, int CC>
void launch_ct(const SpconvArgs& a, hipStream_t st) {
  // enough workgroups to cover the 256 CUs a few times over, otherwise shrink the row tile
  const long long slices = a.cout / CT;
  if constexpr (CT <= 64) {
    if ((long long)cdiv(a.n_out, 128) * slices >= 1024) {
      launch_cfg<CT, 128, 4, CC>(a, st);
      return;
    }
  }
  if ((long long)cdiv(a.n_out, 64) * slices >= 512) {
    launch_cfg<CT, 64, 4, CC>(a, st);
  } else {
    launch_cfg<CT, 32, 2, CC>(a, st);
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
        const float f = idx >= 0 ? a.in[(size_t)idx * a.cin + ci] : 0.0f;
        const float* wk = ws + (kk * a.cin + ci) * COUT;
#pragma unroll
        for (int i = 0; i < COUT; ++i) acc[i] = fmaf(f, wk[i], acc[i]);
      }
    }
  }
  if (ok) {
    float* dst = a.out + (size_t)o * a.ld_out;
#pragma unroll
    for (int i = 0; i < COUT; i += 4) {
      float4 v;
      v.x = acc[i] + a.bias[i]; v.y = acc[i + 1] + a.bias[i + 1];
      v.z = acc[i + 2] + a.bias[i + 2]; v.w = acc[i + 3] + a.bias[i + 3];
      *reinterpret_cast<float4*>(dst + i) = v;
    }
  }
}

}  // namespace

namespace eyoc {

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
  if (a.n_out == 0) return EYOC_OK;
  const bool wide = spconv_cc(a.cin, a.cout) == 64;
  switch (spconv_ct(a.cout)) {
    case 32: wide ? launch_ct<32, 64>(a, st) : launch_ct<32, 32>(a, st); break;
    case 64: wide ? launch_ct<64, 64>(a, st) : launch_ct<64, 32>(a, st); break;
    default: launch_ct<128, 32>(a, st); break;
  }
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

int launch_conv1(const Conv1Args& a, hipStream_t st) {
  EYOC_REQUIRE(a.ks == 1 || a.ks == 3 || a.ks == 5 || a.ks == 7, EYOC_ERR_INVALID, "conv1: kernel size %d", a.ks);
  EYOC_REQUIRE(a.cin >= 1 && a.cin * a.cout <= 8192, EYOC_ERR_INVALID, "conv1: C_in %d x C_out %d too large", a.cin, a.cout);
  EYOC_REQUIRE(a.ld_out % 4 == 0, EYOC_ERR_INVALID, "conv1: ld_out must be a multiple of 4");
  if (a.n == 0) return EYOC_OK;
  dim3 grid(cdiv(a.n, 256));
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

extern "C" {

size_t eyoc_spconv_packed_floats(int K, int cin, int cout) { return (size_t)K * cin * cout; }

// packed[k][slice][cc][nt][jq][lane][e] = W[k][cc*CC + (lane>>4)*(CC/4) + jq*4 + e][slice*CT + nt*16 + (lane&15)] * scale[col]
// with CT = spconv_ct(cout), CC = spconv_cc(cin, cout): one [cc] entry is exactly the LDS image of a weight tile.
int eyoc_spconv_pack_weights(const float* w, const float* scale, int K, int cin, int cout, float* packed) {
  EYOC_REQUIRE(w && packed, EYOC_ERR_INVALID, "pack_weights: NULL argument");
  EYOC_REQUIRE(cin > 0 && cin % 32 == 0 && (cout == 32 || cout == 64 || cout == 128 || cout == 256), EYOC_ERR_INVALID,
               "pack_weights: unsupported shape C_in %d C_out %d", cin, cout);
  const int CT = spconv_ct(cout), CC = spconv_cc(cin, cout);
  const int n_slices = cout / CT, ncc = cin / CC, NT = CT / 16, JQ = CC / 16, APL = CC / 4;
  size_t q = 0;
  for (int k = 0; k < K; ++k)
    for (int s = 0; s < n_slices; ++s)
      for (int cc = 0; cc < ncc; ++cc)
        for (int nt = 0; nt < NT; ++nt)
          for (int jq = 0; jq < JQ; ++jq)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 4; ++e) {
                const int ci = cc * CC + (lane >> 4) * APL + jq * 4 + e;
                const int co = s * CT + nt * 16 + (lane & 15);
                const float v = w[((size_t)k * cin + ci) * cout + co];
                packed[q++] = scale ? v * scale[co] : v;
              }
  return EYOC_OK;
}

int eyoc_spconv(eyoc_ctx* ctx, const int32_t* nbr_dev, int K, int n_out, const float* in_dev, int ld_in, int cin,
                const float* wpacked_dev, int cout, const float* bias_dev, const float* res_dev, int ld_res, int relu,
                float* out_dev, int ld_out, void* stream) {
  EYOC_REQUIRE(ctx, EYOC_ERR_INVALID, "eyoc_spconv: NULL ctx");
  SpconvArgs a;
  a.nbr = nbr_dev; a.K = K; a.n_out = n_out; a.in = in_dev; a.ld_in = ld_in; a.cin = cin; a.w = wpacked_dev;
  a.cout = cout; a.bias = bias_dev; a.res = res_dev; a.ld_res = ld_res; a.relu = relu; a.l2norm = 0;
  a.out = out_dev; a.ld_out = ld_out;
  return launch_spconv(a, (hipStream_t)stream);
}

}  // extern "C"
