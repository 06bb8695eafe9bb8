// Staged sparse convolution on 128-ROW tiles (gfx950, SPLIT16 arithmetic, stride-1 rulebooks of Z-ordered rows).
//
// Same operator, same arithmetic and the same generated offset loop family as spconv_st.hip - what changes is the shape of
// a workgroup, and with it how many of them share a CU.  The 256-row kernel keeps an 80 KB stage and 256 VGPRs per wave:
// TWO workgroups per CU, two waves per SIMD.  Its ablations and traces say a tile's phases - row list, stage
// DMA, offset loops, residual loads, stores - ADD: a workgroup spends ~30 % of its life outside its loops, per CU 49 % of
// the time only one workgroup is in a loop, and one wave per SIMD reaches ~75 % of the matrix pipe.  More co-resident
// workgroups are the cure, and LDS is what forbids them.  A 128-row tile of Z-ordered rows touches 216-265 distinct input
// rows on LiDAR geometry (378-477 for 256 rows; measured on the bench clouds): a 320-row stage = 40 KB, so FOUR
// workgroups fit a CU; each wave owns 64 rows x 32 output channels (32 accumulator registers) inside 128 VGPRs, so four
// waves fit a SIMD.  Per SIMD four independent tiles are then in flight instead of two: while one waits for its stage or
// stores its outputs, two or three others keep the pipe fed (P(at least two in their loops) ~ 0.9 instead of ~0.4).
// Price: the halo of a smaller tile is relatively larger (stage bytes per row x 1.14) and each weight fragment serves half
// as many rows (twice the weight bytes through the L1: ~1.7 MB per 256 rows of a 64 -> 64 layer).
// MEASURED (round 4): no gain - see g_st_tile below.  The kernel stays selectable (eyoc_spconv_st_tile(128)) and tested; the
// effective clock of both kernels is power-managed (GRBM_GUI_ACTIVE / duration: 1.81-1.95 GHz here, 1.88-2.06 for the 256-row
// kernel, scripts/pmc_clock.sh), i.e. the matrix pipe's 529 k busy clocks per SIMD are 60 % of the REAL clocks of the launch.
//
// Tile records (k_local_rulebook_t, 17792 bytes per 128-row tile): the 256-row layout of spconv_st.hip with 32 entries per offset
// and 319 + 1 stage slots per pass; a tile with more than 319 distinct rows takes a second pass (4 % of the level-0 tiles,
// 20 % at the coarsest level), more than 638 is an overflow (the table then falls back to the gathering kernels).
#include <atomic>

#include "spconv.h"

using namespace eyoc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));

// waves per workgroup: 2 row halves (64 rows) x 2 output-channel halves (32 channels); 32-channel layers take 2 waves
// (the record / kernel code below is templated on the tile's row count: 128 for the stride-1 tables, 64 for the strided ones,
// whose 64-row output tiles touch 190-330 distinct fine rows - 128-row tiles would need two passes half of the time)
constexpr int XROWS = 320;                  // stage rows of one 32-channel block (64 B of hi halves + 64 B of lo halves each)
constexpr int UMAX = XROWS - 1;             // slot UMAX holds zeros
constexpr int NPASS_MAX = 3;
constexpr int UCAP = NPASS_MAX * XROWS;     // row-list capacity of a record (both tile shapes)
constexpr int X_BYTES = XROWS * 128;        // 40 KB: four workgroups per CU
constexpr int LO_REGION = XROWS * 64;
constexpr int LOC_OFF = 16 + UCAP * 4;
constexpr int MASK_PASS_BYTES = 56;
template <int TILE>
struct Rec {
  static constexpr int NPASS = TILE == 64 ? 3 : 2;   // staging passes a tile may take: 638 distinct rows for 128-row tiles, 957 for the strided tables' 64-row tiles
  static constexpr int LROWS = TILE / 4;             // rulebook entries per offset: one uint2 (four 16-bit slots) per (64-row group, j)
  static constexpr int MASK_OFF = LOC_OFF + NPASS * 27 * LROWS * 8;
  static constexpr int LR_BYTES = (MASK_OFF + NPASS * MASK_PASS_BYTES + 127) / 128 * 128;   // 17792 (128 rows) / 14464 (64 rows)
  static_assert(LR_BYTES % 128 == 0, "records start on cache lines");
};
constexpr int HSLOTS = 2048;
__host__ __device__ constexpr int slot_addr(int l) { return l * 64 + ((l >> 2) & 3) * 16; }   // as in spconv_st.hip

// One workgroup (128 threads) per 128-row tile: the distinct input rows of the tile's 27-neighbourhoods through an LDS hash
// (slots numbered in table order: deterministic), the stage slot of every (offset, row) as a 16-bit LDS byte address, and
// per (pass, offset) an 8-bit occupancy mask (bit 4 w + c: some row of rows 64 w + 16 c .. + 15 has a neighbour staged in the pass).
template <int TILE>
__global__ __launch_bounds__(TILE) void k_local_rulebook_t(const int32_t* __restrict__ nbr, int K, int n_out, unsigned char* __restrict__ out,
                                                           int* __restrict__ overflow) {
  constexpr int NWB = TILE / 64, LROWS = Rec<TILE>::LROWS, MASK_OFF = Rec<TILE>::MASK_OFF, LR_BYTES = Rec<TILE>::LR_BYTES, NPASS = Rec<TILE>::NPASS;
  __shared__ int hk[HSLOTS];
  __shared__ unsigned short hid[HSLOTS], hpos[HSLOTS];
  __shared__ int wave_cnt[NWB];
  __shared__ int too_many;
  __shared__ unsigned char nib[NPASS][27][NWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x;
  const int row = tile * TILE + (int)threadIdx.x;
  for (int i = threadIdx.x; i < HSLOTS; i += TILE) hk[i] = -1;
  if (threadIdx.x == 0) too_many = 0;
  __syncthreads();
  unsigned short slot[27];
  int idxs[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) idxs[k] = (k < K && row < n_out) ? nbr[(size_t)k * n_out + row] : -1;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int idx = idxs[k];
    unsigned int s = 0xFFFFu;
    if (idx >= 0) {
      s = ((unsigned)idx * 2654435761u) >> 21;
      int probes = 0;
      while (true) {
        const int prev = atomicCAS(&hk[s], -1, idx);
        if (prev == -1 || prev == idx) break;
        s = (s + 1) & (HSLOTS - 1);
        if (++probes >= HSLOTS) { too_many = 1; s = 0xFFFFu; break; }
      }
    }
    slot[k] = (unsigned short)s;
  }
  __syncthreads();
  unsigned char* lr = out + (size_t)tile * LR_BYTES;
  if (too_many) {
    if (threadIdx.x == 0) { reinterpret_cast<int*>(lr)[0] = -1; atomicAdd(overflow, 1); }
    return;
  }
  constexpr int PER_WAVE = HSLOTS / NWB;
  int cnt = 0;
  for (int i0 = 0; i0 < PER_WAVE; i0 += 64) cnt += __popcll(__ballot(hk[wave * PER_WAVE + i0 + lane] >= 0));
  if (lane == 0) wave_cnt[wave] = cnt;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < NWB; ++w) { if (w < wave) base += wave_cnt[w]; total += wave_cnt[w]; }
  int* U = reinterpret_cast<int*>(lr + 16);
  if (total <= UMAX) {                                                 // one pass: any numbering gives the same sums - slot order
    for (int i0 = 0; i0 < PER_WAVE; i0 += 64) {
      const int s = wave * PER_WAVE + i0 + lane;
      const int key = hk[s];
      const unsigned long long m = __ballot(key >= 0);
      const int id = base + __popcll(m & ((1ull << lane) - 1ull));
      if (key >= 0) {
        hid[s] = (unsigned short)id;
        U[id] = key;
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
      const int key = hk[s];
      if (key >= 0) {
        const int id = canonical_slot_id<HSLOTS>(hk, hpos, s);
        hid[s] = (unsigned short)id;
        if (id < NPASS * UMAX) U[id] = key;
      }
    }
  }
  if (threadIdx.x == 0) {
    reinterpret_cast<int*>(lr)[0] = total <= NPASS * UMAX ? total : -1;
    if (total > NPASS * UMAX) atomicAdd(overflow, 1);
  }
  __syncthreads();
  unsigned short* loc = reinterpret_cast<unsigned short*>(lr + LOC_OFF);
  const int r = (int)threadIdx.x, w = r >> 6, c = (r >> 4) & 3, j = r & 15;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int id = slot[k] != 0xFFFFu ? (int)hid[slot[k]] : -1;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      if (p > 0 && total <= p * UMAX) continue;
      const int l = (id >= p * UMAX && id < (p + 1) * UMAX) ? id - p * UMAX : UMAX;
      loc[(((size_t)(p * 27 + k) * LROWS) + w * 16 + j) * 4 + c] = (unsigned short)slot_addr(l);
      const unsigned long long b = __ballot(l != UMAX);
      if (lane == 0)
        nib[p][k][w] = (unsigned char)(((b & 0xFFFFull) != 0) | (((b >> 16) & 0xFFFFull) != 0) << 1 | (((b >> 32) & 0xFFFFull) != 0) << 2 |
                                       (((b >> 48) & 0xFFFFull) != 0) << 3);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NPASS * 28; i += TILE) {
    const int p = i / 28, k = i % 28;
    unsigned short m = 0;
    if (k < 27 && !(p > 0 && total <= p * UMAX)) m = (unsigned short)(nib[p][k][0] | (NWB > 1 ? nib[p][k][NWB - 1] << 4 : 0));
    reinterpret_cast<unsigned short*>(lr + MASK_OFF)[p * 28 + k] = m;
  }
}

#include "spconv_st_loop.inc"

// A workgroup = TILE rows x CTG output channels: TILE / 64 row groups x CTG / 32 channel parts, one wave (64 rows x 32 channels) each.
//   TILE = 128: NWV = 4 -> 64 output channels (>= 64-channel layers), NWV = 2 -> 32 (32-channel layers);
//   TILE = 64 (strided tables): NWV = 2 -> 64 output channels, NWV = 4 -> 128.
template <int TILE, int CC, int SKIP, int NWV>
__global__ __launch_bounds__(NWV * 64, NWV) void spconv_st128_kernel(SpconvArgs a, const unsigned char* __restrict__ local, int n_tiles) {
  constexpr int NTW = 2, CTW = NTW * 16, NC = 4, RW = TILE / 64;
  constexpr int LROWS = Rec<TILE>::LROWS, MASK_OFF = Rec<TILE>::MASK_OFF, LR_BYTES = Rec<TILE>::LR_BYTES;
  constexpr int CTG = CTW * (NWV / RW);
  constexpr int NITV = XROWS / (16 * NWV);
  __shared__ __attribute__((aligned(128))) unsigned char xs[X_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane >> 4, j = lane & 15;
  const int n_cg = a.cout / CTG;
  const int xcd = (int)blockIdx.x & 7, per = 8 / n_cg;
  const int cg = xcd % n_cg, tile = ((int)blockIdx.x >> 3) * per + xcd / n_cg;
  if (tile >= n_tiles) return;
  const int w0 = wave % RW;
  const int ct0 = cg * CTG + (wave / RW) * CTW;
  const int CT = a.cout >= 128 ? 128 : a.cout;
  const int n_slices = a.cout / CT, slice = ct0 / CT, nt0 = (ct0 - slice * CT) / 16;
  constexpr int JQ = CC / 16;
  const int ncc = a.cin / CC;
  const int nqb = a.cin / 32;
  constexpr int K = 27;

  const unsigned char* lr = local + (size_t)tile * LR_BYTES;
  const int* __restrict__ U = reinterpret_cast<const int*>(lr + 16);
  if (threadIdx.x < 8)
    *reinterpret_cast<float4*>(xs + (threadIdx.x >> 2) * LO_REGION + UMAX * 64 + (threadIdx.x & 3) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

  const unsigned long long wbits = (unsigned long long)(size_t)a.w;
  u32x4 wr;
  wr[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wbits);
  wr[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(wbits >> 32) & 0xFFFFu));
  wr[2] = (unsigned)__builtin_amdgcn_readfirstlane(K * a.cin * a.cout * 4);
  wr[3] = 0x00020000u;
  const int tile4 = CC * CT / 4;
  const unsigned int kstride = (unsigned)(n_slices * ncc * tile4 * 16);
  const unsigned int w1off = JQ * 1024;
  if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)xs != 0u) __builtin_trap();   // the loop addresses the stage from LDS byte 0

  f32x16 A0, A1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { A0[i] = 0.f; A1[i] = 0.f; }

  int n_up = 0;
  int Ureg[NITV];
  auto load_rows = [&](int pass) {
#pragma unroll
    for (int it = 0; it < NITV; ++it) Ureg[it] = U[pass * UMAX + (it * NWV + wave) * 16 + (lane >> 2)];
  };
  auto stage = [&](int qb) {
#pragma unroll
    for (int it = 0; it < NITV; ++it) {
      const int l0 = (it * NWV + wave) * 16;
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
  int warm = 0;      // the tile's rulebook entries (6.9 KB per pass) into L2 while the stage is in flight: one dword per line
  if (threadIdx.x < 27 * LROWS * 8 / 128) warm = *reinterpret_cast<const int*>(lr + LOC_OFF + threadIdx.x * 128);
  const int n_u = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(lr)[0]);
  const int n_pass = (n_u + UMAX - 1) / UMAX;
  bool first = true;
  for (int pass = 0; pass < n_pass; ++pass) {
    n_up = min(n_u - pass * UMAX, UMAX);
    if (pass > 0) load_rows(pass);
    const unsigned int* mp = reinterpret_cast<const unsigned int*>(lr + MASK_OFF + pass * MASK_PASS_BYTES);
    u32x8 M0, M1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      M0[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)mp[i]) >> (w0 * 4);
      M1[i] = i < 6 ? (unsigned)__builtin_amdgcn_readfirstlane((int)mp[8 + i]) >> (w0 * 4) : 0u;
    }
    const unsigned char* lb = lr + LOC_OFF + ((size_t)pass * 27 * LROWS + w0 * 16) * 8;
    for (int qb = 0; qb < nqb; ++qb) {
      if (!first) __syncthreads();
      first = false;
      stage(qb);
      const int cc = (qb * 32) / CC, qp = ((qb * 32) % CC) / 32;
      const unsigned int ws0 = (unsigned)__builtin_amdgcn_readfirstlane(((slice * ncc + cc) * tile4 + (nt0 * JQ + 2 * qp) * 64) * 16);
      unsigned int so;
#define EYOC_ST128_ASM(TEXT)                                                                                                                 \
  asm volatile(TEXT : "+{v[96:111]}"(A0), "+{v[112:127]}"(A1), [so] "=&s"(so)                                                                  \
               : [wr] "s"(wr), [ws0] "s"(ws0), [ks] "s"(kstride), [lb] "s"(lb), [w1] "s"(w1off), "{s[36:43]}"(M0), "{s[44:51]}"(M1)         \
               : "memory", "scc", EYOC_ST_LOOP_CLOBBERS_T128)
      if constexpr (TILE == 128 && SKIP) EYOC_ST128_ASM(EYOC_ST_LOOP_T128);
      else if constexpr (TILE == 128) EYOC_ST128_ASM(EYOC_ST_LOOP_T128_NOSKIP);
      else if constexpr (SKIP) EYOC_ST128_ASM(EYOC_ST_LOOP_T64);
      else EYOC_ST128_ASM(EYOC_ST_LOOP_T64_NOSKIP);
#undef EYOC_ST128_ASM
    }
  }
  asm volatile("" :: "v"(warm));

  // ---- epilogue from the registers: lane (g, j) holds, for row 64 w0 + 16 c + j, channels ct0 + 8 g .. + 3 (t = 0) and + 4 .. + 7
  // (t = 1) in accumulator ACC(c, t) = v[96 + (2 c + t) 4 ..]: 16-byte accesses per SPLIT16 half.  Loads, values, stores in
  // three sweeps (everything in flight at once; the blob's registers are free again)
  const float os = a.out_scale ? *a.out_scale : 1.0f;
  const int ch = ct0 + 8 * g;
  float4 b4[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) b4[t] = a.bias ? *reinterpret_cast<const float4*>(a.bias + ch + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
  float mx = 0.f;
  uint4 rh[NC], rl[NC];
  if (a.res) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int o = tile * TILE + w0 * 64 + 16 * c + j;
      if (o < a.n_out) {
        const char* rp = reinterpret_cast<const char*>(a.res + (size_t)o * a.ld_res) + split16_off4(ch);
        rh[c] = *reinterpret_cast<const uint4*>(rp);
        rl[c] = *reinterpret_cast<const uint4*>(rp + SPLIT16_LO);
      }
    }
  }
  float4 v[NC][NTW];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int o = tile * TILE + w0 * 64 + 16 * c + j;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int ai = (c * NTW + t) * 4;
      const f32x16& A = ai < 16 ? A0 : A1;
      v[c][t] = make_float4(A[ai % 16 + 0] * os + b4[t].x, A[ai % 16 + 1] * os + b4[t].y, A[ai % 16 + 2] * os + b4[t].z, A[ai % 16 + 3] * os + b4[t].w);
    }
    if (a.res && o < a.n_out) {
      const float4 q0 = split16_decode4(make_uint2(rh[c].x, rh[c].y), make_uint2(rl[c].x, rl[c].y));
      const float4 q1 = split16_decode4(make_uint2(rh[c].z, rh[c].w), make_uint2(rl[c].z, rl[c].w));
      v[c][0].x += q0.x; v[c][0].y += q0.y; v[c][0].z += q0.z; v[c][0].w += q0.w;
      v[c][1].x += q1.x; v[c][1].y += q1.y; v[c][1].z += q1.z; v[c][1].w += q1.w;
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      if (a.relu) { v[c][t].x = fmaxf(v[c][t].x, 0.f); v[c][t].y = fmaxf(v[c][t].y, 0.f); v[c][t].z = fmaxf(v[c][t].z, 0.f); v[c][t].w = fmaxf(v[c][t].w, 0.f); }
      if (o < a.n_out) split16_track(mx, v[c][t]);
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int o = tile * TILE + w0 * 64 + 16 * c + j;
    if (o >= a.n_out) continue;
    if (a.out_split) {
      uint2 h0, l0, h1, l1;
      split16_encode4(v[c][0], h0, l0);
      split16_encode4(v[c][1], h1, l1);
      char* op = reinterpret_cast<char*>(a.out + (size_t)o * a.ld_out) + split16_off4(ch);
      *reinterpret_cast<uint4*>(op) = make_uint4(h0.x, h0.y, h1.x, h1.y);
      *reinterpret_cast<uint4*>(op + SPLIT16_LO) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    } else {
      float* op = a.out + (size_t)o * a.ld_out + ch;
      *reinterpret_cast<float4*>(op) = v[c][0];
      *reinterpret_cast<float4*>(op + 4) = v[c][1];
    }
  }
  if (a.out_split) split16_report(a.range, mx);
}

// select_st_tile: which tile shape new maps build records for (and the kernel that reads them).  256 by default: measured on the
// 16-pair layer bench (scripts/bench_staged.py, variants interleaved) the 128-row kernel is LEVEL with or up to 7 % behind the
// 256-row one on every layer (level-0 64->64 0.475 vs 0.443 ms, 32->32 0.161 vs 0.148, level-1 0.199 vs 0.197, level-2 128->128
// 0.277 vs 0.259, level-3 256->256 0.335 vs 0.322) although four workgroups share a CU: co-residency was not the limiter.
std::atomic<int> g_st_tile{256};

}  // namespace

namespace eyoc {

int select_st_tile(int rows) { return (rows == 128 || rows == 256) ? g_st_tile.exchange(rows) : g_st_tile.load(); }

size_t local_rulebook128_bytes(int n_out) { return (size_t)cdiv(n_out, 128) * Rec<128>::LR_BYTES; }
size_t local_rulebook64_bytes(int n_out) { return (size_t)cdiv(n_out, 64) * Rec<64>::LR_BYTES; }

int build_local_rulebook128(const int32_t* nbr_dev, int K, int n_out, unsigned char* out_dev, int* overflow_dev, hipStream_t st) {
  if (n_out <= 0) return EYOC_OK;
  hipLaunchKernelGGL(k_local_rulebook_t<128>, dim3(cdiv(n_out, 128)), dim3(128), 0, st, nbr_dev, K, n_out, out_dev, overflow_dev);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

int build_local_rulebook64(const int32_t* nbr_dev, int K, int n_out, unsigned char* out_dev, int* overflow_dev, hipStream_t st) {
  if (n_out <= 0) return EYOC_OK;
  hipLaunchKernelGGL(k_local_rulebook_t<64>, dim3(cdiv(n_out, 64)), dim3(64), 0, st, nbr_dev, K, n_out, out_dev, overflow_dev);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

// tile = 128 (stride-1 tables) or 64 (strided tables: 64 .. 256 output channels)
int launch_spconv_st128(const SpconvArgs& a, const unsigned char* local_dev, int tile, int skip, hipStream_t st) {
  EYOC_REQUIRE(a.math == 1 && local_dev && !a.l2norm && a.K == 27 && !a.perm && (tile == 128 || tile == 64), EYOC_ERR_INVALID,
               "spconv_st128: unsupported layer");
  const int n_tiles = cdiv(a.n_out, tile);
  const int ctg = tile == 128 ? (a.cout >= 64 ? 64 : 32) : (a.cout >= 128 ? 128 : 64);
  const bool wide = spconv_cc(a.cin, a.cout) == 64;
  const int n_cg = a.cout / ctg;
  EYOC_REQUIRE(a.cout % ctg == 0 && n_cg >= 1 && n_cg <= 8 && 8 % n_cg == 0 && a.cin % 32 == 0, EYOC_ERR_INVALID,
               "spconv_st128: %d -> %d channels on %d-row tiles", a.cin, a.cout, tile);
  const dim3 grid((unsigned)(cdiv(n_tiles, 8 / n_cg) * 8));
#define EYOC_ST128(T_, CC_, SK_, NWV_) hipLaunchKernelGGL((spconv_st128_kernel<T_, CC_, SK_, NWV_>), grid, dim3(NWV_ * 64), 0, st, a, local_dev, n_tiles)
#define EYOC_ST128_CC(T_, NWV_)                                                                    \
  do {                                                                                             \
    if (skip) { if (wide) EYOC_ST128(T_, 64, 1, NWV_); else EYOC_ST128(T_, 32, 1, NWV_); }         \
    else { if (wide) EYOC_ST128(T_, 64, 0, NWV_); else EYOC_ST128(T_, 32, 0, NWV_); }              \
  } while (0)
  if (tile == 128) { if (ctg == 64) EYOC_ST128_CC(128, 4); else EYOC_ST128_CC(128, 2); }
  else { if (ctg == 128) EYOC_ST128_CC(64, 4); else EYOC_ST128_CC(64, 2); }
#undef EYOC_ST128_CC
#undef EYOC_ST128
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

}  // namespace eyoc
