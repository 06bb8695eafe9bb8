// Calibration for the hand-scheduled staged loop (gen_st_loop.py): cycles per v_mfma_f32_16x16x32_f16 as a function of
// (a) the distance between two MFMAs on the SAME accumulator, (b) LDS reads / VALU interleaved, (c) scalar branches taken
// around 6-MFMA groups (near stub vs far stub), at 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_dep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define M(acc) "v_mfma_f32_16x16x32_f16 " acc ", v[40:43], v[44:47], " acc "\n\t"

// distance D between MFMAs on the same accumulator: D accumulators used round-robin; 48 MFMAs per iteration
template <int D>
__global__ __launch_bounds__(1024) void k_dist(float* out, int iters, long long* cyc) {
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (D == 1) { asm volatile(M("v[0:3]") M("v[0:3]") M("v[0:3]") M("v[0:3]") M("v[0:3]") M("v[0:3]") M("v[0:3]") M("v[0:3]") ::: "v0","v1","v2","v3"); }
    if constexpr (D == 2) { asm volatile(M("v[0:3]") M("v[4:7]") M("v[0:3]") M("v[4:7]") M("v[0:3]") M("v[4:7]") M("v[0:3]") M("v[4:7]") ::: "v0","v1","v2","v3","v4","v5","v6","v7"); }
    if constexpr (D == 4) { asm volatile(M("v[0:3]") M("v[4:7]") M("v[8:11]") M("v[12:15]") M("v[0:3]") M("v[4:7]") M("v[8:11]") M("v[12:15]") ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15"); }
    if constexpr (D == 8) { asm volatile(M("v[0:3]") M("v[4:7]") M("v[8:11]") M("v[12:15]") M("v[16:19]") M("v[20:23]") M("v[24:27]") M("v[28:31]") ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"); }
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = 0.f;
}

// the chunk group of the generated loop: m m r m m r m m on two accumulators, 4 chunks (8 accumulators) per "half-step",
// MODE 0: exactly that; 1: term-major order over the 4 chunks (distance 8) with the same reads; 2: MODE 0 + 4 VALU per chunk
template <int MODE>
__global__ __launch_bounds__(512) void k_group(float* out, int iters, long long* cyc) {
  __shared__ float lds[4096];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned a0 = (threadIdx.x & 63) * 16, a1 = a0 ^ 64;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define R(dst, a) "ds_read_b128 " dst ", %" a "\n\t"
#define V4 "v_lshlrev_b32 v60, 4, v61\n\tv_and_b32 v60, 0xffff0, v60\n\tv_xad_u32 v60, v60, v61, 0\n\tv_xor_b32 v62, 64, v60\n\t"
#define G(c0, c1, x0, x1) "s_waitcnt lgkmcnt(6)\n\t" M(c0) M(c1) R(x0, "0") M(c0) M(c1) R(x1, "1") M(c0) M(c1)
    if constexpr (MODE == 0)
      asm volatile(G("v[0:3]", "v[4:7]", "v[48:51]", "v[52:55]") G("v[8:11]", "v[12:15]", "v[48:51]", "v[52:55]")
                   G("v[16:19]", "v[20:23]", "v[48:51]", "v[52:55]") G("v[24:27]", "v[28:31]", "v[48:51]", "v[52:55]") "s_waitcnt lgkmcnt(0)\n\t"
                   :: "v"(a0), "v"(a1) : "memory", "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v48","v49","v50","v51","v52","v53","v54","v55");
    if constexpr (MODE == 2)
      asm volatile(V4 G("v[0:3]", "v[4:7]", "v[48:51]", "v[52:55]") V4 G("v[8:11]", "v[12:15]", "v[48:51]", "v[52:55]")
                   V4 G("v[16:19]", "v[20:23]", "v[48:51]", "v[52:55]") V4 G("v[24:27]", "v[28:31]", "v[48:51]", "v[52:55]") "s_waitcnt lgkmcnt(0)\n\t"
                   :: "v"(a0), "v"(a1) : "memory", "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v48","v49","v50","v51","v52","v53","v54","v55","v60","v61","v62");
    if constexpr (MODE == 1)
      asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                   M("v[0:3]") M("v[4:7]") M("v[8:11]") R("v[48:51]", "0") M("v[12:15]") M("v[16:19]") M("v[20:23]") R("v[52:55]", "1") M("v[24:27]") M("v[28:31]")
                   M("v[0:3]") M("v[4:7]") M("v[8:11]") R("v[48:51]", "0") M("v[12:15]") M("v[16:19]") M("v[20:23]") R("v[52:55]", "1") M("v[24:27]") M("v[28:31]")
                   M("v[0:3]") M("v[4:7]") M("v[8:11]") R("v[48:51]", "0") M("v[12:15]") M("v[16:19]") M("v[20:23]") R("v[52:55]", "1") M("v[24:27]") M("v[28:31]")
                   R("v[48:51]", "0") R("v[52:55]", "1")
                   :: "v"(a0), "v"(a1) : "memory", "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v48","v49","v50","v51","v52","v53","v54","v55");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = lds[threadIdx.x];
}

// skip cost: `mask` decides per group whether its 6 MFMAs run; NEAR: forward branch over them; FAR: out-of-line stub 32 KB away
template <bool FAR>
__global__ __launch_bounds__(512) void k_skip(float* out, int iters, unsigned mask, long long* cyc) {
  unsigned m = __builtin_amdgcn_readfirstlane(mask);
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define GN(bit, c0, c1) "s_bitcmp1_b32 %0, " bit "\n\ts_cbranch_scc0 .Ln%=_" bit "\n\t" M(c0) M(c1) M(c0) M(c1) M(c0) M(c1) ".Ln%=_" bit ":\n\t"
#define GF(bit, c0, c1) "s_bitcmp1_b32 %0, " bit "\n\ts_cbranch_scc0 .Lf%=_s" bit "\n\t" M(c0) M(c1) M(c0) M(c1) M(c0) M(c1) ".Lf%=_j" bit ":\n\t"
#define ST(bit) ".Lf%=_s" bit ":\n\ts_nop 0\n\ts_branch .Lf%=_j" bit "\n\t"
    if constexpr (!FAR)
      asm volatile(GN("0", "v[0:3]", "v[4:7]") GN("1", "v[8:11]", "v[12:15]") GN("2", "v[16:19]", "v[20:23]") GN("3", "v[24:27]", "v[28:31]")
                   GN("4", "v[0:3]", "v[4:7]") GN("5", "v[8:11]", "v[12:15]") GN("6", "v[16:19]", "v[20:23]") GN("7", "v[24:27]", "v[28:31]")
                   :: "s"(m) : "scc", "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
    else
      asm volatile(GF("0", "v[0:3]", "v[4:7]") GF("1", "v[8:11]", "v[12:15]") GF("2", "v[16:19]", "v[20:23]") GF("3", "v[24:27]", "v[28:31]")
                   GF("4", "v[0:3]", "v[4:7]") GF("5", "v[8:11]", "v[12:15]") GF("6", "v[16:19]", "v[20:23]") GF("7", "v[24:27]", "v[28:31]")
                   "s_branch .Lf%=_end\n\t"
                   ".fill 8192, 4, 0xbf800000\n\t"      /* 32 KB of s_nop between the loop and its stubs */
                   ST("0") ST("1") ST("2") ST("3") ST("4") ST("5") ST("6") ST("7")
                   ".Lf%=_end:\n\t"
                   :: "s"(m) : "scc", "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = 0.f;
}

int main() {
  float* out; long long* cyc; long long h;
  (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&cyc, 8);
  const int iters = 20000;
#define RUN(name, kern, threads, nm, ...)                                                                    \
  do {                                                                                                       \
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 100, ##__VA_ARGS__, cyc);                \
    (void)hipDeviceSynchronize();                                                                            \
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, ##__VA_ARGS__, cyc);              \
    (void)hipDeviceSynchronize();                                                                            \
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                                      \
    printf("%-44s %d waves/SIMD: %7.2f clk-counter ticks per MFMA slot (%d slots/iter)\n", name, threads / 256, (double)h / iters / nm, nm); \
  } while (0)
  // wall-clock rate of the plain MFMA stream (distance 8) at 1 / 2 / 4 waves per SIMD, one workgroup per CU
  for (int threads : {256, 512, 1024}) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_dist<8>), dim3(256), dim3(threads), 0, 0, out, 1000, cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_dist<8>), dim3(256), dim3(threads), 0, 0, out, 200000, cyc);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = 256.0 * (threads / 64) * 200000.0 * 8;
    printf("plain stream %d waves/SIMD: %.3f ms, %.0f TFLOP/s, wave 0: %.2f ticks per MFMA, tick rate %.2f GHz\n", threads / 256, ms,
           mf * 16384 / ms / 1e9, (double)h / 200000 / 8, (double)h / ms / 1e6);
  }
  for (int threads : {256, 512}) {
    RUN("same-accumulator distance 1", (k_dist<1>), threads, 8);
    RUN("same-accumulator distance 2", (k_dist<2>), threads, 8);
    RUN("same-accumulator distance 4", (k_dist<4>), threads, 8);
    RUN("same-accumulator distance 8", (k_dist<8>), threads, 8);
    RUN("chunk groups m m r m m r m m (dist 2)", (k_group<0>), threads, 24);
    RUN("  + 4 VALU per chunk", (k_group<2>), threads, 24);
    RUN("term-major over 4 chunks (dist 8) + 8 reads", (k_group<1>), threads, 24);
    RUN("near branches, all groups active", (k_skip<false>), threads, 48, 0xFFu);
    RUN("near branches, half skipped (slots=issued+skipped)", (k_skip<false>), threads, 48, 0x55u);
    RUN("near branches, all skipped", (k_skip<false>), threads, 48, 0x00u);
    RUN("far stubs, all active", (k_skip<true>), threads, 48, 0xFFu);
    RUN("far stubs, half skipped", (k_skip<true>), threads, 48, 0x55u);
    RUN("far stubs, all skipped", (k_skip<true>), threads, 48, 0x00u);
  }
  return 0;
}
