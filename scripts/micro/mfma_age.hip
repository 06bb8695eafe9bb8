// Do two waves on one SIMD share the matrix pipe evenly?  (profiles/r6_staged_trace.txt: a staged workgroup's FIRST offset loop - the
// younger of the CU's two workgroups - is 40 % slower than its second.)  One workgroup of 8 waves per CU = two waves per SIMD.  Waves
// 0-3 (one per SIMD) start a dense v_mfma_f32_16x16x32_f16 stream at once; waves 4-7 - launched in the same workgroup, so "younger" only
// by wave slot - sleep first, then run the same stream.  Every wave stamps s_memrealtime (100 MHz) at the start, after each quarter of
// its stream and at the end.  While both waves of a SIMD stream, an even arbiter would give each half the pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int CHAIN>
__global__ __launch_bounds__(512, 1) void k_age(unsigned long long* stamps, float* out, int iters, int sleep_late) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane * 3 + i)); }
  if (wave >= 4) for (int z = 0; z < sleep_late; ++z) __builtin_amdgcn_s_sleep(127);
  unsigned long long t[5];
  t[0] = __builtin_amdgcn_s_memrealtime();
  for (int q = 0; q < 4; ++q) {
    for (int it = 0; it < iters / 4; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[CHAIN ? 0 : i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[CHAIN ? 0 : i], 0, 0, 0);
    }
    asm volatile("s_nop 0" ::: "memory");
    t[q + 1] = __builtin_amdgcn_s_memrealtime();
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0)
    for (int q = 0; q < 5; ++q) stamps[(blockIdx.x * 8 + wave) * 5 + q] = t[q];
}

int main() {
  const int blocks = 256, iters = 40000;
  unsigned long long* st; float* out;
  (void)hipMalloc(&st, blocks * 8 * 5 * 8); (void)hipMalloc(&out, blocks * 512 * 4);
  for (int chain = 0; chain < 2; ++chain)
    for (int sleep_late : {0, 100}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (chain) hipLaunchKernelGGL(k_age<1>, dim3(blocks), dim3(512), 0, 0, st, out, iters, sleep_late);
        else hipLaunchKernelGGL(k_age<0>, dim3(blocks), dim3(512), 0, 0, st, out, iters, sleep_late);
      }
      (void)hipDeviceSynchronize();
      std::vector<unsigned long long> h(blocks * 8 * 5);
      (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
      double q_early[4] = {0, 0, 0, 0}, q_late[4] = {0, 0, 0, 0}, gap = 0;
      for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 8; ++w)
          for (int q = 0; q < 4; ++q) (w < 4 ? q_early : q_late)[q] += (double)(h[(b * 8 + w) * 5 + q + 1] - h[(b * 8 + w) * 5 + q]) / 100.0 / (blocks * 4);
      for (int b = 0; b < blocks; ++b) gap += (double)(h[(b * 8 + 4) * 5] - h[(b * 8) * 5]) / 100.0 / blocks;
      printf("%s, late waves start %.1f us after the early ones: quarters of %d MFMAs per wave\n  early waves %7.2f %7.2f %7.2f %7.2f us\n  late  waves %7.2f %7.2f %7.2f %7.2f us\n",
             chain ? "one dependent accumulator chain" : "8 independent accumulators", gap, iters / 4 * 8, q_early[0], q_early[1], q_early[2], q_early[3], q_late[0],
             q_late[1], q_late[2], q_late[3]);
    }
  return 0;
}
