// Is the fp16 matrix pipe's sustained rate data-dependent (power / clock management)?  The same dense stream of
// v_mfma_f32_16x16x32_f16 - 8 independent accumulators, 8 A and 8 B fragments per wave, no memory traffic - with all-zero
// operands, with one constant, and with pseudo-random fp16 operands (every bit toggling between consecutive instructions).
// hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power && ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline unsigned int mix(unsigned int x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void k(float* out, int iters, int mode) {
  f32x4 acc[8];
  half8 A[8], B[8];
  for (int i = 0; i < 8; ++i) {
    acc[i] = f32x4{0, 0, 0, 0};
    for (int e = 0; e < 8; ++e) {
      const unsigned int r = mix((threadIdx.x * 8 + i) * 8 + e + blockIdx.x * 65536u);
      // mode 0: zeros; 1: the constant 1.0; 2: random values in [-1, 1) with random mantissas
      const float va = mode == 0 ? 0.f : mode == 1 ? 1.f : ((int)(r & 0xFFFF) - 32768) / 32768.f;
      const float vb = mode == 0 ? 0.f : mode == 1 ? 1.f : ((int)(r >> 16) - 32768) / 32768.f;
      A[i][e] = (_Float16)va; B[i][e] = (_Float16)vb;
    }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[(i + r) & 7], B[i], acc[i], 0, 0, 0);
    if (mode == 3) for (int i = 0; i < 8; ++i) acc[i] *= 0.5f;          // (unused)
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same with v_mfma_f32_32x32x16_f16 (twice the flop per instruction and per operand byte read from the register file)
__global__ __launch_bounds__(256) void k32(float* out, int iters, int mode) {
  f32x16 acc[4];
  half8 A[8], B[4];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 8; ++e) {
      const unsigned int r = mix((threadIdx.x * 8 + i) * 8 + e + blockIdx.x * 65536u);
      const float va = mode == 0 ? 0.f : mode == 1 ? 1.f : ((int)(r & 0xFFFF) - 32768) / 32768.f;
      const float vb = mode == 0 ? 0.f : mode == 1 ? 1.f : ((int)(r >> 16) - 32768) / 32768.f;
      A[i][e] = (_Float16)va;
      if (i < 4) B[i][e] = (_Float16)vb;
    }
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[(i + r) & 7], B[i], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out; hipMalloc(&out, (size_t)4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"all-zero operands", "constant 1.0", "random operands"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode)
      for (int blocks : {256, 512}) {
        const int iters = 20000;                                       // ~20-40 ms: long enough for the power management to settle
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1000, mode);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 64 * 16384.0;
        printf("%-20s %d waves/SIMD: %7.2f ms  %7.1f TFLOP/s\n", names[mode], blocks / 256, ms, flop / ms / 1e9);
      }
  for (int mode = 0; mode < 3; ++mode)
    for (int blocks : {256, 512}) {
      const int iters = 20000;
      hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, 1000, mode);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters, mode);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)blocks * 4 * iters * 32 * 32768.0;
      printf("32x32x16 %-20s %d waves/SIMD: %7.2f ms  %7.1f TFLOP/s\n", names[mode], blocks / 256, ms, flop / ms / 1e9);
    }
  return 0;
}
