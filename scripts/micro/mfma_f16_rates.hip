// Issue rates of the fp16 MFMA shapes on gfx950 (which one should the split-fp16 sparse convolution use?).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_f16_rates.hip -o scripts/micro/build/mfma_f16_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)1.0f; }
  float s = 0;
  if (MODE == 0) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char* name, float* d, int waves_per_simd) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 8192 / NACC, blocks = 256 * waves_per_simd;   // 256 CUs x 4 SIMDs: one 256-thread block = one wave per SIMD of a CU
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double n_inst = (double)blocks * 4 * iters * NACC;
  const double flop = n_inst * (MODE == 0 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2);
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * NACC * waves_per_simd);
  printf("%-24s acc %d waves/SIMD %d: %.3f ms  %.0f TFLOP/s  %.1f cycles/instr/SIMD\n", name, NACC, waves_per_simd, ms, flop / ms / 1e9, cyc);
}

int main() {
  float* d; (void)hipMalloc(&d, 1 << 24);
  for (int w : {1, 2, 4}) {
    run<0, 4>("mfma_f32_16x16x32_f16", d, w);
    run<0, 8>("mfma_f32_16x16x32_f16", d, w);
    run<1, 2>("mfma_f32_32x32x16_f16", d, w);
    run<1, 4>("mfma_f32_32x32x16_f16", d, w);
  }
  return 0;
}
