// Calibration: does non-MFMA work overlap with v_mfma_f32_16x16x4_f32 (a) inside one wave, (b) across two waves of a SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (a) one wave per SIMD: NV independent VALU ops (v_fma chains on 8 registers) per MFMA, same wave
template <int NV>
__global__ __launch_bounds__(256) void k_same(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(i + j) & 7] = __builtin_fmaf(v[(i + j) & 7], 1.0001f, 0.5f);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// (b) workgroup of 8 waves = 2 per SIMD: even waves stream MFMAs, odd waves run a VALU chain
__global__ __launch_bounds__(512) void k_cross(float* out, int iters_m, int iters_v, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool mfma_wave = (wave & 4) == 0;   // waves 0-3 -> SIMD 0-3, waves 4-7 -> SIMD 0-3 again
  float s = 0;
  if (mfma_wave) {
    if (mode & 1) {
      f32x4 acc[8];
      for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
      float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
      for (int it = 0; it < iters_m; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else {
    if (mode & 2) {
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
      for (int it = 0; it < iters_v; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
      for (int i = 0; i < 8; ++i) s += v[i];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f();
  (void)hipEventRecord(e0);
  f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
template <int NV>
void same() {
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  const int iters = 4000;
  float ms = timeit([&] { hipLaunchKernelGGL(k_same<NV>, dim3(256), dim3(256), 0, 0, out, iters); });
  printf("same wave, %d VALU per MFMA: %.3f ms (%.1f cycles per MFMA at 2.4 GHz)\n", NV, ms, ms * 1e-3 * 2.4e9 / (iters * 8.0));
  (void)hipFree(out);
}
int main() {
  same<0>(); same<2>(); same<4>(); same<6>(); same<7>(); same<8>(); same<12>();
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int im = 4000, iv = 4000 * 7;   // VALU wave: 7 fma per MFMA slot -> 28 of 32 cycles if it ran alone
  for (int mode = 1; mode <= 3; ++mode) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_cross, dim3(256), dim3(512), 0, 0, out, im, iv, mode); });
    printf("two waves per SIMD, mode %d (1 = MFMA wave only, 2 = VALU wave only, 3 = both): %.3f ms\n", mode, ms);
  }
  return 0;
}
