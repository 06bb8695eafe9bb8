// Calibration: v_mfma_f32_16x16x32_bf16 rate, and whether independent VALU work of the same wave hides behind it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NV>
__global__ __launch_bounds__(256) void k_same(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + threadIdx.x * 3 + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(i + j) & 7] = __builtin_fmaf(v[(i + j) & 7], 1.0001f, 0.5f);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
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
void same(int blocks) {
  float* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 8000;
  float ms = timeit([&] { hipLaunchKernelGGL(k_same<NV>, dim3(blocks), dim3(256), 0, 0, out, iters); });
  const double flop = (double)blocks * 4 * iters * 8 * 16384.0;
  printf("blocks %d: %d VALU per bf16 MFMA: %.3f ms, %.0f TFLOP/s, %.1f ns per MFMA per wave\n", blocks, NV, ms, flop / ms / 1e9,
         ms * 1e6 / (iters * 8.0));
  (void)hipFree(out);
}
int main() {
  same<0>(256); same<1>(256); same<2>(256); same<3>(256); same<4>(256); same<6>(256); same<8>(256);
  same<0>(512); same<2>(512); same<4>(512);
  return 0;
}
