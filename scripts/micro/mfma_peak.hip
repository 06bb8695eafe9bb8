// Calibration: achievable v_mfma_f32_16x16x4_f32 rate on this GPU (no memory traffic).
// hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int threads, const char* name) {
  float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * (threads / 64) * iters * 4 * NACC * 2048.0;
  printf("%s blocks=%d threads=%d nacc=%d: %.3f ms %.1f TFLOP/s\n", name, blocks, threads, NACC, ms, flop / ms / 1e9);
  hipFree(out);
}
int main() {
  run<8>(256, 256, "1 wave/SIMD");
  run<8>(512, 256, "2 waves/SIMD");
  run<8>(1024, 256, "4 waves/SIMD");
  run<2>(512, 256, "2 waves/SIMD");
  run<1>(1024, 256, "4 waves/SIMD dependent");
  run<16>(2048, 256, "many blocks");
  return 0;
}
