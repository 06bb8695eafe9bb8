// Does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs on gfx950, and what is its issue rate next to the fp32
// MFMA?  (Decides how the split-fp16 sparse convolution stores the low halves of its operands.)
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_f16_denorm.hip -o scripts/micro/build/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_denorm(float a_val, float b_val, float* out) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}

// layout probe: A[i][k] = (i == 3 && k == kk), B[k][j] = (k == kk) * (j + 1)  ->  D[3][j] = j + 1
__global__ void k_layout(int kk, float* out) {
  const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
  h8 a, b;
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * g + e;
    a[e] = (_Float16)((j == 3 && k == kk) ? 1.f : 0.f);   // lane (g, i = j) holds A[i][8g + e]
    b[e] = (_Float16)((k == kk) ? (float)(j + 1) : 0.f);  // lane (g, j) holds B[8g + e][j]
  }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + j] = c[r];   // D[4g + r][j]
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)1.0f; }
  const float af = threadIdx.x * 0.001f, bf = 1.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* d; hipMalloc(&d, 1 << 24);
  float h[256];
  const float vals[] = {1.0f, 6.1035e-5f /*min normal*/, 3.0518e-5f /*2^-15 subnormal*/, 9.5367e-7f /*2^-20*/, 5.9605e-8f /*2^-24 smallest*/};
  for (float v : vals) {
    hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, v, 1.0f, d);
    hipMemcpy(h, d, 4, hipMemcpyDeviceToHost);
    printf("a = %.6e (fp16) x b = 1, K = 32: D = %.6e  expected %.6e  %s\n", v, h[0], 32.0 * (double)(float)(_Float16)v,
           fabs(h[0] - 32.0 * (double)(float)(_Float16)v) < 1e-12 ? "exact" : "DIFFERS (flushed?)");
  }
  // subnormal x subnormal-ish product landing in fp32 range
  hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, 9.5367e-7f, 9.5367e-7f, d);
  hipMemcpy(h, d, 4, hipMemcpyDeviceToHost);
  printf("2^-20 x 2^-20 x 32 = %.6e (expected %.6e)\n", h[0], 32.0 * pow(2.0, -40));
  int bad = 0;
  for (int kk = 0; kk < 32; kk += 5) {
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, kk, d);
    hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) bad += h[i * 16 + j] != (i == 3 ? (float)(j + 1) : 0.f);
  }
  printf("layout probe (A lane(g,i)=A[i][8g+e], B lane(g,j)=B[8g+e][j], D lane(g,j)[r]=D[4g+r][j]): %s\n", bad ? "WRONG" : "ok");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096, blocks = 256 * 8;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
      else hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)blocks * 4 * iters * 8 * (mode == 0 ? 16.0 * 16 * 32 * 2 : 16.0 * 16 * 4 * 2);
      if (rep) printf("%s: %.3f ms, %.1f TFLOP/s\n", mode == 0 ? "mfma_f32_16x16x32_f16" : "mfma_f32_16x16x4_f32", ms, flop / ms / 1e9);
    }
  }
  return 0;
}
