// Calibration: issue rate of scalar vs packed fp32 VALU ops (one wave per SIMD and four).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{threadIdx.x * 0.01f + i, threadIdx.x * 0.02f + i};
  const f32x2 c1 = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) v[i].x = __builtin_fmaf(v[i].x, c1.x, c2.x);                       // v_fma_f32
        if (MODE == 1) v[i] = __builtin_elementwise_fma(v[i], c1, c2);                      // v_pk_fma_f32
        if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c2));
        if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
        if (MODE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(c2.x));
      }
  }
  if (MODE == 5) {                                          // v_fma_f64 (RANSAC's k_count: 15-16 of them per residual)
    double d[8];
    for (int i = 0; i < 8; ++i) d[i] = threadIdx.x * 0.01 + i;
    const double a = 1.0001, b = 0.5;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(a), "v"(b));
    for (int i = 0; i < 8; ++i) v[i].x += (float)d[i];
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int blocks) {
  float* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)iters * 32;   // per wave
  printf("%-14s blocks %4d: %.3f ms  %.2f ns per wave-instruction  (%.1f G wave-inst/s chip)\n", name, blocks, ms, ms * 1e6 / insts,
         insts * blocks * 4 / ms / 1e6);
  (void)hipFree(out);
}
int main() {
  for (int blocks : {256, 1024}) {
    run<0>("v_fma_f32", blocks); run<1>("v_pk_fma_f32", blocks); run<2>("v_pk_add_f32", blocks); run<3>("v_pk_mul_f32", blocks);
    run<4>("v_add_f32", blocks); run<5>("v_fma_f64", blocks);
  }
  return 0;
}
