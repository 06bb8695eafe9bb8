// FETCH_SIZE calibration for the two gather patterns of the sparse convolutions (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern").  Every kernel reads N_ROWS distinct 256-byte rows of a
// 1 GB buffer exactly once (no reuse, far beyond the 256 MB Infinity Cache), so the true HBM read volume is known.
//   A  streaming: lane reads 16 contiguous bytes, a wave 1 KB contiguous                     (the guide's x2 case)
//   B  wave-private kernel: lane (g, j) reads 16 B at row[j] * 256 + q * 64 + g * 16, q = 0..3  (4 lanes cover 64 contiguous B)
//   C  row-stationary kernel: lane (g, j) reads 16 B at row[j] * 256 + qb * 128 + g * 32 (+ 16)  (4 lanes, 32-byte stride)
// rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib   ->  compare the counter with bytes_read printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void kA(const uint4* __restrict__ buf, size_t n16, unsigned int* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int acc = 0;
  for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = buf[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
template <int MODE>
__global__ void kBC(const char* __restrict__ buf, const int* __restrict__ rows, int n_rows, unsigned int* sink) {
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  unsigned int acc = 0;
  for (int base = wave * 16; base < n_rows; base += n_waves * 16) {
    const size_t r = (size_t)rows[base + j] * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t off = MODE == 0 ? r + q * 64 + g * 16 : r + (q >> 1) * 128 + g * 32 + (q & 1) * 16;
      const uint4 v = *reinterpret_cast<const uint4*>(buf + off);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) *sink = acc;
}
int main() {
  const size_t bytes = 1ull << 30; const int n_rows = (int)(bytes / 256);
  char* buf; int* rows; unsigned int* sink;
  (void)hipMalloc(&buf, bytes); (void)hipMalloc(&rows, n_rows * 4); (void)hipMalloc(&sink, 4);
  (void)hipMemset(buf, 1, bytes);
  int* h = (int*)malloc(n_rows * 4);
  for (int i = 0; i < n_rows; ++i) h[i] = i;
  unsigned long long s = 88172645463325252ull;           // random permutation: every row once, in random order
  for (int i = n_rows - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; int k = (int)(s % (unsigned)(i + 1)); int t = h[i]; h[i] = h[k]; h[k] = t; }
  (void)hipMemcpy(rows, h, n_rows * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(kA, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
  hipLaunchKernelGGL((kBC<0>), dim3(4096), dim3(256), 0, 0, buf, rows, n_rows, sink);
  hipLaunchKernelGGL((kBC<1>), dim3(4096), dim3(256), 0, 0, buf, rows, n_rows, sink);
  (void)hipDeviceSynchronize();
  printf("bytes_read per kernel = %zu (+ %d for the row list in B, C)\n", bytes, n_rows * 4);
  return 0;
}
