// Stable key sort of row indices (rocPRIM radix sort), used by the coordinate maps to order the row tiles of a
// transposed convolution by neighbour pattern.  Kept in its own translation unit: rocPRIM's templates are the
// slowest thing in the build.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace eyoc {

size_t sort_rows_tmp_bytes(int n, int bits) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned int*)nullptr, (unsigned int*)nullptr, (const int*)nullptr,
                                  (int*)nullptr, (size_t)(n > 0 ? n : 1), 0u, (unsigned)bits, (hipStream_t)0);
  return bytes;
}

// vals_out = vals_in reordered by ascending key; equal keys keep their input order
int sort_rows_by_key(void* tmp, size_t tmp_bytes, const unsigned int* keys_in, unsigned int* keys_out, const int* vals_in,
                     int* vals_out, int n, int bits, hipStream_t st) {
  size_t need = tmp_bytes;
  EYOC_CHECK_HIP(rocprim::radix_sort_pairs(tmp, need, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)bits, st));
  return EYOC_OK;
}

size_t sort_rows64_tmp_bytes(int n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (const int*)nullptr,
                                  (int*)nullptr, (size_t)(n > 0 ? n : 1), 0u, 64u, (hipStream_t)0);
  return bytes;
}

// the same with 64-bit keys (Z-order of the coordinates)
int sort_rows_by_key64(void* tmp, size_t tmp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out, const int* vals_in,
                       int* vals_out, int n, int bits, hipStream_t st) {
  size_t need = tmp_bytes;                                             // sized for 64 bits: enough for any narrower sort
  EYOC_CHECK_HIP(rocprim::radix_sort_pairs(tmp, need, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)bits, st));
  return EYOC_OK;
}

}  // namespace eyoc
