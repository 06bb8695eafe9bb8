// Context, error reporting.
#include "common.h"

namespace eyoc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace eyoc

int eyoc_ctx::ensure_scratch(size_t bytes, hipStream_t st) {
  if (scratch_owned && st != scratch_owner) {
    if (!scratch_ev) EYOC_CHECK_HIP(hipEventCreateWithFlags(&scratch_ev, hipEventDisableTiming));
    EYOC_CHECK_HIP(hipEventRecord(scratch_ev, scratch_owner));
    EYOC_CHECK_HIP(hipStreamWaitEvent(st, scratch_ev, 0));
  }
  scratch_owner = st;
  scratch_owned = true;
  if (bytes <= scratch_bytes) return EYOC_OK;
  size_t want = eyoc::align_up(bytes, 1 << 20);
  if (scratch) EYOC_CHECK_HIP(hipFree(scratch));
  scratch = nullptr;
  scratch_bytes = 0;
  EYOC_CHECK_HIP(hipMalloc(&scratch, want));
  scratch_bytes = want;
  return EYOC_OK;
}

int eyoc_ctx::ensure_pool() {
  if (pool_ready) return EYOC_OK;
  EYOC_CHECK_HIP(hipEventCreateWithFlags(&pool_fork, hipEventDisableTiming));
  for (int i = 0; i < POOL; ++i) {
    EYOC_CHECK_HIP(hipStreamCreateWithFlags(&pool[i], hipStreamNonBlocking));
    EYOC_CHECK_HIP(hipEventCreateWithFlags(&pool_done[i], hipEventDisableTiming));
  }
  pool_ready = true;
  return EYOC_OK;
}

extern "C" {

int eyoc_version(void) { return EYOC_VERSION; }

const char* eyoc_last_error(void) { return eyoc::g_err; }

int eyoc_create(int device, eyoc_ctx** out) {
  EYOC_REQUIRE(out != nullptr, EYOC_ERR_INVALID, "eyoc_create: out is NULL");
  int count = 0;
  EYOC_CHECK_HIP(hipGetDeviceCount(&count));
  EYOC_REQUIRE(device >= 0 && device < count, EYOC_ERR_INVALID, "eyoc_create: device %d of %d", device, count);
  EYOC_CHECK_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  EYOC_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  EYOC_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, EYOC_ERR_INVALID,
               "eyoc_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  eyoc_ctx* ctx = new eyoc_ctx();
  ctx->device = device;
  ctx->pinned_bytes = 4096;
  hipError_t e = hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    delete ctx;
    eyoc::set_error("hipHostMalloc failed: %s", hipGetErrorString(e));
    return EYOC_ERR_HIP;
  }
  *out = ctx;
  return EYOC_OK;
}

int eyoc_destroy(eyoc_ctx* ctx) {
  if (!ctx) return EYOC_OK;
  if (ctx->pool_ready) {
    for (int i = 0; i < eyoc_ctx::POOL; ++i) {
      (void)hipStreamDestroy(ctx->pool[i]);
      (void)hipEventDestroy(ctx->pool_done[i]);
    }
    (void)hipEventDestroy(ctx->pool_fork);
  }
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->scratch_ev) (void)hipEventDestroy(ctx->scratch_ev);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  delete ctx;
  return EYOC_OK;
}

}  // extern "C"
