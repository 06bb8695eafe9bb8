// SC2-PCR registration (Matcher.SC2_PCR, scripts/SC2_PCR/SC2_PCR.py:307-384) - placeholder until the
// fused N^2 kernels land; the entry points exist so the ABI is complete and fail loudly.
#include "common.h"

extern "C" {

size_t eyoc_sc2pcr_workspace_bytes(int n, const eyoc_sc2pcr_params* params) {
  (void)n; (void)params;
  return 0;
}

int eyoc_sc2pcr(eyoc_ctx* ctx, const float* src_dev, const float* tgt_dev, int n, const eyoc_sc2pcr_params* params,
                float* T_dev, float* fitness_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  (void)ctx; (void)src_dev; (void)tgt_dev; (void)n; (void)params; (void)T_dev; (void)fitness_dev;
  (void)workspace_dev; (void)workspace_bytes; (void)stream;
  eyoc::set_error("eyoc_sc2pcr: not implemented in this build");
  return EYOC_ERR_INVALID;
}

}  // extern "C"
