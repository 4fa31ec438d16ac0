// Full-catalog scoring (fp32 MFMA) -- placeholder until the MFMA kernel lands.
#include "rsa_common.hpp"

extern "C" int64_t rsa_fullscore_workspace_bytes(int64_t n_query, int64_t n_items, int32_t k) {
  (void)n_query; (void)n_items; (void)k;
  return 0;
}

extern "C" int rsa_fullscore(const float*, int64_t, int32_t, const float*, int64_t, float*, float*, float*, int64_t*,
                             int32_t, void*, int64_t, rsa_stream_t) {
  rsa::set_error("rsa_fullscore: not implemented in this build");
  return RSA_ERR_UNSUPPORTED;
}
