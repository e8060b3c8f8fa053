// Placeholder translation unit for a16/a17 (CBS + segment z); replaced by the real kernels.
#include "wcx_common.h"

extern "C" {

int wcx_cbs(wcx_ctx *, const double *, const double *, const int64_t *, int, double, int64_t,
            uint64_t, double *, int, int *) {
  wcx_set_error("wcx_cbs: not implemented yet");
  return WCX_ERR_UNSUPPORTED;
}

int wcx_segment_z(wcx_ctx *, const double *, const double *, const double *, int,
                  const int64_t *, int, const double *, int, double *) {
  wcx_set_error("wcx_segment_z: not implemented yet");
  return WCX_ERR_UNSUPPORTED;
}

}  // extern "C"
