// placeholder until the backward pair kernel lands
#include <hip/hip_runtime.h>
#include "enf_layout.h"
extern "C" int enf_launch_pair_bwd(const EnfDims&, const EnfLayout&, const char*, const float*, long long, const float*,
                                   const float*, const float*, const float*, float*, hipStream_t) {
  return ENF_EUNSUPPORTED;
}
