// C-ABI odds and ends: version and error strings.
#include "lsq_common.h"

extern "C" int lsq_abi_version(void) { return LSQ_ABI_VERSION; }

extern "C" const char* lsq_error_string(int code) {
  switch (code) {
    case LSQ_OK: return "ok";
    case LSQ_E_NULL: return "required pointer is NULL";
    case LSQ_E_SHAPE: return "non-positive or inconsistent dimension";
    case LSQ_E_SCHEME: return "unknown scheme or plane count";
    case LSQ_E_TOO_LONG: return "sub-sampled row has 2^22 or more elements";
    case LSQ_E_WORKSPACE: return "workspace too small";
    case LSQ_E_UNSUPPORTED: return "unsupported configuration";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}
