// Library-level entry points of libomnipq_pointops.so (see include/omnipq_pointops.h).
#include "common.h"

extern "C" int omnipq_abi_version(void) { return OMNIPQ_ABI_VERSION; }

extern "C" const char *omnipq_error_string(int code) {
  switch (code) {
    case OMNIPQ_OK: return "ok";
    case OMNIPQ_EINVAL: return "omnipq: invalid argument (shape or null pointer)";
    case OMNIPQ_ETOOLARGE: return "omnipq: problem too large for one launch";
    case OMNIPQ_ETIMEOUT: return "omnipq: in-kernel hand-off timed out";
    default: return hipGetErrorString((hipError_t)code);
  }
}
