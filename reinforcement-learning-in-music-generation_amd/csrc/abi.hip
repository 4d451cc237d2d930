// ABI version + small host-side helpers of libcwlt.so (see include/cwlt.h).
#include "cwlt_common.h"

extern "C" int cwlt_abi_version(void) { return 20; }
