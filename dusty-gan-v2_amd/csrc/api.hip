// ABI bookkeeping for libdgv2.so.
#include "common.h"

extern "C" int dgv2_abi_version(void) { return 47; }
