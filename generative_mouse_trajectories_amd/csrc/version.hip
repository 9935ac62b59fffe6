#include "gmt_common.h"
extern "C" int gmt_version(void) { return GMT_ABI_VERSION; }
