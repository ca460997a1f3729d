// Library identification entry points of the C ABI (include/et_hip.h).
#include "et_device.h"
#include "../../include/et_hip.h"

extern "C" const char* et_build_arch(void) { return "gfx950"; }
extern "C" int et_abi_version(void) { return ET_ABI_VERSION; }
