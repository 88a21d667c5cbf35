// Version / probing entry points of libsfron.so.
#include "common.h"
#include "../../include/sfron.h"
extern "C" {
int sfron_abi_version(void) { return 12; }
const char* sfron_build_arch(void) { return "gfx950"; }
}
