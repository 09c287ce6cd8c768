// ABI identification for libisic_hip.so.
#include "common.h"

extern "C" {
int isic_abi_version(void) { return 1; }
const char* isic_target_arch(void) { return "gfx950"; }
}
