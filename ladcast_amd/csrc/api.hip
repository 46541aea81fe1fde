#include "common.h"

extern "C" int ldc_abi_version(void) { return LDC_ABI_VERSION; }
extern "C" const char* ldc_build_arch(void) { return "gfx950"; }
extern "C" int ldc_sizeof_qkv_epilogue(void) { return static_cast<int>(sizeof(ldc_qkv_epilogue)); }
