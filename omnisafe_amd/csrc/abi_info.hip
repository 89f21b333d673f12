// Build identity of libomnisafe_amd.so: omnisafe_amd/build.py passes the sha256 of the header and of every
// kernel source as OSA_ABI_DIGEST; omnisafe_amd/_lib.py compares it with the digest of the sources it
// sits next to before binding any prototype (ctypes cannot check argument lists).
#include "../../include/omnisafe_amd.h"

#ifndef OSA_ABI_DIGEST
#define OSA_ABI_DIGEST "unknown"
#endif

extern "C" const char* osa_abi_digest(void) { return OSA_ABI_DIGEST; }
