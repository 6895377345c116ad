// Status strings / version of the C ABI (include/iiseg.h).
#include "common.h"

extern "C" const char* iiseg_strerror(int status) {
    switch (status) {
        case IISEG_OK: return "ok";
        case IISEG_ERR_NULL: return "required pointer is NULL";
        case IISEG_ERR_SHAPE: return "inconsistent or unsupported shape";
        case IISEG_ERR_ALIGN: return "pointer is not 16-byte aligned";
        case IISEG_ERR_LAUNCH: return "kernel launch failed";
        case IISEG_ERR_UNSUPPORTED: return "no kernel variant implements this request";
        default: return "unknown iiseg status";
    }
}

extern "C" const char* iiseg_last_hip_error(void) {
    return hipGetErrorString((hipError_t)iiseg_hip_error_slot());
}

extern "C" int iiseg_abi_version(void) { return 29; }
extern "C" const char* iiseg_target_arch(void) { return "gfx950"; }
