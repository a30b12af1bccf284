// Library-wide state of libptamd (none beyond the last HIP error of the calling thread).
#include <hip/hip_runtime.h>

#include "../../include/ptamd.h"

thread_local hipError_t g_pt_last_hip_error = hipSuccess;

extern "C" const char *ptamd_version(void) { return "ptamd 0.1 (gfx950)"; }
extern "C" const char *ptamd_last_hip_error(void) { return hipGetErrorString(g_pt_last_hip_error); }
