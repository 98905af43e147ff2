// Error plumbing and library identity of libcsm_hip.so (C ABI declared in include/csm_hip.h).
// No exception crosses the boundary: every entry point returns 0 on success, a non-zero code otherwise, and
// csm_last_error() gives the text for the calling thread.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void csm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* csm_last_error(void) { return g_err; }

// 2: csm_attn_bwd's scratch doubled (csm_attn_bwd_workspace_bytes), negative clip coefficient = skipped AdamW step
// 3: added entry points only (csm_attn_last_dkv_kernel, ...): every ABI-2 call keeps its meaning
extern "C" int csm_abi_version(void) { return 3; }
extern "C" long long csm_attn_bwd_workspace_bytes(int B, int S, int H) { return 2LL * B * H * S * (long long)sizeof(float); }

// 0 when a gfx950 device is visible to this process, otherwise an error code with text.
extern "C" int csm_device_check(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        csm_set_error("csm_device_check: no HIP device (%s)", hipGetErrorString(e));
        return 3;
    }
    if (device < 0 || device >= n) {
        csm_set_error("csm_device_check: device %d out of range (have %d)", device, n);
        return 1;
    }
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) {
        csm_set_error("csm_device_check: hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return 3;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        csm_set_error("csm_device_check: device %d is %s, this library is built for gfx950 only", device, p.gcnArchName);
        return 4;
    }
    return 0;
}
