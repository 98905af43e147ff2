// Which XCD / CU does each workgroup of a launch land on?  Used by dual_half_probe.py to check what a CU-masked stream
// (hipExtStreamCreateWithCUMask) really restricts.  out[2*b] = XCC id, out[2*b+1] = HW_ID register of block b.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void census_kernel(int* out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15u;     // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);            // HW_REG_HW_ID
        out[2 * blockIdx.x] = (int)xcc;
        out[2 * blockIdx.x + 1] = (int)hw;
    }
    // keep the block resident for a while so that the grid spreads over every CU the queue may use
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)spin) {}
}
extern "C" int census(void* stream, int* out, int nblocks, int spin) {
    hipLaunchKernelGGL(census_kernel, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, out, spin);
    return (int)hipGetLastError();
}
