// lane_xor<O> (common.h: VALU butterflies) against __shfl_xor for every O and random data; wave_sum / wave_max against the
// ds_bpermute loops they replace.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icsm-train-pytorch_amd/csrc -Iinclude tools/probes/lane_xor_probe.hip -o /tmp/lxp && /tmp/lxp
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const uint32_t* in, uint32_t* out) {
    const uint32_t x = in[threadIdx.x];
    uint32_t* o = out + threadIdx.x * 16;
    o[0] = lane_xor_u32<32>(x) ^ (uint32_t)__shfl_xor((int)x, 32, 64);
    o[1] = lane_xor_u32<16>(x) ^ (uint32_t)__shfl_xor((int)x, 16, 64);
    o[2] = lane_xor_u32<8>(x) ^ (uint32_t)__shfl_xor((int)x, 8, 64);
    o[3] = lane_xor_u32<4>(x) ^ (uint32_t)__shfl_xor((int)x, 4, 64);
    o[4] = lane_xor_u32<2>(x) ^ (uint32_t)__shfl_xor((int)x, 2, 64);
    o[5] = lane_xor_u32<1>(x) ^ (uint32_t)__shfl_xor((int)x, 1, 64);
    float v = __uint_as_float((x & 0x007fffffu) | 0x3f800000u) - 1.5f, a = v, b = v;
    for (int q = 32; q > 0; q >>= 1) a += __shfl_xor(a, q, 64);
    for (int q = 32; q > 0; q >>= 1) b = fmaxf(b, __shfl_xor(b, q, 64));
    o[6] = __float_as_uint(a) ^ __float_as_uint(wave_sum(v));
    o[7] = __float_as_uint(b) ^ __float_as_uint(wave_max(v));
}
int main() {
    std::vector<uint32_t> h(256), r(256 * 16);
    uint32_t *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 256 * 64);
    int bad = 0;
    for (int it = 0; it < 50; ++it) {
        for (auto& x : h) x = (uint32_t)rand() * 2654435761u + (uint32_t)rand();
        hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o);
        hipMemcpy(r.data(), o, 256 * 64, hipMemcpyDeviceToHost);
        for (int t = 0; t < 256; ++t) for (int c = 0; c < 8; ++c) if (r[t * 16 + c]) { if (bad < 10) printf("mismatch it %d thread %d check %d\n", it, t, c); ++bad; }
    }
    printf("lane_xor probe: %d mismatches\n", bad);
    return bad != 0;
}
