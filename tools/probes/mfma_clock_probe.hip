// Stand-alone probe: what does the MFMA pipe deliver when nothing else is in the way, and at what shader clock?
// Every CU runs W waves/SIMD of back-to-back v_mfma_f32_16x16x32_bf16 on register operands for ITERS iterations;
// per-workgroup s_memtime (shader clock) and s_memrealtime (100 MHz) deltas give the sustained clock under load.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probes/mfma_clock_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(512) void probe(int iters, unsigned long long* out, float* sink) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("device %s  CUs %d  clockRate %d kHz\n", p.name, cus, p.clockRate);
    unsigned long long* d; float* sink;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * cus * 4); hipMalloc(&sink, 4);
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe<16>, dim3(cus), dim3(threads), 0, 0, iters, d, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(2 * cus);
            hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * cus, hipMemcpyDeviceToHost);
            double cyc = 0, rt = 0; for (int i = 0; i < cus; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
            cyc /= cus; rt /= cus;
            double flops = 2.0 * 16 * 16 * 32 * 16.0 * iters * (threads / 64) * cus;
            printf("threads/WG %d  %.3f ms  %.1f TFLOP/s  memtime cycles %.0f  realtime ticks %.0f (100 MHz) -> memtime clock %.0f MHz; "
                   "MFMA cycles/instr at that clock %.2f\n", threads, ms, flops / ms / 1e9, cyc, rt, cyc / rt * 100.0,
                   cyc / (16.0 * iters * (threads / 256)));
        }
    }
    return 0;
}
