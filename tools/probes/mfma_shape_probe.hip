// Stand-alone probe: does the MFMA shape matter under the power limit?  Register-only loops of v_mfma_f32_16x16x32_bf16
// (32 accumulators of 4 VGPRs) and v_mfma_f32_32x32x16_bf16 (8 accumulators of 16 VGPRs) - the same 128x64x32 wave-tile
// step either way - on pseudo-random bf16 operands (the clock the chip holds depends on the data) and on zeros.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape tools/probes/mfma_shape_probe.hip && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ short rnd_bf16(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    // exponent around 2^-2 .. 2^1, random sign and mantissa
    return (short)(((s >> 16) & 0x807f) | (0x3e80 + ((s >> 8) & 0x180)));
}

template <int SHAPE>
__global__ __launch_bounds__(512) void probe(int iters, int zero, float* sink) {
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 17u;
    bf16x8 a[8], b[4];                       // one k-step of a 128x64 wave tile: 8 A fragments, 4 B fragments
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) a[i][j] = zero ? 0 : rnd_bf16(seed);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[i][j] = zero ? 0 : rnd_bf16(seed);
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[8][4];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)       // 32x32x16: two k-steps of 16 cover the same K = 32
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[2 * j + ks], a[2 * i + ks], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    if (s == 123.456f) sink[0] = s;
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    float* sink; (void)hipMalloc(&sink, 4);
    for (int rep = 0; rep < 2; ++rep)
        for (int zero = 0; zero < 2; ++zero)
            for (int shape : {16, 32}) {
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0);
                if (shape == 16) hipLaunchKernelGGL(probe<16>, dim3(cus), dim3(512), 0, 0, iters, zero, sink);
                else hipLaunchKernelGGL(probe<32>, dim3(cus), dim3(512), 0, 0, iters, zero, sink);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                double flops = 2.0 * 128 * 64 * 32 * (double)iters * 8 * cus;
                printf("mfma %dx%d  %s operands: %.3f ms  %.0f TFLOP/s\n", shape, shape, zero ? "zero  " : "random", ms, flops / ms / 1e9);
            }
    return 0;
}
