// Shared device helpers for the CSM gfx950 (MI355X / CDNA4) kernels.
// wave = 64 lanes everywhere; bf16 is carried as raw uint16_t storage.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA 16x16x32 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // one MFMA 16x16 accumulator

#define CSM_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even; a plain conversion keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&b);
}

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// The same two conversions as ONE v_cvt_pk_bf16_f32 (the scalar form costs a conversion per value plus a shift and an or).
// Identical values; kept separate because swapping it into every kernel shifts hipcc's fma-contraction choices around the
// call sites, and the decode path and its prefix-recompute check (tests/test_configs_gpu.py) compare different kernels bit
// for bit.  Used where the instruction count is exposed: the four-wave GEMM's epilogues.
typedef __attribute__((ext_vector_type(2))) float csm_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 csm_bf16x2;
__device__ __forceinline__ uint32_t pack2bf_pk(float lo, float hi) {
    const csm_f32x2 f = {lo, hi};
    csm_bf16x2 b = __builtin_convertvector(f, csm_bf16x2);
    return *reinterpret_cast<uint32_t*>(&b);
}

// sigmoid / SiLU with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (div_scale, rcp, 4 fma, div_fmas,
// div_fixup: 11 instructions).  They sit in GEMM epilogues where a lane evaluates 64-128 of them per tile and no other
// workgroup on the CU overlaps the epilogue; every SwiGLU site (GEMM epilogues, stand-alone kernels, decode) uses these two.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float silu(float x) { return x * fast_sigmoid(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blocks of up to 16 waves; `red` is >= 16 floats of LDS. All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// 16-byte vector of 8 bf16 as raw words (a native vector type: stays in VGPRs, never in scratch)
typedef __attribute__((ext_vector_type(4))) uint32_t U4;

__device__ __forceinline__ void unpack8(const U4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

__device__ __forceinline__ U4 pack8(const float* f) {
    U4 u = {pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
    return u;
}

__device__ __forceinline__ U4 pack8_pk(const float* f) {
    U4 u = {pack2bf_pk(f[0], f[1]), pack2bf_pk(f[2], f[3]), pack2bf_pk(f[4], f[5]), pack2bf_pk(f[6], f[7])};
    return u;
}

// transposed LDS read: 4 rows x 16 cols of b16 per 16-lane group, lane i receives column i (gfx950)
__device__ __forceinline__ bf16x4 lds_read_tr16(const void* lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(lds_addr));
}

__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) {
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

// host-side error plumbing (csm_api.cpp)
extern "C" void csm_set_error(const char* fmt, ...);
#define CSM_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            csm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return 2;                                                                 \
        }                                                                             \
    } while (0)
#define CSM_REQUIRE(cond, ...)                                                        \
    do {                                                                              \
        if (!(cond)) {                                                                \
            csm_set_error(__VA_ARGS__);                                               \
            return 1;                                                                 \
        }                                                                             \
    } while (0)
