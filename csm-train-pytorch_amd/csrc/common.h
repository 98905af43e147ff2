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

// torchtune's interleaved-pair rotation of one pair, (x0 c - x1 s, x1 c + x0 s), with every product rounded before the add:
// the reference evaluates the expression op by op in fp32 (no fused multiply-add), and left to the compiler two inlined copies
// of the same expression were contracted differently (round 4: one-ulp differences at bf16 ties between the 256 x 256 and the
// 256 x 192 tile kernels, and between the decode kernels and the prefill path).  EVERY rotation in the library - GEMM
// epilogues, csm_rope, the decode kernels, the attention backward's transposed rotation (s -> -s), Mimi - goes through this.
__device__ __forceinline__ void rope_rot(float& x0, float& x1, float c, float s) {
#pragma clang fp contract(off)
    const float a0 = x0 * c - x1 * s, a1 = x1 * c + x0 * s;
    x0 = a0; x1 = a1;
}

// Cross-lane butterflies without the LDS crossbar (round 4).  __shfl_xor compiles to ds_bpermute_b32: an LDS-pipe round trip of
// ~100+ cycles per step, six dependent ones per wave reduction - most of the in-kernel time of the one-row decode kernels
// (tools/probes/decode_stamps.py) and the tail of every norm / loss kernel.  gfx950 has VALU forms for every step:
//   xor 32 / xor 16: v_permlane32_swap / v_permlane16_swap with vdst = src = x leave {lower, upper} pairs: own ^ pair[0] ^ pair[1]
//                    is the partner's value (exact for any x);
//   xor 8:           DPP row_ror:8 (lane (i + 8) mod 16 of the row = i ^ 8: exact);
//   xor 2, xor 1:    DPP quad_perm [2,3,0,1], [1,0,3,2] (exact);
//   xor 4:           DPP row_shl:4 into banks 0, 2 and row_shr:4 into banks 1, 3 (exact; two moves).
// lane_xor<O>(x) therefore returns exactly __shfl_xor(x, O, 64): same operands into the same additions (a + b == b + a bit for
// bit), so every reduction keeps its bits.
__device__ __forceinline__ uint32_t lane_xor_u32_32(uint32_t x) { auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false); return r[0] ^ r[1] ^ x; }
__device__ __forceinline__ uint32_t lane_xor_u32_16(uint32_t x) { auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false); return r[0] ^ r[1] ^ x; }
template <int O>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t x) {
    static_assert(O == 32 || O == 16 || O == 8 || O == 4 || O == 2 || O == 1, "lane_xor: power of two below 64");
    if constexpr (O == 32) return lane_xor_u32_32(x);
    else if constexpr (O == 16) return lane_xor_u32_16(x);
    else if constexpr (O == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false);        // row_ror:8
    else if constexpr (O == 4) {
        int r = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xf, 0x5, false);                                          // row_shl:4 -> lanes 0-3, 8-11
        r = __builtin_amdgcn_update_dpp(r, (int)x, 0x114, 0xf, 0xa, false);                                              // row_shr:4 -> lanes 4-7, 12-15
        return (uint32_t)r;
    } else if constexpr (O == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4e, 0xf, 0xf, false);        // quad_perm [2,3,0,1]
    else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xb1, 0xf, 0xf, false);                                 // quad_perm [1,0,3,2]
}
template <int O> __device__ __forceinline__ float lane_xor(float x) { return __uint_as_float(lane_xor_u32<O>(__float_as_uint(x))); }
template <int O> __device__ __forceinline__ int lane_xor(int x) { return (int)lane_xor_u32<O>((uint32_t)x); }

__device__ __forceinline__ float wave_sum(float v) {       // the butterfly of `for (o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o)`
    v += lane_xor<32>(v); v += lane_xor<16>(v); v += lane_xor<8>(v); v += lane_xor<4>(v); v += lane_xor<2>(v); v += lane_xor<1>(v);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, lane_xor<32>(v)); v = fmaxf(v, lane_xor<16>(v)); v = fmaxf(v, lane_xor<8>(v));
    v = fmaxf(v, lane_xor<4>(v)); v = fmaxf(v, lane_xor<2>(v)); v = fmaxf(v, lane_xor<1>(v));
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
