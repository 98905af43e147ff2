// Shared pieces of the bf16 MFMA GEMM kernels (tile images, staging, fragment loads).  See gemm.hip for the scheme.
#pragma once
#include "common.h"

namespace {

// XOR key of the k-strided ("ks") image: a 32-lane half of a transposed read touches k-rows {8g+q} and {8g+8+q}
// (q = 0..3), so the key must separate rows that differ in bit 3 as well as in bits 0-1.
__device__ __forceinline__ int ks_swz(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; void* C; const bf16_t* R;
    int M, N, K, lda, ldb, ldc, ldr;
    long long sA, sB, sC, sR;  // batch strides in elements
    float alpha;
    int tiles_m, tiles_n;
    int epi_mode; const bf16_t* aux_in; bf16_t* aux_out; int ld_aux;
    int epi_p0, epi_p1;   // EPI_ROPE: columns [0, p0) are rotated, head_dim p1
    const bf16_t* xA; const bf16_t* xB; int kx;   // K-extension (see k_extend): C += xA[M][kx] . xB[N][kx]^T, kx % 32 == 0, 0 = none
};

// ---- staging: each thread moves 4 x 16 B per operand per K-tile --------------------------------
template <int T>
__device__ __forceinline__ void stage_load(const bf16_t* __restrict__ P, int ld, int rows /*M or N*/, int K,
                                           int row0, int k0, U4 (&reg)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = t + 256 * i;
        U4 v = {0u, 0u, 0u, 0u};
        if (T == 0) {
            const int r = idx >> 3, c = idx & 7;
            int gr = row0 + r;
            gr = gr < rows ? gr : rows - 1;                 // clamp: garbage rows are masked at the store
            const int gk = k0 + c * 8;
            if (gk < K) v = *reinterpret_cast<const U4*>(P + (size_t)gr * ld + gk);
        } else {
            const int kr = idx >> 4, c = idx & 15;
            const int gk = k0 + kr, gc = row0 + c * 8;
            if (gk < K && gc < rows) v = *reinterpret_cast<const U4*>(P + (size_t)gk * ld + gc);
        }
        reg[i] = v;
    }
}

template <int T>
__device__ __forceinline__ void stage_store(char* lds, const U4 (&reg)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = t + 256 * i;
        int off;
        if (T == 0) {
            const int r = idx >> 3, c = idx & 7;
            off = r * 128 + ((c ^ (r & 7)) << 4);
        } else {
            const int kr = idx >> 4, c = idx & 15;
            off = kr * 256 + ((((c >> 1) ^ ks_swz(kr)) << 5) | ((c & 1) << 4));
        }
        *reinterpret_cast<U4*>(lds + off) = reg[i];
    }
}

// direct global->LDS staging (LDS-DMA, 16 B per lane, 1 KiB per wave-instruction).  The LDS destination is
// wave-uniform base + lane*16, so the XOR swizzle is applied to the per-lane SOURCE address and the image is the
// same one stage_store writes.  No zero fill: callers guarantee K % 64 == 0; row/column overhang is clamped.
template <int T>
__device__ __forceinline__ void stage_glds(const bf16_t* __restrict__ P, int ld, int rows, int row0, int k0, char* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = wave * 4 + i;  // 1-KiB piece of the 16-KiB tile
        const bf16_t* src;
        if (T == 0) {
            const int r = 8 * j + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            int gr = row0 + r;
            gr = gr < rows ? gr : rows - 1;
            src = P + (size_t)gr * ld + k0 + c * 8;
        } else {
            const int kr = 4 * j + (lane >> 4);
            const int u = lane & 15;
            const int sl = (u >> 1) ^ ks_swz(kr);
            int gc = row0 + sl * 16 + (u & 1) * 8;
            gc = gc < rows ? gc : rows - 8;
            src = P + (size_t)(k0 + kr) * ld + gc;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, 0, 0);
    }
}

// The same tile with the source address split into a wave-uniform base (scalar, advanced per K-tile by scalar adds) and
// per-lane byte offsets computed ONCE per output tile: the K loop spends one vector add per piece on addresses instead of
// a clamped 64-bit row * ld + column per piece (~16 vector instructions each, on the issue port the MFMAs need).
struct GldsSrc {
    const char* base;      // K-tile 0 (uniform)
    unsigned off[4];       // per-lane byte offset of piece i
    long long step;        // bytes per K-tile
};
template <int T>
__device__ __forceinline__ void glds_prepare(const bf16_t* __restrict__ P, int ld, int rows, int row0, GldsSrc& d) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    d.step = T == 0 ? 2 * BK : (long long)ld * 2 * BK;
    const int ub = T == 0 ? min(row0, rows - 1) : min(row0, rows - 8);      // uniform first row / column, clamped like the lanes'
    d.base = reinterpret_cast<const char*>(T == 0 ? P + (size_t)ub * ld : P + ub);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = wave * 4 + i;
        if (T == 0) {
            const int r = 8 * j + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            d.off[i] = (unsigned)(min(r, rows - 1 - ub) * ld + c * 8) * 2u;
        } else {
            const int kr = 4 * j + (lane >> 4);
            const int u = lane & 15;
            const int sl = (u >> 1) ^ ks_swz(kr);
            const int gc = min(row0 + sl * 16 + (u & 1) * 8, rows - 8);
            d.off[i] = (unsigned)(kr * ld + (gc - ub)) * 2u;
        }
    }
}
__device__ __forceinline__ void stage_glds_pre(const GldsSrc& d, const char* b /* d.base + K-tile * d.step */, char* lds) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + d.off[i]),
                                         (__attribute__((address_space(3))) void*)(lds + (wave * 4 + i) * 1024), 16, 0, 0);
}

// ---- fragments ---------------------------------------------------------------------------------------------
// Fragment of 16 tile-rows starting at r0 for k-step ks (32 deep): lane holds row (lane&15), k = 8*(lane>>4)+j.
//
// K-contiguous image (T == 0): one ds_read_b128, issued as a plain load (hipcc tracks its lgkmcnt).
// K-strided image   (T == 1): two ds_read_b64_tr_b16, issued through INLINE ASM: hipcc treats the tr-read builtin
//   as possibly aliasing an in-flight LDS-DMA and puts `s_waitcnt vmcnt(0)` in front of it, which drains the
//   global_load_lds pipeline at every phase.  The asm form is invisible to that pass; in exchange the caller must
//   run frag_wait() (s_waitcnt lgkmcnt(0) + sched_barrier) between the last read and the first MFMA, and only then
//   assemble the two halves (cat4) so that any register copy happens after the data has landed.
struct FragT1 { bf16x4 lo, hi; };

__device__ __forceinline__ int tr_lane_off(int lane) {           // byte offset of (k-row 8g+q, column 4p) in slot 0
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    return (8 * g + q) * 256 + p * 8;
}
__device__ __forceinline__ int tr_lane_key(int lane) {           // ks_swz of every k-row this lane addresses
    return ((lane >> 2) & 3) | (((lane >> 4) & 1) << 2);
}

template <int OFF>
__device__ __forceinline__ bf16x4 tr_read_asm(unsigned addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

// addr = LDS byte address of the image + tr_lane_off + ((slot ^ key) << 5); KS selects the 32-deep k-step
template <int KS>
__device__ __forceinline__ FragT1 load_frag_tr(unsigned addr) {
    FragT1 f;
    f.lo = tr_read_asm<KS * 8192>(addr);
    f.hi = tr_read_asm<KS * 8192 + 1024>(addr);
    return f;
}

__device__ __forceinline__ bf16x8 load_frag_row(const char* lds, int r0, int ks, int lane) {
    const int r = r0 + (lane & 15);
    const int c = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds + r * 128 + ((c ^ (r & 7)) << 4));
}

__device__ __forceinline__ void frag_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Operand fragments of NT 16-row tiles x 2 k-steps, for either image kind.
template <int T, int NT>
struct Frags {
    bf16x8 row[NT][2];
    FragT1 tr[NT][2];
    // r0: first tile-row/column of tile 0 inside the image; tiles are 16 apart
    __device__ __forceinline__ void load(const char* img, int r0, int lane) {
        if constexpr (T == 0) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) row[i][ks] = load_frag_row(img, r0 + 16 * i, ks, lane);
        } else {
            const unsigned base = (unsigned)(uintptr_t)img + tr_lane_off(lane);
            const int key = tr_lane_key(lane);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const unsigned a = base + ((((r0 >> 4) + i) ^ key) << 5);
                tr[i][0] = load_frag_tr<0>(a);
                tr[i][1] = load_frag_tr<1>(a);
            }
        }
    }
    __device__ __forceinline__ bf16x8 get(int i, int ks) const {
        if constexpr (T == 0) return row[i][ks];
        else return cat4(tr[i][ks].lo, tr[i][ks].hi);
    }
};

// ---- epilogue ------------------------------------------------------------------------------------------------
// Result stores of the fast epilogue paths, as inline asm: a store hipcc knows about is a pending vector-memory event at
// the back-edge of the persistent 256x256 kernel's tile loop, and with LDS-DMA requests pending beside it (two event kinds
// on one in-order counter) the compiler's wait insertion degrades to `s_waitcnt vmcnt(0)` in front of the next tile's first
// fragment read - every tile waited for the previous tile's stores to be acknowledged, which is exactly what requesting the
// next tile's operands before the stores was meant to avoid (round 3, found in the ISA).  Unknown to the compiler, the
// stores only make its own counted waits for epilogue LOADS stricter (counts are upper bounds of what may stay in flight).
// (`s_nop 1`: on gfx950 a store of more than 8 bytes needs TWO wait states before its data registers may be overwritten -
// llc's hazard recognizer inserts `s_nop 1` there, and cannot see a store inside inline asm.)
__device__ __forceinline__ void store16_asm(void* p, const U4& v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store8_asm(void* p, const uint2& v) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store4_asm(void* p, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store2_asm(void* p, uint32_t v /* low 16 bits */) {
    asm volatile("global_store_short %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// One 16x16 accumulator: the lane owns C[m][n .. n+3].  mode 0: C = alpha*acc (+R).  mode 1 (SwiGLU forward, bf16 out,
// gate/up interleaved along n): additionally aux_out[m][n/2 .. n/2+1] = silu(gate) * up.  mode 2 (SwiGLU backward,
// GEMM N = F): acc is d(act)[m][n..n+3]; reads gate/up from aux_in[m][2n .. 2n+7] and writes d(gate),d(up)
// interleaved to C[m][2n .. 2n+7] (C is [M][2F]); d(act) itself is never stored.
struct Epi {
    void* C; const bf16_t* R; int ldc, ldr; float alpha;
    int mode; const bf16_t* aux_in; bf16_t* aux_out; int ld_aux;
    int M, N;
    int p0, p1;
};
// mode 3 (RoPE forward, the fused q|k|v projection): columns n < p0 are (q, k) heads of p1 features whose interleaved pairs
// (2i, 2i+1) are rotated by row position m % ld_aux with the fp32 (cos, sin) table aux_in = [P][p1/2][2] - torchtune's
// Llama3ScaledRoPE applied to the fp32 accumulator, one bf16 rounding instead of the stand-alone kernel's two.
enum { EPI_NONE = 0, EPI_SWIGLU_FWD = 1, EPI_SWIGLU_BWD = 2, EPI_ROPE = 3 };

// rotate NP adjacent pairs v[0..2NP) that start at column n of row m (n even, all pairs inside one head)
template <int NP>
__device__ __forceinline__ void epi_rope(const Epi& e, int m, int n, float* v) {
    if (n >= e.p0) return;
    const float* table = reinterpret_cast<const float*>(e.aux_in);
    const int pos = m % e.ld_aux, hd = e.p1;
    const float* t = table + ((size_t)pos * (hd >> 1) + ((n % hd) >> 1)) * 2;
#pragma unroll
    for (int i = 0; i < NP; i += 2) {
        const float4 cs = *reinterpret_cast<const float4*>(t + 2 * i);      // c_i, s_i, c_{i+1}, s_{i+1}
        rope_rot(v[2 * i], v[2 * i + 1], cs.x, cs.y);
        rope_rot(v[2 * i + 2], v[2 * i + 3], cs.z, cs.w);
    }
}

template <typename OutT>
__device__ __forceinline__ void epi_store(const Epi& e, bool vec_ok, int m, int n, const f32x4& a) {
    float v[4] = {a[0] * e.alpha, a[1] * e.alpha, a[2] * e.alpha, a[3] * e.alpha};
    if (e.mode == EPI_SWIGLU_BWD) {
        if constexpr (sizeof(OutT) == 2) {
            float g[8], o[8];
            unpack8(*reinterpret_cast<const U4*>(e.aux_in + (size_t)m * e.ld_aux + 2 * n), g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float gt = g[2 * i], up = g[2 * i + 1];
                const float sg = fast_sigmoid(gt);
                o[2 * i] = v[i] * up * sg * (1.f + gt * (1.f - sg));
                o[2 * i + 1] = v[i] * gt * sg;
            }
            store16_asm(reinterpret_cast<bf16_t*>(e.C) + (size_t)m * e.ldc + 2 * n, pack8(o));
        }
        return;
    }
    if (vec_ok && n + 3 < e.N) {
        if (e.R) {
            const uint2 r2 = *reinterpret_cast<const uint2*>(e.R + (size_t)m * e.ldr + n);
            v[0] += __uint_as_float(r2.x << 16); v[1] += __uint_as_float(r2.x & 0xffff0000u);
            v[2] += __uint_as_float(r2.y << 16); v[3] += __uint_as_float(r2.y & 0xffff0000u);
        }
        if (e.mode == EPI_ROPE) epi_rope<2>(e, m, n, v);
        if constexpr (sizeof(OutT) == 2) {
            uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
            store8_asm(reinterpret_cast<bf16_t*>(e.C) + (size_t)m * e.ldc + n, o);
            if (e.mode == EPI_SWIGLU_FWD) {
                const float a0 = silu(v[0]) * v[1];
                const float a1 = silu(v[2]) * v[3];
                store4_asm(e.aux_out + (size_t)m * e.ld_aux + (n >> 1), pack2bf(a0, a1));
            }
        } else {
            store16_asm(reinterpret_cast<float*>(e.C) + (size_t)m * e.ldc + n,
                        (U4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])});
        }
    } else {
        for (int k = 0; k < 4 && n + k < e.N; ++k) {
            float x = v[k];
            if (e.R) x += bf2f(e.R[(size_t)m * e.ldr + n + k]);
            if constexpr (sizeof(OutT) == 2) store2_asm(reinterpret_cast<bf16_t*>(e.C) + (size_t)m * e.ldc + n + k, f2bf(x));
            else store4_asm(reinterpret_cast<float*>(e.C) + (size_t)m * e.ldc + n + k, __float_as_uint(x));
        }
    }
}

// SwiGLU-backward epilogue for a wave's NI x NJ block of 16x16 accumulators (rows mb + 16 i, d(act) columns nb + 16 j):
// the gate/up vectors of TWO accumulator rows (2 * NJ 16-byte loads per lane) are fetched before any of their results is
// stored - issued one by one between the stores they would sit behind every store's address check and the epilogue
// would pay one full memory latency per accumulator.  The loads of the next row pair are issued before this pair's
// arithmetic and stores (two register buffers), so the memory system always has this wave's reads queued.
#ifndef CSM_ABLATE_EPI
#define CSM_ABLATE_EPI 0        // tools/probes: 1 = no gate/up loads, 2 = no stores, 4 = no sigmoid arithmetic
#endif
template <int NI, int NJ>
__device__ __forceinline__ void epi_swiglu_bwd_block(const Epi& e, int mb, int nb, int lane, const f32x4 (&acc)[NI][NJ]) {
    const int g = lane >> 4;
    bf16_t* C = reinterpret_cast<bf16_t*>(e.C);
    U4 gu[2][2][NJ];                     // [buffer][row of the pair][j]: the NEXT row pair is in flight while this one is worked on
    auto fetch = [&](int i, U4 (&dst)[2][NJ]) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // (lanes outside the matrix read a clamped, valid address and never store: no branch between the loads)
                const int m = min(mb + 16 * (i + ii) + (lane & 15), e.M - 1), n = min(nb + 16 * j + 4 * g, e.N - 4);
                if (CSM_ABLATE_EPI & 1) dst[ii][j] = (U4){0x3f803f80u + (unsigned)m, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + (unsigned)n};
                else dst[ii][j] = *reinterpret_cast<const U4*>(e.aux_in + (size_t)m * e.ld_aux + 2 * n);
            }
    };
    fetch(0, gu[0]);
#pragma unroll
    for (int i = 0; i < NI; i += 2) {
        if (i + 2 < NI) fetch(i + 2, gu[((i >> 1) + 1) & 1]);
        const U4 (&cur)[2][NJ] = gu[(i >> 1) & 1];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int m = mb + 16 * (i + ii) + (lane & 15), n = nb + 16 * j + 4 * g;
                float gv[8], o[8];
                unpack8(cur[ii][j], gv);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = acc[i + ii][j][k] * e.alpha, gt = gv[2 * k], up = gv[2 * k + 1];
                    const float sg = (CSM_ABLATE_EPI & 4) ? gt : fast_sigmoid(gt);
                    o[2 * k] = v * up * sg * (1.f + gt * (1.f - sg));
                    o[2 * k + 1] = v * gt * sg;
                }
                if ((CSM_ABLATE_EPI & 2) ? (m < 0 && o[0] == 12345.f) : (m < e.M && n < e.N)) store16_asm(C + (size_t)m * e.ldc + 2 * n, pack8(o));
            }
    }
}

// Two horizontally adjacent 16x16 accumulators (columns nb .. nb+31 of the same 16 rows) stored with 16-byte
// accesses: v_permlane16_swap exchanges the odd 16-lane rows of the left tile with the even rows of the right tile, so
// even lane-groups end up with 8 consecutive columns of the left tile and odd lane-groups with 8 of the right tile
// (64 contiguous bytes per output row and instruction, half the store instructions of the 8-byte form - the epilogue
// of a one-workgroup-per-CU GEMM is store-issue bound).  Falls back to epi_store for fp32 output, fused epilogues,
// unaligned leading dimensions and column overhang.  All 64 lanes must call this (the swap crosses lanes).
template <typename OutT>
__device__ __forceinline__ void epi_store_pair(const Epi& e, bool vec_ok, int m, int nb, int lane, const f32x4& a, const f32x4& b) {
    const int g = lane >> 4;
    bool wide = false;
    if constexpr (sizeof(OutT) == 2)
        wide = vec_ok && (e.mode == EPI_NONE || e.mode == EPI_ROPE || (e.mode == EPI_SWIGLU_FWD && (e.ld_aux & 3) == 0)) && (e.ldc & 7) == 0 &&
               (!e.R || (e.ldr & 7) == 0) && nb + 31 < e.N && ((reinterpret_cast<uintptr_t>(e.C) & 15) == 0) &&
               (!e.R || (reinterpret_cast<uintptr_t>(e.R) & 15) == 0);
    if (!wide) {
        const int n = nb + 4 * g;
        if (m < e.M && n < e.N) epi_store<OutT>(e, vec_ok, m, n, a);
        if (m < e.M && n + 16 < e.N) epi_store<OutT>(e, vec_ok, m, n + 16, b);
        return;
    }
    float v[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[k]), __float_as_uint(b[k]), false, false);
        v[k] = __uint_as_float(r[0]) * e.alpha;
        v[4 + k] = __uint_as_float(r[1]) * e.alpha;
    }
    if (m >= e.M) return;
    const int n = nb + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);
    if (e.R) {
        float rr[8];
        unpack8(*reinterpret_cast<const U4*>(e.R + (size_t)m * e.ldr + n), rr);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += rr[k];
    }
    if (e.mode == EPI_ROPE) epi_rope<4>(e, m, n, v);
    store16_asm(reinterpret_cast<bf16_t*>(e.C) + (size_t)m * e.ldc + n, pack8(v));
    if (e.mode == EPI_SWIGLU_FWD) {   // v = g0,u0,g1,u1,g2,u2,g3,u3 (gate / up interleaved along N): four activations, one 8-byte store
        float a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = silu(v[2 * k]) * v[2 * k + 1];
        uint2 w; w.x = pack2bf(a[0], a[1]); w.y = pack2bf(a[2], a[3]);
        store8_asm(e.aux_out + (size_t)m * e.ld_aux + (n >> 1), w);
    }
}

// A wave's NI x 4 grid of 16x16 accumulators (rows mb + 16 i, columns nb .. nb+63).  When every 16-byte access of the block
// is legal (uniform test) the block takes the fast path: everything it READS - the residual rows, the RoPE table entries -
// is requested up front (residual) or two rows ahead (RoPE table), never between two stores.  Read one by one, as the
// per-pair form does, each of the 2 NI reads costs a full memory round trip before its store can go (R may alias C, so
// the compiler cannot hoist them itself): 16 serial round trips per wave in the 256x256 kernel, whose epilogue no other
// workgroup on the CU overlaps.
// Returns a lower bound of the vector-memory operations every wave issues when no lane is out of range (the persistent
// 256x256 kernels count their waits past them); 0 = unknown.  Only the inline-asm STORES are counted: `asm volatile` cannot
// be merged or dropped, whereas the loads here are the compiler's (it may fuse or elide them), and a count that is too high
// would let an LDS read overtake its LDS-DMA.  Too low only waits longer - and the loads precede the stores
// that depend on them, so nothing is lost.
// NJP = pairs of horizontally adjacent 16 x 16 accumulators per row of the block (2: 64 columns; 1: the 32-column remainder of
// a 96-column wave block of the 256 x 192 tile).
template <typename OutT, int NI, int NJP = 2>
__device__ __forceinline__ int epi_block(const Epi& e, bool vec_ok, int mb, int nb, int lane, const f32x4 (&acc)[NI][4]) {
    bool fast = false;
    if constexpr (sizeof(OutT) == 2)
        fast = vec_ok && (e.mode == EPI_NONE || (e.mode == EPI_ROPE && (e.p0 & 63) == 0 && (e.p1 & 7) == 0) ||
                          (e.mode == EPI_SWIGLU_FWD && (e.ld_aux & 3) == 0)) &&
               (e.ldc & 7) == 0 && (!e.R || (e.ldr & 7) == 0) && nb + 32 * NJP - 1 < e.N && ((reinterpret_cast<uintptr_t>(e.C) & 15) == 0) &&
               (!e.R || (reinterpret_cast<uintptr_t>(e.R) & 15) == 0);
    if (!fast) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = mb + 16 * i + (lane & 15);
#pragma unroll
            for (int jp = 0; jp < NJP; ++jp) epi_store_pair<OutT>(e, vec_ok, m, nb + 32 * jp, lane, acc[i][2 * jp], acc[i][2 * jp + 1]);
        }
        return 0;
    }
    if constexpr (sizeof(OutT) == 2) {
        const int g = lane >> 4;
        const int nl = nb + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);      // this lane's 8 columns of pair jp start at nl + 32 jp
        const int ml = mb + (lane & 15);                               // this lane's row of accumulator row i is ml + 16 i
        bf16_t* C = reinterpret_cast<bf16_t*>(e.C);
        // permlane swap of one accumulator pair -> this lane's 8 consecutive columns, scaled
        auto gather = [&](int i, int jp, float (&v)[8]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][2 * jp][k]), __float_as_uint(acc[i][2 * jp + 1][k]), false, false);
                v[k] = __uint_as_float(r[0]) * e.alpha;
                v[4 + k] = __uint_as_float(r[1]) * e.alpha;
            }
        };
        if (e.mode == EPI_ROPE && nb < e.p0) {
            // ---- rotated columns (q, k heads): table entries two rows ahead; a residual (LoRA up-projection), rare here, inline
            float4 cs[2][2][2][2];                                      // [buffer][row of the pair][jp][half]: (c, s) of 2 pairs each
            auto rope_fetch = [&](int i, float4 (&dst)[2][2][2]) {
                const float* table = reinterpret_cast<const float*>(e.aux_in);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int jp = 0; jp < NJP; ++jp) {
                        const int pos = (ml + 16 * (i + ii)) % e.ld_aux, n = nl + 32 * jp;
                        const float* t = table + ((size_t)pos * (e.p1 >> 1) + ((n % e.p1) >> 1)) * 2;
                        dst[ii][jp][0] = *reinterpret_cast<const float4*>(t);
                        dst[ii][jp][1] = *reinterpret_cast<const float4*>(t + 4);
                    }
            };
            rope_fetch(0, cs[0]);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if ((i & 1) == 0 && i + 2 < NI) rope_fetch(i + 2, cs[((i >> 1) + 1) & 1]);
                const int m = ml + 16 * i;
#pragma unroll
                for (int jp = 0; jp < NJP; ++jp) {
                    float v[8];
                    gather(i, jp, v);
                    const int n = nl + 32 * jp;
                    if (e.R && m < e.M) {
                        float f[8];
                        unpack8(*reinterpret_cast<const U4*>(e.R + (size_t)m * e.ldr + n), f);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += f[k];
                    }
                    const float4 (&q)[2] = cs[(i >> 1) & 1][i & 1][jp];
                    // (a block of the 256 x 192 tile starts at a multiple of 32: the rotated region may end between its two
                    //  32-column halves - never inside one, p0 being a multiple of 64; uniform per half)
                    if (nb + 32 * jp < e.p0)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {                       // pairs (4h, 4h+1) and (4h+2, 4h+3)
                        rope_rot(v[4 * h], v[4 * h + 1], q[h].x, q[h].y);
                        rope_rot(v[4 * h + 2], v[4 * h + 3], q[h].z, q[h].w);
                    }
                    if (m < e.M) store16_asm(C + (size_t)m * e.ldc + n, pack8(v));
                }
            }
            return NJP * NI;      // the asm stores (the table loads are the compiler's: not counted)
        }
        // ---- everything else: all residual rows of the block first
        U4 rr[NI][2];
        if (e.R) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int jp = 0; jp < NJP; ++jp)    // rows past M read row M-1 (never stored): no branch between the loads
                    rr[i][jp] = *reinterpret_cast<const U4*>(e.R + (size_t)min(ml + 16 * i, e.M - 1) * e.ldr + nl + 32 * jp);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = ml + 16 * i;
#pragma unroll
            for (int jp = 0; jp < NJP; ++jp) {
                float v[8];
                gather(i, jp, v);
                if (e.R) {
                    float f[8];
                    unpack8(rr[i][jp], f);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += f[k];
                }
                const int n = nl + 32 * jp;
                if (m < e.M) {
                    store16_asm(C + (size_t)m * e.ldc + n, pack8(v));
                    if (e.mode == EPI_SWIGLU_FWD) {
                        float a[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) a[k] = silu(v[2 * k]) * v[2 * k + 1];
                        uint2 w; w.x = pack2bf(a[0], a[1]); w.y = pack2bf(a[2], a[3]);
                        store8_asm(e.aux_out + (size_t)m * e.ld_aux + (n >> 1), w);
                    }
                }
            }
        }
        return NJP * NI * (1 + (e.mode == EPI_SWIGLU_FWD ? 1 : 0));    // asm stores only
    }
    return 0;
}

// K-extension: after the main loop, kx / 32 more MFMA k-steps whose operands come straight from global memory - xA [M][kx]
// and xB [N][kx], both row-major with leading dimension kx (64-byte rows per k-step: one 16-byte load per lane and
// fragment, 16 rows x 64 contiguous bytes per instruction).  This is how a LoRA adapter joins the frozen projection's
// GEMM: y = x W0^T + (s x A^T) B^T is one product over K + r with xA = s x A^T (padded to 32 columns) and xB = B (reference
// src/csm/mlx/components/lora.py:85-105), accumulated in fp32 in the same accumulators, before any epilogue - no second
// pass over y, and epilogues that must see the sum (RoPE, SwiGLU) stay fused.  Same for dx = dy W0 + (s dy B) A.
template <int NI>
__device__ __forceinline__ void k_extend(const bf16_t* __restrict__ xA, const bf16_t* __restrict__ xB, int kx, int M, int N, int mb,
                                         int nb, int lane, f32x4 (&acc)[NI][4]) {
    const int c = (lane >> 4) * 8;
    for (int k0 = 0; k0 < kx; k0 += 32) {
        bf16x8 fa[NI], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            fb[j] = *reinterpret_cast<const bf16x8*>(xB + (size_t)min(nb + 16 * j + (lane & 15), N - 1) * kx + k0 + c);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            fa[i] = *reinterpret_cast<const bf16x8*>(xA + (size_t)min(mb + 16 * i + (lane & 15), M - 1) * kx + k0 + c);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
}

}  // namespace
