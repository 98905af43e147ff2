// head_dim-64 causal GQA flash attention for gfx950, second generation (the backbone's shape: S = 2048, 32 q / 8 kv heads).
//
// Same contract as attention.hip (which keeps head_dim 128 and stays the A/B reference for head_dim 64):
// replaces torchtune MultiHeadAttention -> F.scaled_dot_product_attention with the mask of reference
// src/csm/models/model.py:59-76 / src/csm/training/utils.py:90-91 (positions arange(S) => plain causal).
//
// What is different from attention.hip, and why (all of it follows from hd = 64 being VALU-bound on the softmax):
//   * mfma_f32_32x32x16_bf16 instead of 16x16x32: half the MFMA instructions for the same FLOPs, and an MFMA holds the SIMD's
//     vector issue port for 8 of its 32 cycles instead of 8 of 16 - the freed issue slots go to exp / fma / cvt.
//   * S^T = K Q^T with the query on the lane (32 queries per wave): max / sum / lse are per-lane scalars, one
//     v_permlane32_swap joins the two lane halves; the P^T accumulators are the B operand of the PV product as they
//     stand (k order 16s + 8(j>>2) + 4h + (j&3), matched by the transposed V reads).
//   * the row sum l rides on the matrix pipe (an all-ones A operand), which has slack here, instead of 32 v_add per block.
//   * the S^T tile of key block j+1 is computed while block j goes through the softmax (two named accumulator sets,
//     loop unrolled by two), so the wave always has independent MFMA and VALU work to issue.
//   * K / V tiles arrive by LDS-DMA (global_load_lds, 16 B per lane) into a 3-stage ring, issued a whole iteration ahead:
//     no staging registers, no ds_write, one barrier per key block.
//   * ONE LDS image layout for every tile: 128-B rows, 16-B chunk c of row r at chunk c ^ f(r),
//     f(r) = ((r>>1)&1)<<2 | (r>>2)&3 - conflict-free both for ds_read_b128 row fragments of a 32-row MFMA operand and
//     for ds_read_b64_tr_b16 transposed fragments (checked against the gfx950 banking rules: MI355X_MICROARCH.md, LDS).
//     The DMA writes LDS linearly, so the XOR goes on the per-lane SOURCE address.
//   * O leaves through LDS so that every store instruction writes whole 128-B rows.
#include "common.h"
#include <math.h>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
// experiment switches for tools/probes/ablate_attn64.sh (never set in the shipped build): what bounds the forward loop?
// bit0: no exp2; bit1: no barrier / DMA wait in the loop; bit2: no LDS fragment reads in the loop; bit3: no MFMA in the loop;
// bit4: no running-max update
#ifndef CSM_ATT64_ABLATE
#define CSM_ATT64_ABLATE 0
#endif
constexpr bool AB_EXP = CSM_ATT64_ABLATE & 1, AB_BAR = CSM_ATT64_ABLATE & 2, AB_LDS = CSM_ATT64_ABLATE & 4, AB_MFMA = CSM_ATT64_ABLATE & 8,
               AB_MAX = CSM_ATT64_ABLATE & 16;
#define MFMA32_(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define MFMA32(a, b, c) (AB_MFMA ? (c) : MFMA32_(a, b, c))

constexpr int TILE = 64 * 128;          // one 64-row x 64-column bf16 image
constexpr int STAGE = 2 * TILE;         // K image + V image
constexpr int NSTAGE = 3;

__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ float fexp2(float x) { return AB_EXP ? x : __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// max over the two lane halves (lanes l and l ^ 32 hold the same query)
__device__ __forceinline__ float halves_max(float x) {
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    float a;
    asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(__uint_as_float(q[0])), "v"(__uint_as_float(q[1])));
    return a;
}
__device__ __forceinline__ float halves_sum(float x) {
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

template <int OFF>
__device__ __forceinline__ bf16x4 tr_read(unsigned addr) {
    bf16x4 v;
    if (AB_LDS) { asm volatile("" : "=v"(v) : "v"(addr)); return v; }
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ bf16x8 row_read(unsigned addr) {
    bf16x8 v;
    if (AB_LDS) { asm volatile("" : "=v"(v) : "v"(addr)); return v; }
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(n) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VM_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(n) : "memory")

__device__ __forceinline__ bf16x8 pack8f(const f32x16& a, int s) {   // registers 8s .. 8s+7 -> one k-step's B fragment
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)f2bf(a[8 * s + j]);
    return r;
}

// 1-D XCD-aware work order shared by the kernels of this file: a contiguous run of (batch, kv-head) pairs per XCD, and
// inside a run the heaviest causal blocks first (see attention.hip for the reasoning).  Returns (pair, head-in-group, block).
__device__ __forceinline__ void work_item(int nblk, int rep, int& pair, int& hh, int& blk) {
    const int per_pair = rep * nblk;
    const int T = gridDim.x, id = blockIdx.x, xcd = id & 7, within = id >> 3;
    const int q8 = T >> 3, r8 = T & 7;
    const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    pair = nid / per_pair;
    int local = nid % per_pair;
    const int run = (xcd < r8) ? q8 + 1 : q8;
    const int base = nid - within;
    if (run % per_pair == 0 && base % per_pair == 0) {
        const int ncomb = (run / per_pair) * rep;
        const int comb = within % ncomb;
        pair = base / per_pair + comb / rep;
        local = (comb % rep) * nblk + within / ncomb;
    }
    hh = local / nblk;
    blk = nblk - 1 - (local % nblk);
}

// one wave's 2 KiB share (pieces 2w, 2w+1) of a 64-row tile, by LDS-DMA.  The source address is a wave-uniform base
// (tile start: scalar arithmetic) plus a per-lane byte offset computed ONCE (lane_off[i], see dma_lane_off), so a tile
// costs no vector instructions besides the two loads; only a tile that overhangs the sequence end (rows clamped to
// S - 1) recomputes its offsets.
__device__ __forceinline__ void dma_lane_off(int ld, int wave, int lane, unsigned (&off)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (2 * wave + i) + (lane >> 3);
        off[i] = (unsigned)(row * ld + (((lane & 7) ^ swz(row)) << 3)) * 2u;
    }
}
__device__ __forceinline__ void dma_tile(const bf16_t* __restrict__ P, int ld, int S, int row0, char* img, int wave, int lane,
                                         const unsigned (&off)[2]) {
    const char* base = reinterpret_cast<const char*>(P) + (size_t)row0 * ld * 2;      // wave-uniform
    if (row0 + 64 <= S) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[i]),
                                             (__attribute__((address_space(3))) void*)(img + (2 * wave + i) * 1024), 16, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 8 * (2 * wave + i) + (lane >> 3);
            int gr = row0 + row;
            gr = gr < S ? gr : S - 1;
            const bf16_t* src = P + (size_t)gr * ld + (((lane & 7) ^ swz(row)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(img + (2 * wave + i) * 1024), 16, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// forward: workgroup = 128 queries of one (b, h) = 4 waves x 32 queries; key blocks of 64
__global__ __launch_bounds__(256, 3) void attn64_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse, int S, int H, int KV, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rep = H / KV;
    const int nqblk = (S + 127) / 128;
    int pair, hh, qb;
    work_item(nqblk, rep, pair, hh, qb);
    const int kvh = pair % KV, b = pair / KV;
    const int hq = kvh * rep + hh;
    const int ld = (H + 2 * KV) * 64;
    const bf16_t* Qp = qkv + (size_t)b * S * ld + hq * 64;
    const bf16_t* Kp = qkv + (size_t)b * S * ld + (H + kvh) * 64;
    const bf16_t* Vp = qkv + (size_t)b * S * ld + (H + KV + kvh) * 64;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q_base = qb * 128;
    const int qw = q_base + 32 * wave;                 // first query of this wave
    const int qrow = qw + r;
    const int qc = qrow < S ? qrow : S - 1;
    const float c2 = scale * 1.4426950408889634f;

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Qp + (size_t)qc * ld + 16 * ks + 8 * h);

    // per-lane LDS offsets inside a stage.  K row fragments: row 32t + r, chunk 2ks + h (t -> immediate offset 4096 t)
    const unsigned sbase = (unsigned)(uintptr_t)smem;
    unsigned koff[4];
    {
        const int fx = swz(r);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) koff[ks] = sbase + r * 128 + (((2 * ks + h) ^ fx) << 4);
    }
    // V transposed fragments for (dt, u): k-rows 32t + 16s + 8u + 4(g>>1) + q, columns 32dt + 16(g&1) + 4p .. +3
    unsigned voff[2][2];
    {
        const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = 8 * u + 4 * (g >> 1) + q4;
                const int ch = 4 * dt + 2 * (g & 1) + (p >> 1);
                voff[dt][u] = sbase + TILE + row * 128 + ((ch ^ swz(row)) << 4) + (p & 1) * 8;
            }
    }

    int lastq = q_base + 127;
    lastq = lastq < S ? lastq : S - 1;
    const int nkb = lastq / 64 + 1;                     // key blocks (64 keys) of the workgroup
    int lastw = qw + 31;
    lastw = lastw < S ? lastw : S - 1;
    const int ntw = qw < S ? lastw / 32 + 1 : 0;        // key TILES (32 keys) this wave needs

    unsigned dma_off[2];
    dma_lane_off(ld, wave, lane, dma_off);
    auto issue = [&](int kb, int st) {
        dma_tile(Kp, ld, S, kb * 64, smem + st * STAGE, wave, lane, dma_off);
        dma_tile(Vp, ld, S, kb * 64, smem + st * STAGE + TILE, wave, lane, dma_off);
    };

    f32x16 o0, o1, lacc, zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; lacc[i] = 0.f; zero16[i] = 0.f; }
    float m = -INFINITY;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;

#define SB() __builtin_amdgcn_sched_barrier(0)
#define EXP2(S_, I_) { S_[I_] = fexp2(fmaf(S_[I_], c2, nb)); S_[I_ + 1] = fexp2(fmaf(S_[I_ + 1], c2, nb)); }
    // One 32-key tile (index t) of one wave, software-pipelined two tiles deep.  On entry: c = S^T of tile t, mnew = the
    // running max including tile t, kf0..3 = K row fragments of tile t+1.  Instruction order is pinned (sched_barrier
    // between groups): an MFMA is followed by the exp2 of two scores (8 + 2 x 12 issue cycles per 32-cycle MFMA), so
    // matrix pipe and VALU run together inside one wave instead of relying on the SIMD's other waves:
    //   R1  4 MFMA  S^T(t+1)               |  exp2 of scores 0..7 of tile t   (k-step 0 of the PV product)
    //       K fragments of tile t+2 requested (they have all of R2 + R3 + the next R1's head to land)
    //   R2  3 MFMA  row sum + PV, k-step 0 |  exp2 of scores 8..15
    //   R3  3 MFMA  row sum + PV, k-step 1 |  causal mask (diagonal tiles only) and row max of S^T(t+1) -> mnew
    // so nothing a tile needs first (its max, its K fragments) is computed or requested at its own head.
    // A / AN / ANN = byte offsets of tiles t, t+1, t+2 inside the ring: compile-time (the block loop is unrolled over the
    // three stages), so every LDS read is a lane-constant address register plus an immediate.
    bf16x8 kf0, kf1, kf2, kf3;
    float mnew = -INFINITY;
    auto rowmax = [&](const f32x16& x) {
        float x0 = vmax3(x[0], x[1], x[2]), x1 = vmax3(x[3], x[4], x[5]), x2 = vmax3(x[6], x[7], x[8]), x3 = vmax3(x[9], x[10], x[11]);
        x0 = vmax3(x0, x[12], x[13]); x1 = vmax3(x1, x[14], x[15]);
        x0 = vmax3(x0, x1, x2);
        return vmax3(x0, x3, m);
    };
    auto causal = [&](f32x16& x, int key0) {
        if (key0 + 31 > qw) {                           // the tile crosses this wave's diagonal (wave-uniform)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (key0 + (i & 3) + 8 * (i >> 2) + 4 * h > qrow) x[i] = -INFINITY;
        }
    };
    auto step = [&](auto a_tag, auto an_tag, auto ann_tag, int key0, f32x16& c, f32x16& n) {
        constexpr int A = decltype(a_tag)::value, AN = decltype(an_tag)::value, ANN = decltype(ann_tag)::value;
        // V^T fragments: [k-step s][column half dt][row group u]
        const bf16x4 v000 = tr_read<A>(voff[0][0]), v001 = tr_read<A>(voff[0][1]);
        const bf16x4 v010 = tr_read<A>(voff[1][0]), v011 = tr_read<A>(voff[1][1]);
        const bf16x4 v100 = tr_read<A + 2048>(voff[0][0]), v101 = tr_read<A + 2048>(voff[0][1]);
        const bf16x4 v110 = tr_read<A + 2048>(voff[1][0]), v111 = tr_read<A + 2048>(voff[1][1]);
        if (!AB_MAX && !__all(mnew == m)) {             // rescale only when some query's running max moved
            const float alpha = fexp2((m - mnew) * c2);
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            lacc[0] *= alpha;                           // every row of lacc is the same sum; only row 0 is read
            m = mnew;
        }
        const float nb = -m * c2;
        // ---- R1
        LGKM_WAIT(8);                                    // the K fragments (requested one step ago) have landed
        n = MFMA32(kf0, qf[0], zero16); SB(); EXP2(c, 0) SB();
        n = MFMA32(kf1, qf[1], n); SB(); EXP2(c, 2) SB();
        n = MFMA32(kf2, qf[2], n); SB(); EXP2(c, 4) SB();
        n = MFMA32(kf3, qf[3], n); SB(); EXP2(c, 6) SB();
        const bf16x8 p0 = pack8f(c, 0);
        LGKM_WAIT(0);                                    // this tile's V fragments
        kf0 = row_read<ANN>(koff[0]); kf1 = row_read<ANN>(koff[1]); kf2 = row_read<ANN>(koff[2]); kf3 = row_read<ANN>(koff[3]);
        // ---- R2
        lacc = MFMA32(ones, p0, lacc); SB(); EXP2(c, 8) SB();
        o0 = MFMA32(cat4(v000, v001), p0, o0); SB(); EXP2(c, 10) SB();
        o1 = MFMA32(cat4(v010, v011), p0, o1); SB(); EXP2(c, 12) EXP2(c, 14) SB();
        const bf16x8 p1 = pack8f(c, 1);
        // ---- R3
        lacc = MFMA32(ones, p1, lacc); SB();
        causal(n, key0 + 32);
        o0 = MFMA32(cat4(v100, v101), p1, o0); SB();
        const float mx = rowmax(n); SB();
        o1 = MFMA32(cat4(v110, v111), p1, o1); SB();
        mnew = AB_MAX ? m : halves_max(mx);
    };
#undef EXP2

    issue(0, 0);
    if (nkb > 1) { issue(1, 1); VM_WAIT(4); } else { VM_WAIT(0); }
    __builtin_amdgcn_s_barrier();

    // tile positions inside the ring: stage s, tile 0 / 1
    using P00 = std::integral_constant<int, 0>;
    using P01 = std::integral_constant<int, 4096>;
    using P10 = std::integral_constant<int, STAGE>;
    using P11 = std::integral_constant<int, STAGE + 4096>;
    using P20 = std::integral_constant<int, 2 * STAGE>;
    using P21 = std::integral_constant<int, 2 * STAGE + 4096>;
    f32x16 sa, sb;
    if (ntw > 0) {                                       // S^T and row max of tile 0, K fragments of tile 1
        bf16x8 k0 = row_read<0>(koff[0]), k1 = row_read<0>(koff[1]), k2 = row_read<0>(koff[2]), k3 = row_read<0>(koff[3]);
        kf0 = row_read<4096>(koff[0]); kf1 = row_read<4096>(koff[1]); kf2 = row_read<4096>(koff[2]); kf3 = row_read<4096>(koff[3]);
        LGKM_WAIT(4);
        sa = MFMA32(k0, qf[0], zero16); sa = MFMA32(k1, qf[1], sa); sa = MFMA32(k2, qf[2], sa); sa = MFMA32(k3, qf[3], sa);
        causal(sa, 0);
        mnew = halves_max(rowmax(sa));
    }
    // One key block: wait for block kb+1 (issued one block ago), barrier, issue block kb+2 into the stage the barrier has
    // just released, then the block's two tiles.  A step also computes S^T of tile t+1 and requests K of tile t+2 when the
    // wave will not use them (garbage from a stage that holds some other block, never consumed), so that a step has a
    // single code path: two versions of it would merge ~100 live registers through copies.
#define BLOCK(KB, C0, C1, N0, N1, NNIDX)                                                                               \
    {                                                                                                                  \
        if (!AB_BAR) { VM_WAIT(0); __builtin_amdgcn_s_barrier(); }                                                     \
        if (!AB_BAR && (KB) + 2 < nkb) issue((KB) + 2, NNIDX);                                                         \
        if (2 * (KB) < ntw) step(C0{}, C1{}, N0{}, (KB) * 64, sa, sb);                                                 \
        if (2 * (KB) + 1 < ntw) step(C1{}, N0{}, N1{}, (KB) * 64 + 32, sb, sa);                                        \
    }
    for (int kb = 0; kb < nkb; kb += 3) {
        BLOCK(kb, P00, P01, P10, P11, 2)
        if (kb + 1 >= nkb) break;
        BLOCK(kb + 1, P10, P11, P20, P21, 0)
        if (kb + 2 >= nkb) break;
        BLOCK(kb + 2, P20, P21, P00, P01, 1)
    }
#undef BLOCK
#undef SB

    // ---- epilogue: O^T registers -> LDS [q][d] (144-B rows) -> whole 128-B rows to HBM
    __builtin_amdgcn_s_barrier();                        // every wave is done with the stages
    {
        const float l = lacc[0];
        const float inv = 1.f / l;
        char* ob = smem + wave * (32 * 144);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            uint2 w0, w1;
            w0.x = pack2bf(o0[4 * gq] * inv, o0[4 * gq + 1] * inv); w0.y = pack2bf(o0[4 * gq + 2] * inv, o0[4 * gq + 3] * inv);
            w1.x = pack2bf(o1[4 * gq] * inv, o1[4 * gq + 1] * inv); w1.y = pack2bf(o1[4 * gq + 2] * inv, o1[4 * gq + 3] * inv);
            *reinterpret_cast<uint2*>(ob + r * 144 + (8 * gq + 4 * h) * 2) = w0;           // d = 8 gq + 4h .. +3
            *reinterpret_cast<uint2*>(ob + r * 144 + (32 + 8 * gq + 4 * h) * 2) = w1;      // d = 32 + ...
        }
        if (h == 0 && qrow < S) lse[((size_t)b * H + hq) * S + qrow] = m * scale + logf(l);
        // (wave-private LDS region: a wave's own LDS operations execute in order, no wait or barrier needed)
        bf16_t* op = out + ((size_t)b * S) * (H * 64) + hq * 64;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = 8 * pass + (lane >> 3);
            const U4 v = *reinterpret_cast<const U4*>(ob + row * 144 + (lane & 7) * 16);
            if (qw + row < S) *reinterpret_cast<U4*>(op + (size_t)(qw + row) * (H * 64) + (lane & 7) * 8) = v;
        }
    }
}

// Backward of torchtune's interleaved-pair RoPE on two adjacent pairs (x0,x1), (x2,x3) of position p (see attention.hip)
__device__ __forceinline__ void unrope2(const float* __restrict__ table, int p, int hd, int pair, float& g0, float& g1, float& g2,
                                        float& g3) {
    const float4 t = *reinterpret_cast<const float4*>(table + ((size_t)p * (hd >> 1) + pair) * 2);   // c0, s0, c1, s1
    rope_rot(g0, g1, t.x, -t.y);
    rope_rot(g2, g3, t.z, -t.w);
}

// ------------------------------------------------------------------------------------------------------------------
// dQ: the forward's skeleton (128 queries of one (b, h) per workgroup, 32 per wave on the lanes, 32-key tiles from the
// K / V ring) with three products per tile: S^T = K Q^T, dP^T = V dO^T (its accumulators START at -delta, so
// dS = P o dP needs no subtraction), dQ^T += K^T dS^T (K^T by transposed reads of the same K image; dS^T straight from
// the accumulators).  P = exp2(c S - lse) needs no running max.  Also computes delta = rowsum(dO o O) from the same row
// fragments and publishes it for the dK/dV kernel, and applies the RoPE backward in the epilogue (rope != null).
__global__ __launch_bounds__(256, 2) void attn64_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                           const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                           float* __restrict__ delta, bf16_t* __restrict__ dqkv, int S, int H, int KV,
                                                           float scale, const float* __restrict__ rope, long long nstat /* B H S */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rep = H / KV;
    const int nqblk = (S + 127) / 128;
    int pair, hh, qb;
    work_item(nqblk, rep, pair, hh, qb);
    const int kvh = pair % KV, b = pair / KV;
    const int hq = kvh * rep + hh;
    const int ld = (H + 2 * KV) * 64, ldo = H * 64;
    const bf16_t* Qp = qkv + (size_t)b * S * ld + hq * 64;
    const bf16_t* Kp = qkv + (size_t)b * S * ld + (H + kvh) * 64;
    const bf16_t* Vp = qkv + (size_t)b * S * ld + (H + KV + kvh) * 64;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q_base = qb * 128;
    const int qw = q_base + 32 * wave;
    const int qrow = qw + r;
    const int qc = qrow < S ? qrow : S - 1;
    const float c2 = scale * 1.4426950408889634f;

    bf16x8 qf[4], dof[4];
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8*>(Qp + (size_t)qc * ld + 16 * ks + 8 * h);
        dof[ks] = *reinterpret_cast<const bf16x8*>(dout + ((size_t)b * S + qc) * ldo + hq * 64 + 16 * ks + 8 * h);
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(out + ((size_t)b * S + qc) * ldo + hq * 64 + 16 * ks + 8 * h);
#pragma unroll
        for (int j = 0; j < 8; ++j) dsum += bf2f((bf16_t)dof[ks][j]) * bf2f((bf16_t)of[j]);
    }
    const float my_delta = halves_sum(dsum);
    const float nlse2 = -lse[((size_t)b * H + hq) * S + qc] * 1.4426950408889634f;
    // published for the dK/dV kernel in the form it consumes them (it copies them to LDS by DMA, no arithmetic on the way):
    // delta[0 .. nstat) = -delta, delta[nstat .. 2 nstat) = -lse * log2(e), each [B][H][S]
    if (h == 0 && qrow < S) {
        delta[((size_t)b * H + hq) * S + qrow] = -my_delta;
        delta[(size_t)nstat + ((size_t)b * H + hq) * S + qrow] = nlse2;
    }

    const unsigned sbase = (unsigned)(uintptr_t)smem;
    unsigned koff[4];                                    // row fragments of the K image; the V image's are + TILE
    {
        const int fx = swz(r);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) koff[ks] = sbase + r * 128 + (((2 * ks + h) ^ fx) << 4);
    }
    unsigned toff[2][2];                                 // transposed fragments of the K image (see the forward's voff)
    {
        const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = 8 * u + 4 * (g >> 1) + q4;
                const int ch = 4 * dt + 2 * (g & 1) + (p >> 1);
                toff[dt][u] = sbase + row * 128 + ((ch ^ swz(row)) << 4) + (p & 1) * 8;
            }
    }

    int lastq = q_base + 127;
    lastq = lastq < S ? lastq : S - 1;
    const int nkb = lastq / 64 + 1;
    int lastw = qw + 31;
    lastw = lastw < S ? lastw : S - 1;
    const int ntw = qw < S ? lastw / 32 + 1 : 0;

    unsigned dma_off[2];
    dma_lane_off(ld, wave, lane, dma_off);
    auto issue = [&](int kb, int st) {
        dma_tile(Kp, ld, S, kb * 64, smem + st * STAGE, wave, lane, dma_off);
        dma_tile(Vp, ld, S, kb * 64, smem + st * STAGE + TILE, wave, lane, dma_off);
    };

    f32x16 dq0, dq1, zero16, nd16;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq0[i] = 0.f; dq1[i] = 0.f; zero16[i] = 0.f; nd16[i] = -my_delta; }

#define SB() __builtin_amdgcn_sched_barrier(0)
#define DS2(I_) { cs[I_] = fexp2(fmaf(cs[I_], c2, nlse2)) * cdp[I_]; cs[I_ + 1] = fexp2(fmaf(cs[I_ + 1], c2, nlse2)) * cdp[I_ + 1]; }
    // one 32-key tile; on entry cs / cdp = S^T and dP^T - delta of this tile, kf / vf = K and V row fragments of the next
    bf16x8 kf0, kf1, kf2, kf3, vf0, vf1, vf2, vf3;
    auto step = [&](auto a_tag, auto ann_tag, int key0, f32x16& cs, f32x16& cdp, f32x16& ns, f32x16& ndp) {
        constexpr int A = decltype(a_tag)::value, ANN = decltype(ann_tag)::value;
        // K^T fragments of this tile: [k-step s][row half dt][row group u]
        const bf16x4 t000 = tr_read<A>(toff[0][0]), t001 = tr_read<A>(toff[0][1]);
        const bf16x4 t010 = tr_read<A>(toff[1][0]), t011 = tr_read<A>(toff[1][1]);
        const bf16x4 t100 = tr_read<A + 2048>(toff[0][0]), t101 = tr_read<A + 2048>(toff[0][1]);
        const bf16x4 t110 = tr_read<A + 2048>(toff[1][0]), t111 = tr_read<A + 2048>(toff[1][1]);
        if (key0 + 31 > qw) {                           // diagonal tile: exp2(-inf) = 0 kills the masked scores
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (key0 + (i & 3) + 8 * (i >> 2) + 4 * h > qrow) cs[i] = -INFINITY;
        }
        LGKM_WAIT(8);                                    // kf / vf (requested one step ago) have landed
        ns = MFMA32(kf0, qf[0], zero16); SB(); DS2(0) SB();
        ndp = MFMA32(vf0, dof[0], nd16); SB(); DS2(2) SB();
        ns = MFMA32(kf1, qf[1], ns); SB(); DS2(4) SB();
        ndp = MFMA32(vf1, dof[1], ndp); SB(); DS2(6) SB();
        ns = MFMA32(kf2, qf[2], ns); SB(); DS2(8) SB();
        ndp = MFMA32(vf2, dof[2], ndp); SB(); DS2(10) SB();
        ns = MFMA32(kf3, qf[3], ns); SB(); DS2(12) SB();
        ndp = MFMA32(vf3, dof[3], ndp); SB(); DS2(14) SB();
        const bf16x8 p0 = pack8f(cs, 0), p1 = pack8f(cs, 1);
        LGKM_WAIT(0);
        kf0 = row_read<ANN>(koff[0]); kf1 = row_read<ANN>(koff[1]); kf2 = row_read<ANN>(koff[2]); kf3 = row_read<ANN>(koff[3]);
        vf0 = row_read<ANN + TILE>(koff[0]); vf1 = row_read<ANN + TILE>(koff[1]); vf2 = row_read<ANN + TILE>(koff[2]); vf3 = row_read<ANN + TILE>(koff[3]);
        dq0 = MFMA32(cat4(t000, t001), p0, dq0);
        dq1 = MFMA32(cat4(t010, t011), p0, dq1);
        dq0 = MFMA32(cat4(t100, t101), p1, dq0);
        dq1 = MFMA32(cat4(t110, t111), p1, dq1);
    };
#undef DS2

    issue(0, 0);
    if (nkb > 1) { issue(1, 1); VM_WAIT(4); } else { VM_WAIT(0); }
    __builtin_amdgcn_s_barrier();

    using P00 = std::integral_constant<int, 0>;
    using P01 = std::integral_constant<int, 4096>;
    using P10 = std::integral_constant<int, STAGE>;
    using P11 = std::integral_constant<int, STAGE + 4096>;
    using P20 = std::integral_constant<int, 2 * STAGE>;
    using P21 = std::integral_constant<int, 2 * STAGE + 4096>;
    f32x16 sa, sb, da, db;
    if (ntw > 0) {
        bf16x8 k0 = row_read<0>(koff[0]), k1 = row_read<0>(koff[1]), k2 = row_read<0>(koff[2]), k3 = row_read<0>(koff[3]);
        bf16x8 v0 = row_read<TILE>(koff[0]), v1 = row_read<TILE>(koff[1]), v2 = row_read<TILE>(koff[2]), v3 = row_read<TILE>(koff[3]);
        kf0 = row_read<4096>(koff[0]); kf1 = row_read<4096>(koff[1]); kf2 = row_read<4096>(koff[2]); kf3 = row_read<4096>(koff[3]);
        vf0 = row_read<TILE + 4096>(koff[0]); vf1 = row_read<TILE + 4096>(koff[1]); vf2 = row_read<TILE + 4096>(koff[2]); vf3 = row_read<TILE + 4096>(koff[3]);
        LGKM_WAIT(8);
        sa = MFMA32(k0, qf[0], zero16); sa = MFMA32(k1, qf[1], sa); sa = MFMA32(k2, qf[2], sa); sa = MFMA32(k3, qf[3], sa);
        da = MFMA32(v0, dof[0], nd16); da = MFMA32(v1, dof[1], da); da = MFMA32(v2, dof[2], da); da = MFMA32(v3, dof[3], da);
    }
#define BLOCK(KB, C0, C1, N0, N1, NNIDX)                                                                               \
    {                                                                                                                  \
        VM_WAIT(0);                                                                                                    \
        __builtin_amdgcn_s_barrier();                                                                                  \
        if ((KB) + 2 < nkb) issue((KB) + 2, NNIDX);                                                                    \
        if (2 * (KB) < ntw) step(C0{}, N0{}, (KB) * 64, sa, da, sb, db);                                               \
        if (2 * (KB) + 1 < ntw) step(C1{}, N1{}, (KB) * 64 + 32, sb, db, sa, da);                                      \
    }
    for (int kb = 0; kb < nkb; kb += 3) {
        BLOCK(kb, P00, P01, P10, P11, 2)
        if (kb + 1 >= nkb) break;
        BLOCK(kb + 1, P10, P11, P20, P21, 0)
        if (kb + 2 >= nkb) break;
        BLOCK(kb + 2, P20, P21, P00, P01, 1)
    }
#undef BLOCK
#undef SB

    // ---- epilogue: dQ^T registers (x 1/sqrt(hd), RoPE^T) -> LDS [q][d] -> whole 128-B rows of the q block of dqkv
    __builtin_amdgcn_s_barrier();
    {
        char* ob = smem + wave * (32 * 144);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float a0 = dq0[4 * gq] * scale, a1 = dq0[4 * gq + 1] * scale, a2 = dq0[4 * gq + 2] * scale, a3 = dq0[4 * gq + 3] * scale;
            float b0 = dq1[4 * gq] * scale, b1 = dq1[4 * gq + 1] * scale, b2 = dq1[4 * gq + 2] * scale, b3 = dq1[4 * gq + 3] * scale;
            if (rope) {
                unrope2(rope, qc, 64, 4 * gq + 2 * h, a0, a1, a2, a3);
                unrope2(rope, qc, 64, 16 + 4 * gq + 2 * h, b0, b1, b2, b3);
            }
            uint2 w0, w1;
            w0.x = pack2bf(a0, a1); w0.y = pack2bf(a2, a3);
            w1.x = pack2bf(b0, b1); w1.y = pack2bf(b2, b3);
            *reinterpret_cast<uint2*>(ob + r * 144 + (8 * gq + 4 * h) * 2) = w0;
            *reinterpret_cast<uint2*>(ob + r * 144 + (32 + 8 * gq + 4 * h) * 2) = w1;
        }
        bf16_t* op = dqkv + ((size_t)b * S) * ld + hq * 64;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = 8 * pass + (lane >> 3);
            const U4 v = *reinterpret_cast<const U4*>(ob + row * 144 + (lane & 7) * 16);
            if (qw + row < S) *reinterpret_cast<U4*>(op + (size_t)(qw + row) * ld + (lane & 7) * 8) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// dK / dV: workgroup = 64 keys of one (b, kv-head), 4 waves: wave w owns key half (w & 1) - 32 keys on its lanes, their
// K^T / V^T fragments in registers for the whole kernel - and takes the 32-query tile of parity (w >> 1) of every
// 64-query block, so two waves share each key half and their dK^T / dV^T accumulators are added through LDS at the end
// (no atomics, a fixed order: deterministic).  The workgroup walks the 4 q-heads of the group x the query blocks at or
// below the diagonal; Q and dO blocks arrive by LDS-DMA into a 3-stage ring (ONE image each, read by rows for S and dP and
// transposed for dK^T and dV^T), -lse*log2(e) and -delta (written in that form by the dQ kernel) arrive by LDS-DMA too:
// the first is the addend of the exp2 argument, the second is read straight into the accumulators dP starts from.
//   S = Q K^T, dP = dO V^T - delta (query on the register index, key on the lane)  ->  P = exp2(c S - lse), dS = P o dP
//   dV^T += dO^T P, dK^T += Q^T dS   (P / dS leave their accumulators as the B operand, k = query)
template <int OFF>
__device__ __forceinline__ f32x4 stat_read(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
constexpr int KSTAGE = 2 * TILE + 512;      // Q image + dO image + 64 x (-lse*log2e) + 64 x (-delta)
constexpr int KNSTAGE = 4;                  // ring depth of the dK/dV kernel: 66 KiB per workgroup, two workgroups per CU

__global__ __launch_bounds__(256, 2) void attn64_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                            const float* __restrict__ nlse2 /* -lse log2e */, const float* __restrict__ ndelta /* -delta */,
                                                            bf16_t* __restrict__ dqkv, int S, int H, int KV, float scale,
                                                            const float* __restrict__ rope) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rep = H / KV;
    const int nkblk = (S + 63) / 64;
    int kblk, kvh, b;
    {   // XCD-aware 1-D grid: a contiguous run of (b, kv-head) pairs per XCD, the heaviest key blocks of the run first
        const int T = gridDim.x, id = blockIdx.x, xcd = id & 7, within = id >> 3;
        const int q8 = T >> 3, r8 = T & 7;
        const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
        int pair = nid / nkblk;
        kblk = nid % nkblk;
        const int run = (xcd < r8) ? q8 + 1 : q8, base = nid - within;
        if (run % nkblk == 0 && base % nkblk == 0) {
            const int npairs = run / nkblk;
            kblk = within / npairs;
            pair = base / nkblk + within % npairs;
        }
        kvh = pair % KV;
        b = pair / KV;
    }
    const int ld = (H + 2 * KV) * 64, ldo = H * 64;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kh = wave & 1, qp = wave >> 1;
    const int key0w = kblk * 64 + 32 * kh;               // first key of this wave
    const int key = key0w + r;
    const int kc = key < S ? key : S - 1;
    const float c2 = scale * 1.4426950408889634f;

    bf16x8 kf[4], vf[4];                                 // B operands: K^T / V^T [d][key], key on the lane
    {
        const bf16_t* Kp = qkv + ((size_t)b * S + kc) * ld + (H + kvh) * 64;
        const bf16_t* Vp = qkv + ((size_t)b * S + kc) * ld + (H + KV + kvh) * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(Kp + 16 * ks + 8 * h);
            vf[ks] = *reinterpret_cast<const bf16x8*>(Vp + 16 * ks + 8 * h);
        }
    }
    const unsigned sbase = (unsigned)(uintptr_t)smem;
    unsigned roff[4];                                    // row fragments of this wave's query tile (Q image; dO image + TILE)
    {
        const int fx = swz(r);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) roff[ks] = sbase + qp * 4096 + r * 128 + (((2 * ks + h) ^ fx) << 4);
    }
    unsigned toff[2][2];                                 // transposed fragments of the tile (k = query rows)
    {
        const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = 8 * u + 4 * (g >> 1) + q4;
                const int ch = 4 * dt + 2 * (g & 1) + (p >> 1);
                toff[dt][u] = sbase + qp * 4096 + row * 128 + ((ch ^ swz(row)) << 4) + (p & 1) * 8;
            }
    }
    const unsigned soff = sbase + 2 * TILE + (32 * qp + 4 * h) * 4;      // stats of queries 32 qp + 4h + {0..3} (+ 8k: immediate 32 k)

    const int nqb = (S + 63) / 64;
    const int per_head = nqb - kblk;                     // query blocks at or below the diagonal
    const int niter = rep * per_head;

    unsigned dma_q[2], dma_o[2];
    dma_lane_off(ld, wave, lane, dma_q);
    dma_lane_off(ldo, wave, lane, dma_o);
    // The block's 64 x (-lse log2e) and 64 x (-delta), written in that form by the dQ kernel, go to LDS by the same LDS-DMA
    // as the tiles (4 bytes per lane: wave 0 the first row, wave 1 the second), so that no register - and with it no
    // compiler-inserted wait on the freshly issued loads - sits between a request and its use two steps later.
    auto issue = [&](int it, int st) {
        const int hh = it / per_head, qb = kblk + it % per_head;
        const int hq = kvh * rep + hh;
        dma_tile(qkv + (size_t)b * S * ld + hq * 64, ld, S, qb * 64, smem + st * KSTAGE, wave, lane, dma_q);
        dma_tile(dout + (size_t)b * S * ldo + hq * 64, ldo, S, qb * 64, smem + st * KSTAGE + TILE, wave, lane, dma_o);
        if (wave < 2) {
            int q = qb * 64 + lane;
            q = q < S ? q : S - 1;
            const float* sp = (wave == 0 ? nlse2 : ndelta) + ((size_t)b * H + hq) * S + q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                             (__attribute__((address_space(3))) void*)(smem + st * KSTAGE + 2 * TILE + wave * 256), 4, 0, 0);
        }
    };

    f32x16 dk0, dk1, dv0, dv1, zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk0[i] = 0.f; dk1[i] = 0.f; dv0[i] = 0.f; dv1[i] = 0.f; zero16[i] = 0.f; }

#define SB() __builtin_amdgcn_sched_barrier(0)
    // Row fragments and statistics of the NEXT tile are requested while the current tile's last four MFMAs run (their
    // registers are free by then), so a tile never starts by waiting for LDS.  Every wave runs every query block - a tile
    // entirely above the diagonal (one per wave at most) is masked to zero rather than skipped - so there is one code path.
    bf16x8 qr0, qr1, qr2, qr3, or0, or1, or2, or3;
    f32x4 n0, n1, n2, n3, e0, e1, e2, e3;
    auto fetch = [&](auto a_tag) {
        constexpr int A = decltype(a_tag)::value;
        qr0 = row_read<A>(roff[0]); qr1 = row_read<A>(roff[1]); qr2 = row_read<A>(roff[2]); qr3 = row_read<A>(roff[3]);
        or0 = row_read<A + TILE>(roff[0]); or1 = row_read<A + TILE>(roff[1]); or2 = row_read<A + TILE>(roff[2]); or3 = row_read<A + TILE>(roff[3]);
        n0 = stat_read<A>(soff); n1 = stat_read<A + 32>(soff); n2 = stat_read<A + 64>(soff); n3 = stat_read<A + 96>(soff);
        e0 = stat_read<A + 256>(soff); e1 = stat_read<A + 288>(soff); e2 = stat_read<A + 320>(soff); e3 = stat_read<A + 352>(soff);
    };
    auto tile = [&](auto a_tag, auto an_tag, int q0) {
        constexpr int A = decltype(a_tag)::value;
        // dO^T and Q^T fragments: [k-step s][row half dt][row group u]
        const bf16x4 u000 = tr_read<A + TILE>(toff[0][0]), u001 = tr_read<A + TILE>(toff[0][1]);
        const bf16x4 u010 = tr_read<A + TILE>(toff[1][0]), u011 = tr_read<A + TILE>(toff[1][1]);
        const bf16x4 w000 = tr_read<A>(toff[0][0]), w001 = tr_read<A>(toff[0][1]);
        const bf16x4 w010 = tr_read<A>(toff[1][0]), w011 = tr_read<A>(toff[1][1]);
        LGKM_WAIT(8);                                    // rows + stats (requested during the previous tile) have landed
        f32x16 nl, dp, sc;
#pragma unroll
        for (int j = 0; j < 4; ++j) { nl[j] = n0[j]; nl[4 + j] = n1[j]; nl[8 + j] = n2[j]; nl[12 + j] = n3[j];
                                      dp[j] = e0[j]; dp[4 + j] = e1[j]; dp[8 + j] = e2[j]; dp[12 + j] = e3[j]; }
        sc = MFMA32(qr0, kf[0], zero16); dp = MFMA32(or0, vf[0], dp);
        sc = MFMA32(qr1, kf[1], sc); dp = MFMA32(or1, vf[1], dp);
        sc = MFMA32(qr2, kf[2], sc); dp = MFMA32(or2, vf[2], dp);
        sc = MFMA32(qr3, kf[3], sc); dp = MFMA32(or3, vf[3], dp);
        const bf16x4 u100 = tr_read<A + TILE + 2048>(toff[0][0]), u101 = tr_read<A + TILE + 2048>(toff[0][1]);
        const bf16x4 u110 = tr_read<A + TILE + 2048>(toff[1][0]), u111 = tr_read<A + TILE + 2048>(toff[1][1]);
        const bf16x4 w100 = tr_read<A + 2048>(toff[0][0]), w101 = tr_read<A + 2048>(toff[0][1]);
        const bf16x4 w110 = tr_read<A + 2048>(toff[1][0]), w111 = tr_read<A + 2048>(toff[1][1]);
        if ((q0 < key0w + 32) || (q0 + 32 > S)) {       // the tile touches the diagonal or the sequence end (wave-uniform):
#pragma unroll                                          // exp2(-inf) = 0 kills the masked scores, no branch inside the element loop
            for (int i = 0; i < 16; ++i) {
                const int q_ = q0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (key > q_ || q_ >= S) sc[i] = -INFINITY;
            }
        }
#define PD(I_) { const float p_ = fexp2(fmaf(sc[I_], c2, nl[I_])); sc[I_] = p_; dp[I_] = p_ * dp[I_]; }
        PD(0) PD(1) PD(2) PD(3) PD(4) PD(5) PD(6) PD(7)
        const bf16x8 p0 = pack8f(sc, 0), d0 = pack8f(dp, 0);
        LGKM_WAIT(8);                                    // first k-step's transposed fragments
        dv0 = MFMA32(cat4(u000, u001), p0, dv0); SB(); PD(8) PD(9) SB();
        dk0 = MFMA32(cat4(w000, w001), d0, dk0); SB(); PD(10) PD(11) SB();
        dv1 = MFMA32(cat4(u010, u011), p0, dv1); SB(); PD(12) PD(13) SB();
        dk1 = MFMA32(cat4(w010, w011), d0, dk1); SB(); PD(14) PD(15) SB();
#undef PD
        const bf16x8 p1 = pack8f(sc, 1), d1 = pack8f(dp, 1);
        LGKM_WAIT(0);
        fetch(an_tag);                                   // next tile's rows + stats fly under the last four MFMAs
        dv0 = MFMA32(cat4(u100, u101), p1, dv0);
        dk0 = MFMA32(cat4(w100, w101), d1, dk0);
        dv1 = MFMA32(cat4(u110, u111), p1, dv1);
        dk1 = MFMA32(cat4(w110, w111), d1, dk1);
    };
#undef SB

    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, KSTAGE>;
    using K2 = std::integral_constant<int, 2 * KSTAGE>;
    using K3 = std::integral_constant<int, 3 * KSTAGE>;
    if (niter > 0) {
        issue(0, 0);
        // (also the K^T / V^T fragments loaded above: naming them in an asm statement makes the compiler place ITS wait for those
        //  loads here - left pending, it would put an s_waitcnt vmcnt(0) in front of the first MFMA of every tile)
        asm volatile("" :: "v"(kf[0]), "v"(kf[1]), "v"(kf[2]), "v"(kf[3]), "v"(vf[0]), "v"(vf[1]), "v"(vf[2]), "v"(vf[3]));
        VM_WAIT(0);
        __builtin_amdgcn_s_barrier();
        fetch(K0{});                                     // the first tile's rows + stats (later ones ride under the previous tile)
        if (niter > 1) issue(1, 1);
        if (niter > 2) issue(2, 2);
    }
    // One query block per step, FOUR ring stages: a step is only one 32 x 32 tile per wave (16 MFMAs, well under a
    // microsecond), so a block requested one step ahead would still be in flight when it is needed and every step would
    // wait out the rest of a memory latency.  Block it+3 is requested at step it; at the top of step it the wait leaves the
    // youngest request (block it+2: 4 LDS-DMA instructions per wave, plus the statistics row in waves 0-1) in flight and
    // only requires block it+1 to have landed.  The barrier publishes that and releases the stage of block it-1, into which
    // block it+3 goes.  A tile reads block it and, at its
    // end, prefetches its fragments from block it+1.
#define KBLOCK(IT, SC_, SN_, NNNIDX)                                                                                   \
    {                                                                                                                  \
        if (!AB_BAR) {                                                                                                 \
            if ((IT) + 2 < niter) { if (wave < 2) { VM_WAIT(5); } else { VM_WAIT(4); } } else { VM_WAIT(0); }          \
            __builtin_amdgcn_s_barrier();                                                                              \
            if ((IT) + 3 < niter) issue((IT) + 3, NNNIDX);                                                             \
        }                                                                                                              \
        tile(SC_{}, SN_{}, (kblk + (IT) % per_head) * 64 + 32 * qp);                                                   \
    }
    for (int it = 0; it < niter; it += 4) {
        KBLOCK(it, K0, K1, 3)
        if (it + 1 >= niter) break;
        KBLOCK(it + 1, K1, K2, 0)
        if (it + 2 >= niter) break;
        KBLOCK(it + 2, K2, K3, 1)
        if (it + 3 >= niter) break;
        KBLOCK(it + 3, K3, K0, 2)
    }
#undef KBLOCK

    // ---- reduce the two query-parity waves of each key half through LDS, then dK^T / dV^T -> [key][d] rows
    __builtin_amdgcn_s_barrier();
    {
        float* red = reinterpret_cast<float*>(smem) + kh * (64 * 64);    // [reg 0..63][lane] per key half
        if (qp == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                red[(i) * 64 + lane] = dk0[i]; red[(16 + i) * 64 + lane] = dk1[i];
                red[(32 + i) * 64 + lane] = dv0[i]; red[(48 + i) * 64 + lane] = dv1[i];
            }
        }
        __syncthreads();
        if (qp == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                dk0[i] += red[(i) * 64 + lane]; dk1[i] += red[(16 + i) * 64 + lane];
                dv0[i] += red[(32 + i) * 64 + lane]; dv1[i] += red[(48 + i) * 64 + lane];
            }
            // (the sums must be in registers BEFORE the barrier: the row staging below overwrites `red`, and hipcc may sink
            //  these LDS reads to their first use behind it - seen in attention64_asm.hip, round 4; an empty asm pins them)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(dk0[i]), "+v"(dk1[i]), "+v"(dv0[i]), "+v"(dv1[i]));
        }
        __syncthreads();
        if (qp == 0) {
            char* ob = smem + kh * (2 * 32 * 144);      // [K rows | V rows] of this key half, 144-B rows
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float a0 = dk0[4 * gq] * scale, a1 = dk0[4 * gq + 1] * scale, a2 = dk0[4 * gq + 2] * scale, a3 = dk0[4 * gq + 3] * scale;
                float b0 = dk1[4 * gq] * scale, b1 = dk1[4 * gq + 1] * scale, b2 = dk1[4 * gq + 2] * scale, b3 = dk1[4 * gq + 3] * scale;
                if (rope) {
                    unrope2(rope, kc, 64, 4 * gq + 2 * h, a0, a1, a2, a3);
                    unrope2(rope, kc, 64, 16 + 4 * gq + 2 * h, b0, b1, b2, b3);
                }
                uint2 w0, w1, x0, x1;
                w0.x = pack2bf(a0, a1); w0.y = pack2bf(a2, a3);
                w1.x = pack2bf(b0, b1); w1.y = pack2bf(b2, b3);
                x0.x = pack2bf(dv0[4 * gq], dv0[4 * gq + 1]); x0.y = pack2bf(dv0[4 * gq + 2], dv0[4 * gq + 3]);
                x1.x = pack2bf(dv1[4 * gq], dv1[4 * gq + 1]); x1.y = pack2bf(dv1[4 * gq + 2], dv1[4 * gq + 3]);
                *reinterpret_cast<uint2*>(ob + r * 144 + (8 * gq + 4 * h) * 2) = w0;
                *reinterpret_cast<uint2*>(ob + r * 144 + (32 + 8 * gq + 4 * h) * 2) = w1;
                *reinterpret_cast<uint2*>(ob + 32 * 144 + r * 144 + (8 * gq + 4 * h) * 2) = x0;
                *reinterpret_cast<uint2*>(ob + 32 * 144 + r * 144 + (32 + 8 * gq + 4 * h) * 2) = x1;
            }
            bf16_t* kp = dqkv + ((size_t)b * S) * ld + (H + kvh) * 64;
            bf16_t* vp = dqkv + ((size_t)b * S) * ld + (H + KV + kvh) * 64;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = 8 * pass + (lane >> 3);
                const U4 kv_ = *reinterpret_cast<const U4*>(ob + row * 144 + (lane & 7) * 16);
                const U4 vv_ = *reinterpret_cast<const U4*>(ob + 32 * 144 + row * 144 + (lane & 7) * 16);
                if (key0w + row < S) {
                    *reinterpret_cast<U4*>(kp + (size_t)(key0w + row) * ld + (lane & 7) * 8) = kv_;
                    *reinterpret_cast<U4*>(vp + (size_t)(key0w + row) * ld + (lane & 7) * 8) = vv_;
                }
            }
        }
    }
}

}  // namespace

int csm_attn64_fwd_launch(const void* qkv, void* out, float* lse, int B, int S, int H, int KV, hipStream_t stream) {
    const float scale = 0.125f;
    const int lds = NSTAGE * STAGE;
    dim3 grid((unsigned)(((S + 127) / 128) * H * B)), block(256);
    hipLaunchKernelGGL(attn64_fwd_kernel, grid, block, lds, stream, (const bf16_t*)qkv, (bf16_t*)out, lse, S, H, KV, scale);
    return 0;
}

int csm_attn64_dq_launch(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int S,
                         int H, int KV, const float* rope, hipStream_t stream) {
    const float scale = 0.125f;
    const int lds = NSTAGE * STAGE;
    dim3 grid((unsigned)(((S + 127) / 128) * H * B)), block(256);
    hipLaunchKernelGGL(attn64_dq_kernel, grid, block, lds, stream, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, delta,
                       (bf16_t*)dqkv, S, H, KV, scale, rope, (long long)B * H * S);
    return 0;
}

int csm_attn64_dkv_launch(const void* qkv, const void* dout, const float* lse, const float* delta, void* dqkv, int B, int S, int H,
                          int KV, const float* rope, hipStream_t stream) {
    const float scale = 0.125f;
    const int lds = KNSTAGE * KSTAGE;
    static bool done = false;        // more than 64 KiB of dynamic LDS must be requested once
    if (!done) { (void)hipFuncSetAttribute((const void*)attn64_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done = true; }
    dim3 grid((unsigned)(((S + 63) / 64) * KV * B)), block(256);
    // (delta = the dQ kernel's output: [0, BHS) = -delta, [BHS, 2 BHS) = -lse log2e)
    hipLaunchKernelGGL(attn64_dkv_kernel, grid, block, lds, stream, (const bf16_t*)qkv, (const bf16_t*)dout, delta + (size_t)B * H * S, delta,
                       (bf16_t*)dqkv, S, H, KV, scale, rope);
    return 0;
}
