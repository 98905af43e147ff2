// head_dim-64 causal GQA attention backward, dK / dV pass, third generation: ONE wave owns all 64 keys of a key block and walks
// its own stream of 32-query tiles; the loop is a generated, hand-allocated inline-asm block (tools/gen/gen_attn64_dkv_loop.py
// -> attn64_dkv_loop.inc; register map and software pipeline in that script's header).
//
// Same contract as attn64_dkv_kernel of attention64.hip (which stays for shapes this kernel does not take: S % 64 != 0 or a
// GQA group that is not 4 query heads, and as the A/B reference): replaces the backward of torchtune MultiHeadAttention ->
// F.scaled_dot_product_attention with the mask of reference src/csm/models/model.py:59-76 / src/csm/training/utils.py:90-91
// (positions arange(S) => plain causal).  Reads what the dQ kernel published (-delta and -lse * log2e per query).
//
// Workgroup = 64 keys of one (batch, kv head) = 4 waves, one per SIMD (the kernel takes all 512 registers); wave w takes query
// head w of the GQA group: its K^T / V^T operands live in AGPRs for the whole kernel, its Q / dO tiles arrive by LDS-DMA into a
// PRIVATE four-stage ring, so nothing in the loop is a rendezvous - every wait is a counted vmcnt / lgkmcnt on the wave's own
// requests.  The four heads' dK^T / dV^T are added through LDS after the loop in a fixed order (deterministic, no atomics).
//
// Built with -mllvm -amdgpu-spill-vgpr-to-agpr=0 and without any MFMA builtin: the compiler never touches an AGPR itself.
#include "common.h"
#include <string.h>
#include <type_traits>
#include "attn64_dkv_loop.inc"
#include "attn64_dq_loop.inc"

namespace {

constexpr int STAGE = CSM_A64_DKV_STAGE, NSTAGE = CSM_A64_DKV_NSTAGE;
constexpr int WAVE_LDS = STAGE * NSTAGE;             // 33792 B per wave, 135168 B per workgroup

__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

template <int N>
__device__ __forceinline__ float acc_read1() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "i"(N));
    return v;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

__global__ __launch_bounds__(256, 1) void attn64_dkv_asm_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                const float* __restrict__ stats /* [-delta | -lse log2e], nstat each */,
                                                                bf16_t* __restrict__ dqkv, int S, int H, int KV, float scale,
                                                                const float* __restrict__ rope, long long nstat,
                                                                unsigned c2 /* bits of scale * log2(e): the exp2 argument scale */,
                                                                int order /* 0: heaviest key blocks of an XCD's run first; 1: pair by pair */) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int nkblk = S / 64;
    int kblk, kvh, b;
    {   // XCD-aware 1-D grid: a contiguous run of (b, kv-head) pairs per XCD, the heaviest key blocks of the run first
        const int T = gridDim.x, id = blockIdx.x, xcd = id & 7, within = id >> 3;
        const int q8 = T >> 3, r8 = T & 7;
        const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
        int pair = nid / nkblk;
        kblk = nid % nkblk;
        const int run = (xcd < r8) ? q8 + 1 : q8, base = nid - within;
        if (order == 0 && run % nkblk == 0 && base % nkblk == 0) {
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
    const int hq = kvh * 4 + wave;                       // this wave's query head
    const int key0 = kblk * 64;
    const int nsteps = __builtin_amdgcn_readfirstlane(S / 32 - 2 * kblk);      // 32-query tiles at or below the block's diagonal (>= 2)

    const unsigned sbase = (unsigned)(uintptr_t)smem + (unsigned)wave * WAVE_LDS;
    const unsigned wbase = __builtin_amdgcn_readfirstlane(sbase);
    // lane addresses / offsets the loop derives everything else from (see the generator)
    const unsigned roff0 = sbase + r * 128 + ((h ^ swz(r)) << 4);
    unsigned toff00;
    {
        const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3;
        const int row = 4 * (g >> 1) + q4, ch = 2 * (g & 1) + (p >> 1);
        toff00 = sbase + row * 128 + ((ch ^ swz(row)) << 4) + (p & 1) * 8;
    }
    const unsigned soff = sbase + 16 * h;
    unsigned dq[2], dO[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = 8 * p + (lane >> 3);
        dq[p] = (unsigned)(row * ld + (((lane & 7) ^ swz(row)) << 3)) * 2u;
        dO[p] = (unsigned)(row * ldo + (((lane & 7) ^ swz(row)) << 3)) * 2u;
    }
    const unsigned ds = lane < 32 ? (unsigned)(nstat * 4 + 4 * lane) : (unsigned)(4 * (lane - 32));
    unsigned kvo[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) kvo[kh] = (unsigned)((key0 + 32 * kh + r) * ld + 8 * h) * 2u;
    const int m0v = r - 4 * h;
    const char* Qb = reinterpret_cast<const char*>(qkv + ((size_t)b * S + key0) * ld + hq * 64);
    const char* Ob = reinterpret_cast<const char*>(dout + ((size_t)b * S + key0) * ldo + hq * 64);
    const char* Sb = reinterpret_cast<const char*>(stats + ((size_t)b * H + hq) * S + key0);
    const char* Kb = reinterpret_cast<const char*>(qkv + (size_t)b * S * ld + (H + kvh) * 64);
    const char* Vb = reinterpret_cast<const char*>(qkv + (size_t)b * S * ld + (H + KV + kvh) * 64);
    const unsigned qstep = 32u * ld * 2u, ostep = 32u * ldo * 2u;

    asm volatile(CSM_A64_DKV_LOOP
                 ::"v"(roff0), "v"(toff00), "v"(soff), "v"(dq[0]), "v"(dq[1]), "v"(dO[0]), "v"(dO[1]), "v"(ds), "v"(kvo[0]), "v"(kvo[1]), "v"(m0v),
                   "s"(Qb), "s"(Ob), "s"(Sb), "s"(Kb), "s"(Vb), "s"(nsteps), "s"(wbase), "s"(qstep), "s"(ostep), "s"(c2)
                 : CSM_A64_DKV_CLOBBERS);

    // ---- the four heads' accumulators -> LDS [wave][register][lane] (fp32), summed in the fixed order 0 + 1 + 2 + 3
    __syncthreads();                                     // every wave is done with its ring
    float* red = reinterpret_cast<float*>(smem);
    static_for<0, 128>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        red[(wave * 128 + i) * 64 + lane] = acc_read1<i>();
    });
    __syncthreads();
    // wave w finishes one (tensor, key half): w >> 1 = 0: dK, 1: dV; w & 1 = key half.  Tile (dt, kh) of dK is a[16 (2 dt + kh)],
    // of dV a[64 + 16 (2 dt + kh)]
    const int which = wave >> 1, kh = wave & 1;
    float t0[16], t1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r0 = 64 * which + 16 * kh + i, r1 = 64 * which + 16 * (2 + kh) + i;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s0 += red[(w * 128 + r0) * 64 + lane]; s1 += red[(w * 128 + r1) * 64 + lane]; }
        t0[i] = s0; t1[i] = s1;
    }
    // The sums must BE in registers before the barrier: the row staging below overwrites the partial sums, and hipcc otherwise
    // sinks these LDS reads to their first use, behind the barrier (float reads vs uint2 writes do not alias for it) - a wave
    // that waits for its RoPE table entries then sums what a faster wave has already overwritten (found as run-to-run
    // differences in dK with the fused RoPE^T only).  An empty asm that consumes the values pins the reads here.
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(t0[i]), "+v"(t1[i]));
    __syncthreads();                                     // all sums are in registers: the LDS is free for the row staging
    {
        const int keyr = key0 + 32 * kh + r;             // this lane's key (column of the transposed tiles) = its position
        char* ob = smem + wave * (32 * 144);
        const bool unrope = which == 0 && rope != nullptr;
        float4 ta[4], tb[4];                              // RoPE table entries (c0, s0, c1, s1) of this lane's 8 pairs, fetched up front
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            ta[gq] = tb[gq] = make_float4(1.f, 0.f, 1.f, 0.f);
            if (unrope) {
                const float* tr = rope + ((size_t)keyr * 32 + 4 * gq + 2 * h) * 2;
                ta[gq] = *reinterpret_cast<const float4*>(tr);
                tb[gq] = *reinterpret_cast<const float4*>(tr + 32);
            }
        }
        const float mul = which == 0 ? scale : 1.f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float a0 = t0[4 * gq] * mul, a1 = t0[4 * gq + 1] * mul, a2 = t0[4 * gq + 2] * mul, a3 = t0[4 * gq + 3] * mul;
            float b0 = t1[4 * gq] * mul, b1 = t1[4 * gq + 1] * mul, b2 = t1[4 * gq + 2] * mul, b3 = t1[4 * gq + 3] * mul;
            if (unrope) {                                 // backward of the interleaved-pair rotation (attention.hip: unrope2)
                const float4 t = ta[gq], u = tb[gq];
                rope_rot(a0, a1, t.x, -t.y); rope_rot(a2, a3, t.z, -t.w);
                rope_rot(b0, b1, u.x, -u.y); rope_rot(b2, b3, u.z, -u.w);
            }
            uint2 w0, w1;
            w0.x = pack2bf(a0, a1); w0.y = pack2bf(a2, a3);
            w1.x = pack2bf(b0, b1); w1.y = pack2bf(b2, b3);
            *reinterpret_cast<uint2*>(ob + r * 144 + (8 * gq + 4 * h) * 2) = w0;           // d = 8 gq + 4 h .. + 3
            *reinterpret_cast<uint2*>(ob + r * 144 + (32 + 8 * gq + 4 * h) * 2) = w1;      // d = 32 + ...
        }
        // (wave-private LDS region: a wave's own LDS operations execute in order)
        bf16_t* dst = dqkv + ((size_t)b * S + key0 + 32 * kh) * ld + (which == 0 ? (H + kvh) : (H + KV + kvh)) * 64;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = 8 * pass + (lane >> 3);
            const U4 v = *reinterpret_cast<const U4*>(ob + row * 144 + (lane & 7) * 16);
            *reinterpret_cast<U4*>(dst + (size_t)row * ld + (lane & 7) * 8) = v;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// dQ pass, same construction (tools/gen/gen_attn64_dq_loop.py): one wave = 128 queries (four 32-query tiles, the query on the
// lane) of one query head, its Q^T / dO^T operands in AGPRs, walking the 32-key tiles from the diagonal down to key 0.  The four
// waves of a workgroup are the four query heads of one GQA group on the same 128-query block: they consume the same K / V tiles,
// so the workgroup shares ONE eight-stage ring, each wave fetching a quarter of every tile, one barrier per tile.  Persistent:
// a workgroup keeps its (batch, kv head) pair and walks query blocks in a serpentine over the rounds, so that every workgroup's
// total stream length is the same.  Also computes delta = rowsum(dO o O) and publishes -delta and -lse * log2(e) for the dK/dV
// pass, and applies the RoPE backward in the epilogue (rope != null) - as attn64_dq_kernel does.
constexpr int QNQ = CSM_A64_DQ_NQ;
constexpr int QRING = CSM_A64_DQ_STAGE * CSM_A64_DQ_NSTAGE;      // the workgroup's K / V ring: 64 KiB
constexpr int QSTAGING = 2 * CSM_A64_DQ_SG_TILE;               // per wave, behind the ring: the Q / dO / O rows of 64 queries at a time
                                                                // on the way in, its 32-query dQ tiles as [q][d] rows on the way out

__device__ __forceinline__ float halves_sum(float x) {          // lanes l and l ^ 32 hold the same query
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

__global__ __launch_bounds__(256, 1) void attn64_dq_asm_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                               const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                               float* __restrict__ stats, bf16_t* __restrict__ dqkv, int S, int H, int KV,
                                                               float scale, const float* __restrict__ rope, long long nstat, unsigned c2,
                                                               int levels /* query blocks per round of the persistent schedule */,
                                                               unsigned long long* __restrict__ dbg /* tools/probes: cycle stamps, or null */) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int nqblk = S / (32 * QNQ);
    // Persistent schedule: workgroup w = (pair = w % P, level = w / P) keeps its (batch, kv head) pair - its XCD keeps that pair's
    // K / V in L2 - and walks the query blocks nq-1-(L r + level') for r = 0, 1, ...: level' = level in even rounds, L-1-level in
    // odd ones, so every workgroup gets the same total stream length.
    const int P = gridDim.x / levels;
    const int pair = blockIdx.x % P, level = blockIdx.x / P;
    const int kvh = pair % KV, b = pair / KV;
    const int ld = (H + 2 * KV) * 64, ldo = H * 64;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hq = kvh * 4 + wave;
  for (int round = 0; round * levels < nqblk; ++round) {
    const int jb = nqblk - 1 - (round * levels + ((round & 1) ? levels - 1 - level : level));
    if (jb < 0) continue;                                 // (uniform over the workgroup)
    if (round) __syncthreads();                           // the ring is free again: every wave has left the previous block's loop
    unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0, sta = 0, stb = 0;
    if (dbg) { st0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    const int q0 = jb * (32 * QNQ);
    const int ntiles = __builtin_amdgcn_readfirstlane(QNQ * jb + QNQ);         // 32-key tiles at or below the block's diagonal

    const unsigned sbase = (unsigned)(uintptr_t)smem;                            // the ring is the workgroup's
    const unsigned wbase = __builtin_amdgcn_readfirstlane(sbase + (unsigned)wave * 1024u);   // this wave writes piece `wave` of every image
    const unsigned roff0 = sbase + r * 128 + ((h ^ swz(r)) << 4);
    unsigned toff00;
    {
        const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3;
        const int row = 4 * (g >> 1) + q4, ch = 2 * (g & 1) + (p >> 1);
        toff00 = sbase + row * 128 + ((ch ^ swz(row)) << 4) + (p & 1) * 8;
    }
    unsigned dk0;
    {
        const int row = 8 * wave + (lane >> 3);                                // rows 8 wave .. 8 wave + 7 of a tile
        dk0 = (unsigned)(row * ld + (((lane & 7) ^ swz(row)) << 3)) * 2u;
    }
    const int m0v = r - 4 * h;
    const char* Kb = reinterpret_cast<const char*>(qkv + (size_t)b * S * ld + (H + kvh) * 64);
    const char* Vb = reinterpret_cast<const char*>(qkv + (size_t)b * S * ld + (H + KV + kvh) * 64);
    // the block's first Q / dO / O row (the prologue's LDS-DMA descriptors)
    const char* Qb = reinterpret_cast<const char*>(qkv + ((size_t)b * S + q0) * ld + hq * 64);
    const char* Ob = reinterpret_cast<const char*>(dout + ((size_t)b * S + q0) * ldo + hq * 64);
    const char* Pb = reinterpret_cast<const char*>(out + ((size_t)b * S + q0) * ldo + hq * 64);
    const unsigned kstep = 32u * ld * 2u, ostep = 32u * ldo * 2u;
    // staging area of this wave (behind the ring) and its lane addresses / LDS-DMA lane offsets
    const unsigned sgb = __builtin_amdgcn_readfirstlane(sbase + QRING + (unsigned)wave * QSTAGING);
    const unsigned sgoff = sgb + r * 128 + ((h ^ swz(r)) << 4);
    unsigned dqo[2], doo[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = 8 * p + (lane >> 3);
        dqo[p] = (unsigned)(row * ld + (((lane & 7) ^ swz(row)) << 3)) * 2u;
        doo[p] = (unsigned)(row * ldo + (((lane & 7) ^ swz(row)) << 3)) * 2u;
    }
    float nd[QNQ] = {0.f, 0.f, 0.f, 0.f}, nl[QNQ] = {0.f, 0.f, 0.f, 0.f};
#define CSM_A64_DQ_OPERANDS                                                                                              \
    ::"v"(roff0), "v"(toff00), "v"(dk0), "v"(sgoff), "v"(dqo[0]), "v"(nd[0]), "v"(nd[1]), "v"(nd[2]), "v"(nd[3]), "v"(nl[0]), "v"(nl[1]),  \
        "v"(nl[2]), "v"(nl[3]), "v"(m0v), "s"(Kb), "s"(Vb), "s"(Qb), "s"(Ob), "s"(ntiles), "s"(wbase), "s"(kstep), "s"(c2), "s"(ostep),   \
        "v"(dqo[1]), "v"(doo[0]), "v"(doo[1]), "s"(Pb), "s"(sgb)
    // delta = rowsum(dO o O) and the exp2 addend of the lane's queries of tiles 2 half, 2 half + 1, from the staged rows;
    // published for the dK/dV pass
    auto delta_half = [&](int half) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int qt = 2 * half + j;
            const int q = q0 + 32 * qt + r;
            const char* img = smem + (sgb - sbase) + j * CSM_A64_DQ_SG_TILE + r * 128;
            float dsum = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = ((2 * ks + h) ^ swz(r)) << 4;
                const bf16x8 dof = *reinterpret_cast<const bf16x8*>(img + 4096 + ch);
                const bf16x8 of = *reinterpret_cast<const bf16x8*>(img + 8192 + ch);
#pragma unroll
                for (int e = 0; e < 8; ++e) dsum += bf2f((bf16_t)dof[e]) * bf2f((bf16_t)of[e]);
            }
            const float d_ = -halves_sum(dsum);
            const float l_ = -lse[((size_t)b * H + hq) * S + q] * 1.4426950408889634f;
            if (qt == 0) { nd[0] = d_; nl[0] = l_; } else if (qt == 1) { nd[1] = d_; nl[1] = l_; }
            else if (qt == 2) { nd[2] = d_; nl[2] = l_; } else { nd[3] = d_; nl[3] = l_; }
            if (h == 0) {
                stats[((size_t)b * H + hq) * S + q] = d_;
                stats[(size_t)nstat + ((size_t)b * H + hq) * S + q] = l_;
            }
        }
    };
    // Prologue: the wave's Q / dO / O rows arrive as whole 128-B rows by LDS-DMA, 64 queries at a time (the first version's
    // per-lane fragment loads - 32 rows per request, a quarter of each line used - took 11k cycles to issue, and as many again
    // for the delta computation's); this wave's pieces of the first key tiles ride with the first half.
    asm volatile(CSM_A64_DQ_STAGE0 CSM_A64_DQ_OPERANDS : CSM_A64_DQ_PRO_CLOBBERS);
    if (dbg) sta = __builtin_readcyclecounter();
    asm volatile(CSM_A64_DQ_LAND0 CSM_A64_DQ_OPERANDS : CSM_A64_DQ_PRO_CLOBBERS);
    delta_half(0);
    asm volatile(CSM_A64_DQ_STAGE1 CSM_A64_DQ_OPERANDS : CSM_A64_DQ_PRO_CLOBBERS);
    asm volatile(CSM_A64_DQ_LAND1 CSM_A64_DQ_OPERANDS : CSM_A64_DQ_PRO_CLOBBERS);
    delta_half(1);
    if (dbg) { stb = __builtin_readcyclecounter(); st1 = stb; }
    asm volatile(CSM_A64_DQ_LOOP CSM_A64_DQ_OPERANDS : CSM_A64_DQ_CLOBBERS);
#undef CSM_A64_DQ_OPERANDS
    if (dbg) st2 = __builtin_readcyclecounter();

    // ---- epilogue: dQ^T registers (x 1/sqrt(hd), RoPE^T) -> LDS [q][d] (wave-private) -> whole 128-B rows of the q block of dqkv
    char* ob = smem + QRING + wave * QSTAGING;           // (the wave's staging area behind the ring: slower waves may still be reading tiles)
    static_for<0, QNQ>([&](auto qt_) {
        constexpr int qt = decltype(qt_)::value;
        float t0[16], t1[16];
        static_for<0, 16>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            t0[i] = acc_read1<16 * qt + i>() * scale;              // tile (dt = 0, qt)
            t1[i] = acc_read1<16 * (QNQ + qt) + i>() * scale;      // tile (dt = 1, qt)
        });
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float a0 = t0[4 * gq], a1 = t0[4 * gq + 1], a2 = t0[4 * gq + 2], a3 = t0[4 * gq + 3];
            float b0 = t1[4 * gq], b1 = t1[4 * gq + 1], b2 = t1[4 * gq + 2], b3 = t1[4 * gq + 3];
            if (rope) {
                const float* tr = rope + ((size_t)(q0 + 32 * qt + r) * 32 + 4 * gq + 2 * h) * 2;
                const float4 t = *reinterpret_cast<const float4*>(tr), u = *reinterpret_cast<const float4*>(tr + 32);
                rope_rot(a0, a1, t.x, -t.y); rope_rot(a2, a3, t.z, -t.w);
                rope_rot(b0, b1, u.x, -u.y); rope_rot(b2, b3, u.z, -u.w);
            }
            uint2 w0, w1;
            w0.x = pack2bf(a0, a1); w0.y = pack2bf(a2, a3);
            w1.x = pack2bf(b0, b1); w1.y = pack2bf(b2, b3);
            *reinterpret_cast<uint2*>(ob + qt * 4608 + r * 144 + (8 * gq + 4 * h) * 2) = w0;
            *reinterpret_cast<uint2*>(ob + qt * 4608 + r * 144 + (32 + 8 * gq + 4 * h) * 2) = w1;
        }
        // (wave-private LDS region: a wave's own LDS operations execute in order)
        bf16_t* dst = dqkv + ((size_t)b * S + q0 + 32 * qt) * ld + hq * 64;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = 8 * pass + (lane >> 3);
            const U4 v = *reinterpret_cast<const U4*>(ob + qt * 4608 + row * 144 + (lane & 7) * 16);
            *reinterpret_cast<U4*>(dst + (size_t)row * ld + (lane & 7) * 8) = v;
        }
    });
    if (dbg && lane == 0) {                              // per (workgroup, wave, round): stream length, cycle stamps, 100 MHz stamps
        unsigned long long* d = dbg + (((size_t)blockIdx.x * 4 + wave) * 8 + round) * 8;
        d[0] = ntiles; d[1] = st0; d[2] = st1; d[3] = st2; d[4] = __builtin_readcyclecounter(); d[5] = rt0; d[6] = __builtin_amdgcn_s_memrealtime();
        d[7] = ((sta - st0) << 32) | (stb - st0);
    }
  }
}

}  // namespace

// 1 = taken, 0 = shape not supported (the caller falls back to attention64.hip's kernel)
int g_attn64_dkv_asm_order = 0;       // csm_set_attn_variant bit 11
int csm_attn64_dkv_asm_launch(const void* qkv, const void* dout, const float* stats, void* dqkv, int B, int S, int H, int KV,
                              const float* rope, hipStream_t stream) {
    if (S % 64 != 0 || S < 64 || H != 4 * KV) return 0;
    const float scale = 0.125f, c2 = scale * 1.4426950408889634f;
    unsigned c2bits;
    memcpy(&c2bits, &c2, 4);
    const int lds = 4 * WAVE_LDS;
    static bool done = false;        // more than 64 KiB of dynamic LDS must be requested once
    if (!done) { (void)hipFuncSetAttribute((const void*)attn64_dkv_asm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done = true; }
    dim3 grid((unsigned)((S / 64) * KV * B)), block(256);
    hipLaunchKernelGGL(attn64_dkv_asm_kernel, grid, block, lds, stream, (const bf16_t*)qkv, (const bf16_t*)dout, stats, (bf16_t*)dqkv, S, H,
                       KV, scale, rope, (long long)B * H * S, c2bits, g_attn64_dkv_asm_order);
    return 1;
}

int g_attn64_dq_asm_order = 0;
unsigned long long* g_attn64_dq_dbg = nullptr;       // csm_attn64_set_debug (tools/probes)
extern "C" int csm_attn64_set_debug(void* p) { g_attn64_dq_dbg = (unsigned long long*)p; return 0; }
// 1 = taken, 0 = shape not supported (the caller falls back to attention64.hip's dQ kernel)
int csm_attn64_dq_asm_launch(const void* qkv, const void* out, const void* dout, const float* lse, float* stats, void* dqkv, int B, int S,
                             int H, int KV, const float* rope, hipStream_t stream) {
    if (S % (32 * QNQ) != 0 || S < 32 * QNQ || H != 4 * KV) return 0;
    const float scale = 0.125f, c2 = scale * 1.4426950408889634f;
    unsigned c2bits;
    memcpy(&c2bits, &c2, 4);
    const int lds = QRING + 4 * QSTAGING;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)attn64_dq_asm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done = true; }
    // query blocks per round: enough workgroups for every CU (256) when the batch allows, each then walking nq / levels blocks
    const int P = KV * B, nq = S / (32 * QNQ);
    int levels = g_attn64_dq_asm_order ? nq : (256 + P - 1) / P;
    if (levels > nq) levels = nq;
    if (levels < 1) levels = 1;
    dim3 grid((unsigned)(P * levels)), block(256);
    hipLaunchKernelGGL(attn64_dq_asm_kernel, grid, block, lds, stream, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, stats,
                       (bf16_t*)dqkv, S, H, KV, scale, rope, (long long)B * H * S, c2bits, levels, g_attn64_dq_dbg);
    return 1;
}
