// Generation-side kernels for gfx950: top-k / temperature sampler with injected Exp(1) noise, and the Mimi
// split residual vector quantiser (nearest-codeword encode, codebook-sum decode).
#include "common.h"
#include <math.h>

namespace {

// Probe builds only (-DCSM_DECODE_STAMPS, tools/probes/decode_stamps.sh): the first and the last workgroup of a stamped kernel
// record the 100 MHz wall clock at a few points; the product build compiles none of this.
#ifdef CSM_DECODE_STAMPS
__device__ unsigned long long g_decode_stamps[8192];
__device__ unsigned int g_decode_nstamp;
#define STAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); st_[i] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_FLUSH(id) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {                          \
        const unsigned int s0 = atomicAdd(&g_decode_nstamp, 10u);                                                                   \
        if (s0 + 10 <= 8192) { g_decode_stamps[s0] = (id); g_decode_stamps[s0 + 1] = blockIdx.x;                                    \
            for (int q_ = 0; q_ < 8; ++q_) g_decode_stamps[s0 + 2 + q_] = st_[q_]; } } } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH(id)
#endif

struct ValIdx { float v; int i; };

__device__ __forceinline__ ValIdx vi_max(ValIdx a, ValIdx b) {  // larger value wins, then the lower index
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ValIdx vi_min(ValIdx a, ValIdx b) {  // smaller value wins, then the lower index
    return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

template <bool MAX>
__device__ __forceinline__ ValIdx block_arg(ValIdx x, ValIdx* red) {
#define CSM_ARG_STEP(o) { ValIdx y; y.v = lane_xor<o>(x.v); y.i = lane_xor<o>(x.i); x = MAX ? vi_max(x, y) : vi_min(x, y); }
    CSM_ARG_STEP(32) CSM_ARG_STEP(16) CSM_ARG_STEP(8) CSM_ARG_STEP(4) CSM_ARG_STEP(2) CSM_ARG_STEP(1)
#undef CSM_ARG_STEP
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = x;
    __syncthreads();
    ValIdx r = red[0];
    for (int k = 1; k < nw; ++k) r = MAX ? vi_max(r, red[k]) : vi_min(r, red[k]);
    return r;
}

// reference src/csm/models/model.py:79-96: logits/T, keep values >= k-th largest, log_softmax -> softmax,
// argmax(p / q) with q ~ Exp(1) supplied by the caller.  One 256-thread block per row, V <= 256 * NPT.
// A generated frame draws 32 codes one after the other, so this kernel's latency is paid 32 times per 80 ms of audio; it runs as
// ONE workgroup on a cold CU, and what it costs is latency, not work (tools/probes/decode_stamps.py, round 4: of 15 us, 4-5
// were the Exp(1) row arriving from HBM right before the last step, 5-6 the 9-16 rounds of the bisection that found the k-th
// largest value, 2-3 the logits' own arrival).  So:
//  * the Exp(1) row is requested first of all and consumed last;
//  * the k-th largest value comes from an 8-bit radix select on an order-preserving integer image of the floats: per pass the
//    keys that still match the prefix are counted into a 256-bin LDS histogram (ds_add_u32; four histograms, zeroed once, so a
//    pass needs ONE barrier), every wave then scans the bins from the top (4 bins per lane, a DPP row scan + three row totals, no
//    LDS crossbar) and reads the digit, the count above it and the count inside it off the lane that crosses k.  At most four
//    passes; a pass whose digit leaves exactly the wanted number of keys at or above it ends the search (the kept set is decided:
//    typical logits separate their 50th and 51st value in the second pass).  Rounds 2-3 used a bisection with ballots: 2 bits per
//    round, 9-16 rounds of ~0.35 us.
// The kept set is `key >= threshold` on the integer image - the set `value >= k-th largest` keeps (ties included).
constexpr int SMP_PER_THREAD = 16;
__device__ __forceinline__ int dpp_row_shr_add(int v, int ctrl_n) {   // v + (v shifted right by n lanes inside its row of 16, zero fill)
    switch (ctrl_n) {
        case 1: return v + __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
        case 2: return v + __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
        case 4: return v + __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
        default: return v + __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    }
}
template <int NPT>
__global__ __launch_bounds__(256) void sample_topk_kernel(const float* __restrict__ logits, const float* __restrict__ q,
                                                          int* __restrict__ out, int V, int ldl, int topk, float temperature) {
    __shared__ ValIdx red[4];
    __shared__ float fred[16];
    __shared__ __attribute__((aligned(16))) uint32_t hist[4][256];
    const int row = blockIdx.x;
    const float* x = logits + (size_t)row * ldl;
    const float* qq = q + (size_t)row * V;
    const int lane = threadIdx.x & 63;
    STAMP_DECL;
    STAMP(0);
    float qv[NPT], val[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int c = threadIdx.x + 256 * j;
        val[j] = c < V ? x[c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int c = threadIdx.x + 256 * j;
        qv[j] = c < V ? qq[c] : 1.f;
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) hist[ps][threadIdx.x] = 0u;
    uint32_t key[NPT];
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int c = threadIdx.x + 256 * j;
        val[j] = c < V ? val[j] / temperature : -INFINITY;
        const uint32_t u = __float_as_uint(val[j]);
        key[j] = c >= V ? 0u : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
        tmax = fmaxf(tmax, val[j]);
    }
    STAMP(1);
    __syncthreads();                                        // the zeroed histograms
    STAMP(2);
    uint32_t prefix = 0u;
    int krem = topk;                                        // keys still wanted among those that match the prefix
#pragma unroll 1
    for (int ps = 0; ps < 4; ++ps) {
        const int shift = 24 - 8 * ps;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int c = threadIdx.x + 256 * j;
            const bool in = c < V && (ps == 0 || (key[j] >> (shift + 8)) == (prefix >> (shift + 8)));
            if (in) atomicAdd(&hist[ps][(key[j] >> shift) & 255u], 1u);
        }
        __syncthreads();
        // lane l owns bins 252 - 4l .. 255 - 4l; the scan runs from the top bin down (every wave does it: no second barrier)
        const uint4 hv = *reinterpret_cast<const uint4*>(&hist[ps][252 - 4 * lane]);
        const int b3 = (int)hv.w, b2 = (int)hv.z, b1 = (int)hv.y, b0 = (int)hv.x;      // bins 255-4l, 254-4l, 253-4l, 252-4l
        const int mine = b3 + b2 + b1 + b0;
        int pre = mine;
        pre = dpp_row_shr_add(pre, 1); pre = dpp_row_shr_add(pre, 2); pre = dpp_row_shr_add(pre, 4); pre = dpp_row_shr_add(pre, 8);
        const int t0 = __builtin_amdgcn_readlane(pre, 15), t1 = __builtin_amdgcn_readlane(pre, 31), t2 = __builtin_amdgcn_readlane(pre, 47);
        pre += (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);     // keys in my bins and every bin above them
        int above = pre - mine, digit, inside;
        if (above + b3 >= krem) { digit = 255 - 4 * lane; inside = b3; }
        else {
            above += b3;
            if (above + b2 >= krem) { digit = 254 - 4 * lane; inside = b2; }
            else {
                above += b2;
                if (above + b1 >= krem) { digit = 253 - 4 * lane; inside = b1; }
                else { above += b1; digit = 252 - 4 * lane; inside = b0; }
            }
        }
        const unsigned long long cross = __ballot(pre >= krem);
        const int L = cross ? __ffsll((long long)cross) - 1 : 63;                       // (never empty: the matching keys number >= krem)
        digit = __builtin_amdgcn_readlane(digit, L);
        above = __builtin_amdgcn_readlane(above, L);
        inside = __builtin_amdgcn_readlane(inside, L);
        prefix |= (uint32_t)digit << shift;
        const bool exact = above + inside == krem;          // exactly k keys at or above this prefix: the kept set is decided
        krem -= above;
        if (exact) break;                                   // (block-uniform: every wave read the same histogram)
    }
    STAMP(3);
    // ---- the kept values (k of them, more only with ties at the threshold) are compacted - in index order of (thread, slot), by a
    //      block-wide prefix sum of the per-thread counts: deterministic - and ONE wave evaluates log_softmax -> softmax -> p / q ->
    //      argmax on them: two wave butterflies instead of seven block-wide reductions with their fourteen barriers, and 1 exp per
    //      kept value instead of 9 per thread.  Tokens outside the kept set have p = 0 and can never win the race (the largest
    //      kept p is >= 1 / count).  More than 64 kept values (ties, or topk > 64): the block-wide form below.
    __shared__ float cval[64], cq[64];
    __shared__ int cidx[64];
    __shared__ int wtot[4];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < NPT; ++j) mine += (threadIdx.x + 256 * j < V && key[j] >= prefix) ? 1 : 0;
    int pre = mine;
    pre = dpp_row_shr_add(pre, 1); pre = dpp_row_shr_add(pre, 2); pre = dpp_row_shr_add(pre, 4); pre = dpp_row_shr_add(pre, 8);
    {
        const int t0 = __builtin_amdgcn_readlane(pre, 15), t1 = __builtin_amdgcn_readlane(pre, 31), t2 = __builtin_amdgcn_readlane(pre, 47);
        pre += (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
    }
    if (lane == 63) wtot[threadIdx.x >> 6] = pre;
    __syncthreads();
    const int w_ = threadIdx.x >> 6;
    const int before = (w_ > 0 ? wtot[0] : 0) + (w_ > 1 ? wtot[1] : 0) + (w_ > 2 ? wtot[2] : 0);
    const int total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    if (total <= 64) {                                      // (block-uniform)
        int at = before + pre - mine;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int c = threadIdx.x + 256 * j;
            if (c < V && key[j] >= prefix) { cval[at] = val[j]; cq[at] = qv[j]; cidx[at] = c; ++at; }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const bool on = lane < total;
            const float v = on ? cval[lane] : -INFINITY;
            const float top = wave_max(v);
            const float e1 = on ? expf(v - top) : 0.f;
            const float logsum = logf(wave_sum(e1));
            const float ymax = (top - top) - logsum;
            const float e2 = on ? expf(((v - top) - logsum) - ymax) : 0.f;
            const float s2 = wave_sum(e2);
            ValIdx best = {-INFINITY, 0x7fffffff};
            if (on) best = (ValIdx){(e2 / s2) / cq[lane], cidx[lane]};
#define CSM_ARG_STEP(o) { ValIdx y; y.v = lane_xor<o>(best.v); y.i = lane_xor<o>(best.i); best = vi_max(best, y); }
            CSM_ARG_STEP(32) CSM_ARG_STEP(16) CSM_ARG_STEP(8) CSM_ARG_STEP(4) CSM_ARG_STEP(2) CSM_ARG_STEP(1)
#undef CSM_ARG_STEP
            if (lane == 0) out[row] = best.i;
        }
        STAMP(6);
        STAMP_FLUSH(300);
        return;
    }
    const float top = block_max(tmax, fred);
    // log_softmax over kept values, then softmax of that (torch evaluates both)
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < V && key[j] >= prefix) s += expf(val[j] - top);
    }
    s = block_sum(s, fred);
    const float logsum = logf(s);
    const float ymax = (top - top) - logsum;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < V && key[j] >= prefix) s2 += expf(((val[j] - top) - logsum) - ymax);
    }
    s2 = block_sum(s2, fred);
    STAMP(4);
    ValIdx best = {-INFINITY, 0x7fffffff};
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < V) {
            float p = 0.f;
            if (key[j] >= prefix) p = expf(((val[j] - top) - logsum) - ymax) / s2;
            best = vi_max(best, (ValIdx){p / qv[j], c});
        }
    }
    STAMP(5);
    best = block_arg<true>(best, red);
    if (threadIdx.x == 0) out[row] = best.i;
    STAMP(6);
    STAMP_FLUSH(300);
}

// Mimi split RVQ encode (moshi 0.2.2; call site reference src/csm/generator.py:117).  One block per frame; the
// residual lives in registers (4 floats per lane, replicated per wave); each wave scans C/4 codewords with
// coalesced 1-KiB row reads; squared L2 in fp32; the lowest index wins ties.
template <int D>
__global__ __launch_bounds__(256) void rvq_encode_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                         long long* __restrict__ codes, int T, int K, int C, int n_sem) {
    __shared__ ValIdx red[4];
    constexpr int PL = D / 64;
    const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float r[PL];
    for (int k = 0; k < K; ++k) {
        if (k == 0 || k == n_sem) {
#pragma unroll
            for (int j = 0; j < PL; ++j) r[j] = x[(size_t)t * D + lane + 64 * j];
        }
        const float* book = cb + (size_t)k * C * D;
        ValIdx best = {INFINITY, 0x7fffffff};
        for (int c = wave; c < C; c += 4) {
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < PL; ++j) {
                const float e = r[j] - book[(size_t)c * D + lane + 64 * j];
                d += e * e;
            }
            d = wave_sum(d);
            best = vi_min(best, (ValIdx){d, c});
        }
        best = block_arg<false>(best, red);
        if (threadIdx.x == 0) codes[(size_t)k * T + t] = best.i;
#pragma unroll
        for (int j = 0; j < PL; ++j) r[j] -= book[(size_t)best.i * D + lane + 64 * j];
    }
}

// codes [K][T] -> out[t][:] = sum_k cb[k][codes[k][t]] (ascending k, fp32)
__global__ __launch_bounds__(256) void rvq_decode_kernel(const long long* __restrict__ codes, const float* __restrict__ cb,
                                                         float* __restrict__ out, int T, int K, int C, int D) {
    const int t = blockIdx.x;
    for (int col = threadIdx.x; col < D; col += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += cb[((size_t)k * C + codes[(size_t)k * T + t]) * D + col];
        out[(size_t)t * D + col] = acc;
    }
}

}  // namespace

extern "C" int csm_sample_topk(const float* logits, const float* q, int* out, int rows, int V, int ldl, int topk,
                               float temperature, hipStream_t stream) {
    CSM_REQUIRE(logits && q && out && rows > 0 && V > 0 && ldl >= V, "csm_sample_topk: bad arguments");
    CSM_REQUIRE(V <= 256 * SMP_PER_THREAD, "csm_sample_topk: V=%d exceeds %d", V, 256 * SMP_PER_THREAD);
    CSM_REQUIRE(topk > 0 && topk <= V && temperature > 0.f, "csm_sample_topk: bad topk=%d / temperature=%f", topk, temperature);
    if (V <= 256 * 9) hipLaunchKernelGGL(sample_topk_kernel<9>, dim3(rows), dim3(256), 0, stream, logits, q, out, V, ldl, topk, temperature);
    else hipLaunchKernelGGL(sample_topk_kernel<SMP_PER_THREAD>, dim3(rows), dim3(256), 0, stream, logits, q, out, V, ldl, topk, temperature);
    CSM_CHECK_LAUNCH("csm_sample_topk");
    return 0;
}

extern "C" int csm_rvq_encode(const float* x, const float* codebooks, long long* codes, int T, int K, int C, int D,
                              int n_semantic, hipStream_t stream) {
    CSM_REQUIRE(x && codebooks && codes && T > 0 && K > 0 && C > 0, "csm_rvq_encode: bad arguments");
    CSM_REQUIRE(D == 256 || D == 128 || D == 64, "csm_rvq_encode: codebook dim %d unsupported (64/128/256)", D);
    CSM_REQUIRE(n_semantic >= 0 && n_semantic <= K, "csm_rvq_encode: n_semantic out of range");
#define L(DD) hipLaunchKernelGGL((rvq_encode_kernel<DD>), dim3(T), dim3(256), 0, stream, x, codebooks, codes, T, K, C, n_semantic)
    if (D == 256) L(256); else if (D == 128) L(128); else L(64);
#undef L
    CSM_CHECK_LAUNCH("csm_rvq_encode");
    return 0;
}

extern "C" int csm_rvq_decode(const long long* codes, const float* codebooks, float* out, int T, int K, int C, int D,
                              hipStream_t stream) {
    CSM_REQUIRE(codes && codebooks && out && T > 0 && K > 0 && C > 0 && D > 0, "csm_rvq_decode: bad arguments");
    hipLaunchKernelGGL(rvq_decode_kernel, dim3(T), dim3(256), 0, stream, codes, codebooks, out, T, K, C, D);
    CSM_CHECK_LAUNCH("csm_rvq_decode");
    return 0;
}

// ================================================================================================ batch-1 decode path
// Reference Model.generate_frame (src/csm/models/model.py:140-195) runs one backbone position and 31 decoder positions
// per 80-ms frame against KV caches: every projection is a matrix-VECTOR product bound by weight streaming
// (1.95 GB + 31 x 0.22 GB of bf16 weights per frame), attention is a cache read.  These kernels keep the weights'
// [N][K] layout, load 16 B per lane straight to registers (an LDS round trip is pure overhead when nothing is shared
// between waves) and reduce with wave shuffles.
namespace {

// y[b][n] = sum_k x[b][k] * W[n][k] (+ R[b][n]);  one wave per output row, NB <= 4 batch rows share every weight load.
// Two optional fusions remove the tiny kernels that otherwise sit between the matrix-vector products of a decode step:
//   norm_w != NULL : x is RMS-normalised on its way into LDS (same arithmetic and summation order as rmsnorm_fwd_kernel:
//                    one wave per row, lane-strided chunks, wave_sum, one rounding of x * rstd * w to bf16);
//   SWIGLU         : W holds gate/up rows interleaved (w13); a wave computes rows 2i and 2i+1 and writes
//                    y[b][i] = silu(g) * u with g, u rounded to bf16 first (what swiglu_fwd_kernel reads back);
//   row_index      : batch row b of x is row (row_index[b] + row_offset) of a table (the embedding of a sampled code).
template <int NB, typename OutT, bool SWIGLU>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W, OutT* __restrict__ y,
                                                   const bf16_t* __restrict__ R, int N, int K, int ldw, int ldx, int ldy,
                                                   const bf16_t* __restrict__ norm_w, float eps, const int* __restrict__ row_index,
                                                   int row_offset) {
    extern __shared__ __attribute__((aligned(16))) char smem_x[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem_x);            // [NB][K]
    __shared__ float rs[4];
    for (int i = threadIdx.x * 8; i < NB * K; i += blockDim.x * 8) {
        const int b = i / K, k = i - b * K;
        const size_t row = row_index ? (size_t)(row_index[b] + row_offset) : (size_t)b;   // x = table[index[b] + offset]: embedding lookup
        *reinterpret_cast<U4*>(xs + i) = *reinterpret_cast<const U4*>(x + row * ldx + k);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    if (norm_w) {
        const int wv = threadIdx.x >> 6;
        if (wv < NB) {
            float ss = 0.f;
            for (int c = lane; c < (K >> 3); c += 64) {
                float f[8];
                unpack8(*reinterpret_cast<const U4*>(xs + wv * K + c * 8), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
            }
            ss = wave_sum(ss);
            if (lane == 0) rs[wv] = rsqrtf(ss / (float)K + eps);
        }
        __syncthreads();
        for (int i = threadIdx.x * 8; i < NB * K; i += blockDim.x * 8) {
            const int b = i / K, k = i - b * K;
            float f[8], w8[8];
            unpack8(*reinterpret_cast<const U4*>(xs + i), f);
            unpack8(*reinterpret_cast<const U4*>(norm_w + k), w8);
            const float r = rs[b];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = f[j] * r * w8[j];
            *reinterpret_cast<U4*>(xs + i) = pack8(f);
        }
        __syncthreads();
    }
    constexpr int RW = SWIGLU ? 2 : 1;                          // weight rows per output
    const int NO = N / RW;
    for (int n = blockIdx.x * wpb + (threadIdx.x >> 6); n < NO; n += gridDim.x * wpb) {
        float acc[RW][NB];
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
        const bf16_t* w = W + (size_t)n * RW * ldw;
        for (int k = lane * 8; k < K; k += 512) {
            float wf[RW][8];
#pragma unroll
            for (int r = 0; r < RW; ++r) unpack8(*reinterpret_cast<const U4*>(w + (size_t)r * ldw + k), wf[r]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float xf[8];
                unpack8(*reinterpret_cast<const U4*>(xs + b * K + k), xf);
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[r][b] += wf[r][j] * xf[j];
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float v = wave_sum(acc[0][b]);
            if constexpr (SWIGLU) {
                const float g = bf2f(f2bf(v)), u = bf2f(f2bf(wave_sum(acc[1][b])));
                v = silu(g) * u;
            }
            if (lane == 0) {
                if (R) v += bf2f(R[(size_t)b * ldy + n]);
                if constexpr (sizeof(OutT) == 2) y[(size_t)b * ldy + n] = f2bf(v);
                else y[(size_t)b * ldy + n] = v;
            }
        }
    }
}

// The same product for ONE batch row with x held in REGISTERS by every wave (round 4): K = 512 KCH, lane l owns the chunks
// k = 8 l + 512 c, c < KCH - the very chunks it multiplies in gemv_kernel's loop - so the whole of x is in the wave (KCH 16-byte
// loads per lane, L2 hits) and the RMSNorm prologue is a wave reduction: no LDS copy of x, no workgroup barrier (gemv_kernel has
// three in front of its first weight load), and the wave's weight rows are requested before anything else.  Same arithmetic
// in the same order as gemv_kernel<1, ...> (bit-identical: tests/test_e2e_gpu.py decode checks).  NT: non-temporal weight loads
// for matrices that are streamed once per frame (the backbone's 1.9 GB), so that they do not push the depth decoder's 222 MB -
// re-read 31 times per frame - out of the Infinity Cache.
// RPW (round 4): outputs per wave.  The w13 product of the depth decoder (K = 1024, 8192 gate/up pairs, 32 MB) runs one pair per
// wave; two and four pairs per wave (one normalisation of x per wave instead of per pair) were measured at 207.5 / 204 against
// 207.5 frames/s - the launch is bound by its weight stream - so 1 stays the default (csm_set_decode_tuning(2, n)).  A pair's two
// row sums share one butterfly: v_permlane32_swap(gate, up) leaves the gate partials in lanes 0-31 and the up partials in lanes
// 32-63, the remaining xor-16 .. xor-1 steps never cross the halves - the same additions as two separate wave_sum() calls, so the
// same bits.
template <int KCH, typename OutT, bool SWIGLU, bool NT, int RPW = 1>
__global__ __launch_bounds__(256) void gemv_reg_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W, OutT* __restrict__ y,
                                                       const bf16_t* __restrict__ R, int N, int ldw, int ldx, int ldy,
                                                       const bf16_t* __restrict__ norm_w, float eps, const int* __restrict__ row_index,
                                                       int row_offset) {
    constexpr int K = 512 * KCH;
    constexpr int RW = SWIGLU ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW;
    const int NO = N / RW;
    if (n0 >= NO) return;
    STAMP_DECL;
    STAMP(0);
    // the wave's weight rows: requested first (outputs past the end re-read the last one and are never stored)
    U4 wq[RPW][RW][KCH];
#pragma unroll
    for (int o = 0; o < RPW; ++o) {
        const bf16_t* w = W + (size_t)min(n0 + o, NO - 1) * RW * ldw + lane * 8;
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const U4* p = reinterpret_cast<const U4*>(w + (size_t)r * ldw + 512 * c);
                wq[o][r][c] = NT ? __builtin_nontemporal_load(p) : *p;
            }
    }
    const size_t row = row_index ? (size_t)(row_index[0] + row_offset) : (size_t)0;
    U4 xq[KCH];
#pragma unroll
    for (int c = 0; c < KCH; ++c) xq[c] = *reinterpret_cast<const U4*>(x + row * ldx + lane * 8 + 512 * c);
#ifdef CSM_DECODE_STAMPS
    STAMP(1);
#pragma unroll
    for (int c = 0; c < KCH; ++c) asm volatile("" : "+v"(xq[c].x), "+v"(xq[c].y), "+v"(xq[c].z), "+v"(xq[c].w));     // x has arrived
    STAMP(2);
#pragma unroll
    for (int o = 0; o < RPW; ++o)
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int c = 0; c < KCH; ++c) asm volatile("" : "+v"(wq[o][r][c].x), "+v"(wq[o][r][c].y), "+v"(wq[o][r][c].z), "+v"(wq[o][r][c].w));   // weights have arrived
    STAMP(3);
#endif
    if (norm_w) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            float f[8];
            unpack8(xq[c], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
        }
        ss = wave_sum(ss);
        const float rs = rsqrtf(ss / (float)K + eps);
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            float f[8], w8[8];
            unpack8(xq[c], f);
            unpack8(*reinterpret_cast<const U4*>(norm_w + lane * 8 + 512 * c), w8);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = f[j] * rs * w8[j];
            xq[c] = pack8(f);
        }
    }
    float acc[RPW][RW];
#pragma unroll
    for (int o = 0; o < RPW; ++o)
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[o][r] = 0.f;
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        float xf[8];
        unpack8(xq[c], xf);
#pragma unroll
        for (int o = 0; o < RPW; ++o)
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                float wf[8];
                unpack8(wq[o][r][c], wf);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[o][r] += wf[j] * xf[j];
            }
    }
    STAMP(4);
#pragma unroll
    for (int o = 0; o < RPW; ++o) {
        float v;
        if constexpr (SWIGLU) {
            // both row sums in one butterfly: gate partials to lanes 0-31, up partials to lanes 32-63, then xor 16 .. 1 inside the halves
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[o][0]), __float_as_uint(acc[o][1]), false, false);
            float t = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            t += lane_xor<16>(t); t += lane_xor<8>(t); t += lane_xor<4>(t); t += lane_xor<2>(t); t += lane_xor<1>(t);
            const float gs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 0));
            const float us = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 32));
            const float g = bf2f(f2bf(gs)), u = bf2f(f2bf(us));
            v = silu(g) * u;
        } else {
            v = wave_sum(acc[o][0]);
        }
        if (lane == 0 && n0 + o < NO) {
            if (R) v += bf2f(R[n0 + o]);
            if constexpr (sizeof(OutT) == 2) y[n0 + o] = f2bf(v);
            else y[n0 + o] = v;
        }
    }
    STAMP(5);
    STAMP_FLUSH(100 + KCH * 4 + (SWIGLU ? 2 : 0) + (norm_w ? 1 : 0));
}

// Two to four batch rows with x in registers (round 4; generate_batch): wave b of a workgroup prepares row b of x (RMSNorm as a
// wave reduction, rounded to bf16 exactly as gemv_kernel leaves it in LDS) and publishes it as fp32 behind ONE barrier; every wave
// then keeps ALL rows of the elements it multiplies in registers, so a weight chunk is unpacked once and feeds NB fused
// multiply-adds per element, two rows per v_pk_fma_f32 - 24 instructions per 16 bytes of weights at NB = 4, what one row costs -
// instead of re-reading and re-unpacking every row's x from LDS behind three barriers (gemv_kernel<4>: 8.5 us where the one-row
// kernel takes 4.6).  (Every wave preparing all rows itself - no LDS, no barrier - was slower than gemv_kernel: 419 against 439
// frames/s aggregate at B = 4; the preparation is ~130 instructions per row, the products of a K = 1024 row only ~50.)  K = 1024 /
// 2048; K = 8192 stays with gemv_kernel.  Every (row, output) accumulator sees gemv_kernel's products in gemv_kernel's order:
// bit-identical per row to the one-row kernels (test_batched_matrix_vector_kernels_match_single_row).
typedef float csm_f2 __attribute__((ext_vector_type(2)));
// SC = chunks per segment: K = 8192 (KCH = 16) walks x in four segments of four chunks (the fp32 rows stay in LDS, 128 KB at four
// rows; a segment's elements are in registers while its weight chunks are multiplied), K <= 2048 is one segment.
template <int KCH, int NB, typename OutT, bool SWIGLU, bool NT, int SC = KCH>
__global__ __launch_bounds__(256) void gemv_regn_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W, OutT* __restrict__ y,
                                                        const bf16_t* __restrict__ R, int N, int ldw, int ldx, int ldy,
                                                        const bf16_t* __restrict__ norm_w, float eps, const int* __restrict__ row_index,
                                                        int row_offset) {
    constexpr int K = 512 * KCH;
    constexpr int RW = SWIGLU ? 2 : 1;
    constexpr int NP = (NB + 1) / 2;                                        // row pairs (the last one half empty for odd NB)
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int NO = N / RW;
    const bool live = n < NO;                              // (every wave reaches the barrier; rows past the end re-read the last one)
    U4 wq[RW][KCH];
    const bf16_t* w = W + (size_t)(live ? n : NO - 1) * RW * ldw + lane * 8;
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const U4* p = reinterpret_cast<const U4*>(w + (size_t)r * ldw + 512 * c);
            wq[r][c] = NT ? __builtin_nontemporal_load(p) : *p;
        }
    // wave b < NB prepares row b of x (RMSNorm as a wave reduction, rounded to bf16 as gemv_kernel's LDS copy is) and publishes it
    // as fp32; ONE barrier; every lane then takes the elements it multiplies, two rows per register pair
    extern __shared__ __attribute__((aligned(16))) char smem_n[];
    float (*xsh)[K] = reinterpret_cast<float (*)[K]>(smem_n);                // [NB][K]
    const int wv = threadIdx.x >> 6;
    if (wv < NB) {
        const int b = wv;
        const size_t row = row_index ? (size_t)(row_index[b] + row_offset) : (size_t)b;
        U4 xr[KCH];
#pragma unroll
        for (int c = 0; c < KCH; ++c) xr[c] = *reinterpret_cast<const U4*>(x + row * ldx + lane * 8 + 512 * c);
        float rsn = 1.f;
        if (norm_w) {
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                float f[8];
                unpack8(xr[c], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
            }
            ss = wave_sum(ss);
            rsn = rsqrtf(ss / (float)K + eps);
        }
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            float f[8];
            unpack8(xr[c], f);
            if (norm_w) {
                float w8[8];
                unpack8(*reinterpret_cast<const U4*>(norm_w + lane * 8 + 512 * c), w8);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = f[j] * rsn * w8[j];
                unpack8(pack8(f), f);                                        // the bf16 rounding gemv_kernel's LDS copy carries
            }
            *reinterpret_cast<float4*>(&xsh[b][lane * 8 + 512 * c]) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4*>(&xsh[b][lane * 8 + 512 * c + 4]) = make_float4(f[4], f[5], f[6], f[7]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // (LDS only: the weight rows stay in flight)
    csm_f2 acc[RW][NP];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int q = 0; q < NP; ++q) acc[r][q] = (csm_f2){0.f, 0.f};
    static_assert(KCH % SC == 0, "whole segments");
#pragma unroll
    for (int sg = 0; sg < KCH / SC; ++sg) {
        csm_f2 xp[NP][SC][8];                                                // (row 2p, row 2p+1) of the segment's elements this lane multiplies
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int cc = 0; cc < SC; ++cc) {
                const int c = sg * SC + cc;
                const float4 a0 = *reinterpret_cast<const float4*>(&xsh[2 * q][lane * 8 + 512 * c]);
                const float4 a1 = *reinterpret_cast<const float4*>(&xsh[2 * q][lane * 8 + 512 * c + 4]);
                float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
                if (2 * q + 1 < NB) {
                    b0 = *reinterpret_cast<const float4*>(&xsh[2 * q + 1][lane * 8 + 512 * c]);
                    b1 = *reinterpret_cast<const float4*>(&xsh[2 * q + 1][lane * 8 + 512 * c + 4]);
                }
                xp[q][cc][0] = (csm_f2){a0.x, b0.x}; xp[q][cc][1] = (csm_f2){a0.y, b0.y}; xp[q][cc][2] = (csm_f2){a0.z, b0.z}; xp[q][cc][3] = (csm_f2){a0.w, b0.w};
                xp[q][cc][4] = (csm_f2){a1.x, b1.x}; xp[q][cc][5] = (csm_f2){a1.y, b1.y}; xp[q][cc][6] = (csm_f2){a1.z, b1.z}; xp[q][cc][7] = (csm_f2){a1.w, b1.w};
            }
#pragma unroll
        for (int cc = 0; cc < SC; ++cc) {
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                float wf[8];
                unpack8(wq[r][sg * SC + cc], wf);
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int q = 0; q < NP; ++q) acc[r][q] = __builtin_elementwise_fma((csm_f2){wf[j], wf[j]}, xp[q][cc][j], acc[r][q]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float a0 = (b & 1) ? acc[0][b >> 1].y : acc[0][b >> 1].x;
        float v;
        if constexpr (SWIGLU) {
            const float a1 = (b & 1) ? acc[1][b >> 1].y : acc[1][b >> 1].x;
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0), __float_as_uint(a1), false, false);
            float t = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            t += lane_xor<16>(t); t += lane_xor<8>(t); t += lane_xor<4>(t); t += lane_xor<2>(t); t += lane_xor<1>(t);
            const float gs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 0));
            const float us = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 32));
            const float g = bf2f(f2bf(gs)), u = bf2f(f2bf(us));
            v = silu(g) * u;
        } else {
            v = wave_sum(a0);
        }
        if (lane == 0 && live) {
            if (R) v += bf2f(R[(size_t)b * ldy + n]);
            if constexpr (sizeof(OutT) == 2) y[(size_t)b * ldy + n] = f2bf(v);
            else y[(size_t)b * ldy + n] = v;
        }
    }
}

// y[b][n] = sum_k x[b][k] * W[k][n]  (weights stored K-major, e.g. audio_head[i] = [d'][V]): a thread owns 8 columns.
template <int NB, typename OutT>
__global__ __launch_bounds__(256) void gemv_t_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W, OutT* __restrict__ y,
                                                     int N, int K, int ldw, int ldx, int ldy) {
    const int n0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (n0 >= N) return;
    float acc[NB][8];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[b][j] = 0.f;
    for (int k = 0; k < K; ++k) {
        float wf[8];
        unpack8(*reinterpret_cast<const U4*>(W + (size_t)k * ldw + n0), wf);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float xv = bf2f(x[(size_t)b * ldx + k]);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[b][j] += xv * wf[j];
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (n0 + j < N) {
                if constexpr (sizeof(OutT) == 2) y[(size_t)b * ldy + n0 + j] = f2bf(acc[b][j]);
                else y[(size_t)b * ldy + n0 + j] = acc[b][j];
            }
}

// copy the new position's K and V heads (already RoPE'd) from the fused qkv row into the caches [B][KV][S_max][HD]
__global__ __launch_bounds__(256) void kv_append_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                        const int* __restrict__ pos, int H, int KV, int HD, int S_max, int ld) {
    const int b = blockIdx.x;
    const int p = pos[b];
    const int per = KV * HD;
    for (int i = threadIdx.x * 8; i < per; i += blockDim.x * 8) {
        const int kvh = i / HD, d = i - kvh * HD;
        const size_t dst = (((size_t)b * KV + kvh) * S_max + p) * HD + d;
        *reinterpret_cast<U4*>(kc + dst) = *reinterpret_cast<const U4*>(qkv + (size_t)b * ld + H * HD + i);
        *reinterpret_cast<U4*>(vc + dst) = *reinterpret_cast<const U4*>(qkv + (size_t)b * ld + (H + KV) * HD + i);
    }
}

// one query position against the cache: block per (b, q-head); scores for <= 2048 keys live in LDS.
// table != NULL fuses what used to be two more launches per layer: the new position's q and k heads are rotated here
// (torchtune RoPE on interleaved pairs, same arithmetic and bf16 rounding as rope_kernel), and the first q-head block of
// every kv group appends the rotated k and the v of the new position to the caches.  Every block takes the new key /
// value from the qkv row, not from the cache, so blocks of one group do not race with the appending block.
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ kc,
                                                          bf16_t* __restrict__ vc, bf16_t* __restrict__ out,
                                                          const int* __restrict__ pos, int H, int KV, int S_max, int ld,
                                                          float scale, const float* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) char smem_a[];
    float* sc = reinterpret_cast<float*>(smem_a);                 // [S_max] scores, then probabilities
    float* red = sc + S_max;                                       // 16 floats
    float* part = red + 16;                                        // [4 waves][HD] partial outputs
    float* qs = part + 4 * HD;                                     // [HD] query (rotated), bf16-rounded values
    float* kn = qs + HD;                                           // [HD] new key (rotated)
    const int h = blockIdx.x, b = blockIdx.y;
    const int rep = H / KV, kvh = h / rep;
    const int p = pos[b];
    const int n = p + 1;                                           // keys 0 .. pos
    const bf16_t* q = qkv + (size_t)b * ld + h * HD;
    const bf16_t* knew = qkv + (size_t)b * ld + (H + kvh) * HD;
    const bf16_t* vnew = qkv + (size_t)b * ld + (H + KV + kvh) * HD;
    bf16_t* K = kc + ((size_t)b * KV + kvh) * S_max * HD;
    bf16_t* V = vc + ((size_t)b * KV + kvh) * S_max * HD;
    if (threadIdx.x < HD / 2) {
        const int i = threadIdx.x;
        float c = 1.f, sn = 0.f;
        if (table) { c = table[((size_t)p * (HD / 2) + i) * 2]; sn = table[((size_t)p * (HD / 2) + i) * 2 + 1]; }
        const float q0 = bf2f(q[2 * i]), q1 = bf2f(q[2 * i + 1]), k0 = bf2f(knew[2 * i]), k1 = bf2f(knew[2 * i + 1]);
        float q0r = q0, q1r = q1, k0r = k0, k1r = k1;
        rope_rot(q0r, q1r, c, sn);
        rope_rot(k0r, k1r, c, sn);
        const bf16_t rq0 = f2bf(q0r), rq1 = f2bf(q1r);
        const bf16_t rk0 = f2bf(k0r), rk1 = f2bf(k1r);
        qs[2 * i] = bf2f(rq0); qs[2 * i + 1] = bf2f(rq1);
        kn[2 * i] = bf2f(rk0); kn[2 * i + 1] = bf2f(rk1);
        if (table && h % rep == 0) {
            K[(size_t)p * HD + 2 * i] = rk0; K[(size_t)p * HD + 2 * i + 1] = rk1;
            V[(size_t)p * HD + 2 * i] = vnew[2 * i]; V[(size_t)p * HD + 2 * i + 1] = vnew[2 * i + 1];
        }
    }
    __syncthreads();
    float qf[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) qf[c] = qs[c];
    float mx = -INFINITY;
    for (int s = threadIdx.x; s < n; s += blockDim.x) {
        float d = 0.f;
        if (s == p) {
#pragma unroll
            for (int c = 0; c < HD; ++c) d += qf[c] * kn[c];
        } else {
#pragma unroll
            for (int c = 0; c < HD / 8; ++c) {
                float kf[8];
                unpack8(*reinterpret_cast<const U4*>(K + (size_t)s * HD + c * 8), kf);
#pragma unroll
                for (int j = 0; j < 8; ++j) d += qf[c * 8 + j] * kf[j];
            }
        }
        d *= scale;
        sc[s] = d;
        mx = fmaxf(mx, d);
    }
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int s = threadIdx.x; s < n; s += blockDim.x) {
        const float p = __expf(sc[s] - mx);
        sc[s] = p;
        sum += p;
    }
    sum = block_sum(sum, red);
    __syncthreads();
    // out[d] = sum_s p[s] V[s][d]: lane = 8 consecutive d's of a (HD/8)-lane group, groups stride over keys
    constexpr int LPR = HD / 8;                                    // lanes per key row
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR, ngrp = blockDim.x / LPR;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = grp; s < n; s += ngrp) {
        float vf[8];
        unpack8(*reinterpret_cast<const U4*>((s == p ? vnew : V + (size_t)s * HD) + sub * 8), vf);
        const float p = sc[s];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += p * vf[j];
    }
    // reduce the ngrp partials: first inside a wave (groups of a wave differ in lane bits >= log2(LPR)), then across waves
    static_assert(LPR == 8 || LPR == 16, "head_dim 64 or 128");
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if constexpr (LPR == 8) acc[j] += lane_xor<8>(acc[j]);
        acc[j] += lane_xor<16>(acc[j]);
        acc[j] += lane_xor<32>(acc[j]);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < LPR)
#pragma unroll
        for (int j = 0; j < 8; ++j) part[wave * HD + lane * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < HD) {
        float o = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) o += part[w * HD + threadIdx.x];
        out[(size_t)b * H * HD + h * HD + threadIdx.x] = f2bf(o / sum);
    }
}


// The depth decoder's attention folded into the prologue of its output-projection matrix-vector product (round 3): a decoder
// step sees at most 32 keys (one frame's codebooks), so the attention of ALL heads is a few thousand multiply-adds - cheaper to
// recompute in every workgroup of the following product than to launch on its own (124 launches per generated frame, each
// ~7.5 us of launch + ramp for ~0.3 us of work).  Every workgroup: rotates q and the new k (torchtune RoPE, interleaved pairs),
// computes softmax(q K^T / sqrt(HD)) V over cache rows 0 .. pos-1 plus the new key / value taken from the qkv row, writes
// the bf16 result into LDS as the product's input vector, then runs the usual row-per-wave product with the residual.
// Workgroup 0 also appends the rotated k and the v of the new position to the caches (the others never read that row).
// The arithmetic follows attn_decode_kernel operation for operation - the same products in the same order, the same
// reduction trees (scores: one key per lane, a serial dot over HD; sum: a 64-lane butterfly; P.V: 16 key groups of 8-column
// slices, groups summed ((g0+g1)+(g2+g3)) per quartet and the quartets left to right) - so a frame decoded through this
// kernel is bit-identical to one decoded through csm_attn_decode_rope + csm_gemv_bf16 (tests/test_e2e_gpu.py).
template <int NB, int HD>
__global__ __launch_bounds__(512) void gemv_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                        const int* __restrict__ pos, const float* __restrict__ table,
                                                        const bf16_t* __restrict__ W, bf16_t* __restrict__ y,
                                                        const bf16_t* __restrict__ R, int N, int H, int KV, int S_max, int ldq,
                                                        int ldw, int ldy, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_x[];
    const int K = H * HD;
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem_x);                         // [NB][K]   attention output = the product's input
    float* qs = reinterpret_cast<float*>(smem_x + (size_t)NB * K * 2);      // [NB][H][HD]  rotated q (bf16-rounded values)
    float* kn = qs + (size_t)NB * H * HD;                                   // [NB][KV][HD] rotated new k
    float* pw = kn + (size_t)NB * KV * HD;                                  // [waves][64] probabilities
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rep = H / KV;
    // the wave's first weight row: requested before the attention prologue (it depends on nothing here; round 4: behind the two
    // barriers below its latency was the tail of every one of the 128 launches per frame)
    constexpr int KCH = 2;                                                  // K = H * HD = 1024 in the depth decoder: 2 chunks per lane
    const int row0 = blockIdx.x * (int)(blockDim.x >> 6) + wave;
    const bool pre = (K == 512 * KCH) && row0 < N;
    U4 wq0[KCH] = {};
    if (pre) {
#pragma unroll
        for (int c = 0; c < KCH; ++c) wq0[c] = *reinterpret_cast<const U4*>(W + (size_t)row0 * ldw + lane * 8 + 512 * c);
    }
    // ---- RoPE of q (all heads) and of the new k (all kv heads); workgroup 0 appends k, v to the caches
    for (int it = threadIdx.x; it < NB * (H + KV) * (HD / 2); it += blockDim.x) {
        const int i = it % (HD / 2), hh = (it / (HD / 2)) % (H + KV), b = it / ((HD / 2) * (H + KV));
        const int p = pos[b];
        const float c = table[((size_t)p * (HD / 2) + i) * 2], sn = table[((size_t)p * (HD / 2) + i) * 2 + 1];
        const bf16_t* src = qkv + (size_t)b * ldq + hh * HD;                 // q heads, then k heads, contiguous in the fused row
        const float x0 = bf2f(src[2 * i]), x1 = bf2f(src[2 * i + 1]);
        float y0 = x0, y1 = x1;
        rope_rot(y0, y1, c, sn);
        const bf16_t r0 = f2bf(y0), r1 = f2bf(y1);
        if (hh < H) {
            qs[((size_t)b * H + hh) * HD + 2 * i] = bf2f(r0); qs[((size_t)b * H + hh) * HD + 2 * i + 1] = bf2f(r1);
        } else {
            const int kvh = hh - H;
            kn[((size_t)b * KV + kvh) * HD + 2 * i] = bf2f(r0); kn[((size_t)b * KV + kvh) * HD + 2 * i + 1] = bf2f(r1);
            if (blockIdx.x == 0) {
                const bf16_t* vnew = qkv + (size_t)b * ldq + (H + KV + kvh) * HD;
                const size_t dst = (((size_t)b * KV + kvh) * S_max + p) * HD;
                kc[dst + 2 * i] = r0; kc[dst + 2 * i + 1] = r1;
                vc[dst + 2 * i] = vnew[2 * i]; vc[dst + 2 * i + 1] = vnew[2 * i + 1];
            }
        }
    }
    __syncthreads();
    // ---- attention: one wave per (batch row, q head)
    constexpr int LPR = HD / 8;                                              // lanes per value row (16)
    const int sub = lane % LPR, gq = lane / LPR;                             // 8-column slice, group within a quartet
    for (int hb = wave; hb < NB * H; hb += (int)(blockDim.x >> 6)) {
        const int b = hb / H, h = hb % H, kvh = h / rep;
        const int p = pos[b], n = p + 1;
        const float* q = qs + ((size_t)b * H + h) * HD;
        const float* knew = kn + ((size_t)b * KV + kvh) * HD;
        const bf16_t* Kc = kc + ((size_t)b * KV + kvh) * S_max * HD;
        const bf16_t* Vc = vc + ((size_t)b * KV + kvh) * S_max * HD;
        const bf16_t* vnew = qkv + (size_t)b * ldq + (H + KV + kvh) * HD;
        float d = 0.f;
        if (lane < n) {
            if (lane == p) {
#pragma unroll 8
                for (int c = 0; c < HD; ++c) d += q[c] * knew[c];
            } else {
                for (int c = 0; c < HD / 8; ++c) {
                    float kf[8];
                    unpack8(*reinterpret_cast<const U4*>(Kc + (size_t)lane * HD + c * 8), kf);
#pragma unroll
                    for (int j = 0; j < 8; ++j) d += q[c * 8 + j] * kf[j];
                }
            }
            d *= scale;
        }
        const float mx = wave_max(lane < n ? d : -INFINITY);
        const float pr = lane < n ? __expf(d - mx) : 0.f;
        const float sum = wave_sum(pr);
        pw[wave * 64 + lane] = pr;                                           // (wave-private: a wave's own LDS operations are in order)
        float o8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                        // quartet r: key groups 4r .. 4r+3, mine is 4r + gq
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int s_ = 4 * r + gq; s_ < n; s_ += 16) {
                float vf[8];
                unpack8(*reinterpret_cast<const U4*>((s_ == p ? vnew : Vc + (size_t)s_ * HD) + sub * 8), vf);
                const float ps = pw[wave * 64 + s_];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += ps * vf[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] += lane_xor<LPR>(acc[j]);
                acc[j] += lane_xor<2 * LPR>(acc[j]);
                o8[j] += acc[j];                                             // quartets left to right, starting from 0 (0 + W0 is exact)
            }
        }
        if (gq == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xs[(size_t)b * K + h * HD + sub * 8 + j] = f2bf(o8[j] / sum);
        }
    }
    __syncthreads();
    // ---- y = attention . W^T (+ R): one wave per output row (gemv_kernel's loop)
    const int wpb = blockDim.x >> 6;
    for (int nrow = blockIdx.x * wpb + wave; nrow < N; nrow += gridDim.x * wpb) {
        float acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = 0.f;
        const bf16_t* w = W + (size_t)nrow * ldw;
        if (pre && nrow == row0) {                       // (wave-uniform) the prefetched row: the same chunks in the same order
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int k = lane * 8 + 512 * c;
                float wf[8];
                unpack8(wq0[c], wf);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    float xf[8];
                    unpack8(*reinterpret_cast<const U4*>(xs + b * K + k), xf);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[b] += wf[j] * xf[j];
                }
            }
        } else {
            for (int k = lane * 8; k < K; k += 512) {
                float wf[8];
                unpack8(*reinterpret_cast<const U4*>(w + k), wf);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    float xf[8];
                    unpack8(*reinterpret_cast<const U4*>(xs + b * K + k), xf);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[b] += wf[j] * xf[j];
                }
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float v = wave_sum(acc[b]);
            if (lane == 0) {
                if (R) v += bf2f(R[(size_t)b * ldy + nrow]);
                y[(size_t)b * ldy + nrow] = f2bf(v);
            }
        }
    }
}

// The same layer step for ONE utterance whose position the HOST knows (round 4): the depth decoder's step i is always at position
// i, so the captured frame graph can carry it as a kernel argument.  With the position in a register every address of the
// prologue - the RoPE table row, the q|k|v row, the <= 31 cached key rows, the value slices, the first weight row - is known when
// the wave starts, and all of them are requested at once: one memory round trip instead of the dependent chain of
// gemv_attn_kernel (position -> q|k|v row and table -> key rows -> value rows).  The cached key rows are fetched as contiguous
// 16-byte chunks (512 threads x 16 B = a kv head's 31 rows in one coalesced request each; `lane = key` loads touched 32 cache lines
// per instruction and took 1.9 us to issue, tools/probes/decode_stamps.py) and re-read from LDS in the lane = key layout (row
// stride 272 B: conflict-free); the rotated new key is written into the same LDS image, so that lane takes the same code path
// instead of a divergent serial dot product.  The arithmetic is gemv_attn_kernel<1, 128>'s operation for operation (same products,
// same order, same trees): bit-identical.  512 threads: wave w = q head w in the attention (H <= 8); S_max <= 32 (a frame's codebooks).
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load (its fence is
// s_waitcnt vmcnt(0)), which here would pull the prefetched value slices and the weight row - needed last - in front of the first
// barrier.  The compiler still counts vmcnt for the registers those loads fill.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int HD>
__global__ __launch_bounds__(512) void gemv_attn_at_kernel(const bf16_t* __restrict__ qkv, bf16_t* kc, bf16_t* vc, int p,
                                                           const float* __restrict__ table, const bf16_t* __restrict__ W,
                                                           bf16_t* __restrict__ y, const bf16_t* __restrict__ R, int N, int H, int KV,
                                                           int S_max, int ldw, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_x[];
    static_assert(HD == 128, "lane = column pair in the P.V phase");
    constexpr int KROW = HD * 2 + 16;                                       // bytes per key row of the LDS image (conflict-free lane = key reads)
    constexpr int VROW = HD * 2;                                            // value rows are read lane = column pair: no padding
    const int K = H * HD;
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem_x);                         // [K]
    float* qs = reinterpret_cast<float*>(smem_x + (size_t)K * 2);           // [H][HD]
    float* pw = qs + (size_t)H * HD;                                        // [8][32] probabilities (un-normalised), then [8] their sums
    float* psum = pw + 256;
    char* ks = reinterpret_cast<char*>(psum + 8);                           // [KV][32][KROW] key rows 0 .. p (row p: the new, rotated key)
    char* vs = ks + (size_t)KV * 32 * KROW;                                 // [KV][32][VROW] value rows 0 .. p
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rep = H / KV, n = p + 1;
    constexpr int KCH = 2;
    STAMP_DECL;
    STAMP(0);
    // ---- every load of the prologue, back to back, in the order they are consumed (loads return in order: the weight row, which
    //      comes from furthest away and is needed last, goes last)
    const int row0 = blockIdx.x * 8 + wave;
    const int npair = (H + KV) * (HD / 2);                                  // 640 pairs over 512 threads: two rounds
    float rc[2], rs[2], rx0[2], rx1[2];
    uint32_t rv[2] = {0u, 0u};                                              // (k-head pairs: the new value's pair beside it)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int it = threadIdx.x + 512 * t;
        rc[t] = rs[t] = rx0[t] = rx1[t] = 0.f;
        if (it < npair) {
            const int i = it % (HD / 2), hh = it / (HD / 2);
            const float2 cs = *reinterpret_cast<const float2*>(table + ((size_t)p * (HD / 2) + i) * 2);
            const uint32_t xx = *reinterpret_cast<const uint32_t*>(qkv + hh * HD + 2 * i);
            if (hh >= H) rv[t] = *reinterpret_cast<const uint32_t*>(qkv + (hh + KV) * HD + 2 * i);
            rc[t] = cs.x; rs[t] = cs.y; rx0[t] = __uint_as_float(xx << 16); rx1[t] = __uint_as_float(xx & 0xffff0000u);
        }
    }
    // cached key / value rows 0 .. p-1 of both kv heads: thread t = 16-byte chunk t of the head's contiguous rows (HD / 8 chunks per row)
    U4 kg[2] = {}, vg[2] = {};
    const bool kld = (int)threadIdx.x < p * (HD / 8);
    if (kld) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
            if (kh < KV) kg[kh] = *reinterpret_cast<const U4*>(kc + (size_t)kh * S_max * HD + (size_t)threadIdx.x * 8);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
            if (kh < KV) vg[kh] = *reinterpret_cast<const U4*>(vc + (size_t)kh * S_max * HD + (size_t)threadIdx.x * 8);
    }
    U4 wq0[KCH] = {};
    if (row0 < N) {
#pragma unroll
        for (int c = 0; c < KCH; ++c) wq0[c] = *reinterpret_cast<const U4*>(W + (size_t)row0 * ldw + lane * 8 + 512 * c);
    }
    STAMP(1);
#ifdef CSM_DECODE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (probe: everything requested above has arrived)
    STAMP(2);
#endif
    // ---- RoPE of q and of the new k (-> row p of the key image; the new value -> row p of the value image); workgroup 0 appends
    //      k, v to the caches (row p: no load above reads it); then the cached rows into the images
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int it = threadIdx.x + 512 * t;
        if (it < npair) {
            const int i = it % (HD / 2), hh = it / (HD / 2);
            float y0 = rx0[t], y1 = rx1[t];
            rope_rot(y0, y1, rc[t], rs[t]);
            const bf16_t r0 = f2bf(y0), r1 = f2bf(y1);
            if (hh < H) {
                qs[(size_t)hh * HD + 2 * i] = bf2f(r0); qs[(size_t)hh * HD + 2 * i + 1] = bf2f(r1);
            } else {
                const int kh = hh - H;
                const uint32_t kk = (uint32_t)r0 | ((uint32_t)r1 << 16);
                *reinterpret_cast<uint32_t*>(ks + ((size_t)kh * 32 + p) * KROW + 4 * i) = kk;
                *reinterpret_cast<uint32_t*>(vs + ((size_t)kh * 32 + p) * VROW + 4 * i) = rv[t];
                if (blockIdx.x == 0) {
                    const size_t dst = ((size_t)kh * S_max + p) * HD + 2 * i;
                    *reinterpret_cast<uint32_t*>(kc + dst) = kk;
                    *reinterpret_cast<uint32_t*>(vc + dst) = rv[t];
                }
            }
        }
    }
    if (kld) {
        const int row = threadIdx.x / (HD / 8), ch = threadIdx.x % (HD / 8);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
            if (kh < KV) {
                *reinterpret_cast<U4*>(ks + ((size_t)kh * 32 + row) * KROW + ch * 16) = kg[kh];
                *reinterpret_cast<U4*>(vs + ((size_t)kh * 32 + row) * VROW + ch * 16) = vg[kh];
            }
    }
    lds_barrier();
    STAMP(3);
    // ---- scores + softmax: waves 0 .. H/2-1, two heads of one kv group per wave (lanes 0-31: head 2w, lanes 32-63: head 2w+1;
    //      lane & 31 = key).  attn_decode_kernel's arithmetic: a serial dot over HD per key; max and sum as 64-lane butterflies whose
    //      xor-32 step meets only -inf / 0 there (at most 32 keys) and is an identity - the xor-16 .. xor-1 steps stay inside a half.
    if (wave < H / 2) {
        const int hh = 2 * wave + (lane >> 5), key = lane & 31, kvh = hh / rep;
        const float* q = qs + (size_t)hh * HD;
        float d = 0.f;
        if (key < n) {
            const char* krow = ks + ((size_t)kvh * 32 + key) * KROW;
#pragma unroll
            for (int c = 0; c < HD / 8; ++c) {
                float kf[8];
                unpack8(*reinterpret_cast<const U4*>(krow + c * 16), kf);
#pragma unroll
                for (int j = 0; j < 8; ++j) d += q[c * 8 + j] * kf[j];
            }
            d *= scale;
        }
        float mx = key < n ? d : -INFINITY;
        mx = fmaxf(mx, lane_xor<16>(mx)); mx = fmaxf(mx, lane_xor<8>(mx)); mx = fmaxf(mx, lane_xor<4>(mx));
        mx = fmaxf(mx, lane_xor<2>(mx)); mx = fmaxf(mx, lane_xor<1>(mx));
        const float pr = key < n ? __expf(d - mx) : 0.f;
        float sum = pr;
        sum += lane_xor<16>(sum); sum += lane_xor<8>(sum); sum += lane_xor<4>(sum); sum += lane_xor<2>(sum); sum += lane_xor<1>(sum);
        pw[hh * 32 + key] = pr;
        if (key == 0) psum[hh] = sum;
    }
    STAMP(4);
    lds_barrier();
    // ---- P.V: wave = head, lane = column pair, no cross-lane traffic.  The reference order per column (attn_decode_kernel): 16 key
    //      groups (quartet r, group g: keys 4r + g and 4r + g + 16, accumulated from 0 by fused multiply-adds), a quartet's groups
    //      summed (g0 + g1) + (g2 + g3), the quartets added left to right starting from 0 - evaluated here by ONE lane per column
    //      pair instead of 4 x 16 lanes and two cross-lane sums per quartet: the same operations on the same operands.
    if (wave < H) {
        const int h = wave, kvh = h / rep;
        const float* ph = pw + h * 32;
        const char* vcol = vs + (size_t)kvh * 32 * VROW + lane * 4;
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a0[4], a1[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                a0[g4] = 0.f; a1[g4] = 0.f;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int s_ = 4 * r + g4 + 16 * u;
                    if (s_ < n) {                                            // (uniform: n is a launch argument)
                        const uint32_t vv = *reinterpret_cast<const uint32_t*>(vcol + (size_t)s_ * VROW);
                        const float ps = ph[s_];
                        a0[g4] = __builtin_fmaf(ps, __uint_as_float(vv << 16), a0[g4]);
                        a1[g4] = __builtin_fmaf(ps, __uint_as_float(vv & 0xffff0000u), a1[g4]);
                    }
                }
            }
            o0 += (a0[0] + a0[1]) + (a0[2] + a0[3]);
            o1 += (a1[0] + a1[1]) + (a1[2] + a1[3]);
        }
        const float sum = psum[h];
        *reinterpret_cast<uint32_t*>(xs + h * HD + 2 * lane) = (uint32_t)f2bf(o0 / sum) | ((uint32_t)f2bf(o1 / sum) << 16);
    }
    lds_barrier();
    STAMP(5);
    // ---- y = attention . W^T (+ R): one wave per output row
    for (int nrow = row0; nrow < N; nrow += gridDim.x * 8) {
        float acc = 0.f;
        const bf16_t* w = W + (size_t)nrow * ldw;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int k = lane * 8 + 512 * c;
            float wf[8], xf[8];
            unpack8(nrow == row0 ? wq0[c] : *reinterpret_cast<const U4*>(w + k), wf);
            unpack8(*reinterpret_cast<const U4*>(xs + k), xf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += wf[j] * xf[j];
        }
        float v = wave_sum(acc);
        if (lane == 0) {
            if (R) v += bf2f(R[nrow]);
            y[nrow] = f2bf(v);
        }
    }
    STAMP(6);
    STAMP_FLUSH(200);
}

// The depth decoder's attention for two to four utterances (round 4; generate_batch), position known on the host: the scores and
// P.V phases of gemv_attn_at_kernel as a launch of their own - one workgroup per (batch row, kv head), its four q heads on four
// waves - where the fused form would make every workgroup of the output projection recompute all rows' attention (measured in
// round 3: slower from two rows on).  Same loads-at-once prologue, same coalesced key / value images, same arithmetic as
// attn_decode_kernel<128> operation for operation: bit-identical.  H = 4 KV, S_max <= 32.
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_at_kernel(const bf16_t* __restrict__ qkv, bf16_t* kc, bf16_t* vc, bf16_t* __restrict__ out,
                                                             int p, const float* __restrict__ table, int H, int KV, int S_max, int ld,
                                                             float scale) {
    static_assert(HD == 128, "lane = column pair in the P.V phase");
    constexpr int KROW = HD * 2 + 16, VROW = HD * 2;
    __shared__ __attribute__((aligned(16))) float qs[4][HD];
    __shared__ float pw[4][32], psum[4];
    __shared__ __attribute__((aligned(16))) char ks[32 * KROW];
    __shared__ __attribute__((aligned(16))) char vs[32 * VROW];
    const int kvh = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = p + 1;
    const bf16_t* row = qkv + (size_t)b * ld;
    bf16_t* Kc = kc + ((size_t)b * KV + kvh) * S_max * HD;
    bf16_t* Vc = vc + ((size_t)b * KV + kvh) * S_max * HD;
    // ---- every load at once: RoPE inputs of the group's four q heads and its k head (5 x 64 pairs over 256 threads: two rounds),
    //      the new value beside the k pairs, the cached rows as contiguous 16-byte chunks (two per thread and image)
    float rc[2], rs[2], rx0[2], rx1[2];
    uint32_t rv[2] = {0u, 0u};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int it = threadIdx.x + 256 * t;
        rc[t] = rs[t] = rx0[t] = rx1[t] = 0.f;
        if (it < 5 * (HD / 2)) {
            const int i = it % (HD / 2), hh = it / (HD / 2);                 // hh < 4: q head 4 kvh + hh; hh == 4: the k head
            const float2 cs = *reinterpret_cast<const float2*>(table + ((size_t)p * (HD / 2) + i) * 2);
            const int col = (hh < 4 ? (4 * kvh + hh) : (H + kvh)) * HD + 2 * i;
            const uint32_t xx = *reinterpret_cast<const uint32_t*>(row + col);
            if (hh == 4) rv[t] = *reinterpret_cast<const uint32_t*>(row + (H + KV + kvh) * HD + 2 * i);
            rc[t] = cs.x; rs[t] = cs.y; rx0[t] = __uint_as_float(xx << 16); rx1[t] = __uint_as_float(xx & 0xffff0000u);
        }
    }
    U4 kg[2] = {}, vg[2] = {};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ch = threadIdx.x + 256 * t;
        if (ch < p * (HD / 8)) {
            kg[t] = *reinterpret_cast<const U4*>(Kc + (size_t)ch * 8);
            vg[t] = *reinterpret_cast<const U4*>(Vc + (size_t)ch * 8);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int it = threadIdx.x + 256 * t;
        if (it < 5 * (HD / 2)) {
            const int i = it % (HD / 2), hh = it / (HD / 2);
            float y0 = rx0[t], y1 = rx1[t];
            rope_rot(y0, y1, rc[t], rs[t]);
            const bf16_t r0 = f2bf(y0), r1 = f2bf(y1);
            if (hh < 4) {
                qs[hh][2 * i] = bf2f(r0); qs[hh][2 * i + 1] = bf2f(r1);
            } else {
                const uint32_t kk = (uint32_t)r0 | ((uint32_t)r1 << 16);
                *reinterpret_cast<uint32_t*>(ks + (size_t)p * KROW + 4 * i) = kk;
                *reinterpret_cast<uint32_t*>(vs + (size_t)p * VROW + 4 * i) = rv[t];
                *reinterpret_cast<uint32_t*>(Kc + (size_t)p * HD + 2 * i) = kk;        // (row p: no load above reads it)
                *reinterpret_cast<uint32_t*>(Vc + (size_t)p * HD + 2 * i) = rv[t];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ch = threadIdx.x + 256 * t;
        if (ch < p * (HD / 8)) {
            const int r_ = ch / (HD / 8), c_ = ch % (HD / 8);
            *reinterpret_cast<U4*>(ks + (size_t)r_ * KROW + c_ * 16) = kg[t];
            *reinterpret_cast<U4*>(vs + (size_t)r_ * VROW + c_ * 16) = vg[t];
        }
    }
    lds_barrier();
    // ---- scores + softmax: waves 0, 1 - two heads per wave in lane halves (see gemv_attn_at_kernel)
    if (wave < 2) {
        const int hh = 2 * wave + (lane >> 5), key = lane & 31;
        const float* q = qs[hh];
        float d = 0.f;
        if (key < n) {
            const char* krow = ks + (size_t)key * KROW;
#pragma unroll
            for (int c = 0; c < HD / 8; ++c) {
                float kf[8];
                unpack8(*reinterpret_cast<const U4*>(krow + c * 16), kf);
#pragma unroll
                for (int j = 0; j < 8; ++j) d += q[c * 8 + j] * kf[j];
            }
            d *= scale;
        }
        float mx = key < n ? d : -INFINITY;
        mx = fmaxf(mx, lane_xor<16>(mx)); mx = fmaxf(mx, lane_xor<8>(mx)); mx = fmaxf(mx, lane_xor<4>(mx));
        mx = fmaxf(mx, lane_xor<2>(mx)); mx = fmaxf(mx, lane_xor<1>(mx));
        const float pr = key < n ? __expf(d - mx) : 0.f;
        float sum = pr;
        sum += lane_xor<16>(sum); sum += lane_xor<8>(sum); sum += lane_xor<4>(sum); sum += lane_xor<2>(sum); sum += lane_xor<1>(sum);
        pw[hh][key] = pr;
        if (key == 0) psum[hh] = sum;
    }
    lds_barrier();
    // ---- P.V: wave = head of the group, lane = column pair, the reference's summation tree (see gemv_attn_at_kernel)
    {
        const float* ph = pw[wave];
        const char* vcol = vs + lane * 4;
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a0[4], a1[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                a0[g4] = 0.f; a1[g4] = 0.f;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int s_ = 4 * r + g4 + 16 * u;
                    if (s_ < n) {
                        const uint32_t vv = *reinterpret_cast<const uint32_t*>(vcol + (size_t)s_ * VROW);
                        const float ps = ph[s_];
                        a0[g4] = __builtin_fmaf(ps, __uint_as_float(vv << 16), a0[g4]);
                        a1[g4] = __builtin_fmaf(ps, __uint_as_float(vv & 0xffff0000u), a1[g4]);
                    }
                }
            }
            o0 += (a0[0] + a0[1]) + (a0[2] + a0[3]);
            o1 += (a1[0] + a1[1]) + (a1[2] + a1[3]);
        }
        const float sum = psum[wave];
        *reinterpret_cast<uint32_t*>(out + (size_t)b * H * HD + (4 * kvh + wave) * HD + 2 * lane) =
            (uint32_t)f2bf(o0 / sum) | ((uint32_t)f2bf(o1 / sum) << 16);
    }
}

}  // namespace

#ifdef CSM_DECODE_STAMPS
// probe builds: copy out (and reset) the stamp records; returns the number of 10-word records
extern "C" int csm_decode_stamps(unsigned long long* host, int max_records) {
    unsigned int n = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_decode_nstamp), sizeof(n));
    int rec = (int)(n / 10); if (rec > max_records) rec = max_records; if (rec > 819) rec = 819;
    if (host && rec > 0) (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_decode_stamps), (size_t)rec * 10 * sizeof(unsigned long long));
    const unsigned int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_decode_nstamp), &zero, sizeof(zero));
    return rec;
}
#endif
// (g_gemv_rpw: gate/up pairs per wave in the depth decoder's w13 product.  Measured at 1 / 2 / 4: 207.5 / 207.5 / 204 frames/s -
//  that launch streams 32 MB and is bound by its loads, not by the per-wave normalisation: one pair per wave stays the default)
static int g_gemv_reg = 1, g_gemv_nt = 1, g_gemv_rpw = 1, g_gemv_regn = 1;       // csm_set_decode_tuning (A/B: tools/probes)
extern "C" int csm_set_decode_tuning(int key, int value) {
    if (key == 0) g_gemv_reg = value; else if (key == 1) g_gemv_nt = value; else if (key == 2) g_gemv_rpw = value;
    else if (key == 3) g_gemv_regn = value; else return 1;
    return 0;
}
static int gemv_launch(const void* x, const void* W, void* y, const void* residual, int B, int N, int K, int ldw, int ldx, int ldy,
                       int out_f32, const void* norm_w, float eps, int swiglu, const int* row_index, int row_offset,
                       hipStream_t stream) {
    CSM_REQUIRE(x && W && y && B >= 1 && B <= 4 && N > 0 && K > 0 && (K & 7) == 0 && (ldw & 7) == 0 && (ldx & 7) == 0,
                "csm_gemv_bf16: bad arguments (B=%d N=%d K=%d)", B, N, K);
    CSM_REQUIRE((size_t)B * K * 2 <= 65536, "csm_gemv_bf16: B*K too large for the LDS copy of x");
    CSM_REQUIRE(!swiglu || ((N & 1) == 0 && !out_f32), "csm_gemv_bf16_ex: the SwiGLU form needs an even N and bf16 output");
    const int no = swiglu ? N / 2 : N;
    if (B == 1 && g_gemv_reg && (K == 1024 || K == 2048 || K == 8192)) {
        // one batch row: x in registers, no LDS, no barrier (gemv_reg_kernel); matrices of the 2048-wide stack are streamed once
        // per frame: non-temporal
        int grid = (no + 3) / 4;
        const bool nt = g_gemv_nt && (K == 2048 || (K == 8192 && N == 2048));
#define LR(KCH, T, SW, NT_) hipLaunchKernelGGL((gemv_reg_kernel<KCH, T, SW, NT_>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)W, (T*)y, (const bf16_t*)residual, N, ldw, ldx, ldy, (const bf16_t*)norm_w, eps, row_index, row_offset)
#define LK(T, SW, NT_) do { if (K == 1024) LR(2, T, SW, NT_); else if (K == 2048) LR(4, T, SW, NT_); else LR(16, T, SW, NT_); } while (0)
        if (swiglu && K == 1024 && !nt && g_gemv_rpw > 1 && no >= 2048) {
            // the depth decoder's w13: several gate/up pairs per wave (one normalisation of x per wave instead of per pair)
            if (g_gemv_rpw == 2) {
                grid = (no + 7) / 8;
                hipLaunchKernelGGL((gemv_reg_kernel<2, bf16_t, true, false, 2>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)W, (bf16_t*)y, (const bf16_t*)residual, N, ldw, ldx, ldy, (const bf16_t*)norm_w, eps, row_index, row_offset);
            } else {
                grid = (no + 15) / 16;
                hipLaunchKernelGGL((gemv_reg_kernel<2, bf16_t, true, false, 4>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)W, (bf16_t*)y, (const bf16_t*)residual, N, ldw, ldx, ldy, (const bf16_t*)norm_w, eps, row_index, row_offset);
            }
        }
        else if (swiglu) { if (nt) LK(bf16_t, true, true); else LK(bf16_t, true, false); }
        else if (out_f32) { if (nt) LK(float, false, true); else LK(float, false, false); }
        else { if (nt) LK(bf16_t, false, true); else LK(bf16_t, false, false); }
#undef LK
#undef LR
        CSM_CHECK_LAUNCH("csm_gemv_bf16");
        return 0;
    }
    if (B >= 2 && g_gemv_reg && g_gemv_regn && (K == 1024 || K == 2048 || (K == 8192 && !swiglu && !out_f32))) {
        // two to four batch rows, x in registers (gemv_regn_kernel)
        const int gridn = (no + 3) / 4;
        const bool nt = g_gemv_nt && (K == 2048 || (K == 8192 && N == 2048));
        const size_t ldsn = (size_t)B * K * sizeof(float);
#define LN(KCH, NB, T, SW, NT_, SC_) do { auto kf = gemv_regn_kernel<KCH, NB, T, SW, NT_, SC_>;                                       \
            if (ldsn > 65536) { static bool done_ = false; if (!done_) { (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn); done_ = true; } } \
            hipLaunchKernelGGL(kf, dim3(gridn), dim3(256), ldsn, stream, (const bf16_t*)x, (const bf16_t*)W, (T*)y, (const bf16_t*)residual, N, ldw, ldx, ldy, (const bf16_t*)norm_w, eps, row_index, row_offset); } while (0)
#define LNB(KCH, T, SW, NT_, SC_) do { if (B == 2) LN(KCH, 2, T, SW, NT_, SC_); else if (B == 3) LN(KCH, 3, T, SW, NT_, SC_); else LN(KCH, 4, T, SW, NT_, SC_); } while (0)
#define LNK(T, SW) do { if (K == 1024) LNB(2, T, SW, false, 2); else if (nt) LNB(4, T, SW, true, 4); else LNB(4, T, SW, false, 4); } while (0)
        if (K == 8192) { if (nt) LNB(16, bf16_t, false, true, 4); else LNB(16, bf16_t, false, false, 4); }
        else if (swiglu) LNK(bf16_t, true); else if (out_f32) LNK(float, false); else LNK(bf16_t, false);
#undef LNK
#undef LNB
#undef LN
        CSM_CHECK_LAUNCH("csm_gemv_bf16");
        return 0;
    }
    const int grid = no / 4 < 1 ? 1 : (no / 4 > 2048 ? 2048 : no / 4);
    const size_t lds = (size_t)B * K * 2;
#define L(NB, T, SW) hipLaunchKernelGGL((gemv_kernel<NB, T, SW>), dim3(grid), dim3(256), lds, stream, (const bf16_t*)x, (const bf16_t*)W, (T*)y, (const bf16_t*)residual, N, K, ldw, ldx, ldy, (const bf16_t*)norm_w, eps, row_index, row_offset)
    if (swiglu) { if (B == 1) L(1, bf16_t, true); else if (B == 2) L(2, bf16_t, true); else if (B == 3) L(3, bf16_t, true); else L(4, bf16_t, true); }
    else if (out_f32) { if (B == 1) L(1, float, false); else if (B == 2) L(2, float, false); else if (B == 3) L(3, float, false); else L(4, float, false); }
    else { if (B == 1) L(1, bf16_t, false); else if (B == 2) L(2, bf16_t, false); else if (B == 3) L(3, bf16_t, false); else L(4, bf16_t, false); }
#undef L
    CSM_CHECK_LAUNCH("csm_gemv_bf16");
    return 0;
}

extern "C" int csm_gemv_bf16(const void* x, const void* W, void* y, const void* residual, int B, int N, int K, int ldw,
                             int ldx, int ldy, int out_f32, hipStream_t stream) {
    return gemv_launch(x, W, y, residual, B, N, K, ldw, ldx, ldy, out_f32, nullptr, 0.f, 0, nullptr, 0, stream);
}

extern "C" int csm_gemv_bf16_ex(const void* x, const void* W, void* y, const void* residual, int B, int N, int K, int ldw,
                                int ldx, int ldy, int out_f32, const void* norm_scale, float eps, int swiglu, const int* row_index,
                                int row_offset, hipStream_t stream) {
    return gemv_launch(x, W, y, residual, B, N, K, ldw, ldx, ldy, out_f32, norm_scale, eps, swiglu, row_index, row_offset, stream);
}

extern "C" int csm_gemv_t_bf16(const void* x, const void* W, void* y, int B, int N, int K, int ldw, int ldx, int ldy,
                               int out_f32, hipStream_t stream) {
    CSM_REQUIRE(x && W && y && B >= 1 && B <= 4 && N > 0 && K > 0 && (ldw & 7) == 0, "csm_gemv_t_bf16: bad arguments");
    const int grid = ((N + 7) / 8 + 255) / 256;
#define L(NB, T) hipLaunchKernelGGL((gemv_t_kernel<NB, T>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)W, (T*)y, N, K, ldw, ldx, ldy)
    if (out_f32) { if (B == 1) L(1, float); else if (B == 2) L(2, float); else if (B == 3) L(3, float); else L(4, float); }
    else { if (B == 1) L(1, bf16_t); else if (B == 2) L(2, bf16_t); else if (B == 3) L(3, bf16_t); else L(4, bf16_t); }
#undef L
    CSM_CHECK_LAUNCH("csm_gemv_t_bf16");
    return 0;
}

extern "C" int csm_kv_append(const void* qkv, void* kcache, void* vcache, const int* pos, int B, int H, int KV, int HD,
                             int S_max, int ld, hipStream_t stream) {
    CSM_REQUIRE(qkv && kcache && vcache && pos && B > 0 && (HD & 7) == 0, "csm_kv_append: bad arguments");
    hipLaunchKernelGGL(kv_append_kernel, dim3(B), dim3(256), 0, stream, (const bf16_t*)qkv, (bf16_t*)kcache, (bf16_t*)vcache, pos, H,
                       KV, HD, S_max, ld);
    CSM_CHECK_LAUNCH("csm_kv_append");
    return 0;
}

static int attn_decode_launch(const void* qkv, void* kcache, void* vcache, void* out, const int* pos, const float* table, int B,
                              int H, int KV, int HD, int S_max, int ld, hipStream_t stream) {
    CSM_REQUIRE(qkv && kcache && vcache && out && pos && B > 0 && H > 0 && KV > 0 && H % KV == 0, "csm_attn_decode: bad arguments");
    CSM_REQUIRE(HD == 64 || HD == 128, "csm_attn_decode: head_dim %d unsupported", HD);
    CSM_REQUIRE(S_max <= 8192, "csm_attn_decode: S_max too large");
    const size_t lds = (size_t)(S_max + 16 + 6 * HD) * sizeof(float);
    const float scale = 1.f / sqrtf((float)HD);
    if (HD == 64)
        hipLaunchKernelGGL((attn_decode_kernel<64>), dim3(H, B), dim3(256), lds, stream, (const bf16_t*)qkv, (bf16_t*)kcache,
                           (bf16_t*)vcache, (bf16_t*)out, pos, H, KV, S_max, ld, scale, table);
    else
        hipLaunchKernelGGL((attn_decode_kernel<128>), dim3(H, B), dim3(256), lds, stream, (const bf16_t*)qkv, (bf16_t*)kcache,
                           (bf16_t*)vcache, (bf16_t*)out, pos, H, KV, S_max, ld, scale, table);
    CSM_CHECK_LAUNCH("csm_attn_decode");
    return 0;
}

extern "C" int csm_attn_decode(const void* qkv, const void* kcache, const void* vcache, void* out, const int* pos, int B, int H,
                               int KV, int HD, int S_max, int ld, hipStream_t stream) {
    return attn_decode_launch(qkv, const_cast<void*>(kcache), const_cast<void*>(vcache), out, pos, nullptr, B, H, KV, HD, S_max, ld,
                              stream);
}

extern "C" int csm_attn_decode_rope_at(const void* qkv, void* kcache, void* vcache, void* out, int pos, const float* rope_table,
                                       int B, int H, int KV, int HD, int S_max, int ld, hipStream_t stream) {
    CSM_REQUIRE(qkv && kcache && vcache && out && rope_table && B > 0, "csm_attn_decode_rope_at: bad arguments");
    CSM_REQUIRE(HD == 128 && H == 4 * KV && S_max >= 1 && S_max <= 32 && (ld & 7) == 0,
                "csm_attn_decode_rope_at: unsupported shape (H=%d KV=%d HD=%d S_max=%d: needs HD 128, H = 4 KV, S_max <= 32)", H, KV, HD, S_max);
    CSM_REQUIRE(pos >= 0 && pos < S_max, "csm_attn_decode_rope_at: position %d outside the cache (%d rows)", pos, S_max);
    hipLaunchKernelGGL((attn_decode_at_kernel<128>), dim3(KV, B), dim3(256), 0, stream, (const bf16_t*)qkv, (bf16_t*)kcache, (bf16_t*)vcache,
                       (bf16_t*)out, pos, rope_table, H, KV, S_max, ld, 1.f / sqrtf((float)HD));
    CSM_CHECK_LAUNCH("csm_attn_decode_rope_at");
    return 0;
}

extern "C" int csm_attn_decode_rope(const void* qkv, void* kcache, void* vcache, void* out, const int* pos, const float* rope_table,
                                    int B, int H, int KV, int HD, int S_max, int ld, hipStream_t stream) {
    CSM_REQUIRE(rope_table, "csm_attn_decode_rope: null table");
    return attn_decode_launch(qkv, kcache, vcache, out, pos, rope_table, B, H, KV, HD, S_max, ld, stream);
}

// Decoder layer: rotate + cache append + attention over <= 64 cached positions + output projection (+ residual) in ONE launch;
// bit-identical to csm_attn_decode_rope followed by csm_gemv_bf16 (same arithmetic, see gemv_attn_kernel).  HD = 128 only.
extern "C" int csm_gemv_attn_at_bf16(const void* qkv, void* kcache, void* vcache, int pos, const float* rope_table, const void* W,
                                     void* y, const void* residual, int N, int H, int KV, int HD, int S_max, int ldw, hipStream_t stream) {
    CSM_REQUIRE(qkv && kcache && vcache && rope_table && W && y, "csm_gemv_attn_at_bf16: null pointer");
    CSM_REQUIRE(N > 0 && H > 0 && H <= 8 && (H & 1) == 0 && KV > 0 && KV <= 2 && H % KV == 0 && ((H / KV) & 1) == 0 && HD == 128 && H * HD == 1024 && S_max >= 1 && S_max <= 32 && (ldw & 7) == 0,
                "csm_gemv_attn_at_bf16: unsupported shape (H=%d KV=%d HD=%d S_max=%d: needs H*HD = 1024, HD 128, S_max <= 32)", H, KV, HD, S_max);
    CSM_REQUIRE(pos >= 0 && pos < S_max, "csm_gemv_attn_at_bf16: position %d outside the cache (%d rows)", pos, S_max);
    const int K = H * HD;
    const size_t lds = (size_t)K * 2 + ((size_t)H * HD + 256 + 8) * sizeof(float) + (size_t)KV * 32 * (HD * 2 + 16) + (size_t)KV * 32 * HD * 2;
    const int grid = N / 8 < 1 ? 1 : (N / 8 > 2048 ? 2048 : N / 8);
    const float scale = 1.f / sqrtf((float)HD);
    hipLaunchKernelGGL((gemv_attn_at_kernel<128>), dim3(grid), dim3(512), lds, stream, (const bf16_t*)qkv, (bf16_t*)kcache, (bf16_t*)vcache,
                       pos, rope_table, (const bf16_t*)W, (bf16_t*)y, (const bf16_t*)residual, N, H, KV, S_max, ldw, scale);
    CSM_CHECK_LAUNCH("csm_gemv_attn_at_bf16");
    return 0;
}

extern "C" int csm_gemv_attn_bf16(const void* qkv, void* kcache, void* vcache, const int* pos, const float* rope_table, const void* W,
                                  void* y, const void* residual, int B, int N, int H, int KV, int HD, int S_max, int ld_qkv, int ldw,
                                  int ldy, hipStream_t stream) {
    CSM_REQUIRE(qkv && kcache && vcache && pos && rope_table && W && y, "csm_gemv_attn_bf16: null pointer");
    CSM_REQUIRE(B >= 1 && B <= 4 && N > 0 && H > 0 && KV > 0 && H % KV == 0 && HD == 128 && S_max >= 1 && S_max <= 64 && (ldw & 7) == 0 && (ld_qkv & 7) == 0,
                "csm_gemv_attn_bf16: unsupported shape (B=%d H=%d KV=%d HD=%d S_max=%d: needs HD 128, S_max <= 64)", B, H, KV, HD, S_max);
    const int K = H * HD;
    const size_t lds = (size_t)B * K * 2 + ((size_t)B * H * HD + (size_t)B * KV * HD + 512) * sizeof(float);
    CSM_REQUIRE(lds <= 65536, "csm_gemv_attn_bf16: %zu bytes of LDS needed", lds);
    const int grid = N / 8 < 1 ? 1 : (N / 8 > 2048 ? 2048 : N / 8);      // 8 waves per workgroup: one (row, head) each in the prologue
    const float scale = 1.f / sqrtf((float)HD);
#define L(NB) hipLaunchKernelGGL((gemv_attn_kernel<NB, 128>), dim3(grid), dim3(512), lds, stream, (const bf16_t*)qkv, (bf16_t*)kcache, (bf16_t*)vcache, pos, rope_table, (const bf16_t*)W, (bf16_t*)y, (const bf16_t*)residual, N, H, KV, S_max, ld_qkv, ldw, ldy, scale)
    if (B == 1) L(1); else if (B == 2) L(2); else if (B == 3) L(3); else L(4);
#undef L
    CSM_CHECK_LAUNCH("csm_gemv_attn_bf16");
    return 0;
}
