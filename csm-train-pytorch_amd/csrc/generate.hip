// Generation-side kernels for gfx950: top-k / temperature sampler with injected Exp(1) noise, and the Mimi
// split residual vector quantiser (nearest-codeword encode, codebook-sum decode).
#include "common.h"
#include <math.h>

namespace {

struct ValIdx { float v; int i; };

__device__ __forceinline__ ValIdx vi_max(ValIdx a, ValIdx b) {  // larger value wins, then the lower index
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ValIdx vi_min(ValIdx a, ValIdx b) {  // smaller value wins, then the lower index
    return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

template <bool MAX>
__device__ __forceinline__ ValIdx block_arg(ValIdx x, ValIdx* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ValIdx y;
        y.v = __shfl_xor(x.v, o, 64);
        y.i = __shfl_xor(x.i, o, 64);
        x = MAX ? vi_max(x, y) : vi_min(x, y);
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = x;
    __syncthreads();
    ValIdx r = red[0];
    for (int k = 1; k < nw; ++k) r = MAX ? vi_max(r, red[k]) : vi_min(r, red[k]);
    return r;
}

// reference src/csm/models/model.py:79-96: logits/T, keep values >= k-th largest, log_softmax -> softmax,
// argmax(p / q) with q ~ Exp(1) supplied by the caller.  One block per row, V <= 256*16.
constexpr int SMP_PER_THREAD = 16;
__global__ __launch_bounds__(256) void sample_topk_kernel(const float* __restrict__ logits, const float* __restrict__ q,
                                                          int* __restrict__ out, int V, int ldl, int topk, float temperature) {
    __shared__ ValIdx red[4];
    __shared__ float fred[16];
    const int row = blockIdx.x;
    const float* x = logits + (size_t)row * ldl;
    const float* qq = q + (size_t)row * V;
    float val[SMP_PER_THREAD];
    bool removed[SMP_PER_THREAD];
#pragma unroll
    for (int j = 0; j < SMP_PER_THREAD; ++j) {
        const int c = threadIdx.x + 256 * j;
        val[j] = c < V ? x[c] / temperature : -INFINITY;
        removed[j] = c >= V;
    }
    // k-th largest by removing one maximum per round (ties are distinct elements, as in torch.topk)
    float kth = -INFINITY, top = -INFINITY;
    for (int k = 0; k < topk; ++k) {
        ValIdx best = {-INFINITY, 0x7fffffff};
#pragma unroll
        for (int j = 0; j < SMP_PER_THREAD; ++j)
            if (!removed[j]) best = vi_max(best, (ValIdx){val[j], (int)threadIdx.x + 256 * j});
        best = block_arg<true>(best, red);
        if (k == 0) top = best.v;
        kth = best.v;
#pragma unroll
        for (int j = 0; j < SMP_PER_THREAD; ++j)
            if ((int)threadIdx.x + 256 * j == best.i) removed[j] = true;
    }
    // log_softmax over kept values, then softmax of that (torch evaluates both)
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < SMP_PER_THREAD; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < V && val[j] >= kth) s += expf(val[j] - top);
    }
    s = block_sum(s, fred);
    const float logsum = logf(s);
    const float ymax = (top - top) - logsum;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < SMP_PER_THREAD; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < V && val[j] >= kth) s2 += expf(((val[j] - top) - logsum) - ymax);
    }
    s2 = block_sum(s2, fred);
    ValIdx best = {-INFINITY, 0x7fffffff};
#pragma unroll
    for (int j = 0; j < SMP_PER_THREAD; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < V) {
            float p = 0.f;
            if (val[j] >= kth) p = expf(((val[j] - top) - logsum) - ymax) / s2;
            best = vi_max(best, (ValIdx){p / qq[c], c});
        }
    }
    best = block_arg<true>(best, red);
    if (threadIdx.x == 0) out[row] = best.i;
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
    hipLaunchKernelGGL(sample_topk_kernel, dim3(rows), dim3(256), 0, stream, logits, q, out, V, ldl, topk, temperature);
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
