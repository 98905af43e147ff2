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

// ================================================================================================ batch-1 decode path
// Reference Model.generate_frame (src/csm/models/model.py:140-195) runs one backbone position and 31 decoder positions
// per 80-ms frame against KV caches: every projection is a matrix-VECTOR product bound by weight streaming
// (1.95 GB + 31 x 0.22 GB of bf16 weights per frame), attention is a cache read.  These kernels keep the weights'
// [N][K] layout, load 16 B per lane straight to registers (an LDS round trip is pure overhead when nothing is shared
// between waves) and reduce with wave shuffles.
namespace {

// y[b][n] = sum_k x[b][k] * W[n][k] (+ R[b][n]);  one wave per output row, NB <= 4 batch rows share every weight load.
template <int NB, typename OutT>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W, OutT* __restrict__ y,
                                                   const bf16_t* __restrict__ R, int N, int K, int ldw, int ldx, int ldy) {
    extern __shared__ __attribute__((aligned(16))) char smem_x[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem_x);            // [NB][K]
    for (int i = threadIdx.x * 8; i < NB * K; i += blockDim.x * 8) {
        const int b = i / K, k = i - b * K;
        *reinterpret_cast<U4*>(xs + i) = *reinterpret_cast<const U4*>(x + (size_t)b * ldx + k);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int n = blockIdx.x * wpb + (threadIdx.x >> 6); n < N; n += gridDim.x * wpb) {
        float acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = 0.f;
        const bf16_t* w = W + (size_t)n * ldw;
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
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float s = wave_sum(acc[b]);
            if (lane == 0) {
                float v = s;
                if (R) v += bf2f(R[(size_t)b * ldy + n]);
                if constexpr (sizeof(OutT) == 2) y[(size_t)b * ldy + n] = f2bf(v);
                else y[(size_t)b * ldy + n] = v;
            }
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
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ kc,
                                                          const bf16_t* __restrict__ vc, bf16_t* __restrict__ out,
                                                          const int* __restrict__ pos, int H, int KV, int S_max, int ld,
                                                          float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_a[];
    float* sc = reinterpret_cast<float*>(smem_a);                 // [S_max] scores, then probabilities
    float* red = sc + S_max;                                       // 16 floats
    float* part = red + 16;                                        // [4 waves][HD] partial outputs
    const int h = blockIdx.x, b = blockIdx.y;
    const int kvh = h / (H / KV);
    const int n = pos[b] + 1;                                      // keys 0 .. pos (the new one was appended already)
    const bf16_t* q = qkv + (size_t)b * ld + h * HD;
    const bf16_t* K = kc + ((size_t)b * KV + kvh) * S_max * HD;
    const bf16_t* V = vc + ((size_t)b * KV + kvh) * S_max * HD;
    float qf[HD];
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) unpack8(*reinterpret_cast<const U4*>(q + c * 8), qf + c * 8);
    float mx = -INFINITY;
    for (int s = threadIdx.x; s < n; s += blockDim.x) {
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            float kf[8];
            unpack8(*reinterpret_cast<const U4*>(K + (size_t)s * HD + c * 8), kf);
#pragma unroll
            for (int j = 0; j < 8; ++j) d += qf[c * 8 + j] * kf[j];
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
        unpack8(*reinterpret_cast<const U4*>(V + (size_t)s * HD + sub * 8), vf);
        const float p = sc[s];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += p * vf[j];
    }
    // reduce the ngrp partials: first inside a wave (groups of a wave differ in lane bits >= log2(LPR)), then across waves
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
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

}  // namespace

extern "C" int csm_gemv_bf16(const void* x, const void* W, void* y, const void* residual, int B, int N, int K, int ldw,
                             int ldx, int ldy, int out_f32, hipStream_t stream) {
    CSM_REQUIRE(x && W && y && B >= 1 && B <= 4 && N > 0 && K > 0 && (K & 7) == 0 && (ldw & 7) == 0 && (ldx & 7) == 0,
                "csm_gemv_bf16: bad arguments (B=%d N=%d K=%d)", B, N, K);
    CSM_REQUIRE((size_t)B * K * 2 <= 65536, "csm_gemv_bf16: B*K too large for the LDS copy of x");
    const int grid = N / 4 < 1 ? 1 : (N / 4 > 2048 ? 2048 : N / 4);
    const size_t lds = (size_t)B * K * 2;
#define L(NB, T) hipLaunchKernelGGL((gemv_kernel<NB, T>), dim3(grid), dim3(256), lds, stream, (const bf16_t*)x, (const bf16_t*)W, (T*)y, (const bf16_t*)residual, N, K, ldw, ldx, ldy)
    if (out_f32) { if (B == 1) L(1, float); else if (B == 2) L(2, float); else if (B == 3) L(3, float); else L(4, float); }
    else { if (B == 1) L(1, bf16_t); else if (B == 2) L(2, bf16_t); else if (B == 3) L(3, bf16_t); else L(4, bf16_t); }
#undef L
    CSM_CHECK_LAUNCH("csm_gemv_bf16");
    return 0;
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

extern "C" int csm_attn_decode(const void* qkv, const void* kcache, const void* vcache, void* out, const int* pos, int B, int H,
                               int KV, int HD, int S_max, int ld, hipStream_t stream) {
    CSM_REQUIRE(qkv && kcache && vcache && out && pos && B > 0 && H > 0 && KV > 0 && H % KV == 0, "csm_attn_decode: bad arguments");
    CSM_REQUIRE(HD == 64 || HD == 128, "csm_attn_decode: head_dim %d unsupported", HD);
    CSM_REQUIRE(S_max <= 8192, "csm_attn_decode: S_max too large");
    const size_t lds = (size_t)(S_max + 16 + 4 * HD) * sizeof(float);
    const float scale = 1.f / sqrtf((float)HD);
    if (HD == 64)
        hipLaunchKernelGGL((attn_decode_kernel<64>), dim3(H, B), dim3(256), lds, stream, (const bf16_t*)qkv, (const bf16_t*)kcache,
                           (const bf16_t*)vcache, (bf16_t*)out, pos, H, KV, S_max, ld, scale);
    else
        hipLaunchKernelGGL((attn_decode_kernel<128>), dim3(H, B), dim3(256), lds, stream, (const bf16_t*)qkv, (const bf16_t*)kcache,
                           (const bf16_t*)vcache, (bf16_t*)out, pos, H, KV, S_max, ld, scale);
    CSM_CHECK_LAUNCH("csm_attn_decode");
    return 0;
}
