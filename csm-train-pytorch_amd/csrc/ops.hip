// HBM-bound kernels of the CSM train step for gfx950: RMSNorm, RoPE, SwiGLU, masked multi-codebook embedding,
// fused softmax-cross-entropy, gradient-norm / clip and AdamW.  All of them move 16 bytes per lane per access
// and reduce with 64-lane shuffles; none stages through LDS except for block-level scalars.
#include "common.h"
#include <math.h>

namespace {

// ------------------------------------------------------------------------------------------------ RMSNorm
// torchtune RMSNorm (appendix A of SURVEY.md; built by reference src/csm/models/model.py:13-42):
//   y = x * rsqrt(mean(x^2) + eps) * scale, statistics in fp32.  One wave per row, row held in registers.
template <int NC>  // 16-B chunks per lane: D <= NC*512
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd, int M, int D,
                                                          float eps) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int nchunk = D >> 3;
    for (long long row = (long long)blockIdx.x * wpb + (threadIdx.x >> 6); row < M; row += (long long)gridDim.x * wpb) {
        U4 v[NC];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            v[i] = (U4){0u, 0u, 0u, 0u};
            if (c < nchunk) v[i] = *reinterpret_cast<const U4*>(x + (size_t)row * D + c * 8);
            float f[8];
            unpack8(v[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
        }
        ss = wave_sum(ss);
        const float r = rsqrtf(ss / (float)D + eps);
        if (lane == 0 && rstd) rstd[row] = r;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float f[8], s[8];
                unpack8(v[i], f);
                unpack8(*reinterpret_cast<const U4*>(w + c * 8), s);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = f[j] * r * s[j];
                *reinterpret_cast<U4*>(y + (size_t)row * D + c * 8) = pack8(f);
            }
        }
    }
}

// dx = rstd * (dy*w) - x * rstd^3/D * sum(dy*w*x) (+ dres);  per-block partial of dw = sum_rows dy * x * rstd
// One wave per row, 8 waves per block, one block per CU: all three row streams (x, dy, dres) are requested before the
// reduction so 12 x 16 B per lane are in flight, instead of fetching dres after the row's dot product is known.
template <int NC>
__global__ __launch_bounds__(512) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          const float* __restrict__ rstd, const bf16_t* __restrict__ dy,
                                                          const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                          float* __restrict__ dw_acc, int M, int D) {
    extern __shared__ __attribute__((aligned(16))) char smem_dw[];
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int nchunk = D >> 3;
    float dwl[NC][8];
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) dwl[i][j] = 0.f;
    U4 wq[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        wq[i] = (U4){0u, 0u, 0u, 0u};
        if (c < nchunk) wq[i] = *reinterpret_cast<const U4*>(w + c * 8);
    }
    for (long long row = (long long)blockIdx.x * wpb + (threadIdx.x >> 6); row < M; row += (long long)gridDim.x * wpb) {
        U4 xa[NC], ga[NC], ra[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            xa[i] = ga[i] = ra[i] = (U4){0u, 0u, 0u, 0u};
            if (c < nchunk) {
                xa[i] = *reinterpret_cast<const U4*>(x + (size_t)row * D + c * 8);
                ga[i] = *reinterpret_cast<const U4*>(dy + (size_t)row * D + c * 8);
                if (dres) ra[i] = *reinterpret_cast<const U4*>(dres + (size_t)row * D + c * 8);
            }
        }
        const float r = rstd[row];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            float xs[8], gs[8], ws[8];
            unpack8(xa[i], xs); unpack8(ga[i], gs); unpack8(wq[i], ws);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gx = gs[j] * xs[j];
                if (dw_acc) dwl[i][j] += gx * r;
                dot += gx * ws[j];
            }
        }
        dot = wave_sum(dot);
        const float k = r * r * r * dot / (float)D;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float xs[8], gs[8], ws[8], rr[8], o[8];
                unpack8(xa[i], xs); unpack8(ga[i], gs); unpack8(wq[i], ws); unpack8(ra[i], rr);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = r * (gs[j] * ws[j]) - k * xs[j] + rr[j];
                *reinterpret_cast<U4*>(dx + (size_t)row * D + c * 8) = pack8(o);
            }
        }
    }
    if (dw_acc) {
        // block-level reduction through LDS, then one coalesced partial row per block: dw_acc[blockIdx.x][D]
        float* red = reinterpret_cast<float*>(smem_dw);
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) red[wv * D + c * 8 + j] = dwl[i][j];
            }
        }
        __syncthreads();
        for (int col = threadIdx.x; col < D; col += blockDim.x) {
            float t = 0.f;
            for (int k = 0; k < wpb; ++k) t += red[k * D + col];
            dw_acc[(size_t)blockIdx.x * D + col] = t;
        }
    }
}

// dst(bf16)[col] (+)= sum_r partials[r][col]; block = 64 columns x (blockDim.x / 64) row slices (coalesced 256-B row
// reads): 4 slices for the few-row split-K slabs, 16 for the 256 per-block partial rows of the RMSNorm backward, where
// only D / 64 blocks exist and the row loop is the whole latency.
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ partials, int rows, int D, bf16_t* __restrict__ dst,
                                                      int accumulate) {
    __shared__ float red[16][64];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6, ns = blockDim.x >> 6;
    const int col = blockIdx.x * 64 + c;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (col < D) {
        int r = sl;
        for (; r + 3 * ns < rows; r += 4 * ns) {
            t0 += partials[(size_t)r * D + col];
            t1 += partials[(size_t)(r + ns) * D + col];
            t2 += partials[(size_t)(r + 2 * ns) * D + col];
            t3 += partials[(size_t)(r + 3 * ns) * D + col];
        }
        for (; r < rows; r += ns) t0 += partials[(size_t)r * D + col];
    }
    red[sl][c] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (sl == 0 && col < D) {
        float t = 0.f;
        for (int k = 0; k < ns; ++k) t += red[k][c];
        if (accumulate) t += bf2f(dst[col]);
        dst[col] = f2bf(t);
    }
}

// the same reduction for up to 8 (partials, dst) pairs of one shape in one launch (blockIdx.y = pair): the RMSNorm scale gradients
// of the layers whose weight gradients the engine launches together
struct ColsumMulti { const float* partials[8]; bf16_t* dst[8]; };
__global__ __launch_bounds__(1024) void colsum_multi_kernel(ColsumMulti p, int rows, int D, int accumulate) {
    __shared__ float red[16][64];
    const float* __restrict__ partials = p.partials[blockIdx.y];
    bf16_t* __restrict__ dst = p.dst[blockIdx.y];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6, ns = blockDim.x >> 6;
    const int col = blockIdx.x * 64 + c;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (col < D) {
        int r = sl;
        for (; r + 3 * ns < rows; r += 4 * ns) {
            t0 += partials[(size_t)r * D + col];
            t1 += partials[(size_t)(r + ns) * D + col];
            t2 += partials[(size_t)(r + 2 * ns) * D + col];
            t3 += partials[(size_t)(r + 3 * ns) * D + col];
        }
        for (; r < rows; r += ns) t0 += partials[(size_t)r * D + col];
    }
    red[sl][c] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (sl == 0 && col < D) {
        float t = 0.f;
        for (int k = 0; k < ns; ++k) t += red[k][c];
        if (accumulate) t += bf2f(dst[col]);
        dst[col] = f2bf(t);
    }
}

// ------------------------------------------------------------------------------------------------ LoRA side ops
// Inverted dropout on a [M, D] bf16 matrix with row stride ld (reference lora.py:88-90, mlx nn.dropout: kept values are
// scaled by 1/(1-p)).  The keep decision for element (row, col) is a pure function of (seed, row*D + col) - a
// splitmix64 finaliser, 16 bits per element against the threshold p*65536 - so the backward regenerates the very same
// mask from the seed instead of storing it.  accumulate = 1: out += dropout(in).
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* __restrict__ in, int ld_in, bf16_t* __restrict__ out, int ld_out,
                                                      long long M, int D, uint32_t thresh, float scale, uint64_t seed, int accumulate) {
    const int cpr = D >> 3;                                  // 16-byte chunks per row
    const long long total = M * cpr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cpr;
        const int ch = (int)(i - r * cpr);
        const uint64_t e = (uint64_t)(r * D + ch * 8) >> 2;   // one 64-bit draw covers 4 elements
        const uint64_t h0 = mix64(seed ^ (e * 0xd1342543de82ef95ull)), h1 = mix64(seed ^ ((e + 1) * 0xd1342543de82ef95ull));
        float f[8], g[8];
        unpack8(*reinterpret_cast<const U4*>(in + r * ld_in + ch * 8), f);
        if (accumulate) unpack8(*reinterpret_cast<const U4*>(out + r * ld_out + ch * 8), g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t u = (uint32_t)(((j < 4 ? h0 : h1) >> (16 * (j & 3))) & 0xffffu);
            const float v = u >= thresh ? f[j] * scale : 0.f;
            f[j] = accumulate ? g[j] + v : v;
        }
        *reinterpret_cast<U4*>(out + r * ld_out + ch * 8) = pack8(f);
    }
}

// y[r, :] += bias for a [M, D] bf16 matrix with row stride ld (LoRA bias, reference lora.py:100-102)
__global__ __launch_bounds__(256) void bias_add_kernel(bf16_t* __restrict__ y, int ld, const bf16_t* __restrict__ bias, long long M, int D) {
    const int cpr = D >> 3;
    const long long total = M * cpr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cpr;
        const int ch = (int)(i - r * cpr);
        float f[8], b[8];
        unpack8(*reinterpret_cast<const U4*>(y + r * ld + ch * 8), f);
        unpack8(*reinterpret_cast<const U4*>(bias + ch * 8), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += b[j];
        *reinterpret_cast<U4*>(y + r * ld + ch * 8) = pack8(f);
    }
}

// partial column sums of a bf16 [M, D] matrix (row stride ld): block (cb, sl) sums rows sl, sl+S, ... of 64 columns
// into partials[sl][col] (fp32); colsum_kernel finishes.  Used for d(bias) = sum_rows dy.
__global__ __launch_bounds__(256) void colsum_rows_kernel(const bf16_t* __restrict__ x, int ld, long long M, int D,
                                                          float* __restrict__ partials) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int sl = blockIdx.y, S = gridDim.y;
    float t = 0.f;
    if (col < D)
        for (long long r = (long long)sl * 4 + q; r < M; r += (long long)S * 4) t += bf2f(x[r * ld + col]);
    red[q][c] = t;
    __syncthreads();
    if (q == 0 && col < D) partials[(size_t)sl * D + col] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// ------------------------------------------------------------------------------------------------ RoPE
// torchtune Llama3ScaledRoPE on the fused qkv buffer, in place: interleaved pairs (2i, 2i+1) of every q and k
// head rotated by pos*theta'_i (table [P][hd/2][2] = cos,sin fp32, built on the host exactly as the oracle does).
// inverse = 1 applies the transpose rotation (backward).
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ table,
                                                   const int* __restrict__ pos, long long M, int S, int nheads /*H+KV*/,
                                                   int hd, int ld, int inverse) {
    const int cph = hd >> 3;                         // chunks per head
    const long long per_row = (long long)nheads * cph;
    const long long total = M * per_row;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / per_row;
        const int rem = (int)(idx % per_row);
        const int head = rem / cph, c = rem % cph;
        const int p = pos ? pos[row] : (int)(row % S);
        bf16_t* ptr = qkv + (size_t)row * ld + head * hd + c * 8;
        float f[8], o[8];
        unpack8(*reinterpret_cast<const U4*>(ptr), f);
        const float4 t0 = *reinterpret_cast<const float4*>(table + ((size_t)p * (hd >> 1) + c * 4) * 2);
        const float4 t1 = *reinterpret_cast<const float4*>(table + ((size_t)p * (hd >> 1) + c * 4) * 2 + 4);
        const float cs[4] = {t0.x, t0.z, t1.x, t1.z};
        float sn[4] = {t0.y, t0.w, t1.y, t1.w};
        if (inverse) { sn[0] = -sn[0]; sn[1] = -sn[1]; sn[2] = -sn[2]; sn[3] = -sn[3]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = f[2 * i]; o[2 * i + 1] = f[2 * i + 1];
            rope_rot(o[2 * i], o[2 * i + 1], cs[i], sn[i]);
        }
        *reinterpret_cast<U4*>(ptr) = pack8(o);
    }
}

// ------------------------------------------------------------------------------------------------ SwiGLU
// torchtune FeedForward: w2(silu(w1 x) * w3 x).  gu = fused w1/w3 GEMM output with gate and up INTERLEAVED along the
// feature axis ([M][2F]: g0,u0,g1,u1,...), the layout the GEMM's SwiGLU epilogues use.  These stand-alone kernels
// serve the paths the fused epilogues do not cover (LoRA adapters on w1/w3/w2).
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ out, long long M,
                                                         int F) {
    const int cpr = F >> 2;                       // 4 features (8 interleaved values) per thread
    const long long total = M * cpr;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / cpr;
        const int c = (int)(idx % cpr);
        float g[8];
        unpack8(*reinterpret_cast<const U4*>(gu + (size_t)row * 2 * F + c * 8), g);
        uint2 o;
        o.x = pack2bf(silu(g[0]) * g[1], silu(g[2]) * g[3]);
        o.y = pack2bf(silu(g[4]) * g[5], silu(g[6]) * g[7]);
        *reinterpret_cast<uint2*>(out + (size_t)row * F + c * 4) = o;
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dout,
                                                         bf16_t* __restrict__ dgu, long long M, int F) {
    const int cpr = F >> 2;
    const long long total = M * cpr;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / cpr;
        const int c = (int)(idx % cpr);
        float g[8], o[8];
        unpack8(*reinterpret_cast<const U4*>(gu + (size_t)row * 2 * F + c * 8), g);
        const uint2 d2 = *reinterpret_cast<const uint2*>(dout + (size_t)row * F + c * 4);
        const float d[4] = {__uint_as_float(d2.x << 16), __uint_as_float(d2.x & 0xffff0000u), __uint_as_float(d2.y << 16),
                            __uint_as_float(d2.y & 0xffff0000u)};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float gt = g[2 * i], up = g[2 * i + 1];
            const float sg = fast_sigmoid(gt);
            o[2 * i] = d[i] * up * sg * (1.f + gt * (1.f - sg));
            o[2 * i + 1] = d[i] * gt * sg;
        }
        *reinterpret_cast<U4*>(dgu + (size_t)row * 2 * F + c * 8) = pack8(o);
    }
}

// ------------------------------------------------------------------------------------------------ embedding
// reference src/csm/models/model.py:202-217 + mask-mul-sum src/csm/training/utils.py:85-87:
//   h[row] = sum over live slots c of  (c < K ? audio_emb[tok + c*V_a] : text_emb[tok]).   One block per row.
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ tokens, const uint8_t* __restrict__ mask,
                                                        const bf16_t* __restrict__ text_emb, const bf16_t* __restrict__ audio_emb,
                                                        bf16_t* __restrict__ out, int K, int D, int audio_vocab) {
    const long long row = blockIdx.x;
    const long long* tok = tokens + row * (K + 1);
    const uint8_t* mk = mask + row * (K + 1);
    for (int c0 = threadIdx.x * 8; c0 < D; c0 += blockDim.x * 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s <= K; ++s) {
            if (!mk[s]) continue;
            const bf16_t* src = (s < K) ? audio_emb + ((size_t)tok[s] + (size_t)s * audio_vocab) * D : text_emb + (size_t)tok[s] * D;
            float f[8];
            unpack8(*reinterpret_cast<const U4*>(src + c0), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
        *reinterpret_cast<U4*>(out + (size_t)row * D + c0) = pack8(acc);
    }
}

// Deterministic, scratch-free embedding backward: the (row, source) occurrence list is sorted by embedding row on the
// host side (torch.sort, no sync); one wave per occurrence index, only the FIRST occurrence of a row does work: it sums
// the bf16 gradient rows of every occurrence of that row in fp32 registers (fixed order) and adds the result into the
// bf16 gradient table once.  Sources: index < M -> dh[index] (backbone input gradient), else dseq[index - M]
// (depth-decoder input gradient).  Rows >= n_rows are padding (masked-out slots).
template <int NC>
__global__ __launch_bounds__(256) void embed_bwd_sorted_kernel(const long long* __restrict__ rows, const long long* __restrict__ src,
                                                               long long n_occ, const bf16_t* __restrict__ dh,
                                                               const bf16_t* __restrict__ dseq, long long M,
                                                               bf16_t* __restrict__ g_text, bf16_t* __restrict__ g_audio,
                                                               long long text_rows, long long n_rows, int D) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n_occ) return;
    const long long r = rows[i];
    if (r >= n_rows || (i > 0 && rows[i - 1] == r)) return;      // padding, or not the first occurrence of this row
    float acc[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
    const int nchunk = D >> 3;
    for (long long k = i; k < n_occ && rows[k] == r; ++k) {
        const long long sidx = src[k];
        const bf16_t* p = sidx < M ? dh + (size_t)sidx * D : dseq + (size_t)(sidx - M) * D;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nchunk) {
                float f[8];
                unpack8(*reinterpret_cast<const U4*>(p + ch * 8), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[c][j] += f[j];
            }
        }
    }
    bf16_t* dst = r < text_rows ? g_text + (size_t)r * D : g_audio + (size_t)(r - text_rows) * D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
            float f[8];
            unpack8(*reinterpret_cast<const U4*>(dst + ch * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += acc[c][j];
            *reinterpret_cast<U4*>(dst + ch * 8) = pack8(f);
        }
    }
}

// dst[rows[n]][:] += src[n * src_stride_rows][:]   (rows are unique: no atomics; one wave per n; rows[n] < 0 = padding, skipped)
__global__ __launch_bounds__(256) void rows_add_kernel(bf16_t* __restrict__ dst, const int* __restrict__ rows,
                                                       const bf16_t* __restrict__ src, long long N, int src_stride_rows, int D) {
    const int lane = threadIdx.x & 63;
    const long long n = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    if (rows[n] < 0) return;
    bf16_t* d = dst + (size_t)rows[n] * D;
    const bf16_t* s = src + (size_t)n * src_stride_rows * D;
    for (int c0 = lane * 8; c0 < D; c0 += 512) {
        float a[8], b[8];
        unpack8(*reinterpret_cast<const U4*>(d + c0), a);
        unpack8(*reinterpret_cast<const U4*>(s + c0), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j];
        *reinterpret_cast<U4*>(d + c0) = pack8(a);
    }
}

// out[n][:] = table[rows[n]][:]; table[rows[n]][:] = 0   (rows unique; rows[n] < 0 = padding: out[n] = 0).  The data-parallel
// exchange of the text-embedding gradient rows: a rank lifts the rows it touched out of its gradient table, every rank
// then adds all ranks' rows back in rank order (csm_rows_add_bf16), which leaves bit-identical tables everywhere.
__global__ __launch_bounds__(256) void rows_take_kernel(bf16_t* __restrict__ table, const int* __restrict__ rows,
                                                        bf16_t* __restrict__ out, long long N, int D) {
    const int lane = threadIdx.x & 63;
    const long long n = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const int r = rows[n];
    bf16_t* o = out + (size_t)n * D;
    const U4 z = {0u, 0u, 0u, 0u};
    for (int c0 = lane * 8; c0 < D; c0 += 512) {
        if (r < 0) { *reinterpret_cast<U4*>(o + c0) = z; continue; }
        bf16_t* t = table + (size_t)r * D + c0;
        *reinterpret_cast<U4*>(o + c0) = *reinterpret_cast<const U4*>(t);
        *reinterpret_cast<U4*>(t) = z;
    }
}

// build the depth-decoder input rows [N][K][D]: position 0 = backbone state h[row], position i>=1 = audio_emb of
// code i-1 of the target frame (teacher forcing of reference src/csm/models/model.py:175-189).
__global__ __launch_bounds__(256) void decoder_input_kernel(const bf16_t* __restrict__ hidden, const int* __restrict__ rows,
                                                            const long long* __restrict__ codes /*[N][K]*/,
                                                            const bf16_t* __restrict__ audio_emb, bf16_t* __restrict__ out,
                                                            int K, int D, int audio_vocab) {
    const long long n = blockIdx.x / K;
    const int i = blockIdx.x % K;
    const bf16_t* src = (i == 0) ? hidden + (size_t)rows[n] * D
                                 : audio_emb + ((size_t)codes[n * K + i - 1] + (size_t)(i - 1) * audio_vocab) * D;
    for (int c0 = threadIdx.x * 8; c0 < D; c0 += blockDim.x * 8)
        *reinterpret_cast<U4*>(out + ((size_t)n * K + i) * D + c0) = *reinterpret_cast<const U4*>(src + c0);
}

// ------------------------------------------------------------------------------------------------ cross-entropy
// F.cross_entropy(mean) of reference src/csm/training/utils.py:102-105, fused with its backward:
// loss_row = logsumexp(x) - x[t];  dlogits = (softmax(x) - onehot(t)) * gscale, bf16, pad columns zeroed.
// target < 0 marks a row that is not part of the loss (last position of each sequence).
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const long long* __restrict__ targets,
                                                 float* __restrict__ loss_rows, bf16_t* __restrict__ dlogits, int V, int ldl,
                                                 int ldd, float gscale) {
    __shared__ float red[16];
    const long long row = blockIdx.x;
    const float* x = logits + (size_t)row * ldl;
    const long long t = targets[row];
    if (t < 0) {
        if (threadIdx.x == 0) loss_rows[row] = 0.f;
        if (dlogits)
            for (int c = threadIdx.x; c < ldd; c += blockDim.x) dlogits[(size_t)row * ldd + c] = 0;
        return;
    }
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < V; c += blockDim.x) mx = fmaxf(mx, x[c]);
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int c = threadIdx.x; c < V; c += blockDim.x) sum += __expf(x[c] - mx);
    sum = block_sum(sum, red);
    const float lse = mx + logf(sum);
    if (threadIdx.x == 0) loss_rows[row] = lse - x[t];
    if (dlogits) {
        for (int c = threadIdx.x; c < ldd; c += blockDim.x) {
            float g = 0.f;
            if (c < V) g = (__expf(x[c] - lse) - (c == t ? 1.f : 0.f)) * gscale;
            dlogits[(size_t)row * ldd + c] = f2bf(g);
        }
    }
}

// The same with ONE WAVE per row and the row in registers (NC float4 per lane): one pass over the logits instead of three
// block-wide ones with two barriers each, 16-byte loads, 8-byte stores.  Used when the row fits (V <= 256 NC) and the
// leading dimensions allow the vector accesses.
template <int NC>
__global__ __launch_bounds__(256) void ce_wave_kernel(const float* __restrict__ logits, const long long* __restrict__ targets,
                                                      float* __restrict__ loss_rows, bf16_t* __restrict__ dlogits, long long R, int V,
                                                      int ldl, int ldd, float gscale) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* x = logits + (size_t)row * ldl;
    const long long t = targets[row];
    if (t < 0) {
        if (lane == 0) loss_rows[row] = 0.f;
        if (dlogits)
            for (int c = lane * 4; c < ldd; c += 256) *reinterpret_cast<uint2*>(dlogits + (size_t)row * ldd + c) = make_uint2(0u, 0u);
        return;
    }
    float v[NC][4];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = (lane + 64 * i) * 4;
        float4 q = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (c < ldl) q = *reinterpret_cast<const float4*>(x + c);
        v[i][0] = c + 0 < V ? q.x : -INFINITY; v[i][1] = c + 1 < V ? q.y : -INFINITY;
        v[i][2] = c + 2 < V ? q.z : -INFINITY; v[i][3] = c + 3 < V ? q.w : -INFINITY;
        mx = fmaxf(fmaxf(mx, fmaxf(v[i][0], v[i][1])), fmaxf(v[i][2], v[i][3]));
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += __expf(v[i][j] - mx);        // exp(-inf) = 0 for the masked columns
    sum = wave_sum(sum);
    const float lse = mx + logf(sum);
    if (lane == 0) loss_rows[row] = lse - x[t];
    if (dlogits) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < ldd) {
                float g[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = c + j < V ? (__expf(v[i][j] - lse) - (c + j == t ? 1.f : 0.f)) * gscale : 0.f;
                uint2 o; o.x = pack2bf(g[0], g[1]); o.y = pack2bf(g[2], g[3]);
                *reinterpret_cast<uint2*>(dlogits + (size_t)row * ldd + c) = o;
            }
        }
    }
}

// deterministic single-block sum: out[0] = scale * sum(x[0..n))
__global__ __launch_bounds__(1024) void reduce_sum_kernel(const float* __restrict__ x, long long n, float scale,
                                                          float* __restrict__ out) {
    __shared__ float red[16];
    float acc = 0.f;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) acc += x[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = acc * scale;
}

// ------------------------------------------------------------------------------------------------ optimiser
// sum of squares of a bf16 gradient range -> one fp32 partial per block (deterministic two-stage reduction)
__global__ __launch_bounds__(256) void sumsq_kernel(const bf16_t* __restrict__ g, long long n, float* __restrict__ partials) {
    __shared__ float red[16];
    float acc = 0.f;
    const long long nvec = n >> 3;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // four 16-byte loads in flight per thread (one was 5.4 TB/s: 16 waves x 1 KB per CU do not cover the memory latency)
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        const U4 q0 = *reinterpret_cast<const U4*>(g + i * 8), q1 = *reinterpret_cast<const U4*>(g + (i + stride) * 8);
        const U4 q2 = *reinterpret_cast<const U4*>(g + (i + 2 * stride) * 8), q3 = *reinterpret_cast<const U4*>(g + (i + 3 * stride) * 8);
        float f0[8], f1[8], f2[8], f3[8];
        unpack8(q0, f0); unpack8(q1, f1); unpack8(q2, f2); unpack8(q3, f3);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += f0[j] * f0[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += f1[j] * f1[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += f2[j] * f2[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += f3[j] * f3[j];
    }
    for (; i < nvec; i += stride) {
        float f[8];
        unpack8(*reinterpret_cast<const U4*>(g + i * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += f[j] * f[j];
    }
    if (blockIdx.x == 0)
        for (long long i = (nvec << 3) + threadIdx.x; i < n; i += blockDim.x) { const float f = bf2f(g[i]); acc += f * f; }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// norm = sqrt(sum partials); coef = min(1, max_norm / (norm + 1e-6))  (torch clip_grad_norm_, reference
// src/csm/training/trainer.py:271-274).  max_norm <= 0 disables clipping.  out[0] = norm, out[1] = coef.
__global__ __launch_bounds__(1024) void clip_coef_kernel(const float* __restrict__ partials, int n, float max_norm,
                                                         float* __restrict__ out) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partials[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(acc);
        out[0] = norm;
        out[1] = (max_norm > 0.f) ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
    }
}

// One element of torch.optim.AdamW (decoupled decay, bias-corrected, eps outside the sqrt ratio).  Floating-point contraction
// is off so that every kernel that inlines this rounds exactly the same way (csm_adamw_step and csm_adamw_step_split are
// bit-identical in master / m / v).
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float lr, float beta1, float beta2, float eps,
                                            float wd, float bc1, float bc2_sqrt) {
#pragma clang fp contract(off)
    p = p * (1.f - lr * wd);
    m = beta1 * m + (1.f - beta1) * g;
    v = beta2 * v + ((1.f - beta2) * g) * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - (lr / bc1) * (m / denom);
}

// A NEGATIVE clip coefficient on the device (norm_and_coef[1] < 0) means "this optimiser step does not happen": weights and
// moments stay as they are, only the zero_grad side of the pass runs.  It is how a step whose gradients are known on the
// device to be incomplete (data-parallel text-row exchange over capacity, training/dp.py) is dropped on every rank without
// the host having to look at the flag before it launches the update.
__device__ __forceinline__ void adamw_skipped(bf16_t* __restrict__ grad, long long nvec, int zero_grad) {
    if (!zero_grad) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x)
        *reinterpret_cast<U4*>(grad + i * 8) = (U4){0u, 0u, 0u, 0u};
}

// torch.optim.AdamW over one contiguous parameter range: fp32 master / m / v, bf16 gradient (times the device-side
// clip coefficient), writes the bf16 working copy.  28 B/param of HBM traffic.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ param, bf16_t* __restrict__ grad, long long n,
                                                    float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                    float bc2_sqrt, const float* __restrict__ coef_ptr, float gmul,
                                                    int zero_grad) {
    // plain (default cache policy) 16-byte accesses: measured 5.6 TB/s; non-temporal loads/stores measured 4.5-5.0 TB/s here
    const float coef = (coef_ptr ? coef_ptr[1] : 1.f) * gmul;
    const long long nvec = n >> 3;
    if (coef_ptr && coef_ptr[1] < 0.f) { adamw_skipped(grad, nvec, zero_grad); return; }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        float g[8];
        unpack8(*reinterpret_cast<const U4*>(grad + i * 8), g);
        float4 p0 = *reinterpret_cast<const float4*>(master + i * 8), p1 = *reinterpret_cast<const float4*>(master + i * 8 + 4);
        float4 m0 = *reinterpret_cast<const float4*>(m + i * 8), m1 = *reinterpret_cast<const float4*>(m + i * 8 + 4);
        float4 v0 = *reinterpret_cast<const float4*>(v + i * 8), v1 = *reinterpret_cast<const float4*>(v + i * 8 + 4);
        float p[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            adam_update(p[j], mm[j], vv[j], g[j] * coef, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
        }
        *reinterpret_cast<float4*>(master + i * 8) = make_float4(p[0], p[1], p[2], p[3]);
        *reinterpret_cast<float4*>(master + i * 8 + 4) = make_float4(p[4], p[5], p[6], p[7]);
        *reinterpret_cast<float4*>(m + i * 8) = make_float4(mm[0], mm[1], mm[2], mm[3]);
        *reinterpret_cast<float4*>(m + i * 8 + 4) = make_float4(mm[4], mm[5], mm[6], mm[7]);
        *reinterpret_cast<float4*>(v + i * 8) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        *reinterpret_cast<float4*>(v + i * 8 + 4) = make_float4(vv[4], vv[5], vv[6], vv[7]);
        *reinterpret_cast<U4*>(param + i * 8) = pack8(p);
        if (zero_grad) *reinterpret_cast<U4*>(grad + i * 8) = (U4){0u, 0u, 0u, 0u};
    }
}

// The same update with the fp32 master weight held as TWO 16-bit halves: the bf16 working copy IS the upper half (rounded
// half-up: hi = (bits + 0x8000) >> 16) and `lo` the lower 16 bits, so master = ((hi - (lo >> 15)) << 16) | lo exactly - no
// separate 4-byte master to read and write: 26 instead of 28 B/param.  The arithmetic is the fp32 sequence above, bit for
// bit; the only visible difference is that an exact tie rounds the bf16 working weight away from zero instead of to even.
__global__ __launch_bounds__(256) void adamw_split_kernel(uint16_t* __restrict__ lo, float* __restrict__ m, float* __restrict__ v,
                                                          bf16_t* __restrict__ param, bf16_t* __restrict__ grad, long long n,
                                                          float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                          float bc2_sqrt, const float* __restrict__ coef_ptr, float gmul,
                                                          int zero_grad) {
    const float coef = (coef_ptr ? coef_ptr[1] : 1.f) * gmul;
    const long long nvec = n >> 3;
    if (coef_ptr && coef_ptr[1] < 0.f) { adamw_skipped(grad, nvec, zero_grad); return; }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        float g[8];
        unpack8(*reinterpret_cast<const U4*>(grad + i * 8), g);
        const U4 hw = *reinterpret_cast<const U4*>(param + i * 8), lw = *reinterpret_cast<const U4*>(lo + i * 8);
        float4 m0 = *reinterpret_cast<const float4*>(m + i * 8), m1 = *reinterpret_cast<const float4*>(m + i * 8 + 4);
        float4 v0 = *reinterpret_cast<const float4*>(v + i * 8), v1 = *reinterpret_cast<const float4*>(v + i * 8 + 4);
        const uint32_t hww[4] = {hw.x, hw.y, hw.z, hw.w}, lww[4] = {lw.x, lw.y, lw.z, lw.w};
        float p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t h16 = (hww[j >> 1] >> ((j & 1) * 16)) & 0xffffu, l16 = (lww[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
            p[j] = __uint_as_float(((h16 - (l16 >> 15)) << 16) | l16);
        }
        float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        uint32_t ho[4] = {0u, 0u, 0u, 0u}, lout[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            adam_update(p[j], mm[j], vv[j], g[j] * coef, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
            const uint32_t bits = __float_as_uint(p[j]);
            ho[j >> 1] |= (((bits + 0x8000u) >> 16) & 0xffffu) << ((j & 1) * 16);
            lout[j >> 1] |= (bits & 0xffffu) << ((j & 1) * 16);
        }
        *reinterpret_cast<float4*>(m + i * 8) = make_float4(mm[0], mm[1], mm[2], mm[3]);
        *reinterpret_cast<float4*>(m + i * 8 + 4) = make_float4(mm[4], mm[5], mm[6], mm[7]);
        *reinterpret_cast<float4*>(v + i * 8) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        *reinterpret_cast<float4*>(v + i * 8 + 4) = make_float4(vv[4], vv[5], vv[6], vv[7]);
        *reinterpret_cast<U4*>(param + i * 8) = (U4){ho[0], ho[1], ho[2], ho[3]};
        *reinterpret_cast<U4*>(lo + i * 8) = (U4){lout[0], lout[1], lout[2], lout[3]};
        if (zero_grad) *reinterpret_cast<U4*>(grad + i * 8) = (U4){0u, 0u, 0u, 0u};
    }
}

inline int grid_for(long long work_items, int block, int cap = 256 * 8) {
    long long b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

}  // namespace

// ================================================================================================ C ABI
extern "C" int csm_rmsnorm_fwd(const void* x, const void* scale, void* y, float* rstd, int M, int D, float eps,
                               hipStream_t stream) {
    CSM_REQUIRE(x && scale && y, "csm_rmsnorm_fwd: null pointer");
    CSM_REQUIRE(M > 0 && D > 0 && (D & 7) == 0 && D <= 4096, "csm_rmsnorm_fwd: D=%d must be a multiple of 8 and <= 4096", D);
    const int grid = grid_for(M, 4, 4096);
    const int nc = (D + 511) / 512;
#define L(NC) hipLaunchKernelGGL((rmsnorm_fwd_kernel<NC>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)scale, (bf16_t*)y, rstd, M, D, eps)
    if (nc <= 1) L(1); else if (nc <= 2) L(2); else if (nc <= 4) L(4); else L(8);
#undef L
    CSM_CHECK_LAUNCH("csm_rmsnorm_fwd");
    return 0;
}

#define CSM_RMSNORM_BWD_BLOCKS 256
extern "C" int csm_rmsnorm_bwd_blocks(void) { return CSM_RMSNORM_BWD_BLOCKS; }

// dscale_partials: [csm_rmsnorm_bwd_blocks()][D] fp32 workspace (every row is written), or NULL when the scale is frozen
extern "C" int csm_rmsnorm_bwd(const void* x, const void* scale, const float* rstd, const void* dy, const void* dres,
                               void* dx, float* dscale_partials, int M, int D, hipStream_t stream) {
    CSM_REQUIRE(x && scale && rstd && dy && dx, "csm_rmsnorm_bwd: null pointer");
    CSM_REQUIRE(M > 0 && D > 0 && (D & 7) == 0 && D <= 4096, "csm_rmsnorm_bwd: D=%d must be a multiple of 8 and <= 4096", D);
    const int grid = CSM_RMSNORM_BWD_BLOCKS;
    const int nc = (D + 511) / 512;
    const size_t lds = dscale_partials ? (size_t)8 * D * sizeof(float) : 0;      // 8 waves x D floats (64 KiB at D = 2048)
    if (lds > 65536) {   // only the D > 2048 instance needs more than the default dynamic-LDS limit
        static bool done = false;
        if (!done) (void)hipFuncSetAttribute((const void*)rmsnorm_bwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(8 * 4096 * sizeof(float)));
        done = true;
    }
#define L(NC) hipLaunchKernelGGL((rmsnorm_bwd_kernel<NC>), dim3(grid), dim3(512), lds, stream, (const bf16_t*)x, (const bf16_t*)scale, rstd, (const bf16_t*)dy, (const bf16_t*)dres, (bf16_t*)dx, dscale_partials, M, D)
    if (nc <= 1) L(1); else if (nc <= 2) L(2); else if (nc <= 4) L(4); else L(8);
#undef L
    CSM_CHECK_LAUNCH("csm_rmsnorm_bwd");
    return 0;
}

extern "C" int csm_colsum_bf16(const float* partials, int rows, int D, void* dst, int accumulate, hipStream_t stream) {
    CSM_REQUIRE(partials && dst && rows > 0 && D > 0, "csm_colsum_bf16: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3((D + 63) / 64), dim3(rows >= 64 ? 1024 : 256), 0, stream, partials, rows, D, (bf16_t*)dst,
                       accumulate);
    CSM_CHECK_LAUNCH("csm_colsum_bf16");
    return 0;
}

extern "C" int csm_colsum_bf16_multi(int n, const float* const* partials, void* const* dst, int rows, int D, int accumulate,
                                     hipStream_t stream) {
    CSM_REQUIRE(n >= 1 && n <= 8 && partials && dst && rows > 0 && D > 0, "csm_colsum_bf16_multi: 1..8 pairs, rows, D > 0");
    ColsumMulti p;
    for (int i = 0; i < 8; ++i) {
        CSM_REQUIRE(i >= n || (partials[i] && dst[i]), "csm_colsum_bf16_multi: null pointer in pair %d", i);
        p.partials[i] = partials[i < n ? i : 0]; p.dst[i] = (bf16_t*)dst[i < n ? i : 0];
    }
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((D + 63) / 64, n), dim3(rows >= 64 ? 1024 : 256), 0, stream, p, rows, D, accumulate);
    CSM_CHECK_LAUNCH("csm_colsum_bf16_multi");
    return 0;
}

extern "C" int csm_dropout_bf16(const void* in, int ld_in, void* out, int ld_out, long long M, int D, float p,
                                unsigned long long seed, int accumulate, hipStream_t stream) {
    CSM_REQUIRE(in && out && M > 0 && D > 0 && D % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && ld_in >= D && ld_out >= D,
                "csm_dropout_bf16: bad arguments (D and strides must be multiples of 8)");
    CSM_REQUIRE(p >= 0.f && p < 1.f, "csm_dropout_bf16: p must be in [0, 1)");
    const uint32_t thresh = (uint32_t)(p * 65536.0f + 0.5f);
    const long long total = M * (D / 8);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, M, D,
                       thresh, 1.0f / (1.0f - thresh / 65536.0f), (uint64_t)seed, accumulate);
    CSM_CHECK_LAUNCH("csm_dropout_bf16");
    return 0;
}

extern "C" int csm_bias_add_bf16(void* y, int ld, const void* bias, long long M, int D, hipStream_t stream) {
    CSM_REQUIRE(y && bias && M > 0 && D > 0 && D % 8 == 0 && ld % 8 == 0 && ld >= D, "csm_bias_add_bf16: bad arguments");
    const long long total = M * (D / 8);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(bias_add_kernel, dim3(blocks), dim3(256), 0, stream, (bf16_t*)y, ld, (const bf16_t*)bias, M, D);
    CSM_CHECK_LAUNCH("csm_bias_add_bf16");
    return 0;
}

extern "C" int csm_colsum_rows_bf16(const void* x, int ld, long long M, int D, float* partials, int slices, hipStream_t stream) {
    CSM_REQUIRE(x && partials && M > 0 && D > 0 && ld >= D && slices > 0 && slices <= 65535, "csm_colsum_rows_bf16: bad arguments");
    hipLaunchKernelGGL(colsum_rows_kernel, dim3((D + 63) / 64, slices), dim3(256), 0, stream, (const bf16_t*)x, ld, M, D, partials);
    CSM_CHECK_LAUNCH("csm_colsum_rows_bf16");
    return 0;
}

extern "C" int csm_rope(void* qkv, const float* table, const int* pos, long long M, int S, int n_heads_qk, int head_dim,
                        int ld, int inverse, hipStream_t stream) {
    CSM_REQUIRE(qkv && table, "csm_rope: null pointer");
    CSM_REQUIRE(M > 0 && S > 0 && n_heads_qk > 0 && (head_dim & 7) == 0 && (ld & 7) == 0, "csm_rope: bad shape");
    const long long total = M * n_heads_qk * (head_dim >> 3);
    hipLaunchKernelGGL(rope_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, stream, (bf16_t*)qkv, table, pos, M, S,
                       n_heads_qk, head_dim, ld, inverse);
    CSM_CHECK_LAUNCH("csm_rope");
    return 0;
}

extern "C" int csm_swiglu_fwd(const void* gu, void* out, long long M, int F, hipStream_t stream) {
    CSM_REQUIRE(gu && out && M > 0 && F > 0 && (F & 7) == 0, "csm_swiglu_fwd: bad arguments");
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(M * (F >> 2), 256, 8192)), dim3(256), 0, stream, (const bf16_t*)gu,
                       (bf16_t*)out, M, F);
    CSM_CHECK_LAUNCH("csm_swiglu_fwd");
    return 0;
}

extern "C" int csm_swiglu_bwd(const void* gu, const void* dout, void* dgu, long long M, int F, hipStream_t stream) {
    CSM_REQUIRE(gu && dout && dgu && M > 0 && F > 0 && (F & 7) == 0, "csm_swiglu_bwd: bad arguments");
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(M * (F >> 2), 256, 8192)), dim3(256), 0, stream, (const bf16_t*)gu,
                       (const bf16_t*)dout, (bf16_t*)dgu, M, F);
    CSM_CHECK_LAUNCH("csm_swiglu_bwd");
    return 0;
}

extern "C" int csm_embed_fwd(const long long* tokens, const uint8_t* mask, const void* text_emb, const void* audio_emb,
                             void* out, long long M, int K, int D, int audio_vocab, hipStream_t stream) {
    CSM_REQUIRE(tokens && mask && text_emb && audio_emb && out, "csm_embed_fwd: null pointer");
    CSM_REQUIRE(M > 0 && M < (1ll << 31) && K > 0 && (D & 7) == 0, "csm_embed_fwd: bad shape");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)M), dim3(256), 0, stream, tokens, mask, (const bf16_t*)text_emb,
                       (const bf16_t*)audio_emb, (bf16_t*)out, K, D, audio_vocab);
    CSM_CHECK_LAUNCH("csm_embed_fwd");
    return 0;
}

extern "C" int csm_embed_bwd_sorted(const long long* sorted_rows, const long long* src_index, long long n_occ, const void* dh,
                                    const void* dseq, long long M, void* g_text, void* g_audio, long long text_rows,
                                    long long n_rows, int D, hipStream_t stream) {
    CSM_REQUIRE(sorted_rows && src_index && dh && g_text && g_audio && n_occ > 0, "csm_embed_bwd_sorted: bad arguments");
    CSM_REQUIRE((D & 7) == 0 && D <= 4096, "csm_embed_bwd_sorted: D=%d must be a multiple of 8 and <= 4096", D);
    const long long blocks = (n_occ + 3) / 4;
    CSM_REQUIRE(blocks < (1ll << 31), "csm_embed_bwd_sorted: too many occurrences");
    const int nc = (D + 511) / 512;
#define L(NC) hipLaunchKernelGGL((embed_bwd_sorted_kernel<NC>), dim3((unsigned)blocks), dim3(256), 0, stream, sorted_rows, src_index, n_occ, (const bf16_t*)dh, (const bf16_t*)dseq, M, (bf16_t*)g_text, (bf16_t*)g_audio, text_rows, n_rows, D)
    if (nc <= 1) L(1); else if (nc <= 2) L(2); else if (nc <= 4) L(4); else L(8);
#undef L
    CSM_CHECK_LAUNCH("csm_embed_bwd_sorted");
    return 0;
}

extern "C" int csm_rows_add_bf16(void* dst, const int* rows, const void* src, long long N, int src_stride_rows, int D,
                                 hipStream_t stream) {
    CSM_REQUIRE(dst && rows && src && N > 0 && (D & 7) == 0 && src_stride_rows > 0, "csm_rows_add_bf16: bad arguments");
    hipLaunchKernelGGL(rows_add_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, (bf16_t*)dst, rows, (const bf16_t*)src,
                       N, src_stride_rows, D);
    CSM_CHECK_LAUNCH("csm_rows_add_bf16");
    return 0;
}

extern "C" int csm_rows_take_bf16(void* table, const int* rows, void* out, long long N, int D, hipStream_t stream) {
    CSM_REQUIRE(table && rows && out && N > 0 && (D & 7) == 0, "csm_rows_take_bf16: bad arguments");
    hipLaunchKernelGGL(rows_take_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, (bf16_t*)table, rows, (bf16_t*)out, N, D);
    CSM_CHECK_LAUNCH("csm_rows_take_bf16");
    return 0;
}

extern "C" int csm_decoder_input_fwd(const void* hidden, const int* rows, const long long* codes, const void* audio_emb,
                                     void* out, long long N, int K, int D, int audio_vocab, hipStream_t stream) {
    CSM_REQUIRE(hidden && rows && codes && audio_emb && out && N > 0 && N * K < (1ll << 31) && (D & 7) == 0,
                "csm_decoder_input_fwd: bad arguments");
    hipLaunchKernelGGL(decoder_input_kernel, dim3((unsigned)(N * K)), dim3(256), 0, stream, (const bf16_t*)hidden, rows, codes,
                       (const bf16_t*)audio_emb, (bf16_t*)out, K, D, audio_vocab);
    CSM_CHECK_LAUNCH("csm_decoder_input_fwd");
    return 0;
}

extern "C" int csm_ce_fwd_bwd(const float* logits, const long long* targets, float* loss_rows, void* dlogits, long long R,
                              int V, int ldl, int ldd, float grad_scale, hipStream_t stream) {
    CSM_REQUIRE(logits && targets && loss_rows && R > 0 && R < (1ll << 31) && V > 0 && ldl >= V, "csm_ce_fwd_bwd: bad arguments");
    CSM_REQUIRE(!dlogits || ldd >= V, "csm_ce_fwd_bwd: ldd < V");
    const int width = ldl > ldd && dlogits ? ldl : (dlogits ? ldd : ldl);          // columns a lane pattern must cover
    const bool vec = (ldl & 3) == 0 && ((uintptr_t)logits & 15) == 0 && (!dlogits || ((ldd & 3) == 0 && ((uintptr_t)dlogits & 7) == 0 && ldd <= ldl + 0));
    if (vec && width <= 256 * 12) {
        const dim3 grid((unsigned)((R + 3) / 4)), block(256);
#define L(NC) hipLaunchKernelGGL((ce_wave_kernel<NC>), grid, block, 0, stream, logits, targets, loss_rows, (bf16_t*)dlogits, R, V, ldl, ldd, grad_scale)
        if (width <= 256 * 4) L(4); else if (width <= 256 * 9) L(9); else L(12);
#undef L
    } else {
        hipLaunchKernelGGL(ce_kernel, dim3((unsigned)R), dim3(256), 0, stream, logits, targets, loss_rows, (bf16_t*)dlogits, V, ldl,
                           ldd, grad_scale);
    }
    CSM_CHECK_LAUNCH("csm_ce_fwd_bwd");
    return 0;
}

extern "C" int csm_reduce_sum_f32(const float* x, long long n, float scale, float* out, hipStream_t stream) {
    CSM_REQUIRE(x && out && n > 0, "csm_reduce_sum_f32: bad arguments");
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, stream, x, n, scale, out);
    CSM_CHECK_LAUNCH("csm_reduce_sum_f32");
    return 0;
}

#define CSM_SUMSQ_BLOCKS 8192
extern "C" int csm_sumsq_blocks(void) { return CSM_SUMSQ_BLOCKS; }

// partials must hold csm_sumsq_blocks() floats per call; several ranges can be reduced into consecutive slots.
extern "C" int csm_sumsq_bf16(const void* g, long long n, float* partials, hipStream_t stream) {
    CSM_REQUIRE(g && partials && n > 0 && ((uintptr_t)g & 15) == 0, "csm_sumsq_bf16: bad arguments");
    hipLaunchKernelGGL(sumsq_kernel, dim3(CSM_SUMSQ_BLOCKS), dim3(256), 0, stream, (const bf16_t*)g, n, partials);
    CSM_CHECK_LAUNCH("csm_sumsq_bf16");
    return 0;
}

extern "C" int csm_clip_coef(const float* partials, int n_partials, float max_norm, float* norm_and_coef,
                             hipStream_t stream) {
    CSM_REQUIRE(partials && norm_and_coef && n_partials > 0, "csm_clip_coef: bad arguments");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1024), 0, stream, partials, n_partials, max_norm, norm_and_coef);
    CSM_CHECK_LAUNCH("csm_clip_coef");
    return 0;
}

static int g_adamw_blocks = 1 << 20;     // one 256-thread block per 2048 elements, no grid stride: 6.80 vs 6.96 ms per 1.55 G parameters (tools/probes/adamw_blocks.py)
extern "C" int csm_set_adamw_blocks(int b) { g_adamw_blocks = b; return 0; }

// zero_grad != 0 clears the gradient range in the same pass (the line is already being read)
extern "C" int csm_adamw_step(float* master, float* m, float* v, void* param, void* grad, long long n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step,
                              const float* norm_and_coef, float grad_mul, int zero_grad, hipStream_t stream) {
    CSM_REQUIRE(master && m && v && param && grad && n > 0 && step > 0, "csm_adamw_step: bad arguments");
    CSM_REQUIRE((n & 7) == 0, "csm_adamw_step: n must be a multiple of 8 (pad the arena)");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n >> 3, 256, g_adamw_blocks)), dim3(256), 0, stream, master, m, v, (bf16_t*)param,
                       (bf16_t*)grad, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, norm_and_coef, grad_mul, zero_grad);
    CSM_CHECK_LAUNCH("csm_adamw_step");
    return 0;
}

// AdamW with the master weight split into the bf16 working copy (upper half, rounded half-up) and `master_lo` (lower 16
// bits): same update as csm_adamw_step, 26 B/param.  Replaces optim.AdamW of reference src/csm/training/trainer.py:166-173.
extern "C" int csm_adamw_step_split(void* master_lo, float* m, float* v, void* param, void* grad, long long n, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int step, const float* norm_and_coef, float grad_mul,
                                    int zero_grad, hipStream_t stream) {
    CSM_REQUIRE(master_lo && m && v && param && grad && n > 0 && step >= 1, "csm_adamw_step_split: bad arguments");
    CSM_REQUIRE((n & 7) == 0, "csm_adamw_step_split: n must be a multiple of 8 (pad the arena)");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_split_kernel, dim3(grid_for(n >> 3, 256, g_adamw_blocks)), dim3(256), 0, stream, (uint16_t*)master_lo, m, v,
                       (bf16_t*)param, (bf16_t*)grad, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, norm_and_coef, grad_mul, zero_grad);
    CSM_CHECK_LAUNCH("csm_adamw_step_split");
    return 0;
}
