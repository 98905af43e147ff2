// Mimi codec kernels for gfx950 (fp32): SEANet causal convolutions / transposed convolutions with fused ELU, bias and
// residual, LayerNorm, a tiled fp32 linear layer with fused GELU / layer-scale / residual epilogues, rotate-half RoPE and
// sliding-window causal attention for the 8-layer codec transformers.
//
// Replaces the moshi 0.2.2 `MimiModel.encode / decode` calls behind reference src/csm/generator.py:67-70,117,209
// (third-party, not vendored; restated from the published architecture and cross-checked against the HF port).
// fp32 on purpose: the encoder ends in a nearest-codeword search whose integer output must match a fp32 CPU run, and the
// whole codec is ~40 GFLOP per 10 s of audio - noise next to the language model - so no MFMA / bf16 here.
#include "common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }

// y[co][t] = bias[co] + sum_{ci in group} sum_j w[co][ci][j] * act(xpad[ci][t*stride + j*dil - pad_left]) (+ res[co][t])
// xpad: zero (pad_mode 0) or edge-replicated (pad_mode 1) outside [0, T_in).  One wave-row of threads shares `co`, so the
// weights are wave-uniform (scalar loads) and the input reads are coalesced along time.
__global__ __launch_bounds__(256) void conv1d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ res,
                                                     float* __restrict__ y, int C_in, int C_out, int T_in, int T_out, int k,
                                                     int stride, int dil, int pad_left, int pad_mode, int groups, int elu_in) {
    const int co = blockIdx.y;
    const int cin_g = C_in / groups, cout_g = C_out / groups;
    const int grp = co / cout_g;
    const float* wrow = w + (size_t)co * cin_g * k;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T_out; t += gridDim.x * blockDim.x) {
        float acc = bias ? bias[co] : 0.f;
        const int base = t * stride - pad_left;
        for (int ci = 0; ci < cin_g; ++ci) {
            const float* xr = x + (size_t)(grp * cin_g + ci) * T_in;
            for (int j = 0; j < k; ++j) {
                int p = base + j * dil;
                float v;
                if (p >= 0 && p < T_in) v = xr[p];
                else if (pad_mode == 1) v = xr[p < 0 ? 0 : T_in - 1];
                else v = 0.f;
                if (elu_in) v = elu1(v);   // ELU(0) = 0, so applying it to zero padding changes nothing
                acc += wrow[ci * k + j] * v;
            }
        }
        if (res) acc += res[(size_t)co * T_out + t];
        y[(size_t)co * T_out + t] = acc;
    }
}

// ConvTranspose1d (torch weight layout [C_in][C_out/groups][k]) cropped to [crop_left, crop_left + T_out):
// y[co][t] = bias[co] + sum_ci sum_{j : (t + crop_left - j) % stride == 0} act(x[ci][(t + crop_left - j)/stride]) w[ci][co_g][j]
__global__ __launch_bounds__(256) void conv_transpose1d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y, int C_in,
                                                               int C_out, int T_in, int T_out, int k, int stride, int crop_left,
                                                               int groups, int elu_in) {
    const int co = blockIdx.y;
    const int cin_g = C_in / groups, cout_g = C_out / groups;
    const int grp = co / cout_g, co_g = co % cout_g;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T_out; t += gridDim.x * blockDim.x) {
        float acc = bias ? bias[co] : 0.f;
        const int tf = t + crop_left;
        for (int j = tf % stride; j < k; j += stride) {
            const int ti = (tf - j) / stride;
            if (ti < 0 || ti >= T_in) continue;
            for (int ci = 0; ci < cin_g; ++ci) {
                float v = x[(size_t)(grp * cin_g + ci) * T_in + ti];
                if (elu_in) v = elu1(v);
                acc += v * w[((size_t)(grp * cin_g + ci) * cout_g + co_g) * k + j];
            }
        }
        y[(size_t)co * T_out + t] = acc;
    }
}

// y[t][:] = (x[t] - mean) * rsqrt(var + eps) * w + b      (one wave per row)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, int T, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= T) return;
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mean = wave_sum(s) / D;
    float v = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = xr[c] - mean; v += d * d; }
    const float r = rsqrtf(wave_sum(v) / D + eps);
    for (int c = lane; c < D; c += 64) y[(size_t)row * D + c] = (xr[c] - mean) * r * w[c] + b[c];
}

// y[T][N] = epilogue(x[T][K] W[N][K]^T): 64x64 tile per 256-thread block, 4x4 outputs per thread, K in steps of 16
// epilogue: act 1 = exact GELU; scale != NULL -> y = res + scale[n] * acc (layer scale + residual); else y = acc (+ res)
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ scale, const float* __restrict__ res,
                                                         float* __restrict__ y, int T, int N, int K, int ldx, int act) {
    __shared__ float xs[16][64 + 1], ws[16][64 + 1];
    const int t0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // 16 x 16 threads, each 4 (t) x 4 (n)
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int r = i >> 4, c = i & 15;
            xs[c][r] = (t0 + r < T && k0 + c < K) ? x[(size_t)(t0 + r) * ldx + k0 + c] : 0.f;
            ws[c][r] = (n0 + r < N && k0 + c < K) ? W[(size_t)(n0 + r) * K + k0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = xs[c][ty * 4 + i]; b[i] = ws[c][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty * 4 + i;
        if (t >= T) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float v = acc[i][j];
            if (act == 1) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
            if (scale) v = res[(size_t)t * N + n] + scale[n] * v;
            else if (res) v += res[(size_t)t * N + n];
            y[(size_t)t * N + n] = v;
        }
    }
}

// rotate-half RoPE (HF / moshi convention) in place on the q and k parts of qkv [T][3*H*hd]; theta_i = base^(-2i/hd)
__global__ __launch_bounds__(256) void rope_half_kernel(float* __restrict__ qkv, int T, int H, int hd, float base, int pos0) {
    const int half = hd >> 1;
    const long long total = (long long)T * 2 * H * half;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(idx % half);
        const int hh = (int)((idx / half) % (2 * H));          // q heads then k heads
        const int t = (int)(idx / ((long long)half * 2 * H));
        const float inv = powf(base, -2.f * i / hd);
        float sn, cs;
        sincosf((float)(pos0 + t) * inv, &sn, &cs);
        float* p = qkv + (size_t)t * 3 * H * hd + (size_t)hh * hd;
        const float a = p[i], b = p[i + half];
        float ra = a, rb = b;
        rope_rot(ra, rb, cs, sn);
        p[i] = ra;
        p[i + half] = rb;
    }
}

// causal sliding-window attention, one block per (query, head); keys in (q - window, q]
template <int HD>
__global__ __launch_bounds__(64) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T, int H, int window) {
    extern __shared__ float sc[];   // [window]
    const int q = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int ld = 3 * H * HD;
    const float* qp = qkv + (size_t)q * ld + h * HD;
    const int k_lo = q - window + 1 > 0 ? q - window + 1 : 0;
    const int n = q - k_lo + 1;
    const float scale = rsqrtf((float)HD);
    float mx = -INFINITY;
    for (int s = lane; s < n; s += 64) {
        const float* kp = qkv + (size_t)(k_lo + s) * ld + (H + h) * HD;
        float d = 0.f;
#pragma unroll 8
        for (int c = 0; c < HD; ++c) d += qp[c] * kp[c];
        d *= scale;
        sc[s] = d;
        mx = fmaxf(mx, d);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int s = lane; s < n; s += 64) { const float p = expf(sc[s] - mx); sc[s] = p; sum += p; }
    sum = wave_sum(sum);
    __syncthreads();
    for (int c = lane; c < HD; c += 64) {
        float acc = 0.f;
        for (int s = 0; s < n; ++s) acc += sc[s] * qkv[(size_t)(k_lo + s) * ld + (2 * H + h) * HD + c];
        out[(size_t)q * H * HD + h * HD + c] = acc / sum;
    }
}

// out[c][r] = in[r][c]
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cn) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < Cn) tile[i][tx] = in[(size_t)(r0 + i) * Cn + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < Cn && r0 + tx < R) out[(size_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

}  // namespace

extern "C" int csm_conv1d_f32(const float* x, const float* w, const float* bias, const float* residual, float* y, int C_in,
                              int C_out, int T_in, int T_out, int k, int stride, int dilation, int pad_left, int pad_mode,
                              int groups, int elu_in, hipStream_t stream) {
    CSM_REQUIRE(x && w && y && C_in > 0 && C_out > 0 && T_in > 0 && T_out > 0 && k > 0 && stride > 0 && dilation > 0 && groups > 0 &&
                    C_in % groups == 0 && C_out % groups == 0 && C_out <= 65535, "csm_conv1d_f32: bad arguments");
    int bx = (T_out + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(conv1d_kernel, dim3(bx, C_out), dim3(256), 0, stream, x, w, bias, residual, y, C_in, C_out, T_in, T_out, k,
                       stride, dilation, pad_left, pad_mode, groups, elu_in);
    CSM_CHECK_LAUNCH("csm_conv1d_f32");
    return 0;
}

extern "C" int csm_conv_transpose1d_f32(const float* x, const float* w, const float* bias, float* y, int C_in, int C_out, int T_in,
                                        int T_out, int k, int stride, int crop_left, int groups, int elu_in, hipStream_t stream) {
    CSM_REQUIRE(x && w && y && C_in > 0 && C_out > 0 && T_in > 0 && T_out > 0 && k > 0 && stride > 0 && groups > 0 &&
                    C_in % groups == 0 && C_out % groups == 0 && C_out <= 65535, "csm_conv_transpose1d_f32: bad arguments");
    int bx = (T_out + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(conv_transpose1d_kernel, dim3(bx, C_out), dim3(256), 0, stream, x, w, bias, y, C_in, C_out, T_in, T_out, k,
                       stride, crop_left, groups, elu_in);
    CSM_CHECK_LAUNCH("csm_conv_transpose1d_f32");
    return 0;
}

extern "C" int csm_layernorm_f32(const float* x, const float* w, const float* b, float* y, int T, int D, float eps, hipStream_t stream) {
    CSM_REQUIRE(x && w && b && y && T > 0 && D > 0, "csm_layernorm_f32: bad arguments");
    hipLaunchKernelGGL(layernorm_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, x, w, b, y, T, D, eps);
    CSM_CHECK_LAUNCH("csm_layernorm_f32");
    return 0;
}

extern "C" int csm_linear_f32(const float* x, const float* W, const float* scale, const float* residual, float* y, int T, int N,
                              int K, int ldx, int act, hipStream_t stream) {
    CSM_REQUIRE(x && W && y && T > 0 && N > 0 && K > 0 && ldx >= K && (!scale || residual), "csm_linear_f32: bad arguments");
    hipLaunchKernelGGL(linear_f32_kernel, dim3((N + 63) / 64, (T + 63) / 64), dim3(256), 0, stream, x, W, scale, residual, y, T, N,
                       K, ldx, act);
    CSM_CHECK_LAUNCH("csm_linear_f32");
    return 0;
}

extern "C" int csm_rope_half_f32(float* qkv, int T, int H, int head_dim, float base, int pos0, hipStream_t stream) {
    CSM_REQUIRE(qkv && T > 0 && H > 0 && head_dim > 0 && (head_dim & 1) == 0, "csm_rope_half_f32: bad arguments");
    const long long total = (long long)T * 2 * H * (head_dim / 2);
    long long b = (total + 255) / 256;
    hipLaunchKernelGGL(rope_half_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(256), 0, stream, qkv, T, H, head_dim, base, pos0);
    CSM_CHECK_LAUNCH("csm_rope_half_f32");
    return 0;
}

extern "C" int csm_attn_window_f32(const float* qkv, float* out, int T, int H, int head_dim, int window, hipStream_t stream) {
    CSM_REQUIRE(qkv && out && T > 0 && H > 0 && window > 0 && window <= 8192, "csm_attn_window_f32: bad arguments");
    CSM_REQUIRE(head_dim == 64, "csm_attn_window_f32: head_dim %d unsupported (64)", head_dim);
    hipLaunchKernelGGL((attn_f32_kernel<64>), dim3(T, H), dim3(64), (size_t)window * sizeof(float), stream, qkv, out, T, H, window);
    CSM_CHECK_LAUNCH("csm_attn_window_f32");
    return 0;
}

extern "C" int csm_transpose_f32(const float* in, float* out, int R, int C, hipStream_t stream) {
    CSM_REQUIRE(in && out && R > 0 && C > 0, "csm_transpose_f32: bad arguments");
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, stream, in, out, R, C);
    CSM_CHECK_LAUNCH("csm_transpose_f32");
    return 0;
}
