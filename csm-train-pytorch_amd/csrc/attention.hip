// Flash-style causal GQA attention for gfx950, forward + backward, head_dim 64 (backbone) / 128 (decoder).
//
// Replaces torchtune MultiHeadAttention -> F.scaled_dot_product_attention with the boolean mask built by
// reference src/csm/models/model.py:59-76 and src/csm/training/utils.py:90-91 (positions arange(S) => plain
// causal); GQA: q-head j uses kv-head j / (H/KV).  Nothing of size S x S is ever stored.
//
// Layout: qkv is the fused projection output [B*S][(H + 2*KV) * HD] (q heads | k heads | v heads),
// RoPE already applied in place.  out is [B*S][H*HD]; lse [B][H][S] fp32 (natural log).
//
// All three kernels use mfma_f32_16x16x32_bf16 and keep the softmax row on ONE lane:
//   forward / dQ :  S^T = K Q^T  (key on the register index, query on the lane)  -> the running max, sum,
//                   lse and delta are per-lane scalars; P^T (resp. dS^T) accumulators feed the second MFMA
//                   as its B operand with no data movement; V^T (resp. K^T) A-fragments come from
//                   ds_read_b64_tr_b16 on a [key][d] LDS image.
//   dK/dV        :  S = Q K^T    (query on the register index, key on the lane)  -> dV^T and dK^T accumulate in
//                   registers over every q-block and every q-head of the kv group; no atomics, deterministic.
// LDS images: "row" image (16-B chunk c of row r at chunk c ^ (r & (HD/8-1))) for ds_read_b128 fragments,
// "tr" image (32-B slot s of row r at s ^ swz(r)) for the transposed reads; both conflict-free.
#include "common.h"
#include <math.h>

namespace {

template <int HD>
struct Img {
    static constexpr int ROWB = HD * 2;
    static constexpr int CPR = HD / 8;          // 16-B chunks per row
    static constexpr int BYTES = 64 * ROWB;     // one 64-row image
    __device__ static __forceinline__ int row_off(int r, int c) { return r * ROWB + ((c ^ (r & (CPR - 1))) << 4); }
    __device__ static __forceinline__ int tr_off(int r, int slot) {
        const int sw = (HD == 64) ? ((r >> 1) & 3) : (r & 7);
        return r * ROWB + ((slot ^ sw) << 5);
    }
};

// global -> registers: 64 rows x HD starting at row r0 of a [S][ld] panel (rows clamped to S-1)
template <int HD>
__device__ __forceinline__ void tile_load(const bf16_t* __restrict__ P, int ld, int S, int r0, U4 (&reg)[HD / 32]) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int i = 0; i < HD / 32; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx / CPR, c = idx % CPR;
        int gr = r0 + r;
        gr = gr < S ? gr : S - 1;
        reg[i] = *reinterpret_cast<const U4*>(P + (size_t)gr * ld + c * 8);
    }
}

template <int HD, bool ROW, bool TR>
__device__ __forceinline__ void tile_store(char* row_img, char* tr_img, const U4 (&reg)[HD / 32]) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int i = 0; i < HD / 32; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx / CPR, c = idx % CPR;
        if (ROW) *reinterpret_cast<U4*>(row_img + Img<HD>::row_off(r, c)) = reg[i];
        if (TR) *reinterpret_cast<U4*>(tr_img + Img<HD>::tr_off(r, c >> 1) + ((c & 1) << 4)) = reg[i];
    }
}

// A/B fragment (row = r0 + lane&15, k = 32*ks + 8*(lane>>4) + j) from a row image
template <int HD>
__device__ __forceinline__ bf16x8 frag_row(const char* img, int r0, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + Img<HD>::row_off(r0 + (lane & 15), ks * 4 + (lane >> 4)));
}

// A fragment of the TRANSPOSED tile: row index = column (16*dt + lane&15) of the image,
// k index 8g+j  <->  image row  32*s2 + (j<4 ? 4g+j : 16+4g+j-4)   (matches frag_from_acc below)
template <int HD>
__device__ __forceinline__ bf16x8 frag_tr(const char* img, int dt, int s2, int lane) {
    const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3;
    const int r = 32 * s2 + 4 * g + q4;
    bf16x4 lo = lds_read_tr16(img + Img<HD>::tr_off(r, dt) + p * 8);
    bf16x4 hi = lds_read_tr16(img + Img<HD>::tr_off(r + 16, dt) + p * 8);
    return cat4(lo, hi);
}

// two 16x16 accumulators (register index = k) -> one B fragment of a 32-deep k-step
__device__ __forceinline__ bf16x8 frag_from_acc(const f32x4& a, const f32x4& b) {
    bf16x8 r;
    r[0] = (short)f2bf(a[0]); r[1] = (short)f2bf(a[1]); r[2] = (short)f2bf(a[2]); r[3] = (short)f2bf(a[3]);
    r[4] = (short)f2bf(b[0]); r[5] = (short)f2bf(b[1]); r[6] = (short)f2bf(b[2]); r[7] = (short)f2bf(b[3]);
    return r;
}

template <int HD>
__device__ __forceinline__ void load_rowfrags(const bf16_t* __restrict__ P, int ld, int row, bf16x8 (&f)[HD / 32], int lane) {
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks)
        f[ks] = *reinterpret_cast<const bf16x8*>(P + (size_t)row * ld + ks * 32 + (lane >> 4) * 8);
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------------
// forward (BWD=false) and dQ (BWD=true) share one skeleton: block = 64 queries of one (b, h); wave = 16 queries.
template <int HD, bool BWD>
__global__ __launch_bounds__(256) void attn_q_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                     float* __restrict__ lse, const bf16_t* __restrict__ dout,
                                                     const float* __restrict__ delta, bf16_t* __restrict__ dqkv,
                                                     int S, int H, int KV, float scale) {
    constexpr int NKS = HD / 32, NDT = HD / 16, NST = HD / 32;
    using I = Img<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_row = smem;
    char* v_img = smem + I::BYTES;          // fwd: V tr image;  bwd: V row image
    char* k_tr = smem + 2 * I::BYTES;       // bwd only

    const int qb = gridDim.x - 1 - blockIdx.x;  // heaviest (most key blocks) first
    const int h = blockIdx.y, b = blockIdx.z;
    const int kvh = h / (H / KV);
    const int ld = (H + 2 * KV) * HD;
    const bf16_t* Qp = qkv + (size_t)b * S * ld + h * HD;
    const bf16_t* Kp = qkv + (size_t)b * S * ld + (H + kvh) * HD;
    const bf16_t* Vp = qkv + (size_t)b * S * ld + (H + KV + kvh) * HD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int qrow = qb * 64 + wave * 16 + (lane & 15);
    const int qrow_c = qrow < S ? qrow : S - 1;
    const float c2 = scale * 1.4426950408889634f;

    bf16x8 qf[NKS], dof[NKS];
    load_rowfrags<HD>(Qp, ld, qrow_c, qf, lane);
    float my_lse = 0.f, my_delta = 0.f;
    if (BWD) {
        load_rowfrags<HD>(dout + (size_t)b * S * (H * HD) + h * HD, H * HD, qrow_c, dof, lane);
        my_lse = lse[((size_t)b * H + h) * S + qrow_c] * 1.4426950408889634f;
        my_delta = delta[((size_t)b * H + h) * S + qrow_c];
    }

    f32x4 o[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;

    U4 kreg[NST], vreg[NST];
    tile_load<HD>(Kp, ld, S, 0, kreg);
    tile_load<HD>(Vp, ld, S, 0, vreg);

    for (int kb = 0; kb <= qb; ++kb) {
        __syncthreads();
        tile_store<HD, true, BWD>(k_row, k_tr, kreg);
        tile_store<HD, BWD, !BWD>(v_img, v_img, vreg);
        __syncthreads();
        if (kb < qb) {
            tile_load<HD>(Kp, ld, S, (kb + 1) * 64, kreg);
            tile_load<HD>(Vp, ld, S, (kb + 1) * 64, vreg);
        }
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) s[kt] = MFMA(frag_row<HD>(k_row, 16 * kt, ks, lane), qf[ks], s[kt]);
            if (BWD) {
                dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) dp[kt] = MFMA(frag_row<HD>(v_img, 16 * kt, ks, lane), dof[ks], dp[kt]);
            }
        }
        // s[kt][r] = S[key = kb*64 + 16kt + 4g + r][q = qrow]
        if (kb == qb) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kb * 64 + 16 * kt + 4 * g + r > qrow) s[kt][r] = -INFINITY;
        }
        if (!BWD) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m, mx);
            const float alpha = exp2f((m - m_new) * c2);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = exp2f((s[kt][r] - m_new) * c2);
                    s[kt][r] = p;
                    sum += p;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            l = l * alpha + sum;
            m = m_new;
#pragma unroll
            for (int i = 0; i < NDT; ++i) o[i] *= alpha;
        } else {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = exp2f(s[kt][r] * c2 - my_lse);   // masked: exp2(-inf) = 0
                    s[kt][r] = p * (dp[kt][r] - my_delta) * scale;   // dS^T
                }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = frag_from_acc(s[2 * s2], s[2 * s2 + 1]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) o[dt] = MFMA(frag_tr<HD>(BWD ? k_tr : v_img, dt, s2, lane), pf, o[dt]);
        }
    }

    // o[dt][r] = O^T (or dQ^T) [d = 16dt + 4g + r][q = qrow]
    if (qrow < S) {
        if (!BWD) {
            const float inv = 1.f / l;
            bf16_t* op = out + ((size_t)b * S + qrow) * (H * HD) + h * HD;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                uint2 w; w.x = pack2bf(o[dt][0] * inv, o[dt][1] * inv); w.y = pack2bf(o[dt][2] * inv, o[dt][3] * inv);
                *reinterpret_cast<uint2*>(op + 16 * dt + 4 * g) = w;
            }
            if (g == 0) lse[((size_t)b * H + h) * S + qrow] = m * scale + logf(l);
        } else {
            bf16_t* op = dqkv + ((size_t)b * S + qrow) * ld + h * HD;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                uint2 w; w.x = pack2bf(o[dt][0], o[dt][1]); w.y = pack2bf(o[dt][2], o[dt][3]);
                *reinterpret_cast<uint2*>(op + 16 * dt + 4 * g) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dK / dV: block = 64*KT keys of one (b, kv-head); wave = KT tiles of 16 keys; loops q-heads of the group and
// q-blocks at or below the diagonal.
template <int HD, int KT>
__global__ __launch_bounds__(256) void attn_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                       const float* __restrict__ lse, const float* __restrict__ delta,
                                                       bf16_t* __restrict__ dqkv, int S, int H, int KV, float scale) {
    constexpr int NKS = HD / 32, NDT = HD / 16, NST = HD / 32, KB = 64 * KT;
    using I = Img<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* q_row = smem;
    char* q_tr = smem + I::BYTES;
    char* do_row = smem + 2 * I::BYTES;
    char* do_tr = smem + 3 * I::BYTES;

    const int kblk = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int rep = H / KV;
    const int ld = (H + 2 * KV) * HD, ldo = H * HD;
    const bf16_t* Kp = qkv + (size_t)b * S * ld + (H + kvh) * HD;
    const bf16_t* Vp = qkv + (size_t)b * S * ld + (H + KV + kvh) * HD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const float c2 = scale * 1.4426950408889634f;

    bf16x8 kf[KT][NKS], vf[KT][NKS];
    int key[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        key[t] = kblk * KB + (wave * KT + t) * 16 + (lane & 15);
        const int kc = key[t] < S ? key[t] : S - 1;
        load_rowfrags<HD>(Kp, ld, kc, kf[t], lane);
        load_rowfrags<HD>(Vp, ld, kc, vf[t], lane);
    }
    f32x4 dk[KT][NDT], dv[KT][NDT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int i = 0; i < NDT; ++i) { dk[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int nqb = (S + 63) / 64;
    const int qb0 = (kblk * KB) / 64;
    const int niter = rep * (nqb - qb0);
    U4 qreg[NST], dreg[NST];
    if (niter > 0) {
        tile_load<HD>(qkv + (size_t)b * S * ld + (kvh * rep) * HD, ld, S, qb0 * 64, qreg);
        tile_load<HD>(dout + (size_t)b * S * ldo + (kvh * rep) * HD, ldo, S, qb0 * 64, dreg);
    }
    for (int it = 0; it < niter; ++it) {
        const int hh = it / (nqb - qb0), qb = qb0 + it % (nqb - qb0);
        const int h = kvh * rep + hh;
        __syncthreads();
        tile_store<HD, true, true>(q_row, q_tr, qreg);
        tile_store<HD, true, true>(do_row, do_tr, dreg);
        __syncthreads();
        if (it + 1 < niter) {
            const int hh1 = (it + 1) / (nqb - qb0), qb1 = qb0 + (it + 1) % (nqb - qb0);
            tile_load<HD>(qkv + (size_t)b * S * ld + (kvh * rep + hh1) * HD, ld, S, qb1 * 64, qreg);
            tile_load<HD>(dout + (size_t)b * S * ldo + (kvh * rep + hh1) * HD, ldo, S, qb1 * 64, dreg);
        }
        // per-register row constants: q = qb*64 + 16qt + 4g + r
        float lse4[4][4], del4[4][4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = qb * 64 + 16 * qt + 4 * g + r;
                const int qc = q < S ? q : S - 1;
                lse4[qt][r] = lse[((size_t)b * H + h) * S + qc] * 1.4426950408889634f;
                del4[qt][r] = delta[((size_t)b * H + h) * S + qc];
            }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            f32x4 s[4], dp[4];
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                s[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dp[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    s[qt] = MFMA(frag_row<HD>(q_row, 16 * qt, ks, lane), kf[t][ks], s[qt]);
                    dp[qt] = MFMA(frag_row<HD>(do_row, 16 * qt, ks, lane), vf[t][ks], dp[qt]);
                }
                // s[qt][r] = S[q = qb*64+16qt+4g+r][key = key[t]]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qb * 64 + 16 * qt + 4 * g + r;
                    float p = exp2f(s[qt][r] * c2 - lse4[qt][r]);
                    if (key[t] > q || q >= S) p = 0.f;
                    s[qt][r] = p;
                    dp[qt][r] = p * (dp[qt][r] - del4[qt][r]) * scale;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = frag_from_acc(s[2 * s2], s[2 * s2 + 1]);
                const bf16x8 dsf = frag_from_acc(dp[2 * s2], dp[2 * s2 + 1]);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    dv[t][dt] = MFMA(frag_tr<HD>(do_tr, dt, s2, lane), pf, dv[t][dt]);
                    dk[t][dt] = MFMA(frag_tr<HD>(q_tr, dt, s2, lane), dsf, dk[t][dt]);
                }
            }
        }
    }
    // dk[t][dt][r] = dK^T[d = 16dt + 4g + r][key = key[t]]
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        if (key[t] >= S) continue;
        bf16_t* kp = dqkv + ((size_t)b * S + key[t]) * ld + (H + kvh) * HD;
        bf16_t* vp = dqkv + ((size_t)b * S + key[t]) * ld + (H + KV + kvh) * HD;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            uint2 w; w.x = pack2bf(dk[t][dt][0], dk[t][dt][1]); w.y = pack2bf(dk[t][dt][2], dk[t][dt][3]);
            *reinterpret_cast<uint2*>(kp + 16 * dt + 4 * g) = w;
            uint2 u; u.x = pack2bf(dv[t][dt][0], dv[t][dt][1]); u.y = pack2bf(dv[t][dt][2], dv[t][dt][3]);
            *reinterpret_cast<uint2*>(vp + 16 * dt + 4 * g) = u;
        }
    }
}

// delta[b,h,s] = sum_d dO * O   (one 16-lane group per (row, head))
__global__ void attn_delta_kernel(const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout, float* __restrict__ delta,
                                  int B, int S, int H, int HD) {
    const long long gid = (long long)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const long long total = (long long)B * S * H;
    const int sub = threadIdx.x & 15;
    float acc = 0.f;
    long long row = 0; int h = 0;
    const bool ok = gid < total;
    if (ok) {
        row = gid / H; h = (int)(gid % H);
        const bf16_t* o = out + (size_t)row * H * HD + h * HD;
        const bf16_t* d = dout + (size_t)row * H * HD + h * HD;
        for (int c = sub * 8; c < HD; c += 128) {
            float a[8], bb[8];
            unpack8(*reinterpret_cast<const U4*>(o + c), a);
            unpack8(*reinterpret_cast<const U4*>(d + c), bb);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += a[i] * bb[i];
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (ok && sub == 0) {
        const long long bidx = row / S, s = row % S;
        delta[((size_t)bidx * H + h) * S + s] = acc;
    }
}

}  // namespace

static int check_attn(const char* name, int B, int S, int H, int KV, int HD) {
    CSM_REQUIRE(B > 0 && S > 0 && H > 0 && KV > 0 && H % KV == 0, "%s: bad shape B=%d S=%d H=%d KV=%d", name, B, S, H, KV);
    CSM_REQUIRE(HD == 64 || HD == 128, "%s: head_dim %d unsupported (64 or 128)", name, HD);
    return 0;
}

extern "C" int csm_attn_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int KV, int HD,
                            hipStream_t stream) {
    if (int e = check_attn("csm_attn_fwd", B, S, H, KV, HD)) return e;
    CSM_REQUIRE(qkv && out && lse, "csm_attn_fwd: null pointer");
    dim3 grid((S + 63) / 64, H, B), block(256);
    const float scale = 1.f / sqrtf((float)HD);
    if (HD == 64)
        hipLaunchKernelGGL((attn_q_kernel<64, false>), grid, block, 2 * Img<64>::BYTES, stream, (const bf16_t*)qkv,
                           (bf16_t*)out, lse, nullptr, nullptr, nullptr, S, H, KV, scale);
    else
        hipLaunchKernelGGL((attn_q_kernel<128, false>), grid, block, 2 * Img<128>::BYTES, stream, (const bf16_t*)qkv,
                           (bf16_t*)out, lse, nullptr, nullptr, nullptr, S, H, KV, scale);
    CSM_CHECK_LAUNCH("csm_attn_fwd");
    return 0;
}

extern "C" int csm_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                            float* delta_ws, int B, int S, int H, int KV, int HD, hipStream_t stream) {
    if (int e = check_attn("csm_attn_bwd", B, S, H, KV, HD)) return e;
    CSM_REQUIRE(qkv && out && dout && lse && dqkv && delta_ws, "csm_attn_bwd: null pointer");
    const float scale = 1.f / sqrtf((float)HD);
    {
        const long long total = (long long)B * S * H;
        const int per_block = 256 / 16;
        hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + per_block - 1) / per_block)), dim3(256), 0, stream,
                           (const bf16_t*)out, (const bf16_t*)dout, delta_ws, B, S, H, HD);
        CSM_CHECK_LAUNCH("csm_attn_bwd(delta)");
    }
    dim3 block(256);
    if (HD == 64) {
        hipLaunchKernelGGL((attn_dkv_kernel<64, 2>), dim3((S + 127) / 128, KV, B), block, 4 * Img<64>::BYTES, stream,
                           (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta_ws, (bf16_t*)dqkv, S, H, KV, scale);
        CSM_CHECK_LAUNCH("csm_attn_bwd(dkv)");
        hipLaunchKernelGGL((attn_q_kernel<64, true>), dim3((S + 63) / 64, H, B), block, 3 * Img<64>::BYTES, stream,
                           (const bf16_t*)qkv, nullptr, const_cast<float*>(lse), (const bf16_t*)dout, delta_ws,
                           (bf16_t*)dqkv, S, H, KV, scale);
    } else {
        hipLaunchKernelGGL((attn_dkv_kernel<128, 1>), dim3((S + 63) / 64, KV, B), block, 4 * Img<128>::BYTES, stream,
                           (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta_ws, (bf16_t*)dqkv, S, H, KV, scale);
        CSM_CHECK_LAUNCH("csm_attn_bwd(dkv)");
        hipLaunchKernelGGL((attn_q_kernel<128, true>), dim3((S + 63) / 64, H, B), block, 3 * Img<128>::BYTES, stream,
                           (const bf16_t*)qkv, nullptr, const_cast<float*>(lse), (const bf16_t*)dout, delta_ws,
                           (bf16_t*)dqkv, S, H, KV, scale);
    }
    CSM_CHECK_LAUNCH("csm_attn_bwd(dq)");
    return 0;
}
