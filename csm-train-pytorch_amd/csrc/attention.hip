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
// LDS images: ONE layout serves both kinds of read of a 64 x HD tile: 32-B slot s of row r sits at slot s ^ swz(r)
// (swz = (r>>1)&3 for HD 64, r&7 for HD 128).  ds_read_b128 row fragments (k along d) and ds_read_b64_tr_b16 transposed
// fragments (k along the rows) are both bank-conflict-free on it (checked lane group by lane group against the gfx950
// banking rules), so a tile that is needed in both forms - K in the dQ kernel, Q and dO in the dK/dV kernel - is
// written to LDS once.
#include "common.h"
#include <math.h>

namespace {

template <int HD>
struct Img {
    static constexpr int ROWB = HD * 2;
    static constexpr int CPR = HD / 8;          // 16-B chunks per row
    static constexpr int BYTES = 64 * ROWB;     // one 64-row image
    __device__ static __forceinline__ int tr_off(int r, int slot) {
        const int sw = (HD == 64) ? ((r >> 1) & 3) : (r & 7);
        return r * ROWB + ((slot ^ sw) << 5);
    }
    __device__ static __forceinline__ int row_off(int r, int c) { return tr_off(r, c >> 1) + ((c & 1) << 4); }   // 16-B chunk c
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

template <int HD>
__device__ __forceinline__ void tile_store(char* img, const U4 (&reg)[HD / 32]) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int i = 0; i < HD / 32; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx / CPR, c = idx % CPR;
        *reinterpret_cast<U4*>(img + Img<HD>::row_off(r, c)) = reg[i];
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

// raw v_exp_f32 (no denormal-range fix-up: arguments here are <= ~0 and a flushed tiny result is exact enough)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max / sum across the four 16-lane rows of a wave with v_permlane16_swap + v_permlane32_swap (VALU, no LDS
// round trip like __shfl_xor's ds_bpermute).  vdst = src = x: the swap leaves {even-row value, odd-row value} pairs.
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // no canonicalising v_max in front
    return r;
}
__device__ __forceinline__ float rows_max(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    float a;
    asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(__uint_as_float(r[0])), "v"(__uint_as_float(r[1])));
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
    asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(__uint_as_float(q[0])), "v"(__uint_as_float(q[1])));
    return a;
}
__device__ __forceinline__ float rows_sum(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// Backward of torchtune's interleaved-pair RoPE on two adjacent pairs (x0,x1), (x2,x3) of position p: the transpose
// rotation, (g0 c + g1 s, g1 c - g0 s).  table = [P][hd/2][2] (cos, sin) fp32 as for csm_rope; pair = first pair index.
// Fusing it into the dQ / dK epilogues saves the separate in-place pass over the [M, (H+2KV) hd] gradient.
__device__ __forceinline__ void unrope2(const float* __restrict__ table, int p, int hd, int pair, float& g0, float& g1, float& g2,
                                        float& g3) {
    const float4 t = *reinterpret_cast<const float4*>(table + ((size_t)p * (hd >> 1) + pair) * 2);   // c0, s0, c1, s1
    rope_rot(g0, g1, t.x, -t.y);
    rope_rot(g2, g3, t.z, -t.w);
}

// experiment switches for tools/probes (never set in the shipped build): what bounds the dK/dV loop?
// bit0: no global prefetch / LDS commit after the first tile; bit1: no softmax VALU work; bit2: no LDS fragment reads
#ifndef CSM_ATT_ABLATE
#define CSM_ATT_ABLATE 0
#endif
constexpr bool AAB_G = CSM_ATT_ABLATE & 1, AAB_V = CSM_ATT_ABLATE & 2, AAB_L = CSM_ATT_ABLATE & 4;

// ------------------------------------------------------------------------------------------------
// forward (BWD=false) and dQ (BWD=true) share one skeleton: block = 64*QT queries of one (b, h); wave = QT tiles of
// 16 queries, so every K / V fragment read from LDS feeds QT MFMAs.  K/V tiles are double-buffered in LDS
// (global -> registers one key block ahead -> LDS after the compute), one barrier per key block.
// GRP (round 4, short sequences: S <= 16 QT, four q heads per kv head - the depth decoder's 32-position frames): a workgroup is one
// (batch, kv head) pair and wave w takes ALL queries of q head w of the group, so the four waves share one staging of the
// group's K / V (the block-per-head mapping staged them once per head and left half of every workgroup's waves without queries).
// Every query tile still sees the same key tiles in the same order: the same bits.
template <int HD, bool BWD, int QT, bool GRP = false>
__global__ __launch_bounds__(256, 2) void attn_q_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                     float* __restrict__ lse, const bf16_t* __restrict__ dout,
                                                     float* __restrict__ delta, bf16_t* __restrict__ dqkv,
                                                     int S, int H, int KV, float scale, int lpt, const float* __restrict__ rope) {
    constexpr int NKS = HD / 32, NDT = HD / 16, NST = HD / 32;
    constexpr int NIMG = 2;
    using I = Img<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto k_img = [&](int st) { return smem + st * NIMG * I::BYTES; };                  // row frags (S^T) and, in dQ, tr frags (K^T)
    auto v_img = [&](int st) { return smem + st * NIMG * I::BYTES + I::BYTES; };      // fwd: tr frags (V^T); dQ: row frags

    // 1-D grid, XCD-aware: workgroup ids are dealt round-robin over the 8 XCDs; the bijective remap below hands every
    // XCD a contiguous run of work items, ordered (batch, kv-head) pair -> q-head of the group -> q-block (heaviest
    // first), so the workgroups resident on one XCD read the same K/V panels through the same 4-MiB L2.
    const int nqblk = (S + 64 * QT - 1) / (64 * QT);
    const int rep_ = H / KV;
    const int per_pair = rep_ * nqblk;
    int pair, local;
    {
        const int T = gridDim.x, id = blockIdx.x, xcd = id & 7, within = id >> 3;
        const int q8 = T >> 3, r8 = T & 7;
        const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
        pair = nid / per_pair;
        local = nid % per_pair;
        const int run = (xcd < r8) ? q8 + 1 : q8;
        const int base = nid - within;
        if (lpt && run % per_pair == 0 && base % per_pair == 0) {   // longest first over the whole run: q-block major, (pair, head) minor
            const int ncomb = (run / per_pair) * rep_;
            const int comb = within % ncomb;
            pair = base / per_pair + comb / rep_;
            local = (comb % rep_) * nqblk + within / ncomb;
        }
    }
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (GRP) { pair = blockIdx.x; local = 0; }
    const int qb = nqblk - 1 - (local % nqblk);
    const int kvh_ = pair % KV, b = pair / KV;
    const int h = GRP ? kvh_ * rep_ + wave : kvh_ * rep_ + local / nqblk;
    const int kvh = kvh_;
    const int ld = (H + 2 * KV) * HD;
    const bf16_t* Qp = qkv + (size_t)b * S * ld + h * HD;
    const bf16_t* Kp = qkv + (size_t)b * S * ld + (H + kvh) * HD;
    const bf16_t* Vp = qkv + (size_t)b * S * ld + (H + KV + kvh) * HD;
    const int q_base = GRP ? 0 : qb * 64 * QT;
    const float c2 = scale * 1.4426950408889634f;

    int r0[QT], qrow[QT];
    bf16x8 qf[QT][NKS], dof[QT][NKS];
    float nlse2[QT], my_delta[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        r0[qt] = GRP ? qt * 16 : q_base + (wave * QT + qt) * 16;
        qrow[qt] = r0[qt] + (lane & 15);
        const int qc = qrow[qt] < S ? qrow[qt] : S - 1;
        load_rowfrags<HD>(Qp, ld, qc, qf[qt], lane);
        if (BWD) {
            load_rowfrags<HD>(dout + (size_t)b * S * (H * HD) + h * HD, H * HD, qc, dof[qt], lane);
            nlse2[qt] = -lse[((size_t)b * H + h) * S + qc] * 1.4426950408889634f;
            // delta = rowsum(dO * O) of this query: computed here from the same row fragments and published for the
            // dK/dV kernel, which runs after this one (saves a separate pass over O and dO)
            bf16x8 of[NKS];
            load_rowfrags<HD>(out + (size_t)b * S * (H * HD) + h * HD, H * HD, qc, of, lane);
            float dsum = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) dsum += bf2f((bf16_t)dof[qt][ks][j]) * bf2f((bf16_t)of[ks][j]);
            my_delta[qt] = rows_sum(dsum);
            if (g == 0 && qrow[qt] < S) delta[((size_t)b * H + h) * S + qrow[qt]] = my_delta[qt];
        }
    }

    f32x4 o[QT][NDT];
    float m[QT], l[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m[qt] = -INFINITY; l[qt] = 0.f;
#pragma unroll
        for (int i = 0; i < NDT; ++i) o[qt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    int last = GRP ? 16 * QT - 1 : q_base + 64 * QT - 1;
    last = last < S ? last : S - 1;
    const int nkb = last / 64 + 1;
    U4 kreg[NST], vreg[NST];
    tile_load<HD>(Kp, ld, S, 0, kreg);
    tile_load<HD>(Vp, ld, S, 0, vreg);
    tile_store<HD>(k_img(0), kreg);
    tile_store<HD>(v_img(0), vreg);
    __syncthreads();

    for (int kb = 0; kb < nkb; ++kb) {
        const int st = kb & 1;
        if (kb + 1 < nkb) {
            tile_load<HD>(Kp, ld, S, (kb + 1) * 64, kreg);
            tile_load<HD>(Vp, ld, S, (kb + 1) * 64, vreg);
        }
        const int key0 = kb * 64;
        bool active[QT];
        bool any = false;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) { active[qt] = key0 <= r0[qt] + 15; any |= active[qt]; }
        if (any) {
            f32x4 s[QT][4], dp[QT][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                bf16x8 kfr[NKS], vfr[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    kfr[ks] = frag_row<HD>(k_img(st), 16 * kt, ks, lane);
                    if (BWD) vfr[ks] = frag_row<HD>(v_img(st), 16 * kt, ks, lane);
                }
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    s[qt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (BWD) dp[qt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (active[qt]) {
#pragma unroll
                        for (int ks = 0; ks < NKS; ++ks) {
                            s[qt][kt] = MFMA(kfr[ks], qf[qt][ks], s[qt][kt]);
                            if (BWD) dp[qt][kt] = MFMA(vfr[ks], dof[qt][ks], dp[qt][kt]);
                        }
                    }
                }
            }
            bf16x8 pf[QT][2];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (!active[qt]) {
                    pf[qt][0] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    pf[qt][1] = pf[qt][0];
                    continue;
                }
                // s[qt][kt][r] = S[key = key0 + 16kt + 4g + r][q = qrow[qt]]
                if (key0 + 63 > r0[qt]) {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (key0 + 16 * kt + 4 * g + r > qrow[qt]) s[qt][kt][r] = -INFINITY;
                }
                if (!BWD) {
                    float mx = max3(s[qt][0][0], s[qt][0][1], s[qt][0][2]);
                    mx = max3(mx, s[qt][0][3], s[qt][1][0]);
                    mx = max3(mx, s[qt][1][1], s[qt][1][2]);
                    mx = max3(mx, s[qt][1][3], s[qt][2][0]);
                    mx = max3(mx, s[qt][2][1], s[qt][2][2]);
                    mx = max3(mx, s[qt][2][3], s[qt][3][0]);
                    mx = max3(mx, s[qt][3][1], s[qt][3][2]);
                    mx = rows_max(max3(mx, s[qt][3][3], m[qt]));     // running max folded in
                    const float m_new = mx;
                    if (!__all(m_new == m[qt])) {          // wave-uniform: rescale only when some row's max moved
                        const float alpha = fast_exp2((m[qt] - m_new) * c2);
                        l[qt] *= alpha;
#pragma unroll
                        for (int i = 0; i < NDT; ++i) o[qt][i] *= alpha;
                        m[qt] = m_new;
                    }
                    const float nb = -m_new * c2;
                    float sum = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = fast_exp2(fmaf(s[qt][kt][r], c2, nb));
                            s[qt][kt][r] = p;
                            sum += p;
                        }
                    l[qt] += rows_sum(sum);
                } else {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = fast_exp2(fmaf(s[qt][kt][r], c2, nlse2[qt]));   // masked: exp2(-inf) = 0
                            s[qt][kt][r] = p * (dp[qt][kt][r] - my_delta[qt]);  // dS^T / scale (the factor goes on dQ below)
                        }
                }
                pf[qt][0] = frag_from_acc(s[qt][0], s[qt][1]);
                pf[qt][1] = frag_from_acc(s[qt][2], s[qt][3]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const bf16x8 vt = frag_tr<HD>(BWD ? k_img(st) : v_img(st), dt, s2, lane);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        if (active[qt]) o[qt][dt] = MFMA(vt, pf[qt][s2], o[qt][dt]);
                }
        }
        if (kb + 1 < nkb) {
            tile_store<HD>(k_img(st ^ 1), kreg);
            tile_store<HD>(v_img(st ^ 1), vreg);
        }
        __syncthreads();
    }

    // o[qt][dt][r] = O^T (or dQ^T) [d = 16dt + 4g + r][q = qrow[qt]]
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        if (qrow[qt] >= S) continue;
        if (!BWD) {
            const float inv = 1.f / l[qt];
            bf16_t* op = out + ((size_t)b * S + qrow[qt]) * (H * HD) + h * HD;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                uint2 w; w.x = pack2bf(o[qt][dt][0] * inv, o[qt][dt][1] * inv); w.y = pack2bf(o[qt][dt][2] * inv, o[qt][dt][3] * inv);
                *reinterpret_cast<uint2*>(op + 16 * dt + 4 * g) = w;
            }
            if (g == 0) lse[((size_t)b * H + h) * S + qrow[qt]] = m[qt] * scale + logf(l[qt]);
        } else {
            bf16_t* op = dqkv + ((size_t)b * S + qrow[qt]) * ld + h * HD;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                float v0 = o[qt][dt][0] * scale, v1 = o[qt][dt][1] * scale, v2 = o[qt][dt][2] * scale, v3 = o[qt][dt][3] * scale;
                if (rope) unrope2(rope, qrow[qt], HD, 8 * dt + 2 * g, v0, v1, v2, v3);   // gradient w.r.t. the un-rotated q
                uint2 w; w.x = pack2bf(v0, v1); w.y = pack2bf(v2, v3);
                *reinterpret_cast<uint2*>(op + 16 * dt + 4 * g) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dK / dV: block = 64*KT keys of one (b, kv-head); wave = KT tiles of 16 keys; loops the q-heads of the group and the
// q-blocks at or below the diagonal.  Q / dO tiles (one image each, read both ways) and the block's lse / delta are
// double-buffered in LDS; one barrier per q-block.
template <int HD, int KT>
__global__ __launch_bounds__(256, 2) void attn_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          bf16_t* __restrict__ dqkv, int S, int H, int KV, float scale, int map,
                                                          const float* __restrict__ rope) {
    constexpr int NKS = HD / 32, NDT = HD / 16, NST = HD / 32, KB = 64 * KT;
    using I = Img<HD>;
    constexpr int STAGE = 2 * I::BYTES + 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto q_img = [&](int st) { return smem + st * STAGE; };                    // row frags (S) and tr frags (dK)
    auto do_img = [&](int st) { return smem + st * STAGE + I::BYTES; };        // row frags (dP) and tr frags (dV)
    auto stat = [&](int st) { return reinterpret_cast<float*>(smem + st * STAGE + 2 * I::BYTES); };  // [0..63] lse*log2e, [64..127] delta

    // 1-D XCD-aware grid (see attn_q_kernel): contiguous run of (batch, kv-head) pairs per XCD, heaviest key block first
    const int nkblk = (S + KB - 1) / KB;
    int kblk, kvh, b;
    {
        const int T = gridDim.x, id = blockIdx.x, xcd = id & 7, within = id >> 3;
        const int q8 = T >> 3, r8 = T & 7;
        const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
        const int pair = nid / nkblk;
        kblk = nid % nkblk;
        // Causal work falls linearly with the key block and the grid is one wave of workgroups (two per CU), so the
        // two workgroups that share a CU should hold complementary key blocks.  map 1 assumes the dispatcher deals an
        // XCD's workgroups breadth-first (w and w + run/2 share a CU), map 2 depth-first (w and w ^ 1 share a CU).
        const int run = (xcd < r8) ? q8 + 1 : q8, half = run >> 1;
        if (map == 1 && (run & 1) == 0 && half % nkblk == 0 && within >= half) kblk = nkblk - 1 - kblk;
        if (map == 2 && (nkblk & 1) == 0 && (within & 1)) kblk = nkblk - 1 - (kblk ^ 1);
        int pair_ = pair;
        const int base = nid - within;
        if (map == 3 && run % nkblk == 0 && base % nkblk == 0) {      // heaviest key blocks of every (b, kv-head) of the run first
            const int npairs = run / nkblk;
            kblk = within / npairs;
            pair_ = base / nkblk + within % npairs;
        }
        kvh = pair_ % KV;
        b = pair_ / KV;
    }
    const int rep = H / KV;
    const int ld = (H + 2 * KV) * HD, ldo = H * HD;
    const bf16_t* Kp = qkv + (size_t)b * S * ld + (H + kvh) * HD;
    const bf16_t* Vp = qkv + (size_t)b * S * ld + (H + KV + kvh) * HD;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float c2 = scale * 1.4426950408889634f;

    bf16x8 kf[KT][NKS], vf[KT][NKS];
    int key[KT], key0[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        key0[t] = kblk * KB + (wave * KT + t) * 16;
        key[t] = key0[t] + (lane & 15);
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
    const int per_head = nqb - qb0;
    const int niter = rep * per_head;
    U4 qreg[NST], dreg[NST];
    float streg = 0.f;
    auto prefetch = [&](int it) {
        const int hh = it / per_head, qb = qb0 + it % per_head;
        const int h = kvh * rep + hh;
        tile_load<HD>(qkv + (size_t)b * S * ld + h * HD, ld, S, qb * 64, qreg);
        tile_load<HD>(dout + (size_t)b * S * ldo + h * HD, ldo, S, qb * 64, dreg);
        if (threadIdx.x < 128) {
            int q = qb * 64 + (threadIdx.x & 63);
            q = q < S ? q : S - 1;
            const size_t idx = ((size_t)b * H + h) * S + q;
            streg = threadIdx.x < 64 ? lse[idx] * 1.4426950408889634f : delta[idx];
        }
    };
    auto commit = [&](int st) {
        tile_store<HD>(q_img(st), qreg);
        tile_store<HD>(do_img(st), dreg);
        if (threadIdx.x < 128) stat(st)[threadIdx.x] = streg;
    };
    if (niter > 0) { prefetch(0); commit(0); }
    __syncthreads();

    for (int it = 0; it < niter; ++it) {
        const int st = AAB_G ? 0 : (it & 1);
        const int qb = qb0 + it % per_head;
        if (!AAB_G && it + 1 < niter) prefetch(it + 1);
        bool active[KT];
        bool any = false;
#pragma unroll
        for (int t = 0; t < KT; ++t) { active[t] = key0[t] <= qb * 64 + 63; any |= active[t]; }
        if (any) {
            f32x4 s[KT][4], dp[KT][4];
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const float4 l4 = *reinterpret_cast<const float4*>(stat(st) + 16 * qt + 4 * g);
                const float4 d4 = *reinterpret_cast<const float4*>(stat(st) + 64 + 16 * qt + 4 * g);
                const float lse4[4] = {l4.x, l4.y, l4.z, l4.w}, del4[4] = {d4.x, d4.y, d4.z, d4.w};
                bf16x8 qfr[NKS], dofr[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    qfr[ks] = frag_row<HD>(q_img(st), 16 * qt, ks, lane);
                    dofr[ks] = frag_row<HD>(do_img(st), 16 * qt, ks, lane);
                }
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    s[t][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    dp[t][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (!active[t]) continue;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        s[t][qt] = MFMA(qfr[ks], kf[t][ks], s[t][qt]);
                        dp[t][qt] = MFMA(dofr[ks], vf[t][ks], dp[t][qt]);
                    }
                    // s[t][qt][r] = S[q = qb*64+16qt+4g+r][key = key[t]]
                    if (!AAB_V) {
                        // the causal / tail mask only bites on the diagonal q-tile and in the last q-block (wave-uniform
                        // test); everywhere else an element costs fma + exp2 + sub + mul (dS is left unscaled: the
                        // 1/sqrt(hd) factor is applied once to dK in the epilogue)
                        const int qlo = qb * 64 + 16 * qt;
                        const bool edge = (qlo < key0[t] + 16) || (qlo + 16 > S);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float p = fast_exp2(fmaf(s[t][qt][r], c2, -lse4[r]));
                            if (edge) {
                                const int q = qlo + 4 * g + r;
                                if (key[t] > q || q >= S) p = 0.f;
                            }
                            s[t][qt][r] = p;
                            dp[t][qt][r] = p * (dp[t][qt][r] - del4[r]);
                        }
                    }
                }
            }
            bf16x8 pf[KT][2], dsf[KT][2];
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    pf[t][s2] = frag_from_acc(s[t][2 * s2], s[t][2 * s2 + 1]);
                    dsf[t][s2] = frag_from_acc(dp[t][2 * s2], dp[t][2 * s2 + 1]);
                }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const bf16x8 dot = frag_tr<HD>(do_img(st), dt, s2, lane);
                    const bf16x8 qtt = frag_tr<HD>(q_img(st), dt, s2, lane);
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        if (!active[t]) continue;
                        dv[t][dt] = MFMA(dot, pf[t][s2], dv[t][dt]);
                        dk[t][dt] = MFMA(qtt, dsf[t][s2], dk[t][dt]);
                    }
                }
        }
        if (!AAB_G && it + 1 < niter) commit(st ^ 1);
        __syncthreads();
    }
    // dk[t][dt][r] = dK^T[d = 16dt + 4g + r][key = key[t]]
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        if (key[t] >= S) continue;
        bf16_t* kp = dqkv + ((size_t)b * S + key[t]) * ld + (H + kvh) * HD;
        bf16_t* vp = dqkv + ((size_t)b * S + key[t]) * ld + (H + KV + kvh) * HD;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            float v0 = dk[t][dt][0] * scale, v1 = dk[t][dt][1] * scale, v2 = dk[t][dt][2] * scale, v3 = dk[t][dt][3] * scale;
            if (rope) unrope2(rope, key[t], HD, 8 * dt + 2 * g, v0, v1, v2, v3);         // gradient w.r.t. the un-rotated k
            uint2 w; w.x = pack2bf(v0, v1); w.y = pack2bf(v2, v3);
            *reinterpret_cast<uint2*>(kp + 16 * dt + 4 * g) = w;
            uint2 u; u.x = pack2bf(dv[t][dt][0], dv[t][dt][1]); u.y = pack2bf(dv[t][dt][2], dv[t][dt][3]);
            *reinterpret_cast<uint2*>(vp + 16 * dt + 4 * g) = u;
        }
    }
}

}  // namespace

static int check_attn(const char* name, int B, int S, int H, int KV, int HD) {
    CSM_REQUIRE(B > 0 && S > 0 && H > 0 && KV > 0 && H % KV == 0, "%s: bad shape B=%d S=%d H=%d KV=%d", name, B, S, H, KV);
    CSM_REQUIRE(HD == 64 || HD == 128, "%s: head_dim %d unsupported (64 or 128)", name, HD);
    return 0;
}

static int g_attn_dkv_map = 3, g_attn_dkv_kt1 = 1, g_attn_q_lpt = 1;   // scheduling switches (csm_set_attn_variant)
template <int HD, bool BWD, int QT, bool GRP = false>
static void launch_q(const void* qkv, void* out, float* lse, const void* dout, float* delta, void* dqkv, int B, int S,
                     int H, int KV, float scale, hipStream_t stream, const float* rope = nullptr) {
    constexpr int lds = 2 * 2 * Img<HD>::BYTES;
    auto k = attn_q_kernel<HD, BWD, QT, GRP>;
    static bool done = false;
    if (!done && lds > 65536) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }
    done = true;
    dim3 grid(GRP ? (unsigned)(KV * B) : (unsigned)(((S + 64 * QT - 1) / (64 * QT)) * H * B)), block(256);
    hipLaunchKernelGGL(k, grid, block, lds, stream, (const bf16_t*)qkv, (bf16_t*)out, lse, (const bf16_t*)dout, delta,
                       (bf16_t*)dqkv, S, H, KV, scale, g_attn_q_lpt, rope);
}
static int g_attn_grp = 1;      // csm_set_attn_variant bit 14 switches the grouped short-sequence mapping off (A/B)

template <int HD, int KT>
static void launch_dkv(const void* qkv, const void* dout, const float* lse, const float* delta, void* dqkv, int B, int S, int H,
                       int KV, float scale, hipStream_t stream, const float* rope = nullptr) {
    constexpr int lds = 2 * (2 * Img<HD>::BYTES + 512);
    auto k = attn_dkv_kernel<HD, KT>;
    static bool done = false;
    if (!done && lds > 65536) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }
    done = true;
    dim3 grid((unsigned)(((S + 64 * KT - 1) / (64 * KT)) * KV * B)), block(256);
    hipLaunchKernelGGL(k, grid, block, lds, stream, (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, S, H, KV,
                       scale, g_attn_dkv_map, rope);
}

static int g_attn_qt_fwd = 2, g_attn_qt_bwd = 1;   // query tiles per wave of the head_dim-64 forward / dQ kernels
static int g_attn_gen2 = 3;                          // head_dim 64: second-generation kernels of attention64.hip (bit0 fwd, bit1 bwd)
int csm_attn64_fwd_launch(const void* qkv, void* out, float* lse, int B, int S, int H, int KV, hipStream_t stream);
int csm_attn64_dkv_launch(const void* qkv, const void* dout, const float* lse, const float* delta, void* dqkv, int B, int S, int H,
                          int KV, const float* rope, hipStream_t stream);
int csm_attn64_dkv_asm_launch(const void* qkv, const void* dout, const float* stats, void* dqkv, int B, int S, int H, int KV,
                              const float* rope, hipStream_t stream);
extern int g_attn64_dkv_asm_order, g_attn64_dq_asm_order;
int csm_attn64_dq_asm_launch(const void* qkv, const void* out, const void* dout, const float* lse, float* stats, void* dqkv, int B, int S,
                             int H, int KV, const float* rope, hipStream_t stream);
// The asm dQ kernel is OFF by default: isolated it beats the compiler-scheduled one (122 vs 134 us at B=4, S=2048, 32/8 heads),
// inside the train step it does not (130.6 vs 128.1 us, profiles/r04_step_attention_kernels.txt): with one wave per SIMD nothing
// covers the moment at which all 256 workgroups fetch their Q / dO / O rows at once.  csm_set_attn_variant bit 12 switches it on.
static int g_attn_last_dkv = 0, g_attn_last_dq = 0, g_attn_dq_asm = 0;
// bit 0: the dK/dV pass, bit 1: the dQ pass of the most recent csm_attn_bwd* call ran the generated-asm kernel
extern "C" int csm_attn_last_dkv_kernel(void) { return g_attn_last_dkv | (g_attn_last_dq << 1); }
static int g_attn_dkv_asm = 1;                       // csm_set_attn_variant bit 10 switches the asm dK/dV kernel off (A/B)
int csm_attn64_dq_launch(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int S,
                         int H, int KV, const float* rope, hipStream_t stream);
extern "C" int csm_set_attn_variant(int v) {
    // 0 restores the defaults.  Otherwise an experiment word: bits 0..1 / 2..3 query tiles per wave of the head_dim-64
    // forward / dQ kernel (1 | 2; 0 = 1); bits 4..5 dK/dV work order (0 plain, 1 / 2 complementary pairing, 3 heaviest key blocks first);
    // bit 6 dK/dV key tile (1: 64 keys per workgroup, 0: 128); bit 7 forward / dQ work order heaviest q-blocks first.
    // bits 8..9: 0 = second-generation head_dim-64 kernels (attention64.hip; the default), 1 = first generation forward,
    // 2 = first generation backward, 3 = both first generation (A/B reference).  bit 10: dK/dV through attention64.hip's
    // second-generation kernel instead of the generated-asm one of attention64_asm.hip; bit 11: that kernel pair by pair;
    // bit 12: dQ through the generated-asm kernel (off by default); bit 13: that kernel one query block per workgroup.
    if (v == 0) v = 2 | (1 << 2) | (3 << 4) | (1 << 6) | (1 << 7);
    g_attn_gen2 = 3 & ~((v >> 8) & 3);
    g_attn_qt_fwd = (v & 3) == 2 ? 2 : 1;
    g_attn_qt_bwd = ((v >> 2) & 3) == 2 ? 2 : 1;
    g_attn_dkv_map = (v >> 4) & 3;
    g_attn_dkv_kt1 = (v >> 6) & 1;
    g_attn_q_lpt = (v >> 7) & 1;
    g_attn_dkv_asm = !((v >> 10) & 1);               // bit 10: second-generation dK/dV kernel instead of the asm one
    g_attn_dq_asm = (v >> 12) & 1;                   // bit 12: the asm dQ kernel instead of the second-generation one (see above)
    g_attn64_dq_asm_order = (v >> 13) & 1;           // bit 13: asm dQ kernel one query block per workgroup (not persistent)
    g_attn64_dkv_asm_order = (v >> 11) & 1;          // bit 11: asm dK/dV kernel walks an XCD's (batch, kv head) pairs one after the other
    g_attn_grp = !((v >> 14) & 1);                   // bit 14: head_dim 128, S <= 32: block-per-head mapping instead of the grouped one
    return 0;
}

extern "C" int csm_attn_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int KV, int HD,
                            hipStream_t stream) {
    if (int e = check_attn("csm_attn_fwd", B, S, H, KV, HD)) return e;
    CSM_REQUIRE(qkv && out && lse, "csm_attn_fwd: null pointer");
    const float scale = 1.f / sqrtf((float)HD);
    if (HD == 64 && (g_attn_gen2 & 1)) {
        csm_attn64_fwd_launch(qkv, out, lse, B, S, H, KV, stream);
    } else if (HD == 64) {
        if (S > 64 && g_attn_qt_fwd == 2) launch_q<64, false, 2>(qkv, out, lse, nullptr, nullptr, nullptr, B, S, H, KV, scale, stream);
        else launch_q<64, false, 1>(qkv, out, lse, nullptr, nullptr, nullptr, B, S, H, KV, scale, stream);
    } else if (g_attn_grp && S <= 32 && H == 4 * KV) {
        launch_q<128, false, 2, true>(qkv, out, lse, nullptr, nullptr, nullptr, B, S, H, KV, scale, stream);
    } else {
        launch_q<128, false, 1>(qkv, out, lse, nullptr, nullptr, nullptr, B, S, H, KV, scale, stream);
    }
    CSM_CHECK_LAUNCH("csm_attn_fwd");
    return 0;
}

static int attn_bwd_impl(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta_ws,
                         const float* rope, int B, int S, int H, int KV, int HD, hipStream_t stream) {
    if (int e = check_attn("csm_attn_bwd", B, S, H, KV, HD)) return e;
    g_attn_last_dkv = g_attn_last_dq = 0;
    CSM_REQUIRE(qkv && out && dout && lse && dqkv && delta_ws, "csm_attn_bwd: null pointer");
    const float scale = 1.f / sqrtf((float)HD);
    // dQ first: it also writes delta = rowsum(dO * O), which the dK/dV kernel reads
    void* o = const_cast<void*>(out);
    float* l = const_cast<float*>(lse);
    if (HD == 64) {
        if (g_attn_gen2 & 2) {
            g_attn_last_dq = g_attn_dq_asm && csm_attn64_dq_asm_launch(qkv, out, dout, lse, delta_ws, dqkv, B, S, H, KV, rope, stream);
            if (!g_attn_last_dq) csm_attn64_dq_launch(qkv, out, dout, lse, delta_ws, dqkv, B, S, H, KV, rope, stream);
        } else if (S > 64 && g_attn_qt_bwd == 2) launch_q<64, true, 2>(qkv, o, l, dout, delta_ws, dqkv, B, S, H, KV, scale, stream, rope);
        else launch_q<64, true, 1>(qkv, o, l, dout, delta_ws, dqkv, B, S, H, KV, scale, stream, rope);
        CSM_CHECK_LAUNCH("csm_attn_bwd(dq)");
        if (g_attn_gen2 & 2) {
            // third generation (attention64_asm.hip: one wave per 64 keys and query head, generated asm loop) where the shape allows
            g_attn_last_dkv = g_attn_dkv_asm && csm_attn64_dkv_asm_launch(qkv, dout, delta_ws, dqkv, B, S, H, KV, rope, stream);
            if (!g_attn_last_dkv) csm_attn64_dkv_launch(qkv, dout, lse, delta_ws, dqkv, B, S, H, KV, rope, stream);
        } else if (g_attn_dkv_kt1) launch_dkv<64, 1>(qkv, dout, lse, delta_ws, dqkv, B, S, H, KV, scale, stream, rope);
        else launch_dkv<64, 2>(qkv, dout, lse, delta_ws, dqkv, B, S, H, KV, scale, stream, rope);
    } else {
        if (g_attn_grp && S <= 32 && H == 4 * KV) launch_q<128, true, 2, true>(qkv, o, l, dout, delta_ws, dqkv, B, S, H, KV, scale, stream, rope);
        else launch_q<128, true, 1>(qkv, o, l, dout, delta_ws, dqkv, B, S, H, KV, scale, stream, rope);
        CSM_CHECK_LAUNCH("csm_attn_bwd(dq)");
        launch_dkv<128, 1>(qkv, dout, lse, delta_ws, dqkv, B, S, H, KV, scale, stream, rope);
    }
    CSM_CHECK_LAUNCH("csm_attn_bwd(dkv)");
    return 0;
}

extern "C" int csm_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                            float* delta_ws, int B, int S, int H, int KV, int HD, hipStream_t stream) {
    return attn_bwd_impl(qkv, out, dout, lse, dqkv, delta_ws, nullptr, B, S, H, KV, HD, stream);
}

extern "C" int csm_attn_bwd_rope(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                 float* delta_ws, const float* rope_table, int B, int S, int H, int KV, int HD,
                                 hipStream_t stream) {
    CSM_REQUIRE(rope_table, "csm_attn_bwd_rope: null table");
    return attn_bwd_impl(qkv, out, dout, lse, dqkv, delta_ws, rope_table, B, S, H, KV, HD, stream);
}
