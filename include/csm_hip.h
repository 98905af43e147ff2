/* libcsm_hip.so - C ABI of the MI355X (gfx950) hot path for CSM training / generation.
 *
 * The reference (imaginateit/csm-train-pytorch) is pure Python and has no FFI: its boundary for this path is
 * the Python API of csm.models.model / csm.training.utils / csm.training.trainer / csm.training.lora_trainer /
 * csm.generator.  The host-side mirror of that API lives in csm-train-pytorch_amd/csm/ and calls ONLY the entry
 * points below (through ctypes).  Each entry names the reference arithmetic it replaces (paths relative to the
 * reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (torch-allocated); the library owns nothing;
 *   - bf16 tensors are passed as void* (raw uint16 storage), row-major, 16-byte aligned;
 *   - all work is enqueued on the caller's hipStream_t; nothing synchronises, allocates or frees (graph-capturable);
 *   - return value 0 = ok, non-zero = error (1 bad argument, 2 launch failure, 3 no device, 4 wrong arch);
 *     csm_last_error() returns the text for the calling thread.  No exception crosses the boundary.
 */
#ifndef CSM_HIP_H
#define CSM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* csm_stream_t; /* == hipStream_t */

int csm_abi_version(void);
const char* csm_last_error(void);
int csm_device_check(int device);

/* ---- K3/K6/K7/K8/K9/K10/K11: every dense contraction -------------------------------------------------------
 * C[M,N] = alpha * opA(A) . opB(B)^T (+ R), bf16 in, fp32 accumulate, bf16 or fp32 out, optional batch.
 *   transA=0: A is [M][K] (lda)      transA=1: A is [K][M] (lda)
 *   transB=0: B is [N][K] (ldb)      transB=1: B is [K][N] (ldb)
 * Replaces torchtune nn.Linear q/k/v/output_proj, w1/w2/w3 (model.py:13-42), Model.projection, codebook0_head
 * and torch.mm(decoder_h, audio_head[i-1]) (src/csm/models/model.py:124-126,172,184,187), their autograd
 * backward products, and the LoRA A/B products of src/csm/mlx/components/lora.py:71-105.
 * R (bf16, ldr) may alias C (accumulate).  lda/ldb multiples of 8; the contiguous dimension of A and of B a
 * multiple of 8. */
int csm_gemm_bf16(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb, int ldc,
                  int ldr, int transA, int transB, int out_f32, float alpha, int batch, long long strideA,
                  long long strideB, long long strideC, long long strideR, csm_stream_t stream);

/* Same product with a fused epilogue (bf16 output only):
 *   epilogue 1, SwiGLU forward : C = [M][N] with gate/up interleaved along N (g0,u0,g1,u1,...) AND aux_out[M][N/2] =
 *                                silu(gate)*up  - the w1/w3 projection of torchtune FeedForward in one pass;
 *   epilogue 2, SwiGLU backward: the GEMM result is d(act) [M][N] (never stored); aux_in = gate/up [M][2N]; C receives
 *                                d(gate),d(up) interleaved [M][2N] (ldc >= 2N) - the w2 dgrad fused with the activation backward. */
int csm_gemm_bf16_ex(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb, int ldc,
                     int ldr, int transA, int transB, int out_f32, float alpha, int batch, long long strideA,
                     long long strideB, long long strideC, long long strideR, int epilogue, const void* aux_in,
                     void* aux_out, int ld_aux, csm_stream_t stream);
/* Fused q|k|v projection + RoPE forward: C[M][N] = A[M][K] W[N][K]^T, interleaved pairs of columns [0, n_rope_cols) (the q and
 * k heads, head_dim features each) rotated by position (row % rows_per_seq) with the fp32 (cos, sin) table
 * [P][head_dim/2][2].  Replaces torchtune q_proj / k_proj / v_proj + Llama3ScaledRoPE as configured at reference
 * src/csm/models/model.py:13-25,30-42 (rope_base 500000, scale_factor 32) for positions arange(S)
 * (src/csm/training/utils.py:81-82). */
int csm_gemm_bf16_rope(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                       const float* rope_table, int rows_per_seq, int n_rope_cols, int head_dim, csm_stream_t stream);
/* A frozen projection and its LoRA adapters as ONE product: C[M][N] = A . B (layouts by transA / transB as in csm_gemm_bf16)
 * + xA[M][kx] . xB[N][kx]^T (+ R), the extra kx / 32 k-steps taken after the main loop, in fp32, in the same accumulators,
 * before the epilogue (0 none; 1 / 2 SwiGLU forward / backward: aux_out / aux_in, ld_aux as in csm_gemm_bf16_ex; 3 RoPE: aux_in = table, ld_aux =
 * rows per sequence, rope_cols / head_dim as in csm_gemm_bf16_rope).  xA / xB: bf16, row-major, leading dimension kx (a
 * multiple of 32, <= 256; ranks padded with zeros), 16-byte aligned.  Replaces the two extra matmuls and the add of
 * LoRALinear.__call__ (reference src/csm/mlx/components/lora.py:85-105: y = x W0^T + scale (x A^T) B^T, with xA = scale x A^T
 * and xB = B) and of its input gradient (dx = dy W0 + (scale dy B) A: xA = scale dy B, xB = A^T). */
int csm_gemm_bf16_kext(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb, int ldc, int ldr,
                       int transA, int transB, const void* xA, const void* xB, int kx, int epilogue, const void* aux_in,
                       void* aux_out, int ld_aux, int rope_cols, int head_dim, csm_stream_t stream);
/* out[M][N] = alpha * X[M][K] Wt[N][K]^T for N = 32 or 64 and K % 128 == 0: the skinny products of a LoRA group (x A^T and dy B of
 * reference src/csm/mlx/components/lora.py:85-105 with the group's ranks side by side), read-once bandwidth kernel. */
int csm_skinny_nt_bf16(const void* X, const void* Wt, void* out, int M, int N, int K, int ldx, int ldw, int ldo, float alpha,
                       csm_stream_t stream);
/* The two backward products of a Linear layer in one launch, their tiles interleaved over the chip:
 *   dX[M][Kin] = dY[M][Nout] W[Nout][Kin]   (dx_epilogue 0), or the SwiGLU backward of that product (dx_epilogue 2: W = w2,
 *   aux_in = gate/up [M][2 Kin], dX = d(gate/up) [M][2 Kin]);   dW[Nout][Kin] (+)= alpha_w * dY^T X[M][Kin].
 * Replaces autograd's two matmuls per nn.Linear in `loss.backward()` (reference src/csm/training/trainer.py:261-263). */
int csm_gemm_bf16_dgrad_wgrad(const void* dY, const void* W, void* dX, const void* X, void* dW, int M, int Nout, int Kin, int ld_dy,
                              int ldw, int ld_dx, int ldx, int ld_dw, int dx_epilogue, const void* aux_in, int ld_aux,
                              int accumulate, float alpha_w, csm_stream_t stream);
/* Two weight gradients sharing the token dimension, dW_i[N_i][K_i] (+)= alpha * dY_i[M][N_i]^T X_i[M][K_i], in one launch (for
 * outputs too small to fill the chip alone: attention output projection + fused q|k|v projection of one layer). */
int csm_gemm_bf16_two_wgrad(const void* dY1, const void* X1, void* dW1, int N1, int K1, int ld_dy1, int ldx1, int ld_dw1,
                            const void* dY2, const void* X2, void* dW2, int N2, int K2, int ld_dy2, int ldx2, int ld_dw2,
                            int M, int accumulate, float alpha, csm_stream_t stream);
/* n (1..12) such weight gradients in one launch, arrays of n entries each.  The attention projections' gradients of one layer
 * are 160 tiles of 256 x 256 - 0.63 of a round of the 256 CUs; the engine defers them and launches three layers' worth (480
 * tiles, 1.9 rounds) together.  Tile arithmetic as csm_gemm_bf16_two_wgrad: results do not depend on the grouping.
 * (autograd's per-Linear weight-gradient matmuls, reference src/csm/training/trainer.py:261-263) */
int csm_gemm_bf16_multi_wgrad(int n, const void* const* dY, const void* const* X, void* const* dW, const int* N, const int* K,
                              const int* ld_dy, const int* ldx, const int* ld_dw, int M, int accumulate, float alpha,
                              csm_stream_t stream);


/* tuning switch (A/B benchmarking): 0 register staging 128x128; 1 LDS-DMA 128x128; 2 auto = the 256x256 pipelined kernel
 * where its tiles fill the chip, else 128x128 (default); 3 force the 256x256 kernel; 4 force the four-wave 256x256 kernel
 * with the hand-scheduled K loop where it applies (batch 1, no K-extension), else as 3 */
int csm_set_gemm_variant(int v);
/* tuning switch: 1 (default) the 256x256 kernel runs one persistent workgroup per CU over its tile list (the next tile's first
 * loads are requested before the finished tile is stored); 0 one tile per workgroup */
int csm_set_gemm256_persistent(int v);
int csm_get_gemm256_persistent(void);
/* tuning switches for A/B runs: key 0 = the eight-wave 256x256 kernel touches the tile of a fused epilogue's read operand
 * (gate/up of the SwiGLU backward, a bf16 residual) during its K loop so that the epilogue's loads hit cache (default 1);
 * key 1 = the auto variant hands batch-1 products without K-extension to the four-wave kernel (default 1);
 * key 8 (round 4) = the four-wave kernel uses 256 x 192 output tiles where they fill the rounds of the 256 CUs better (default 1) */
int csm_set_gemm_tuning(int key, int value);
/* name of the kernel (rocprofv3 spelling, without the argument list) the most recent csm_gemm_* call on this host thread's
 * library instance launched - for benchmarks that attribute time to kernels; not thread-safe */
const char* csm_gemm_last_kernel(void);

/* ---- K2: torchtune RMSNorm (sa_norm / mlp_norm / norm; eps=1e-5 at model.py:22,39) -------------------------- */
int csm_rmsnorm_fwd(const void* x, const void* scale, void* y, float* rstd, int M, int D, float eps, csm_stream_t stream);
int csm_rmsnorm_bwd_blocks(void);
int csm_rmsnorm_bwd(const void* x, const void* scale, const float* rstd, const void* dy, const void* dres, void* dx,
                    float* dscale_partials /* [csm_rmsnorm_bwd_blocks()][D] or NULL */, int M, int D, csm_stream_t stream);
int csm_colsum_bf16(const float* partials, int rows, int D, void* dst, int accumulate, csm_stream_t stream);
/* the same reduction for n (1..8) pairs of one shape in one launch (the RMSNorm scale gradients of the layers whose small weight
 * gradients the engine launches together); per column the arithmetic of csm_colsum_bf16 */
int csm_colsum_bf16_multi(int n, const float* const* partials, void* const* dst, int rows, int D, int accumulate, csm_stream_t stream);

/* ---- LoRA side ops (reference mlx/components/lora.py:85-102): inverted dropout of the adapter input (mask is a pure
 * function of (seed, row*D+col), so the backward regenerates it; accumulate=1: out += dropout(in)), the optional
 * lora_bias add, and the bf16 column sums that give d(lora_bias) (finish with csm_colsum_bf16 on the partials). */
int csm_dropout_bf16(const void* in, int ld_in, void* out, int ld_out, long long M, int D, float p, unsigned long long seed,
                     int accumulate, csm_stream_t stream);
int csm_bias_add_bf16(void* y, int ld, const void* bias, long long M, int D, csm_stream_t stream);
int csm_colsum_rows_bf16(const void* x, int ld, long long M, int D, float* partials /* [slices][D] */, int slices,
                         csm_stream_t stream);

/* ---- K4: torchtune Llama3ScaledRoPE (rope_base=500000, scale_factor=32: model.py:23-24,40-41), interleaved
 * pairs, in place on the q and k heads of the fused qkv buffer; inverse=1 is the backward rotation.
 * table = [P][head_dim/2][2] (cos,sin) fp32; pos = int32 [M] or NULL (position = row % S). */
int csm_rope(void* qkv, const float* table, const int* pos, long long M, int S, int n_heads_qk, int head_dim, int ld,
             int inverse, csm_stream_t stream);

/* ---- K5: causal GQA attention replacing torchtune MultiHeadAttention -> F.scaled_dot_product_attention with the
 * mask of model.py:59-76 / training/utils.py:90-91.  qkv [B*S][(H+2KV)*HD]; out [B*S][H*HD]; lse [B][H][S]. */
int csm_set_attn_variant(int v); /* scheduling experiments: 0 = defaults; else bit fields (query tiles per wave, dK/dV key
                                  * tile and work order, heaviest-first order of the forward / dQ grid) - see attention.hip */
int csm_attn_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int KV, int HD, csm_stream_t stream);
/* scratch of csm_attn_bwd / csm_attn_bwd_rope in BYTES (since ABI 2: 2 x [B][H][S] floats - the dQ pass leaves -delta and
 * -lse*log2(e) per query there for the dK/dV pass; ABI 1 needed half).  Allocate at least this much for delta_ws. */
long long csm_attn_bwd_workspace_bytes(int B, int S, int H);
int csm_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                 float* delta_ws /* scratch, csm_attn_bwd_workspace_bytes(B, S, H) */, int B, int S, int H, int KV, int HD, csm_stream_t stream);
/* the same with the backward of csm_rope fused into the dQ / dK epilogues (table as for csm_rope, position = row index
 * inside the sequence): dqkv comes out as the gradient of the UN-rotated projection output, no separate inverse pass. */
int csm_attn_bwd_rope(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta_ws,
                      const float* rope_table, int B, int S, int H, int KV, int HD, csm_stream_t stream);
/* which kernels the most recent csm_attn_bwd* call on this library instance launched (for tests and benchmarks; not
 * thread-safe; since ABI 3): bit 0 = the dK/dV pass, bit 1 = the dQ pass ran the generated-asm kernel of attention64_asm.hip
 * (head_dim 64, S % 64 == 0, 4 query heads per kv head); a clear bit = the compiler-scheduled kernel of attention64.hip /
 * attention.hip */
int csm_attn_last_dkv_kernel(void);
/* tools/probes only: a device buffer (>= grid x 4 waves x 8 rounds x 8 u64) that the asm dQ kernel fills with cycle stamps, or NULL */
int csm_attn64_set_debug(void* device_buffer);

/* ---- K7: SwiGLU of torchtune FeedForward: out = silu(gate) * up, gu = gate/up INTERLEAVED ([M][2F]: g0,u0,g1,u1,..) - */
int csm_swiglu_fwd(const void* gu, void* out, long long M, int F, csm_stream_t stream);
int csm_swiglu_bwd(const void* gu, const void* dout, void* dgu, long long M, int F, csm_stream_t stream);

/* ---- K1: Model._embed_tokens + mask-mul-sum (model.py:202-217, training/utils.py:85-87) ---------------------- *
 * tokens int64 [M][K+1] (K audio slots then the text slot), mask uint8 [M][K+1]; out bf16 [M][D].
 * The backward is csm_embed_bwd_sorted below. */
int csm_embed_fwd(const long long* tokens, const uint8_t* mask, const void* text_emb, const void* audio_emb, void* out,
                  long long M, int K, int D, int audio_vocab, csm_stream_t stream);

/* Deterministic, scratch-free form of the embedding backward: occurrences (embedding row, source row) sorted by
 * embedding row; text rows are [0, text_rows), audio rows follow; rows >= n_rows are padding.  Source index < M reads
 * dh[index], otherwise dseq[index - M].  Sums in fp32 registers, adds into the bf16 gradient tables once per row. */
int csm_embed_bwd_sorted(const long long* sorted_rows, const long long* src_index, long long n_occ, const void* dh,
                         const void* dseq, long long M, void* g_text, void* g_audio, long long text_rows, long long n_rows,
                         int D, csm_stream_t stream);
/* dst[rows[n]] += src[n * src_stride_rows] for unique rows (scatter of the decoder's position-0 gradient); rows[n] < 0 is
 * padding and is skipped */
int csm_rows_add_bf16(void* dst, const int* rows, const void* src, long long N, int src_stride_rows, int D, csm_stream_t stream);
/* out[n] = table[rows[n]], table[rows[n]] = 0 for unique rows (rows[n] < 0: out[n] = 0).  New capability (the reference has
 * no distributed code): the fixed-capacity exchange of touched text-embedding gradient rows between
 * data-parallel ranks (csm/training/dp.py), which replaces a 525 MB all-reduce of reference model.py:119's table. */
int csm_rows_take_bf16(void* table, const int* rows, void* out, long long N, int D, csm_stream_t stream);

/* depth-decoder teacher forcing (model.py:175-189): out[n][0]=hidden[rows[n]], out[n][i]=audio_emb[code_{i-1}+(i-1)V] */
int csm_decoder_input_fwd(const void* hidden, const int* rows, const long long* codes, const void* audio_emb, void* out,
                          long long N, int K, int D, int audio_vocab, csm_stream_t stream);

/* ---- K9/K11: F.cross_entropy (training/utils.py:102-105) fused with its backward ------------------------------ *
 * logits fp32 [R][ldl]; targets int64 [R] (<0 = row excluded); loss_rows fp32 [R]; dlogits bf16 [R][ldd] or NULL. */
int csm_ce_fwd_bwd(const float* logits, const long long* targets, float* loss_rows, void* dlogits, long long R, int V,
                   int ldl, int ldd, float grad_scale, csm_stream_t stream);
int csm_reduce_sum_f32(const float* x, long long n, float scale, float* out, csm_stream_t stream);

/* ---- K12/K13: clip_grad_norm_ + AdamW (training/trainer.py:166-173,271-277) ------------------------------------ */
int csm_sumsq_blocks(void);
int csm_sumsq_bf16(const void* g, long long n, float* partials /* [csm_sumsq_blocks()] */, csm_stream_t stream);
int csm_clip_coef(const float* partials, int n_partials, float max_norm, float* norm_and_coef /* [2] */, csm_stream_t stream);
/* norm_and_coef[1] < 0 on the device = skip this step: parameters and moments are left untouched, only zero_grad is honoured
 * (how a data-parallel step with incomplete gradients is dropped on every rank without a host sync). */
int csm_adamw_step(float* master, float* m, float* v, void* param, void* grad, long long n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, const float* norm_and_coef /* or NULL */,
                   float grad_mul, int zero_grad /* clear grad in the same pass */, csm_stream_t stream);
/* the same update with the fp32 master held as two 16-bit halves: `param` (bf16 working copy = upper half, rounded half-up)
 * and `master_lo` (lower 16 bits); master = ((hi - (lo >> 15)) << 16) | lo exactly.  26 instead of 28 bytes per parameter. */
int csm_adamw_step_split(void* master_lo, float* m, float* v, void* param, void* grad, long long n, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int step, const float* norm_and_coef, float grad_mul,
                         int zero_grad, csm_stream_t stream);

int csm_set_adamw_blocks(int blocks); /* tuning switch */

/* ---- K14: sample_topk + _multinomial_sample_one_no_sync (model.py:79-96), Exp(1) noise q supplied ---------------- */
int csm_sample_topk(const float* logits, const float* q, int* out, int rows, int V, int ldl, int topk, float temperature,
                    csm_stream_t stream);

/* ---- K15: batch-1 decode of Model.generate_frame (model.py:161-195) against KV caches ------------------------------ *
 * y[b][n] = sum_k x[b][k] W[n][k] (+ residual[b][n]), B <= 4 (weight-streaming matrix-vector product) */
/* tuning switches for A/B runs (since ABI 3): key 0 = one-row products keep x in registers (no LDS copy, no barrier; default 1),
 * key 1 = the 2048-wide stack's weights are loaded non-temporally in decode (default 1),
 * key 2 = gate/up pairs per wave in the depth decoder's w13 product (1, 2 or 4; default 1: measured equal or slower above),
 * key 3 = two to four batch rows: K = 1024 / 2048 products keep every row of x in registers (default 1; 0 = the LDS kernel) */
int csm_set_decode_tuning(int key, int value);
int csm_gemv_bf16(const void* x, const void* W, void* y, const void* residual, int B, int N, int K, int ldw, int ldx, int ldy,
                  int out_f32, csm_stream_t stream);
/* csm_gemv_bf16 with the neighbouring element-wise steps of a decode layer fused in: norm_scale != NULL applies torchtune
 * RMSNorm (eps) to x first; swiglu = 1 reads W as interleaved gate/up rows (N = 2F) and writes y[b][F] = silu(g) * u;
 * row_index != NULL takes batch row b of x from row (row_index[b] + row_offset) of the table x (embedding lookup of a
 * sampled code, model.py:189-191). */
int csm_gemv_bf16_ex(const void* x, const void* W, void* y, const void* residual, int B, int N, int K, int ldw, int ldx, int ldy,
                     int out_f32, const void* norm_scale, float eps, int swiglu, const int* row_index, int row_offset,
                     csm_stream_t stream);
/* y[b][n] = sum_k x[b][k] W[k][n]  (K-major weights: audio_head[i] = [decoder_dim][vocab]) */
int csm_gemv_t_bf16(const void* x, const void* W, void* y, int B, int N, int K, int ldw, int ldx, int ldy, int out_f32,
                    csm_stream_t stream);
/* caches are [B][KV][S_max][HD] bf16; pos = int32 [B] on the device (graph-replayable): the new key/value row index */
/* a depth-decoder layer's rotate + cache append + attention (<= 64 cached positions, head_dim 128) + output projection
 * (+ residual) as ONE launch: y[B][N] = attention(qkv row, caches) . W[N][H*HD]^T + residual; bit-identical to
 * csm_attn_decode_rope followed by csm_gemv_bf16 (model.py:181-187 runs 31 such steps per frame). */
int csm_gemv_attn_bf16(const void* qkv, void* kcache, void* vcache, const int* pos, const float* rope_table, const void* W, void* y,
                       const void* residual, int B, int N, int H, int KV, int HD, int S_max, int ld_qkv, int ldw, int ldy,
                       csm_stream_t stream);
/* csm_attn_decode_rope for rows that share a position the host knows (round 4: the depth decoder's step i of a batch of utterances):
 * all loads of the prologue at once, coalesced key / value images; HD = 128, H = 4 KV, S_max <= 32; bit-identical to it. */
int csm_attn_decode_rope_at(const void* qkv, void* kcache, void* vcache, void* out, int pos, const float* rope_table, int B, int H,
                            int KV, int HD, int S_max, int ld, csm_stream_t stream);
/* the same for one utterance (B = 1) whose position the host knows (round 4: the depth decoder's step i is always at position i,
 * so a captured frame graph carries it as an argument): every address of the prologue is known at launch and all its loads
 * go out at once.  S_max <= 32, H * HD = 1024; bit-identical to csm_gemv_attn_bf16. */
int csm_gemv_attn_at_bf16(const void* qkv, void* kcache, void* vcache, int pos, const float* rope_table, const void* W, void* y,
                          const void* residual, int N, int H, int KV, int HD, int S_max, int ldw, csm_stream_t stream);
int csm_kv_append(const void* qkv, void* kcache, void* vcache, const int* pos, int B, int H, int KV, int HD, int S_max, int ld,
                  csm_stream_t stream);
/* out[b][h*HD..] = softmax(q . K[0..pos[b]]^T / sqrt(HD)) V   for the single query row in qkv[b] */
int csm_attn_decode(const void* qkv, const void* kcache, const void* vcache, void* out, const int* pos, int B, int H, int KV,
                    int HD, int S_max, int ld, csm_stream_t stream);
/* the same with what precedes it in a decode step fused in: q and the new k are rotated inside (table as for csm_rope,
 * position = pos[b]) and the new k / v are appended to the caches at pos[b] by the kernel itself - one launch instead of
 * csm_rope + csm_kv_append + csm_attn_decode. qkv is the UNROTATED fused projection row. */
int csm_attn_decode_rope(const void* qkv, void* kcache, void* vcache, void* out, const int* pos, const float* rope_table, int B,
                         int H, int KV, int HD, int S_max, int ld, csm_stream_t stream);

/* ---- K16 (RVQ part): Mimi split residual VQ behind generator.py:117,209 ------------------------------------------ */
int csm_rvq_encode(const float* x, const float* codebooks, long long* codes, int T, int K, int C, int D, int n_semantic,
                   csm_stream_t stream);
int csm_rvq_decode(const long long* codes, const float* codebooks, float* out, int T, int K, int C, int D,
                   csm_stream_t stream);

/* ---- K16 (rest): Mimi codec around the RVQ - SEANet convolutions and the two 8-layer codec transformers, fp32 -------- *
 * (moshi MimiModel.encode / decode behind generator.py:67-70,117,209).  Activations are [C][T] for the convolutions and
 * [T][D] for the transformers. */
int csm_conv1d_f32(const float* x, const float* w, const float* bias, const float* residual, float* y, int C_in, int C_out,
                   int T_in, int T_out, int k, int stride, int dilation, int pad_left, int pad_mode /*0 zero,1 replicate*/,
                   int groups, int elu_in, csm_stream_t stream);
int csm_conv_transpose1d_f32(const float* x, const float* w, const float* bias, float* y, int C_in, int C_out, int T_in, int T_out,
                             int k, int stride, int crop_left, int groups, int elu_in, csm_stream_t stream);
int csm_layernorm_f32(const float* x, const float* w, const float* b, float* y, int T, int D, float eps, csm_stream_t stream);
/* y = act(x W^T); scale != NULL: y = residual + scale[n] * y (layer scale); else y += residual when given; act 1 = GELU */
int csm_linear_f32(const float* x, const float* W, const float* scale, const float* residual, float* y, int T, int N, int K,
                   int ldx, int act, csm_stream_t stream);
int csm_rope_half_f32(float* qkv, int T, int H, int head_dim, float base, int pos0, csm_stream_t stream);
int csm_attn_window_f32(const float* qkv, float* out, int T, int H, int head_dim, int window, csm_stream_t stream);
int csm_transpose_f32(const float* in, float* out, int R, int C, csm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CSM_HIP_H */
