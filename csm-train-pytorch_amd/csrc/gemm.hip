// bf16 MFMA GEMM for gfx950:  C[M,N] = alpha * opA(A) . opB(B)^T (+ R)
//
//   TA == 0 : A is [M][K] row-major (K contiguous, "kc")      TA == 1 : A is [K][M] row-major ("ks")
//   TB == 0 : B is [N][K] row-major (torch Linear weight)     TB == 1 : B is [K][N] row-major
//
// so one kernel family covers the three products of a Linear layer without materialised transposes:
//   forward  Y  = X  W^T      (TA=0,TB=0)      A=X[M,K]    B=W[N,K]
//   dgrad    dX = dY W        (TA=0,TB=1)      A=dY[M,N']  B=W[N',K'] read as [K=N'][N=K']
//   wgrad    dW = dY^T X      (TA=1,TB=1)      A=dY[M',N'] read as [K=M'][M=N'],  B=X[M',K'] as [K=M'][N=K']
//
// Replaces the torchtune nn.Linear calls configured at reference src/csm/models/model.py:13-42 and
// the projection / heads at src/csm/models/model.py:124-126,172,184,187.
//
// Tile: 128x128x64 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave, 16 accumulators of
// mfma_f32_16x16x32_bf16).  LDS: 2 x (16 KiB A + 16 KiB B) double buffer.  Global->register->LDS
// staging one tile ahead; one barrier per K-tile.  LDS images are XOR-swizzled so that both the
// ds_read_b128 row fragments (kc) and the ds_read_b64_tr_b16 transposed fragments (ks) are
// bank-conflict free:
//   kc image [128 rows][64 k]   : 16-B chunk c of row r lives at chunk  c ^ (r & 7)
//   ks image [64 k][128 cols]   : 32-B slot  s of k-row r lives at slot s ^ ks_swz(r)
// The MFMA is issued with the operands swapped (D = Btile . Atile^T) so every lane ends up with
// four CONSECUTIVE output columns of one row: 8-byte bf16 / 16-byte fp32 stores, no LDS epilogue.
// Workgroup ids are remapped so that each XCD (private 4 MiB L2) walks a contiguous, GROUP_M-rastered
// run of tiles.
#include "gemm_common.h"

const char* g_last_gemm_kernel = "";   // see csm_gemm_last_kernel below (shared with gemm256.hip)

namespace {

template <int TA, int TB, typename OutT, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- XCD-aware, GROUP_M-rastered tile id -------------------------------------------------
    const int nwg = g.tiles_m * g.tiles_n;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, within = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * g.tiles_n;
    const int grp = id / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(g.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (id % per_group) % gsz;
    const int tn = (id % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int bz = blockIdx.z;
    const bf16_t* A = g.A + (size_t)bz * g.sA;
    const bf16_t* B = g.B + (size_t)bz * g.sB;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto ldsA = [&](int b) { return smem + b * 2 * TILE_BYTES; };
    auto ldsB = [&](int b) { return smem + b * 2 * TILE_BYTES + TILE_BYTES; };

    const int nt = (g.K + BK - 1) / BK;
    auto compute = [&](int cur) {
        Frags<TA, 4> fa;
        Frags<TB, 4> fb;
        fa.load(ldsA(cur), wm0, lane);
        fb.load(ldsB(cur), wn0, lane);
        if constexpr (TA != 0 || TB != 0) frag_wait();   // asm tr-reads are invisible to hipcc's lgkmcnt bookkeeping
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb.get(j, ks), fa.get(i, ks), acc[i][j], 0, 0, 0);
    };
    if constexpr (GLDS) {
        GldsSrc sa, sb;
        glds_prepare<TA>(A, g.lda, g.M, m0, sa);
        glds_prepare<TB>(B, g.ldb, g.N, n0, sb);
        const char *pa = sa.base, *pb = sb.base;          // running pointers: K-tile t + 1 inside the loop
        stage_glds_pre(sa, pa, ldsA(0));
        stage_glds_pre(sb, pb, ldsB(0));
        __syncthreads();   // waits vmcnt(0) for the LDS-DMA, then the barrier
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            pa += sa.step; pb += sb.step;
            if (t + 1 < nt) {
                stage_glds_pre(sa, pa, ldsA(cur ^ 1));
                stage_glds_pre(sb, pb, ldsB(cur ^ 1));
            }
            compute(cur);
            __syncthreads();
        }
    } else {
        U4 ra[4], rb[4];
        stage_load<TA>(A, g.lda, g.M, g.K, m0, 0, ra);
        stage_load<TB>(B, g.ldb, g.N, g.K, n0, 0, rb);
        stage_store<TA>(ldsA(0), ra);
        stage_store<TB>(ldsB(0), rb);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt) {
                stage_load<TA>(A, g.lda, g.M, g.K, m0, (t + 1) * BK, ra);
                stage_load<TB>(B, g.ldb, g.N, g.K, n0, (t + 1) * BK, rb);
            }
            compute(cur);
            if (t + 1 < nt) {
                stage_store<TA>(ldsA(cur ^ 1), ra);
                stage_store<TB>(ldsB(cur ^ 1), rb);
            }
            __syncthreads();
        }
    }

    if (g.kx) k_extend<4>(g.xA, g.xB, g.kx, g.M, g.N, m0 + wm0, n0 + wn0, lane, acc);

    // ---- epilogue: lane owns C[m = ..+(lane&15)][n = ..+4*(lane>>4) .. +3] per 16x16 tile ----------
    Epi e;
    e.C = reinterpret_cast<OutT*>(g.C) + (size_t)bz * g.sC;
    e.R = g.R ? g.R + (size_t)bz * g.sR : nullptr;
    e.ldc = g.ldc; e.ldr = g.ldr; e.alpha = g.alpha; e.mode = g.epi_mode; e.aux_in = g.aux_in; e.aux_out = g.aux_out;
    e.ld_aux = g.ld_aux; e.M = g.M; e.N = g.N; e.p0 = g.epi_p0; e.p1 = g.epi_p1;
    const bool vec_ok = ((g.ldc & 3) == 0) && (!e.R || (g.ldr & 3) == 0);
    (void)epi_block<OutT, 4>(e, vec_ok, m0 + wm0, n0 + wn0, lane, acc);
}

// ---------------------------------------------------------------------------------------------------------------
// Skinny product out[M][N] = alpha * X[M][K] . Wt[N][K]^T for N = 32 or 64 (a LoRA group's ranks): a pure bandwidth problem -
// X is read once (M K 2 bytes), Wt (N K 2 bytes, tens of KB) stays in L2 - that a 128x128 GEMM tile serves badly (M / 128
// workgroups, each walking K serially, 3/4 of every MFMA wasted).  Here a workgroup owns 16 rows and its four waves a quarter
// of K each (M / 16 x 4 waves = 16 waves per CU at M = 16k).  X is fetched with fully coalesced instructions - two rows x 512
// contiguous bytes per instruction - and turned into MFMA fragments through a wave-private LDS staging area (an MFMA fragment
// read straight from global memory puts 16 rows, 4 KB apart, behind every instruction: measured 2.5 TB/s); the Wt
// fragments, L2 residents, are read directly.  The four partial sums meet in LDS at the end, in a fixed order.
template <int NT /* N / 16 */>
__global__ __launch_bounds__(256) void skinny_nt_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wt, bf16_t* __restrict__ out,
                                                        int M, int K, int ldx, int ldw, int ldo, float alpha) {
    constexpr int RS = 528;                                 // staged row: 512 B of k + 16 B pad (fragment reads hit 64 distinct banks)
    __shared__ __attribute__((aligned(16))) char stage[4][16 * RS];
    const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 16;
    const int c = (lane >> 4) * 8;
    const int kbeg = kq * (K >> 2), kend = kbeg + (K >> 2);
    const bf16_t* wp = Wt + (size_t)(lane & 15) * ldw + c;
    char* st = stage[kq];
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kend; k0 += 256) {             // 256 k-values (8 k-steps of 32) per round; the last may be shorter
        const int kn = min(256, kend - k0);                 // multiple of 32
        U4 xr[8];
        bf16x8 fw[8][NT];
#pragma unroll
        for (int i = 0; i < 8; ++i) {                       // instruction i: rows 2i, 2i+1, 32 lanes x 16 B each
            const int r = min(m0 + 2 * i + (lane >> 5), M - 1);
            const int kk = min((lane & 31) * 8, kn - 8);    // (lanes past a short round re-read its last chunk; never used)
            xr[i] = *reinterpret_cast<const U4*>(X + (size_t)r * ldx + k0 + kk);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + min(32 * u, kn - 32);
#pragma unroll
            for (int j = 0; j < NT; ++j) fw[u][j] = *reinterpret_cast<const bf16x8*>(wp + (size_t)(16 * j) * ldw + k);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<U4*>(st + (2 * i + (lane >> 5)) * RS + (lane & 31) * 16) = xr[i];
        // (wave-private staging: a wave's own LDS operations execute in order, no barrier)
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (32 * u < kn) {
                const bf16x8 fx = *reinterpret_cast<const bf16x8*>(st + (lane & 15) * RS + u * 64 + (lane >> 4) * 16);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[u][j], fx, acc[j], 0, 0, 0);
            }
    }
    // lane holds out[m0 + (lane & 15)][16 j + 4 (lane >> 4) .. + 3]; add the four K quarters through LDS (fixed order)
    __syncthreads();                                        // every wave is done with its staging area: reuse it
    float* red = reinterpret_cast<float*>(&stage[0][0]);    // [3][NT][64][4]
    if (kq > 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((kq - 1) * NT + j) * 64 + lane) * 4 + r] = acc[j][r];
    }
    __syncthreads();
    if (kq == 0 && m0 + (lane & 15) < M) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                v[r] = (((acc[j][r] + red[((0 * NT + j) * 64 + lane) * 4 + r]) + red[((1 * NT + j) * 64 + lane) * 4 + r]) +
                        red[((2 * NT + j) * 64 + lane) * 4 + r]) * alpha;
            uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(out + (size_t)(m0 + (lane & 15)) * ldo + 16 * j + 4 * (lane >> 4)) = o;
        }
    }
}

// symbol (as rocprofv3 prints it, argument list dropped) of the kernel the most recent GEMM entry point launched: bench.py
// names the kernels behind its per-kind timings with it (csm_gemm_last_kernel)
#define CSM_KNAME(base, TA, TB, f32, extra) \
    ((f32) ? (TA ? (TB ? base "<1, 1, float" extra ">" : base "<1, 0, float" extra ">") : (TB ? base "<0, 1, float" extra ">" : base "<0, 0, float" extra ">")) \
           : (TA ? (TB ? base "<1, 1, unsigned short" extra ">" : base "<1, 0, unsigned short" extra ">") : (TB ? base "<0, 1, unsigned short" extra ">" : base "<0, 0, unsigned short" extra ">")))

int g_gemm_variant = 2;   // 0 register staging, 1 LDS-DMA 128x128, 2 auto (256x256 where it fills the chip), 3 force 256x256

template <int TA, int TB>
int launch(const GemmArgs& g, int out_f32, int batch, hipStream_t stream) {
    dim3 grid(g.tiles_m * g.tiles_n, 1, batch), block(256);
    const size_t lds = 4 * TILE_BYTES;
    const bool glds = g_gemm_variant >= 1 && (g.K % BK) == 0 && g.M >= 8 && g.N >= 8;
    g_last_gemm_kernel = glds ? CSM_KNAME("gemm_kernel", TA, TB, out_f32, ", true") : CSM_KNAME("gemm_kernel", TA, TB, out_f32, ", false");
    if (glds) {
        if (out_f32) hipLaunchKernelGGL((gemm_kernel<TA, TB, float, true>), grid, block, lds, stream, g);
        else hipLaunchKernelGGL((gemm_kernel<TA, TB, bf16_t, true>), grid, block, lds, stream, g);
    } else {
        if (out_f32) hipLaunchKernelGGL((gemm_kernel<TA, TB, float, false>), grid, block, lds, stream, g);
        else hipLaunchKernelGGL((gemm_kernel<TA, TB, bf16_t, false>), grid, block, lds, stream, g);
    }
    CSM_CHECK_LAUNCH("csm_gemm_bf16");
    return 0;
}

}  // namespace

int csm_gemm256_launch(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb, int ldc,
                       int ldr, int transA, int transB, int out_f32, float alpha, int batch, long long sA, long long sB,
                       long long sC, long long sR, int epi_mode, const void* aux_in, void* aux_out, int ld_aux,
                       hipStream_t stream, int epi_p0 = 0, int epi_p1 = 0, const void* xA = nullptr, const void* xB = nullptr, int kx = 0);

int csm_gemm256w4_launch(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb, int ldc, int ldr,
                         int transA, int transB, int out_f32, float alpha, int batch, long long sA, long long sB, long long sC,
                         long long sR, int epi_mode, const void* aux_in, void* aux_out, int ld_aux,
                         hipStream_t stream, int epi_p0, int epi_p1, const void* xA = nullptr, const void* xB = nullptr, int kx = 0);

// 256x256 tiles run one workgroup per CU: use them when the tile count fills (most of) a whole number of rounds over
// the 256 CUs and little of the tile area hangs over the matrix edge.
int g_gemm_w4 = 1;     // csm_set_gemm_tuning(1, v)
int g_w4_kext = 1;     // csm_set_gemm_tuning(7, v): K-extension products on the four-wave kernel
static bool prefer_256(int M, int N, int K, int batch, double need = 0.80) {
    if (K % 64 != 0 || M < 8 || N < 8) return false;
    const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256) * batch;
    const long long rounds = (tiles + 255) / 256;
    const double fill = (double)tiles / (double)(rounds * 256);
    const double area = ((double)M * N * batch) / ((double)tiles * 65536.0);
    return fill * area >= need;
}

// tuning / A-B switch (tools/gemm_bench.py, tests): 0 = 128x128 kernel with register staging, 1 = 128x128 kernel with
// LDS-DMA staging, 2 = auto (default: the 256x256 kernel where its tiles fill the chip), 3 = force the 256x256 kernel
extern int g_persistent;
// A/B switch: 1 (default) = the 256x256 kernel runs one persistent workgroup per CU over its tile list, 0 = one tile per workgroup
extern "C" int csm_set_gemm256_persistent(int v) { g_persistent = v ? 1 : 0; return 0; }
extern "C" int csm_get_gemm256_persistent(void) { return g_persistent; }
extern "C" const char* csm_gemm_last_kernel(void) { return g_last_gemm_kernel; }

// tuning switches of the 256x256 kernels for A/B runs: key 0 = epilogue-read prefetch of the eight-wave kernel (default 1);
// key 1 = the auto variant gives batch-1 products without K-extension to the four-wave kernel (default 1)
extern int g_gemm_touch;
extern int g_w4_stagger[4];
extern int g_w4_fast_epi, g_w4_n6;
extern "C" int csm_set_gemm_tuning(int key, int value) {
    CSM_REQUIRE(key >= 0 && key <= 8, "csm_set_gemm_tuning: unknown key %d", key);
    if (key == 0) g_gemm_touch = value ? 1 : 0;
    else if (key == 1) g_gemm_w4 = value ? 1 : 0;
    else if (key == 6) g_w4_fast_epi = value ? 1 : 0;
    else if (key == 7) g_w4_kext = value ? 1 : 0;
    else if (key == 8) g_w4_n6 = value ? 1 : 0;
    else {      // 2: number of start groups of a persistent four-wave launch; 3 / 4 / 5: offset between groups in 10 ns ticks
        CSM_REQUIRE(value >= 0 && value <= (key == 2 ? 32 : 20000), "csm_set_gemm_tuning: key %d value %d out of range", key, value);
        g_w4_stagger[key - 2] = key == 2 ? (value < 1 ? 1 : value) : value;
    }
    return 0;
}

extern "C" int csm_set_gemm_variant(int v) {
    CSM_REQUIRE(v >= 0 && v <= 4, "csm_set_gemm_variant: %d is not one of 0..4", v);
    g_gemm_variant = v;
    return 0;
}

// epilogue: 0 none; 1 SwiGLU forward (C = gate/up interleaved [M][N], aux_out = act [M][N/2]); 2 SwiGLU backward
// (GEMM computes d(act) [M][N]; aux_in = gate/up [M][2N]; C = d(gate/up) interleaved [M][2N], ldc >= 2N)
static int gemm_dispatch(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda,
                         int ldb, int ldc, int ldr, int transA, int transB, int out_f32, float alpha, int batch,
                         long long strideA, long long strideB, long long strideC, long long strideR, int epilogue,
                         const void* aux_in, void* aux_out, int ld_aux, hipStream_t stream, int epi_p0, int epi_p1,
                         const void* xA = nullptr, const void* xB = nullptr, int kx = 0) {
    CSM_REQUIRE(A && B && C, "csm_gemm_bf16: null operand");
    CSM_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "csm_gemm_bf16: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
    CSM_REQUIRE((lda & 7) == 0 && (ldb & 7) == 0, "csm_gemm_bf16: lda/ldb must be multiples of 8 (lda=%d ldb=%d)", lda, ldb);
    CSM_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "csm_gemm_bf16: A/B must be 16-byte aligned");
    CSM_REQUIRE((strideA & 7) == 0 && (strideB & 7) == 0, "csm_gemm_bf16: batch strides of A/B must be multiples of 8");
    if (!transA) CSM_REQUIRE((K & 7) == 0, "csm_gemm_bf16: K must be a multiple of 8 when A is [M][K] (K=%d)", K);
    else CSM_REQUIRE((M & 7) == 0, "csm_gemm_bf16: M must be a multiple of 8 when A is [K][M] (M=%d)", M);
    if (!transB) CSM_REQUIRE((K & 7) == 0, "csm_gemm_bf16: K must be a multiple of 8 when B is [N][K] (K=%d)", K);
    else CSM_REQUIRE((N & 7) == 0, "csm_gemm_bf16: N must be a multiple of 8 when B is [K][N] (N=%d)", N);
    CSM_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? N : K) && ldc >= (epilogue == 2 ? 2 * N : N), "csm_gemm_bf16: leading dimension too small");
    // (R is allowed with the SwiGLU-forward epilogue: gate/up = alpha * acc + R, activation from the sum - the port through
    //  which a LoRA adapter's (alpha/r) t B^T, written first, joins the frozen w1/w3 product without leaving the fused path)
    if (epilogue == 1) CSM_REQUIRE(!out_f32 && aux_out && (N & 3) == 0 && (ldc & 3) == 0 && ld_aux >= N / 2 && (ld_aux & 1) == 0, "csm_gemm_bf16_ex: bad SwiGLU-forward epilogue arguments");
    if (epilogue == 2) CSM_REQUIRE(!out_f32 && aux_in && (N & 3) == 0 && (ldc & 7) == 0 && (ld_aux & 7) == 0 && ld_aux >= 2 * N && !R && ((uintptr_t)aux_in & 15) == 0 && ((uintptr_t)C & 15) == 0, "csm_gemm_bf16_ex: bad SwiGLU-backward epilogue arguments");
    CSM_REQUIRE(epilogue >= 0 && epilogue <= 3, "csm_gemm_bf16_ex: unknown epilogue %d", epilogue);
    // variant 4: the four-wave 256x256 kernel with the hand-scheduled K loop (gemm256w4.hip), where it applies
    // (the four-wave kernel beats the 128x128 one from 1.5 rounds of tiles on: fused q|k|v forward, 384 tiles, 97 vs 115 us)
    if (((g_gemm_variant == 2 && g_gemm_w4 && prefer_256(M, N, K, batch, 0.70)) || (g_gemm_variant == 4 && K % 64 == 0 && M >= 8 && N >= 8)) && (kx == 0 || g_w4_kext))
        return csm_gemm256w4_launch(A, B, C, R, M, N, K, lda, ldb, ldc, ldr, transA, transB, out_f32, alpha, batch, strideA, strideB,
                                    strideC, strideR, epilogue, aux_in, aux_out, ld_aux, stream, epi_p0, epi_p1, xA, xB, kx);
    if ((g_gemm_variant == 2 && prefer_256(M, N, K, batch)) || (g_gemm_variant >= 3 && K % 64 == 0 && M >= 8 && N >= 8))
        return csm_gemm256_launch(A, B, C, R, M, N, K, lda, ldb, ldc, ldr, transA, transB, out_f32, alpha, batch, strideA,
                                  strideB, strideC, strideR, epilogue, aux_in, aux_out, ld_aux, stream, epi_p0, epi_p1, xA, xB, kx);
    GemmArgs g;
    g.epi_p0 = epi_p0; g.epi_p1 = epi_p1;
    g.xA = (const bf16_t*)xA; g.xB = (const bf16_t*)xB; g.kx = kx;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.R = (const bf16_t*)R;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
    g.sA = strideA; g.sB = strideB; g.sC = strideC; g.sR = strideR;
    g.alpha = alpha;
    g.epi_mode = epilogue; g.aux_in = (const bf16_t*)aux_in; g.aux_out = (bf16_t*)aux_out; g.ld_aux = ld_aux;
    g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
    if (!transA && !transB) return launch<0, 0>(g, out_f32, batch, stream);
    if (!transA && transB) return launch<0, 1>(g, out_f32, batch, stream);
    if (transA && transB) return launch<1, 1>(g, out_f32, batch, stream);
    return launch<1, 0>(g, out_f32, batch, stream);
}

extern "C" int csm_gemm_bf16_ex(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda,
                                int ldb, int ldc, int ldr, int transA, int transB, int out_f32, float alpha, int batch,
                                long long strideA, long long strideB, long long strideC, long long strideR, int epilogue,
                                const void* aux_in, void* aux_out, int ld_aux, hipStream_t stream) {
    CSM_REQUIRE(epilogue >= 0 && epilogue <= 2, "csm_gemm_bf16_ex: unknown epilogue %d", epilogue);
    return gemm_dispatch(A, B, C, R, M, N, K, lda, ldb, ldc, ldr, transA, transB, out_f32, alpha, batch, strideA, strideB, strideC,
                         strideR, epilogue, aux_in, aux_out, ld_aux, stream, 0, 0);
}

// Fused q|k|v projection + RoPE (torchtune Llama3ScaledRoPE, reference model.py:23-24,40-41): C[M][N] = A[M][K] W[N][K]^T with the
// interleaved pairs of columns [0, n_rope_cols) - the q and k heads, head_dim features each - rotated by position
// (row % rows_per_seq) using the fp32 (cos, sin) table [P][head_dim/2][2].  Positions are the row index inside a sequence
// (training / prefill); callers with explicit positions use csm_rope after a plain csm_gemm_bf16.
extern "C" int csm_gemm_bf16_rope(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                                  const float* rope_table, int rows_per_seq, int n_rope_cols, int head_dim, hipStream_t stream) {
    CSM_REQUIRE(rope_table && rows_per_seq > 0 && head_dim >= 8 && (head_dim & 7) == 0 && n_rope_cols >= 0 && n_rope_cols <= N &&
                n_rope_cols % head_dim == 0 && (N & 7) == 0 && (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0,
                "csm_gemm_bf16_rope: bad arguments");
    return gemm_dispatch(A, W, C, nullptr, M, N, K, lda, ldw, ldc, 0, 0, 0, 0, 1.f, 1, 0, 0, 0, 0, EPI_ROPE, rope_table, nullptr,
                         rows_per_seq, stream, n_rope_cols, head_dim);
}

// A frozen projection and its LoRA adapters as ONE product (reference LoRALinear.__call__, src/csm/mlx/components/lora.py:85-105:
// y = x W0^T + scale * (x A^T) B^T; the same for the input gradient dx = dy W0 + scale * (dy B) A):
//   C[M][N] = A . B  (operand layouts by transA / transB as in csm_gemm_bf16)  +  xA[M][kx] . xB[N][kx]^T   (+ R)
// xA / xB are row-major with leading dimension kx (a multiple of 32; adapter ranks padded with zeros), bf16, 16-byte
// aligned; the extra k-steps run after the main loop, in fp32, in the same accumulators, before the epilogue.
// epilogue: 0 none, 1 / 2 SwiGLU forward / backward (aux_out / aux_in, ld_aux as in csm_gemm_bf16_ex), 3 RoPE (aux_in = table,
// ld_aux = rows per sequence, rope_cols, head_dim as in csm_gemm_bf16_rope).
extern "C" int csm_gemm_bf16_kext(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb, int ldc,
                                  int ldr, int transA, int transB, const void* xA, const void* xB, int kx, int epilogue,
                                  const void* aux_in, void* aux_out, int ld_aux, int rope_cols, int head_dim, hipStream_t stream) {
    CSM_REQUIRE(xA && xB && kx > 0 && kx % 32 == 0 && kx <= 256, "csm_gemm_bf16_kext: kx must be a multiple of 32 in [32, 256] (kx=%d)", kx);
    CSM_REQUIRE(((uintptr_t)xA & 15) == 0 && ((uintptr_t)xB & 15) == 0, "csm_gemm_bf16_kext: xA / xB must be 16-byte aligned");
    CSM_REQUIRE(epilogue >= 0 && epilogue <= 3, "csm_gemm_bf16_kext: epilogue must be 0..3");
    if (epilogue == 3)
        CSM_REQUIRE(aux_in && ld_aux > 0 && head_dim >= 8 && (head_dim & 7) == 0 && rope_cols >= 0 && rope_cols <= N && rope_cols % head_dim == 0 &&
                    (N & 7) == 0 && (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 && !transA && !transB, "csm_gemm_bf16_kext: bad RoPE epilogue arguments");
    return gemm_dispatch(A, B, C, R, M, N, K, lda, ldb, ldc, ldr, transA, transB, 0, 1.f, 1, 0, 0, 0, 0, epilogue, aux_in, aux_out, ld_aux,
                         stream, epilogue == 3 ? rope_cols : 0, epilogue == 3 ? head_dim : 0, xA, xB, kx);
}

// out[M][N] = alpha * X[M][K] . Wt[N][K]^T, N = 32 or 64, K a multiple of 128 - the skinny products of a LoRA group (reference
// LoRALinear: x A^T and dy B, src/csm/mlx/components/lora.py:85-105), shaped for bandwidth instead of MFMA tiles.
extern "C" int csm_skinny_nt_bf16(const void* X, const void* Wt, void* out, int M, int N, int K, int ldx, int ldw, int ldo, float alpha,
                                  hipStream_t stream) {
    CSM_REQUIRE(X && Wt && out && M > 0, "csm_skinny_nt_bf16: null operand");
    CSM_REQUIRE((N == 32 || N == 64) && K >= 128 && K % 128 == 0, "csm_skinny_nt_bf16: N must be 32 or 64 and K a multiple of 128 (N=%d K=%d)", N, K);
    CSM_REQUIRE((ldx & 7) == 0 && (ldw & 7) == 0 && (ldo & 3) == 0 && ldx >= K && ldw >= K && ldo >= N, "csm_skinny_nt_bf16: bad leading dimensions");
    CSM_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)Wt & 15) == 0 && ((uintptr_t)out & 7) == 0, "csm_skinny_nt_bf16: operands must be 16-byte aligned");
    const dim3 grid((M + 15) / 16), block(256);
    if (N == 32) hipLaunchKernelGGL((skinny_nt_kernel<2>), grid, block, 0, stream, (const bf16_t*)X, (const bf16_t*)Wt, (bf16_t*)out, M, K, ldx, ldw, ldo, alpha);
    else hipLaunchKernelGGL((skinny_nt_kernel<4>), grid, block, 0, stream, (const bf16_t*)X, (const bf16_t*)Wt, (bf16_t*)out, M, K, ldx, ldw, ldo, alpha);
    CSM_CHECK_LAUNCH("csm_skinny_nt_bf16");
    return 0;
}

extern "C" int csm_gemm_bf16(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda,
                             int ldb, int ldc, int ldr, int transA, int transB, int out_f32, float alpha, int batch,
                             long long strideA, long long strideB, long long strideC, long long strideR,
                             hipStream_t stream) {
    return csm_gemm_bf16_ex(A, B, C, R, M, N, K, lda, ldb, ldc, ldr, transA, transB, out_f32, alpha, batch, strideA, strideB,
                            strideC, strideR, 0, nullptr, nullptr, 0, stream);
}

int csm_gemm256_pair_launch(const void* dY, const void* W, void* dX, int M, int Nout, int Kin, int ld_dy, int ldw, int ld_dx,
                            int dx_epi, const void* aux_in, int ld_aux,
                            const void* X, int ldx, void* dW, int ld_dw, int accumulate, float alpha_w, hipStream_t stream);

// The two backward products of a Linear layer y = x W^T in ONE launch (torchtune nn.Linear backward; reference loop
// src/csm/training/trainer.py:261-263 `loss.backward()`):
//   dX[M][Kin]    = dY[M][Nout] . W[Nout][Kin]            (epilogue 0)
//                 or, epilogue 2, the SwiGLU backward of it: W = w2 [d][F], aux_in = gate/up [M][2F], dX = d(gate/up) [M][2F]
//   dW[Nout][Kin] (+)= alpha_w * dY^T . X[M][Kin]
// Tiles of the two products are interleaved over the chip (gemm256.hip: gemm256pair_kernel).
extern "C" int csm_gemm_bf16_dgrad_wgrad(const void* dY, const void* W, void* dX, const void* X, void* dW, int M, int Nout, int Kin,
                                         int ld_dy, int ldw, int ld_dx, int ldx, int ld_dw, int dx_epilogue, const void* aux_in,
                                         int ld_aux, int accumulate, float alpha_w, hipStream_t stream) {
    CSM_REQUIRE(dY && W && dX && X && dW, "csm_gemm_bf16_dgrad_wgrad: null operand");
    CSM_REQUIRE(M > 0 && Nout > 0 && Kin > 0 && M % 64 == 0 && Nout % 64 == 0 && (Kin & 7) == 0,
                "csm_gemm_bf16_dgrad_wgrad: M and Nout must be multiples of 64, Kin of 8 (M=%d Nout=%d Kin=%d)", M, Nout, Kin);
    CSM_REQUIRE((ld_dy & 7) == 0 && (ldw & 7) == 0 && (ldx & 7) == 0 && (ld_dw & 7) == 0 && (ld_dx & 7) == 0 && ld_dy >= Nout && ldw >= Kin &&
                ldx >= Kin && ld_dw >= Kin, "csm_gemm_bf16_dgrad_wgrad: bad leading dimensions");
    CSM_REQUIRE(((uintptr_t)dY & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)dX & 15) == 0 &&
                ((uintptr_t)dW & 15) == 0, "csm_gemm_bf16_dgrad_wgrad: operands must be 16-byte aligned");
    CSM_REQUIRE(dx_epilogue == 0 || dx_epilogue == 2, "csm_gemm_bf16_dgrad_wgrad: dx_epilogue must be 0 or 2");
    if (dx_epilogue == 2)
        CSM_REQUIRE(aux_in && (Kin & 3) == 0 && ld_dx >= 2 * Kin && (ld_aux & 7) == 0 && ld_aux >= 2 * Kin && ((uintptr_t)aux_in & 15) == 0,
                    "csm_gemm_bf16_dgrad_wgrad: bad SwiGLU-backward epilogue arguments");
    else
        CSM_REQUIRE(ld_dx >= Kin, "csm_gemm_bf16_dgrad_wgrad: ld_dx too small");
    return csm_gemm256_pair_launch(dY, W, dX, M, Nout, Kin, ld_dy, ldw, ld_dx, dx_epilogue, aux_in, ld_aux, X, ldx, dW, ld_dw, accumulate,
                                   alpha_w, stream);
}

int csm_gemm256_two_wgrad_launch(const void* dY1, const void* X1, void* dW1, int N1, int K1, int ld_dy1, int ldx1, int ld_dw1,
                                 const void* dY2, const void* X2, void* dW2, int N2, int K2, int ld_dy2, int ldx2, int ld_dw2,
                                 int M, int accumulate, float alpha, hipStream_t stream);

// Two weight gradients that share the token dimension M, in ONE launch of 256x256 tiles (autograd's dW = dY^T X of two
// nn.Linear layers; reference loop src/csm/training/trainer.py:261-263).  For outputs too small to fill 256 CUs alone.
int csm_gemm256w4_two_wgrad_launch(const void* dY1, const void* X1, void* dW1, int N1, int K1, int ld_dy1, int ldx1, int ld_dw1,
                                   const void* dY2, const void* X2, void* dW2, int N2, int K2, int ld_dy2, int ldx2, int ld_dw2,
                                   int M, int accumulate, float alpha, hipStream_t stream);
extern "C" int csm_gemm_bf16_two_wgrad(const void* dY1, const void* X1, void* dW1, int N1, int K1, int ld_dy1, int ldx1, int ld_dw1,
                                       const void* dY2, const void* X2, void* dW2, int N2, int K2, int ld_dy2, int ldx2, int ld_dw2,
                                       int M, int accumulate, float alpha, hipStream_t stream) {
    CSM_REQUIRE(dY1 && X1 && dW1 && dY2 && X2 && dW2, "csm_gemm_bf16_two_wgrad: null operand");
    CSM_REQUIRE(M > 0 && M % 64 == 0 && N1 > 0 && N2 > 0 && K1 > 0 && K2 > 0 && ((N1 | N2 | K1 | K2) & 7) == 0,
                "csm_gemm_bf16_two_wgrad: M must be a multiple of 64, the other dimensions of 8");
    CSM_REQUIRE(((ld_dy1 | ldx1 | ld_dw1 | ld_dy2 | ldx2 | ld_dw2) & 7) == 0 && ld_dy1 >= N1 && ldx1 >= K1 && ld_dw1 >= K1 && ld_dy2 >= N2 &&
                ldx2 >= K2 && ld_dw2 >= K2, "csm_gemm_bf16_two_wgrad: bad leading dimensions");
    CSM_REQUIRE((((uintptr_t)dY1 | (uintptr_t)X1 | (uintptr_t)dW1 | (uintptr_t)dY2 | (uintptr_t)X2 | (uintptr_t)dW2) & 15) == 0,
                "csm_gemm_bf16_two_wgrad: operands must be 16-byte aligned");
    if (g_gemm_w4 && g_gemm_variant != 3)
        return csm_gemm256w4_two_wgrad_launch(dY1, X1, dW1, N1, K1, ld_dy1, ldx1, ld_dw1, dY2, X2, dW2, N2, K2, ld_dy2, ldx2, ld_dw2, M,
                                              accumulate, alpha, stream);
    return csm_gemm256_two_wgrad_launch(dY1, X1, dW1, N1, K1, ld_dy1, ldx1, ld_dw1, dY2, X2, dW2, N2, K2, ld_dy2, ldx2, ld_dw2, M,
                                        accumulate, alpha, stream);
}

// n weight gradients dW_i[N_i][K_i] (+)= alpha dY_i[M][N_i]^T X_i[M][K_i] in ONE launch of 256 x 256 tiles (four-wave kernel): products
// that each fill a fraction of a round of the 256 CUs (the attention projections' gradients of a layer: 160 tiles) fill whole
// rounds together.  Every tile's arithmetic is that of csm_gemm_bf16_two_wgrad / csm_gemm_bf16 on the same product.
int csm_gemm256w4_multi_wgrad_launch(int n, const void* const* dY, const void* const* X, void* const* dW, const int* N, const int* K,
                                     const int* ld_dy, const int* ldx, const int* ld_dw, int M, int accumulate, float alpha,
                                     hipStream_t stream);
extern "C" int csm_gemm_bf16_multi_wgrad(int n, const void* const* dY, const void* const* X, void* const* dW, const int* N, const int* K,
                                         const int* ld_dy, const int* ldx, const int* ld_dw, int M, int accumulate, float alpha,
                                         hipStream_t stream) {
    CSM_REQUIRE(n >= 1 && n <= 12 && dY && X && dW && N && K && ld_dy && ldx && ld_dw, "csm_gemm_bf16_multi_wgrad: 1..12 products, no null array");
    CSM_REQUIRE(M > 0 && M % 64 == 0, "csm_gemm_bf16_multi_wgrad: M must be a positive multiple of 64");
    for (int i = 0; i < n; ++i) {
        CSM_REQUIRE(dY[i] && X[i] && dW[i], "csm_gemm_bf16_multi_wgrad: null operand in product %d", i);
        CSM_REQUIRE(N[i] > 0 && K[i] > 0 && ((N[i] | K[i] | ld_dy[i] | ldx[i] | ld_dw[i]) & 7) == 0 && ld_dy[i] >= N[i] && ldx[i] >= K[i] &&
                    ld_dw[i] >= K[i], "csm_gemm_bf16_multi_wgrad: product %d: dimensions must be multiples of 8, leading dimensions >= the row length", i);
        CSM_REQUIRE((((uintptr_t)dY[i] | (uintptr_t)X[i] | (uintptr_t)dW[i]) & 15) == 0, "csm_gemm_bf16_multi_wgrad: product %d: operands must be 16-byte aligned", i);
    }
    if (!(g_gemm_w4 && g_gemm_variant != 3)) {        // kernel A/B switches: one product per launch through the ordinary dispatch
        for (int i = 0; i < n; ++i) {
            const int rc = csm_gemm_bf16(dY[i], X[i], dW[i], accumulate ? dW[i] : nullptr, N[i], K[i], M, ld_dy[i], ldx[i], ld_dw[i], ld_dw[i], 1, 1,
                                         0, alpha, 1, 0, 0, 0, 0, stream);
            if (rc) return rc;
        }
        return 0;
    }
    return csm_gemm256w4_multi_wgrad_launch(n, dY, X, dW, N, K, ld_dy, ldx, ld_dw, M, accumulate, alpha, stream);
}
