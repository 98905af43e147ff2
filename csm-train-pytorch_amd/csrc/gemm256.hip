// 256x256x64 deep-pipelined bf16 MFMA GEMM for gfx950 (same contract and operand layouts as gemm.hip).
//
// 512 threads = 8 waves as 2(M) x 4(N); each wave owns a 128x64 output block = 8x4 accumulators of
// mfma_f32_16x16x32_bf16 (128 VGPRs).  One workgroup per CU with all 160 KiB of LDS: A double-buffered, B triple-buffered
// half-tiles of 16 KiB (each exactly the 128x64 / 64x128 swizzled image of gemm_common.h), filled by LDS-DMA
// (global_load_lds, 2 instructions per wave and half-tile), never drained to zero inside the loop: ONE counted
// `s_waitcnt vmcnt(4)` and ONE raw s_barrier per K-tile (a __syncthreads() would drain the DMA queue).  See the kernel
// comment for the phase schedule.  Two earlier forms - four 64x32 quadrant phases without register prefetch, and a
// persistent workgroup walking a tile list with the DMA look-ahead running across tile boundaries - measured slower
// (the persistent one because loads, LDS-DMA and stores share one in-order vmcnt: the next tile's counted waits also wait
// for the previous tile's stores) and were removed; they are in the history.
#include "gemm_common.h"
#include <type_traits>

int g_persistent = 1;     // csm_set_gemm256_persistent (gemm.hip)
int g_gemm_touch = 1;     // csm_set_gemm_tuning(0, v): epilogue-read prefetch in the 256x256 kernel
extern const char* g_last_gemm_kernel;   // csm_gemm_last_kernel (gemm.hip)
#define CSM_KNAME256(TA, TB, f32) \
    ((f32) ? (TA ? (TB ? "gemm256p_kernel<1, 1, float>" : "gemm256p_kernel<1, 0, float>") : (TB ? "gemm256p_kernel<0, 1, float>" : "gemm256p_kernel<0, 0, float>")) \
           : (TA ? (TB ? "gemm256p_kernel<1, 1, unsigned short>" : "gemm256p_kernel<1, 0, unsigned short>") : (TB ? "gemm256p_kernel<0, 1, unsigned short>" : "gemm256p_kernel<0, 0, unsigned short>")))

namespace {

constexpr int HALF = 16384;            // bytes per half-tile
constexpr int BUF = 4 * HALF;          // bytes per K-tile buffer

struct Gemm256Args {
    const bf16_t* A; const bf16_t* B; void* C; const bf16_t* R;
    int M, N, K, lda, ldb, ldc, ldr;
    long long sA, sB, sC, sR;
    float alpha;
    int tiles_m, tiles_n;
    int epi_mode; const bf16_t* aux_in; bf16_t* aux_out; int ld_aux;
    int epi_p0, epi_p1;
    const bf16_t* xA; const bf16_t* xB; int kx;      // K-extension operands (gemm_common.h: k_extend)
    int touch;                                       // epilogue-read prefetch on (csm_set_gemm_tuning(0, v))
};

// one 16-KiB half-tile by LDS-DMA: 16 pieces of 1 KiB over 8 waves
template <int T>
__device__ __forceinline__ void issue_half(const bf16_t* __restrict__ P, int ld, int rows, int row0, int k0, char* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = wave * 2 + i;
        const bf16_t* src;
        if (T == 0) {
            const int r = 8 * j + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            int gr = row0 + r;
            gr = gr < rows ? gr : rows - 1;
            src = P + (size_t)gr * ld + k0 + c * 8;
        } else {
            const int kr = 4 * j + (lane >> 4);
            const int u = lane & 15;
            const int sl = (u >> 1) ^ ks_swz(kr);
            int gc = row0 + sl * 16 + (u & 1) * 8;
            gc = gc < rows ? gc : rows - 8;
            src = P + (size_t)(k0 + kr) * ld + gc;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, 0, 0);
    }
}

// The same half-tile with the address split into a wave-uniform base (scalar registers, advanced per K-tile by scalar adds)
// and a per-lane byte offset computed ONCE per output tile: the K loop then spends no vector instructions on addresses
// (recomputing row * ld + column per piece cost ~130 VALU instructions per K-tile and wave - 64-bit multiplies among them -
// on the issue port the MFMAs need).
struct DmaSrc {
    const char* base[2];      // K-tile 0 of half h (uniform)
    unsigned off[2][2];       // per-lane byte offset of piece i of half h
    long long step;           // bytes per K-tile
};
template <int T>
__device__ __forceinline__ void dma_prepare(const bf16_t* __restrict__ P, int ld, int rows, int row0, DmaSrc& d) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    d.step = T == 0 ? 128 : (long long)ld * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (T == 0) {
            const int ub = min(row0 + h * 128, rows - 1);                       // uniform first row (clamped like the lanes' rows)
            d.base[h] = reinterpret_cast<const char*>(P + (size_t)ub * ld);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = wave * 2 + i;
                const int r = 8 * j + (lane >> 3);
                const int c = (lane & 7) ^ (r & 7);
                const int rel = min(r, rows - 1 - ub);                          // row ub + rel = min(row0 + 128 h + r, rows - 1)
                d.off[h][i] = (unsigned)(rel * ld + c * 8) * 2u;
            }
        } else {
            const int ub = min(row0 + h * 128, rows - 8);
            d.base[h] = reinterpret_cast<const char*>(P + ub);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = wave * 2 + i;
                const int kr = 4 * j + (lane >> 4);
                const int u = lane & 15;
                const int sl = (u >> 1) ^ ks_swz(kr);
                const int gc = min(row0 + h * 128 + sl * 16 + (u & 1) * 8, rows - 8);
                d.off[h][i] = (unsigned)(kr * ld + (gc - ub)) * 2u;
            }
        }
    }
}
// (round 3) The request itself is the BUFFER form of LDS-DMA: `buffer_load_dwordx4 v_off, s[rsrc], 0 offen lds` - a 128-bit
// resource descriptor in scalar registers (rebuilt from the running scalar pointer: two scalar instructions) plus ONE 32-bit
// per-lane offset.  The flat form (`global_load_lds_dwordx4 v[addr:addr+1]`) needed a 64-bit vector add per piece inside the K
// loop and - worse - made hipcc keep eight 64-bit per-lane offsets and eight 64-bit running vector pointers alive: ~32 VGPRs
// in a kernel that has none to spare (256 allocated, several values spilled to scratch and reloaded, each reload behind an
// `s_waitcnt vmcnt(0)`, in the first K-tile of every output tile).
__device__ __forceinline__ void issue_half_pre(const DmaSrc& d, int h, const char* b /* d.base[h] + K-tile * d.step, uniform */, char* lds) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (num_records = 2^32 - 1: the range check never fires; rows / columns past the matrix were clamped into it by dma_prepare)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(b), 0, 0xffffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (wave * 2 + i) * 1024), 16,
                                                 (int)d.off[h][i], 0, 0, 0);
}

#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define BARRIER() __builtin_amdgcn_s_barrier()

// compile-time loop: the index reaches inline-asm "i" operands as a constant
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int OFF>
__device__ __forceinline__ bf16x8 row_read_asm(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

// Operand fragments whose EVERY LDS read is inline asm (round 3).  With plain loads for the K-contiguous image hipcc kept its
// own lgkmcnt bookkeeping, and because it cannot see the asm transposed reads in between it fell back to `s_waitcnt lgkmcnt(0)`
// in front of every MFMA group fed by plain loads - the prefetched pair was waited for in full in ph1 and ph3 - and put
// `s_waitcnt vmcnt(0)` in front of the first plain LDS load after the prologue's LDS-DMA (possible alias), which also drained
// K-tile 1's requests once per output tile.  Now nothing here is visible to that pass: the phase code counts its own waits.
// Loads take a tile range [I0, I1) and a k-step (0, 1, or 2 = both) so that a phase can start its first MFMAs on the
// fragments that arrive first.
template <int T, int NT>
struct AFrags {
    bf16x8 row[NT][2];
    FragT1 tr[NT][2];
    template <int I0, int I1, int KS>
    __device__ __forceinline__ void load(const char* img, int r0, int lane) {
        if constexpr (T == 0) {
            const int r = r0 + (lane & 15);
            const unsigned a0 = (unsigned)(uintptr_t)img + r * 128 + (((lane >> 4) ^ (r & 7)) << 4);
            const unsigned a1 = a0 ^ 64u;                      // k-step 1: 16-byte chunk index + 4, under the XOR swizzle
            static_for<I0, I1>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (KS != 1) row[i][0] = row_read_asm<i * 2048>(a0);
                if constexpr (KS != 0) row[i][1] = row_read_asm<i * 2048>(a1);
            });
        } else {
            const unsigned base = (unsigned)(uintptr_t)img + tr_lane_off(lane);
            const int key = tr_lane_key(lane);
            static_for<I0, I1>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const unsigned a = base + ((((r0 >> 4) + i) ^ key) << 5);
                if constexpr (KS != 1) tr[i][0] = load_frag_tr<0>(a);
                if constexpr (KS != 0) tr[i][1] = load_frag_tr<1>(a);
            });
        }
    }
    // LDS read instructions of load<I0, I1, KS>
    static constexpr int reads(int tiles, int ksteps) { return tiles * ksteps * (T == 0 ? 1 : 2); }
    __device__ __forceinline__ bf16x8 get(int i, int ks) const {
        if constexpr (T == 0) return row[i][ks];
        else return cat4(tr[i][ks].lo, tr[i][ks].hi);
    }
};

// ---------------------------------------------------------------------------------------------------------------
// A K-tile is four phases.  A phase multiplies ONE pair of A m-tiles (32 rows) by all four B n-tiles of the wave
// (16 MFMAs) while the NEXT pair's fragments are already being read from LDS into the other A register set, so fragment
// reads run under the MFMAs instead of in front of them.  Only the 8 B fragment reads at the start of a K-tile are
// exposed.  Slot liveness: the B slot of tile t is dead after ph1(t)'s reads, the A slots after ph3(t) (which drains its
// prefetch reads before the barrier); the single barrier at the end of ph3 releases both and publishes tile t+1.  The
// LDS-DMA of later tiles is spread over the phases (order per operand layout: see EARLY / EARLYB in the kernel), e.g.
//     ph1(t): B0(t+2)    ph2(t): B1(t+2)    ph3(t): -    [wait, barrier]    ph4(t): A0(t+2), A1(t+2)
// with one counted wait per K-tile at the end of ph3 (vmcnt(4): all but the two youngest B half-tiles have landed).
// Experiment switches for tools/probes/ablate_gemm.sh (never set in the shipped build): drop one ingredient of the main
// loop to see what bounds it.  bit0: no LDS-DMA after the prologue; bit1: no fragment reads; bit2: no MFMA; bit3: no barrier.
#ifndef CSM_ABLATE
#define CSM_ABLATE 0
#endif
constexpr bool ABL_G = CSM_ABLATE & 1, ABL_L = CSM_ABLATE & 2, ABL_M = CSM_ABLATE & 4, ABL_B = CSM_ABLATE & 8;
constexpr bool NO_PRIO = CSM_ABLATE & 64;    // no s_setprio around the MFMA groups
constexpr bool EARLY_B = CSM_ABLATE & 32;    // flip the layout's default for the B half-tile issue phases (see the kernel)
constexpr bool EARLY_A1 = CSM_ABLATE & 16;   // flip: A1 of the next tile in ph1 instead of with A0 in ph4

#define WAIT_VM_DYN(n) do { switch (n) { case 63: WAIT_VM(63); break; case 56: WAIT_VM(56); break; case 52: WAIT_VM(52); break; \
    case 40: WAIT_VM(40); break; case 36: WAIT_VM(36); break; case 24: WAIT_VM(24); break; case 20: WAIT_VM(20); break;       \
    default: WAIT_VM(4); break; } } while (0)

// ``id`` = first tile of this workgroup, ``stride`` = tiles between its consecutive tiles (the grid size of a persistent
// launch; anything >= the tile count for one tile per workgroup).  A workgroup that has a next tile requests that tile's
// first two K-tiles BEFORE it stores the finished one: the loads enter the memory system ahead of the 128 KiB of stores
// every CU emits at the same moment (in a fresh workgroup they queue behind the previous round's stores), the store drain
// overlaps the next main loop's first K-tiles, and no workgroup is torn down and launched between rounds.  Loads and
// stores share one in-order counter, so the waits for those early loads are counted PAST the epilogue's memory
// operations (``pend`` = a lower bound of their number, known for full tiles on the 16-byte epilogue paths; 0 otherwise,
// which degrades to waiting for the stores as well).
template <int TA, int TB, typename OutT>
__device__ __forceinline__ void gemm256_body(const Gemm256Args& g, int id, const int stride, const int bz, char* smem) {
    const int nwg = g.tiles_m * g.tiles_n;
    // m-tiles per raster group (the tiles an XCD works on at one time are GROUP_M m-tiles x 32 / GROUP_M n-tiles).
    // tools/probes/group_m_probe.sh: 8 and 16 lose 1-4 %; 2 wins 4 % on the bare N = 16384 GEMM but nothing in the real
    // step (tools/probes/ab_group_m_step.sh, where that GEMM carries the SwiGLU epilogue).  CSM_GROUP_M pins it for the probes.
#ifndef CSM_GROUP_M
#define CSM_GROUP_M 4
#endif
    constexpr int GROUP_M = CSM_GROUP_M;
    auto coords = [&](int t_id, int& m0_, int& n0_) {
        {
            const int q = nwg >> 3, r = nwg & 7, xcd = t_id & 7, within = t_id >> 3;
            t_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
        }
        const int per_group = GROUP_M * g.tiles_n;
        const int grp = t_id / per_group;
        const int first_m = grp * GROUP_M;
        const int gsz = min(g.tiles_m - first_m, GROUP_M);
        m0_ = (first_m + (t_id % per_group) % gsz) * 256;
        n0_ = ((t_id % per_group) / gsz) * 256;
    };
    int m0, n0;
    coords(id, m0, n0);

    const bf16_t* A = g.A + (size_t)bz * g.sA;
    const bf16_t* B = g.B + (size_t)bz * g.sB;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int a_half = wr, b_half = wc >> 1, b_off = (wc & 1) * 64;

    f32x4 acc[8][4];

    // Issue order of the LDS-DMA half-tiles, measured per operand layout (tools/probes/ablate_gemm.sh, 5..9-round medians
    // on the train-step shapes; bits 16 / 32 of CSM_ABLATE flip the two choices):
    //   EARLY  - both A halves of tile t+2 in ph4(t) instead of A0 in ph4(t), A1 in ph1(t+1): the younger half gets one
    //            more phase of latency slack.  nn: 4-6 % faster; with EARLYB also nt (2-5 %) and tn (7 %).
    //   EARLYB - the B halves of tile t+2 in ph1/ph2 instead of ph2/ph3.  Helps nt and tn, costs nn 5 % -> off for nn.
    // Putting all four in one phase (ph4: A(t+2), B(t+3)) was 4-12 % slower: the DMA wants to be spread out.
    // Re-measured after the DMA addresses left the vector pipe (round 2, tools/probes/ablate_gemm2.sh): tn is now 2-3 % faster
    // with neither (A1 of the next tile in ph1, B halves in ph2 / ph3); nt is indifferent; nn unchanged.
    constexpr bool NN = (TA == 0 && TB == 1), TN = (TA == 1 && TB == 1);
    constexpr bool EARLY = EARLY_A1 ? TN : !TN;
    constexpr bool EARLYB = EARLY_B ? (NN || TN) : !(NN || TN);
    const int nt = g.K / 64;
    // LDS map (160 KiB): A double-buffered [2][2 halves] at 0..64 KiB, B TRIPLE-buffered [3][2 halves] at 64..160 KiB
    auto slotA = [&](int b, int h) { return smem + b * (2 * HALF) + h * HALF; };
    auto slotB = [&](int b3, int h) { return smem + 4 * HALF + b3 * (2 * HALF) + h * HALF; };
    DmaSrc srcA, srcB;
    const char *pa[2], *pb[2];                  // K-tile t + 2 of each half: running pointers, advanced by scalar adds (no multiplies in the loop)
    // ``rel``: K-tile relative to the running pointers' tile (0 = tile t + 2 inside the loop, -1 = tile t + 1)
    auto issueA = [&](int h, int rel, int tile, int b) { if (ABL_G && tile >= 2) return; issue_half_pre(srcA, h, pa[h] + rel * srcA.step, slotA(b, h)); };
    auto issueB = [&](int h, int rel, int tile, int b3) { if (ABL_G && tile >= 2) return; issue_half_pre(srcB, h, pb[h] + rel * srcB.step, slotB(b3, h)); };
    constexpr int NRA = (TA == 0) ? 4 : 8;     // LDS read instructions per prefetched A pair
    // the first two K-tiles of the output tile at (m0, n0): 8 + 8 (or 6) DMA instructions per wave
    // ``a1`` : request A1 of K-tile 1 here as well (always for the EARLY layouts; for the others only in front of a later tile
    // of a persistent workgroup, see ``a1_pre``)
    auto prologue = [&](bool a1) {
        dma_prepare<TA>(A, g.lda, g.M, m0, srcA);
        dma_prepare<TB>(B, g.ldb, g.N, n0, srcB);
#pragma unroll
        for (int h = 0; h < 2; ++h) { pa[h] = srcA.base[h] + 2 * srcA.step; pb[h] = srcB.base[h] + 2 * srcB.step; }
        issueA(0, -2, 0, 0); issueA(1, -2, 0, 0); issueB(0, -2, 0, 0); issueB(1, -2, 0, 0);
        if (nt > 1) {
            issueB(0, -1, 1, 1); issueB(1, -1, 1, 1); issueA(0, -1, 1, 1);
            if (a1) issueA(1, -1, 1, 1);
        }
    };
    prologue(EARLY);
    int pend = 0;                               // memory operations of the previous tile's epilogue issued after the prologue
    // ---- epilogue-read prefetch ("touches").  The fused epilogues that READ a tile-sized operand - gate/up of the
    // SwiGLU-backward epilogue (256 KiB per tile), a bf16 residual (128 KiB) - used to start those reads when the main loop
    // was over, on every CU at the same moment, at HBM latency and into a saturated HBM.  Now each lane touches one dword of
    // each 64-byte piece of the tile's operand during the last K-tiles (one `global_load_dword` per wave and K-tile, in ph4):
    // the lines travel HBM -> Infinity Cache / L2 under the main loop and the epilogue's loads hit cache.  The loaded value
    // is never used; its register stays reserved until the loop's final `vmcnt(0)`.
    const char* tch_base = nullptr;
    long long tch_ld = 0, tch_col0 = 0, tch_row_bytes = 0;
    int tch_n = 0, tch_cl2 = 0;
    if constexpr (sizeof(OutT) == 2) {
        if (g.touch && nt >= 16) {
            if (g.epi_mode == EPI_SWIGLU_BWD) {
                tch_base = reinterpret_cast<const char*>(g.aux_in); tch_ld = (long long)g.ld_aux * 2; tch_n = 8; tch_cl2 = 4;
                tch_row_bytes = (long long)g.N * 4;
            } else if (g.R) {
                tch_base = reinterpret_cast<const char*>(g.R + (size_t)bz * g.sR); tch_ld = (long long)g.ldr * 2; tch_n = 4; tch_cl2 = 3;
                tch_row_bytes = (long long)g.N * 2;
            }
        }
    }
    // spread over the whole main loop (all of them within its last third made the loop's own operand fetches queue behind 67 MB
    // of prefetch: same total time); the last one is at least two K-tiles old when the loop's final wait drains the queue
    const int tch_stride = tch_n ? max(1, (nt - 4) / tch_n) : 1;
    int tch_next = 1, tch_q = 0;                // K-tile of the next touch, touches done (per output tile)
    bool tch_prev = false;                      // a touch went out in this K-tile's ph1
    unsigned junk = 0;
    auto touch = [&](int q) {
        const int idx = q * 512 + (int)threadIdx.x;
        const int row = idx >> tch_cl2, ch = idx & ((1 << tch_cl2) - 1);
        const long long colb = min(tch_col0 + ch * 64, tch_row_bytes - 4);
        const char* ptr = tch_base + (long long)min(m0 + row, g.M - 1) * tch_ld + colb;
        asm volatile("global_load_dword %0, %1, off" : "+v"(junk) : "v"(ptr) : "memory");
    };
    // Whether K-tile 1 is COMPLETE in the prologue (8 instructions) or lacks A1 (6; then ph1(0) requests it).  The counted
    // waits past the previous tile's epilogue (``pend``) are only sound when everything the wait is for was requested BEFORE
    // that epilogue: A1(1) requested in ph1(0) is younger than the stores, and `vmcnt(4 + pend)` at ph3(0) would not wait for
    // it (round-2 race, tn layout: waves with a_half = 1 could read A half-tile 1 of K-tile 1 before it landed).  So later
    // tiles of a workgroup always take the complete prologue.
    bool a1_pre = EARLY;
  for (;;) {
    tch_col0 = (long long)n0 * (tch_cl2 == 4 ? 4 : 2);          // gate/up: columns 2 n0 .. of 2-byte elements; residual: n0 ..
    tch_next = 1; tch_q = 0; tch_prev = false;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K-tile 0 has landed (K-tile 1's 8 / 6 instructions and ``pend`` younger operations may still be in flight)
    if (nt > 1) {
        if (a1_pre) { if (pend > 0) { const int w = min(63, 8 + pend); WAIT_VM_DYN(w); } else { WAIT_VM(8); } }
        else { WAIT_VM(6); }                    // (first tile of a non-EARLY layout: pend == 0)
    } else { WAIT_VM(0); }
    BARRIER();

    AFrags<TB, 4> fb;
    AFrags<TA, 2> fa0, fa1;
    if (!ABL_L) { fb.template load<0, 2, 2>(slotB(0, b_half), b_off, lane); fa0.template load<0, 2, 2>(slotA(0, a_half), 0, lane); }
    if (ABL_L) { fb.template load<0, 4, 2>(slotB(0, b_half), b_off, lane); fa1.template load<0, 2, 2>(slotA(0, a_half), 32, lane); }
    int b3 = 0;   // t % 3
#define WAIT_LGKM(n) do { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"((n) > 15 ? 15 : (n)) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    constexpr int RB = AFrags<TB, 4>::reads(2, 1);            // B reads of two n-tiles, one k-step
    // MFMAs of A pair FA (rows MI, MI + 1 of the wave's 8 m-tiles) x n-tiles [J0, J1) at k-step KS
#define MFMA_GROUP(FA, MI, J0, J1, KS)                                                                                 \
    if (!ABL_M) {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                     \
        _Pragma("unroll") for (int j = J0; j < J1; ++j)                                                               \
            acc[MI + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb.get(j, KS), FA.get(i, KS), acc[MI + i][j], 0, 0, 0); }
#define MFMA_PAIR(FA, MI)                                                                                              \
    if (!ABL_M) { if (!NO_PRIO) __builtin_amdgcn_s_setprio(1);                                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                  \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                 \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
                acc[MI + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb.get(j, ks), FA.get(i, ks), acc[MI + i][j], 0, 0, 0); \
    if (!NO_PRIO) __builtin_amdgcn_s_setprio(0); }

    // (no unrolling AND no peeling: hipcc peeled the first K-tile - the `t == 0` conditions - and hoisted that copy's LDS
    // addresses out of the tile loop, where they were spilled to scratch: a dozen reloads, each behind `s_waitcnt vmcnt(0)`,
    // in the first K-tile of every output tile)
#pragma clang loop unroll(disable)
    for (int t = 0; t < nt; ++t) {
        const int b = t & 1;
        const int b3n = b3 == 0 ? 2 : b3 - 1;            // (t + 2) % 3
        const char* la = slotA(b, a_half);
        const char* lb = slotB(b3, b_half);
        // ---- ph1: pair 0 (set 0).  B fragments: n-tiles 0, 1 were requested in ph4 of the previous K-tile (below) and have
        // had its last 8 MFMAs to arrive; n-tiles 2, 3 are requested here and arrive under this phase's first 8 MFMAs.
        if (!EARLY && t + 1 < nt && !(t == 0 && a1_pre)) issueA(1, -1, t + 1, b ^ 1);
        // (the touch goes out AFTER every A request of tile t+2 - for the non-EARLY layouts A1 is requested just above - so that the
        // wait at the end of ph3, which is for those A pieces, can leave it in flight: vmcnt(5) instead of vmcnt(4))
        tch_prev = false;
        if (tch_q < tch_n && t == tch_next && t + 3 < nt) { touch(tch_q); ++tch_q; tch_next += tch_stride; tch_prev = true; }
        if (EARLYB && t + 2 < nt) issueB(0, 0, t + 2, b3n);
        if (!ABL_L) {
            fb.template load<2, 4, 0>(lb, b_off, lane);
            fb.template load<2, 4, 1>(lb, b_off, lane);
            WAIT_LGKM(2 * RB);                                  // ph4's requests (B n-tiles 0, 1 and A pair 0) are in
            if (!NO_PRIO) __builtin_amdgcn_s_setprio(1);
            MFMA_GROUP(fa0, 0, 0, 2, 0)
            MFMA_GROUP(fa0, 0, 0, 2, 1)
            fa1.template load<0, 2, 2>(la, 32, lane);
            WAIT_LGKM(RB + NRA);
            MFMA_GROUP(fa0, 0, 2, 4, 0)
            WAIT_LGKM(NRA);
            MFMA_GROUP(fa0, 0, 2, 4, 1)
            if (!NO_PRIO) __builtin_amdgcn_s_setprio(0);
        } else {
            MFMA_PAIR(fa0, 0)
        }
        // ---- ph2: pair 1 (set 1)   | prefetch pair 2 -> set 0
        if (t + 2 < nt) issueB(EARLYB ? 1 : 0, 0, t + 2, b3n);
        if (!ABL_L) fa0.template load<0, 2, 2>(la, 64, lane);
        WAIT_LGKM(NRA);
        MFMA_PAIR(fa1, 2)
        // (no barrier: the B slots were released by ph1's barrier, nothing new has to be visible yet)
        // ---- ph3: pair 2 (set 0)   | prefetch pair 3 -> set 1
        if (!EARLYB && t + 2 < nt) issueB(1, 0, t + 2, b3n);
        if (!ABL_L) fa1.template load<0, 2, 2>(la, 96, lane);
        WAIT_LGKM(NRA);
        MFMA_PAIR(fa0, 4)
        WAIT_LGKM(0);                                           // pair-3 reads done: the A slots of this buffer are dead
        // tile t+1 has landed (for t = 0 of a later tile of this workgroup, counted past the previous tile's stores: K-tile 1
        // was requested in full before them, and only the two B halves of K-tile 2 - 4 instructions - after them)
        // (a touch issued in this K-tile's ph1 is younger than the A pieces this wait is for: it stays in flight too)
        if (t + 2 < nt) {
            if (t == 0 && pend > 0) { const int w = min(63, 4 + pend); WAIT_VM_DYN(w); }
            else if (tch_prev) { WAIT_VM(5); }
            else { WAIT_VM(4); }
        } else { WAIT_VM(0); }
        if (!ABL_B) BARRIER();
        // ---- ph4: pair 3 (set 1)   | after its first 8 MFMAs (n-tiles 0, 1, both k-steps) those B registers are free: the B
        // fragments of n-tiles 0, 1 of K-tile t+1 (published by the barrier above) and A pair 0 of t+1 are requested under
        // the other 8 MFMAs - ph1(t+1) then starts on operands that are already there instead of waiting for its first read
        if (t + 2 < nt) { issueA(0, 0, t + 2, b); if (EARLY) issueA(1, 0, t + 2, b); }
        if (!ABL_L) {
            if (!NO_PRIO) __builtin_amdgcn_s_setprio(1);
            MFMA_GROUP(fa1, 6, 0, 2, 0)
            MFMA_GROUP(fa1, 6, 0, 2, 1)
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < nt) {
                fb.template load<0, 2, 2>(slotB(b3 == 2 ? 0 : b3 + 1, b_half), b_off, lane);
                fa0.template load<0, 2, 2>(slotA(b ^ 1, a_half), 0, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            MFMA_GROUP(fa1, 6, 2, 4, 0)
            MFMA_GROUP(fa1, 6, 2, 4, 1)
            if (!NO_PRIO) __builtin_amdgcn_s_setprio(0);
        } else {
            MFMA_PAIR(fa1, 6)
        }
        // (no barrier: ph1(t+1) only touches slots ph3's barrier has already released / published)
        b3 = b3 == 2 ? 0 : b3 + 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) { pa[h] += srcA.step; pb[h] += srcB.step; }
    }
#undef MFMA_GROUP
#undef MFMA_PAIR
    asm volatile("" ::"v"(junk));               // (the touches' destination register was reserved up to here)

    if (g.kx) k_extend<8>(g.xA, g.xB, g.kx, g.M, g.N, m0 + wr * 128, n0 + wc * 64, lane, acc);

    // next tile of this workgroup: its first K-tiles are requested before this tile's results are stored (every LDS read of
    // this tile is behind the last K-tile's barrier, so the slots are free)
    const int em0 = m0, en0 = n0;
    id += stride;
    const bool more = id < nwg;
    if (more) {
        coords(id, m0, n0); prologue(true); a1_pre = true;
        // ``pend`` counts epilogue operations issued AFTER these requests: nothing of the epilogue may move above them
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }

    Epi e;
    e.C = reinterpret_cast<OutT*>(g.C) + (size_t)bz * g.sC;
    e.R = g.R ? g.R + (size_t)bz * g.sR : nullptr;
    e.ldc = g.ldc; e.ldr = g.ldr; e.alpha = g.alpha; e.mode = g.epi_mode; e.aux_in = g.aux_in; e.aux_out = g.aux_out;
    e.ld_aux = g.ld_aux; e.M = g.M; e.N = g.N; e.p0 = g.epi_p0; e.p1 = g.epi_p1;
    const bool vec_ok = ((g.ldc & 3) == 0) && (!e.R || (g.ldr & 3) == 0);
    const bool full = em0 + 256 <= g.M && en0 + 256 <= g.N;      // no lane of this tile skips a memory operation
    bool swb = false;
    if constexpr (sizeof(OutT) == 2) swb = e.mode == EPI_SWIGLU_BWD;
    if (swb) {
        if constexpr (sizeof(OutT) == 2) epi_swiglu_bwd_block<8, 4>(e, em0 + wr * 128, en0 + wc * 64, lane, acc);
        pend = full ? 32 : 0;          // the block's 32 asm stores (its 32 gate/up loads are the compiler's: not counted)
    } else {
        pend = epi_block<OutT, 8>(e, vec_ok, em0 + wr * 128, en0 + wc * 64, lane, acc);
        if (!full) pend = 0;
    }
    if (!more) break;
  }
}

template <int TA, int TB, typename OutT>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(Gemm256Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256_body<TA, TB, OutT>(g, blockIdx.x, gridDim.x, blockIdx.z, smem);
}

// Two independent GEMMs in ONE launch: the dgrad (nn) and the wgrad (tn) of the same Linear layer, which share dY and
// nothing else.  Their tiles take different times (different K, different epilogues), and the work list interleaves them
// in blocks of 8 workgroups (one per XCD) in the ratio of their tile counts - so the chip stops marching in lockstep:
// while some CUs sit in an HBM-bound epilogue (the SwiGLU-backward one reads and writes 512 KB per tile) the others are in
// their MFMA main loops, instead of all 256 CUs hitting HBM together and then all idling the memory system together.
// It also fills the chip for wgrads with few tiles (attention projections) without fp32 split-K slabs.
struct Gemm256Pair {
    Gemm256Args a, b;         // a: nn (dgrad), b: tn (wgrad); both bf16 out
    int na, nb;               // tiles of each
    int ra, rb;               // interleave: ra workgroups of a, then rb of b, ... (multiples of 8)
};

__global__ __launch_bounds__(512, 2) void gemm256pair_kernel(Gemm256Pair p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // list position -> (kind, index within kind); once one kind runs out the rest of the list is the other kind
    const int per = p.ra + p.rb;
    const int full = min(p.na / p.ra, p.nb / p.rb);            // whole interleave periods
    const int pos = blockIdx.x;
    int kind, j;
    if (pos < full * per) {
        const int q = pos / per, r = pos % per;
        if (r < p.ra) { kind = 0; j = q * p.ra + r; } else { kind = 1; j = q * p.rb + (r - p.ra); }
    } else {
        const int rest = pos - full * per, left_a = p.na - full * p.ra;
        if (rest < left_a) { kind = 0; j = full * p.ra + rest; } else { kind = 1; j = full * p.rb + (rest - left_a); }
    }
    if (kind == 0) gemm256_body<0, 1, bf16_t>(p.a, j, 1 << 30, 0, smem);
    else gemm256_body<1, 1, bf16_t>(p.b, j, 1 << 30, 0, smem);
}


template <int TA, int TB>
int launch256(const Gemm256Args& g, int out_f32, int batch, hipStream_t stream) {
    // persistent: at most one workgroup per CU, each walking tiles id, id + 256, ... (csm_set_gemm256_persistent(0): one tile each)
    const int tiles = g.tiles_m * g.tiles_n;
    dim3 grid(g_persistent && tiles > 256 ? 256 : tiles, 1, batch), block(512);
    const size_t lds = 10 * HALF;   // 160 KiB: the whole LDS of a CU
    static bool done[2] = {false, false};
    g_last_gemm_kernel = CSM_KNAME256(TA, TB, out_f32);
    if (out_f32) {
        auto k = gemm256p_kernel<TA, TB, float>;
        if (!done[1]) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done[1] = true; }
        hipLaunchKernelGGL(k, grid, block, lds, stream, g);
    } else {
        auto k = gemm256p_kernel<TA, TB, bf16_t>;
        if (!done[0]) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done[0] = true; }
        hipLaunchKernelGGL(k, grid, block, lds, stream, g);
    }
    CSM_CHECK_LAUNCH("csm_gemm_bf16(256)");
    return 0;
}

// Two weight-gradient products (tn, bf16 out) in one launch: neither fills the chip alone (attention output projection:
// 64 tiles, fused q|k|v projection: 96 tiles on 256 CUs), together they are one round of 160 tiles - instead of a
// 128x128-tile launch at 0.7-0.8 PF/s plus fp32 split-K slabs and a column-sum pass.
struct Gemm256Two { Gemm256Args a, b; int na; };
__global__ __launch_bounds__(512, 2) void gemm256two_tn_kernel(Gemm256Two p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int id = blockIdx.x;
    // longest first is moot (same K); keep each problem's tiles contiguous so that an XCD's L2 sees one operand set at a time
    if (id < p.na) gemm256_body<1, 1, bf16_t>(p.a, id, 1 << 30, 0, smem);
    else gemm256_body<1, 1, bf16_t>(p.b, id - p.na, 1 << 30, 0, smem);
}

}  // namespace

// called by csm_gemm_bf16 (gemm.hip) when the tile heuristic picks the 256x256 kernel; same argument meaning
int csm_gemm256_launch(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb, int ldc,
                       int ldr, int transA, int transB, int out_f32, float alpha, int batch, long long sA, long long sB,
                       long long sC, long long sR, int epi_mode, const void* aux_in, void* aux_out, int ld_aux,
                       hipStream_t stream, int epi_p0, int epi_p1, const void* xA, const void* xB, int kx) {
    Gemm256Args g;
    g.epi_p0 = epi_p0; g.epi_p1 = epi_p1;
    g.xA = (const bf16_t*)xA; g.xB = (const bf16_t*)xB; g.kx = kx; g.touch = g_gemm_touch;
    g.epi_mode = epi_mode; g.aux_in = (const bf16_t*)aux_in; g.aux_out = (bf16_t*)aux_out; g.ld_aux = ld_aux;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.R = (const bf16_t*)R;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
    g.sA = sA; g.sB = sB; g.sC = sC; g.sR = sR; g.alpha = alpha;
    g.tiles_m = (M + 255) / 256; g.tiles_n = (N + 255) / 256;
    if (!transA && !transB) return launch256<0, 0>(g, out_f32, batch, stream);
    if (!transA && transB) return launch256<0, 1>(g, out_f32, batch, stream);
    if (transA && transB) return launch256<1, 1>(g, out_f32, batch, stream);
    return launch256<1, 0>(g, out_f32, batch, stream);
}

static void fill256(Gemm256Args& g, const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb, int ldc,
                    int ldr, float alpha, int epi_mode, const void* aux_in, void* aux_out, int ld_aux) {
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.R = (const bf16_t*)R;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
    g.sA = g.sB = g.sC = g.sR = 0; g.alpha = alpha;
    g.xA = g.xB = nullptr; g.kx = 0; g.touch = g_gemm_touch;
    g.tiles_m = (M + 255) / 256; g.tiles_n = (N + 255) / 256;
    g.epi_mode = epi_mode; g.aux_in = (const bf16_t*)aux_in; g.aux_out = (bf16_t*)aux_out; g.ld_aux = ld_aux;
    g.epi_p0 = g.epi_p1 = 0;
}

// dX[M][Kin] = dY[M][Nout] W[Nout][Kin] (epilogue dx_epi: 0 none, 2 SwiGLU backward with aux gu / output d(gu))  together with
// dW[Nout][Kin] (+)= alpha_w * dY^T X.   Shapes must suit the 256x256 kernel (contractions multiples of 64).
int csm_gemm256_pair_launch(const void* dY, const void* W, void* dX, int M, int Nout, int Kin, int ld_dy, int ldw, int ld_dx,
                            int dx_epi, const void* aux_in, int ld_aux,
                            const void* X, int ldx, void* dW, int ld_dw, int accumulate, float alpha_w, hipStream_t stream) {
    Gemm256Pair p;
    // a: C = dY . W  (A = dY [M][Nout] K-contiguous, B = W read as [K = Nout][N = Kin])
    fill256(p.a, dY, W, dX, nullptr, M, Kin, Nout, ld_dy, ldw, ld_dx, 0, 1.f, dx_epi, aux_in, nullptr, ld_aux);
    // b: C = dY^T . X  (A = dY read as [K = M][M' = Nout], B = X read as [K = M][N = Kin])
    fill256(p.b, dY, X, dW, accumulate ? dW : nullptr, Nout, Kin, M, ld_dy, ldx, ld_dw, ld_dw, alpha_w, 0, nullptr, nullptr, 0);
    p.na = p.a.tiles_m * p.a.tiles_n;
    p.nb = p.b.tiles_m * p.b.tiles_n;
    // interleave in blocks of 8 in (roughly) the ratio of the tile counts, longest-running kind first inside a period
    int ra = 8, rb = 8;
    if (p.na >= 2 * p.nb) ra = 8 * (int)((p.na + p.nb / 2) / p.nb > 8 ? 8 : (p.na + p.nb / 2) / p.nb);
    else if (p.nb >= 2 * p.na) rb = 8 * (int)((p.nb + p.na / 2) / p.na > 8 ? 8 : (p.nb + p.na / 2) / p.na);
    p.ra = ra; p.rb = rb;
    static bool done = false;
    const size_t lds = 10 * HALF;
    if (!done) { (void)hipFuncSetAttribute((const void*)gemm256pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
    g_last_gemm_kernel = "gemm256pair_kernel";
    hipLaunchKernelGGL(gemm256pair_kernel, dim3(p.na + p.nb), dim3(512), lds, stream, p);
    CSM_CHECK_LAUNCH("csm_gemm_bf16_dgrad_wgrad");
    return 0;
}

// dW1[N1][K1] (+)= alpha dY1[M][N1]^T X1[M][K1]  and  dW2[N2][K2] (+)= alpha dY2[M][N2]^T X2[M][K2]  in one launch
int csm_gemm256_two_wgrad_launch(const void* dY1, const void* X1, void* dW1, int N1, int K1, int ld_dy1, int ldx1, int ld_dw1,
                                 const void* dY2, const void* X2, void* dW2, int N2, int K2, int ld_dy2, int ldx2, int ld_dw2,
                                 int M, int accumulate, float alpha, hipStream_t stream) {
    Gemm256Two p;
    fill256(p.a, dY1, X1, dW1, accumulate ? dW1 : nullptr, N1, K1, M, ld_dy1, ldx1, ld_dw1, ld_dw1, alpha, 0, nullptr, nullptr, 0);
    fill256(p.b, dY2, X2, dW2, accumulate ? dW2 : nullptr, N2, K2, M, ld_dy2, ldx2, ld_dw2, ld_dw2, alpha, 0, nullptr, nullptr, 0);
    p.na = p.a.tiles_m * p.a.tiles_n;
    const int nb = p.b.tiles_m * p.b.tiles_n;
    static bool done = false;
    const size_t lds = 10 * HALF;
    if (!done) { (void)hipFuncSetAttribute((const void*)gemm256two_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
    g_last_gemm_kernel = "gemm256two_tn_kernel";
    hipLaunchKernelGGL(gemm256two_tn_kernel, dim3(p.na + nb), dim3(512), lds, stream, p);
    CSM_CHECK_LAUNCH("csm_gemm_bf16_two_wgrad");
    return 0;
}
