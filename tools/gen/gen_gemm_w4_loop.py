#!/usr/bin/env python3
"""Generates csrc/gemm256w4_loop.inc: the hand-scheduled K loop of the 4-wave 256x256 bf16 GEMM (gemm256w4.hip) as ONE
inline-asm block per operand-layout pair.

Why generated assembly: a 128 x 128 wave tile (four waves per workgroup, one per SIMD) needs 256 accumulator registers; hipcc's
register allocator cannot keep them in AGPRs across a K loop written with the MFMA builtin (round 2: ~650 accvgpr moves and
170 scratch accesses per two K-tiles).  Here the accumulators are a[0:255], owned by the asm block; the fragments live in
v[64:191] (two sets, ping-pong per 32-deep k-step), and every LDS read, LDS-DMA request, wait and the one barrier per K-tile
has a fixed slot between the MFMAs.

Schedule of K-tile t (tile = 64 deep = 2 k-steps; A double-buffered, B triple-buffered in LDS, as in gemm256.hip):
  ks0: 64 MFMAs on fragment set P | reads (t, ks1) -> set Q | LDS-DMA of B(t+2) (its stage held tile t-1: free)
  s_waitcnt lgkmcnt(0); s_waitcnt vmcnt(8 | 0)  [my pieces of tile t+1 have landed]; s_barrier  [tile t+1 published, tile t's A stage free]
  ks1: 64 MFMAs on Q | reads (t+1, ks0) -> P | LDS-DMA of A(t+2) into tile t's A stage
Operands (see ASM_OPERANDS in gemm256w4.hip): %0 A lane address (LDS, stage 0, my half, k-step 0), %1 B lane address,
%2 B tr key (T=1) / unused, %3..%18 LDS-DMA lane offsets [A0 x4, A1 x4, B0 x4, B1 x4], %19 A base ptr, %20 B base ptr,
%21 A step, %22 B step, %23 nt, %24 wave * 4096.
"""
import sys

# ---- fixed registers ------------------------------------------------------------------------------------------------
PA, PB, QA, QB = 64, 96, 128, 160          # fragment sets: A tiles i at +4i, B tiles j at +4j
VB = 192                                    # v[192:199] B per-n-tile lane addresses (T=1), stage 0;  v192 for T=0
VBC = 200                                   # v[200:207] the same + current stage
VAC0, VAC1 = 208, 209                       # A current-stage lane addresses, k-step 0 / 1
VBT0, VBT1 = 210, 211                       # B (T=0) current-stage lane addresses, k-step 0 / 1
VA1 = 212                                   # A stage-0 k-step 1 address (T=0)
VB1 = 213                                   # B stage-0 k-step 1 address (T=0)
VAT = 214                                   # v[214:221] A per-m-tile lane addresses (T=1), stage 0
VATC = 222                                  # v[222:229] the same + current stage
S_RA, S_RB = 40, 44                         # buffer resource descriptors (4 SGPRs each)
S_STA, S_STB = 48, 50                       # steps (64-bit)
S_T, S_NT, S_SLA, S_SLB, S_TMP, S_WOFF, S_B3, S_T2 = 52, 53, 54, 55, 56, 57, 58, 59
S_WOFFB = 68                                # this wave's LDS offset inside a B half-tile (= S_WOFF for 4 pieces per wave; wave * 3072 for 3)
S_LA, S_LB = 60, 61                         # LDS byte offsets of the CURRENT tile's A / B stage (for the reads)


NJ = 8          # n-tiles (16 columns each) of a wave's block: 8 (256-column tiles) or 6 (192-column tiles, K-contiguous B only)


def frag_reads(T, dst, cur0, cur1, curt, ks, n=8):
    """LDS reads of n tiles' fragments for k-step ks into registers dst + 4 * tile."""
    out = []
    for i in range(n):
        r = dst + 4 * i
        if T == 0:
            out.append(f"ds_read_b128 v[{r}:{r + 3}], v{cur1 if ks else cur0} offset:{i * 2048}")
        else:
            out.append(f"ds_read_b64_tr_b16 v[{r}:{r + 1}], v{curt + i} offset:{ks * 8192}")
            out.append(f"ds_read_b64_tr_b16 v[{r + 2}:{r + 3}], v{curt + i} offset:{ks * 8192 + 1024}")
    return out


def dma_half(which, h, op_base):
    """my pieces of half-tile h of operand `which` ('A' | 'B') for the tile the resource descriptor points at: 4 of the 16 of a
    128-row half; 3 of the 12 of a 96-row B half (NJ = 6: the wave's LDS offset inside the half is then wave * 3072, s[S_WOFFB])"""
    out = []
    rs = S_RA if which == "A" else S_RB
    sl = S_SLA if which == "A" else S_SLB
    for i in range(4 if which == "A" else NJ // 2):
        out.append(f"s_add_u32 m0, s{sl}, {h * 16384 + i * 1024}")
        out.append("s_nop 0")
        out.append(f"buffer_load_dwordx4 %{op_base + 4 * h + i}, s[{rs}:{rs + 3}], 0 offen lds")
    return out


def mfmas(sa, sb, zero=False):
    """64 MFMAs of one k-step; ``zero``: the accumulators START here (C = 0: the first k-step of an output tile)"""
    out = []
    for i in range(8):
        for j in range(NJ):
            c = (i * 8 + j) * 4
            src_c = "0" if zero else f"a[{c}:{c + 3}]"
            out.append(f"v_mfma_f32_16x16x32_bf16 a[{c}:{c + 3}], v[{sb + 4 * j}:{sb + 4 * j + 3}], v[{sa + 4 * i}:{sa + 4 * i + 3}], {src_c}")
    return out


def interleave(mf, side, first=0, last=None):
    """side instructions spread evenly over MFMAs [first, last) of the list (the first one after MFMA `first`)"""
    out = []
    last = len(mf) if last is None else last
    n, m = last - first, len(side)
    k = 0
    for idx, ins in enumerate(mf):
        out.append(ins)
        if first <= idx < last:
            want = (idx - first + 1) * m // n
            while k < want:
                out.append(side[k]); k += 1
    out += side[k:]
    return out


def merge(a, b):
    """two instruction streams already interleaved with the same MFMA list: not used"""
    raise NotImplementedError


def weave(mf, streams):
    """several side streams, each (instructions, first MFMA, last MFMA), woven between the MFMAs of one phase"""
    slots = [[] for _ in mf]
    for side, first, last in streams:
        n, m = last - first, len(side)
        k = 0
        for idx in range(first, last):
            want = (idx - first + 1) * m // n
            while k < want:
                slots[idx].append(side[k]); k += 1
        slots[last - 1] += side[k:]
    out = []
    for ins, sl in zip(mf, slots):
        out.append(ins)
        out += sl
    return out


def set_cur(TA, TB):
    """lane addresses of the current tile's stages (s60 / s61 = LDS byte offsets of its A / B stage)"""
    out = []
    if TA == 0:
        out += [f"v_add_u32 v{VAC0}, s{S_LA}, %0", f"v_add_u32 v{VAC1}, s{S_LA}, v{VA1}"]
    else:
        out += [f"v_add_u32 v{VATC + i}, s{S_LA}, v{VAT + i}" for i in range(8)]
    if TB == 0:
        out += [f"v_add_u32 v{VBT0}, s{S_LB}, %1", f"v_add_u32 v{VBT1}, s{S_LB}, v{VB1}"]
    else:
        out += [f"v_add_u32 v{VBC + j}, s{S_LB}, v{VB + j}" for j in range(8)]
    return out


def next_stage_offsets():
    """s60 / s61 -> stage of tile t+1 (A alternates 0 / 32768; B cycles 65536 + {0, 32768, 65536} by s58 = (t+1) % 3)"""
    return [f"s_xor_b32 s{S_LA}, s{S_LA}, 0x8000",
            f"s_add_u32 s{S_B3}, s{S_B3}, 1", f"s_cmp_eq_u32 s{S_B3}, 3", f"s_cselect_b32 s{S_B3}, 0, s{S_B3}",
            f"s_lshl_b32 s{S_LB}, s{S_B3}, 15", f"s_add_u32 s{S_LB}, s{S_LB}, 0x10000"]


ADV = []
for _rs, _st in ((S_RA, S_STA), (S_RB, S_STB)):
    ADV += [f"s_add_u32 s{_rs}, s{_rs}, s{_st}", f"s_addc_u32 s{_rs + 1}, s{_rs + 1}, s{_st + 1}"]

# touch state (epilogue-read prefetch): s62 countdown, s63 period - 1, s64 touches left, s[66:67] byte stride; v230 junk, v[232:233] address
S_TCD, S_TPER, S_TLEFT, S_TFLAG, S_TSTR = 62, 63, 64, 65, 66
V_JUNK, V_TADDR = 230, 232


def set_records(tile_reg_or_imm):
    """descriptors get -1 records when the K-tile exists, 0 (every request dropped by the range check) when it does not"""
    return [f"s_cmp_lt_u32 {tile_reg_or_imm}, s{S_NT}", f"s_cselect_b32 s{S_TMP}, -1, 0",
            f"s_mov_b32 s{S_RA + 2}, s{S_TMP}", f"s_mov_b32 s{S_RB + 2}, s{S_TMP}"]


def descriptors(opA, opB, opStA, opStB, opNt, opWoff):
    L = []
    e = L.append
    e(f"s_mov_b64 s[{S_RA}:{S_RA + 1}], %{opA}"); e(f"s_mov_b32 s{S_RA + 2}, -1"); e(f"s_mov_b32 s{S_RA + 3}, 0x00020000")
    e(f"s_mov_b64 s[{S_RB}:{S_RB + 1}], %{opB}"); e(f"s_mov_b32 s{S_RB + 2}, -1"); e(f"s_mov_b32 s{S_RB + 3}, 0x00020000")
    e(f"s_and_b32 s{S_RA + 1}, s{S_RA + 1}, 0xffff"); e(f"s_and_b32 s{S_RB + 1}, s{S_RB + 1}, 0xffff")
    e(f"s_mov_b64 s[{S_STA}:{S_STA + 1}], %{opStA}"); e(f"s_mov_b64 s[{S_STB}:{S_STB + 1}], %{opStB}")
    e(f"s_mov_b32 s{S_NT}, %{opNt}"); e(f"s_mov_b32 s{S_WOFF}, %{opWoff}")
    if NJ == 8:
        e(f"s_mov_b32 s{S_WOFFB}, s{S_WOFF}")
    else:       # operand = sbase + wave * 4096 (4 pieces of 1 KiB per wave and half); with 3 pieces: sbase + wave * 3072
        e(f"s_and_b32 s{S_TMP}, s{S_WOFF}, 0x3fff"); e(f"s_lshr_b32 s{S_TMP}, s{S_TMP}, 2")      # wave * 1024  (sbase is a multiple of 16 KiB)
        e(f"s_sub_u32 s{S_WOFFB}, s{S_WOFF}, s{S_TMP}")
    return L


def gen_pro():
    """The first two K-tiles of an output tile (A stage 0 / 1, B stage 0 / 1): requests only, no wait.  Operands: %0..%15 LDS-DMA
    lane offsets [A0 x4, A1 x4, B0 x4, B1 x4], %16 A base, %17 B base, %18 A step, %19 B step, %20 nt, %21 wave LDS offset."""
    L = descriptors(16, 17, 18, 19, 20, 21)
    e = L.append
    e(f"s_mov_b32 s{S_SLA}, s{S_WOFF}"); e(f"s_add_u32 s{S_SLB}, s{S_WOFFB}, 0x10000")
    L += dma_half("A", 0, 0) + dma_half("A", 1, 0) + dma_half("B", 0, 8) + dma_half("B", 1, 8)
    L += ADV
    L += set_records("1")
    e(f"s_add_u32 s{S_SLA}, s{S_WOFF}, 0x8000"); e(f"s_add_u32 s{S_SLB}, s{S_WOFFB}, 0x18000")
    L += dma_half("A", 0, 0) + dma_half("A", 1, 0) + dma_half("B", 0, 8) + dma_half("B", 1, 8)
    return L


def gen(TA, TB):
    """The K loop of one output tile whose first two K-tiles have been requested (gen_pro) and whose K-tile 0 has LANDED (the
    caller waits: the count depends on what it issued since).  Operands: %0 A lane address, %1 B lane address, %2 tr key,
    %3..%18 LDS-DMA lane offsets, %19 A base, %20 B base, %21 A step, %22 B step, %23 nt, %24 wave LDS offset, %25 touch address
    (64-bit per lane), %26 touch stride (64-bit), %27 touches, %28 K-tiles between touches - 1."""
    L = descriptors(19, 20, 21, 22, 23, 24)
    e = L.append
    L += ADV + ADV                                                                 # descriptors point at K-tile 2
    if TA == 0:
        e(f"v_xor_b32 v{VA1}, 64, %0")
    else:
        for i in range(8):
            e(f"v_xor_b32 v{VAT + i}, {i}, %2"); e(f"v_lshl_add_u32 v{VAT + i}, v{VAT + i}, 5, %0")
    if TB == 0:
        e(f"v_xor_b32 v{VB1}, 64, %1")
    else:
        for j in range(8):
            e(f"v_xor_b32 v{VB + j}, {j}, %2"); e(f"v_lshl_add_u32 v{VB + j}, v{VB + j}, 5, %1")
    e(f"v_lshl_add_u64 v[{V_TADDR}:{V_TADDR + 1}], %25, 0, 0")
    e(f"s_mov_b64 s[{S_TSTR}:{S_TSTR + 1}], %26"); e(f"s_mov_b32 s{S_TLEFT}, %27"); e(f"s_mov_b32 s{S_TPER}, %28"); e(f"s_mov_b32 s{S_TCD}, 0"); e(f"s_mov_b32 s{S_TFLAG}, 0")
    # (no zeroing of the 256 accumulators: the first k-step of the tile is a peeled copy of ks0 whose MFMAs take C = 0)
    # The loop is branch-free but for the touches: a request for a K-tile past the last one goes through a descriptor with ZERO
    # records (the range check drops it: no memory traffic, and it still counts in vmcnt), a fragment read past the last tile
    # returns stale bytes nobody multiplies - so every wave issues the same instructions every trip and the waits are constants.
    e("s_barrier")
    e(f"s_mov_b32 s{S_T}, 0"); e(f"s_mov_b32 s{S_B3}, 0"); e(f"s_mov_b32 s{S_LA}, 0"); e(f"s_mov_b32 s{S_LB}, 0x10000")
    L += set_cur(TA, TB)
    L += frag_reads(TA, PA, VAC0, VAC1, VATC, 0) + frag_reads(TB, PB, VBT0, VBT1, VBC, 0, NJ)
    e("s_waitcnt lgkmcnt(0)")
    def ks0(zero):
        K0 = []
        k = K0.append
        k(f"s_add_u32 s{S_T2}, s{S_T}, 2")
        K0 += set_records(f"s{S_T2}")
        # B stage of tile t+2 = the stage tile t-1 used = (s58 + 2) % 3
        k(f"s_add_u32 s{S_TMP}, s{S_B3}, 2"); k(f"s_cmp_ge_u32 s{S_TMP}, 3"); k(f"s_cselect_b32 s{S_SLB}, {-3 & 0xffffffff}, 0")
        k(f"s_add_u32 s{S_SLB}, s{S_SLB}, s{S_TMP}")
        k(f"s_lshl_b32 s{S_SLB}, s{S_SLB}, 15"); k(f"s_add_u32 s{S_SLB}, s{S_SLB}, 0x10000"); k(f"s_add_u32 s{S_SLB}, s{S_SLB}, s{S_WOFFB}")
        # fragments (t, ks1) -> Q within the first 44 MFMAs, the 8 B(t+2) requests spread over all 64
        reads_q = frag_reads(TA, QA, VAC0, VAC1, VATC, 1) + frag_reads(TB, QB, VBT0, VBT1, VBC, 1, NJ)
        dmaB = dma_half("B", 0, 11) + dma_half("B", 1, 11)
        return K0 + weave(mfmas(PA, PB, zero), [(reads_q, 0, 44 * NJ // 8), (dmaB, 2, 8 * NJ)])
    # ---- K-tile 0's first k-step, peeled: its MFMAs start the accumulators (C = 0); then into the loop at its mid-tile point
    L += ks0(True)
    e("s_branch 5f")
    # ---- the loop: one K-tile per trip
    e("1:")
    L += ks0(False)
    e("5:")
    # my pieces of tile t+1 have landed: everything but the 8 B(t+2) requests of this phase - and, when the previous K-tile
    # ended with a touch (s65), that touch, which is younger than every piece this wait is for
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_cmp_eq_u32 s{S_TFLAG}, 0"); e("s_cbranch_scc1 8f")
    e(f"s_waitcnt vmcnt({NJ + 1})"); e(f"s_mov_b32 s{S_TFLAG}, 0"); e("s_branch 9f")
    e("8:"); e(f"s_waitcnt vmcnt({NJ})")
    e("9:"); e("s_barrier")
    # ---- ks1: fragments of (t+1, ks0) -> P and A(t+2) into the stage tile t just left
    e(f"s_add_u32 s{S_SLA}, s{S_LA}, s{S_WOFF}")                                     # A stage of tile t (free now) for the t+2 requests
    L += next_stage_offsets()
    reads_p = set_cur(TA, TB) + frag_reads(TA, PA, VAC0, VAC1, VATC, 0) + frag_reads(TB, PB, VBT0, VBT1, VBC, 0, NJ)
    dmaA = dma_half("A", 0, 3) + dma_half("A", 1, 3) + ADV
    L += weave(mfmas(QA, QB), [(reads_p, 0, 44 * NJ // 8), (dmaA, 2, 8 * NJ)])
    # ---- a touch (epilogue-read prefetch) every (s63 + 1) K-tiles while any are left, AFTER this K-tile's last A request: it is
    # then younger than everything the next K-tile's wait is for and stays in flight across it (vmcnt(9) there), i.e. it has
    # two K-tiles to come back from HBM before a wait insists on it
    e(f"s_sub_u32 s{S_TCD}, s{S_TCD}, 1"); e("s_cbranch_scc0 7f")
    e(f"s_mov_b32 s{S_TCD}, s{S_TPER}")
    e(f"s_cmp_eq_u32 s{S_TLEFT}, 0"); e("s_cbranch_scc1 7f")
    e(f"s_sub_u32 s{S_TLEFT}, s{S_TLEFT}, 1")
    e(f"global_load_dword v{V_JUNK}, v[{V_TADDR}:{V_TADDR + 1}], off")
    e(f"v_lshl_add_u64 v[{V_TADDR}:{V_TADDR + 1}], s[{S_TSTR}:{S_TSTR + 1}], 0, v[{V_TADDR}:{V_TADDR + 1}]")
    e(f"s_mov_b32 s{S_TFLAG}, 1")
    e("7:")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_add_u32 s{S_T}, s{S_T}, 1"); e(f"s_cmp_lt_u32 s{S_T}, s{S_NT}"); e("s_cbranch_scc1 1b")
    e("s_waitcnt vmcnt(0)")                                                        # (dropped requests, touches: nothing of mine stays in flight)
    e("s_nop 7"); e("s_nop 7"); e("s_nop 3")                                       # last MFMA results before v_accvgpr_read
    return [x for x in L if x is not None]


def main(path):
    global NJ
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen/gen_gemm_w4_loop.py - do not edit.  The K loop of gemm256w4.hip as inline-asm text,\n"
                "// one macro per operand-layout pair (TA, TB: 0 = K-contiguous LDS image, 1 = K-strided image).\n")
        for TA in (0, 1):
            for TB in (0, 1):
                body = gen(TA, TB)
                f.write(f"#define CSM_W4_LOOP_{TA}{TB} \\\n")
                for ins in body:
                    f.write(f'    "{ins}\\n\\t" \\\n')
                f.write('    ""\n')
        f.write("#define CSM_W4_ZERO \\\n")
        for n in range(256):
            f.write(f'    "v_accvgpr_write_b32 a{n}, 0\\n\\t" \\\n')
        f.write('    ""\n')
        f.write("#define CSM_W4_ZERO_CLOBBERS " + ", ".join(f'"a{n}"' for n in range(256)) + "\n")
        f.write("#define CSM_W4_PRO \\\n")
        for ins in gen_pro():
            f.write(f'    "{ins}\\n\\t" \\\n')
        f.write('    ""\n')
        # 256 x 192 tiles (round 4): the wave's block is 8 x 6 accumulators, the B half-tile 96 rows = 3 LDS-DMA pieces per wave;
        # K-contiguous B only (the fused q|k|v projections: 384 tiles of 256 x 256 are 1.5 rounds of the 256 CUs, 512 of 256 x 192 two)
        NJ = 6
        for TA in (0, 1):
            f.write(f"#define CSM_W4N6_LOOP_{TA}0 \\\n")
            for ins in gen(TA, 0):
                f.write(f'    "{ins}\\n\\t" \\\n')
            f.write('    ""\n')
        f.write("#define CSM_W4N6_PRO \\\n")
        for ins in gen_pro():
            f.write(f'    "{ins}\\n\\t" \\\n')
        f.write('    ""\n')
        NJ = 8
        # one 32-deep k-step whose fragments arrive as operands (K-extension: a LoRA group's columns after the main loop):
        # %0..%7 the wave's 8 A-row fragments, %8..%15 its 8 B-row fragments
        f.write("#define CSM_W4_KEXT \\\n")
        for i in range(8):
            for j in range(8):
                c = (i * 8 + j) * 4
                f.write(f'    "v_mfma_f32_16x16x32_bf16 a[{c}:{c + 3}], %{8 + j}, %{i}, a[{c}:{c + 3}]\\n\\t" \\\n')
        f.write('    "s_nop 7\\n\\ts_nop 7\\n\\t" \\\n    ""\n')
        f.write("#define CSM_W4_KEXT_CLOBBERS " + ", ".join(f'"a{n}"' for n in range(256)) + "\n")
        f.write("#define CSM_W4_CLOBBERS " + ", ".join([f'"v{n}"' for n in range(64, 234)] + [f'"a{n}"' for n in range(256)] +
                                                      [f'"s{n}"' for n in range(40, 70)] + ['"scc"', '"vcc"', '"memory"']) + "\n")
        f.write("#define CSM_W4_PRO_CLOBBERS " + ", ".join([f'"s{n}"' for n in range(40, 62)] + ['"s68"', '"s69"', '"scc"', '"memory"']) + "\n")


if __name__ == "__main__":
    main(sys.argv[1])
