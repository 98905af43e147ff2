#!/usr/bin/env python3
"""Generates csrc/attn64_dq_loop.inc: the key loop of the head_dim-64 dQ attention-backward kernel (attention64_asm.hip,
attn64_dq_asm_kernel) as inline-asm text with hand-allocated registers - the dK/dV generator's construction
(gen_attn64_dkv_loop.py) with the roles of queries and keys exchanged.

One wave owns 128 queries (four 32-query tiles, the query on the lane) of one (batch, query head) and walks the 32-key tiles
from the diagonal DOWN to key 0, so that the four tiles that need the causal mask come first.  The four waves of a workgroup
are the four query heads of one GQA group on the same 128-query block: they need the SAME K / V tiles, so the workgroup keeps
ONE eight-stage LDS ring (K image + V image per stage) and every wave fetches a quarter of each tile (one 1-KiB piece of K, one
of V) by LDS-DMA, five tiles ahead; one s_barrier per tile - behind the wave's counted wait for its pieces of tile t+2 -
publishes that tile.  No cross-wave sum.  (First versions, measured: private rings per wave 130 us, shared ring with 64
queries per wave 127 us against 137 for the compiler-scheduled kernel - at 64 queries a tile's 24 MFMAs = 768 cycles sit under
~1300 cycles of exp2 / pack / LDS / LDS-DMA issue; 128 queries per wave halve everything but the exp2 work per MFMA.)

  a[0:127]   dQ^T accumulators, tile (dt, qt) at 16 (4 dt + qt)          (dt: d half, qt: query tile)
  a[128:191] Q^T B-operands [qt][ks] (the wave's 128 queries, loaded once), a[192:255] dO^T B-operands
  v[32:95]   S^T tiles [qt] -> P in place, v[96:159] dP^T tiles [qt] -> dS in place
  v[160:191] packed bf16 dS B-operands [qt][s]  (single-buffered: written behind the MFMAs that read the previous tile's)
  v[192:223] K / V row fragments of the NEXT tile [ks], v[224:239] K^T fragments [dt][s] of the CURRENT tile

Software pipeline, iteration t (48 MFMAs; tile = 32 keys x 128 queries):
  MFMA  0..31  S^T(t+1), dP^T(t+1), query tile by query tile
  MFMA 32..47  dQ^T += K^T dS(t), query tile by query tile
  VALU         exp2 / -delta / multiply of tile t+1: query tile 0 in slots 10..19, 1 in 20..29, 2 in 30..41, 3 in 42..47 and
               0..7 of the next iteration; the bf16 packs of query tile q behind MFMA 35 + 4 q
  LDS          K^T fragments of tile t (slots 1..11), K / V row fragments of tile t+2 (34..47, behind the barrier)
  LDS-DMA      this wave's two pieces of tile t+5 (slots 34..39)
"""
import os
import sys

ABLATE = int(os.environ.get("CSM_A64DQ_ABLATE", "0"))     # bit0 no LDS-DMA in the loop, bit1 no VALU, bit2 no LDS reads, bit3 no MFMA, bit4 no barrier
STAGE = 8192                     # K image 4096 + V image 4096
NSTAGE = 8
LEAD = 5                         # tiles between the LDS-DMA requests and the tile whose dQ update runs
NQ = 4
NSLOT = 48


def A_DQ(dt, qt): return 16 * (4 * dt + qt)
def A_QF(qt, ks): return 128 + 16 * qt + 4 * ks
def A_OF(qt, ks): return 192 + 16 * qt + 4 * ks
def V_SC(qt): return 32 + 16 * qt
def V_DP(qt): return 96 + 16 * qt
def V_D(qt, s): return 160 + 8 * qt + 4 * s
def V_KR(ks): return 192 + 4 * ks
def V_VR(ks): return 208 + 4 * ks
def V_KT(dt, s): return 224 + 4 * (2 * dt + s)
V_NINF, V_M1 = 240, 241
V_R = 242                        # v[242:245] row-fragment lane addresses [ks]
V_T = 246                        # v[246:249] transposed-fragment lane addresses [dt][u]
V_SG = 250                       # v[250:253] staging row-fragment lane addresses [ks]; v[254:255] LDS-DMA lane offsets of pieces 2, 3
LAST_V = 255
S_RK, S_RV = 40, 44              # buffer descriptors
S_T, S_N, S_WB, S_KST, S_C2, S_KOFF, S_TMP, S_T4, S_OST = 48, 49, 50, 51, 52, 53, 54, 55, 60
S_QB, S_OB = 56, 58              # (descriptors of the Q / dO / O tiles of the prologue: s[56:59], s[64:67], s[68:71])
S_RQ, S_RO, S_RP = 56, 64, 68
LAST_S = 71

OP_ROFF, OP_TOFF, OP_DK0, OP_SG, OP_DQ0, OP_ND0, OP_ND1, OP_ND2, OP_ND3, OP_NL0, OP_NL1, OP_NL2, OP_NL3, OP_M0 = range(14)
OP_KB, OP_VB, OP_QB, OP_OB, OP_N, OP_WB, OP_KST, OP_C2, OP_OST, OP_DQ1, OP_DO0, OP_DO1, OP_PB, OP_SGB = range(14, 28)
# OP_SG: staging lane address (row r, chunk h); OP_DQ0/1, OP_DO0/1: LDS-DMA lane offsets of pieces 0 / 1 of a 32-row tile with the
# Q (qkv) / dO, O leading dimension; OP_QB / OP_OB / OP_PB: the block's first Q / dO / O row; OP_SGB: the wave's staging base
OP_ND = (OP_ND0, OP_ND1, OP_ND2, OP_ND3)
OP_NL = (OP_NL0, OP_NL1, OP_NL2, OP_NL3)


def rng(base, n):
    return f"[{base}:{base + n - 1}]"


def mfma(dst, a, b, c):
    return f"v_mfma_f32_32x32x16_bf16 {dst}, {a}, {b}, {c}"


def mfmas_A(qt):
    """S^T and dP^T of one query tile against the next key tile's row fragments: 8 MFMAs"""
    out = []
    sc, dp = "v" + rng(V_SC(qt), 16), "v" + rng(V_DP(qt), 16)
    for ks in range(4):
        out.append(mfma(sc, "v" + rng(V_KR(ks), 4), "a" + rng(A_QF(qt, ks), 4), "0" if ks == 0 else sc))
    for ks in range(4):
        out.append(mfma(dp, "v" + rng(V_VR(ks), 4), "a" + rng(A_OF(qt, ks), 4), "0" if ks == 0 else dp))
    return out


def mfmas_B(zero):
    """dQ^T += K^T dS of the current tile, query tile by query tile: 16 MFMAs"""
    out = []
    for qt in range(NQ):
        for s in range(2):
            for dt in range(2):
                dq = "a" + rng(A_DQ(dt, qt), 16)
                out.append(mfma(dq, "v" + rng(V_KT(dt, s), 4), "v" + rng(V_D(qt, s), 4), "0" if (zero and s == 0) else dq))
    return out


def valu(qt, elems, mask_reg=None):
    """P = exp2(c2 S + nl), dS = P (dP - delta), in place in the dP tile, for accumulator elements ``elems`` of query tile qt.
    Masked (diagonal) tiles: key 32 kt + c + 4 h > query  <=>  c > mask_reg.  (The add sits between a transcendental and the
    first reader of its result: gfx940+ trans forwarding hazard.)"""
    out = []
    sc, dp = V_SC(qt), V_DP(qt)
    for i in range(elems[0], elems[1], 2):
        for j in (i, i + 1):
            if mask_reg is not None:
                c = (j & 3) + 8 * (j >> 2)
                out.append(f"v_cmp_gt_i32 vcc, {c}, {mask_reg}")
                out.append(f"v_cndmask_b32 v{sc + j}, v{sc + j}, v{V_NINF}, vcc")
        for j in (i, i + 1):
            out.append(f"v_fma_f32 v{sc + j}, v{sc + j}, s{S_C2}, %{OP_NL[qt]}")
        for j in (i, i + 1):
            out.append(f"v_exp_f32 v{sc + j}, v{sc + j}")
        for j in (i, i + 1):
            out.append(f"v_add_f32 v{dp + j}, v{dp + j}, %{OP_ND[qt]}")
        for j in (i, i + 1):
            out.append(f"v_mul_f32 v{dp + j}, v{sc + j}, v{dp + j}")
    return out


def packs(qt):
    """dS of query tile qt -> its two bf16 B-operands"""
    dp = V_DP(qt)
    return [f"v_cvt_pk_bf16_f32 v{V_D(qt, i // 8) + (i % 8) // 2}, v{dp + i}, v{dp + i + 1}" for i in range(0, 16, 2)]


def reads_T(k):
    """K^T fragments [dt][s] of the tile in stage k: two ds_read_b64_tr_b16 each"""
    out = []
    for s in range(2):
        for dt in range(2):
            for u in range(2):
                r = V_KT(dt, s) + 2 * u
                out.append(f"ds_read_b64_tr_b16 v{rng(r, 2)}, v{V_T + 2 * dt + u} offset:{k * STAGE + 2048 * s}")
    return out


def reads_R(k):
    """K and V row fragments of the tile in stage k"""
    out = []
    for ks in range(4):
        out.append(f"ds_read_b128 v{rng(V_KR(ks), 4)}, v{V_R + ks} offset:{k * STAGE}")
    for ks in range(4):
        out.append(f"ds_read_b128 v{rng(V_VR(ks), 4)}, v{V_R + ks} offset:{k * STAGE + 4096}")
    return out


def dma(k, tile_expr=None):
    """This wave's quarter of one key tile into stage k of the workgroup's ring: piece `wave` of the K rows and of the V rows
    (s[S_WB] = LDS base + wave * 1024; the lane offsets address rows 8 wave .. 8 wave + 7).  The stream runs from the diagonal
    down, so the row offset DEcreases.  A tile past the stream's end goes through descriptors with zero records (dropped,
    still counted in vmcnt)."""
    out = []
    if tile_expr is not None:
        out += [f"s_cmp_lt_u32 {tile_expr}, s{S_N}", f"s_cselect_b32 s{S_TMP}, -1, 0",
                f"s_mov_b32 s{S_RK + 2}, s{S_TMP}", f"s_mov_b32 s{S_RV + 2}, s{S_TMP}"]
    for img, rs in ((0, S_RK), (4096, S_RV)):
        out += [f"s_add_u32 m0, s{S_WB}, {k * STAGE + img}", "s_nop 0",
                f"buffer_load_dwordx4 %{OP_DK0}, s{rng(rs, 4)}, s{S_KOFF} offen lds"]
    out += [f"s_sub_u32 s{S_KOFF}, s{S_KOFF}, s{S_KST}"]
    return out


def weave(mf, streams):
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


ALL, DIAG = f"v{V_M1}", f"%{OP_M0}"


def mask_regs(tile):
    """per query tile, for the first four tiles of the (descending) stream: tile j = keys q0 + 32 (3 - j) .. + 31: query tiles
    below 3 - j see none of them, query tile 3 - j the diagonal, the rest all"""
    if tile > 3:
        return [None] * NQ
    return [ALL if qt < 3 - tile else (DIAG if qt == 3 - tile else None) for qt in range(NQ)]


V3_SPLIT = 4                     # elements of query tile 3 computed in the tail of the iteration (the rest: head of the next)
WINDOWS = {"v3b": (0, 8), "p3": (8, 10), "rT": (1, 12), "v0": (10, 20), "v1": (20, 30), "v2": (30, 42), "v3a": (42, 48),
           "p0": (37, 40), "p1": (41, 44), "p2": (45, 48), "dma": (34, 40), "rR": (34, 48)}
MASK_CUR = [None] * NQ           # masks of the tile whose query tile 3 is finished in the body being generated


def body(k, tile_next_masks, zero_acc):
    """iteration t with t % 8 == k: tile t in stage k, tile t+2 (row fragments) in stage k+2, requests for tile t+LEAD"""
    mf = mfmas_A(0) + mfmas_A(1) + mfmas_A(2) + mfmas_A(3) + mfmas_B(zero_acc)
    vs = {"v3b": valu(3, (V3_SPLIT, 16), MASK_CUR[3]), "p3": packs(3),
          "v0": valu(0, (0, 16), tile_next_masks[0]), "v1": valu(1, (0, 16), tile_next_masks[1]),
          "v2": valu(2, (0, 16), tile_next_masks[2]), "v3a": valu(3, (0, V3_SPLIT), tile_next_masks[3]),
          "p0": packs(0), "p1": packs(1), "p2": packs(2)}
    d = [f"s_add_u32 s{S_T4}, s{S_T}, {LEAD}"] + dma((k + LEAD) % NSTAGE, f"s{S_T4}")
    if ABLATE & 1:
        d = [x for x in d if not x.startswith("buffer_load")]
    if ABLATE & 2:
        vs = {n: [] for n in vs}
    rT, rR = reads_T(k), reads_R((k + 2) % NSTAGE)
    if ABLATE & 4:
        rT, rR = [], []
    if ABLATE & 8:
        mf = ["s_nop 0"] * NSLOT
    W = WINDOWS
    # my pieces of tile t+2 have landed (those of tiles t+3, t+4: 4 requests, may fly); the barrier publishes everybody's
    sync = ["s_waitcnt vmcnt(0)" if ABLATE & 1 else f"s_waitcnt vmcnt({2 * (LEAD - 3)})"] + ([] if ABLATE & 16 else ["s_barrier"])
    streams = [(vs[n], *W[n]) for n in ("v3b", "p3")]
    streams += [(rT, *W["rT"]), (["s_waitcnt lgkmcnt(0)"], 31, 32)]            # tile t's K^T fragments, in front of MFMA 32
    streams += [(vs[n], *W[n]) for n in ("v0", "v1", "v2", "p0", "v3a", "p1", "p2")]
    streams += [(sync, 33, 34), (d, *W["dma"]), (rR, *W["rR"])]
    return weave(mf, streams) + ["s_waitcnt lgkmcnt(0)"]


def scalars(L):
    """descriptors and scalar state from the operands (both asm blocks start with this: the compiler owns the SGPRs in between)"""
    e = L.append
    for rs, op in ((S_RK, OP_KB), (S_RV, OP_VB)):
        e(f"s_mov_b64 s{rng(rs, 2)}, %{op}"); e(f"s_and_b32 s{rs + 1}, s{rs + 1}, 0xffff")
        e(f"s_mov_b32 s{rs + 2}, -1"); e(f"s_mov_b32 s{rs + 3}, 0x00020000")
    e(f"s_mov_b32 s{S_N}, %{OP_N}"); e(f"s_mov_b32 s{S_WB}, %{OP_WB}"); e(f"s_mov_b32 s{S_KST}, %{OP_KST}"); e(f"s_mov_b32 s{S_C2}, %{OP_C2}")
    e(f"s_mov_b32 s{S_OST}, %{OP_OST}")


SG_TILE = 12288                  # staging per 32-query tile: Q image, dO image, O image (4 KiB each); two tiles per wave


def stage_descriptors(L):
    e = L.append
    for rs, op in ((S_RQ, OP_QB), (S_RO, OP_OB), (S_RP, OP_PB)):
        e(f"s_mov_b64 s{rng(rs, 2)}, %{op}"); e(f"s_and_b32 s{rs + 1}, s{rs + 1}, 0xffff")
        e(f"s_mov_b32 s{rs + 2}, -1"); e(f"s_mov_b32 s{rs + 3}, 0x00020000")
    e(f"s_mov_b32 s{S_KST}, %{OP_KST}"); e(f"s_mov_b32 s{S_OST}, %{OP_OST}")
    # pieces 2, 3 of a tile = pieces 0, 1 sixteen rows further down (the swizzle repeats every 16 rows)
    e(f"s_lshr_b32 s{S_TMP}, s{S_KST}, 1"); e(f"v_add_u32 v{V_SG + 4}, s{S_TMP}, %{OP_DQ0}"); e(f"v_add_u32 v{V_SG + 5}, s{S_TMP}, %{OP_DQ1}")


def stage_half(L, half):
    """LDS-DMA of the Q, dO and O rows of query tiles 2 half, 2 half + 1 into the wave's staging area: whole 128-B rows, 8 per
    request (the per-lane fragment loads of the first version touched 32 rows per request and used a quarter of each line: 11k
    cycles to issue 32 of them)."""
    e = L.append
    for j in range(2):
        qt = 2 * half + j
        e(f"s_mul_i32 s{S_KOFF}, s{S_KST}, {qt}")                                   # row offset of the tile: 32 qt rows
        e(f"s_mul_i32 s{S_T4}, s{S_OST}, {qt}")
        for p in range(4):
            vq = [f"%{OP_DQ0}", f"%{OP_DQ1}", f"v{V_SG + 4}", f"v{V_SG + 5}"][p]
            e(f"s_add_u32 m0, %{OP_SGB}, {j * SG_TILE + p * 1024}"); e("s_nop 0")
            e(f"buffer_load_dwordx4 {vq}, s{rng(S_RQ, 4)}, s{S_KOFF} offen lds")
        # dO / O rows: their own leading dimension -> their own lane offsets (pieces 2, 3 through a scalar offset of 16 rows)
        e(f"s_lshr_b32 s{S_TMP}, s{S_OST}, 1"); e(f"s_add_u32 s{S_TMP}, s{S_TMP}, s{S_T4}")
        for img, rs in ((4096, S_RO), (8192, S_RP)):
            for p in range(4):
                vo = f"%{OP_DO0}" if p % 2 == 0 else f"%{OP_DO1}"
                so = f"s{S_T4}" if p < 2 else f"s{S_TMP}"
                e(f"s_add_u32 m0, %{OP_SGB}, {j * SG_TILE + img + p * 1024}"); e("s_nop 0")
                e(f"buffer_load_dwordx4 {vo}, s{rng(rs, 4)}, {so} offen lds")


def stage_frags(L, half):
    """the staged Q / dO rows of query tiles 2 half, 2 half + 1 as B-operand fragments, straight into AGPRs"""
    e = L.append
    for ks in range(4):
        e(f"v_xor_b32 v{V_SG + ks}, {32 * ks}, %{OP_SG}")
    for j in range(2):
        qt = 2 * half + j
        for ks in range(4):
            e(f"ds_read_b128 a{rng(A_QF(qt, ks), 4)}, v{V_SG + ks} offset:{j * SG_TILE}")
            e(f"ds_read_b128 a{rng(A_OF(qt, ks), 4)}, v{V_SG + ks} offset:{j * SG_TILE + 4096}")


def gen_stage(half):
    """Prologue block of one half (64 queries): [half 1: the fragments of half 0 have left the staging area] requests; half 0
    also requests this wave's pieces of key tiles 0 .. LEAD-1.  No vmcnt wait in it."""
    L = []
    e = L.append
    if half == 1:
        e("s_waitcnt lgkmcnt(0)")
    stage_descriptors(L)
    stage_half(L, half)
    if half == 0:
        scalars(L)
        e(f"s_sub_u32 s{S_TMP}, s{S_N}, 1"); e(f"s_mul_i32 s{S_KOFF}, s{S_TMP}, s{S_KST}")     # first tile = key tile n - 1
        L += dma(0)
        L += dma(1)
        for j in range(2, LEAD):
            e(f"s_mov_b32 s{S_T4}, {j}")
            L += dma(j, f"s{S_T4}")
    return L


def gen_land(half):
    """the half's rows have landed: fragments -> AGPRs (the delta computation that follows reads dO and O from the same images)"""
    L = ["s_waitcnt vmcnt(0)"]
    stage_frags(L, half)
    return L


def gen():
    global MASK_CUR
    L = []
    e = L.append
    scalars(L)
    # the requests of tiles 0 .. LEAD-1 are out (gen_pro): the next one is tile LEAD = key tile n - 1 - LEAD
    e(f"s_sub_u32 s{S_TMP}, s{S_N}, {1 + LEAD}"); e(f"s_mul_i32 s{S_KOFF}, s{S_TMP}, s{S_KST}")
    for ks in range(4):
        e(f"v_xor_b32 v{V_R + ks}, {32 * ks}, %{OP_ROFF}")
    e(f"v_mov_b32 v{V_T}, %{OP_TOFF}")
    e(f"v_xor_b32 v{V_T + 1}, 32, %{OP_TOFF}"); e(f"v_add_u32 v{V_T + 1}, 1024, v{V_T + 1}")
    e(f"v_xor_b32 v{V_T + 2}, 64, v{V_T}"); e(f"v_xor_b32 v{V_T + 3}, 64, v{V_T + 1}")
    e(f"v_mov_b32 v{V_NINF}, 0xff800000"); e(f"v_subrev_u32 v{V_M1}, 32, %{OP_M0}")
    # everything requested before this block has landed - the wave's operands, its pieces of tiles 0 .. LEAD-1 and whatever the
    # compiler issued in between (a count would have to know those): the loop's own waits below are counted again
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    L += reads_R(0)
    e("s_waitcnt lgkmcnt(0)")
    # ---- iteration -1: S / dP / dS of tile 0 (masked), row fragments of tile 1
    m0 = mask_regs(0)
    for qt in range(NQ):
        L += mfmas_A(qt)
    e("s_barrier")                                                                # tile 1 (everybody's pieces landed above)
    L += reads_R(1)
    for qt in range(3):
        L += valu(qt, (0, 16), m0[qt]) + packs(qt)
    L += valu(3, (0, V3_SPLIT), m0[3])
    e("s_waitcnt lgkmcnt(0)")
    # ---- iterations 0 .. 3 peeled: their tiles (and the next ones) carry the causal masks, iteration 0 starts the accumulators
    for t in range(4):
        e(f"s_mov_b32 s{S_T}, {t}")
        MASK_CUR = mask_regs(t)
        L += body(t, mask_regs(t + 1), t == 0)
    MASK_CUR = [None] * NQ
    e(f"s_mov_b32 s{S_T}, 4"); e(f"s_cmp_ge_u32 s{S_T}, s{S_N}"); e("s_cbranch_scc1 19f")
    e("12:")
    for k in (4, 5, 6, 7, 0, 1, 2, 3):
        L += body(k, [None] * NQ, False)
        e(f"s_add_u32 s{S_T}, s{S_T}, 1"); e(f"s_cmp_ge_u32 s{S_T}, s{S_N}"); e("s_cbranch_scc1 19f")
    e("s_branch 12b")
    e("19:")
    e("s_waitcnt vmcnt(0)")
    e("s_nop 7"); e("s_nop 7"); e("s_nop 7")
    return L


def main(path):
    body_ = gen()
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen/gen_attn64_dq_loop.py - do not edit.  The key loop of attn64_dq_asm_kernel\n"
                "// (attention64_asm.hip) as inline-asm text; register map and schedule in the generator's header.\n")
        f.write("#ifndef CSM_A64_DQ_LOOP      // (tools/probes pre-include an ablated copy)\n")
        for name, ins_list in (("CSM_A64_DQ_STAGE0", gen_stage(0)), ("CSM_A64_DQ_LAND0", gen_land(0)), ("CSM_A64_DQ_STAGE1", gen_stage(1)),
                               ("CSM_A64_DQ_LAND1", gen_land(1)), ("CSM_A64_DQ_LOOP", body_)):
            f.write(f"#define {name} \\\n")
            for ins in ins_list:
                f.write(f'    "{ins}\\n\\t" \\\n')
            f.write('    ""\n')
        f.write("#define CSM_A64_DQ_CLOBBERS " + ", ".join([f'"v{n}"' for n in range(32, LAST_V + 1)] + [f'"a{n}"' for n in range(256)] +
                                                          [f'"s{n}"' for n in range(40, LAST_S + 1)] + ['"scc"', '"vcc"', '"memory"']) + "\n")
        f.write("#define CSM_A64_DQ_PRO_CLOBBERS " + ", ".join([f'"v{n}"' for n in range(V_SG, V_SG + 6)] + [f'"a{n}"' for n in range(128, 256)] +
                                                              [f'"s{n}"' for n in range(40, LAST_S + 1)] + ['"scc"', '"memory"']) + "\n")
        f.write(f"#define CSM_A64_DQ_SG_TILE {SG_TILE}\n")
        f.write(f"#define CSM_A64_DQ_STAGE {STAGE}\n#define CSM_A64_DQ_NSTAGE {NSTAGE}\n#define CSM_A64_DQ_NQ {NQ}\n")
        f.write(f"// {sum(1 for x in body_ if x.startswith('v_mfma'))} MFMAs, {len(body_)} instructions\n")
        f.write("#endif\n")


if __name__ == "__main__":
    main(sys.argv[1])
