#!/usr/bin/env python3
"""Generates csrc/attn64_dq_loop.inc: the key loop of the head_dim-64 dQ attention-backward kernel (attention64_asm.hip,
attn64_dq_asm_kernel) as ONE inline-asm block with hand-allocated registers - the dK/dV generator's structure
(gen_attn64_dkv_loop.py) with the roles of queries and keys exchanged.

One wave owns 64 queries (two 32-query tiles, the query on the lane) of one (batch, query head) and walks the stream of
32-key tiles from the diagonal DOWN to key 0, so that the two tiles that need the causal mask come first.  The four waves of a
workgroup are the four query heads of one GQA group on the same 64-query block: they need the SAME K / V tiles, so the
workgroup keeps ONE eight-stage LDS ring (K image + V image per stage) and every wave fetches a quarter of each tile (one
1-KiB piece of K, one of V) by LDS-DMA, five tiles ahead.  One s_barrier per iteration - after the wave's counted wait for its
pieces of tile t+2 - publishes that tile and frees the stage of the tile everybody has left; private rings (first version)
issued four times the requests and lost more to their issue cost than the rendezvous costs (130 vs 137 us for the
compiler-scheduled kernel).  No cross-wave sum.

  a[0:63]    dQ^T accumulators, tile (dt, qt) at 16 (2 dt + qt)        (dt: d half, qt: query tile)
  a[64:95]   Q^T B-operands [qt][ks] (the wave's 64 queries, loaded once), a[96:127] dO^T B-operands
  a[128:143] K row fragments of the NEXT tile [ks], a[144:159] V row fragments      (ds_read_b128 straight into AGPRs)
  a[160:175] K^T fragments [dt][s] of the CURRENT tile                               (ds_read_b64_tr_b16 into AGPRs)
  v[32:63]   S^T tiles [qt] -> P in place, v[64:95] dP^T tiles [qt] -> dS in place (their accumulators start from -delta)
  v[96:127]  packed bf16 dS B-operands [parity][qt][s]
  v[128:159] -delta of the lane's query as a 16-register tile [qt] (the C operand of dP's first MFMA)

Software pipeline, iteration t (24 MFMAs; tile = 32 keys x 64 queries):
  MFMA  0..15  S^T(t+1), dP^T(t+1) for both query tiles
  MFMA 16..23  dQ^T += K^T dS(t)
  VALU         exp2 / multiply / pack of tile t+1 (query tile 0: slots 10..23; query tile 1: its head in 18..23, the rest in
               slots 0..7 of the next iteration)
  LDS          K^T fragments of tile t (slots 1..8), K / V row fragments of tile t+2 (17..23, behind the barrier)
  LDS-DMA      this wave's two pieces of tile t+5 (slots 17..23)
"""
import os
import sys

ABLATE = int(os.environ.get("CSM_A64DQ_ABLATE", "0"))     # bit0 no LDS-DMA in the loop, bit1 no VALU, bit2 no LDS reads, bit3 no MFMA
STAGE = 8192                     # K image 4096 + V image 4096
NSTAGE = 8
LEAD = 5                         # tiles between the LDS-DMA requests and the tile whose dQ update runs
NSLOT = 24


def A_DQ(dt, qt): return 16 * (2 * dt + qt)
def A_QF(qt, ks): return 64 + 16 * qt + 4 * ks
def A_OF(qt, ks): return 96 + 16 * qt + 4 * ks
def A_KR(ks): return 128 + 4 * ks
def A_VR(ks): return 144 + 4 * ks
def A_KT(dt, s): return 160 + 4 * (2 * dt + s)
def V_SC(qt): return 32 + 16 * qt
def V_DP(qt): return 64 + 16 * qt
def V_D(par, qt, s): return 96 + 16 * par + 8 * qt + 4 * s
def V_ND(qt): return 128 + 16 * qt
V_NINF, V_M1 = 160, 161
V_R = 162                        # v[162:165] row-fragment lane addresses [ks]
V_T = 166                        # v[166:169] transposed-fragment lane addresses [dt][u]
LAST_V = 169
S_RK, S_RV = 40, 44              # buffer descriptors
S_T, S_N, S_WB, S_KST, S_C2, S_KOFF, S_TMP, S_T4 = 48, 49, 50, 51, 52, 53, 54, 55
S_QB, S_OB = 56, 58
LAST_S = 59

OP_ROFF, OP_TOFF, OP_DK0, OP_UNUSED, OP_Q0, OP_Q1, OP_O0, OP_O1, OP_ND0, OP_ND1, OP_NL0, OP_NL1, OP_M0 = range(13)
OP_KB, OP_VB, OP_QB, OP_OB, OP_N, OP_WB, OP_KST, OP_C2 = range(13, 21)
OP_NL = (OP_NL0, OP_NL1)


def rng(base, n):
    return f"[{base}:{base + n - 1}]"


def mfma(dst, a, b, c):
    return f"v_mfma_f32_32x32x16_bf16 {dst}, {a}, {b}, {c}"


def mfmas_A(qt):
    """S^T and dP^T of one query tile against the next key tile's row fragments: 8 MFMAs"""
    out = []
    sc, dp = "v" + rng(V_SC(qt), 16), "v" + rng(V_DP(qt), 16)
    for ks in range(4):
        out.append(mfma(sc, "a" + rng(A_KR(ks), 4), "a" + rng(A_QF(qt, ks), 4), "0" if ks == 0 else sc))
    for ks in range(4):
        out.append(mfma(dp, "a" + rng(A_VR(ks), 4), "a" + rng(A_OF(qt, ks), 4), ("v" + rng(V_ND(qt), 16)) if ks == 0 else dp))
    return out


def mfmas_B(par, zero):
    """dQ^T += K^T dS of the current tile: 8 MFMAs"""
    out = []
    for s in range(2):
        for dt in range(2):
            for qt in range(2):
                dq = "a" + rng(A_DQ(dt, qt), 16)
                out.append(mfma(dq, "a" + rng(A_KT(dt, s), 4), "v" + rng(V_D(par, qt, s), 4), "0" if (zero and s == 0) else dq))
    return out


def valu(qt, par, elems, mask_reg=None):
    """P = exp2(c2 S + nl), dS = P dP (dP already holds dP - delta), packed to bf16, for accumulator elements ``elems`` of
    query tile qt.  Masked (diagonal) tiles: key 32 kt + c + 4 h > query  <=>  c > mask_reg."""
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
            out.append(f"v_mul_f32 v{dp + j}, v{sc + j}, v{dp + j}")
        s, q = i // 8, (i % 8) // 2
        out.append(f"v_cvt_pk_bf16_f32 v{V_D(par, qt, s) + q}, v{dp + i}, v{dp + i + 1}")
    return out


def reads_T(k):
    """K^T fragments [dt][s] of the tile in stage k: two ds_read_b64_tr_b16 each"""
    out = []
    for s in range(2):
        for dt in range(2):
            for u in range(2):
                r = A_KT(dt, s) + 2 * u
                out.append(f"ds_read_b64_tr_b16 a{rng(r, 2)}, v{V_T + 2 * dt + u} offset:{k * STAGE + 2048 * s}")
    return out


def reads_R(k):
    """K and V row fragments (into AGPRs) of the tile in stage k"""
    out = []
    for ks in range(4):
        out.append(f"ds_read_b128 a{rng(A_KR(ks), 4)}, v{V_R + ks} offset:{k * STAGE}")
    for ks in range(4):
        out.append(f"ds_read_b128 a{rng(A_VR(ks), 4)}, v{V_R + ks} offset:{k * STAGE + 4096}")
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


def mask_regs(tile):
    """per query tile, for the first two tiles of the (descending) stream: tile 0 = keys q0+32 .. q0+63, tile 1 = keys q0 .. q0+31"""
    if tile == 0:
        return [f"v{V_M1}", f"%{OP_M0}"]          # query tile 0: every key is ahead; query tile 1: the diagonal
    if tile == 1:
        return [f"%{OP_M0}", None]                # query tile 0: the diagonal; query tile 1: all visible
    return [None, None]


WINDOWS = {"v1b": (0, 8), "rT": (1, 9), "dma": (17, 24), "v0": (10, 24), "rR": (17, 24), "v1a": (18, 24)}
V1_SPLIT = int(os.environ.get("CSM_A64DQ_V1SPLIT", "4"))
MASK_CUR = [None, None]


def body(k, tile_next_masks, zero_acc):
    """iteration t with t % 8 == k: tile t in stage k, tile t+2 (row fragments) in stage k+2, requests for tile t+LEAD"""
    par_r, par_w = k % 2, (k + 1) % 2
    mf = mfmas_A(0) + mfmas_A(1) + mfmas_B(par_r, zero_acc)
    v1b = valu(1, par_r, (V1_SPLIT, 16), MASK_CUR[1])
    v0 = valu(0, par_w, (0, 16), tile_next_masks[0])
    v1a = valu(1, par_w, (0, V1_SPLIT), tile_next_masks[1])
    d = [f"s_add_u32 s{S_T4}, s{S_T}, {LEAD}"] + dma((k + LEAD) % NSTAGE, f"s{S_T4}")
    if ABLATE & 1:
        d = [x for x in d if not x.startswith("buffer_load")]
    if ABLATE & 2:
        v1b, v0, v1a = [], [], []
    rT, rR = reads_T(k), reads_R((k + 2) % NSTAGE)
    if ABLATE & 4:
        rT, rR = [], []
    if ABLATE & 8:
        mf = ["s_nop 0"] * NSLOT
    W = WINDOWS
    # my pieces of tile t+2 have landed (those of tiles t+3, t+4: 4 requests, may fly); the barrier publishes everybody's and tells
    # that every wave has left tile t-1 (in fact t-3 is what the requests below overwrite)
    sync = ["s_waitcnt vmcnt(0)" if ABLATE & 1 else f"s_waitcnt vmcnt({2 * (LEAD - 3)})"] + ([] if ABLATE & 16 else ["s_barrier"])
    streams = [
        (v1b, *W["v1b"]),
        (rT, *W["rT"]),
        (["s_waitcnt lgkmcnt(0)"], W["rT"][1], W["rT"][1] + 1),    # tile t's K^T fragments have landed
        (v0, *W["v0"]),
        (sync, 16, 17),
        (d, *W["dma"]),
        (rR, *W["rR"]),
        (v1a, *W["v1a"]),
    ]
    return weave(mf, streams) + ["s_waitcnt lgkmcnt(0)"]


def scalars(L):
    """descriptors and scalar state from the operands (both asm blocks start with this: the compiler owns the SGPRs in between)"""
    e = L.append
    for rs, op in ((S_RK, OP_KB), (S_RV, OP_VB)):
        e(f"s_mov_b64 s{rng(rs, 2)}, %{op}"); e(f"s_and_b32 s{rs + 1}, s{rs + 1}, 0xffff")
        e(f"s_mov_b32 s{rs + 2}, -1"); e(f"s_mov_b32 s{rs + 3}, 0x00020000")
    e(f"s_mov_b64 s{rng(S_QB, 2)}, %{OP_QB}"); e(f"s_mov_b64 s{rng(S_OB, 2)}, %{OP_OB}")
    e(f"s_mov_b32 s{S_N}, %{OP_N}"); e(f"s_mov_b32 s{S_WB}, %{OP_WB}"); e(f"s_mov_b32 s{S_KST}, %{OP_KST}"); e(f"s_mov_b32 s{S_C2}, %{OP_C2}")


def gen_pro():
    """Block 1 (no wait in it): the wave's Q^T / dO^T operands straight into AGPRs and its pieces of tiles 0 .. LEAD-1.  Whatever the
    compiler issues between this block and the loop (the delta computation's loads, the RoPE table entries) overlaps them."""
    L = []
    e = L.append
    scalars(L)
    # first tile of the stream = key tile n - 1: row offset (n - 1) * step
    e(f"s_sub_u32 s{S_TMP}, s{S_N}, 1"); e(f"s_mul_i32 s{S_KOFF}, s{S_TMP}, s{S_KST}")
    for qt, (oq, oo) in enumerate(((OP_Q0, OP_O0), (OP_Q1, OP_O1))):
        for ks in range(4):
            e(f"global_load_dwordx4 a{rng(A_QF(qt, ks), 4)}, %{oq}, s{rng(S_QB, 2)} offset:{32 * ks}")
            e(f"global_load_dwordx4 a{rng(A_OF(qt, ks), 4)}, %{oo}, s{rng(S_OB, 2)} offset:{32 * ks}")
    L += dma(0)
    L += dma(1)
    for j in range(2, LEAD):
        e(f"s_mov_b32 s{S_T4}, {j}")
        L += dma(j, f"s{S_T4}")
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
    for qt, op in ((0, OP_ND0), (1, OP_ND1)):
        for i in range(16):
            e(f"v_mov_b32 v{V_ND(qt) + i}, %{op}")
    # everything requested before this block has landed - the wave's operands, its pieces of tiles 0 .. LEAD-1 and whatever the
    # compiler issued in between (a count would have to know those): the loop's own waits below are counted again
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    L += reads_R(0)
    e("s_waitcnt lgkmcnt(0)")
    m0 = mask_regs(0)
    L += mfmas_A(0) + mfmas_A(1)
    e("s_barrier")                                                                # tile 1 (everybody's pieces landed above)
    L += reads_R(1)
    L += valu(0, 0, (0, 16), m0[0])
    L += valu(1, 0, (0, V1_SPLIT), m0[1])
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_mov_b32 s{S_T}, 0")
    MASK_CUR = m0
    L += body(0, mask_regs(1), True)
    e(f"s_mov_b32 s{S_T}, 1")
    MASK_CUR = mask_regs(1)
    L += body(1, [None, None], False)
    MASK_CUR = [None, None]
    e(f"s_add_u32 s{S_T}, s{S_T}, 1"); e(f"s_cmp_ge_u32 s{S_T}, s{S_N}"); e("s_cbranch_scc1 19f")
    e("s_branch 13f")
    e("12:")
    L += body(1, [None, None], False)
    e(f"s_add_u32 s{S_T}, s{S_T}, 1"); e(f"s_cmp_ge_u32 s{S_T}, s{S_N}"); e("s_cbranch_scc1 19f")
    e("13:")
    for k in list(range(2, NSTAGE)) + [0]:
        L += body(k, [None, None], False)
        e(f"s_add_u32 s{S_T}, s{S_T}, 1"); e(f"s_cmp_ge_u32 s{S_T}, s{S_N}"); e("s_cbranch_scc1 19f")
    e("s_branch 12b")
    e("19:")
    e("s_waitcnt vmcnt(0)")
    e("s_nop 7"); e("s_nop 7"); e("s_nop 7")
    return L


def main(path):
    body_, pro = gen(), gen_pro()
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen/gen_attn64_dq_loop.py - do not edit.  The key loop of attn64_dq_asm_kernel\n"
                "// (attention64_asm.hip) as inline-asm text; register map and schedule in the generator's header.\n")
        f.write("#ifndef CSM_A64_DQ_LOOP      // (tools/probes pre-include an ablated copy)\n")
        for name, ins_list in (("CSM_A64_DQ_PRO", pro), ("CSM_A64_DQ_LOOP", body_)):
            f.write(f"#define {name} \\\n")
            for ins in ins_list:
                f.write(f'    "{ins}\\n\\t" \\\n')
            f.write('    ""\n')
        f.write("#define CSM_A64_DQ_CLOBBERS " + ", ".join([f'"v{n}"' for n in range(32, LAST_V + 1)] + [f'"a{n}"' for n in range(176)] +
                                                          [f'"s{n}"' for n in range(40, LAST_S + 1)] + ['"scc"', '"vcc"', '"memory"']) + "\n")
        f.write("#define CSM_A64_DQ_PRO_CLOBBERS " + ", ".join([f'"a{n}"' for n in range(64, 128)] +
                                                              [f'"s{n}"' for n in range(40, LAST_S + 1)] + ['"scc"', '"memory"']) + "\n")
        f.write(f"#define CSM_A64_DQ_STAGE {STAGE}\n#define CSM_A64_DQ_NSTAGE {NSTAGE}\n")
        f.write(f"// {sum(1 for x in body_ if x.startswith('v_mfma'))} MFMAs, {len(body_)} instructions\n")
        f.write("#endif\n")


if __name__ == "__main__":
    main(sys.argv[1])
