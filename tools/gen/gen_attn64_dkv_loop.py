#!/usr/bin/env python3
"""Generates csrc/attn64_dkv_loop.inc: the whole query loop of the head_dim-64 dK/dV attention-backward kernel
(attention64_asm.hip) as ONE inline-asm block with hand-allocated registers.

Why generated assembly (DESIGN 'tried' table r3): the compiler-allocated dK/dV kernel (attention64.hip) spends
its time in the per-step rendezvous of its four waves (214 -> 148 us without barrier + wait) and in LDS fragment reads that
serve one 32-key half each; the form that removes both - ONE wave owns all 64 keys of the block and walks its own stream of
query tiles - needs ~400 live registers, which hipcc turns into AGPR shuffles.  Here every register is assigned by hand:

  a[0:63]    dK^T accumulators, tile (dt, kh) at 16 (2 dt + kh)      (dt: d half, kh: key half; 32 x 32 tiles)
  a[64:127]  dV^T accumulators
  a[128:159] K^T B-operands [kh][ks] (the wave's 64 keys, loaded once), a[160:191] V^T B-operands
  a[192:207] Q row fragments of the NEXT tile [ks], a[208:223] dO row fragments     (ds_read_b128 straight into AGPRs)
  a[224:239] dO^T fragments [dt][s] of the CURRENT tile, a[240:255] Q^T fragments   (ds_read_b64_tr_b16 into AGPRs)
  v[32:63]   S^T tiles [kh] -> P in place, v[64:95] dP^T tiles [kh] -> dS in place
  v[96:159]  packed bf16 P / dS B-operands [parity][kh]{P[s], dS[s]}  (double-buffered: written for tile t+1 while tile t's are read)
  v[160:191] -lse*log2e of the tile's 32 queries as 16 per-lane values [parity], v[192:207] -delta likewise (dP's accumulators start from it)

Work split (attention64_asm.hip): workgroup = 64 keys of one (batch, kv head); wave w = query head w of the GQA group, with a
PRIVATE four-stage LDS ring of 32-query tiles (Q image, dO image, 2 x 32 statistics) that it fills by LDS-DMA and consumes
alone - no barrier anywhere in the loop, every wait is a counted vmcnt / lgkmcnt of the wave's own requests.  The four
waves' dK / dV are summed through LDS once, after the loop (fixed order: deterministic).

Software pipeline, iteration t (32 MFMAs; tile = 32 queries x 64 keys):
  MFMA  0..15  S^T(t+1), dP^T(t+1) for both key halves        (operands: row fragments read during iteration t-1)
  MFMA 16..31  dV^T += dO^T P(t), dK^T += Q^T dS(t)            (operands: packed P / dS of tile t from iteration t-1 and the head of t)
  VALU         exp2 / multiply / pack of tile t+1's key half 0 (slots 10..23) and the first half of key half 1 (24..31); the
               second half of tile t's key half 1 rides in slots 0..7
  LDS          transposed fragments of tile t (slots 8..13), row fragments + statistics of tile t+2 (slots 17..23)
  LDS-DMA      tile t+3 into the stage tile t-1 left (slots 0..9), 9 pieces of 1 KiB / 256 B
The first two tiles of a stream touch the diagonal: those copies of the body carry the causal mask (-inf before the exp2).
"""
import os
import sys

# experiment switch for tools/probes/ablate_a64dkv.sh (never set in the shipped build): what bounds the loop?
# bit0: no LDS-DMA inside the loop; bit1: no exp2 / multiply / pack; bit2: no LDS fragment reads inside the loop; bit3: no MFMA
ABLATE = int(os.environ.get("CSM_A64DKV_ABLATE", "0"))

STAGE = 8448                     # Q image 4096 + dO image 4096 + 64 floats
NSTAGE = 4

# ---- registers -----------------------------------------------------------------------------------------------------
def A_DK(dt, kh): return 16 * (2 * dt + kh)
def A_DV(dt, kh): return 64 + 16 * (2 * dt + kh)
def A_KF(kh, ks): return 128 + 16 * kh + 4 * ks
def A_VF(kh, ks): return 160 + 16 * kh + 4 * ks
def A_QR(ks): return 192 + 4 * ks
def A_OR(ks): return 208 + 4 * ks
def A_U(dt, s): return 224 + 4 * (2 * dt + s)
def A_W(dt, s): return 240 + 4 * (2 * dt + s)
def V_SC(kh): return 32 + 16 * kh
def V_DP(kh): return 64 + 16 * kh
def V_P(par, kh, s): return 96 + 32 * par + 16 * kh + 4 * s
def V_D(par, kh, s): return 96 + 32 * par + 16 * kh + 8 + 4 * s
def V_N(par): return 160 + 16 * par
V_E = 192
V_NINF, V_M1 = 208, 209
V_R = 210                        # v[210:213] row-fragment lane addresses [ks]
V_T = 214                        # v[214:217] transposed-fragment lane addresses [dt][u]
V_DQ = 218                       # v[218:219] LDS-DMA lane offsets of Q pieces 2, 3
V_DO = 220                       # v[220:221] ... of dO pieces 2, 3
LAST_V = 221
S_RQ, S_RO, S_RS = 40, 44, 48    # buffer descriptors
S_T, S_N, S_WB, S_QST, S_OST, S_C2, S_QOFF, S_OOFF, S_SOFF, S_TMP, S_T3 = 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62
S_KB, S_VB = 64, 66
LAST_S = 67

# operands (see the asm statement in attention64_asm.hip)
OP_ROFF, OP_TOFF, OP_SOFF, OP_DQ0, OP_DQ1, OP_DO0, OP_DO1, OP_DS, OP_KV0, OP_KV1, OP_M0 = range(11)
OP_QB, OP_OB, OP_SB, OP_KB, OP_VB, OP_N, OP_WB, OP_QST, OP_OST, OP_C2 = range(11, 21)


def rng(base, n):
    return f"[{base}:{base + n - 1}]"


def mfma(dst, a, b, c):
    return f"v_mfma_f32_32x32x16_bf16 {dst}, {a}, {b}, {c}"


def mfmas_A(kh):
    """S^T and dP^T of one key half: 8 MFMAs"""
    out = []
    sc, dp = "v" + rng(V_SC(kh), 16), "v" + rng(V_DP(kh), 16)
    for ks in range(4):
        out.append(mfma(sc, "a" + rng(A_QR(ks), 4), "a" + rng(A_KF(kh, ks), 4), "0" if ks == 0 else sc))
    for ks in range(4):
        out.append(mfma(dp, "a" + rng(A_OR(ks), 4), "a" + rng(A_VF(kh, ks), 4), ("v" + rng(V_E, 16)) if ks == 0 else dp))
    return out


def mfmas_B(kh, par, zero):
    """dV^T / dK^T updates of one key half from tile t's packed P / dS: 8 MFMAs"""
    out = []
    for s in range(2):
        for dt in range(2):
            dv, dk = "a" + rng(A_DV(dt, kh), 16), "a" + rng(A_DK(dt, kh), 16)
            out.append(mfma(dv, "a" + rng(A_U(dt, s), 4), "v" + rng(V_P(par, kh, s), 4), "0" if (zero and s == 0) else dv))
            out.append(mfma(dk, "a" + rng(A_W(dt, s), 4), "v" + rng(V_D(par, kh, s), 4), "0" if (zero and s == 0) else dk))
    return out


def valu(kh, par, elems, mask_reg=None):
    """P = exp2(c2 S + nl), dS = P dP, packed to bf16, for accumulator elements ``elems`` (pairs) of key half kh.
    Order inside a pair: both fma, both exp2, both multiplies, then the two packs - one independent instruction sits between a
    transcendental and the first reader of its result (gfx940+ trans forwarding hazard)."""
    out = []
    sc, dp, n = V_SC(kh), V_DP(kh), V_N(par)
    for i in range(elems[0], elems[1], 2):
        for j in (i, i + 1):
            if mask_reg is not None:
                c = (j & 3) + 8 * (j >> 2)
                out.append(f"v_cmp_lt_i32 vcc, {c}, {mask_reg}")
                out.append(f"v_cndmask_b32 v{sc + j}, v{sc + j}, v{V_NINF}, vcc")
        for j in (i, i + 1):
            out.append(f"v_fma_f32 v{sc + j}, v{sc + j}, s{S_C2}, v{n + j}")
        for j in (i, i + 1):
            out.append(f"v_exp_f32 v{sc + j}, v{sc + j}")
        for j in (i, i + 1):
            out.append(f"v_mul_f32 v{dp + j}, v{sc + j}, v{dp + j}")
        s, q = i // 8, (i % 8) // 2
        out.append(f"v_cvt_pk_bf16_f32 v{V_P(par, kh, s) + q}, v{sc + i}, v{sc + i + 1}")
        out.append(f"v_cvt_pk_bf16_f32 v{V_D(par, kh, s) + q}, v{dp + i}, v{dp + i + 1}")
    return out


def reads_T(k):
    """transposed fragments of the tile in stage k: dO^T [dt][s] and Q^T [dt][s], two ds_read_b64_tr_b16 each"""
    out = []
    for img, areg in ((4096, A_U), (0, A_W)):
        for s in range(2):
            for dt in range(2):
                for u in range(2):
                    r = areg(dt, s) + 2 * u
                    out.append(f"ds_read_b64_tr_b16 a{rng(r, 2)}, v{V_T + 2 * dt + u} offset:{k * STAGE + img + 2048 * s}")
    return out


def reads_R(k, par):
    """row fragments (into AGPRs) and statistics of the tile in stage k"""
    out = []
    for ks in range(4):
        out.append(f"ds_read_b128 a{rng(A_QR(ks), 4)}, v{V_R + ks} offset:{k * STAGE}")
    for ks in range(4):
        out.append(f"ds_read_b128 a{rng(A_OR(ks), 4)}, v{V_R + ks} offset:{k * STAGE + 4096}")
    for g in range(4):
        out.append(f"ds_read_b128 v{rng(V_N(par) + 4 * g, 4)}, %{OP_SOFF} offset:{k * STAGE + 8192 + 32 * g}")
    for g in range(4):
        out.append(f"ds_read_b128 v{rng(V_E + 4 * g, 4)}, %{OP_SOFF} offset:{k * STAGE + 8192 + 128 + 32 * g}")
    return out


def dma(k, tile_expr=None):
    """LDS-DMA of one tile into stage k: 4 pieces of the Q rows, 4 of the dO rows, the 64 statistics.  ``tile_expr``: scalar
    register holding the tile index - a tile past the stream's end goes through descriptors with ZERO records (dropped by the
    range check, still counted in vmcnt: the waits stay constants)."""
    out = []
    if tile_expr is not None:
        out += [f"s_cmp_lt_u32 {tile_expr}, s{S_N}", f"s_cselect_b32 s{S_TMP}, -1, 0",
                f"s_mov_b32 s{S_RQ + 2}, s{S_TMP}", f"s_mov_b32 s{S_RO + 2}, s{S_TMP}", f"s_mov_b32 s{S_RS + 2}, s{S_TMP}"]
    qv = [f"%{OP_DQ0}", f"%{OP_DQ1}", f"v{V_DQ}", f"v{V_DQ + 1}"]
    ov = [f"%{OP_DO0}", f"%{OP_DO1}", f"v{V_DO}", f"v{V_DO + 1}"]
    for p in range(4):
        out += [f"s_add_u32 m0, s{S_WB}, {k * STAGE + p * 1024}", "s_nop 0",
                f"buffer_load_dwordx4 {qv[p]}, s{rng(S_RQ, 4)}, s{S_QOFF} offen lds"]
    for p in range(4):
        out += [f"s_add_u32 m0, s{S_WB}, {k * STAGE + 4096 + p * 1024}", "s_nop 0",
                f"buffer_load_dwordx4 {ov[p]}, s{rng(S_RO, 4)}, s{S_OOFF} offen lds"]
    out += [f"s_add_u32 m0, s{S_WB}, {k * STAGE + 8192}", "s_nop 0",
            f"buffer_load_dword %{OP_DS}, s{rng(S_RS, 4)}, s{S_SOFF} offen lds"]
    out += [f"s_add_u32 s{S_QOFF}, s{S_QOFF}, s{S_QST}", f"s_add_u32 s{S_OOFF}, s{S_OOFF}, s{S_OST}",
            f"s_add_u32 s{S_SOFF}, s{S_SOFF}, 128"]
    return out


def weave(mf, streams):
    """side streams, each (instructions, first slot, last slot): spread evenly over the slots AFTER MFMAs [first, last)"""
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


def cost(ins):
    """issue cost in cycles of a filler instruction beside MFMAs (MI355X_MICROARCH.md, per-instruction constants)"""
    op = ins.split()[0]
    if op == "v_exp_f32":
        return 8
    if op == "v_cvt_pk_bf16_f32":
        return 5
    if op.startswith("v_"):
        return 4
    if op.startswith("ds_read"):
        return 8
    if op.startswith("buffer_load"):
        return 40
    if op.startswith("s_waitcnt"):
        return 0
    return 1                                      # scalar instructions issue beside the vector ones


def mask_regs(tile):
    """per key half: the lane register to compare the element constants with, for the diagonal tiles 0 and 1 of a stream
    (key_local - 4h - 32 (tile - kh) > c  <=>  masked); None = nothing masked"""
    if tile == 0:
        return [f"%{OP_M0}", f"v{V_M1}"]          # kh 0: the diagonal; kh 1: every key is ahead of every query
    if tile == 1:
        return [None, f"%{OP_M0}"]                # kh 0: all visible; kh 1: the diagonal
    return [None, None]


def body(k, tile_next_masks, zero_acc):
    """iteration t with t % 4 == k (tile t in stage k).  ``tile_next_masks`` = mask_regs of tile t+1."""
    par_r, par_w = k % 2, (k + 1) % 2
    mf = mfmas_A(0) + mfmas_A(1) + mfmas_B(0, par_r, zero_acc) + mfmas_B(1, par_r, zero_acc)
    v1b = valu(1, par_r, (V1_SPLIT, 16), MASK_CUR[1])
    v0 = valu(0, par_w, (0, 16), tile_next_masks[0])
    v1a = valu(1, par_w, (0, V1_SPLIT), tile_next_masks[1])
    d = [f"s_add_u32 s{S_T3}, s{S_T}, 4"] + dma(k, f"s{S_T3}")
    if ABLATE & 1:
        d = [x for x in d if not x.startswith("buffer_load")]
    if ABLATE & 2:
        v1b, v0, v1a = [], [], []
    rT, rR = reads_T(k), reads_R((k + 2) % 4, par_r)
    if ABLATE & 4:
        rT, rR = [], []
    if ABLATE & 8:
        mf = ["s_nop 0"] * 32
    W = WINDOWS
    streams = [
        (v1b, *W["v1b"]),                                      # must be done before MFMA 8 rewrites S^T / dP^T of key half 1
        (rT, *W["rT"]),
        (["s_waitcnt lgkmcnt(0)"], W["rT"][1], W["rT"][1] + 1),   # tile t's transposed fragments have landed: its stage is free
        (d, *W["dma"]),                                        # tile t+4 into the stage tile t has just left
        (v0, *W["v0"]),
        (["s_waitcnt vmcnt(@)"], 16, 17),                      # tile t+2 has landed (count patched below)
        (rR, *W["rR"]),                                        # row fragments / statistics of tile t+2 (parity of t+2 == parity of t)
        (v1a, *W["v1a"]),
    ]
    out = weave(mf, streams) + ["s_waitcnt lgkmcnt(0)"]
    # the wait for tile t+2 must let exactly the younger requests fly: tile t+3's 9 and those of tile t+4 issued before it
    i_wait = out.index("s_waitcnt vmcnt(@)")
    n_before = sum(1 for x in out[:i_wait] if x.startswith("buffer_load"))
    out[i_wait] = "s_waitcnt vmcnt(0)" if ABLATE & 1 else f"s_waitcnt vmcnt({9 + n_before})"
    return out


# filler windows [first slot, last slot) of the iteration's streams (slot i = behind MFMA i); see the header for what limits each
WINDOWS = {"v1b": (0, 8), "rT": (1, 9), "dma": (10, 32), "v0": (10, 22), "rR": (17, 32), "v1a": (22, 32)}
V1_SPLIT = int(os.environ.get("CSM_A64DKV_V1SPLIT", "8"))    # elements of key half 1 handled in the tail of the iteration (rest: head of the next)
MASK_CUR = [None, None]           # masks of the tile whose second key-half part (v1b) runs in the body being generated


def gen():
    global MASK_CUR
    L = []
    e = L.append
    # ---- descriptors, scalars, derived lane addresses
    for rs, op in ((S_RQ, OP_QB), (S_RO, OP_OB), (S_RS, OP_SB)):
        e(f"s_mov_b64 s{rng(rs, 2)}, %{op}"); e(f"s_and_b32 s{rs + 1}, s{rs + 1}, 0xffff")
        e(f"s_mov_b32 s{rs + 2}, -1"); e(f"s_mov_b32 s{rs + 3}, 0x00020000")
    e(f"s_mov_b64 s{rng(S_KB, 2)}, %{OP_KB}"); e(f"s_mov_b64 s{rng(S_VB, 2)}, %{OP_VB}")
    e(f"s_mov_b32 s{S_N}, %{OP_N}"); e(f"s_mov_b32 s{S_WB}, %{OP_WB}"); e(f"s_mov_b32 s{S_QST}, %{OP_QST}"); e(f"s_mov_b32 s{S_OST}, %{OP_OST}")
    e(f"s_mov_b32 s{S_C2}, %{OP_C2}")
    e(f"s_mov_b32 s{S_QOFF}, 0"); e(f"s_mov_b32 s{S_OOFF}, 0"); e(f"s_mov_b32 s{S_SOFF}, 0")
    # K^T / V^T operands of the wave's 64 keys, straight into AGPRs (older than every LDS-DMA request below)
    for kh, op in ((0, OP_KV0), (1, OP_KV1)):
        for ks in range(4):
            e(f"global_load_dwordx4 a{rng(A_KF(kh, ks), 4)}, %{op}, s{rng(S_KB, 2)} offset:{32 * ks}")
            e(f"global_load_dwordx4 a{rng(A_VF(kh, ks), 4)}, %{op}, s{rng(S_VB, 2)} offset:{32 * ks}")
    # pieces 2, 3 of a tile = pieces 0, 1 sixteen rows further down (same swizzle: it repeats every 16 rows)
    e(f"s_lshr_b32 s{S_TMP}, s{S_QST}, 1"); e(f"v_add_u32 v{V_DQ}, s{S_TMP}, %{OP_DQ0}"); e(f"v_add_u32 v{V_DQ + 1}, s{S_TMP}, %{OP_DQ1}")
    e(f"s_lshr_b32 s{S_TMP}, s{S_OST}, 1"); e(f"v_add_u32 v{V_DO}, s{S_TMP}, %{OP_DO0}"); e(f"v_add_u32 v{V_DO + 1}, s{S_TMP}, %{OP_DO1}")
    # row-fragment addresses: chunk (2 ks + h) ^ f(r) = ((h ^ f(r)) ^ 2 ks  ->  address ^ (ks << 5)
    for ks in range(4):
        e(f"v_xor_b32 v{V_R + ks}, {32 * ks}, %{OP_ROFF}")
    # transposed-fragment addresses [dt][u]: dt flips chunk bit 2 (^ 64); u adds 8 rows, whose swizzle differs in bit 1 (^ 32, + 1024)
    e(f"v_mov_b32 v{V_T}, %{OP_TOFF}")
    e(f"v_xor_b32 v{V_T + 1}, 32, %{OP_TOFF}"); e(f"v_add_u32 v{V_T + 1}, 1024, v{V_T + 1}")
    e(f"v_xor_b32 v{V_T + 2}, 64, v{V_T}"); e(f"v_xor_b32 v{V_T + 3}, 64, v{V_T + 1}")
    e(f"v_mov_b32 v{V_NINF}, 0xff800000"); e(f"v_add_u32 v{V_M1}, 32, %{OP_M0}")
    # ---- tiles 0, 1, 2 (tile 2 only if it exists: n >= 2 always)
    L += dma(0)
    L += dma(1)
    for j in (2, 3):
        e(f"s_mov_b32 s{S_T3}, {j}")
        L += dma(j, f"s{S_T3}")
    e("s_waitcnt vmcnt(27)")                                                     # K / V operands and tile 0
    L += reads_R(0, 0)
    e("s_waitcnt lgkmcnt(0)")
    # ---- iteration -1: S / dP / P / dS of tile 0 (masked), row fragments of tile 1
    m0 = mask_regs(0)
    L += mfmas_A(0) + mfmas_A(1)
    e("s_waitcnt vmcnt(18)")                                                     # tile 1
    L += reads_R(1, 1)
    L += valu(0, 0, (0, 16), m0[0])
    L += valu(1, 0, (0, V1_SPLIT), m0[1])
    e("s_waitcnt lgkmcnt(0)")
    # ---- iteration 0 (peeled: its accumulators start from zero, tile 1 carries the other diagonal mask), then the loop
    e(f"s_mov_b32 s{S_T}, 0")
    MASK_CUR = m0
    L += body(0, mask_regs(1), True)
    e(f"s_mov_b32 s{S_T}, 1")
    e("s_branch 11f")
    MASK_CUR = [None, None]
    e("10:")
    L += body(0, [None, None], False)
    e(f"s_add_u32 s{S_T}, s{S_T}, 1"); e(f"s_cmp_ge_u32 s{S_T}, s{S_N}"); e("s_cbranch_scc1 19f")
    e("11:")
    # iteration 1's v1b still belongs to tile 1 (masked): a peeled copy, then the steady-state copies
    MASK_CUR = mask_regs(1)
    L += body(1, [None, None], False)
    MASK_CUR = [None, None]
    e(f"s_add_u32 s{S_T}, s{S_T}, 1"); e(f"s_cmp_ge_u32 s{S_T}, s{S_N}"); e("s_cbranch_scc1 19f")
    e("s_branch 13f")
    e("12:")
    L += body(1, [None, None], False)
    e(f"s_add_u32 s{S_T}, s{S_T}, 1"); e(f"s_cmp_ge_u32 s{S_T}, s{S_N}"); e("s_cbranch_scc1 19f")
    e("13:")
    for k in (2, 3):
        L += body(k, [None, None], False)
        e(f"s_add_u32 s{S_T}, s{S_T}, 1"); e(f"s_cmp_ge_u32 s{S_T}, s{S_N}"); e("s_cbranch_scc1 19f")
    L += body(0, [None, None], False)
    e(f"s_add_u32 s{S_T}, s{S_T}, 1"); e(f"s_cmp_ge_u32 s{S_T}, s{S_N}"); e("s_cbranch_scc1 19f")
    e("s_branch 12b")
    e("19:")
    e("s_waitcnt vmcnt(0)")
    e("s_nop 7"); e("s_nop 7"); e("s_nop 7")                                       # the last MFMA results, before v_accvgpr_read
    return L


def main(path):
    body_ = gen()
    # the unused first loop copy ('10:') is never branched to: drop it (kept above only so that labels read in order)
    i0, i1 = body_.index("10:"), body_.index("11:")
    body_ = body_[:i0] + body_[i1:]
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen/gen_attn64_dkv_loop.py - do not edit.  The query loop of attn64_dkv_asm_kernel\n"
                "// (attention64_asm.hip) as inline-asm text; register map and schedule in the generator's header.\n")
        f.write("#ifndef CSM_A64_DKV_LOOP      // (tools/probes pre-include an ablated copy)\n")
        f.write("#define CSM_A64_DKV_LOOP \\\n")
        for ins in body_:
            f.write(f'    "{ins}\\n\\t" \\\n')
        f.write('    ""\n')
        f.write("#define CSM_A64_DKV_CLOBBERS " + ", ".join([f'"v{n}"' for n in range(32, LAST_V + 1)] + [f'"a{n}"' for n in range(256)] +
                                                           [f'"s{n}"' for n in range(40, LAST_S + 1)] + ['"scc"', '"vcc"', '"memory"']) + "\n")
        f.write(f"#define CSM_A64_DKV_STAGE {STAGE}\n#define CSM_A64_DKV_NSTAGE {NSTAGE}\n")
        f.write(f"// {sum(1 for x in body_ if x.startswith('v_mfma'))} MFMAs, {len(body_)} instructions\n")
        f.write("#endif\n")


if __name__ == "__main__":
    main(sys.argv[1])
