"""Explicit forward / backward schedule of the CSM train step over libcsm_hip.so.

There is no autograd tape underneath: the layer structure is fixed, so the engine walks it forwards saving the
activations the backward needs, then backwards issuing dgrad / wgrad GEMMs and the fused element-wise backward
kernels, accumulating straight into the bf16 gradient arena (wgrad GEMM epilogue ``C += alpha * dY^T X``).
It restates what autograd does for the reference hot loop ``compute_loss -> loss.backward()``
(reference src/csm/training/utils.py:56-119, src/csm/training/trainer.py:245-263).

Activations kept per layer (bf16 unless noted): x (block input), xn, qkv (post-RoPE), attn out, lse (fp32),
h (post-attention residual), hn, gate|up, swiglu out, rstd x2 (fp32)  ~= 76 KB per position per backbone layer.
With 288 GB of HBM nothing is recomputed.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .hip import ops

BF16 = torch.bfloat16
F32 = torch.float32
# SwiGLU fused into the GEMM epilogues (csm_gemm_bf16_ex).  A/B on one MI355X box, B=4 S=2048: 88.0 ms/step fused vs
# 88.9 ms with the stand-alone kernels (the 256x256 GEMM runs one workgroup per CU, so most of the epilogue work is
# exposed; the gain is what remains after that).  CSM_FUSE_SWIGLU=0 restores the stand-alone path for re-evaluation.
import os as _os
FUSE_SWIGLU = _os.environ.get("CSM_FUSE_SWIGLU", "1") == "1"
FUSE_ROPE_BWD = _os.environ.get("CSM_FUSE_ROPE_BWD", "1") == "1"      # A/B switch: RoPE backward inside the attention backward
FUSE_ROPE_FWD = _os.environ.get("CSM_FUSE_ROPE_FWD", "1") == "1"      # A/B switch: RoPE forward inside the q|k|v GEMM epilogue
# A/B switch: dgrad + wgrad of a Linear layer as one launch with interleaved tiles (bit mask: 1 w2, 2 w13, 4 output_proj, 8 qkv).
# OFF by default: measured on the B=4, S=2048 step (tools/probes/pair_ab.sh) every pairing is SLOWER than the two separate
# launches (73.7 ms/step -> 75.7 with w2 alone, 77.4 w13, 77.1 output_proj, 78.3 qkv, 83.2 all four): de-phasing the CUs'
# epilogues does not pay for two operand sets competing for each XCD's 4 MiB L2.  Kept as a tested entry point.
PAIR_DX_DW = int(_os.environ.get("CSM_PAIR_DX_DW", "0"))
# A/B switch: the two small weight gradients of a layer (attention output projection: 64 tiles, fused q|k|v: 96 tiles of
# 256x256) as ONE launch after the attention backward, instead of split-K slabs + column sum and a 128x128-tile launch
GROUP_ATTN_DW = _os.environ.get("CSM_GROUP_ATTN_DW", "1") == "1"
# the attention projections' weight gradients of this many layers go out in one launch (1 = per layer); one layer's are 160
# tiles of 256 x 256 - 0.63 of a round of the 256 CUs - three layers' 480: 1.9 rounds
DEFER_ATTN_DW = int(_os.environ.get("CSM_DEFER_ATTN_DW", "3"))
# the RMSNorm scale gradients' column sums of a layer (and of the layers whose attention gradients are deferred) in one launch
DEFER_NORM_DW = _os.environ.get("CSM_DEFER_NORM_DW", "1") == "1"
# the depth decoder's random frame subset is drawn on the host (A/B: 0 = device randperm)
ROWS_ON_HOST = _os.environ.get("CSM_ROWS_ON_HOST", "1") == "1"
# embedding backward: one sort of (row, source) keys instead of a stable argsort + two gathers (A/B: 0 = the latter)
EMB_KEYSORT = _os.environ.get("CSM_EMB_KEYSORT", "1") == "1"
# the depth decoder's fused attention + output projection takes its position as a launch argument (A/B: 0 = from device memory)
DECODE_POS_HOST = _os.environ.get("CSM_DECODE_POS_HOST", "1") == "1"
# LoRA groups ride on the frozen projections' GEMMs as K-extension operands (training/lora.py); 0 = per-adapter products
LORA_FUSE = _os.environ.get("CSM_LORA_FUSE", "1") != "0"


class _Stack:
    """One Llama stack (backbone or decoder) bound to a model's arenas."""

    def __init__(self, model, prefix: str):
        self.m, self.prefix = model, prefix
        self.c = model.bb if prefix == "backbone" else model.dc
        self.acts: List[Dict[str, torch.Tensor]] = []

    def w(self, name, grad=False):
        return self.m.block(f"{self.prefix}.{name}", grad)

    def _lora(self, layer: int, module: str):
        lo = self.m.lora
        return lo.get(self.prefix, layer, module) if lo is not None else None

    def _group(self, layer: int, gname: str):
        """(group, fused?) - the LoRA adapters of ``layer`` on one fused projection (lora.py GROUPS); fused = the whole
        group enters that projection's GEMMs as one K-extension operand pair."""
        lo = self.m.lora
        G = lo.group(self.prefix, layer, gname) if lo is not None else None
        return G, (G is not None and LORA_FUSE and G.fusable())

    # -------------------------------------------------------------------------------------------- forward
    def forward(self, x: torch.Tensor, B: int, S: int, save: bool, pos: Optional[torch.Tensor] = None,
                fuse_rope: bool = True, on_layer_start=None) -> torch.Tensor:
        c, dev = self.c, x.device
        M, d = x.shape
        H, KV, hd, F = c.num_heads, c.num_kv_heads, c.head_dim, c.intermediate_dim
        table = self.m.rope_table(self.prefix)
        self.acts = []
        for i in range(c.num_layers):
            if on_layer_start is not None:
                on_layer_start(self.prefix, i)       # ZeRO-1: this layer's parameters have arrived (training/dp.py wait_params)
            a: Dict[str, torch.Tensor] = {}
            xn = torch.empty(M, d, dtype=BF16, device=dev)
            rstd1 = torch.empty(M, dtype=F32, device=dev)
            ops.rmsnorm_fwd(x, self.w(f"layers.{i}.sa_norm.scale"), xn, rstd1, c.norm_eps)
            qkv = torch.empty(M, c.qkv_dim, dtype=BF16, device=dev)
            hq, hk = H * hd, KV * hd
            G, fused = self._group(i, "attn_in")
            rope_in_gemm = FUSE_ROPE_FWD and fuse_rope and pos is None
            if G is None and rope_in_gemm:
                # positions = arange(S) and nothing to add before the rotation: RoPE rides in the projection's epilogue
                ops.linear_rope_fwd(xn, self.w(f"layers.{i}.attn.qkv"), qkv, table, S, hq + hk, hd)
            elif fused:
                # the adapters' up-projections are extra k-steps of the same product, so the rotation still sees the sum
                a["tx_attn_in"] = G.project(xn)
                ops.gemm_kext(xn, self.w(f"layers.{i}.attn.qkv"), qkv, a["tx_attn_in"], G.Bx,
                              rope=(table, S, hq + hk, hd) if rope_in_gemm else None)
                if not rope_in_gemm:
                    ops.rope(qkv, table, S, H + KV, hd, pos=pos)
            else:
                ops.linear_fwd(xn, self.w(f"layers.{i}.attn.qkv"), qkv)
                for mod, lo_, hi_ in (("q_proj", 0, hq), ("k_proj", hq, hq + hk), ("v_proj", hq + hk, hq + 2 * hk)):
                    ad = self._lora(i, mod)
                    if ad is not None:
                        a[f"t_{mod}"] = ad.forward(xn, qkv[:, lo_:hi_])
                ops.rope(qkv, table, S, H + KV, hd, pos=pos)
            o = torch.empty(M, H * hd, dtype=BF16, device=dev)
            lse = torch.empty(B, H, S, dtype=F32, device=dev)
            ops.attn_fwd(qkv, o, lse, B, S, H, KV, hd)
            h = torch.empty(M, d, dtype=BF16, device=dev)
            G, fused = self._group(i, "attn_out")
            if fused:
                a["tx_attn_out"] = G.project(o)
                ops.gemm_kext(o, self.w(f"layers.{i}.attn.output_proj.weight"), h, a["tx_attn_out"], G.Bx, R=x)
            else:
                ops.linear_fwd(o, self.w(f"layers.{i}.attn.output_proj.weight"), h, residual=x)
                ad = self._lora(i, "output_proj")
                if ad is not None:
                    a["t_output_proj"] = ad.forward(o, h)
            hn = torch.empty(M, d, dtype=BF16, device=dev)
            rstd2 = torch.empty(M, dtype=F32, device=dev)
            ops.rmsnorm_fwd(h, self.w(f"layers.{i}.mlp_norm.scale"), hn, rstd2, c.norm_eps)
            gu = torch.empty(M, 2 * F, dtype=BF16, device=dev)     # gate/up interleaved: g0,u0,g1,u1,...
            act = torch.empty(M, F, dtype=BF16, device=dev)
            ad1, ad3 = self._lora(i, "w1"), self._lora(i, "w3")
            G, fused = self._group(i, "mlp_in")
            if FUSE_SWIGLU and G is None:
                ops.linear_swiglu_fwd(hn, self.w(f"layers.{i}.mlp.w13"), gu, act)   # activation fused into the GEMM epilogue
            elif FUSE_SWIGLU and fused:
                # adapters on w1 / w3: extra k-steps of the w13 product (Bx rows interleaved like w13), SwiGLU from the sum
                a["tx_mlp_in"] = G.project(hn)
                ops.gemm_kext(hn, self.w(f"layers.{i}.mlp.w13"), gu, a["tx_mlp_in"], G.Bx, swiglu_act=act)
            elif FUSE_SWIGLU and not any(ad is not None and ad.bias is not None for ad in (ad1, ad3)):
                # (dropout: one mask per adapter) the adapters' (alpha/r) t B^T is written FIRST, for both at once -
                # t13 = the adapters' projections side by side, as the group's Bx expects them - and the frozen product takes it
                # in through the residual port of its SwiGLU epilogue: gate/up = acc + R, act from the sum
                rp = (ad1 or ad3).r
                t13 = torch.zeros(M, G.kx, dtype=BF16, device=dev)
                for j, (mod, ad) in enumerate(G.adapters.items()):
                    a[f"t_{mod}"] = ad.project(hn, t13[:, j * rp:(j + 1) * rp])
                ops.gemm(t13, G.Bx, gu, None, alpha=(ad1 or ad3).scaling)
                ops.linear_swiglu_fwd(hn, self.w(f"layers.{i}.mlp.w13"), gu, act, residual=gu)
            else:
                ops.linear_fwd(hn, self.w(f"layers.{i}.mlp.w13"), gu)
                if ad1 is not None or ad3 is not None:
                    gv = gu.view(M, F, 2)
                    for mod, col, ad in (("w1", 0, ad1), ("w3", 1, ad3)):
                        if ad is not None:
                            tmp = torch.zeros(M, F, dtype=BF16, device=dev)
                            a[f"t_{mod}"] = ad.forward(hn, tmp)
                            gv[:, :, col] += tmp
                ops.swiglu_fwd(gu, act)
            out = torch.empty(M, d, dtype=BF16, device=dev)
            G, fused = self._group(i, "mlp_out")
            if fused:
                a["tx_mlp_out"] = G.project(act)
                ops.gemm_kext(act, self.w(f"layers.{i}.mlp.w2.weight"), out, a["tx_mlp_out"], G.Bx, R=h)
            else:
                ops.linear_fwd(act, self.w(f"layers.{i}.mlp.w2.weight"), out, residual=h)
                ad = self._lora(i, "w2")
                if ad is not None:
                    a["t_w2"] = ad.forward(act, out)
            if save:
                a.update(x=x, xn=xn, rstd1=rstd1, qkv=qkv, o=o, lse=lse, h=h, hn=hn, rstd2=rstd2, gu=gu, act=act)
                self.acts.append(a)
            x = out
        xf = torch.empty(M, d, dtype=BF16, device=dev)
        rstdf = torch.empty(M, dtype=F32, device=dev)
        ops.rmsnorm_fwd(x, self.w("norm.scale"), xf, rstdf, c.norm_eps)
        if save:
            self.final = dict(x=x, rstd=rstdf)
        return xf

    # -------------------------------------------------------------------------------------------- backward
    def backward(self, dxf: torch.Tensor, B: int, S: int, train_base: bool, alpha: float,
                 pos: Optional[torch.Tensor] = None, on_layer_done=None, acc: bool = True) -> torch.Tensor:
        """dxf = gradient w.r.t. the final-norm output.  Returns the gradient w.r.t. the stack input.
        Weight gradients are added to the gradient arena (``acc``) or overwrite what it holds (``acc=False``: the first
        backward after a lazy optimizer step, which left consumed gradients behind instead of zeros); the incoming
        gradient is already scaled, so ``alpha`` stays 1 unless a caller rescales."""
        c, dev = self.c, dxf.device
        M, d = dxf.shape
        H, KV, hd, F = c.num_heads, c.num_kv_heads, c.head_dim, c.intermediate_dim
        table = self.m.rope_table(self.prefix)
        nb = ops.lib.csm_rmsnorm_bwd_blocks()
        parts = torch.empty(nb, d, dtype=F32, device=dev) if train_base else None
        delta = torch.empty(2, B, H, S, dtype=F32, device=dev)   # attention-backward scratch (-delta, -lse log2e)

        pend_norm = []   # (per-block partial sums, scale gradient) of the layers' norms: reduced together, one launch per flush

        def norm_bwd(x, name, rstd, dy, dres, defer=False):
            dx = torch.empty(M, d, dtype=BF16, device=dev)
            if train_base and defer:
                pp = torch.empty(nb, d, dtype=F32, device=dev)
                ops.rmsnorm_bwd(x, self.w(name), rstd, dy, dx, dres, pp)
                pend_norm.append((pp, self.w(name, grad=True)))
                return dx
            ops.rmsnorm_bwd(x, self.w(name), rstd, dy, dx, dres, parts)
            if train_base:
                ops.colsum_bf16(parts, self.w(name, grad=True), accumulate=acc)
            return dx

        def flush_norms():
            if pend_norm:
                ops.colsum_bf16_multi(pend_norm, accumulate=acc)
                pend_norm.clear()

        pend = []        # deferred (dqkv, xn, dW_qkv, dh, o, dW_o, layer): the operands stay alive until their launch
        # A narrow stack (the depth decoder, d = 1024): every weight gradient but w13's is a fraction of a round of 256 x 256
        # tiles (o_proj 16, q|k|v 24, w2 128) and would go through fp32 split-K slabs + a column-sum launch each.  Deferred to
        # the end of the stack they are one launch of 160 tiles (attention, all layers) and one of 256 per two layers (w2).
        small = (train_base and DEFER_ATTN_DW > 1 and GROUP_ATTN_DW and not PAIR_DX_DW and c.embed_dim < 2048 and M % 64 == 0
                 and M >= 4096)
        pend_w2 = []     # (dy, act, dW_w2) of a narrow stack

        def flush_w2():
            if pend_w2 and not (len(pend_w2) > 1 and ops.multi_linear_dw(pend_w2, accumulate=acc, alpha=alpha)):
                for dy_, x_, g_ in pend_w2:
                    ops.linear_dw(dy_, x_, g_, accumulate=acc, alpha=alpha)
            pend_w2.clear()

        def flush_attn_dw():
            for c0 in range(0, len(pend), 6):                # at most 12 products per launch
                chunk = pend[c0:c0 + 6]
                probs = []
                for t in chunk:
                    probs += [(t[0], t[1], t[2]), (t[3], t[4], t[5])]
                if not (len(chunk) > 1 and ops.multi_linear_dw(probs, accumulate=acc, alpha=alpha)):
                    for dq_, xn_, gq_, dh_, o_, go_, _ in chunk:
                        if not ops.two_linear_dw(dq_, xn_, gq_, dh_, o_, go_, accumulate=acc, alpha=alpha):
                            ops.linear_dw(dh_, o_, go_, accumulate=acc, alpha=alpha)
                            ops.linear_dw(dq_, xn_, gq_, accumulate=acc, alpha=alpha)
            flush_w2()
            flush_norms()
            if on_layer_done is not None:
                for t in pend:
                    on_layer_done(self.prefix, t[6])
            pend.clear()

        dx = norm_bwd(self.final["x"], "norm.scale", self.final["rstd"], dxf, None)
        for i in reversed(range(c.num_layers)):
            a = self.acts[i]
            deferred = False
            hq, hk = H * hd, KV * hd
            # ---- MLP: out = h + w2(act)
            dgu = torch.empty(M, 2 * F, dtype=BF16, device=dev)
            ad = self._lora(i, "w2")
            G, fused = self._group(i, "mlp_out")
            fused = fused and "tx_mlp_out" in a
            w2_done = False
            if FUSE_SWIGLU and ad is None and train_base and (PAIR_DX_DW & 1):
                # dgrad (with the SwiGLU backward in its epilogue) and wgrad of w2 share dx and nothing else: one launch
                w2_done = ops.linear_dx_dw(dx, self.w(f"layers.{i}.mlp.w2.weight"), dgu, a["act"], self.w(f"layers.{i}.mlp.w2.weight", True),
                                           accumulate=acc, alpha=alpha, swiglu_gu=a["gu"])
            if w2_done:
                pass
            elif FUSE_SWIGLU and ad is None:
                ops.linear_dx_swiglu_bwd(dx, self.w(f"layers.{i}.mlp.w2.weight"), a["gu"], dgu)   # d(act) never stored
            elif FUSE_SWIGLU and fused:
                # d(act) = dx w2 + (s dx Bx) At^T inside one product, SwiGLU backward in its epilogue
                dts = G.backward(a["act"], dx, a["tx_mlp_out"])
                ops.gemm_kext(dx, self.w(f"layers.{i}.mlp.w2.weight"), dgu, dts, G.At, transB=True, swiglu_bwd_gu=a["gu"])
            else:
                dact = torch.empty(M, F, dtype=BF16, device=dev)
                if fused:
                    dts = G.backward(a["act"], dx, a["tx_mlp_out"])
                    ops.gemm_kext(dx, self.w(f"layers.{i}.mlp.w2.weight"), dact, dts, G.At, transB=True)
                else:
                    ops.linear_dx(dx, self.w(f"layers.{i}.mlp.w2.weight"), dact)
                    if ad is not None:
                        ad.backward(a["act"], dx, a["t_w2"], dact)
                ops.swiglu_bwd(a["gu"], dact, dgu)
                del dact
            if train_base and not w2_done:
                if small:
                    pend_w2.append((dx, a["act"], self.w(f"layers.{i}.mlp.w2.weight", True)))
                    if len(pend_w2) == 2:
                        flush_w2()
                else:
                    ops.linear_dw(dx, a["act"], self.w(f"layers.{i}.mlp.w2.weight", True), accumulate=acc, alpha=alpha)
            dhn = torch.empty(M, d, dtype=BF16, device=dev)
            G, fused = self._group(i, "mlp_in")
            fused = fused and "tx_mlp_in" in a
            if fused:
                dts = G.backward(a["hn"], dgu, a["tx_mlp_in"])
                ops.gemm_kext(dgu, self.w(f"layers.{i}.mlp.w13"), dhn, dts, G.At, transB=True)
                if train_base:
                    ops.linear_dw(dgu, a["hn"], self.w(f"layers.{i}.mlp.w13", True), accumulate=acc, alpha=alpha)
            else:
                if not (train_base and (PAIR_DX_DW & 2) and ops.linear_dx_dw(dgu, self.w(f"layers.{i}.mlp.w13"), dhn, a["hn"],
                                                                            self.w(f"layers.{i}.mlp.w13", True), accumulate=acc, alpha=alpha)):
                    ops.linear_dx(dgu, self.w(f"layers.{i}.mlp.w13"), dhn)
                    if train_base:
                        ops.linear_dw(dgu, a["hn"], self.w(f"layers.{i}.mlp.w13", True), accumulate=acc, alpha=alpha)
                for mod, col in (("w1", 0), ("w3", 1)):
                    ad = self._lora(i, mod)
                    if ad is not None:
                        ad.backward(a["hn"], dgu.view(M, F, 2)[:, :, col].contiguous(), a[f"t_{mod}"], dhn)
            del dgu
            dh = norm_bwd(a["h"], f"layers.{i}.mlp_norm.scale", a["rstd2"], dhn, dx, defer=DEFER_NORM_DW)   # + residual path
            # ---- attention: h = x + output_proj(o)
            do = torch.empty(M, H * hd, dtype=BF16, device=dev)
            group_dw = False
            G, fused = self._group(i, "attn_out")
            fused = fused and "tx_attn_out" in a
            if fused:
                dts = G.backward(a["o"], dh, a["tx_attn_out"])
                ops.gemm_kext(dh, self.w(f"layers.{i}.attn.output_proj.weight"), do, dts, G.At, transB=True)
                if train_base:
                    ops.linear_dw(dh, a["o"], self.w(f"layers.{i}.attn.output_proj.weight", True), accumulate=acc, alpha=alpha)
            else:
                if not (train_base and (PAIR_DX_DW & 4) and ops.linear_dx_dw(dh, self.w(f"layers.{i}.attn.output_proj.weight"), do, a["o"],
                                                                            self.w(f"layers.{i}.attn.output_proj.weight", True),
                                                                            accumulate=acc, alpha=alpha)):
                    ops.linear_dx(dh, self.w(f"layers.{i}.attn.output_proj.weight"), do)
                    # (the output projection's dW waits for the q|k|v projection's below when the two can share one launch)
                    group_dw = train_base and GROUP_ATTN_DW and not (PAIR_DX_DW & 8) and M % 64 == 0 and M >= 4096 and (c.embed_dim >= 2048 or small)
                    if train_base and not group_dw:
                        ops.linear_dw(dh, a["o"], self.w(f"layers.{i}.attn.output_proj.weight", True), accumulate=acc, alpha=alpha)
                ad = self._lora(i, "output_proj")
                if ad is not None:
                    ad.backward(a["o"], dh, a["t_output_proj"], do)
            dqkv = torch.empty(M, c.qkv_dim, dtype=BF16, device=dev)
            if pos is None and FUSE_ROPE_BWD:     # positions = arange(S): the RoPE backward rides in the dQ / dK epilogues
                ops.attn_bwd(a["qkv"], a["o"], do, a["lse"], dqkv, delta, B, S, H, KV, hd, rope_table=table)
            else:
                ops.attn_bwd(a["qkv"], a["o"], do, a["lse"], dqkv, delta, B, S, H, KV, hd)
                ops.rope(dqkv, table, S, H + KV, hd, pos=pos, inverse=True)
            dxn = torch.empty(M, d, dtype=BF16, device=dev)
            G, fused = self._group(i, "attn_in")
            fused = fused and "tx_attn_in" in a
            if fused:
                dts = G.backward(a["xn"], dqkv, a["tx_attn_in"])
                ops.gemm_kext(dqkv, self.w(f"layers.{i}.attn.qkv"), dxn, dts, G.At, transB=True)
                if train_base:
                    if group_dw:
                        ops.linear_dw(dh, a["o"], self.w(f"layers.{i}.attn.output_proj.weight", True), accumulate=acc, alpha=alpha)
                    ops.linear_dw(dqkv, a["xn"], self.w(f"layers.{i}.attn.qkv", True), accumulate=acc, alpha=alpha)
            else:
                if not (train_base and (PAIR_DX_DW & 8) and ops.linear_dx_dw(dqkv, self.w(f"layers.{i}.attn.qkv"), dxn, a["xn"],
                                                                            self.w(f"layers.{i}.attn.qkv", True), accumulate=acc, alpha=alpha)):
                    ops.linear_dx(dqkv, self.w(f"layers.{i}.attn.qkv"), dxn)
                    if train_base and group_dw and DEFER_ATTN_DW > 1:
                        pend.append((dqkv, a["xn"], self.w(f"layers.{i}.attn.qkv", True), dh, a["o"],
                                     self.w(f"layers.{i}.attn.output_proj.weight", True), i))
                        deferred = True
                    elif train_base and group_dw and ops.two_linear_dw(dqkv, a["xn"], self.w(f"layers.{i}.attn.qkv", True), dh, a["o"],
                                                                     self.w(f"layers.{i}.attn.output_proj.weight", True), accumulate=acc, alpha=alpha):
                        pass
                    elif train_base:
                        if group_dw:
                            ops.linear_dw(dh, a["o"], self.w(f"layers.{i}.attn.output_proj.weight", True), accumulate=acc, alpha=alpha)
                        ops.linear_dw(dqkv, a["xn"], self.w(f"layers.{i}.attn.qkv", True), accumulate=acc, alpha=alpha)
                for mod, lo_, hi_ in (("q_proj", 0, hq), ("k_proj", hq, hq + hk), ("v_proj", hq + hk, hq + 2 * hk)):
                    ad = self._lora(i, mod)
                    if ad is not None:
                        ad.backward(a["xn"], dqkv[:, lo_:hi_], a[f"t_{mod}"], dxn)
            dx = norm_bwd(a["x"], f"layers.{i}.sa_norm.scale", a["rstd1"], dxn, dh, defer=DEFER_NORM_DW)
            self.acts[i] = None
            if deferred:
                # (layer 0 goes alone: what is launched last is what a data-parallel all-reduce cannot hide behind compute;
                #  a narrow stack keeps everything to its end - the wide stack's backward follows and hides its all-reduce)
                if not small and (len(pend) >= DEFER_ATTN_DW or i <= 1):
                    flush_attn_dw()
            else:
                if pend or pend_w2:
                    flush_attn_dw()          # (mixed stacks: nothing of an earlier layer may stay pending behind this layer's hook)
                flush_norms()                # this layer's two norms in one launch
                if on_layer_done is not None:
                    on_layer_done(self.prefix, i)
        if pend or pend_w2:
            flush_attn_dw()
        flush_norms()
        self.acts = []
        return dx


class Engine:
    def __init__(self, model):
        self.m = model
        self.backbone = _Stack(model, "backbone")
        self.decoder = _Stack(model, "decoder")
        self.saved = None
        self.grad_hook = None   # called as hook(prefix, layer) when a layer's weight gradients are final (DP overlap)
        # called as hook(prefix, layer) right before a forward first reads that bucket's parameters, hook(None, None) before
        # anything else reads parameters (ZeRO-1: the updated shards are all-gathered behind the optimiser step, dp.py)
        self.param_hook = None
        # a list here makes the cache-free generate path append the logits [B, V] every code is drawn from (parity tests)
        self.capture_logits = None
        self._consts = {}               # index tensors that depend on shapes only (built once, reused every step)

    def _need(self, group=None, layer=None):
        if self.param_hook is not None:
            self.param_hook(group, layer)

    # -------------------------------------------------------------------------------------------- loss forward
    def forward_loss(self, tokens: torch.Tensor, masks: torch.Tensor, targets: torch.Tensor, semantic_weight: float,
                     acoustic_weight: float, save: bool, acoustic_rows: Optional[torch.Tensor] = None):
        """Forward of ``compute_loss``.  Returns (total, semantic, acoustic) as 0-d fp32 GPU tensors."""
        m, a = self.m, self.m.args
        dev = m.device
        B, S, K1 = tokens.shape
        K, V, Vp = a.audio_num_codebooks, a.audio_vocab_size, m.vocab_pad
        assert K1 == K + 1, f"tokens last dim must be {K + 1}"
        if targets.shape[1] < S - 1:
            raise ValueError(f"target_audio_tokens has {targets.shape[1]} frames, needs at least seq_len-1 = {S - 1}")
        if S > m.bb.max_seq_len:
            raise ValueError(f"sequence length {S} exceeds max_seq_len {m.bb.max_seq_len}")
        M, d = B * S, m.bb.embed_dim
        tk = tokens.reshape(M, K1).to(device=dev, dtype=torch.int64).contiguous()
        mk = masks.reshape(M, K1).to(device=dev, dtype=torch.uint8).contiguous()
        ign = getattr(m, "target_ignore_index", None)     # None: reference behaviour, every row counts (utils.py:102-105)
        self._validate_batch(tokens, masks, targets, ign)
        tg = targets.to(device=dev, dtype=torch.int64)

        if m.lora is not None and LORA_FUSE:
            m.lora.refresh()
        h0 = torch.empty(M, d, dtype=BF16, device=dev)
        self._need("embeddings", -1)
        ops.embed_fwd(tk, mk, m.block("text_embeddings.weight"), m.block("audio_embeddings.weight"), h0, V)
        hidden = self.backbone.forward(h0, B, S, save, on_layer_start=self.param_hook)

        # codebook-0 head + CE over positions [0, S-1) of every sequence (reference utils.py:96-106)
        logits = torch.empty(M, Vp, dtype=F32, device=dev)
        self._need("other", -1)
        ops.linear_fwd(hidden, m.block("codebook0_head.padded"), logits)
        t0 = torch.full((B, S), -1, dtype=torch.int64, device=dev)
        t0[:, :S - 1] = tg[:, :S - 1, 0]
        t0 = t0.reshape(M).contiguous()
        # with an ignore index the mean runs over the labelled rows only (torch CE semantics); the kernel skips t < 0
        n_sem = B * (S - 1) if ign is None else max(1, int((t0 >= 0).sum()))
        rows_loss = torch.empty(M, dtype=F32, device=dev)
        ops.ce_fwd_bwd(logits, t0, rows_loss, None, V, 0.0)
        sem = torch.empty(1, dtype=F32, device=dev)
        ops.reduce_sum(rows_loss, sem, 1.0 / n_sem)

        ac = torch.zeros(1, dtype=F32, device=dev)
        dec = None
        if m.acoustic_mode != "off":
            rows = self._acoustic_rows(B, S, acoustic_rows, t0 if ign is not None else None)
            dec = self._decoder_forward(hidden, rows, tg, B, S, save)
            ops.reduce_sum(dec["rows_loss"], ac, 1.0 / dec["n_rows"])
        total = semantic_weight * sem + acoustic_weight * ac
        if save:
            self.saved = dict(B=B, S=S, tk=tk, mk=mk, hidden=hidden, logits=logits, t0=t0, n_sem=n_sem, dec=dec,
                              sw=float(semantic_weight), aw=float(acoustic_weight))
        return total[0], sem[0], ac[0]

    def _validate_batch(self, tokens, masks, targets, ign):
        """Range checks of the integer inputs: the kernels index embedding tables, logits rows and gradient tables with
        them unchecked, so an id outside its table is an out-of-bounds device access (the reference's nn.Embedding /
        F.cross_entropy would assert).  Host batches (what a DataLoader hands over) are checked every time - min / max of
        a few hundred thousand integers on the CPU, no device sync.  Device-resident batches (benchmarks, gradient
        accumulation over one batch) cost a device sync, so a verdict is remembered for that exact tensor version."""
        a = self.m.args
        V, TV, K = a.audio_vocab_size, a.text_vocab_size, a.audio_num_codebooks

        def check(tk, mk, tg):
            if int(tg.max()) >= V or (ign is None and int(tg.min()) < 0):
                raise ValueError("target_audio_tokens out of range for audio_vocab_size")
            if ign is not None and bool(((tg < 0) & (tg != ign)).any()):
                raise ValueError(f"negative target_audio_tokens other than the ignore index {ign}")
            live = mk.bool()
            au, tx = tk[..., :K], tk[..., K]
            au_live, tx_live = au[live[..., :K]], tx[live[..., K]]
            if au_live.numel() and (int(au_live.min()) < 0 or int(au_live.max()) >= V):
                raise ValueError("input_tokens: audio code out of range for audio_vocab_size")
            if tx_live.numel() and (int(tx_live.min()) < 0 or int(tx_live.max()) >= TV):
                raise ValueError("input_tokens: text id out of range for text_vocab_size")

        if not (tokens.is_cuda or targets.is_cuda):
            check(tokens, masks, targets)
            return
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (tokens, masks, targets)) + (ign,)
        if key != getattr(self, "_validated_batch", None):
            check(tokens, masks, targets)
            self._validated_batch = key

    def _acoustic_rows(self, B, S, rows, t0=None):
        """Flattened (b*S + p) indices, p < S-1, of the positions whose frame trains the depth decoder.  ``t0`` (the
        flattened codebook-0 labels, negative = ignored) restricts the choice to labelled frames."""
        dev = self.m.device
        if rows is not None:
            r = rows.to(device=dev, dtype=torch.int64)            # given over the B*(S-1) row space of compute_loss
            b, p = r // (S - 1), r % (S - 1)
            r = b * S + p
            if t0 is not None:
                r = r[t0[r] >= 0]
            return r.to(torch.int32).contiguous()
        if t0 is None and self.m.acoustic_mode != "all" and ROWS_ON_HOST:
            # a random subset of all frames: drawn on the host (torch's CPU generator) and copied over - one small copy instead of
            # ~20 launches of device randperm / sort / index kernels in front of the depth decoder
            n_all = B * (S - 1)
            n = max(1, int(round(n_all * self.m.acoustic_fraction)))
            perm = torch.randperm(n_all)[:n].sort().values
            return ((perm // (S - 1)) * S + perm % (S - 1)).to(torch.int32).to(dev, non_blocking=True)
        ck = ("allr", B, S)
        allr = self._consts.get(ck)
        if allr is None:
            allr = self._consts[ck] = (torch.arange(B, device=dev)[:, None] * S + torch.arange(S - 1, device=dev)[None, :]).reshape(-1)
        if t0 is not None:
            allr = allr[t0[allr] >= 0]
            if allr.numel() == 0:
                raise ValueError("no labelled audio frame in the batch")
        if self.m.acoustic_mode == "all":
            return allr.to(torch.int32).contiguous()
        n = max(1, int(round(allr.numel() * self.m.acoustic_fraction)))
        perm = torch.randperm(allr.numel(), device=dev)[:n].sort().values
        return allr[perm].to(torch.int32).contiguous()

    def _decoder_forward(self, hidden, rows, tg, B, S, save):
        m, a = self.m, self.m.args
        dev = m.device
        K, V, Vp = a.audio_num_codebooks, a.audio_vocab_size, m.vocab_pad
        N = rows.numel()
        d, dd = m.bb.embed_dim, m.dc.embed_dim
        codes = tg[:, :S, :].reshape(-1, K) if tg.shape[1] >= S else None
        if codes is None:   # targets may be exactly S-1 long: index through (b, p)
            b, p = rows.long() // S, rows.long() % S
            codes = tg[b, p]
        else:
            codes = codes[rows.long()]
        codes = codes.contiguous()
        seq = torch.empty(N * K, d, dtype=BF16, device=dev)
        ops.decoder_input_fwd(hidden, rows, codes, m.block("audio_embeddings.weight"), seq, V)
        x0 = torch.empty(N * K, dd, dtype=BF16, device=dev)
        ops.linear_fwd(seq, m.block("projection.weight"), x0)
        xf = self.decoder.forward(x0, N, K, save, on_layer_start=self.param_hook)   # [N*K, dd]
        logits = torch.empty(K - 1, N, Vp, dtype=F32, device=dev)
        ah = m.block("audio_head.padded")                                           # [K-1, dd, Vp]
        xf2 = xf.view(N, K * dd)
        ops.gemm(xf2[:, dd:2 * dd], ah[0], logits[0], None, False, True, batch=K - 1, sA=dd, sB=dd * Vp, sC=N * Vp)
        tgt = codes[:, 1:].t().contiguous().reshape(-1)                             # [(K-1)*N]
        rows_loss = torch.empty((K - 1) * N, dtype=F32, device=dev)
        ops.ce_fwd_bwd(logits.view(-1, Vp), tgt, rows_loss, None, V, 0.0)
        out = dict(rows_loss=rows_loss, n_rows=(K - 1) * N)
        if save:
            out.update(rows=rows, codes=codes, seq=seq, xf=xf, logits=logits, tgt=tgt, N=N)
        return out

    # -------------------------------------------------------------------------------------------- backward
    def backward(self, gscale: float = 1.0):
        """Gradients of ``gscale * total`` accumulated into the gradient arena (and LoRA gradient tensors).
        Which groups receive weight gradients follows ``model.trainable`` (freeze flags / LoRA)."""
        if self.saved is None:
            raise RuntimeError("backward() without a saved forward (was the forward run under no_grad?)")
        s, m, a = self.saved, self.m, self.m.args
        self.saved = None
        dev = m.device
        B, S = s["B"], s["S"]
        K, V, Vp = a.audio_num_codebooks, a.audio_vocab_size, m.vocab_pad
        M, d, dd = B * S, m.bb.embed_dim, m.dc.embed_dim
        tr = m.trainable
        train_embeddings, train_other = tr["embeddings"], tr["other"]
        m.ensure_grads()
        # Gradient-buffer states (see Model.grad_state): "live" buffers are accumulated into; "stale" ones hold the
        # gradients a lazy optimizer step consumed and are overwritten by this backward (the weight-gradient GEMMs then
        # neither read their output nor need a zero fill) - or zeroed here if this backward will not write them.
        gst = m.grad_state
        dec_runs = s["dec"] is not None and s["aw"] != 0.0
        writes = {"backbone": tr["backbone"], "decoder": dec_runs and tr["decoder"], "other": train_other,
                  "embeddings": train_embeddings}
        acc = {}
        for g_, w_ in writes.items():
            acc[g_] = gst[g_] == "live"
            if gst[g_] == "stale" and (not w_ or g_ == "embeddings"):
                o_, n_ = m.group_range(g_)
                m.grad_arena[o_:o_ + n_].zero_()
                gst[g_] = "zero"
            if w_:
                gst[g_] = "live"
        if train_other and not acc["other"] and not dec_runs:          # the decoder-side tensors of "other" get no gradient
            m.block("projection.weight", True).zero_()
            m.block("audio_head.padded", True).zero_()
        hidden = s["hidden"]
        dseq = None
        dec_pos0 = None
        # ---- depth decoder (acoustic term)
        dec = s["dec"]
        if dec is not None and s["aw"] != 0.0:
            N = dec["N"]
            dl = torch.empty(K - 1, N, Vp, dtype=BF16, device=dev)
            ops.ce_fwd_bwd(dec["logits"].view(-1, Vp), dec["tgt"], dec["rows_loss"], dl.view(-1, Vp), V,
                           gscale * s["aw"] / dec["n_rows"])
            ah = m.block("audio_head.padded")
            dxf = torch.zeros(N, K * dd, dtype=BF16, device=dev)                   # position 0 has no head: stays 0
            ops.gemm(dl[0], ah[0], dxf[:, dd:2 * dd], None, False, False, batch=K - 1, sA=N * Vp, sB=dd * Vp, sC=dd)
            xf2 = dec["xf"].view(N, K * dd)
            if train_other:
                gah = m.block("audio_head.padded", True)
                ops.gemm(xf2[:, dd:2 * dd], dl[0], gah[0], gah[0] if acc["other"] else None, True, True, batch=K - 1, sA=dd,
                         sB=N * Vp, sC=dd * Vp, sR=dd * Vp)
            del dl
            dx0 = self.decoder.backward(dxf.view(N * K, dd), N, K, tr["decoder"], 1.0, on_layer_done=self.grad_hook,
                                        acc=acc["decoder"])
            dseq = torch.empty(N * K, d, dtype=BF16, device=dev)
            ops.linear_dx(dx0, m.block("projection.weight"), dseq)
            if train_other:
                ops.linear_dw(dx0, dec["seq"], m.block("projection.weight", True), accumulate=acc["other"])
            dec_pos0 = (dseq, dec["rows"], K)          # position 0 of every frame is the backbone state: scattered below
            if self.grad_hook is not None:
                self.grad_hook("decoder", -1)

        # ---- codebook-0 head (semantic term)
        dlog = torch.empty(M, Vp, dtype=BF16, device=dev)
        rows_loss = torch.empty(M, dtype=F32, device=dev)
        ops.ce_fwd_bwd(s["logits"], s["t0"], rows_loss, dlog, V, gscale * s["sw"] / s["n_sem"])
        dhid = torch.empty(M, d, dtype=BF16, device=dev)
        ops.linear_dx(dlog, m.block("codebook0_head.padded"), dhid)
        if train_other:
            ops.linear_dw(dlog, hidden, m.block("codebook0_head.padded", True), accumulate=acc["other"])
        del dlog
        if dec_pos0 is not None:
            ops.rows_add_bf16(dhid, dec_pos0[1], dec_pos0[0], dec_pos0[2])   # rows are unique: plain bf16 read-modify-write
        if self.grad_hook is not None:
            self.grad_hook("other", -1)

        # ---- backbone
        dh0 = self.backbone.backward(dhid, B, S, tr["backbone"], 1.0, on_layer_done=self.grad_hook, acc=acc["backbone"])
        if train_embeddings:
            self._embedding_backward(s, dh0, dseq if (dec is not None and s["aw"] != 0.0) else None)
        if self.grad_hook is not None:
            self.grad_hook("embeddings", -1)

    def _embedding_backward(self, s, dh0, dseq):
        """d(text_embeddings), d(audio_embeddings): every (embedding row, gradient source row) occurrence - live slots of
        the backbone input and, when the decoder was trained, positions 1..K-1 of its input - is sorted by embedding row
        (torch.sort: no host sync) and reduced by ``csm_embed_bwd_sorted`` in a fixed order, straight into the bf16
        gradient arena."""
        m, a = self.m, self.m.args
        dev = m.device
        K, V = a.audio_num_codebooks, a.audio_vocab_size
        TV = a.text_vocab_size
        n_rows = TV + K * V
        tk, mk = s["tk"], s["mk"]
        M = tk.shape[0]
        # shape-only index tensors (slot offsets, source ids) are built once per shape, not once per step
        ck = ("embbwd", M, K, V, TV)
        c = self._consts.get(ck)
        if c is None:
            slot = torch.arange(K + 1, device=dev)
            c = self._consts[ck] = dict(off=torch.where(slot < K, TV + slot * V, torch.zeros_like(slot)),            # [K+1] row offset of a slot
                                        src=torch.arange(M, device=dev).unsqueeze(1).expand(M, K + 1).reshape(-1).contiguous())
        rows = torch.where(mk.bool(), tk + c["off"], torch.full_like(tk, n_rows)).reshape(-1)   # [M (K+1)]; masked-out slots -> padding id
        src = c["src"]
        n_src = M
        if dseq is not None:
            codes, N = s["dec"]["codes"], s["dec"]["N"]
            dk = ("embbwd_dec", M, N, K, V, TV)
            d = self._consts.get(dk)
            if d is None:
                d = self._consts[dk] = dict(off=TV + torch.arange(K - 1, device=dev) * V,
                                            src=(M + torch.arange(N, device=dev).unsqueeze(1) * K + torch.arange(1, K, device=dev)).reshape(-1))
            rows = torch.cat([rows, (codes[:, :K - 1] + d["off"]).reshape(-1)])
            src = torch.cat([src, d["src"]])
            n_src = M + N * K
        if n_src <= (1 << 20) and EMB_KEYSORT:
            # (row, source) pairs are unique, so ONE key sort gives the order a stable sort by row gives (sources ascend within a
            # row exactly as the occurrences were listed) - a radix sort of 64-bit keys instead of a stable merge sort (18 launches)
            # plus two gathers
            key = torch.sort((rows << 20) | src).values
            rows_s, src_s = key >> 20, key & ((1 << 20) - 1)
        else:
            order = torch.argsort(rows, stable=True)
            rows_s, src_s = rows[order].contiguous(), src[order].contiguous()
        ops.embed_bwd_sorted(rows_s, src_s, dh0, dseq,
                             m.block("text_embeddings.weight", True), m.block("audio_embeddings.weight", True))

    # -------------------------------------------------------------------------------------------- generation
    @torch.no_grad()
    def hidden_states(self, tokens: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
        """Backbone hidden states [B, S, D] (bf16) for full sequences - used by generation and by parity tests."""
        m, a = self.m, self.m.args
        B, S, K1 = tokens.shape
        M = B * S
        tk = tokens.reshape(M, K1).to(device=m.device, dtype=torch.int64).contiguous()
        mk = masks.reshape(M, K1).to(device=m.device, dtype=torch.uint8).contiguous()
        h0 = torch.empty(M, m.bb.embed_dim, dtype=BF16, device=m.device)
        self._need()
        ops.embed_fwd(tk, mk, m.block("text_embeddings.weight"), m.block("audio_embeddings.weight"), h0, a.audio_vocab_size)
        return self.backbone.forward(h0, B, S, False).view(B, S, -1)

    @torch.no_grad()
    def generate_frame(self, tokens, tokens_mask, input_pos, temperature, topk, noise=None):
        """Reference model.py:140-195: backbone position(s) -> c0 -> 31 depth-decoder steps, against KV caches.

        The prompt (``input_pos`` starting at 0) is prefilled with the training forward kernels and its post-RoPE K/V
        rows are copied into the backbone cache; every later call is a single-position decode step (matrix-vector
        kernels + cache attention).  The decoder cache is reset every frame, as the reference does (model.py:181).
        ``model.use_kv_cache = False`` selects the cache-free prefix-recompute path (same arithmetic, kept as a check).
        """
        m, a = self.m, self.m.args
        self._need()
        if not getattr(m, "use_kv_cache", True):
            return self._generate_frame_recompute(tokens, tokens_mask, input_pos, temperature, topk, noise)
        dev = m.device
        K, V, Vp = a.audio_num_codebooks, a.audio_vocab_size, m.vocab_pad
        d, dd = m.bb.embed_dim, m.dc.embed_dim
        tokens, tokens_mask = tokens.to(dev), tokens_mask.to(dev)
        Bn, Sn, K1 = tokens.shape
        first = int(input_pos[0, 0]) == 0
        st = getattr(m, "_decode_state", None)
        if not first and (st is None or st.B != Bn or st.cur < 0):
            raise RuntimeError("generate_frame: a non-first call (input_pos > 0) needs the state of a prompt prefilled with the "
                               "same batch size (call it with input_pos starting at 0 first)")
        if first:
            st = m._decode_state = DecodeState(self, Bn)
        if first:
            last_h = st.prefill(tokens, tokens_mask)
            return self._frame_tail(st, last_h, temperature, topk, noise)
        if Sn != 1:
            raise ValueError("generate_frame with caches: after the prompt, feed one position per call")
        if getattr(m, "use_hip_graph", True):
            return st.graph_frame(tokens, tokens_mask, temperature, topk, noise)
        if st.cur + 1 >= m.bb.max_seq_len:          # the decode kernels index LDS and the KV caches with the position
            raise ValueError("sequence exceeds max_seq_len")
        st.cur += 1
        return self._decode_frame(st, tokens, tokens_mask, temperature, topk, noise)

    @torch.no_grad()
    def generate_first_frames(self, tokens_list, masks_list, temperature, topk, noise=None):
        """Batched generation (up to 4 utterances): prefill B prompts of different lengths and sample the
        first frame of each; later frames go through ``generate_frame`` with ``[B, 1, K+1]`` tokens and a non-zero
        ``input_pos``, exactly as for one utterance."""
        m = self.m
        self._need()
        st = m._decode_state = DecodeState(self, len(tokens_list))
        last_h = st.prefill_ragged(tokens_list, masks_list)
        return self._frame_tail(st, last_h, temperature, topk, noise)

    def _decode_frame(self, st, tokens, tokens_mask, temperature, topk, noise):
        """One decode frame, eagerly."""
        st.fill_noise(noise)
        return self._decode_frame_body(st, tokens, tokens_mask, temperature, topk)

    def _decode_frame_body(self, st, tokens, tokens_mask, temperature, topk):
        """One decode frame with no host-side dependence on device data and no random draw: this is the body a HIP graph
        captures (the frame's Exp(1) noise sits in ``st.noise_buf``, filled before the body runs / the graph replays)."""
        last_h = st.backbone_step(tokens, tokens_mask)
        return self._frame_tail_body(st, last_h, temperature, topk)

    def _frame_tail(self, st, last_h, temperature, topk, noise):
        st.fill_noise(noise)
        return self._frame_tail_body(st, last_h, temperature, topk)

    def _frame_tail_body(self, st, last_h, temperature, topk):
        m, a = self.m, self.m.args
        K, V = a.audio_num_codebooks, a.audio_vocab_size
        from .models.model import sample_topk

        qall = st.noise_buf          # all Exp(1) draws of the frame (the reference draws them one codebook at a time, model.py:79-82)

        def draw(lg, i):
            return sample_topk(lg[:, :V], topk, temperature, qall[i])

        ops.gemv(last_h, m.block("codebook0_head.padded"), st.logits)
        samples = [draw(st.logits, 0)]
        dnorm = m.block("decoder.norm.scale")
        for i in range(K):
            # decoder position i: input = backbone state (i = 0) or the embedding of code i-1; the stack's final norm is
            # applied inside the head product (position 0 has no head)
            hres = (st.decoder_step(last_h, i, final_norm=False) if i == 0 else
                    st.decoder_step(None, i, code=samples[-1].view(-1), final_norm=False))
            if i >= 1:
                # [V][d'] copy of audio_head[i-1]: row-per-wave GEMV
                ops.gemv_ex(hres, st.head_t[i - 1], st.logits, norm_scale=dnorm, eps=m.dc.norm_eps)
                samples.append(draw(st.logits, i))
        return torch.cat(samples, dim=1)

    @torch.no_grad()
    def _generate_frame_recompute(self, tokens, tokens_mask, input_pos, temperature, topk, noise=None):
        """Reference model.py:140-195.  The KV state is the token history (prefix recompute)."""
        m, a = self.m, self.m.args
        dev = m.device
        K, V, Vp = a.audio_num_codebooks, a.audio_vocab_size, m.vocab_pad
        d, dd = m.bb.embed_dim, m.dc.embed_dim
        tokens = tokens.to(dev)
        tokens_mask = tokens_mask.to(dev)
        first = int(input_pos[0, 0]) == 0
        if first or m._gen_hist is None:
            hist_t, hist_m = tokens, tokens_mask
        else:
            hist_t = torch.cat([m._gen_hist[0], tokens], dim=1)
            hist_m = torch.cat([m._gen_hist[1], tokens_mask], dim=1)
        m._gen_hist = (hist_t, hist_m)
        Bn = hist_t.shape[0]
        hidden = self.hidden_states(hist_t, hist_m)                 # [B, S, d]
        last_h = hidden[:, -1, :].contiguous()                      # [B, d]
        logits = torch.empty(Bn, Vp, dtype=F32, device=dev)
        ops.linear_fwd(last_h, m.block("codebook0_head.padded"), logits)
        from .models.model import sample_topk

        def draw(lg, i):
            q = None if noise is None else noise[i].to(dev)
            if self.capture_logits is not None:                     # (parity tests: the logits each code of the frame was drawn from)
                self.capture_logits.append(lg[:, :V].clone())
            return sample_topk(lg[:, :V], topk, temperature, q)

        c0 = draw(logits, 0)                                         # [B,1] int32
        samples = [c0]
        aemb = m.block("audio_embeddings.weight")
        seq = [last_h.unsqueeze(1), aemb[c0.long() + 0 * V]]        # [B,1,d] each
        ah = m.block("audio_head.padded")
        for i in range(1, K):
            cur = torch.cat(seq, dim=1)                              # [B, L, d]
            L = cur.shape[1]
            x0 = torch.empty(Bn * L, dd, dtype=BF16, device=dev)
            ops.linear_fwd(cur.reshape(Bn * L, d).contiguous(), m.block("projection.weight"), x0)
            # (fuse_rope=False: round q/k to bf16 before the rotation, exactly as the decode kernels of the KV-cache path do, so
            #  that the two paths stay comparable bit for bit on the frame they both compute from scratch)
            xf = self.decoder.forward(x0, Bn, L, False, fuse_rope=False).view(Bn, L, dd)
            lg = torch.empty(Bn, Vp, dtype=F32, device=dev)
            ops.gemm(xf[:, -1, :].contiguous(), ah[i - 1], lg, None, False, True)
            ci = draw(lg, i)
            samples.append(ci)
            seq.append(aemb[ci.long() + i * V])
        return torch.cat(samples, dim=1)


class _DecodeStack:
    """KV caches + preallocated single-position buffers of one stack."""

    def __init__(self, stack: _Stack, B: int, s_max: int):
        c, dev = stack.c, stack.m.device
        self.stack, self.B, self.s_max = stack, B, s_max
        H, KV, hd, F, d = c.num_heads, c.num_kv_heads, c.head_dim, c.intermediate_dim, c.embed_dim
        z = lambda *shape, dt=BF16: torch.empty(*shape, dtype=dt, device=dev)   # noqa: E731
        self.k = [torch.zeros(B, KV, s_max, hd, dtype=BF16, device=dev) for _ in range(c.num_layers)]
        self.v = [torch.zeros(B, KV, s_max, hd, dtype=BF16, device=dev) for _ in range(c.num_layers)]
        self.pos = torch.zeros(B, dtype=torch.int32, device=dev)
        self.xn, self.qkv, self.o, self.h, self.hn = z(B, d), z(B, c.qkv_dim), z(B, H * hd), z(B, d), z(B, d)
        self.gu, self.act, self.xa, self.xb, self.xf = z(B, 2 * F), z(B, F), z(B, d), z(B, d), z(B, d)
        # four launches per layer instead of five where the cache is short (csm_gemv_attn_bf16: S_max <= 64, head_dim 128).  Only
        # for one utterance: every workgroup of the fused launch recomputes the attention of all (row, head) pairs, which costs
        # more than the launch it saves from two rows on (measured: 149 vs 147 frames/s at B = 1, 331 vs 402 aggregate at B = 4)
        import os
        self.fuse_attn = B == 1 and s_max <= 64 and hd == 128 and os.environ.get("CSM_DECODE_FUSE_ATTN", "1") == "1"
        # the position as a host integer where the caller knows it (the depth decoder: step i is at position i): see ops.gemv_attn
        self.pos_host = None

    def fill_from(self, acts, B, S):
        """Copy the post-RoPE K / V rows of a prefilled prompt into the caches."""
        c = self.stack.c
        H, KV, hd = c.num_heads, c.num_kv_heads, c.head_dim
        for i, a in enumerate(acts):
            qkv = a["qkv"].view(B, S, -1)
            self.k[i][:, :, :S] = qkv[:, :, H * hd:(H + KV) * hd].reshape(B, S, KV, hd).permute(0, 2, 1, 3)
            self.v[i][:, :, :S] = qkv[:, :, (H + KV) * hd:].reshape(B, S, KV, hd).permute(0, 2, 1, 3)

    def fill_row(self, acts, b, S):
        """The same for ONE sequence (a [1, S] prefill) into batch row ``b`` of the caches: ragged batched prompts."""
        c = self.stack.c
        H, KV, hd = c.num_heads, c.num_kv_heads, c.head_dim
        for i, a in enumerate(acts):
            qkv = a["qkv"].view(S, -1)
            self.k[i][b, :, :S] = qkv[:, H * hd:(H + KV) * hd].reshape(S, KV, hd).permute(1, 0, 2)
            self.v[i][b, :, :S] = qkv[:, (H + KV) * hd:].reshape(S, KV, hd).permute(1, 0, 2)

    def step(self, x: torch.Tensor, final_norm: bool = True) -> torch.Tensor:
        """One position per batch row at ``self.pos`` (device int32).  x [B, d] -> final-normed hidden [B, d]
        (``final_norm=False``: the un-normed residual stream, for a caller that fuses the norm into its next product)."""
        st, c = self.stack, self.stack.c
        H, KV, hd = c.num_heads, c.num_kv_heads, c.head_dim
        table = st.m.rope_table(st.prefix)
        cur, nxt = x, self.xa
        for i in range(c.num_layers):
            # five launches per layer (four in the depth decoder): the norms ride in the prologue of the following matrix-vector
            # product, RoPE and the cache append inside the attention kernel, SwiGLU in the epilogue of the w13 product
            ops.gemv_ex(cur, st.w(f"layers.{i}.attn.qkv"), self.qkv, norm_scale=st.w(f"layers.{i}.sa_norm.scale"), eps=c.norm_eps)
            if self.fuse_attn:
                # (depth decoder: <= 32 cached positions - the attention rides in the prologue of the output projection)
                ops.gemv_attn(self.qkv, self.k[i], self.v[i], self.pos, table, st.w(f"layers.{i}.attn.output_proj.weight"), self.h, cur,
                              H, KV, hd, pos_host=self.pos_host)
            else:
                ops.attn_decode_rope(self.qkv, self.k[i], self.v[i], self.o, self.pos, table, H, KV, hd, pos_host=self.pos_host)
                ops.gemv(self.o, st.w(f"layers.{i}.attn.output_proj.weight"), self.h, residual=cur)
            ops.gemv_ex(self.h, st.w(f"layers.{i}.mlp.w13"), self.act, norm_scale=st.w(f"layers.{i}.mlp_norm.scale"), eps=c.norm_eps,
                        swiglu=True)
            ops.gemv(self.act, st.w(f"layers.{i}.mlp.w2.weight"), nxt, residual=self.h)
            cur, nxt = nxt, (self.xb if nxt is self.xa else self.xa)
        if not final_norm:
            return cur
        ops.rmsnorm_fwd(cur, st.w("norm.scale"), self.xf, None, c.norm_eps)
        return self.xf


class DecodeState:
    """Everything ``generate_frame`` keeps between calls: the two stacks' caches and small persistent buffers."""

    def __init__(self, engine: "Engine", B: int):
        m = engine.m
        if m.lora is not None:
            raise NotImplementedError("generate with un-merged LoRA adapters: call model.merge_lora_weights() first")
        if B > 4:
            raise ValueError("the decode kernels handle up to 4 sequences at a time")
        self.e, self.B = engine, B
        dev = m.device
        self.bb = _DecodeStack(engine.backbone, B, m.bb.max_seq_len)
        self.dc = _DecodeStack(engine.decoder, B, m.args.audio_num_codebooks)
        self.h0 = torch.empty(B, m.bb.embed_dim, dtype=BF16, device=dev)
        self.proj = torch.empty(B, m.dc.embed_dim, dtype=BF16, device=dev)
        self.logits = torch.empty(B, m.vocab_pad, dtype=F32, device=dev)
        self.dpos = [torch.full((B,), i, dtype=torch.int32, device=dev) for i in range(m.args.audio_num_codebooks)]
        self.cur = -1
        self.graph, self.graph_key, self.warm = None, None, 0
        # the frame's Exp(1) draws [K, B, V]: a PERSISTENT buffer, refilled before every frame outside the captured graph -
        # so a replayed frame can be given the same noise as an eager one (parity tests) or fresh draws (generation)
        self.noise_buf = torch.empty(m.args.audio_num_codebooks, B, m.args.audio_vocab_size, dtype=F32, device=dev)
        # audio_head is stored [K-1][d'][V] (reference layout, a K-major matrix for x @ W); the decode path wants one
        # output row per wave, so it keeps a [K-1][V][d'] copy made from the current weights when the state is created
        self.head_t = m.block("audio_head.padded").transpose(1, 2).contiguous()

    def fill_noise(self, noise=None):
        """``noise``: K tensors [B, V] of Exp(1) draws (pins the sampler), or None for fresh draws from torch's generator."""
        if noise is None:
            self.noise_buf.exponential_(1)
        else:
            for i, q in enumerate(noise):
                self.noise_buf[i].copy_(q.reshape(self.noise_buf[i].shape), non_blocking=True)

    def prefill(self, tokens, masks):
        e, m = self.e, self.e.m
        B, S, K1 = tokens.shape
        if S > m.bb.max_seq_len:
            raise ValueError("prompt longer than max_seq_len")
        M = B * S
        tk = tokens.reshape(M, K1).to(torch.int64).contiguous()
        mk = masks.reshape(M, K1).to(torch.uint8).contiguous()
        h0 = torch.empty(M, m.bb.embed_dim, dtype=BF16, device=m.device)
        ops.embed_fwd(tk, mk, m.block("text_embeddings.weight"), m.block("audio_embeddings.weight"), h0, m.args.audio_vocab_size)
        hidden = e.backbone.forward(h0, B, S, True)
        self.bb.fill_from(e.backbone.acts, B, S)
        e.backbone.acts = []
        self.bb.pos.fill_(S - 1)
        self.cur = S - 1                       # host mirror of the device-side position (no sync per frame)
        return hidden.view(B, S, -1)[:, -1, :].contiguous()

    def prefill_ragged(self, tokens_list, masks_list):
        """Prompts of different lengths, one per batch row: each is prefilled on its own ([1, S_b] through the training
        forward) into its row of the caches; positions are per row from then on (``pos`` is a device vector)."""
        e, m = self.e, self.e.m
        last = []
        for b, (tk, mk) in enumerate(zip(tokens_list, masks_list)):
            S = tk.shape[0]
            if S > m.bb.max_seq_len:
                raise ValueError("prompt longer than max_seq_len")
            tk = tk.to(device=m.device, dtype=torch.int64).contiguous()
            mk = mk.to(device=m.device, dtype=torch.uint8).contiguous()
            h0 = torch.empty(S, m.bb.embed_dim, dtype=BF16, device=m.device)
            ops.embed_fwd(tk, mk, m.block("text_embeddings.weight"), m.block("audio_embeddings.weight"), h0, m.args.audio_vocab_size)
            hidden = e.backbone.forward(h0, 1, S, True)
            self.bb.fill_row(e.backbone.acts, b, S)
            e.backbone.acts = []
            self.bb.pos[b] = S - 1
            last.append(hidden[-1])
            self.cur = max(self.cur, S - 1)
        return torch.stack(last).contiguous()

    def backbone_step(self, tokens, masks):
        m = self.e.m
        B = tokens.shape[0]
        tk = tokens.reshape(B, -1).to(torch.int64).contiguous()
        mk = masks.reshape(B, -1).to(torch.uint8).contiguous()
        ops.embed_fwd(tk, mk, m.block("text_embeddings.weight"), m.block("audio_embeddings.weight"), self.h0, m.args.audio_vocab_size)
        self.bb.pos.add_(1)
        return self.bb.step(self.h0)

    def graph_frame(self, tokens, masks, temperature, topk, noise=None):
        """Replay one decode frame (~1.8 k kernel launches) as a single HIP graph.  The first decode frame runs eagerly
        (warm-up: lazy function attributes, allocator), the second is captured, later ones are replays; positions, input
        tokens and the frame's noise live in persistent device buffers, so the same graph serves every frame.  Re-captured
        when temperature / top-k change."""
        m = self.e.m
        if self.cur + 1 >= m.bb.max_seq_len:
            raise ValueError("sequence exceeds max_seq_len")
        self.cur += 1
        key = (float(temperature), int(topk))
        if self.graph is None or self.graph_key != key:
            if self.warm < 1 or self.graph_key not in (None, key):
                self.warm, self.graph, self.graph_key = 1, None, None
                return self.e._decode_frame(self, tokens, masks, temperature, topk, noise)
            self.in_tok = tokens.to(torch.int64).clone()
            self.in_msk = masks.to(torch.uint8).clone()
            self.fill_noise(noise)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # No cyclic garbage collection while the stream is capturing: a collection that happens to run inside the ~700
            # launches may finalise objects whose destructors call the runtime (an older model's captured graph, events) - not
            # allowed during capture, the process aborts (seen once a test run's allocation pattern moved a collection in
            # there).  torch.cuda.graph() collects before it starts capturing; reference-counted frees are unaffected.
            import gc
            gc_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(g):
                    self.out_static = self.e._decode_frame_body(self, self.in_tok, self.in_msk, temperature, topk)
            finally:
                if gc_on:
                    gc.enable()
            self.graph, self.graph_key = g, key
            g.replay()
            return self.out_static.clone()
        self.in_tok.copy_(tokens)
        self.in_msk.copy_(masks)
        self.fill_noise(noise)
        self.graph.replay()
        return self.out_static.clone()

    def decoder_reset(self):
        pass   # positions restart at 0 every frame; stale cache rows beyond the current position are never read

    def decoder_step(self, x_in, i, code=None, final_norm=True):
        """Decoder position i.  Input = ``x_in`` [B, d] (the backbone state, i = 0) or, with ``code`` (int32 [B] on the
        device), the audio embedding of code i-1 gathered inside the projection product (model.py:189-191)."""
        m = self.e.m
        if code is None:
            ops.gemv(x_in.contiguous(), m.block("projection.weight"), self.proj)
        else:
            ops.gemv_ex(m.block("audio_embeddings.weight"), m.block("projection.weight"), self.proj, row_index=code,
                        row_offset=(i - 1) * m.args.audio_vocab_size)
        self.dc.pos = self.dpos[i]
        self.dc.pos_host = i if DECODE_POS_HOST else None
        return self.dc.step(self.proj, final_norm=final_norm)
