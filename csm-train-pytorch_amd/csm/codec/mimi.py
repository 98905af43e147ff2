"""Mimi audio codec on MI355X - the audio tokenizer ``Generator`` uses (reference src/csm/generator.py:67-70,117,209).

The reference obtains Mimi from ``moshi==0.2.2`` (``loaders.get_mimi`` + ``set_num_codebooks(32)``), a third-party package
that is neither vendored nor installed here, and whose weights come from the hub.  This module re-implements the
*architecture* (SEANet causal conv encoder / decoder with ratios 8-6-5-4, two 8-layer causal transformers with layer
scale and a 250-frame window, stride-2 down / up-sampling to 12.5 Hz, split residual VQ with 1 semantic + 31 acoustic
codebooks of 2048 x 256) on hand-written fp32 HIP kernels, taking the weights as a state dict with the key names of the
Hugging Face port (``transformers.MimiModel.state_dict()``; the moshi checkpoint maps onto it 1:1).  It exposes exactly
what ``Generator`` touches: ``encode([1,1,N]) -> [1,K,T]`` int64, ``decode([1,K,T]) -> [1,1,N]``, ``sample_rate``.
"""
import math
from typing import Dict, Optional

import torch

from ..hip import check, lib, ops

F32 = torch.float32


def _s():
    return torch.cuda.current_stream().cuda_stream


class MimiCodec:
    sample_rate = 24000
    frame_rate = 12.5

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", num_codebooks: int = 32, ratios=(8, 6, 5, 4),
                 num_filters: int = 64, hidden: int = 512, heads: int = 8, window: int = 250, n_layers: int = 8,
                 codebook_dim: int = 256, n_semantic: int = 1, rope_theta: float = 10000.0, norm_eps: float = 1e-5):
        self.dev = torch.device(device)
        self.K, self.ratios, self.nf, self.hidden, self.heads = num_codebooks, tuple(ratios), num_filters, hidden, heads
        self.window, self.n_layers, self.cdim, self.n_sem = window, n_layers, codebook_dim, n_semantic
        self.theta, self.eps = rope_theta, norm_eps
        self.w = {k: v.detach().to(self.dev, F32).contiguous() for k, v in state_dict.items() if v.dtype.is_floating_point}
        w = self.w
        for tr in ("encoder_transformer", "decoder_transformer"):          # fuse q|k|v rows once
            for i in range(n_layers):
                p = f"{tr}.layers.{i}.self_attn"
                w[f"{p}.qkv"] = torch.cat([w[f"{p}.q_proj.weight"], w[f"{p}.k_proj.weight"], w[f"{p}.v_proj.weight"]], 0).contiguous()
        self.cb = {}
        for name, n in (("semantic", n_semantic), ("acoustic", num_codebooks - n_semantic)):
            q = f"quantizer.{name}_residual_vector_quantizer"
            books = [w[f"{q}.layers.{i}.codebook.embed_sum"] / w[f"{q}.layers.{i}.codebook.cluster_usage"].clamp(min=1e-5)[:, None]
                     for i in range(n)]
            self.cb[name] = torch.stack(books).contiguous()
            w[f"{q}.in"] = w[f"{q}.input_proj.weight"].squeeze(-1).contiguous()      # [256, 512]
            w[f"{q}.out"] = w[f"{q}.output_proj.weight"].squeeze(-1).contiguous()    # [512, 256]

    def set_num_codebooks(self, n: int):
        self.K = n

    # ------------------------------------------------------------------ kernels
    def _conv(self, x, name, k, stride=1, dil=1, elu=False, res=None, pad_mode=0):
        wt, b = self.w[f"{name}.conv.weight"], self.w.get(f"{name}.conv.bias")
        C_out, cin_g, kk = wt.shape
        assert kk == k
        C_in, T_in = x.shape
        groups = C_in // cin_g
        k_eff = (k - 1) * dil + 1
        pad_total = k_eff - stride
        n_frames = math.ceil((T_in - k_eff + pad_total) / stride + 1) - 1
        extra = n_frames * stride + k_eff - pad_total - T_in
        T_out = (T_in + pad_total + extra - k_eff) // stride + 1
        y = torch.empty(C_out, T_out, dtype=F32, device=self.dev)
        check(lib.csm_conv1d_f32(x.data_ptr(), wt.data_ptr(), b.data_ptr() if b is not None else None,
                                 res.data_ptr() if res is not None else None, y.data_ptr(), C_in, C_out, T_in, T_out, k, stride,
                                 dil, pad_total, pad_mode, groups, int(elu), _s()), "csm_conv1d_f32")
        return y

    def _convt(self, x, name, k, stride, elu=False):
        wt, b = self.w[f"{name}.conv.weight"], self.w.get(f"{name}.conv.bias")
        C_in, cout_g, kk = wt.shape
        assert kk == k and x.shape[0] == C_in
        groups = 1 if cout_g != 1 or C_in == 1 else C_in
        C_out = cout_g * groups
        T_in = x.shape[1]
        T_out = T_in * stride                              # causal: the k - stride overhang is trimmed on the right
        y = torch.empty(C_out, T_out, dtype=F32, device=self.dev)
        check(lib.csm_conv_transpose1d_f32(x.data_ptr(), wt.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(),
                                           C_in, C_out, T_in, T_out, k, stride, 0, groups, int(elu), _s()), "csm_conv_transpose1d_f32")
        return y

    def _resblock(self, x, name):
        h = self._conv(x, f"{name}.block.1", 3, elu=True)
        return self._conv(h, f"{name}.block.3", 1, elu=True, res=x)

    def _linear(self, x, W, scale=None, res=None, act=0):
        T, K = x.shape
        N = W.shape[0]
        y = torch.empty(T, N, dtype=F32, device=self.dev)
        check(lib.csm_linear_f32(x.data_ptr(), W.data_ptr(), scale.data_ptr() if scale is not None else None,
                                 res.data_ptr() if res is not None else None, y.data_ptr(), T, N, K, x.stride(0), act, _s()),
              "csm_linear_f32")
        return y

    def _transpose(self, x):
        R, Cn = x.shape
        y = torch.empty(Cn, R, dtype=F32, device=self.dev)
        check(lib.csm_transpose_f32(x.data_ptr(), y.data_ptr(), R, Cn, _s()), "csm_transpose_f32")
        return y

    def _transformer(self, x, tr):
        """x [T, hidden] -> [T, hidden]: pre-LN, rotate-half RoPE, causal window attention, GELU MLP, layer scale."""
        T, D = x.shape
        H, hd, w = self.heads, D // self.heads, self.w
        for i in range(self.n_layers):
            p = f"{tr}.layers.{i}"
            xn = torch.empty_like(x)
            check(lib.csm_layernorm_f32(x.data_ptr(), w[f"{p}.input_layernorm.weight"].data_ptr(), w[f"{p}.input_layernorm.bias"].data_ptr(),
                                        xn.data_ptr(), T, D, self.eps, _s()), "csm_layernorm_f32")
            qkv = self._linear(xn, w[f"{p}.self_attn.qkv"])
            check(lib.csm_rope_half_f32(qkv.data_ptr(), T, H, hd, self.theta, 0, _s()), "csm_rope_half_f32")
            o = torch.empty(T, D, dtype=F32, device=self.dev)
            check(lib.csm_attn_window_f32(qkv.data_ptr(), o.data_ptr(), T, H, hd, self.window, _s()), "csm_attn_window_f32")
            x = self._linear(o, w[f"{p}.self_attn.o_proj.weight"], scale=w[f"{p}.self_attn_layer_scale.scale"], res=x)
            check(lib.csm_layernorm_f32(x.data_ptr(), w[f"{p}.post_attention_layernorm.weight"].data_ptr(),
                                        w[f"{p}.post_attention_layernorm.bias"].data_ptr(), xn.data_ptr(), T, D, self.eps, _s()),
                  "csm_layernorm_f32")
            h1 = self._linear(xn, w[f"{p}.mlp.fc1.weight"], act=1)
            x = self._linear(h1, w[f"{p}.mlp.fc2.weight"], scale=w[f"{p}.mlp_layer_scale.scale"], res=x)
        return x

    # ------------------------------------------------------------------ public protocol (Mimi's)
    @torch.no_grad()
    def encode_latent(self, wav: torch.Tensor) -> torch.Tensor:
        """[1,1,N] waveform -> pre-quantiser latent [T, hidden] at 12.5 Hz."""
        x = wav.reshape(1, -1).to(self.dev, F32).contiguous()
        x = self._conv(x, "encoder.layers.0", 7)
        idx = 1
        for r in reversed(self.ratios):
            x = self._resblock(x, f"encoder.layers.{idx}")
            x = self._conv(x, f"encoder.layers.{idx + 2}", 2 * r, stride=r, elu=True)
            idx += 3
        x = self._conv(x, f"encoder.layers.{idx + 1}", 3, elu=True)                  # [hidden, T25]
        x = self._transformer(self._transpose(x), "encoder_transformer")
        x = self._conv(self._transpose(x), "downsample", 4, stride=2, pad_mode=1)     # [hidden, T]
        return self._transpose(x)

    @torch.no_grad()
    def encode(self, wav: torch.Tensor) -> torch.Tensor:
        lat = self.encode_latent(wav)
        T = lat.shape[0]
        codes = torch.empty(self.K, T, dtype=torch.int64, device=self.dev)
        q = "quantizer.semantic_residual_vector_quantizer"
        ops.rvq_encode(self._linear(lat, self.w[f"{q}.in"]), self.cb["semantic"], codes[:self.n_sem], self.n_sem)
        if self.K > self.n_sem:
            q = "quantizer.acoustic_residual_vector_quantizer"
            na = self.K - self.n_sem
            ops.rvq_encode(self._linear(lat, self.w[f"{q}.in"]), self.cb["acoustic"][:na].contiguous(), codes[self.n_sem:], 0)
        return codes.unsqueeze(0)

    @torch.no_grad()
    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        c = codes[0].to(self.dev, torch.int64).contiguous()
        K, T = c.shape
        lat = torch.zeros(T, self.hidden, dtype=F32, device=self.dev)
        for name, lo, hi in (("semantic", 0, min(K, self.n_sem)), ("acoustic", self.n_sem, K)):
            if hi <= lo:
                continue
            q = f"quantizer.{name}_residual_vector_quantizer"
            zq = torch.empty(T, self.cdim, dtype=F32, device=self.dev)
            ops.rvq_decode(c[lo:hi].contiguous(), self.cb[name][:hi - lo].contiguous(), zq)
            lat = self._linear(zq, self.w[f"{q}.out"], res=lat)
        x = self._convt(self._transpose(lat), "upsample", 4, 2)                       # [hidden, 2T]
        x = self._transformer(self._transpose(x), "decoder_transformer")
        x = self._conv(self._transpose(x), "decoder.layers.0", 7)
        idx = 1
        for r in self.ratios:
            x = self._convt(x, f"decoder.layers.{idx + 1}", 2 * r, r, elu=True)
            x = self._resblock(x, f"decoder.layers.{idx + 2}")
            idx += 3
        x = self._conv(x, f"decoder.layers.{idx + 1}", 3, elu=True)                  # [1, N]
        return x.reshape(1, 1, -1)
