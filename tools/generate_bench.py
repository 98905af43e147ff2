#!/usr/bin/env python3
"""BASELINE config 5: generate() - Mimi encode of the context + AR multi-codebook decode of 10 s of audio (125 frames) +
Mimi decode, on one MI355X, CSM-1B random init.  Neither tokenizer can be fetched offline: the text side is a byte-level
stand-in, the audio side is the real GPU codec (csm.codec.MimiCodec) with seeded random weights in the Hugging Face
layout (GEN_CODEC=rvq swaps in the quantiser-only stand-in when transformers is unavailable)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.generator import Generator, Segment
from csm.models.model import Model
from csm.training.trainer import csm_1b_args
from csm.hip import ops


class ByteTokenizer:
    def encode(self, text):
        return [128000] + [b + 1000 for b in text.encode()] + [128001]


class RvqOnlyCodec:
    """Mimi's quantiser stage only: 1920-sample frames are folded into 256-d latents by a fixed random projection."""
    sample_rate = 24000

    def __init__(self, device, K=32):
        g = torch.Generator(device=device).manual_seed(0)
        self.cb = torch.randn(K, 2048, 256, device=device, generator=g)
        self.proj = torch.randn(1920, 256, device=device, generator=g) / 44.0
        self.K = K

    def encode(self, wav):                      # [1,1,N] -> [1,K,T]
        T = wav.shape[-1] // 1920
        lat = (wav[0, 0, :T * 1920].view(T, 1920) @ self.proj).contiguous()
        codes = torch.empty(self.K, T, dtype=torch.int64, device=wav.device)
        ops.rvq_encode(lat, self.cb, codes, 1)
        return codes.unsqueeze(0)

    def decode(self, codes):                    # [1,K,T] -> [1,1,N]
        c = codes[0].clamp(0, 2047).contiguous()
        out = torch.empty(c.shape[1], 256, dtype=torch.float32, device=codes.device)
        ops.rvq_decode(c, self.cb, out)
        return (out @ self.proj.t()).reshape(1, 1, -1)


def make_codec(dev):
    if os.environ.get("GEN_CODEC", "mimi") == "mimi":
        from transformers import MimiConfig, MimiModel
        from csm.codec import MimiCodec
        torch.manual_seed(0)
        hf = MimiModel(MimiConfig()).eval()
        with torch.no_grad():
            for name, buf in hf.named_buffers():
                if name.endswith("embed_sum"):
                    buf.copy_(torch.randn(buf.shape))
        return MimiCodec(hf.state_dict(), device=dev)
    return RvqOnlyCodec(dev)


def run(model, frames=125, repeats=2):
    """BASELINE config 5 on an existing model (bench.py's extra leg): 5 s of context audio tokenised by Mimi, ``frames`` AR
    frames, Mimi decode.  Returns the best of ``repeats`` timed runs after one short warm-up."""
    dev = model.device
    if not model.caches_are_enabled():
        model.setup_caches(1)
    gen = Generator(model, text_tokenizer=ByteTokenizer(), audio_tokenizer=make_codec(dev))
    ctx = [Segment(0, "hello there", torch.randn(5 * 24000, device=dev) * 0.1)]
    text = "the quick brown fox jumps over the lazy dog"
    gen.generate(text, 1, ctx, max_audio_length_ms=80 * 5)                  # warm-up: lazy attributes, allocator, graph capture
    best = None
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.time()
        audio = gen.generate(text, 1, ctx, max_audio_length_ms=80 * frames)
        torch.cuda.synchronize()
        dt = time.time() - t0
        nfr = audio.numel() / 1920
        if best is None or dt / max(nfr, 1) < best[0] / max(best[1], 1):
            best = (dt, nfr)
    dt, nfr = best
    return {"frames": nfr, "seconds": round(dt, 4), "frames_per_s": round(nfr / dt, 1), "ms_per_frame": round(dt / max(nfr, 1) * 1e3, 3),
            "x_real_time": round(nfr * 0.08 / dt, 2), "includes": "Mimi encode of the context + prefill + decode frames + Mimi decode"}


def run_batch(model, nb=4, frames=125):
    """generate_batch(): ``nb`` utterances decoded together (ragged prompts, shared weight loads); aggregate frames/s of the second
    of two runs (the first captures the batch's frame graph)."""
    dev = model.device
    gen = Generator(model, text_tokenizer=ByteTokenizer(), audio_tokenizer=make_codec(dev))
    ctx = [Segment(0, "hello there", torch.randn(5 * 24000, device=dev) * 0.1)]
    texts = [f"utterance number {i}: the quick brown fox jumps over the lazy dog" for i in range(nb)]
    ctxs = [ctx if i % 2 == 0 else [] for i in range(nb)]
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        outs = gen.generate_batch(texts, list(range(nb)), ctxs, max_audio_length_ms=80 * frames)
        torch.cuda.synchronize()
        dt = time.time() - t0
    tot = sum(o.numel() for o in outs) / 1920
    return {"utterances": nb, "frames": tot, "seconds": round(dt, 4), "frames_per_s_aggregate": round(tot / dt, 1),
            "x_real_time_aggregate": round(tot * 0.08 / dt, 2)}


def main():
    dev = "cuda:0"
    model = Model(csm_1b_args(), device=dev, seed=0)
    codec = make_codec(dev)
    gen = Generator(model, text_tokenizer=ByteTokenizer(), audio_tokenizer=codec)
    ctx = [Segment(0, "hello there", torch.randn(5 * 24000, device=dev) * 0.1)]        # 5 s of context audio to tokenise
    frames = int(os.environ.get("GEN_FRAMES", 125))
    for n in (5, frames):
        torch.cuda.synchronize()
        t0 = time.time()
        audio = gen.generate("the quick brown fox jumps over the lazy dog", 1, ctx, max_audio_length_ms=80 * n)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"{n} frames requested: {audio.numel() / 24000:.2f} s of audio in {dt:.2f} s -> {audio.numel() / 1920 / dt:.1f} frames/s "
              f"({audio.numel() / 24000 / dt:.2f}x real time)")
    nb = int(os.environ.get("GEN_BATCH", 4))
    if nb > 1:
        texts = [f"utterance number {i}: the quick brown fox jumps over the lazy dog" for i in range(nb)]
        ctxs = [ctx if i % 2 == 0 else [] for i in range(nb)]
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            outs = gen.generate_batch(texts, list(range(nb)), ctxs, max_audio_length_ms=80 * frames)
            torch.cuda.synchronize()
            dt = time.time() - t0
        tot = sum(o.numel() for o in outs)
        print(f"batch of {nb}: {tot / 24000:.2f} s of audio in {dt:.2f} s -> {tot / 1920 / dt:.1f} frames/s aggregate ({tot / 24000 / dt:.2f}x real time)")


if __name__ == "__main__":
    main()
