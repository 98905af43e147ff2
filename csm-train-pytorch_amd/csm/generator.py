"""Generation API of reference ``src/csm/generator.py`` (``Segment``, ``Generator``, ``load_csm_1b``) on MI355X.

The autoregressive core (``Model.generate_frame`` driven by ``Generator.generate``, generator.py:147-218) and the frame
tokenisation (generator.py:77-145) are implemented here.  Two inputs of the reference cannot exist in this build
environment (no network): the Llama-3 text tokenizer (``AutoTokenizer.from_pretrained``, generator.py:35-36) and the
Mimi codec weights (``hf_hub_download``, generator.py:67).  Both are therefore *injected*: ``Generator(model,
text_tokenizer=..., audio_tokenizer=...)``; when omitted the reference's own loading calls are attempted and their
failure is reported as-is.  The audio tokenizer protocol is Mimi's: ``encode([1,1,N]) -> [1,K,T]`` int64,
``decode([1,K,T]) -> [1,1,N]``, ``sample_rate``.  Watermarking (silentcipher) is post-processing outside the hot path
and is not applied here (out of scope).
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from .models.model import Model, ModelArgs


@dataclass
class Segment:
    """A segment of speech (reference generator.py:18-25)."""

    speaker: int
    text: str
    audio: torch.Tensor  # (num_samples,), sample_rate = 24_000


def load_llama3_tokenizer(path: str = "meta-llama/Llama-3.2-1B"):
    """Reference generator.py:28-45 (needs the Hugging Face hub, a local cache of meta-llama/Llama-3.2-1B, or ``path`` =
    a local directory holding that tokenizer's files)."""
    import os
    from tokenizers.processors import TemplateProcessing
    from transformers import AutoTokenizer

    tokenizer = AutoTokenizer.from_pretrained(path, local_files_only=os.path.isdir(path))
    bos, eos = tokenizer.bos_token, tokenizer.eos_token
    tokenizer._tokenizer.post_processor = TemplateProcessing(
        single=f"{bos}:0 $A:0 {eos}:0", pair=f"{bos}:0 $A:0 {eos}:0 {bos}:1 $B:1 {eos}:1",
        special_tokens=[(f"{bos}", tokenizer.bos_token_id), (f"{eos}", tokenizer.eos_token_id)])
    return tokenizer


class Generator:
    """Speech generator using the CSM model (reference generator.py:48)."""

    def __init__(self, model: Model, text_tokenizer=None, audio_tokenizer=None):
        self._model = model
        self._model.setup_caches(1)
        self._text_tokenizer = text_tokenizer if text_tokenizer is not None else load_llama3_tokenizer()
        if audio_tokenizer is None:
            raise RuntimeError("Generator needs an audio tokenizer with Mimi's encode/decode protocol: the reference fetches "
                               "Mimi weights from the hub (generator.py:67), which this environment cannot do")
        self._audio_tokenizer = audio_tokenizer
        self.sample_rate = audio_tokenizer.sample_rate
        self.device = model.device

    def _tokenize_text_segment(self, text: str, speaker: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Reference generator.py:77-100: ``f"[{speaker}]{text}"`` ids into the last column."""
        K1 = self._model.args.audio_num_codebooks + 1
        ids = self._text_tokenizer.encode(f"[{speaker}]{text}")
        frame = torch.zeros(len(ids), K1).long()
        mask = torch.zeros(len(ids), K1).bool()
        frame[:, -1] = torch.tensor(ids)
        mask[:, -1] = True
        return frame.to(self.device), mask.to(self.device)

    def _tokenize_audio(self, audio: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Reference generator.py:102-130: Mimi codes [K,T] + one all-zero EOS frame into the first K columns."""
        K1 = self._model.args.audio_num_codebooks + 1
        audio = audio.to(self.device)
        codes = self._audio_tokenizer.encode(audio.unsqueeze(0).unsqueeze(0))[0]
        eos = torch.zeros(codes.size(0), 1, dtype=codes.dtype, device=codes.device)
        codes = torch.cat([codes, eos], dim=1)
        frame = torch.zeros(codes.size(1), K1).long().to(self.device)
        mask = torch.zeros(codes.size(1), K1).bool().to(self.device)
        frame[:, :-1] = codes.transpose(0, 1)
        mask[:, :-1] = True
        return frame, mask

    def _tokenize_segment(self, segment: Segment) -> Tuple[torch.Tensor, torch.Tensor]:
        tt, tm = self._tokenize_text_segment(segment.text, segment.speaker)
        at, am = self._tokenize_audio(segment.audio)
        return torch.cat([tt, at], dim=0), torch.cat([tm, am], dim=0)

    @torch.inference_mode()
    def generate(self, text: str, speaker: int, context: List[Segment], max_audio_length_ms: float = 90_000,
                 temperature: float = 0.9, topk: int = 50, eos_check_every: int = 8) -> torch.Tensor:
        """Reference generator.py:147-218.  The reference tests every frame for EOS on the host (one device sync per
        frame, generator.py:196-199); here the all-zero test runs on the device and the host looks at it once per
        ``eos_check_every`` frames, so the frame graphs are enqueued back to back.  The audio returned is the same: frames
        sampled past the EOS frame are dropped."""
        self._model.reset_caches()
        max_audio_frames = int(max_audio_length_ms / 80)
        tokens, masks = [], []
        for seg in context:
            t, m = self._tokenize_segment(seg)
            tokens.append(t)
            masks.append(m)
        t, m = self._tokenize_text_segment(text, speaker)
        tokens.append(t)
        masks.append(m)
        prompt_tokens = torch.cat(tokens, dim=0).long().to(self.device)
        prompt_mask = torch.cat(masks, dim=0).bool().to(self.device)
        samples = []
        curr_tokens, curr_mask = prompt_tokens.unsqueeze(0), prompt_mask.unsqueeze(0)
        curr_pos = torch.arange(0, prompt_tokens.size(0)).unsqueeze(0).long().to(self.device)
        max_seq_len = self._model.bb.max_seq_len - max_audio_frames
        if curr_tokens.size(1) >= max_seq_len:
            raise ValueError(f"Inputs too long, must be below max_seq_len - max_audio_frames: {max_seq_len}")
        K = self._model.args.audio_num_codebooks
        audio_mask = torch.cat([torch.ones(1, K, dtype=torch.bool), torch.zeros(1, 1, dtype=torch.bool)], dim=1).unsqueeze(1).to(self.device)
        pad = torch.zeros(1, 1, dtype=torch.long, device=self.device)
        checked = 0                                   # frames [0, checked) are known not to be EOS
        step = max(1, int(eos_check_every))
        for i in range(max_audio_frames):
            sample = self._model.generate_frame(curr_tokens, curr_mask, curr_pos, temperature, topk)
            samples.append(sample)
            if len(samples) - checked >= step or i == max_audio_frames - 1:
                eos = (torch.cat(samples[checked:], 0) == 0).all(dim=1)                 # one host look per chunk
                hit = eos.nonzero()
                if hit.numel():
                    samples = samples[:checked + int(hit[0])]
                    break
                checked = len(samples)
            curr_tokens = torch.cat([sample.long(), pad], dim=1).unsqueeze(1)
            curr_mask = audio_mask
            curr_pos = curr_pos[:, -1:] + 1
        if not samples:
            return torch.zeros(0, device=self.device)
        codes = torch.stack(samples).permute(1, 2, 0).long()
        return self._audio_tokenizer.decode(codes).squeeze(0).squeeze(0)

    @torch.inference_mode()
    def generate_batch(self, texts: List[str], speakers: List[int], contexts: List[List[Segment]],
                       max_audio_length_ms: float = 90_000, temperature: float = 0.9, topk: int = 50,
                       eos_check_every: int = 8) -> List[torch.Tensor]:
        """``generate`` for up to 4 utterances at once (not in the reference, whose loop is single-utterance): the prompts
        (different lengths) are prefilled one by one into their rows of the KV caches, then every decode frame advances all
        rows together - the decode kernels share each weight load between the batch rows, so B utterances cost about as
        much as one.  A row stops contributing at its own EOS frame; the loop ends when every row has one."""
        B = len(texts)
        if not (1 <= B <= 4 and len(speakers) == B and len(contexts) == B):
            raise ValueError("generate_batch takes 1..4 utterances with one speaker id and one context list each")
        self._model.reset_caches()
        max_audio_frames = int(max_audio_length_ms / 80)
        K = self._model.args.audio_num_codebooks
        toks, msks = [], []
        for text, spk, ctx in zip(texts, speakers, contexts):
            t_, m_ = [], []
            for seg in ctx:
                t, m = self._tokenize_segment(seg)
                t_.append(t)
                m_.append(m)
            t, m = self._tokenize_text_segment(text, spk)
            t_.append(t)
            m_.append(m)
            toks.append(torch.cat(t_, 0).long().to(self.device))
            msks.append(torch.cat(m_, 0).bool().to(self.device))
            if toks[-1].size(0) >= self._model.bb.max_seq_len - max_audio_frames:
                raise ValueError(f"Inputs too long, must be below max_seq_len - max_audio_frames: {self._model.bb.max_seq_len - max_audio_frames}")
        frames = [self._model.engine.generate_first_frames(toks, msks, temperature, topk)]          # [B, K] each
        mask = torch.cat([torch.ones(B, K, dtype=torch.bool), torch.zeros(B, 1, dtype=torch.bool)], 1).unsqueeze(1).to(self.device)
        pad = torch.zeros(B, 1, dtype=torch.long, device=self.device)
        pos = torch.ones(B, 1, dtype=torch.long, device=self.device)       # only "not the prompt" matters: positions live on the device
        eos_at = [None] * B
        step = max(1, int(eos_check_every))
        for i in range(1, max_audio_frames + 1):
            if i % step == 0 or i == max_audio_frames:
                allz = (torch.stack(frames, 1) == 0).all(dim=2).cpu()                               # [B, frames so far]
                for b in range(B):
                    hit = allz[b].nonzero()
                    eos_at[b] = int(hit[0]) if hit.numel() else None
                if all(e is not None for e in eos_at) or i == max_audio_frames:
                    break
            tokens = torch.cat([frames[-1].long(), pad], dim=1).unsqueeze(1)
            frames.append(self._model.generate_frame(tokens, mask, pos, temperature, topk))
        out = []
        codes_all = torch.stack(frames, 2).long()                                                    # [B, K, T]
        for b in range(B):
            n = eos_at[b] if eos_at[b] is not None else min(len(frames), max_audio_frames)
            if n == 0:
                out.append(torch.zeros(0, device=self.device))
            else:
                out.append(self._audio_tokenizer.decode(codes_all[b:b + 1, :, :n]).squeeze(0).squeeze(0))
        return out

    def save_wav(self, path: str, audio: torch.Tensor):
        """16-bit PCM writer (torchaudio is not available in this image)."""
        import wave
        pcm = (audio.detach().float().cpu().clamp(-1, 1) * 32767.0).to(torch.int16).numpy().tobytes()
        with wave.open(path, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(int(self.sample_rate))
            w.writeframes(pcm)


def load_csm_1b(ckpt_path: str = "ckpt.pt", device: str = "cuda", text_tokenizer=None, audio_tokenizer=None,
                mimi_weights: str = None, tokenizer_path: str = None) -> Generator:
    """Reference generator.py:221-244.  ``mimi_weights`` / ``tokenizer_path`` name local files for the two tokenizers the
    reference pulls from the hub."""
    if audio_tokenizer is None and mimi_weights:
        from .codec import load_mimi
        audio_tokenizer = load_mimi(mimi_weights, device=device)
    if text_tokenizer is None and tokenizer_path:
        text_tokenizer = load_llama3_tokenizer(tokenizer_path)
    args = ModelArgs(backbone_flavor="llama-1B", decoder_flavor="llama-100M", text_vocab_size=128256,
                     audio_vocab_size=2051, audio_num_codebooks=32)
    model = Model(args, device=device)
    model.load_state_dict(torch.load(ckpt_path, map_location="cpu", weights_only=False))
    return Generator(model, text_tokenizer=text_tokenizer, audio_tokenizer=audio_tokenizer)
