"""Data front end of CSM training: audio loading, segmentation, contextual examples, frame tokenisation, collation.

Mirrors reference ``src/csm/data/training_data.py`` (classes, signatures and results) without torchaudio, which this
image does not have: WAV files are read with ``scipy.io.wavfile`` and resampled by the windowed-sinc interpolation
``torchaudio.functional.resample`` documents (Hann window, ``lowpass_filter_width=6``, ``rolloff=0.99``), restated
below.  Audio tokenisation goes through any object with Mimi's ``encode([1,1,N]) -> [1,K,T]`` protocol - on the GPU box
that is ``csm.codec.MimiCodec`` (HIP kernels); nothing here touches the GPU by itself, so the module imports on CPU.

Frame layout (reference ``generator.py:77-145`` = ``training_data.py:304-335``): a text token occupies column 32 of a
33-wide frame, an audio frame columns 0..31, and every audio segment ends in one all-zero EOS frame.
"""
import json
import math
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Sequence, Union

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, Sampler

IGNORE_INDEX = -100


# ------------------------------------------------------------------------------------------------ audio I/O
def resample(wav: torch.Tensor, orig_sr: int, new_sr: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    """Band-limited resampling of ``[..., N]`` by windowed-sinc interpolation (the algorithm of
    ``torchaudio.functional.resample`` with its default ``sinc_interp_hann`` kernel, which the reference calls at
    training_data.py:60)."""
    if orig_sr == new_sr:
        return wav
    g = math.gcd(int(orig_sr), int(new_sr))
    o, n = int(orig_sr) // g, int(new_sr) // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = torch.arange(-width, width + o, dtype=torch.float64) / o
    t = (torch.arange(0, -n, -1, dtype=torch.float64)[:, None] / n + idx[None, :]) * base
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernel = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / o)      # [n, 2*width + o]
    shape = wav.shape
    x = wav.reshape(-1, 1, shape[-1]).to(torch.float64)
    x = torch.nn.functional.pad(x, (width, width + o))
    y = torch.nn.functional.conv1d(x, kernel[:, None, :], stride=o)                                 # [B, n, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    length = math.ceil(n * shape[-1] / o)
    return y[:, :length].reshape(shape[:-1] + (length,)).to(wav.dtype)


def load_audio(path: Union[str, Path]):
    """``(waveform [channels, N] float32 in [-1, 1], sample_rate)`` of a WAV file (what ``torchaudio.load`` returns)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(str(path))
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    elif np.issubdtype(data.dtype, np.integer):
        x = data.astype(np.float32) / float(2 ** (8 * data.dtype.itemsize - 1))
    else:
        x = data.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.T)), int(sr)


@dataclass
class TrainingExample:
    """One utterance: transcript, mono waveform at the processor's sample rate, speaker id (training_data.py:16-23)."""
    text: str
    audio: torch.Tensor
    speaker_id: int
    metadata: Optional[Dict] = None


# ------------------------------------------------------------------------------------------------ segmentation
class CSMDataProcessor:
    """Cuts a recording + transcript into training utterances (reference training_data.py:26-176)."""

    def __init__(self, sample_rate: int = 24000, segment_duration_ms: int = 10000, overlap_ms: int = 2000):
        self.sample_rate = sample_rate
        self.segment_duration_samples = int(segment_duration_ms * sample_rate / 1000)
        self.overlap_samples = int(overlap_ms * sample_rate / 1000)

    def prepare_from_audio_file(self, audio_path, transcript_path, speaker_id: int, alignment_path=None) -> List[TrainingExample]:
        audio, sr = load_audio(audio_path)
        if sr != self.sample_rate:
            audio = resample(audio, sr, self.sample_rate)
        if audio.size(0) > 1:
            audio = audio.mean(dim=0, keepdim=True)
        audio = audio.squeeze(0)
        transcript = Path(transcript_path).read_text().strip()
        if alignment_path:
            return self._segment_with_alignments(audio, transcript, speaker_id, self._load_alignments(alignment_path))
        return self._segment_basic(audio, transcript, speaker_id)

    def _keep(self, text: str, n_samples: int) -> bool:
        return len(text) >= 10 and n_samples >= self.sample_rate       # at least 10 characters and one second

    def _segment_basic(self, audio: torch.Tensor, transcript: str, speaker_id: int) -> List[TrainingExample]:
        """Overlapping fixed-length windows; the text of a window is the proportional slice of the transcript
        (characters per sample), reference training_data.py:81-114."""
        n = audio.size(0)
        cps = len(transcript) / n
        out = []
        for s in range(0, n, self.segment_duration_samples - self.overlap_samples):
            e = min(s + self.segment_duration_samples, n)
            text = transcript[int(s * cps):int(e * cps)]
            if self._keep(text.strip(), e - s):
                out.append(TrainingExample(text, audio[s:e], speaker_id, {"start_sample": s, "end_sample": e}))
        return out

    @staticmethod
    def _load_alignments(path) -> Dict:
        with open(path, "r") as f:
            return json.load(f)

    def _segment_with_alignments(self, audio, transcript, speaker_id, alignments) -> List[TrainingExample]:
        """Greedy grouping of aligned words into windows no longer than the segment duration, measured from the
        window's first word (the initial window is anchored at sample 0); reference training_data.py:121-176."""
        words = alignments.get("words", [])
        if not words:
            return self._segment_basic(audio, transcript, speaker_id)
        spans, start, end, text = [], 0, 0, ""
        for w in words:
            ws, we = int(w["start"] * self.sample_rate), int(w["end"] * self.sample_rate)
            if we - start > self.segment_duration_samples:
                if text:
                    spans.append((start, end, text))
                start, end, text = ws, we, w["word"] + " "
            else:
                end, text = we, text + w["word"] + " "
        if text:
            spans.append((start, end, text))
        out = []
        for s, e, text in spans:
            text = text.strip()
            if self._keep(text, e - s):
                out.append(TrainingExample(text, audio[s:e], speaker_id, {"start_sample": s, "end_sample": e}))
        return out


class ContextualExampleGenerator:
    """Turn i of a conversation becomes a target with up to ``max_context_turns`` preceding turns as context
    (reference training_data.py:179-224)."""

    def __init__(self, max_context_turns: int = 3, include_audio_context: bool = True):
        self.max_context_turns = max_context_turns
        self.include_audio_context = include_audio_context

    def create_contextual_examples(self, conversation: Sequence[TrainingExample]) -> List[Dict]:
        return [{"context": list(conversation[max(0, i - self.max_context_turns):i]), "target": conversation[i]}
                for i in range(len(conversation))]


# ------------------------------------------------------------------------------------------------ dataset
class CSMDataset(Dataset):
    """Tokenised contextual examples (reference training_data.py:227-358): input = context segments (text + audio
    frames) followed by the target's text frames; ``target_audio_tokens`` = the target's Mimi codes ``[T, K]``.

    ``truncate="reference"`` reproduces the reference's over-length rule literally (only the target's text frames
    survive, lines 289-295); ``truncate="keep_context"`` keeps the last ``max_seq_len`` frames instead."""

    def __init__(self, examples: List[Dict], text_tokenizer, audio_tokenizer, max_seq_len: int = 2048, truncate: str = "reference"):
        if truncate not in ("reference", "keep_context"):
            raise ValueError("truncate must be 'reference' or 'keep_context'")
        for name, tok in (("text_tokenizer", text_tokenizer), ("audio_tokenizer", audio_tokenizer)):
            if not hasattr(tok, "encode"):
                raise TypeError(f"{name} needs an encode() method")
        self.examples, self.text_tokenizer, self.audio_tokenizer = examples, text_tokenizer, audio_tokenizer
        self.max_seq_len, self.truncate = max_seq_len, truncate

    def __len__(self):
        return len(self.examples)

    def _codes(self, audio: torch.Tensor) -> torch.Tensor:
        """Mimi codes ``[K, T]`` int64 (on the CPU) of a mono waveform."""
        codes = self.audio_tokenizer.encode(audio.reshape(1, 1, -1))
        if isinstance(codes, (list, tuple)):
            codes = codes[0]
        if codes.dim() == 3:
            codes = codes[0]
        return codes.detach().to("cpu", torch.long)

    def _tokenize_text_segment(self, text: str, speaker: int):
        ids = self.text_tokenizer.encode(f"[{speaker}]{text}")
        frame = torch.zeros(len(ids), 33, dtype=torch.long)
        mask = torch.zeros(len(ids), 33, dtype=torch.bool)
        frame[:, -1] = torch.as_tensor(ids, dtype=torch.long)
        mask[:, -1] = True
        return frame, mask

    def _tokenize_audio(self, audio: torch.Tensor):
        codes = self._codes(audio)                                                      # [K, T]
        codes = torch.cat([codes, torch.zeros(codes.size(0), 1, dtype=torch.long)], 1)   # + EOS frame
        frame = torch.zeros(codes.size(1), 33, dtype=torch.long)
        mask = torch.zeros(codes.size(1), 33, dtype=torch.bool)
        frame[:, :codes.size(0)] = codes.t()
        mask[:, :-1] = True
        return frame, mask

    def _tokenize_segment(self, text: str, speaker: int, audio: torch.Tensor):
        tt, tm = self._tokenize_text_segment(text, speaker)
        at, am = self._tokenize_audio(audio)
        return torch.cat([tt, at], 0), torch.cat([tm, am], 0)

    def _tokenize_audio_for_target(self, audio: torch.Tensor) -> torch.Tensor:
        return self._codes(audio).t().contiguous()                                       # [T, K]

    def __getitem__(self, idx):
        ex = self.examples[idx]
        toks, masks = [], []
        for ctx in ex.get("context", []):
            t, m = self._tokenize_segment(ctx.text, ctx.speaker_id, ctx.audio)
            toks.append(t)
            masks.append(m)
        target = ex["target"]
        tt, tm = self._tokenize_text_segment(target.text, target.speaker_id)
        toks.append(tt)
        masks.append(tm)
        tokens, mask = torch.cat(toks, 0), torch.cat(masks, 0)
        if tokens.size(0) > self.max_seq_len:
            keep = min(self.max_seq_len, tt.size(0)) if self.truncate == "reference" else self.max_seq_len
            tokens, mask = tokens[tokens.size(0) - keep:], mask[mask.size(0) - keep:]
        return {"input_tokens": tokens, "input_masks": mask, "target_audio_tokens": self._tokenize_audio_for_target(target.audio)}

    def lengths(self) -> List[int]:
        """Input length of every item without running the audio tokenizer twice: frames = text ids + ceil(N/1920)+1
        per context turn (an estimate used only for bucketing)."""
        out = []
        for ex in self.examples:
            n = len(self.text_tokenizer.encode(f"[{ex['target'].speaker_id}]{ex['target'].text}"))
            for c in ex.get("context", []):
                n += len(self.text_tokenizer.encode(f"[{c.speaker_id}]{c.text}")) + math.ceil(c.audio.numel() / 1920) + 1
            out.append(min(n, self.max_seq_len))
        return out


# ------------------------------------------------------------------------------------------------ collation
def collate_variable_length(batch: List[Dict[str, torch.Tensor]], target_pad: int = 0) -> Dict[str, torch.Tensor]:
    """Zero-pad ``input_tokens`` [S,33] / ``input_masks`` [S,33] / ``target_audio_tokens`` [T,K] to the batch max
    (reference training_data.py:379-408).  ``target_pad=IGNORE_INDEX`` pads the targets with -100 instead, which the
    HIP cross-entropy skips (rows with a negative target contribute neither loss nor gradient and are left out of the
    mean) - the reference's zero padding trains padded rows towards the EOS class."""
    max_s = max(b["input_tokens"].shape[0] for b in batch)
    max_t = max(b["target_audio_tokens"].shape[0] for b in batch)
    n = len(batch)
    k1 = batch[0]["input_tokens"].shape[1]
    k = batch[0]["target_audio_tokens"].shape[1]
    tokens = torch.zeros(n, max_s, k1, dtype=torch.long)
    masks = torch.zeros(n, max_s, k1, dtype=torch.bool)
    targets = torch.full((n, max_t, k), target_pad, dtype=torch.long)
    for i, b in enumerate(batch):
        s, t = b["input_tokens"].shape[0], b["target_audio_tokens"].shape[0]
        tokens[i, :s] = b["input_tokens"]
        masks[i, :s] = b["input_masks"]
        targets[i, :t] = b["target_audio_tokens"]
    return {"input_tokens": tokens, "input_masks": masks, "target_audio_tokens": targets}


class LengthBucketSampler(Sampler):
    """Batches of similar length: indices are sorted by length inside windows of ``bucket_batches`` batches, cut into
    batches, and the batches shuffled - so zero padding (and the work spent on it) stays small.  Deterministic per
    (seed, epoch); with ``world_size > 1`` every rank takes every ``world_size``-th batch."""

    def __init__(self, lengths: Sequence[int], batch_size: int, bucket_batches: int = 50, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False, rank: int = 0, world_size: int = 1):
        self.lengths, self.batch_size, self.bucket_batches = list(lengths), batch_size, bucket_batches
        self.shuffle, self.seed, self.drop_last, self.rank, self.world_size, self.epoch = shuffle, seed, drop_last, rank, world_size, 0

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def _batches(self) -> List[List[int]]:
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        n = len(self.lengths)
        order = torch.randperm(n, generator=g).tolist() if self.shuffle else list(range(n))
        window = self.batch_size * self.bucket_batches
        batches = []
        for w in range(0, n, window):
            chunk = sorted(order[w:w + window], key=lambda i: self.lengths[i])
            batches += [chunk[i:i + self.batch_size] for i in range(0, len(chunk), self.batch_size)]
        if self.drop_last:
            batches = [b for b in batches if len(b) == self.batch_size]
        if self.shuffle:
            batches = [batches[i] for i in torch.randperm(len(batches), generator=g).tolist()]
        usable = len(batches) - len(batches) % self.world_size
        return batches[self.rank:usable:self.world_size] if self.world_size > 1 else batches

    def __iter__(self) -> Iterator[List[int]]:
        return iter(self._batches())

    def __len__(self):
        return len(self._batches())


def create_dataloader(dataset: Dataset, batch_size: int, shuffle: bool = True, num_workers: int = 2, pin_memory: bool = True,
                      bucket_by_length: bool = False, target_pad: int = 0) -> DataLoader:
    """Reference ``create_dataloader`` (training_data.py:361-376).  ``bucket_by_length`` swaps the plain shuffle for
    :class:`LengthBucketSampler` (needs ``dataset.lengths()``)."""
    from functools import partial
    collate = collate_variable_length if target_pad == 0 else partial(collate_variable_length, target_pad=target_pad)
    if bucket_by_length:
        sampler = LengthBucketSampler(dataset.lengths(), batch_size, shuffle=shuffle)
        return DataLoader(dataset, batch_sampler=sampler, num_workers=num_workers, pin_memory=pin_memory, collate_fn=collate)
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, pin_memory=pin_memory,
                      collate_fn=collate)
