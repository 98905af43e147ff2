"""Data front end (csm.data) on the CPU: segmentation, contextual examples, frame tokenisation, collation, bucketing.
The expectations restate reference ``src/csm/data/training_data.py`` (line numbers in the assertions' comments) and the
reference's own data tests (``tests/unit/test_training.py``: mock tokenizers, shapes of one dataset item)."""
import json
import math
import wave

import numpy as np
import pytest
import torch

from csm.data import (CSMDataProcessor, CSMDataset, ContextualExampleGenerator, LengthBucketSampler, TrainingExample,
                      collate_variable_length, create_dataloader, load_audio, resample)
from csm.data.training_data import IGNORE_INDEX


class MockTextTokenizer:                     # the reference tests' MockTextTokenizer protocol: encode(str) -> ids
    def encode(self, text):
        return [1] + [3 + (ord(c) % 50) for c in text] + [2]


class MockAudioTokenizer:                    # Mimi protocol: [1,1,N] -> [1,32,ceil(N/1920)]
    def encode(self, wav):
        t = math.ceil(wav.shape[-1] / 1920)
        base = (wav.reshape(-1)[:t].abs() * 1000).long() % 2048 if wav.shape[-1] >= t else torch.zeros(t, dtype=torch.long)
        return (base[None, None, :] + torch.arange(32)[None, :, None]) % 2048


def write_wav(path, x, sr):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1 if x.ndim == 1 else x.shape[0])
        w.setsampwidth(2)
        w.setframerate(sr)
        data = (np.clip(x, -1, 1) * 32767).astype("<i2")
        w.writeframes(data.T.tobytes() if x.ndim == 2 else data.tobytes())


def test_resample_and_load(tmp_path):
    sr, new = 16000, 24000
    t = np.arange(sr) / sr
    x = 0.5 * np.sin(2 * np.pi * 440 * t)
    write_wav(tmp_path / "a.wav", np.stack([x, x]), sr)
    wav, got_sr = load_audio(tmp_path / "a.wav")
    assert got_sr == sr and wav.shape == (2, sr) and wav.dtype == torch.float32
    assert abs(float(wav.abs().max()) - 0.5) < 1e-3
    y = resample(wav, sr, new)
    assert y.shape == (2, new)                                          # ceil(new * N / orig)
    ref = 0.5 * np.sin(2 * np.pi * 440 * np.arange(new) / new)
    assert np.abs(y[0, 200:-200].numpy() - ref[200:-200]).max() < 2e-3   # band-limited tone is reproduced away from the edges
    from scipy.signal import resample_poly
    z = resample_poly(wav[0].numpy().astype(np.float64), 3, 2)
    assert np.abs(y[0, 200:-200].numpy() - z[200:-200]).max() < 5e-3
    assert resample(wav, sr, sr) is wav
    d = resample(wav, new, sr)                                          # down-sampling path: length and finite values
    assert d.shape == (2, math.ceil(sr * sr / new)) and torch.isfinite(d).all()


def test_segment_basic_and_alignments(tmp_path):
    sr = 24000
    proc = CSMDataProcessor(sample_rate=sr, segment_duration_ms=10000, overlap_ms=2000)
    audio = torch.randn(25 * sr) * 0.1
    transcript = "".join(chr(97 + i % 26) for i in range(500))
    ex = proc._segment_basic(audio, transcript, speaker_id=3)
    # stride = 10 s - 2 s = 8 s -> windows at 0, 8, 16 s (24 s start gives a 1 s tail: kept only if >= 1 s and >= 10 chars)
    starts = [e.metadata["start_sample"] for e in ex]
    assert starts[:3] == [0, 8 * sr, 16 * sr]
    assert all(e.speaker_id == 3 and e.audio.numel() == e.metadata["end_sample"] - e.metadata["start_sample"] for e in ex)
    cps = len(transcript) / audio.numel()
    assert ex[1].text == transcript[int(8 * sr * cps):int(18 * sr * cps)]
    assert ex[-1].metadata["end_sample"] == audio.numel()
    # too-short text or audio is dropped (training_data.py:103-105)
    assert proc._segment_basic(torch.zeros(sr // 2), "long enough transcript text", 0) == []
    assert proc._segment_basic(torch.zeros(3 * sr), "short", 0) == []

    words = [{"word": f"w{i:02d}", "start": 0.5 * i, "end": 0.5 * i + 0.4} for i in range(50)]    # 25 s of speech
    al = proc._segment_with_alignments(audio, transcript, 1, {"words": words})
    assert len(al) == 3
    assert al[0].metadata == {"start_sample": 0, "end_sample": int(9.9 * sr)}        # words 0..19 end within 10 s of sample 0
    assert al[0].text.split()[0] == "w00" and al[0].text.split()[-1] == "w19"
    assert al[1].metadata["start_sample"] == int(10.0 * sr) and al[1].text.split()[0] == "w20"
    assert proc._segment_with_alignments(audio, transcript, 1, {"words": []})[0].text == ex[0].text   # falls back (127-129)

    # end to end from files, with resampling and the alignment JSON
    t = np.arange(12 * 16000) / 16000
    write_wav(tmp_path / "r.wav", 0.3 * np.sin(2 * np.pi * 220 * t), 16000)
    (tmp_path / "r.txt").write_text("  " + transcript[:200] + "\n")
    got = proc.prepare_from_audio_file(tmp_path / "r.wav", tmp_path / "r.txt", 7)
    assert len(got) == 2 and got[0].audio.numel() == 10 * sr and got[1].metadata["end_sample"] == 12 * sr
    (tmp_path / "r.json").write_text(json.dumps({"words": words[:20]}))
    got = proc.prepare_from_audio_file(tmp_path / "r.wav", tmp_path / "r.txt", 7, tmp_path / "r.json")
    assert len(got) == 1 and got[0].speaker_id == 7


def conversation(n, sr=24000):
    return [TrainingExample(f"utterance number {i} of the talk", torch.randn(sr * (1 + i % 3)) * 0.1, i % 2) for i in range(n)]


def test_contextual_examples_and_dataset_items():
    conv = conversation(6)
    ctx = ContextualExampleGenerator(max_context_turns=3).create_contextual_examples(conv)
    assert [len(c["context"]) for c in ctx] == [0, 1, 2, 3, 3, 3]
    assert ctx[5]["context"] == conv[2:5] and ctx[5]["target"] is conv[5]
    tt, at = MockTextTokenizer(), MockAudioTokenizer()
    ds = CSMDataset(ctx, tt, at, max_seq_len=2048)
    assert len(ds) == 6
    it = ds[0]                                                          # no context: target text frames only
    n_text = len(tt.encode("[0]" + conv[0].text))
    assert it["input_tokens"].shape == (n_text, 33) and it["input_masks"].dtype == torch.bool
    assert bool(it["input_masks"][:, -1].all()) and not bool(it["input_masks"][:, :-1].any())
    assert it["input_tokens"][:, -1].tolist() == tt.encode("[0]" + conv[0].text)
    assert it["target_audio_tokens"].shape == (math.ceil(conv[0].audio.numel() / 1920), 32)
    it = ds[2]                                                          # two context turns: text + audio + EOS frame each
    lens = [len(tt.encode(f"[{c.speaker_id}]{c.text}")) + math.ceil(c.audio.numel() / 1920) + 1 for c in conv[:2]]
    n_t = len(tt.encode(f"[{conv[2].speaker_id}]{conv[2].text}"))
    assert it["input_tokens"].shape[0] == sum(lens) + n_t
    a0 = len(tt.encode(f"[{conv[0].speaker_id}]{conv[0].text}"))
    first_audio = it["input_tokens"][a0:lens[0]]
    assert bool(it["input_masks"][a0:lens[0], :32].all()) and not bool(it["input_masks"][a0:lens[0], 32].any())
    assert torch.equal(first_audio[:-1, :32], at.encode(conv[0].audio.reshape(1, 1, -1))[0].t())
    assert int(first_audio[-1].abs().sum()) == 0                        # the EOS frame is all zeros (training_data.py:319-321)
    assert ds.lengths()[2] == it["input_tokens"].shape[0]
    # over-length items: the reference keeps exactly the target's text frames (289-295); keep_context keeps the tail
    short = CSMDataset(ctx, tt, at, max_seq_len=40)
    assert short[3]["input_tokens"].shape[0] == min(40, len(tt.encode(f"[{conv[3].speaker_id}]{conv[3].text}")))
    tail = CSMDataset(ctx, tt, at, max_seq_len=40, truncate="keep_context")[3]
    assert tail["input_tokens"].shape[0] == 40 and torch.equal(tail["input_tokens"], ds[3]["input_tokens"][-40:])
    with pytest.raises(TypeError):
        CSMDataset(ctx, tt, object())


def test_collate_padding_and_bucketing():
    conv = conversation(12)
    ds = CSMDataset(ContextualExampleGenerator(2).create_contextual_examples(conv), MockTextTokenizer(), MockAudioTokenizer())
    items = [ds[i] for i in range(4)]
    b = collate_variable_length(items)
    S, T = max(i["input_tokens"].shape[0] for i in items), max(i["target_audio_tokens"].shape[0] for i in items)
    assert b["input_tokens"].shape == (4, S, 33) and b["input_masks"].shape == (4, S, 33) and b["target_audio_tokens"].shape == (4, T, 32)
    s0 = items[0]["input_tokens"].shape[0]
    assert int(b["input_tokens"][0, s0:].abs().sum()) == 0 and not bool(b["input_masks"][0, s0:].any())
    t0 = items[0]["target_audio_tokens"].shape[0]
    assert int(b["target_audio_tokens"][0, t0:].abs().sum()) == 0       # reference: zero padding (399-402)
    bi = collate_variable_length(items, target_pad=IGNORE_INDEX)
    assert bool((bi["target_audio_tokens"][0, t0:] == IGNORE_INDEX).all()) and torch.equal(bi["input_tokens"], b["input_tokens"])
    lengths = ds.lengths()
    smp = LengthBucketSampler(lengths, batch_size=3, bucket_batches=4, seed=1)
    batches = list(smp)
    assert sorted(i for bt in batches for i in bt) == list(range(12)) and all(len(bt) == 3 for bt in batches)
    spread = np.mean([max(lengths[i] for i in bt) - min(lengths[i] for i in bt) for bt in batches])
    plain = np.mean([max(lengths[i:i + 3]) - min(lengths[i:i + 3]) for i in range(0, 12, 3)])
    assert spread <= plain
    assert list(LengthBucketSampler(lengths, 3, 4, seed=1)) == batches   # deterministic per (seed, epoch)
    smp.set_epoch(1)
    assert list(smp) != batches
    r0, r1 = (list(LengthBucketSampler(lengths, 3, 4, seed=1, rank=r, world_size=2)) for r in (0, 1))
    assert len(r0) == len(r1) == 2 and not set(map(tuple, r0)) & set(map(tuple, r1))
    dl = create_dataloader(ds, batch_size=3, shuffle=True, num_workers=0, pin_memory=False, bucket_by_length=True, target_pad=IGNORE_INDEX)
    got = [x["input_tokens"].shape[0] for x in dl]
    assert sum(got) == 12
    dl = create_dataloader(ds, batch_size=5, shuffle=False, num_workers=0, pin_memory=False)
    assert [x["input_tokens"].shape[0] for x in dl] == [5, 5, 2]


def test_against_reference_fixture():
    """tests/golden/golden_data.npz holds what the REFERENCE's own data module (loaded by file path in the build container,
    see tests/golden/make_golden_data.py) produced for these seeded inputs: segment bounds and texts, contextual examples,
    every dataset item (frame layout, EOS frame, over-length rule) and a collated batch.  Ours must reproduce it exactly,
    with the tokenizer returning its codes as a list, as a [1,K,T] tensor (Mimi's form, on which the reference itself
    breaks) or as a [K,T] tensor."""
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(here, "golden_data.npz"))
    meta = json.load(open(os.path.join(here, "golden_data_meta.json")))
    sr = 24000
    g = torch.Generator().manual_seed(2024)
    audio = torch.randn(31 * sr + 777, generator=g) * 0.1
    transcript = " ".join(f"word{i:03d}" for i in range(130))
    words = [{"word": f"w{i:02d}", "start": 0.45 * i + 0.02, "end": 0.45 * i + 0.40} for i in range(66)]
    conv_audio = [torch.randn(sr * (1 + i % 3) + 100 * i, generator=g) * 0.1 for i in range(5)]
    conv_text = [f"turn {i}: the rain in spain stays mainly in the plain" for i in range(5)]

    proc = CSMDataProcessor(sample_rate=sr, segment_duration_ms=10000, overlap_ms=2000)
    basic = proc._segment_basic(audio, transcript, 5)
    assert [[e.metadata["start_sample"], e.metadata["end_sample"]] for e in basic] == z["basic_bounds"].tolist()
    assert [e.text for e in basic] == meta["basic_texts"]
    al = proc._segment_with_alignments(audio, transcript, 2, {"words": words})
    assert [[e.metadata["start_sample"], e.metadata["end_sample"]] for e in al] == z["aligned_bounds"].tolist()
    assert [e.text for e in al] == meta["aligned_texts"]
    conv = [TrainingExample(t, a, i % 2) for i, (t, a) in enumerate(zip(conv_text, conv_audio))]
    ctx = ContextualExampleGenerator(max_context_turns=2).create_contextual_examples(conv)
    assert [len(c["context"]) for c in ctx] == meta["context_lens"]
    assert [c["context"][0].text if c["context"] else None for c in ctx] == meta["context_first_text"]

    class Tok:
        def __init__(self, form):
            self.form = form

        def encode(self, wav):
            t = math.ceil(wav.shape[-1] / 1920)
            base = (wav.reshape(-1)[:t].abs() * 1000).long() % 2048
            codes = (base[None, :] + torch.arange(32)[:, None]) % 2048
            return {"list": [codes], "b": codes[None], "kt": codes}[self.form]

    for form in ("list", "b", "kt"):
        for name, max_len in (("full", 2048), ("short", 40)):
            ds = CSMDataset(ctx, MockTextTokenizer(), Tok(form), max_seq_len=max_len)
            items = [ds[i] for i in range(len(ds))]
            for i, it in enumerate(items):
                assert torch.equal(it["input_tokens"], torch.from_numpy(z[f"{name}_tokens_{i}"]).long()), (form, name, i)
                assert torch.equal(it["input_masks"], torch.from_numpy(z[f"{name}_masks_{i}"])), (form, name, i)
                assert torch.equal(it["target_audio_tokens"], torch.from_numpy(z[f"{name}_targets_{i}"]).long()), (form, name, i)
            if name == "full":
                b = collate_variable_length(items[1:4])
                assert torch.equal(b["input_tokens"], torch.from_numpy(z["collate_tokens"]).long())
                assert torch.equal(b["input_masks"], torch.from_numpy(z["collate_masks"]))
                assert torch.equal(b["target_audio_tokens"], torch.from_numpy(z["collate_targets"]).long())
