"""Mint tests/golden/golden_data.npz: outputs of the REFERENCE's own data front end (run ONLY in the build container).

    python tests/golden/make_golden_data.py

``/root/reference/src/csm/data/training_data.py`` is loaded by file path.  It imports ``torchaudio`` (absent here; used
only by ``prepare_from_audio_file``, which this script does not call) and ``csm.generator.Segment`` (a three-field
dataclass whose module drags in torchtune / moshi); both are given as stub modules, exactly as make_golden.py does
for torchtune.  What the fixture pins is the pure-Python logic the MI355X data module mirrors: ``_segment_basic``,
``_segment_with_alignments``, ``ContextualExampleGenerator.create_contextual_examples``, ``CSMDataset.__getitem__``
(frame layout, EOS frame, over-length rule) and ``collate_variable_length``.  Inputs are regenerated from seeds by the
test; only expected outputs (integers / index lists) are stored.
"""
import importlib.util
import json
import math
import os
import sys
import types
from dataclasses import dataclass

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/csm/data/training_data.py"


def load_reference():
    sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))
    csm = types.ModuleType("csm")
    gen = types.ModuleType("csm.generator")

    @dataclass
    class Segment:
        speaker: int
        text: str
        audio: torch.Tensor

    gen.Segment = Segment
    csm.generator = gen
    sys.modules["csm"], sys.modules["csm.generator"] = csm, gen
    spec = importlib.util.spec_from_file_location("ref_training_data", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class MockTextTokenizer:
    def encode(self, text):
        return [1] + [3 + (ord(c) % 50) for c in text] + [2]


class MockAudioTokenizer:
    """[1,1,N] -> LIST holding one [32, ceil(N/1920)] code matrix: the only return form with which both of the reference's
    call sites work (``encode(...)[0]`` at training_data.py:318 and the ``isinstance(list)`` branch at 343-347; a
    [1,K,T] tensor, which is what Mimi returns, reaches ``collate_variable_length`` as a 3-D target and fails there)."""

    def encode(self, wav):
        t = math.ceil(wav.shape[-1] / 1920)
        base = (wav.reshape(-1)[:t].abs() * 1000).long() % 2048
        return [(base[None, :] + torch.arange(32)[:, None]) % 2048]


def inputs():
    """Deterministic inputs shared with tests/test_data_cpu.py::test_against_reference_fixture."""
    sr = 24000
    g = torch.Generator().manual_seed(2024)
    audio = torch.randn(31 * sr + 777, generator=g) * 0.1
    transcript = " ".join(f"word{i:03d}" for i in range(130))
    words = [{"word": f"w{i:02d}", "start": 0.45 * i + 0.02, "end": 0.45 * i + 0.40} for i in range(66)]
    conv_audio = [torch.randn(sr * (1 + i % 3) + 100 * i, generator=g) * 0.1 for i in range(5)]
    conv_text = [f"turn {i}: the rain in spain stays mainly in the plain" for i in range(5)]
    return sr, audio, transcript, words, conv_audio, conv_text


def main():
    R = load_reference()
    sr, audio, transcript, words, conv_audio, conv_text = inputs()
    out = {}
    proc = R.CSMDataProcessor(sample_rate=sr, segment_duration_ms=10000, overlap_ms=2000)
    basic = proc._segment_basic(audio, transcript, 5)
    out["basic_bounds"] = np.array([[e.metadata["start_sample"], e.metadata["end_sample"]] for e in basic], dtype=np.int64)
    meta = {"basic_texts": [e.text for e in basic]}
    al = proc._segment_with_alignments(audio, transcript, 2, {"words": words})
    out["aligned_bounds"] = np.array([[e.metadata["start_sample"], e.metadata["end_sample"]] for e in al], dtype=np.int64)
    meta["aligned_texts"] = [e.text for e in al]
    conv = [R.TrainingExample(text=t, audio=a, speaker_id=i % 2) for i, (t, a) in enumerate(zip(conv_text, conv_audio))]
    ctx = R.ContextualExampleGenerator(max_context_turns=2).create_contextual_examples(conv)
    meta["context_lens"] = [len(c["context"]) for c in ctx]
    meta["context_first_text"] = [c["context"][0].text if c["context"] else None for c in ctx]
    for name, max_len in (("full", 2048), ("short", 40)):
        ds = R.CSMDataset(ctx, MockTextTokenizer(), MockAudioTokenizer(), max_seq_len=max_len)
        items = [ds[i] for i in range(len(ds))]
        for i, it in enumerate(items):
            out[f"{name}_tokens_{i}"] = it["input_tokens"].numpy().astype(np.int32)
            out[f"{name}_masks_{i}"] = it["input_masks"].numpy()
            out[f"{name}_targets_{i}"] = it["target_audio_tokens"].numpy().astype(np.int32)
        if name == "full":
            b = R.collate_variable_length(items[1:4])
            out["collate_tokens"] = b["input_tokens"].numpy().astype(np.int32)
            out["collate_masks"] = b["input_masks"].numpy()
            out["collate_targets"] = b["target_audio_tokens"].numpy().astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "golden_data.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "golden_data_meta.json"), "w"), indent=1)
    print("wrote golden_data.npz:", {k: v.shape for k, v in list(out.items())[:6]}, "...", len(out), "arrays")


if __name__ == "__main__":
    main()
