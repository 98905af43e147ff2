"""``csm-generate`` on MI355X: flag names and defaults of reference ``src/csm/cli/generate.py:28-105``.  The two things the
reference downloads from the hub are passed as local files (``--mimi-weights``, ``--text-tokenizer``); WAV I/O goes through
``csm.data.load_audio`` / ``Generator.save_wav`` because torchaudio is not installed."""
import argparse
import os

from ..data import load_audio, resample
from ..generator import Segment, load_csm_1b

# reference cli/generate.py:16-25
VOICE_PRESETS = {"neutral": 0, "warm": 1, "deep": 2, "bright": 3, "soft": 4, "energetic": 5, "calm": 6, "clear": 7,
                 "resonant": 8, "authoritative": 9}


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Generate speech with CSM (MI355X)")
    p.add_argument("--model-path", type=str, required=True, help="Path to the model checkpoint (the hub download is not available offline)")
    p.add_argument("--text", type=str, required=True, help="Text to generate speech for")
    voice = p.add_mutually_exclusive_group()
    voice.add_argument("--speaker", type=int, default=0, help="Speaker ID (default: 0)")
    voice.add_argument("--voice", type=str, choices=VOICE_PRESETS.keys(), help="Voice preset to use")
    p.add_argument("--output", type=str, default="audio.wav", help="Output file path (default: audio.wav)")
    p.add_argument("--context-audio", type=str, nargs="*", help="Path(s) to audio file(s) to use as context")
    p.add_argument("--context-text", type=str, nargs="*", help="Text(s) corresponding to the context audio files")
    p.add_argument("--context-speaker", type=int, nargs="*", help="Speaker ID(s) for the context segments")
    p.add_argument("--max-audio-length-ms", type=int, default=10000)
    p.add_argument("--temperature", type=float, default=0.9)
    p.add_argument("--topk", type=int, default=50)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--mimi-weights", type=str, required=True, help="local Mimi weights (transformers.MimiModel state dict)")
    p.add_argument("--text-tokenizer", type=str, required=True, help="local directory of the Llama-3.2 tokenizer files")
    return p.parse_args(argv)


def build_context(args, sample_rate):
    """Reference generate.py:129-156: (audio, text, speaker) triples -> Segments at the generator's sample rate."""
    context = []
    if args.context_audio:
        if not (args.context_text and args.context_speaker):
            raise ValueError("If context audio is provided, context text and speaker must also be provided")
        if not (len(args.context_audio) == len(args.context_text) == len(args.context_speaker)):
            raise ValueError("The number of context audio, text, and speaker entries must be the same")
        for path, text, speaker in zip(args.context_audio, args.context_text, args.context_speaker):
            wav, sr = load_audio(path)
            wav = resample(wav.mean(0) if wav.size(0) > 1 else wav.squeeze(0), sr, sample_rate)
            context.append(Segment(text=text, speaker=speaker, audio=wav))
    return context


def main(argv=None):
    args = parse_args(argv)
    speaker_id = VOICE_PRESETS[args.voice] if args.voice else args.speaker
    generator = load_csm_1b(args.model_path, args.device, mimi_weights=args.mimi_weights, tokenizer_path=args.text_tokenizer)
    context = build_context(args, generator.sample_rate)
    audio = generator.generate(text=args.text, speaker=speaker_id, context=context, max_audio_length_ms=args.max_audio_length_ms,
                               temperature=args.temperature, topk=args.topk)
    os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
    generator.save_wav(args.output, audio)
    print(f"Audio saved to {args.output} ({audio.numel() / generator.sample_rate:.2f} s at {generator.sample_rate} Hz)")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
