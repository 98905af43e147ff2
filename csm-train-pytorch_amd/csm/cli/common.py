"""Dataset plumbing shared by the CLIs.

Three sources, in this order:
* ``--audio-dir`` / ``--transcript-dir`` (/ ``--alignment-dir``): the reference flow (``src/csm/cli/train.py:228-329``) -
  ``CSMDataProcessor`` cuts every ``*.wav`` + ``*.txt`` pair into utterances, which are tokenised ONCE, up front, with
  the Llama-3 tokenizer (``--text-tokenizer`` = local directory) and Mimi on the GPU (``--mimi-weights`` = local file;
  the reference downloads both from the hub).  Tokenising inside DataLoader workers, as the reference does, would
  put the codec on the CPU and stall the step.
* ``--token-file``: a ``torch.save``d list of dicts with ``input_tokens [S,33]`` int64, ``input_masks [S,33]`` bool,
  ``target_audio_tokens [T,32]`` int64 (exactly what ``CSMDataset.__getitem__`` yields, training_data.py:245-302).
* ``--synthetic N``: seeded synthetic sequences.
"""
from pathlib import Path

import torch
from torch.utils.data import Dataset

from ..data import SyntheticCSMDataset, collate_variable_length


class TokenFileDataset(Dataset):
    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def get_batch(self, batch_idx, batch_size):   # protocol of CSMLoRATrainer.train
        return collate_variable_length([self.items[(batch_idx * batch_size + j) % len(self.items)] for j in range(batch_size)])


def add_data_args(parser, context_turns: int = 0):
    g = parser.add_argument_group("Data")
    g.add_argument("--token-file", type=str, default=None, help="torch.save'd list of tokenised examples")
    g.add_argument("--synthetic", type=int, default=0, help="use N seeded synthetic sequences instead of a token file")
    g.add_argument("--max-seq-len", type=int, default=2048, help="sequence length of synthetic data / truncation length")
    g.add_argument("--val-split", type=float, default=0.1, help="fraction of the examples held out for validation")
    g.add_argument("--audio-dir", type=str, default=None, help="directory of *.wav recordings (reference flag)")
    g.add_argument("--transcript-dir", type=str, default=None, help="directory of *.txt transcripts, same stems (reference flag)")
    g.add_argument("--alignment-dir", type=str, default=None, help="optional directory of word-alignment *.json (reference flag)")
    g.add_argument("--speaker-id", type=int, default=0)
    g.add_argument("--mimi-weights", type=str, default=None, help="local Mimi weights (transformers.MimiModel state dict)")
    g.add_argument("--text-tokenizer", type=str, default=None, help="local directory of the Llama-3.2 tokenizer files")
    g.add_argument("--context-turns", type=int, default=context_turns, help="previous utterances of the same file given as context")
    g.add_argument("--ignore-padding", action="store_true", help="pad targets with -100 so padded frames leave the loss")


def load_raw_examples(args, logger=None):
    """Reference ``load_data`` (cli/train.py:228-278): every ``<audio-dir>/<stem>.wav`` with ``<transcript-dir>/<stem>.txt``."""
    from ..data import CSMDataProcessor
    proc, per_file = CSMDataProcessor(), []
    for wav in sorted(Path(args.audio_dir).glob("*.wav")):
        txt = Path(args.transcript_dir) / f"{wav.stem}.txt"
        if not txt.exists():
            if logger:
                logger.warning(f"No transcript found for {wav}, skipping")
            continue
        al = Path(args.alignment_dir) / f"{wav.stem}.json" if args.alignment_dir else None
        per_file.append(proc.prepare_from_audio_file(wav, txt, args.speaker_id, al if al is not None and al.exists() else None))
    return per_file


def tokenise_examples(per_file, text_tokenizer, audio_tokenizer, max_seq_len, context_turns=0):
    """Contextual examples (reference ``prepare_dataset`` cli/train.py:281-329: no context) -> tokenised items."""
    from ..data import ContextualExampleGenerator, CSMDataset
    gen = ContextualExampleGenerator(max_context_turns=context_turns)
    ctx = []
    for examples in per_file:
        ctx += gen.create_contextual_examples(examples) if context_turns > 0 else [{"context": [], "target": e} for e in examples]
    ds = CSMDataset(ctx, text_tokenizer, audio_tokenizer, max_seq_len=max_seq_len)
    return [ds[i] for i in range(len(ds))]


def load_datasets(args):
    if args.synthetic:
        n_val = max(1, int(args.synthetic * args.val_split)) if args.val_split > 0 else 0
        train = SyntheticCSMDataset(args.synthetic - n_val, args.max_seq_len, seed=1234)
        val = SyntheticCSMDataset(n_val, args.max_seq_len, seed=4321) if n_val else None
        return train, val
    if args.audio_dir:
        if not (args.transcript_dir and args.mimi_weights and args.text_tokenizer):
            raise SystemExit("--audio-dir needs --transcript-dir, --mimi-weights and --text-tokenizer (local files: the hub "
                             "downloads of the reference are not available offline)")
        from ..codec import load_mimi
        from ..generator import load_llama3_tokenizer
        items = tokenise_examples(load_raw_examples(args), load_llama3_tokenizer(args.text_tokenizer),
                                  load_mimi(args.mimi_weights), args.max_seq_len, args.context_turns)
        n_val = int(len(items) * args.val_split)              # reference split: the tail is the validation set (train.py:293-300)
        n_train = len(items) - n_val
        return TokenFileDataset(items[:n_train]), (TokenFileDataset(items[n_train:]) if n_val else None)
    if not args.token_file:
        raise SystemExit("give --audio-dir/--transcript-dir (+ --mimi-weights, --text-tokenizer), --token-file or --synthetic N")
    items = torch.load(args.token_file, map_location="cpu", weights_only=False)
    items = [{k: v[:args.max_seq_len] for k, v in it.items()} for it in items]
    n_val = int(len(items) * args.val_split)
    return TokenFileDataset(items[n_val:]), (TokenFileDataset(items[:n_val]) if n_val else None)
