"""Dataset plumbing shared by the CLIs.

The reference CLIs tokenise ``--audio-dir`` / ``--transcript-dir`` with Mimi and the Llama-3 tokenizer
(``src/csm/cli/train.py:228-329``), both of which need files this environment cannot fetch.  The MI355X CLIs therefore
take the frames *already tokenised*: ``--token-file`` = a ``torch.save``d list of dicts with ``input_tokens [S,33]``
int64, ``input_masks [S,33]`` bool, ``target_audio_tokens [T,32]`` int64 (exactly what ``CSMDataset.__getitem__``
yields, ``src/csm/data/training_data.py:245-302``), or ``--synthetic N`` for seeded synthetic sequences.
"""
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


def add_data_args(parser):
    g = parser.add_argument_group("Data")
    g.add_argument("--token-file", type=str, default=None, help="torch.save'd list of tokenised examples")
    g.add_argument("--synthetic", type=int, default=0, help="use N seeded synthetic sequences instead of a token file")
    g.add_argument("--max-seq-len", type=int, default=2048, help="sequence length of synthetic data / truncation length")
    g.add_argument("--val-split", type=float, default=0.1, help="fraction of the examples held out for validation")
    # accepted for command-line compatibility with the reference; raw audio cannot be tokenised offline
    for name in ("--audio-dir", "--transcript-dir", "--alignment-dir"):
        g.add_argument(name, type=str, default=None, help="(reference flag) raw data directory - needs Mimi + tokenizer weights")
    g.add_argument("--speaker-id", type=int, default=0)


def load_datasets(args):
    if args.synthetic:
        n_val = max(1, int(args.synthetic * args.val_split)) if args.val_split > 0 else 0
        train = SyntheticCSMDataset(args.synthetic - n_val, args.max_seq_len, seed=1234)
        val = SyntheticCSMDataset(n_val, args.max_seq_len, seed=4321) if n_val else None
        return train, val
    if not args.token_file:
        raise SystemExit("give --token-file (pre-tokenised examples) or --synthetic N; tokenising --audio-dir needs Mimi and "
                         "Llama-3 tokenizer weights that are not available offline")
    items = torch.load(args.token_file, map_location="cpu", weights_only=False)
    items = [{k: v[:args.max_seq_len] for k, v in it.items()} for it in items]
    n_val = int(len(items) * args.val_split)
    return TokenFileDataset(items[n_val:]), (TokenFileDataset(items[:n_val]) if n_val else None)
