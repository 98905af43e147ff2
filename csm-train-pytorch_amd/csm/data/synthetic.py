"""Seeded synthetic interleaved batches - what bench.py and the plumbing tests train on."""
import torch
from torch.utils.data import Dataset

from .training_data import collate_variable_length


class SyntheticCSMDataset(Dataset):
    """Seeded interleaved text+audio token sequences of a fixed length - what bench.py trains on."""

    def __init__(self, n_items: int, seq_len: int, text_vocab: int = 128256, audio_vocab: int = 2051, n_codebooks: int = 32,
                 seed: int = 1234, n_segments: int = 2):
        self.n, self.S, self.tv, self.av, self.K, self.seed, self.nseg = n_items, seq_len, text_vocab, audio_vocab, n_codebooks, seed, n_segments

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        S, K = self.S, self.K
        tokens = torch.zeros(S, K + 1, dtype=torch.long)
        mask = torch.zeros(S, K + 1, dtype=torch.bool)
        p = 0
        for sgi in range(self.nseg):
            end = S if sgi == self.nseg - 1 else (S * (sgi + 1)) // self.nseg
            hi = max(2, min(48, (end - p) // 2))
            nt = int(torch.randint(min(16, hi - 1) if hi > 1 else 1, hi + 1, (1,), generator=g))
            nt = min(nt, end - p)
            tokens[p:p + nt, K] = torch.randint(0, self.tv, (nt,), generator=g)
            mask[p:p + nt, K] = True
            p += nt
            if end > p:
                codes = torch.randint(0, self.av - 3, (end - p, K), generator=g)
                codes[-1] = 0   # EOS frame
                tokens[p:end, :K] = codes
                mask[p:end, :K] = True
            p = end
        targets = torch.randint(0, self.av - 3, (S, K), generator=g)
        return {"input_tokens": tokens, "input_masks": mask, "target_audio_tokens": targets}

    # protocol consumed by CSMLoRATrainer.train (reference training/data.py:364-388)
    def get_batch(self, batch_idx: int, batch_size: int):
        items = [self[(batch_idx * batch_size + j) % self.n] for j in range(batch_size)]
        return collate_variable_length(items)
