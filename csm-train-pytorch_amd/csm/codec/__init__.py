"""Audio codec (Mimi) on the HIP path."""
import os

import torch

from .mimi import MimiCodec  # noqa: F401


def load_mimi(weights_path: str, device="cuda", num_codebooks: int = 32) -> MimiCodec:
    """Mimi from a local weights file (the reference downloads them from the hub, generator.py:67-70): a
    ``.safetensors`` / ``torch.save`` state dict with the key names of ``transformers.MimiModel`` (``kyutai/mimi``)."""
    if not os.path.exists(weights_path):
        raise FileNotFoundError(f"Mimi weights not found: {weights_path}")
    if weights_path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(weights_path)
    else:
        sd = torch.load(weights_path, map_location="cpu", weights_only=False)
        sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
    return MimiCodec(sd, device=device, num_codebooks=num_codebooks)
