"""Training utilities - same surface as reference ``src/csm/training/utils.py`` (torch half; the MLX half of that
file has no meaning on ROCm and is out of scope)."""
import logging
import os
from typing import Dict, Optional, Tuple

import torch


def setup_logger(name: str, log_file: Optional[str] = None, level: int = logging.INFO) -> logging.Logger:
    """Console + optional file logger (reference utils.py:14-53)."""
    logger = logging.getLogger(name)
    logger.setLevel(level)
    for handler in logger.handlers[:]:
        logger.removeHandler(handler)
    fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    console = logging.StreamHandler()
    console.setLevel(level)
    console.setFormatter(fmt)
    logger.addHandler(console)
    if log_file:
        d = os.path.dirname(log_file)
        if d:
            os.makedirs(d, exist_ok=True)
        fh = logging.FileHandler(log_file)
        fh.setLevel(level)
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    return logger


class _EngineLoss(torch.autograd.Function):
    """Ties the explicit HIP forward/backward schedule into ``loss.backward()`` so the reference's inner loop
    (``compute_loss`` -> ``(loss / accumulation_steps).backward()``, trainer.py:253-263) runs unchanged."""

    @staticmethod
    def forward(ctx, anchor, model, total):
        ctx.model = model
        return total.clone()

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.model
        m.engine.backward(float(grad_out))
        return None, None, None


def compute_loss(model, input_tokens: torch.Tensor, input_masks: torch.Tensor, target_audio_tokens: torch.Tensor,
                 semantic_weight: float = 100.0, acoustic_weight: float = 1.0,
                 acoustic_rows: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """Loss of reference ``compute_loss`` (utils.py:56-119) on the HIP path.

    semantic = mean CE of codebook-0 logits at positions [0, S-1) vs ``target_audio_tokens[:, :S-1, 0]`` (no ignore
    index, exactly as the reference).  The acoustic term is the reference's literal 0 placeholder unless
    ``model.acoustic_mode`` is "all" / "amortized" (teacher-forced depth-decoder CE over every / a 1/16 sample of the
    frames, the recipe the reference documents but never implemented); ``acoustic_rows`` pins the sampled rows.
    Returns (total, {"semantic_loss", "acoustic_loss"}).  ``total.backward()`` accumulates into ``param.grad``.
    """
    need_grad = torch.is_grad_enabled()
    total, sem, ac = model.engine.forward_loss(input_tokens, input_masks, target_audio_tokens, semantic_weight,
                                               acoustic_weight, save=need_grad, acoustic_rows=acoustic_rows)
    losses = {"semantic_loss": sem, "acoustic_loss": ac}
    if need_grad:
        if not hasattr(model, "_anchor"):
            model._anchor = torch.zeros((), device=model.device, requires_grad=True)
        total = _EngineLoss.apply(model._anchor, model, total)
    return total, losses


def save_checkpoint(model, optimizer, epoch: int, global_step: int, loss: float, save_dir: str, name: str = "checkpoint",
                    optimizer_state=None) -> str:
    """Reference utils.py:526-574: one ``.pt`` dict written as ``{name}_epoch{e}_step{s}.pt`` and ``{name}_latest.pt``.
    ``optimizer_state``: an already gathered optimiser state dict (a sharded optimiser's ``state_dict`` is a collective the
    caller runs on every rank; only the writing rank comes here)."""
    os.makedirs(save_dir, exist_ok=True)
    ckpt = {
        "model": {k: v.cpu() for k, v in model.state_dict().items()},
        "optimizer": optimizer_state if optimizer_state is not None else (optimizer.state_dict() if optimizer is not None else None),
        "epoch": epoch,
        "global_step": global_step,
        "loss": float(loss),
    }
    path = os.path.join(save_dir, f"{name}_epoch{epoch}_step{global_step}.pt")
    for dst in (path, os.path.join(save_dir, f"{name}_latest.pt")):
        tmp = dst + ".tmp"                       # write-then-rename: a reader (or a crash) never sees half a checkpoint
        torch.save(ckpt, tmp)
        os.replace(tmp, dst)
    return path


def load_checkpoint(checkpoint_path: str, model, optimizer=None, device: Optional[str] = None) -> Dict:
    """Reference utils.py:864-895: restores model (+ optimizer) and returns {"epoch","global_step","loss"}."""
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    model.load_state_dict(ckpt["model"])
    if optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    return {"epoch": ckpt.get("epoch", 0), "global_step": ckpt.get("global_step", 0), "loss": ckpt.get("loss", float("inf"))}
