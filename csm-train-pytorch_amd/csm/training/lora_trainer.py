"""``CSMLoRATrainer`` - API of reference ``src/csm/training/lora_trainer.py`` (+ the ``train`` loop it inherits from
``CSMMLXTrainer``, mlx_trainer.py:733-876) on MI355X.

The reference class is MLX-only and raises ImportError without MLX (lora_trainer.py:71-75); its MLX transformer is
numerically different from ``Model`` and swallows every exception.  This class keeps the
constructor, attributes, method names, dataset protocol (``len(ds)``, ``ds.get_batch(i, bs)``) and return types,
takes the block numerics from ``Model`` (the PyTorch path the north star calls the reference CPU path) and the LoRA
maths from ``mlx/components/lora.py``, and raises on errors instead of substituting constants.
"""
import json
import math
import os
import time
from pathlib import Path
from typing import List, Optional

import numpy as np
import torch

from ..models.model import Model
from .dp import GradSync
from .lora import apply_lora_to_model, merge_lora_weights
from .optim import FusedAdamW
from .trainer import csm_1b_args
from .utils import compute_loss, setup_logger


def _to_torch(x):
    if torch.is_tensor(x):
        return x
    return torch.as_tensor(np.asarray(x))


class CSMLoRATrainer:
    """LoRA trainer for CSM models (reference lora_trainer.py:29)."""

    def __init__(self, model_path: str, output_dir: str, log_file: Optional[str] = None, learning_rate: float = 1e-4,
                 semantic_weight: float = 100.0, acoustic_weight: float = 1.0, weight_decay: float = 0.01,
                 lora_r: int = 8, lora_alpha: float = 16.0, lora_dropout: float = 0.0,
                 target_modules: Optional[List[str]] = None, target_layers: Optional[List[int]] = None,
                 lora_use_bias: bool = False, device: str = "cuda", model: Optional[Model] = None):
        self.model_path = model_path
        self.output_dir = Path(output_dir)
        self.output_dir.mkdir(parents=True, exist_ok=True)
        self.logger = setup_logger("csm_lora_trainer", log_file or str(self.output_dir / "lora_training.log"))
        self.learning_rate = learning_rate
        self.semantic_weight = semantic_weight
        self.acoustic_weight = acoustic_weight
        self.weight_decay = weight_decay
        self.lora_r = lora_r
        self.lora_alpha = lora_alpha
        self.lora_dropout = lora_dropout
        self.target_modules = target_modules
        self.target_layers = target_layers
        self.lora_use_bias = lora_use_bias
        self.device = device
        self.logger.info(f"Loading model from {model_path} and applying LoRA")
        self.model = model
        self.optimizer = None
        self.grad_sync = None
        self.max_grad_norm = 0.0
        self._load_model_with_lora()
        self.epoch = 0
        self.global_step = 0
        self.best_loss = float("inf")

    def _load_model_with_lora(self):
        """Reference lora_trainer.py:112-303: load (safetensors or torch ``.pt``), then ``apply_lora_to_model``."""
        if self.model is None:
            self.model = Model(csm_1b_args(), device=self.device)
            if self.model_path:
                if self.model_path.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    sd = load_file(self.model_path)
                else:
                    sd = torch.load(self.model_path, map_location="cpu", weights_only=False)
                    if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
                        sd = sd["model"]
                self.model.load_state_dict(sd)
        apply_lora_to_model(self.model, r=self.lora_r, alpha=self.lora_alpha, dropout=self.lora_dropout,
                            target_modules=self.target_modules, target_layers=self.target_layers,
                            use_bias=self.lora_use_bias)

    def prepare_optimizer(self):
        """Reference lora_trainer.py:305-372: Adam(lr) over ``model.get_lora_params()`` only (no weight decay)."""
        n = self.model.lora.num_params()
        if GradSync.active():
            GradSync.broadcast_parameters(self.model)      # base weights and adapters of rank 0 everywhere, before the master copy
            # adapter dropout must differ between ranks (each rank sees its own shard): fold the rank into the seeds
            rk = torch.distributed.get_rank()
            for ad in self.model.lora.adapters.values():
                ad.seed = ad.seed * 31 + rk
        self.optimizer = FusedAdamW(self.model, {}, lora_lr=self.learning_rate, lora_weight_decay=0.0)
        base = sum(self.model.group_range(g)[1] for g in ("backbone", "decoder"))
        self.logger.info(f"Training with {n:,} LoRA parameters ({100.0 * n / base:.3f}% of the transformer stacks)")
        self._ensure_grad_sync()

    def _ensure_grad_sync(self):
        """(Re-)attach the gradient exchange; ``train`` closes it when it returns (GradSync.close)."""
        if self.grad_sync is None and GradSync.active():
            self.grad_sync = GradSync.for_model(self.model)

    def train_step(self, batch):
        """Reference lora_trainer.py:374-457: loss + grads w.r.t. LoRA params -> optional clip -> Adam.  Returns the
        loss as a 0-d tensor.  Errors propagate (no fallback constants)."""
        if self.optimizer is None:
            self.prepare_optimizer()
        m = self.model
        if self.grad_sync is not None:
            self.grad_sync.arm(True)
        loss, _ = compute_loss(m, _to_torch(batch["input_tokens"]), _to_torch(batch["input_masks"]),
                               _to_torch(batch["target_audio_tokens"]), self.semantic_weight, self.acoustic_weight)
        m.engine.backward(1.0 / (self.grad_sync.world_size if self.grad_sync is not None else 1))
        if self.grad_sync is not None:
            self.grad_sync.finish()
        if self.max_grad_norm and self.max_grad_norm > 0:
            self.optimizer.clip_grad_norm(self.max_grad_norm)
        self.optimizer.step(zero_grad=True)
        return loss.detach()

    def train(self, train_dataset, val_dataset=None, batch_size: int = 2, epochs: int = 5, val_every: int = 100,
              save_every: int = 500, max_grad_norm: float = 1.0, resume_from: Optional[str] = None) -> float:
        """Reference ``CSMMLXTrainer.train`` (mlx_trainer.py:733-876); unlike it, ``max_grad_norm`` takes effect
         and checkpoints are written."""
        if self.optimizer is None:
            self.prepare_optimizer()
        self._ensure_grad_sync()
        self.max_grad_norm = max_grad_norm
        if resume_from:
            self.load_lora_weights(resume_from)
        self.logger.info("Starting LoRA training")
        # data parallel (new capability): rank r takes batches r, r + world, r + 2 world, ... of the get_batch protocol, so
        # the ranks see disjoint data and one optimiser step covers world * batch_size sequences; replicas are identical,
        # so rank 0 alone writes files
        world = torch.distributed.get_world_size() if GradSync.active() else 1
        rank = torch.distributed.get_rank() if GradSync.active() else 0

        def save(path):
            if rank == 0:
                self.save_model(path, "lora")
            if world > 1:
                torch.distributed.barrier()

        for epoch in range(self.epoch, self.epoch + epochs):
            t0 = time.time()
            losses = []
            n_batches = len(train_dataset) // (batch_size * world)
            for batch_idx in range(n_batches):
                loss = self.train_step(train_dataset.get_batch(batch_idx * world + rank, batch_size))
                losses.append(loss)
                self.global_step += 1
                if val_dataset is not None and self.global_step % val_every == 0:
                    val_loss = GradSync.mean_scalar(self._validate(val_dataset, batch_size))
                    if rank == 0:
                        self.logger.info(f"Epoch {epoch + 1}, Step {self.global_step}, Val Loss: {val_loss:.6f}")
                    if val_loss < self.best_loss:
                        self.best_loss = val_loss
                        save(str(self.output_dir / "best"))
                if self.global_step % save_every == 0:
                    save(str(self.output_dir / f"checkpoint_step_{self.global_step}"))
            avg = float(torch.stack(losses).mean()) if losses else float("nan")
            avg = GradSync.mean_scalar(avg)
            if not math.isfinite(avg):
                raise FloatingPointError(f"non-finite training loss in epoch {epoch + 1}")
            if rank == 0:
                self.logger.info(f"Epoch {epoch + 1} completed in {time.time() - t0:.2f}s, Avg Loss: {avg:.6f}")
            self.epoch = epoch + 1
        if rank == 0:
            self.logger.info("Training completed")
        if self.grad_sync is not None:
            self.grad_sync.close()       # gives the process-global GEMM schedule switch back (a later train() re-attaches)
            self.grad_sync = None
        return self.best_loss

    def _validate(self, val_dataset, batch_size: int) -> float:
        """Reference mlx_trainer.py:878-972 (at most 10 batches, line 899)."""
        n = min(10, len(val_dataset) // batch_size)
        total = 0.0
        self.model.lora.training = False          # adapter dropout off while scoring
        try:
            with torch.no_grad():
                for i in range(n):
                    b = val_dataset.get_batch(i, batch_size)
                    loss, _ = compute_loss(self.model, _to_torch(b["input_tokens"]), _to_torch(b["input_masks"]),
                                           _to_torch(b["target_audio_tokens"]), self.semantic_weight, self.acoustic_weight)
                    total += float(loss)
        finally:
            self.model.lora.training = True
        return total / max(1, n)

    def save_model(self, save_path: str, save_mode: str = "lora"):
        """Reference lora_trainer.py:459-570: ``lora`` -> adapter safetensors + ``_metadata.json``; ``full`` -> merged
        model; ``both`` -> both."""
        from safetensors.torch import save_file
        if save_mode not in ("lora", "full", "both"):
            raise ValueError(f"unknown save_mode {save_mode!r}")
        base = save_path[:-len(".safetensors")] if save_path.endswith(".safetensors") else save_path
        d = os.path.dirname(base)
        if d:
            os.makedirs(d, exist_ok=True)
        lo = self.model.lora
        if save_mode in ("lora", "both"):
            path = base + ("_lora" if save_mode == "both" else "") + ".safetensors"
            save_file({k: v.detach().cpu().contiguous() for k, v in lo.named_tensors()}, path)
            meta = {"lora_r": self.lora_r, "lora_alpha": self.lora_alpha, "lora_dropout": self.lora_dropout,
                    "target_modules": lo.target_modules, "target_layers": self.target_layers,
                    "lora_use_bias": self.lora_use_bias, "params_count": lo.num_params()}
            with open(path.replace(".safetensors", "_metadata.json"), "w") as f:
                json.dump(meta, f, indent=2)
        if save_mode in ("full", "both"):
            path = base + ("_full" if save_mode == "both" else "") + ".safetensors"
            backup = self.model.arena.clone()
            merge_lora_weights(self.model)
            sd = {k: v.detach().cpu().contiguous() for k, v in self.model._views(self.model.arena).items()}
            self.model.arena.copy_(backup)
            self.model.params_rewritten(lora=False)
            save_file(sd, path)
        return base

    def load_lora_weights(self, lora_path: str):
        """Reference lora_trainer.py:572-633."""
        from safetensors.torch import load_file
        if not lora_path.endswith(".safetensors"):
            lora_path = lora_path + ".safetensors"
        sd = load_file(lora_path)
        names = dict(self.model.lora.named_tensors())
        missing = [k for k in names if k not in sd]
        if missing:
            raise KeyError(f"LoRA file lacks {len(missing)} tensors, e.g. {missing[:3]}")
        with torch.no_grad():
            for k, dst in names.items():
                dst.copy_(sd[k].to(device=dst.device, dtype=dst.dtype))
        if self.optimizer is not None and "lora" in self.optimizer.state:
            self.optimizer.set_master("lora", self.model.lora.arena.float())

    def generate_sample(self, text: str, speaker_id: int = 0, output_path: str = "sample.wav") -> str:
        """Reference lora_trainer.py:635-700."""
        from ..generator import Generator
        gen = Generator(self.model)
        audio = gen.generate(text=text, speaker=speaker_id, context=[])
        gen.save_wav(output_path, audio)
        return output_path
