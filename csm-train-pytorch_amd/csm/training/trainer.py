"""``CSMTrainer`` - the "csm-train" loop of reference ``src/csm/training/trainer.py`` on MI355X.

Constructor, ``prepare_optimizer`` and ``train`` keep the reference's signatures and loop semantics (per-group
learning rates by name substring, gradient accumulation, global-norm clipping, validation / checkpoint cadence);
underneath, the step is HIP kernels + a fused AdamW, and - new capability, the reference has no distributed code -
one process per GPU with bucketed RCCL all-reduce of the bf16 gradient arena overlapped with the backward.
"""
import time
from pathlib import Path
from typing import Optional

import numpy as np
import os

import torch

from ..data import create_dataloader
from ..models.model import Model, ModelArgs
from .dp import GradSync
from .optim import FusedAdamW
from .utils import compute_loss, load_checkpoint, save_checkpoint, setup_logger


def csm_1b_args() -> ModelArgs:
    """Hyper-parameters hard-coded at reference trainer.py:100-106 / generator.py:232-238."""
    return ModelArgs(backbone_flavor="llama-1B", decoder_flavor="llama-100M", text_vocab_size=128256,
                     audio_vocab_size=2051, audio_num_codebooks=32)


class CSMTrainer:
    """PyTorch-API trainer for CSM models (reference trainer.py:26)."""

    def __init__(self, model_path: str, output_dir: str, device: str = "cuda", log_file: Optional[str] = None,
                 learning_rate: float = 1e-5, backbone_lr_multiplier: float = 0.1, decoder_lr_multiplier: float = 1.0,
                 embedding_lr_multiplier: float = 0.5, semantic_weight: float = 100.0, acoustic_weight: float = 1.0,
                 weight_decay: float = 0.01):
        self.model_path = model_path
        self.output_dir = Path(output_dir)
        self.output_dir.mkdir(parents=True, exist_ok=True)
        self.device = device
        self.logger = setup_logger("csm_trainer", log_file or str(self.output_dir / "training.log"))
        self.learning_rate = learning_rate
        self.backbone_lr_multiplier = backbone_lr_multiplier
        self.decoder_lr_multiplier = decoder_lr_multiplier
        self.embedding_lr_multiplier = embedding_lr_multiplier
        self.semantic_weight = semantic_weight
        self.acoustic_weight = acoustic_weight
        self.weight_decay = weight_decay
        self.logger.info(f"Loading model from {model_path}")
        self.model: Optional[Model] = None
        self.optimizer: Optional[FusedAdamW] = None
        self.grad_sync: Optional[GradSync] = None
        self.zero1: Optional[bool] = None          # data parallel only: None = the CSM_DP_ZERO1 switch, True / False pins it
        self._load_model()
        self.epoch = 0
        self.global_step = 0
        self.best_loss = float("inf")

    def _load_model(self):
        """Reference trainer.py:90-121: empty path -> caller sets ``.model``; ``.pt`` -> state dict of ``Model``."""
        if not self.model_path:
            self.logger.warning("Empty model path provided. Model will need to be set manually.")
            return
        self.model = Model(csm_1b_args(), device=self.device)
        sd = torch.load(self.model_path, map_location="cpu", weights_only=False)
        if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
            sd = sd["model"]
        self.model.load_state_dict(sd)
        self.model.setup_caches(4)

    def prepare_optimizer(self, freeze_backbone: bool = False, freeze_decoder: bool = False, freeze_embeddings: bool = False):
        """Reference trainer.py:123-173: four groups (backbone lr x0.1, decoder x1.0, embeddings x0.5, other x1)."""
        m = self.model
        m.trainable.update(backbone=not freeze_backbone, decoder=not freeze_decoder, embeddings=not freeze_embeddings, other=True)
        lrs = {"backbone": self.learning_rate * self.backbone_lr_multiplier,
               "decoder": self.learning_rate * self.decoder_lr_multiplier,
               "embeddings": self.learning_rate * self.embedding_lr_multiplier,
               "other": self.learning_rate}
        if GradSync.active():
            # replicas start from rank 0's parameters (do not rely on equal seeds / files), BEFORE the optimiser takes its
            # fp32 master copy of them
            GradSync.broadcast_parameters(m)
        if GradSync.active() and (self.zero1 if self.zero1 is not None else os.environ.get("CSM_DP_ZERO1", "0") == "1"):
            # ZeRO-1 (training/dp.py, training/zero.py): reduce-scatter -> AdamW on this rank's shards -> parameter all-gather
            from .zero import ZeroAdamW
            self.grad_sync = GradSync.for_model(m, zero1=True)
            self.grad_sync.on_skip.append(lambda step: self.optimizer.uncount_step())
            self.optimizer = ZeroAdamW(m, lrs, self.grad_sync, weight_decay=self.weight_decay)
            self.logger.info(f"ZeRO-1: this rank updates {self.optimizer.num_owned():,} of {self.optimizer.num_trainable():,} parameters")
        else:
            self.optimizer = FusedAdamW(m, lrs, weight_decay=self.weight_decay)
        total = sum(p.numel() for n, p in m.named_parameters()
                    if m.trainable["backbone" if "backbone" in n else "decoder" if "decoder" in n else
                                   "embeddings" if "embeddings" in n else "other"])
        self.logger.info(f"Training with {total:,} trainable parameters")
        self._ensure_grad_sync()

    def _ensure_grad_sync(self):
        """(Re-)attach the gradient exchange: ``train`` closes it when it returns (GradSync.close), a later ``train`` call or a
        direct ``train_step`` after ``prepare_optimizer`` gets a fresh one."""
        if self.grad_sync is None and GradSync.active():
            if getattr(self.optimizer, "sharded", False):
                self.grad_sync = self.optimizer.sync          # the shard plan lives in it: re-attach, never re-plan
                self.grad_sync.attach()
                return
            self.grad_sync = GradSync.for_model(self.model, zero1=False)
            self.grad_sync.on_skip.append(lambda step: self.optimizer.uncount_step())

    def train_step(self, batch, accumulation_steps: int = 1, is_boundary: bool = True, max_grad_norm: float = 1.0):
        """One micro-batch: loss -> backward (-> on the boundary micro-batch: all-reduce, clip, AdamW, zero_grad)."""
        m = self.model
        if self.grad_sync is not None:
            self.grad_sync.note_batch(batch["input_tokens"], batch["input_masks"])
            self.grad_sync.arm(is_boundary)
        loss, details = compute_loss(m, batch["input_tokens"], batch["input_masks"], batch["target_audio_tokens"],
                                     self.semantic_weight, self.acoustic_weight)
        scale = 1.0 / accumulation_steps
        if self.grad_sync is not None:
            scale /= self.grad_sync.world_size
        m.engine.backward(scale)
        if is_boundary:
            if self.grad_sync is not None:
                self.grad_sync.finish()
            if max_grad_norm and max_grad_norm > 0:
                self.optimizer.clip_grad_norm(max_grad_norm)
            # (data parallel: a step whose text-row exchange ran over capacity is dropped on the device, on every rank)
            self.optimizer.step(zero_grad=True if os.environ.get("CSM_EAGER_ZERO_GRAD") else "lazy",
                                skip=self.grad_sync.skip_flag() if self.grad_sync is not None else None)
        return loss.detach(), details

    def train(self, train_dataset, val_dataset=None, batch_size: int = 2, accumulation_steps: int = 4, epochs: int = 5,
              val_every: int = 100, save_every: int = 500, max_grad_norm: float = 1.0, resume_from: Optional[str] = None):
        """Reference trainer.py:175-357."""
        nw = getattr(self, "num_workers", 2)
        # ignore_padding (not in the reference): pad targets with -100 so padded frames leave the loss
        pad = 0
        if getattr(self, "ignore_padding", False):
            from ..data.training_data import IGNORE_INDEX
            pad = self.model.target_ignore_index = IGNORE_INDEX
        sampler = None
        rank0 = (not GradSync.active()) or torch.distributed.get_rank() == 0
        if GradSync.active():
            # data parallel: every rank walks its own shard of the dataset (new capability, see training/dp.py)
            from functools import partial
            from torch.utils.data import DataLoader
            from torch.utils.data.distributed import DistributedSampler
            from ..data import collate_variable_length
            sampler = DistributedSampler(train_dataset, shuffle=True, drop_last=True)
            train_loader = DataLoader(train_dataset, batch_size=batch_size, sampler=sampler, num_workers=nw,
                                      collate_fn=partial(collate_variable_length, target_pad=pad), pin_memory=True, drop_last=True)
        else:
            train_loader = create_dataloader(train_dataset, batch_size=batch_size, shuffle=True, num_workers=nw, target_pad=pad)
        val_loader = (create_dataloader(val_dataset, batch_size=batch_size, shuffle=False, num_workers=nw, target_pad=pad)
                      if val_dataset else None)
        if self.optimizer is None:
            self.prepare_optimizer()
        self._ensure_grad_sync()
        if resume_from:
            self.logger.info(f"Resuming from checkpoint: {resume_from}")
            meta = load_checkpoint(resume_from, self.model, self.optimizer, self.device)
            self.epoch, self.global_step, self.best_loss = meta["epoch"], meta["global_step"], meta["loss"]
            GradSync.assert_replicas_equal(self.model, what=f"parameters after resuming from {resume_from}")
        self.logger.info("Starting training")
        self.model.train()
        avg_loss = float("nan")
        def save(epoch_, loss_, name="checkpoint"):
            # replicas are bit-identical: one writer (rank 0), the others wait so that nobody races ahead into a resume
            if self.grad_sync is not None:
                self.grad_sync.check_overflow(block=True)    # every dropped step is known (and un-counted) before state is written
            # (a sharded optimiser's state_dict is a collective: every rank calls it, rank 0 receives the tensors)
            opt_sd = self.optimizer.state_dict() if getattr(self.optimizer, "sharded", False) else None
            if rank0:
                save_checkpoint(self.model, self.optimizer, epoch_, self.global_step, loss_, str(self.output_dir), name,
                                optimizer_state=opt_sd)
            if GradSync.active():
                torch.distributed.barrier()

        for epoch in range(self.epoch, self.epoch + epochs):
            t0 = time.time()
            losses = []
            if sampler is not None:
                sampler.set_epoch(epoch)                 # a different shuffle every epoch
            for batch_idx, batch in enumerate(train_loader):
                boundary = (batch_idx + 1) % accumulation_steps == 0
                loss, _ = self.train_step(batch, accumulation_steps, boundary, max_grad_norm)
                losses.append(loss)
                if not boundary:
                    continue
                self.global_step += 1
                if val_loader is not None and self.global_step % val_every == 0:
                    val_loss = GradSync.mean_scalar(self._validate(val_loader))      # every rank scores the same set: mean == each
                    if rank0:
                        self.logger.info(f"Epoch {epoch + 1}, Step {self.global_step}, Val Loss: {val_loss:.6f}")
                    if val_loss < self.best_loss:
                        self.best_loss = val_loss
                        save(epoch + 1, val_loss, "best")
                if self.global_step % save_every == 0:
                    recent = GradSync.mean_scalar(float(torch.stack(losses[-accumulation_steps:]).mean()))
                    save(epoch + 1, recent)
            avg_loss = float(torch.stack(losses).mean()) if losses else float("nan")   # one host sync per epoch
            avg_loss = GradSync.mean_scalar(avg_loss)                                    # over the ranks' shards
            if rank0:
                self.logger.info(f"Epoch {epoch + 1} completed in {time.time() - t0:.2f}s, Avg Loss: {avg_loss:.6f}")
            save(epoch + 1, avg_loss, f"epoch_{epoch + 1}")
            self.epoch = epoch + 1
        if rank0:
            self.logger.info("Training completed")
        save(self.epoch, avg_loss, "final")
        if self.grad_sync is not None and self.grad_sync.skipped_steps and rank0:
            self.logger.warning(f"{len(self.grad_sync.skipped_steps)} optimiser step(s) were dropped (text-row exchange over capacity): "
                                f"{self.grad_sync.skipped_steps}")
        if self.grad_sync is not None:
            # detach from the engine and give the process-global GEMM schedule switch back: a later single-GPU model in this
            # process must not inherit the data-parallel setting.  A second train() call re-creates it.
            self.grad_sync.close()
            self.grad_sync = None
        return self.best_loss

    def _validate(self, val_loader) -> float:
        """Reference trainer.py:359-394."""
        self.model.eval()
        total, n = 0.0, 0
        with torch.no_grad():
            for batch in val_loader:
                loss, _ = compute_loss(self.model, batch["input_tokens"], batch["input_masks"], batch["target_audio_tokens"],
                                       self.semantic_weight, self.acoustic_weight)
                total += float(loss)
                n += 1
        self.model.train()
        return total / max(1, n)

    def generate_sample(self, text: str, speaker_id: int = 0, output_path: Optional[str] = None) -> str:
        """Reference trainer.py:396-434; needs the text tokenizer and Mimi weights, which this build cannot fetch."""
        from ..generator import Generator
        generator = Generator(self.model)
        audio = generator.generate(text=text, speaker=speaker_id, context=[])
        output_path = output_path or str(self.output_dir / f"sample_step_{self.global_step}.wav")
        generator.save_wav(output_path, audio)
        return output_path
