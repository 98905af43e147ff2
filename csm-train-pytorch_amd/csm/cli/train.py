"""``csm-train`` on MI355X: flag names and defaults of reference ``src/csm/cli/train.py:23-225``; single GPU or, under
``torchrun --nproc-per-node N``, data parallel over N GPUs (RCCL)."""
import argparse
import logging

from ..training.dp import init_distributed
from ..training.trainer import CSMTrainer
from .common import add_data_args, load_datasets


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train CSM model (MI355X)")
    p.add_argument("--model-path", type=str, required=True, help="Path to the CSM model checkpoint ('' = random init)")
    p.add_argument("--output-dir", type=str, default="csm_trained")
    add_data_args(p)
    t = p.add_argument_group("Training")
    t.add_argument("--learning-rate", type=float, default=1e-5)
    t.add_argument("--backbone-lr-multiplier", type=float, default=0.1)
    t.add_argument("--decoder-lr-multiplier", type=float, default=1.0)
    t.add_argument("--embedding-lr-multiplier", type=float, default=0.5)
    t.add_argument("--epochs", type=int, default=5)
    t.add_argument("--batch-size", type=int, default=2)
    t.add_argument("--accumulation-steps", type=int, default=4)
    t.add_argument("--semantic-weight", type=float, default=100.0)
    t.add_argument("--acoustic-weight", type=float, default=1.0)
    t.add_argument("--weight-decay", type=float, default=0.01)
    t.add_argument("--max-grad-norm", type=float, default=1.0)
    t.add_argument("--freeze-backbone", action="store_true")
    t.add_argument("--freeze-decoder", action="store_true")
    t.add_argument("--freeze-embeddings", action="store_true")
    t.add_argument("--resume-from", type=str, default=None)
    t.add_argument("--acoustic-mode", choices=["off", "all", "amortized"], default="off",
                   help="depth-decoder loss: off = the reference's placeholder, all / amortized (1/16 of frames)")
    m = p.add_argument_group("Misc")
    m.add_argument("--device", type=str, default="cuda")
    m.add_argument("--num-workers", type=int, default=2)
    m.add_argument("--save-every", type=int, default=500)
    m.add_argument("--val-every", type=int, default=100)
    m.add_argument("--log-file", type=str, default=None)
    m.add_argument("--debug", action="store_true")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    rank, world, local = init_distributed()
    device = f"cuda:{local}" if args.device.startswith("cuda") else args.device
    trainer = CSMTrainer(model_path=args.model_path, output_dir=args.output_dir, device=device, log_file=args.log_file,
                         learning_rate=args.learning_rate, backbone_lr_multiplier=args.backbone_lr_multiplier,
                         decoder_lr_multiplier=args.decoder_lr_multiplier, embedding_lr_multiplier=args.embedding_lr_multiplier,
                         semantic_weight=args.semantic_weight, acoustic_weight=args.acoustic_weight, weight_decay=args.weight_decay)
    if args.debug:
        trainer.logger.setLevel(logging.DEBUG)
    if trainer.model is None:
        from ..models.model import Model
        from ..training.trainer import csm_1b_args
        trainer.model = Model(csm_1b_args(), device=device, seed=0)
    trainer.model.acoustic_mode = args.acoustic_mode
    trainer.num_workers = args.num_workers
    trainer.ignore_padding = args.ignore_padding
    train_ds, val_ds = load_datasets(args)
    trainer.prepare_optimizer(freeze_backbone=args.freeze_backbone, freeze_decoder=args.freeze_decoder,
                              freeze_embeddings=args.freeze_embeddings)
    best = trainer.train(train_ds, val_ds, batch_size=args.batch_size, accumulation_steps=args.accumulation_steps, epochs=args.epochs,
                         val_every=args.val_every, save_every=args.save_every, max_grad_norm=args.max_grad_norm,
                         resume_from=args.resume_from)
    if rank == 0:
        trainer.logger.info(f"best validation loss: {best}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
