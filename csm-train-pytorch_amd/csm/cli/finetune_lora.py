"""``csm-finetune-lora`` on MI355X: flag names and defaults of reference ``src/csm/cli/finetune_lora.py:32-237``."""
import argparse
import logging

from ..training.dp import init_distributed
from ..training.lora_trainer import CSMLoRATrainer
from .common import add_data_args, load_datasets


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Fine-tune CSM with LoRA (MI355X)")
    p.add_argument("--model-path", type=str, required=True, help="checkpoint (.pt or .safetensors); '' = random init")
    p.add_argument("--output-dir", type=str, default="csm_lora")
    add_data_args(p, context_turns=2)   # reference default of csm-finetune-lora
    lo = p.add_argument_group("LoRA")
    lo.add_argument("--lora-r", type=int, default=8)
    lo.add_argument("--lora-alpha", type=float, default=16.0)
    lo.add_argument("--lora-dropout", type=float, default=0.0)
    lo.add_argument("--target-modules", type=str, nargs="+", default=["q_proj", "v_proj"])
    lo.add_argument("--target-layers", type=int, nargs="+", default=None)
    lo.add_argument("--lora-bias", action="store_true")
    t = p.add_argument_group("Training")
    t.add_argument("--learning-rate", type=float, default=1e-4)
    t.add_argument("--semantic-weight", type=float, default=100.0)
    t.add_argument("--acoustic-weight", type=float, default=1.0)
    t.add_argument("--weight-decay", type=float, default=0.01)
    t.add_argument("--batch-size", type=int, default=2)
    t.add_argument("--epochs", type=int, default=5)
    t.add_argument("--val-every", type=int, default=100)
    t.add_argument("--save-every", type=int, default=500)
    t.add_argument("--max-grad-norm", type=float, default=1.0)
    t.add_argument("--resume-from", type=str, default=None)
    t.add_argument("--save-mode", choices=["lora", "full", "both"], default="lora")
    t.add_argument("--acoustic-mode", choices=["off", "all", "amortized"], default="off")
    p.add_argument("--log-level", type=str, default="info")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--generate-samples", action="store_true")
    p.add_argument("--sample-prompt", type=str, default="Hello, this is a test of the fine-tuned voice.")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    rank, world, local = init_distributed()
    model = None
    if not args.model_path:
        from ..models.model import Model
        from ..training.trainer import csm_1b_args
        model = Model(csm_1b_args(), device=f"cuda:{local}", seed=0)
    trainer = CSMLoRATrainer(model_path=args.model_path, output_dir=args.output_dir, learning_rate=args.learning_rate,
                             semantic_weight=args.semantic_weight, acoustic_weight=args.acoustic_weight,
                             weight_decay=args.weight_decay, lora_r=args.lora_r, lora_alpha=args.lora_alpha,
                             lora_dropout=args.lora_dropout, target_modules=args.target_modules,
                             target_layers=args.target_layers, lora_use_bias=args.lora_bias, device=f"cuda:{local}", model=model)
    trainer.logger.setLevel(logging.DEBUG if args.debug else getattr(logging, args.log_level.upper(), logging.INFO))
    trainer.model.acoustic_mode = args.acoustic_mode
    train_ds, val_ds = load_datasets(args)
    trainer.prepare_optimizer()
    best = trainer.train(train_ds, val_ds, batch_size=args.batch_size, epochs=args.epochs, val_every=args.val_every,
                         save_every=args.save_every, max_grad_norm=args.max_grad_norm, resume_from=args.resume_from)
    if rank == 0:
        trainer.save_model(f"{args.output_dir}/final", args.save_mode)
        trainer.logger.info(f"best validation loss: {best}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
