"""Command-line front ends (``python -m csm.cli.train`` / ``python -m csm.cli.finetune_lora``)."""
