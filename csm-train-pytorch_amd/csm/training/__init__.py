"""Training entry points (reference ``src/csm/training/__init__.py`` exports the same names)."""
from .utils import compute_loss, load_checkpoint, save_checkpoint, setup_logger  # noqa: F401
from .trainer import CSMTrainer  # noqa: F401
from .lora_trainer import CSMLoRATrainer  # noqa: F401
