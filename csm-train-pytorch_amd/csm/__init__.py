"""csm - MI355X-native training / generation path for CSM-1B.

Drop-in for the hot path of imaginateit/csm-train-pytorch: ``csm.models.model``, ``csm.training.utils``,
``csm.training.trainer``, ``csm.training.lora_trainer`` and ``csm.generator`` keep the reference's names and
signatures; the arithmetic runs in hand-written gfx950 kernels (``csm/hip/libcsm_hip.so``).
"""
__version__ = "0.1.0"
