from .model import Model, ModelArgs, sample_topk  # noqa: F401
