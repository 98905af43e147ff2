from .mimi import MimiCodec  # noqa: F401
