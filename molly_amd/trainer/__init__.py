from .zero2 import Zero2Optimizer, linear_warmup_lr  # noqa: F401
