from .zero2 import Zero2Optimizer, linear_warmup_lr  # noqa: F401
from .loop import TrainArgs, Trainer, save_model  # noqa: F401
