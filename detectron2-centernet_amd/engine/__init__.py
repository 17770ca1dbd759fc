from .launch import launch
from .train_loop import SimpleTrainer

__all__ = ["SimpleTrainer", "launch"]
