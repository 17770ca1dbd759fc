from .train_loop import SimpleTrainer

__all__ = ["SimpleTrainer"]
