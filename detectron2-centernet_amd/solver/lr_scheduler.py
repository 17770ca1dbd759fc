"""WarmupMultiStepLR (detectron2/solver/lr_scheduler.py:16-49, warm-up factor :90-116)."""
from bisect import bisect_right


def warmup_multistep_factor(it, milestones, gamma, warmup_factor, warmup_iters, warmup_method="linear"):
    if it >= warmup_iters:
        w = 1.0
    elif warmup_method == "constant":
        w = warmup_factor
    elif warmup_method == "linear":
        alpha = it / warmup_iters
        w = warmup_factor * (1 - alpha) + alpha
    else:
        raise ValueError(f"Unknown warmup method: {warmup_method}")
    return w * gamma ** bisect_right(list(milestones), it)


class WarmupMultiStepLR:
    """scheduler over a FlatSGD: `step()` advances one iteration and writes the new learning rates."""

    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear",
                 last_epoch=-1):
        if not list(milestones) == sorted(milestones):
            raise ValueError(f"Milestones should be a list of increasing integers. Got {milestones}")
        self.optimizer, self.milestones, self.gamma = optimizer, list(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        self.last_epoch = last_epoch
        self.step()

    def get_factor(self):
        return warmup_multistep_factor(self.last_epoch, self.milestones, self.gamma, self.warmup_factor,
                                       self.warmup_iters, self.warmup_method)

    def step(self):
        self.last_epoch += 1
        self.optimizer.set_lr_factor(self.get_factor())

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = sd["last_epoch"]
        self.optimizer.set_lr_factor(self.get_factor())
