"""SGD with momentum over flat parameter / gradient / momentum buffers, updated by the HIP kernel `sgd_kernel`.

Hyper-parameter grouping follows detectron2/solver/build.py:93-137: norm-layer parameters use WEIGHT_DECAY_NORM,
`bias` parameters use BASE_LR*BIAS_LR_FACTOR and WEIGHT_DECAY_BIAS, everything else BASE_LR / WEIGHT_DECAY;
momentum SOLVER.MOMENTUM, no Nesterov (defaults.py).  Design for MI355X: all parameters live in ONE contiguous f32
buffer ordered by reverse registration (roughly the order backward produces gradients), so (i) ONE kernel launch updates everything (per-run
learning rate / weight decay tables in device memory), (ii) the gradient buffer is directly the RCCL all-reduce operand, cut
into large contiguous buckets (engine/reducer.py).
"""
import torch

from .. import ops

NORM_TYPES = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.SyncBatchNorm,
              torch.nn.GroupNorm, torch.nn.InstanceNorm2d, torch.nn.LayerNorm, torch.nn.LocalResponseNorm)


def param_groups(cfg, model):
    """[(param, lr_factor, weight_decay)] in module order (solver/build.py:100-131)."""
    out, memo = [], set()
    for module in model.modules():
        for key, value in module.named_parameters(recurse=False):
            if not value.requires_grad or value in memo:
                continue
            memo.add(value)
            lr_factor, wd = 1.0, cfg.SOLVER.WEIGHT_DECAY
            if isinstance(module, NORM_TYPES):
                wd = cfg.SOLVER.WEIGHT_DECAY_NORM
            elif key == "bias":
                lr_factor = cfg.SOLVER.BIAS_LR_FACTOR
                wd = cfg.SOLVER.WEIGHT_DECAY_BIAS
            out.append((value, float(lr_factor), float(wd)))
    return out


class FlatSGD:
    def __init__(self, groups, base_lr, momentum=0.9, device=None):
        """groups: [(param, lr_factor, weight_decay)].  Parameters are re-pointed into one flat buffer."""
        self.base_lr, self.momentum = float(base_lr), float(momentum)
        groups = list(reversed(groups))  # heads first: the order gradients become ready in backward
        # contiguous segments per (lr_factor, wd) would break the backward-order layout; instead keep the order and
        # record maximal runs of equal hyper-parameters (a handful for DLA-34: weights / norm+bias alternate per layer
        # type, so runs are merged by sorting *within* a bucket-sized window only if adjacent).  Simplest exact form:
        # one run per change of (lr_factor, wd).
        self.params = [g[0] for g in groups]
        device = device or self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat_param = torch.empty(total, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_mom = torch.zeros(total, dtype=torch.float32, device=device)
        self.offsets, self.runs = [], []
        off = 0
        for p, lf, wd in groups:
            n = p.numel()
            self.flat_param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + n].view_as(p.data)
            p.grad = self.flat_grad[off:off + n].view_as(p.data)
            p._ctdet_flat_grad = p.grad      # ops_train.grad_slot: backward kernels accumulate here directly
            if self.runs and self.runs[-1][2] == lf and self.runs[-1][3] == wd:
                self.runs[-1][1] = off + n
            else:
                self.runs.append([off, off + n, lf, wd])
            self.offsets.append((off, n))
            off += n
        self.lr_factors = sorted({r[2] for r in self.runs})
        # device-side tables for the one-launch update: run ends, weight decay, index into the lr table
        self._lr_table = torch.zeros(len(self.lr_factors), dtype=torch.float32, device=device)
        self._run_end = torch.tensor([r[1] for r in self.runs], dtype=torch.int64, device=device)
        self._run_wd = torch.tensor([r[3] for r in self.runs], dtype=torch.float32, device=device)
        self._run_lr_index = torch.tensor([self.lr_factors.index(r[2]) for r in self.runs], dtype=torch.int32, device=device)
        self._first = True
        self.set_lr_factor(1.0)
        # zero-initialised scratch for the atomically accumulated weight gradients (ops_train.ZeroArena): a little larger
        # than the parameter count (channel padding), cleared together with the gradient buffer
        self._arena = self._packs = None
        if torch.device(device).type == "cuda":
            from .. import ops_train
            self._arena = ops_train.ARENA = ops_train.ZeroArena(int(total * 1.25) + (1 << 20), torch.device(device))
            self._packs = ops.PACK_PLAN = ops.PackPlan(self.flat_param)

    def set_lr_factor(self, f):
        self._sched_factor = float(f)
        for i, lf in enumerate(self.lr_factors):
            self._lr_table[i:i + 1].fill_(self.base_lr * lf * self._sched_factor)

    @property
    def lr(self):
        return self.base_lr * self._sched_factor

    def zero_grad(self):
        self.flat_grad.zero_()
        if self._arena is not None:       # the weight-gradient accumulators of this backward pass: one fill for all of them
            from .. import ops_train
            ops_train.PENDING.clear()     # leftovers of a backward pass that raised ...
            ops_train._END_QUEUED[0] = False   # ... whose end-of-backward callback autograd dropped with it: without this
                                               # reset no later pass would queue the flush of its weight gradients again
            self._arena.begin_step()
        for p, (off, n) in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = p._ctdet_flat_grad = self.flat_grad[off:off + n].view_as(p.data)

    def step(self):
        ops.sgd_momentum_runs_(self.flat_param, self.flat_grad, self.flat_mom, self._run_end, self._run_lr_index,
                               self._run_wd, self._lr_table, self.momentum, self._first)
        self._first = False
        for p in self.params:  # raw-pointer update: tell autograd / the packed-weight caches the values changed
            torch.autograd.graph.increment_version(p)
        if self._packs is not None:   # the f16 operands of every conv weight the last step used, in one launch
            self._packs.run()

    def state_dict(self):
        return {"momentum": self.flat_mom.clone(), "first": self._first}

    def load_state_dict(self, sd):
        """this optimizer's own state, or the `torch.optim.SGD` state dict a reference checkpoint carries under "optimizer"
        ({"state": {index: {"momentum_buffer": t}}, "param_groups": [{"params": [indices], ...}]}): the reference builds one
        group per parameter in module order (solver/build.py:100-137), the order of `param_groups` here, so buffer i
        belongs to parameter i of that order"""
        if "momentum" in sd:
            self.flat_mom.copy_(sd["momentum"])
            self._first = sd["first"]
            return
        if "state" not in sd or "param_groups" not in sd:
            raise KeyError("optimizer state: neither FlatSGD's {'momentum', 'first'} nor a torch.optim.SGD state dict")
        order = [i for g in sd["param_groups"] for i in g["params"]]
        if len(order) != len(self.params):
            raise ValueError(f"optimizer state holds {len(order)} parameters, the model has {len(self.params)} trainable ones")
        # self.params is the reversed module order
        loaded = 0
        for k, idx in enumerate(order):
            st = sd["state"].get(idx, {})
            buf = st.get("momentum_buffer")
            off, n = self.offsets[len(order) - 1 - k]
            if buf is None:
                self.flat_mom[off:off + n].zero_()
                continue
            if buf.numel() != n:
                raise ValueError(f"momentum buffer {idx}: {buf.numel()} elements, parameter has {n}")
            self.flat_mom[off:off + n].copy_(buf.reshape(-1).to(self.flat_mom.device, torch.float32))
            loaded += 1
        self._first = loaded == 0


def build_optimizer(cfg, model):
    """solver/build.py:93-137.  Options of the reference's builder that this optimizer does not implement are refused
    instead of being silently dropped."""
    if cfg.SOLVER.get("NESTEROV", False):
        raise NotImplementedError("SOLVER.NESTEROV: the fused SGD kernel implements plain momentum")
    clip = cfg.SOLVER.get("CLIP_GRADIENTS", None)
    if clip is not None and clip.get("ENABLED", False):
        raise NotImplementedError("SOLVER.CLIP_GRADIENTS: gradient clipping is not implemented")
    return FlatSGD(param_groups(cfg, model), cfg.SOLVER.BASE_LR, cfg.SOLVER.MOMENTUM)
