"""detectron2/layers/batch_norm.py:13-131 (`FrozenBatchNorm2d`, `get_norm`) and the `Conv2d` wrapper of
detectron2/layers/wrappers.py:37-101 -- parameter containers with the reference's names (`conv1.weight`,
`conv1.norm.{weight,bias,running_mean,running_var}`), so reference ResNet checkpoints load key-for-key.  The hot
path never calls these modules' forward: `hipnn.conv_module` folds the norm into the conv kernel's epilogue."""
import torch
from torch import nn


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters, all buffers (batch_norm.py:13-99):
    y = x * weight / sqrt(running_var + eps) + (bias - running_mean * weight / sqrt(running_var + eps))."""

    _version = 3
    frozen = True

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        bias = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1).to(x.dtype) + bias.reshape(1, -1, 1, 1).to(x.dtype)

    def __repr__(self):
        return f"FrozenBatchNorm2d(num_features={self.num_features}, eps={self.eps})"


def get_norm(norm, out_channels):
    """batch_norm.py:102-131 for the norms the CenterNet configs use."""
    if norm is None or norm == "":
        return None
    if isinstance(norm, str):
        norm = {"BN": nn.BatchNorm2d, "FrozenBN": FrozenBatchNorm2d}[norm]
    return norm(out_channels)


class Conv2d(nn.Conv2d):
    """nn.Conv2d carrying an optional `norm` sub-module and `activation` (wrappers.py:37-101)."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation
