"""Deterministic, name-keyed weights shared by make_golden.py and the tests (so 78 MB of DLA-34 weights never
need to be stored): every state-dict entry is generated from crc32(key) + seed."""
import zlib

import torch


def fill_state_dict(sd, seed):
    """deterministic, name-keyed weights so the test side can rebuild them without storing 78 MB."""
    out = {}
    for k in sorted(sd.keys()):
        v = sd[k]
        g = torch.Generator().manual_seed((zlib.crc32(k.encode()) + seed) % (2 ** 31))
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            out[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif k.endswith("running_mean"):
            out[k] = torch.randn(v.shape, generator=g) * 0.1
        elif "/norm.weight" in k:
            out[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75      # VoVNet's FrozenBN scales (keys carry a '/')
        elif "actf.0.weight" in k or ".bn" in k and k.endswith("weight") or k.endswith(".1.weight") and v.dim() == 1:
            out[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif v.dim() == 1:
            out[k] = torch.randn(v.shape, generator=g) * 0.1
        elif "up_" in k and v.dim() == 4 and v.shape[1] == 1:
            out[k] = v.clone()  # keep the bilinear fill_up_weights initialisation (it is part of what is pinned)
        else:
            fan_in = v[0].numel()
            scale = 0.5 if "conv_offset_mask" in k else 1.4
            out[k] = torch.randn(v.shape, generator=g) * (scale / fan_in ** 0.5)
    return out


