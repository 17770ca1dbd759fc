"""Probe (round 4): which kernels of the training step give different results when foreign waves share the GPU?

One eager training step (forward + backward, no optimizer update: the parameters stay fixed) is repeated; every op wrapper's
outputs are reduced to checksums on the device, in launch order.  The first run (alone) is the reference; then a second
process runs training steps on the same GPU and the traces are compared entry by entry.  Outputs that end in f32 atomics
(weight gradients, the DCNv2 scatter) are compared with a tolerance, everything else bit for bit."""
import os
import subprocess
import sys
import time

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
import bench
from detectron2_centernet_amd import ops, ops_train as ot
from detectron2_centernet_amd.engine.bench_train import synthetic_batch

TRACE = []
ATOMIC = ("conv_wgrad", "dcn_col2im_coord", "dwconvT_bwd")


def _sig(t):
    return tuple(t.shape)


def wrap(mod, name):
    fn = getattr(mod, name)

    def w(*a, **k):
        out = fn(*a, **k)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for i, o in enumerate(outs):
            if isinstance(o, torch.Tensor) and o.is_floating_point() and o.numel() > 0:
                d = o.detach().double()
                TRACE.append((f"{name}[{i}]", _sig(o), torch.stack([d.sum(), d.abs().sum()])))
        return out
    setattr(mod, name, w)


for m, names in ((ops, ("conv2d", "dcnv2", "dcnv2_offset", "dwconvT_add", "maxpool2x2", "focal_loss", "reg_l1_loss", "preprocess")),
                 (ot, ("conv_wgrad", "bn_train_fwd", "bn_train_bwd", "dcn_cols", "dcn_col2im_coord", "maxpool2x2_bwd", "dwconvT_bwd"))):
    for n in names:
        wrap(m, n)


def one_step(model, batch):
    TRACE.clear()
    model.zero_grad(set_to_none=True)
    losses = model.train_batch_tensor(*batch)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    return [(n, s, c.cpu()) for n, s, c in TRACE]


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dev = torch.device("cuda:0")
    model, cfg = bench.build_model(prec, dev, calibrate=False)
    model.train()
    batch = synthetic_batch(16, 512, 0, dev)
    one_step(model, batch)
    ref = one_step(model, batch)
    again = one_step(model, batch)
    child = subprocess.Popen([sys.executable, os.path.join(here, "probe_contention.py"), "hammer", str(20 + 2 * reps)])
    time.sleep(15)
    bad = {}
    for rep in range(-1, reps):
        cur = again if rep < 0 else one_step(model, batch)
        assert len(cur) == len(ref)
        first = None
        for i, ((n, s, c), (n2, s2, c2)) in enumerate(zip(ref, cur)):
            assert n == n2 and s == s2
            scale = max(1e-30, float(c[1]))
            d = float((c - c2).abs().max()) / scale
            lim = 1e-5 if any(n.startswith(a) for a in ATOMIC) else 0.0
            if d > lim:
                key = (n, s)
                bad.setdefault(key, [0, 0.0, i])
                bad[key][0] += 1
                bad[key][1] = max(bad[key][1], d)
                if first is None:
                    first = (i, n, s, d)
        print(("alone, second run" if rep < 0 else f"shared GPU, run {rep}") + f": first deviating op: {first}", flush=True)
    print(f"{prec}: ops whose outputs deviated (count of runs, worst relative checksum deviation, position in the step of {len(ref)} ops):")
    for (n, s), (cnt, d, pos) in sorted(bad.items(), key=lambda kv: kv[1][2]):
        print(f"  #{pos:4d} {n:22s} {str(s):28s} {cnt:3d} runs  {d:.3e}")
    child.wait()


main()
