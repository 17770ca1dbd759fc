"""does the eager training step leave device memory behind?  allocated bytes after every step, with and without gc.collect()"""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["CTDET_TRAIN_GRAPH"] = "0"
import bench  # noqa: E402
from detectron2_centernet_amd.engine.bench_train import synthetic_batch  # noqa: E402
from detectron2_centernet_amd.engine.train_loop import SimpleTrainer  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
config = sys.argv[2] if len(sys.argv) > 2 else "dla34"
dev = torch.device("cuda:0")
model, cfg = bench.build_model(prec, dev, calibrate=False, config=config)
model.train()
tr = SimpleTrainer(model, None, cfg)
batch = synthetic_batch(16 if config == "dla34" else 4, 512, 0, dev)
for i in range(12):
    tr.run_step_tensors(*batch)
    torch.cuda.synchronize()
    a = torch.cuda.memory_allocated() / 2**30
    n = gc.collect() if i >= 8 else -1
    b = torch.cuda.memory_allocated() / 2**30
    print(f"step {i}: allocated {a:.2f} GiB" + (f", after gc.collect() ({n} objects) {b:.2f} GiB" if n >= 0 else ""), flush=True)
