"""Two eager (no HIP graph) eval forwards of the bench workload: the target of the rocprofv3 --pmc passes, which
do not get along with graph replay.  usage: python3 tools/eager_forward.py [batch] [passes] [f16x3|f16|f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from detectron2_centernet_amd.modeling.meta_arch.centernet import _EvalEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
model, cfg = bench.build_model(sys.argv[3] if len(sys.argv) > 3 else "f16x3", dev)
model.eval()
images = bench.synthetic_images(B, 512, 0, dev)
eng = _EvalEngine(model, B, 512, 512, 512, 512, images.dtype, use_graph=False)
eng.images.copy_(images)
with torch.no_grad():
    for _ in range(passes):
        eng()
torch.cuda.synchronize()
print("done", flush=True)
