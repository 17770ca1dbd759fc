"""Which launches of one eager training step are NOT kernels (hipMemcpyAsync / hipMemsetAsync)?  Captured into the step's HIP
graph these become memcpy / memset nodes; `profiles/r04_graph_memset_node.txt` records why the step should have none.

    python tools/find_copy_nodes.py [precision]
"""
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    from detectron2_centernet_amd.engine.bench_train import synthetic_batch
    from detectron2_centernet_amd.engine.train_loop import SimpleTrainer
    dev = torch.device("cuda:0")
    os.environ["CTDET_TRAIN_GRAPH"] = "0"
    model, cfg = bench.build_model(prec, dev, calibrate=False)
    model.train()
    trainer = SimpleTrainer(model, None, cfg)
    batch = synthetic_batch(2, 128, 0, dev)
    for _ in range(2):
        trainer.run_step_tensors(*batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        trainer.run_step_tensors(*batch)
        torch.cuda.synchronize()
    n = 0
    for ev in prof.events():
        name = ev.name
        if name in ("hipMemcpyAsync", "hipMemsetAsync", "hipMemcpyWithStream", "hipMemset"):
            n += 1
            chain, p = [], ev.cpu_parent
            while p is not None:
                chain.append(f"{p.name}{p.input_shapes if p.input_shapes else ''}")
                p = p.cpu_parent
            print(name, "<-", " <- ".join(chain))
    print(n, "non-kernel launches")


main()
