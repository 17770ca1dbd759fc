"""Which launches of one eager training step are NOT kernels (hipMemcpyAsync / hipMemsetAsync)?  Captured into the step's HIP
graph these become memcpy / memset nodes; `profiles/r04_graph_memset_node.txt` records why the step should have none.

    python tools/find_copy_nodes.py [precision] [dla34|res50|vovnet]
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
    if len(sys.argv) > 2 and sys.argv[2] == "vovnet":
        import tempfile
        sys.path.insert(0, os.path.join(root, "tests"))
        sys.path.insert(0, os.path.join(root, "tests", "golden"))
        from test_vovnet_gpu import BASE, VOV_YAML
        from weights import fill_state_dict
        from detectron2_centernet_amd.config import get_cfg
        from detectron2_centernet_amd.data.catalog import register_synthetic
        from detectron2_centernet_amd.modeling import build_model
        tmp = tempfile.mkdtemp()
        open(os.path.join(tmp, "Base-CenterNet.yaml"), "w").write(BASE)
        open(os.path.join(tmp, "v.yaml"), "w").write(VOV_YAML)
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(tmp, "v.yaml"))
        cfg.MODEL.CENTERNET.HIP_PRECISION = prec
        register_synthetic("bulb_train", num_classes=80)
        model = build_model(cfg)
        sd0 = fill_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, seed=19)
        model.load_state_dict({k: v.to(model.device) for k, v in sd0.items()})
        cfg.SOLVER.IMS_PER_BATCH = 2
    else:
        model, cfg = bench.build_model(prec, dev, calibrate=False, config=sys.argv[2] if len(sys.argv) > 2 else "dla34")
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
