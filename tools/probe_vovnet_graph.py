"""Probe (round 4): what is in the captured VoVNet training step that makes hipGraphInstantiate crash?
The step is captured with keep_graph=True (no instantiation), its nodes are listed through the HIP graph API (type; grid /
block of kernel nodes; extents of memset / memcpy nodes), anomalies are printed, and only then is instantiation tried."""
import ctypes as C
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
sys.path.insert(0, os.path.join(root, "tests", "golden"))


class Dim3(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("z", C.c_uint32)]


class KernelNodeParams(C.Structure):
    _fields_ = [("blockDim", Dim3), ("extra", C.c_void_p), ("func", C.c_void_p), ("gridDim", Dim3), ("kernelParams", C.c_void_p),
                ("sharedMemBytes", C.c_uint32)]


class MemsetParams(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("elementSize", C.c_uint32), ("height", C.c_size_t), ("pitch", C.c_size_t),
                ("value", C.c_uint32), ("width", C.c_size_t)]


def list_nodes(raw_graph):
    hip = C.CDLL("libamdhip64.so")
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(C.c_void_p(raw_graph), None, C.byref(n)) == 0
    nodes = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(C.c_void_p(raw_graph), nodes, C.byref(n)) == 0
    names = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "waitEvent", 7: "eventRecord",
             8: "extSemSignal", 9: "extSemWait", 10: "memAlloc", 11: "memFree", 12: "memcpyFromSymbol", 13: "memcpyToSymbol"}
    counts, odd = {}, []
    for i in range(n.value):
        t = C.c_int(-1)
        hip.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(t))
        counts[names.get(t.value, t.value)] = counts.get(names.get(t.value, t.value), 0) + 1
        if t.value == 0:
            p = KernelNodeParams()
            rc = hip.hipGraphKernelNodeGetParams(C.c_void_p(nodes[i]), C.byref(p))
            g, b = p.gridDim, p.blockDim
            if rc != 0 or 0 in (g.x, g.y, g.z, b.x, b.y, b.z) or b.x * b.y * b.z > 1024 or p.func is None:
                odd.append((i, "kernel", rc, (g.x, g.y, g.z), (b.x, b.y, b.z), p.sharedMemBytes, p.func))
        elif t.value == 2:
            p = MemsetParams()
            rc = hip.hipGraphMemsetNodeGetParams(C.c_void_p(nodes[i]), C.byref(p))
            if rc != 0 or p.width == 0 or p.height == 0 or p.elementSize not in (1, 2, 4) or not p.dst:
                odd.append((i, "memset", rc, p.dst, p.elementSize, p.width, p.height, p.pitch))
        elif t.value not in (0, 1, 2, 5):
            odd.append((i, names.get(t.value, t.value)))
    return n.value, counts, odd


def main():
    from test_vovnet_gpu import BASE, VOV_YAML
    from weights import fill_state_dict
    from detectron2_centernet_amd.config import get_cfg
    from detectron2_centernet_amd.data.catalog import register_synthetic
    from detectron2_centernet_amd.engine.bench_train import synthetic_batch
    from detectron2_centernet_amd.engine.train_loop import SimpleTrainer
    from detectron2_centernet_amd.modeling import build_model
    import tempfile
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
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
    model.train()
    if "testflow" in sys.argv:
        # what tests/test_vovnet_gpu.py does before it builds the trainer: one eager step through the list-of-dicts interface
        from detectron2_centernet_amd.data.catalog import synthetic_sample
        from detectron2_centernet_amd.structures import Boxes, Instances
        inputs = []
        for i in range(2):
            smp = synthetic_sample(i, size=128, num_classes=80, max_boxes=6)
            inst = Instances((128, 128))
            inst.gt_boxes, inst.gt_classes = Boxes(smp["boxes"]), smp["classes"]
            inputs.append({"image": smp["image"], "instances": inst})
        losses = model(inputs)
        sum(losses.values()).backward()
        model.zero_grad(set_to_none=True)
        cfg.SOLVER.IMS_PER_BATCH = 2
        tr = SimpleTrainer(model, None, cfg)
        dev = torch.device("cuda:0")
        batch = synthetic_batch(2, 128, 0, dev)
        orig = torch.cuda.CUDAGraph
        torch.cuda.CUDAGraph = lambda: orig(keep_graph=True)
        for _ in range(2):
            tr.run_step_tensors(*batch)
        key = tuple((tuple(t.shape), t.dtype) for t in batch)
        g = tr._graphs[key]
        tr._capture(g, *batch, with_step=True)
        print("capture state:", "failed: " + str(g.get("failed")) if g.get("failed") else "ok", flush=True)
        graph = g["graph"]
        n, counts, odd = list_nodes(graph.raw_cuda_graph())
        print(f"captured {n} nodes: {counts}", flush=True)
        print(f"{len(odd)} nodes look wrong:", flush=True)
        for o in odd[:40]:
            print("  ", o, flush=True)
        print("instantiating ...", flush=True)
        graph.instantiate()
        print("instantiated; replaying", flush=True)
        graph.replay()
        torch.cuda.synchronize()
        print("replayed: losses", {k: float(v) for k, v in g["losses"].items()}, flush=True)
        return
    cfg.SOLVER.IMS_PER_BATCH = 2
    tr = SimpleTrainer(model, None, cfg)
    dev = torch.device("cuda:0")
    batch = synthetic_batch(2, 128, 0, dev)
    for _ in range(2):
        tr._finish_step(model.train_batch_tensor(*batch))
    torch.cuda.synchronize()
    inputs = [t.clone() for t in batch]
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(graph):
        loss_dict = model.train_batch_tensor(*inputs)
        losses = sum(loss_dict.values())
        tr.optimizer.zero_grad()
        losses.backward()
        tr.optimizer.step()
    n, counts, odd = list_nodes(graph.raw_cuda_graph())
    print(f"captured {n} nodes: {counts}", flush=True)
    print(f"{len(odd)} nodes look wrong:", flush=True)
    for o in odd[:40]:
        print("  ", o, flush=True)
    print("instantiating ...", flush=True)
    graph.instantiate()
    print("instantiated; replaying", flush=True)
    graph.replay()
    torch.cuda.synchronize()
    print("replayed: losses", {k: float(v) for k, v in loss_dict.items()}, flush=True)


main()
