"""Node types of a captured HIP graph, read through the HIP graph API (ctypes on libamdhip64; no torch types involved).

Why the trainer looks: a replay of the captured training step that was launched on an IDLE stream ran the step's one
hipMemsetAsync node (the clear of the heat-map target) out of order with the kernel after it -- two ranks sharing a GPU, the
data-parallel step whose SGD launch is eager.  The step is therefore built from kernels only, and SimpleTrainer reports any
other node it finds after a capture (profiles/r04_graph_memset_node.txt has the measurements).
"""
import ctypes as C

_NAMES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "waitEvent", 7: "eventRecord",
          8: "extSemSignal", 9: "extSemWait", 10: "memAlloc", 11: "memFree", 12: "memcpyFromSymbol", 13: "memcpyToSymbol"}
_hip = None


def node_types(raw_graph):
    """{type name: count} of the nodes of hipGraph_t `raw_graph` (an int, e.g. torch.cuda.CUDAGraph.raw_cuda_graph())"""
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipGraphGetNodes.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
        _hip.hipGraphNodeGetType.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    n = C.c_size_t(0)
    rc = _hip.hipGraphGetNodes(C.c_void_p(raw_graph), None, C.byref(n))
    if rc != 0:
        raise RuntimeError(f"hipGraphGetNodes: error {rc}")
    nodes = (C.c_void_p * max(n.value, 1))()
    rc = _hip.hipGraphGetNodes(C.c_void_p(raw_graph), nodes, C.byref(n))
    if rc != 0:
        raise RuntimeError(f"hipGraphGetNodes: error {rc}")
    out = {}
    for i in range(n.value):
        t = C.c_int(-1)
        rc = _hip.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(t))
        if rc != 0:
            raise RuntimeError(f"hipGraphNodeGetType: error {rc}")
        name = _NAMES.get(t.value, str(t.value))
        out[name] = out.get(name, 0) + 1
    return out
