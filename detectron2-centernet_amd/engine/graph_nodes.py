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


def inspect(graph, what):
    """node types of a torch.cuda.CUDAGraph captured with keep_graph=True, with the warning about anything that is not a kernel;
    a failure of the inspection itself (an API that is not there) is logged and does not touch the capture: returns None"""
    import logging
    log = logging.getLogger(__name__)
    try:
        nodes = node_types(graph.raw_cuda_graph())
    except Exception as e:   # noqa: BLE001 -- the look at the graph is a check, not a step of the capture
        log.warning("could not list the nodes of the captured %s: %r", what, e)
        return None
    other = {k: v for k, v in nodes.items() if k not in ("kernel", "empty")}
    if other:
        log.warning("the captured %s holds non-kernel nodes %s: replace the hipMemsetAsync / contiguous copy_ / large torch "
                    "reduction behind them by kernels", what, other)
    return nodes
