"""Round 5: can a captured graph's small memset nodes be FOUND and REPAIRED from outside?  (profiles/memset_node_repro.py:
a hipMemsetAsync of <= 4 KiB captured into a hipGraph writes 0xA0 bytes instead of its recorded value from the second
replay on.)  torch.cuda.CUDAGraph(keep_graph=True) keeps the hipGraph_t / hipGraphExec_t handles; the HIP graph API is
called through ctypes.
    python profiles/memset_node_fix_experiment.py"""
import ctypes

import torch

hip = ctypes.CDLL("libamdhip64.so")


class MemsetParams(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_void_p), ("elementSize", ctypes.c_uint), ("height", ctypes.c_size_t),
                ("pitch", ctypes.c_size_t), ("value", ctypes.c_uint), ("width", ctypes.c_size_t)]


hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
hip.hipGraphNodeGetType.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
hip.hipGraphMemsetNodeGetParams.argtypes = [ctypes.c_void_p, ctypes.POINTER(MemsetParams)]
hip.hipGraphExecMemsetNodeSetParams.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(MemsetParams)]
dev = torch.device("cuda:0")


def memset_nodes(graph):
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(graph, None, ctypes.byref(n)) == 0
    nodes = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(graph, nodes, ctypes.byref(n)) == 0
    out, kinds = [], []
    for node in nodes:
        t = ctypes.c_int(-1)
        assert hip.hipGraphNodeGetType(node, ctypes.byref(t)) == 0
        kinds.append(t.value)
        if t.value == 2:      # hipGraphNodeTypeMemset
            p = MemsetParams()
            assert hip.hipGraphMemsetNodeGetParams(node, ctypes.byref(p)) == 0
            out.append((node, p))
    return out, kinds


for mode in ("plain", "reset params before every replay", "reset params once after instantiation"):
    buf = torch.full((1,), 7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        hip.hipMemsetAsync(buf.data_ptr(), 0, 4, torch.cuda.current_stream().cuda_stream)
        buf.add_(1)
    g.instantiate()
    found, kinds = memset_nodes(g.raw_cuda_graph())
    ex = g.raw_cuda_graph_exec()
    if mode == "plain":
        print("node types of the captured graph:", kinds, "memset params (value, elementSize, width, height):",
              [(p.value, p.elementSize, p.width, p.height) for _, p in found], flush=True)
    if mode.startswith("reset params once"):
        for node, p in found:
            print("  set rc", hip.hipGraphExecMemsetNodeSetParams(ex, node, ctypes.byref(p)))
    seen = []
    for k in range(4):
        if mode.startswith("reset params before"):
            for node, p in found:
                rc = hip.hipGraphExecMemsetNodeSetParams(ex, node, ctypes.byref(p))
                assert rc == 0, rc
        g.replay()
        torch.cuda.synchronize()
        seen.append(int(buf.item()))
    print("%-40s buffer after replays 1..4: %s" % (mode, seen), flush=True)
