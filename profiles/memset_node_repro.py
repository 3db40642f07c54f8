"""Round 5: is a hipMemsetAsync captured into a hipGraph re-executed on every replay?  (csrc/common.hpp's zero_async is a
kernel because round 2 saw dirty allocators on the second replay; library kernels that clear a semaphore / split-K
workspace with hipMemsetAsync before they run depend on the same thing.)  No d3feat code involved.
    python profiles/memset_node_repro.py"""
import ctypes

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
dev = torch.device("cuda:0")
for nbytes, label in ((4, "one word"), (4096, "4 KiB"), (1 << 20, "1 MiB")):
    buf = torch.full((nbytes // 4,), 7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        buf.add_(1)                      # a kernel node behind the memset node
    assert rc == 0, rc
    seen = []
    for k in range(4):
        g.replay()
        torch.cuda.synchronize()
        seen.append((int(buf.min()), int(buf.max())))
    ok = all(s == (1, 1) for s in seen)
    print("memset node, %-8s: buffer after replays 1..4 (min, max) = %s -> %s" % (
        label, seen, "re-executed on every replay" if ok else "NOT re-executed as recorded"), flush=True)
# the same with two graphs replayed concurrently on two streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bufs = [torch.full((1024,), 7, dtype=torch.int32, device=dev) for _ in range(2)]
graphs = []
for b in bufs:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        hip.hipMemsetAsync(b.data_ptr(), 0, 4096, torch.cuda.current_stream().cuda_stream)
        for _ in range(50):
            b.add_(1)
    graphs.append(g)
bad = 0
for k in range(100):
    with torch.cuda.stream(s1):
        graphs[0].replay()
    with torch.cuda.stream(s2):
        graphs[1].replay()
    torch.cuda.synchronize()
    bad += int(not all(int(b.min()) == 50 == int(b.max()) for b in bufs))
print("two graphs with memset nodes replayed concurrently 100 x: %d bad replays" % bad, flush=True)
