"""Round 5: which side bounds the second form of the A^T B kernel?  D3F_ATB2_DBG=1 runs it without its loads (MFMA +
LDS reads alone), =2 without its MFMAs (LDS-DMA alone); results are garbage in both, only the time counts.
    python profiles/atb_diag.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["D3F_ATB_SWEEP"] = "1"     # the library re-reads its D3F_ATB* tunables at every call
import torch  # noqa: E402
from d3feat_pytorch_amd import _native  # noqa: E402

L = _native.lib()
dev = torch.device("cuda:0")
SHAPES = [(6208, 512, 512), (23872, 256, 256), (23872, 960, 64), (114688, 480, 32), (114688, 128, 64), (6208, 128, 256)]
CONFIGS = [("k4s3", {"D3F_ATB2_KS": "4", "D3F_ATB2_S": "3"}), ("k4s2", {"D3F_ATB2_KS": "4", "D3F_ATB2_S": "2"}),
           ("k2s4", {"D3F_ATB2_KS": "2", "D3F_ATB2_S": "4"}),
           ("k2s4w1024", {"D3F_ATB2_KS": "2", "D3F_ATB2_S": "4", "D3F_ATB2_WGS": "1024"}),
           ("k4s2t8", {"D3F_ATB2_KS": "4", "D3F_ATB2_S": "2", "D3F_ATB2_TMAX": "8"}),
           ("k2s2t8", {"D3F_ATB2_KS": "2", "D3F_ATB2_S": "2", "D3F_ATB2_TMAX": "8"}),
           ("k2s2t8w512", {"D3F_ATB2_KS": "2", "D3F_ATB2_S": "2", "D3F_ATB2_TMAX": "8", "D3F_ATB2_WGS": "512"})]
KEYS = ["D3F_ATB2_KS", "D3F_ATB2_S", "D3F_ATB2_WGS", "D3F_ATB2_TMAX", "D3F_ATB2_DBG"]


def timed(R, M, N):
    A = torch.randn(R, M, device=dev)
    B = torch.randn(R, N, device=dev)
    C = torch.empty(M, N, device=dev)
    nb = L.d3f_linear_grad_weight_ws_bytes(R, N, M)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)

    def fn():
        L.d3f_linear_grad_weight(B.data_ptr(), A.data_ptr(), R, N, M, C.data_ptr(), ws.data_ptr(), nb,
                                 torch.cuda.current_stream().cuda_stream)
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 50


os.environ["D3F_ATB_V"] = "2"
print("%-22s %-12s %9s %9s %9s   (us: full | no loads | no MFMAs); MFMA floor = flops / 157.3 TF" % ("shape", "config", "full", "noload", "nomfma"))
for R, M, N in SHAPES:
    floor = 2.0 * R * M * N / 157.3e6
    for name, env in CONFIGS:
        row = []
        for dbg in ("0", "1", "2"):
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            os.environ["D3F_ATB2_DBG"] = dbg
            row.append(timed(R, M, N))
        print("%6d x %4d x %4d   %-12s %9.2f %9.2f %9.2f   floor %.1f" % (R, M, N, name, row[0], row[1], row[2], floor), flush=True)
