"""Round 5: the two forms of the A^T B weight-gradient kernel (csrc/linear.hip) on the 27 shapes of a 3-pair stack's
training step, every ring / tile / partition configuration of the second form against the first.

    python profiles/atb_sweep.py [quick]

Each (shape, configuration): result checked against torch.mm in float64, then 10 launches captured in a hipGraph and
replayed 5 times (HIP events).  Prints one line per shape with every configuration's time, the best one, and the sums.
Configurations are process environment variables the library reads at every call (D3F_ATB_V, D3F_ATB2_*)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["D3F_ATB_SWEEP"] = "1"     # the library re-reads its D3F_ATB* tunables at every call
import torch  # noqa: E402
from d3feat_pytorch_amd import _native  # noqa: E402

# (rows R, M, N) of C [M, N] = A^T B as bench.py's per-launch table lists them for --lanes 4 --stack 3 (r04_bench.json)
SHAPES = [(114688, 32, 384), (23872, 256, 256), (6208, 512, 512), (6208, 128, 512), (6208, 512, 128), (6208, 512, 256),
          (6208, 128, 256), (6208, 256, 64), (6208, 960, 64), (23872, 64, 256), (23872, 256, 64), (23872, 960, 64),
          (23872, 256, 128), (23872, 64, 128), (23872, 128, 32), (23872, 480, 32), (114688, 32, 128), (114688, 128, 32),
          (114688, 480, 32), (114688, 128, 64), (114688, 32, 64), (114688, 16, 64), (6208, 1920, 128)]
# multiplicity of each shape in one step (the table above lists distinct shapes)
COUNT = {(6208, 128, 512): 2, (6208, 512, 128): 2, (23872, 64, 256): 2, (23872, 256, 64): 2, (23872, 960, 64): 2}

CONFIGS = [("v1", {"D3F_ATB_V": "1"}),
           ("k2s4", {"D3F_ATB_V": "2", "D3F_ATB2_KS": "2", "D3F_ATB2_S": "4"}),
           ("k2s3", {"D3F_ATB_V": "2", "D3F_ATB2_KS": "2", "D3F_ATB2_S": "3"}),
           ("k2s2", {"D3F_ATB_V": "2", "D3F_ATB2_KS": "2", "D3F_ATB2_S": "2"}),
           ("k4s2", {"D3F_ATB_V": "2", "D3F_ATB2_KS": "4", "D3F_ATB2_S": "2"}),
           ("k4s3", {"D3F_ATB_V": "2", "D3F_ATB2_KS": "4", "D3F_ATB2_S": "3"}),
           ("k2s4g2", {"D3F_ATB_V": "2", "D3F_ATB2_KS": "2", "D3F_ATB2_S": "4", "D3F_ATB2_MIN_GROUPS": "2"}),
           ("k2s4g8", {"D3F_ATB_V": "2", "D3F_ATB2_KS": "2", "D3F_ATB2_S": "4", "D3F_ATB2_MIN_GROUPS": "8"}),
           ("k2s4w1k", {"D3F_ATB_V": "2", "D3F_ATB2_KS": "2", "D3F_ATB2_S": "4", "D3F_ATB2_WGS": "1024"}),
           ("k2s4w256", {"D3F_ATB_V": "2", "D3F_ATB2_KS": "2", "D3F_ATB2_S": "4", "D3F_ATB2_WGS": "256"}),
           ("k2s4t8", {"D3F_ATB_V": "2", "D3F_ATB2_KS": "2", "D3F_ATB2_S": "4", "D3F_ATB2_TMAX": "8"}),
           ("k2s2t8", {"D3F_ATB_V": "2", "D3F_ATB2_KS": "2", "D3F_ATB2_S": "2", "D3F_ATB2_TMAX": "8"})]
KEYS = sorted({k for _, e in CONFIGS for k in e})
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    CONFIGS = CONFIGS[:3]

L = _native.lib()
dev = torch.device("cuda:0")


def set_env(env):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)


def run(shape, env):
    R, M, N = shape
    set_env(env)
    g = torch.Generator(device=dev).manual_seed(R + 7 * M + 13 * N)
    A = torch.randn(R, M, device=dev, generator=g)      # grad_out [R, Cout = M]
    B = torch.randn(R, N, device=dev, generator=g)      # x [R, Cin = N]
    C = torch.full((M, N), float("nan"), device=dev)
    nb = L.d3f_linear_grad_weight_ws_bytes(R, N, M)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)

    def fn():
        rc = L.d3f_linear_grad_weight(B.data_ptr(), A.data_ptr(), R, N, M, C.data_ptr(), ws.data_ptr(), nb,
                                      torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    fn()
    torch.cuda.synchronize()
    ref = torch.mm(A.double().t(), B.double())
    err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
    C2 = C.clone()
    fn()
    torch.cuda.synchronize()
    same = bool(torch.equal(C, C2))
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
    return e0.elapsed_time(e1) * 1e3 / 50, err, same


print("%-22s" % "R x M x N" + "".join("%10s" % n for n, _ in CONFIGS) + "   best")
sums = {n: 0.0 for n, _ in CONFIGS}
best_sum = v1_sum = 0.0
flops = 0.0
bad = []
for shape in SHAPES:
    mult = COUNT.get(shape, 1)
    row, best = [], None
    for name, env in CONFIGS:
        try:
            us, err, same = run(shape, env)
        except Exception as e:  # noqa: BLE001
            us, err, same = float("nan"), float("nan"), False
            bad.append((shape, name, repr(e)[:80]))
        if not (err < 2e-5) or not same:
            bad.append((shape, name, "relerr %.2e reproducible %s" % (err, same)))
        row.append(us)
        sums[name] += us * mult
        if us == us and (best is None or us < best[0]):
            best = (us, name)
    best_sum += best[0] * mult
    flops += 2.0 * shape[0] * shape[1] * shape[2] * mult
    print("%6d x %4d x %4d  " % shape + "".join("%10.2f" % u for u in row) + "   %s" % best[1], flush=True)
print("%-22s" % "sum over the step (us)" + "".join("%10.1f" % sums[n] for n, _ in CONFIGS) + "   %.1f" % best_sum)
print("%-22s" % "TFLOP/s" + "".join("%10.1f" % (flops / sums[n] / 1e6) for n, _ in CONFIGS) + "   %.1f" % (
    flops / best_sum / 1e6))
print("f32 MFMA peak 157.3 TFLOP/s; launches per step:", sum(COUNT.get(s, 1) for s in SHAPES))
if bad:
    print("PROBLEMS:")
    for b in bad:
        print("  ", b)
else:
    print("all results within 2e-5 of float64 and bit-reproducible")
