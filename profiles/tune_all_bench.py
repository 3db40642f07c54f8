"""A TunableOp table tuned FROM SCRATCH over the library-GEMM shapes of the current bench.py steps (every shape tuned
while the graphs are captured), e.g. with hipBLASLt candidates in the race:
    PYTORCH_TUNABLEOP_HIPBLASLT_ENABLED=1 python profiles/tune_all_bench.py gpurun_out/table_lt.csv [bench.py arguments]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
out = sys.argv[1]
sys.argv = [os.path.join(REPO, "bench.py"), "--no-tuned-gemm"] + (sys.argv[2:] or ["--quick", "--steps", "10"])
import tempfile  # noqa: E402
import torch  # noqa: E402
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(False)
torch.cuda.tunable.set_filename(os.path.join(tempfile.gettempdir(), "d3f_tunableop_all_%d.csv" % os.getpid()))
import bench  # noqa: E402

try:
    bench.main()
except SystemExit:
    pass
res = torch.cuda.tunable.get_results()
with open(out, "w") as fh:
    for k, v in torch.cuda.tunable.get_validators():
        fh.write("Validator,%s,%s\n" % (k, v))
    for r in res:
        fh.write("%s,%s,%s,%s\n" % (r[0], r[1], r[2], r[3]))
print("wrote", out, len(res), "entries", file=sys.stderr)
