"""The shipped TunableOp table extended by the library-GEMM shapes of the CURRENT bench.py steps (4 x 3 stacked lanes, the
one-pair schedule, the trainer-path legs) that it lacks: bench.py tunes missing shapes while it captures; this runs it and
writes the merged table (the format PyTorch reads back):
    python profiles/tune_missing_bench.py gpurun_out/tunableop_merged.csv [bench.py arguments]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
out = sys.argv[1]
sys.argv = [os.path.join(REPO, "bench.py")] + (sys.argv[2:] or ["--no-cpu-baseline", "--steps", "10"])
import torch  # noqa: E402
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
