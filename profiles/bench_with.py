"""bench.py with module constants of the package overridden first (the package reads no environment):
    python profiles/bench_with.py ops._FUSED_LINEAR_MAX_CIN=256 train.PairLanes.JOIN_ON_HOST=True -- --quick --steps 20
Everything before `--` is `<module>.<attr>[.<attr>]=<python literal>` (or `tunables.<field>=<int>` for the library's
d3f_tunables), everything after goes to bench.py."""
import ast
import importlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
import d3feat_pytorch_amd  # noqa: E402,F401
for spec in args[:cut]:
    path, value = spec.split("=", 1)
    parts = path.split(".")
    if parts[0] == "tunables":       # a field of the library's d3f_tunables: tunables.atb_task_us=80
        from d3feat_pytorch_amd import _native
        _native.set_tunables(**{parts[1]: ast.literal_eval(value)})
        print("set d3f_tunables.%s = %s" % (parts[1], value), file=sys.stderr)
        continue
    obj = importlib.import_module("d3feat_pytorch_amd." + parts[0]) if parts[0] != "d3f" else d3feat_pytorch_amd
    for i, p in enumerate(parts[1:-1]):
        if not hasattr(obj, p):      # a sub-module that nothing imported yet (models.architectures)
            importlib.import_module("d3feat_pytorch_amd." + ".".join(parts[:i + 2]))
        obj = getattr(obj, p)
    setattr(obj, parts[-1], ast.literal_eval(value))
    print("set %s = %r" % (path, getattr(obj, parts[-1])), file=sys.stderr)
sys.argv = [os.path.join(REPO, "bench.py")] + args[cut + 1:]
import bench  # noqa: E402
bench.main()
