"""d3feat_pytorch_amd -- MI355X-native D3Feat hot path behind the reference's operator API.

Layout mirrors the reference's module paths so its callers keep working:
  models.blocks            KPConv, max_pool, closest_pool, gather, block_decider, *Block classes
  models.architectures     KPFCNN (forward + detection_scores)
  datasets.dataloader      batch_neighbors_kpconv, batch_grid_subsampling_kpconv, collate_fn_descriptor, calibrate_neighbors
  utils.loss               cdist, CircleLoss, DetLoss, ContrastiveLoss
  cpp_wrappers.cpp_neighbors.radius_neighbors.batch_query / cpp_wrappers.cpp_subsampling.grid_subsampling.subsample_batch
  geometric_registration.common.build_correspondence          geometric_registration.evaluate (test.py's protocol)
  datasets.ThreeDMatch     ThreeDMatchDataset / ThreeDMatchTestset          trainer.Trainer (trainer.py's epoch loop)
``install_reference_aliases()`` registers those names in ``sys.modules`` so the reference's own
``models/architectures.py`` (``from models.blocks import *``) runs on top of this package unchanged.
"""
import os as _os

# The training step keeps several HIP streams busy at once (a training stream and a pyramid side stream per pair in
# flight, train.PairLanes).  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and
# streams that share a queue run back to back -- measured: two pairs in flight 244 pairs/s on 4 queues, 302 on 16 or more
# (profiles/r03_queue_pipes.txt).  Read when the runtime initialises (first HIP call), so it is set at import; a value
# the user exported wins.
_HWQ_SET_HERE = "GPU_MAX_HW_QUEUES" not in _os.environ
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
# set when the variable had to be set HERE and the HIP runtime was already up (it is then ignored: train.PairLanes warns)
import sys as _sys
HIP_WAS_INITIALISED_AT_IMPORT = bool(_HWQ_SET_HERE and 'torch' in _sys.modules
                                     and _sys.modules['torch'].cuda.is_initialized())

from . import _native  # noqa: F401,E402

__version__ = "0.1.0"

_ALIASES = ["models", "models.blocks", "models.architectures", "datasets", "datasets.dataloader", "utils",
            "utils.loss", "kernels", "kernels.kernel_points", "cpp_wrappers", "cpp_wrappers.cpp_neighbors",
            "cpp_wrappers.cpp_neighbors.radius_neighbors", "cpp_wrappers.cpp_subsampling",
            "cpp_wrappers.cpp_subsampling.grid_subsampling", "geometric_registration",
            "geometric_registration.common", "datasets.ThreeDMatch", "trainer"]


def install_reference_aliases(names=None, overwrite=False):
    """Make ``import models.blocks`` etc. resolve to this package (drop-in for the reference's import paths)."""
    import importlib
    import sys
    for n in (names or _ALIASES):
        if n in sys.modules and not overwrite:
            continue
        sys.modules[n] = importlib.import_module(__name__ + "." + n)


def build(verbose=False):
    return _native.build(verbose=verbose)


def enable_tuned_gemms(path=None, tune_missing=False):
    """Library-GEMM kernel selection for the network's shapes (PyTorch TunableOp over rocBLAS / hipBLASLt).

    About 95 library GEMMs remain in a training step (unary blocks of the coarse levels, the few-point KPConv weight /
    input gradients); their shapes are odd for a BLAS heuristic (160..2000 rows against 512..7680 reduction or output
    columns) and the default pick is 6 % of the step slower than the best solution.  ``tuned/tunableop_gfx950.csv``
    holds the winners for the static-capacity shapes of S1-class pairs, produced by
    ``PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 python bench.py``; it carries validators (PyTorch, HIP,
    rocBLAS / hipBLASLt versions, gfx arch) and is ignored by PyTorch when they do not match.  Shapes that are not in
    the table use the library default unless ``tune_missing``.  Returns True when the table was loaded."""
    import os
    import torch
    tunable = torch.cuda.tunable
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "tunableop_gfx950.csv")
    # rocBLAS candidates only (368 of the 375 shipped rows are rocBLAS winners).  Rounds 4-5 kept hipBLASLt out of the race
    # because its winners stalled under concurrent replay; that stall was the shared BLAS handle (round 6,
    # profiles/r06_stall_root_cause.txt), but the shipped table was tuned without them.  Read by PyTorch when TunableOp
    # first runs; a value the user exported wins.  Set HERE, by the opt-in, not at import.
    os.environ.setdefault("PYTORCH_TUNABLEOP_HIPBLASLT_ENABLED", "0")
    if not os.path.exists(path):
        return False
    tunable.enable(True)
    tunable.tuning_enable(bool(tune_missing))
    try:
        import tempfile
        # anything PyTorch decides to write back at exit goes to a scratch file, never into the package
        tunable.set_filename(os.path.join(tempfile.gettempdir(), "d3f_tunableop_%d.csv" % os.getpid()))
        return bool(tunable.read_file(path))
    except Exception:  # pragma: no cover - an unreadable table must not take the training step down
        tunable.enable(False)
        return False


import contextlib as _contextlib

TUNE_MISSING_GEMMS = True


@_contextlib.contextmanager
def tuning_missing_gemms(max_ms=10, max_iterations=20):
    """While active, library-GEMM shapes that are NOT in the loaded TunableOp table are tuned at their first call (a few
    solutions' worth of milliseconds each) instead of running on the library's default pick.  ``TrainStep.capture``
    wraps its warm-up steps in this: the shipped table holds the shapes of S1-class pairs at exact capacities, and any
    other capacity set (a trainer's size classes, head-room) changes every row count -- measured at 10 % head-room:
    221.9 pairs/s on the default picks, 238.6 after 18 s of tuning 85 shapes during the capture
    (profiles/tune_on_capture_experiment.py).  No effect (yields False) unless ``enable_tuned_gemms`` is on."""
    import torch
    tunable = torch.cuda.tunable
    if not (torch.cuda.is_available() and tunable.is_enabled()) or not TUNE_MISSING_GEMMS:
        yield False      # (TUNE_MISSING_GEMMS = False: kernel traces of bench.py without thousands of tuning candidates)
        return
    prev = (tunable.tuning_is_enabled(), tunable.get_max_tuning_duration(), tunable.get_max_tuning_iterations())
    tunable.set_max_tuning_duration(int(max_ms))
    tunable.set_max_tuning_iterations(int(max_iterations))
    tunable.tuning_enable(True)
    try:
        yield True
    finally:
        tunable.tuning_enable(prev[0])
        tunable.set_max_tuning_duration(prev[1])
        tunable.set_max_tuning_iterations(prev[2])

