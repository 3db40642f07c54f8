"""d3feat_pytorch_amd -- MI355X-native D3Feat hot path behind the reference's operator API.

Layout mirrors the reference's module paths so its callers keep working:
  models.blocks            KPConv, max_pool, closest_pool, gather, block_decider, *Block classes
  models.architectures     KPFCNN (forward + detection_scores)
  datasets.dataloader      batch_neighbors_kpconv, batch_grid_subsampling_kpconv, collate_fn_descriptor, calibrate_neighbors
  utils.loss               cdist, CircleLoss, DetLoss, ContrastiveLoss
  cpp_wrappers.cpp_neighbors.radius_neighbors.batch_query / cpp_wrappers.cpp_subsampling.grid_subsampling.subsample_batch
  geometric_registration.common.build_correspondence
``install_reference_aliases()`` registers those names in ``sys.modules`` so the reference's own
``models/architectures.py`` (``from models.blocks import *``) runs on top of this package unchanged.
"""
from . import _native  # noqa: F401

__version__ = "0.1.0"

_ALIASES = ["models", "models.blocks", "models.architectures", "datasets", "datasets.dataloader", "utils",
            "utils.loss", "kernels", "kernels.kernel_points", "cpp_wrappers", "cpp_wrappers.cpp_neighbors",
            "cpp_wrappers.cpp_neighbors.radius_neighbors", "cpp_wrappers.cpp_subsampling",
            "cpp_wrappers.cpp_subsampling.grid_subsampling", "geometric_registration",
            "geometric_registration.common"]


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
