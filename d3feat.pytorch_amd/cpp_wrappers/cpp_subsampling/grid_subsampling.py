"""``grid_subsampling`` module mirror (reference cpp_wrappers/cpp_subsampling/wrapper.cpp:28-56)."""
import numpy as np
import torch

from ...datasets.dataloader import batch_grid_subsampling_kpconv


def subsample_batch(points, batches, features=None, classes=None, sampleDl=0.1, method="barycenters", max_p=0,
                    verbose=0):
    """(points f32 [N',3], batches i32 [B]) -- wrapper.cpp:62-333, points-only; `method` is validated and ignored
    exactly like the reference does (wrapper.cpp:92)."""
    if method not in ("barycenters", "voxelcenters"):
        raise RuntimeError('Error parsing method. Valid method names are "barycenters" and "voxelcenters" ')
    if features is not None or classes is not None:
        raise NotImplementedError("feature / label subsampling is outside the D3Feat hot path")
    as_numpy = not (isinstance(points, torch.Tensor) and points.is_cuda)
    p, b = batch_grid_subsampling_kpconv(points, batches, sampleDl=sampleDl, max_p=max_p)
    return (p.cpu().numpy(), b.cpu().numpy()) if as_numpy else (p, b)


def subsample(points, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0):
    """Single-cloud form (wrapper.cpp:338-565)."""
    n = int(points.shape[0])
    p, _ = subsample_batch(points, np.array([n], dtype=np.int32), features=features, classes=classes,
                           sampleDl=sampleDl, method=method, verbose=verbose)
    return p
