"""``grid_subsampling`` module mirror (reference cpp_wrappers/cpp_subsampling/wrapper.cpp:28-56)."""
import numpy as np
import torch

from ...datasets.dataloader import batch_grid_subsampling_kpconv


def subsample_batch(points, batches, features=None, classes=None, sampleDl=0.1, method="barycenters", max_p=0,
                    verbose=0):
    """(points f32 [N',3], batches i32 [B][, features f32 [N',d]][, classes i32 [N',ld]]) -- wrapper.cpp:62-333: the
    four return shapes of the native module (:316-322) on d3f_grid_subsample / d3f_grid_subsample_ex; `method` is
    validated and ignored exactly like the reference does (wrapper.cpp:92)."""
    if method not in ("barycenters", "voxelcenters"):
        raise RuntimeError('Error parsing method. Valid method names are "barycenters" and "voxelcenters" ')
    as_numpy = not (isinstance(points, torch.Tensor) and points.is_cuda)
    res = batch_grid_subsampling_kpconv(points, batches, features=features, labels=classes, sampleDl=sampleDl,
                                        max_p=max_p)
    return tuple(t.cpu().numpy() for t in res) if as_numpy else res


def subsample(points, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0):
    """Single-cloud form (wrapper.cpp:338-565)."""
    n = int(points.shape[0])
    res = subsample_batch(points, np.array([n], dtype=np.int32), features=features, classes=classes,
                          sampleDl=sampleDl, method=method, verbose=verbose)
    # wrapper.cpp:550-556: points alone, or (points, features) / (points, classes) / (points, features, classes)
    return res[0] if len(res) == 2 else (res[0],) + tuple(res[2:])
