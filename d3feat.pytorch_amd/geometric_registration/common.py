"""Mirror of the reference's ``geometric_registration/common.py:5-21`` (build_correspondence) and the score top-k
of ``test.py:56-57`` on the MFMA matching kernel."""
import numpy as np
import torch

from .. import ops


def build_correspondence(source_desc, target_desc):
    """Mutually closest descriptor pairs [n_mutual, 2] (source index, target index)."""
    as_numpy = not (isinstance(source_desc, torch.Tensor) and source_desc.is_cuda)
    if as_numpy:
        if not torch.cuda.is_available():
            raise RuntimeError("build_correspondence runs on the GPU (no CPU path)")
        s = torch.as_tensor(np.ascontiguousarray(source_desc), dtype=torch.float32).cuda()
        t = torch.as_tensor(np.ascontiguousarray(target_desc), dtype=torch.float32).cuda()
    else:
        s, t = source_desc, target_desc
    row, col, mutual = ops.mutual_nn(s, t)
    i = torch.nonzero(mutual, as_tuple=False).view(-1)
    res = torch.stack([i, row[i].long()], dim=1)
    return res.cpu().numpy() if as_numpy else res


def select_keypoints(scores, num_points):
    """Indices of the `num_points` highest-scoring points, ascending score like ``np.argsort(score)[-k:]``."""
    s = scores.reshape(-1)
    k = min(int(num_points), int(s.numel()))
    return torch.argsort(s, stable=True)[-k:] if isinstance(s, torch.Tensor) else np.argsort(s, kind='stable')[-k:]
