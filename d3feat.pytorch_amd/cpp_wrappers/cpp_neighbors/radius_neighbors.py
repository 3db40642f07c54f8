"""``radius_neighbors`` module mirror (reference cpp_wrappers/cpp_neighbors/wrapper.cpp:25-52): ``batch_query``."""
import torch

from ... import ops
from ...datasets.dataloader import _device, _to_dev


def batch_query(queries, supports, q_batches, s_batches, radius=0.1):
    """int32 ndarray [Nq, max_count], like the CPython module (wrapper.cpp:58-238).  Device tensors in -> device
    tensor out; array-likes in -> NumPy array out."""
    as_numpy = not (isinstance(queries, torch.Tensor) and queries.is_cuda)
    dev = _device() if as_numpy else queries.device
    q, s = _to_dev(queries, torch.float32, dev), _to_dev(supports, torch.float32, dev)
    grid = ops.RadiusGrid(s, s_batches, radius)
    _, mx = grid.query(q, q_batches, 1, want_max=True)
    width = int(mx.item())
    grid.status.raise_if_set()
    if width < 1:
        raise RuntimeError("Error")
    idx = grid.query(q, q_batches, width)
    return idx.cpu().numpy() if as_numpy else idx
