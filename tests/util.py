"""Shared helpers for the parity tests."""
import hashlib

import numpy as np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def d2_exact(q, s):
    """float32 ((dx*dx)+(dy*dy))+(dz*dz), the reference's evaluation order (nanoflann.hpp:433-441)."""
    q = q.astype(np.float32)
    s = s.astype(np.float32)
    d = (q - s).astype(np.float32)
    sq = (d * d).astype(np.float32)
    return ((sq[..., 0] + sq[..., 1]).astype(np.float32) + sq[..., 2]).astype(np.float32)


def neighbor_d2(queries, supports, table):
    """d2 of every table entry (inf for the shadow index)."""
    ns = supports.shape[0]
    pad = np.concatenate([supports, np.full((1, 3), np.nan, np.float32)], 0)
    d2 = d2_exact(queries[:, None, :], pad[table])
    d2[table >= ns] = np.inf
    return d2


def assert_neighbors_equal_tie_aware(queries, supports, ours, ref, what="", stats=None):
    """Neighbor tables must agree exactly except for the ORDER inside groups of equal d2, which the reference leaves
    unspecified (std::sort on distance only, nanoflann.hpp:208-214).  Returns (#tie rows, #rows).  A tie group that
    reaches the last column may have been cut by the column limit and keep different members; such rows are counted in
    ``stats['cut']`` (a dict the caller passes) so that a test can insist that none occurred."""
    ours = np.asarray(ours).astype(np.int64)
    ref = np.asarray(ref).astype(np.int64)
    assert ours.shape == ref.shape, "%s shape %s vs %s" % (what, ours.shape, ref.shape)
    da = neighbor_d2(queries, supports, ours)
    db = neighbor_d2(queries, supports, ref)
    assert np.array_equal(da, db), "%s: per-slot squared distances differ (bit-exact check)" % what
    bad = np.nonzero((ours != ref).any(axis=1))[0]
    width = ours.shape[1]
    for r in bad:
        a, b, d = ours[r], ref[r], da[r]
        start = 0
        while start < width:
            end = start
            while end + 1 < width and d[end + 1] == d[start]:
                end += 1
            last_group_cut = end == width - 1  # a tie group cut by the column limit may keep different members
            if not last_group_cut:
                assert sorted(a[start:end + 1]) == sorted(b[start:end + 1]), "%s row %d: tie group differs" % (what, r)
            elif sorted(a[start:end + 1]) != sorted(b[start:end + 1]) and stats is not None:
                stats['cut'] = stats.get('cut', 0) + 1
            start = end + 1
    return len(bad), ours.shape[0]


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def grad_mismatch(a, b, elem_tol=5e-3):
    """(fraction of elements off by more than elem_tol * max|b|, relative L2 error).  A training step is not smooth
    where a LeakyReLU input is ~0: a 1e-7 rounding difference there picks the other branch, which moves one column of
    that layer's bias gradient by ~1/sqrt(rows) and everything upstream by a little.  Networks with BatchNorm (zero-mean
    activations, few-row levels) hit such edges in almost every run, so their gradients are compared by these two
    statistics instead of the max norm: a wrong formula moves most elements, a branch flip moves few."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    scale = max(np.abs(b).max(), 1e-30)
    return float((np.abs(a - b) > elem_tol * scale).mean()), float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


class Cfg:
    pass


def gate_margins(x, idx):
    """Per point: max_c ( x[n,c] - max over the OTHER entries of row n of x'[idx[n,h], c] ), x' = x with the zero shadow
    row (index == N) -- the eval-mode detector keeps a point iff this is >= 0, i.e. iff some channel of the point is
    the maximum of its neighborhood (architectures.py:361-366: `features == max_h features[idx]`, self included).
    x: [N,C] float tensor / array (the normalisation by a positive global maximum does not move the comparison),
    idx: [N,H] integer table whose entries >= N are shadow."""
    import torch
    x = torch.as_tensor(x).double().cpu()
    idx = torch.as_tensor(idx).long().cpu().clamp(max=x.shape[0])
    n = x.shape[0]
    pad = torch.cat([x, torch.zeros((1, x.shape[1]), dtype=x.dtype)], 0)
    out = torch.empty(n, dtype=torch.float64)
    rows = torch.arange(n)
    for a in range(0, n, 4096):   # chunked: [4096, H, C] at a time
        b = min(n, a + 4096)
        nb = pad[idx[a:b]]                                              # [b-a, H, C]
        others = idx[a:b] != rows[a:b, None]
        nb = torch.where(others[:, :, None], nb, torch.full_like(nb, -float('inf')))
        out[a:b] = (x[a:b] - nb.max(dim=1).values).max(dim=1).values
    return out.numpy()


def assert_gate_flips_are_ties(got_scores, ref_scores, x_raw, idx, rel_tol=2e-4, max_share=1e-3):
    """Eval-mode scores against the reference's: the zero / non-zero pattern may differ only at points whose gate is a
    floating-point tie -- the margin by which the point is (or misses being) a channel-wise local maximum is within
    `rel_tol` of the feature scale, i.e. inside the 1e-4 descriptor tolerance -- and such points are rare."""
    import numpy as np
    got = np.asarray(got_scores).reshape(-1)
    ref = np.asarray(ref_scores).reshape(-1)
    flips = np.nonzero((got != 0) != (ref != 0))[0]
    assert flips.size <= max_share * got.size, (flips.size, got.size)
    if flips.size:
        import torch
        xr = torch.as_tensor(x_raw).double().cpu()
        scale = float(xr.abs().max())
        m = gate_margins(xr, idx)
        assert np.abs(m[flips]).max() <= rel_tol * scale, (np.abs(m[flips]).max(), scale, flips[:10])
    return int(flips.size)
