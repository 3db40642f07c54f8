"""Device-resident batch construction: mirror of the reference's ``datasets/dataloader.py``.

Reference functions and what replaces them:
  batch_grid_subsampling_kpconv (dataloader.py:12-50)   -> HIP voxel-hash subsampling (csrc/grid_subsample.hip)
  batch_neighbors_kpconv        (dataloader.py:52-67)   -> HIP cell-list radius search (csrc/radius_neighbors.hip)
  collate_fn_descriptor         (dataloader.py:69-189)  -> the same block walk, executed on the GPU: the voxel levels
                                                           are chained on the device (one host read-back of the level
                                                           sizes per pair instead of 17 CPU calls), one cell list per
                                                           level serves its conv / pool / upsample searches
  calibrate_neighbors           (dataloader.py:191-223) -> count-only queries + device histogram
In the reference these run on CPU inside DataLoader worker processes and gate the GPU; here they are stream-ordered
kernels in the training process.  Inputs may be NumPy arrays / CPU tensors (as the reference's datasets yield) or
device tensors; outputs are device tensors (the reference's ``.to(device)`` loop becomes a no-op).

Index tensors are int32 by default (the reference casts to int64, dataloader.py:161-163, which doubles the bytes the
KPConv gather has to read); pass ``index_dtype=torch.int64`` for bit-for-bit dtype parity.
"""
import numpy as np
import torch

from .. import ops


def _device(device=None):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise RuntimeError("d3feat_pytorch_amd needs a HIP device: the batch builder runs on the GPU (no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(a, dtype, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(device)


def batch_grid_subsampling_kpconv(points, batches_len, features=None, labels=None, sampleDl=0.1, max_p=0, verbose=0,
                                  random_grid_orient=True, order=ops.ORDER_REFERENCE):
    """(s_points float32 [N',3], s_len int32 [B][, s_features float32 [N',d]][, s_labels int32 [N',l]]) -- reference
    dataloader.py:12-50: barycentres per voxel, member-mean features, majority-vote labels.

    Rows come in the reference's order (``order=ops.ORDER_REFERENCE``).  D3Feat's collate takes the points-only
    branch; the feature / label branches are the same device call with two more passes."""
    dev = points.device if isinstance(points, torch.Tensor) and points.is_cuda else _device()
    p = _to_dev(points, torch.float32, dev)
    f = _to_dev(features, torch.float32, dev) if features is not None else None
    c = _to_dev(labels, torch.int32, dev) if labels is not None else None
    res = ops.grid_subsample_raw(p, batches_len, sampleDl, max_p=max_p, order=order, features=f, labels=c)
    out, out_len, total, status = res[:4]
    n = int(total.item())  # the one read-back this standalone form needs
    status.raise_if_set()
    if n < 1:
        raise RuntimeError("Error")  # cpp_subsampling/wrapper.cpp:266
    extra = [t[:n] for t in res[4:]]
    # the reference returns classes as [N', ldim] even for a label vector (wrapper.cpp:282-284,308-310)
    return (out[:n], out_len) + tuple(extra)


def batch_neighbors_kpconv(queries, supports, q_batches, s_batches, radius, max_neighbors):
    """int32 [Nq, min(max_neighbors, max_count)] neighbor table -- reference dataloader.py:52-67.

    Rows: supports of the same cloud with d2 < radius^2, ascending d2 (ties by index), padded with len(supports)."""
    dev = queries.device if isinstance(queries, torch.Tensor) and queries.is_cuda else _device()
    q = _to_dev(queries, torch.float32, dev)
    s = _to_dev(supports, torch.float32, dev)
    grid = ops.RadiusGrid(s, s_batches, radius)
    if max_neighbors > 0:
        idx, mx = grid.query(q, q_batches, int(max_neighbors), want_max=True)
        width = min(int(mx.item()), int(max_neighbors))
    else:
        _, mx = grid.query(q, q_batches, 1, want_max=True)
        width = int(mx.item())
        idx = grid.query(q, q_batches, max(width, 1)) if width > 0 else None
    grid.status.raise_if_set()
    if width < 1:
        raise RuntimeError("Error")  # cpp_neighbors/wrapper.cpp:201-205: an all-empty result is an error
    return idx if idx.shape[1] == width else idx[:, :width].contiguous()


# north-star spellings
batch_neighbors = batch_neighbors_kpconv
batch_grid_subsampling = batch_grid_subsampling_kpconv


class _Walk:
    """Which pyramid levels convolve, which pool, and with what radii -- the schedule the reference derives while it
    walks ``config.architecture`` (dataloader.py:98-178), here computed up front as one entry per level:
    ``conv_r`` (None when the level has no convolution), ``pool`` and, for pooling levels, ``dl`` / ``pool_r`` / ``up_r``."""

    def __init__(self, config):
        self.layers = [self._entry(config, run, closing, level)
                       for level, (run, closing) in enumerate(self._levels(config.architecture))]

    @staticmethod
    def _levels(architecture):
        """[(same-resolution blocks, closing pool / strided block or None)] of the encoder (everything before the first
        'global' / 'upsample' block)."""
        levels, run = [], []
        for name in architecture:
            if 'global' in name or 'upsample' in name:
                break
            if 'pool' in name or 'strided' in name:
                levels.append((run, name))
                run = []
            else:
                run.append(name)
        if run:
            levels.append((run, None))
        return levels

    @staticmethod
    def _entry(config, run, closing, level):
        r_normal = config.first_subsampling_dl * config.conv_radius * 2 ** level
        r_deform = r_normal * config.deform_radius / config.conv_radius
        entry = {'conv_r': None, 'pool': closing is not None}
        if run:
            # the reference looks at every block of the level but the last one (`layer_blocks[:-1]`, dataloader.py:118)
            entry['conv_r'] = r_deform if any('deformable' in name for name in run[:-1]) else r_normal
        if closing is not None:
            entry['dl'] = 2 * r_normal / config.conv_radius
            entry['pool_r'] = r_deform if 'deformable' in closing else r_normal
            entry['up_r'] = 2 * entry['pool_r']
        return entry


def _conv_table(grid_for, pts, lens, level, e, lim, reverse_tables, want_max=False, group=0):
    """neighbors[l] (dataloader.py:122-128); with ``want_max`` a pair (table, device max neighbor count).  With ``reverse_tables`` the same search also leaves the table's transpose
    in search form (its whole ranked list + the key of the last kept entry): the in-radius relation of a cloud with
    itself is symmetric, so the list of s IS the candidate set of rev(s) -- no transposition pass."""
    if e['conv_r'] is None:
        empty = torch.zeros((0, 1), dtype=torch.int32, device=pts[level].device)
        return (empty, None) if want_max else empty
    grid = grid_for(level, e['conv_r'])
    n = pts[level].shape[0]
    if not (reverse_tables and ops.wants_reverse_table(n)):
        return grid.query(pts[level], lens[level], lim, want_max=want_max, max_group=group)
    res = grid.query(pts[level], lens[level], lim, want_max=want_max, wide=ops.REV_WIDTH_CONV, want_last_key=True,
                     max_group=group)
    tab, wide, lkey = res[0], res[-2], res[-1]
    # ... evaluated once here into the exact form (membership of every entry, compacted {q - s, q} rows): the training
    # stream then reads each neighborhood as one coalesced run
    rev = ops.filter_reverse_table(ops.ReverseTable(wide, n, lim, n, last_key=lkey, status=grid.status),
                                   pts[level], pts[level])
    ops.attach_reverse_table(tab, rev)
    return (tab, res[1]) if want_max else tab


def _pool_tables(grid_for, pts, lens, level, e, lim, reverse_tables, status, group=0, engine_upsamples=False,
                 mx_out=None, tr_counts=None):
    """(pools[l], device max count, upsamples[l]) (dataloader.py:141-152).  The transpose of the pooling table (coarse
    points around every fine point, radius r) is the leading part of the rows of the upsampling table (same point
    pairs, radius 2r, nearest first): nothing extra is searched, the pooling query only adds its last-kept keys."""
    grid = grid_for(level, e['pool_r'])
    ns = pts[level].shape[0]
    if engine_upsamples and ops.UPSAMPLES_FROM_POOL and reverse_tables and ops.wants_reverse_table(ns):
        # the pooling search (coarse queries over the fine cloud) finds every (fine, coarse) pair within the pooling radius;
        # the engine's upsampling rows are the same pairs seen from the fine side: the pooling search appends every pair to the
        # fine point's list as it goes, and the rows are ranked from those lists instead of searched for a second time (one wave per FINE point scanning the coarse cell list: 114 us at
        # level 0 of a 3-pair stack)
        bound = min(float(e['up_r']), max(float(e['pool_r']), 1.1 * float(e['dl']) * 3.0 ** 0.5))
        tab, mx, lkey, transposed = grid.query_pool_transposed(pts[level + 1], lens[level + 1], lim, max_group=group,
                                                               mx_out=mx_out, counts=tr_counts)
        up = grid_for(level + 1, e['up_r']).prefix_rows_from_transposed(pts[level], lens[level], lim, e['pool_r'],
                                                                        transposed, nearest_bound=bound)
        rev = ops.filter_reverse_table(ops.ReverseTable(up, pts[level + 1].shape[0], lim, ns, last_key=lkey,
                                                        radius=e['pool_r'], status=status), pts[level + 1], pts[level])
        ops.attach_reverse_table(tab, rev)
        return tab, mx, up
    if engine_upsamples:
        # inside the training engine the upsampling table is read in two places only: column 0 (closest_pool) and the part
        # of every row within the POOLING radius (the transpose below) -- the prefix form ranks just that (the 2 r rows
        # are ~68 entries wide, the part within r ~9: no LDS sorting network, 156 -> 80 us at level 0 of a 3-pair stack)
        # the nearest coarse point of a fine point is at most its own voxel's barycentre away: within the voxel diagonal
        # dl sqrt(3) (dl = 0.8 r for the reference's radii: 1.39 r); 1.1 of that, capped by the search radius
        bound = min(float(e['up_r']), max(float(e['pool_r']), 1.1 * float(e['dl']) * 3.0 ** 0.5))
        up = grid_for(level + 1, e['up_r']).query_prefix(pts[level], lens[level], lim, e['pool_r'], nearest_bound=bound)
    else:
        up = grid_for(level + 1, e['up_r']).query(pts[level], lens[level], lim)
    if not (reverse_tables and ops.wants_reverse_table(ns)):
        tab, mx = grid.query(pts[level + 1], lens[level + 1], lim, want_max=True, max_group=group, mx_out=mx_out)
        return tab, mx, up
    tab, mx, lkey = grid.query(pts[level + 1], lens[level + 1], lim, want_max=True, want_last_key=True,
                               max_group=group, mx_out=mx_out)
    rev = ops.filter_reverse_table(ops.ReverseTable(up, pts[level + 1].shape[0], lim, ns, last_key=lkey,
                                                    radius=e['pool_r'], status=status), pts[level + 1], pts[level])
    ops.attach_reverse_table(tab, rev)
    return tab, mx, up


def build_pyramid_static(points, lengths, config, neighborhood_limits, capacities, order=ops.ORDER_REFERENCE,
                         reverse_tables=False, status=None, conv_widths=True, group=0, engine_upsamples=False,
                         clear_status=False):
    """Capacity-shaped pyramid: every level l has ``capacities[l]`` rows, the live row counts stay on the device.

    No host synchronisation at all (hipGraph-capturable): voxel levels write into fixed-capacity buffers (rows past
    the live count are zero), searches give rows past the live count an all-shadow row, and the shadow index of a
    table is the support CAPACITY, so the operators run unchanged on the padded shapes and padded rows never reach a
    live row.  A level outgrowing its capacity sets D3F_ST_CAPACITY in the returned status word.

    ``group`` > 0: the batch stacks several reference batches of ``group`` clouds each (8 fragment pairs: 16 clouds,
    group 2).  Clouds never see each other in any search; what a reference batch shares is the WIDTH of its tables
    (min(limit, max count of that batch), dataloader.py:64-66) and the detector's normaliser, so the width entries come
    per group (int32 [B/group]) and the batch carries ``_group`` for the operators that need it."""
    # ``engine_upsamples``: the upsampling tables in their prefix form (ops.RadiusGrid.query_prefix) -- for consumers that
    # read column 0 and the part within the pooling radius only (train.TrainStep); the reference's rows otherwise.
    dev = points.device
    walk = _Walk(config)
    status = status if status is not None else ops.DeviceStatus(dev)   # (a caller's word collects flags across builds)
    pts, lens = [points], [ops._lens(lengths, dev, "lengths")]
    # one cell list per level (conv / pool searches of level l and the upsampling search of level l - 1 share its radius);
    # their bucket counters, the pooling tables' max-count words and (``clear_status``) the build's status word are
    # cleared by ONE launch in front of everything else (ten launches before)
    n_pool = sum(1 for e in walk.layers if e['pool'])
    rows = [int(points.shape[0])] + [int(capacities[l]) for l in range(1, n_pool + 1)]
    n_mx = -(-int(lens[0].numel()) // int(group)) if group else 1
    mx_all = torch.empty(max(1, n_pool * n_mx), dtype=torch.int32, device=dev)
    spaces = [ops.RadiusGrid.workspace(n, dev) for n in rows]
    # (the per-fine-point counters of the pooling searches' transposes, all levels in one buffer: cleared with the rest)
    tr_all = None
    if engine_upsamples and ops.UPSAMPLES_FROM_POOL and reverse_tables and len(spaces) <= 5:
        tr_all = torch.empty(sum(rows[:n_pool]), dtype=torch.int32, device=dev)
    ops.zero_buffers([z for _, z in spaces] + [mx_all] + ([tr_all] if tr_all is not None else []) +
                     ([status.word] if clear_status else []))
    tr_off = [sum(rows[:l]) for l in range(n_pool + 1)]
    for e in walk.layers:
        if e['pool']:
            out, out_len, _, _ = ops.grid_subsample_raw(pts[-1], lens[-1], e['dl'], order=order, status=status,
                                                        out_cap=int(capacities[len(pts)]))
            pts.append(out)
            lens.append(out_len)
    grids = {}
    used = set()

    def grid_for(level, radius):
        key = (level, float(radius))
        if key not in grids:
            ws = None
            if level not in used:      # (a second radius on the same level builds, and clears, a list of its own)
                used.add(level)
                ws = spaces[level][0]
            grids[key] = ops.RadiusGrid(pts[level], lens[level], radius, status=status, ws=ws)
        return grids[key]

    empty_idx = torch.empty((0, 1), dtype=torch.int32, device=dev)
    neighbors, neighbors_width, pools, pools_width, upsamples = [], [], [], [], []
    level = 0
    for li, e in enumerate(walk.layers):
        lim = int(neighborhood_limits[li])
        # the level-0 table's max count is what the detector's eval-mode gate needs (conv_widths=False: training only --
        # 38k waves reducing into one word cost the level-0 search 25 us)
        if li == 0 and conv_widths:
            tab, tab_max = _conv_table(grid_for, pts, lens, level, e, lim, reverse_tables, want_max=True, group=group)
        else:
            tab, tab_max = _conv_table(grid_for, pts, lens, level, e, lim, reverse_tables), None
        neighbors.append(tab)
        neighbors_width.append(tab_max)
        if e['pool']:
            # static width = the limit; the reference trims to min(limit, max_count) (dataloader.py:64-66).  Only max_pool
            # can tell the difference (a full row gains zero-valued shadow candidates): it gets max_count, on the device
            tab, mx, up = _pool_tables(grid_for, pts, lens, level, e, lim, reverse_tables, status, group=group,
                                       engine_upsamples=engine_upsamples, mx_out=mx_all[level * n_mx:(level + 1) * n_mx],
                                       tr_counts=tr_all[tr_off[level]:tr_off[level + 1]] if tr_all is not None else None)
            pools.append(tab)
            pools_width.append(mx)
            upsamples.append(up)
            level += 1
        else:
            pools.append(empty_idx)
            pools_width.append(None)
            upsamples.append(empty_idx)
    n_levels = len(neighbors)
    return {'points': [pts[min(i, len(pts) - 1)] for i in range(n_levels)], 'neighbors': neighbors, 'pools': pools,
            'pools_width': pools_width, 'neighbors_width': neighbors_width, 'upsamples': upsamples,
            'stack_lengths': [lens[min(i, len(lens) - 1)] for i in range(n_levels)], '_status': status, '_static': True,
            '_group': int(group)}


def build_pyramid(points, lengths, config, neighborhood_limits, index_dtype=torch.int32, exact_width=False,
                  order=ops.ORDER_REFERENCE, reverse_tables=False):
    """points [N0,3] + stack lengths [B] -> dict(points, neighbors, pools, upsamples, stack_lengths) on the device.

    One host synchronisation (level sizes) per call; with ``exact_width`` a second one trims every neighbor table
    to the reference's width min(limit, max_count)."""
    dev = points.device
    walk = _Walk(config)
    L = len(walk.layers)
    status = ops.DeviceStatus(dev)
    # 1. chain the voxel levels on the device
    cap_pts, lens, totals = [points], [ops._lens(lengths, dev, "lengths")], []
    for li, e in enumerate(walk.layers):
        if not e['pool']:
            continue
        out, out_len, total, _ = ops.grid_subsample_raw(cap_pts[-1], lens[-1], e['dl'], order=order, status=status)
        cap_pts.append(out)
        lens.append(out_len)
        totals.append(total)
    if totals:
        sizes = torch.cat(totals + [status.word]).tolist()  # the single read-back
        if sizes[-1]:
            status.raise_if_set()
        pts = [points] + [cap_pts[i + 1][:sizes[i]] for i in range(len(totals))]
    else:
        pts = [points]
    # 2. neighbor tables; one cell list per (level, radius)
    grids = {}

    def grid_for(level, radius):
        key = (level, float(radius))
        if key not in grids:
            grids[key] = ops.RadiusGrid(pts[level], lens[level], radius, status=status)
        return grids[key]

    empty_idx = torch.zeros((0, 1), dtype=index_dtype, device=dev)
    neighbors, neighbors_width, pools, pools_width, upsamples, maxima = [], [], [], [], [], []
    search_form = reverse_tables and not exact_width and index_dtype == torch.int32
    level = 0
    for li, e in enumerate(walk.layers):
        lim = int(neighborhood_limits[li])

        def run(qlevel, slevel, radius, keep_max=None):
            res = grid_for(slevel, radius).query(pts[qlevel], lens[qlevel], lim,
                                                 want_max=exact_width or keep_max is not None)
            if exact_width:
                maxima.append(res[1])
                return res[0]
            if keep_max is not None:
                keep_max.append(res[1])
                return res[0]
            return res

        def run_pool(qlevel, slevel, radius):
            if exact_width:
                return run(qlevel, slevel, radius), None
            return grid_for(slevel, radius).query(pts[qlevel], lens[qlevel], lim, want_max=True)

        if search_form:   # the training path: tables at the limit's width + their transposes straight from the searches
            if li == 0:
                tab, tab_max = _conv_table(grid_for, pts, lens, level, e, lim, True, want_max=True)
            else:
                tab, tab_max = _conv_table(grid_for, pts, lens, level, e, lim, True), None
            neighbors.append(tab)
            neighbors_width.append(tab_max)
            if e['pool']:
                tab, tab_width, up = _pool_tables(grid_for, pts, lens, level, e, lim, True, status)
                pools.append(tab)
                pools_width.append(tab_width)
                upsamples.append(up)
                level += 1
            else:
                pools.append(empty_idx)
                pools_width.append(None)
                upsamples.append(empty_idx)
            continue
        if e['conv_r'] is not None and li > 0 and not exact_width:
            neighbors.append(run(level, level, e['conv_r']))
            neighbors_width.append(None)
        elif e['conv_r'] is not None:
            neighbors.append(run(level, level, e['conv_r'], keep_max=None if exact_width else neighbors_width))
        else:
            neighbors.append(empty_idx)
            if not exact_width:
                neighbors_width.append(None)
        if e['pool']:
            tab, tab_width = run_pool(level + 1, level, e['pool_r'])
            pools.append(tab)
            pools_width.append(tab_width)
            upsamples.append(run(level, level + 1, e['up_r']))
            level += 1
        else:
            pools.append(empty_idx)
            pools_width.append(None)
            upsamples.append(empty_idx)
    if exact_width and maxima:
        mx = torch.cat(maxima).tolist()  # second read-back; walk in the order the maxima were appended
        k = 0
        for li, e in enumerate(walk.layers):
            for name, lst, present in (("n", neighbors, e['conv_r'] is not None), ("p", pools, e['pool']),
                                       ("u", upsamples, e['pool'])):
                if not present:
                    continue
                w = mx[k]
                k += 1
                if w < 1:
                    raise RuntimeError("Error")
                if w < lst[li].shape[1]:
                    lst[li] = lst[li][:, :w].contiguous()
    if reverse_tables and not search_form and index_dtype == torch.int32:
        # reference-width (trimmed) tables: CSR transposes of the tables KPConv runs on (conv + pooling), attached to the
        # tables themselves (ops.build_reverse_table), so the batch dict keeps the reference's keys
        level = 0
        for li, e in enumerate(walk.layers):
            if e['conv_r'] is not None and ops.wants_reverse_table(pts[level].shape[0]):
                ops.build_reverse_table(neighbors[li], pts[level].shape[0])
            if e['pool']:
                if ops.wants_reverse_table(pts[level].shape[0]):
                    ops.build_reverse_table(pools[li], pts[level].shape[0])
                level += 1
    if index_dtype != torch.int32:
        neighbors = [t.to(index_dtype) for t in neighbors]
        pools = [t.to(index_dtype) for t in pools]
        upsamples = [t.to(index_dtype) for t in upsamples]
    n_levels = len(neighbors)
    out_pts = [pts[min(i, len(pts) - 1)] for i in range(n_levels)]
    out_lens = [lens[min(i, len(lens) - 1)] for i in range(n_levels)]
    out = {'points': out_pts, 'neighbors': neighbors, 'pools': pools, 'upsamples': upsamples,
           'stack_lengths': out_lens, '_status': status}
    if not exact_width:   # full-width tables: max_pool / the detector get each table's max neighbor count (device int32[1])
        out['pools_width'] = pools_width
        out['neighbors_width'] = neighbors_width
    return out


def collate_fn_descriptor(list_data, config, neighborhood_limits, device=None, index_dtype=torch.int32,
                          exact_width=True, reverse_tables=False, keep_status=False):
    """One fragment pair -> the multi-scale batch dict of the reference (dataloader.py:69-189), built on the GPU."""
    assert len(list_data) == 1
    dev = _device(device)
    pts0, pts1, feat0, feat1, sel_corr, dist_keypts = list_data[0]
    p0, p1 = _to_dev(pts0, torch.float32, dev), _to_dev(pts1, torch.float32, dev)
    points = torch.cat([p0, p1], dim=0)
    feats = torch.cat([_to_dev(feat0, torch.float32, dev), _to_dev(feat1, torch.float32, dev)], dim=0)
    lengths = torch.tensor([p0.shape[0], p1.shape[0]], dtype=torch.int32, device=dev)
    d = build_pyramid(points, lengths, config, neighborhood_limits, index_dtype=index_dtype, exact_width=exact_width,
                      reverse_tables=reverse_tables)
    if not keep_status:   # (the voxel levels were checked at the size read-back; search flags need a caller that syncs)
        d.pop('_status')
    d['features'] = feats
    d['corr'] = sel_corr.to(dev) if isinstance(sel_corr, torch.Tensor) else torch.from_numpy(np.asarray(sel_corr)).to(dev)
    d['dist_keypts'] = dist_keypts.to(dev) if isinstance(dist_keypts, torch.Tensor) \
        else torch.from_numpy(np.asarray(dist_keypts)).to(dev)
    return d


def calibrate_neighbors(dataset, config, collate_fn=None, keep_ratio=0.8, samples_threshold=2000, device=None):
    """Per-layer neighbor cap = the keep_ratio quantile of the neighborhood-size histogram (dataloader.py:191-223).

    The reference materialises uncapped neighbor matrices on CPU; here only per-query counts leave the kernel."""
    dev = _device(device)
    hist_n = int(np.ceil(4 / 3 * np.pi * (config.deform_radius + 1) ** 3))
    walk = _Walk(config)
    n_layers = config.num_layers
    hists = torch.zeros((n_layers, hist_n), dtype=torch.int64, device=dev)
    for i in range(len(dataset)):
        pts0, pts1 = dataset[i][0], dataset[i][1]
        p = torch.cat([_to_dev(pts0, torch.float32, dev), _to_dev(pts1, torch.float32, dev)], dim=0)
        lens = torch.tensor([len(pts0), len(pts1)], dtype=torch.int32, device=dev)
        for li, e in enumerate(walk.layers[:n_layers]):
            if e['conv_r'] is not None:
                _, counts = ops.RadiusGrid(p, lens, e['conv_r']).query(p, lens, 1, want_counts=True)
                c = counts.long()
                hists[li] += torch.bincount(c[c < hist_n], minlength=hist_n)[:hist_n]
            if e['pool']:
                p, lens = batch_grid_subsampling_kpconv(p, lens, sampleDl=e['dl'])
        if int(hists.sum(dim=1).min().item()) > samples_threshold:
            break
    h = hists.cpu().numpy()
    cumsum = np.cumsum(h.T, axis=0)
    return np.sum(cumsum < (keep_ratio * cumsum[hist_n - 1, :]), axis=0)


class _PairLoader:
    """Minimal stand-in for the reference's DataLoader (dataloader.py:225-238): batch_size 1, collate on device."""

    def __init__(self, dataset, config, neighborhood_limits, shuffle):
        self.dataset, self.config, self.limits, self.shuffle = dataset, config, neighborhood_limits, shuffle
        self.batch_size = 1

    def __len__(self):
        return len(self.dataset)

    def __iter__(self):
        order = np.random.permutation(len(self.dataset)) if self.shuffle else np.arange(len(self.dataset))
        for i in order:
            yield collate_fn_descriptor([self.dataset[int(i)]], self.config, self.limits)


def get_dataloader(dataset, batch_size=1, num_workers=0, shuffle=True, neighborhood_limits=None):
    if neighborhood_limits is None:
        neighborhood_limits = calibrate_neighbors(dataset, dataset.config, collate_fn=collate_fn_descriptor)
    return _PairLoader(dataset, dataset.config, neighborhood_limits, shuffle), neighborhood_limits
