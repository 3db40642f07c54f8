"""CPU restatement (PyTorch fp32, autograd-able) of the PyTorch side of the D3Feat hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file; the product
never does.  Each function names the reference lines it restates.  It is pinned against golden vectors produced by
importing the real reference in the build container (tests/golden/make_golden.py -> tests/golden/*.npz).
"""
import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------------
# operators
# ---------------------------------------------------------------------------------------------------------------
def kpconv(q_pts, s_pts, neighb_inds, x, kernel_points, weights, extent, influence='linear', aggregation='sum'):
    """models/blocks.py:237-382, rigid path; influence / aggregation modes of :327-352."""
    idx = neighb_inds.long()
    s_pad = torch.cat([s_pts, torch.full_like(s_pts[:1], 1e6)], 0)          # :277 shadow point
    rel = s_pad[idx] - q_pts[:, None, :]                                     # :280-283  [n,H,3]
    diff = rel[:, :, None, :] - kernel_points[None, None, :, :]              # :293-294  [n,H,K,3]
    sq = (diff ** 2).sum(dim=3)                                              # :297
    if influence == 'constant':
        w = torch.ones_like(sq).transpose(1, 2)                              # :329-331
    elif influence == 'linear':
        w = torch.clamp(1 - torch.sqrt(sq) / extent, min=0.0).transpose(1, 2)    # :336-337  [n,K,H]
    elif influence == 'gaussian':
        sigma = extent * 0.3                                                 # :341
        w = torch.exp(-sq / (2 * sigma ** 2 + 1e-9)).transpose(1, 2)         # :342 -> radius_gaussian :66-73
    else:
        raise ValueError('Unknown influence function type (config.KP_influence)')
    if aggregation == 'closest':
        nearest = torch.argmin(sq, dim=2)                                    # :349
        w = w * F.one_hot(nearest, kernel_points.shape[0]).transpose(1, 2)   # :350
    elif aggregation != 'sum':
        raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
    x_pad = torch.cat([x, torch.zeros_like(x[:1])], 0)                       # :356
    nx = x_pad[idx]                                                          # :359      [n,H,Cin]
    wf = torch.matmul(w, nx)                                                 # :362      [n,K,Cin]
    out = torch.einsum('nkc,kco->no', wf, weights)                           # :369-374
    nn = torch.clamp((nx.sum(dim=-1) > 0).sum(dim=-1), min=1)                # :377-379
    return out / nn[:, None].to(out.dtype)


def kpconv_deformable(q_pts, s_pts, neighb_inds, x, kernel_points, weights, extent, offset_features, modulated=False,
                      influence='linear', aggregation='sum'):
    """models/blocks.py:243-387, deformable=True: ``offset_features`` [n, (3 or 4) K] is the output of the offset
    convolution plus its bias (:244).  Returns (out, min_d2 [n,K], deformed_KP [n,K,3]).  Instead of compacting the
    in-range neighbors with topk (:305-321, an order-only device) the out-of-range ones are masked to the shadow."""
    K = kernel_points.shape[0]
    idx = neighb_inds.long()
    if modulated:                                                            # :247-250
        unscaled = offset_features[:, :3 * K].reshape(-1, K, 3)
        modulations = 2 * torch.sigmoid(offset_features[:, 3 * K:])
    else:                                                                    # :252-253
        unscaled, modulations = offset_features.reshape(-1, K, 3), None
    deformed = unscaled * extent + kernel_points                             # :256,287
    ns = s_pts.shape[0]
    s_pad = torch.cat([s_pts, torch.full_like(s_pts[:1], 1e6)], 0)
    rel = s_pad[idx] - q_pts[:, None, :]
    sq = ((rel[:, :, None, :] - deformed[:, None, :, :]) ** 2).sum(dim=3)    # :293-297  [n,H,K]
    min_d2 = sq.min(dim=1)[0]                                                # :301
    in_range = (sq < extent ** 2).any(dim=2)                                 # :304
    idx = torch.where(in_range, idx, torch.full_like(idx, ns))               # :316-321
    sq = torch.where(in_range[:, :, None], sq, torch.full_like(sq, 1e12))    # (the shadow rows topk would have dropped)
    if influence == 'constant':
        w = torch.ones_like(sq).transpose(1, 2)
    elif influence == 'linear':
        w = torch.clamp(1 - torch.sqrt(sq) / extent, min=0.0).transpose(1, 2)
    elif influence == 'gaussian':
        w = torch.exp(-sq / (2 * (extent * 0.3) ** 2 + 1e-9)).transpose(1, 2)
    else:
        raise ValueError('Unknown influence function type (config.KP_influence)')
    if aggregation == 'closest':
        w = w * F.one_hot(torch.argmin(sq, dim=2), K).transpose(1, 2)
    elif aggregation != 'sum':
        raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
    x_pad = torch.cat([x, torch.zeros_like(x[:1])], 0)
    nx = x_pad[idx]
    wf = torch.matmul(w, nx)
    if modulations is not None:
        wf = wf * modulations.unsqueeze(2)                                   # :365-366
    out = torch.einsum('nkc,kco->no', wf, weights)
    nn = torch.clamp((nx.sum(dim=-1) > 0).sum(dim=-1), min=1)
    return out / nn[:, None].to(out.dtype), min_d2, deformed


def max_pool(x, inds):
    """models/blocks.py:94-110."""
    x_pad = torch.cat([x, torch.zeros_like(x[:1])], 0)
    return x_pad[inds.long()].max(dim=1)[0]


def closest_pool(x, inds):
    """models/blocks.py:79-91."""
    x_pad = torch.cat([x, torch.zeros_like(x[:1])], 0)
    return x_pad[inds.long()[:, 0]]


def detection_scores(features, neighbors, training=True):
    """models/architectures.py:322-368."""
    n = features.shape[0]
    nb = torch.cat([neighbors.long(), torch.full_like(neighbors[:1].long(), n)], 0)     # :334-335
    f = torch.cat([features, torch.zeros_like(features[:1])], 0)                        # :332-333
    f = f / (torch.max(f) + 1e-6)                                                       # :342
    nf = f[nb]                                                                          # :345
    num = torch.clamp((nf.sum(dim=-1) != 0).sum(dim=-1, keepdim=True), min=1)           # :346-348
    mean = nf.sum(dim=1) / num                                                          # :349
    alpha = F.softplus(f - mean)                                                        # :350
    beta = f / (1e-6 + f.max(dim=1, keepdim=True)[0])                                   # :353-354
    scores = (alpha * beta).max(dim=1, keepdim=True)[0]                                 # :356-358
    if not training:
        detected = (f == nf.max(dim=1)[0]).float().max(dim=1, keepdim=True)[0]          # :361-365
        scores = scores * detected
    return scores[:-1]


def cdist_euclidean(a, b):
    """utils/loss.py:35-39."""
    d = a[:, None, :] - b[None, :, :]
    return torch.sqrt((d ** 2).sum(dim=-1) + 1e-12)


def cdist(a, b, metric='euclidean'):
    """utils/loss.py:8-44, every metric."""
    if metric == 'cosine':
        return torch.sqrt(2 - 2 * a @ b.t())
    if metric == 'arccosine':
        return torch.acos(a @ b.t())
    d = a[:, None, :] - b[None, :, :]
    if metric == 'sqeuclidean':
        return (d ** 2).sum(-1)
    if metric == 'euclidean':
        return torch.sqrt((d ** 2).sum(-1) + 1e-12)
    if metric == 'cityblock':
        return d.abs().sum(-1)
    raise NotImplementedError('The following metric is not implemented by `cdist` yet: {}'.format(metric))


def circle_loss(anchor, positive, dist_keypts, log_scale=10.0, safe_radius=0.1, pos_margin=0.1, neg_margin=1.4,
                metric='euclidean'):
    """utils/loss.py:111-141 -> (loss, accuracy, furthest_positive, average_negative, dists)."""
    dists = cdist_euclidean(anchor, positive) if metric == 'euclidean' else cdist(anchor, positive, metric)
    m = dists.shape[0]
    eye = torch.eye(m, dtype=torch.bool)
    neg_mask = dist_keypts > safe_radius
    fp = (dists * eye.float()).max(dim=1)[0]
    cn = (dists + 1e5 * eye.float()).min(dim=1)[0]
    avg_neg = (dists.sum(dim=-1) - fp) / (m - 1)
    acc = ((fp - cn) < 0).sum() * 100.0 / m
    pos = dists - 1e5 * neg_mask.float()
    pw = torch.clamp((pos - pos_margin).detach(), min=0)
    lpr = torch.logsumexp(log_scale * (pos - pos_margin) * pw, dim=-1)
    lpc = torch.logsumexp(log_scale * (pos - pos_margin) * pw, dim=-2)
    neg = dists + 1e5 * (~neg_mask).float()
    nw = torch.clamp((neg_margin - neg).detach(), min=0)
    lnr = torch.logsumexp(log_scale * (neg_margin - neg) * nw, dim=-1)
    lnc = torch.logsumexp(log_scale * (neg_margin - neg) * nw, dim=-2)
    loss = F.softplus(lpr + lnr) / log_scale + F.softplus(lpc + lnc) / log_scale
    return loss.mean(), acc, fp, avg_neg, dists


def det_loss(dists, anc_score, pos_score):
    """utils/loss.py:149-158."""
    m = dists.shape[0]
    eye = torch.eye(m, dtype=torch.float32)
    fp = (dists * eye).max(dim=1)[0]
    cn = (dists + 1e5 * eye).min(dim=1)[0]
    return ((fp - cn) * (anc_score + pos_score).squeeze(-1)).mean()


def build_correspondence(source_desc, target_desc):
    """geometric_registration/common.py:5-21 (NumPy float32)."""
    with np.errstate(invalid='ignore'):
        distance = np.sqrt(2 - 2 * (source_desc @ target_desc.T))
    source_idx = np.argmin(distance, axis=1)
    target_idx = np.argmin(distance, axis=0)
    keep = target_idx[source_idx] == np.arange(len(source_idx))
    i = np.nonzero(keep)[0]
    return np.stack([i, source_idx[i]], axis=1)


# ---------------------------------------------------------------------------------------------------------------
# batch construction (datasets/dataloader.py:69-189) on the native CPU oracle
# ---------------------------------------------------------------------------------------------------------------
def collate(pts0, pts1, config, limits, native, use_ref=False):
    """The reference's block walk with the CPU checkers; returns the batch dict with NumPy-backed CPU tensors."""
    bq = native.ref_batch_query if use_ref else (lambda q, s, qb, sb, radius: native.batch_query(q, s, qb, sb, radius))
    ss = native.ref_subsample_batch if use_ref else native.subsample_batch

    def search(q, s, qb, sb, r, lim):
        idx = bq(q, s, qb, sb, radius=r)
        return idx[:, :lim] if lim > 0 else idx

    pts = np.concatenate([pts0, pts1], 0).astype(np.float32)
    lens = np.array([len(pts0), len(pts1)], dtype=np.int32)
    r_normal = config.first_subsampling_dl * config.conv_radius
    out = {'points': [], 'neighbors': [], 'pools': [], 'upsamples': [], 'stack_lengths': []}
    layer_blocks, layer = [], 0
    arch = config.architecture
    for bi, block in enumerate(arch):
        if 'global' in block or 'upsample' in block:
            break
        if not ('pool' in block or 'strided' in block):
            layer_blocks.append(block)
            if bi < len(arch) - 1 and 'upsample' not in arch[bi + 1]:
                continue
        conv = search(pts, pts, lens, lens, r_normal, limits[layer]) if layer_blocks else np.zeros((0, 1), np.int64)
        if 'pool' in block or 'strided' in block:
            dl = 2 * r_normal / config.conv_radius
            pp, pb = ss(pts, lens, sampleDl=dl)
            pool = search(pp, pts, pb, lens, r_normal, limits[layer])
            up = search(pts, pp, lens, pb, 2 * r_normal, limits[layer])
        else:
            pp, pb = np.zeros((0, 3), np.float32), np.zeros((0,), np.int32)
            pool, up = np.zeros((0, 1), np.int64), np.zeros((0, 1), np.int64)
        out['points'].append(torch.from_numpy(pts.copy()))
        out['neighbors'].append(torch.from_numpy(np.ascontiguousarray(conv)).long())
        out['pools'].append(torch.from_numpy(np.ascontiguousarray(pool)).long())
        out['upsamples'].append(torch.from_numpy(np.ascontiguousarray(up)).long())
        out['stack_lengths'].append(torch.from_numpy(lens.copy()))
        pts, lens = pp, pb
        r_normal *= 2
        layer += 1
        layer_blocks = []
    return out


# ---------------------------------------------------------------------------------------------------------------
# the network (models/architectures.py:195-320, models/blocks.py:481-686) as a function of a state_dict
# ---------------------------------------------------------------------------------------------------------------
def _unary(sd, prefix, x, relu):
    x = F.linear(x, sd[prefix + '.mlp.weight'], sd[prefix + '.mlp.bias'])
    if prefix + '.batch_norm.bias' in sd:
        x = x + sd[prefix + '.batch_norm.bias']
    return F.leaky_relu(x, 0.1) if relu else x


def kpfcnn_forward(sd, batch, config, training=True):
    """features [N0,32] (unit norm), scores [N0,1] from a reference-layout state_dict (CPU tensors)."""
    arch = list(config.architecture)
    x = batch['features']
    layer, r = 0, config.first_subsampling_dl * config.conv_radius
    skips, enc_i, first_up = [], 0, len(arch)

    def conv(prefix, block, feats, layer, r):
        if 'strided' in block:
            q, s, idx = batch['points'][layer + 1], batch['points'][layer], batch['pools'][layer]
        else:
            q, s, idx = batch['points'][layer], batch['points'][layer], batch['neighbors'][layer]
        ext = r * config.KP_extent / config.conv_radius
        return kpconv(q, s, idx, feats, sd[prefix + '.KPConv.kernel_points'], sd[prefix + '.KPConv.weights'], ext), idx

    for i, block in enumerate(arch):
        if any(t in block for t in ('pool', 'strided', 'upsample', 'global')):
            skips.append(x)
        if 'upsample' in block:
            first_up = i
            break
        p = 'encoder_blocks.%d' % enc_i
        if block.startswith('simple'):
            y, _ = conv(p, block, x, layer, r)
            x = F.leaky_relu(y + sd[p + '.batch_norm.bias'], 0.1)
        else:
            y = _unary(sd, p + '.unary1', x, True) if (p + '.unary1.mlp.weight') in sd else x
            y, idx = conv(p, block, y, layer, r)
            y = F.leaky_relu(y + sd[p + '.batch_norm_conv.bias'], 0.1)
            y = _unary(sd, p + '.unary2', y, False)
            sc = max_pool(x, idx) if 'strided' in block else x
            if (p + '.unary_shortcut.mlp.weight') in sd:
                sc = _unary(sd, p + '.unary_shortcut', sc, False)
            x = F.leaky_relu(y + sc, 0.1)
        enc_i += 1
        if 'pool' in block or 'strided' in block:
            layer += 1
            r *= 2
    skips = skips[:-1] if len(skips) and first_up < len(arch) else skips  # the entry appended for the upsample block
    for j, block in enumerate(arch[first_up:]):
        p = 'decoder_blocks.%d' % j
        if j > 0 and 'upsample' in arch[first_up + j - 1]:
            x = torch.cat([x, skips.pop()], dim=1)
        if 'upsample' in block:
            x = closest_pool(x, batch['upsamples'][layer - 1])
            layer -= 1
            r *= 0.5
        elif block == 'unary':
            x = _unary(sd, p, x, True)
        elif block == 'last_unary':
            x = F.linear(x, sd[p + '.mlp.weight'], sd[p + '.mlp.bias'])
    scores = detection_scores(x, batch['neighbors'][0], training=training)
    return F.normalize(x, p=2, dim=-1), scores
