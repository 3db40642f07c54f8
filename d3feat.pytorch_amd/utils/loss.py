"""Mirror of the reference's ``utils/loss.py`` on the fused HIP loss kernel.

``CircleLoss`` / ``DetLoss`` keep the reference's constructor arguments and return shapes
(loss.py:100-141: ``(loss, accuracy, furthest_positive.tolist(), average_negative.tolist(), 0, dists)``;
loss.py:144-158: scalar).  In the reference the trainer calls the two modules back to back on the same
``dists`` (trainer.py:96-97); here ``CircleLoss.forward`` runs ONE fused launch that also evaluates the detector
term, and ``DetLoss.forward`` picks that result up when it is handed the ``dists`` object CircleLoss returned and the
scores were registered with :func:`attach_scores` (what ``d3feat_pytorch_amd.train.train_step`` does); otherwise it
launches the fused kernel again.  The two Python lists are produced lazily (`LazyList`) so that no host
synchronisation happens unless a caller actually reads them (the reference's ``.tolist()`` syncs every step).
"""
import torch
import torch.nn as nn

from .. import ops


class LazyList(list):
    """A list whose elements are fetched from the device on first access."""

    def __init__(self, tensor):
        super().__init__()
        self._t = tensor

    def _load(self):
        if self._t is not None:
            t, self._t = self._t, None
            super().extend(t.tolist())

    def __iter__(self):
        self._load()
        return super().__iter__()

    def __len__(self):
        return int(self._t.numel()) if self._t is not None else super().__len__()

    def __getitem__(self, i):
        self._load()
        return super().__getitem__(i)

    def __repr__(self):
        self._load()
        return super().__repr__()

    def device_mean(self):
        """Device-side mean (what trainer.py:99-100 computes with np.mean) without the read-back."""
        return self._t.mean() if self._t is not None else torch.tensor(list(super().__iter__())).mean()


def cdist(a, b, metric='euclidean'):
    """All-pairs distance (reference loss.py:8-44).  Only the matrix itself; the training path never calls this
    (the distance is fused into the loss kernel)."""
    if metric == 'cosine':
        return torch.sqrt(2 - 2 * torch.matmul(a, b.T))
    if metric == 'arccosine':
        return torch.acos(torch.matmul(a, b.T))
    diffs = torch.unsqueeze(a, dim=1) - torch.unsqueeze(b, dim=0)
    if metric == 'sqeuclidean':
        return torch.sum(diffs ** 2, dim=-1)
    if metric == 'euclidean':
        return torch.sqrt(torch.sum(diffs ** 2, dim=-1) + 1e-12)
    if metric == 'cityblock':
        return torch.sum(torch.abs(diffs), dim=-1)
    raise NotImplementedError('The following metric is not implemented by `cdist` yet: {}'.format(metric))


class CircleLoss(nn.Module):
    def __init__(self, dist_type='cosine', log_scale=10, safe_radius=0.10, pos_margin=0.1, neg_margin=1.4):
        super(CircleLoss, self).__init__()
        self.log_scale = log_scale
        self.pos_margin = pos_margin
        self.neg_margin = neg_margin
        self.pos_optimal = pos_margin
        self.neg_optimal = neg_margin
        self.dist_type = dist_type
        self.safe_radius = safe_radius
        if dist_type not in ('euclidean', 'cosine', 'arccosine', 'sqeuclidean', 'cityblock'):
            raise NotImplementedError('The following metric is not implemented by `cdist` yet: {}'.format(dist_type))

    def _forward_other_metric(self, anchor, positive, dist_keypts):
        """dist_type other than the configuration's 'euclidean' (config.py:50): the fused kernel is written for that
        metric; the rest of loss.py:111-141 is metric-agnostic tensor algebra on the [M, M] matrix (M = 128 rows), run
        as such on the device."""
        dists = cdist(anchor, positive, metric=self.dist_type)
        m = dists.shape[0]
        eye = torch.eye(m, dtype=torch.float32, device=dists.device)
        far = (dist_keypts.to(dists.device) > self.safe_radius)
        furthest_positive = (dists * eye).max(dim=1)[0]
        closest_negative = (dists + 1e5 * eye).min(dim=1)[0]
        average_negative = (dists.sum(dim=-1) - furthest_positive) / (m - 1)
        accuracy = ((furthest_positive - closest_negative) < 0).sum() * 100.0 / m
        pos = dists - 1e5 * far.float()
        pos_arg = self.log_scale * (pos - self.pos_margin) * torch.clamp((pos - self.pos_optimal).detach(), min=0)
        neg = dists + 1e5 * (~far).float()
        neg_arg = self.log_scale * (self.neg_margin - neg) * torch.clamp((self.neg_optimal - neg).detach(), min=0)
        by_row = torch.nn.functional.softplus(torch.logsumexp(pos_arg, dim=-1) + torch.logsumexp(neg_arg, dim=-1))
        by_col = torch.nn.functional.softplus(torch.logsumexp(pos_arg, dim=-2) + torch.logsumexp(neg_arg, dim=-2))
        loss = (by_row + by_col) / self.log_scale
        return torch.mean(loss), accuracy, LazyList(furthest_positive), LazyList(average_negative), 0, dists

    def forward(self, anchor, positive, dist_keypts, anc_score=None, pos_score=None):
        if self.dist_type != 'euclidean':
            return self._forward_other_metric(anchor, positive, dist_keypts)
        M = anchor.shape[0]
        zeros = None
        if anc_score is None or pos_score is None:
            zeros = torch.zeros(M, dtype=torch.float32, device=anchor.device)
        scalars, dists, fp, an = ops.circle_det_loss(
            anchor, positive, dist_keypts, zeros if anc_score is None else anc_score,
            zeros if pos_score is None else pos_score, self.log_scale, self.safe_radius, self.pos_margin,
            self.neg_margin)
        if anc_score is not None and pos_score is not None:
            dists._d3f_det = (scalars, anc_score, pos_score)
        dists._d3f_ctx = (anchor, positive, dist_keypts, self.log_scale, self.safe_radius, self.pos_margin,
                          self.neg_margin)
        return scalars[0], scalars[2], LazyList(fp), LazyList(an), 0, dists


class DetLoss(nn.Module):
    def __init__(self, metric='euclidean'):
        super(DetLoss, self).__init__()
        self.metric = metric

    def forward(self, dists, anc_score, pos_score):
        fused = getattr(dists, '_d3f_det', None)
        if fused is not None and fused[1] is anc_score and fused[2] is pos_score:
            return fused[0][1]  # already evaluated by the same launch as the circle loss
        ctx = getattr(dists, '_d3f_ctx', None)
        if ctx is None:
            # a plain distance matrix (CircleLoss with a non-default metric, or a caller's own): loss.py:149-158 as is
            m = dists.shape[0]
            eye = torch.eye(m, dtype=torch.float32, device=dists.device)
            furthest_positive = (dists * eye).max(dim=1)[0]
            closest_negative = (dists + 1e5 * eye).min(dim=1)[0]
            return torch.mean((furthest_positive - closest_negative) * (anc_score + pos_score).squeeze(-1))
        # the reference's call order (desc loss first, det loss second): run the fused kernel again with the scores;
        # only the detector scalar is used, so its backward carries exactly the detector term's gradients.
        anchor, positive, dist_keypts, s, sr, pm, nm = ctx
        scalars, _, _, _ = ops.circle_det_loss(anchor, positive, dist_keypts, anc_score, pos_score, s, sr, pm, nm)
        return scalars[1]


class ContrastiveLoss(nn.Module):
    """Batch-hard contrastive loss (reference loss.py:47-97); not the default (config.py:51) -- plain PyTorch."""

    def __init__(self, pos_margin=0.1, neg_margin=1.4, metric='euclidean', safe_radius=0.25):
        super(ContrastiveLoss, self).__init__()
        self.pos_margin, self.neg_margin, self.metric, self.safe_radius = pos_margin, neg_margin, metric, safe_radius

    def forward(self, anchor, positive, dist_keypts):
        M = anchor.shape[0]
        eye = torch.eye(M, dtype=torch.float32, device=anchor.device)
        dist = cdist(anchor, positive, metric=self.metric)
        near = (dist_keypts.to(anchor.device).double() + 10 * eye.double()) < self.safe_radius
        dists = dist + 10.0 * near.float()
        furthest_positive = torch.max(dists * eye, dim=1)[0]
        closest_negative = torch.min(dists + 1e5 * eye, dim=1)[0]
        diff = furthest_positive - closest_negative
        accuracy = (diff < 0).sum() * 100.0 / diff.shape[0]
        loss = torch.clamp(furthest_positive - self.pos_margin, min=0) + torch.clamp(self.neg_margin - closest_negative,
                                                                                    min=0)
        average_negative = (torch.sum(dists, dim=-1) - furthest_positive) / (M - 1)
        return torch.mean(loss), accuracy, LazyList(furthest_positive), LazyList(average_negative), 0, dists
