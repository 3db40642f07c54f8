"""One training step of D3Feat on the device, single- or multi-GPU.

Mirrors the reference's step (trainer.py:89-111): zero grads -> model(inputs) -> gather anchor/positive descriptors
and scores by `corr` (:91-94) -> circle + detector loss (:96-98) -> backward (:103) -> skip optimizer.step() when any
gradient is non-finite (:104-111) -> SGD(lr 0.01, momentum 0.98, weight_decay 1e-6) (training_3DMatch.py:62-76).
Differences that are the point of this build:
  * the batch dict is built on the GPU from the raw pair (datasets/dataloader.py of this package) -- no CPU collate;
  * one fused loss launch produces both losses;
  * parameters and gradients live in ONE flat fp32 buffer each (97 MB at full width): the optimizer is three
    elementwise kernels, the data-parallel exchange is a bucketed RCCL all-reduce of that buffer over xGMI, and the
    NaN/Inf guard is decided from the REDUCED gradients so every rank takes the same decision without a host sync.
One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm.
"""
import torch
import torch.distributed as dist

from .datasets import dataloader as dl
from .models.architectures import KPFCNN
from .utils.loss import CircleLoss


class FlatParams:
    """All trainable parameters (and their gradients) of a module as views into two flat fp32 buffers."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.data = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.data[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.data[off:off + k].view_as(p.data)
            p.grad = self.grad[off:off + k].view_as(p.data)
            off += k
        self.numel = n

    def zero_grad(self):
        self.grad.zero_()
        off = 0
        for p in self.params:  # autograd may have replaced .grad; re-point it at the flat buffer
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad[off:off + k].data_ptr():
                p.grad = self.grad[off:off + k].view_as(p.data)
            off += k


def allreduce_mean_(flat, world_size, n_buckets=4, group=None):
    """Bucketed all-reduce(SUM) / world_size, in place.  xGMI rings are per-link bound, so a few large buckets
    (>= 16 MB each at full width) keep every link busy while bounding the latency of the first bucket."""
    if world_size <= 1:
        return flat
    n = flat.numel()
    step = (n + n_buckets - 1) // n_buckets
    works = []
    for b in range(n_buckets):
        chunk = flat[b * step:min(n, (b + 1) * step)]
        if chunk.numel():
            works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    flat.mul_(1.0 / world_size)
    return flat


class GuardedSGD:
    """SGD with momentum + weight decay on flat buffers; the update is skipped (momentum untouched) when the
    gradient holds a non-finite value -- the reference's guard (trainer.py:104-111) evaluated on the device."""

    def __init__(self, flat, lr=0.01, momentum=0.98, weight_decay=1e-6):
        self.flat, self.lr, self.momentum, self.weight_decay = flat, lr, momentum, weight_decay
        self.buf = torch.zeros_like(flat.data)
        self.skipped = torch.zeros(1, dtype=torch.int32, device=flat.data.device)

    @torch.no_grad()
    def step(self):
        g = self.flat.grad
        ok = torch.isfinite(g).all()
        d = torch.add(g, self.flat.data, alpha=self.weight_decay)       # g + wd * p
        new_buf = torch.add(d, self.buf, alpha=self.momentum)            # mom * buf + d
        self.buf.copy_(torch.where(ok, new_buf, self.buf))
        self.flat.data.sub_(torch.where(ok, new_buf, torch.zeros_like(new_buf)), alpha=self.lr)
        self.skipped += (~ok).to(torch.int32)
        return ok


class TrainStep:
    """Owns model + optimizer state for one rank and runs fragment pairs through the whole hot path."""

    def __init__(self, config, neighborhood_limits, device, world_size=1, seed=0, model=None):
        self.config, self.limits, self.device, self.world = config, [int(x) for x in neighborhood_limits], device, world_size
        if model is None:
            import numpy as np
            np.random.seed(seed)
            torch.manual_seed(seed)
            model = KPFCNN(config)
        self.model = model.to(device).train()
        if world_size > 1:  # identical start on every rank (kernel points are RNG-initialised per process)
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                dist.broadcast(t.data, src=0)
        self.flat = FlatParams(self.model)
        self.opt = GuardedSGD(self.flat, lr=config.lr, momentum=config.momentum, weight_decay=config.weight_decay)
        self.circle = CircleLoss(dist_type='euclidean', log_scale=config.log_scale, safe_radius=config.safe_radius,
                                 pos_margin=config.pos_margin, neg_margin=config.neg_margin)
        self.w_desc, self.w_det = float(config.desc_loss_weight), float(config.det_loss_weight)

    def build_batch(self, item):
        return dl.collate_fn_descriptor([item], self.config, self.limits, device=self.device, exact_width=False)

    def forward_loss(self, batch):
        feats, scores = self.model(batch)
        corr = batch['corr'].long()
        n0 = batch['n0'] if 'n0' in batch else int(batch['stack_lengths'][0][0])
        ia, ip = corr[:, 0], corr[:, 1] + n0
        desc, acc, fp, an, _, dists = self.circle(feats[ia], feats[ip], batch['dist_keypts'], scores[ia], scores[ip])
        det = dists._d3f_det[0][1]
        return desc * self.w_desc + det * self.w_det, desc, det, acc

    # -- pyramid construction on a side stream ---------------------------------------------------------------
    # build_pyramid reads the level sizes back once; done on the training stream that read-back would wait for the
    # whole previous step.  prefetch() runs it on its own stream (the host blocks only on that stream), so the 13
    # searches + 4 voxel levels of pair k+1 overlap the network of pair k.
    def prefetch(self, item):
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._side):  # inputs are long-lived device tensors: no dependency on the main stream
            batch = self.build_batch(item)
            batch['n0'] = int(item[0].shape[0])
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._pending = (batch, ev)

    def step(self, item=None, next_item=None):
        """item = (pts0, pts1, feat0, feat1, sel_corr, dist_keypts), host arrays or device tensors.  With
        ``next_item`` the following pair's pyramid is started on the side stream before this step's network runs."""
        pending = getattr(self, '_pending', None)
        if pending is not None:
            batch, ev = pending
            self._pending = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            for v in batch.values():  # tensors produced on the side stream are consumed on this one
                for t in (v if isinstance(v, list) else [v]):
                    if isinstance(t, torch.Tensor):
                        t.record_stream(torch.cuda.current_stream(self.device))
        else:
            batch = self.build_batch(item)
            batch['n0'] = int(item[0].shape[0])
        if next_item is not None:
            self.prefetch(next_item)
        self.flat.zero_grad()
        loss, desc, det, acc = self.forward_loss(batch)
        loss.backward()
        allreduce_mean_(self.flat.grad, self.world)
        self.opt.step()
        return loss.detach(), desc.detach(), det.detach(), acc.detach()
