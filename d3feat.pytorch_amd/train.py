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
import os

import torch
import torch.distributed as dist

from . import ops
from .datasets import dataloader as dl
from .models.architectures import KPFCNN
from .utils.loss import CircleLoss


# Other threads of the process (the RCCL watchdog of torch.distributed polls events) must not invalidate a capture
# that is under way on this thread.
_CAPTURE_MODE = "thread_local"


class FlatParams:
    """All trainable parameters of a module as views into one flat fp32 buffer, plus a flat gradient buffer.

    Gradients are NOT accumulated into the flat buffer by autograd.  The owner clears ``p.grad`` before every backward
    (``zero_grad``); the weight-gradient kernels of this package then write straight into the parameter's place in the
    flat gradient buffer (``_d3f_grad_slot``, picked up by ops) and autograd adopts that view as ``p.grad``; whatever
    else the engine produced (bias gradients, library-computed ones) is moved in by one multi-tensor copy
    (``gather_grads``).  No per-parameter accumulation launches, no 97 MB concatenation.
    Contract: exactly one backward between ``zero_grad`` calls (the slots are overwritten, not accumulated)."""

    def __init__(self, module):
        every = list(module.parameters())
        self.params = [p for p in every if p.requires_grad]
        # position of each trainable parameter in module.parameters() -- the index torch.optim uses in its state_dict
        self.module_index = [i for i, p in enumerate(every) if p.requires_grad]
        self.n_module_params = len(every)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.data = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.slots = []
        off = 0
        for p in self.params:
            k = p.numel()
            self.data[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.data[off:off + k].view_as(p.data)
            p.grad = None
            slot = self.grad[off:off + k].view_as(p.data)
            self.slots.append(slot)
            if p.dim() >= 2 and dev.type == "cuda":  # weight matrices / KPConv kernels: written in place by ops
                p._d3f_grad_slot = slot
            off += k
        self.numel = n
        self.lanes = [(self.grad, self.slots)]

    def add_lane(self):
        """A further gradient buffer over the same parameters: one per pair in flight on this GPU (PairLanes).  Returns
        its index for ``bind``."""
        grad = torch.zeros_like(self.grad)
        slots, off = [], 0
        for p in self.params:
            slots.append(grad[off:off + p.numel()].view_as(p.data))
            off += p.numel()
        self.lanes.append((grad, slots))
        return len(self.lanes) - 1

    def bind(self, lane):
        """Backward passes run (or captured) from here on leave their gradients in buffer ``lane``."""
        self.grad, self.slots = self.lanes[lane]
        for p, slot in zip(self.params, self.slots):
            if hasattr(p, '_d3f_grad_slot'):
                p._d3f_grad_slot = slot

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def gather_grads(self, first=0, last=None):
        """Make ``grad`` hold the gradients of the last backward for parameters ``first:last`` (a gradient bucket):
        in-place ones are already there, the others are moved by one multi-tensor copy, unused parameters get zeros."""
        ps, slots = self.params[first:last], self.slots[first:last]
        src, dst = [], []
        for p, slot in zip(ps, slots):
            if p.grad is None:
                slot.zero_()
            elif p.grad.data_ptr() != slot.data_ptr():
                src.append(p.grad.reshape(slot.shape))
                dst.append(slot)
        if dst:
            torch._foreach_copy_(dst, src)
        a = sum(p.numel() for p in self.params[:first])
        return self.grad[a:a + sum(p.numel() for p in ps)]


def allreduce_mean_(flat, world_size, n_buckets=4, group=None, average=True):
    """Bucketed all-reduce(SUM) (/ world_size unless ``average=False``), in place.  xGMI rings are per-link bound, so a few large buckets
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
    if average:
        flat.mul_(1.0 / world_size)
    return flat


class GuardedSGD:
    """SGD with momentum + weight decay on flat buffers; the update is skipped (momentum untouched) when the
    gradient holds a non-finite value -- the reference's guard (trainer.py:104-111) evaluated on the device.
    On the GPU this is d3f_sgd_guarded_step (two launches, no host sync); the torch expression below is the same
    arithmetic for host tensors (the gloo tests of the data-parallel logic)."""

    def __init__(self, flat, lr=0.01, momentum=0.98, weight_decay=1e-6):
        self.flat = flat
        self.buf = torch.zeros_like(flat.data)
        # {scratch, skipped steps, OR of pair-status flags that caused a skip, number of those skips}
        self.state = torch.zeros(4, dtype=torch.int32, device=flat.data.device)
        # {lr, momentum, weight_decay} live on the device: the kernel reads them when it runs, so a schedule changes the
        # step size of an already captured graph (a scalar argument would be frozen at capture)
        self.hyper = torch.tensor([lr, momentum, weight_decay, 1.0], dtype=torch.float32, device=flat.data.device)
        self._hyper = [float(lr), float(momentum), float(weight_decay), 1.0]
        self.initial_lr = float(lr)

    def _set(self, i, v):
        self._hyper[i] = float(v)
        self.hyper[i] = float(v)   # stream-ordered fill, no sync

    lr = property(lambda self: self._hyper[0], lambda self, v: self._set(0, v))
    momentum = property(lambda self: self._hyper[1], lambda self, v: self._set(1, v))
    weight_decay = property(lambda self: self._hyper[2], lambda self, v: self._set(2, v))
    # the gradient is multiplied by this first: 1/world_size makes the mean out of an all-reduced SUM inside the step
    grad_scale = property(lambda self: self._hyper[3], lambda self, v: self._set(3, v))

    def use_grad_scale(self, v):
        """Make ``v`` the gradient scale of the steps launched from here on: 1 / (pairs whose gradients are summed into
        one update, over all lanes, stacked pairs and ranks).  Every engine asks for ITS scale before it launches a step,
        so engines with different pair counts (a stacked graph engine, the eager single-pair fallback) can share one
        optimizer.  The value lives on the device and is read by the update kernel when it runs: a change first drains
        the device (an update still in flight on another stream must see the old value) -- rare, engines alternate only
        when a pair leaves the graph path."""
        v = float(v)
        if self._hyper[3] == v:
            return
        if self.hyper.is_cuda:
            torch.cuda.synchronize(self.hyper.device)
        self.grad_scale = v
        if self.hyper.is_cuda:
            torch.cuda.current_stream(self.hyper.device).synchronize()

    @property
    def param_groups(self):
        """Read-only view in torch.optim's shape (the reference reads ``param_groups[0]['lr']``, trainer.py:224)."""
        return [{'lr': self.lr, 'momentum': self.momentum, 'dampening': 0, 'weight_decay': self.weight_decay,
                 'nesterov': False, 'initial_lr': self.initial_lr, 'params': list(range(self.flat.n_module_params))}]

    def state_dict(self):
        """Same layout as ``torch.optim.SGD.state_dict()`` over ``model.parameters()`` (what the reference snapshots,
        trainer.py:196): momentum buffers keyed by the parameter's position in ``model.parameters()``."""
        state, off = {}, 0
        for idx, p in zip(self.flat.module_index, self.flat.params):
            state[idx] = {'momentum_buffer': self.buf[off:off + p.numel()].view_as(p).detach().clone()}
            off += p.numel()
        return {'state': state, 'param_groups': self.param_groups}

    def load_state_dict(self, sd):
        group = sd['param_groups'][0]
        if len(sd['param_groups']) != 1 or group.get('dampening', 0) != 0 or group.get('nesterov', False):
            raise ValueError("GuardedSGD loads a single-group torch.optim.SGD state without dampening / nesterov")
        if len(group['params']) != self.flat.n_module_params:
            raise ValueError("optimizer state covers %d parameters, the model has %d" % (
                len(group['params']), self.flat.n_module_params))
        self.lr, self.momentum, self.weight_decay = group['lr'], group['momentum'], group['weight_decay']
        self.initial_lr = float(group.get('initial_lr', self.initial_lr))
        slot, off = {}, 0
        for idx, p in zip(self.flat.module_index, self.flat.params):
            slot[idx] = (off, p)
            off += p.numel()
        self.buf.zero_()   # a parameter without state has not been stepped yet: first step sets buf = d = 0*m + d
        for key, st in sd['state'].items():
            mb = st.get('momentum_buffer')
            if mb is None:
                continue
            if int(key) not in slot:
                raise ValueError("optimizer state for parameter %s, which is not trainable here" % key)
            off, p = slot[int(key)]
            if tuple(mb.shape) != tuple(p.shape):
                raise ValueError("momentum buffer %s has shape %s, parameter has %s" % (key, tuple(mb.shape), tuple(p.shape)))
            self.buf[off:off + p.numel()].copy_(mb.reshape(-1))

    @property
    def skipped(self):
        return self.state[1]

    @torch.no_grad()
    def step(self, want_ok=True, pair_status=None, grads=None):
        """Returns a 0-dim bool tensor: True when the update was applied (None with want_ok=False: the training
        step only consults the skipped-step counter).  ``grads``: the gradient buffers of several pairs in flight
        (PairLanes) -- the step uses their sum.  ``pair_status``: device status word (int32[1]) of the pair the
        gradient came from; when it is set the update is skipped like for a non-finite gradient (a pyramid that
        outgrew a graph capacity, say, never reaches the parameters) and the flags are kept in ``state[2:4]``."""
        g = self.flat.grad if grads is None else list(grads)
        on_gpu = (g if grads is None else g[0]).is_cuda
        if grads is not None and not on_gpu:    # host tensors (gloo tests): the same fixed-order sum
            total = g[0].clone()
            for other in g[1:]:
                total += other
            g = total
        if on_gpu:
            before = self.state[1].clone() if want_ok else None
            ops.sgd_guarded_step(g, self.flat.data, self.buf, self.lr, self.momentum, self.weight_decay, self.state,
                                 hyper=self.hyper, pair_status=pair_status)
            return (self.state[1] == before) if want_ok else None
        ok = torch.isfinite(g).all()
        if pair_status is not None:
            bad = pair_status.reshape(-1)[0] != 0
            self.state[2] |= pair_status.reshape(-1)[0].to(self.state.dtype)
            self.state[3] += bad.to(self.state.dtype)
            ok = ok & ~bad
        if self.grad_scale != 1.0:
            g = g * self.grad_scale
        d = torch.add(g, self.flat.data, alpha=self.weight_decay)       # g + wd * p
        new_buf = torch.add(d, self.buf, alpha=self.momentum)            # d + mom * buf
        self.buf.copy_(torch.where(ok, new_buf, self.buf))
        self.flat.data.sub_(torch.where(ok, new_buf, torch.zeros_like(new_buf)), alpha=self.lr)
        self.state[1] += (~ok).to(torch.int32)
        return ok


KEEP_GRAPH_TEMPLATES = True


def new_graph():
    """A torch.cuda.CUDAGraph whose hipGraph_t template is KEPT next to the instantiated executable graph.

    Round 5 (profiles/memset_node_repro.py, memset_node_fix_experiment.py): on this HIP runtime a memset node of an
    executable graph keeps reading its fill value from the TEMPLATE graph's node, and PyTorch destroys the template right
    after instantiation (keep_graph=False, the default) -- from the second replay on a small hipMemsetAsync captured into
    a graph writes 0xA0 bytes (freed memory) instead of its value; with the template alive it replays correctly.  Nothing
    in this package captures a memset (every clear is a kernel, csrc/common.hpp), but a library call inside a captured
    step may.  (This does NOT cure the library-GEMM solutions that never finish on a later replay -- hipBLASLt winners,
    the default pick of one 4 x 3 stacked shape: they hang with the templates kept as well, profiles/r05_hipblaslt_hang.txt;
    the engine captures TunableOp-selected rocBLAS solutions.  Round 6 found that cause: one BLAS handle behind all lanes'
    graphs, profiles/r06_stall_root_cause.txt.)  KEEP_GRAPH_TEMPLATES = False: PyTorch's default form (experiments)."""
    if not KEEP_GRAPH_TEMPLATES:
        return torch.cuda.CUDAGraph()
    try:
        return torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError:      # (an older PyTorch without the option)
        return torch.cuda.CUDAGraph()


def finish_graph(g):
    """After the capture: a kept-template graph is instantiated explicitly (the default form does it in capture_end)."""
    inst = getattr(g, 'instantiate', None)
    if inst is not None and KEEP_GRAPH_TEMPLATES:
        try:
            inst()
        except RuntimeError:       # (already instantiated: the default form)
            pass
    return g


_WARM = {}


def _warm_stream(device):
    """One warm-up stream per device for every capture of the process (each used stream holds a hardware queue)."""
    key = str(device)
    if key not in _WARM:
        _WARM[key] = torch.cuda.Stream(device=device)
    return _WARM[key]


def fresh_streams(n, device):
    """``n`` HIP streams whose hardware queues are created HERE, one after the other.

    Streams that are busy at the same time must not share a compute pipe of the GPU: the four pipes of an XCD's
    dispatcher each work through the queues mapped to them one dispatch at a time, and the driver deals queues to pipes
    round-robin in the order the queues come into existence -- which for a HIP stream is its first use, not its
    creation.  Measured (profiles/r03_queue_pipes.txt): two pairs in flight run at 361 pairs/s when the two training
    streams and the two pyramid streams sit on four different pipes and at 290-304 when a pyramid stream shares a pipe
    with a training stream; which of the two a process got depended on how many other streams had been used before.
    Touching the streams back to back gives them consecutive queues, hence different pipes for any four of them."""
    streams = [torch.cuda.Stream(device=device) for _ in range(n)]
    for s in streams:
        with torch.cuda.stream(s):
            torch.zeros(1, device=device).add_(1)       # the first launch on the stream brings its queue to life
        s.synchronize()
    return streams


def lane_streams(lanes, device):
    """(network streams, pyramid streams) for ``lanes`` pairs in flight on one GPU.

    The lanes' network streams sit on different compute pipes (fresh_streams); what is left of the four pipes goes to
    the pyramid builds: two lanes have a pyramid stream each (361 pairs/s in training), three share one (their 0.6-ms
    builds run back to back under a 7-ms step: 414), and with four lanes every pipe runs a network and each lane builds
    its next pyramid on its OWN stream, behind its network graph (434) -- a stream that SHARES a pipe with another busy
    stream is far worse than either (a fifth stream for the pyramids: 256).  Never more streams than that: every stream
    that has run a kernel keeps a hardware queue, and a process with more queues than the GPU has slots for gets them
    time-sliced (profiles/r03_queue_pipes.txt)."""
    if not 1 <= int(lanes) <= 4:
        raise ValueError("1..4 pairs in flight (the dispatcher has four compute pipes: one per network stream), got %r"
                         % (lanes,))
    streams = fresh_streams(min(2 * lanes, 4), device)
    nets = streams[:lanes]
    sides = streams[lanes:] if lanes <= 2 else [streams[3]] * lanes if lanes == 3 else nets
    return nets, sides


def is_stack(item):
    """A stack of pairs (tuple of dataset items) rather than one dataset item (tuple of arrays)."""
    return isinstance(item, (tuple, list)) and len(item) > 0 and isinstance(item[0], (tuple, list))


def same_item(a, b):
    """Identity of a step's input: the same item object, or stacks of the same item objects in the same order."""
    if a is b:
        return True
    if a is None or b is None or not (is_stack(a) and is_stack(b)) or len(a) != len(b):
        return False
    return all(x is y for x, y in zip(a, b))


class TrainStep:
    """Owns model + optimizer state for one rank and runs fragment pairs through the whole hot path.

    ``stack`` (``enable_graph(..., stack=Q)``) > 1: the static-shape step trains on Q fragment pairs STACKED into one
    batch of 2Q clouds -- one pyramid, one network forward / backward, one loss launch sequence for all of them, the
    gradient is the sum of the pairs' gradients and the optimizer's gradient scale 1 / (Q * ranks) makes their mean:
    the update a data-parallel step over Q times as many ranks makes.  The reference refuses batches of more than one
    pair (datasets/dataloader.py:73); what a reference batch of ONE pair shares stays per pair: the neighbor-table widths
    (dataloader.py:64-66), the detector's feature normaliser (architectures.py:342), the loss's M x M problem
    (trainer.py:91-98).  A stacked step's input is a tuple of Q dataset items."""

    stack = 1

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
        self._init_training_state(config, world_size)

    def _init_training_state(self, config, world_size):
        self.flat = FlatParams(self.model)
        self.opt = GuardedSGD(self.flat, lr=config.lr, momentum=config.momentum, weight_decay=config.weight_decay)
        self.opt.grad_scale = 1.0 / max(1, world_size)
        self.circle = CircleLoss(dist_type='euclidean', log_scale=config.log_scale, safe_radius=config.safe_radius,
                                 pos_margin=config.pos_margin, neg_margin=config.neg_margin)
        self.w_desc, self.w_det = float(config.desc_loss_weight), float(config.det_loss_weight)
        # Gradient buckets for the data-parallel exchange.  Backward visits decoder -> coarse encoder levels -> fine
        # encoder levels; 97% of the gradient BYTES belong to the first two, while the fine levels (encoder blocks
        # 0..CUT-1, 38k/8k points) still have ~1/3 of the backward's run time ahead of them.  With split_backward the
        # network is cut at the input of encoder block CUT: the "deep" bucket (blocks CUT.. + decoder) is all-reduced
        # over xGMI while the backward of the fine levels runs.
        self.split_backward = world_size > 1 or self.SPLIT_BACKWARD_ON_ONE_RANK
        self.CUT = 5
        shallow = [p for b in list(self.model.encoder_blocks)[:self.CUT] for p in b.parameters() if p.requires_grad]
        self.n_shallow = len(shallow)
        assert all(a is b for a, b in zip(shallow, self.flat.params[:self.n_shallow])), "parameter order"
        self.numel_shallow = sum(p.numel() for p in shallow)

    ITEM_DTYPES = (torch.float32, torch.float32, torch.float32, torch.float32, torch.int64, torch.float64)

    def upload(self, item):
        """Host item (NumPy arrays as a dataset returns them) -> device tensors in the dtypes of the step's buffers.

        The copy runs on a stream of its own and the HOST waits for it: the pipelined step reads the next pair on its
        side stream, which does not wait for the training stream (step_graph), and a copy queued on the training
        stream would sit behind a whole network step.  ~0.6 MB per pair."""
        if all(isinstance(t, torch.Tensor) and t.device == self.device for t in item):
            return item
        if self.device.type != 'cuda':
            return tuple(torch.as_tensor(t).to(dtype=k) for t, k in zip(item, self.ITEM_DTYPES))
        import numpy as np
        if getattr(self, '_h2d', None) is None:
            self._h2d = torch.cuda.Stream(device=self.device)
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._h2d):
            out = tuple(torch.as_tensor(np.ascontiguousarray(t) if isinstance(t, np.ndarray) else t).to(
                device=self.device, dtype=k) for t, k in zip(item, self.ITEM_DTYPES))
        self._h2d.synchronize()
        for t in out:   # consumed on other streams than the one that allocated them
            t.record_stream(torch.cuda.current_stream(self.device))
            t.record_stream(self._side)
        return out

    # training builds the transposes of the KPConv tables with the pyramid (gather-form grad-input); inference does not
    reverse_tables = True
    # the captured / static-shape pyramids carry their upsampling tables in the prefix form (column 0 + the part within
    # the pooling radius: all the step reads; datasets.dataloader._pool_tables).  False: the full 2 r rows (A/B experiments)
    engine_upsamples = True
    # the two-stage backward of the multi-rank step on ONE rank (what its cut costs without an exchange: measurements)
    SPLIT_BACKWARD_ON_ONE_RANK = False
    # one launch for a step's input loads (ops.copy_buffers); False: one copy_() per tensor (A/B experiments)
    FUSED_INPUT_LOAD = True

    def build_batch(self, item):
        batch = dl.collate_fn_descriptor([item], self.config, self.limits, device=self.device, exact_width=False,
                                         reverse_tables=self.reverse_tables, keep_status=True)
        # flags the searches raised after the level-size read-back (candidate overflow, ...): checked at the caller's
        # next check_status() instead of being dropped
        pend = getattr(self, '_eager_status', None)
        if pend is None:
            pend = self._eager_status = []
        status = batch.pop('_status')
        pend.append(status)
        del pend[:-64]
        # ... and the word itself rides with the batch: the eager step gates its update on it exactly like the graph step
        # does on its set's word (a truncated search-form transpose must not reach the parameters)
        batch['_pair_status'] = status.word
        return batch

    def _loss_from_raw(self, x, scores, batch):
        """Reference trainer.py:91-98 on the un-normalised descriptors: the 2M sampled rows are gathered and
        normalised by one launch (the other rows never enter the loss)."""
        c = self.circle
        if batch.get('_pairs', 1) > 1:     # stacked pairs: every pair's own M x M problem, total = their sum
            total, desc, det, acc, fp, an = ops.train_loss_pairs(
                x, scores, batch['corr'], batch['stack_lengths'][0], None, c.log_scale, c.safe_radius, c.pos_margin,
                c.neg_margin, self.w_desc, self.w_det, neg_mask=batch['neg_mask'])
            self.last_distances = (fp, an)                     # [Q, M] each
            self.last_pair_losses = (desc, det, acc)           # [Q] each
            return total, desc, det, acc
        n0 = batch['n0'] if 'n0' in batch else batch['stack_lengths'][0][:1]  # host int, or a device scalar (no sync)
        loss, desc, det, acc, fp, an = ops.train_loss(x, scores, batch['corr'], n0, batch['dist_keypts'], c.log_scale,
                                                      c.safe_radius, c.pos_margin, c.neg_margin, self.w_desc, self.w_det,
                                                      neg_mask=batch.get('neg_mask'))
        # per-row furthest-positive / average-negative distances [M] (trainer.py:99-100 averages them on the host);
        # kept on the device for whoever wants the statistics -- no extra launches in the step itself
        self.last_distances = (fp, an)
        return loss, desc, det, acc

    @torch.no_grad()
    def evaluate(self, item):
        """Validation pass on one pair (trainer.py:164-176): eval-mode forward (the detector applies its local-maximum
        mask, architectures.py:361-366) + both losses, no backward.  Returns device scalars
        (loss, desc_loss, det_loss, accuracy, d_pos, d_neg)."""
        self.model.eval()
        try:
            batch = self.build_batch(item)
            batch['n0'] = int(item[0].shape[0])
            loss, desc, det, acc = self.forward_loss(batch)
            fp, an = self.last_distances
            return loss, desc, det, acc, fp.mean(), an.mean()
        finally:
            self.model.train()

    def _seed(self, loss):
        """d loss / d loss = 1 as a cached device scalar (loss.backward() would fill a fresh one every step)."""
        if getattr(self, '_one', None) is None or self._one.device != loss.device:
            self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
        return self._one

    def forward_loss(self, batch):
        x, scores = self.model.forward_raw(batch)
        return self._loss_from_raw(x, scores, batch)

    def _forward_loss_cut(self, batch):
        """forward_loss with the autograd graph cut at the input of encoder block CUT: everything downstream (coarse
        encoder, decoder, loss) hangs off detached leaves.  Returns the loss tuple and [(tensor, leaf), ...]."""
        m = self.model
        x = batch['features'].detach()
        skips, cuts = [], []
        for i, op in enumerate(m.encoder_blocks):
            if i == self.CUT:
                def leaf_of(t):
                    leaf = t.detach().requires_grad_(True)
                    cuts.append((t, leaf))
                    return leaf
                x = leaf_of(x)
                skips = [leaf_of(t) for t in skips]
            if i in m.encoder_skips:
                skips.append(m.mark_skip(x, op))
            x = op(x, batch)
        x = m._decode(x, skips, batch)
        scores = m.detection_scores(batch, x)
        return self._loss_from_raw(x, scores, batch), cuts

    def _lane_stages(self, st):
        """(stage 1, stage 2) of a LANE's split step on the set's pair(s): gradients into the lane's own buffer, no
        exchange and no optimizer step (the join owns both); a flagged pair poisons the deep bucket at the end of
        stage 1, before the join sums and exchanges it."""
        lane = self.lane

        def stage1():
            self.flat.bind(lane)
            try:
                out = self._backward_deep(self._set_batch(st))
                self._poison_if_flagged(self.flat.grad[self.numel_shallow:], st.status.word)
            finally:
                self.flat.bind(0)
            return out

        def stage2():
            self.flat.bind(lane)
            try:
                self._backward_shallow()
            finally:
                self.flat.bind(0)
        return stage1, stage2

    def _backward_deep(self, batch):
        """Stage 1: forward + backward of the deep bucket; leaves the cut gradients in self._cuts."""
        self.flat.zero_grad()
        (loss, desc, det, acc), cuts = self._forward_loss_cut(batch)
        deep = self.flat.params[self.n_shallow:]
        with ops.weight_grad_group():     # the deep bucket's weight gradients: ONE grouped launch at the stage's end,
            torch.autograd.backward(loss, self._seed(loss), inputs=deep + [leaf for _, leaf in cuts])   # before its all-reduce
        self._cuts = cuts
        self.flat.gather_grads(self.n_shallow, None)
        return loss.detach(), desc.detach(), det.detach(), acc.detach()

    def _backward_shallow(self):
        """Stage 2: backward of the fine encoder levels from the cut gradients."""
        cuts, self._cuts = self._cuts, None
        with ops.weight_grad_group():
            torch.autograd.backward([t for t, _ in cuts], [leaf.grad for _, leaf in cuts],
                                    inputs=self.flat.params[:self.n_shallow])
        self.flat.gather_grads(0, self.n_shallow)

    def _poison_if_flagged(self, grad, pair_status):
        """A flagged pair makes its gradient non-finite before the exchange (d3f_poison_gradient_if_status) and parks
        the flags in the optimizer state: the guard on the REDUCED gradient then skips the step on every rank."""
        ops.poison_gradient_if_status(grad, pair_status, self.opt.state)

    def _exchange_and_step(self, after_deep, after_shallow, pair_status=None):
        """after_deep(): runs/launches stage 1; after_shallow(): stage 2.  The deep bucket's all-reduce is in flight
        while stage 2 executes.  ``pair_status``: this rank's pair status word -- with several ranks a flagged pair
        poisons its gradient BEFORE the exchange, so the guard on the reduced gradient skips the step on every rank."""
        out = after_deep()
        g = self.flat.grad
        works = []
        if self.world > 1 and pair_status is not None:
            self._poison_if_flagged(g[self.numel_shallow:], pair_status)
            pair_status = None
        if self.world > 1:
            deep = g[self.numel_shallow:]
            nb = 3
            step = (deep.numel() + nb - 1) // nb
            for b in range(nb):
                chunk = deep[b * step:min(deep.numel(), (b + 1) * step)]
                if chunk.numel():
                    works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))
        after_shallow()
        if self.world > 1:
            works.append(dist.all_reduce(g[:self.numel_shallow], op=dist.ReduceOp.SUM, async_op=True))
            for w in works:
                w.wait()                 # SUM over ranks; the 1/world of the mean is opt.grad_scale
        self.opt.step(want_ok=False, pair_status=pair_status)
        return out

    # -- static shapes + hipGraph ---------------------------------------------------------------------------
    # The eager step costs ~10 ms of host enqueue time (600+ launches) against ~7 ms of GPU work.  In graph mode
    # every level of the pyramid has a fixed row CAPACITY (live counts stay on the device, see
    # build_pyramid_static), so the whole step has static shapes and addresses and is replayed with ONE launch.
    # The step is software-pipelined over two streams,
    #     training stream:  zero grads -> network forward -> fused loss -> backward -> optimizer  on pair k   (set i)
    #     side stream:      4 voxel levels + 13 radius searches                                   of pair k+1 (set 1-i)
    # so the latency-bound pyramid kernels (2-workgroup voxel ordering, hash builds) run UNDER the network's
    # MFMA/HBM-bound kernels instead of in front of them.  (Two branches inside ONE graph were measured to execute
    # back to back, hence separate graphs on separate streams; see step_graph for the ordering.)
    @staticmethod
    def capacities_for(level_sizes, slack=1.06):
        """Per-level row capacities from observed level sizes [[N0, N1, ...], ...] (rounded up to 64 rows)."""
        n = len(level_sizes[0])
        return [int(-(-int(max(s[l] for s in level_sizes) * slack + 32) // 64) * 64) for l in range(n)]

    NSETS = 3

    class _Set:
        """Inputs of one pair (static addresses) + its capacity-shaped pyramid."""

        def __init__(self, caps, num_corr, dev, stack=1):
            self.pts = torch.zeros((caps[0], 3), dtype=torch.float32, device=dev)
            self.feat = None      # [caps[0], in_features_dim]: the pair's input features (set by enable_graph)
            self.lens = torch.zeros(2 * stack, dtype=torch.int32, device=dev)
            # stack lengths travel through pinned memory: a copy from pageable memory blocks the host until everything
            # queued on the stream before it has run -- with four lanes that stream holds the lane's network graph
            # (measured: the step stalled for good, profiles/r04_notes.txt)
            self.lens_host = torch.zeros(2 * stack, dtype=torch.int32).pin_memory() \
                if (stack > 1 and torch.device(dev).type == 'cuda') else None
            self.lens_ev = None
            lead = (stack,) if stack > 1 else ()      # stacked pairs: every pair's own table / matrix
            self.corr = torch.zeros(lead + (num_corr, 2), dtype=torch.int64, device=dev)
            self.dk = torch.zeros(lead + (num_corr, num_corr), dtype=torch.float64, device=dev)
            self.mask = torch.zeros(lead + (num_corr, num_corr), dtype=torch.uint8, device=dev)   # dk > safe_radius
            self.batch = None     # persistent pyramid tensors (filled by the side branch of the OTHER graph)
            self.status = None
            self.loaded = None    # the item whose pyramid `batch` holds
            # host mirror of the status word after the pair's network step (async copy queued behind the step) and the
            # pair it belongs to: a pair whose update the optimizer skipped (capacity overflow) can be re-run eagerly
            self.status_host = torch.zeros(1, dtype=torch.int32).pin_memory() if torch.device(dev).type == 'cuda' else None
            self.status_item = None

    def enable_graph(self, capacities, num_corr, stack=None):
        """Static shapes for the captured step: per-level row capacities (of the whole stack), correspondences per
        pair, and ``stack`` = fragment pairs per step (default: as before, 1 unless set)."""
        if stack is not None:
            if not 1 <= int(stack) <= 16:
                raise ValueError("1..16 stacked pairs per step, got %r" % (stack,))
            self.stack = int(stack)
        if getattr(self.config, 'use_batch_norm', False) and self.model.training:
            # capacity-shaped levels carry ~10 % zero rows: batch statistics over them are not the reference's
            # BatchNorm1d over the live points (the blocks do not hand the live row count to the normalisation)
            raise RuntimeError("use_batch_norm=True trains on the eager path only: the captured graphs run on "
                               "capacity-padded levels, which would bias the batch statistics")
        dev = self.device
        self.caps = [int(c) for c in capacities]
        self.sets = [TrainStep._Set(self.caps, num_corr, dev, self.stack) for _ in range(self.NSETS)]
        fdim = int(getattr(self.config, 'in_features_dim', 1))
        for st in self.sets:    # input features at static addresses (the reference feeds ones, or zeros under
            st.feat = torch.ones((self.caps[0], fdim), dtype=torch.float32, device=dev)   # self_augment)
        self.graphs = None
        self.cur = 0

    NO_PREFETCH = object()   # step_graph(item, NO_PREFETCH): do not build any pyramid for the following step

    def clone_for_capacities(self, capacities, num_corr, stack=None):
        """A second engine over the SAME model, flat buffers, optimizer and streams with its own static buffer sets and
        graphs for other level capacities -- one per size class of the dataset (trainer.Trainer): real 3DMatch pairs
        vary several-fold in size, and a single capacity set makes every small pair pay for the largest.  ``stack``:
        pairs per step of the clone (default: this engine's)."""
        import copy
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        if getattr(self, '_h2d', None) is None and self.device.type == 'cuda':
            self._h2d = torch.cuda.Stream(device=self.device)
        other = copy.copy(self)
        for name in ('sets', 'graphs', 'g_net', 'g_net_b', 'g_pyr', '_graph_out', '_graph_dist', 'ev_net', 'ev_pyr',
                     '_pending', '_cuts', 'caps', 'cur', '_overflowed'):
            other.__dict__.pop(name, None)
        other.enable_graph(capacities, num_corr, stack=stack)
        return other

    def clone_for_lane(self, lane, stream=None, side=None, split=None):
        """An engine over the SAME model, parameters and optimizer whose network step leaves its gradient in buffer
        ``lane`` of the flat parameters and does NOT step the optimizer, with streams, static buffer sets and graphs of
        its own: one of the pairs in flight of ``PairLanes``."""
        import copy
        other = copy.copy(self)
        for name in ('sets', 'graphs', 'g_net', 'g_net_b', 'g_pyr', '_graph_out', '_graph_dist', 'ev_net', 'ev_pyr',
                     '_pending', '_cuts', 'caps', 'cur', '_overflowed', '_eager_status'):
            other.__dict__.pop(name, None)
        while len(self.flat.lanes) <= lane:
            self.flat.add_lane()
        other.lane = int(lane)
        # several ranks: the lane's backward is cut like the one-pair engine's (stage 1 down to encoder block CUT, stage 2
        # the fine levels) so that the join can exchange the deep gradient bucket under every lane's stage 2
        # (PairLanes.step_graph); one rank: a single network graph per lane
        other.split_backward = (self.world > 1) if split is None else bool(split)
        if stream is None or side is None:
            stream, side = fresh_streams(2, self.device)
        other.stream, other._side = stream, side
        if getattr(self, '_h2d', None) is None and self.device.type == 'cuda':
            self._h2d = torch.cuda.Stream(device=self.device)    # uploads: shared by the lanes (copy engines)
        return other

    def _collect_status(self, i):
        """Set i is about to be reused (the caller has waited for its last network step): if that pair was flagged --
        the optimizer then skipped its update -- remember it for take_overflowed()."""
        st = self.sets[i]
        item = getattr(st, 'status_item', None)
        if item is not None and st.status_host is not None and getattr(self, 'opt', None) is not None:
            flags = int(st.status_host[0])
            if flags:
                self.__dict__.setdefault('_overflowed', []).append((item, flags))
        st.status_item = None

    def take_overflowed(self, drain=False):
        """[(item, flags)] of pairs whose graph step was skipped because a level outgrew its capacity (or a search
        overflowed), oldest first; ``drain`` also waits for and checks the sets still in flight (end of an epoch)."""
        if drain and getattr(self, 'ev_net', None) is not None:
            for i in range(len(self.sets)):
                self.ev_net[i].synchronize()
                self._collect_status(i)
        out, self._overflowed = self.__dict__.get('_overflowed', []), []
        return out

    @staticmethod
    def pairs_of(item):
        """The dataset items of a step's input (a stack's members, or the one pair)."""
        return list(item) if is_stack(item) else [item]

    def preload(self, item):
        """Build ``item``'s pyramid on the side stream into the set this engine trains on next (used when the previous
        pair belonged to another size class, whose step_graph cannot prefetch into this engine's sets)."""
        i = self.cur
        st = self.sets[i]
        if same_item(st.loaded, item):
            return
        self.ev_net[i].synchronize()
        self._collect_status(i)
        with torch.cuda.stream(self._side):
            self._load_inputs(st, item)
            self.g_pyr[i].replay()
            self.ev_pyr[i].record(self._side)
        st.loaded = item

    def fits(self, item):
        """Whether the pair (the stack of pairs) can go through the captured graphs (level-0 capacity and correspondence
        count; deeper levels are checked on the device, D3F_ST_CAPACITY -> check_status)."""
        pairs = self.pairs_of(item)
        if len(pairs) != self.stack or is_stack(item) != (self.stack > 1):
            return False
        m = tuple(self.sets[0].corr.shape[-2:])
        return (sum(int(it[0].shape[0]) + int(it[1].shape[0]) for it in pairs) <= self.caps[0]
                and all(tuple(it[4].shape) == m for it in pairs))

    def _load_inputs(self, st, item):
        if not self.fits(item):
            pairs = self.pairs_of(item)
            raise RuntimeError("input does not fit the captured shapes (%d pair(s), %d points, corr %s; the graphs take "
                               "%d pair(s), capacity %d, corr %s)" % (
                                   len(pairs), sum(int(it[0].shape[0]) + int(it[1].shape[0]) for it in pairs),
                                   tuple(pairs[0][4].shape), self.stack, self.caps[0], tuple(self.sets[0].corr.shape[-2:])))
        if self._load_inputs_fused(st, item):
            return
        if self.stack == 1:
            p0, p1, _, _, corr, dk = item
            n0, n1 = int(p0.shape[0]), int(p1.shape[0])
            st.pts[:n0].copy_(p0, non_blocking=True)
            st.pts[n0:n0 + n1].copy_(p1, non_blocking=True)
            if len(item) >= 6 and item[2] is not None and item[3] is not None:   # (feat0, feat1) of the dataset item
                st.feat[:n0].copy_(item[2].reshape(n0, -1), non_blocking=True)
                st.feat[n0:n0 + n1].copy_(item[3].reshape(n1, -1), non_blocking=True)
            st.lens[0] = n0
            st.lens[1] = n1
            st.corr.copy_(corr, non_blocking=True)
            st.dk.copy_(dk, non_blocking=True)
        else:       # clouds 2q, 2q + 1 of the stack = pair q
            off, lens = 0, []
            for q, it in enumerate(item):
                for p, f in ((it[0], it[2]), (it[1], it[3])):
                    n = int(p.shape[0])
                    st.pts[off:off + n].copy_(p, non_blocking=True)
                    if f is not None:
                        st.feat[off:off + n].copy_(f.reshape(n, -1), non_blocking=True)
                    off += n
                    lens.append(n)
                st.corr[q].copy_(it[4], non_blocking=True)
                st.dk[q].copy_(it[5], non_blocking=True)
            if st.lens_host is not None:
                if st.lens_ev is not None:     # the set's previous load (three steps ago in the pipelined loop) has read it
                    st.lens_ev.synchronize()
                st.lens_host.copy_(torch.tensor(lens, dtype=torch.int32))
                st.lens.copy_(st.lens_host, non_blocking=True)          # one small copy, not 2Q fills
                st.lens_ev = st.lens_ev or torch.cuda.Event()
                st.lens_ev.record(torch.cuda.current_stream(self.device))
            else:
                st.lens.copy_(torch.tensor(lens, dtype=torch.int32))
        st.mask.copy_(st.dk > self.circle.safe_radius)   # with the upload, off the training stream (utils/loss.py:119)

    def _load_inputs_fused(self, st, item):
        """The same loads as ONE launch (ops.copy_buffers: clouds, features, correspondences, keypoint distances and the
        loss's neighbor mask) when every tensor of the item already sits on the device in the buffers' dtypes and is
        contiguous -- what ``upload`` returns; ~25 copy launches per stacked step before, on the stream that also replays
        the lane's graphs.  Returns False (nothing done) otherwise."""
        if self.device.type != 'cuda' or not self.FUSED_INPUT_LOAD:
            return False
        pairs = self.pairs_of(item)
        jobs, off, lens = [], 0, []

        def ok(t, dtype):
            return isinstance(t, torch.Tensor) and t.device == self.device and t.dtype == dtype and t.is_contiguous()
        if len(pairs) != self.stack:
            return False
        for q, it in enumerate(pairs):
            corr, dk = it[4], it[5]
            if not (ok(corr, torch.int64) and ok(dk, torch.float64)):
                return False
            dst_corr, dst_dk, dst_mask = (st.corr, st.dk, st.mask) if self.stack == 1 else (st.corr[q], st.dk[q], st.mask[q])
            # the one launch copies BYTE COUNTS of the sources to raw destination pointers: every source must have
            # exactly its destination's extent (the copy_() path raised on a mismatched item; here it would be an
            # out-of-bounds device write into the graph-fed static buffers)
            if tuple(corr.shape) != tuple(dst_corr.shape) or tuple(dk.shape) != tuple(dst_dk.shape) \
                    or dst_mask.numel() != dk.numel():
                return False
            for p, f in ((it[0], it[2]), (it[1], it[3])):
                if not ok(p, torch.float32) or (f is not None and not ok(f, torch.float32)):
                    return False
                n = int(p.shape[0])
                if p.dim() != 2 or p.shape[1] != 3 or off + n > int(st.pts.shape[0]):
                    return False
                jobs.append((p, st.pts[off:off + n]))
                if f is not None:
                    if f.numel() != n * st.feat.shape[1]:
                        return False
                    jobs.append((f, st.feat[off:off + n]))
                off += n
                lens.append(n)
            jobs += [(corr, dst_corr), (dk, dst_dk), (dk, dst_mask, 'mask')]
        if self.stack == 1:
            st.lens[0] = lens[0]
            st.lens[1] = lens[1]
        elif st.lens_host is not None:
            if st.lens_ev is not None:     # the set's previous load (three steps ago in the pipelined loop) has read it
                st.lens_ev.synchronize()
            st.lens_host.copy_(torch.tensor(lens, dtype=torch.int32))
            st.lens.copy_(st.lens_host, non_blocking=True)
            st.lens_ev = st.lens_ev or torch.cuda.Event()
            st.lens_ev.record(torch.cuda.current_stream(self.device))
        else:
            st.lens.copy_(torch.tensor(lens, dtype=torch.int32))
        ops.copy_buffers(jobs, threshold=float(self.circle.safe_radius))
        return True

    def _build_set(self, st, adopt=False):
        """Pyramid of the pair in ``st``'s input buffers -> ``st.batch``.  Every kernel of the build (and of the
        network step on that pair) ORs its flags into the set's own status word; the optimizer skips the update of a
        flagged pair and keeps the flags (GuardedSGD.step), ``check_status`` reports them.  ``adopt`` (the first build, and the
        captured one: a graph's outputs have static addresses) makes the new tensors the set's batch; otherwise they
        are copied into the existing ones so that an already captured network graph keeps seeing its addresses."""
        if st.status is None:
            st.status = ops.DeviceStatus(self.device)
        # (the word describes THIS pair -- what an earlier one caused is kept by the optimizer, state[2:4] --: it is cleared
        # by the build's first launch, together with the cell lists' counters)
        batch = dl.build_pyramid_static(st.pts, st.lens, self.config, self.limits, self.caps,
                                        reverse_tables=self.reverse_tables, status=st.status,
                                        conv_widths=not self.reverse_tables,   # (the eval-mode gate's input: inference)
                                        group=2 if self.stack > 1 else getattr(self, 'group', 0),
                                        engine_upsamples=self.engine_upsamples, clear_status=True)
        batch.pop('_status')
        if st.batch is None or adopt:
            st.batch = batch
            return
        done = set()
        for key in ('points', 'neighbors', 'pools', 'pools_width', 'neighbors_width', 'upsamples', 'stack_lengths'):
            for dst, src in zip(st.batch[key], batch[key]):
                if dst is not None and dst.data_ptr() not in done and dst.numel():
                    dst.copy_(src)
                    done.add(dst.data_ptr())
                    rd, rs = getattr(dst, '_d3f_rev', None), getattr(src, '_d3f_rev', None)
                    if rd is not None and rs is not None:   # the table's transpose lives at static addresses as well
                        for td, tsrc in zip(rd.tensors(), rs.tensors()):
                            if td.data_ptr() not in done:
                                td.copy_(tsrc)
                                done.add(td.data_ptr())

    def _set_batch(self, st):
        batch = dict(st.batch)
        batch['features'], batch['corr'], batch['dist_keypts'] = st.feat, st.corr, st.dk
        batch['neg_mask'] = st.mask
        if self.stack > 1:
            batch['_pairs'] = self.stack
        return batch

    def _use_scale(self):
        """The optimizer's gradient scale for THIS engine's steps (lanes: the join sets it)."""
        if getattr(self, 'lane', None) is None and getattr(self, 'opt', None) is not None:   # (forward-only engines: none)
            self.opt.use_grad_scale(1.0 / (self.stack * max(1, self.world)))

    def _net_step(self, st):
        """Forward + loss + backward (+ guarded update) on the set's pair(s).  Returns (loss, desc, det, accuracy) device
        scalars; with stacked pairs ``loss`` is the sum over the stack and the other three are per pair [Q]."""
        batch = self._set_batch(st)
        lane = getattr(self, 'lane', None)
        if lane is not None:      # one of several pairs in flight (PairLanes): gradient into the lane's own buffer,
            self.flat.bind(lane)  # the optimizer step belongs to the join
        try:
            self.flat.zero_grad()
            loss, desc, det, acc = self.forward_loss(batch)
            with ops.weight_grad_group():     # every weight gradient of the step in one grouped launch (ops.WeightGradGroup)
                torch.autograd.backward(loss, self._seed(loss))
            grad = self.flat.gather_grads()
            if lane is None:
                self.opt.step(want_ok=False, pair_status=st.status.word)
            else:   # a flagged pair turns its lane non-finite: the guard of the joint step then skips the update
                self._poison_if_flagged(grad, st.status.word)
        finally:
            if lane is not None:
                self.flat.bind(0)
        return loss.detach(), desc.detach(), det.detach(), acc.detach()

    def _static_step(self, item):
        """One step on static shapes without a graph (pyramid, then network, on the current stream)."""
        st = self.sets[0]
        self._use_scale()
        self._load_inputs(st, item)
        self._build_set(st)
        st.loaded = item
        return self._net_step(st)

    def capture(self, item):
        """Warm up, then record the graphs (torch.cuda.CUDAGraph = hipGraph): one network step and one pyramid build
        per buffer set.  Network graphs replay on the training stream, pyramid graphs on a side stream."""
        dev = self.device
        main = torch.cuda.current_stream(dev)
        self._use_scale()
        for st in self.sets:
            self._load_inputs(st, item)
            self._build_set(st)
            st.loaded = item
        warm = _warm_stream(dev)
        warm.wait_stream(main)
        from . import tuning_missing_gemms
        # (library-GEMM shapes of THESE capacities that the loaded TunableOp table lacks are tuned by the warm-up steps)
        with torch.cuda.stream(warm), tuning_missing_gemms():
            for k in range(3):
                self._build_set(self.sets[(k + 1) % self.NSETS])
                if self.split_backward and getattr(self, 'lane', None) is not None:
                    s1, s2 = self._lane_stages(self.sets[k % self.NSETS])
                    out = s1()
                    s2()
                elif self.split_backward:
                    batch = self._set_batch(self.sets[k % self.NSETS])
                    out = self._exchange_and_step(lambda: self._backward_deep(batch), self._backward_shallow,
                                                  pair_status=self.sets[k % self.NSETS].status.word)
                else:
                    out = self._net_step(self.sets[k % self.NSETS])
        main.wait_stream(warm)
        torch.cuda.synchronize(dev)
        self.check_status()
        # (stream priorities were tried: the range here is {0, -1}; replaying the network graphs on a priority -1 stream
        # made the step 3x slower, so both streams stay at the default priority)
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=dev)
        self.g_net, self.g_net_b, self.g_pyr, self._graph_out, self._graph_dist = [], [], [], [], []
        # pyramid graphs first: the tensors a captured build produces have static addresses, so they BECOME the set's
        # batch (no copies into separately held buffers: 34 launches per pyramid), and the network graphs are then
        # recorded against them.  Each pyramid graph owns its memory pool: in a shared pool the scratch of one build
        # would be laid over the adopted outputs of another set, which a network graph may be reading at that moment.
        # (The network graphs never run concurrently and replay in capture order: they do share a pool.)
        # A lane's graphs are captured ON THE LANE'S OWN STREAM, not on torch.cuda.graph's default capture stream (one per
        # process): PyTorch keeps one BLAS workspace per (handle, stream) and a captured library GEMM has the workspace of
        # its CAPTURE stream baked in -- graphs of different lanes captured on the shared default stream would all hold
        # the same workspace and then replay CONCURRENTLY on the lanes' streams, their split-K partial sums / semaphores
        # on top of each other (round 5: the root of round 4's "hipBLASLt winner never finishes on a later replay",
        # profiles/r05_hipblaslt_hang.txt; with rocBLAS split-K solutions the same sharing is a silent race).
        cap = getattr(self, 'stream', None) if getattr(self, 'lane', None) is not None else None
        for i in range(self.NSETS):
            g = new_graph()
            with torch.cuda.graph(g, stream=cap, capture_error_mode=_CAPTURE_MODE):
                self._build_set(self.sets[i], adopt=True)
            self.g_pyr.append(finish_graph(g))
        for g in self.g_pyr:    # a capture records, it does not run: fill the adopted tensors (inputs are loaded)
            g.replay()
        torch.cuda.synchronize(dev)
        self._choose_side_stream()
        for i in range(self.NSETS):
            g = new_graph()
            lane_split = self.split_backward and getattr(self, 'lane', None) is not None
            s1, s2 = self._lane_stages(self.sets[i]) if lane_split else (None, None)
            with torch.cuda.graph(g, pool=self.g_net[0].pool() if self.g_net else None, stream=cap,
                                  capture_error_mode=_CAPTURE_MODE):
                if lane_split:
                    self._graph_out.append(s1())
                elif self.split_backward:
                    self._graph_out.append(self._backward_deep(self._set_batch(self.sets[i])))
                else:
                    self._graph_out.append(self._net_step(self.sets[i]))
                self._graph_dist.append(self.last_distances)   # held: the pool keeps these addresses for this graph
            self.g_net.append(finish_graph(g))
            if self.split_backward:  # stage 2 of the same step: continues in the same pool
                g = new_graph()
                with torch.cuda.graph(g, pool=self.g_net[0].pool(), stream=cap, capture_error_mode=_CAPTURE_MODE):
                    if lane_split:
                        s2()
                    else:
                        self._backward_shallow()
                self.g_net_b.append(finish_graph(g))
        torch.cuda.synchronize(dev)
        self.ev_net = [torch.cuda.Event() for _ in range(self.NSETS)]
        self.ev_pyr = [torch.cuda.Event() for _ in range(self.NSETS)]
        for i in range(self.NSETS):
            self.ev_net[i].record(main)
            self.ev_pyr[i].record(main)
        self.cur = 0
        return out

    def _choose_side_stream(self):
        """The pyramid stream must not sit on the compute pipe of the stream the network graphs replay on
        (fresh_streams; measured: 251 pairs/s against 188-218 when they share one).  The training stream is the
        caller's, so its pipe is not ours to pick: four candidates with consecutive queues cover all four pipes, and a
        probe -- two pyramid graphs at once, one on the training stream, one on the candidate -- tells which to avoid.
        The lanes of PairLanes bring their own streams (consecutive queues by construction)."""
        if getattr(self, 'lane', None) is not None or self.__dict__.get('_side_probe') is not None:
            return
        import time
        dev = self.device

        def timed(stream):
            best = float('inf')
            for _ in range(4):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                self.g_pyr[0].replay()
                with torch.cuda.stream(stream):
                    self.g_pyr[1].replay()
                torch.cuda.synchronize(dev)
                best = min(best, time.perf_counter() - t0)
            return best
        serial = timed(torch.cuda.current_stream(dev))     # both on the training stream: what sharing a pipe costs
        probe, chosen = [], None
        for _ in range(4):          # consecutive queues: at most one of four shares the training stream's pipe
            cand = fresh_streams(1, dev)[0]
            probe.append(timed(cand))
            if chosen is None or probe[-1] < chosen[0]:
                chosen = (probe[-1], cand)
            if probe[-1] < 0.8 * serial:
                break
        self._side = chosen[1]
        self._side_probe = {'serial_ms': round(serial * 1e3, 4), 'candidates_ms': [round(t * 1e3, 4) for t in probe]}

    def step_graph(self, item, next_item=None):
        """Train on ``item``; the pyramid of ``next_item`` (default: ``item`` again) is built on the side stream
        meanwhile.  When ``item`` was the previous call's ``next_item`` its pyramid is already there (or under way);
        otherwise it is built first.  ``next_item``'s tensors must be complete when this is called (host arrays or
        device tensors produced on an already synchronised stream): the side stream does not wait for the training
        stream.

        Ordering uses HOST waits on events, not stream waits (a graph launch behind a pending cross-stream wait was
        measured to block the host for the whole step).  With three buffer sets both waits are on work launched at
        least one step earlier, so the host stays a step ahead of the GPU and the training stream never idles:
          * the pyramid of pair k+1 overwrites the set last read by the network step of pair k-2,
          * the network step of pair k needs the pyramid launched during step k-1."""
        self._ensure_loaded(item)
        self._prefetch_next(item if next_item is None else next_item)
        return self._launch_net(item)

    # step_graph in three host phases (PairLanes interleaves them over its lanes: every network graph is launched before
    # any lane spends host time on the following pair's pyramid)
    def _ensure_loaded(self, item):
        i = self.cur
        st = self.sets[i]
        if not same_item(st.loaded, item):
            self.ev_net[i].synchronize()
            self._collect_status(i)
            with torch.cuda.stream(self._side):
                self._load_inputs(st, item)
                self.g_pyr[i].replay()
                self.ev_pyr[i].record(self._side)
            st.loaded = item

    def _prefetch_next(self, nxt, n=None):
        """Pyramid of the following pair into set ``n`` (default: the one after the current) on the side stream."""
        if nxt is TrainStep.NO_PREFETCH:
            return
        n = (self.cur + 1) % self.NSETS if n is None else n
        nx = self.sets[n]
        self.ev_net[n].synchronize()
        self._collect_status(n)
        with torch.cuda.stream(self._side):
            self._load_inputs(nx, nxt)
            self.g_pyr[n].replay()
            self.ev_pyr[n].record(self._side)
        nx.loaded = nxt

    def _launch_net(self, item):
        i = self.cur
        st = self.sets[i]
        main = torch.cuda.current_stream(self.device)
        self._use_scale()
        side = getattr(self, '_side', None)
        if not (side is not None and side.cuda_stream == main.cuda_stream):
            self.ev_pyr[i].synchronize()
        # (else -- four lanes: a lane builds its pyramids on its OWN stream -- stream order already puts the set's pyramid
        # in front of this launch.  The host wait made the host follow the GPU step by step (round 4, bench
        # `host_enqueue_ms_per_step` 20.6 of a 21.5 ms step): every lane's launch waited for that lane's previous step to
        # drain, and for the lanes before it.)

        def done():   # queued behind the pair's network step: host mirror of its status word, then the set's event
            if st.status_host is not None and st.status is not None:
                st.status_host.copy_(st.status.word, non_blocking=True)
                st.status_item = item
            self.ev_net[i].record(main)
        if self.split_backward and getattr(self, 'lane', None) is not None:
            # a lane among several ranks: stage 1 now, stage 2 when PairLanes calls _launch_stage2 (the join exchanges
            # the deep bucket in between)
            self.g_net[i].replay()
            self._stage2 = (i, done)
        elif self.split_backward:
            def stage1():
                self.g_net[i].replay()
                return self._graph_out[i]

            def stage2():
                self.g_net_b[i].replay()
                done()
            self._exchange_and_step(stage1, stage2, pair_status=st.status.word)
        else:
            self.g_net[i].replay()
            done()
        self.cur = (i + 1) % self.NSETS
        self.last_distances = self._graph_dist[i]
        return self._graph_out[i]

    def _launch_stage2(self):
        """Second half of a lane's split step (see _launch_net), on the current stream."""
        i, done = self._stage2
        self._stage2 = None
        self.g_net_b[i].replay()
        done()

    def check_status(self, raise_on_skip=True):
        """One synchronisation: what the device flagged since the last call.

        Graph mode: a pair whose pyramid overflowed a capacity (D3F_ST_CAPACITY, ...) has had its update SKIPPED by the
        optimizer (GuardedSGD.step, ``pair_status``) -- the parameters never saw it; the flags and the number of such
        pairs are returned as ``(flags, count)`` and, with ``raise_on_skip``, raised as the RuntimeError the reference's
        native modules would have thrown.  Eager pyramids (``step``) raise at build time already; the status words they
        left behind (searches) are checked here as well."""
        flags = count = 0
        opt = getattr(self, 'opt', None)
        if opt is not None and opt.state.is_cuda:
            s = opt.state.tolist()
            flags, count = int(s[2]), int(s[3])
            opt.state[2:4] = 0
        for st in getattr(self, 'sets', None) or []:
            if st.status is not None and opt is None:    # forward-only engines have no optimizer to park the flags in
                flags |= int(st.status.word.item())
                st.status.word.zero_()
        for status in getattr(self, '_eager_status', []):
            flags |= int(status.word.item())
        self._eager_status = []
        if flags and raise_on_skip:
            from . import _native
            raise RuntimeError("%s (%d training pair(s) skipped)" % (_native.status_message(flags) or
                                                                     ("device status %d" % flags), count))
        return flags, count

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
        self.opt.use_grad_scale(1.0 / max(1, self.world))      # one pair per eager step
        pending = getattr(self, '_pending', None)
        if pending is not None:
            batch, ev = pending
            self._pending = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            cur = torch.cuda.current_stream(self.device)
            for v in batch.values():  # tensors produced on the side stream are consumed on this one
                for t in (v if isinstance(v, list) else [v]):
                    if isinstance(t, torch.Tensor):
                        t.record_stream(cur)
                        rev = getattr(t, '_d3f_rev', None)     # the table's transpose is read by the backward pass
                        for r in (rev.tensors() if rev is not None else ()):
                            r.record_stream(cur)
        else:
            batch = self.build_batch(item)
            batch['n0'] = int(item[0].shape[0])
        if next_item is not None:
            self.prefetch(next_item)
        pair_status = batch.get('_pair_status')
        if self.split_backward:
            return self._exchange_and_step(lambda: self._backward_deep(batch), self._backward_shallow,
                                           pair_status=pair_status)
        self.flat.zero_grad()
        loss, desc, det, acc = self.forward_loss(batch)
        with ops.weight_grad_group():
            torch.autograd.backward(loss, self._seed(loss))
        if self.world > 1 and pair_status is not None:
            self._poison_if_flagged(self.flat.gather_grads(), pair_status)
            pair_status = None
        allreduce_mean_(self.flat.gather_grads(), self.world, average=False)   # the mean is opt.grad_scale
        self.opt.step(want_ok=False, pair_status=pair_status)
        return loss.detach(), desc.detach(), det.detach(), acc.detach()


class LaneThread(object):
    """A host thread that belongs to one lane of PairLanes: everything the lane CAPTURES runs here (PairLanes.capture),
    so that the lane's library GEMMs are recorded with the BLAS handle -- and its device workspace -- of this thread and
    of no other lane.  Jobs run with autograd's multithreading off: backward nodes execute on the calling thread, not on
    autograd's one worker per device (whose single handle every lane's backward GEMMs would share)."""

    def __init__(self, device, index):
        import queue
        import threading
        self.device = device
        self._jobs = queue.Queue()
        self._thread = threading.Thread(target=self._loop, name="d3f-lane-%d" % index, daemon=True)
        self._thread.start()

    def _loop(self):
        if self.device.type == 'cuda':
            torch.cuda.set_device(self.device)
        while True:
            fn, box, done = self._jobs.get()
            try:
                with torch.autograd.set_multithreading_enabled(False):
                    box.append((True, fn()))
            except BaseException as e:   # noqa: B036 -- handed to the caller of run()
                box.append((False, e))
            finally:
                done.set()

    def run(self, fn):
        """fn() on the lane's thread; returns its result (or raises what it raised)."""
        import threading
        box, done = [], threading.Event()
        self._jobs.put((fn, box, done))
        done.wait()
        ok, val = box[0]
        if not ok:
            raise val
        return val


class PairLanes:
    """Several fragment pairs IN FLIGHT on one GPU, meeting at one optimizer step.

    One pair's network step leaves most of an MI355X idle: the kernels of the coarse levels launch 50-200 workgroups on
    256 CUs, and the fine-level KPConv kernels saturate no unit (profiles/r03_pmc_kpconv.txt) -- they wait on gather
    latency.  Two independent steps replayed on two streams fill those holes (1.43x the pairs/s of one,
    profiles/r03_queue_pipes.txt).  ``lanes`` engines share the model, the flat parameter buffer and the optimizer;
    each has its own training stream, side stream, static buffer sets, graphs and GRADIENT buffer (FlatParams.add_lane).
    A step trains on ``lanes * stack`` pairs at once: every lane replays forward + loss + backward of its pair -- or of
    its STACK of ``stack`` pairs (TrainStep ``stack``: one pyramid, one network graph for all of them) --, then the join
    (on lane 0's stream) applies ONE guarded SGD step on the sum of the lane gradients scaled by
    1 / (lanes * stack * world) -- exactly the update a data-parallel step over that many ranks makes
    (d3f_sgd_guarded_step_lanes forms the sum inside the update kernel).
    With several ranks every lane's backward is captured in two stages and the join moves to a stream of its own: the
    lanes' deep gradient buckets are summed and all-reduced while every lane's stage 2 (the fine levels) executes, the
    shallow buckets follow, then the guard on the reduced gradient and the one update (``step_graph``).
    The reference trains one pair per optimizer step (dataloader.py:73); ``lanes=1, stack=1`` keeps that."""

    def __init__(self, ts, lanes=2, stack=1, split=None):
        """``split``: two-stage lane graphs with the join on a stream of its own -- the multi-rank form; default: when
        there are several ranks (True on one rank runs the same schedule without the all-reduces: tests)."""
        self.ts, self.P, self.Q = ts, int(lanes), int(stack)
        self.split = (ts.world > 1) if split is None else bool(split)
        if ts.world > 1 and not self.split:
            # (the one-rank join -- sum of the lanes inside the update kernel -- exchanges nothing: several ranks would
            # silently train unsynchronised replicas at a gradient scale of 1 / (lanes x stack x ranks))
            raise ValueError("PairLanes over several ranks needs split=True: the join that exchanges the gradients")
        nets, sides = lane_streams(self.P, ts.device)
        self.engines = [ts.clone_for_lane(k, nets[k], sides[k], split=self.split) for k in range(self.P)]
        for eng in self.engines:
            eng.stack = self.Q
        self.ev_lane = [torch.cuda.Event() for _ in range(self.P)]
        self.ev_stage1 = [torch.cuda.Event() for _ in range(self.P)]
        self.exchange = True     # (False: the multi-rank step without its all-reduces -- bench.py's overlap measurement)
        # the last joint update -- shared with the clones for other capacity classes (they step the same parameters)
        self._join = {'ev_step': torch.cuda.Event(), 'stepped': False}
        self._groups = {}

    caps = property(lambda self: self.engines[0].caps)
    pairs_per_step = property(lambda self: self.P * self.Q)
    CAPTURE_IN_THREADS = True     # (False: every lane captured from the calling thread, rounds 3-5 -- A/B experiments)
    JOIN_ON_HOST = False          # (True: the host waits for the lanes' events instead of the join stream: measurements)
    PROBE_DEADLINE_S = 20.0       # probe_overlap: longest a concurrent replay of the lanes' graphs may take

    def _exchange_stream(self):
        """The stream the multi-rank join runs on (shared with the clones for other capacity classes)."""
        if self._join.get('xs') is None:
            self._join['xs'] = torch.cuda.Stream(device=self.ts.device)
        return self._join['xs']

    def clone_for_capacities(self, capacities, num_corr):
        """The same lanes (streams, gradient buffers, join) with buffer sets and graphs for other level capacities: one
        per size class of the dataset (TrainStep.clone_for_capacities)."""
        import copy
        other = copy.copy(self)
        other.engines = [e.clone_for_capacities(capacities, num_corr) for e in self.engines]
        other._groups = {}
        return other

    def probe_overlap(self, reps=3, redeal=True):
        """Do the lanes' network graphs really run side by side?  After ``capture``: the P stage-1 graphs replayed one
        after the other (a synchronisation in between) against all at once; the ratio is the overlap factor (1 = the
        streams share a compute pipe and serialise).  Nothing in HIP says which pipe a stream's
        hardware queue landed on, so this is measured: below 1.15 the lanes are dealt fresh streams once (graphs replay
        on whatever stream is current) and the better deal is kept.  Lane graphs do not touch the parameters, so the
        probe is free of side effects.  Returns {'serial_ms', 'concurrent_ms', 'factor', 'redealt'}."""
        import time
        import warnings
        from . import HIP_WAS_INITIALISED_AT_IMPORT
        if self.P < 2 or self.ts.device.type != 'cuda':
            return {'factor': 1.0, 'redealt': False}
        if HIP_WAS_INITIALISED_AT_IMPORT:
            warnings.warn("the HIP runtime was initialised before d3feat_pytorch_amd was imported: GPU_MAX_HW_QUEUES=32 was "
                          "not in effect, streams beyond four share hardware queues and the lanes may serialise")
        dev = self.ts.device

        def measure():
            graphs = [eng.g_net[eng.cur] for eng in self.engines]
            ser = con = float('inf')
            for _ in range(reps):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for eng, g in zip(self.engines, graphs):
                    with torch.cuda.stream(eng.stream):
                        g.replay()
                    eng.stream.synchronize()
                ser = min(ser, time.perf_counter() - t0)
                t0 = time.perf_counter()
                marks = []
                for eng, g in zip(self.engines, graphs):
                    with torch.cuda.stream(eng.stream):
                        g.replay()
                        ev = torch.cuda.Event()
                        ev.record(eng.stream)
                        marks.append(ev)
                # bounded: the first concurrent re-replay of the lanes' graphs is where a captured library kernel that
                # cannot run beside its own copies would stall for good (rounds 4-5) -- poll, never block
                while not all(ev.query() for ev in marks):
                    if time.perf_counter() - t0 > self.PROBE_DEADLINE_S:
                        raise RuntimeError(
                            "PairLanes: the lanes' network graphs did not finish a concurrent replay within %.0f s (the "
                            "same graphs replay one after the other in %.1f ms): a captured kernel does not survive "
                            "running beside its copies in the other lanes' graphs.  Known cause: library GEMMs of several "
                            "lanes recorded with ONE BLAS handle (shared device workspace) -- PairLanes.CAPTURE_IN_THREADS "
                            "= True gives every lane its own.  The GPU queue is stuck; end the process."
                            % (self.PROBE_DEADLINE_S, ser * 1e3))
                    time.sleep(0.0005)
                con = min(con, time.perf_counter() - t0)
            return ser, con
        ser, con = measure()
        out = {'serial_ms': round(ser * 1e3, 3), 'concurrent_ms': round(con * 1e3, 3), 'factor': round(ser / con, 3),
               'redealt': False}
        if redeal and ser / con < 1.15:
            old = [(e.stream, e._side) for e in self.engines]
            nets, sides = lane_streams(self.P, dev)
            for e, sn, sd in zip(self.engines, nets, sides):
                e.stream, e._side = sn, sd
            ser2, con2 = measure()
            if ser2 / con2 > ser / con:
                out.update({'serial_ms': round(ser2 * 1e3, 3), 'concurrent_ms': round(con2 * 1e3, 3),
                            'factor': round(ser2 / con2, 3), 'redealt': True})
            else:
                for e, (sn, sd) in zip(self.engines, old):
                    e.stream, e._side = sn, sd
        self.overlap = out
        return out

    def deal(self, items):
        """The step's ``lanes * stack`` pairs as one input per lane: the pair itself, or a stack of ``stack`` pairs."""
        if len(items) != self.P * self.Q:
            raise ValueError("%d pairs for %d lanes x %d stacked pairs" % (len(items), self.P, self.Q))
        if self.Q == 1:
            return list(items)
        return [tuple(items[k * self.Q:(k + 1) * self.Q]) for k in range(self.P)]

    def preload(self, items):
        """Pyramids of the lanes' next pairs into the sets they train on next (the previous step ran on another class)."""
        for eng, item in zip(self.engines, self.deal(items)):
            eng.preload(item)

    def enable_graph(self, capacities, num_corr):
        for eng in self.engines:
            eng.enable_graph(capacities, num_corr, stack=self.Q)

    def fits(self, item):
        """Whether one pair fits a lane's share of the captured capacities (a lane's stack holds ``stack`` of them)."""
        eng = self.engines[0]
        m = tuple(eng.sets[0].corr.shape[-2:])
        return (self.Q * (int(item[0].shape[0]) + int(item[1].shape[0])) <= eng.caps[0]
                and tuple(item[4].shape) == m) if self.Q > 1 else eng.fits(item)

    def fits_group(self, items):
        """Whether the ``lanes * stack`` pairs of a step fit the captured shapes as dealt."""
        return len(items) == self.P * self.Q and all(e.fits(it) for e, it in zip(self.engines, self.deal(items)))

    def capture(self, item):
        """``item``: one pair (repeated to fill every stack) or the ``lanes * stack`` pairs of a step."""
        per_lane = self.deal(item) if is_stack(item) and len(item) == self.P * self.Q and self.P * self.Q > 1 else [
            (tuple([item] * self.Q) if self.Q > 1 else item)] * self.P
        if not self.CAPTURE_IN_THREADS or self.P < 2:
            out = None
            for eng, it in zip(self.engines, per_lane):
                eng.stream.wait_stream(torch.cuda.current_stream(self.ts.device))
                with torch.cuda.stream(eng.stream):
                    out = eng.capture(it)
                eng.stream.synchronize()
            return out
        # Every lane is captured on a host thread of ITS OWN (LaneThread: alive as long as the lanes; also used by the
        # clones for other capacity classes), with autograd's backward on that thread instead of its per-device worker.
        # Why: PyTorch hands every host THREAD its own BLAS handle, and a rocBLAS handle owns the device workspace
        # (split-K partial sums, stream-K flags) whose address a captured library GEMM gets baked in.  Captured from one
        # thread -- and with every backward GEMM issued by autograd's one device thread -- the graphs of ALL lanes held
        # the SAME workspace and then replayed concurrently: with the library's default solution picks the first joint
        # replay of 4 lanes x 3 stacked pairs stalled for good, 6 of 6 runs in round 5 and 2 of 2 in round 6 before this
        # change, 0 of 4 after it (profiles/r06_stall_root_cause.txt).  One handle per lane makes the condition
        # impossible by construction instead of avoided by solution selection.
        dev = self.ts.device
        cur = torch.cuda.current_stream(dev)
        lane_threads = self._join.setdefault('threads', [])
        while len(lane_threads) < self.P:
            lane_threads.append(LaneThread(dev, len(lane_threads)))
        out = None
        for eng, it, th in zip(self.engines, per_lane, lane_threads):
            def job(eng=eng, it=it):
                eng.stream.wait_stream(cur)
                with torch.cuda.stream(eng.stream):
                    r = eng.capture(it)
                eng.stream.synchronize()
                return r
            out = th.run(job)            # one capture at a time; the thread keeps its handle afterwards
        # the first concurrent replay of what was just recorded, against a deadline: a captured kernel that cannot run
        # beside its copies shows up HERE as a RuntimeError, not as a training stream that stalls for good later
        self.probe_overlap(reps=1, redeal=False)
        return out

    def step_graph(self, items, next_items=None):
        """Train on ``items`` (``lanes * stack`` pairs: lane k takes items[k*stack:(k+1)*stack]); ``next_items``
        (default: the same again) are the pairs of the following call, whose pyramids are built on the lanes' side
        streams meanwhile.  Returns the lanes' (loss, desc_loss, det_loss, accuracy) device scalars (stacked lanes: see
        TrainStep._net_step); they are complete after ``synchronize()``."""
        items = self.deal(items)
        if next_items is TrainStep.NO_PREFETCH:
            nxt = [TrainStep.NO_PREFETCH] * self.P
        else:
            nxt = items if next_items is None else self.deal(next_items)
        self.ts.opt.use_grad_scale(1.0 / (self.P * self.Q * max(1, self.ts.world)))
        outs = []
        host_join = self.JOIN_ON_HOST
        split = self.split              # several ranks: two-stage lanes, the deep bucket exchanged under stage 2
        for k, eng in enumerate(self.engines):
            eng._ensure_loaded(items[k])
        for k, eng in enumerate(self.engines):     # every network graph first ...
            with torch.cuda.stream(eng.stream):
                if (k > 0 or split) and self._join['stepped']:      # the parameters of the previous joint update
                    if host_join:
                        self._join['ev_step'].synchronize()
                    else:
                        eng.stream.wait_event(self._join['ev_step'])
                outs.append(eng._launch_net(items[k]))
                (self.ev_stage1 if split else self.ev_lane)[k].record(eng.stream)
        flat, opt = self.ts.flat, self.ts.opt
        grads = [flat.lanes[k][0] for k in range(self.P)]
        if split:
            # the join runs on a stream of its own: deep buckets of the lanes summed into lane 0's and all-reduced in
            # three chunks (xGMI rings are per-link bound: a few >= 30 MB messages) WHILE every lane's stage 2 -- the
            # backward of the fine levels, about a third of a step -- executes; then the small shallow bucket, the guard on
            # the reduced gradient and ONE update.  No pass of its own over the 97 MB for the mean (opt.grad_scale).
            ns = self.ts.numel_shallow
            xs = self._exchange_stream()
            with torch.cuda.stream(xs):
                for k in range(self.P):
                    xs.wait_event(self.ev_stage1[k])
                deep = grads[0][ns:]
                for g in grads[1:]:
                    deep.add_(g[ns:])
                works, nb = [], 3
                step = (deep.numel() + nb - 1) // nb
                for b in range(nb):
                    chunk = deep[b * step:min(deep.numel(), (b + 1) * step)]
                    if chunk.numel() and self.exchange and self.ts.world > 1:
                        works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))
            for k, eng in enumerate(self.engines):
                with torch.cuda.stream(eng.stream):
                    eng._launch_stage2()
                    self.ev_lane[k].record(eng.stream)
        for k, eng in enumerate(self.engines):     # ... then the following pairs' pyramids, under the running networks
            eng._prefetch_next(nxt[k], n=eng.cur)
        if split:
            with torch.cuda.stream(xs):
                for k in range(self.P):
                    xs.wait_event(self.ev_lane[k])
                shallow = grads[0][:ns]
                for g in grads[1:]:
                    shallow.add_(g[:ns])
                if self.exchange and self.ts.world > 1:
                    works.append(dist.all_reduce(shallow, op=dist.ReduceOp.SUM, async_op=True))
                for w in works:
                    w.wait()                 # SUM over ranks; the 1 / (pairs x ranks) of the mean is opt.grad_scale
                opt.step(want_ok=False, grads=grads[:1])
                self._join['ev_step'].record(xs)
        else:
            s0 = self.engines[0].stream
            with torch.cuda.stream(s0):
                for k in range(1, self.P):
                    if host_join:
                        self.ev_lane[k].synchronize()
                    else:
                        s0.wait_event(self.ev_lane[k])
                opt.step(want_ok=False, grads=grads)
                self._join['ev_step'].record(s0)
        self._join['stepped'] = True
        group = tuple(items)
        for it in group:
            self._groups[id(it)] = group
        while len(self._groups) > 16 * self.P:
            self._groups.pop(next(iter(self._groups)))
        return outs

    def synchronize(self):
        for eng in self.engines:
            eng.stream.synchronize()
        if self._join.get('xs') is not None:
            self._join['xs'].synchronize()

    def resync(self):
        """After the parameters were updated OUTSIDE the lanes (an eager step on the current stream; the caller has
        ``synchronize()``d the lanes before it): the lanes' next graphs wait for that update."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.ts.device))
        for eng in self.engines:
            eng.stream.wait_event(ev)
        if self._join.get('xs') is not None:
            self._join['xs'].wait_event(ev)

    def make_visible(self, stream=None):
        """The lanes' latest outputs (losses, distances) become readable on ``stream`` (default: the current one)."""
        stream = torch.cuda.current_stream(self.ts.device) if stream is None else stream
        for ev in self.ev_lane:
            stream.wait_event(ev)

    def check_status(self, raise_on_skip=True):
        return self.engines[0].check_status(raise_on_skip)

    def take_overflowed(self, drain=False):
        """[(item, flags)]: pairs whose joint update the optimizer skipped -- the flagged pair AND the pairs that shared
        its step (flags 0), for the caller to run again."""
        flagged = []
        for eng in self.engines:
            flagged += eng.take_overflowed(drain)
        out, seen = [], set()
        for item, flags in flagged:
            for mate in self._groups.get(id(item), (item,)):
                for pair in TrainStep.pairs_of(mate):
                    if id(pair) not in seen:
                        seen.add(id(pair))
                        out.append((pair, flags if mate is item else 0))
        return out
