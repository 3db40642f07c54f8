"""Epoch loop around :class:`TrainStep` with the reference ``Trainer``'s semantics (trainer.py:9-225).

What is kept from the reference, by name: ``Trainer(args)`` reading ``max_epoch, training_max_iter, val_max_iter,
save_dir, verbose, scheduler_interval, snapshot_interval, pretrain, train_loader, val_loader``; ``train()``,
``train_epoch(epoch)``, ``evaluate(epoch)``, ``_snapshot(epoch, name)``, ``_load_pretrain(path)``, ``_get_lr()``; the
best-loss / best-accuracy snapshots (trainer.py:48-54), the ExponentialLR step every ``scheduler_interval`` epochs
(:59-60, training_3DMatch.py:78-81), the non-finite-gradient guard (:104-111, on the device here) and the snapshot
file layout ``{'epoch', 'state_dict', 'optimizer', 'scheduler', 'best_loss'}`` (:193-206) with torch.optim.SGD /
ExponentialLR shaped state, so snapshots move between the two code bases in either direction.

What differs: one step is ``TrainStep.step_graph`` (the pipelined hipGraph replay) whenever the pair fits the captured
capacities, the eager stream-ordered step otherwise; running statistics are accumulated on the device and read back
every ``log_interval`` iterations (the reference reads five scalars back per step, trainer.py:115-119); scalars go to
any object with ``add_scalar(tag, value, step)`` (``args.writer``; tensorboardX is not a dependency), by default a
JSON-lines file under ``args.tboard_dir``.  With ``world_size > 1`` every rank walks the same shuffled order and takes
every ``world_size``-th pair; rank 0 writes snapshots and logs.

``args.pairs_in_flight`` (L, default 1) and ``args.stacked_pairs`` (Q, default 1) train on L x Q pairs per optimizer
step and GPU: Q pairs STACKED into one pyramid + one network graph (TrainStep ``stack``), L such graphs in flight on
streams of their own (train.PairLanes), one SGD step on the mean of the L x Q gradients -- the update a data-parallel
step over L x Q times as many ranks makes; graph mode only.  L = Q = 1 is the reference's one pair per optimizer step
(dataloader.py:73).  NOTE on the learning rate: the mean gradient of P pairs is the gradient of a batch of P, so the
reference's schedule (lr 0.01, momentum 0.98, training_3DMatch.py:62-76), tuned for batches of one pair, sees P times
fewer and less noisy updates per epoch; scale ``lr`` (or the epoch count) accordingly -- the trainer does not.
With several ranks every per-step decision that changes which collectives a rank issues (graph step or eager
fallback) is agreed between the ranks first (``_agree``), and pairs whose update was skipped are NOT re-run (a rank
re-running alone would issue all-reduces nobody matches): the consistent skip of rounds 1-2.
"""
import json
import os

import numpy as np
import torch
import torch.distributed as dist

from .train import PairLanes, TrainStep


class ExponentialLR:
    """``torch.optim.lr_scheduler.ExponentialLR`` on a :class:`GuardedSGD` (same recursion ``lr <- lr * gamma``, same
    state_dict keys), writing the new rate to the optimizer's device-resident hyper-parameters."""

    def __init__(self, optimizer, gamma, last_epoch=0):
        self.optimizer, self.gamma, self.last_epoch = optimizer, float(gamma), int(last_epoch)
        self.base_lrs = [optimizer.initial_lr]
        self._last_lr = [optimizer.lr]

    def step(self):
        self.last_epoch += 1
        self.optimizer.lr = self.optimizer.lr * self.gamma
        self._last_lr = [self.optimizer.lr]

    def get_last_lr(self):
        return list(self._last_lr)

    def state_dict(self):
        return {'gamma': self.gamma, 'base_lrs': list(self.base_lrs), 'last_epoch': self.last_epoch,
                '_step_count': self.last_epoch + 1, '_get_lr_called_within_step': False,
                '_last_lr': list(self._last_lr)}

    def load_state_dict(self, sd):
        self.gamma = float(sd['gamma'])
        self.base_lrs = [float(x) for x in sd['base_lrs']]
        self.last_epoch = int(sd['last_epoch'])
        last = sd.get('_last_lr')
        self.optimizer.lr = float(last[0]) if last else self.base_lrs[0] * self.gamma ** self.last_epoch
        self._last_lr = [self.optimizer.lr]


class ScalarLog:
    """``add_scalar`` sink: one JSON object per line in ``<log_dir>/scalars.jsonl``."""

    def __init__(self, log_dir):
        self.path = None
        if log_dir:
            os.makedirs(log_dir, exist_ok=True)
            self.path = os.path.join(log_dir, 'scalars.jsonl')

    def add_scalar(self, tag, value, step):
        if self.path is not None:
            with open(self.path, 'a') as f:
                f.write(json.dumps({'tag': tag, 'value': float(value), 'step': int(step)}) + '\n')


class _Meters:
    """Running sums of (desc_loss, det_loss, accuracy, d_pos, d_neg) on the device; one read-back per report."""
    NAMES = ('desc_loss', 'det_loss', 'accuracy', 'd_pos', 'd_neg')

    def __init__(self, device):
        self.sum = torch.zeros(5, dtype=torch.float64, device=device)
        self.count = 0

    def update(self, desc, det, acc, d_pos, d_neg):
        self.sum += torch.stack([desc.double().reshape(()), det.double().reshape(()), acc.double().reshape(()),
                                 d_pos.double().reshape(()), d_neg.double().reshape(())])
        self.count += 1

    def averages(self):
        v = (self.sum / max(1, self.count)).tolist()   # the only synchronisation
        return dict(zip(self.NAMES, v))


def _get(args, name, default):
    return getattr(args, name, default) if getattr(args, name, None) is not None else default


class Trainer(object):
    AUTO_LANES, AUTO_STACK, AUTO_MIN_STEPS = 4, 3, 8      # args.fast_schedule (see __init__)
    FAST_LR_SCALE = 1.0     # learning-rate factor of fast_schedule (profiles/r06_train_curve_schedules.txt)

    def __init__(self, args):
        self.config = args
        self.start_epoch = 0
        self.max_epoch = args.max_epoch
        self.training_max_iter = _get(args, 'training_max_iter', 3500)
        self.val_max_iter = _get(args, 'val_max_iter', 500)
        self.save_dir = _get(args, 'save_dir', None)
        self.verbose = _get(args, 'verbose', False)
        self.log_interval = _get(args, 'log_interval', 100)
        self.scheduler_interval = _get(args, 'scheduler_interval', 1)
        self.snapshot_interval = _get(args, 'snapshot_interval', 1)
        self.best_acc = 0
        self.best_loss = 10000000
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.device = torch.device(_get(args, 'device', 'cuda'))

        self.train_loader = args.train_loader
        self.val_loader = _get(args, 'val_loader', None)
        limits = _get(args, 'neighborhood_limits', None)
        if limits is None:
            limits = self.train_loader.limits
        self.engine = TrainStep(args, limits, self.device, world_size=self.world, seed=_get(args, 'seed', 0),
                                model=_get(args, 'model', None))
        self.model = self.engine.model
        self.optimizer = self.engine.opt
        self.scheduler = ExponentialLR(self.optimizer, gamma=args.scheduler_gamma)
        self.writer = _get(args, 'writer', None) or ScalarLog(_get(args, 'tboard_dir', None) if self.rank == 0 else None)
        self.use_graph = bool(_get(args, 'graph', self.device.type == 'cuda'))
        if self.use_graph and _get(args, 'use_batch_norm', False):
            # batch statistics must run over the live points: the capacity-padded graph path is not used with BatchNorm
            if self.rank == 0:
                print("note: use_batch_norm=True -> eager training path (no hipGraph replay)")
            self.use_graph = False
        self._captured = False
        # Pairs per optimizer step and GPU.  DEFAULT: the reference's schedule -- one pair per optimizer step
        # (dataloader.py:73 batch size 1, trainer.py:89-111), so a run with the reference's config reproduces the
        # reference's optimisation trajectory (hyper-parameters of training_3DMatch.py:62-81 are tuned for it).
        # OPT-IN: ``args.fast_schedule = True`` trains on AUTO_LANES x AUTO_STACK = 12 pairs per step and GPU (4 network
        # graphs in flight x 3 pairs stacked in each: what an MI355X needs to be busy, bench.py 2.1x the pairs/s); every
        # update is then the MEAN gradient of 12 x ranks pairs, i.e. 12x fewer updates per epoch.
        # profiles/r06_train_curve_schedules.txt compares the schedules on one pair stream at equal pairs seen; the rule
        # it supports is applied when ``args.fast_schedule_lr_scale`` is left at None (see FAST_LR_SCALE), and printed.
        # ``pairs_in_flight`` / ``stacked_pairs`` pin any other schedule (no scaling applied: the caller's business).
        self.lanes = max(1, int(_get(args, 'pairs_in_flight', 1)))
        self.stack = max(1, int(_get(args, 'stacked_pairs', 1)))
        pinned = _get(args, 'pairs_in_flight', None) is not None or _get(args, 'stacked_pairs', None) is not None
        self.lr_scale = 1.0
        if _get(args, 'reference_schedule', False):
            self.lanes = self.stack = 1
        elif _get(args, 'fast_schedule', False) and not pinned and self.use_graph and self.device.type == 'cuda':
            per_rank = len(self.train_loader.dataset) // max(1, getattr(self.train_loader, 'batch_size', 1)) // self.world
            if per_rank >= self.AUTO_MIN_STEPS * self.AUTO_LANES * self.AUTO_STACK:
                self.lanes, self.stack = self.AUTO_LANES, self.AUTO_STACK
                scale = _get(args, 'fast_schedule_lr_scale', None)
                self.lr_scale = float(self.FAST_LR_SCALE if scale is None else scale)
                if self.lr_scale != 1.0:
                    self.optimizer.lr = self.optimizer.lr * self.lr_scale
                    self.optimizer.initial_lr = self.optimizer.initial_lr * self.lr_scale
                    self.scheduler.base_lrs = [lr * self.lr_scale for lr in self.scheduler.base_lrs]
                    self.scheduler._last_lr = [self.optimizer.lr]
                if self.rank == 0:
                    print("note: fast_schedule -- %d x %d = %d fragment pairs per optimizer step and GPU; every update is "
                          "the MEAN gradient of those pairs x %d rank(s), a batch of %d where the reference steps once per "
                          "pair (dataloader.py:73); learning rate x %g -> %g (profiles/r06_train_curve_schedules.txt)"
                          % (self.lanes, self.stack, self.lanes * self.stack, self.world,
                             self.lanes * self.stack * self.world, self.lr_scale,
                             float(self.optimizer.param_groups[0]['lr'])))
            elif self.rank == 0:
                print("note: fast_schedule needs %d pairs per rank and epoch (%d here): training one pair per step"
                      % (self.AUTO_MIN_STEPS * self.AUTO_LANES * self.AUTO_STACK, per_rank))
        if self.lanes * self.stack > 1 and _get(args, 'tuned_gemms', False):
            # TunableOp-selected library GEMMs (the shipped table + tuning of the missing shapes while a class is
            # captured) are an OPTIMISATION (~6 % of a step) the caller opts into -- process-wide state is not toggled
            # behind their back.  They are NOT needed for safety any more: every lane records its GEMMs with a BLAS handle
            # of its own (train.PairLanes.capture), and an unreadable / version-mismatched table is reported, not ignored.
            from . import enable_tuned_gemms
            if not enable_tuned_gemms() and self.rank == 0:
                print("WARNING: tuned_gemms requested but tuned/tunableop_gfx950.csv was not accepted by this PyTorch / "
                      "ROCm stack (validators differ): the library's default GEMM picks are used")
        if self.lanes * self.stack > 1 and not self.use_graph:
            if self.rank == 0:
                print("note: pairs_in_flight=%d / stacked_pairs=%d need the hipGraph path; training one pair per step"
                      % (self.lanes, self.stack))
            self.lanes = self.stack = 1
        self.group = self.lanes * self.stack          # pairs per optimizer step and GPU
        self._agree_stream = None
        if _get(args, 'pretrain', ''):
            self._load_pretrain(args.pretrain)

    # ---------------------------------------------------------------------------------------------------------
    def _capacity_classes(self, dataset, samples=32, slack=1.10, classes=3):
        """[(capacities, dataset index of a member)] ascending by level-0 capacity: the level sizes of up to ``samples``
        pairs spread over the dataset, sorted by their level-0 size and cut into up to ``classes`` groups of equal
        count, each with its own head-room -- real 3DMatch pairs (reference datasets/ThreeDMatch.py:93-149) vary
        several-fold in size, and ONE capacity set would make every small pair pay the largest pair's kernels.  Groups
        whose level-0 capacities end up within 20 % of each other are merged (a class costs three buffer sets and six
        graphs).  ``graph_capacities`` in the config (one list, or a list of lists) overrides the sampling."""
        caps = _get(self.config, 'graph_capacities', None)
        n = len(dataset)
        picks = sorted(set(int(round(k * (n - 1) / max(1, min(samples, n) - 1))) for k in range(min(samples, n))))
        sized = []
        for i in picks:
            b = self.engine.build_batch(self._fetch(dataset, i))
            sized.append(([int(t.shape[0]) for t in b['points']], i))
        sized.sort(key=lambda t: t[0][0])
        if caps is not None:
            lists = [list(map(int, c)) for c in caps] if isinstance(caps[0], (list, tuple)) else [list(map(int, caps))]
            out = []
            for c in sorted(lists, key=lambda c: c[0]):
                fit = [i for sz, i in sized if all(a <= b for a, b in zip(sz, c))]
                if not fit:
                    raise ValueError("no sampled pair fits graph_capacities %s" % (c,))
                out.append((c, fit[-1]))
            return out
        k = max(1, min(int(classes), len(sized)))
        groups = [sized[len(sized) * g // k: len(sized) * (g + 1) // k] for g in range(k)]
        out = []
        for grp in (g for g in groups if g):
            c = TrainStep.capacities_for([sz for sz, _ in grp], slack=slack)
            if out and c[0] <= 1.2 * out[-1][0][0]:
                merged = [max(a, b) for a, b in zip(out[-1][0], c)]
                out[-1] = (merged, grp[-1][1])
            else:
                out.append((c, grp[-1][1]))
        return out

    def _shuffles(self, loader):
        """Whether the loader asks for a random order: our ``_PairLoader.shuffle``, or a ``torch.utils.data.DataLoader``
        built with ``shuffle=True`` (it then carries a RandomSampler).  Only ``loader.dataset`` is used: the pairs are
        collated on the device by the engine, the loader's own collate_fn / workers are not involved."""
        if hasattr(loader, 'shuffle'):
            return bool(loader.shuffle)
        sampler = getattr(loader, 'sampler', None)
        if sampler is None:
            return False
        if type(sampler).__name__ == 'DistributedSampler':
            return bool(getattr(sampler, 'shuffle', True))     # DistributedSampler(shuffle=False) keeps the order
        return type(sampler).__name__ == 'RandomSampler'

    def _order(self, loader, epoch):
        n = len(loader.dataset)
        if _get(self.config, 'shuffle', None) if _get(self.config, 'shuffle', None) is not None else self._shuffles(loader):
            return np.random.RandomState(_get(self.config, 'seed', 0) * 100003 + epoch).permutation(n)
        return np.arange(n)

    def _report_skipped(self):
        self._rerun_overflowed(drain=True)
        flags, count = self.engine.check_status(raise_on_skip=False)
        rerun = getattr(self, 'rerun_pairs', 0) - getattr(self, '_rerun_reported', 0)
        rerun_steps = getattr(self, 'rerun_steps', 0) - getattr(self, '_rerun_steps_reported', 0)
        self._rerun_reported = getattr(self, 'rerun_pairs', 0)
        self._rerun_steps_reported = getattr(self, 'rerun_steps', 0)
        if rerun and self.rank == 0:
            print("note: %d pair(s) outgrew their capacity class and were trained on the eager path instead" % rerun)
        count = max(0, count - rerun_steps)       # skipped steps that were NOT made up for
        if rerun and not count:
            flags = 0
        if (count or flags) and self.rank == 0:
            from . import _native
            print("warning: %d step(s) of %d pair(s) skipped -- %s" % (
                count, self.group, _native.status_message(flags) or ("status %d" % flags)))
        self.skipped_pairs = getattr(self, 'skipped_pairs', 0) + count * self.group
        return count

    def _fetch(self, dataset, i):
        return self.engine.upload(dataset[int(i)])

    def _capture_classes(self, item):
        """One graph engine per size class (TrainStep.clone_for_capacities), each captured on a member of its class.
        capture() warms the step up by running it: parameters, momentum and counters are put back afterwards."""
        eng = self.engine
        ds = self.train_loader.dataset
        self._engines = []
        keep = (eng.flat.data.clone(), eng.opt.buf.clone(), eng.opt.state.clone())
        for caps, member in self._capacity_classes(ds, classes=int(_get(self.config, 'capacity_classes', 3))):
            caps = [self.stack * int(c) for c in caps]     # a stack holds `stack` pairs of the class
            if self.lanes > 1:        # several graphs in flight: the lanes of every class share streams and the join
                if not self._engines:
                    e = PairLanes(eng, self.lanes, stack=self.stack)
                    e.enable_graph(caps, num_corr=int(item[4].shape[0]))
                else:
                    e = self._engines[0].clone_for_capacities(caps, num_corr=int(item[4].shape[0]))
            elif not self._engines:   # the first class lives in the engine itself, the others in clones of it
                e = eng
                e.enable_graph(caps, num_corr=int(item[4].shape[0]), stack=self.stack)
            else:
                e = eng.clone_for_capacities(caps, num_corr=int(item[4].shape[0]), stack=self.stack)
            one = self._fetch(ds, member)
            e.capture(one if self.lanes > 1 or self.stack == 1 else tuple([one] * self.stack))
            if self.lanes > 1 and not self._engines:
                self.lanes_overlap = e.probe_overlap()       # do the lanes' streams really run side by side?
            self._engines.append(e)
        for dst, src in zip((eng.flat.data, eng.opt.buf, eng.opt.state), keep):
            dst.copy_(src)
        self._captured = True

    def _class_of(self, item):
        for e in self._engines:       # ascending capacities: the smallest class the pair fits
            if e.fits(item):
                return e
        return None

    def _rerun_overflowed(self, drain=False):
        """Pairs whose graph step the optimizer skipped because a deeper level outgrew its class capacity are trained
        on the eager path (exact shapes) a few steps late instead of being dropped -- the reference trains on every
        pair (trainer.py:89-111)."""
        again = [x for e in getattr(self, '_engines', []) for x in e.take_overflowed(drain=drain)]
        if again and self.world > 1:
            return      # (a rank re-running alone would issue collectives no other rank matches: the skip stands)
        if again:
            pairs = [p for entry, _ in again for p in TrainStep.pairs_of(entry)]
            self._eager_steps(pairs)
            self.rerun_pairs = getattr(self, 'rerun_pairs', 0) + len(pairs)
            # the optimizer counts skipped STEPS (state[3]); a step holds `group` pairs
            self.rerun_steps = getattr(self, 'rerun_steps', 0) + max(1, len(pairs) // self.group)

    def _eager_steps(self, items):
        """Eager steps on the current stream.  With pairs in flight the lanes are drained first and wait for these
        updates afterwards (their graphs run on streams of their own)."""
        lanes = self._engines[0] if self.lanes > 1 and getattr(self, '_engines', None) else None
        if lanes is not None:
            lanes.synchronize()
        res = []
        for it in items:
            _, desc, det, acc = self.engine.step(it)
            fp, an = self.engine.last_distances
            res.append((desc, det, acc, fp.mean(), an.mean()))
        if lanes is not None:
            lanes.resync()
        return res

    def _agree(self, flag):
        """Logical AND of ``flag`` over the ranks (one tiny all-reduce on a stream of its own, so that reading it does
        not wait for the training streams): decisions that change which collectives a rank issues must be common."""
        if self.world == 1:
            return bool(flag)
        if self.device.type != 'cuda':
            t = torch.tensor([1 if flag else 0], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))
        if self._agree_stream is None:
            self._agree_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._agree_stream):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = bool(int(t.item()))
        return ok

    # The agreement for a step is POSTED one step ahead (its pairs are known then: the caller hands every step the
    # following step's pairs) as an asynchronous all-reduce and READ when the step starts: the read waits for a
    # collective issued a whole step earlier -- queued behind the exchange of the step before that one, long finished --
    # so the host keeps its run-ahead over the device (an all-reduce + .item() inside every step made the host
    # wait for the previous step's gradient exchange to drain).  Every rank posts and reads in the same order.
    def _post_agreement(self, key, flag):
        if self.world == 1:
            return
        if self.device.type != 'cuda':
            t = torch.tensor([1 if flag else 0], dtype=torch.int32)
            self._pending_agree = (key, t, dist.all_reduce(t, op=dist.ReduceOp.MIN, async_op=True))
            return
        if self._agree_stream is None:
            self._agree_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._agree_stream):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device)
            work = dist.all_reduce(t, op=dist.ReduceOp.MIN, async_op=True)
        self._pending_agree = (key, t, work)

    def _agreed(self, key, flag):
        """The ranks' common decision for the step ``key`` (logical AND of ``flag``): the posted one when there is one."""
        if self.world == 1:
            return bool(flag)
        pend, self._pending_agree = getattr(self, '_pending_agree', None), None
        if pend is not None and pend[0] == key:
            _, t, work = pend
            if t.is_cuda:
                # work.wait() orders the CURRENT stream behind the collective: it has to be the stream the flag is read
                # on (waited for on the training stream, the .item() copy on the agree stream could read this rank's own
                # un-reduced flag while another rank posts late -- ranks would then disagree on graph vs eager path)
                with torch.cuda.stream(self._agree_stream):
                    work.wait()
                    return bool(int(t.item()))
            work.wait()
            return bool(int(t.item()))
        if pend is not None:          # (a posted agreement nobody reads would leave the ranks' collectives out of step)
            pend[2].wait()
        return self._agree(flag)

    def _group_class(self, group):
        """The smallest capacity class whose graphs take the ``group`` pairs of a step (None: eager fallback)."""
        if group is None:
            return None
        for e in self._engines:
            if (e.fits_group(group) if self.lanes > 1 else e.fits(tuple(group))):
                return e
        return None

    def _lanes_step(self, items, next_items):
        """One joint step on ``pairs_in_flight x stacked_pairs`` pairs: the smallest capacity class that holds all of
        them (a group the graphs cannot take goes through the eager step pair by pair -- on EVERY rank when there are
        several).  Returns [(desc, det, acc, d_pos, d_neg)] per pair, readable on the current stream."""
        if not self._captured:
            self._capture_classes(items[0])
        e = self._group_class(items)
        self._step_no = getattr(self, '_step_no', 0) + 1
        ok = self._agreed(self._step_no, e is not None)
        nxt_e = self._group_class(next_items)
        if next_items is not None:
            self._post_agreement(self._step_no + 1, nxt_e is not None)
        if not ok:
            return self._eager_steps(items)
        feed = (lambda g: g) if self.lanes > 1 else (lambda g: tuple(g))
        if nxt_e is e:
            outs = e.step_graph(feed(items), feed(next_items))
        else:
            if nxt_e is not None:
                nxt_e.preload(feed(next_items))
            outs = e.step_graph(feed(items), TrainStep.NO_PREFETCH)
        res = []
        if self.lanes > 1:
            e.make_visible()
            lanes = list(zip(e.engines, outs))
        else:
            lanes = [(e, outs)]
        for lane, out in lanes:
            fp, an = lane.last_distances
            desc, det, acc = out[1].reshape(-1), out[2].reshape(-1), out[3].reshape(-1)
            fpm, anm = fp.reshape(self.stack, -1).mean(dim=1), an.reshape(self.stack, -1).mean(dim=1)
            for q in range(self.stack):
                res.append((desc[q], det[q], acc[q], fpm[q], anm[q]))
        if self.lanes > 1:
            e.resync()   # the lanes' next replays (which overwrite these static outputs three steps on) follow the reads
        self._rerun_overflowed()      # a flagged pair and the pairs that shared its (skipped) step: eager, one by one
        return res

    def _one_step(self, item, next_item):
        eng = self.engine
        if self.use_graph:
            if not self._captured:
                self._capture_classes(item)
            e = self._class_of(item)
            self._step_no = getattr(self, '_step_no', 0) + 1
            ok = self._agreed(self._step_no, e is not None)
            nxt_e = self._class_of(next_item) if next_item is not None else None
            if next_item is not None:
                self._post_agreement(self._step_no + 1, nxt_e is not None)
            if ok:
                if nxt_e is e:
                    out = e.step_graph(item, next_item)
                else:
                    if nxt_e is not None:
                        nxt_e.preload(next_item)       # the next pair belongs to another class: its sets, its graph
                    out = e.step_graph(item, TrainStep.NO_PREFETCH)
                eng.last_distances = e.last_distances
                self._rerun_overflowed()
                return out
        return eng.step(item)

    def train(self):
        self.model.train()
        for epoch in range(self.start_epoch, self.max_epoch):
            self.train_epoch(epoch + 1)
            res = self.evaluate(epoch + 1)
            if res['desc_loss'] < self.best_loss:
                self.best_loss = res['desc_loss']
                self._snapshot(epoch + 1, 'best_loss')
            if res['accuracy'] > self.best_acc:
                self.best_acc = res['accuracy']
                self._snapshot(epoch + 1, 'best_acc')
            for k, v in res.items():
                self.writer.add_scalar('val/%s' % k, v, epoch + 1)
            if (epoch + 1) % self.scheduler_interval == 0:
                self.scheduler.step()
            if (epoch + 1) % self.snapshot_interval == 0:
                self._snapshot(epoch + 1)
        if self.rank == 0:
            print("Training finish!... save training results")

    def train_epoch(self, epoch):
        ds = self.train_loader.dataset
        order = self._order(self.train_loader, epoch)
        P = self.group
        num_iter = min(self.training_max_iter,
                       len(ds) // max(1, getattr(self.train_loader, 'batch_size', 1)) // self.world // P)
        meters = _Meters(self.device)

        def group(it):      # pair k of step `it` on this rank (P = 1: the reference's one pair per step)
            return [self._fetch(ds, order[(it * P + k) * self.world + self.rank]) for k in range(P)]
        items = group(0) if num_iter else None
        for it in range(num_iter):
            nxt = group(it + 1) if it + 1 < num_iter else None
            if P == 1:
                _, desc, det, acc = self._one_step(items[0], nxt[0] if nxt else None)
                fp, an = self.engine.last_distances
                meters.update(desc, det, acc, fp.mean(), an.mean())
            else:
                for stats in self._lanes_step(items, nxt):
                    meters.update(*stats)
            items = nxt
            if (it + 1) % self.log_interval == 0 and self.verbose:
                avg = meters.averages()
                self._report_skipped()
                if self.rank == 0:
                    cur = num_iter * (epoch - 1) + it
                    for tag, key in (('Desc_Loss', 'desc_loss'), ('Det_Loss', 'det_loss'), ('D_pos', 'd_pos'),
                                     ('D_neg', 'd_neg'), ('Accuracy', 'accuracy')):
                        self.writer.add_scalar('train/' + tag, avg[key], cur)
                    print("Epoch: %d [%4d/%d] desc loss: %.2f det loss: %.2f acc:  %.2f d_pos: %.2f d_neg: %.2f "
                          "lr: %.3g skipped steps: %d" % (epoch, it + 1, num_iter, avg['desc_loss'], avg['det_loss'],
                                                          avg['accuracy'], avg['d_pos'], avg['d_neg'],
                                                          self._get_lr(), int(self.optimizer.skipped)))
        avg = meters.averages()
        self._report_skipped()
        if self.group > 1 and self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)      # the last joint update, before evaluation / snapshots
        if self.rank == 0:
            print("Epoch %d: Desc Loss: %.2f, Det Loss : %.2f, Accuracy: %.2f, D_pos: %.2f, D_neg: %.2f" % (
                epoch, avg['desc_loss'], avg['det_loss'], avg['accuracy'], avg['d_pos'], avg['d_neg']))
        return avg

    def evaluate(self, epoch):
        loader = self.val_loader if self.val_loader is not None else self.train_loader
        ds = loader.dataset
        num_iter = min(self.val_max_iter, len(ds) // max(1, getattr(loader, 'batch_size', 1)))
        meters = _Meters(self.device)
        for it in range(self.rank, num_iter, self.world):
            _, desc, det, acc, d_pos, d_neg = self.engine.evaluate(self._fetch(ds, it))
            meters.update(desc, det, acc, d_pos, d_neg)
        if self.world > 1:
            cnt = torch.tensor([float(meters.count)], dtype=torch.float64, device=self.device)
            dist.all_reduce(meters.sum)
            dist.all_reduce(cnt)
            meters.count = int(cnt.item())
        res = meters.averages()
        if self.rank == 0:
            print("Evaluation: Epoch %d: Desc Loss %s, Det Loss %s, Accuracy %s" % (
                epoch, res['desc_loss'], res['det_loss'], res['accuracy']))
        return res

    # ---------------------------------------------------------------------------------------------------------
    def _snapshot(self, epoch, name=None):
        if self.rank != 0 or not self.save_dir:
            return None
        os.makedirs(self.save_dir, exist_ok=True)
        state = {
            'epoch': epoch,
            # parameters are views into one flat buffer: clone so that each entry owns exactly its own storage
            'state_dict': {k: v.detach().clone() for k, v in self.model.state_dict().items()},
            'optimizer': self.optimizer.state_dict(),
            'scheduler': self.scheduler.state_dict(),
            'best_loss': self.best_loss,
        }
        filename = os.path.join(self.save_dir, 'model_%s.pth' % (epoch if name is None else name))
        print("Save model to %s" % filename)
        torch.save(state, filename)
        return filename

    def _load_pretrain(self, resume):
        if not os.path.isfile(resume):
            raise ValueError("=> no checkpoint found at '%s'" % resume)
        print("=> loading checkpoint %s" % resume)
        state = torch.load(resume, map_location=self.device, weights_only=True)
        self.start_epoch = state['epoch']
        self.model.load_state_dict(state['state_dict'])   # copies into the flat parameter buffer (views stay valid)
        self.scheduler.load_state_dict(state['scheduler'])
        self.optimizer.load_state_dict(state['optimizer'])
        self.best_loss = state['best_loss']

    def _get_lr(self, group=0):
        return self.optimizer.param_groups[group]['lr']
