"""CPU: epoch-loop host logic -- snapshot interchange with torch.optim.SGD / ExponentialLR (the reference's snapshot
layout, trainer.py:193-218), the learning-rate recursion, the scalar log.  No kernel launches."""
import json
import os

import numpy as np
import torch

from d3feat_pytorch_amd import config as cfgmod
from d3feat_pytorch_amd.models.architectures import KPFCNN
from d3feat_pytorch_amd.train import FlatParams, GuardedSGD
from d3feat_pytorch_amd.trainer import ExponentialLR, ScalarLog, Trainer


def _small_model(seed=0):
    np.random.seed(seed)
    torch.manual_seed(seed)
    return KPFCNN(cfgmod.default_config(first_features_dim=16))


def _fake_grads(model, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(p.shape, generator=g) * 1e-2 if p.requires_grad else None for p in model.parameters()]


def _reference_side(model, steps, gamma):
    """What the reference's training script builds (training_3DMatch.py:62-81), stepped `steps` epochs."""
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.98, weight_decay=1e-6)
    sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=gamma)
    for s in range(steps):
        for p, g in zip(model.parameters(), _fake_grads(model, s)):
            p.grad = g
        opt.step()
        sch.step()
    return opt, sch


class _Loader:
    def __init__(self):
        self.dataset, self.batch_size, self.shuffle, self.limits = [], 1, False, [5, 5, 5, 5, 5]


def _args(tmp, **kw):
    cfg = cfgmod.default_config(first_features_dim=16)
    cfg.max_epoch, cfg.save_dir, cfg.tboard_dir, cfg.device, cfg.graph = 2, str(tmp / 'snap'), str(tmp / 'tb'), 'cpu', False
    cfg.train_loader = _Loader()
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def test_reference_snapshot_loads_and_training_continues_identically(tmp_path):
    gamma = 0.1 ** (1 / 80)
    ref_model = _small_model()
    ref_opt, ref_sch = _reference_side(ref_model, steps=3, gamma=gamma)
    path = tmp_path / 'model_3.pth'
    torch.save({'epoch': 3, 'state_dict': ref_model.state_dict(), 'optimizer': ref_opt.state_dict(),
                'scheduler': ref_sch.state_dict(), 'best_loss': 1.25}, path)

    tr = Trainer(_args(tmp_path, model=_small_model(seed=7), pretrain=str(path)))
    assert tr.start_epoch == 3 and tr.best_loss == 1.25
    assert tr._get_lr() == ref_opt.param_groups[0]['lr'] == ref_sch.get_last_lr()[0]
    for (k, a), b in zip(tr.model.state_dict().items(), ref_model.state_dict().values()):
        assert torch.equal(a, b), k
    # parameters are still views of the flat buffer after load_state_dict
    assert all(p.data_ptr() >= tr.engine.flat.data.data_ptr() for p in tr.engine.flat.params)
    assert tr.engine.flat.params[0].untyped_storage().data_ptr() == tr.engine.flat.data.untyped_storage().data_ptr()

    # one more epoch on both sides with the same gradients: same parameters, same momentum, same rate
    grads = _fake_grads(ref_model, 99)
    for p, g in zip(ref_model.parameters(), grads):
        p.grad = g
    ref_opt.step()
    ref_sch.step()
    flat = tr.engine.flat
    off = 0
    for p, g in zip(tr.model.parameters(), grads):
        if p.requires_grad:
            flat.grad[off:off + p.numel()] = g.reshape(-1)
            off += p.numel()
    assert bool(tr.optimizer.step())
    tr.scheduler.step()
    for (k, a), b in zip(tr.model.state_dict().items(), ref_model.state_dict().values()):
        assert torch.allclose(a, b, rtol=0, atol=1e-7), k
    assert tr._get_lr() == ref_opt.param_groups[0]['lr']

    # ... and back: our snapshot loads into the reference's optimizer / scheduler objects
    out = tr._snapshot(4)
    state = torch.load(out, weights_only=True)
    assert set(state) == {'epoch', 'state_dict', 'optimizer', 'scheduler', 'best_loss'}
    m2 = _small_model(seed=11)
    m2.load_state_dict(state['state_dict'])
    o2 = torch.optim.SGD(m2.parameters(), lr=1.0, momentum=0.5)
    s2 = torch.optim.lr_scheduler.ExponentialLR(o2, gamma=0.5)
    s2.load_state_dict(state['scheduler'])
    o2.load_state_dict(state['optimizer'])
    assert o2.param_groups[0]['lr'] == ref_opt.param_groups[0]['lr'] and o2.param_groups[0]['momentum'] == 0.98
    assert s2.last_epoch == ref_sch.last_epoch and s2.gamma == gamma
    for i, p in enumerate(ref_model.parameters()):
        if p.requires_grad:
            a = o2.state[list(m2.parameters())[i]]['momentum_buffer']
            assert torch.allclose(a, ref_opt.state[p]['momentum_buffer'], rtol=0, atol=1e-7)
    # every saved tensor owns its storage (no 100 MB flat buffer behind a 16-element bias)
    assert all(v.untyped_storage().nbytes() == v.numel() * v.element_size() for v in state['state_dict'].values())


def test_exponential_lr_follows_torch_recursion_bit_for_bit():
    lin = torch.nn.Linear(3, 2)
    opt = GuardedSGD(FlatParams(lin), lr=0.01)
    ours = ExponentialLR(opt, gamma=0.1 ** (1 / 80))
    t_opt = torch.optim.SGD(torch.nn.Linear(3, 2).parameters(), lr=0.01, momentum=0.98)
    theirs = torch.optim.lr_scheduler.ExponentialLR(t_opt, gamma=0.1 ** (1 / 80))
    for _ in range(200):
        t_opt.step()
        theirs.step()
        ours.step()
        assert ours.get_last_lr() == theirs.get_last_lr()
    assert float(opt.hyper[0]) == np.float32(ours.get_last_lr()[0])   # what the kernel will read
    sd = ours.state_dict()
    assert sd['last_epoch'] == theirs.state_dict()['last_epoch'] and sd['_step_count'] == theirs.state_dict()['_step_count']


def test_optimizer_state_rejects_foreign_layouts():
    lin = torch.nn.Linear(3, 2)
    opt = GuardedSGD(FlatParams(lin))
    sd = opt.state_dict()
    assert set(sd['state']) == {0, 1} and sd['param_groups'][0]['params'] == [0, 1]
    bad = {'state': {}, 'param_groups': [dict(sd['param_groups'][0], nesterov=True)]}
    for broken in (bad, {'state': {}, 'param_groups': [dict(sd['param_groups'][0], params=[0])]},
                   {'state': {0: {'momentum_buffer': torch.zeros(5)}}, 'param_groups': sd['param_groups']}):
        try:
            opt.load_state_dict(broken)
        except ValueError:
            continue
        raise AssertionError("accepted %r" % (broken,))


def test_scalar_log_and_missing_checkpoint(tmp_path):
    log = ScalarLog(str(tmp_path / 'tb'))
    log.add_scalar('val/accuracy', 12.5, 3)
    rows = [json.loads(l) for l in open(os.path.join(str(tmp_path / 'tb'), 'scalars.jsonl'))]
    assert rows == [{'tag': 'val/accuracy', 'value': 12.5, 'step': 3}]
    try:
        Trainer(_args(tmp_path, model=_small_model(), pretrain=str(tmp_path / 'nope.pth')))
    except ValueError as e:
        assert 'no checkpoint' in str(e)
    else:
        raise AssertionError


def test_gradient_scale_is_the_data_parallel_mean():
    a, b = torch.nn.Linear(5, 3), torch.nn.Linear(5, 3)
    b.load_state_dict(a.state_dict())
    fa, fb = FlatParams(a), FlatParams(b)
    oa, ob = GuardedSGD(fa, lr=0.1, momentum=0.9, weight_decay=1e-3), GuardedSGD(fb, lr=0.1, momentum=0.9, weight_decay=1e-3)
    oa.grad_scale = 0.25
    for s in range(3):
        g = torch.randn(fa.numel, generator=torch.Generator().manual_seed(s))
        fa.grad.copy_(g)            # the SUM over 4 ranks
        fb.grad.copy_(g * 0.25)     # the mean, scaled beforehand
        assert bool(oa.step()) and bool(ob.step())
        assert torch.equal(fa.data, fb.data)


def test_trainer_reads_the_loaders_shuffle_request():
    """A torch DataLoader built with shuffle=True carries a RandomSampler; our pair loader has a .shuffle attribute."""
    import torch.utils.data as tud
    from d3feat_pytorch_amd.trainer import Trainer
    ds = list(range(7))
    assert Trainer._shuffles(None, tud.DataLoader(ds, shuffle=True)) is True
    assert Trainer._shuffles(None, tud.DataLoader(ds, shuffle=False)) is False

    class L:
        shuffle = True
    assert Trainer._shuffles(None, L()) is True
    # a DistributedSampler shuffles only when it was built to (reference-style loaders with shuffle=False keep the order)
    from torch.utils.data.distributed import DistributedSampler
    for flag in (True, False):
        smp = DistributedSampler(ds, num_replicas=2, rank=0, shuffle=flag)
        assert Trainer._shuffles(None, tud.DataLoader(ds, sampler=smp)) is flag


def test_pairs_in_flight_needs_the_graph_path_and_groups_the_epoch(tmp_path, capsys):
    """``pairs_in_flight`` > 1 (several pairs per optimizer step, train.PairLanes) is a property of the hipGraph path:
    without it the trainer says so and trains one pair per step; with it an epoch is cut into groups of that many pairs
    per rank (the tail that does not fill a group is left out, like a DataLoader's drop_last)."""
    from d3feat_pytorch_amd.trainer import Trainer
    tr = Trainer(_args(tmp_path, model=_small_model(), pairs_in_flight=3))
    assert tr.lanes == 1 and tr.group == 1 and "pairs_in_flight=3 / stacked_pairs=1 need the hipGraph path" in \
        capsys.readouterr().out
    tr = Trainer(_args(tmp_path, model=_small_model(), stacked_pairs=4))
    assert tr.stack == 1 and tr.group == 1 and "stacked_pairs=4 need the hipGraph path" in capsys.readouterr().out
    one = Trainer(_args(tmp_path, model=_small_model()))
    assert one.lanes == 1 and one.stack == 1 and one.group == 1              # the reference's one pair per step

    class _Probe(Trainer):       # the epoch loop's grouping, without a device: record what each step is handed
        def __init__(self, n, lanes, stack=1):
            self.lanes, self.stack, self.group = lanes, stack, lanes * stack
            self.world, self.rank, self.training_max_iter = 1, 0, 10 ** 9
            self.train_loader = type('L', (), {'dataset': list(range(n)), 'batch_size': 1, 'shuffle': False})()
            self.config, self.device, self.verbose, self.log_interval = type('C', (), {})(), torch.device('cpu'), False, 100
            self.steps = []

        def _fetch(self, ds, i):
            return int(i)

        def _lanes_step(self, items, nxt):
            self.steps.append((list(items), None if nxt is None else list(nxt)))
            z = torch.zeros(())
            return [(z, z, z, z, z) for _ in items]

        def _report_skipped(self):
            return 0
    p = _Probe(11, 4)
    p.train_epoch(1)
    assert [s[0] for s in p.steps] == [[0, 1, 2, 3], [4, 5, 6, 7]] and p.steps[0][1] == [4, 5, 6, 7] and p.steps[1][1] is None
    p = _Probe(11, 2, stack=2)     # 2 graphs in flight x 2 stacked pairs: the same groups of four
    p.train_epoch(1)
    assert [s[0] for s in p.steps] == [[0, 1, 2, 3], [4, 5, 6, 7]]
