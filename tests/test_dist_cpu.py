"""CPU, 2 processes over gloo: the data-parallel exchange of the training step (flat gradient buffer, bucketed
all-reduce mean, NaN/Inf guard decided on the REDUCED gradient so every rank takes the same decision)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from d3feat_pytorch_amd.train import FlatParams, GuardedSGD, allreduce_mean_
    torch.manual_seed(0)  # identical initial model on every rank
    model = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.LeakyReLU(0.1), torch.nn.Linear(13, 5))
    flat = FlatParams(model)
    opt = GuardedSGD(flat, lr=0.1, momentum=0.9, weight_decay=1e-3)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.LeakyReLU(0.1), torch.nn.Linear(13, 5))
    ref.load_state_dict(model.state_dict())
    ref_opt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    ok = True
    for step in range(4):
        # every rank sees its own data; the reference model sees the concatenation of both ranks' data
        g = torch.Generator().manual_seed(100 * step)
        xs = [torch.randn(6, 7, generator=g) for _ in range(world)]
        ys = [torch.randn(6, 5, generator=g) for _ in range(world)]
        flat.zero_grad()
        loss = ((model(xs[rank]) - ys[rank]) ** 2).mean()
        loss.backward()
        flat.gather_grads()
        poison = step == 2 and rank == 1  # a non-finite gradient on ONE rank must stop the step on ALL ranks
        if poison:
            flat.grad[3] = float("nan")
        allreduce_mean_(flat.grad, world, n_buckets=3)
        stepped = bool(opt.step().item())
        ref_opt.zero_grad()
        rl = sum(((ref(xs[r]) - ys[r]) ** 2).mean() for r in range(world)) / world
        rl.backward()
        if step != 2:
            ref_opt.step()
        ok &= stepped == (step != 2)
        for p, q in zip(model.parameters(), ref.parameters()):
            ok &= torch.allclose(p, q, atol=1e-6)
    # all ranks hold identical parameters
    mine = flat.data.clone()
    other = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    ok &= all(torch.equal(o, other[0]) for o in other)
    ok &= int(opt.skipped.item()) == 1
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def _split_worker(rank, world, port, ret):
    """TrainStep._exchange_and_step (the split-backward path of the multi-GPU step): deep bucket in 3 chunks + shallow
    bucket, mean folded into the optimizer's gradient scale, pair-status poison -- against one plain all-reduce."""
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from d3feat_pytorch_amd.train import FlatParams, GuardedSGD, TrainStep
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.LeakyReLU(0.1), torch.nn.Linear(13, 11),
                                torch.nn.LeakyReLU(0.1), torch.nn.Linear(11, 5))
    ref = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.LeakyReLU(0.1), torch.nn.Linear(13, 11),
                              torch.nn.LeakyReLU(0.1), torch.nn.Linear(11, 5))
    ref.load_state_dict(model.state_dict())
    eng = TrainStep.__new__(TrainStep)
    eng.world = world
    eng.flat = FlatParams(model)
    eng.opt = GuardedSGD(eng.flat, lr=0.1, momentum=0.9, weight_decay=1e-3)
    eng.opt.grad_scale = 1.0 / world
    eng.numel_shallow = sum(p.numel() for p in model[0].parameters())      # "fine levels" = the first layer

    def host_poison(grad, pair_status):
        """d3f_poison_gradient_if_status restated on host tensors (the product calls the HIP kernel): first element of
        the bucket -> NaN when the status word is set, flags OR-ed into state[2], skip count into state[3]."""
        bad = pair_status.reshape(-1)[0] != 0
        grad[0] = torch.where(bad, torch.full((), float('nan')), grad[0])
        eng.opt.state[2] |= pair_status.reshape(-1)[0].to(eng.opt.state.dtype)
        eng.opt.state[3] += bad.to(eng.opt.state.dtype)
    eng._poison_if_flagged = host_poison
    ref_opt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    ok = True
    for step in range(4):
        g = torch.Generator().manual_seed(7 * step)
        xs = [torch.randn(6, 7, generator=g) for _ in range(world)]
        ys = [torch.randn(6, 5, generator=g) for _ in range(world)]
        eng.flat.zero_grad()
        ((model(xs[rank]) - ys[rank]) ** 2).mean().backward()
        eng.flat.gather_grads()
        status = torch.tensor([8 if (step == 1 and rank == 1) else 0], dtype=torch.int32)   # rank 1's pair overflowed
        eng._exchange_and_step(lambda: None, lambda: None, pair_status=status)
        ref_opt.zero_grad()
        (sum(((ref(xs[r]) - ys[r]) ** 2).mean() for r in range(world)) / world).backward()
        if step != 1:
            ref_opt.step()
        for p, q in zip(model.parameters(), ref.parameters()):
            ok &= torch.allclose(p, q, atol=1e-6)
    ok &= int(eng.opt.skipped.item()) == 1                       # every rank skipped the same step
    ok &= int(eng.opt.state[3]) == (1 if rank == 1 else 0) and int(eng.opt.state[2]) == (8 if rank == 1 else 0)
    mine = eng.flat.data.clone()
    other = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    ok &= all(torch.equal(o, other[0]) for o in other)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def _agree_worker(rank, world, port, ret):
    """trainer.Trainer._agree (logical AND over the ranks) and the stacked / laned join arithmetic on host tensors:
    every rank sums the deep and shallow buckets of its lanes, the buckets are all-reduced (3 deep chunks + 1 shallow, as
    PairLanes.step_graph does), ONE guarded update with scale 1 / (pairs x ranks) -- against a single model that sees
    the mean gradient of all lanes x ranks batches."""
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from d3feat_pytorch_amd.train import FlatParams, GuardedSGD
    from d3feat_pytorch_amd.trainer import Trainer
    tr = Trainer.__new__(Trainer)
    tr.world, tr.device, tr._agree_stream = world, torch.device('cpu'), None
    ok = tr._agree(True) is True and tr._agree(rank == 0) is False and tr._agree(False) is False
    # the agreement of step k + 1 is POSTED during step k (asynchronous all-reduce) and READ when step k + 1 starts
    # (round 5: no collective + .item() inside the step that needs the answer); an unposted step falls back to _agree,
    # a posted-but-unread one is drained so that the ranks' collectives stay in step
    ok &= tr._agreed(1, True) is True                      # nothing posted: synchronous
    tr._post_agreement(2, rank == 0)                       # rank 1 cannot take the graph path for step 2
    ok &= tr._agreed(2, True) is False                     # ... so nobody does (the local flag at read time is not used)
    tr._post_agreement(3, True)
    ok &= tr._agreed(3, False) is True                     # the posted (common) decision wins
    tr._post_agreement(5, True)
    ok &= tr._agreed(4, rank == 1) is False                # a stale post is waited for, then a fresh agreement is made
    ok &= getattr(tr, '_pending_agree', None) is None
    torch.manual_seed(0)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.LeakyReLU(0.1), torch.nn.Linear(13, 5))   # noqa: E731
    model, ref = mk(), mk()
    ref.load_state_dict(model.state_dict())
    flat = FlatParams(model)
    lanes, stack = 2, 2                      # 2 lanes x 2 stacked pairs per rank: 8 "pairs" per update over 2 ranks
    flat.add_lane()
    opt = GuardedSGD(flat, lr=0.1, momentum=0.9, weight_decay=1e-3)
    ref_opt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    ns = sum(p.numel() for p in model[0].parameters())        # "shallow" bucket = the first layer
    for step in range(3):
        g = torch.Generator().manual_seed(11 * step)
        xs = [[[torch.randn(4, 7, generator=g) for _ in range(stack)] for _ in range(lanes)] for _ in range(world)]
        ys = [[[torch.randn(4, 5, generator=g) for _ in range(stack)] for _ in range(lanes)] for _ in range(world)]
        for lane in range(lanes):            # a lane's buffer = the SUM over its stack (total loss = sum of pair losses)
            flat.bind(lane)
            flat.zero_grad()
            sum(((model(xs[rank][lane][q]) - ys[rank][lane][q]) ** 2).mean() for q in range(stack)).backward()
            flat.gather_grads()
        flat.bind(0)
        grads = [flat.lanes[k][0] for k in range(lanes)]
        opt.use_grad_scale(1.0 / (lanes * stack * world))
        deep = grads[0][ns:]
        for other in grads[1:]:
            deep.add_(other[ns:])
        nb = 3
        chunk = (deep.numel() + nb - 1) // nb
        works = [dist.all_reduce(deep[b * chunk:min(deep.numel(), (b + 1) * chunk)], async_op=True) for b in range(nb)]
        shallow = grads[0][:ns]
        for other in grads[1:]:
            shallow.add_(other[:ns])
        works.append(dist.all_reduce(shallow, async_op=True))
        for w in works:
            w.wait()
        ok &= bool(opt.step(grads=grads[:1]))
        ref_opt.zero_grad()
        (sum(((ref(xs[r][k][q]) - ys[r][k][q]) ** 2).mean() for r in range(world) for k in range(lanes)
             for q in range(stack)) / (world * lanes * stack)).backward()
        ref_opt.step()
        for p_, q_ in zip(model.parameters(), ref.parameters()):
            ok &= torch.allclose(p_, q_, atol=1e-6)
    # an eager one-pair step afterwards runs at ITS scale (it used to inherit 1 / pairs)
    opt.use_grad_scale(1.0 / world)
    ok &= opt.grad_scale == 1.0 / world
    mine = flat.data.clone()
    other = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    ok &= all(torch.equal(o, other[0]) for o in other)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_rank_agreement_and_laned_stacked_join_over_two_ranks():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_agree_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_stack_inputs_and_gradient_scale_host_logic():
    """What a stacked step is handed (train.is_stack / same_item / TrainStep.pairs_of, PairLanes.deal) and
    GuardedSGD.use_grad_scale on host tensors."""
    sys.path.insert(0, REPO)
    from d3feat_pytorch_amd.train import FlatParams, GuardedSGD, PairLanes, TrainStep, is_stack, same_item
    a, b, c = (torch.zeros(3, 3),) * 6, (torch.ones(2, 3),) * 6, (torch.ones(4, 3),) * 6
    assert not is_stack(a) and is_stack((a, b)) and is_stack([a])
    assert same_item(a, a) and not same_item(a, b) and same_item((a, b), (a, b)) and not same_item((a, b), (b, a))
    assert not same_item((a, b), (a, b, c)) and not same_item(None, a) and not same_item((a, b), a)
    assert TrainStep.pairs_of(a) == [a] and TrainStep.pairs_of((a, b)) == [a, b]
    lanes = PairLanes.__new__(PairLanes)
    lanes.P, lanes.Q = 2, 3
    assert lanes.deal([a, b, c, c, b, a]) == [(a, b, c), (c, b, a)] and lanes.pairs_per_step == 6
    with pytest.raises(ValueError):
        lanes.deal([a, b, c])
    lanes.Q = 1
    assert lanes.deal([a, b]) == [a, b]
    m = torch.nn.Linear(4, 2)
    f = FlatParams(m)
    o = GuardedSGD(f, lr=0.5, momentum=0.0, weight_decay=0.0)
    f.grad.fill_(1.0)
    before = f.data.clone()
    o.use_grad_scale(0.25)
    assert o.grad_scale == 0.25 and bool(o.step())
    assert torch.allclose(f.data, before - 0.5 * 0.25)
    o.use_grad_scale(0.25)            # asking for the scale in effect is free
    o.use_grad_scale(1.0)
    before = f.data.clone()
    assert bool(o.step()) and torch.allclose(f.data, before - 0.5)


def test_split_bucket_exchange_equals_one_allreduce_and_status_poison_is_collective():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_split_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_two_rank_gradient_exchange_and_guard():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_flat_params_views_track_module():
    sys.path.insert(0, REPO)
    from d3feat_pytorch_amd.train import FlatParams
    m = torch.nn.Linear(3, 2)
    f = FlatParams(m)
    assert f.numel == 8 and m.weight.data_ptr() == f.data.data_ptr()
    (m(torch.ones(1, 3)).sum()).backward()
    f.gather_grads()
    assert torch.equal(f.grad[:6].view(2, 3), m.weight.grad) and float(f.grad.abs().sum()) > 0
    f.zero_grad()
    assert m.weight.grad is None
    assert float(f.gather_grads().abs().sum()) == 0.0  # parameters without a gradient contribute zeros


def test_gradient_lanes_on_host_tensors():
    """FlatParams.add_lane / bind and GuardedSGD.step(grads=[...]) on host tensors: the step uses the fixed-order sum of
    the lane buffers (the arithmetic d3f_sgd_guarded_step_lanes does on the device)."""
    sys.path.insert(0, REPO)
    from d3feat_pytorch_amd.train import FlatParams, GuardedSGD
    torch.manual_seed(0)
    m, twin = torch.nn.Linear(5, 3), torch.nn.Linear(5, 3)
    twin.load_state_dict(m.state_dict())
    f, tf = FlatParams(m), FlatParams(twin)
    o, to = GuardedSGD(f, lr=0.1, momentum=0.9, weight_decay=1e-3), GuardedSGD(tf, lr=0.1, momentum=0.9, weight_decay=1e-3)
    o.grad_scale = to.grad_scale = 0.5
    assert f.add_lane() == 1 and len(f.lanes) == 2 and f.lanes[1][0].shape == f.grad.shape
    f.bind(1)
    assert f.grad is f.lanes[1][0] and f.slots[0].data_ptr() == f.lanes[1][0].data_ptr()
    f.bind(0)
    assert f.grad is f.lanes[0][0]
    for step in range(3):
        a, b = torch.randn(f.numel), torch.randn(f.numel)
        f.lanes[0][0].copy_(a)
        f.lanes[1][0].copy_(b)
        tf.grad.copy_(a + b)
        assert bool(o.step(grads=[f.lanes[0][0], f.lanes[1][0]])) and bool(to.step())
        assert torch.equal(f.data, tf.data) and torch.equal(o.buf, to.buf)
    f.lanes[1][0][2] = float('nan')
    before = f.data.clone()
    assert not bool(o.step(grads=[f.lanes[0][0], f.lanes[1][0]])) and torch.equal(before, f.data)


def test_bench_launched_plainly_with_gpus_2_runs_two_ranks():
    """`python bench.py --gpus 2` without a launcher must yield n_gpus == 2 (it spawns its ranks through
    torch.distributed.run) and a rank count that disagrees with --gpus must fail instead of running 1 rank; the
    rendezvous-only mode does exactly the launch + process-group part, over gloo, without a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--rendezvous-only"], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 2, "ranks": 2, "rendezvous_only": True}
    # a launcher that provides a different world size than --gpus asks for is an error, not a silent 1-rank run
    bad = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--rendezvous-only"], cwd=REPO,
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)
