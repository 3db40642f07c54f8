"""Worker of test_gpu_model.py::test_two_rank_lanes_join_equals_the_eager_mean_gradient_step: one of two ranks that share
cuda:0 (gloo stands in for RCCL, which refuses two ranks on one device).  Every rank drives the REAL multi-rank join of
train.PairLanes.step_graph (two-stage lane graphs, deep buckets summed and all-reduced under stage 2, shallow bucket,
guard on the reduced gradient, one update; reference trainer.py:89-111 per rank + the data-parallel exchange of SURVEY
8e) over 2 lanes x 2 stacked pairs and checks, after every step, the parameters against an EAGER mean-gradient SGD step
over all ranks x lanes x stacked pairs: every rank recomputes the eager gradients of ITS pairs at the step's
parameters, the sums are all-reduced, and the torch arithmetic of GuardedSGD is applied to a copy of the parameters."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import d3feat_pytorch_amd  # noqa: F401,E402
from d3feat_pytorch_amd import config as cfgmod, synthetic  # noqa: E402
from d3feat_pytorch_amd.datasets import dataloader as dl  # noqa: E402
from d3feat_pytorch_amd.train import PairLanes, TrainStep  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    g = np.load(os.path.join(REPO, "tests", "golden", "s0_small.npz"))
    cfg = cfgmod.default_config(first_features_dim=16, num_node=64)
    limits = [int(x) for x in g['limits']]

    def gpu_subsample(points, lengths, dlen):
        p, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(points).to(dev), torch.as_tensor(lengths).to(dev),
                                                sampleDl=dlen)
        return p.cpu().numpy(), b.cpu().numpy()

    def to_dev(item):
        return tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in item)
    m = 64
    pool = [to_dev(synthetic.make_pair(11 + 10 * rank + 2 * k, 12 + 10 * rank + 2 * k, gpu_subsample, n_raw=20000 + 4000 * k,
                                       scale=0.12 + 0.02 * k, num_node=m)) for k in range(3)]
    n_lanes, stack = 2, 2
    n = n_lanes * stack

    def fresh(world_size):
        np.random.seed(0)
        torch.manual_seed(0)
        return TrainStep(cfg, limits, dev, world_size=world_size, seed=0)
    ts = fresh(world)                    # (broadcasts rank 0's state: identical replicas)
    ref = fresh(1)
    ref.flat.data.copy_(ts.flat.data)
    sizes = [[int(t.shape[0]) for t in ref.build_batch(it)['points']] for it in pool]
    local = torch.tensor([stack * max(s[l] for s in sizes) for l in range(5)], dtype=torch.int64)
    dist.all_reduce(local, op=dist.ReduceOp.MAX)          # every rank captures the same capacities
    caps = TrainStep.capacities_for([local.tolist()], slack=1.1)
    lanes = PairLanes(ts, n_lanes, stack=stack)
    assert lanes.split, "several ranks: the lanes run their two-stage form"
    lanes.enable_graph(caps, num_corr=m)
    steps = [[pool[(k + j) % 3] for j in range(n)] for k in range(3)]
    lanes.capture(tuple(steps[0]))
    torch.cuda.synchronize()
    assert torch.equal(ts.flat.data, ref.flat.data), "a lane capture must not move the parameters"
    buf = torch.zeros_like(ref.flat.data)
    lr, mom, wd = ref.opt.lr, ref.opt.momentum, ref.opt.weight_decay
    worst = 0.0
    for k, pairs in enumerate(steps):
        nxt = steps[k + 1] if k + 1 < len(steps) else None
        outs = lanes.step_graph(pairs, nxt)
        lanes.synchronize()
        torch.cuda.synchronize()
        gsum = torch.zeros_like(ref.flat.data)
        descs = []
        for it in pairs:                  # eager gradient of each of THIS rank's pairs at the step's parameters
            batch = ref.build_batch(it)
            batch['n0'] = int(it[0].shape[0])
            ref.flat.zero_grad()
            loss, desc, det, _ = ref.forward_loss(batch)
            torch.autograd.backward(loss, ref._seed(loss))
            gsum += ref.flat.gather_grads()
            descs.append(float(desc))
        got_desc = torch.cat([o[1].reshape(-1) for o in outs]).tolist()
        for q in range(n):
            assert abs(got_desc[q] - descs[q]) < 1e-4 * max(1.0, abs(descs[q])), (rank, k, q, got_desc[q], descs[q])
        total = gsum.cpu()
        dist.all_reduce(total, op=dist.ReduceOp.SUM)       # over the ranks: all ranks x lanes x stacked pairs
        gmean = total.to(dev) * (1.0 / (n * world))
        d = gmean + wd * ref.flat.data
        buf = buf * mom + d
        ref.flat.data.sub_(lr * buf)
        err = float((ts.flat.data - ref.flat.data).abs().max())
        bound = 1e-6 + 1e-4 * lr * float(buf.abs().max())
        worst = max(worst, err / bound)
        assert err < bound, (rank, k, err, bound)
        # replicas agree bit for bit after the joint update
        mine = ts.flat.data.double().sum().cpu()
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert all(float(b) == float(both[0]) for b in both), (rank, k, [float(b) for b in both])
    assert lanes.check_status() == (0, 0) and int(ts.opt.skipped) == 0
    assert ts.opt.grad_scale == 1.0 / (n * world)
    print("JOIN_OK rank %d: 3 joint updates of %d pairs match the eager mean-gradient step (worst %.2f of the bound)" % (
        rank, n * world, worst), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
