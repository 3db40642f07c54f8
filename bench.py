#!/usr/bin/env python
"""Headline benchmark: fragment-pairs/sec (fwd+bwd) on 3DMatch-shaped synthetic pairs -- BASELINE.json configs[2]:
full D3Feat U-Net forward+backward on one fragment pair (~19k + 19k points, 32-d descriptors), circle + detector loss,
radius search + grid subsampling ON THE DEVICE, SGD step included.

    python bench.py --gpus 1 --steps 20 --warmup 5            # default: 4 graphs in flight x 3 stacked pairs per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over lanes x stack pairs per GPU (every pair: pyramid build with 13 radius
searches + 4 voxel levels -> KPFCNN forward -> fused loss -> backward), one (all-reduce +) guarded SGD update on the mean
of their gradients; `one_pair_in_flight` = the reference's schedule, one pair per optimizer step.  Inputs (the two raw fragments, the sampled
correspondences and their distance matrix) are resident in HBM before the timed region.  Each rank processes its own
pairs (weak scaling); the only data-path collective is the RCCL gradient all-reduce.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
# one hardware queue per busy stream (see d3feat.pytorch_amd/__init__.py); before the HIP runtime starts
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
os.environ.setdefault("PYTORCH_TUNABLEOP_HIPBLASLT_ENABLED", "0")   # (same place: rocBLAS candidates only)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
F32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak (v_mfma_f32_16x16x4_f32), same guide


def dx_kernel_cost(Nq, Ns, H, Cin, Cout, K):
    """(algorithmic bytes, flops, unique bytes) of ONE launch of the scatter-form KPConv grad-input kernel.  Algorithmic =
    SURVEY.md 8d convention (gather/scatter-expanded: a row counts once per (query, neighbor) slot); unique = every
    operand tensor once (8d's lower-bound figure).  Cout == 0 marks the variant that reads gW = (g/nn) W^T from a GEMM."""
    common = 12 * Nq + 4 * Nq * H + 16 * Nq * H + 4 * Nq * H * Cin       # queries, index rows, packed supports, scatter rows
    uniq = 12 * Nq + 4 * Nq * H + 16 * Ns + 4 * Ns * Cin
    if Cout == 0:
        return common + 4 * Nq * K * Cin, 2 * Nq * H * K * Cin, uniq + 4 * Nq * K * Cin
    return (common + 4 * Nq * Cout + 4 * Nq + 4 * K * Cin * Cout, 2 * Nq * K * Cin * Cout + 2 * Nq * H * K * Cin,
            uniq + 4 * Nq * Cout + 4 * Nq + 4 * K * Cin * Cout)


def fwd_kernel_cost(Nq, Ns, H, Cin, Cout, K):
    uniq = 12 * Nq + 4 * Nq * H + 16 * Ns + 4 * Ns * Cin + 4 * K * Cin * Cout + 4 * Nq * Cout
    return kpconv_fwd_bytes(Nq, Ns, H, K, Cin, Cout), 2 * Nq * H * K * Cin + 2 * Nq * K * Cin * Cout, uniq


def gather_kernel_cost(edges, widths):
    """Cost model of the gather-form grad-input kernel (the forward operator on the transposed graph).
    ``edges[(Nq, Ns)]`` = TRUE number of (query, support) pairs of the table it transposes -- the valid entries of the
    forward table, counted on the device; the search-form transpose stores a superset (every in-radius point / the 2r
    upsampling row) that the kernel filters by membership BEFORE it gathers anything Cout-wide, so only true edges are
    charged.  Per edge: index 4 + query position 12 + 1/nn 4 + membership key 8 + gathered gradient row 4 Cout; per support
    row: position 12 + written grad_x row 4 Cin; weights once.  ``widths[(Nq, Ns)]`` = stored table width (unique bytes)."""
    def cost(Nq, Ns, H, Cin, Cout, K):
        E = edges.get((Nq, Ns), Nq * 42)
        W = widths.get((Nq, Ns), 96)
        b = Ns * (12 + 4 * Cin) + E * (4 + 12 + 4 + 8 + 4 * Cout) + 4 * K * Cin * Cout
        uniq = 12 * Ns + 4 * Ns * W + 12 * Nq + 4 * Nq + 8 * Nq + 4 * Nq * Cout + 4 * K * Cin * Cout + 4 * Ns * Cin
        return b, 2 * E * K * Cout + 2 * Ns * K * Cout * Cin, uniq
    return cost


def atb_kernel_cost(R, _ns, _h, M, N, _k):
    """A^T B weight gradient (linear.hip): both operands streamed once, the output block written once (SURVEY 8d counts
    a GEMM operand once: nothing is gathered here)."""
    b = 4 * R * (M + N) + 4 * M * N
    return b, 2 * R * M * N, b


def atb_group_cost(_n, mi_flop, ki_bytes, _tasks, _z, _k):
    """One GROUPED weight-gradient launch pair (linear.hip, round 6: atb_grouped_kernel + atb_grouped_reduce_kernel over
    every queued problem of a backward stage).  The library records the sums over its problems: 2 R M N in MiFLOP and the
    A^T B byte model above, 4 R (M + N) + 4 M N, in KiB."""
    b = ki_bytes * 1024.0
    return b, mi_flop * 1048576.0, b


def agg_fwd_kernel_cost(Nq, Ns, H, Cin, _cout, K):
    """Forward aggregation kernel (kpconv_aggregate.hip): SURVEY 8d's KPConv forward bytes without the weights and the
    output row, plus the written wf [Nq, K Cin]."""
    b = 12 * Nq + 4 * Nq * H + Nq * H * (16 + 4 * Cin) + 4 * Nq * K * Cin
    return b, 2 * Nq * H * K * Cin, 12 * Nq + 4 * Nq * H + 16 * Ns + 4 * Ns * Cin + 4 * Nq * K * Cin


def agg_rev_kernel_cost(edges):
    """Transposed aggregation over the exact-form reverse table: per TRUE edge the entry (16 B), 1/nn (4 B) and the
    gathered gradient row; per support row the written [K Cout] block."""
    def cost(Nq, Ns, W, _cin, Cout, K):
        E = edges.get((Nq, Ns), Nq * 42)
        b = E * (16 + 4 + 4 * Cout) + 4 * Ns * K * Cout
        return b, 2 * E * K * Cout, 16 * Ns * W + 4 * Nq + 4 * Nq * Cout + 4 * Ns * K * Cout
    return cost


KERNEL_NAMES = {4: "atb_partial_kernel + atb_reduce_kernel (weight gradients C = A^T B of the unary blocks and of KPConv "
                   "from the saved aggregation: reduction over the points spread over the chip, f32 MFMA, fixed-order sum; "
                   "round 5: both forms of the partial kernel -- direct loads / per-wave LDS rings fed by LDS-DMA, chosen "
                   "per shape -- and the second stage that also finishes the block's bias gradient)",
                7: "atb_grouped_kernel + atb_grouped_reduce_kernel (round 6: EVERY weight gradient C = A^T B of a backward "
                   "stage -- unary blocks and KPConv from the saved aggregation -- in ONE launch over all problems' (row "
                   "partition, output block) tasks, per-wave LDS rings fed by LDS-DMA, f32 MFMA, plus ONE launch that sums "
                   "all slabs in a fixed order and finishes the bias gradients; a 'launch' below = one such pair)",
                5: "kpconv_agg_fwd_kernel (KPConv neighbor aggregation wf = sum_h w x, registers -> HBM; contraction by GEMM)",
                6: "kpconv_agg_rev_kernel (KPConv grad-input aggregation over the reverse table, registers -> HBM; "
                   "contraction by GEMM, no atomics)",
                1: "kpconv_fwd_fused_kernel (KPConv forward: gather + influence + aggregation + contraction on f32 MFMA)",
                2: "kpconv_bwd_dx_kernel (KPConv grad-input, scatter form: gW tile on f32 MFMA + float atomics)",
                3: "kpconv_dx_gather_kernel (KPConv grad-input, gather form over the reverse neighbor table: "
                   "aggregation + W^T contraction on f32 MFMA, no atomics)"}


def timed_kernels(lib, run_steps, costs, n_steps):
    """HIP events on the launch stream around every launch of the hand-written KPConv kernels
    (d3f_debug_kernel_timing_*), grouped by kernel: {which: stats}."""
    import ctypes
    cap = 2048
    if lib.d3f_debug_kernel_timing_begin(-127, cap) != 0:
        return {}
    run_steps()
    torch.cuda.synchronize()
    ms = (ctypes.c_float * cap)()
    sh = (ctypes.c_int32 * (6 * cap))()
    n = min(lib.d3f_debug_kernel_timing_end(ms, sh, cap), cap)
    groups = {}
    for i in range(max(n, 0)):
        shape = [int(sh[6 * i + j]) for j in range(6)]
        which, shape[5] = shape[5] >> 8, shape[5] & 255
        b, f, u = costs[which](*shape)
        g = groups.setdefault(which, [0, 0.0, 0.0, 0.0, 0.0, []])
        g[0] += 1
        g[1] += ms[i]
        g[2] += b
        g[3] += f
        g[4] += u
        keys = ("problems", "MiFLOP", "KiB", "workgroups", "_", "__") if which == 7 else ("Nq", "Ns", "H", "Cin", "Cout", "K")
        g[5].append({"shape": dict(zip(keys, shape)), "us": round(ms[i] * 1e3, 2),
                     "bytes": int(b), "flops": int(f)})
    return {w: {"launches": g[0], "avg_us": g[1] / g[0] * 1e3, "us_per_step": g[1] / n_steps * 1e3,
                "bytes_per_launch": g[2] / g[0], "flops_per_launch": g[3] / g[0], "unique_bytes_per_launch": g[4] / g[0],
                "gbs": g[2] / (g[1] * 1e-3) / 1e9, "tflops": g[3] / (g[1] * 1e-3) / 1e12,
                "unique_gbs": g[4] / (g[1] * 1e-3) / 1e9, "per_launch": g[5][:len(g[5]) // n_steps]}
            for w, g in groups.items()}


def kpconv_fwd_bytes(Nq, Ns, H, K, Cin, Cout):
    """Algorithmic bytes of one KPConv forward (gather-expanded logical bytes, int32 indices)."""
    return 12 * Nq + 4 * Nq * H + Nq * H * (12 + 4 * Cin) + 4 * K * Cin * Cout + 180 + 4 * Nq * Cout


def kpconv_bwd_bytes(Nq, Ns, H, K, Cin, Cout):
    return kpconv_fwd_bytes(Nq, Ns, H, K, Cin, Cout) + 4 * Nq * Cout + 4 * Ns * Cin + 4 * K * Cin * Cout


class _RepeatPair(torch.utils.data.Dataset):
    """The same raw pair n times (the CPU baseline's DataLoader-mode leg)."""

    def __init__(self, item, n):
        self.item, self.n = item, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.item


def _one_thread(_worker_id):
    torch.set_num_threads(1)


def cpu_baseline(item, cfg, limits, budget_s=90.0):
    """The CPU oracle (C++ radius search / voxel subsampling restatement + PyTorch-CPU restatement of the network,
    losses, backward, SGD) timed on this host -- a reported baseline, not the thing shipped.  SURVEY 8d protocol:
    5 warm-up steps (the first at the default thread count, the others double as a sweep over intra-op thread counts:
    the all-cores default oversubscribes a many-core host) + 20 timed steps at the best count, median; serial s/pair =
    collate + forward + loss + backward + SGD.  Then the reference's OPERATING MODE (config.py:86, dataloader.py:225-237):
    a torch DataLoader with min(16, nproc) worker processes running the collate in front of the training process,
    measured, not computed.  ``budget_s`` bounds the whole leg (fewer timed steps are reported as such)."""
    from oracle import native as onat, ops_ref
    from d3feat_pytorch_amd.models.architectures import KPFCNN
    np.random.seed(0)
    torch.manual_seed(0)
    model = KPFCNN(cfg)
    sd = {k: v.detach().clone().requires_grad_(v.requires_grad and not k.endswith('kernel_points'))
          for k, v in model.state_dict().items()}
    for k, p in model.named_parameters():
        sd[k].requires_grad_(p.requires_grad)
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.SGD(params, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
    pts0, pts1, _, _, corr, dk = item
    corr_t, dk_t = torch.from_numpy(corr).long(), torch.from_numpy(dk)
    nproc = os.cpu_count() or 1

    def collate_list(list_data):   # the reference's collate_fn signature (batch size 1, dataloader.py:73)
        p0, p1 = list_data[0][0], list_data[0][1]
        batch = ops_ref.collate(p0, p1, cfg, limits, onat)
        batch['features'] = torch.ones((p0.shape[0] + p1.shape[0], 1))
        return batch

    def collate():
        t0 = time.time()
        batch = collate_list([item])
        return batch, time.time() - t0

    def net(batch):
        t0 = time.time()
        opt.zero_grad()
        feats, scores = ops_ref.kpfcnn_forward(sd, batch, cfg, training=True)
        n0 = pts0.shape[0]
        loss, _, _, _, dists = ops_ref.circle_loss(feats[corr_t[:, 0]], feats[corr_t[:, 1] + n0], dk_t)
        det = ops_ref.det_loss(dists, scores[corr_t[:, 0]], scores[corr_t[:, 1] + n0])
        (loss + det).backward()
        opt.step()
        return time.time() - t0

    t_start = time.time()
    default_threads = torch.get_num_threads()
    batch, t_col = collate()
    net(batch)                                            # warm-up 1 (allocator, thread pools)
    sweep = {}
    cand = sorted(set(t for t in (8, 16, 32, default_threads) if 1 <= t <= nproc))[:4]
    for th in cand:                                       # warm-ups 2..5 = the thread sweep
        torch.set_num_threads(th)
        sweep[th] = net(batch)
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    n_warm = 1 + len(sweep)
    while n_warm < 5:                                     # (hosts with few candidates: plain warm-ups up to 5)
        net(batch)
        n_warm += 1
    serial, nets, cols = [], [], [t_col]
    while len(serial) < 20 and (len(serial) < 3 or time.time() - t_start < 0.7 * budget_s):
        batch, tc = collate()
        tn = net(batch)
        cols.append(tc)
        nets.append(tn)
        serial.append(tc + tn)
    med, mnet, mcol = float(np.median(serial)), float(np.median(nets)), float(np.median(cols))
    # DataLoader mode, measured: worker processes collate (1 thread each), this process trains on `best` threads
    workers = min(16, nproc)
    pipe = {"collate_workers": workers}
    try:
        n_pipe = 2 + max(4, min(10, int((budget_s - (time.time() - t_start)) / max(mnet, 1e-3)) - 2))
        loader = torch.utils.data.DataLoader(_RepeatPair(item, n_pipe), batch_size=1, shuffle=False, num_workers=workers,
                                             collate_fn=collate_list, worker_init_fn=_one_thread, timeout=180)
        times, t_prev = [], None
        for i, b in enumerate(loader):
            net(b)
            now = time.time()
            if i >= 2:                # two steps to fill the pipeline
                times.append(now - t_prev)
            t_prev = now
        pipe.update({"pairs_per_s": round(1.0 / float(np.median(times)), 4), "timed_steps": len(times),
                     "s_per_pair": round(float(np.median(times)), 3)})
    except Exception as e:  # pragma: no cover - the serial figure must not depend on this leg
        pipe.update({"error": "%s: %s" % (type(e).__name__, e),
                     "pairs_per_s_computed": round(1.0 / max(mnet, mcol / workers), 4)})
    torch.set_num_threads(default_threads)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(1.0 / med, 4), "unit": "fragment-pairs/s", "cores": int(best), "kind": "port",
            "serial_s_per_pair": round(med, 3), "collate_s": round(mcol, 3), "network_s": round(mnet, 3),
            "warmup_steps": n_warm, "timed_steps": len(serial),
            "pipelined": pipe, "pipelined_pairs_per_s": pipe.get("pairs_per_s", pipe.get("pairs_per_s_computed")),
            "thread_sweep_s_per_step": {str(k): round(v, 3) for k, v in sorted(sweep.items())},
            "host": {"cpu": cpu_model, "logical_cores": nproc, "torch": torch.__version__},
            "sample": "%d warm-up + %d timed steps (median) of the same S1-class pair: CPU oracle collate (C++ cell-list "
                      "search + unordered_map voxel subsampling, 1 thread, %.2f s) + PyTorch-CPU fwd/loss/bwd/SGD on %d "
                      "intra-op threads (best of the warm-up sweep, %.2f s); value = serial; pipelined = measured with a "
                      "torch DataLoader of %d collate worker processes in front of the training process (the "
                      "reference's mode, config.py:86)" % (n_warm, len(serial), mcol, best, mnet, workers)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=8, help="distinct synthetic pairs per rank (cycled)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tuned-gemm", action="store_true", help="library-default GEMM kernels instead of the shipped "
                                                                 "TunableOp table (d3feat.pytorch_amd/tuned/)")
    ap.add_argument("--no-tune-missing", action="store_true",
                    help="library-GEMM shapes the shipped table lacks run on the library's default pick instead of being "
                         "tuned while the graphs are captured (kernel traces without thousands of tuning candidates)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--lanes", type=int, default=4,
                    help="network graphs in flight per GPU (train.PairLanes): each on streams and graphs of its own, one "
                         "optimizer step on the mean of all their pairs' gradients per step.  --lanes 1 --stack 1 = the "
                         "reference's one pair per optimizer step (also measured and reported otherwise: "
                         "one_pair_in_flight)")
    ap.add_argument("--stack", type=int, default=3,
                    help="fragment pairs STACKED into one pyramid + one network graph per lane (TrainStep stack): a step "
                         "trains on lanes x stack pairs, one optimizer step on the mean of their gradients")
    ap.add_argument("--quick", action="store_true", help="headline legs only (value, blocks, one_pair_in_flight): no "
                                                         "trainer-path / evaluation / matching / roofline / CPU legs")
    ap.add_argument("--cpu-budget", type=float, default=90.0)
    ap.add_argument("--blocks", type=int, default=5, help="extra timed blocks of --steps steps after the contract region "
                                                         "(median / min / max reported next to `value`)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch / rendezvous check only (no GPU work): every rank joins the process group and rank 0 "
                         "prints {n_gpus, ranks}; what tests/test_dist_cpu.py runs on a GPU-less host")
    args = ap.parse_args()
    # D3F_BENCH_WATCHDOG=<seconds>: dump every thread's Python stack and exit if the run is still going by then (a hung
    # capture / tuning pass on a rented GPU box costs the whole call's budget otherwise); D3F_BENCH_LOG=1: stage stamps
    # Default 30 minutes (a full default run takes ~2): a blind multi-GPU submission that stalls -- e.g. lane replays beside
    # RCCL kernels, which no one-GPU box could try -- ends with stacks on stderr instead of holding the driver's lease.
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("D3F_BENCH_WATCHDOG", "1800")), exit=True)
    t_begin = time.time()

    def stage(msg):
        if os.environ.get("D3F_BENCH_LOG"):
            print("[bench %7.1fs] %s" % (time.time() - t_begin, msg), file=sys.stderr, flush=True)

    # `python bench.py --gpus N` launched plainly (no WORLD_SIZE in the environment): spawn the N ranks ourselves, exactly
    # as the documented command does, instead of silently running one rank
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Test hook (tests/test_gpu_model.py): D3F_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and exchanges gradients over
    # gloo, so the N > 1 control flow (broadcast, split backward, bucketed exchange, max-over-ranks timing) can be
    # exercised on a one-GPU box.  RCCL refuses two ranks on one device; the timing of such a run means nothing.
    share_gpu = os.environ.get("D3F_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- one rank per GPU, launch with `python -m "
                         "torch.distributed.run --nproc-per-node %d ...` or plainly (the ranks are then spawned here)"
                         % (args.gpus, world, args.gpus))
    if args.rendezvous_only:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        n = torch.ones(1)
        if world > 1:
            dist.all_reduce(n)
        if int(n.item()) != args.gpus:
            raise SystemExit("bench.py: %d ranks joined, --gpus %d" % (int(n.item()), args.gpus))
        if rank == 0:
            print(json.dumps({"n_gpus": world, "ranks": int(n.item()), "rendezvous_only": True}))
        if world > 1:
            dist.destroy_process_group()
        return
    if not share_gpu and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) visible" % (world, torch.cuda.device_count()))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from d3feat_pytorch_amd import _native, config as cfgmod, ops, synthetic
    from d3feat_pytorch_amd.datasets import dataloader as dl
    from d3feat_pytorch_amd.train import TrainStep
    _native.lib()  # fail loudly if the HIP library is missing
    import d3feat_pytorch_amd as d3f
    tuned = False
    if args.no_tune_missing:
        d3f.TUNE_MISSING_GEMMS = False
    if not args.no_tuned_gemm and not os.environ.get("PYTORCH_TUNABLEOP_ENABLED"):
        tuned = d3f.enable_tuned_gemms()
        if not tuned:     # loud: the 4 x 3 step is safe without the table (one BLAS handle per lane), only ~4 % slower
            print("WARNING: tuned/tunableop_gfx950.csv was not accepted by this PyTorch / ROCm stack (validators differ): "
                  "library-default GEMM picks", file=sys.stderr)

    cfg = cfgmod.default_config()

    def gpu_subsample(points, lengths, dlen):
        p, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(points).to(dev), torch.as_tensor(lengths).to(dev),
                                                sampleDl=dlen)
        return p.cpu().numpy(), b.cpu().numpy()

    host_items, items = [], []
    for i in range(args.pairs):
        sa, sb = 100 * rank + 2 * i + 1, 100 * rank + 2 * i + 2   # rank r uses seeds (100r+2i+1, 100r+2i+2)
        it = synthetic.make_pair(sa, sb, gpu_subsample)
        host_items.append(it)
        items.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in it))

    class _DS:
        config = cfg

        def __len__(self):
            return 1

        def __getitem__(self, i):
            return host_items[0]
    limits = [int(x) for x in dl.calibrate_neighbors(_DS(), cfg, samples_threshold=10 ** 9)]

    stage("items + limits ready")
    ts = TrainStep(cfg, limits, dev, world_size=world, seed=0)
    prof = ops.EventProfiler()
    n_total = args.warmup + args.steps
    use_graph = not args.no_graph
    if use_graph:
        # static-capacity shapes + hipGraph replay of the whole step (forward, loss, backward, optimizer of pair k on the
        # main branch, pyramid of pair k+1 on a side branch)
        sizes = []
        for it in items:
            b = ts.build_batch(it)
            sizes.append([int(t.shape[0]) for t in b['points']])
        # every pair that will be run has been measured: the capacities need no slack beyond the 64-row rounding
        # (a real data loader calibrates them like neighborhood_limits; overflow is detected, D3F_ST_CAPACITY)
        ts.enable_graph(TrainStep.capacities_for(sizes, slack=1.0), num_corr=int(items[0][4].shape[0]))
        try:
            stage("capturing the one-pair engine, capacities %s" % (ts.caps,))
            ts.capture(items[0])
            stage("captured")
        except Exception as e:  # pragma: no cover - keep the benchmark alive on a capture problem
            print("hipGraph capture failed (%s: %s); falling back to eager launches" % (type(e).__name__, e),
                  file=sys.stderr)
            use_graph = False

    def run_one(k):
        # the pyramid of pair k+1 is built on a side stream / side branch of the graph meanwhile
        nxt = items[(k + 1) % len(items)] if k + 1 < n_total else None
        if use_graph:
            return ts.step_graph(items[k % len(items)], nxt)
        return ts.step(items[k % len(items)], next_item=nxt)

    # several pairs in flight (graph mode only): lane j of step k trains on pair P*k + j of the rank's cycle
    L = max(1, args.lanes) if use_graph else 1      # lanes: network graphs in flight
    Q = max(1, args.stack) if use_graph else 1      # pairs stacked into each of them
    P = L * Q                                       # pairs per step and GPU
    lanes = None

    def stacked_caps(slack):
        """Capacities of a stack of Q pairs: Q times the largest level sizes seen (any Q of the cycled pairs fit)."""
        if Q == 1:
            return TrainStep.capacities_for(sizes, slack=slack)
        return TrainStep.capacities_for([[Q * max(sz[l] for sz in sizes) for l in range(len(sizes[0]))]], slack=slack)
    if P > 1:
        from d3feat_pytorch_amd.train import PairLanes
        try:
            lanes = PairLanes(ts, L, stack=Q)
            lanes.enable_graph(stacked_caps(1.0), num_corr=int(items[0][4].shape[0]))
            stage("capturing %d lanes x %d stacked pairs, capacities %s" % (L, Q, lanes.caps))
            lanes.capture(tuple(items[j % len(items)] for j in range(P)))
            stage("captured")
            lanes.probe_overlap()      # (after init_process_group: RCCL's queues are in place)
            stage("lanes overlap %s" % (lanes.overlap,))
        except Exception as e:  # pragma: no cover - keep the benchmark alive: one pair in flight, as in rounds 1-2
            print("pairs in flight unavailable (%s: %s); running one pair per step" % (type(e).__name__, e),
                  file=sys.stderr)
            lanes, P, L, Q = None, 1, 1, 1
            ts.flat.bind(0)

    # EXPERIMENT knob (never a benchmark number: the JSON line says so): every step trains on the pairs the graphs were
    # captured on and builds no pyramid -- an upper bound on what shortening the pyramid builds could buy
    frozen = os.environ.get("D3F_BENCH_FROZEN_PYRAMIDS") == "1" and lanes is not None

    def run(k):
        if lanes is None:
            return run_one(k)
        if frozen:
            return lanes.step_graph([items[j % len(items)] for j in range(P)], TrainStep.NO_PREFETCH)[0]
        cur = [items[(P * k + j) % len(items)] for j in range(P)]
        nxt = [items[(P * (k + 1) + j) % len(items)] for j in range(P)]
        return lanes.step_graph(cur, nxt)[0]

    for w in range(args.warmup):
        run(w)
    torch.cuda.synchronize()
    stage("warm-up done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = run(args.warmup + k)
    t_enq = time.perf_counter()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    stage("timed region done")
    loss_val = float(out[0].item()) / Q      # (a stacked lane reports the sum over its stack)
    if use_graph:
        ts.check_status()
    # run-to-run spread: the same K steps a few more times (single shots of 90 ms differ by ~1 %)
    block_rates = []
    if world == 1:
        for _ in range(max(0, args.blocks)):
            torch.cuda.synchronize()
            tb0 = time.perf_counter()
            for k in range(args.steps):
                run(args.warmup + k)
            torch.cuda.synchronize()
            block_rates.append(P * args.steps / (time.perf_counter() - tb0))
        if use_graph:
            ts.check_status()
    # the reference's schedule -- ONE pair per optimizer step -- on the same engine, same K steps
    one_in_flight = None
    if lanes is not None:
        torch.cuda.synchronize()
        for k in range(3):
            run_one(k)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        to0 = time.perf_counter()
        for k in range(args.steps):
            run_one(3 + k)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt1 = torch.tensor([time.perf_counter() - to0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(dt1, op=dist.ReduceOp.MAX)
        one_in_flight = {"value": round(world * args.steps / float(dt1.item()), 3), "unit": "fragment-pairs/s",
                         "ms_per_step": round(float(dt1.item()) / args.steps * 1e3, 3),
                         "note": "one pair per optimizer step (dataloader.py:73 batch size 1), one network graph in "
                                 "flight per GPU: the schedule of rounds 1-2"}
        ts.check_status()
    # The reference's boundary hands HOST arrays to the step (dataset item -> collate).  Same K steps again with every
    # pair uploaded from pageable NumPy memory inside the timed region (TrainStep.upload); reported next to `value`,
    # never as `value`.
    pcie = None
    if world == 1 and not args.quick:
        def run_host(k):
            if lanes is not None:     # P pairs per step, each uploaded inside the timed region
                cur = run_host.cur if run_host.cur is not None else [
                    ts.upload(host_items[(P * k + j) % len(host_items)]) for j in range(P)]
                run_host.cur = [ts.upload(host_items[(P * (k + 1) + j) % len(host_items)]) for j in range(P)]
                return lanes.step_graph(cur, run_host.cur)[0]
            cur = run_host.cur if run_host.cur is not None else ts.upload(host_items[k % len(host_items)])
            run_host.cur = ts.upload(host_items[(k + 1) % len(host_items)])
            return ts.step_graph(cur, run_host.cur) if use_graph else ts.step(cur, next_item=run_host.cur)
        run_host.cur = None
        for k in range(2):
            run_host(k)
        torch.cuda.synchronize()
        th0 = time.perf_counter()
        for k in range(args.steps):
            run_host(2 + k)
        torch.cuda.synchronize()
        th1 = time.perf_counter()
        pcie = {"value": round(P * args.steps / (th1 - th0), 3), "unit": "fragment-pairs/s",
                "ms_per_step": round((th1 - th0) / args.steps * 1e3, 3),
                "note": "same step, every pair uploaded from pageable host arrays (points, correspondences, keypoint "
                        "distances: ~0.6 MB) inside the timed region"}
    # What a USER of trainer.Trainer gets: capacities sampled with 10 % head-room instead of sized to the
    # exact pairs, and a stream of pairs of different sizes through capacity classes (one graph engine per class,
    # the next pair's pyramid prefetched into its own class's sets).  Reported next to `value`, never as `value`.
    trainer_path = None
    if world == 1 and use_graph and not args.quick:
        try:
            keep = (ts.flat.data.clone(), ts.opt.buf.clone(), ts.opt.state.clone())
            eng110 = ts.clone_for_capacities(TrainStep.capacities_for(sizes, slack=1.10), num_corr=int(items[0][4].shape[0]))
            eng110.capture(items[0])
            for k in range(3):
                eng110.step_graph(items[k % len(items)], items[(k + 1) % len(items)])
            torch.cuda.synchronize()
            tt0 = time.perf_counter()
            for k in range(args.steps):
                eng110.step_graph(items[(3 + k) % len(items)], items[(4 + k) % len(items)])
            torch.cuda.synchronize()
            tt1 = time.perf_counter()
            trainer_path = {"capacity_slack_1.10": {"value": round(args.steps / (tt1 - tt0), 3), "unit": "fragment-pairs/s",
                                                     "capacities": eng110.caps}}
            if lanes is not None:      # the same head-room with pairs in flight (Trainer(pairs_in_flight=P))
                lanes110 = lanes.clone_for_capacities(stacked_caps(1.10), num_corr=int(items[0][4].shape[0]))
                lanes110.capture(tuple(items[j % len(items)] for j in range(P)))

                def run110(k):
                    return lanes110.step_graph([items[(P * k + j) % len(items)] for j in range(P)],
                                               [items[(P * (k + 1) + j) % len(items)] for j in range(P)])
                for k in range(3):
                    run110(k)
                torch.cuda.synchronize()
                tt0 = time.perf_counter()
                for k in range(args.steps):
                    run110(3 + k)
                torch.cuda.synchronize()
                tt1 = time.perf_counter()
                trainer_path["capacity_slack_1.10_pairs_in_flight"] = {
                    "value": round(P * args.steps / (tt1 - tt0), 3), "unit": "fragment-pairs/s", "pairs_in_flight": P,
                    "lanes": L, "stacked_pairs_per_lane": Q}
                del lanes110
            del eng110
            # mixed sizes: the S1-class pairs alternating with pairs a quarter of their size, two capacity classes
            small_items = []
            for i in range(2):
                it = synthetic.make_pair(900 + 2 * i, 901 + 2 * i, gpu_subsample, n_raw=80000, scale=0.31)
                small_items.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in it))
            ssz = [[int(t.shape[0]) for t in ts.build_batch(it)['points']] for it in small_items]
            eng_s = ts.clone_for_capacities(TrainStep.capacities_for(ssz, slack=1.10), num_corr=int(items[0][4].shape[0]))
            eng_s.capture(small_items[0])
            stream_items = [items[0], small_items[0], items[1], small_items[1]]

            def cls(it):
                return eng_s if eng_s.fits(it) else ts

            def mixed(k):
                cur, nxt = stream_items[k % 4], stream_items[(k + 1) % 4]
                e, ne = cls(cur), cls(nxt)
                if ne is e:
                    return e.step_graph(cur, nxt)
                ne.preload(nxt)
                return e.step_graph(cur, TrainStep.NO_PREFETCH)
            for k in range(4):
                mixed(k)
            torch.cuda.synchronize()
            tm0 = time.perf_counter()
            for k in range(args.steps):
                mixed(4 + k)
            torch.cuda.synchronize()
            tm1 = time.perf_counter()
            pts_mix = [int(it[0].shape[0] + it[1].shape[0]) for it in stream_items]
            trainer_path["mixed_sizes_two_classes"] = {
                "value": round(args.steps / (tm1 - tm0), 3), "unit": "fragment-pairs/s", "points_per_pair": pts_mix,
                "capacities": [eng_s.caps, ts.caps],
                "skipped_or_rerun": len(eng_s.take_overflowed(drain=True)) + len(ts.take_overflowed(drain=True))}
            if lanes is not None:
                # the same two classes with pairs in flight (Trainer(pairs_in_flight=P)): groups of P large pairs
                # alternating with groups of P small ones -- every step changes class, so every group's pyramids are
                # preloaded into the other class's sets instead of prefetched by the running step
                lanes_s = lanes.clone_for_capacities(
                    TrainStep.capacities_for([[Q * max(sz[l] for sz in ssz) for l in range(len(ssz[0]))]], slack=1.10),
                    num_corr=int(items[0][4].shape[0]))
                lanes_s.capture(small_items[0])
                groups = [[items[j % len(items)] for j in range(P)], [small_items[j % 2] for j in range(P)],
                          [items[(P + j) % len(items)] for j in range(P)], [small_items[(j + 1) % 2] for j in range(P)]]

                def mixed_lanes(k):
                    e, ne = (lanes, lanes_s)[k % 2], (lanes, lanes_s)[(k + 1) % 2]
                    ne.preload(groups[(k + 1) % 4])
                    return e.step_graph(groups[k % 4], TrainStep.NO_PREFETCH)
                for k in range(4):
                    mixed_lanes(k)
                torch.cuda.synchronize()
                tl0 = time.perf_counter()
                for k in range(args.steps):
                    mixed_lanes(4 + k)
                torch.cuda.synchronize()
                tl1 = time.perf_counter()
                trainer_path["mixed_sizes_two_classes_pairs_in_flight"] = {
                    "value": round(P * args.steps / (tl1 - tl0), 3), "unit": "fragment-pairs/s", "pairs_in_flight": P,
                    "groups": "P large pairs / P small pairs alternating (a class change every step)",
                    "skipped_or_rerun": len(lanes.take_overflowed(drain=True)) + len(lanes_s.take_overflowed(drain=True))}
                del lanes_s
            del eng_s
            for dst, src in zip((ts.flat.data, ts.opt.buf, ts.opt.state), keep):
                dst.copy_(src)
        except Exception as e:  # pragma: no cover - the headline number must not depend on this leg
            trainer_path = {"error": "%s: %s" % (type(e).__name__, e)}
    # N > 1: how much of the gradient exchange hides under the backward of the fine levels.  Three legs of the same K
    # steps: (a) the timed region above; (b) the steps with the all-reduces left out (ranks drift apart: run LAST, after
    # the replica check below would be too late, so parameters are saved and restored); (c) the all-reduces alone.
    exchange = None
    if world > 1:
        try:
            keep = (ts.flat.data.clone(), ts.opt.buf.clone())
            ts_world = ts.world
            if lanes is not None:
                lanes.exchange = False                # the same two-stage step without its all-reduces
            else:
                ts.world = 1                          # no exchange, same split graphs
            for k in range(2):
                run(args.warmup + k)
            torch.cuda.synchronize()
            dist.barrier()
            tn0 = time.perf_counter()
            for k in range(args.steps):
                run(args.warmup + k)
            torch.cuda.synchronize()
            t_noex = (time.perf_counter() - tn0) / args.steps
            ts.world = ts_world
            if lanes is not None:
                lanes.exchange = True
            ts.flat.data.copy_(keep[0])
            ts.opt.buf.copy_(keep[1])
            g = ts.flat.grad
            deep = g[ts.numel_shallow:]
            step = (deep.numel() + 2) // 3

            def exchange_only():     # (lanes: the same buckets, of the lanes' summed gradient)
                works = [dist.all_reduce(deep[b * step:min(deep.numel(), (b + 1) * step)], op=dist.ReduceOp.SUM,
                                         async_op=True) for b in range(3)]
                works.append(dist.all_reduce(g[:ts.numel_shallow], op=dist.ReduceOp.SUM, async_op=True))
                for w in works:
                    w.wait()
            for _ in range(2):
                exchange_only()
            torch.cuda.synchronize()
            dist.barrier()
            tc0 = time.perf_counter()
            for _ in range(args.steps):
                exchange_only()
            torch.cuda.synchronize()
            t_comm = (time.perf_counter() - tc0) / args.steps
            # ... and the same three legs for the REFERENCE's schedule (one pair per optimizer step and rank: the hard case,
            # 97 MB of gradients per ~4 ms step): the one-pair engine's split graphs without their all-reduces
            if lanes is not None and one_in_flight is not None:
                try:
                    keep1 = (ts.flat.data.clone(), ts.opt.buf.clone())
                    ts.world = 1
                    for k in range(2):
                        run_one(k)
                    torch.cuda.synchronize()
                    dist.barrier()
                    tq0 = time.perf_counter()
                    for k in range(args.steps):
                        run_one(2 + k)
                    torch.cuda.synchronize()
                    t_noex1 = (time.perf_counter() - tq0) / args.steps
                    ts.world = ts_world
                    ts.flat.data.copy_(keep1[0])
                    ts.opt.buf.copy_(keep1[1])
                    t_step1 = one_in_flight["ms_per_step"] * 1e-3
                    exposed1 = max(0.0, t_step1 - t_noex1)
                    one_in_flight["exchange"] = {
                        "step_ms": round(t_step1 * 1e3, 3), "step_without_exchange_ms": round(t_noex1 * 1e3, 3),
                        "exchange_alone_ms": round(t_comm * 1e3, 3), "exposed_ms": round(exposed1 * 1e3, 3),
                        "overlap_frac": round(1.0 - exposed1 / t_comm, 3) if t_comm > 0 else None,
                        "buckets": "3 deep chunks (overlapped with the stage-2 backward graph) + 1 shallow"}
                except Exception as e:  # pragma: no cover
                    ts.world = ts_world
                    one_in_flight["exchange"] = {"error": "%s: %s" % (type(e).__name__, e)}
            t_step = (t1 - t0) / args.steps
            exposed = max(0.0, t_step - t_noex)
            exchange = {"rccl_ranks": world, "backend": dist.get_backend(), "bytes_per_step": int(g.numel() * 4),
                        "buckets": "3 deep chunks of the lanes' summed gradient (exchanged under every lane's stage-2 "
                                   "backward graph) + 1 shallow" if lanes is not None
                        else "3 deep chunks (overlapped with the stage-2 backward graph) + 1 shallow",
                        "step_ms": round(t_step * 1e3, 3), "step_without_exchange_ms": round(t_noex * 1e3, 3),
                        "exchange_alone_ms": round(t_comm * 1e3, 3), "exposed_ms": round(exposed * 1e3, 3),
                        "overlap_frac": round(1.0 - exposed / t_comm, 3) if t_comm > 0 else None}
        except Exception as e:  # pragma: no cover - the headline number must not depend on this leg
            exchange = {"rccl_ranks": world, "error": "%s: %s" % (type(e).__name__, e)}
    # data-parallel sanity: after the timed steps every rank must hold bit-identical parameters
    replica_spread = None
    if world > 1:
        chk = ts.flat.data.double().abs().sum().reshape(1)
        hi, lo = chk.clone(), chk.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        replica_spread = float((hi - lo).item())
    # Per-operator HIP-event timing (events on the launch stream around every C-ABI call).  Events cannot be recorded
    # inside a replayed graph, so the same steps are run eagerly right after the timed region, same process and data.
    if not args.quick:
        ops.set_profiler(prof)
        ts._pending = None
        for k in range(3):
            ts.step(items[k % len(items)])
        torch.cuda.synchronize()
        ops.set_profiler(None)
    prof_steps = 3
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    # Roofline leg: the hand-written KPConv kernels are the largest kernels of the training stream (profiles/r02*);
    # every launch of them in 3 eager steps is bracketed by HIP events inside the library, on the stream it is launched
    # on, and the one with the most time per step is reported as `roofline`.
    # ... at the shapes the timed region ran: with stacked pairs, a lane's step on its stack (static shapes, eager launches)
    roof_eng = lanes.engines[0] if (lanes is not None and Q > 1) else None
    roof_items = tuple(items[j % len(items)] for j in range(Q))

    def _three_steps():
        ts._pending = None
        for k in range(3):
            if roof_eng is not None:
                roof_eng._static_step(roof_items)
            else:
                ts.step(items[k % len(items)])
    edges, widths = {}, {}
    if not args.quick:   # TRUE edges of the transposed tables = valid entries of the forward tables
        if roof_eng is not None:
            roof_eng._static_step(roof_items)
            torch.cuda.synchronize()
            batches = [roof_eng.sets[0].batch]
        else:
            batches = [ts.build_batch(items[k]) for k in range(min(3, len(items)))]
        for b_k in batches:
            for tabs in (b_k['neighbors'], b_k['pools']):
                for t in tabs:
                    r = getattr(t, '_d3f_rev', None)
                    if r is not None:
                        edges[(r.Nq, r.Ns)] = int((t < r.Ns).sum())
                        widths[(r.Nq, r.Ns)] = int(r.width)
    kt = {} if args.quick else timed_kernels(
        _native.lib(), _three_steps, {1: fwd_kernel_cost, 2: dx_kernel_cost, 3: gather_kernel_cost(edges, widths),
                                      4: atb_kernel_cost, 5: agg_fwd_kernel_cost, 6: agg_rev_kernel_cost(edges),
                                      7: atb_group_cost}, 3)
    dom = max(kt, key=lambda w: kt[w]["us_per_step"]) if kt else None
    dx_t = kt.get(dom)

    # dense mutual-NN matching of the pair's descriptors (19k x 19k x 32, the distance matrix is
    # never materialised) -- the one MFMA-bound kernel of the path; timed with HIP events on the current stream.
    matching = None
    if rank == 0 and not args.quick:
        n0m, n1m = int(items[0][0].shape[0]), int(items[0][1].shape[0])
        gen = torch.Generator(device=dev).manual_seed(0)
        da = torch.nn.functional.normalize(torch.randn(n0m, 32, device=dev, generator=gen), dim=1)
        db = torch.nn.functional.normalize(torch.randn(n1m, 32, device=dev, generator=gen), dim=1)
        for _ in range(2):
            ops.mutual_nn(da, db)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            ops.mutual_nn(da, db)
        e1.record()
        torch.cuda.synchronize()
        mms = e0.elapsed_time(e1) / reps
        mfl = 2.0 * n0m * n1m * 32        # S T^T once -- row and column arg-min come out of the same sweep
        matching = {"workload": "mutual-NN of %d x %d unit descriptors (32-d): row argmin + column argmin + mutual flag, "
                                "one sweep over the S x T tiles" % (n0m, n1m), "ms": round(mms, 3), "bound": "mfma",
                    "achieved": round(mfl / (mms * 1e-3) / 1e12, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(mfl / (mms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                    "flops": int(mfl), "pairs_per_s": round(1e3 / mms, 1)}

    # evaluation pass of one pair -- pyramid + eval-mode forward (descriptors, gated scores), top-k
    # keypoints by score, mutual-NN matching of the selected descriptors (build_correspondence); eager launches.
    evaluation = None
    if rank == 0 and not args.quick:
        from d3feat_pytorch_amd.geometric_registration.common import build_correspondence, select_keypoints
        ts.model.eval()
        ev_item = items[0]
        n0e = int(ev_item[0].shape[0])

        def eval_pass(k):
            with torch.no_grad():
                b = ts.build_batch(ev_item)
                feats, scores = ts.model(b)
                si = select_keypoints(scores[:n0e], k)
                ti = select_keypoints(scores[n0e:], k)
                return build_correspondence(feats[:n0e][si], feats[n0e:][ti])
        evaluation = {}
        for k in (250, 5000):
            for _ in range(2):
                corr_k = eval_pass(k)
            torch.cuda.synchronize()
            t_e0 = time.perf_counter()
            for _ in range(5):
                corr_k = eval_pass(k)
            torch.cuda.synchronize()
            evaluation["top%d" % k] = {"ms_per_pair": round((time.perf_counter() - t_e0) / 5 * 1e3, 3),
                                       "mutual_matches": int(corr_k.shape[0])}
        evaluation["note"] = ("pyramid + eval forward + top-k + mutual-NN per pair, eager launches (host-bound), random-init "
                              "weights")
        # the same pass on the inference engine: forward-only network graph, the next pair's pyramid graph on the side
        # stream (infer.InferStep); descriptors + scores of both fragments, then top-250 + mutual-NN
        try:
            from d3feat_pytorch_amd.infer import InferStep
            eng = InferStep(ts.model, cfg, limits, dev, clouds=2)
            eng.enable_graph(ts.caps if use_graph else TrainStep.capacities_for(
                [[int(t.shape[0]) for t in ts.build_batch(it)['points']] for it in items], slack=1.0))
            clouds = [(it[0], it[1]) for it in items]

            def eval_graph(k):
                cur, nxt = clouds[k % len(clouds)], clouds[(k + 1) % len(clouds)]
                feats, scores = eng.describe(cur, nxt)
                n0g = int(cur[0].shape[0])
                si = select_keypoints(scores[:n0g], 250)
                ti = select_keypoints(scores[n0g:], 250)
                return build_correspondence(feats[:n0g][si], feats[n0g:][ti])
            for k in range(3):
                corr_g = eval_graph(k)
            torch.cuda.synchronize()
            t_g0 = time.perf_counter()
            for k in range(10):
                corr_g = eval_graph(3 + k)
            torch.cuda.synchronize()
            eng.check_status()
            evaluation["top250_pipelined"] = {"ms_per_pair": round((time.perf_counter() - t_g0) / 10 * 1e3, 3),
                                              "mutual_matches": int(corr_g.shape[0]),
                                              "launch": "hipGraph replay, pyramid of the next pair on a side stream"}
            # the same with keypoints and matches left on the device (d3f_topk_scores + d3f_mutual_nn_batched, P = 1):
            # no torch.nonzero, hence no host synchronisation per pair -- the pipeline runs ahead
            def eval_device(k):
                cur, nxt = clouds[k % len(clouds)], clouds[(k + 1) % len(clouds)]
                feats, scores = eng.describe(cur, nxt)
                return eng.match(cur, feats, scores, num_points=250)
            for k in range(3):
                m_dev = eval_device(k)
            torch.cuda.synchronize()
            t_d0 = time.perf_counter()
            for k in range(20):
                m_dev = eval_device(3 + k)
            torch.cuda.synchronize()
            evaluation["top250_pipelined_device_outputs"] = {
                "ms_per_pair": round((time.perf_counter() - t_d0) / 20 * 1e3, 3), "mutual_matches": int(m_dev[1].sum()),
                "note": "keypoint table [2,250] and mutual flags stay on the device (one read-back per scene instead of "
                        "one per pair)"}
        except Exception as e:  # pragma: no cover - the headline number must not depend on this leg
            evaluation["top250_pipelined"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # BASELINE configs[3]: 8 fragment pairs per inference batch -- 16 clouds stacked into one forward graph (per-pair
        # detector normaliser / table widths), top-250 keypoints of all 16 clouds by one launch, the 8 matchings by one
        # pair of launches; and the dense all-points matching of the 8 pairs (8 x 19k x 19k x 32 on the f32 matrix cores)
        try:
            from d3feat_pytorch_amd.infer import InferStep
            eng8 = InferStep(ts.model, cfg, limits, dev, clouds=16, group=2)
            one = ts.caps if use_graph else TrainStep.capacities_for(
                [[int(t.shape[0]) for t in ts.build_batch(it)['points']] for it in items], slack=1.0)
            eng8.enable_graph([8 * int(c) for c in one])
            stacks = [tuple(c for j in range(8) for c in items[(j + s0) % len(items)][:2]) for s0 in range(2)]

            def batch8(k):
                cur, nxt = stacks[k % 2], stacks[(k + 1) % 2]
                feats, scores = eng8.describe(cur, nxt)
                return cur, feats, scores, eng8.match(cur, feats, scores, num_points=250)
            for k in range(3):
                cur, feats8, scores8, m8 = batch8(k)
            torch.cuda.synchronize()
            t80 = time.perf_counter()
            for k in range(6):
                cur, feats8, scores8, m8 = batch8(3 + k)
            torch.cuda.synchronize()
            ms8 = (time.perf_counter() - t80) / 6 * 1e3
            eng8.check_status()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(2):
                dense = eng8.match(cur, feats8, scores8)
            e0.record()
            for _ in range(3):
                dense = eng8.match(cur, feats8, scores8)
            e1.record()
            torch.cuda.synchronize()
            dms = e0.elapsed_time(e1) / 3
            dfl = sum(2.0 * int(cur[2 * j].shape[0]) * int(cur[2 * j + 1].shape[0]) * 32 for j in range(8))
            evaluation["batched_8_pairs"] = {
                "workload": "16 clouds (%d points) in one forward graph + top-250 keypoints + 8 mutual-NN matchings"
                            % sum(int(c.shape[0]) for c in cur),
                "ms_per_batch": round(ms8, 3), "pairs_per_s": round(8e3 / ms8, 1),
                "mutual_matches_top250": int(m8[1].sum()),
                "dense_matching": {"workload": "8 x (19k x 19k x 32) row + column arg-min, one sweep (one launch) for all",
                                   "ms": round(dms, 3), "bound": "mfma",
                                   "achieved": round(dfl / (dms * 1e-3) / 1e12, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": round(dfl / (dms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                                   "pairs_per_s": round(8e3 / dms, 1), "mutual_matches": int(dense[1].sum())}}
            del eng8
        except Exception as e:  # pragma: no cover
            evaluation["batched_8_pairs"] = {"error": "%s: %s" % (type(e).__name__, e)}
        ts.model.train()

    if rank == 0:
        n_pts = [int(it[0].shape[0] + it[1].shape[0]) for it in items]
        summary = prof.summary()
        roofline = None
        if dx_t is not None:
            traffic, traffic_stale = None, None
            tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    entry = json.load(f).get(KERNEL_NAMES[dom].split(" ")[0], {})
                traffic = entry.get("hbm_bytes_per_launch_pair", entry.get("hbm_bytes_per_launch"))
                # the counters were collected on the kernel source whose SHA is stored with them (profiles/pmc_traffic.py)
                src = entry.get("source")
                if traffic is not None and src:
                    import hashlib
                    with open(os.path.join(REPO, "d3feat.pytorch_amd", "csrc", src), "rb") as fh:
                        traffic_stale = hashlib.sha256(fh.read()).hexdigest()[:16] != entry.get("source_sha16")
            counters = None   # L2 hit rate / MFMA-pipe busy of the same kernels (separate rocprofv3 --pmc passes)
            cpath = os.path.join(REPO, "profiles", "r06_pmc_kernels.json")
            for older in ("r05_pmc_kernels.json", "r04_pmc_kernels.json", "r03_pmc_kpconv.json"):
                if not os.path.exists(cpath):
                    cpath = os.path.join(REPO, "profiles", older)
            if os.path.exists(cpath):
                with open(cpath) as f:
                    counters = json.load(f).get(KERNEL_NAMES[dom].split(" ")[0])
            f_hbm = dx_t["gbs"] / HBM_PEAK_GBS
            f_mfma = dx_t["tflops"] / F32_MFMA_PEAK_TFLOPS
            mfma_bound = f_mfma > f_hbm
            roofline = {
                "bound": "mfma" if mfma_bound else "hbm",
                "kernel": KERNEL_NAMES[dom] + "; all channel-width instantiations, every layer it runs on",
                "achieved": round(dx_t["tflops"] if mfma_bound else dx_t["gbs"], 2),
                "peak": F32_MFMA_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS,
                "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": round(max(f_mfma, f_hbm), 4),
                "traffic": traffic, "traffic_source_stale": traffic_stale,
                "avg_us": round(dx_t["avg_us"], 2), "launches_timed": dx_t["launches"],
                "us_per_step": round(dx_t["us_per_step"], 1),
                "algorithmic_bytes_per_launch": int(dx_t["bytes_per_launch"]),
                "algorithmic_flops_per_launch": int(dx_t["flops_per_launch"]),
                "byte_model": "SURVEY 8d: gather-expanded logical bytes, int32 indices, a gathered row charged once per "
                              "TRUE (query, support) edge (counted on the device on the forward tables)",
                "hbm": {"achieved_GBs": round(dx_t["gbs"], 1), "frac": round(f_hbm, 4)},
                "unique_bytes": {"per_launch": int(dx_t["unique_bytes_per_launch"]),
                                 "achieved_GBs": round(dx_t["unique_gbs"], 1),
                                 "frac": round(dx_t["unique_gbs"] / HBM_PEAK_GBS, 4),
                                 "note": "8d's lower-bound figure: every operand tensor once"},
                "counters": counters,
                "per_launch": dx_t["per_launch"],
                "mfma_f32": {"achieved_TFLOPs": round(dx_t["tflops"], 2), "frac": round(f_mfma, 4)},
                "also_timed": [{"kernel": KERNEL_NAMES[w].split(" ")[0], "avg_us": round(v["avg_us"], 2),
                                "launches_timed": v["launches"], "us_per_step": round(v["us_per_step"], 1),
                                "achieved_GBs": round(v["gbs"], 1), "hbm_frac": round(v["gbs"] / HBM_PEAK_GBS, 4),
                                "unique_GBs": round(v["unique_gbs"], 1),
                                "achieved_TFLOPs": round(v["tflops"], 2),
                                "mfma_frac": round(v["tflops"] / F32_MFMA_PEAK_TFLOPS, 4)}
                               for w, v in sorted(kt.items()) if w != dom],
                "pairs_per_timed_step": Q,
                "measured": "hipEventRecord on the launch stream immediately before/after each launch of the kernel "
                            "(d3f_debug_kernel_timing_*), 3 eager steps after the timed region on the shapes it ran (a "
                            "lane's stack of %d pair(s)); averages are time-weighted over all launches; the kernel "
                            "reported is the hand-written kernel with the most time per training step" % Q}
        res = {
            "metric": "fragment-pairs/sec (fwd+bwd) on 3DMatch-shaped pairs" + (
                " -- EXPERIMENT, NOT A BENCHMARK: pyramids frozen (D3F_BENCH_FROZEN_PYRAMIDS)" if frozen else ""),
            "value": round(P * args.steps * world / elapsed, 3),
            "unit": "fragment-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "pairs_per_step": P * world,
            # whole-job value / ranks: what one GPU of this run sustains (cross-check against the N = 1 line)
            "value_per_gpu": round(P * args.steps / elapsed, 3),
            "one_pair_in_flight": one_in_flight,
            "value_blocks": None if not block_rates else {
                "blocks": len(block_rates), "median": round(float(np.median(block_rates)), 3),
                "min": round(min(block_rates), 3), "max": round(max(block_rates), 3),
                "note": "the same K steps repeated after the contract region (value is the contract region alone)"},
            "host_enqueue_ms_per_step": round((t_enq - t0) / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: full D3Feat KPFCNN fwd+bwd on fragment pairs "
                                   "(%d stacked points avg, 128 correspondences, 32-d descriptors, circle+detector loss, "
                                   "on-device radius search + grid subsample, SGD step); %s" % (int(np.mean(n_pts)), (
                                       "%d pairs per step and GPU = %d network graph(s) in flight x %d pairs stacked into "
                                       "each (one pyramid, one forward + loss + backward for the stack), every pair a whole "
                                       "forward + loss + backward, ONE guarded SGD step per %d pairs on the mean of their "
                                       "gradients (the update a %d-rank data-parallel step makes); one_pair_in_flight = one "
                                       "pair per optimizer step" % (P, L, Q, P * world, P * world)) if P > 1 else
                                       "one pair per optimizer step"),
                       "points_per_pair": n_pts, "neighbor_limits": limits, "pairs_per_rank": len(items),
                       "pairs_in_flight_per_gpu": P, "lanes": L, "stacked_pairs_per_lane": Q,
                       "lanes_overlap_factor": None if lanes is None else getattr(lanes, "overlap", {}).get("factor"),
                       "lanes_overlap_probe": None if lanes is None else getattr(lanes, "overlap", None),
                       "peak_hbm_allocated_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                       "side_stream_probe_ms": getattr(ts, "_side_probe", None),
                       "parallelism": "dp%d" % world if P == 1 else ("dp%d x %d lanes" % (world, L)) + (
                           " x %d stacked" % Q if Q > 1 else ""),
                       "final_loss": round(loss_val, 5),
                       "replica_param_checksum_spread": replica_spread,
                       "skipped_steps": int(ts.opt.skipped),
                       "library_gemms": "TunableOp table tuned/tunableop_gfx950.csv" if tuned else "library default",
                       "launch": "hipGraph replay: network step on the %s, next pair's pyramid graph %s "
                                 "(static level capacities %s)" % (
                                     "training stream" if P == 1 else "lane's stream",
                                     "on a side stream" if L <= 2 else "on a pyramid stream the lanes share" if L == 3
                                     else "on the lane's own stream behind it (all four compute pipes train)",
                                     lanes.caps if lanes is not None else ts.caps)
                                 if use_graph else "eager launches, pyramid on a side stream"},
            "pcie_inclusive": pcie,
            "trainer_path": trainer_path,
            "exchange": exchange,
            "roofline": roofline,
            "matching": matching,
            "evaluation": evaluation,
            "kernels": {k: {"avg_us": round(v["avg_ms"] * 1e3, 2), "calls": v["calls"],
                            "total_ms": round(v["total_ms"], 3)} for k, v in
                        sorted(summary.items(), key=lambda kv: -kv[1]["total_ms"])[:12]},
        }
        # the step as a whole against the matrix peak: SURVEY 8d's algorithmic FLOPs of one S1-class pair (KPConv 16.56
        # + Linear 19.78 GFLOP forward; x3 for forward + the two backward products) times the measured pairs/s per GPU
        if roofline is not None:
            fl_pair = 3.0 * (16.56e9 + 19.78e9)
            per_gpu = res["value"] / world
            roofline["whole_step"] = {
                "flops_per_pair": int(fl_pair), "achieved_TFLOPs": round(per_gpu * fl_pair / 1e12, 2),
                "mfma_f32_frac": round(per_gpu * fl_pair / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                "one_pair_in_flight_frac": None if one_in_flight is None else round(
                    one_in_flight["value"] / world * fl_pair / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                "note": "SURVEY 8d algorithmic FLOPs (forward KPConv + Linear, x3) x pairs/s per GPU; `frac` above is the "
                        "dominant hand-written kernel running alone"}
        if world == 1 and not args.no_cpu_baseline and not args.quick:
            res["cpu_baseline"] = cpu_baseline(host_items[0], cfg, limits, budget_s=args.cpu_budget)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
