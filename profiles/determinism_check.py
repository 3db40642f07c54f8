"""How far do two runs of the same training sequence drift apart?  eager vs eager (atomics order only), graph vs
eager, graph vs graph.  Used to set the tolerance of tests/test_gpu_model.py::test_graph_mode_matches_eager_step.
Run on the GPU box: python profiles/determinism_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from d3feat_pytorch_amd import config as cfgmod
from d3feat_pytorch_amd.train import TrainStep

DEV = "cuda:0"


def main():
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "s0_small.npz"))
    n0, n1 = g['pts0'].shape[0], g['pts1'].shape[0]
    raw = (g['pts0'], g['pts1'], np.ones((n0, 1), np.float32), np.ones((n1, 1), np.float32), g['sel_corr'], g['dist_keypts_in'])
    item = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(DEV) for a in raw)
    swapped = (item[1], item[0], item[3], item[2], item[4].flip(1).contiguous(), item[5].t().contiguous())
    cfg = cfgmod.default_config(first_features_dim=16, num_node=64)
    limits = [int(x) for x in g['limits']]
    sizes = [[int(g['batch.points.%d' % l].shape[0]) for l in range(5)]]
    seq = [item, swapped, item, swapped, swapped, item, item, swapped]

    def fresh():
        np.random.seed(0)
        torch.manual_seed(0)
        return TrainStep(cfg, limits, torch.device(DEV), seed=0)

    def run_eager():
        t = fresh()
        losses = [float(t.step(item)[0]) for _ in range(3)]
        losses += [float(t.step(it)[0]) for it in seq]
        return t.flat.data.clone(), losses

    def run_graph():
        t = fresh()
        t.enable_graph(TrainStep.capacities_for(sizes, slack=1.3), num_corr=item[4].shape[0])
        t.capture(item)
        losses = [0, 0, 0]
        for k, it in enumerate(seq):
            losses.append(float(t.step_graph(it, seq[k + 1] if k + 1 < len(seq) else None)[0]))
        t.check_status()
        return t.flat.data.clone(), losses

    e1, le1 = run_eager()
    e2, le2 = run_eager()
    g1, lg1 = run_graph()
    g2, lg2 = run_graph()

    def run_graph_ondemand():
        """No prefetch hint on the 4th step, then a pair object the pipeline has not seen."""
        t = fresh()
        t.enable_graph(TrainStep.capacities_for(sizes, slack=1.3), num_corr=item[4].shape[0])
        t.capture(item)
        losses = [0, 0, 0]
        for k, it in enumerate(seq[:4]):
            losses.append(float(t.step_graph(it, seq[k + 1] if k + 1 < 4 else None)[0]))
        for it in seq[4:]:
            losses.append(float(t.step_graph(tuple(x.clone() for x in it))[0]))
        return t.flat.data.clone(), losses
    g3, lg3 = run_graph_ondemand()
    print("graph on-demand/eager: %.3e" % (float((g3 - e1).abs().max()) / float(e1.abs().max())))
    print("graph on-demand losses:", ["%.5f" % v for v in lg3[3:]])
    scale = float(e1.abs().max())
    for name, a, b in (("eager/eager", e1, e2), ("graph/eager", g1, e1), ("graph/graph", g1, g2)):
        print("%s: max |dparam| / max|param| = %.3e" % (name, float((a - b).abs().max()) / scale))
    print("eager losses :", ["%.5f" % v for v in le1[3:]])
    print("eager losses2:", ["%.5f" % v for v in le2[3:]])
    print("graph losses :", ["%.5f" % v for v in lg1[3:]])
    print("graph losses2:", ["%.5f" % v for v in lg2[3:]])


if __name__ == "__main__":
    main()
