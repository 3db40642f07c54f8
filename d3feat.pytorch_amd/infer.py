"""Descriptor extraction as a pipelined hipGraph replay -- the inference side of the hot path (reference test.py:107-120:
one eager ``model(inputs)`` per fragment behind a CPU collate).

``InferStep`` reuses the training step's machinery (``TrainStep``: three static buffer sets, the pyramid graph of cloud
k+1 on a side stream under the network graph of cloud k, host-side event waits) with a forward-only network graph in
eval mode: un-normalised descriptors -> F.normalize, detector scores with the local-maximum gate.  An input is a tuple
of 1 (a fragment, ``geometric_registration.evaluate``) or more clouds stacked into one batch (a pair); outputs are
capacity-shaped static tensors, valid until the same buffer set comes round again (two calls later).
"""
import torch
import torch.nn.functional as F

from .train import TrainStep


class InferStep(TrainStep):
    reverse_tables = False   # forward only

    def __init__(self, model, config, neighborhood_limits, device, clouds=1):
        self.clouds = int(clouds)
        super().__init__(config, neighborhood_limits, device, world_size=1, model=model)
        self.model.eval()

    def _init_training_state(self, config, world_size):   # no flat parameter buffer, optimizer or loss
        self.split_backward = False
        self.last_distances = None
        self.opt = None

    def _build_set(self, st, adopt=False):
        # forward only: nothing consumes the set's flags between builds, so they stay sticky until check_status()
        keep = st.status.word.clone() if st.status is not None else None
        super()._build_set(st, adopt)
        if keep is not None:
            st.status.word.bitwise_or_(keep)

    # ---- static inputs: `clouds` point arrays per item ------------------------------------------------------------
    def enable_graph(self, capacities, num_corr=1):
        super().enable_graph(capacities, 1)
        self.g_net = None
        for st in self.sets:
            st.lens = torch.zeros(self.clouds, dtype=torch.int32, device=self.device)

    def fits(self, item):
        return len(item) == self.clouds and sum(int(p.shape[0]) for p in item) <= self.caps[0]

    def _load_inputs(self, st, item):
        if not self.fits(item):
            raise RuntimeError("input does not fit the captured shapes (%d clouds, %d points; %d clouds, capacity %d)" % (
                len(item), sum(int(p.shape[0]) for p in item), self.clouds, self.caps[0]))
        off = 0
        for c, p in enumerate(item):
            n = int(p.shape[0])
            st.pts[off:off + n].copy_(torch.as_tensor(p), non_blocking=True)
            st.lens[c] = n
            off += n

    def _net_step(self, st):
        batch = self._set_batch(st)
        with torch.no_grad():
            x, scores = self.model.forward_raw(batch)
            return F.normalize(x, p=2, dim=-1), scores

    # ---- convenience ------------------------------------------------------------------------------------------------
    def describe(self, item, next_item=None):
        """(descriptors [n,32], scores [n,1]) of the live rows of ``item`` (views of static outputs)."""
        if getattr(self, 'sets', None) is None:
            raise RuntimeError("call enable_graph(capacities) first (per-level row capacities, TrainStep.capacities_for)")
        if self.g_net is None:
            self.capture(item)
        feats, scores = self.step_graph(item, next_item)
        n = sum(int(p.shape[0]) for p in item)
        return feats[:n], scores[:n]
