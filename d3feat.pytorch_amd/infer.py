"""Descriptor extraction as a pipelined hipGraph replay -- the inference side of the hot path (reference test.py:107-120:
one eager ``model(inputs)`` per fragment behind a CPU collate).

``InferStep`` reuses the training step's machinery (``TrainStep``: three static buffer sets, the pyramid graph of cloud
k+1 on a side stream under the network graph of cloud k, host-side event waits) with a forward-only network graph in
eval mode: un-normalised descriptors -> F.normalize, detector scores with the local-maximum gate.  An input is a tuple
of 1 (a fragment, ``geometric_registration.evaluate``) or more clouds stacked into one batch (a pair); outputs are
capacity-shaped static tensors, valid until the same buffer set comes round again (two calls later).

BASELINE configs[3] -- 8 fragment pairs per inference batch: ``InferStep(..., clouds=16, group=2)`` stacks the 16 clouds
into ONE batch (every search and voxel level is per cloud anyway) and keeps what a reference batch of one pair shares
per group of 2 clouds: the detector's feature normaliser (architectures.py:342) and the widths of the neighbor tables
(dataloader.py:64-66).  ``match`` then selects keypoints and matches all 8 pairs with one top-k launch and one pair of
matching launches (ops.topk_scores, ops.mutual_nn_batched).
"""
import torch
import torch.nn.functional as F

from .train import TrainStep


class InferStep(TrainStep):
    reverse_tables = False   # forward only

    def __init__(self, model, config, neighborhood_limits, device, clouds=1, group=0):
        self.clouds = int(clouds)
        self.group = int(group)        # clouds per reference batch when several are stacked (0: the whole batch is one)
        if self.group and self.clouds % self.group:
            raise ValueError("clouds (%d) must be a multiple of group (%d)" % (self.clouds, self.group))
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
        for p in item:
            n = int(p.shape[0])
            st.pts[off:off + n].copy_(torch.as_tensor(p), non_blocking=True)
            off += n
        st.lens.copy_(torch.tensor([int(p.shape[0]) for p in item], dtype=torch.int32))   # one small copy, not 16 fills

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

    # ---- 8 pairs per batch: keypoints + mutual-NN of every pair --------------------------------------------------------
    def segments(self, item):
        """int32 [clouds, 2] (offset, length) of every cloud of ``item`` in the stacked rows (level 0 = the input itself,
        so the host knows them), on the device."""
        lens = [int(p.shape[0]) for p in item]
        offs = [0]
        for n in lens[:-1]:
            offs.append(offs[-1] + n)
        return torch.tensor([[o, n] for o, n in zip(offs, lens)], dtype=torch.int32, device=self.device)

    def match(self, item, feats, scores, num_points=None):
        """Mutual nearest-neighbor correspondences of every pair (clouds 2p, 2p+1) of a stacked batch -- the reference's
        per-pair ``build_correspondence`` (geometric_registration/common.py:5-21), on all descriptors or, with
        ``num_points``, on the top-scoring keypoints of each cloud (test.py:56-57).  Returns (row_argmin, mutual, seg):
        without keypoints indexed by stacked row (pair-local target indices); with keypoints [P, num_points] tables
        over the selected rows plus the selection itself (cloud-local row indices, ascending score)."""
        from . import ops
        seg = self.segments(item)
        P = seg.shape[0] // 2
        if num_points is None:
            pair = torch.cat([seg[0::2], seg[1::2]], dim=1).contiguous()            # src_off, src_len, tgt_off, tgt_len
            mx = max(int(p.shape[0]) for p in item)
            row, col, mutual = ops.mutual_nn_batched(feats, feats, pair, mx, mx)
            return row, mutual, seg
        k = int(num_points)
        sel = ops.topk_scores(scores, seg, k)                                       # [2P, k] cloud-local, -1 padded
        live = (sel >= 0)
        rows = (sel.clamp(min=0) + seg[:, :1]).long().reshape(-1)                    # stacked rows of the keypoints
        desc = feats.index_select(0, rows)                                           # [2P*k, C]
        n_live = live.sum(dim=1).to(torch.int32)                                     # keypoints per cloud (<= k)
        base = torch.arange(2 * P, device=self.device, dtype=torch.int32) * k + (k - n_live)   # live ones are the tail
        pair = torch.stack([base[0::2], n_live[0::2], base[1::2], n_live[1::2]], dim=1).contiguous()
        row, col, mutual = ops.mutual_nn_batched(desc, desc, pair, k, k)
        return row.view(2 * P, k)[0::2], mutual.view(2 * P, k)[0::2], sel

