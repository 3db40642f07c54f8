"""KPFCNN encoder/decoder + detector head: host-side mirror of the reference's ``models/architectures.py``.

The reference file itself runs unchanged on top of ``d3feat_pytorch_amd.models.blocks`` (that is the drop-in
boundary); this module is the copy-free equivalent that travels with the repo, because
the reference tree is not present on the GPU box.  Same constructor (``KPFCNN(config)``), same sub-module names
(``encoder_blocks`` / ``decoder_blocks`` -> identical ``state_dict`` keys), same forward contract
(architectures.py:299-320: ``(F.normalize(x), scores)``), and ``detection_scores`` (architectures.py:322-368) is the
fused HIP kernel instead of ~25 PyTorch ops over a gathered [N,H,C] tensor.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .blocks import (KPConv, LastUnaryBlock, NearestUpsampleBlock, ResnetBottleneckBlock, UnaryBlock,  # noqa: F401
                     block_decider)

# False: the descriptor head runs on the concatenated level-0 matrix as in rounds 1-5 (A/B measurements)
UPSAMPLED_HEAD = True


def p2p_fitting_regularizer(net, deform_fitting_power=None, repulse_extent=None):
    """Regulariser of the deformable kernel points (reference architectures.py:22-55; used by its classification
    network, plain tensor algebra on what each deformable KPConv left on itself): 2 x the mean normalised squared
    distance of every deformed kernel point to its nearest input point (``min_d2``, through which the gradient reaches
    the offsets) + a repulsion between deformed kernel points of one query closer than ``repulse_extent``."""
    power = net.deform_fitting_power if deform_fitting_power is None else deform_fitting_power
    repulse = net.repulse_extent if repulse_extent is None else repulse_extent
    fitting, repulsive = 0, 0
    for m in net.modules():
        if not (isinstance(m, KPConv) and m.deformable):
            continue
        fitting = fitting + (m.min_d2 / (m.KP_extent ** 2)).abs().mean()
        locs = m.deformed_KP / m.KP_extent
        K = locs.shape[1]
        for i in range(K):
            others = torch.cat([locs[:, :i, :], locs[:, i + 1:, :]], dim=1).detach()
            dist = torch.sqrt(torch.sum((others - locs[:, i:i + 1, :]) ** 2, dim=2))
            rep = torch.sum(torch.clamp_max(dist - repulse, max=0.0) ** 2, dim=1)
            repulsive = repulsive + rep.abs().mean() / K
    return power * (2 * fitting + repulsive)


def _moves_level(block):
    return any(tag in block for tag in ('pool', 'strided', 'upsample', 'global'))


class KPFCNN(nn.Module):
    def __init__(self, config, verbose=False):
        super(KPFCNN, self).__init__()
        arch = list(config.architecture)
        layer = 0
        r = config.first_subsampling_dl * config.conv_radius
        in_dim = config.in_features_dim
        out_dim = config.first_features_dim
        self.K = config.num_kernel_points

        # ---- encoder: every block up to the first upsampling one (architectures.py:209-254)
        self.encoder_blocks = nn.ModuleList()
        self.encoder_skip_dims = []
        self.encoder_skips = []
        first_up = len(arch)
        for i, block in enumerate(arch):
            if 'equivariant' in block and out_dim % 3 != 0:
                raise ValueError('Equivariant block but features dimension is not a factor of 3')
            if _moves_level(block):
                self.encoder_skips.append(i)
                self.encoder_skip_dims.append(in_dim)
            if 'upsample' in block:
                first_up = i
                break
            self.encoder_blocks.append(block_decider(block, r, in_dim, out_dim, layer, config))
            in_dim = out_dim // 2 if 'simple' in block else out_dim
            if 'pool' in block or 'strided' in block:
                layer += 1
                r *= 2
                out_dim *= 2

        # ---- decoder: skip features are concatenated in front of the block that follows an upsampling
        self.decoder_blocks = nn.ModuleList()
        self.decoder_concats = []
        for j, block in enumerate(arch[first_up:]):
            if j > 0 and 'upsample' in arch[first_up + j - 1]:
                in_dim += self.encoder_skip_dims[layer]
                self.decoder_concats.append(j)
            self.decoder_blocks.append(block_decider(block, r, in_dim, out_dim, layer, config))
            in_dim = out_dim
            if 'upsample' in block:
                layer -= 1
                r *= 0.5
                out_dim = out_dim // 2
        if verbose:
            print(self)

    def forward_raw(self, batch):
        """(un-normalised descriptors [N,C], scores [N,1]) -- `forward` without the final F.normalize, for callers
        that only need a few normalised rows (the training step: ops.select_normalize)."""
        x = batch['features'].detach()   # (the reference clones, architectures.py:301; nothing writes into it here)
        skips = []
        for i, op in enumerate(self.encoder_blocks):
            if i in self.encoder_skips:
                skips.append(self.mark_skip(x, op))
            x = op(x, batch)
        x = self._decode(x, skips, batch)
        return x, self.detection_scores(batch, x)

    @staticmethod
    def mark_skip(x, op):
        """A skip tensor is consumed by the decoder (first in backward) and by the strided block `op` that follows it:
        the decoder deposits its gradient, the block's pooling backward accumulates on top of it (ops.GradHolder)
        instead of autograd adding the two."""
        if x.requires_grad and isinstance(op, ResnetBottleneckBlock) and 'strided' in op.block_name \
                and op.fuses_gradients(x):
            x._d3f_grad_in = ops.GradHolder()
        return x

    def _decode(self, x, skips, batch):
        """Decoder blocks with the skip concatenations of the reference (architectures.py:309-314)."""
        pending = None  # an upsampling block whose output is concatenated right away: one launch does both
        for j, op in enumerate(self.decoder_blocks):
            if j in self.decoder_concats:
                if pending is not None:
                    inds = batch['upsamples'][pending.layer_ind - 1]
                    pending = None
                    if isinstance(op, UnaryBlock) and not op.use_bn and (x.shape[1] * 4) % 16 == 0:
                        # unary block right after upsample + concat: the upsampled half of the product is computed
                        # on the coarse rows and upsampled in the epilogue (ops.upsample_linear_bias_act)
                        skip = skips.pop()
                        x = ops.upsample_linear_bias_act(x, inds, skip, op.mlp.weight, op.mlp.bias,
                                                         op.batch_norm.bias, slope=1.0 if op.no_relu else 0.1,
                                                         skip_grad_deposit=getattr(skip, '_d3f_grad_in', None))
                        continue
                    if isinstance(op, LastUnaryBlock) and UPSAMPLED_HEAD and (x.shape[1] * 4) % 16 == 0 \
                            and op.mlp.bias is not None:
                        # the descriptor head (plain Linear, reference blocks.py:518-541) behind the last upsample +
                        # concat: the same split -- the 256 upsampled channels are multiplied on the 4.8x fewer coarse
                        # rows, the level-0 concatenation [N, 384] is never formed, and the backward pools 32 gradient
                        # columns back to the coarse rows instead of scattering 256 (round 6)
                        skip = skips.pop()
                        x = ops.upsample_linear_bias_act(x, inds, skip, op.mlp.weight, op.mlp.bias, None, slope=1.0,
                                                         skip_grad_deposit=getattr(skip, '_d3f_grad_in', None))
                        continue
                    x = ops.closest_pool(x, inds, skip=skips.pop())
                else:
                    x = torch.cat([x, skips.pop()], dim=1)
            if isinstance(op, NearestUpsampleBlock) and (j + 1) in self.decoder_concats and x.is_cuda:
                pending = op
                continue
            x = op(x, batch)
        return x

    def forward(self, batch):
        x, scores = self.forward_raw(batch)
        return F.normalize(x, p=2, dim=-1), scores

    def detection_scores(self, inputs, features):
        """Saliency score of every point [N,1] (reference architectures.py:322-368); eval mode adds the
        local-maximum gate."""
        lens = inputs['stack_lengths'][0] if inputs.get('_static', False) else None  # capacity-shaped batch
        widths = inputs.get('neighbors_width')   # full-limit tables: the level-0 table's max count, on the device
        return ops.detection_scores(features, inputs['neighbors'][0], training=self.training, lens=lens,
                                    width=widths[0] if widths else None, group=inputs.get('_group', 0) if lens is not None else 0)
