"""Drop-in mirror of the reference's ``models/blocks.py`` operator API on the HIP kernels.

Same names, constructor arguments, parameter names (``weights`` [K,Cin,Cout], ``kernel_points`` [K,3] -- fixed by
checkpoints) and call signatures as the reference (models/blocks.py: gather:35, closest_pool:79, max_pool:94,
global_average:113, KPConv:143, block_decider:395, BatchNormBlock:441, UnaryBlock:481, LastUnaryBlock:518,
SimpleBlock:544, ResnetBottleneckBlock:601, GlobalAverageBlock:689, NearestUpsampleBlock:702, MaxPoolBlock:720), so
``models/architectures.py`` (``from models.blocks import *``; ``block_decider(...)``; ``isinstance(m, KPConv)``)
runs on top of it unchanged.  The operators themselves are the hand-written HIP kernels of
``libd3feat_hip.so`` -- there is no PyTorch implementation of the hot path in this file.

The 'constant' / 'gaussian' influences and the 'closest' aggregation (blocks.py:327-352; never enabled by D3Feat's
config, config.py:39-41) run on the general-path kernels, and so does the deformable / modulated KPConv
(blocks.py:187-203,243-324,365-366; config.py:45-46): offsets from a rigid KPConv on the ordinary kernels, aggregation
with per-query kernel points and its gradients in csrc/kpconv_deform.hip.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.init import kaiming_uniform_
from torch.nn.parameter import Parameter

from .. import ops
from ..kernels.kernel_points import load_kernels
# north-star spellings of the two preprocessing operators live next to KPConv as well
from ..datasets.dataloader import (batch_grid_subsampling_kpconv, batch_neighbors_kpconv,  # noqa: F401
                                   batch_grid_subsampling, batch_neighbors)

__all__ = ['gather', 'radius_gaussian', 'closest_pool', 'max_pool', 'global_average', 'KPConv', 'block_decider',
           'BatchNormBlock', 'UnaryBlock', 'LastUnaryBlock', 'SimpleBlock', 'ResnetBottleneckBlock',
           'GlobalAverageBlock', 'NearestUpsampleBlock', 'MaxPoolBlock', 'batch_neighbors_kpconv',
           'batch_grid_subsampling_kpconv', 'batch_neighbors', 'batch_grid_subsampling',
           'nn', 'torch', 'math', 'Parameter']


def gather(x, idx, method=2):
    """x[idx] with shape idx.shape + x.shape[1:] (reference blocks.py:35-66; all three methods are equal in value)."""
    if method not in (0, 1, 2):
        raise ValueError('Unkown method')
    return x[idx.long()]


def radius_gaussian(sq_r, sig, eps=1e-9):
    return torch.exp(-sq_r / (2 * sig ** 2 + eps))


def closest_pool(x, inds):
    """Features of the closest neighbor: x'[inds[:, 0]] with a zero shadow row (reference blocks.py:79-91)."""
    return ops.closest_pool(x, inds)


def max_pool(x, inds):
    """Channel-wise max over each neighborhood, zero shadow row (reference blocks.py:94-110)."""
    return ops.max_pool(x, inds)


def global_average(x, batch_lengths):
    """Per-cloud mean of the stacked features (reference blocks.py:113-134)."""
    out, i0 = [], 0
    for length in batch_lengths:
        length = int(length)
        out.append(torch.mean(x[i0:i0 + length], dim=0))
        i0 += length
    return torch.stack(out)


class KPConv(nn.Module):
    """Kernel point convolution (reference blocks.py:143-387), rigid or deformable (+ modulated)."""

    def __init__(self, kernel_size, p_dim, in_channels, out_channels, KP_extent, radius,
                 fixed_kernel_points='center', KP_influence='linear', aggregation_mode='sum',
                 deformable=False, modulated=False):
        super(KPConv, self).__init__()
        self.K = kernel_size
        self.p_dim = p_dim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.radius = radius
        self.KP_extent = KP_extent
        self.fixed_kernel_points = fixed_kernel_points
        self.KP_influence = KP_influence
        self.aggregation_mode = aggregation_mode
        self.deformable = deformable
        self.modulated = modulated
        self.min_d2 = None
        self.deformed_KP = None
        self.offset_features = None
        ops.kpconv_mode(KP_influence, aggregation_mode)   # unknown names raise like the reference (:344,:352)
        if p_dim != 3 or kernel_size > 16:
            raise NotImplementedError('HIP KPConv supports 3-D points and at most 16 kernel points')
        self.weights = Parameter(torch.zeros((self.K, in_channels, out_channels), dtype=torch.float32),
                                 requires_grad=True)
        if deformable:
            # construction order = the reference's (:187-203): the offset convolution draws its weights and its kernel
            # points from the global RNGs BEFORE this layer initialises its own
            self.offset_dim = (self.p_dim + 1) * self.K if modulated else self.p_dim * self.K
            self.offset_conv = KPConv(self.K, self.p_dim, self.in_channels, self.offset_dim, KP_extent, radius,
                                      fixed_kernel_points=fixed_kernel_points, KP_influence=KP_influence,
                                      aggregation_mode=aggregation_mode)
            self.offset_bias = Parameter(torch.zeros(self.offset_dim, dtype=torch.float32), requires_grad=True)
        else:
            self.offset_dim = None
            self.offset_conv = None
            self.offset_bias = None
        self.reset_parameters()
        self.kernel_points = self.init_KP()

    def reset_parameters(self):
        kaiming_uniform_(self.weights, a=math.sqrt(5))
        if self.deformable:
            nn.init.zeros_(self.offset_bias)

    def init_KP(self):
        kp = load_kernels(self.radius, self.K, dimension=self.p_dim, fixed=self.fixed_kernel_points)
        return Parameter(torch.tensor(kp, dtype=torch.float32), requires_grad=False)

    def forward(self, q_pts, s_pts, neighb_inds, x):
        if not self.deformable:
            return ops.kpconv(q_pts, s_pts, neighb_inds, x, self.kernel_points, self.weights, self.KP_extent,
                              self.KP_influence, self.aggregation_mode)
        # offsets (and modulations) from the rigid offset convolution (:244-256)
        self.offset_features = self.offset_conv(q_pts, s_pts, neighb_inds, x) + self.offset_bias
        if self.modulated:
            unscaled = self.offset_features[:, :self.p_dim * self.K].reshape(-1, self.K, self.p_dim)
            modulations = 2 * torch.sigmoid(self.offset_features[:, self.p_dim * self.K:])
        else:
            unscaled = self.offset_features.view(-1, self.K, self.p_dim)
            modulations = None
        out, self.min_d2, self.deformed_KP = ops.kpconv_deformable(
            q_pts, s_pts, neighb_inds, x, self.kernel_points, self.weights, self.KP_extent,
            unscaled * self.KP_extent, modulations, self.KP_influence, self.aggregation_mode)
        return out

    def __repr__(self):
        return 'KPConv(radius: {:.2f}, extent: {:.2f}, in_feat: {:d}, out_feat: {:d})'.format(
            self.radius, self.KP_extent, self.in_channels, self.out_channels)


def block_decider(block_name, radius, in_dim, out_dim, layer_ind, config):
    """Block factory with the reference's block vocabulary (blocks.py:395-438)."""
    if block_name == 'unary':
        return UnaryBlock(in_dim, out_dim, config.use_batch_norm, config.batch_norm_momentum)
    if block_name == 'last_unary':
        return LastUnaryBlock(in_dim, 32, config.use_batch_norm, config.batch_norm_momentum)
    if block_name.startswith('simple'):
        return SimpleBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    if block_name.startswith('resnetb'):
        return ResnetBottleneckBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    if block_name in ('max_pool', 'max_pool_wide'):
        return MaxPoolBlock(layer_ind)
    if block_name == 'global_average':
        return GlobalAverageBlock()
    if block_name == 'nearest_upsample':
        return NearestUpsampleBlock(layer_ind)
    raise ValueError('Unknown block name in the architecture definition : ' + block_name)


class BatchNormBlock(nn.Module):
    """BatchNorm1d over the stacked points, or a learned bias when use_bn is False (reference blocks.py:441-478)."""

    def __init__(self, in_dim, use_bn, bn_momentum):
        super(BatchNormBlock, self).__init__()
        self.bn_momentum = bn_momentum
        self.use_bn = use_bn
        self.in_dim = in_dim
        if self.use_bn:
            self.batch_norm = nn.BatchNorm1d(in_dim, momentum=bn_momentum)
        else:
            self.bias = Parameter(torch.zeros(in_dim, dtype=torch.float32), requires_grad=True)

    def reset_parameters(self):
        nn.init.zeros_(self.bias)

    def forward(self, x, slope=1.0):
        """``slope`` != 1 fuses the block's LeakyReLU behind the normalisation (device path only)."""
        if self.use_bn:
            bn = self.batch_norm
            if x.is_cuda and x.dim() == 2:
                if self.training and bn.track_running_stats:
                    bn.num_batches_tracked.add_(1)       # nn.BatchNorm1d's own counter (state_dict parity)
                return ops.batch_norm(x, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                      self.training or not bn.track_running_stats, bn.momentum, bn.eps, slope=slope)
            y = bn(x.unsqueeze(2).transpose(0, 2)).transpose(0, 2).squeeze()
            return y if slope == 1.0 else F.leaky_relu(y, slope)
        if x.is_cuda and x.dim() == 2:
            return ops.bias_act(x, self.bias, slope=slope)
        y = x + self.bias
        return y if slope == 1.0 else F.leaky_relu(y, slope)

    def __repr__(self):
        return 'BatchNormBlock(in_feat: {:d}, momentum: {:.3f}, only_bias: {:s})'.format(
            self.in_dim, self.bn_momentum, str(not self.use_bn))


class UnaryBlock(nn.Module):
    """Linear + (BN | bias) + LeakyReLU(0.1) (reference blocks.py:481-515)."""

    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False):
        super(UnaryBlock, self).__init__()
        self.bn_momentum = bn_momentum
        self.use_bn = use_bn
        self.no_relu = no_relu
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.mlp = nn.Linear(in_dim, out_dim, bias=True)
        self.batch_norm = BatchNormBlock(out_dim, self.use_bn, self.bn_momentum)
        if not no_relu:
            self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, x, batch=None, residual=None, grad_holder=None, grad_deposit=None, pack_for=None):
        if not self.use_bn and x.is_cuda and x.dim() == 2:
            # Linear without its bias; both biases (+ residual) + LeakyReLU go into ONE epilogue launch whose backward
            # also yields the (shared) bias gradient -- no separate add / leaky / column-reduce kernels.  pack_for: the
            # output feeds a KPConv; the epilogue leaves its packed supports behind (ops.bias_act)
            return ops.linear_bias_act(x, self.mlp.weight, self.mlp.bias, residual, self.batch_norm.bias,
                                       slope=1.0 if (self.no_relu and residual is None) else 0.1,
                                       grad_holder=grad_holder, grad_deposit=grad_deposit, pack_for=pack_for)
        if residual is None:                     # BN (+ LeakyReLU) in one normalisation pass on the device
            return self.batch_norm(self.mlp(x), slope=1.0 if self.no_relu else 0.1)
        x = self.batch_norm(self.mlp(x))
        if x.is_cuda and x.dim() == 2:
            return ops.bias_act(x, add=residual, slope=0.1)
        return self.leaky_relu_res(x + residual)

    @staticmethod
    def leaky_relu_res(x):
        return F.leaky_relu(x, 0.1)

    def __repr__(self):
        return 'UnaryBlock(in_feat: {:d}, out_feat: {:d}, BN: {:s}, ReLU: {:s})'.format(
            self.in_dim, self.out_dim, str(self.use_bn), str(not self.no_relu))


class LastUnaryBlock(nn.Module):
    """Plain Linear head (reference blocks.py:518-541)."""

    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False):
        super(LastUnaryBlock, self).__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.mlp = nn.Linear(in_dim, out_dim, bias=True)

    def forward(self, x, batch=None):
        if x.is_cuda and x.dim() == 2:
            # same arithmetic as nn.Linear; the bias gradient comes from the fused epilogue's column reduction instead
            # of a multi-block library reduction (whose memset-initialised semaphores do not survive hipGraph replay)
            return ops.linear_bias_act(x, self.mlp.weight, self.mlp.bias, slope=1.0)
        return self.mlp(x)

    def __repr__(self):
        return 'LastUnaryBlock(in_feat: {:d}, out_feat: {:d})'.format(self.in_dim, self.out_dim)


def _layer_inputs(block_name, layer_ind, batch):
    """(queries, supports, neighbor table) of a block: strided blocks pool onto the next level (blocks.py:588-595)."""
    if 'strided' in block_name:
        return batch['points'][layer_ind + 1], batch['points'][layer_ind], batch['pools'][layer_ind]
    return batch['points'][layer_ind], batch['points'][layer_ind], batch['neighbors'][layer_ind]


def _pool_width(layer_ind, batch):
    """Keyword arguments for ops.max_pool of a strided block: the device-resident max neighbor count of the pooling
    table when the batch keeps its tables wider than the reference would (static shapes) -- ops.max_pool then reads the
    columns the reference's table has -- and, for a batch that stacks several reference batches (``_group``), the stack
    lengths of the pooled level so that every group uses its own width."""
    w = batch.get('pools_width') if isinstance(batch, dict) else None
    if w is None:
        return {}
    group = batch.get('_group', 0)
    if group:
        return {'width': w[layer_ind], 'groups': (batch['stack_lengths'][layer_ind + 1], group)}
    return {'width': w[layer_ind]}


def _make_kpconv(block_name, in_dim, out_dim, radius, config):
    extent = radius * config.KP_extent / config.conv_radius
    return KPConv(config.num_kernel_points, config.in_points_dim, in_dim, out_dim, extent, radius,
                  fixed_kernel_points=config.fixed_kernel_points, KP_influence=config.KP_influence,
                  aggregation_mode=config.aggregation_mode, deformable='deform' in block_name,
                  modulated=config.modulated)


class SimpleBlock(nn.Module):
    """KPConv(in -> out/2) + bias/BN + LeakyReLU (reference blocks.py:544-598)."""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super(SimpleBlock, self).__init__()
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.layer_ind = layer_ind
        self.block_name = block_name
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.KPConv = _make_kpconv(block_name, in_dim, out_dim // 2, radius, config)
        self.batch_norm = BatchNormBlock(out_dim // 2, self.use_bn, self.bn_momentum)
        self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, x, batch):
        q_pts, s_pts, inds = _layer_inputs(self.block_name, self.layer_ind, batch)
        if not self.use_bn and x.is_cuda and not self.KPConv.deformable:
            return ops.kpconv_bias_act(q_pts, s_pts, inds, x, self.KPConv.kernel_points, self.KPConv.weights,
                                       self.KPConv.KP_extent, self.batch_norm.bias, slope=0.1,
                                       influence=self.KPConv.KP_influence, aggregation=self.KPConv.aggregation_mode)
        y = self.KPConv(q_pts, s_pts, inds, x)
        return self.batch_norm(y, slope=0.1)


class ResnetBottleneckBlock(nn.Module):
    """unary(in -> out/4) -> KPConv -> unary(out/4 -> out) + shortcut (reference blocks.py:601-686)."""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super(ResnetBottleneckBlock, self).__init__()
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.block_name = block_name
        self.layer_ind = layer_ind
        self.in_dim = in_dim
        self.out_dim = out_dim
        mid = out_dim // 4
        self.unary1 = UnaryBlock(in_dim, mid, self.use_bn, self.bn_momentum) if in_dim != mid else nn.Identity()
        self.KPConv = _make_kpconv(block_name, mid, mid, radius, config)
        self.batch_norm_conv = BatchNormBlock(mid, self.use_bn, self.bn_momentum)
        self.unary2 = UnaryBlock(mid, out_dim, self.use_bn, self.bn_momentum, no_relu=True)
        if in_dim != out_dim:
            self.unary_shortcut = UnaryBlock(in_dim, out_dim, self.use_bn, self.bn_momentum, no_relu=True)
        else:
            self.unary_shortcut = nn.Identity()
        self.leaky_relu = nn.LeakyReLU(0.1)

    def fuses_gradients(self, features):
        """Whether forward(features) takes the training path that routes the gradients of `features` through
        ops.GradHolder (a skip tensor may only be marked for it, architectures.KPFCNN.mark_skip, when this holds)."""
        return ((not self.use_bn) and features.is_cuda and isinstance(self.unary1, UnaryBlock) and
                features.requires_grad and not self.KPConv.deformable)

    def forward(self, features, batch):
        q_pts, s_pts, inds = _layer_inputs(self.block_name, self.layer_ind, batch)
        # `features` feeds unary1 and the shortcut: the shortcut branch deposits its gradient, unary1's grad-input GEMM
        # adds it (ops.GradHolder) -- no separate accumulation launch
        fuse = self.fuses_gradients(features)
        holder = ops.GradHolder() if fuse else None
        pack = None
        if not self.use_bn and features.is_cuda and isinstance(self.unary1, UnaryBlock) and not self.KPConv.deformable:
            # unary1's epilogue packs the supports of the KPConv it feeds; the grad_x scatter target is cleared along
            # with it unless that layer's grad-input is the gather form (reverse table present, many rows)
            gather = getattr(inds, '_d3f_rev', None) is not None and ops.wants_reverse_table(s_pts.shape[0])
            pack = (s_pts, features.requires_grad and not gather)
        if isinstance(self.unary1, UnaryBlock):
            x = self.unary1(features, grad_holder=holder, pack_for=pack)
        else:
            x = self.unary1(features)
        if fuse:
            strided = 'strided' in self.block_name
            # a skip tensor also feeds the decoder, whose gradient arrives first: the pooling backward scatters on top
            incoming = getattr(features, '_d3f_grad_in', None) if strided else None
            shortcut = ops.max_pool(features, inds, grad_deposit=holder, grad_incoming=incoming,
                                    **_pool_width(self.layer_ind, batch)) if strided else features
            pair = isinstance(self.unary_shortcut, UnaryBlock) and ops.linear_pair_supported(
                q_pts.shape[0], self.unary2.in_dim, shortcut.shape[1], self.out_dim, shortcut, self.unary2.mlp.weight,
                self.unary_shortcut.mlp.weight)
            if isinstance(self.unary_shortcut, UnaryBlock) and not pair:
                shortcut = self.unary_shortcut(shortcut, grad_deposit=None if strided else holder)
            elif not strided and not pair:
                shortcut = ops.grad_tap(features, holder)
            x = ops.kpconv_bias_act(q_pts, s_pts, inds, x, self.KPConv.kernel_points, self.KPConv.weights,
                                    self.KPConv.KP_extent, self.batch_norm_conv.bias, slope=0.1,
                                    influence=self.KPConv.KP_influence, aggregation=self.KPConv.aggregation_mode)
            if pair:
                # unary2 and the shortcut unary write ONE output in one launch: the shortcut tensor is never formed
                u2, us = self.unary2, self.unary_shortcut
                return ops.linear_pair_bias_act(x, u2.mlp.weight, u2.mlp.bias, u2.batch_norm.bias, shortcut, us.mlp.weight,
                                                us.mlp.bias, us.batch_norm.bias, slope=0.1,
                                                grad_deposit2=None if strided else holder)
            return self.unary2(x, residual=shortcut)
        if not self.use_bn and x.is_cuda and not self.KPConv.deformable:
            x = ops.kpconv_bias_act(q_pts, s_pts, inds, x, self.KPConv.kernel_points, self.KPConv.weights,
                                    self.KPConv.KP_extent, self.batch_norm_conv.bias, slope=0.1,
                                    influence=self.KPConv.KP_influence, aggregation=self.KPConv.aggregation_mode)
        else:
            x = self.batch_norm_conv(self.KPConv(q_pts, s_pts, inds, x), slope=0.1)
        shortcut = ops.max_pool(features, inds, **_pool_width(self.layer_ind, batch)) \
            if 'strided' in self.block_name else features
        shortcut = self.unary_shortcut(shortcut)
        return self.unary2(x, residual=shortcut)  # leaky(unary2(x) + shortcut); in unary2's epilogue without BN


class GlobalAverageBlock(nn.Module):
    def __init__(self):
        super(GlobalAverageBlock, self).__init__()

    def forward(self, x, batch):
        return global_average(x, batch['stack_lengths'][-1])


class NearestUpsampleBlock(nn.Module):
    """Nearest-neighbor upsampling onto the finer level (reference blocks.py:702-717)."""

    def __init__(self, layer_ind):
        super(NearestUpsampleBlock, self).__init__()
        self.layer_ind = layer_ind

    def forward(self, x, batch):
        return closest_pool(x, batch['upsamples'][self.layer_ind - 1])

    def __repr__(self):
        return 'NearestUpsampleBlock(layer: {:d} -> {:d})'.format(self.layer_ind, self.layer_ind - 1)


class MaxPoolBlock(nn.Module):
    def __init__(self, layer_ind):
        super(MaxPoolBlock, self).__init__()
        self.layer_ind = layer_ind

    def forward(self, x, batch):
        return ops.max_pool(x, batch['pools'][self.layer_ind + 1], **_pool_width(self.layer_ind + 1, batch))
