"""Fused f32 GEMM (csrc/gemm.hip) through the C ABI against an fp64 restatement of the reference's arithmetic:
nn.Linear + bias + LeakyReLU of the unary blocks (models/blocks.py:481-541,686), their autograd backward, and the
`weighted_features @ weights` contraction of KPConv (blocks.py:375-380).  Tolerance: 2e-5 of the output's max magnitude
(f32 FMA chains against fp64; the north star asks 1e-4 on descriptors)."""
import numpy as np
import pytest
import torch

from d3feat_pytorch_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-5


def _rel(a, b):
    b = b.double()
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _rand(g, *shape):
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1)


SHAPES = [(154, 512, 7680), (571, 1024, 3072), (2053, 128, 256), (7961, 64, 128), (33, 20, 36), (64, 64, 32),
          (1, 4, 4), (130, 132, 100), (2053, 512, 1920)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("a_ks,b_ks", [(False, False), (False, True), (True, True), (True, False)])
def test_gemm_layouts_match_fp64(M, N, K, a_ks, b_ks):
    if a_ks and M % 4:
        M = (M + 3) // 4 * 4   # a KS operand is contiguous along its output index: multiple of 4
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A, B = _rand(g, M, K), _rand(g, K, N)
    ref = A @ B
    Ad = (A.t().contiguous() if a_ks else A).float().to(DEV)
    Bd = (B if b_ks else B.t().contiguous()).float().to(DEV)
    out = ops.gemm(Ad, Bd, a_ks=a_ks, b_ks=b_ks)
    assert out.shape == (M, N)
    assert _rel(out.cpu(), ref) < TOL
    # split reductions are summed in a fixed order: bit-identical on a second call
    assert torch.equal(out, ops.gemm(Ad, Bd, a_ks=a_ks, b_ks=b_ks))


@pytest.mark.parametrize("M,N,K", [(571, 256, 512), (154, 2048, 512), (2053, 128, 512), (70, 64, 64)])
def test_gemm_epilogue_is_the_unary_block(M, N, K):
    """act(x W^T / d + b1 + add + b2) and the nearest-upsampled residual (architectures.py:311-314)."""
    g = torch.Generator().manual_seed(5)
    x, W = _rand(g, M, K), _rand(g, N, K)
    b1, b2, add = _rand(g, N), _rand(g, N), _rand(g, M, N)
    d = torch.randint(1, 40, (M,), generator=g).double()
    lrelu = lambda v, s: torch.where(v > 0, v, v * s)   # noqa: E731
    xd, Wd = x.float().to(DEV), W.float().to(DEV)
    zi = torch.ones(77, device=DEV)
    out = ops.gemm(xd, Wd, bias1=b1.float().to(DEV), bias2=b2.float().to(DEV), add=add.float().to(DEV),
                   row_div=d.float().to(DEV), slope=0.1, zero_init=zi)
    ref = lrelu((x @ W.t()) / d[:, None] + b1 + add + b2, 0.1)
    assert _rel(out.cpu(), ref) < TOL
    assert float(zi.abs().max()) == 0.0
    # coarse residual through an index column, shadow index -> nothing added; out is a column block of a wider matrix
    Nc = 37
    coarse = _rand(g, Nc, N)
    idx = torch.randint(0, Nc + 1, (M, 3), generator=g).to(torch.int32)
    wide = torch.zeros((M, N + 8), device=DEV)
    ops.gemm(xd, Wd, out=wide[:, 4:4 + N], add=coarse.float().to(DEV), add_idx=idx.to(DEV), add_rows=Nc, slope=1.0)
    pad = torch.cat([coarse, torch.zeros(1, N, dtype=torch.float64)])
    ref = x @ W.t() + pad[idx[:, 0].long()]
    assert _rel(wide[:, 4:4 + N].cpu(), ref) < TOL
    assert float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + N:].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(571, 256, 1024), (154, 512, 2048), (7961, 64, 256), (2053, 512, 128)])
def test_gemm_backward_of_the_unary_block(M, N, K):
    """grad_x = (g * act'(out)) W (+ deposited), grad_W = (g * act'(out))^T x, grad_b = column sums -- the mask is
    evaluated inside the operand staging, the bias gradient is a by-product of the weight-gradient GEMM."""
    g = torch.Generator().manual_seed(9)
    x, W = _rand(g, M, K), _rand(g, N, K)
    out, go, dep = _rand(g, M, N), _rand(g, M, N), _rand(g, M, K)
    gm = go * torch.where(out > 0, 1.0, 0.1)
    xd, Wd, outd, god = (t.float().to(DEV) for t in (x, W, out, go))
    gx = ops.gemm(god, Wd, b_ks=True, a_mask=outd, mask_slope=0.1, add=dep.float().to(DEV))
    assert _rel(gx.cpu(), gm @ W + dep) < TOL
    gb, gb2 = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    gw = ops.gemm(god, xd, a_ks=True, b_ks=True, a_mask=outd, mask_slope=0.1, rowsum=gb, rowsum2=gb2)
    assert _rel(gw.cpu(), gm.t() @ x) < TOL
    assert _rel(gb.cpu(), gm.sum(0)) < TOL and torch.equal(gb, gb2)
    assert torch.equal(gw, ops.gemm(god, xd, a_ks=True, b_ks=True, a_mask=outd, mask_slope=0.1))


def test_gemm_rejects_what_it_cannot_run():
    a = torch.zeros((8, 6), device=DEV)     # reduction length 6: not a multiple of 4 floats
    with pytest.raises(RuntimeError):
        ops.gemm(a, torch.zeros((4, 6), device=DEV))
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros((8, 8)), torch.zeros((4, 8)))   # host tensors: there is no CPU path
