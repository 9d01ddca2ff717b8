"""GPU checks of the bf16x3 linear kernels (csrc/linear_x3.hip) against float64 torch on random data, every layer shape
they serve, one- and two-input forms: forward, input gradient, weight and bias gradients.  Tolerance: 1e-4 of the
largest entry (bf16x3 products carry ~1e-5 relative error)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(64, 128), (128, 64), (64, 64), (64, 32), (32, 64), (32, 32)]


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('M,K', SHAPES)
@pytest.mark.parametrize('two', [False, True])
def test_linear_x3_forward_backward_vs_float64(M, K, two):
    dev = _dev()
    from deepgate import ops
    if ops.PRECISION != 'x3':
        pytest.skip('bf16x3 mode only')
    assert ops._lin_x3(M, K)
    torch.manual_seed(M * 1000 + K + int(two))
    N = 64 * 37 + 19                                  # a partial last tile
    K1 = K // 2 if two else K
    x1 = torch.randn(N, K1, device=dev, requires_grad=True)
    x2 = torch.randn(N, K - K1, device=dev, requires_grad=True) if two else None
    W = (torch.randn(M, K, device=dev) * 0.2).requires_grad_(True)
    b = torch.randn(M, device=dev, requires_grad=True)
    y = ops.linear(x1, W, b, x2=x2)
    gy = torch.randn(N, M, device=dev)
    y.backward(gy)
    xd = torch.cat([x1, x2], 1).detach().double() if two else x1.detach().double()
    Wd, bd, gd = W.detach().double(), b.detach().double(), gy.double()
    assert _rel(y.detach(), xd @ Wd.t() + bd) <= 1e-4
    gx = gd @ Wd
    assert _rel(x1.grad, gx[:, :K1]) <= 1e-4
    if two:
        assert _rel(x2.grad, gx[:, K1:]) <= 1e-4
    assert _rel(W.grad, gd.t() @ xd) <= 1e-4
    assert _rel(b.grad, gd.sum(0)) <= 1e-4
