"""The C++ autograd bindings (ffwm_amd/lib/ffwm_torch_ext.so) against the ctypes / Python autograd Functions they
shadow: same kernels, same results -- forward values and every gradient (run with ``-m gpu``)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ext():
    from ffwm_amd import _ext
    m = _ext.get()
    assert m is not None, "ffwm_torch_ext.so is not built (python -m ffwm_amd.build)"
    return m


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _pair(fa, fb, inputs, tol=0.0):
    """Run both callables on cloned leaf inputs (None entries pass through), compare outputs and all gradients."""
    outs, grads = [], []
    for f in (fa, fb):
        xs = [None if t is None else t.clone().requires_grad_(t.is_floating_point()) for t in inputs]
        y = f(*xs)
        go = torch.rand(y.shape, generator=_g(99)).to(y.device)
        y.backward(go)
        outs.append(y.detach())
        grads.append([None if t is None else t.grad for t in xs])
    assert (outs[0] - outs[1]).abs().max().item() <= tol * (1 + outs[1].abs().max().item())
    for a, b in zip(*grads):
        assert (a is None) == (b is None)
        if a is not None:
            assert (a - b).abs().max().item() <= max(tol, 1e-6) * (1 + b.abs().max().item())


@pytest.mark.parametrize("shape", [(4, 37, 9, 11), (8, 64, 32, 32), (2, 5, 128, 128)])
def test_bn_lrelu(ext, shape):
    from ffwm_amd.norm import _BnLreluFunction
    g = _g(1)
    x = torch.randn(*shape, generator=g).to(DEV)
    w, b = torch.rand(shape[1], generator=g).to(DEV) + 0.5, torch.randn(shape[1], generator=g).to(DEV)
    rm_a, rv_a = torch.zeros(shape[1], device=DEV), torch.ones(shape[1], device=DEV)
    rm_b, rv_b = rm_a.clone(), rv_a.clone()
    _pair(lambda x_, w_, b_: ext.bn_lrelu(x_, w_, b_, rm_a, rv_a, 1e-5, 0.1, 0.2),
          lambda x_, w_, b_: _BnLreluFunction.apply(x_, w_, b_, rm_b, rv_b, 1e-5, 0.1, 0.2), [x, w, b], tol=1e-6)
    assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)


def test_bias_relu_and_mfm(ext):
    from ffwm_amd.external_function import BiasReLUFunction, MaxFeatureMapFunction
    g = _g(2)
    h, b = torch.randn(3, 10, 7, 9, generator=g).to(DEV), torch.randn(10, generator=g).to(DEV)
    _pair(ext.bias_relu, BiasReLUFunction.apply, [h, b])
    x, b2 = torch.randn(4, 96, 16, 16, generator=g).to(DEV), torch.randn(96, generator=g).to(DEV)
    _pair(ext.mfm, MaxFeatureMapFunction.apply, [x, b2])
    _pair(lambda x_: ext.mfm(x_, None), lambda x_: MaxFeatureMapFunction.apply(x_, None), [x])
