"""NeuralAutomataAgent sensing, CPU side: the oracle restatement of ConvolutionModel (core/agent/evo.py:45-118) against
torch's own Conv2d (the primitive the reference calls), and the host-side mirror of the reference's unit tests
(test/unit/agent.py: model construction, shapes, serialisation) — nothing here launches a kernel."""
import os
import tempfile

import numpy as np
import pytest

th = pytest.importorskip('torch')
from torch import nn                      # noqa: E402

from oracle import cpu_ref as R           # noqa: E402

kernel_sizes_test = ((3,), (5,), (3, 3), (3, 5), (3, 5, 3),)      # test/unit/agent.py:11


@pytest.mark.parametrize('field_size', [(12, 12), (96, 96), (12, 8)])
@pytest.mark.parametrize('kernel_sizes', kernel_sizes_test)
def test_oracle_conv_stack_equals_torch_conv2d(field_size, kernel_sizes):
    th.manual_seed(sum(field_size) + len(kernel_sizes))
    convs = [nn.Conv2d(3, 3, k, padding='same', padding_mode='circular', bias=False).double() for k in kernel_sizes]
    x = th.rand(1, 3, *field_size, dtype=th.float64)
    y = x
    for c in convs:
        y = c(y)
    want = th.tanh(y)[0].detach().numpy()
    got = R.nca_sense(x[0].numpy(), [c.weight.detach().numpy() for c in convs])
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize('boundary', ['circular', 'zeros', 'reflect', 'replicate'])
@pytest.mark.parametrize('field_size,kernel_sizes', [((12, 8), (3, 5)), ((9, 7), (7,)), ((16, 16), (3, 3, 3))])
def test_oracle_conv_boundaries_equal_torch_padding_modes(boundary, field_size, kernel_sizes):
    """ConvolutionModel(boundary=…) (core/agent/evo.py:51,86): every padding_mode torch's Conv2d knows, oracle vs torch."""
    th.manual_seed(len(boundary) + sum(field_size))
    convs = [nn.Conv2d(3, 3, k, padding='same', padding_mode=boundary, bias=False).double() for k in kernel_sizes]
    x = th.rand(1, 3, *field_size, dtype=th.float64)
    y = x
    for c in convs:
        y = c(y)
    want = th.tanh(y)[0].detach().numpy()
    got = R.nca_sense(x[0].numpy(), [c.weight.detach().numpy() for c in convs], boundary=boundary)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14)


def test_oracle_conv_known_answers():
    """A single 3×3 kernel with one non-zero tap is a circular shift; the identity tap returns the input; channels mix
    by the (o, i) entry."""
    x = np.arange(2 * 4 * 5, dtype=np.float64).reshape(2, 4, 5)
    w = np.zeros((2, 2, 3, 3))
    w[0, 0, 1, 1] = 1.0                 # out0 = in0
    w[1, 0, 0, 2] = 2.0                 # out1[x, y] = 2·in0[x − 1, y + 1]
    out = R.conv2d_circular(x, w)
    assert np.array_equal(out[0], x[0])
    assert np.array_equal(out[1], 2.0 * np.roll(x[0], (1, -1), axis=(0, 1)))


def test_model_and_agent_interface_like_the_reference_tests():
    """test/unit/agent.py:14-23,72-96 against die_amd's classes (needs the library to import the package, no GPU)."""
    die_amd = pytest.importorskip('die_amd')
    model = die_amd.ConvolutionModel(num_act_channels=3, num_obs_channels=3, kernel_sizes=(3, 5), p_agent_dropout=0.)
    assert not all(th.all(k.weight == 0) for k in model.conv_layers())
    model.init_weights()
    out = model.forward(th.rand(1, 3, 12, 8))
    assert out.shape == (1, 3, 12, 8) and float(out.abs().max()) <= 1.0
    agent = die_amd.NeuralAutomataAgent(kernel_sizes=(3, 5))
    path = tempfile.mktemp()
    try:
        agent.save(path)
        assert os.path.exists(path)
        agent2 = die_amd.NeuralAutomataAgent.load(path)
        assert agent.init_params == agent2.init_params
        x = th.rand(1, 3, 16, 12)
        assert th.allclose(agent.model.forward(x), agent2.model.forward(x))
    finally:
        if os.path.exists(path):
            os.remove(path)
    # an already-open stream (allowed by the signature, as by the reference's single th.load): ONE read
    import io
    buf = io.BytesIO()
    agent.save(buf)
    buf.seek(0)
    agent3 = die_amd.NeuralAutomataAgent.load(buf)
    assert agent.init_params == agent3.init_params
    assert th.allclose(agent.model.forward(x), agent3.model.forward(x))
    assert agent.render()[0].shape == (2, 2, 3)
