"""Short runs of the randomised parity sweeps in tests/fuzz_cases.py (long runs:
`python -m tests.fuzz_cases step|forward|paths|init <cases> <seed>`).  This round they found the
signed-zero rule of masked gradients, the re-homing of starved slots in decomposed worlds and a
contraction-dependent rounding difference between the fused and the stand-alone forward."""
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def fuzz():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from tests import fuzz_cases
    return fuzz_cases


def test_fuzz_env_step_vs_oracle(fuzz):
    assert fuzz.fuzz_step(150, seed=101) == 0


def test_fuzz_forward_vs_oracle(fuzz):
    assert fuzz.fuzz_forward(100, seed=102) == 0


def test_fuzz_all_device_paths_identical(fuzz):
    assert fuzz.fuzz_paths(40, seed=103) == 0


def test_fuzz_init_vs_oracle(fuzz):
    assert fuzz.fuzz_init(30, seed=104) == 0


def test_fuzz_binned_step_vs_classic_step(fuzz):
    assert fuzz.fuzz_binned(12, seed=105) == 0


def test_fuzz_batched_replicas_vs_stand_alone_runs(fuzz):
    assert fuzz.fuzz_batched(10, seed=106, verbose=False) == 0


def test_fuzz_neural_automata_sensing_vs_oracle(fuzz):
    assert fuzz.fuzz_nca(25, seed=107, verbose=False) == 0
