import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def oracle():
    """CPU oracle (test infrastructure) - built on demand with gcc."""
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(autouse=True)
def _library_selection_state():
    """Tests must not depend on the order they run in: the pipelines switch MIOpen's find mode (`cudnn.benchmark`), its
    deterministic-solver flag and TunableOp on for the process; whatever a test changed is put back after it, so a later parity
    test sees the library defaults (an algorithm picked by timing - split-K with atomics, Winograd - moves float32 results in the
    last digits, enough to trip a 2e-3 gradient tolerance once in a while)."""
    import sys
    torch = sys.modules.get('torch')
    if torch is None:
        yield
        return
    saved = (torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic)
    tun = torch.cuda.tunable.is_enabled() if torch.cuda.is_available() else None
    yield
    torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = saved
    if tun is not None and torch.cuda.tunable.is_enabled() != tun:
        torch.cuda.tunable.enable(tun)
