import os
import sys

import pytest

# The CPU oracle is OpenMP-parallel PyTorch.  On a shared host the default active spin-waiting of idle OpenMP workers turned
# this suite's 40 s into 12 min twice (user time 80 min: eight threads spinning against a neighbour's load); passive waiting and
# a thread count capped at the cores we can actually see keep it bounded.  (Set before torch is imported anywhere.)
os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')
os.environ.setdefault('OMP_NUM_THREADS', str(max(1, min(16, len(os.sched_getaffinity(0))))))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _library_switches_follow_the_environment(request):
    """The library caches its MSCL_* tuning switches (csrc/common.h, MsclTune).  A GPU test that flips one with
    monkeypatch.setenv calls lib.tune() itself; this autouse fixture is torn down AFTER monkeypatch has restored the
    environment and makes the library read it again, so a forced kernel path never leaks into the next test."""
    yield
    if request.node.get_closest_marker('gpu') is not None:
        from mscl_amd import lib
        if lib._lib is not None:
            lib.tune()
