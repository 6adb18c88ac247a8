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
