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


def _poison(dev, gib):
    """MSCL_TEST_POISON=<GiB>: fill that much device memory with NaN bit patterns, in block sizes of both pools of the caching
    allocator, and hand it back -- half to the allocator's cache (every later torch.empty is carved out of it), half to the driver
    (graph-private pools and the library's own allocations).  A kernel that reads a workspace element nobody wrote then sees NaN
    instead of whatever a fresh box happens to hold (a probe for reads of uninitialised memory; off by default)."""
    import torch
    held = []
    big = max(1, int(gib))
    for _ in range(big):
        held.append(torch.full((1 << 28,), float('nan'), device=dev))                       # 1 GiB blocks (large pool)
    for _ in range(2048):
        held.append(torch.full((1 << 17,), float('nan'), device=dev))                       # 512 KiB blocks (small pool)
    torch.cuda.synchronize()
    del held[::2]
    torch.cuda.empty_cache()
    del held
    torch.cuda.synchronize()


def _jitter(cycles):
    """MSCL_TEST_JITTER=<cycles>: every `with torch.cuda.stream(s):` region starts with a spin kernel of that many GPU cycles on s,
    so whatever a side stream produces arrives LATE -- a consumer on another stream that lacks a wait then reads stale data every time
    instead of once in a blue moon (a probe for missing cross-stream edges; off by default).  A negative count delays the stream
    that opens the region instead."""
    import torch
    enter = torch.cuda.StreamContext.__enter__

    def late(self):
        if cycles < 0 and self.stream is not None:         # negative: the stream that forks is the late one
            torch.cuda._sleep(-cycles)
        r = enter(self)
        if cycles > 0 and self.stream is not None:
            torch.cuda._sleep(cycles)
        return r
    torch.cuda.StreamContext.__enter__ = late


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    d = torch.device('cuda:0')
    if os.environ.get('MSCL_TEST_POISON'):
        _poison(d, float(os.environ['MSCL_TEST_POISON']))
    if os.environ.get('MSCL_TEST_JITTER'):
        _jitter(int(os.environ['MSCL_TEST_JITTER']))
    return d


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
