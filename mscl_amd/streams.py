"""Choosing side streams that really run beside the main stream (and beside the RCCL communicator).

HIP multiplexes streams onto a few hardware queues (4 by default) in creation order, and two streams that land on one
queue serialise: with a process group alive (its own stream plus RCCL's) a freshly created side stream aliased the main
stream and the three-chain overlap of the MSCL step was gone (13.2 instead of 9.4 ms per step; rocprofv3 kernel trace:
both streams on queue 4).  So a few candidates are created and those that a spin kernel shows to overlap with everything
chosen so far are kept; with a process group up, candidates that also overlap with a large all-reduce come first.

The probe's RESULT is a per-rank timing outcome and may differ between ranks.  The collectives it issues must not:
every rank calls `comm.agree_min` exactly once, and the number of `comm.timed` calls that follow depends only on the
agreed value.  (An earlier form skipped the communicator probe when its local candidate count was <= 1 and looped over
its local candidates: two ranks disagreeing by one candidate issued different collective sequences and would hang.)
The timing primitives are passed in, so the control flow is unit-tested on the CPU with scripted timings
(tests/test_host.py::test_stream_probe_collectives_do_not_depend_on_local_timings).
"""

REPEATS = 2          # timed repetitions per communicator measurement (fixed: part of the collective sequence)


def pick_side_streams(cand, n, spin, comm=None):
    """cand: candidate stream objects (opaque here); n: how many are wanted.
    spin(streams, cycles) -> seconds for a spin kernel of `cycles` on the main stream and on each of `streams` at once.
    comm: None, or an object with
        agree_min(v) -> int   MIN of v over all ranks (ONE collective),
        timed(streams, cycles) -> seconds for [barrier; spin on `streams`; one large all-reduce] (TWO collectives).
    Returns (streams in preference order, cut to n; report dict)."""
    cycles = 200000
    spin(list(cand), 1000)                                     # first use of a stream may set up its hardware queue
    spin([], cycles)
    t = spin([], cycles)
    cycles = int(cycles * max(1.0, 1.5e-3 / max(t, 1e-5)))    # ~1.5 ms per spin: far above launch latency
    one = min(spin([], cycles) for _ in range(2))
    chosen = []                                                # one representative per hardware queue other than main's
    for c in cand:
        if spin(chosen + [c], cycles) < 1.4 * one:
            chosen.append(c)
    report = dict(spin_ms=1e3 * one, queues_beside_main=len(chosen), beside_comm=None, wanted=n, candidates=len(cand),
                  probed_with_comm=0)
    if comm is not None:
        k = int(comm.agree_min(len(chosen)))                   # issued by every rank, whatever its local result
        if k > 1:
            # the communicator's stream sits on one of the queues too: a side stream sharing it would stall behind every
            # gradient bucket.  Same test, with a large all-reduce as the other party, on the first k local candidates.
            comm.timed([], 0)
            t_c = min(comm.timed([], 0) for _ in range(REPEATS))
            cyc = max(1000, int(cycles * t_c / one))            # spin about as long as the all-reduce
            t_s = min(spin([], cyc) for _ in range(2))
            free = []
            for c in chosen[:k]:
                if min(comm.timed([c], cyc) for _ in range(REPEATS)) < max(t_c, t_s) + 0.5 * min(t_c, t_s):
                    free.append(c)
            report['beside_comm'], report['probed_with_comm'] = len(free), k
            chosen = free + [c for c in chosen if all(c is not x for x in free)]
    rest = [c for c in cand if all(c is not x for x in chosen)]
    return (chosen + rest)[:n], report
