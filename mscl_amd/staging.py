"""Per-step host -> device words (EMA momenta, learning rate, shuffle-BN index rows) without a host/device race.

The step never synchronises the host (logs stay on the device, collectives wait stream-side, graph replay is
asynchronous), so the host runs several steps ahead of the GPU.  A pinned word that is rewritten every step while an
asynchronous H2D copy of it is still queued hands step n the values of step n+k: the wrong momentum / learning rate, or
-- for the shuffle-BN index rows -- rows paired with the wrong split sizes, i.e. keys assigned to the wrong samples.

`StagingRing` owns a small ring of pinned slots.  `push()` picks the next slot, waits for the event recorded after that
slot's previous copy (normally long complete), fills it, queues the copy into the ONE device tensor the kernels read, and
records the slot's event.  The device tensor is overwritten in stream order, after every kernel of the previous step
that read it.  `push()` must run outside a HIP-graph capture: a captured memcpy node keeps reading one host address on
every replay, which is the race this class exists to remove -- captured steps upload before `graph.replay()` instead.
"""
import torch


class StagingRing:
    def __init__(self, shape, dtype, device, slots=4):
        self.dev = torch.zeros(shape, dtype=dtype, device=device)
        self._host = [torch.zeros(shape, dtype=dtype).pin_memory() for _ in range(slots)]
        self._events = [None] * slots
        self._next = 0
        self.pushes = 0

    def slot(self):
        """the pinned tensor the next push() will send (fill it, then call push()); waits until its last copy is done"""
        i = self._next
        ev = self._events[i]
        if ev is not None:
            ev.synchronize()
            self._events[i] = None
        return self._host[i]

    def push(self, values=None):
        """copy the current slot (after `values`, if given, were written into it) to the device tensor on the current stream"""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('StagingRing.push() inside a graph capture would bake one host address into the graph')
        h = self.slot()
        if values is not None:
            h.copy_(values)
        self.dev.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[self._next] = ev
        self._next = (self._next + 1) % len(self._host)
        self.pushes += 1
        return self.dev
