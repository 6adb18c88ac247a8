"""Data-parallel pieces: one process per GPU, torch.distributed (backend 'nccl' = RCCL over xGMI on ROCm;
'gloo' in the CPU tests).  The path shards by samples; the exchange steps are
  * shuffle-BN of key clips (recognizers/moco.py:146-191): all ranks derive the SAME permutation from a
    counter-based seed (no rank-0 randperm + broadcast), gather the key clips and keep their slice;
  * the negative-key all-gather before the queue write (moco.py:426), so every replica's queue stays
    bit-identical;
  * the gradient all-reduce of the flat fp32 arena, in a few large buckets sized for the 7-link xGMI mesh.
All helpers are device-agnostic and are exercised on CPU tensors with 2-rank gloo in tests/.
"""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


def single():
    """True when no exchange step is needed.  MSCL_FORCE_DIST=1 makes an initialised one-rank group run every collective
    anyway: the only way to drive the RCCL code paths (all-to-all, all-gather, async AVG all-reduce) on a one-GPU box."""
    if not is_dist():
        return True
    return dist.get_world_size() == 1 and os.environ.get('MSCL_FORCE_DIST') != '1'


def settle_before_capture():
    """Call before a HIP-graph capture when an RCCL process group is alive.  ProcessGroupNCCL's watchdog thread polls the
    completion events of recent collectives every 100 ms; if one of those events belongs to a stream that has meanwhile
    joined a capture, the query fails ("event last recorded in a capturing stream") and the watchdog aborts the process.
    After a device synchronize every collective is complete, and two watchdog periods later all of them are retired."""
    if is_dist() and dist.get_backend() == 'nccl':
        import time
        torch.cuda.synchronize()
        time.sleep(0.25)


@torch.no_grad()
def all_gather_cat(t):
    """ref: recognizers/moco.py:558-568 (concat_all_gather); identity for a single replica."""
    if single():
        return t
    t = t.contiguous()
    out = torch.empty((world_size() * t.shape[0], *t.shape[1:]), dtype=t.dtype, device=t.device)
    if dist.get_backend() == 'nccl':
        dist.all_gather_into_tensor(out, t)
    else:
        dist.all_gather(list(out.chunk(world_size())), t)
    return out


def balanced_world(n, world=None):
    """the world size for which shuffle_perm deals a BALANCED permutation of n = W * B rows (0: none): W ranks, B a multiple of W"""
    W = world if world is not None else (dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1)
    return W if (W >= 1 and n % W == 0 and (n // W) % W == 0) else 0


def shuffle_perm(n, step, slot, seed=20221, world=None):
    """permutation of range(n) shared by all ranks: CPU generator seeded from (seed, step, slot).

    Round 6: when the per-rank batch B = n / W is a multiple of the world size W (the shipped config: 32 clips on 4 or 8 GPUs) the
    permutation is dealt BALANCED: every rank encodes exactly B / W rows of every owner, in random order -- each owner shuffles its
    rows and deals them round-robin to the encoders, each encoder shuffles what it received.  Still a permutation of the global
    batch whose every encoder batch mixes all replicas (what shuffle-BN is for: recognizers/moco.py:146-172 draws an unconstrained
    torch.randperm; the shared seed was a documented deviation already), and the all-to-all that carries it has EQUAL, CONSTANT
    split sizes: it can be captured into the whole-step HIP graph, where the unconstrained form had to fall back to an all-gather
    of W times the rows.  world = None: the initialised process group's size (1 without one: plain randperm, as before)."""
    g = torch.Generator().manual_seed(seed + 7919 * step + 104729 * slot)
    W = balanced_world(n, world)
    if W <= 1:
        return torch.randperm(n, generator=g)
    B = n // W
    k = B // W
    dealt = [torch.randperm(B, generator=g) + s * B for s in range(W)]              # owner s: its rows, shuffled
    rows = []
    for r in range(W):                                                                   # encoder r: k rows of every owner, shuffled
        got = torch.cat([dealt[s][r * k:(r + 1) * k] for s in range(W)])
        rows.append(got[torch.randperm(B, generator=g)])
    return torch.cat(rows)


@torch.no_grad()
def shuffle_select(x, step, slot):
    """rows of the all-gathered batch this rank encodes with the key encoder (moco.py:146-172)."""
    W = world_size()
    if W == 1:
        return x          # a within-batch permutation does not change per-GPU BN statistics
    b = x.shape[0]
    perm = shuffle_perm(W * b, step, slot)
    idx = perm.view(W, b)[rank()].to(x.device)
    return all_gather_cat(x).index_select(0, idx)


@torch.no_grad()
def unshuffle_select(k, step, slot):
    """undo shuffle_select on the encoded keys (moco.py:174-191)."""
    W = world_size()
    if W == 1:
        return k
    b = k.shape[0]
    inv = torch.argsort(shuffle_perm(W * b, step, slot))
    idx = inv.view(W, b)[rank()].to(k.device)
    return all_gather_cat(k).index_select(0, idx)


class ShufflePlan:
    """Shuffle-BN as two all-to-alls instead of two all-gathers (moco.py:146-191 gathers the whole W*B batch on every
    rank and keeps B rows of it: W times the traffic that is needed; over point-to-point xGMI links the key clips are
    the largest exchange of the step).  Everything is derived on the host from the permutation all ranks share.

    forward:  rank r must encode global rows perm[r] (in that order); every row travels once, owner -> encoder.
    backward: the encoded keys travel back, encoder -> owner, and land at their local index.
    """

    def __init__(self, W, B, rank, perm):
        perm = perm.view(W, B)
        flat_pos = torch.argsort(perm.flatten())            # flat_pos[g] = r * B + p : who encodes global row g, and where
        mine = flat_pos[rank * B:(rank + 1) * B]             # ... for the rows this rank owns
        self.send_order = torch.argsort(mine)                # own rows sorted by (destination, position there)
        self.send_splits = torch.bincount(mine // B, minlength=W).tolist()
        src = perm[rank] // B                                # owner of each row this rank encodes
        self.recv_splits = torch.bincount(src, minlength=W).tolist()
        arrival = torch.argsort(src, stable=True)            # arrival[j] = position p of the j-th received row
        self.recv_order = torch.argsort(arrival)             # received buffer -> position order
        # way back: keys sorted by global id (= by owner, then local index); owners receive, per encoder, ascending ids
        self.back_send_order = torch.argsort(perm[rank])
        ids = torch.cat([torch.sort(perm[r][(perm[r] // B) == rank]).values for r in range(W)])
        self.back_recv_order = torch.argsort(ids)            # ids is a permutation of this rank's own global ids

    def index_rows(self):
        """the four index vectors, stacked (4, B) int64, for one upload"""
        return torch.stack([self.send_order, self.recv_order, self.back_send_order, self.back_recv_order])


@torch.no_grad()
def exchange_rows(x, send_order, recv_order, send_splits, recv_splits):
    """out[p] = the p-th row this rank must hold after the exchange; x rows are this rank's."""
    xs = x.index_select(0, send_order).contiguous()
    if dist.get_backend() == 'gloo' and xs.is_cuda:
        # gloo has no all-to-all for device tensors: same result from all-gathers (test path: two ranks on one GPU)
        W, r = world_size(), rank()
        allx = all_gather_cat(xs).view(W, xs.shape[0], *xs.shape[1:])
        sp = all_gather_cat(torch.tensor([send_splits], device=xs.device)).tolist()
        parts = []
        for src in range(W):
            a = sum(sp[src][:r])
            parts.append(allx[src, a:a + sp[src][r]])
        out = torch.cat(parts)
    else:
        out = torch.empty_like(xs)
        dist.all_to_all_single(out, xs, output_split_sizes=recv_splits, input_split_sizes=send_splits)
    return out.index_select(0, recv_order)


@torch.no_grad()
def gather_unshuffle(keys, inv):
    """The way back of shuffle-BN for several key blocks at once (moco.py:174-191 + the gather of moco.py:426).
    keys: list of (B, dim) blocks this rank ENCODED (rows in shuffled order); inv: (len(keys), W*B) int64, inv[i] =
    argsort of block i's permutation.  One all-gather moves all blocks; returns (full, own): full[i] = every replica's
    keys of block i in global sample order (what the queue enqueues), own[i] = this rank's B rows of it."""
    B, dim, r = keys[0].shape[0], keys[0].shape[1], rank()
    allk = all_gather_cat(torch.cat(keys, dim=1))
    full = [allk[:, i * dim:(i + 1) * dim].index_select(0, inv[i]) for i in range(len(keys))]
    return full, [f[r * B:(r + 1) * B] for f in full]


def bucket_plan(total, bucket_elems):
    """[(start, end)] covering [0, total) in buckets of at most bucket_elems (last one short)."""
    out, a = [], 0
    while a < total:
        b = min(total, a + bucket_elems)
        out.append((a, b))
        a = b
    return out


def _sum_then_scale():
    """gloo has no ReduceOp.AVG (CPU tests, and the 1-GPU 2-rank GPU test); RCCL does"""
    return dist.get_backend() != 'nccl'


@torch.no_grad()
def grad_rs_ag(seg, pad=None, async_op=False):
    """`mscl_grad_rs_ag` of SURVEY.md section 8(b) / 5.8: the gradient SUM of one arena bucket as an explicit reduce-scatter +
    all-gather pair over the process group instead of one all-reduce (reference: the DDP reducer wired at
    mmaction/apis/train.py:84-88).  Each rank ends up owning the sum of its 1/W share (reduce-scatter), then every rank collects
    all shares (all-gather): on a point-to-point xGMI mesh both halves can run as direct exchanges between every pair of GPUs
    over all 7 links at once, where a ring all-reduce is paced by one link.  The bucket is padded to a multiple of W in a
    scratch buffer (`pad`: reused between steps).  Selected with MSCL_GRAD_COLLECTIVE=rs_ag (GradReducer); the default stays
    all_reduce until an 8-GPU node has measured the two against each other.  Returns (handle | None, scratch): the caller copies
    scratch[:n] back into `seg` after waiting (async) -- or gets `seg` updated in place (sync).
    gloo (CPU tests) has no reduce-scatter: W reduces, one per share, stand in for it."""
    W, r = world_size(), rank()
    n = seg.numel()
    share = (n + W - 1) // W
    if pad is None or pad.numel() < share * W or pad.dtype != seg.dtype or pad.device != seg.device:
        pad = torch.empty(share * W, dtype=seg.dtype, device=seg.device)
    buf = pad[:share * W]
    buf[:n].copy_(seg.reshape(-1))
    if share * W > n:
        buf[n:].zero_()
    mine = buf[r * share:(r + 1) * share]
    if dist.get_backend() == 'nccl':
        # both halves asynchronous: they run in order on the communicator's stream, and a synchronous call would make the
        # calling (backward) stream wait for the wire
        dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM, async_op=True)       # the own share of buf receives the sum
        h = dist.all_gather_into_tensor(buf, mine, async_op=True)
        if not async_op:
            h.wait()
    else:
        for d in range(W):
            dist.reduce(buf[d * share:(d + 1) * share], dst=d, op=dist.ReduceOp.SUM)
        h = dist.all_gather(list(buf.chunk(W)), mine.clone(), async_op=async_op)
    if not async_op:
        seg.reshape(-1).copy_(buf[:n])
        return None, pad
    return h, pad


class GradReducer:
    """Bucketed, overlapped mean all-reduce of the flat gradient arena.

    Buckets are contiguous arena ranges in the order backward finishes them (projection MLP + neck, layer4,
    layer3, layer2 .. stem; the flow trunk last).  `bucket_done(i)` is called from the backward of the module
    that completes a bucket: the all-reduce is issued at once (async) and runs on the communicator's stream
    while backward continues with the earlier layers; `finish()` waits for all of them before the optimizer.
    Layer 4 alone is 100 MB of the 150 MB: it is reduced under the whole layer3..stem backward."""

    def __init__(self, flat, ranges, need=None, transport=None, collective=None):
        """transport: 'fp32' (default: the reference's DDP averages fp32 gradients) or 'bf16' (MSCL_GRAD_TRANSPORT=bf16): each
        bucket travels as bf16 -- half the bytes over the xGMI links, 75 instead of 150 MB per step -- and is widened and
        averaged on arrival.  Every rank receives the same sums, so the replicas stay bit-identical; the values differ from the
        fp32 mean by one bf16 rounding of each rank's gradient and of the sum (relative 2^-8 per element, direction cosine
        ~0.99999).  Opt-in: unmeasured on a multi-GPU node (SURVEY.md section 5.8 names it as the transport to try)."""
        self.flat, self.ranges = flat, list(ranges)
        self.need = list(need) if need is not None else [1] * len(self.ranges)   # trigger calls that complete a bucket
        self.works, self.launched, self.hits = [], set(), [0] * len(self.ranges)
        self.counters = []                          # nn.BucketCounter objects that fire buckets of this reducer (reset in finish())
        self.transport = transport or os.environ.get('MSCL_GRAD_TRANSPORT', 'fp32')
        if self.transport not in ('fp32', 'bf16'):
            raise ValueError(f'gradient transport {self.transport!r}: fp32 or bf16')
        # all_reduce (default) | rs_ag: the explicit reduce-scatter + all-gather pair (grad_rs_ag), to A/B on a multi-GPU node
        self.collective = collective or os.environ.get('MSCL_GRAD_COLLECTIVE', 'all_reduce')
        if self.collective not in ('all_reduce', 'rs_ag'):
            raise ValueError(f'gradient collective {self.collective!r}: all_reduce or rs_ag')
        self._pads = {}
        # measurement aid (bench.py, "exposed_wire_ms_measured"): buckets are not sent at all -- the replicas then drift apart,
        # so only ever for a few timed steps at the very end of a run
        self.skip = False
        # measurement aid (tools/chain_times.py): called with the bucket index whenever a trigger completes a bucket, also at
        # world size 1 -- where in backward each bucket's all-reduce would start
        self.on_fire = None

    def bucket_done(self, i, force=False):
        if self.on_fire is not None and not force:
            self.on_fire(i)
        if single() or i in self.launched:
            return
        self.hits[i] += 1
        if self.hits[i] < self.need[i] and not force:
            return                                  # e.g. the flow trunk is traversed twice per step
        self.launched.add(i)
        if self.skip:
            return
        a, b = self.ranges[i]
        seg = self.flat[a:b]
        if self.collective == 'rs_ag':
            src = seg.to(torch.bfloat16) if self.transport == 'bf16' else seg
            h, self._pads[i] = grad_rs_ag(src, self._pads.get(i), async_op=True)
            n = seg.numel()
            self.works.append((h, seg, self._pads[i][:n]))         # finish(): seg <- sums / W
        elif self.transport == 'bf16':
            buf = seg.to(torch.bfloat16)
            self.works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), seg, buf))
        elif _sum_then_scale():
            self.works.append((dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True), seg, None))
        else:
            self.works.append((dist.all_reduce(seg, op=dist.ReduceOp.AVG, async_op=True), None, None))

    def finish(self):
        if single():
            return
        for i in range(len(self.ranges)):          # anything a trigger missed (e.g. unused branches)
            self.bucket_done(i, force=True)
        for w, seg, buf in self.works:
            w.wait()
            if buf is not None:
                seg.copy_(buf)
            if seg is not None:
                seg.div_(world_size())
        self.works, self.launched, self.hits = [], set(), [0] * len(self.ranges)
        for c in self.counters:
            c.pending = 0


def exposed_wire_ms(bucket_bytes, fire_ms, backward_ms, world, link_gbps=153.0):
    """The rule behind the gradient transport and the bucket split, checkable without a multi-GPU node:
    buckets travel one after the other on the communicator's stream, a ring all-reduce moves 2 (W-1)/W of a bucket's bytes over
    one xGMI link (~153 GB/s), bucket i starts when its trigger fires (`fire_ms` into backward, from the
    measured chain) or when its predecessor is done.  Returns the milliseconds of wire time left after backward ends -- what the
    step actually pays.  fp32 is kept while this stays under ~1 % of the step; bf16 transport halves `bucket_bytes`."""
    t = 0.0
    for nbytes, fire in sorted(zip(bucket_bytes, fire_ms), key=lambda p: p[1]):
        wire = 2.0 * (world - 1) / world * nbytes / (link_gbps * 1e9) * 1e3
        t = max(t, fire) + wire
    return max(0.0, t - backward_ms)


@torch.no_grad()
def allreduce_mean_(flat, bucket_elems=8 << 20):
    """average a flat gradient buffer over the replicas, bucket by bucket (async, waited at the end).
    149.8 MB of fp32 gradients -> 5 buckets of 32 MiB: large enough to run every xGMI link at rate,
    small enough that the first bucket's reduction overlaps the rest being issued."""
    W = world_size()
    if W == 1:
        return flat
    works = []
    for a, b in bucket_plan(flat.numel(), bucket_elems):
        seg = flat[a:b]
        if not _sum_then_scale():
            works.append(dist.all_reduce(seg, op=dist.ReduceOp.AVG, async_op=True))
        else:
            works.append((dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True), seg))
    for w in works:
        if isinstance(w, tuple):
            w[0].wait(); w[1].div_(W)
        else:
            w.wait()
    return flat
