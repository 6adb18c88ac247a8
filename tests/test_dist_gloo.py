"""world_size-2 gloo tests on CPU for the data-parallel exchange steps: shuffle-BN
select/unselect, key all-gather + replicated queue write, gradient averaging."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from mscl_amd import parallel
        from oracle import mscl as om
        torch.manual_seed(7 + rank)
        b = 4
        x = torch.arange(b, dtype=torch.float32).view(b, 1) + 100 * rank           # sample ids
        # shuffle: every sample lands on exactly one rank, unshuffle restores the owner's order
        xs = parallel.shuffle_select(x, step=5, slot=0)
        allx = parallel.all_gather_cat(xs)
        assert sorted(allx.flatten().tolist()) == sorted((torch.arange(b).repeat(world) + 100 * torch.arange(world).repeat_interleave(b)).tolist())
        k = xs * 2.0                                                              # "encode"
        ku = parallel.unshuffle_select(k, step=5, slot=0)
        assert torch.equal(ku, x * 2.0)
        # gradient averaging over buckets
        g = torch.full((1000,), float(rank + 1))
        parallel.allreduce_mean_(g, bucket_elems=300)
        assert torch.allclose(g, torch.full((1000,), (1 + world) * world / 2 / world))
        # bucketed reducer, fp32 and bf16 transport: the mean on every rank, identical across ranks
        # ... and with the explicit reduce-scatter + all-gather pair in place of the all-reduce (`mscl_grad_rs_ag`; bucket
        # sizes that do not divide by the world size exercise its padding)
        for transport, tol, coll in (('fp32', 0.0, 'all_reduce'), ('bf16', 2 ** -7, 'all_reduce'), ('fp32', 0.0, 'rs_ag'),
                                     ('bf16', 2 ** -7, 'rs_ag')):
            flat = torch.linspace(-3.0, 5.0, 1001)[:1000] * (rank + 1)
            want = torch.linspace(-3.0, 5.0, 1001)[:1000] * (1 + world) / 2
            red = parallel.GradReducer(flat, [(601, 1000), (0, 601)], need=[1, 2], transport=transport, collective=coll)
            red.bucket_done(0); red.bucket_done(1); red.bucket_done(1)           # the second bucket needs two trigger calls
            red.finish()
            assert torch.allclose(flat, want, rtol=tol, atol=1e-6 + tol * 0.01), (transport, coll, float((flat - want).abs().max()))
            both = parallel.all_gather_cat(flat[None])
            assert torch.equal(both[0], both[1]), (transport, coll)
            red.bucket_done(0); red.bucket_done(1); red.bucket_done(1); red.finish()      # a second step reuses the padded scratch
        # bench.py's "rccl" object (what a driver verifies an N-rank line with): its shape under this 2-rank group
        import sys
        import types
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        six = parallel.GradReducer(torch.zeros(6000), [(i * 1000, (i + 1) * 1000) for i in range(6)])
        info = bench.rccl_probe(torch.device('cpu'), world, types.SimpleNamespace(reducer=six))
        assert info['backend'] == 'gloo' and info['world_size'] == world and info['ranks_seen'] == list(range(world))
        assert info['allreduce_busbw_GBps'] > 0 and info['allreduce_100MB_ms'] > 0
        assert info['grad_collective'] == 'all_reduce' and info['grad_transport'] == 'fp32' and len(info['grad_buckets_MB']) == 6
        assert isinstance(info['exposed_wire_ms_model'], float)
        import json as _json
        _json.dumps(info)
        # the stand-alone form: the sum of a bucket, in place
        seg = torch.arange(7, dtype=torch.float32) * (rank + 1)
        parallel.grad_rs_ag(seg)
        assert torch.equal(seg, torch.arange(7, dtype=torch.float32) * (1 + world) * world / 2)
        # replicated queue: the oracle's enqueue under 2 ranks keeps queue/ptr/count identical everywhere
        rec = om.MoCoV2('flow', 128, K=16, max_iters=100)
        with torch.no_grad():
            rec.queue.copy_(torch.zeros_like(rec.queue))
        keys = torch.nn.functional.normalize(torch.randn(b, 128), dim=1)
        rec.dequeue_and_enqueue(keys)
        gathered = parallel.all_gather_cat(keys)
        assert torch.equal(rec.queue[:, :world * b], gathered.T) and int(rec.queue_ptr) == world * b
        assert rec.batch_size == world * b and torch.equal(rec.count[:world * b], torch.ones(world * b, dtype=torch.long))
        # all-to-all shuffle: same rows as the all-gather formulation, and the way back restores the owner's order
        for step in (0, 3):
            plan = parallel.ShufflePlan(world, b, rank, parallel.shuffle_perm(world * b, step, 1))
            got = parallel.exchange_rows(x, plan.send_order, plan.recv_order, plan.send_splits, plan.recv_splits)
            assert torch.equal(got, parallel.shuffle_select(x, step=step, slot=1))
            back = parallel.exchange_rows(got * 2.0, plan.back_send_order, plan.back_recv_order, plan.recv_splits, plan.send_splits)
            assert torch.equal(back, x * 2.0)
        # keys of several blocks come back in ONE all-gather: own rows in the owner's order + the global order for the queue
        perms = [parallel.shuffle_perm(world * b, 9, slot) for slot in range(3)]
        inv = torch.stack([torch.argsort(p) for p in perms])
        mine = [(torch.arange(b, dtype=torch.float32).view(b, 1) + 100 * rank + 1000 * i).repeat(1, 3) for i in range(3)]
        enc = [parallel.all_gather_cat(m).index_select(0, p.view(world, b)[rank]) * 2.0 for m, p in zip(mine, perms)]
        full, own = parallel.gather_unshuffle(enc, inv)
        for i in range(3):
            assert torch.equal(own[i], mine[i] * 2.0)
            assert torch.equal(full[i], parallel.all_gather_cat(mine[i]) * 2.0)
        state = torch.cat([rec.queue.flatten(), rec.count.float(), rec.queue_ptr.float()])
        states = parallel.all_gather_cat(state[None])
        assert torch.equal(states[0], states[1])
        q.put((rank, 'ok'))
    except Exception as e:      # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r, msg in res:
        assert msg == 'ok', f'rank {r}: {msg}'
