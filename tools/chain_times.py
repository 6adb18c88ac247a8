"""Solo duration of each chain of the MSCL step, each replayed from its own HIP graph (so launch overhead is the graph's, as
in the whole-step graph), next to the whole-step graph on three streams and on one stream.  Tells which chain bounds the
forward phase and the backward phase of the three-stream step (rocprofv3's tracer serialises the streams, so a trace
cannot).  usage: python tools/chain_times.py [--iters 20]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mscl_amd import ClipSGD, Config, build_model, kernels as K      # noqa: E402
from mscl_amd.fill import fill_module                                  # noqa: E402
from mscl_amd.nn import pool                                           # noqa: E402
from mscl_amd.recognizers import KeyGraph, QueryGraph                  # noqa: E402
from mscl_amd.synthetic import synthetic_batch                         # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--no-step', action='store_true')
    ap.add_argument('--only', default=None, help="'rgbq': replay only the RGB query forward + backward graphs (for a rocprofv3 kernel trace of that chain)")
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
    cfg.model.sup_head.t = 8
    model = build_model(cfg.model)
    fill_module(model)
    model.materialize(dev).train()
    opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
    batch = synthetic_batch(8, 16, 112, 112, 0, 0, device=dev)
    rec, recf, aug = model.recognizer, model.recognizer_flow, model.aug_gpu
    ids = model.sup_head.mlvl_ids
    out = {}

    def eager():
        o = model.train_step(batch, sync_logs=False)
        opt.zero_grad()
        o['loss'].backward()
        opt.step()
    for _ in range(2):
        eager()
    torch.cuda.synchronize()
    sc = model._scal.dev
    im_q, im_k = batch['imgs']
    fq, fk = batch['flow_imgs']
    Th = fq.shape[2] // 2
    x_rgb = aug.pack_rgb(im_q, None)
    x_flow = aug.pack_flow(fq, 0, Th, None)
    anchor = torch.zeros(1, device=dev, requires_grad=True)

    if a.only == 'rgbq':
        def rgb_body0(x):
            q, maps = rec.encode_q(x, levels=(ids[0],))
            m = maps[ids[0]]
            return q, pool(m, m.shape[0] * m.shape[1], m.shape[2] * m.shape[3]), tuple(m.shape)
        qg = QueryGraph(warmup=0)
        qg.capture(rgb_body0, x_rgb, rec.encoder_q.stem)
        print('rgb query fwd %.3f ms, bwd %.3f ms' % (timed(qg.fwd.replay, a.iters), timed(qg.bwd.replay, a.iters)))
        return
    # key chains (EMA + forward, no gradient)
    for name, r, x, m in (('rgb key (EMA+fwd)', rec, x_rgb, sc[0:1]), ('flow key, one pass (EMA+fwd)', recf, x_flow, sc[1:2])):
        g = KeyGraph(warmup=0)
        g.run(r, x, m)
        assert g.graph is not None, getattr(g, 'error', None)
        out[name] = timed(g.graph.replay, a.iters)

    def rgb_body(x):
        q, maps = rec.encode_q(x, levels=(ids[0],))
        m = maps[ids[0]]
        return q, pool(m, m.shape[0] * m.shape[1], m.shape[2] * m.shape[3]), tuple(m.shape)
    for name, body, x, trig in (('rgb query', rgb_body, x_rgb, rec.encoder_q.stem), ('flow query, one pass', model._flow_query_body, x_flow, recf.encoder_q.stem)):
        qg = QueryGraph(warmup=0)
        qg.capture(body, x, trig)
        out[name + ' fwd'] = timed(qg.fwd.replay, a.iters)
        out[name + ' bwd'] = timed(qg.bwd.replay, a.iters)

    # where in backward each gradient bucket's all-reduce would start (GradReducer.on_fire): events in two eager three-stream steps,
    # as fractions of that backward, scaled to the graph-replayed RGB backward above (eager launches are host-paced).
    # bench.py's ring model (rccl.exposed_wire_ms_model) reads this line from the newest profiles/r0N_chain_times.txt.
    fired = {}
    def on_fire(i):
        if i not in fired:
            e = torch.cuda.Event(enable_timing=True); e.record(); fired[i] = e
    fracs = None
    for rep in range(2):
        fired.clear()
        o = model.train_step(batch, sync_logs=False)
        opt.zero_grad()
        model.reducer.on_fire = on_fire
        model.reducer.hits = [0] * len(model.reducer.ranges)
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        o['loss'].backward()
        model.sync_streams()
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        model.reducer.on_fire = None
        opt.step()
        torch.cuda.synchronize()
        total = e0.elapsed_time(e1)
        fracs = [e0.elapsed_time(fired[i]) / total if i in fired else 1.0 for i in range(len(model.reducer.ranges))]
    bwd_ms = out['rgb query bwd']
    fire_line = 'bucket fire times into backward (ms) [layer4, layer3, layer2, stem+layer1, neck+heads, flow]: ' + \
        ' '.join('%.3f' % (f * bwd_ms) for f in fracs) + '; backward %.3f' % bwd_ms

    # optimizer + EMA-free tail
    def tail():
        opt.zero_grad()
        opt.step()
    out['zero_grad + clip + SGD + shadow refresh (eager)'] = timed(tail, a.iters)

    if not a.no_step:
        from mscl_amd.graph import GraphedStep
        for streams in ('3', '1'):
            model.two_streams = streams != '1'
            gs = GraphedStep(model, opt, batch, warmup=1)
            out[f'whole step, graph, {streams} stream(s)'] = timed(lambda: gs.step(batch), a.iters)
    for k, v in out.items():
        print(f'{k:55s} {v:8.3f} ms', flush=True)
    print(fire_line, flush=True)
    f = out
    print('sum of chains (rgb q f+b, rgb k, 2 flow q f+b, 2 flow k): %.3f ms' % (
        f['rgb query fwd'] + f['rgb query bwd'] + f['rgb key (EMA+fwd)'] + 2 * (f['flow query, one pass fwd'] + f['flow query, one pass bwd'])
        + 2 * f['flow key, one pass (EMA+fwd)']))
    print('flow stream forward phase (2 q fwd + 2 k): %.3f ms; rgb query fwd %.3f ms; flow bwd (2 passes) %.3f ms; rgb bwd %.3f ms' % (
        2 * f['flow query, one pass fwd'] + 2 * f['flow key, one pass (EMA+fwd)'], f['rgb query fwd'], 2 * f['flow query, one pass bwd'], f['rgb query bwd']))


if __name__ == '__main__':
    main()
