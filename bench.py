"""MSCL hot-path benchmark: clip-pairs / second / node for the full MSCLWithAug training step
(dual-stream R3D-18 + r2d_18, MoCo queues, cross-modal InfoNCE, LMCL, backward, grad-clip + SGD).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic batch of 8 clip-pairs per GPU (16 frames, 112x112,
RGB q/k + visualised flow q/k as base||rotated), inputs resident in HBM, weights from the closed-form
fill.  Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      dominant kernel = the layer-1 3x3x3 64->64 conv (halo-resident MFMA kernel, MFMA-bound), timed with
                event pairs on its launch stream in two single-stream eager steps right after the timed region
  cpu_baseline  the oracle/ restatement ("port") timed on this box's host cores on a bounded sample
"""
import argparse
import glob
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
T_FRAMES, SIDE, BATCH = 16, 112, 8
# GFLOP (2 * MACs) of one step at B=8, T=16, 112^2, from SURVEY.md Appendix A-C's per-layer forward counts:
#   reference step (section 8d): 3302 = (RGB trunk 651.14 + neck 145.28 + flow trunk 2 x 14.51) x (3 on the query side + 1 on the
#   key side).  What this build EXECUTES is less: no stem input gradients (22.66 + 2 x 0.94), the key branch's pyramid is
#   skipped (145.28: k_mlvl has no reader), and levels 1-2 of the last PConv3D are not computed (13.18 forward; they never had
#   a backward: their gradient is None in the reference too), nor, since round 5, level 2 of the first PConv3D, which only those
#   fed (1.39 forward: P1 on the 7 x 7 level + the strided P2 into it).  step_frac prices the executed work only.
_TRUNK, _NECK, _NECK_DEAD, _FLOW, _STEM_DG, _FSTEM_DG = 651.14, 145.28, 13.18 + 1.387, 14.51, 22.659, 0.944
STEP_GFLOP_REFERENCE = 4 * (_TRUNK + _NECK + 2 * _FLOW)
STEP_GFLOP_EXECUTED = (3 * _TRUNK - _STEM_DG) + 3 * (_NECK - _NECK_DEAD) + 2 * (3 * _FLOW - _FSTEM_DG) + 2 * _FLOW + _TRUNK


def stage_table(events):
    """per conv shape class of the RGB trunk: algorithmic GFLOP, summed event time and TFLOP/s for forward / input gradient /
    weight gradient launches of the two single-stream eager steps (events: kernels.PROFILE['events'])"""
    names = {(64, 64, 56): 'layer1 64->64 @16x56x56', (64, 128, 28): 'layer2.0.conv1 64->128 s2', (128, 128, 28): '128->128 @8x28x28 (layer2, SEPC)',
             (128, 256, 14): 'layer3.0.conv1 128->256 s2', (256, 256, 14): 'layer3 256->256 @4x14x14',
             (256, 512, 7): 'layer4.0.conv1 256->512 s2', (512, 512, 7): 'layer4 512->512 @2x7x7'}
    acc = {}
    for ev in events:
        mode, d, e0, e1 = ev[:4]
        share = ev[4] if len(ev) > 4 else 1.0          # a grouped launch (nn.WGradQueue): its time split over the layers by FLOPs
        N, T, H, W, C, To, Ho, Wo, K, kT, kH, kW, sT, sH, sW = d
        if C == 8:                                  # 3-channel stems, padded (and W-paired: kW 4 stands for 7 taps)
            name, cin, taps = ('stem rgb' if kT == 3 else 'stem flow'), 3, kT * kH * 7
        elif kT == 3 and kH == 3 and (C, K, Ho) in names:
            name, cin, taps = names[(C, K, Ho)], C, 27
        else:
            name, cin, taps = ('flow trunk' if C <= 64 and kT == 1 else 'neck / shortcuts (1x1x1, 1x3x3, small 3x3x3)'), C, kT * kH * kW
        gf = 2.0 * N * To * Ho * Wo * K * taps * cin * 1e-9
        a = acc.setdefault(name, {}).setdefault(mode, [0.0, 0.0, 0, 0])
        a[0] += gf; a[1] += e0.elapsed_time(e1) * share; a[2] += 1; a[3] += 1 if len(ev) > 4 else 0
    out = {}
    for name, modes in acc.items():
        out[name] = {m: dict({'launches': v[2], 'gflop': round(v[0], 2), 'ms': round(v[1], 4),
                              'tflops': round(v[0] / max(v[1], 1e-9), 1), 'frac': round(v[0] / max(v[1], 1e-9) / PEAK_BF16_TFLOPS, 4)},
                             # layers that went out in a grouped launch (nn.WGradQueue): their time is a FLOP-share split of ONE event pair
                             # around the whole group -- an approximation, not a per-layer measurement
                             **({'flop_share_of_grouped_launch': v[3]} if v[3] else {}))
                     for m, v in modes.items()}
    return out


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-batch', type=int, default=8)
    p.add_argument('--no-graph', action='store_true', help='eager launches instead of one captured HIP graph per step')
    p.add_argument('--deterministic', action='store_true',
                   help='variant: the library\'s deterministic mode (fixed-order sums instead of float atomics; the reference\'s --deterministic)')
    p.add_argument('--no-variants', action='store_true', help='skip the deterministic-mode leg that follows the headline region')
    p.add_argument('--stochastic-aug', action='store_true',
                   help='variant (not the BASELINE.json workload): random flip / colour jitter / grayscale / blur per step')
    return p.parse_args()


HOST_THREADS = 0        # torch's default intra-op thread count, recorded by main() before it is lowered for the GPU part


def cpu_baseline(cpu_batch):
    """oracle/ (pure-PyTorch fp32 restatement of the reference step) on the host cores.  The box shows 256 logical CPUs but a
    one-GPU job owns a share of them: with torch's default 128 threads the step ran 15.5 s, with 16 threads 1.3 s (B=4).  So the
    thread count is chosen by one timed step each at 8 / 16 / 32 threads (MSCL_CPU_THREADS overrides), then 3 steps are timed."""
    from mscl_amd.synthetic import synthetic_batch
    from oracle import fill as ofill, mscl as om
    orc = om.MSCLWithAug(num_frames=T_FRAMES)
    ofill.fill_module(orc)
    orc.train()
    opt = om.SGDClip(orc.parameters())

    def step(s):
        batch = synthetic_batch(cpu_batch, T_FRAMES, SIDE, SIDE, 0, s)
        t0 = time.perf_counter()
        out = orc.train_step(batch)
        opt.zero_grad()
        out['loss'].backward()
        opt.step()
        return time.perf_counter() - t0
    default_threads = torch.get_num_threads()
    avail = max(HOST_THREADS, default_threads) if os.environ.get('OMP_NUM_THREADS') is None else default_threads
    env = os.environ.get('MSCL_CPU_THREADS')
    cands = [int(env)] if env else sorted({min(n, avail) for n in (8, 16, 32)})
    torch.set_num_threads(cands[len(cands) // 2])
    step(0)                                                    # warm-up (allocator, oneDNN primitives)
    trial = {}
    for n in cands:
        torch.set_num_threads(n)
        trial[n] = step(1)
    threads = min(trial, key=trial.get)
    torch.set_num_threads(threads)
    times = [step(2 + i) for i in range(3)]
    torch.set_num_threads(default_threads)
    mean = sum(times) / len(times)
    return dict(value=cpu_batch / mean, unit='clip-pairs/s', cores=threads, kind='port',
                sample=f'oracle MSCLWithAug step (fwd+bwd+clip+SGD), fp32, B={cpu_batch}, T={T_FRAMES}, {SIDE}x{SIDE}, '
                       f'1 warm-up, thread count picked from {cands} by one step each, then 3 timed steps '
                       f'(mean {mean:.2f} s on {threads} threads)')


def chain_fire_times(profiles=None):
    """(fire times [ms into backward] of the six gradient buckets, backward length, source) from the newest
    profiles/r0N_chain_times.txt that carries a 'bucket fire times' line (tools/chain_times.py); the round-3 constants otherwise"""
    import glob
    import re
    profiles = profiles or os.path.join(ROOT, 'profiles')
    for path in sorted(glob.glob(os.path.join(profiles, 'r[0-9][0-9]_chain_times.txt')), reverse=True):
        for ln in open(path):
            m = re.match(r'bucket fire times into backward \(ms\) \[[^\]]*\]: ([0-9. ]+); backward ([0-9.]+)', ln)
            if m:
                return [float(v) for v in m.group(1).split()], float(m.group(2)), os.path.basename(path)
    return [0.9, 1.25, 1.8, 3.78, 0.5, 1.7], 3.78, 'constants of profiles/r03_chain_times.txt'


def rccl_probe(dev, world, model):
    """What RCCL saw, so that a driver can verify an N-rank line from the line itself (world size > 1, or a one-rank group with
    MSCL_FORCE_DIST=1): backend, world size, the ranks an all-gather returned, the bus bandwidth of one 100-MB fp32 all-reduce
    timed before the warm-up (busbw = 2 (W-1)/W x bytes / time, the ring figure RCCL's own tests quote; 0 at W = 1), and the
    ring-model estimate of the wire time the step leaves exposed (parallel.exposed_wire_ms on the bucket sizes of THIS model and
    the fire times of profiles/r03_chain_times.txt)."""
    from mscl_amd import parallel
    sync = torch.cuda.synchronize if dev.type == 'cuda' else (lambda: None)      # (CPU + gloo: the 2-rank test of this function)
    out = {'backend': dist.get_backend(), 'world_size': dist.get_world_size()}
    seen = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(seen, torch.tensor([dist.get_rank()], dtype=torch.int64, device=dev))
    out['ranks_seen'] = [int(t) for t in seen]
    buf = torch.ones(25 << 20, dtype=torch.float32, device=dev)          # 100 MiB
    dist.all_reduce(buf)                                                   # warm-up: connections, proxy threads
    sync(); dist.barrier()
    t0 = time.perf_counter()
    dist.all_reduce(buf)
    sync()
    dt = time.perf_counter() - t0
    out['allreduce_100MB_ms'] = round(dt * 1e3, 3)
    out['allreduce_busbw_GBps'] = round(2.0 * (world - 1) / world * buf.numel() * 4 / dt / 1e9, 1) if world > 1 else 0.0
    red = getattr(model, 'reducer', None)
    if red is not None:
        nbytes = [(b - a) * 4 for a, b in red.ranges]
        out['grad_buckets_MB'] = [round(n / 1e6, 1) for n in nbytes]
        out['grad_collective'], out['grad_transport'] = red.collective, red.transport
        # fire times into backward (ms) of [layer 4, layer 3, layer 2, stem + layer 1, neck + heads, flow] and the backward's length,
        # measured on one GPU by tools/chain_times.py: read from the newest committed profiles/r0N_chain_times.txt
        fire, bwd, src = chain_fire_times()
        out['fire_times_source'] = src
        if len(nbytes) == len(fire):
            out['exposed_wire_ms_model'] = round(parallel.exposed_wire_ms(nbytes, fire, bwd, max(world, 2)), 4)
    return out


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if 'WORLD_SIZE' in os.environ or 'RANK' in os.environ:
            raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
        # called bare with --gpus N > 1: start the N ranks ourselves (one process per GPU over RCCL, as the driver's own launch line
        # does) as a CHILD of this process, before anything here has touched the GPU, and pass its output and exit code on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    # the GPU path needs no CPU parallelism; 128 OpenMP workers spin-waiting beside the launching thread only cost (a one-GPU job
    # owns ~16 cores of the box; torch.distributed.run already sets OMP_NUM_THREADS=1 for N > 1).  cpu_baseline picks its own count.
    global HOST_THREADS
    HOST_THREADS = torch.get_num_threads()
    torch.set_num_threads(max(1, min(8, HOST_THREADS)))
    forced = world == 1 and os.environ.get('MSCL_FORCE_DIST') == '1'    # diagnostic: run every collective on a 1-rank RCCL group
    if world > 1 or forced:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if forced:
            os.environ.setdefault('MASTER_PORT', '29531')
            dist.init_process_group('nccl', device_id=dev, rank=0, world_size=1)
        else:
            dist.init_process_group('nccl', device_id=dev)

    from mscl_amd import ClipSGD, Config, build_model, kernels
    from mscl_amd.fill import fill_module
    from mscl_amd.synthetic import synthetic_batch

    cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
    cfg.model.sup_head.t = T_FRAMES // 2            # the config derives it from num_frames (16 here, 8 as shipped)
    model = build_model(cfg.model)
    fill_module(model)
    model.materialize(dev).train()
    if args.deterministic:
        from mscl_amd import lib as _lib
        _lib.set_deterministic(True)
    if args.stochastic_aug:
        model.aug_gpu.stochastic = True
        model.aug_gpu.seed(1234 + rank)
    opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
    nbatch = 4
    batches = [synthetic_batch(BATCH, T_FRAMES, SIDE, SIDE, rank, s, device=dev) for s in range(nbatch)]
    torch.cuda.synchronize()

    def eager_step(i):
        out = model.train_step(batches[i % nbatch], sync_logs=False)
        opt.zero_grad()
        out['loss'].backward()
        opt.step()
        return out['loss']

    # World size > 1 launches eagerly unless MSCL_GRAPH_DP=1: capturing RCCL collectives issued from three streams
    # into one HIP graph could only be validated on a single-GPU box this round.
    graphed = None
    if not args.no_graph and (world == 1 or os.environ.get('MSCL_GRAPH_DP') == '1'):
        try:
            from mscl_amd.graph import GraphedStep
            graphed = GraphedStep(model, opt, batches[0], warmup=2)
        except Exception as e:      # noqa: BLE001 -- a capture failure must not lose the measurement
            print(f'[bench] HIP-graph capture failed ({type(e).__name__}: {e}); falling back to eager launches', file=sys.stderr)
            graphed = None

    def step(i):
        if graphed is not None:
            return graphed.step(batches[i % nbatch])[0]
        return eager_step(i)

    if graphed is None:
        # eager launches replay the key branches and the flow query passes from HIP sub-graphs, captured on the third
        # call; like the whole-step capture above this happens before the W warm-up steps, whatever W is
        for i in range(3):
            eager_step(i)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rccl = None
    if world > 1 or forced:
        try:
            rccl = rccl_probe(dev, world, model)
        except Exception as e:      # noqa: BLE001 -- a failed probe must not lose the measurement
            rccl = {'error': f'{type(e).__name__}: {e}'}
    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss_t = step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    # dominant-kernel timing: event pairs around the layer-1 conv launches of 2 eager steps issued right after the
    # timed region (same kernels, same data; the timed region itself is one graph launch per step)
    # (one stream for these two steps: an event pair on one stream would otherwise also time the other streams' kernels
    #  sharing the CUs -- 154 us instead of 131 us for this launch; the in-graph average is in profiles/*kernel_stats*)
    kernels.PROFILE = dict(events=[])
    streams_were, keyg_were, qg_were = model.two_streams, model.key_graphs, model.query_graphs
    model.two_streams = model.key_graphs = model.query_graphs = False   # plain launches on one stream: events inside a capture are not timeable
    for i in range(2):
        eager_step(i)
    torch.cuda.synchronize()
    model.two_streams, model.key_graphs, model.query_graphs = streams_were, keyg_were, qg_were
    prof = kernels.PROFILE
    kernels.PROFILE = None
    # -- world size > 1: the step with its gradient buckets NOT sent (a few steps at the very end: the replicas drift apart);
    # step time - that = the wire time the overlap leaves exposed, measured
    dt_nocoll = None
    if (world > 1 or forced) and getattr(model, 'reducer', None) is not None and graphed is None:
        try:
            model.reducer.skip = True
            for i in range(2):
                step(i)
            fence()
            t1 = time.perf_counter()
            for i in range(max(4, args.steps // 2)):
                step(i)
            fence()
            dt_nocoll = (time.perf_counter() - t1) / max(4, args.steps // 2)
        except Exception as e:      # noqa: BLE001
            print(f'[bench] no-collective leg failed ({type(e).__name__}: {e})', file=sys.stderr)
        finally:
            model.reducer.skip = False
    # -- variant: the library's deterministic mode (the reference's --deterministic), from a graph of its own, >= 10 replays after
    # the headline region.  One GPU only; never the headline value.
    det_variant = None
    if world == 1 and not forced and graphed is not None and not args.deterministic and not args.no_variants:
        try:
            from mscl_amd import lib as _lib
            from mscl_amd.graph import GraphedStep
            _lib.set_deterministic(True)
            gd = GraphedStep(model, opt, batches[0], warmup=2)
            for i in range(3):
                gd.step(batches[i % nbatch])
            torch.cuda.synchronize()
            nrep = max(10, args.steps)
            t1 = time.perf_counter()
            for i in range(nrep):
                ld = gd.step(batches[i % nbatch])[0]
            torch.cuda.synchronize()
            dtd = time.perf_counter() - t1
            det_variant = {'value': BATCH * nrep / dtd, 'unit': 'clip-pairs/s', 'ms_per_step': 1e3 * dtd / nrep, 'steps': nrep,
                           'final_loss': float(ld.detach()),
                           'what': 'mscl_set_deterministic(1): fixed-order sums instead of float atomics, one captured HIP graph per step'}
            del gd
        except Exception as e:      # noqa: BLE001
            det_variant = {'error': f'{type(e).__name__}: {e}'}
        finally:
            _lib.set_deterministic(False)
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    loss = float(loss_t.detach())
    if not (loss == loss):
        raise SystemExit('loss is NaN')

    if rank == 0:
        ms = [ev[2].elapsed_time(ev[3]) for ev in prof['events']
              if ev[0] == 'fwd' and (ev[1][4], ev[1][8], ev[1][9], ev[1][2]) == (64, 64, 3, SIDE // 2)]          # layer-1 3x3x3 64->64 forward launches
        avg_ms = sum(ms) / max(1, len(ms))
        flops = 2.0 * BATCH * T_FRAMES * (SIDE // 2) ** 2 * 64 * 27 * 64       # 88.8 GFLOP per launch
        achieved = flops / (avg_ms * 1e-3) / 1e12 if ms else 0.0
        # second kernel by total time: conv_pp_kernel<128> (layers 2-4, SEPC, FPN); priced on its largest shape, the 128 -> 128
        # 3x3x3 conv on (8,8,28,28,128) = 44.4 GFLOP per launch, forward launches by the same event pairs
        ms2 = [ev[2].elapsed_time(ev[3]) for ev in prof['events']
               if ev[0] == 'fwd' and (ev[1][4], ev[1][8], ev[1][9], ev[1][2]) == (128, 128, 3, SIDE // 4)]
        avg2 = sum(ms2) / max(1, len(ms2))
        flops2 = 2.0 * BATCH * (T_FRAMES // 2) * (SIDE // 4) ** 2 * 128 * 27 * 128
        ach2 = flops2 / (avg2 * 1e-3) / 1e12 if ms2 else 0.0
        traffic = None
        tname = None
        # PMC passes (FETCH_SIZE + WRITE_SIZE, separate runs: tools/evidence_r0N.sh), the newest committed round first; see the file
        for tp in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_traffic_layer1.json')), reverse=True):
            traffic = json.load(open(tp)).get('traffic_bytes_per_launch')
            tname = os.path.basename(tp)
            break
        step_s = dt / args.steps
        line = {
            'metric': 'clip-pairs/sec/node (R3D-18, 16x112^2, bs8/gpu)',
            'value': world * BATCH * args.steps / dt, 'unit': 'clip-pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'full MSCLWithAug step (dual-stream R3D-18 + r2d_18, MoCo queues K=65536, '
                                   'cross-modal InfoNCE, LMCL, backward, clip+SGD), mscl_r18 config with T=16',
                       'clip': f'{T_FRAMES}x{SIDE}x{SIDE}', 'batch_per_gpu': BATCH, 'global_batch': BATCH * world,
                       'parallelism': f'dp{world}' + (' (collectives forced on a 1-rank RCCL group)' if forced else ''), 'weights': 'closed-form fill, fp32 masters + bf16 shadows',
                       'launch': 'one captured HIP graph per step' if graphed is not None
                                 else 'eager launches, %d of %d collective-free branches replayed from HIP sub-graphs' % (
                                     sum(g.graph is not None for g in model._key_graph) + sum(g.fwd is not None for g in model.active_query_graphs()),
                                     len(model._key_graph) + len(model.active_query_graphs())),
                       **({'deterministic': 'fixed-order sums instead of float atomics (variant, not the headline line)'} if args.deterministic else {}),
                       'aug': 'stochastic flip+jitter+grayscale+blur (variant)' if args.stochastic_aug
                              else 'normalise only (BASELINE.json workload)'},
            'final_loss': loss,
            'roofline': {'bound': 'mfma', 'kernel': 'conv_halo64b_kernel fwd (+BN statistics), 3x3x3 64->64 on (8,16,56,56,64)',
                         'achieved': achieved, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_BF16_TFLOPS,
                         'launches_timed': len(ms), 'avg_launch_ms': avg_ms, 'traffic': traffic,
                         'traffic_unit': f'HBM bytes per launch (rocprofv3 PMC, profiles/{tname})',
                         'traffic_source': 'file' if traffic is not None else None,       # read from the committed PMC summary, not measured in this run

                         'algorithmic_bytes': 2 * BATCH * T_FRAMES * (SIDE // 2) ** 2 * 64 * 2 + 64 * 27 * 64 * 2,
                         'also': [{'bound': 'mfma', 'kernel': 'conv_pp_kernel<128> fwd (+BN statistics), 3x3x3 128->128 on (8,8,28,28,128)',
                                   'achieved': ach2, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach2 / PEAK_BF16_TFLOPS,
                                   'launches_timed': len(ms2), 'avg_launch_ms': avg2,
                                   'algorithmic_bytes': 2 * BATCH * (T_FRAMES // 2) * (SIDE // 4) ** 2 * 128 * 2 + 128 * 27 * 128 * 2}],
                         # the whole step against the same peak: executed conv GFLOP of one step / timed step duration
                         'step_gflop': round(STEP_GFLOP_EXECUTED, 1), 'step_gflop_reference': round(STEP_GFLOP_REFERENCE, 1),
                         'step_tflops': STEP_GFLOP_EXECUTED * 1e-3 / step_s,
                         'step_frac': STEP_GFLOP_EXECUTED * 1e-3 / step_s / PEAK_BF16_TFLOPS,
                         # every conv shape class, forward / input gradient / weight gradient, from event pairs in the same two
                         # single-stream eager steps (so the >= 0.70 target of north_star is tracked by this line)
                         'stages': stage_table(prof['events'])},
        }
        if det_variant is not None:
            line['variants'] = {'deterministic': det_variant}
        if rccl is not None:
            if dt_nocoll is not None:
                rccl['step_ms_without_grad_collectives'] = round(1e3 * dt_nocoll, 4)
                rccl['exposed_wire_ms_measured'] = round(1e3 * (step_s - dt_nocoll), 4)
            line['rccl'] = rccl
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.cpu_batch)
        print(json.dumps(line), flush=True)
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
