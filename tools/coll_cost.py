"""Host and device cost of one c10d/RCCL collective call on a 1-rank group (what every eager data-parallel step pays per call,
before any wire time).  python tools/coll_cost.py"""
import os, time, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
dev = torch.device('cuda', 0)
small = torch.ones(1024, device=dev); big = torch.ones(25 << 20, device=dev)
out = torch.empty_like(small)
def bench(name, fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f'{name:42s} host {1e6 * (t1 - t0) / n:7.1f} us/call   device {1e3 * e0.elapsed_time(e1) / n:7.1f} us/call', flush=True)
bench('all_reduce AVG 4 KB (sync op)', lambda: dist.all_reduce(small, op=dist.ReduceOp.AVG))
bench('all_reduce AVG 4 KB async + wait', lambda: dist.all_reduce(small, op=dist.ReduceOp.AVG, async_op=True).wait())
bench('all_reduce AVG 100 MB', lambda: dist.all_reduce(big, op=dist.ReduceOp.AVG), n=50)
bench('all_gather_into_tensor 4 KB', lambda: dist.all_gather_into_tensor(out, small))
bench('all_to_all_single 4 KB', lambda: dist.all_to_all_single(out, small, output_split_sizes=[1024], input_split_sizes=[1024]))
bench('index_select 19 MB (for scale)', lambda: torch.ones(8, 602112, device=dev).index_select(0, torch.arange(8, device=dev)), n=50)
x = torch.ones(8, 602112, device=dev)
s2 = torch.cuda.Stream()
def cross():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        dist.all_reduce(small, op=dist.ReduceOp.AVG)
    torch.cuda.current_stream().wait_stream(s2)
bench('all_reduce from a side stream + joins', cross)
dist.destroy_process_group()
