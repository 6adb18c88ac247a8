"""Layer-1 input gradient with / without the fused BatchNorm-backward reduce (mscl_conv_halo64_dgrad_bn) and the BatchNorm
backward passes it replaces, one launch each on the (8,16,56,56,64) map.  usage: python tools/bench_fused_bn.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from mscl_amd import kernels as K
dev = torch.device('cuda:0')
C = 64; shape = (8, 16, 56, 56, C)
d = K.conv_desc(shape, C, (3, 3, 3), (1, 1, 1), (1, 1, 1))
dy = torch.randn(shape, device=dev).to(torch.bfloat16); wT = (torch.randn((C, 3, 3, 3, C), device=dev) * 0.05).to(torch.bfloat16)
add = torch.randn(shape, device=dev).to(torch.bfloat16); y = torch.randn(shape, device=dev).to(torch.bfloat16)
out = torch.relu(y); g = torch.ones(C, device=dev); mean = torch.zeros(C, device=dev); inv = torch.ones(C, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
scr = torch.zeros(K.STAT_SLOTS * 4 * C, device=dev)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print('dgrad plain        %.1f us' % t(lambda: K.conv3d_dgrad(dy, wT, d, addend=add)))
print('dgrad fused reduce %.1f us' % t(lambda: K.conv_halo64_dgrad_bn(dy, wT, d, y, out, mean, inv, scr, addend=add)))
dz = K.conv_halo64_dgrad_bn(dy, wT, d, y, out, mean, inv, scr, addend=add)
print('bn bwd full +dres  %.1f us' % t(lambda: K.bn_act_bwd(dz, out, y, g, mean, inv, dg, db, True, scr, want_identity_dres=True)))
print('bn bwd full        %.1f us' % t(lambda: K.bn_act_bwd(dz, out, y, g, mean, inv, dg, db, True, scr)))
print('bn bwd apply only  %.1f us' % t(lambda: K.bn_act_bwd(dz, None, y, g, mean, inv, dg, db, 2, scr)))
