"""margins of the oracle's LMCL rows at B = 32 (test_step_at_the_shipped_batch_of_32): how close is each row's label score to the
top-1 / top-5 boundary, against the accuracy the HIP path reports"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_model_gpu import build
from mscl_amd.synthetic import synthetic_batch
from oracle import fill as ofill, mscl as om
dev = torch.device('cuda:0')
B, T, H, Kq = 32, 8, 112, 65536
batch = synthetic_batch(B, T, H, H, 0, 0)
for rep in range(3):
    model, _ = build(T, Kq, dev)
    out = model.train_step({k: [t.to(dev) for t in v] for k, v in batch.items()})
    print('hip', rep, out['log_vars']['top1_acc_pos'] * 128, out['log_vars']['top5_acc_pos'] * 128, out['log_vars']['loss_pos'])
orc = om.MSCLWithAug(num_frames=T, K=Kq); ofill.fill_module(orc); orc.train()
torch.manual_seed(100)
oo = orc.train_step(batch)
print('oracle', float(oo['log_vars']['top1_acc_pos']) * 128, float(oo['log_vars']['top5_acc_pos']) * 128, float(oo['log_vars']['loss_pos']))
f = orc._features
scores, labels = om.lmcl_scores(f['img']['q_mlvl'][0], f['base']['q_mlvl'][-1], f['aug']['q_mlvl'][-1], orc.T, getattr(orc.sup_head, 'trans_flow', None))
scores = scores.detach().double(); n = scores.shape[0]
lab = scores[torch.arange(n), labels]
others = scores.clone(); others[torch.arange(n), labels] = -1e9
srt = others.sort(dim=1, descending=True).values
for k in (1, 5):
    m = (lab - srt[:, k - 1]).abs()
    print('k', k, 'scores std', float(scores.std()), 'margins sorted', [round(float(v), 4) for v in m.sort().values[:24]])
    for eps in (0.005, 0.01, 0.02, 0.05):
        print('   eps', eps, 'rows within', int((m < eps).sum()))
