"""G10: the retrieval metric of the reference, tools/test_retrival.py:283-303, on seeded clustered features.

The script cannot be imported (mmcv at module level, everything inside main()), so the metric's own source lines are read
from the reference file, checked to be the block that starts at `ks = [1,5,10,20,50]` and ends with the per-k loop, and
executed here with the four tensors bound -- the reference's arithmetic, not a restatement.  Inputs regenerate from the
seed; tests/golden/retrieval_g10.json keeps the accuracies (and the oracle restatement is asserted equal while doing so).
Development container only."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import retrieval as oret                          # noqa: E402

REF = '/root/reference/tools/test_retrival.py'


features = oret.clustered_features


def main():
    lines = open(REF).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.strip() == 'ks = [1,5,10,20,50]')
    end = next(i for i in range(start, len(lines)) if 'NN acc' in lines[i])
    block = [l[4:] if l.startswith('    ') else l for l in lines[start:end + 1]]
    assert any('topk' in l for l in block) and any('normalize' in l for l in block), 'unexpected reference block'
    src = '\n'.join(block)
    cases = {}
    for seed, noise in ((7, 2.0), (8, 6.0), (9, 12.0)):
        tf, tl, sf, sl = features(seed, noise=noise)
        env = dict(torch=torch, F=F, np=np, train_feature=tf.clone(), test_feature=sf.clone(), train_label=tl, test_label=sl, print=lambda *a: None)
        exec(src, env)
        ref = dict(zip(env['ks'], env['NN_acc']))
        ora = oret.knn_accuracy(tf.numpy(), tl.numpy(), sf.numpy(), sl.numpy())
        assert all(abs(ref[k] - ora[k]) < 1e-6 for k in ref), (ref, ora)
        cases[f'seed{seed}'] = dict(seed=seed, noise=noise, acc={str(k): v for k, v in ref.items()})
    out = os.path.join(ROOT, 'tests', 'golden', 'retrieval_g10.json')
    json.dump(dict(source='tools/test_retrival.py:%d-%d executed verbatim' % (start + 1, end + 1), cases=cases), open(out, 'w'), indent=1)
    print('wrote', out, cases)


if __name__ == '__main__':
    main()
