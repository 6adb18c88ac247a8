import sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import importlib
tm = importlib.import_module('test_model_gpu')
dev = torch.device('cuda:0')
from mscl_amd import nn
for arm in sys.argv[1:]:
    exec(arm)
    ok = 0; bad = 0
    for i in range(6):
        try:
            tm.test_graphed_step_equals_eager_bitwise_in_deterministic_mode(dev); ok += 1
        except AssertionError as e:
            bad += 1; print('FAIL', arm, str(e)[:200], flush=True)
    print(arm, 'ok', ok, 'bad', bad, flush=True)
