"""tests/golden/*r50* from the REFERENCE's own Python (via ref_harness) for configs/recognition/moco/mscl_r50_cosm_lr3e-2.py
(BASELINE.json configs[4]: ResNet3dSlowOnly-50 + resnet_flow.r2d_50).  Run only in the development container:
  python tools/oracle/make_golden_r50.py
While generating, oracle.mscl.MSCLWithAug(arch='r50') is asserted equal to the reference (step 0 to 1e-6: losses, every
gradient, the clip norm, integer state).  Fixtures are data: inputs regenerate from seeds (mscl_amd.synthetic)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as rh                     # noqa: E402
import make_golden as mg                     # noqa: E402


def main():
    rh.install()
    name = mg.CFG_OF['r50']
    cfg = rh.load_ref_cfg(name)
    keep = {k: cfg[k] for k in ('model', 'optimizer', 'optimizer_config', 'lr_config', 'total_epochs', 'dataset_size',
                                'num_frames', 'find_unused_parameters')}
    with open(os.path.join(mg.OUT, 'ref_config_r50.json'), 'w') as f:
        json.dump(keep, f, indent=1, sort_keys=True)
    ref, _ = rh.build_ref_model(num_frames=8, cfg_name=name)
    man = [[n, list(t.shape), str(t.dtype)] for n, t in ref.state_dict().items()]
    with open(os.path.join(mg.OUT, 'state_dict_manifest_r50.json'), 'w') as f:
        json.dump(man, f)
    del ref
    # reduced spatial size (the trunk divides H by 32), then 112^2; one step each: with 53 batch-2 BatchNorm layers the second step
    # of reference and oracle already differ by fp32 rounding chaos (a 0.9 relative gap on the stem gradient), so it pins nothing
    mg.run_steps(B=2, T=8, H=64, n_steps=1, K=4096, tag='r50_step_b2_t8_h64', arch='r50')
    mg.run_steps(B=2, T=8, H=112, n_steps=1, K=65536, tag='r50_step_b2_t8_h112', arch='r50')


if __name__ == '__main__':
    main()
