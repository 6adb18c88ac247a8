"""oracle/ -- CPU restatement of the reference's MSCL training hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package; the
product (mscl_amd/) never does, and it fails loudly when its HIP library is missing rather than
falling back to anything in here.

What it restates (every function cites the reference file:line it follows):
  nets.py   torchvision-style VideoResNet trunks (RGB r3d_18, flow r2d_18), FPN / SEPC / TPNMoCo,
            BaseMoCo
  mscl.py   MoCoV2, MoCoHead, MSCLWithAugMxHead, MSCLWithAugPosHeadV2 (LMCL), MSCLWithAug,
            _parse_losses, top_k_accuracy, grad-clip + SGD step
  fill.py   the closed-form, RNG-free parameter fill used on both sides of every parity test

Parity status: PINNED against the reference's own Python for everything that lives under
/root/reference (recognizers, necks, heads, loss, flow trunk, vendored R3D twin): the script
tools/oracle/make_golden.py imports those files in the development container, runs them on the
same filled weights / seeded inputs, asserts agreement with this package and writes
tests/golden/*.npz.  UNPINNED by any reference test (the reference has none for this path,
SURVEY.md §4) and for the third-party pieces not vendored in the reference: torchvision's r3d_18
(restated from the reference's structurally identical twin mmaction/models/backbones/r3d.py) and
mmcv's ConvModule / xavier_init / OptimizerHook (restated from their documented behaviour).
The only reference-pinned known-answer vectors on this path are top_k_accuracy's
(tests/test_metrics/test_accuracy.py:118-163), reproduced in tests/test_oracle.py.
"""
