#!/bin/bash
# round 4, GPU call 7: PMC counters of the layer-1 kernels (LDS activity / conflicts / waits), the copy sites of a step, one-stage 1x1x1 A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c7; rm -rf $O; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_l1a -- python3 $R/tools/bench_conv.py --only l1_64_64,l2_128_128 --iters 3 --modes fwd,dgrad,wgrad > $O/pmc_l1a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/pmc_l1b -- python3 $R/tools/bench_conv.py --only l1_64_64,l2_128_128 --iters 3 --modes fwd,dgrad,wgrad > $O/pmc_l1b.log 2>&1
cd $R
python3 - <<'PY' | tee $O/pmc_l1.txt
import csv, glob, collections, os, re
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/c7'
for d in ('pmc_l1a','pmc_l1b'):
    f=glob.glob(f'{O}/{d}/**/*counter_collection.csv', recursive=True)
    if not f: print(d,'no csv'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=re.sub(r'\(.*$','',re.sub(r'^void\s+','',r['Kernel_Name']))[:40]
        if any(s in k for s in ('halo','conv_pp','wgrad')): acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(d,k,' '.join(f'{c}={sum(x)/len(x):.3g}' for c,x in sorted(v.items())))
PY
timeout -k 10 300 python3 -u tools/find_copies.py 2>&1 | grep -v amdgpu | tee $O/find_copies.log
run() { name=$1; shift; echo "== $name"; timeout -k 10 420 "$@" 2>&1 | grep -v amdgpu | tee $O/$name.log; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1; }
run sweep_one_stage python3 -u tools/bench_conv.py --r50 --sweep MSCL_ONE_STAGE=0,1 --modes fwd,dgrad
timeout -k 10 300 python3 -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 300 -k "fwd_dgrad_wgrad" 2>&1 | tail -3
