#!/bin/bash
# kernel time and SQ counters of the window-resident stride-2 input gradient on the layer-2 entry shape (scratch: gpurun_out/stemprof)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stemprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/tools/bench_conv.py --only stem_rgb_pairw --modes wgrad --iters 10 > $O/st.log 2>&1
f=$(ls $O/st/*/*kernel_stats.csv | head -1); grep "wgrad_stem_kernel\|Name" $f | cut -d, -f1-8 | cut -c1-200
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc -- python3 $R/tools/bench_conv.py --only stem_rgb_pairw --modes wgrad --iters 3 --rounds 1 > $O/pmc.log 2>&1
f=$(ls $O/pmc/*/*counter_collection.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'wgrad_stem_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']]['v'] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in acc: print(k, acc[k]['v'] / max(n[k], 1))
PY
rm -rf $O/st/*/*.db $O/pmc/*/*.db
