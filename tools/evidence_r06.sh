#!/bin/bash
# Round-6 evidence from ONE gpurun call on the final code.  usage (dev container):
#   gpurun --timeout 1200 -- "GIT_HEAD=$(git rev-parse --short HEAD) bash tools/evidence_r06.sh"
# Every file it writes carries the commit (GIT_HEAD) it was measured on; tools/copy_evidence_r06.py copies gpurun_out/ev_r06/* to profiles/r06_*.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev_r06; rm -rf $O; mkdir -p $O
H=${GIT_HEAD:-unknown}; echo "$H" > $O/HEAD
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err
# (--no-variants: the deterministic-mode leg would otherwise be the LAST step of the trace, and the launch counts / loss-phase timeline below
#  would describe deterministic mode -- as rounds 5's and the first round-6 copy of these two files did)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_line_under_rocprof.json 2> $O/stats.err
f=$(ls $O/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv
t=$(ls $O/stats/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$t" ] && python3 $R/tools/count_step_kernels.py $O/stats > $O/step_launch_counts.txt 2>&1
[ -n "$t" ] && { echo "# commit $H: loss phase of one graph-replayed step (tools/loss_phase_trace.py on the kernel trace of the run above; the tracer serialises the streams)"; python3 $R/tools/loss_phase_trace.py $t 2>&1; } > $O/loss_phase.txt
rm -rf $O/stats
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d' ' -f1)
  MSCL_STREAMS=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > $O/pmc_$n.json 2> $O/pmc_$n.err
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/l1_$c -- python3 $R/tools/bench_conv.py --only l1_64_64 --iters 3 --modes fwd > $O/l1_$c.log 2>&1
done
cd $R
{ echo "# commit $H: per-kernel MFMA busy and HBM traffic of the whole step (three PMC passes)"; echo "# NOTE: this is the EAGER step on ONE stream (MSCL_STREAMS=1 --no-graph: the counters need serialised kernels), so it carries the eager path's glue (~65 __amd_rocclr_copyBuffer per step) and no overlap.  The headline is a three-stream graph replay: its launch counts follow the table."; python3 tools/pmc_step_summary.py $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE 11; echo; echo "## launches of ONE graph-replayed step (tools/count_step_kernels.py on the kernel trace of the bench run above)"; echo; cat $O/step_launch_counts.txt; } > $O/step_utilisation.md 2> $O/step_utilisation.err
{ echo "# commit $H: every conv stage alone (tools/bench_conv.py --iters 20, min of 3 rounds), TFLOP/s against the 2500 TFLOP/s dense bf16 peak"; python3 tools/bench_conv.py --iters 20 2>/dev/null | grep -v amdgpu; } > $O/conv_stage.log
{ echo "# commit $H: SlowOnly-50 conv shapes one at a time (tools/bench_conv.py --r50 --iters 10, min of 3 rounds); GB/s = (input + output map bytes) / time"; python3 tools/bench_conv.py --r50 --iters 10 2>/dev/null | grep -v amdgpu; } > $O/conv_stage_r50.log
{ echo "# commit $H"; python3 tools/chain_times.py 2>/dev/null | grep -v amdgpu; } > $O/chain_times.txt
python3 tools/bench_trunk.py > $O/trunk_r18.json 2>/dev/null
python3 tools/bench_trunk.py --r50 > $O/trunk_r50.json 2>/dev/null
python3 bench.py --deterministic --no-cpu-baseline > $O/bench_line_deterministic.json 2>/dev/null
python3 tools/bench_step_r50.py > $O/step_config5_r50_32x224.json 2>/dev/null
python3 tools/bench_step_r50.py --frames 8 > $O/step_config5_r50_8x224.json 2>/dev/null
python3 tools/traffic_json.py $O/l1_FETCH_SIZE $O/l1_WRITE_SIZE $H > $O/traffic_layer1.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/nce -- python3 $R/tools/bench_nce.py > $O/nce_graph.txt 2>/dev/null
f=$(ls $O/nce/*/*kernel_stats.csv 2>/dev/null | head -1)
{ echo "# commit $H: InfoNCE passes, K = 65536, dim 128 (tools/bench_nce.py): graph-replayed call pairs, then the kernels alone (rocprofv3 --kernel-trace --stats, us per launch)"; grep "R=" $O/nce_graph.txt; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r['Name'].startswith(('void nce_', 'nce_')):
        print(f"{r['Name'][:48]:48s} {r['Calls']:>5s} launches  {float(r['AverageNs']) / 1e3:7.1f} us   {128 * 65536 * 4 / float(r['AverageNs']):7.2f} GB/s of queue reads" if 'finish' not in r['Name'] and 'reduce' not in r['Name'] else f"{r['Name'][:48]:48s} {r['Calls']:>5s} launches  {float(r['AverageNs']) / 1e3:7.1f} us")
PY
} > $O/nce_passes.md
rm -rf $O/nce
cd $R
{ echo "# commit $H: aten device ops of one eager step by call site (tools/glue_launches.py)"; python3 tools/glue_launches.py 2>/dev/null | grep -v amdgpu; } > $O/glue_launches.txt
{ echo "# commit $H: BatchNorm passes alone (tools/bench_bn.py)"; python3 tools/bench_bn.py 2>/dev/null | grep -v amdgpu; } > $O/bn_passes.txt
{ echo "# commit $H: the grouped weight-gradient launch against per-layer launches, two captured whole-step graphs replayed alternately in one process (tools/ab_step.py, clip-pairs/s)"; python3 tools/ab_step.py "nn.GROUP_WGRADS[0]=False" "nn.GROUP_WGRADS[0]=True" 2>/dev/null | grep -v amdgpu; } > $O/ab_group_wgrad.txt
{ echo "# commit $H: alternating whole-step graphs in one process (tools/ab_step.py, clip-pairs/s): window-resident stride-2 input gradient off / on; side-chain split-K cap 16 / 4 / 1"
  python3 tools/ab_step.py "os.environ.__setitem__('MSCL_DGRAD_S2','0'); lib.call_raw('mscl_tuning_reload')" "os.environ.__setitem__('MSCL_DGRAD_S2','1'); lib.call_raw('mscl_tuning_reload')" 2>/dev/null | grep -v amdgpu
  # (the InfoNCE vector / MFMA pair was measured with the switch MSCL_NCE_MFMA that commit abbe824 still had: 1179.3 -> 1184.5; the
  #  switch is gone, the vector kernels take the shapes the MFMA kernels do not)
  python3 tools/ab_step.py "model.set_side_split(16,16)" "model.set_side_split(4,4)" "model.set_side_split(1,1)" 2>/dev/null | grep -v amdgpu; } > $O/ab_round6.txt
{ echo "# commit $H: the stand-alone reproducer of the round-5 dropped corner against the PRODUCT library (tools/diag/flake_repro: 1 000 000 comparisons), then its VALU probe in the compiled kernel's packed-fp32 form beside the RGB key trunk's convs on this box"
  tools/diag/flake_repro --replays 50000 2>&1 | grep "device\|reference\|SUMMARY"; tools/diag/flake_repro --replays 20000 --probe 0 --side convs --streams B 2>&1 | grep "PROBE"; rocm-smi --showuniqueid 2>/dev/null | grep -i unique; } > $O/flake_repro.txt
{ echo "# commit $H: the GPU suite under the two hazard probes of tests/conftest.py"
  for v in "MSCL_TEST_POISON=2" "MSCL_TEST_JITTER=200000"; do echo "\$ $v python -m pytest tests -q -m gpu"; env $v python3 -m pytest tests -q -m gpu > $O/probe_$v.log 2>&1; grep -E "^(FAILED|E  )" $O/probe_$v.log | head -20; tail -1 $O/probe_$v.log; done; } > $O/probe_runs.txt
F=$R/tests/golden/oracle_curve_b8_t16_112_k65536.json
if [ -f $F ]; then
  n=$(python3 -c "import json;print(len(json.load(open('$F'))['steps']))")
  { echo "# commit $H: HIP path in deterministic mode at the BENCHMARK size against the fp32 oracle's curve (made on CPU by tools/train_curve.py --oracle-only, committed fixture: $n steps)"; python3 tools/train_curve.py --steps $n --batch 8 --frames 16 --side 112 --queue 65536 --every 50 --deterministic --oracle-json $F 2>/dev/null | grep -v amdgpu; } > $O/training_curve.md
fi
rm -rf $O/pmc_*/*/*.db 2>/dev/null
ls -la $O | head -40; echo finished
