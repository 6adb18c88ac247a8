#!/bin/bash
# Round-4 evidence from ONE gpurun call on the final code.  usage (dev container):
#   gpurun --timeout 1200 -- "GIT_HEAD=$(git rev-parse --short HEAD) bash tools/evidence_r04.sh"
# Every file it writes carries the commit (GIT_HEAD) it was measured on; copy gpurun_out/ev_r04/* to profiles/r04_*.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev_r04; rm -rf $O; mkdir -p $O
H=${GIT_HEAD:-unknown}; echo "$H" > $O/HEAD
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_line_under_rocprof.json 2> $O/stats.err
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d' ' -f1)
  MSCL_STREAMS=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > $O/pmc_$n.json 2> $O/pmc_$n.err
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/l1_$c -- python3 $R/tools/bench_conv.py --only l1_64_64 --iters 3 --modes fwd > $O/l1_$c.log 2>&1
done
cd $R
{ echo "# commit $H: per-kernel MFMA busy and HBM traffic of the whole step (three PMC passes, single stream, eager)"; python3 tools/pmc_step_summary.py $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE 11; } > $O/step_utilisation.md 2> $O/step_utilisation.err
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
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/l1w_$c -- python3 $R/tools/bench_conv.py --only l1_64_64 --iters 3 --modes wgrad > $O/l1w_$c.log 2>&1
done
cd $R
python3 tools/traffic_json.py $O/l1w_FETCH_SIZE $O/l1w_WRITE_SIZE $H wgrad_halo64 > $O/traffic_layer1_wgrad.json 2> $O/traffic_layer1_wgrad.err
{ echo "# commit $H: aten device ops of one eager step by call site (tools/glue_launches.py)"; python3 tools/glue_launches.py 2>/dev/null | grep -v amdgpu; } > $O/glue_launches.txt
{ echo "# commit $H"; MSCL_LIB=$R/mscl_amd/csrc/build/libmscl_hip_stamp.so python3 tools/pp_stamps.py 2>/dev/null | grep -v amdgpu; } > $O/pp_stamps.txt
{ echo "# commit $H: BatchNorm passes alone (tools/bench_bn.py)"; python3 tools/bench_bn.py 2>/dev/null | grep -v amdgpu; } > $O/bn_passes.txt
f=$(ls $O/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv
rm -rf $O/stats $O/pmc_*/*/*.db 2>/dev/null
ls -la $O | head -40; echo finished
