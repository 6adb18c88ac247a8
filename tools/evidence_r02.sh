set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_v3.json 2> $O/bench_v3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_v3_prof.json 2> $O/bench_v3_prof.err
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d' ' -f1)
  MSCL_STREAMS=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > $O/pmc_$n.json 2> $O/pmc_$n.err
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/l1_$c -- python3 $R/tools/bench_conv.py --only l1_64 --iters 3 --modes fwd > $O/l1_$c.log 2>&1
done
python3 $R/tools/bench_conv.py --iters 20 > $O/conv_stage.log 2>&1
python3 $R/tools/bench_trunk.py > $O/trunk_r18.json 2>/dev/null
echo finished
