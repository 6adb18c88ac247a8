#!/bin/bash
# One GPU call of the dropped-corner hunt (profiles/r06_flake.md).  Logs under gpurun_out/flake/.
# usage: bash tools/diag/run_hunt.sh [REPLAYS [STEP_RUNS]]
set -o pipefail
cd "$(dirname "$0")/../.."
OUT=gpurun_out/flake; mkdir -p $OUT
REPLAYS=${1:-50000}; RUNS=${2:-150}
ORIG=tools/diag/libmscl_hip_orig.so; DIAG=tools/diag/libups_diag.so; R=tools/diag/flake_repro
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; rocm-smi --showserial 2>/dev/null | grep -i serial; } > $OUT/box.txt 2>&1
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 240 "$@" > $OUT/$name.log 2>&1; echo "   rc $? $(grep SUMMARY $OUT/$name.log)"; }
run repro_orig_graph      $R --lib $ORIG --replays $REPLAYS
run repro_orig_convs      $R --lib $ORIG --replays $REPLAYS --side convs
run repro_orig_elem       $R --lib $ORIG --replays $REPLAYS --side elem
run repro_orig_none       $R --lib $ORIG --replays $REPLAYS --side none
run repro_orig_onestream  $R --lib $ORIG --replays $((REPLAYS / 3)) --one-stream
run repro_orig_eager      $R --lib $ORIG --replays $((REPLAYS / 5)) --eager
run repro_ship_graph      $R --replays $REPLAYS
run repro_diag_graph      $R --lib $ORIG --diag $DIAG --replays $REPLAYS
GPU_MAX_HW_QUEUES=1 run repro_orig_hwq1 $R --lib $ORIG --replays $REPLAYS
echo "== step-level probe, original kernel, three streams ($RUNS runs)"
MSCL_LIB=$ORIG timeout -k 10 420 python tools/flake_det.py $RUNS graph > $OUT/step_orig.log 2>&1; echo "   rc $? $(tail -1 $OUT/step_orig.log)"
echo "== step-level probe through the self-checking kernel ($RUNS runs)"
FLAKE_DIAG=1 timeout -k 10 420 python tools/flake_det.py $RUNS graph > $OUT/step_diag.log 2>&1; echo "   rc $? $(tail -1 $OUT/step_diag.log)"
grep -h "first difference\|rec [0-9]*:" $OUT/step_*.log | head -40
