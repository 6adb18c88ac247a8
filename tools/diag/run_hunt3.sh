#!/bin/bash
# Third GPU call of the hunt (profiles/r06_flake.md): the instruction forms of the VALU probe beside the conv chains, on another box.
set -o pipefail
cd "$(dirname "$0")/../.."
OUT=gpurun_out/flake3; mkdir -p $OUT
REPLAYS=${1:-20000}
ORIG=tools/diag/libmscl_hip_orig.so; R=tools/diag/flake_repro
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; } > $OUT/box.txt 2>&1
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 200 "$@" > $OUT/$name.log 2>&1; echo "   rc $? $(grep 'SUMMARY\|PROBE' $OUT/$name.log | tr '\n' ' ')"; }
for m in 0 1 2 3 4 5 6; do run probe$m $R --lib $ORIG --replays $REPLAYS --probe $m --side convs --streams B; done
run probe0_none $R --lib $ORIG --replays $REPLAYS --probe 0 --side none
run probe0_l3   $R --lib $ORIG --replays $REPLAYS --probe 0 --side convs --streams B --layers 12:13
MSCL_PP_KSPLIT=1 run probe0_l3_nosplit $R --lib $ORIG --replays $REPLAYS --probe 0 --side convs --streams B --layers 12:13
