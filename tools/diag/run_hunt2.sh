#!/bin/bash
# Second GPU call of the hunt (profiles/r06_flake.md): which neighbours trigger the event, and which instruction form it hits.
set -o pipefail
cd "$(dirname "$0")/../.."
OUT=gpurun_out/flake2; mkdir -p $OUT
REPLAYS=${1:-30000}
ORIG=tools/diag/libmscl_hip_orig.so; R=tools/diag/flake_repro
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; } > $OUT/box.txt 2>&1
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 200 "$@" > $OUT/$name.log 2>&1; echo "   rc $? $(grep 'SUMMARY\|PROBE' $OUT/$name.log | tr '\n' ' ')"; }
run base_orig        $R --lib $ORIG --replays $REPLAYS
run probe_convs      $R --lib $ORIG --replays $REPLAYS --probe --side convs
run probe_none       $R --lib $ORIG --replays $REPLAYS --probe --side none
run nopk_orig        $R --lib tools/diag/libmscl_hip_orig_nopk.so --replays $REPLAYS
run nopk_ship        $R --lib tools/diag/libmscl_hip_ship_nopk.so --replays $REPLAYS
run streamB          $R --lib $ORIG --replays $REPLAYS --side convs --streams B
run streamC          $R --lib $ORIG --replays $REPLAYS --side convs --streams C
run B_stem           $R --lib $ORIG --replays $REPLAYS --side convs --streams B --layers 0:1
run B_layer1         $R --lib $ORIG --replays $REPLAYS --side convs --streams B --layers 1:5
run B_layer2         $R --lib $ORIG --replays $REPLAYS --side convs --streams B --layers 5:10
run B_layer3         $R --lib $ORIG --replays $REPLAYS --side convs --streams B --layers 10:15
run B_layer4         $R --lib $ORIG --replays $REPLAYS --side convs --streams B --layers 15:20
run C_stem           $R --lib $ORIG --replays $REPLAYS --side convs --streams C --layers 0:1
run C_layer1         $R --lib $ORIG --replays $REPLAYS --side convs --streams C --layers 1:5
run C_layer34        $R --lib $ORIG --replays $REPLAYS --side convs --streams C --layers 10:20
