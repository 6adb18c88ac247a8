#!/bin/bash
# round 4, GPU call 9: BatchNorm grid caps re-swept inside the step with the two-block layer-1 kernels; eager forced-dist leg
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c9; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
b() { tag=$1; shift; env "$@" timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_$tag.json 2> $O/bench_$tag.err || { tail -5 $O/bench_$tag.err; exit 1; }
  python - $O/bench_$tag.json $tag <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), round(d['ms_per_step'],3), flush=True)
PY
}
for rep in 1 2; do
b base_$rep X=1
b f256_$rep MSCL_BN_FWD_CAP=256
b f1024_$rep MSCL_BN_FWD_CAP=1024
b a256_$rep MSCL_BN_APPLY_CAP=256
b a1024_$rep MSCL_BN_APPLY_CAP=1024
b r256_$rep MSCL_BN_RED_CAP=256
b r1024_$rep MSCL_BN_RED_CAP=1024
done
echo "== eager forced one-rank group (the world-size > 1 code path incl. the no-collective leg)"
MSCL_FORCE_DIST=1 timeout -k 10 300 python -u bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $O/bench_forced_eager.json 2> $O/bench_forced_eager.err || { tail -8 $O/bench_forced_eager.err; exit 1; }
python - $O/bench_forced_eager.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('forced eager', round(d['value'],1), d['config']['launch'][:70], json.dumps(d.get('rccl')))
PY
