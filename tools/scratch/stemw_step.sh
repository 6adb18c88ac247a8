#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 -m pytest tests -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
  echo "new  $(python3 bench.py --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1))")"
  echo "old  $(MSCL_WGRAD_STEM=0 python3 bench.py --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1))")"
done
for i in 1 2; do
  echo "r50 trunk new $(python3 tools/bench_trunk.py --r50 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1))")"
  echo "r50 trunk old $(MSCL_WGRAD_STEM=0 python3 tools/bench_trunk.py --r50 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1))")"
done
echo "r18 trunk new $(python3 tools/bench_trunk.py 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1))")"
echo "r18 trunk old $(MSCL_WGRAD_STEM=0 python3 tools/bench_trunk.py 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1))")"
