#!/bin/bash
# A/B of an environment switch on one box:   bash tools/ab_env.sh VAR v1 v2 ...   (two rounds, bench.py defaults without extras)
VAR=$1; shift
for round in 1 2; do
  for v in "$@"; do
    export $VAR=$v
    python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json, sys, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ['$VAR'], round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v['avg_ms'] * 1e3, 1) for k, v in d['kernels'].items()})"
  done
done
