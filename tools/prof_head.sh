#!/bin/bash
# Per-kernel averages of ECHOBlock's dense tail alone (tools/time_head.py), forward and backward in separate profiler runs.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for shape in net dp; do
  for which in f b; do
    rm -rf /tmp/head_prof
    SHAPE=$shape WHICH=$which rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/head_prof -o h -- python3 "$ROOT/tools/time_head.py" 2>&1 | grep -E "forward|backward"
    python3 - /tmp/head_prof/h_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'fc_' in r['Name']:
        print(f"    {r['Name'][:60]:60s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:7.1f} us")
PY
  done
done
