#!/bin/bash
# Per-kernel GPU time of the config-5 step (bench.py --mode dp, one rank): rocprofv3 kernel trace, summary printed per step.
# Usage (on the GPU box): bash tools/prof_dp.sh [rows]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__; __graft_entry__.build()" || exit 1   # never build under the profiler
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dp_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp_prof -o dp -- python3 "$ROOT/bench.py" --mode dp --steps 20 --warmup 5 --cold --no-cpu-baseline --no-extras --no-kernel-events 2>&1 | tail -1 | cut -c1-300
mkdir -p "$ROOT/gpurun_out/dp_prof"
cp /tmp/dp_prof/*kernel_stats.csv "$ROOT/gpurun_out/dp_prof/"
python3 - "$ROOT/gpurun_out/dp_prof/dp_kernel_stats.csv" "${1:-70}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# one finishing launch per convolution whatever the backward arrangement (17 convolutions per step)
convs = sum(int(r['Calls']) for r in rows if 'fc_reduce_param_grads_kernel' in r['Name'])
steps = convs / 17.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
print(f'steps seen {steps:.1f}; GPU busy {tot / steps / 1e6:.3f} ms/step, {calls / steps:.1f} launches/step')
for r in rows[:int(sys.argv[2])]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']) / steps:6.1f}/step {float(r['TotalDurationNs']) / steps / 1e3:8.1f} us/step  avg {float(r['AverageNs']) / 1e3:7.1f}")
PY
