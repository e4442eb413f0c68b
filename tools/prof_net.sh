#!/bin/bash
# Per-kernel GPU time of the config-3 network step: rocprofv3 kernel trace of tools/bench_net.py, summary csv copied to
# gpurun_out/net_prof/.  Usage (on the GPU box): bash tools/prof_net.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__; __graft_entry__.build()" || exit 1   # never build under the profiler
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/net_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/net_prof -o net -- python3 "$ROOT/tools/bench_net.py" 2>&1 | grep "segmentation net"
mkdir -p "$ROOT/gpurun_out/net_prof"
cp /tmp/net_prof/*kernel_stats.csv "$ROOT/gpurun_out/net_prof/"
python3 - "$ROOT/gpurun_out/net_prof/net_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 40
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
print(f'GPU busy {tot / steps / 1e6:.3f} ms/step, {calls / steps:.1f} launches/step')
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print(f"{r['Name'][:64]:64s} {int(r['Calls']) / steps:5.1f}/step {float(r['TotalDurationNs']) / steps / 1e3:8.1f} us/step  avg {float(r['AverageNs']) / 1e3:7.1f}")
PY
