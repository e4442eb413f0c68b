#!/bin/bash
# Per-kernel GPU time of any python tool: bash tools/prof_script.sh tools/<script>.py [rows]   (on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SCRIPT=$1; ROWS=${2:-30}
NAME=$(basename "$SCRIPT" .py)
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__; __graft_entry__.build()" || exit 1   # never build under the profiler
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -o p -- python3 "$ROOT/$SCRIPT" 2>&1 | grep -v "rocprofv3\|amdgpu.ids" | tail -4
mkdir -p "$ROOT/gpurun_out/prof_$NAME"
cp /tmp/prof_$NAME/*kernel_stats.csv "$ROOT/gpurun_out/prof_$NAME/"
python3 "$ROOT/tools/kernel_stats_summary.py" "$ROOT/gpurun_out/prof_$NAME/p_kernel_stats.csv" "$ROWS"
