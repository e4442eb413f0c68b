#!/bin/bash
# Everything the judged numbers come from, in one GPU-box call:   bash tools/round_evidence.sh <tag>      (e.g. r05_h)
#   rocprofv3 kernel stats + counter passes of bench.py's command (tools/profile_bench.sh) -> profiles-ready files under gpurun_out/,
#   pmc_counters.json put in place for the bench lines that follow, the three bench lines, per-kernel profiles of the config-3 and
#   config-5 steps.  Copy gpurun_out/<tag>_* to profiles/ afterwards.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-evidence}
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd "$ROOT"
bash tools/profile_bench.sh "$TAG" > "$OUT/${TAG}_profile_bench.log" 2>&1
cp "$OUT/${TAG}_pmc_counters.json" "$ROOT/profiles/pmc_counters.json"
python3 bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_protocol.json" 2> "$OUT/${TAG}_bench_driver_protocol.err"
python3 bench.py --no-cpu-baseline --no-extras > "$OUT/${TAG}_bench_default.json" 2> /dev/null
python3 bench.py --mode dp --steps 20 --warmup 5 > "$OUT/${TAG}_bench_dp.json" 2> /dev/null
python3 bench.py --mode net --steps 20 --warmup 5 > "$OUT/${TAG}_bench_net.json" 2> /dev/null
bash tools/prof_net.sh > "$OUT/${TAG}_net_prof.log" 2>&1
bash tools/prof_dp.sh > "$OUT/${TAG}_dp_prof.log" 2>&1
head -c 400 "$OUT/${TAG}_bench_driver_protocol.json"; echo
tail -3 "$OUT/${TAG}_net_prof.log"
