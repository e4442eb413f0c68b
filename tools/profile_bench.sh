#!/bin/bash
# rocprofv3 evidence for bench.py's default command, on the GPU box:   bash tools/profile_bench.sh <tag>
#   pass 0: --kernel-trace --stats                         -> gpurun_out/<tag>_kernel_stats.csv
#   pass 1: --pmc FETCH_SIZE        pass 2: --pmc WRITE_SIZE          (TCC slots: not both in one pass)
#   pass 3: --pmc SQ busy / instruction counters
# and gpurun_out/<tag>_pmc_counters.json (copy to profiles/pmc_counters.json: bench.py reads it and checks the digest).
# Counter passes never carry a trace option (the pool refuses that combination).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-prof}
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
# build first, from a process no profiler has touched (bench.py refuses to build under rocprofv3)
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__; __graft_entry__.build()" || exit 1
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10"
rm -rf /tmp/pb_*
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_stats -o p -- $CMD > "$OUT/${TAG}_bench_under_rocprof.json" 2>/dev/null
cp /tmp/pb_stats/*kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null || cp $(find /tmp/pb_stats -name '*kernel_stats.csv' | head -1) "$OUT/${TAG}_kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pb_fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pb_write -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_INSTS_SALU \
    --output-format csv -d /tmp/pb_sq -o p -- $CMD > /dev/null 2>&1
python3 "$ROOT/tools/pmc_summary.py" /tmp/pb_fetch /tmp/pb_write /tmp/pb_sq > "$OUT/${TAG}_pmc_summary.txt"
python3 - "$ROOT" "$OUT/${TAG}_pmc_summary.txt" "$OUT/${TAG}_pmc_counters.json" "$OUT/${TAG}_kernel_stats.csv" <<'PY'
import csv, json, re, sys
root, summary, out, stats = sys.argv[1:5]
sys.path.insert(0, root)
from fieldconv_amd.build import _source_digest
kern, cur = {}, None
for line in open(summary):
    if not line.startswith(' '):
        cur = line.strip()
        kern[cur] = {}
    else:
        m = re.match(r'\s+(\S+)\s+mean\s+([0-9.eE+-]+)', line)
        if m:
            kern[cur][m.group(1)] = float(m.group(2))
avg_ns = {}
try:
    for r in csv.DictReader(open(stats)):
        avg_ns[r['Name'].split('(')[0].replace('void ', '')] = float(r['AverageNs'])
except Exception:
    pass
names = {'fc_forward': ('fc_forward_factored_kernel', 'fc_forward_ring_kernel', 'fc_forward_kernel'), 'fc_backward_data': ('fc_backward_gather_kernel', 'fc_backward_data_kernel'),
         'fc_backward_filter': ('fc_backward_stream_kernel', 'fc_backward_filter_half2_kernel', 'fc_backward_filter_half_kernel', 'fc_backward_filter_kernel')}
res = {}
for short, cands in names.items():
    for k, c in kern.items():
        if any(('::' + cand + '<') in k or k.endswith('::' + cand) for cand in cands) and 'SQ_BUSY_CYCLES' in c:
            cyc = c['SQ_BUSY_CYCLES'] / 32.0            # summed over the 32 shader engines: cycles of the launch
            e = {'kernel': k, 'launch_cycles': cyc,
                 'valu_busy': c['SQ_ACTIVE_INST_VALU'] * 4.0 / (1024 * cyc),      # quad-cycles, summed over 1024 SIMDs
                 'mfma_busy': c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc),       # cycles, summed over 1024 SIMDs
                 'wave_wait_frac': c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'],
                 'valu_insts': c['SQ_INSTS_VALU'], 'mfma_insts': c['SQ_INSTS_MFMA'], 'salu_insts': c['SQ_INSTS_SALU']}
            if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
                e['hbm_read_bytes'] = int(c['FETCH_SIZE'] * 1024 * 2)            # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
                e['hbm_write_bytes'] = int(c['WRITE_SIZE'] * 1024)
                e['hbm_bytes_per_launch'] = e['hbm_read_bytes'] + e['hbm_write_bytes']
            if k in avg_ns:
                e['rocprof_avg_us'] = avg_ns[k] / 1e3
            res[short] = e
json.dump({'library_source_digest': _source_digest(), 'command': 'python3 bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10',
           'formulas': 'launch_cycles = SQ_BUSY_CYCLES / 32 shader engines; valu_busy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * launch_cycles); '
                       'mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * launch_cycles); HBM read = 2 * FETCH_SIZE KiB (gfx950 calibration for '
                       'wide streaming reads; narrower gathers are uncalibrated: upper estimate), write = WRITE_SIZE KiB',
           'kernels': res}, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
