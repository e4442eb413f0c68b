#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6j
o=gpurun_out/r6j
timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/bench.json 2> $o/bench.err
timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/bench2.json 2> $o/bench2.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$o/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline --no-kernel-events > $GRAFT_REPO_ROOT/$o/prof_bench.json 2> $GRAFT_REPO_ROOT/$o/prof.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob,csv
for f in ('gpurun_out/r6j/bench.json','gpurun_out/r6j/bench2.json','gpurun_out/r6j/prof_bench.json'):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step', round(d['ms_per_step'],4), 'value', round(d['value'],1), 'settled', d.get('settled') and round(d['settled']['ms_per_step'],4), d.get('settled') and d['settled'].get('kernel_us'))
    except Exception as e: print(f, 'ERR', e)
for f in glob.glob('gpurun_out/r6j/prof/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:12]: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
