#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6l
o=gpurun_out/r6l
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config2 or whole_tensors or small_cotangent" > $o/pytest.log 2>&1
echo "pytest rc=$?" >> $o/pytest.log
WHICH=b timeout 300 python tools/time_kernels.py geo > $o/t_prod.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/bench.json 2> $o/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline --no-kernel-events > $GRAFT_REPO_ROOT/$o/prof_bench.json 2> $GRAFT_REPO_ROOT/$o/prof.err
cd $GRAFT_REPO_ROOT
cp $(find /tmp/pk -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
tail -3 $o/pytest.log; grep -h "gather\|stream\|checksum" $o/t_prod.log
python - <<'PY'
import json,csv
for f in ('gpurun_out/r6l/bench.json','gpurun_out/r6l/prof_bench.json'):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step', round(d['ms_per_step'],4), 'value', round(d['value'],1), 'settled', d.get('settled') and round(d['settled']['ms_per_step'],4), d.get('settled') and d['settled'].get('kernel_us'))
    except Exception as e: print(f, 'ERR', e)
rows=list(csv.DictReader(open('gpurun_out/r6l/kernel_stats.csv')))
for r in rows[:10]: print(r['Name'][:80], r['Calls'], r['AverageNs'])
PY
