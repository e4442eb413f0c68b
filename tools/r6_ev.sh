#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/round_evidence.sh r06 > gpurun_out/r06_round_evidence.log 2>&1
FIELDCONV_DEV=1 timeout 300 python tools/stamps.py stream --wave 0 8 --tiles 3 --warm 300 > gpurun_out/r06_stream_stamps.txt 2>&1
FIELDCONV_DEV=1 timeout 300 python tools/stamps.py fwd --wave 0 4 --tiles 2 --warm 300 > gpurun_out/r06_fwd_stamps.txt 2>&1
tail -5 gpurun_out/r06_round_evidence.log
cat gpurun_out/r06_pmc_summary.txt | head -60
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_driver_protocol.json').read().strip().splitlines()[-1])
print('literal', d['ms_per_step'], d['value'], 'settled', d['settled']['ms_per_step'], d['roofline'])
PY
