#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "config2 or whole_tensors or small_cotangent or two_arithmetic" 2>&1 | tail -2
for i in 1 2 3; do WHICH=b timeout 300 python tools/time_kernels.py geo 2>&1 | grep "gather\|stream"; done
timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('literal', d['ms_per_step'], 'settled', d['settled']['ms_per_step'], d['settled']['kernel_us'])"
