#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6suite
timeout 2400 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r6suite/gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r6suite/gpu_tests.log
tail -30 gpurun_out/r6suite/gpu_tests.log
