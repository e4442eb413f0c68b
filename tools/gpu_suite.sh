#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r06_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r06_gpu_tests.log
tail -22 gpurun_out/r06_gpu_tests.log
