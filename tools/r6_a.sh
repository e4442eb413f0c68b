#!/bin/bash
# round 6, first GPU contact of the H-streaming arrangement: parity at config 2, isolated kernel times, a short bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config2 or whole_tensors or small_cotangent" > gpurun_out/r6a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6a/pytest.log
timeout 300 python tools/time_kernels.py geo > gpurun_out/r6a/time_new.log 2>&1
FC_BWD_STREAM=0 timeout 300 python tools/time_kernels.py geo > gpurun_out/r6a/time_old.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r6a/bench.log 2>&1
tail -5 gpurun_out/r6a/pytest.log; cat gpurun_out/r6a/time_new.log gpurun_out/r6a/time_old.log; tail -c 1500 gpurun_out/r6a/bench.log
