#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
o=gpurun_out/r6h
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config2 or whole_tensors or small_cotangent" > $o/pytest.log 2>&1
echo "pytest rc=$?" >> $o/pytest.log
WHICH=b timeout 300 python tools/time_kernels.py geo > $o/t_prod.log 2>&1
for d in 2 4 6 16 22; do
  FIELDCONV_DEV=1 FC_DEBUG_BWD=$d WHICH=b timeout 300 python tools/time_kernels.py geo 2>&1 | grep "gather\|stream" > $o/t_dbg$d.log
done
FIELDCONV_DEV=1 FC_STAMP_KERNEL=stream timeout 300 python tools/stamps.py stream --wave 0 8 --tiles 4 --warm 50 > $o/stamps.log 2>&1
tail -4 $o/pytest.log; grep -h "gather\|stream\|checksum" $o/t_prod.log; for d in 2 4 6 16 22; do echo "dbg $d"; cat $o/t_dbg$d.log; done; cat $o/stamps.log | tail -75
