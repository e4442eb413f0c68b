#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
o=gpurun_out/r6b
WHICH=b timeout 300 python tools/time_kernels.py geo > $o/t_prod.log 2>&1
for d in 0 2 4 6 16 22 32 64 96; do
  FC_DEBUG_BWD=$d WHICH=b timeout 300 python tools/time_kernels.py geo 2>&1 | grep "gather\|stream" > $o/t_dbg$d.log
done
FC_STAMP_KERNEL=stream timeout 300 python tools/stamps.py stream --wave 0 8 --tiles 3 --warm 50 > $o/stamps.log 2>&1
grep -h "gather\|stream" $o/t_prod.log; for d in 0 2 4 6 16 22 32 64 96; do echo "dbg $d"; cat $o/t_dbg$d.log; done; cat $o/stamps.log | tail -80
