#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
o=gpurun_out/r6c
FC_STAMP_KERNEL=stream timeout 300 python tools/stamps.py stream --wave 0 8 --tiles 4 --warm 50 > $o/stamps.log 2>&1
FC_DEBUG_BWD=22 FC_STAMP_KERNEL=stream timeout 300 python tools/stamps.py stream --wave 0 8 --tiles 4 --warm 50 > $o/stamps22.log 2>&1
cat $o/stamps.log | tail -90; tail -60 $o/stamps22.log
