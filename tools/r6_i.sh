#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
o=gpurun_out/r6i
for d in 0 6 22; do
FIELDCONV_DEV=1 FC_DEBUG_BWD=$d FC_STAMP_KERNEL=stream timeout 300 python tools/stamps.py stream --wave 0 3 8 14 --tiles 3 --warm 50 > $o/stamps$d.log 2>&1
done
