#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for d in 0 64 2 66 10 74 11; do
  FC_DEBUG_BWD=$d WHICH=b python3 tools/time_kernels.py geo 2>&1 | grep "bwd_data"
done
