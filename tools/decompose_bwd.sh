#!/bin/bash
# Phase decomposition of the backward kernels by switching phases off (FC_DEBUG_BWD bits: 1 gather, 2 data-kernel MFMA,
# 4 filter-kernel MFMA, 8 slab copy).  Run on the GPU box: bash tools/decompose_bwd.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
python3 -c "import __graft_entry__; __graft_entry__.build()" || exit 1
for d in 0 1 2 8 3 10 11 4; do
  FC_DEBUG_BWD=$d WHICH=b python3 tools/time_kernels.py geo 2>&1 | grep "bwd"
done
