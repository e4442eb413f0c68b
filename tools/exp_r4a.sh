#!/bin/bash
# Round 4, experiment A (GPU box): 24-bit row offsets and issue priority by remaining slots, against the previous library.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out
python3 -c "import __graft_entry__; __graft_entry__.build(); __graft_entry__.smoke()" 2>&1 | tail -2
echo "== base library"
FIELDCONV_HIP_LIB=$ROOT/fieldconv_amd/_native/libfc_base.so python3 tools/time_kernels.py geo 2>&1 | grep median
echo "== mad24"
python3 tools/time_kernels.py geo 2>&1 | grep median
echo "== mad24 + dynamic priority"
FC_DEBUG=8 FC_DEBUG_BWD=32 python3 tools/time_kernels.py geo 2>&1 | grep median
echo "== no static priority in fwd"
FC_DEBUG=4 WHICH=f python3 tools/time_kernels.py geo 2>&1 | grep median
echo "== base again"
FIELDCONV_HIP_LIB=$ROOT/fieldconv_amd/_native/libfc_base.so python3 tools/time_kernels.py geo 2>&1 | grep median
echo "== mad24 again"
python3 tools/time_kernels.py geo 2>&1 | grep median
echo "== dyn again"
FC_DEBUG=8 FC_DEBUG_BWD=32 python3 tools/time_kernels.py geo 2>&1 | grep median
