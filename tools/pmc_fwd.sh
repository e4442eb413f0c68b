#!/bin/bash
# SQ counters of the convolution kernels (three passes of 8 counters); run on the GPU box: bash tools/pmc_fwd.sh <tag> [env...]
# The library is built FIRST, by a process no profiler has touched: under rocprofv3 the preloaded tool library has
# initialised the GPU before python starts, and a build from there would start hipcc children that exec clang -- the
# exec hop this pool forbids (bench.py itself refuses to build under a profiler).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__; __graft_entry__.build()" || exit 1
cd /tmp && export TMPDIR=/tmp
for pass in 1 2 3; do
  case $pass in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA";;
    2) C="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES";;
    3) C="SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_IFETCH";;
  esac
  rm -rf "/tmp/pmc_${TAG}_$pass"
  env "$@" rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$pass -o p -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras --steps 12 --warmup 3 > /dev/null 2>&1
done
python3 "$ROOT/tools/pmc_summary.py" /tmp/pmc_${TAG}_1 /tmp/pmc_${TAG}_2 /tmp/pmc_${TAG}_3 --filter fc_forward
