#!/bin/bash
# Development: libfieldconv_hip.so variants with -DFC_ROLES_EXP=n (fc_backward_roles.hpp) beside the normal library, selected at
# run time with FIELDCONV_HIP_LIB.  Usage (CPU): bash tools/build_variants.sh 1 2 3 4
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/fieldconv_amd/csrc
OUT=$ROOT/fieldconv_amd/_native
TMP=$(mktemp -d)
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -fno-gpu-rdc -Wno-unused-result -w"
for f in $CSRC/*.hip; do
  b=$(basename $f .hip)
  [ $b = fc_backward_ring ] && continue
  /opt/rocm/bin/hipcc $FLAGS -c -o $TMP/$b.o $f &
done
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DFC_ROLES_EXP=$n -c -o $TMP/ring_$n.o $CSRC/fc_backward_ring.hip &
done
wait
for n in "$@"; do
  objs=$(ls $TMP/*.o | grep -v ring_)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o $OUT/exp_$n.so $objs $TMP/ring_$n.o && echo built $OUT/exp_$n.so
done
rm -rf $TMP
