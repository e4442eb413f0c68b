#!/bin/bash
# Build the stand-alone microbenchmarks next to their sources (gfx950).
cd "$(dirname "$0")" || exit 1
for f in walk issue_mix; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -o $f $f.hip 2>&1 | grep -E "error" -A6
done
ls -la walk issue_mix
