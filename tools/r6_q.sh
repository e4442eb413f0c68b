#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "two_arithmetic" 2>&1 | grep -v "^$" | tail -12
