#!/bin/bash
# run a -k selection of the GPU parity tests without -x; prints the failures
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout -k 10 500 python -u -m pytest ${TFILES:-tests/test_gpu_parity.py} -m gpu -q --timeout 120 --timeout-method thread -k "$1" > $O/t.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $O/t.log | cut -c1-220
