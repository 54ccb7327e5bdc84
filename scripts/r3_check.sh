#!/bin/bash
# round 3: GPU tests, the default bench line (with its general_geometry sub-record), the bare N-rank launch rehearsed
# with two gloo ranks on the one card
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
if [ "$1" != "notests" ]; then
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/tests.log 2>&1; rc=$?
tail -3 $O/tests.log
[ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
cat $O/bench_default.json
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --share-gpu --nel 48 --steps 3 --warmup 1 > $O/bench_g2.json 2> $O/bench_g2.err || { tail -20 $O/bench_g2.err; exit 1; }
cat $O/bench_g2.json
