#!/bin/bash
# randomised GPU-vs-oracle parity over all kinds with the final build (tests/fuzz_gpu.py), incl. multi-slab split calls
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout -k 10 200 python3 -u tests/fuzz_gpu.py --kinds p3,e2,s3 --seconds 150 --seed 2026 > $O/fuzz_mixed.log 2>&1 || { grep -v "^ok" $O/fuzz_mixed.log | tail -20; exit 1; }
tail -1 $O/fuzz_mixed.log
timeout -k 10 150 python3 -u tests/fuzz_gpu.py --kinds p3,e2,s3 --seconds 90 --seed 2027 --split-only > $O/fuzz_split.log 2>&1 || { grep -v "^ok" $O/fuzz_split.log | tail -20; exit 1; }
tail -1 $O/fuzz_split.log
timeout -k 10 150 python3 -u tests/fuzz_gpu.py --kinds enr --seconds 90 --seed 2028 > $O/fuzz_enr.log 2>&1 || { grep -v "^ok" $O/fuzz_enr.log | tail -20; exit 1; }
tail -1 $O/fuzz_enr.log
