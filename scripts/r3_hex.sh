#!/bin/bash
# round 3: the level-1 elastic kernel (k_assemble_s3_hex): its tests, then the perturbed bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "hexahedron or general_geometry or sheared or moved_interior or kernel_levels or asynchronous_assembly_is_final or element_soa" > $O/hex_tests.log 2>&1; rc=$?
tail -15 $O/hex_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python3 bench.py --perturb --no-cpu-baseline --no-peak > $O/bench_hex.json 2> $O/bench_hex.err || { tail -20 $O/bench_hex.err; exit 1; }
cat $O/bench_hex.json
