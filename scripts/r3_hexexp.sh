#!/bin/bash
# k_assemble_s3_hex: tests, then K-only timings of the shipped build and of the phase-removal builds (HX_EXP)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "hexahedron or general_geometry or sheared or moved_interior or kernel_levels or asynchronous_assembly_is_final or element_soa" > $O/hex_tests.log 2>&1; rc=$?
tail -3 $O/hex_tests.log
[ $rc -ne 0 ] && { tail -40 $O/hex_tests.log; exit $rc; }
for v in "" $HXV; do
  lib=$R/petsc_amd/csrc/libfemasm.so; [ -n "$v" ] && lib=$R/petsc_amd/csrc/exp/libfemasm_hx$v.so
  FEMASM_LIB=$lib timeout -k 10 120 python3 bench.py --perturb --k-only --no-cpu-baseline --no-peak --steps 10 > $O/bench_hex_$v.json 2> $O/bench_hex_$v.err || { tail -20 $O/bench_hex_$v.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$O/bench_hex_$v.json')); print('variant [$v]', d['roofline']['kernel'], 'level', d['roofline']['kernel_level'], 'main_ms %.3f'%d['roofline']['avg_launch_ms'], 'k_ms %.3f'%d['k_ms'])"
done
