#!/bin/bash
# k_assemble_s3_hex and its successors: K-only timings of the shipped build and of the profiling builds named in HXV
# (petsc_amd/csrc/exp/libfemasm_hx<v>.so); the phase-clock build prints its clocks on stderr
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
for v in "" $HXV; do
  lib=$R/petsc_amd/csrc/libfemasm.so; [ -n "$v" ] && lib=$R/petsc_amd/csrc/exp/libfemasm_hx$v.so
  FEMASM_LIB=$lib timeout -k 10 120 python3 bench.py --perturb --k-only --no-cpu-baseline --no-peak --steps 10 $BARGS > $O/bench_hex_$v.json 2> $O/bench_hex_$v.err || { tail -20 $O/bench_hex_$v.err; exit 1; }
  grep "femasm hex" $O/bench_hex_$v.err
  python3 -c "import json,sys; d=json.load(open('$O/bench_hex_$v.json')); print('variant [$v]', d['roofline']['kernel'], 'level', d['roofline']['kernel_level'], 'main_ms %.3f'%d['roofline']['avg_launch_ms'], 'k_ms %.3f'%d['k_ms'])"
done
