#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
for zc in $ZCS; do
FEMASM_LIB=$R/petsc_amd/csrc/exp/libfemasm_hx9.so timeout -k 10 120 python3 bench.py --perturb --k-only --no-cpu-baseline --no-peak --steps 10 --option zc=$zc > $O/bench_hex_9.json 2> $O/bench_hex_9.err
grep "femasm hex" $O/bench_hex_9.err
python3 -c "import json,sys; d=json.load(open('$O/bench_hex_9.json')); print('zc=$zc', d['roofline']['kernel'], 'main_ms %.3f'%d['roofline']['avg_launch_ms'])"
done
