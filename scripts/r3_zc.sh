#!/bin/bash
# K-only timing of the perturbed elastic bench for several values of an option
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
for z in $ZCS; do
  timeout -k 10 120 python3 bench.py --perturb --k-only --no-cpu-baseline --no-peak --steps 10 --option ${OPT:-zc}=$z $BARGS > $O/bench_zc.json 2> $O/bench_zc.err || { tail -20 $O/bench_zc.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$O/bench_zc.json')); print('${OPT:-zc}=$z', d['roofline']['kernel'], 'main_ms %.3f'%d['roofline']['avg_launch_ms'], 'k_ms %.3f'%d['k_ms'])"
done
