#!/bin/bash
# Poisson general: node layers per workgroup
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
for zc in 0 16 22 32 43 64 128; do
opt=""; [ $zc != 0 ] && opt="--option zc=$zc"
timeout -k 10 200 python3 bench.py --workload poisson3d --perturb --no-cpu-baseline --no-peak --no-general --steps 10 --k-only $opt 2>$O/p3_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('zc $zc K %.3f main %.3f'%(d['k_ms'], d['roofline']['avg_launch_ms']))" || { tail $O/p3_bench.err; exit 1; }
done
