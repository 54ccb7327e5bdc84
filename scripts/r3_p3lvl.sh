#!/bin/bash
# Poisson 256^3 on the uniform mesh at the three kernel levels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
for lv in 2 1 0 1 0; do
timeout -k 10 200 python3 bench.py --workload poisson3d --no-cpu-baseline --no-peak --no-general --steps 10 --fast-level $lv 2>$O/p3_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('level $lv: step %.3f K %.3f rhs %.3f'%(d['ms_per_step'], d['k_ms'], d['rhs_ms']), d['roofline']['kernel'], 'main %.3f'%d['roofline']['avg_launch_ms'])" || { tail $O/p3_bench.err; exit 1; }
done
