#!/bin/bash
# enriched kind: its tests, then the bench (K ms, rhs ms, kernel level)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
if [ "$1" != "notests" ]; then
timeout -k 10 300 python -u -m pytest tests/test_gpu_enriched.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/enr_tests.log 2>&1 || { tail -30 $O/enr_tests.log; exit 1; }
tail -1 $O/enr_tests.log
fi
for i in 1 2; do
timeout -k 10 200 python3 bench.py --workload enriched3d --no-cpu-baseline --no-peak --steps 5 $BARGS 2>$O/enr_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step %.3f K %.3f rhs %.3f level'%(d['ms_per_step'], d['k_ms'], d['rhs_ms']), d['roofline']['kernel_level'], 'frac %.3f'%d['roofline']['frac'])" || { tail $O/enr_bench.err; exit 1; }
done
