#!/bin/bash
# E2 level 2 (k_assemble_e2_rect): tests that reach it, a fuzz run, the bench with the record stream and with lane loads
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
if [ "$1" != "notests" ]; then
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 150 --timeout-method thread -k "e2 or E2 or plane or elastic2d or config2" > $O/e2_tests.log 2>&1 || { tail -30 $O/e2_tests.log; exit 1; }
tail -1 $O/e2_tests.log
timeout -k 10 200 python3 -u tests/fuzz_gpu.py --kinds e2 --seconds 100 --seed 93 > $O/e2_fuzz.log 2>&1 || { grep -v "^ok" $O/e2_fuzz.log | tail -20; exit 1; }
tail -1 $O/e2_fuzz.log
fi
for v in "" "--option e2_lane_loads=1" "" "--option e2_lane_loads=1" "--k-only" "--k-only --option e2_lane_loads=1"; do
timeout -k 10 200 python3 bench.py --workload elastic2d --no-cpu-baseline --no-peak --no-general --steps 20 $v $BARGS 2>$O/e2_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$v] step %.4f K %.4f rhs %.4f'%(d['ms_per_step'], d['k_ms'], d['rhs_ms']), d['roofline']['kernel'], 'main %.4f frac %.3f'%(d['roofline']['avg_launch_ms'], d['roofline']['frac']))" || { tail $O/e2_bench.err; exit 1; }
done
