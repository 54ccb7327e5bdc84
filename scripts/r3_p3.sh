#!/bin/bash
# Poisson kind on general hexahedra (k_assemble_p3_hex): the parity tests that reach it, a fuzz run, the perturbed bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
if [ "$1" != "notests" ]; then
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q --timeout 120 --timeout-method thread -k "p3 or poisson or P3 or general or perturb" > $O/p3_tests.log 2>&1 || { tail -30 $O/p3_tests.log; exit 1; }
tail -1 $O/p3_tests.log
timeout -k 10 200 python3 -u tests/fuzz_gpu.py --kinds p3 --seconds 100 --seed 91 > $O/p3_fuzz.log 2>&1 || { grep -v "^ok" $O/p3_fuzz.log | tail -20; exit 1; }
tail -1 $O/p3_fuzz.log
fi
for v in "--perturb" "--perturb --option p3_hex2w=1" "--perturb" "--perturb --option p3_hex2w=1" "--perturb --option p3_sweep=2"; do
timeout -k 10 200 python3 bench.py --workload poisson3d --no-cpu-baseline --no-peak --no-general --steps 10 $v $BARGS 2>$O/p3_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v step %.3f K %.3f rhs %.3f level'%(d['ms_per_step'], d['k_ms'], d['rhs_ms']), d['roofline']['kernel_level'], d['roofline']['kernel'], 'main %.3f frac %.3f'%(d['roofline']['avg_launch_ms'], d['roofline']['frac']))" || { tail $O/p3_bench.err; exit 1; }
done
