#!/bin/bash
# enriched kind, level 1 (general hexahedra, one material per element): tests, a fuzz run, the perturbed bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_enriched.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/enr_tests.log 2>&1 || { tail -30 $O/enr_tests.log; exit 1; }
tail -1 $O/enr_tests.log
timeout -k 10 200 python3 -u tests/fuzz_gpu.py --kinds enr --seconds 120 --seed 77 > $O/enr1_fuzz.log 2>&1 || { grep -v "^ok" $O/enr1_fuzz.log | tail -20; exit 1; }
tail -1 $O/enr1_fuzz.log
for pert in "--perturb" "--perturb --fast-level 0" ""; do
timeout -k 10 200 python3 bench.py --workload enriched3d --no-cpu-baseline --no-peak --steps 5 $pert 2>$O/enr_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$pert step %.3f K %.3f rhs %.3f level'%(d['ms_per_step'], d['k_ms'], d['rhs_ms']), d['roofline']['kernel_level'], d['roofline']['kernel'], 'frac %.3f'%d['roofline']['frac'])" || { tail $O/enr_bench.err; exit 1; }
done
