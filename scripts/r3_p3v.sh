#!/bin/bash
# Poisson general bench for the profiling builds named in EV (petsc_amd/csrc/exp/libfemasm_p<v>.so, built with
# hipcc ... -DFEMASM_DEV -DPH_EXP=<v> -o exp/libfemasm_p<v>.so femasm.hip -ldl), K only
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
for v in "" $EV; do
lib=$R/petsc_amd/csrc/libfemasm.so; [ -n "$v" ] && lib=$R/petsc_amd/csrc/exp/libfemasm_p$v.so
FEMASM_LIB=$lib timeout -k 10 200 python3 bench.py --workload poisson3d --perturb --no-cpu-baseline --no-peak --no-general --steps 10 --k-only $BARGS 2>$O/p3_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] K %.3f main %.3f level'%(d['k_ms'], d['roofline']['avg_launch_ms']), d['roofline']['kernel_level'])" || { tail $O/p3_bench.err; exit 1; }
done
