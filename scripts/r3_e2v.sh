#!/bin/bash
# E2 bench for the builds named in EV (petsc_amd/csrc/exp/libfemasm_<v>.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
for v in "" $EV "" $EV; do
lib=$R/petsc_amd/csrc/libfemasm.so; [ -n "$v" ] && lib=$R/petsc_amd/csrc/exp/libfemasm_$v.so
FEMASM_LIB=$lib timeout -k 10 200 python3 bench.py --workload elastic2d --no-cpu-baseline --no-peak --no-general --steps 20 $BARGS 2>$O/e2_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] step %.4f K %.4f main %.4f'%(d['ms_per_step'], d['k_ms'], d['roofline']['avg_launch_ms']))" || { tail $O/e2_bench.err; exit 1; }
done
