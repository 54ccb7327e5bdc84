#!/bin/bash
# enriched K for the builds named in EV (petsc_amd/csrc/exp/libfemasm_<v>.so): node order of k_enr_box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
for v in "" $EV "" $EV; do
lib=$R/petsc_amd/csrc/libfemasm.so; [ -n "$v" ] && lib=$R/petsc_amd/csrc/exp/libfemasm_$v.so
FEMASM_LIB=$lib timeout -k 10 200 python3 bench.py --workload enriched3d --no-cpu-baseline --no-peak --steps 5 --k-only 2>$O/enr_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] K %.3f'%(d['k_ms']))" || { tail $O/enr_bench.err; exit 1; }
done
