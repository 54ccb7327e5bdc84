#!/bin/bash
# time K of the enriched workload for the profiling builds libfemasm_exp<N>.so that exist: exp_enr.sh "11 12 13" [bench args]
vs=$1; shift
for v in "" $vs; do
  L=$PWD/petsc_amd/csrc/libfemasm${v:+_exp$v}.so
  [ -f $L ] || continue
  echo -n "variant '$v': "
  FEMASM_LIB=$L timeout -k 10 120 python -u bench.py --workload enriched3d --steps 5 --warmup 1 --k-only --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_ms'], 'ms', d['roofline']['frac'])" || exit 1
done
