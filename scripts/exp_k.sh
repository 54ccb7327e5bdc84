#!/bin/bash
# time the K kernel of experiment builds: prints avg_launch_ms per variant
for v in "" _exp1 _exp2 _exp3; do
  L=$PWD/petsc_amd/csrc/libfemasm$v.so
  [ -f $L ] || continue
  echo -n "variant '$v': "
  FEMASM_LIB=$L timeout -k 10 120 python -u bench.py --steps 5 --warmup 1 --k-only --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_ms'], 'ms', d['roofline']['frac'])" || exit 1
done
