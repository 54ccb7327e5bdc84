#!/bin/bash
# full-step timing of experiment builds
for v in "" _exp1 _exp2 _exp3 _exp4 _exp5 _exp7 _exp8; do
  L=$PWD/petsc_amd/csrc/libfemasm$v.so
  [ -f $L ] || continue
  echo -n "variant '$v': "
  FEMASM_LIB=$L timeout -k 10 120 python -u bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K', round(d['roofline']['avg_launch_ms'],3), 'rhs', round(d['rhs_ms'],3))" || exit 1
done
