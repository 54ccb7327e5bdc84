#!/bin/bash
# one-line summaries of the three workloads
for w in elastic3d poisson3d elastic2d; do
  timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $w "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s step %.3f ms  K %.3f (main %.3f) ms (%.0f GB/s, frac %.3f)  rhs %.3f ms  %.3g el/s'%('$w', d['ms_per_step'], d['k_ms'], r['avg_launch_ms'], r['achieved'], r['frac'], d['rhs_ms'], d['value']))" || exit 1
done
