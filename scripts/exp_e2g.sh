# E2 general geometry: element-matrix kernel (default) against the brick kernel (FEMASM_E2_BRICK=1)
run() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],4), 'K', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for e in 0 1 0 1; do
  if [ $e = 1 ]; then export FEMASM_E2_BRICK=1; else unset FEMASM_E2_BRICK; fi
  echo -n "brick=$e: "; timeout -k 10 120 python -u bench.py --workload elastic2d --perturb --steps 20 --warmup 3 --no-cpu-baseline "$@" | run || exit 1
done
