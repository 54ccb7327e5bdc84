# E2 level-2 kernel: node rows per wave (FEMASM_E2_YC), whole step and K kernel at 2048^2
for yc in 1 2 4 8 16 32 64; do
  echo -n "yc $yc: "
  timeout -k 10 120 python -u bench.py --workload elastic2d --steps 20 --warmup 3 --no-cpu-baseline --option e2_yc=$yc "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],4), 'K', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3))" || exit 1
done
