# P3 general geometry: element-matrix kernel (default) against the node-row kernel (--option p3_sweep=1)
B="python -u bench.py --workload poisson3d --perturb --steps 10 --warmup 2 --no-cpu-baseline"
run() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],4), 'K', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for r in 1 2; do
echo -n "ke: "; timeout -k 10 120 $B "$@" | run || exit 1
echo -n "sweep: "; timeout -k 10 120 $B --option p3_sweep=1 "$@" | run || exit 1
done
