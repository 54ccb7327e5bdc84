# K kernels with the compact per-element material (FEMASM_MAT_ELEMENT_SOA) against the drivers' AoS records, K only
run() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],4), 'ms', round(d['roofline']['frac'],3), d['config']['material'][:30])"; }
for wl in elastic3d elastic2d; do
for f in "" --element-soa "" --element-soa; do
  echo -n "$wl $f: "; timeout -k 10 120 python -u bench.py --workload $wl --steps 10 --warmup 2 --k-only --no-cpu-baseline $f | run || exit 1
done
done
echo -n "elastic3d general: "; timeout -k 10 120 python -u bench.py --workload elastic3d --perturb --steps 5 --warmup 2 --k-only --no-cpu-baseline | run
echo -n "elastic3d general soa: "; timeout -k 10 120 python -u bench.py --workload elastic3d --perturb --steps 5 --warmup 2 --k-only --no-cpu-baseline --element-soa | run
