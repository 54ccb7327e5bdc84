# E2 general geometry: element-matrix kernel (default) against the brick kernel (--option e2_brick=1) and other tile shapes
run() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],4), 'K', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
B="python -u bench.py --workload elastic2d --perturb --steps 20 --warmup 3 --no-cpu-baseline"
for r in 1 2; do
for v in "" _t31 _t7; do
  [ -f petsc_amd/csrc/libfemasm$v.so ] || continue
  echo -n "lib '$v': "; FEMASM_LIB=$PWD/petsc_amd/csrc/libfemasm$v.so timeout -k 10 120 $B "$@" | run || exit 1
done
done
echo -n "brick: "; timeout -k 10 120 $B --option e2_brick=1 "$@" | run
