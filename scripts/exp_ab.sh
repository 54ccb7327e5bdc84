# same-box A/B of two builds over bench workloads: libfemasm.so (tree) against libfemasm_exp8.so (an earlier commit)
# usage: bash scripts/exp_ab.sh "--workload elastic2d" "--workload elastic2d --k-only" ...
run() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],4), 'K', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
if [ $# = 0 ]; then set -- "--workload elastic3d" "--workload poisson3d" "--workload elastic2d" "--workload elastic3d --perturb" "--workload elastic2d --perturb"; fi
for args in "$@"; do
for v in "" _exp8 "" _exp8; do
  echo -n "$args lib '$v': "; FEMASM_LIB=$PWD/petsc_amd/csrc/libfemasm$v.so timeout -k 10 120 python -u bench.py $args --steps 20 --warmup 3 --no-cpu-baseline | run || exit 1
done
done
