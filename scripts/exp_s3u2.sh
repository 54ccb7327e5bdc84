run() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],4), 'ms', round(d['roofline']['frac'],3))"; }
B="python -u bench.py --steps 10 --warmup 2 --k-only --no-cpu-baseline"
for v in "" _exp1 _exp2 "" _exp1 _exp2; do
echo -n "unit copy-out variant '$v': "; FEMASM_LIB=$PWD/petsc_amd/csrc/libfemasm$v.so timeout -k 10 120 $B | run || exit 1
done
