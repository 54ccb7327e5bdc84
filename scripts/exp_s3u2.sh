# the elastic level-2 K kernels at other mesh sizes: unit kernel (default) against the row-sweep kernel
run() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],4), 'ms', round(d['roofline']['frac'],3))"; }
for nel in 32 64 96 160 192; do
B="python -u bench.py --steps 10 --warmup 2 --k-only --no-cpu-baseline --nel $nel"
echo -n "nel $nel unit: "; timeout -k 10 120 $B | run || exit 1
echo -n "nel $nel rowsweep: "; FEMASM_S3_ROWSWEEP=1 timeout -k 10 120 $B | run || exit 1
done
