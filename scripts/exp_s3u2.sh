run() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],4), 'ms', round(d['roofline']['frac'],3))"; }
B="python -u bench.py --steps 10 --warmup 2 --k-only --no-cpu-baseline"
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "unit_and_rowsweep or system_matches" --timeout 120 --timeout-method thread 2>&1 | tail -1
echo -n "unit: "; timeout -k 10 120 $B | run || exit 1
echo -n "unit nostore: "; FEMASM_LIB=$PWD/petsc_amd/csrc/libfemasm_exp1.so timeout -k 10 120 $B | run || exit 1
echo -n "rowsweep: "; FEMASM_S3_ROWSWEEP=1 timeout -k 10 120 $B | run || exit 1
