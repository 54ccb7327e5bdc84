# A/B of the elastic level-2 unit kernel: x-consecutive units per wave (FEMASM_S3_UNITS) against the row-sweep kernel
for nu in 1 2 4 8 16; do
  echo -n "units/wave $nu: "
  timeout -k 10 120 python -u bench.py --steps 10 --warmup 2 --k-only --no-cpu-baseline --option s3_units=$nu "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_ms'], 'ms', d['roofline']['frac'])" || exit 1
done
echo -n "rowsweep: "
timeout -k 10 120 python -u bench.py --steps 10 --warmup 2 --k-only --no-cpu-baseline --option s3_rowsweep=1 "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_ms'], 'ms', d['roofline']['frac'])"
