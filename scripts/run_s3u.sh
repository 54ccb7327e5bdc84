# full GPU suite, then the headline step with the unit kernel and with the row-sweep kernel (--option s3_rowsweep=1)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/s3u_test.log 2>&1 || { tail -30 gpurun_out/s3u_test.log; exit 1; }
tail -2 gpurun_out/s3u_test.log
for rep in 1 2; do
for e in 0 1; do
  timeout -k 10 120 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --option s3_rowsweep=$e | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rowsweep=$e step', round(d['ms_per_step'],3), 'K', round(d['roofline']['avg_launch_ms'],3), 'frac', round(d['roofline']['frac'],3), 'rhs', round(d['rhs_ms'],3))"
done
done
