set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_enriched.py -x -q --timeout 120 --timeout-method thread > gpurun_out/enr_node_test.log 2>&1 || { tail -30 gpurun_out/enr_node_test.log; exit 1; }
tail -3 gpurun_out/enr_node_test.log
timeout -k 10 200 python bench.py --workload enriched3d --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/enr_node_bench.log 2>&1 && tail -1 gpurun_out/enr_node_bench.log
timeout -k 10 200 python bench.py --workload enriched3d --enr-contraction valu --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/enr_valu_bench.log 2>&1 && tail -1 gpurun_out/enr_valu_bench.log
