set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_enriched.py -x -q --timeout 120 --timeout-method thread > gpurun_out/enr_node_test.log 2>&1 || { tail -30 gpurun_out/enr_node_test.log; exit 1; }
tail -3 gpurun_out/enr_node_test.log
bash scripts/exp_enr.sh "$1"
