#!/bin/bash
# PMC passes over the K-only bench (one pass per run, as rocprofv3 does not split counters)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run() { # name counters...
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -- python3 $R/bench.py --steps 3 --warmup 1 $KONLY --no-cpu-baseline $EXTRA > $R/gpurun_out/pmc_$name.json 2> $R/gpurun_out/pmc_$name.err || { echo "pass $name failed"; tail -5 $R/gpurun_out/pmc_$name.err; exit 1; }
  python3 - $R/gpurun_out/pmc_$name <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        import os
        if os.environ.get('KPAT','k_assemble') in r['Kernel_Name']:
            d[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items():
    print(k)
    for c,x in sorted(v.items()): print('   %-28s n=%d mean=%.6g'%(c,len(x),sum(x)/len(x)))
PY
}
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INST_CYCLES_VMEM &&
run wr WRITE_SIZE &&
run rd FETCH_SIZE &&
run grbm GRBM_GUI_ACTIVE
