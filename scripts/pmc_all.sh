#!/bin/bash
# PMC passes over bench.py; prints per-kernel means for every femasm kernel.  usage: pmc_all.sh [bench args]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run() { name=$1; shift; ctr=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_$name.json 2> $R/gpurun_out/pmc_$name.err || { echo "pass $name failed"; tail -5 $R/gpurun_out/pmc_$name.err; exit 1; }
  python3 - $R/gpurun_out/pmc_$name <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'femasm' in r['Kernel_Name'] and 'pattern' not in r['Kernel_Name']:
            d[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items():
    print(k)
    for c,x in sorted(v.items()): print('   %-28s n=%d mean=%.6g'%(c,len(x),sum(x)/len(x)))
PY
}
run sq1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@" &&
run sq2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" "$@" &&
run wr "WRITE_SIZE" "$@" &&
run rd "FETCH_SIZE" "$@"
