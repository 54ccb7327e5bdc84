#!/bin/bash
# rocprofv3 kernel-trace stats of one bench.py run:  prof_stats.sh <name> [bench args...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1; shift
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$name -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$name.json 2> $R/gpurun_out/prof_$name.err || { echo "profile $name failed"; tail -5 $R/gpurun_out/prof_$name.err; exit 1; }
f=$(find $R/gpurun_out/prof_$name -name '*kernel_stats.csv' | head -1)
cp $f $R/gpurun_out/prof_${name}_kernel_stats.csv
python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-90s calls %5s avg %10.1f us  %5s%%'%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
