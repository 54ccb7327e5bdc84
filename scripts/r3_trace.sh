#!/bin/bash
# kernel-trace stats of one bench command (arguments = bench arguments); prints the per-kernel table
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp
rm -rf $O/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --no-peak --steps 10 "$@" > $O/trace.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
f=$(find $O/trace -name '*kernel_stats.csv' | head -1)
python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-70s calls %5s mean_us %10.1f  pct %5s'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
