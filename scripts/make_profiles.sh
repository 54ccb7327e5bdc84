#!/bin/bash
# Round profiles of the default bench command: kernel-trace stats, then WRITE_SIZE and FETCH_SIZE in separate
# --pmc passes.  Output: gpurun_out/profiles/<tag>_*  (copy what is to be kept into profiles/<round>/).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-elastic3d}; shift
O=$R/gpurun_out/profiles; mkdir -p $O
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_trace -- python3 $R/bench.py "$@" > $O/${tag}_bench.json 2> $O/${tag}_trace.err || { echo "stats run failed"; tail -5 $O/${tag}_trace.err; exit 1; }
if grep -E "SIGSEGV|Aborted|Segmentation fault|core dumped" $O/${tag}_trace.err; then echo "stats run: a process of the profiled command crashed (see ${tag}_trace.err)"; exit 1; fi
cp $(find $O/${tag}_trace -name '*kernel_stats.csv' | head -1) $O/${tag}_kernel_stats.csv
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $O/${tag}_pmc_$c -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $O/${tag}_pmc_$c.json 2> $O/${tag}_pmc_$c.err || { echo "pmc $c failed"; tail -5 $O/${tag}_pmc_$c.err; exit 1; }
  if grep -E "SIGSEGV|Aborted|Segmentation fault|core dumped" $O/${tag}_pmc_$c.err; then echo "pmc $c: a process crashed"; exit 1; fi
  python3 - $O/${tag}_pmc_$c $c > $O/${tag}_pmc_$c.txt <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']==sys.argv[2] and 'femasm' in r['Kernel_Name']:
            d[r['Kernel_Name']].append(float(r['Counter_Value']))
for k,v in sorted(d.items()): print('%s,%s,%d,%.6g'%(sys.argv[2],k.replace(',',';'),len(v),sum(v)/len(v)))
PY
  cat $O/${tag}_pmc_$c.txt
done
rm -rf $O/${tag}_trace $O/${tag}_pmc_WRITE_SIZE $O/${tag}_pmc_FETCH_SIZE
cat $O/${tag}_bench.json
