#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py (program directly after --), per-grid summary.  usage: prof_bench.sh <outdir-name> [bench args]
set -e
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/bench_under_rocprof.json 2> $O/err.txt
f=$(ls $O/raw/*/*kernel_trace.csv | head -1)
python3 $R/scripts/trace_summary.py $f > $O/kernel_summary_by_grid.txt
cp $(ls $O/raw/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/raw
cat $O/kernel_summary_by_grid.txt
