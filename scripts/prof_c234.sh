set -x
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r4c
mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -- python3 $R/scripts/c4_bench.py 10 2 > $O/c4_bench_under_rocprof.json 2> $O/c4.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python3 $R/scripts/c3_bench.py 20 3 > $O/c3_bench_under_rocprof.json 2> $O/c3.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python3 $R/bench.py --batch 65536 --noise 10 --control-noise 1 --steps 10 --warmup 2 --no-cpu-baseline --no-survey-window > $O/c2_bench_under_rocprof.json 2> $O/c2.err || exit 1
for c in c2 c3 c4; do
  f=$(ls $O/$c/*/*kernel_trace.csv | head -1)
  python3 $R/scripts/trace_summary.py $f > $O/${c}_kernel_summary_by_grid.txt
  cp $(ls $O/$c/*/*kernel_stats.csv | head -1) $O/${c}_kernel_stats.csv
  rm -rf $O/$c
done
# the same three without the profiler (the numbers to quote)
cd $R
timeout -k 10 200 python3 scripts/c4_bench.py 10 2 > $O/c4_bench.json 2>> $O/c4.err
timeout -k 10 200 python3 scripts/c3_bench.py 20 3 > $O/c3_bench.json 2>> $O/c3.err
timeout -k 10 300 python3 bench.py --batch 65536 --noise 10 --control-noise 1 --steps 10 --warmup 2 --no-cpu-baseline --no-survey-window > $O/c2_bench.json 2>> $O/c2.err
ls -la $O
