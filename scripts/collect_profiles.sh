#!/bin/bash
# Everything profiles/r05_* is made from, in ONE gpurun session (copy gpurun_out/r05/* into profiles/ as r05_<name> afterwards):
#   pmc_traffic.json / _c4.json / _c1plain.json, kernel_sheet*.txt   two rocprofv3 --pmc passes each (FETCH_SIZE / WRITE_SIZE) over scripts/qp_bench.py:
#                                      C1's whole-batch launch (nt variant by workspace size), the same with plain accesses forced, and
#                                      BASELINE config 4's problem (7-DoF, N = 40, row on every node) at one sub-batch launch
#   mfma_counters.txt                  SQ_INSTS_VALU_MFMA_MOPS_F32 over the network pass of config 4 (k_mlp_wave, the default for large row counts)
#   bench.json                         python bench.py                                  (the line the driver records)
#   bench_kernel_stats.csv, bench_kernel_summary_by_grid.txt, bench_under_rocprof.json    rocprofv3 --kernel-trace --stats of the same
#   c2_*, c3_*, c4_*                   BASELINE configs 2-4: bench lines (with roofline + in-loop kernel split) and kernel traces
#   policy_bench.txt                   run_mpc(on_device=True) per controller
set -x
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r05
rm -rf $O; mkdir -p $O
cd $R
bash scripts/prof_pmc.sh r05/pmc c1 > $O/pmc.log 2>&1 && cp $O/pmc/pmc_traffic.json $O/pmc_traffic.json && cp $O/pmc/kernel_sheet.txt $O/kernel_sheet.txt
bash scripts/prof_pmc.sh r05/pmc_c4 c4 > $O/pmc_c4.log 2>&1 && cp $O/pmc_c4/pmc_traffic.json $O/pmc_traffic_c4.json && cp $O/pmc_c4/kernel_sheet.txt $O/kernel_sheet_c4.txt
bash scripts/prof_pmc.sh r05/pmc_c1plain c1plain > $O/pmc_c1plain.log 2>&1 && cp $O/pmc_c1plain/pmc_traffic.json $O/pmc_traffic_c1plain.json
# (the bench lines read profiles/r05_pmc_traffic*.json: make this session's files visible to them)
cp $O/pmc_traffic.json $R/profiles/r05_pmc_traffic.json; cp $O/pmc_traffic_c4.json $R/profiles/r05_pmc_traffic_c4.json
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
bash scripts/prof_bench.sh r05/benchprof > /dev/null 2>&1
cp $O/benchprof/kernel_stats.csv $O/bench_kernel_stats.csv; cp $O/benchprof/kernel_summary_by_grid.txt $O/bench_kernel_summary_by_grid.txt; cp $O/benchprof/bench_under_rocprof.json $O/bench_under_rocprof.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -- python3 $R/scripts/c4_bench.py 10 2 > $O/c4_bench_under_rocprof.json 2> $O/c4.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python3 $R/scripts/c3_bench.py 20 3 > $O/c3_bench_under_rocprof.json 2> $O/c3.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python3 $R/bench.py --batch 65536 --noise 10 --control-noise 1 --steps 10 --warmup 2 --no-cpu-baseline --no-survey-window > $O/c2_bench_under_rocprof.json 2> $O/c2.err
for c in c2 c3 c4; do
  f=$(ls $O/$c/*/*kernel_trace.csv | head -1)
  python3 $R/scripts/trace_summary.py $f > $O/${c}_kernel_summary_by_grid.txt
  cp $(ls $O/$c/*/*kernel_stats.csv | head -1) $O/${c}_kernel_stats.csv
  rm -rf $O/$c
done
cd $R
bash scripts/prof_mfma.sh r05/mfma $O/c4_kernel_summary_by_grid.txt > $O/mfma.log 2>&1 && cp $O/mfma/mfma_counters.txt $O/mfma_counters.txt
timeout -k 10 200 python3 scripts/c4_bench.py 10 2 > $O/c4_bench.json 2>> $O/c4.err
timeout -k 10 200 python3 scripts/c3_bench.py 20 3 > $O/c3_bench.json 2>> $O/c3.err
timeout -k 10 300 python3 bench.py --batch 65536 --noise 10 --control-noise 1 --steps 10 --warmup 2 --no-cpu-baseline --no-survey-window > $O/c2_bench.json 2>> $O/c2.err
SMPC_BENCH_CONTROLLER=constraint_everywhere timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench_constraint_everywhere.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_20steps.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --graphs 1 --no-cpu-baseline > $O/bench_graphs.json 2>> $O/bench.err
SMPC_STEPS=60 timeout -k 10 400 python3 scripts/policy_bench.py st htwa receding real_receding > $O/policy_bench.txt 2>&1
timeout -k 10 300 python3 scripts/rollout_bench.py > $O/rollout_bench.txt 2>/dev/null
rm -rf $O/pmc $O/pmc_c4 $O/pmc_c1plain $O/benchprof $O/mfma
ls -la $O
