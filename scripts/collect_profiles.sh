#!/bin/bash
# Everything profiles/r06_* is made from, in ONE gpurun session (copy gpurun_out/r06/* into profiles/ as r06_<name> afterwards):
#   pmc_traffic.json / _c4.json / _wg.json  two rocprofv3 --pmc passes each (FETCH_SIZE / WRITE_SIZE) over scripts/qp_bench.py: C1's whole-batch
#                                      launch (k_qp_ipm, nt variant by workspace size), BASELINE config 4's problem at one sub-batch launch, and
#                                      512 instances of C1 through k_qp_ipm_wg (the latency form, round 6)
#   bench.json                         python bench.py                                  (the line the driver records)
#   bench_kernel_stats.csv, bench_kernel_summary_by_grid.txt, bench_under_rocprof.json    rocprofv3 --kernel-trace --stats of the same
#   bench_b512*.json / _kernel_*       the same for bench.py --batch 512 (one GPU's share of the headline on 8 GPUs: the latency form's regime)
#   c2_*, c3_*, c4_*                   BASELINE configs 2-4 through the one launcher: bench lines and kernel traces
#   strong_scaling_proxy.json, latency.json   scripts/latency_scaling.py
#   wg_phase_profile_B*.txt, qp_phase_profile.txt, lat_probe.txt, policy_bench.txt
#   `collect_profiles.sh wg`: only the files that depend on k_qp_ipm_wg (pmc_traffic_wg, strong_scaling_proxy, latency, bench.json --
#   which embeds the two --, bench_b512*, wg_phase_profile_B*, form_sweep, bench_800steps_latency_form) -- re-collected after the last change of that kernel
set -x
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r06
rm -rf $O; mkdir -p $O
cd $R
if [ "$1" = wg ]; then
  bash scripts/prof_pmc.sh r06/pmc_wg wg > $O/pmc_wg.log 2>&1 && cp $O/pmc_wg/pmc_traffic.json $O/pmc_traffic_wg.json && cp $O/pmc_wg/kernel_sheet.txt $O/kernel_sheet_wg.txt
  cp $O/pmc_traffic_wg.json $R/profiles/r06_pmc_traffic_wg.json
  python3 scripts/latency_scaling.py --out-dir $O --tag r06 --batches 4096,2048,1536,1024,512 > $O/latency_scaling.log 2>&1
  mv $O/r06_strong_scaling_proxy.json $O/strong_scaling_proxy.json; mv $O/r06_latency.json $O/latency.json
  cp $O/strong_scaling_proxy.json $R/profiles/r06_strong_scaling_proxy.json
  timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
  timeout -k 10 300 python3 bench.py --batch 512 --no-latency > $O/bench_b512.json 2>> $O/bench.err
  bash scripts/prof_bench.sh r06/benchprof512 --batch 512 --no-latency > /dev/null 2>&1
  cp $O/benchprof512/kernel_stats.csv $O/bench_b512_kernel_stats.csv; cp $O/benchprof512/kernel_summary_by_grid.txt $O/bench_b512_kernel_summary_by_grid.txt; cp $O/benchprof512/bench_under_rocprof.json $O/bench_b512_under_rocprof.json
  SMPC_B=1 python3 scripts/qp_wg_phase_profile.py > $O/wg_phase_profile_B1.txt 2>&1
  SMPC_B=256 python3 scripts/qp_wg_phase_profile.py > $O/wg_phase_profile_B256.txt 2>&1
  mkdir -p gpurun_out/r6f; rm -f gpurun_out/r6f/sweep.txt; bash scripts/form_sweep.sh > /dev/null 2>&1; cp gpurun_out/r6f/sweep.txt $O/form_sweep.txt
  for b in 512 1024 1536; do timeout -k 10 150 python3 bench.py --batch $b --steps 800 --warmup 10 --no-cpu-baseline --no-loop-timing --no-survey-window --no-latency 2>> $O/bench.err | grep '^{' >> $O/bench_800steps_latency_form.jsonl; done
  rm -rf $O/pmc_wg $O/benchprof512
  ls -la $O
  exit 0
fi
bash scripts/prof_pmc.sh r06/pmc c1 > $O/pmc.log 2>&1 && cp $O/pmc/pmc_traffic.json $O/pmc_traffic.json && cp $O/pmc/kernel_sheet.txt $O/kernel_sheet.txt
bash scripts/prof_pmc.sh r06/pmc_wg wg > $O/pmc_wg.log 2>&1 && cp $O/pmc_wg/pmc_traffic.json $O/pmc_traffic_wg.json && cp $O/pmc_wg/kernel_sheet.txt $O/kernel_sheet_wg.txt
bash scripts/prof_pmc.sh r06/pmc_c4 c4 > $O/pmc_c4.log 2>&1 && cp $O/pmc_c4/pmc_traffic.json $O/pmc_traffic_c4.json
# (the bench lines read profiles/r06_pmc_traffic*.json: make this session's files visible to them)
cp $O/pmc_traffic.json $R/profiles/r06_pmc_traffic.json; cp $O/pmc_traffic_c4.json $R/profiles/r06_pmc_traffic_c4.json; cp $O/pmc_traffic_wg.json $R/profiles/r06_pmc_traffic_wg.json
python3 scripts/latency_scaling.py --out-dir $O --tag r06 --batches 4096,2048,1536,1024,512 > $O/latency_scaling.log 2>&1
mv $O/r06_strong_scaling_proxy.json $O/strong_scaling_proxy.json; mv $O/r06_latency.json $O/latency.json
cp $O/strong_scaling_proxy.json $R/profiles/r06_strong_scaling_proxy.json
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
bash scripts/prof_bench.sh r06/benchprof > /dev/null 2>&1
cp $O/benchprof/kernel_stats.csv $O/bench_kernel_stats.csv; cp $O/benchprof/kernel_summary_by_grid.txt $O/bench_kernel_summary_by_grid.txt; cp $O/benchprof/bench_under_rocprof.json $O/bench_under_rocprof.json
timeout -k 10 300 python3 bench.py --batch 512 --no-latency > $O/bench_b512.json 2>> $O/bench.err
bash scripts/prof_bench.sh r06/benchprof512 --batch 512 --no-latency > /dev/null 2>&1
cp $O/benchprof512/kernel_stats.csv $O/bench_b512_kernel_stats.csv; cp $O/benchprof512/kernel_summary_by_grid.txt $O/bench_b512_kernel_summary_by_grid.txt; cp $O/benchprof512/bench_under_rocprof.json $O/bench_b512_under_rocprof.json
cd /tmp
for c in c2 c3 c4; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$c -- python3 $R/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-survey-window > $O/${c}_bench_under_rocprof.json 2> $O/$c.err
  f=$(ls $O/$c/*/*kernel_trace.csv | head -1)
  python3 $R/scripts/trace_summary.py $f > $O/${c}_kernel_summary_by_grid.txt
  cp $(ls $O/$c/*/*kernel_stats.csv | head -1) $O/${c}_kernel_stats.csv
  rm -rf $O/$c
done
cd $R
timeout -k 10 400 python3 bench.py --config c2 --steps 10 --warmup 2 --no-survey-window > $O/c2_bench.json 2>> $O/c2.err
timeout -k 10 300 python3 bench.py --config c3 --steps 20 --warmup 3 --no-survey-window > $O/c3_bench.json 2>> $O/c3.err
timeout -k 10 300 python3 bench.py --config c4 --steps 10 --warmup 2 --no-survey-window > $O/c4_bench.json 2>> $O/c4.err
SMPC_BENCH_CONTROLLER=constraint_everywhere timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-latency > $O/bench_constraint_everywhere.json 2>> $O/bench.err
SMPC_B=1 python3 scripts/qp_wg_phase_profile.py > $O/wg_phase_profile_B1.txt 2>&1
SMPC_B=256 python3 scripts/qp_wg_phase_profile.py > $O/wg_phase_profile_B256.txt 2>&1
SMPC_B=256 python3 scripts/qp_phase_profile.py > $O/qp_phase_profile.txt 2>&1
./build/lat_probe > $O/lat_probe.txt 2>&1
SMPC_STEPS=60 timeout -k 10 400 python3 scripts/policy_bench.py st htwa receding real_receding > $O/policy_bench.txt 2>&1
SMPC_WARM=1 SMPC_SQP_ITERS=40 SMPC_STEPS=60 timeout -k 10 500 python3 scripts/policy_bench.py st htwa receding real_receding >> $O/policy_bench.txt 2>&1
rm -rf $O/pmc $O/pmc_c4 $O/pmc_wg $O/benchprof $O/benchprof512
ls -la $O
