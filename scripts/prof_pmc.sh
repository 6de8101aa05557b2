#!/bin/bash
# HBM traffic of the solve kernels: two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over scripts/qp_bench.py -- only
# --kernel-trace next to --pmc, the program directly after `--` -- then scripts/pmc_traffic.py.
#   usage: prof_pmc.sh <outdir-name> [case: c1 (default) | c1nt | c1plain | c4 | wg]
#   wg    512 instances of C1 (the per-GPU share of the 4096-instance headline on 8 GPUs): k_qp_ipm_wg, the latency form (round 6)
#   c1    C1's whole-batch launch (4096 instances, Z1, N = 30), workspace-size policy -> nt variant from round 5 on (733 MB > 384 MB)
#   c1nt  the same with SMPC_QP_NT=1 forced / c1plain with SMPC_QP_NT=0 forced (A/B of the two access variants)
#   c4    BASELINE config 4's problem (7-DoF, N = 40, row on every node) at one sub-batch launch of bench.py --config c4 (5461 instances)
set -e
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/$1
CASE=${2:-c1}
B=4096; LABEL="C1 state after 5 closed-loop steps, Z1, N=30"; KN="k_qp_ipm<6,6,true>"; KEY=k_qp_ipm
case $CASE in
  c1) ;;
  c1nt) export SMPC_QP_NT=1; LABEL="$LABEL, non-temporal variant forced" ;;
  c1plain) export SMPC_QP_NT=0; LABEL="$LABEL, plain-access variant forced"; KN="k_qp_ipm<6,6,false>" ;;
  wg) export SMPC_B=512; B=512; LABEL="C1 state after 5 closed-loop steps, Z1, N=30, one GPU's share of the headline on 8 GPUs (latency form, a workgroup per instance)"; KN="k_qp_ipm_wg<6,6,4>"; KEY=k_qp_ipm_wg ;;
  c4) export SMPC_QPB_PROBLEM=fr7; B=5461; LABEL="C4 problem (7-DoF, N=40, NN row on every node) after 5 closed-loop steps, one sub-batch launch of bench.py --config c4"; KN="k_qp_ipm<7,4,true>" ;;
esac
mkdir -p $O
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/scripts/qp_bench.py > $O/qp_bench_f.txt 2> $O/err_f.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $R/scripts/qp_bench.py > $O/qp_bench_w.txt 2> $O/err_w.txt
cat $O/qp_bench_f.txt
IT=$(python3 -c "import re,sys; print(re.search(r'iters mean ([0-9.]+)', open('$O/qp_bench_f.txt').read()).group(1))")
python3 $R/scripts/pmc_traffic.py $O/f $O/w $B $IT $O/pmc_traffic.json "$LABEL" "$KN" $KEY > /dev/null
python3 $R/scripts/kernel_sheet.py $O/f $O/w > $O/kernel_sheet.txt || true
rm -rf $O/f $O/w
python3 -c "import json; d=json.load(open('$O/pmc_traffic.json')); print('bytes per instance-iteration', d['bytes_per_instance_iteration']); print({k: (round(v['read_bytes_per_launch']/1e9,3), round(v['write_bytes_per_launch']/1e9,3), round(v['avg_duration_ms'],3)) for k,v in d.items() if isinstance(v, dict)})"
cat $O/kernel_sheet.txt
