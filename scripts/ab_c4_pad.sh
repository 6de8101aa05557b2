#!/bin/bash
# BASELINE config 4 with the QP kernel's residency capped by LDS padding (6 -> 4 -> 3 blocks of k_qp_ipm<7,4> per CU) x the network
# kernel for large row counts (SMPC_MLP_FUSED_MAX: 8192 = the default one-wave kernel, 100000000 = the four-wave fused kernel),
# in ONE gpurun session.  Needs the experiment build (the shipped library reads no SMPC_QP_PAD_LDS):
#   make -C safe_mpc_amd/csrc libsmpc_hip_exp.so
# Round 5: 24.4 -> 26.4 -> 27.4 -> 31.2 ms per step -- the 7-DoF solve needs every slot it can get.
D=$(cd "$(dirname "$0")/.." && pwd)
export SMPC_HIP_LIB=$D/safe_mpc_amd/csrc/libsmpc_hip_exp.so
mkdir -p $D/gpurun_out/c4pad
for pad in 0 8192 13824 22016; do
 for fm in 8192 100000000; do
  SMPC_QP_PAD_LDS=$pad SMPC_MLP_FUSED_MAX=$fm python $D/bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline --no-survey-window > $D/gpurun_out/c4pad/c4_${pad}_${fm}.json 2>>$D/gpurun_out/c4pad/err.txt
  python -c "
import json
d=json.load(open('$D/gpurun_out/c4pad/c4_${pad}_${fm}.json')); print('pad', $pad, 'fused_max', $fm, 'ms/step %.2f' % d['ms_per_step'], {k: round(v,2) for k,v in d['kernel_ms_in_loop'].items()})"
 done
done
