#!/bin/bash
# A/B of two builds of the engine inside ONE gpurun session (boxes differ by up to 10 %): alternates
# safe_mpc_amd/csrc/libsmpc_hip_base.so and libsmpc_hip.so.   usage: ab_bench.sh [rounds] [extra bench args]
R=${1:-2}; shift
D=$(cd "$(dirname "$0")/.." && pwd)
one() { SMPC_HIP_LIB=$1 python $D/bench.py --no-cpu-baseline --no-loop-timing --steps 40 --warmup 5 "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d.get('survey_window') or {}; r=d['roofline']['kernel_ms']
print('%.3f ms/step  survey %.3f  probe: lin %.3f setup %.3f ipm %.3f  it %.2f' % (d['ms_per_step'], s.get('ms_per_step',0), r['linearise'], r['qp_setup'], r['qp_ipm'], d['config']['mean_ipm_iterations']))"; }
for i in $(seq $R); do
  echo "base: $(one $D/safe_mpc_amd/csrc/libsmpc_hip_base.so "$@")"
  echo "new : $(one $D/safe_mpc_amd/csrc/libsmpc_hip.so "$@")"
done
