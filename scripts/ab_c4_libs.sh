#!/bin/bash
# BASELINE config 4 over several builds of the engine in ONE gpurun session: the 7-DoF solve alone (scripts/qp_bench.py) and the
# three-stream loop (bench.py --config c4).  usage: ab_c4_libs.sh <rounds> lib1.so lib2.so ...
R=$1; shift
D=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq $R); do for l in "$@"; do
  a=$(SMPC_HIP_LIB=$D/safe_mpc_amd/csrc/$l SMPC_QPB_PROBLEM=fr7 python $D/scripts/qp_bench.py 2>/dev/null | tail -1)
  b=$(SMPC_HIP_LIB=$D/safe_mpc_amd/csrc/$l python $D/bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline --no-survey-window 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('loop %.2f ms/step' % d['ms_per_step'], {k: round(v,2) for k,v in (d['roofline'].get('kernel_ms_in_loop') or {}).items() if k in ('mlp','qp_ipm','linearise')}, 'it %.2f' % d['config']['mean_ipm_iterations'])")
  echo "$l: $a | $b"
done; done
