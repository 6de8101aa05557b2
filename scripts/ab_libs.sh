#!/bin/bash
# bench.py over several builds of the engine in ONE gpurun session.  usage: ab_libs.sh <rounds> lib1.so lib2.so ... [-- bench args]
R=$1; shift
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" == "--" ] && shift
D=$(cd "$(dirname "$0")/.." && pwd)
one() { SMPC_HIP_LIB=$1 python $D/bench.py --no-cpu-baseline --no-loop-timing --steps 40 --warmup 5 "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d.get('survey_window') or {}; r=d['roofline']['kernel_ms']
print('%.3f ms/step  survey %.3f  probe: lin %.3f mlp %.3f setup %.3f ipm %.3f  it %.2f' % (d['ms_per_step'], s.get('ms_per_step',0), r['linearise'], r['mlp'], r['qp_setup'], r['qp_ipm'], d['config']['mean_ipm_iterations']))"; }
for i in $(seq $R); do for l in "${LIBS[@]}"; do echo "$(basename $l): $(one $D/safe_mpc_amd/csrc/$l "$@")"; done; done
