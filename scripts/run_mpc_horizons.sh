#!/bin/bash
# horizon sweep of the reference (scripts/run_mpc_horizons.sh:19-34): one batched run per horizon
cont=${1:-st}
for N in 20 25 30 35 40; do
  python "$(dirname "$0")/guess_acados.py" -c "$cont" --horizon "$N" && python "$(dirname "$0")/mpc.py" -c "$cont" --horizon "$N"
done
