#!/bin/bash
# alpha sweep of the reference (scripts/run_mpc_alphas.sh:19-34): one batched run per safety margin
cont=${1:-st}
for A in 20 30 40 50; do
  python "$(dirname "$0")/guess_acados.py" -c "$cont" --alpha "$A" && python "$(dirname "$0")/mpc.py" -c "$cont" --alpha "$A"
done
