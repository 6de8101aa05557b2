mkdir -p gpurun_out/r05f
for pad in 0 8192 13824 22016; do
 for fm in 8192 100000000; do
  SMPC_QP_PAD_LDS=$pad SMPC_MLP_FUSED_MAX=$fm python scripts/c4_bench.py 10 2 > gpurun_out/r05f/c4_${pad}_${fm}.json 2>>gpurun_out/r05f/err.txt
  python -c "
import json
d=json.load(open('gpurun_out/r05f/c4_${pad}_${fm}.json')); print('pad', $pad, 'fused_max', $fm, 'ms/step %.2f' % d['ms_per_step'], {k: round(v,2) for k,v in d['kernel_ms_in_loop'].items()})"
 done
done
