for d in 0 3 0 1 2 3; do
  echo "dup=$d $(SMPC_DUP_KERNELS=$d python bench.py --no-cpu-baseline --no-loop-timing --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['survey_window']['ms_per_step'])")"
done
