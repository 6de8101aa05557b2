set -e
mkdir -p gpurun_out/r6f
run() { # batch streams form
  timeout -k 10 120 python bench.py --batch $1 --streams $2 --qp-form $3 --no-cpu-baseline --no-loop-timing --no-survey-window --no-latency 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B $1 S $2 $3: %.3f ms  %.0f inst-steps/s  its %.2f' % (d['ms_per_step'], d['value'], d['config']['mean_ipm_iterations']))" | tee -a gpurun_out/r6f/sweep.txt
}
run 1536 3 latency
run 1536 4 latency
run 1536 6 latency
run 2048 4 latency
run 2048 6 latency
run 2048 8 latency
run 2048 3 throughput
run 2048 2 throughput
run 2048 4 throughput
run 1024 2 latency
run 1024 4 latency
run 768 2 latency
run 768 3 latency
