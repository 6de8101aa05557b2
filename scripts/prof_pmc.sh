#!/bin/bash
# HBM traffic of the solve kernels: two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over scripts/qp_bench.py -- only
# --kernel-trace next to --pmc, the program directly after `--` -- then scripts/pmc_traffic.py.   usage: prof_pmc.sh <outdir-name>
set -e
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/scripts/qp_bench.py > $O/qp_bench_f.txt 2> $O/err_f.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $R/scripts/qp_bench.py > $O/qp_bench_w.txt 2> $O/err_w.txt
cat $O/qp_bench_f.txt
IT=$(python3 -c "import re,sys; print(re.search(r'iters mean ([0-9.]+)', open('$O/qp_bench_f.txt').read()).group(1))")
python3 $R/scripts/pmc_traffic.py $O/f $O/w 4096 $IT $O/pmc_traffic.json > /dev/null
python3 $R/scripts/kernel_sheet.py $O/f $O/w > $O/kernel_sheet.txt || true
rm -rf $O/f $O/w
python3 -c "import json; d=json.load(open('$O/pmc_traffic.json')); print('bytes per instance-iteration', d['bytes_per_instance_iteration']); print({k: (round(v['read_bytes_per_launch']/1e9,3), round(v['write_bytes_per_launch']/1e9,3), round(v['avg_duration_ms'],3)) for k,v in d.items() if isinstance(v, dict)})"
cat $O/kernel_sheet.txt
