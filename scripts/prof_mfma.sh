#!/bin/bash
# MFMA utilisation of the network pass's kernels: one rocprofv3 --pmc pass (SQ_INSTS_VALU_MFMA_MOPS_F32; only --kernel-trace beside
# it, the program directly after `--`) over scripts/qp_bench.py on BASELINE config 4's problem (row on every node: 5461 x 40 network
# rows per launch), kernels alone on the GPU; TFLOP/s = MOPS_F32 x 512 flop / kernel duration (the counter's definition), against the
# dense fp32 MFMA peak of 157.3 TFLOP/s.  "In the loop": the same MOPS over the kernel's average duration inside bench.py --config c4's
# three-stream loop (rocprofv3 --kernel-trace of the loop: profiles/rNN_c4_kernel_summary_by_grid.txt).
#   usage: prof_mfma.sh <outdir-name> [trace summary of the loop]
set -e
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/$1
LOOP=${2:-}
mkdir -p $O
cd /tmp
export SMPC_QPB_PROBLEM=fr7
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/m -- python3 $R/scripts/qp_bench.py > $O/qp_bench_m.txt 2> $O/err_m.txt
python3 - <<PY
import csv, glob, collections, re
cc = (glob.glob('$O/m/*counter_collection.csv') + glob.glob('$O/m/*/*counter_collection.csv'))[0]
val, dur, grid = collections.defaultdict(list), collections.defaultdict(list), {}
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'SQ_INSTS_VALU_MFMA_MOPS_F32':
        continue
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('smpc::', '').replace(' ', '')
    val[k].append(float(r['Counter_Value'])); dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp'])); grid[k] = r.get('Grid_Size', '')
loop = {}
if '$LOOP':
    for line in open('$LOOP'):          # scripts/trace_summary.py: kernel, blocks, calls, avg_us, min_us, max_us, %time
        p_ = line.rsplit(None, 6)
        if len(p_) == 7 and p_[1].isdigit():
            loop.setdefault(p_[0].replace('smpc::', '').replace(' ', ''), float(p_[3]))
with open('$O/mfma_counters.txt', 'w') as f:
    f.write('# rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace over SMPC_QPB_PROBLEM=fr7 scripts/qp_bench.py (C4 problem, 5461 instances x 40 network rows per launch);\n')
    f.write('# TFLOP/s = MOPS_F32 x 512 flop / kernel duration; dense fp32 MFMA peak 157.3 TFLOP/s.  ' + open('$O/qp_bench_m.txt').read())
    f.write('%-34s %10s %8s %10s %12s %9s %8s %14s %10s\n' % ('kernel', 'grid', 'launches', 'avg_us', 'MOPS_F32', 'TFLOP/s', '% peak', 'in-loop avg_us', '% in loop'))
    for k in sorted(val):
        if max(val[k]) <= 0:
            continue
        h = len(val[k]) // 2
        v, d = sum(val[k][h:]) / len(val[k][h:]), sum(dur[k][h:]) / len(dur[k][h:])
        tf = v * 512 / (d * 1e-9) / 1e12
        lu = loop.get(k)
        f.write('%-34s %10s %8d %10.1f %12.4g %9.1f %7.1f%% %14s %10s\n' % (k, grid[k], len(val[k]), d / 1e3, v, tf, 100 * tf / 157.3,
                ('%.1f' % lu) if lu else '-', ('%.1f%%' % (100 * v * 512 / (lu * 1e-6) / 1e12 / 157.3)) if lu else '-'))
print(open('$O/mfma_counters.txt').read())
PY
rm -rf $O/m
