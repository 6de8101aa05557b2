#!/bin/bash
# Issue-side counters of the solve kernels (one rocprofv3 --pmc pass per counter, only --kernel-trace beside it, the program directly
# after `--`): VALU / LDS / memory instructions per launch and the busy / wait cycles they sit in.  usage: prof_valu.sh <outdir-name> [env for qp_bench]
set -e
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
for c in SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python3 $R/scripts/qp_bench.py > $O/qp_bench_$c.txt 2> $O/err_$c.txt || echo "counter $c failed"
done
python3 - <<PY
import csv, glob, collections
out = collections.defaultdict(dict)
dur = {}
for d in sorted(glob.glob('$O/*/')):
    files = glob.glob(d + '*/*counter_collection.csv') + glob.glob(d + '*counter_collection.csv')
    if not files: continue
    vals, durs = collections.defaultdict(list), collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('smpc::', '').split('<')[0]
        vals[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
        durs[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for (k, c), v in vals.items():
        h = v[len(v) // 2:]
        out[k][c] = sum(h) / len(h)
    for k, v in durs.items():
        h = v[len(v) // 2:]
        dur[k] = sum(h) / len(h)
with open('$O/issue_counters.txt', 'w') as f:
    f.write(open('$O/qp_bench_SQ_INSTS_VALU.txt').read())
    for k in ('k_qp_ipm', 'k_stage_build', 'k_mlp_fused'):
        if k in out:
            f.write('%s  (avg launch %.1f us)\n' % (k, dur.get(k, 0) / 1e3))
            for c, v in sorted(out[k].items()):
                f.write('    %-24s %.4g\n' % (c, v))
print(open('$O/issue_counters.txt').read())
PY
rm -rf $O/SQ_* 
