"""Per-kernel, per-grid-size summary of a rocprofv3 --kernel-trace CSV (bench.py launches k_qp_ipm at two sizes: the timed
closed loop runs it per sub-batch stream, the roofline probe once over the whole batch -- rocprofv3's own --stats averages
the two together).  usage: trace_summary.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    acc[(name, int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in acc.values())
print('%-58s %8s %6s %11s %11s %11s %7s' % ('kernel', 'blocks', 'calls', 'avg_us', 'min_us', 'max_us', '%time'))
for (name, blocks), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) < 0.002 * tot:
        continue
    print('%-58s %8d %6d %11.1f %11.1f %11.1f %6.2f%%' % (name[:58], blocks, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot))
