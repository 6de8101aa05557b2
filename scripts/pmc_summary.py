"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv (+ durations from kernel_trace.csv)."""
import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + '/*/*counter_collection.csv')[0]
kt = glob.glob(d + '/*/*kernel_trace.csv')[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[r['Kernel_Name'].split('(')[0][-40:]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    vals[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
pick = sys.argv[2] if len(sys.argv) > 2 else 'qp_ipm'
for k in vals:
    if pick not in k:
        continue
    print(k, 'calls', len(dur[k]), 'avg_us', sum(dur[k]) / len(dur[k]) / 1e3, 'per-call us', [round(x / 1e3) for x in dur[k]])
    for c, v in vals[k].items():
        print('   %-24s avg %.4g   per-call %s' % (c, sum(v) / len(v), ['%.3g' % x for x in v]))
