"""Per-kernel sheet of one solve (scripts/qp_bench.py launches): average duration, HBM bytes per launch and the rate they
imply, from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output, --kernel-trace only next to --pmc).
FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (profiles/r01_pmc_calibration.txt);
WRITE_SIZE is exact.  usage: kernel_sheet.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>"""
import collections, csv, glob, sys


def load(d, counter):
    cc = glob.glob(d + '/*/*counter_collection.csv')[0]
    dur, val = collections.defaultdict(list), collections.defaultdict(list)
    for r in csv.DictReader(open(cc)):
        if r['Counter_Name'] != counter:
            continue
        k = (r['Kernel_Name'].split('(')[0].replace('void ', '').replace('smpc::', '')[:44], int(r['Grid_Size']))
        dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        val[k].append(float(r['Counter_Value']))
    return dur, val


df, vf = load(sys.argv[1], 'FETCH_SIZE')
dw, vw = load(sys.argv[2], 'WRITE_SIZE')
half = lambda v: v[len(v) // 2:]           # the steady-state launches
print('%-46s %9s %6s %10s %10s %10s %9s' % ('kernel', 'grid', 'calls', 'avg_us', 'read_MB', 'write_MB', 'GB/s'))
rows = []
for k in vf:
    if k not in vw or not vf[k] or not vw[k]:
        continue
    f = sum(half(vf[k])) / len(half(vf[k])) * 2.0 * 1024.0
    w = sum(half(vw[k])) / len(half(vw[k])) * 1024.0
    d = 0.5 * (sum(half(df[k])) / len(half(df[k])) + sum(half(dw[k])) / len(half(dw[k])))
    rows.append((d * len(vf[k]), k, len(vf[k]), d, f, w))
for tot, k, n, d, f, w in sorted(rows, reverse=True):
    if d < 2000 and f + w < 1e6:
        continue
    print('%-46s %9d %6d %10.1f %10.1f %10.1f %9.0f' % (k[0], k[1], n, d / 1e3, f / 1e6, w / 1e6, (f + w) / d))
