"""HBM traffic of k_qp_ipm per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), with the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE counts half the bytes of wide coalesced reads -> doubled.
usage: pmc_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <out.json>"""
import csv, glob, json, sys

def avg(d, counter):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'k_qp_ipm' in r['Kernel_Name'] and r['Counter_Name'] == counter]
    return sum(v) / len(v), len(v)

fetch_kb, n1 = avg(sys.argv[1], 'FETCH_SIZE')
write_kb, n2 = avg(sys.argv[2], 'WRITE_SIZE')
out = {'kernel': 'k_qp_ipm<6,6>', 'workload': 'C1 closed loop, B=4096, N=30, 6 launches (scripts/profile_solve.py)',
       'FETCH_SIZE_KB_raw': fetch_kb, 'WRITE_SIZE_KB': write_kb, 'launches': min(n1, n2),
       'traffic_bytes_per_launch': (2.0 * fetch_kb + write_kb) * 1024.0,
       'note': 'FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request on 16-B/lane streams); separate --pmc passes'}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(out)
