"""HBM traffic of the QP kernels per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, only
--kernel-trace next to --pmc), corrected as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes: on gfx950
FETCH_SIZE tallies 64 B per 128-B request -> doubled; WRITE_SIZE is exact.  The factor 2 was re-checked for this kernel's own
access widths (8 and 16 bytes per lane, 208-byte pieces) with scripts/pmc_calib.hip: profiles/r01_pmc_calibration.txt.

usage: pmc_traffic.py <FETCH_SIZE pass .db> <WRITE_SIZE pass .db> <out.json>"""
import json, sqlite3, sys


def avg(db, kernel, counter):
    con = sqlite3.connect(db)
    rows = list(con.execute("select value, duration from counters_collection where kernel_name like ? and counter_name = ?",
                            ('%' + kernel + '%', counter)))
    rows = rows[len(rows) // 2:]      # the steady-state launches (qp_bench: identical inputs, repeated)
    return sum(r[0] for r in rows) / len(rows), sum(r[1] for r in rows) / len(rows) * 1e-9, len(rows)


out = {'workload': 'scripts/qp_bench.py: C1 state after 5 closed-loop steps, B=4096, N=30, identical launches',
       'note': 'FETCH_SIZE doubled (gfx950 tallies 64 B per 128-B request; checked for 8/16 B per lane loads), WRITE_SIZE exact; '
               'separate --pmc passes'}
for k in ('k_qp_ipm', 'k_qp_setup'):
    f_kb, dur_f, n1 = avg(sys.argv[1], k, 'FETCH_SIZE')
    w_kb, dur_w, n2 = avg(sys.argv[2], k, 'WRITE_SIZE')
    byt = (2.0 * f_kb + w_kb) * 1024.0
    out[k] = {'FETCH_SIZE_KB_raw': f_kb, 'WRITE_SIZE_KB': w_kb, 'launches': min(n1, n2), 'traffic_bytes_per_launch': byt,
              'avg_duration_ms': 0.5e3 * (dur_f + dur_w), 'traffic_GBps': byt / (0.5 * (dur_f + dur_w)) / 1e9}
out['kernel'] = 'k_qp_ipm<6,6>'
out['traffic_bytes_per_launch'] = out['k_qp_ipm']['traffic_bytes_per_launch']
out['traffic_GBps_in_pmc_run'] = out['k_qp_ipm']['traffic_GBps']
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
