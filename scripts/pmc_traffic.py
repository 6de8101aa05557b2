"""HBM traffic of the QP kernels per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, only
--kernel-trace next to --pmc, CSV output), corrected as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes: on
gfx950 FETCH_SIZE tallies 64 B per 128-B request -> doubled; WRITE_SIZE is exact.  The factor 2 was re-checked for this
kernel's own access widths (8 and 16 bytes per lane) with scripts/pmc_calib.hip: profiles/r01_pmc_calibration.txt.

k_qp_ipm's traffic is proportional to the IPM iterations it runs, so the file also records BYTES PER INSTANCE-ITERATION
(traffic / (instances x mean iterations of the profiled launches)); bench.py multiplies that by its own probe's iterations.

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <instances> <mean iterations> <out.json>"""
import collections, csv, glob, json, sys


def load(d, counter):
    cc = (glob.glob(d + '/*counter_collection.csv') + glob.glob(d + '/*/*counter_collection.csv'))[0]
    dur, val = collections.defaultdict(list), collections.defaultdict(list)
    for r in csv.DictReader(open(cc)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('smpc::', '')
        k = k.replace(' ', '') if k.startswith('k_gemm_f32') else k.split('<')[0]      # (the GEMM's epilogue variants stay apart)
        dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        val[k].append(float(r['Counter_Value']))
    return dur, val


df, vf = load(sys.argv[1], 'FETCH_SIZE')
dw, vw = load(sys.argv[2], 'WRITE_SIZE')
B, iters = int(sys.argv[3]), float(sys.argv[4])
label = sys.argv[6] if len(sys.argv) > 6 else 'C1 state after 5 closed-loop steps, N=30'
kname = sys.argv[7] if len(sys.argv) > 7 else 'k_qp_ipm<6,6>'
key = sys.argv[8] if len(sys.argv) > 8 else 'k_qp_ipm'       # k_qp_ipm (a wavefront per two instances) or k_qp_ipm_wg (a workgroup per instance)
half = lambda v: v[len(v) // 2:]      # the steady-state launches (qp_bench: identical inputs, repeated)
mean = lambda v: sum(v) / len(v)
out = {'workload': 'scripts/qp_bench.py: %s, B=%d, identical launches, mean %.2f IPM iterations' % (label, B, iters),
       'note': 'FETCH_SIZE doubled (gfx950 tallies 64 B per 128-B request; checked for 8/16 B per lane loads), WRITE_SIZE exact; '
               'separate --pmc passes; counter units are KB'}
for k in [k_ for k_ in sorted(vf) if k_ in vw and (k_.startswith('k_qp') or k_.startswith('k_stage') or k_.startswith('k_mlp') or k_.startswith('k_gemm') or k_.startswith('k_nn') or k_ == 'k_node_linearise')]:
    f_kb, w_kb = mean(half(vf[k])), mean(half(vw[k]))
    dur = 0.5 * (mean(half(df[k])) + mean(half(dw[k]))) * 1e-9
    byt = (2.0 * f_kb + w_kb) * 1024.0
    out[k] = {'FETCH_SIZE_KB_raw': f_kb, 'WRITE_SIZE_KB': w_kb, 'launches': min(len(half(vf[k])), len(half(vw[k]))),
              'read_bytes_per_launch': 2.0 * f_kb * 1024.0, 'write_bytes_per_launch': w_kb * 1024.0,
              'traffic_bytes_per_launch': byt, 'avg_duration_ms': dur * 1e3, 'traffic_GBps': byt / dur / 1e9}
out['kernel'] = kname
out['instances'], out['mean_iterations'] = B, iters
out['traffic_bytes_per_launch'] = out[key]['traffic_bytes_per_launch']
out['bytes_per_instance_iteration'] = out[key]['traffic_bytes_per_launch'] / (B * iters)
out['traffic_GBps_in_pmc_run'] = out[key]['traffic_GBps']
json.dump(out, open(sys.argv[5], 'w'), indent=1)
print(json.dumps(out, indent=1))
