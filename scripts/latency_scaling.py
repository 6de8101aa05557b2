#!/usr/bin/env python3
"""Two measurements VERDICT r5 asked for (items 2, "what's missing" 2 and 3), both on ONE GPU:

(a) the strong-scaling PROXY of the headline metric: instances are independent and the only collective is one gather per rollout
    (DESIGN section 6), so the per-GPU time of the 4096-instance job on G GPUs is the one-GPU time of its per-GPU share.
    `bench.py --batch {4096, 2048, 1024, 512}` on SURVEY 8(d)'s window (10 + 100 steps) -> predicted speed-up at 1 / 2 / 4 / 8 GPUs.
(b) the latency of ONE solve for a small batch -- the statistic the reference itself prints (99 % quantile of the per-step solver
    time of one instance, scripts/mpc.py:300-303; budget dt = 5 ms, config.yaml:7): the `HipOcpSolver` drop-in (B = 1, numpy in and
    out as acados has it) and `BatchedOcpSolver` at B = 1 / 8 / 64 with device-resident inputs, N = 10 (BASELINE config 0) and
    N = 30, p50 / p99 over a 200-step closed loop, next to the CPU port's one-thread figure on the same states.

Usage (GPU box):  python scripts/latency_scaling.py [--out-dir profiles] [--tag r06] [--skip-proxy] [--skip-latency]
Writes <tag>_strong_scaling_proxy.json and <tag>_latency.json.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def proxy(batches, extra_env=None):
    rows = []
    for b in batches:
        env = dict(os.environ, **(extra_env or {}))
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--batch', str(b), '--no-cpu-baseline', '--no-loop-timing', '--no-latency']
        t0 = time.time()
        out = subprocess.run(cmd, env=env, capture_output=True, text=True)
        if out.returncode != 0:
            sys.stderr.write(out.stderr[-2000:])
            raise SystemExit(f'bench.py --batch {b} failed')
        line = json.loads(out.stdout.strip().splitlines()[-1])
        rows.append({'batch': b, 'ms_per_step': line['ms_per_step'], 'instance_steps_per_s': line['value'],
                     'mean_ipm_iterations': line['config']['mean_ipm_iterations'], 'streams': line['config']['streams_per_gpu'],
                     'failed_instance_steps': line['config']['failed_instance_steps'], 'wall_s': time.time() - t0})
        print('[proxy]', rows[-1], flush=True)
    t_full = rows[0]['ms_per_step']
    for r in rows:
        r['gpus_at_4096_total'] = round(batches[0] / r['batch'], 2)      # (1536 = the share on 2.67 GPUs: a point of the curve, not a node size)
        r['predicted_strong_speedup'] = t_full / r['ms_per_step']
        r['predicted_strong_efficiency'] = r['predicted_strong_speedup'] / r['gpus_at_4096_total']
    return rows


def latency(n_steps=200):
    import bench
    from safe_mpc_amd.acados_facade import HipOcpSolver
    out = []
    for N in (10, 30):
        states = []
        for r in bench.small_batch_latency(N=N, batches=(1, 8, 64), n_steps=n_steps, keep_states=states):
            r['solver'] = 'BatchedOcpSolver (device-resident inputs, wall clock around solve + sync)'
            out.append(r)
            print('[latency]', r, flush=True)
        # the drop-in facade: numpy in / out through the C ABI's host path, as AcadosOcpSolver is used at controller.py:141-165
        par, prob, net = bench.build_problem(N=N, controller='st')
        hs = HipOcpSolver(prob, net, device=0)
        lat = []
        for k, (x, xg, ug, p) in enumerate(states):
            t0 = time.perf_counter()
            hs.reset()
            hs.constraints_set(0, 'lbx', x[0]); hs.constraints_set(0, 'ubx', x[0])
            for i in range(N):
                hs.set(i, 'x', xg[0, i]); hs.set(i, 'u', ug[0, i]); hs.set(i, 'p', p[0, i])
            hs.set(N, 'x', xg[0, N]); hs.set(N, 'p', p[0, N])
            t1 = time.perf_counter()
            hs.solve()
            t2 = time.perf_counter()
            if k >= 5:
                lat.append([t2 - t1, t2 - t0])
        lat = np.array(lat) * 1e3
        row = {'solver': 'HipOcpSolver (drop-in of AcadosOcpSolver, host arrays, B = 1)', 'N': N, 'B': 1,
               'p50_ms': float(np.quantile(lat[:, 0], 0.5)), 'p99_ms': float(np.quantile(lat[:, 0], 0.99)),
               'p50_ms_with_set_calls': float(np.quantile(lat[:, 1], 0.5)), 'solves': len(lat)}
        out.append(row)
        print('[latency]', row, flush=True)
        # the CPU port on ONE thread over the same closed-loop states (bench.py's cpu_baseline leg; a restatement, not acados)
        row = bench.cpu_baseline_latency(prob, net, states[5:])
        row.update({'solver': 'CPU port (oracle/, one thread; a restatement, not acados)', 'N': N, 'B': 1})
        out.append(row)
        print('[latency]', row, flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out-dir', default=os.path.join(ROOT, 'gpurun_out'))
    ap.add_argument('--tag', default='r06')
    ap.add_argument('--skip-proxy', action='store_true')
    ap.add_argument('--skip-latency', action='store_true')
    ap.add_argument('--batches', default='4096,2048,1024,512')
    args = ap.parse_args()
    os.makedirs(args.out_dir, exist_ok=True)
    if not args.skip_proxy:
        rows = proxy([int(b) for b in args.batches.split(',')])
        json.dump({'what': 'one-GPU proxy of the strong-scaling curve of the headline metric (4096 Z1 N=30 instances in total): '
                           'bench.py --batch <per-GPU share>, 10 warm-up + 100 timed steps; instances are independent and the only '
                           'collective is one gather per rollout, so the G-GPU step time is the one-GPU time of the share',
                   'rows': rows}, open(os.path.join(args.out_dir, f'{args.tag}_strong_scaling_proxy.json'), 'w'), indent=1)
    if not args.skip_latency:
        rows = latency()
        json.dump({'what': 'per-solve latency of a small batch over a 200-step closed loop (controller st, EXT cost), the statistic of '
                           'scripts/mpc.py:300-303; budget dt = 5 ms (config.yaml:7)', 'rows': rows},
                  open(os.path.join(args.out_dir, f'{args.tag}_latency.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
