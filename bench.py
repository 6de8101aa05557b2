#!/usr/bin/env python3
"""bench.py -- RTI-MPC throughput of the HIP engine on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W

One "step" = one closed-loop RTI-MPC step of the whole batch, all inputs resident in HBM:
<Controller>.step (guessCorrection -> SQP-RTI solve: linearise, MLP, QP -> accept test -> provideControl) -> plant step
(reference controller.py:274-284 + scripts/mpc.py:151,240 for every instance at once).  The timed loop launches engine kernels
only: the policy step is smpc_policy_step, the counters are kept by smpc_accumulate_stats.

Workload (config.workload = "C1"): Z1-class 6-DoF, N = 30, 4096 OCP instances per GPU, controller 'st'
(terminal soft safe-set row through the 12-256-256-256-1 GELU MLP), EXTERNAL cost with exact Hessian, 6 capsule pairs,
Halton initial states, constant first guess -- SURVEY 8(d).
  --scaling weak   (default) every rank owns 4096 instances
  --scaling strong 4096 instances in total, split over the ranks by shard_range (the north star's wording)
The only exchange is ONE RCCL gather of the rollout log (applied controls + statuses of every timed step) at the end of the
timed region.

Launch: `python bench.py --gpus N` starts N fresh child processes itself (one per GPU, before anything touches a GPU) when
no launcher environment is present; under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it
reads RANK / LOCAL_RANK / WORLD_SIZE.  Asking for more GPUs than are visible is an error, never a silent 1-GPU run.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline:     algorithmic HBM bytes of the dominant kernel (k_qp_ipm) / its HIP-event duration vs 8 TB/s (a probe launch over
                the whole batch, alone on the GPU, at the state the timed loop ended in); `traffic` = measured HBM bytes per
                instance-iteration (rocprofv3 --pmc passes, profiles/) x the probe's own iteration count; kernel_ms_in_loop =
                HIP-event durations of the same kernels in a continuation of the loop after the timed region (per sub-batch
                launch, three streams sharing the chip; the timed region itself records no events) with the per-solve time
                quantiles the reference prints (scripts/mpc.py:300-303); the FP64 / MFMA FLOP fractions SURVEY 8(d) asks for;
                the launch's load balance
  cpu_baseline: the CPU oracle (a port, not acados) timed on this host on a bounded sample of the SAME closed-loop state:
                all-core throughput and 1-thread single-instance latency (p50 / p99, comparable to scripts/mpc.py:300-303).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

# one hardware queue per sub-batch stream (the HIP default of 4 makes two of them share a queue and serialise)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 4096
HORIZON = 30
CONTROLLER = os.environ.get('SMPC_BENCH_CONTROLLER', 'st')   # 'constraint_everywhere': the safe-set row on ALL nodes (reported separately)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3       # same guide: FP32 matrix (v_mfma_f32_32x32x2_f32) = FP32 vector peak
FP64_VALU_PEAK_TF = 78.6       # vendor figure for MI355X vector FP64 (the guide does not list it): 256 CUs x 4 SIMDs
                               # x 16 FP64 FMA lanes/clk x 2 flop x 2.4 GHz


def qp_flops_per_iteration(nq, N, nh):
    """N [7/3 nx^3 + 4 nx^2 nu + 2 nx nu^2 + nu^3/3 + (nx+nu)^2 (nh+nx)]  (0.486 MFLOP at Z1, N = 30, nh = 13)"""
    nx, nu = 2 * nq, nq
    return N * (7.0 / 3.0 * nx ** 3 + 4.0 * nx * nx * nu + 2.0 * nx * nu * nu + nu ** 3 / 3.0 + (nx + nu) ** 2 * (nh + nx))


def mlp_flops_per_row(nq, H=256, L=3):
    """forward + one VJP = 4 (2 nq H + (L-1) H^2 + H)  (537 600 at nq = 6)"""
    return 4.0 * (2 * nq * H + (L - 1) * H * H + H)


def algorithmic_bytes(nq, N):
    """inputs read once + outputs written once, FP64, per instance-step."""
    nx, nu = 2 * nq, nq
    return 8 * (nx + (N + 1) * nx + N * nu + (N + 1) * 5) + 8 * ((N + 1) * nx + N * nu) + 4


def in_loop_kernel_times(solvers, step_fn, sync_fn, n_steps):
    """Continue a closed loop for n_steps with the engines' HIP-event rings on (smpc_enable_timing(2)) and return, per solver,
    the mean per-launch durations in ms: [linearise, mlp, qp_setup, qp_ipm, solve_total] (None where nothing was read).  Used
    AFTER a script's timed region, so that the timed region itself records no events."""
    n_steps = min(n_steps, 60)
    for sv in solvers:
        sv.enable_timing(2)
    sync_fn()
    for _ in range(n_steps):
        step_fn()
    sync_fn()
    out = []
    for sv in solvers:
        rows = []
        for back in range(n_steps):
            tm = sv.timing_history(back)
            if tm is not None:
                rows.append([tm['time_lin'], tm['time_nn'], tm['time_qp_setup'], tm['time_qp_ipm'], tm['time_tot']])
        sv.enable_timing(0)
        out.append(1e3 * np.mean(rows, axis=0) if rows else None)
    return out


def roofline_of_launches(alg_bytes, ipm_ms, traffic_bytes_per_instance_iteration=None, instance_iterations=None):
    """`roofline` block for a set of k_qp_ipm launches: alg_bytes[i] = algorithmic HBM bytes of launch i (SURVEY 8(d) per
    instance-step x its instances), ipm_ms[i] = its HIP-event duration.  achieved = bytes per launch / average launch duration."""
    alg, ms = float(np.sum(alg_bytes)), float(np.sum(ipm_ms))
    ach = alg / (ms * 1e-3) / 1e9
    traffic = None
    if traffic_bytes_per_instance_iteration and instance_iterations:
        traffic = traffic_bytes_per_instance_iteration * instance_iterations / max(len(alg_bytes), 1)
    return {'bound': 'hbm', 'kernel': 'k_qp_ipm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
            'traffic': traffic, 'algorithmic_bytes_per_launch': alg / max(len(alg_bytes), 1),
            'avg_launch_ms': ms / max(len(alg_bytes), 1), 'launches': len(alg_bytes),
            'note': 'per sub-batch launch INSIDE the closed loop (the streams share the chip): algorithmic bytes / HIP-event duration'}


def build_problem(N=None, controller=None):
    from safe_mpc_amd.parser import Parameters
    from safe_mpc_amd.problem import OcpProblem
    from safe_mpc_amd.safe_set import SafeSetNet
    N = HORIZON if N is None else N
    par = Parameters({}, 'z1')
    par.nq, par.n_dof_safe_set, par.N = 6, 6, N
    par.net_size = [12, 256, 1]
    prob = OcpProblem(par, controller or CONTROLLER, 'ext', N=N)
    net = SafeSetNet.from_params(par, prob.x_min, prob.x_max)
    prob.set_normalisation(net.mean, net.std)
    return par, prob, net


def halton(n, dim, skip):
    primes = [2, 3, 5, 7, 11, 13, 17]
    idx = np.arange(skip, skip + n)
    out = np.zeros((n, dim))
    for d in range(dim):
        b, f, k = primes[d], 1.0, idx.copy()
        r = np.zeros(n)
        while k.max() > 0:
            f /= b
            r += f * (k % b)
            k //= b
        out[:, d] = r
    return out


def initial_states(solver, prob, B, rank):
    """Halton q0 in the joint box (guess_acados.py:79,100), collision filter through the engine (guess_acados.py:109)."""
    nq = prob.nq
    lo, hi = prob.lbx[:nq] + 0.05, prob.ubx[:nq] - 0.05
    xs, skip = [], 1 + rank * 16 * B
    while sum(len(x) for x in xs) < B:
        q = lo + halton(2 * B, nq, skip) * (hi - lo)
        skip += 2 * B
        x = np.hstack([q, np.zeros((2 * B, nq))])
        ok = solver.check_trajectory(x[:, None, :], tol_x=0.0, row_lb=prob.row_lb, row_ub=prob.row_ub)
        xs.append(x[ok])
    return np.vstack(xs)[:B]


def usable_cores():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands out a share of
    its host's cores; running one thread per visible core oversubscribes the quota and gets throttled)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]                     # cgroup v2
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())                # cgroup v1
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(prob, net, x0, xg, ug, p, budget_s=10.0, latency_solves=160):
    """CPU port (oracle/) on this host's cores, on a bounded sample of the closed-loop state the GPU leg was timed on:
    (1) OpenMP over instances on all cores -> instance-steps/s; (2) ONE thread, ONE instance per call -> latency p50 / p99
    (what scripts/mpc.py:300-303 prints per step for the reference's sequential loop)."""
    from oracle.oracle import Oracle, build, set_num_threads
    build()
    o = Oracle(prob, (net.weights, net.biases))
    cores = set_num_threads(usable_cores())
    chunk, done, iters = min(len(x0), 8 * cores), 0, 0
    o.solve_batch(x0[:cores], xg[:cores], ug[:cores], p[:cores])      # warm the code path
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:                         # cycle through the sample until the budget is used
        lo = done % (len(x0) - chunk + 1)
        sl = slice(lo, lo + chunk)
        iters += int(o.solve_batch(x0[sl], xg[sl], ug[sl], p[sl])[3].sum())
        done += chunk
    dt = time.perf_counter() - t0
    lat = []
    set_num_threads(1)                                                 # no team to wake: a plain single-threaded call
    for i in range(latency_solves):
        j = i % len(x0)
        t1 = time.perf_counter()
        o.solve_batch(x0[j:j + 1], xg[j:j + 1], ug[j:j + 1], p[j:j + 1])
        lat.append(time.perf_counter() - t1)
    lat = np.sort(np.array(lat)) * 1e3
    return {'value': done / dt, 'unit': 'instance-steps/s', 'cores': cores, 'kind': 'port',
            'mean_ipm_iterations': iters / max(done, 1),
            'latency_1thread_ms': {'p50': float(np.quantile(lat, 0.5)), 'p99': float(np.quantile(lat, 0.99)),
                                   'mean': float(lat.mean()), 'solves': int(len(lat))},
            'sample': f'{done} instance-solves drawn cyclically from {len(x0)} C1 instances in the closed-loop state the GPU leg ended '
                      f'in (same x, shifted guess, p), OpenMP over instances, {dt:.1f} s; then {len(lat)} single-instance solves on one '
                      f'thread; oracle/smpc_oracle.cpp -O3 (dual-number derivatives, dense stage algebra), a CPU restatement -- not acados; '
                      f'PARITY UNPINNED: the oracle is held by first principles and its own frozen output, by no vector of the reference (DESIGN.md section 5)'}


def cpu_baseline_latency(prob, net, states):
    """The one-thread, one-instance leg of `cpu_baseline` on a given list of (x, x_guess, u_guess, p) states (numpy, B = 1)."""
    from oracle.oracle import Oracle, build, set_num_threads
    build()
    o = Oracle(prob, (net.weights, net.biases))
    set_num_threads(1)
    lat = []
    for x, xg, ug, p in states:
        t1 = time.perf_counter()
        o.solve_batch(x, xg, ug, p)
        lat.append(time.perf_counter() - t1)
    lat = np.array(lat) * 1e3
    return {'p50_ms': float(np.quantile(lat, 0.5)), 'p99_ms': float(np.quantile(lat, 0.99)), 'mean_ms': float(lat.mean()),
            'solves': int(len(lat))}


def small_batch_latency(N=HORIZON, batches=(1,), n_steps=200, device=0, keep_states=None):
    """Per-solve latency of a SMALL batch over an n_steps closed loop of controller 'st' (the statistic the reference prints for
    its one-instance loop: quantiles of the solver time per step, scripts/mpc.py:239,300-303; budget dt = 5 ms, config.yaml:7).
    Device-resident inputs, wall clock around `solve` + sync (launch latency included).  GPU only; the CPU port's one-thread
    figure on the same kind of state is `cpu_baseline.latency_1thread_ms`."""
    import torch
    from safe_mpc_amd.solver import BatchedOcpSolver
    dev = torch.device('cuda', device)
    t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    par, prob, net = build_problem(N=N, controller='st')
    rows = []
    for B in batches:
        sv = BatchedOcpSolver(prob, net, device=device)
        x0 = initial_states(sv, prob, B, 3)
        xg = np.repeat(x0[:, None, :], N + 1, axis=1)
        ug = np.zeros((B, N, prob.nu))
        p = np.zeros((B, N + 1, 5))
        p[:, :, :3], p[:, :, 3], p[:, :, 4] = prob.ee_ref, par.alpha, 1.0
        x, xg, ug, p = t(x0), t(xg), t(ug), t(p)
        lat, its = [], []
        for k in range(n_steps + 5):
            if keep_states is not None and B == 1:
                keep_states.append([a.cpu().numpy().copy() for a in (x, xg, ug, p)])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            xo, uo, st, it = sv.solve(x, xg, ug, p)
            sv.sync()
            dt_ = time.perf_counter() - t0
            if k >= 5:
                lat.append(dt_)
                its.append(float(it.double().mean().item()))
            xg, ug, ua = sv.provide_control((st == 0).to(torch.int32), xo, uo, xg, ug)
            x, _ = sv.plant_step(x, ua)
            xg = sv.guess_correction(xg, ug)
            sv.sync()
        lat = np.array(lat) * 1e3
        rows.append({'N': N, 'B': B, 'p50_ms': float(np.quantile(lat, 0.5)), 'p99_ms': float(np.quantile(lat, 0.99)),
                     'mean_ms': float(lat.mean()), 'mean_ipm_iterations': float(np.mean(its)), 'solves': int(len(lat))})
        sv.close()
    return rows


def launch_children(args):
    """`python bench.py --gpus N` without a launcher: N fresh children, one per GPU, started before this process touches a GPU
    (device_count() only counts -- where it does go through the HIP runtime in the parent that is harmless here: the ranks are
    fresh child processes, never an exec of this one).  Children inherit stdout: rank 0 prints the JSON line."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write(f'bench.py: --gpus {args.gpus} requested but {have} GPU(s) visible; refusing to run on fewer\n')
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        for pr in procs:
            rc = max(rc, abs(pr.wait()))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults = the window SURVEY 8(d) / BASELINE.md section 3 quote C1 on (10 warm-up + 100 timed steps): the closed loop gets
    # more expensive with time, and the 3 + 20 window of rounds 1-4 flattered `value` by ~18 %
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', choices=['c1', 'c2', 'c3', 'c4'], default='c1',
                    help="BASELINE.json's configs: c1 Z1 N=30 x 4096 (the headline); c2 the same with one perturbed plant per instance x 65536; "
                         'c3 the horizon x alpha grid, 32768 over 8 GPUs grouped by horizon; c4 7-DoF N=40, row on every node, 131072 over 8 GPUs')
    ap.add_argument('--batch', type=int, default=None, help='instances per GPU (weak) / in total (strong); default: the config\'s own size')
    ap.add_argument('--graphs', type=int, default=0,
                    help='1: replay each sub-batch step as a captured hipGraph (one host launch per sub-batch and step instead of '
                         '~25; measured on one GPU: no gain, DESIGN.md section 7 -- kept for A/B runs on a node whose ranks share '
                         'the host; legal under torch.distributed: the log slot comes from a device-side step counter)')
    ap.add_argument('--streams', type=int, default=None,
                    help='sub-batches per GPU (default: 3 above 1024 instances, 2 up to 1024, 1 up to 512), each on its own HIP stream: independent instances, so the sub-batches advance '
                         'independently and the long tail of one QP launch overlaps the bulk of another (round 4, DESIGN.md '
                         'section 8: 2 / 3 / 4 / 5 / 6 streams = 3.15 / 2.92 / 3.17 / 3.22 / 3.14 ms per step)')
    ap.add_argument('--qp-form', choices=['auto', 'throughput', 'latency'], default='auto',
                    help="form of the interior-point solve (smpc_set_qp_mode); auto: the latency form (a workgroup per instance) up to "
                         "1536 instances per GPU in sub-batches of at most 512, the throughput form above -- the others are for A/B runs")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-latency', action='store_true', help='skip the one-instance latency measurement added to the C1 line')
    ap.add_argument('--noise', type=float, default=0.0,
                    help='model noise in percent (BASELINE config 2): one perturbed plant per instance, the draws of utils.py:138-166 '
                         'seeded by the instance id')
    ap.add_argument('--control-noise', type=float, default=0.0, help='torque noise in percent of tau_max (env_model.py:196)')
    ap.add_argument('--no-loop-timing', action='store_true',
                    help='skip the extra pass (after the timed region) that records per-kernel HIP events of the loop')
    ap.add_argument('--no-survey-window', action='store_true',
                    help="skip the extra untimed-for-`value` pass over SURVEY 8(d)'s window (10 warm-up + 100 timed steps)")
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help='weak: --batch instances per GPU; strong: --batch instances in total, split over the GPUs')
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_children(args))

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to stdout when its first communicator comes
    # up: keep the real stdout aside for the result line and send everything else written to fd 1 to stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU (or none: bench.py spawns them)')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the engine has no CPU fallback')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    use_dist = world > 1 or os.environ.get('SMPC_FORCE_DIST') == '1'      # the env knob exercises the RCCL path with one rank
    init_only = os.environ.get('SMPC_DIST_INIT_ONLY') == '1'             # diagnostic: communicator up, no per-step exchange
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    if init_only:
        use_dist = False

    from safe_mpc_amd.sharding import gather_to_root, shard_by_horizon, shard_range
    from safe_mpc_amd.solver import BatchedOcpSolver
    from safe_mpc_amd.controller import get_controller
    cfg = args.config
    controller = 'constraint_everywhere' if cfg == 'c4' else CONTROLLER
    # ---- the instances of this rank: a list of groups, each one problem (one horizon) with the global ids of its instances ----------
    #   c1 / c2  Z1, N = 30; weak: --batch per GPU; strong: --batch in total, split by shard_range
    #   c3       the horizon x alpha grid of run_mpc_horizons.sh:19-34 / run_mpc_alphas.sh:19-34 (N in 20..40 x alpha in 20..50), grouped
    #            by horizon (a handle has one N) and every group sliced over the ranks (shard_by_horizon); weak: every rank owns one
    #            eighth of an 8 x --batch grid (BASELINE: 32 768 over 8 GPUs), strong: --batch in total over the ranks that are there
    #   c4       7-DoF Franka-class, N = 40, row on every node; weak: --batch per GPU (BASELINE: 131 072 over 8 = 16 384), strong: split
    per_gpu_default = {'c1': B_PER_GPU, 'c2': 65536, 'c3': 4096, 'c4': 16384}[cfg]
    if args.batch is None:
        args.batch = per_gpu_default if args.scaling == 'weak' else per_gpu_default * (8 if cfg in ('c3', 'c4') else 1)
    if cfg == 'c2':
        args.noise = args.noise if args.noise > 0 else 10.0                 # BASELINE config 2: model-noise Monte Carlo
        args.control_noise = args.control_noise if args.control_noise > 0 else 1.0
    if cfg == 'c4':
        from safe_mpc_amd.parser import Parameters
        from safe_mpc_amd.problem import OcpProblem
        from safe_mpc_amd.safe_set import SafeSetNet
        par = Parameters({}, 'fr7', filename=os.path.join(ROOT, 'config_fr7.yaml'))
        par.N = 40
        prob = OcpProblem(par, controller, 'ext', N=40)
        net = SafeSetNet.from_params(par, prob.x_min, prob.x_max)
        prob.set_normalisation(net.mean, net.std)
    else:
        par, prob, net = build_problem()
    nx, nu, nq = prob.nx, prob.nu, prob.nq

    def partition(r):
        """{horizon: global instance ids} owned by rank r, and the job's total"""
        if cfg == 'c3':
            n_full = args.batch * 8 if args.scaling == 'weak' else args.batch
            horizons = np.repeat([20, 25, 30, 35, 40], n_full // 5 + 1)[:n_full]
            return shard_by_horizon(horizons, 8 if args.scaling == 'weak' else world, r % 8 if args.scaling == 'weak' else r), \
                (args.batch * world if args.scaling == 'weak' else n_full)
        if args.scaling == 'strong':
            lo_r, hi_r = shard_range(args.batch, world, r)
            return {prob.N: np.arange(lo_r, hi_r)}, args.batch
        return {prob.N: np.arange(args.batch) + r * args.batch}, args.batch * world
    owned, B_total = partition(rank)
    B = int(sum(len(v) for v in owned.values()))
    if B < 1:
        raise SystemExit('strong scaling: fewer instances than ranks')
    N = prob.N
    # (up to 512 instances the engine launches the latency form of the QP solve, a workgroup per instance: its launches have no tail
    #  worth overlapping, and one launch over all instances lets the engine see the whole batch when it picks the form)
    # (up to 1024: two launches of up to 512 workgroups, the second one's starting as the first one's retire -- measured 1.65 ms per step
    #  against 1.80 with three sub-batches and 2.05 with one launch of 1024)
    S = (1 if B <= 512 else (2 if B <= 1024 else max(1, min(3, B // 256 or 1)))) if args.streams is None else max(1, min(args.streams, B))

    t = lambda a, dt=torch.float64: torch.tensor(a, dtype=dt, device=dev)

    # Results for rank 0: every sub-batch writes the applied control and the status of each step into its slice of a rollout
    # log -- straight from the engine's kernels (u_out / status pointers of smpc_policy_step), no copy -- and ONE gather of the
    # log ends the timed region ("once per step ... or once per rollout").  A per-step collective would need an
    # event hand-off between the sub-batch streams and the collective's stream every step; measured on one GPU that
    # re-synchronises the streams and costs 0.5-1.5 ms per step (round 2, on a 3.9 ms step) even with the collective itself skipped.
    K_log = max(args.steps, 1)
    log_bytes = K_log * B * (nu * 8 + 4)
    gather_in = torch.empty((log_bytes,), dtype=torch.uint8, device=dev)
    u_log = gather_in[:K_log * B * nu * 8].view(torch.float64).view(K_log, B, nu)
    st_log = gather_in[K_log * B * nu * 8:].view(torch.int32).view(K_log, B)

    class Sub:
        pass
    subs, off = [], 0
    probe_sv = BatchedOcpSolver(prob, net, device=local)                 # (collision filter of the initial states; the roofline probe)
    starts512 = initial_states(probe_sv, prob, 512, rank) if cfg == 'c3' else None
    for Ng, gids in owned.items():
        if cfg == 'c3':
            par_g, prob_g, net_g = build_problem(N=int(Ng))
            # one sub-batch (= handle = stream) per horizon group: five of them share the GPU
            pieces = [(0, len(gids))]
        else:
            par_g, prob_g, net_g = par, prob, net
            pieces = [shard_range(len(gids), S, i) for i in range(S)]
        # model noise (BASELINE config 2, generate_urdf_noise.py:20-36): one plant per instance, seed = GLOBAL instance id (SURVEY 8(d)
        # C2; the reference perturbs and reseeds per model, utils.py:126-171), + one torque-noise draw per instance (env_model.py:196)
        jt_g = tn_g = None
        if args.noise > 0:
            from safe_mpc_amd.closed_loop import perturbed_joint_tables_batched
            jt_g = np.ascontiguousarray(perturbed_joint_tables_batched(par_g, nq, args.noise, gids)).view(np.float64).reshape(len(gids), nq, -1)
        if args.control_noise > 0:
            tn_g = np.random.default_rng(1 + rank).normal(0.0, prob_g.tau_max * args.control_noise / 100, (len(gids), nu))
        if cfg == 'c3':
            x0_g = starts512[gids % 512]
        else:
            x0_g = initial_states(probe_sv, prob_g, len(gids), rank)
            if cfg == 'c4':
                x0_g[:, nq:] = 0.1 * np.random.default_rng(rank).uniform(-1, 1, (len(gids), nq)) * prob_g.ubx[nq:]
        p_g = np.zeros((len(gids), Ng + 1, 5))
        p_g[:, :, :3], p_g[:, :, 3], p_g[:, :, 4] = prob_g.ee_ref, par_g.alpha, 1.0
        if cfg == 'c3':
            p_g[:, :, 3] = np.array([20.0, 30.0, 40.0, 50.0])[gids % 4][:, None]          # alpha rides per instance (run_mpc_alphas.sh:19)
        for lo, hi in pieces:
            sv = BatchedOcpSolver(prob_g, net_g, device=local)
            sb = Sub()
            sb.solver, sb.n, sb.off, sb.N = sv, hi - lo, off + lo, int(Ng)
            sb.par, sb.prob, sb.net = par_g, prob_g, net_g
            sb.x0_h, sb.p_h = x0_g[lo:hi], p_g[lo:hi]
            sb.stream = torch.cuda.ExternalStream(sv.L.smpc_stream(sv.h), device=dev)
            with torch.cuda.stream(sb.stream):
                # the reference's controller object for these instances, all of its state in HBM (safe_mpc_amd/controller.py)
                sb.ctrl = get_controller(controller, par_g, sb.n, cost='ext', N=int(Ng), solver=sv, net=net_g, device=local, device_state=True)
                sb.ctrl.setGuess(t(np.repeat(sb.x0_h[:, None, :], Ng + 1, axis=1)), t(np.zeros((sb.n, Ng, nu))))
                sb.ctrl.p.copy_(t(sb.p_h))
                sb.x_sim, sb.x_next = t(sb.x0_h), t(sb.x0_h)
                sb.u_eff = torch.empty((sb.n, nu), dtype=torch.float64, device=dev)
                sb.acc = torch.zeros((3,), dtype=torch.int64, device=dev)      # [sum of IPM iterations, failed solves, solves]
                sb.jt = t(jt_g[lo:hi]) if jt_g is not None else None
                sb.tn = t(tn_g[lo:hi]) if tn_g is not None else None
                sb.status_home = sb.ctrl.last_status
                sb.u_stage = torch.empty((sb.n, nu), dtype=torch.float64, device=dev)       # (--graphs 1: the step's control before it
                sb.slot_dev = torch.zeros((1,), dtype=torch.int64, device=dev)             #  is copied to log row slot_dev)
            sb.tsum, sb.tcnt = np.zeros(5), 0
            subs.append(sb)
        off += len(gids)
    solvers = [sb.solver for sb in subs]
    S = len(subs)
    # The form of the QP solve (smpc_set_qp_mode): a handle picks it from ITS launch size, the bench from the rank's whole batch -- the
    # latency form (a workgroup per instance) while the sub-batches are at most 512 instances (1536 per GPU), the throughput form above
    # (three sub-batches of 683 on the latency form: 2.81 ms per step against 2.60)
    qp_form = 'latency' if max(sb.n for sb in subs) <= 512 and B <= 1536 else 'throughput'
    if args.qp_form != 'auto':
        qp_form = args.qp_form
    for sv in solvers + [probe_sv]:
        sv.set_qp_mode(qp_form)
    sizes_all = [int(sum(len(v) for v in partition(r)[0].values())) for r in range(world)]
    if cfg == 'c3' and args.scaling == 'weak':
        B_total = int(sum(sizes_all))        # (the horizon groups do not divide evenly: the first ranks take the remainders)
    # receive buffers of the per-step gather are allocated once, outside the timed loop
    gather_bufs = None
    if use_dist:
        from safe_mpc_amd.sharding import gather_buffers
        gather_bufs = gather_buffers(gather_in.view(K_log * B, nu * 8 + 4), [K_log * n_r for n_r in sizes_all], rank)

    def sub_step(sb, first):
        """One closed-loop step of one sub-batch, entirely on its stream: engine kernels only (graph-capturable when the log
        slot is fixed).  <Controller>.step = guessCorrection, RTI solve, accept test, provideControl (controller.py:274-284 for
        'st'; :651-661 for 'constraint_everywhere') as smpc_policy_step; then the plant (env_model.py:192-206)."""
        sv, ctrl = sb.solver, sb.ctrl
        fixed = bool(args.graphs)            # (--graphs 1: a captured step bakes its addresses in -- state copied back, and the
                                             #  log slot taken from a device-side step counter instead of the host's)
        if fixed:
            ctrl.step_on_device(sb.x_sim, u_out=sb.u_stage)
            sv.plant_step(sb.x_sim, sb.u_stage, sb.jt, sb.tn, out=(sb.x_next, sb.u_eff))
            sb.x_sim.copy_(sb.x_next)
            # rollout log: row `slot_dev` of this sub-batch's columns, then slot_dev = (slot_dev + 1) mod K_log -- all on the device
            u_log[:, sb.off:sb.off + sb.n].index_copy_(0, sb.slot_dev, sb.u_stage.unsqueeze(0))
            st_log[:, sb.off:sb.off + sb.n].index_copy_(0, sb.slot_dev, ctrl.last_status.unsqueeze(0))
            sb.slot_dev.add_(1).remainder_(K_log)
        else:
            slot = step_no[0] % K_log
            u_slot = u_log[slot, sb.off:sb.off + sb.n]
            if use_dist:      # the status of this step lands in the rollout log as well
                object.__setattr__(ctrl, 'last_status', st_log[slot, sb.off:sb.off + sb.n])
            ctrl.step_on_device(sb.x_sim, u_out=u_slot)
            sv.plant_step(sb.x_sim, u_slot, sb.jt, sb.tn, out=(sb.x_next, sb.u_eff))
            sb.x_sim, sb.x_next = sb.x_next, sb.x_sim
        sv.accumulate_stats(ctrl.last_status, ctrl.qp_iter, sb.acc)

    step_no = [0]
    loop_timing = [False]

    def step(first):
        for sb in subs:
            with torch.cuda.stream(sb.stream):
                if sb.graph is not None and not first:
                    sb.graph.replay()       # the ~25 launches of one sub-batch step as ONE hipGraphLaunch
                else:
                    sub_step(sb, first)
        step_no[0] += 1

    def gather_results():
        """the single result gather, inside the timed region: the rollout log of this rank's instances to rank 0"""
        cur = torch.cuda.current_stream()
        for sb in subs:
            cur.wait_stream(sb.stream)
        if os.environ.get('SMPC_SKIP_NCCL') != '1':      # (diagnostic knob: everything but the collective itself)
            gather_to_root(gather_in.view(K_log * B, nu * 8 + 4), sizes=[K_log * n_r for n_r in sizes_all], bufs=gather_bufs, concat=False)

    def barrier():
        for sb in subs:
            sb.solver.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for sb in subs:
        sb.graph = None
    for i in range(args.warmup):
        step(first=(i == 0))
    barrier()
    # hipGraph capture of one sub-batch step (after the warm-up: every workspace exists, nothing allocates or syncs).
    # With S sub-batches in flight the host has to issue S x 25 launches per step; as graphs that is S launches, and the
    # streams drift apart so that the long tail of one sub-batch's QP kernel is filled by the bulk of another's.
    if args.graphs and args.warmup >= 2:
        try:
            for sb in subs:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=sb.stream, capture_error_mode='relaxed'):
                    sub_step(sb, False)
                sb.graph = g
        except Exception as e:      # report, fall back to eager launches
            print(f'[bench] graph capture failed ({type(e).__name__}: {e}); eager launches', file=sys.stderr)
            for sb in subs:
                sb.graph = None
        barrier()
    for sb in subs:
        with torch.cuda.stream(sb.stream):
            sb.acc.zero_()
    # (the timed region runs uninstrumented: no HIP events inside it.  kernel_ms_in_loop comes from a pass of its own, below)
    for sb in subs:
        with torch.cuda.stream(sb.stream):
            sb.slot_dev.zero_()
    barrier()
    t0 = time.perf_counter()
    step_no[0] = 0
    for i in range(args.steps):
        step(first=(args.warmup == 0 and i == 0))
    # what the host spent enqueueing the timed steps (~25 launches per sub-batch and step, from Python): when this approaches
    # ms_per_step the loop is bound by the host, not by the GPU -- the thing to look at first when 8 ranks share one host
    host_issue = time.perf_counter() - t0
    if use_dist:
        gather_results()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    acc_all = sum(sb.acc.cpu().numpy().astype(np.int64) for sb in subs)
    log_check = None
    if os.environ.get('SMPC_BENCH_CHECK_LOG') == '1' and use_dist:
        # (test hook: every timed step must have left its own row in the rollout log -- with --graphs 1 the row index is the
        #  device-side counter's)
        ul, sl = u_log.cpu().numpy(), st_log.cpu().numpy()
        distinct = len({ul[k].tobytes() for k in range(K_log)})
        log_check = (f'ok: {distinct} distinct log rows, statuses all zero' if distinct == K_log and not sl.any()
                     else f'BAD: {distinct} distinct rows of {K_log}, {int((sl != 0).sum())} non-zero statuses')
    # Per-kernel HIP-event durations of the loop's own launches: the SAME loop continued for a few more steps with the engine's
    # event ring switched on (5 hipEventRecord per solve), outside the timed region -- `value` is measured without them.  The
    # per-solve `time_tot` of this pass is also what the reference reports per step (scripts/mpc.py:239,300-303).
    loop_timing[0] = not args.no_loop_timing and subs[0].graph is None and world == 1
    tt_all, t_ev = [], 0.0
    if loop_timing[0]:
        n_ev = min(max(args.steps, 8), 48)
        for sb in subs:
            sb.solver.enable_timing(2)
        barrier()
        t_ev = time.perf_counter()
        for i in range(n_ev):
            step(first=False)
        barrier()
        t_ev = (time.perf_counter() - t_ev) / n_ev
        for sb in subs:
            for back in range(n_ev):
                tm = sb.solver.timing_history(back)
                if tm is not None:
                    sb.tsum += [tm['time_lin'], tm['time_nn'], tm['time_qp_setup'], tm['time_qp_ipm'], tm['time_tot']]
                    sb.tcnt += 1
                    tt_all.append([tm['time_lin'], tm['time_qp_setup'] + tm['time_qp_ipm'], tm['time_qp_ipm'], tm['time_tot']])
            sb.solver.enable_timing(0)
    loop_timing[0] = False
    for sb in subs:
        if use_dist:
            object.__setattr__(sb.ctrl, 'last_status', sb.status_home)
    mean_iter = float(acc_all[0]) / max(int(acc_all[2]), 1)
    fails = float(acc_all[1])
    in_loop = None
    if sum(sb.tcnt for sb in subs) > 0:
        ts = sum(sb.tsum for sb in subs) / sum(sb.tcnt for sb in subs) * 1e3
        tq = np.array(tt_all) * 1e3
        q = lambda c: {'p50': float(np.quantile(tq[:, c], 0.5)), 'p99': float(np.quantile(tq[:, c], 0.99))}
        in_loop = {'linearise': ts[0], 'mlp': ts[1], 'qp_setup': ts[2], 'qp_ipm': ts[3], 'solve_total': ts[4],
                   'launches_sampled': int(sum(sb.tcnt for sb in subs)), 'instances_per_launch': [sb.n for sb in subs],
                   'ms_per_step_with_events': 1e3 * t_ev,
                   # the reference's own report for this path (scripts/mpc.py:300-303: 99 % quantile of controller.getTime() per
                   # step), here per sub-batch solve: time_lin / time_qp / time_qp_solver_call / time_tot in ms
                   'solve_time_quantiles_ms': {'time_lin': q(0), 'time_qp': q(1), 'time_qp_solver_call': q(2), 'time_tot': q(3)},
                   'note': 'HIP-event durations per SUB-BATCH launch in a continuation of the timed loop with the event ring on '
                           '(outside the timed region; the sub-batch streams share the chip, so these overlap each other and '
                           'their sum over the streams is not the step time)'}

    # roofline probe of the dominant kernel (k_qp_ipm): ONE launch over the whole batch, alone on the GPU, HIP events on
    # the engine's stream around each phase (the per-launch duration rocprofv3 --kernel-trace reports for the same launch)
    roof = None
    # (a grid of horizons, C3, is probed on its N = 30 group: a handle has one horizon)
    probe_N = 30 if any(sb.N == 30 for sb in subs) else subs[0].N
    psubs = [sb for sb in subs if sb.N == probe_N]
    prob, net, N = psubs[0].prob, psubs[0].net, probe_N
    Bp = int(sum(sb.n for sb in psubs))
    if rank == 0:
        sv = probe_sv if probe_sv.N == probe_N else BatchedOcpSolver(prob, net, device=local)
        sv.set_qp_mode(qp_form)
        st_ = torch.cuda.ExternalStream(sv.L.smpc_stream(sv.h), device=dev)
        with torch.cuda.stream(st_):
            xs = torch.cat([sb.x_sim for sb in psubs]); xgf = torch.cat([sb.ctrl.x_guess for sb in psubs])
            ugf = torch.cat([sb.ctrl.u_guess for sb in psubs]); pf = torch.cat([sb.ctrl.p for sb in psubs])
        torch.cuda.synchronize()
        sv.enable_timing(True)
        acc = np.zeros(8)
        probes, it_probe = 5, 0.0
        with torch.cuda.stream(st_):      # (one untimed pass: the new handle's workspaces are allocated inside its first solve)
            sv.guess_correction(xgf, ugf)
            sv.solve(xs, xgf, ugf, pf)
        sv.sync()
        for _ in range(probes):
            with torch.cuda.stream(st_):
                sv.guess_correction(xgf, ugf)
                out_p = sv.solve(xs, xgf, ugf, pf)
            tm = sv.timing()
            it_probe += float(out_p[3].double().mean().item())
            acc += [tm['time_lin'], tm['time_nn'], tm['time_qp'], tm['time_tot'], tm['time_qp_setup'], tm['time_qp_ipm'],
                    tm['qp_wave_busy_mean'], tm['qp_wave_span']]
        acc /= probes
        it_probe /= probes
        alg = algorithmic_bytes(prob.nq, N) * Bp
        ach = alg / acc[5] / 1e9
        # the form of the QP solve the engine picks for a launch of this size (engine.hip: qp_wg_choice; smpc_set_qp_mode)
        qp_kernel = 'k_qp_ipm_wg' if qp_form == 'latency' else 'k_qp_ipm'
        # FLOP fractions: algorithmic flops of the launch / its HIP-event duration / peak
        nh = prob.nq + prob.desc.n_rows + 1
        qp_fl = qp_flops_per_iteration(prob.nq, N, nh) * it_probe * Bp
        mlp_rows = Bp if prob.desc.nn_mode == 1 else (Bp * N if prob.desc.nn_mode == 2 else 0)
        mlp_fl = mlp_flops_per_row(prob.nq) * mlp_rows
        # HBM bytes come from separate rocprofv3 --pmc passes over scripts/qp_bench.py (counters cannot be read from inside this
        # process).  The committed file holds bytes per INSTANCE-ITERATION of k_qp_ipm (the kernel's traffic is proportional to
        # the iterations it runs); traffic of THIS probe launch = that x the probe's own iteration count x instances, so that
        # traffic / kernel time is the rate of the launch that was timed here.  null when there is no file for this round's kernel.
        traffic, traffic_rate, traffic_src, bpi = None, None, None, None
        tag_ = ('_wg' if qp_kernel == 'k_qp_ipm_wg' else '') + ('_c4' if cfg == 'c4' else '')
        for name in (f'r06_pmc_traffic{tag_}.json', f'r05_pmc_traffic{tag_}.json'):
            tf = os.path.join(ROOT, 'profiles', name)
            if bpi is None and os.path.exists(tf) and controller in ('st', 'constraint_everywhere'):
                tj = json.load(open(tf))
                bpi = tj.get('bytes_per_instance_iteration')
                if bpi:
                    bpi = bpi * (N + 1) / (41.0 if cfg == 'c4' else 31.0)       # (the kernel moves a fixed number of bytes per stage)
                    traffic = bpi * it_probe * Bp
                    traffic_rate, traffic_src = traffic / acc[5] / 1e9, 'profiles/' + name
        roof = {'bound': 'hbm', 'kernel': qp_kernel, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': ach / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                'traffic_bytes_per_instance_iteration': bpi, 'traffic_GBps': traffic_rate,
                'flop_frac_qp_fp64': qp_fl / acc[5] / 1e12 / FP64_VALU_PEAK_TF,
                'flop_frac_mlp_mfma': (mlp_fl / acc[1] / 1e12 / MFMA_F32_PEAK_TF) if mlp_rows and acc[1] > 0 else None,
                'flops': {'qp_per_launch': qp_fl, 'qp_TFLOPs': qp_fl / acc[5] / 1e12, 'fp64_valu_peak_TFLOPs': FP64_VALU_PEAK_TF,
                          'mlp_per_launch': mlp_fl, 'mlp_TFLOPs': (mlp_fl / acc[1] / 1e12) if mlp_rows and acc[1] > 0 else None,
                          'mfma_f32_peak_TFLOPs': MFMA_F32_PEAK_TF, 'mlp_rows': mlp_rows,
                          'note': 'mlp time = features + GEMMs + chain kernels of the network pass (HIP events)'},
                'load_balance': {'span_over_mean_busy': acc[7] / acc[6] if acc[6] > 0 else None,
                                 'mean_instance_busy_ms': acc[6] * 1e3, 'launch_span_ms': acc[7] * 1e3,
                                 'note': 'k_qp_ipm: first start -> last end of any half-wavefront over the mean busy time of one '
                                         '(smpc_get_qp_wave_stats); 1.0 = no time spent waiting for the slowest instances'},
                'mean_ipm_iterations_in_probe': it_probe,
                'kernel_ms': {'linearise': acc[0] * 1e3, 'mlp': acc[1] * 1e3, 'qp_setup': acc[4] * 1e3, 'qp_ipm': acc[5] * 1e3,
                              'solve_total': acc[3] * 1e3,
                              'note': 'ONE launch over the whole batch, alone on the GPU (the roofline probe) -- not the timed loop'},
                'kernel_ms_in_loop': in_loop,
                'algorithmic_bytes_per_launch': alg, 'launch': f'one launch, B={Bp}, N={N}, alone on the GPU, closed-loop state after the timed steps'}
        if len(psubs) < len(subs):
            # several horizons share the GPU (C3): every group's launch inside the loop, algorithmic bytes / HIP-event duration
            tl = [(sb, sb.tsum / sb.tcnt) for sb in subs if sb.tcnt > 0]
            if tl:
                roof['in_loop_all_groups'] = roofline_of_launches([algorithmic_bytes(sb.prob.nq, sb.N) * sb.n for sb, _ in tl], [1e3 * t_[3] for _, t_ in tl])

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        nb = min(Bp, 1024)
        h = lambda a: a[:nb].cpu().numpy()
        cpu = cpu_baseline(prob, net, h(xs), h(xgf), h(ugf), h(pf))

    # SURVEY 8(d) / BASELINE.md section 3 quote C1 over "100 timed steps after 10 warm-up"; the closed loop gets more expensive
    # with time (more IPM iterations per solve), so a shorter window flatters the number.  When the run's own K / W differ, the
    # same loop is run once more over exactly that window -- from the same initial state, after everything above, outside the
    # timed region that `value` reports -- and added to the line.
    survey = None
    if (args.steps, args.warmup) == (100, 10):
        survey = {'steps': 100, 'warmup': 10, 'ms_per_step': 1e3 * elapsed / 100, 'value': B_total * 100 / elapsed,
                  'mean_ipm_iterations': mean_iter, 'failed_instance_steps': fails,
                  'note': 'the window SURVEY 8(d) quotes C1 on IS this run\'s timed region (the default since round 5)'}
    elif world == 1 and not args.no_survey_window and not args.graphs:
        for sb in subs:
            with torch.cuda.stream(sb.stream):
                sb.ctrl.setGuess(t(np.repeat(sb.x0_h[:, None, :], sb.N + 1, axis=1)), t(np.zeros((sb.n, sb.N, nu))))
                sb.ctrl.p.copy_(t(sb.p_h))
                sb.x_sim.copy_(t(sb.x0_h))
                if hasattr(sb.ctrl, 'fails'):
                    sb.ctrl.fails.zero_()
        barrier()
        for i in range(10):
            step(first=(i == 0))
        for sb in subs:
            with torch.cuda.stream(sb.stream):
                sb.acc.zero_()
        barrier()
        t1 = time.perf_counter()
        for i in range(100):
            step(first=False)
        barrier()
        el2 = time.perf_counter() - t1
        acc2 = sum(sb.acc.cpu().numpy().astype(np.int64) for sb in subs)
        survey = {'steps': 100, 'warmup': 10, 'ms_per_step': 1e3 * el2 / 100, 'value': B_total * 100 / el2,
                  'mean_ipm_iterations': float(acc2[0]) / max(int(acc2[2]), 1), 'failed_instance_steps': float(acc2[1]),
                  'note': 'the window SURVEY 8(d) quotes C1 on, run after the timed region from the same initial state'}

    if rank == 0:
        total = B_total * args.steps
        per = '/GPU' if args.scaling == 'weak' else ' in total'
        nn_txt = ' (soft terminal NN row, ' if controller == 'st' else ' (NN row as configured, '
        if cfg in ('c1', 'c2'):
            std = args.batch == (65536 if cfg == 'c2' else B_PER_GPU)
            tag = cfg.upper() if std else f'{cfg.upper()}-like ({args.batch} instead of {65536 if cfg == "c2" else B_PER_GPU} instances)'
            workload = (tag + f': Z1-class 6-DoF, N=30, {args.batch} instances{per}' +
                        (f', model noise {args.noise}% (one plant per instance, seed = instance id) + torque noise {args.control_noise}%' if args.noise > 0 or args.control_noise > 0 else '') +
                        ', controller ' + controller + nn_txt + 'MLP 12-256-256-256-1 fp32), EXT cost exact Hessian, 6 capsule pairs, Halton x0')
            metric = f'RTI-MPC instance-steps/s (batch={args.batch}{" per GPU" if args.scaling == "weak" else " in total"}, Z1 N=30)'
        elif cfg == 'c3':
            workload = (f'C3: horizon x alpha grid of run_mpc_horizons.sh / run_mpc_alphas.sh (N in 20..40 x alpha in 20..50), {B_total} instances in total, '
                        f'grouped by horizon and sliced over {8 if args.scaling == "weak" else world} shares (this rank: {B} instances, ' +
                        ', '.join(f'N={sb.N}: {sb.n}' for sb in subs) + '), Z1-class 6-DoF, controller ' + controller + nn_txt +
                        'MLP 12-256-256-256-1 fp32), EXT cost exact Hessian, 6 capsule pairs, Halton x0')
            metric = f'RTI-MPC instance-steps/s (horizon x alpha grid, {B} instances per GPU, Z1 N=20..40)'
        else:
            workload = (f'C4: 7-DoF Franka-class (config_fr7.yaml), N=40, {args.batch} instances{per}, controller {controller} (NN row on every node, '
                        'MLP 14-256-256-256-1 fp32), EXT cost exact Hessian, sphere obstacle + floor rows, Halton x0')
            metric = f'RTI-MPC instance-steps/s (batch={args.batch}{" per GPU" if args.scaling == "weak" else " in total"}, 7-DoF N=40)'
        extra = {}
        if cfg == 'c1' and world == 1 and not args.no_latency:
            # the reference's own statistic for this path: the solver time per step of ONE instance (scripts/mpc.py:300-303; budget dt = 5 ms)
            try:
                extra['latency_b1_ms'] = small_batch_latency(N=HORIZON, batches=(1,), n_steps=100, device=local)[0]
            except Exception as e:      # (never lose the line to the side measurement)
                extra['latency_b1_ms'] = {'error': f'{type(e).__name__}: {e}'}
            pf_ = os.path.join(ROOT, 'profiles', 'r06_strong_scaling_proxy.json')
            if os.path.exists(pf_):
                rows_ = json.load(open(pf_))['rows']
                extra['strong_proxy'] = {'source': 'profiles/r06_strong_scaling_proxy.json (scripts/latency_scaling.py: bench.py --batch <per-GPU share> on one GPU; '
                                                   'instances are independent, one gather per rollout)',
                                         'ms_per_step_by_share': {str(r_['batch']): r_['ms_per_step'] for r_ in rows_},
                                         'predicted_speedup_by_gpus': {('%g' % r_['gpus_at_4096_total']): r_['predicted_strong_speedup'] for r_ in rows_}}
        line = {
            'metric': metric,
            'value': total / elapsed, 'unit': 'instance-steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / max(args.steps, 1), 'higher_is_better': True,
            'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': workload, 'name': cfg,
                       'batch_per_gpu': B, 'batch_total': B_total, 'horizon': sorted({sb.N for sb in subs}) if cfg == 'c3' else N,
                       'controller': controller, 'streams_per_gpu': S, 'qp_form': qp_form,
                       'hip_graphs': bool(subs[0].graph is not None),
                       'batched_steps_per_s': args.steps / elapsed, 'mean_ipm_iterations': mean_iter,
                       'failed_instance_steps': fails},
            # host side of the loop: time this rank's Python spent enqueueing one step (all sub-batches) against
            # the step itself, and the cores it may use -- eight ranks of an 8-GPU run share one host
            'host_issue_ms_per_step': 1e3 * host_issue / max(args.steps, 1), 'usable_cores': usable_cores(),
            'roofline': roof, 'cpu_baseline': cpu, 'survey_window': survey,
        }
        line.update(extra)
        if log_check is not None:
            line['log_check'] = log_check
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + '\n').encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
