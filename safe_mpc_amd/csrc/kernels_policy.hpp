// kernels_policy.hpp -- the accept / reject / abort automata of the reference's controller classes and the safe-abort loop of
// its driver script, one thread per instance, all state in HBM.  A closed-loop step of a policy is then a
// dozen kernel launches around the solve instead of ~130 array operations.  The readable statement of the same automata is the
// numpy path of safe_mpc_amd/controller.py and closed_loop.py, which the GPU tests run side by side with these kernels.
#pragma once
#include "device_model.hpp"

namespace smpc {

// RecedingController._set_flags (controller.py:452-469): the safe-set row is switched on at node r of each instance and at the
// terminal node; RealReceding.step (controller.py:524-536): node r is boxed to the previously planned state +- tube instead.
// One thread per (instance, node).
__global__ void k_policy_pre(int B, int N, int nx, int kind, const uint8_t* __restrict__ stepping,
                             const int64_t* __restrict__ r_all, double* __restrict__ p, const double* __restrict__ xg,
                             const double* __restrict__ lo_st, const double* __restrict__ hi_st, double tube,
                             double* __restrict__ lo_b, double* __restrict__ hi_b, int32_t* __restrict__ zero_flag,
                             int32_t* __restrict__ ok_fill) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && zero_flag) *zero_flag = 0;       // (RealReceding has no guessCorrection launch to do this)
    if (t >= (long)B * (N + 1)) return;
    const long b = t / (N + 1);
    const int k = (int)(t - b * (N + 1));
    if (ok_fill && k == 0) ok_fill[b] = 1;         // (... nor to start the state test's verdicts at "ok")
    if (stepping && !stepping[b]) return;
    const int r = (int)r_all[b];
    if (kind == SMPC_POLICY_RECEDING) {
        const bool on = k == N || (k == r && r < N) || k == 0;   // (node 0 keeps the default flag: it never carries the row)
        p[t * SMPC_NP + 4] = on ? 1.0 : -1.0;
    } else {
        // model bounds everywhere (terminal node: its own), except the tube at node r
        const bool at_r = k == r && r < N;
        const int rc = (r < N - 1 ? r : N - 1) + 1;
        const double* c = xg + (b * (N + 1) + rc) * nx;
        for (int i = 0; i < nx; i++) {
            lo_b[t * nx + i] = at_r ? c[i] - tube : lo_st[(size_t)k * nx + i];
            hi_b[t * nx + i] = at_r ? c[i] + tube : hi_st[(size_t)k * nx + i];
        }
    }
}

// solve() feeds every node its column of the reference trajectory, p[i][0:3] = cost.traj[:, current_step + i]
// (controller.py:153-156; cost_definition.py:30-31, 89: traj has n_steps + 1 + N columns).  One thread per (instance, node);
// launched only when the caller gave a trajectory.
__global__ void k_policy_traj(int B, int N, const uint8_t* __restrict__ stepping, const int64_t* __restrict__ current_step,
                              const double* __restrict__ traj, long traj_len, double* __restrict__ p) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * (N + 1)) return;
    const long b = t / (N + 1);
    const int k = (int)(t - b * (N + 1));
    if (stepping && !stepping[b]) return;
    long c = (long)current_step[b] + k;
    c = c < 0 ? 0 : (c < traj_len ? c : traj_len - 1);
    for (int i = 0; i < 3; i++) p[t * SMPC_NP + i] = traj[(size_t)i * traj_len + c];
}

// The receding policies move r to the last safe node of the NEW trajectory, looking at nodes r + 2 .. N only
// (controller.py:491-494) -- in steady state one or two nodes per instance.  This builds the list of those (instance, node) pairs
// so that the network is evaluated there and nowhere else.  r is taken after this step's decrement / abort reset, exactly as
// k_policy_post will see it.  One thread per (instance, node).
__global__ void k_policy_safe_list(int B, int N, int abort_flag, const uint8_t* __restrict__ stepping, const int64_t* __restrict__ r_all,
                                   int32_t* __restrict__ idx, int32_t* __restrict__ count) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * (N + 1)) return;
    const long b = t / (N + 1);
    const int k = (int)(t - b * (N + 1));
    if (stepping && !stepping[b]) return;
    int64_t r = r_all[b];
    if (abort_flag) {
        r -= 1;
        if (r == 0) r += N;
    } else {
        r -= r > 0 ? 1 : 0;
    }
    if (k >= r + 2) idx[atomicAdd(count, 1)] = (int32_t)t;
}

// What follows the solve in <Controller>.step, per instance:
//   NAIVE        controller.py:274-284   fails = status == 0 ? 0 : fails + 1
//   STATE_CHECK  controller.py:651-661   the same with ok = status == 0 and checkStateConstraints(x_temp)
//   STWA         controller.py:375-388   first failure stores x_guess[-2] as viable state; N - 1 failures in a row: abort
//   RECEDING /   controller.py:471-498   receding index r, abort at r == 0 (x_guess[1] becomes the viable state), r moved to
//   REAL_RECEDING                         the last safe node of the new trajectory
// accept[b] = fails == 0 afterwards (provideControl, controller.py:172); active[b] = stepped and did not abort (its guess is
// shifted); abort_out[b]; *any_abort |= abort.  Instances with stepping[b] == 0 are left alone.
__global__ void k_policy_post(int B, int N, int nx, int kind, int abort_flag, const uint8_t* __restrict__ stepping,
                              const int32_t* __restrict__ status, const int32_t* __restrict__ state_ok,
                              const int32_t* __restrict__ safe, const double* __restrict__ xg, int64_t* __restrict__ fails,
                              int64_t* __restrict__ current_step, int64_t* __restrict__ r_all, double* __restrict__ x_viable,
                              int32_t* __restrict__ accept, uint8_t* __restrict__ active, uint8_t* __restrict__ abort_out,
                              int32_t* __restrict__ any_abort, int32_t* __restrict__ zero_cnt) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0 && zero_cnt) *zero_cnt = 0;     // (the length of the safe-set test's node list: every kernel that read it has finished)
    if (b >= B) return;
    if (stepping && !stepping[b]) {
        accept[b] = 0;
        active[b] = 0;
        abort_out[b] = 0;
        return;
    }
    const double* xgb = xg + (size_t)b * (N + 1) * nx;
    int64_t f = fails[b];
    bool abort = false;
    if (kind == SMPC_POLICY_NAIVE || kind == SMPC_POLICY_STATE_CHECK) {
        const bool ok = status[b] == 0 && (kind == SMPC_POLICY_NAIVE || state_ok[b] != 0);
        f = ok ? 0 : f + 1;
    } else if (kind == SMPC_POLICY_STWA) {
        const bool ok = status[b] == 0 && state_ok[b] != 0;
        if (!ok && f == 0)
            for (int i = 0; i < nx; i++) x_viable[(size_t)b * nx + i] = xgb[(size_t)(N - 1) * nx + i];
        abort = !ok && f == N - 1;
        f = ok ? 0 : (abort ? f : f + 1);
    } else {
        int64_t r = r_all[b];
        if (abort_flag) {
            r -= 1;
            abort = r == 0;
        } else {
            r -= r > 0 ? 1 : 0;
        }
        if (abort) {
            for (int i = 0; i < nx; i++) x_viable[(size_t)b * nx + i] = xgb[(size_t)nx + i];
            r += N;
        }
        const bool ok = status[b] == 0 && state_ok[b] != 0 && !abort;
        // r <- i - 1 for the largest i in r+2 .. N whose node passes the safe-set test (controller.py:491-494)
        int best = -1;
        for (int i = (int)r + 2; i <= N; i++)
            if (safe[(size_t)b * (N + 1) + i]) best = i;
        if (ok && best >= 0) r = best - 1;
        r_all[b] = r;
        f = abort ? f : (ok ? 0 : f + 1);
    }
    fails[b] = f;
    current_step[b] += abort ? 0 : 1;
    accept[b] = f == 0;
    active[b] = !abort;
    abort_out[b] = abort;
    if (abort) atomicOr(any_abort, 1);
}

// The driver's safe-abort tracking (scripts/mpc.py:130-151): instances that follow the backup trajectory get the PD law on
// its nodes, then hold its last state until they are at rest, then resume MPC.  Writes u_other (the control of the instances
// that do not step their controller), the stepping mask, and the receding-index log row of this step.
__global__ void k_loop_pre(int B, int nq, int Nb, const double* __restrict__ x_cur, const uint8_t* __restrict__ alive,
                           uint8_t* __restrict__ sa, int64_t* __restrict__ ja, const double* __restrict__ x_abort,
                           const double* __restrict__ u_abort, const int64_t* __restrict__ r_all, const int64_t* __restrict__ step,
                           int64_t* __restrict__ r_log, double* __restrict__ u_other, uint8_t* __restrict__ stepping,
                           const uint8_t* __restrict__ pending, uint8_t* __restrict__ resumed) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int nx = 2 * nq;
    const double kp = 1.0, kd = 1e2, kd_hold = 3e2;                                      // mpc.py:97, 143
    const double* x = x_cur + (size_t)b * nx;
    const bool in_abort = sa[b] && alive[b];
    const int64_t j = ja[b];
    const bool follow = in_abort && j < Nb, hold = in_abort && j >= Nb;
    bool resume = hold;
    for (int i = 0; i < nq; i++) resume = resume && x[nq + i] < 5e-3;                    // mpc.py:138 (signed, as there)
    const bool still = hold && !resume;
    const int idx = (int)(j < Nb - 1 ? j : Nb - 1);
    const double* xa = x_abort + ((size_t)b * (Nb + 1) + idx) * nx;
    const double* ua = u_abort + ((size_t)b * Nb + idx) * nq;
    const double* xe = x_abort + ((size_t)b * (Nb + 1) + Nb) * nx;
    for (int i = 0; i < nq; i++) {
        double u = 0.0;
        if (follow) u = ua[i] - (kp * (x[i] - xa[i]) + kd * (x[nq + i] - xa[nq + i]));
        if (still) u = -(kp * (x[i] - xe[i]) + kd_hold * (x[nq + i] - xe[nq + i]));
        u_other[(size_t)b * nq + i] = u;
    }
    const bool s_new = sa[b] && !resume;
    sa[b] = s_new;
    ja[b] = j + (in_abort ? 1 : 0);
    // (pending: raised abort in the previous step, its backup OCP is still being solved on another stream -- whatever the
    //  outcome, it does not step: it will either follow the backup trajectory or be dead)
    const bool stp = alive[b] && !s_new && !(pending && pending[b]);
    stepping[b] = stp;
    if (resumed) resumed[b] = resume && stp;
    if (r_log) r_log[(size_t)step[0] * B + b] = (stp && r_all) ? r_all[b] : -1;
}

// After the plant step (scripts/mpc.py:240-264): logs, outcome of the new state, next current state.
__global__ void k_loop_post(int B, int nq, const double* __restrict__ u, const double* __restrict__ x_next,
                            const int32_t* __restrict__ ok_next, const int64_t* __restrict__ step, double* __restrict__ x_log,
                            double* __restrict__ u_log, uint8_t* __restrict__ alive, uint8_t* __restrict__ collided,
                            int64_t* __restrict__ last_x, int64_t* __restrict__ last_u, double* __restrict__ x_cur) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int nx = 2 * nq;
    const int64_t j = step[0];
    for (int i = 0; i < nq; i++) u_log[((size_t)j * B + b) * nq + i] = u[(size_t)b * nq + i];
    for (int i = 0; i < nx; i++) x_log[((size_t)(j + 1) * B + b) * nx + i] = x_next[(size_t)b * nx + i];
    const bool was = alive[b];
    const bool bad = was && !ok_next[b];
    if (bad) {                              // the failing state stays logged (mpc.py:246-264)
        last_x[b] = j + 1;
        last_u[b] = j;
        collided[b] = 1;
        alive[b] = 0;
    }
    if (was && !bad)
        for (int i = 0; i < nx; i++) x_cur[(size_t)b * nx + i] = x_next[(size_t)b * nx + i];
}

__global__ void k_step_advance(int64_t* __restrict__ step) { step[0] += 1; }

// scripts/mpc.py:137-141 vs :161-190: an abort raised on the step an instance resumed MPC opens no abort event -- the
// instance is in safe abort again with its old backup trajectory and its abort clock still running (smpc.h,
// smpc_loop_classify_aborts).
__global__ void k_loop_classify_aborts(int B, int quirks, const uint8_t* __restrict__ resumed, uint8_t* __restrict__ sa,
                                       uint8_t* __restrict__ abort, int32_t* __restrict__ any_event) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B || !abort[b]) return;
    if (quirks && resumed && resumed[b]) {
        sa[b] = 1;
        abort[b] = 0;
    } else {
        atomicOr(any_event, 1);
    }
}

// The abort events of the PREVIOUS step (scripts/mpc.py:161-190), applied once their backup OCPs are solved: one thread per
// event c (instance rows[c]).  Solved: the instance follows the backup trajectory from now on (this step applies its first node
// with the PD law, as the tracking branch of mpc.py:130-136 would); failed: the instance is lost at the step of the event.
__global__ void k_loop_apply_backup(int n_c, int nq, int Nb, const int64_t* __restrict__ rows, const int32_t* __restrict__ status_c,
                                    const double* __restrict__ x_c, const double* __restrict__ u_c, const double* __restrict__ x_cur,
                                    const int64_t* __restrict__ step, uint8_t* __restrict__ alive, uint8_t* __restrict__ sa,
                                    uint8_t* __restrict__ collided, uint8_t* __restrict__ viable, int64_t* __restrict__ ja,
                                    int64_t* __restrict__ last_x, int64_t* __restrict__ last_u, double* __restrict__ x_abort,
                                    double* __restrict__ u_abort, double* __restrict__ u, uint8_t* __restrict__ pending) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_c) return;
    const int nx = 2 * nq;
    const int64_t b = rows[c];
    pending[b] = 0;
    if (status_c[c] != 0) {
        const int64_t j = step[0] - 1;      // the step of the event (mpc.py:186-190 breaks there)
        collided[b] = 1;
        alive[b] = 0;
        last_x[b] = j;
        last_u[b] = j;
        return;
    }
    const double* xs = x_c + (size_t)c * (Nb + 1) * nx;
    const double* us = u_c + (size_t)c * Nb * nq;
    double* xa = x_abort + (size_t)b * (Nb + 1) * nx;
    double* ua = u_abort + (size_t)b * Nb * nq;
    for (int i = 0; i < (Nb + 1) * nx; i++) xa[i] = xs[i];
    for (int i = 0; i < Nb * nq; i++) ua[i] = us[i];
    sa[b] = 1;
    viable[b] = viable[b] < 255 ? viable[b] + 1 : 255;
    ja[b] = 1;
    const double kp = 1.0, kd = 1e2;
    const double* x = x_cur + (size_t)b * nx;
    for (int i = 0; i < nq; i++) u[(size_t)b * nq + i] = us[i] - (kp * (x[i] - xs[i]) + kd * (x[nq + i] - xs[nq + i]));
}

}  // namespace smpc
