// kernels_callers.hpp -- the callers on either side of the solve: warm-start
// roll-out and shift, feasibility predicates, plant step.  All embarrassingly parallel over instances.
#pragma once
#include "device_model.hpp"

namespace smpc {

// guessCorrection (controller.py:226-231): one thread per (instance, joint), sequential in k.  `mask` (or null): instances
// with mask[b] == 0 are left alone (the policy layer: instances that do not step their controller this time).
// zero_flag (may be null): one int32 this launch resets -- smpc_policy_step's any_abort, which a later kernel of the same step
// raises with atomics (a hipMemsetAsync would be one more launch in a chain of short ones)
// ok_fill (may be null): [B] verdicts of the state test that follows the solve, set to "ok" here (one fill launch fewer in the chain)
__global__ void k_guess_correction(int B, int N, int nq, double dt, double* __restrict__ xg,
                                   const double* __restrict__ ug, const uint8_t* __restrict__ mask, int32_t* __restrict__ zero_flag,
                                   int32_t* __restrict__ ok_fill) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && zero_flag) *zero_flag = 0;
    if (t >= (long)B * nq) return;
    const long b = t / nq;
    const int i = (int)(t % nq);
    if (ok_fill && i == 0) ok_fill[b] = 1;
    if (mask && !mask[b]) return;
    const int nx = 2 * nq;
    double* x = xg + b * (N + 1) * nx;
    const double* u = ug + b * N * nq;
    double q = x[i], v = x[nq + i];
    const double c = 0.5 * dt * dt;
    for (int k = 0; k < N; k++) {
        const double uk = u[k * nq + i];
        q = q + dt * v + c * uk;
        v = v + dt * uk;
        x[(k + 1) * nx + i] = q;
        x[(k + 1) * nx + nq + i] = v;
    }
}

// provideControl (controller.py:169-184): one thread per (instance, column of x or u).  The policy layer passes two masks (or
// null): an instance with stepping[b] == 0 keeps its guess and gets u_other[b] (the driver's PD law); one that stepped but
// aborted (active[b] == 0) keeps its guess and applies its first row (the early `return self.u_guess[0], True` of
// controller.py:384-385, 483-487).
__global__ void k_provide_control(int B, int N, int nq, const int32_t* __restrict__ accept,
                                  const double* __restrict__ xt, const double* __restrict__ ut,
                                  double* __restrict__ xg, double* __restrict__ ug, double* __restrict__ u_apply,
                                  const uint8_t* __restrict__ stepping, const uint8_t* __restrict__ active,
                                  const double* __restrict__ u_other) {
    const int nx = 2 * nq, ncol = nx + nq;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * ncol) return;
    const long b = t / ncol;
    const int c = (int)(t % ncol);
    const bool acc = accept[b] != 0;
    const bool stp = !stepping || stepping[b], act = stp && (!active || active[b]);
    if (!act) {
        if (c >= nx) u_apply[b * nq + c - nx] = stp ? ug[b * N * nq + c - nx] : u_other[b * nq + c - nx];
        return;
    }
    if (c < nx) {
        double* x = xg + b * (N + 1) * nx;
        const double* src = acc ? xt + b * (N + 1) * nx : x;
        for (int k = 0; k < N; k++) x[k * nx + c] = src[(k + 1) * nx + c];
        x[N * nx + c] = x[(N - 1) * nx + c];
    } else {
        const int i = c - nx;
        double* u = ug + b * N * nq;
        const double* src = acc ? ut + b * N * nq : u;
        u_apply[b * nq + i] = src[i];
        for (int k = 0; k + 1 < N; k++) u[k * nq + i] = src[(k + 1) * nq + i];
        if (N > 1) u[(N - 1) * nq + i] = u[(N - 2) * nq + i];
        else u[i] = src[i];
    }
}

// checkStateConstraints over trajectories (env_model.py:170-173, 236-243): bounds with tolerance + collision rows within
// the check bounds.  One thread per (instance, node); instance verdicts are AND-ed with an atomic.
template <int NQ>
__global__ void k_check_nodes(const smpc_problem_desc* __restrict__ D, int B, int n_nodes, const double* __restrict__ x,
                              const double* __restrict__ x_min, const double* __restrict__ x_max, double tol_x,
                              const double* __restrict__ row_lb, const double* __restrict__ row_ub,
                              int32_t* __restrict__ state_ok, int coll_nodes) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * n_nodes) return;
    constexpr int NX = 2 * NQ;
    const double* xk = x + t * NX;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < NX; i++) ok = ok && (xk[i] >= x_min[i] - tol_x) && (xk[i] <= x_max[i] + tol_x);
    // (the reference's checkCollision returns after the first row of a trajectory, env_model.py:238-243: callers that keep
    //  that quirk collision-test the leading coll_nodes nodes only)
    if ((int)(t % n_nodes) >= coll_nodes) {
        if (!ok) atomicAnd(&state_ok[t / n_nodes], 0);
        return;
    }
    double q[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) q[i] = xk[i];
    Mat3<double> Rw[NQ];
    Vec3<double> pw[NQ], zw[NQ];
    fk_world<NQ>(D->joints, q, Rw, pw, zw);
    for (int r = 0; r < D->n_rows; r++) {
        const smpc_row& row = D->rows[r];
        double v;
        switch (row.kind) {
        case SMPC_ROW_SEG_FIXEDSEG:
            v = segment_dist2<NQ>(point_with_jacobian<NQ>(D->points[row.pa], Rw, pw, zw),
                                  point_with_jacobian<NQ>(D->points[row.pb], Rw, pw, zw), dv_const<NQ>(row.C),
                                  dv_const<NQ>(row.D)).v;
            break;
        case SMPC_ROW_SEG_SEG:
            v = segment_dist2<NQ>(point_with_jacobian<NQ>(D->points[row.pa], Rw, pw, zw),
                                  point_with_jacobian<NQ>(D->points[row.pb], Rw, pw, zw),
                                  point_with_jacobian<NQ>(D->points[row.pc], Rw, pw, zw),
                                  point_with_jacobian<NQ>(D->points[row.pd], Rw, pw, zw)).v;
            break;
        case SMPC_ROW_SEG_POINT:
            v = ball_segment_dist2<NQ>(point_with_jacobian<NQ>(D->points[row.pa], Rw, pw, zw),
                                       point_with_jacobian<NQ>(D->points[row.pb], Rw, pw, zw), row.len2,
                                       dv_const<NQ>(row.C)).v;
            break;
        case SMPC_ROW_POINT_POINT: {
            DV3<NQ> w = point_with_jacobian<NQ>(D->points[row.pa], Rw, pw, zw) - dv_const<NQ>(row.C);
            v = dot(w, w).v;
            break;
        }
        default: {
            DV3<NQ> P = point_with_jacobian<NQ>(D->points[row.pa], Rw, pw, zw);
            v = (row.axis == 0 ? P.x.v : (row.axis == 1 ? P.y.v : P.z.v)) - row.offset;
            break;
        }
        }
        ok = ok && (row_lb[r] <= v) && (v <= row_ub[r]);
    }
    if (!ok) atomicAnd(&state_ok[t / n_nodes], 0);
}

// safe-set acceptance test (safe_set.py:61-68): g(x, alpha) within [-tol, 1e6 + tol]
template <int NQ>
__global__ void k_check_nn(const smpc_problem_desc* __restrict__ D, int M, const double* __restrict__ x, double alpha,
                           double tol_safe, const float* __restrict__ y, int32_t* __restrict__ nn_ok,
                           const int32_t* __restrict__ idx, const int32_t* __restrict__ m_live) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    if (m_live && m >= *m_live) return;
    const long node = idx ? idx[m] : m;        // (compacted list of nodes, kernels_mlp.hpp mode 3; verdicts land at the nodes)
    const double* xk = x + (size_t)node * 2 * NQ;
    const int nd = D->nn_dof;
    double vn2 = 0.0;
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const double v = i < nd ? xk[NQ + i] + (i == 0 ? D->nn_eps : 0.0) : 0.0;
        vn2 += v * v;
    }
    const double g = (double)y[m] * (100.0 - alpha) / 100.0 - sqrt(vn2);
    nn_ok[node] = (g >= -tol_safe) && (g <= 1e6 + tol_safe);
}

// NaiveController.step's bookkeeping (controller.py:279-283): fails = status == 0 ? 0 : fails + 1; accept = fails == 0
__global__ void k_accept(int B, const int32_t* __restrict__ status, int32_t* __restrict__ fails, int32_t* __restrict__ accept) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int f = status[b] == 0 ? 0 : fails[b] + 1;
    fails[b] = f;
    accept[b] = f == 0;
}

// plant step AdamModel.integrate (env_model.py:192-206).  NQ + 2 lanes per instance, one inverse-dynamics call each: lane 0
// tau(q, qd, u), lane 1 the bias h(q, qd), lanes 2.. the columns of M (unit accelerations, no velocity, no gravity) -- the
// same code with different inputs, so the wavefront does not diverge; lane 0 then solves M a = clamp(tau + noise) - h.
template <int NQ>
__global__ __launch_bounds__(64) void k_plant_step(const smpc_problem_desc* __restrict__ D, int B, const double* __restrict__ x,
                             const double* __restrict__ u, const smpc_joint* __restrict__ joints_noisy,
                             const double* __restrict__ tau_noise, double* __restrict__ x_next,
                             double* __restrict__ u_eff) {
    constexpr int NX = 2 * NQ, RL = NQ + 2, PER = 64 / RL;
    __shared__ double sm[PER][RL][NQ];
    const int li = threadIdx.x / RL, role = threadIdx.x % RL;
    const int b = blockIdx.x * PER + li;
    const bool live = li < PER && b < B;
    const int bb = live ? b : 0;
    const smpc_joint* J = joints_noisy ? joints_noisy + (size_t)bb * NQ : D->joints;
    const double* xb = x + (size_t)bb * NX;
    double q[NQ], qd[NQ], qdd[NQ], out[NQ];
    const bool dyn = role < 2;
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        q[i] = xb[i];
        qd[i] = dyn ? xb[NQ + i] : 0.0;
        qdd[i] = role == 0 ? u[(size_t)bb * NQ + i] : (role - 2 == i ? 1.0 : 0.0);
    }
    const double g[3] = {dyn ? D->gravity[0] : 0.0, dyn ? D->gravity[1] : 0.0, dyn ? D->gravity[2] : 0.0};
    rnea_world<NQ, double>(J, g, q, qd, qdd, out);
    if (live) {
#pragma unroll
        for (int i = 0; i < NQ; i++) sm[li][role][i] = out[i];
    }
    __syncthreads();
    if (!live || role != 0) return;
    double rhs[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        double t = sm[li][0][i] + (tau_noise ? tau_noise[(size_t)b * NQ + i] : 0.0);
        const double tm = J[i].tau_max;
        t = fmin(fmax(t, -tm), tm);
        rhs[i] = t - sm[li][1][i];
    }
    // M is symmetric positive definite (column j of M sits in sm[li][2 + j]): Cholesky solve
    double L[NQ][NQ];
#pragma unroll
    for (int j = 0; j < NQ; j++) {
        double d = sm[li][2 + j][j];
#pragma unroll
        for (int t = 0; t < j; t++) d = fma(-L[j][t], L[j][t], d);
        const double lj = sqrt(d);
        L[j][j] = lj;
#pragma unroll
        for (int i = j + 1; i < NQ; i++) {
            double v = 0.5 * (sm[li][2 + j][i] + sm[li][2 + i][j]);
#pragma unroll
            for (int t = 0; t < j; t++) v = fma(-L[i][t], L[j][t], v);
            L[i][j] = v / lj;
        }
    }
    double y[NQ], acc[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        double v = rhs[i];
#pragma unroll
        for (int t = 0; t < i; t++) v = fma(-L[i][t], y[t], v);
        y[i] = v / L[i][i];
    }
#pragma unroll
    for (int i = NQ - 1; i >= 0; i--) {
        double v = y[i];
#pragma unroll
        for (int t = i + 1; t < NQ; t++) v = fma(-L[t][i], acc[t], v);
        acc[i] = v / L[i][i];
    }
    const double dt = D->dt, c = 0.5 * dt * dt;
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        x_next[(size_t)b * NX + i] = q[i] + dt * qd[i] + c * acc[i];
        x_next[(size_t)b * NX + NQ + i] = qd[i] + dt * acc[i];
        if (u_eff) u_eff[(size_t)b * NQ + i] = acc[i];
    }
}

// sums over a batch, accumulated across calls (a closed loop's counters stay on the device):
// acc[0] += sum of IPM iterations, acc[1] += solves with status != 0, acc[2] += solves
// (one wavefront per block, like every short kernel of the loop: the reduction below is a wave-wide xor-shuffle)
__global__ __launch_bounds__(64) void k_accumulate_stats(int B, const int32_t* __restrict__ status, const int32_t* __restrict__ qp_iter,
                                                          unsigned long long* __restrict__ acc) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long it = 0, bad = 0, n = 0;
    if (b < B) {
        it = qp_iter ? (unsigned long long)max(qp_iter[b], 0) : 0ull;
        bad = status[b] != 0;
        n = 1;
    }
    for (int o = 32; o > 0; o >>= 1) {
        it += __shfl_xor(it, o);
        bad += __shfl_xor(bad, o);
        n += __shfl_xor(n, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], it);
        atomicAdd(&acc[1], bad);
        atomicAdd(&acc[2], n);
    }
}

}  // namespace smpc
