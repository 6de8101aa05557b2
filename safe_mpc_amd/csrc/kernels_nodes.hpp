// kernels_nodes.hpp -- HOT LOOP A: per-node linearisation (what acados evaluates through CasADi-generated C,
// SURVEY 3.2): torque row and its Jacobians, EE cost terms, collision rows.  Work items are (instance, node) pairs,
// flattened so that a wavefront covers 64 consecutive nodes: inputs x[B][N+1][nx] / u[B][N][nu] are then read as one
// contiguous span per wave.
#pragma once
#include "device_model.hpp"
#include "rnea_deriv.hpp"

namespace smpc {

// ---- K1: geometry + cost, one thread per (b, k), k = 0..N ---------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(128) void k_node_geometry(const smpc_problem_desc* __restrict__ D, int B, int N,
                                                       const double* __restrict__ xg, const double* __restrict__ p,
                                                       smpc_node_eval* __restrict__ out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * (N + 1)) return;
    constexpr int NX = 2 * NQ;
    const double* x = xg + t * NX;
    const double* pk = p + t * SMPC_NP;
    smpc_node_eval* o = out + t;

    double q[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) q[i] = x[i];
    Mat3<double> Rw[NQ];
    Vec3<double> pw[NQ], zw[NQ];
    fk_world<NQ>(D->joints, q, Rw, pw, zw);

    // EE point, cost gradient and Hessian (cost_definition.py:69-96)
    {
        DV3<NQ> ee = point_with_jacobian<NQ>(D->points[D->ee_point], Rw, pw, zw);
        o->ee[0] = ee.x.v; o->ee[1] = ee.y.v; o->ee[2] = ee.z.v;
        if (D->cost_kind == SMPC_COST_REACH) {
            const double Q2 = 2.0 * D->Q;
            const double dx = ee.x.v - pk[0], dy = ee.y.v - pk[1], dz = ee.z.v - pk[2];
            const int link = D->points[D->ee_point].link;
            Vec3<double> P(ee.x.v, ee.y.v, ee.z.v), del(dx, dy, dz);
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                o->cost_grad_q[i] = Q2 * (ee.x.d[i] * dx + ee.y.d[i] * dy + ee.z.d[i] * dz);
#pragma unroll
                for (int j = 0; j < NQ; j++) {
                    double h = ee.x.d[i] * ee.x.d[j] + ee.y.d[i] * ee.y.d[j] + ee.z.d[i] * ee.z.d[j];
                    if (D->hessian == SMPC_HESS_EXACT) {
                        // d2 P / dq_i dq_j = z_min x (z_max x (P - p_max)) for min(i,j) <= max(i,j) <= link
                        const int lo = i < j ? i : j, hi = i < j ? j : i;
                        if (hi <= link) {
                            Vec3<double> Jhi = cross(zw[hi], P - pw[hi]);
                            h += dot(del, cross(zw[lo], Jhi));
                        }
                    }
                    o->cost_hess_qq[i * NQ + j] = Q2 * h;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                o->cost_grad_q[i] = 0.0;
#pragma unroll
                for (int j = 0; j < NQ; j++) o->cost_hess_qq[i * NQ + j] = 0.0;
            }
        }
    }
    // collision rows (env_model.py:263-316)
    const int nrows = D->n_rows;
    for (int r = 0; r < nrows; r++) {
        const smpc_row& row = D->rows[r];
        DQ<NQ> v;
        switch (row.kind) {
        case SMPC_ROW_SEG_FIXEDSEG:
            v = segment_dist2<NQ>(point_with_jacobian<NQ>(D->points[row.pa], Rw, pw, zw),
                                  point_with_jacobian<NQ>(D->points[row.pb], Rw, pw, zw), dv_const<NQ>(row.C),
                                  dv_const<NQ>(row.D));
            break;
        case SMPC_ROW_SEG_SEG:
            v = segment_dist2<NQ>(point_with_jacobian<NQ>(D->points[row.pa], Rw, pw, zw),
                                  point_with_jacobian<NQ>(D->points[row.pb], Rw, pw, zw),
                                  point_with_jacobian<NQ>(D->points[row.pc], Rw, pw, zw),
                                  point_with_jacobian<NQ>(D->points[row.pd], Rw, pw, zw));
            break;
        case SMPC_ROW_SEG_POINT:
            v = ball_segment_dist2<NQ>(point_with_jacobian<NQ>(D->points[row.pa], Rw, pw, zw),
                                       point_with_jacobian<NQ>(D->points[row.pb], Rw, pw, zw), row.len2,
                                       dv_const<NQ>(row.C));
            break;
        case SMPC_ROW_POINT_POINT: {
            DV3<NQ> w = point_with_jacobian<NQ>(D->points[row.pa], Rw, pw, zw) - dv_const<NQ>(row.C);
            v = dot(w, w);
            break;
        }
        default: {
            DV3<NQ> P = point_with_jacobian<NQ>(D->points[row.pa], Rw, pw, zw);
            v = row.axis == 0 ? P.x : (row.axis == 1 ? P.y : P.z);
            v.v -= row.offset;
            break;
        }
        }
        o->row_val[r] = v.v;
#pragma unroll
        for (int i = 0; i < NQ; i++) o->row_grad[r * NQ + i] = v.d[i];
    }
    // fields owned by other kernels start from a defined value
    o->nn_val = 0.0;
#pragma unroll
    for (int i = 0; i < NX; i++) o->nn_grad[i] = 0.0;
    const int k = (int)(t % (N + 1));
    if (k == N) {
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            o->tau[i] = 0.0;
#pragma unroll
            for (int j = 0; j < NQ; j++) {
                o->M[i * NQ + j] = 0.0;
                o->dtau_dq[i * NQ + j] = 0.0;
                o->dtau_dv[i * NQ + j] = 0.0;
            }
        }
    }
}

// ---- K2: torque row and Jacobians, one thread per (b, k < N) ---------------------------------------------------------------
// One forward and one backward recursion give tau, M = dtau/du, dtau/dq and dtau/dqd in closed form (rnea_deriv.hpp); every
// entry is stored once, straight from the recursion into the node's record.  (Round 1 ran 3 NQ single-tangent dual-number
// passes per node in 3 NQ threads: 0.48 ms per 4096 x 30 nodes.)
template <int NQ>
__global__ __launch_bounds__(64) void k_node_torque(const smpc_problem_desc* __restrict__ D, int B, int N,
                                                    const double* __restrict__ xg, const double* __restrict__ ug,
                                                    smpc_node_eval* __restrict__ out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * N) return;
    constexpr int NX = 2 * NQ;
    const long b = t / N;
    const int k = (int)(t % N);
    const double* x = xg + (b * (N + 1) + k) * NX;
    const double* u = ug + t * NQ;
    smpc_node_eval* o = out + b * (N + 1) + k;
    double q[NQ], qd[NQ], qdd[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        q[i] = x[i];
        qd[i] = x[NQ + i];
        qdd[i] = u[i];
    }
    rd::rnea_with_derivatives<NQ>(D->joints, D->gravity, q, qd, qdd, o->tau, o->M, o->dtau_dq, o->dtau_dv);
}

// Layout experiment (diagnostic entry smpc_debug_torque_layout): the same work with the outputs written
// field-major across the batch axis -- out[f][node], consecutive lanes to consecutive addresses -- instead of into each node's
// 2.6 KB record.
template <int NQ>
__global__ __launch_bounds__(64) void k_node_torque_soa(const smpc_problem_desc* __restrict__ D, int B, int N,
                                                        const double* __restrict__ xg, const double* __restrict__ ug,
                                                        double* __restrict__ out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)B * N;
    if (t >= n) return;
    constexpr int NX = 2 * NQ;
    const long b = t / N;
    const int k = (int)(t % N);
    const double* x = xg + (b * (N + 1) + k) * NX;
    const double* u = ug + t * NQ;
    double q[NQ], qd[NQ], qdd[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        q[i] = x[i];
        qd[i] = x[NQ + i];
        qdd[i] = u[i];
    }
    double* o = out + t;
    rd::rnea_with_derivatives<NQ>(D->joints, D->gravity, q, qd, qdd, o, o + (long)NQ * n, o + (long)(NQ + NQ * NQ) * n,
                                  o + (long)(NQ + 2 * NQ * NQ) * n, n);
}

}  // namespace smpc
