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
__device__ __forceinline__ void node_geometry(const smpc_problem_desc* __restrict__ D, int B, int N, const double* __restrict__ xg,
                                              const double* __restrict__ p, double* __restrict__ out, const long t) {
    if (t >= (long)B * (N + 1)) return;
    constexpr int NX = 2 * NQ;
    const double* x = xg + t * NX;
    const double* pk = p + t * SMPC_NP;
    double* const o = ev_node(out, t);          // (interleaved tile: element f at o[f * EV_TILE])
    auto put = [&](int f, double v) { o[f * EV_TILE] = v; };
    constexpr int F_EE = SMPC_EV_OFF(ee), F_CG = SMPC_EV_OFF(cost_grad_q), F_CH = SMPC_EV_OFF(cost_hess_qq), F_RV = SMPC_EV_OFF(row_val),
                  F_RG = SMPC_EV_OFF(row_grad), F_NV = SMPC_EV_OFF(nn_val), F_NG = SMPC_EV_OFF(nn_grad), F_TAU = SMPC_EV_OFF(tau),
                  F_M = SMPC_EV_OFF(M), F_DQ = SMPC_EV_OFF(dtau_dq), F_DV = SMPC_EV_OFF(dtau_dv);

    double q[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) q[i] = x[i];
    Mat3<double> Rw[NQ];
    Vec3<double> pw[NQ], zw[NQ];
    fk_world<NQ>(D->joints, q, Rw, pw, zw);

    // EE point, cost gradient and Hessian (cost_definition.py:69-96)
    {
        DV3<NQ> ee = point_with_jacobian<NQ>(D->points[D->ee_point], Rw, pw, zw);
        put(F_EE, ee.x.v); put(F_EE + 1, ee.y.v); put(F_EE + 2, ee.z.v);
        if (D->cost_kind == SMPC_COST_REACH) {
            const double Q2 = 2.0 * D->Q;
            const double dx = ee.x.v - pk[0], dy = ee.y.v - pk[1], dz = ee.z.v - pk[2];
            const int link = D->points[D->ee_point].link;
            Vec3<double> P(ee.x.v, ee.y.v, ee.z.v), del(dx, dy, dz);
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                put(F_CG + i, Q2 * (ee.x.d[i] * dx + ee.y.d[i] * dy + ee.z.d[i] * dz));
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
                    put(F_CH + i * NQ + j, Q2 * h);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                put(F_CG + i, 0.0);
#pragma unroll
                for (int j = 0; j < NQ; j++) put(F_CH + i * NQ + j, 0.0);
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
        put(F_RV + r, v.v);
#pragma unroll
        for (int i = 0; i < NQ; i++) put(F_RG + r * NQ + i, v.d[i]);
    }
    // fields owned by other kernels start from a defined value
    put(F_NV, 0.0);
#pragma unroll
    for (int i = 0; i < NX; i++) put(F_NG + i, 0.0);
    const int k = (int)(t % (N + 1));
    if (k == N) {
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            put(F_TAU + i, 0.0);
#pragma unroll
            for (int j = 0; j < NQ; j++) {
                put(F_M + i * NQ + j, 0.0);
                put(F_DQ + i * NQ + j, 0.0);
                put(F_DV + i * NQ + j, 0.0);
            }
        }
    }
}

// ---- K2: torque row and Jacobians, one thread per (b, k) ------------------------------------------------------------------------
// One forward and one backward recursion give tau, M = dtau/du, dtau/dq and dtau/dqd in closed form (rnea_deriv.hpp); every
// entry is stored once, straight from the recursion into the node's (interleaved) record.  The thread index runs over ALL nodes,
// so that EV_TILE neighbouring lanes always belong to one tile; the terminal nodes (no torque row) idle.  (Round 1 ran 3 NQ
// single-tangent dual-number passes per node in 3 NQ threads: 0.48 ms per 4096 x 30 nodes.)
template <int NQ>
__device__ __forceinline__ void node_torque(const smpc_problem_desc* __restrict__ D, int B, int N, const double* __restrict__ xg,
                                            const double* __restrict__ ug, double* __restrict__ out, const long t) {
    if (t >= (long)B * (N + 1)) return;
    constexpr int NX = 2 * NQ;
    const long b = t / (N + 1);
    const int k = (int)(t - b * (N + 1));
    if (k == N) return;
    const double* x = xg + t * NX;
    const double* u = ug + (b * N + k) * NQ;
    double* const o = ev_node(out, t);
    double q[NQ], qd[NQ], qdd[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        q[i] = x[i];
        qd[i] = x[NQ + i];
        qdd[i] = u[i];
    }
    rd::rnea_with_derivatives<NQ>(D->joints, D->gravity, q, qd, qdd, o + SMPC_EV_OFF(tau) * EV_TILE, o + SMPC_EV_OFF(M) * EV_TILE,
                                  o + SMPC_EV_OFF(dtau_dq) * EV_TILE, o + SMPC_EV_OFF(dtau_dv) * EV_TILE, (long)EV_TILE);
}

// The linearisation kernel: geometry + cost, then torque row and Jacobians, one thread per node, ONE launch.  As two kernels
// (rounds 1-2) each of them waited for SIMD room of its own in the three-stream closed loop -- a wavefront of either needs a
// SIMD that holds no QP wavefront (310 / 512 registers per lane) -- about 0.13 ms per wait against 0.04 / 0.05 ms of work.
template <int NQ>
__global__ __launch_bounds__(64) void k_node_linearise(const smpc_problem_desc* __restrict__ D, int B, int N,
                                                       const double* __restrict__ xg, const double* __restrict__ ug,
                                                       const double* __restrict__ p, double* __restrict__ out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    node_geometry<NQ>(D, B, N, xg, p, out, t);
    node_torque<NQ>(D, B, N, xg, ug, out, t);
}

// smpc_eval_nodes: the interleaved tiles back into plain records, one thread per (node, element)
__global__ void k_ev_untile(long nodes, const double* __restrict__ tiled, double* __restrict__ plain) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nodes * EV_D) return;
    const long n = t / EV_D;
    const int f = (int)(t - n * EV_D);
    plain[t] = ev_node(tiled, n)[f * EV_TILE];
}

}  // namespace smpc
