// rnea_deriv.hpp -- inverse dynamics tau = M(q) qdd + h(q, qd) of a serial chain of revolute joints TOGETHER WITH its
// closed-form first derivatives dtau/dq, dtau/dqd and M = dtau/dqdd, in one forward and one backward recursion.
// Replaces what the reference obtains from CasADi's algorithmic differentiation of adam's RNEA graph
// (src/safe_mpc/env_model.py:80-83: tau = mass(q) u + bias(q, qd); its Jacobian enters the torque rows of the stage QP).
//
// Formulation: world-frame spatial algebra (every quantity expressed in world axes about the world origin).
//   motion vectors  m = (w; v)   angular part, linear velocity of the body-fixed point passing the origin
//   force vectors   f = (n; f)   moment about the origin, force
//   S_k = (z_k; p_k x z_k)                                  joint axis as a motion vector
//   v_i = v_{i-1} + S_i qd_i,  a_i = a_{i-1} + S_i qdd_i + v_i x S_i qd_i,  a_0 = (0; -g)
//   f_i = Y_i a_i + v_i x* (Y_i v_i),   tau_j = S_j . F_j,   F_j = sum_{l >= j} f_l
// Derivatives (first principles; with psi_k = v_k x S_k, chi_k = a_k x S_k - psi_k x v_k and the per-body operator
//   B_l m = Y_l (m x v_l) + m x* (Y_l v_l) + v_l x* (Y_l m),  composites Yc_m = sum_{l >= m} Y_l, Bc_m = sum_{l >= m} B_l):
//   d tau_j / d qdd_k = S_j . Yc_max(j,k) S_k
//   d tau_j / d qd_k  = S_j . c_{max(j,k), k},                                c_{m,k} = Bc_m S_k + 2 Yc_m psi_k
//   d tau_j / d q_k   = S_j . d_{j,k}                       (k < j)
//                     = S_j . (d_{k,k} + S_k x* F_k)        (k >= j),         d_{m,k} = Bc_m psi_k + Yc_m chi_k
// (the S_k x* F_j term of the k < j case cancels against d S_j / d q_k = S_k x S_j).  Identities used: d S_l / d q_k =
// S_k x S_l (k < l), d Y_l / d q_k = S_k x* Y_l - Y_l S_k x (k <= l), d v_l / d q_k = S_k x (v_l - v_k),
// d a_l / d q_k = S_k x (a_l - a_k) + psi_k x (v_l - v_k), d a_l / d qd_k = 2 psi_k + S_k x v_l  (l >= k).
// The test oracle differentiates a link-frame recursion with dual numbers instead: nothing is shared with this file.
//
// The same source compiles for the device (hipcc) and for the host (g++, tests/test_rnea_deriv.py checks it there).
#pragma once
#include "../../include/smpc.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SMPC_FN __host__ __device__ __forceinline__
#else
#include <cmath>
#define SMPC_FN inline
#endif

namespace smpc {
namespace rd {

struct V3 {
    double x, y, z;
};
SMPC_FN V3 v3(double x, double y, double z) { return V3{x, y, z}; }
SMPC_FN V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
SMPC_FN V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
SMPC_FN V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
SMPC_FN double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
SMPC_FN V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct SV {   // spatial vector: (a; l) = (angular; linear) for motions, (moment; force) for forces
    V3 a, l;
};
SMPC_FN SV operator+(SV p, SV q) { return SV{p.a + q.a, p.l + q.l}; }
SMPC_FN SV operator-(SV p, SV q) { return SV{p.a - q.a, p.l - q.l}; }
SMPC_FN SV operator*(SV p, double s) { return SV{p.a * s, p.l * s}; }
SMPC_FN SV mxm(SV m1, SV m2) { return SV{cross(m1.a, m2.a), cross(m1.a, m2.l) + cross(m1.l, m2.a)}; }   // m1 x m2
SMPC_FN SV mxf(SV m, SV f) { return SV{cross(m.a, f.a) + cross(m.l, f.l), cross(m.a, f.l)}; }            // m x* f
SMPC_FN double mdotf(SV m, SV f) { return dot(m.a, f.a) + dot(m.l, f.l); }

// rigid-body (or composite) inertia about the world origin: mass, first moment m c, rotational inertia about the origin
struct Inertia {
    double m;
    V3 mc;
    double I[6];   // xx xy xz yy yz zz
};
SMPC_FN SV apply(const Inertia& Y, SV v) {   // momentum / force of a motion vector
    SV o;
    o.l = v.l * Y.m + cross(v.a, Y.mc);
    o.a = V3{Y.I[0] * v.a.x + Y.I[1] * v.a.y + Y.I[2] * v.a.z, Y.I[1] * v.a.x + Y.I[3] * v.a.y + Y.I[4] * v.a.z,
             Y.I[2] * v.a.x + Y.I[4] * v.a.y + Y.I[5] * v.a.z} + cross(Y.mc, v.l);
    return o;
}

// R <- R * R0 * Rot(axis, q), row-major 3x3
SMPC_FN void advance(double* R, const smpc_joint& J, double q) {
    const double s = sin(q), c = cos(q), v = 1.0 - c;
    const double ax = J.axis[0], ay = J.axis[1], az = J.axis[2];
    const double Q[9] = {c + v * ax * ax,      v * ax * ay - s * az, v * ax * az + s * ay,
                         v * ay * ax + s * az, c + v * ay * ay,      v * ay * az - s * ax,
                         v * az * ax - s * ay, v * az * ay + s * ax, c + v * az * az};
    double A[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) A[3 * i + j] = R[3 * i] * J.R0[j] + R[3 * i + 1] * J.R0[3 + j] + R[3 * i + 2] * J.R0[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R[3 * i + j] = A[3 * i] * Q[j] + A[3 * i + 1] * Q[3 + j] + A[3 * i + 2] * Q[6 + j];
}
SMPC_FN V3 rot(const double* R, const double* v) {
    return V3{R[0] * v[0] + R[1] * v[1] + R[2] * v[2], R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
              R[6] * v[0] + R[7] * v[1] + R[8] * v[2]};
}

// tau[NQ]; M, dq, dv row-major NQ x NQ with leading dimension NQ:  M[j*NQ+k] = d tau_j / d qdd_k, dq[..] = d tau_j / d q_k, ...
// es: element stride of the four outputs (1: packed as above; the layout experiment of DESIGN.md section 4 passes the node count)
template <int NQ>
SMPC_FN void rnea_with_derivatives(const smpc_joint* J, const double* grav, const double* q, const double* qd,
                                   const double* qdd, double* tau, double* M, double* dq, double* dv, long es = 1) {
    SV S[NQ], v[NQ], a[NQ];
    Inertia Y[NQ];
    // ---- forward: kinematics and body inertias in world coordinates ---------------------------------------------------
    {
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        V3 p = v3(0, 0, 0);
        SV vp{v3(0, 0, 0), v3(0, 0, 0)}, ap{v3(0, 0, 0), v3(-grav[0], -grav[1], -grav[2])};
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const smpc_joint& Ji = J[i];
            p = p + rot(R, Ji.p0);
            advance(R, Ji, q[i]);
            const V3 z = rot(R, Ji.axis);
            S[i] = SV{z, cross(p, z)};
            const SV jv = S[i] * qd[i];
            v[i] = vp + jv;
            a[i] = ap + S[i] * qdd[i] + mxm(v[i], jv);
            vp = v[i];
            ap = a[i];
            // inertia about the origin: R I R^T + m (|c|^2 1 - c c^T)
            const V3 c = p + rot(R, Ji.com);
            const double* I = Ji.inertia;
            double T[9];   // T = R * I_local
#pragma unroll
            for (int r = 0; r < 3; r++) {
                T[3 * r + 0] = R[3 * r] * I[0] + R[3 * r + 1] * I[1] + R[3 * r + 2] * I[2];
                T[3 * r + 1] = R[3 * r] * I[1] + R[3 * r + 1] * I[3] + R[3 * r + 2] * I[4];
                T[3 * r + 2] = R[3 * r] * I[2] + R[3 * r + 1] * I[4] + R[3 * r + 2] * I[5];
            }
            const double m = Ji.mass, cc = dot(c, c);
            Inertia& Yi = Y[i];
            Yi.m = m;
            Yi.mc = c * m;
            Yi.I[0] = T[0] * R[0] + T[1] * R[1] + T[2] * R[2] + m * (cc - c.x * c.x);
            Yi.I[1] = T[0] * R[3] + T[1] * R[4] + T[2] * R[5] - m * c.x * c.y;
            Yi.I[2] = T[0] * R[6] + T[1] * R[7] + T[2] * R[8] - m * c.x * c.z;
            Yi.I[3] = T[3] * R[3] + T[4] * R[4] + T[5] * R[5] + m * (cc - c.y * c.y);
            Yi.I[4] = T[3] * R[6] + T[4] * R[7] + T[5] * R[8] - m * c.y * c.z;
            Yi.I[5] = T[6] * R[6] + T[7] * R[7] + T[8] * R[8] + m * (cc - c.z * c.z);
        }
    }
    // ---- backward: composites, torques and the three Jacobians --------------------------------------------------------
    Inertia Yc;
    Yc.m = 0.0;
    Yc.mc = v3(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 6; i++) Yc.I[i] = 0.0;
    double Bc[36];   // composite of B_l, row-major 6x6: rows (moment xyz, force xyz), columns (angular xyz, linear xyz)
#pragma unroll
    for (int i = 0; i < 36; i++) Bc[i] = 0.0;
    SV F{v3(0, 0, 0), v3(0, 0, 0)};
    auto applyB = [&](SV m) {
        const double x[6] = {m.a.x, m.a.y, m.a.z, m.l.x, m.l.y, m.l.z};
        double y[6];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 6; c++) s += Bc[6 * r + c] * x[c];
            y[r] = s;
        }
        return SV{v3(y[0], y[1], y[2]), v3(y[3], y[4], y[5])};
    };
#pragma unroll
    for (int m = NQ - 1; m >= 0; m--) {
        const Inertia& Ym = Y[m];
        const SV vm = v[m];
        const SV hm = apply(Ym, vm);
        F = F + apply(Ym, a[m]) + mxf(vm, hm);
        Yc.m += Ym.m;
        Yc.mc = Yc.mc + Ym.mc;
#pragma unroll
        for (int i = 0; i < 6; i++) Yc.I[i] += Ym.I[i];
        // B_m column by column: B e = Y (e x v) + e x* h + v x* (Y e)
#pragma unroll
        for (int c = 0; c < 6; c++) {
            SV e{v3(c == 0, c == 1, c == 2), v3(c == 3, c == 4, c == 5)};
            const SV col = apply(Ym, mxm(e, vm)) + mxf(e, hm) + mxf(vm, apply(Ym, e));
            Bc[0 * 6 + c] += col.a.x; Bc[1 * 6 + c] += col.a.y; Bc[2 * 6 + c] += col.a.z;
            Bc[3 * 6 + c] += col.l.x; Bc[4 * 6 + c] += col.l.y; Bc[5 * 6 + c] += col.l.z;
        }
        tau[m * es] = mdotf(S[m], F);
#pragma unroll
        for (int k = 0; k <= m; k++) {
            const SV Sk = S[k];
            const SV psi = mxm(v[k], Sk);
            const SV chi = mxm(a[k], Sk) - mxm(psi, v[k]);
            const SV yS = apply(Yc, Sk);
            const SV cv = applyB(Sk) + apply(Yc, psi) * 2.0;
            SV dd = applyB(psi) + apply(Yc, chi);
            const double mm = mdotf(S[m], yS);
            M[(m * NQ + k) * es] = mm;
            M[(k * NQ + m) * es] = mm;
            if (k < m) {
                dv[(m * NQ + k) * es] = mdotf(S[m], cv);
                dq[(m * NQ + k) * es] = mdotf(S[m], dd);
            } else {
                dd = dd + mxf(Sk, F);
#pragma unroll
                for (int j = 0; j <= m; j++) {
                    dv[(j * NQ + m) * es] = mdotf(S[j], cv);
                    dq[(j * NQ + m) * es] = mdotf(S[j], dd);
                }
            }
        }
    }
}

}  // namespace rd
}  // namespace smpc
