// device_model.hpp -- rigid-body device functions for gfx950 (HIP, FP64).
//
// World-frame spatial-vector formulation: every quantity is expressed in world coordinates about the world origin,
// so the recursion carries no per-link rotation of tangents and the backward sweep collapses into
//   tau_j = S_j . sum_{l >= j} f_l
// accumulated on the fly (no stored per-link forces).  Replaces the adam/CasADi graphs of the reference
// (src/safe_mpc/env_model.py:40-45, 80-83, 92-95, 131-163) -- N1/N2 in SURVEY section 2.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>

#include "../../include/smpc.h"

namespace smpc {

// ---- node records in HBM: interleaved tiles of EV_TILE nodes --------------------------------------------------------------------
// The linearisation kernels are thread-per-node, and a thread that stores one double into its own node's 2.6 KB record makes the
// wave write 64 scattered 8-byte pieces: measured, HBM then sees 4 x the bytes (446 MB per launch for 112 MB of torque
// Jacobians: writes go out in 32-byte sectors; profiles/r02_kernel_sheet_before_tiles.txt).  So the records of EV_TILE
// consecutive nodes are interleaved element by element -- element f of node n sits at
// (n / EV_TILE) * EV_TILE * EV_D + f * EV_TILE + n % EV_TILE -- and EV_TILE neighbouring lanes fill whole sectors per store.
// k_qp_setup loads a tile with one block (EV_TILE half-waves, one node each) and takes it apart in LDS; smpc_eval_nodes returns
// plain smpc_node_eval records (k_ev_untile).  EV_TILE = 4 (one 32-byte sector per store group): with 8 the producers gain
// nothing more, and k_qp_setup's 46 KB blocks no longer fit next to the resident QP wavefronts (148 KB of LDS per CU) -- in the
// three-stream closed loop it ran 2.2 x longer and the step 3 % slower; 2 / 4 / 8: 3.89 / 3.85 / 4.02 ms per step against 3.90
// with plain records (three alternations in one session).
constexpr int EV_D = (int)(sizeof(smpc_node_eval) / sizeof(double));
#ifndef SMPC_EV_TILE
#define SMPC_EV_TILE 4
#endif
constexpr int EV_TILE = SMPC_EV_TILE;
#define SMPC_EV_OFF(field) ((int)(offsetof(smpc_node_eval, field) / sizeof(double)))
__host__ __device__ inline size_t ev_tiles(size_t nodes) { return (nodes + EV_TILE - 1) / EV_TILE; }
// pointer to element 0 of node n's record; element f is at [f * EV_TILE]
__device__ __forceinline__ double* ev_node(double* base, long n) { return base + (n / EV_TILE) * (EV_TILE * (long)EV_D) + n % EV_TILE; }
__device__ __forceinline__ const double* ev_node(const double* base, long n) { return base + (n / EV_TILE) * (EV_TILE * (long)EV_D) + n % EV_TILE; }


// ---- scalar with one tangent direction ------------------------------------------------------------------------------
struct D1 {
    double v, d;
    __device__ __forceinline__ D1() : v(0.0), d(0.0) {}
    __device__ __forceinline__ D1(double a) : v(a), d(0.0) {}
    __device__ __forceinline__ D1(double a, double b) : v(a), d(b) {}
};
__device__ __forceinline__ D1 operator+(D1 a, D1 b) { return D1(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ D1 operator-(D1 a, D1 b) { return D1(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ D1 operator-(D1 a) { return D1(-a.v, -a.d); }
__device__ __forceinline__ D1 operator*(D1 a, D1 b) { return D1(a.v * b.v, fma(a.d, b.v, a.v * b.d)); }
__device__ __forceinline__ D1 operator*(D1 a, double b) { return D1(a.v * b, a.d * b); }
__device__ __forceinline__ D1 operator*(double a, D1 b) { return D1(a * b.v, a * b.d); }
__device__ __forceinline__ double value(double a) { return a; }
__device__ __forceinline__ double value(D1 a) { return a.v; }
__device__ __forceinline__ void sincos_t(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ void sincos_t(D1 x, D1* s, D1* c) {
    double sv, cv;
    sincos(x.v, &sv, &cv);
    *s = D1(sv, cv * x.d);
    *c = D1(cv, -sv * x.d);
}

template <class T> struct Vec3 {
    T x, y, z;
    __device__ __forceinline__ Vec3() : x(0.0), y(0.0), z(0.0) {}
    __device__ __forceinline__ Vec3(T a, T b, T c) : x(a), y(b), z(c) {}
};
template <class T> __device__ __forceinline__ Vec3<T> operator+(Vec3<T> a, Vec3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> __device__ __forceinline__ Vec3<T> operator-(Vec3<T> a, Vec3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> __device__ __forceinline__ Vec3<T> operator*(Vec3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T> __device__ __forceinline__ T dot(Vec3<T> a, Vec3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> __device__ __forceinline__ Vec3<T> cross(Vec3<T> a, Vec3<T> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class T> struct Mat3 {
    T m[9];
};
template <class T> __device__ __forceinline__ Vec3<T> mul(const Mat3<T>& R, Vec3<T> v) {
    return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
            R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
template <class T> __device__ __forceinline__ Vec3<T> mulc(const Mat3<T>& R, const double* v) {
    return {R.m[0] * v[0] + R.m[1] * v[1] + R.m[2] * v[2], R.m[3] * v[0] + R.m[4] * v[1] + R.m[5] * v[2],
            R.m[6] * v[0] + R.m[7] * v[1] + R.m[8] * v[2]};
}

// R <- R * R0 * Rot(axis, q)   (R0 and axis are plain constants of the joint table)
template <class T> __device__ __forceinline__ void advance_rotation(Mat3<T>& R, const smpc_joint& J, T q) {
    T s, c;
    sincos_t(q, &s, &c);
    T v = T(1.0) - c;
    const double ax = J.axis[0], ay = J.axis[1], az = J.axis[2];
    T Q[9];
    Q[0] = c + v * (ax * ax);      Q[1] = v * (ax * ay) - s * az; Q[2] = v * (ax * az) + s * ay;
    Q[3] = v * (ay * ax) + s * az; Q[4] = c + v * (ay * ay);      Q[5] = v * (ay * az) - s * ax;
    Q[6] = v * (az * ax) - s * ay; Q[7] = v * (az * ay) + s * ax; Q[8] = c + v * (az * az);
    // A = R * R0
    T A[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            A[3 * i + j] = R.m[3 * i] * J.R0[j] + R.m[3 * i + 1] * J.R0[3 + j] + R.m[3 * i + 2] * J.R0[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            R.m[3 * i + j] = A[3 * i] * Q[j] + A[3 * i + 1] * Q[3 + j] + A[3 * i + 2] * Q[6 + j];
}

// Inverse dynamics tau = M(q) qdd + h(q, qd) with gravity, forward-only world-frame recursion.
//   spatial velocity  (w, vo): angular velocity and linear velocity of the body-fixed point passing the world origin
//   spatial accel     (wd, ao): its time derivative in world coordinates; base acceleration = -gravity
//   S_i = (z_i, p_i x z_i)
template <int NQ, class T>
__device__ __forceinline__ void rnea_world(const smpc_joint* __restrict__ J, const double* __restrict__ grav, const T* q,
                                           const T* qd, const T* qdd, T* tau) {
    Mat3<T> R;
#pragma unroll
    for (int i = 0; i < 9; i++) R.m[i] = T((i % 4 == 0) ? 1.0 : 0.0);
    Vec3<T> p, w, vo, wd, ao(T(-grav[0]), T(-grav[1]), T(-grav[2]));
    Vec3<T> Sz[NQ], So[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) tau[i] = T(0.0);
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const smpc_joint& Ji = J[i];
        p = p + mulc(R, Ji.p0);
        advance_rotation(R, Ji, q[i]);
        Vec3<T> z = mulc(R, Ji.axis);
        Vec3<T> so = cross(p, z);
        Sz[i] = z;
        So[i] = so;
        // v_i = v_{i-1} + S qd ;  a_i = a_{i-1} + S qdd + v_i x (S qd)
        Vec3<T> jw = z * qd[i], jo = so * qd[i];
        w = w + jw;
        vo = vo + jo;
        wd = wd + z * qdd[i] + cross(w, jw);
        ao = ao + so * qdd[i] + cross(w, jo) + cross(vo, jw);
        // body inertia in world axes
        Vec3<T> c = p + mulc(R, Ji.com);
        const double* I = Ji.inertia;
        // Ic * x = R (I (R^T x))
        auto apply_I = [&](Vec3<T> x) {
            T lx = R.m[0] * x.x + R.m[3] * x.y + R.m[6] * x.z;
            T ly = R.m[1] * x.x + R.m[4] * x.y + R.m[7] * x.z;
            T lz = R.m[2] * x.x + R.m[5] * x.y + R.m[8] * x.z;
            Vec3<T> y(lx * I[0] + ly * I[1] + lz * I[2], lx * I[1] + ly * I[3] + lz * I[4],
                      lx * I[2] + ly * I[4] + lz * I[5]);
            return mul(R, y);
        };
        T m = T(Ji.mass);
        Vec3<T> l = (vo + cross(w, c)) * m;           // linear momentum
        Vec3<T> k = apply_I(w) + cross(c, l);          // angular momentum about the world origin
        Vec3<T> fl = (ao + cross(wd, c)) * m;          // I a, linear part
        Vec3<T> fn = apply_I(wd) + cross(c, fl);       // I a, moment about the origin
        Vec3<T> n = fn + cross(w, k) + cross(vo, l);   // + v x* (I v)
        Vec3<T> f = fl + cross(w, l);
#pragma unroll
        for (int j = 0; j <= i; j++) tau[j] = tau[j] + dot(Sz[j], n) + dot(So[j], f);
    }
}

// World poses of the actuated link frames: R[i] (row-major), p[i], and world joint axes z[i]
template <int NQ>
__device__ __forceinline__ void fk_world(const smpc_joint* __restrict__ J, const double* q, Mat3<double>* Rw,
                                         Vec3<double>* pw, Vec3<double>* zw) {
    Mat3<double> R;
#pragma unroll
    for (int i = 0; i < 9; i++) R.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
    Vec3<double> p;
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        p = p + mulc(R, J[i].p0);
        advance_rotation(R, J[i], q[i]);
        Rw[i] = R;
        pw[i] = p;
        zw[i] = mulc(R, J[i].axis);
    }
}

// ---- value + NQ tangents (directions = joints), used for the clamped distance expressions ------------------------------
template <int NQ> struct DQ {
    double v;
    double d[NQ];
};
template <int NQ> __device__ __forceinline__ DQ<NQ> dq_const(double a) {
    DQ<NQ> r;
    r.v = a;
#pragma unroll
    for (int i = 0; i < NQ; i++) r.d[i] = 0.0;
    return r;
}
template <int NQ> __device__ __forceinline__ DQ<NQ> operator+(const DQ<NQ>& a, const DQ<NQ>& b) {
    DQ<NQ> r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < NQ; i++) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <int NQ> __device__ __forceinline__ DQ<NQ> operator-(const DQ<NQ>& a, const DQ<NQ>& b) {
    DQ<NQ> r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < NQ; i++) r.d[i] = a.d[i] - b.d[i];
    return r;
}
template <int NQ> __device__ __forceinline__ DQ<NQ> operator*(const DQ<NQ>& a, const DQ<NQ>& b) {
    DQ<NQ> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < NQ; i++) r.d[i] = fma(a.d[i], b.v, a.v * b.d[i]);
    return r;
}
template <int NQ> __device__ __forceinline__ DQ<NQ> operator/(const DQ<NQ>& a, const DQ<NQ>& b) {
    DQ<NQ> r;
    double inv = 1.0 / b.v;
    r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < NQ; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
// clamp to [0,1] with CasADi's tie rule (fmin: first argument wins at x<=y; fmax: at x>=y)
template <int NQ> __device__ __forceinline__ DQ<NQ> clamp01(const DQ<NQ>& t) {
    DQ<NQ> m = (t.v <= 1.0) ? t : dq_const<NQ>(1.0);
    return (m.v >= 0.0) ? m : dq_const<NQ>(0.0);
}
template <int NQ> struct DV3 {
    DQ<NQ> x, y, z;
};
template <int NQ> __device__ __forceinline__ DV3<NQ> operator-(const DV3<NQ>& a, const DV3<NQ>& b) {
    return {a.x - b.x, a.y - b.y, a.z - b.z};
}
template <int NQ> __device__ __forceinline__ DQ<NQ> dot(const DV3<NQ>& a, const DV3<NQ>& b) {
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
template <int NQ> __device__ __forceinline__ DV3<NQ> scale(const DV3<NQ>& a, const DQ<NQ>& s) {
    return {a.x * s, a.y * s, a.z * s};
}
template <int NQ> __device__ __forceinline__ DV3<NQ> dv_const(const double* c) {
    return {dq_const<NQ>(c[0]), dq_const<NQ>(c[1]), dq_const<NQ>(c[2])};
}

// world position of a robot point and its Jacobian  dP/dq_j = z_j x (P - p_j)  for j <= link
template <int NQ>
__device__ __forceinline__ DV3<NQ> point_with_jacobian(const smpc_point& P, const Mat3<double>* Rw,
                                                       const Vec3<double>* pw, const Vec3<double>* zw) {
    DV3<NQ> out = dv_const<NQ>(P.local);
    if (P.link < 0) return out;
    Vec3<double> w = pw[P.link] + mulc(Rw[P.link], P.local);
    out.x.v = w.x; out.y.v = w.y; out.z.v = w.z;
#pragma unroll
    for (int j = 0; j < NQ; j++) {
        if (j <= P.link) {
            Vec3<double> Jc = cross(zw[j], w - pw[j]);
            out.x.d[j] = Jc.x; out.y.d[j] = Jc.y; out.z.d[j] = Jc.z;
        }
    }
    return out;
}

// squared segment-segment distance exactly as the reference writes it (utils.py:94-113)
template <int NQ>
__device__ __forceinline__ DQ<NQ> segment_dist2(const DV3<NQ>& A, const DV3<NQ>& B, const DV3<NQ>& Cc, const DV3<NQ>& Dd) {
    DV3<NQ> ab = B - A, cd = Dd - Cc, ac = Cc - A;
    DQ<NQ> R = dot(ab, cd), S1 = dot(ab, ac), D1 = dot(ab, ab), S2 = dot(cd, ac), D2 = dot(cd, cd);
    DQ<NQ> t = (S1 * D2 - S2 * R) / (D1 * D2 - (R * R + dq_const<NQ>(1e-5)));
    t = clamp01(t);
    DQ<NQ> u = (t * R - S2) / D2;
    u = clamp01(u);
    t = (u * R + S1) / D1;
    t = clamp01(t);
    DV3<NQ> w = scale(ab, t) - scale(cd, u) - ac;
    return dot(w, w);
}
// squared point-segment distance (utils.py:115-118); note fmin(fmax(.,0),1) order
template <int NQ>
__device__ __forceinline__ DQ<NQ> ball_segment_dist2(const DV3<NQ>& A, const DV3<NQ>& B, double len2, const DV3<NQ>& P) {
    DQ<NQ> t = dot(P - A, B - A) / dq_const<NQ>(len2);
    DQ<NQ> m = (t.v >= 0.0) ? t : dq_const<NQ>(0.0);
    t = (m.v <= 1.0) ? m : dq_const<NQ>(1.0);
    DV3<NQ> w = P - DV3<NQ>{A.x + (B.x - A.x) * t, A.y + (B.y - A.y) * t, A.z + (B.z - A.z) * t};
    return dot(w, w);
}

}  // namespace smpc
