/*
 * smpc_oracle.cpp -- CPU restatement of the safe-MPC RTI hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (safe_mpc_amd/, the HIP library) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (idra-lab/safe-mpc) evaluates this path inside third-party native code that is
 * neither vendored in /root/reference nor installable here -- acados (+HPIPM/BLASFEO, version unpinned, README.md:13),
 * CasADi 3.7.0, l4casadi 1.3.0, adam_robotics 0.3.1 (requirements.txt:1-3) -- and its repository holds no golden
 * vectors, assertions or fixtures for it.  This file restates the *published algorithms* behind the
 * reference's call sites and is pinned instead by first-principles tests (tests/test_oracle_*.py): closed forms,
 * finite differences, torch autograd for the MLP, scipy for the QP, KKT residuals.
 *
 * What follows which reference line:
 *   double integrator f_disc ............ src/safe_mpc/env_model.py:63-67
 *   tau = M(q) u + h(q,qd) .............. env_model.py:80-83   (adam mass_matrix / bias_force == RNEA, fixed base)
 *   EE point t_glob ..................... env_model.py:92-95
 *   capsule / sphere / plane rows ....... env_model.py:246-319, utils.py:94-124
 *   NN safe-set row ..................... safe_set.py:26-43, 72-104; switch utils.py:207-210
 *   cost ................................ cost_definition.py:34-100
 *   OCP assembly, options ............... controller.py:13-125, parser.py:115-121,139, config.yaml:15-21
 *   solve() ............................. controller.py:136-167
 *   guessCorrection / provideControl .... controller.py:226-231, 169-184
 *   feasibility predicates .............. env_model.py:170-243, safe_set.py:61-68
 *   plant step .......................... env_model.py:192-206
 *
 * Formulation differences from the GPU code are deliberate (they make agreement meaningful): this file uses the
 * classic link-frame Newton-Euler recursion, forward-mode dual numbers for every derivative, dense per-stage
 * matrices and a textbook Riccati recursion; the HIP kernels use world-frame spatial algebra, closed-form
 * derivatives of the recursion (no dual numbers) and wave-cooperative LDS tiles.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/smpc.h"
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int MAXQ = SMPC_MAX_NQ;
constexpr int MAXT = 3 * MAXQ;  // most tangent directions ever needed: q, qd, u

/* ------------------------------------------------------------------------------------------------------------------
 * forward-mode dual numbers
 * ---------------------------------------------------------------------------------------------------------------- */
struct Dual {
    double v;
    double d[MAXT];
    int n;
    Dual() : v(0), n(0) {}
    Dual(double x) : v(x), n(0) {}
    static Dual var(double x, int idx, int n) {
        Dual r;
        r.v = x;
        r.n = n;
        for (int i = 0; i < n; i++) r.d[i] = 0.0;
        r.d[idx] = 1.0;
        return r;
    }
};
inline int dn(const Dual& a, const Dual& b) { return a.n > b.n ? a.n : b.n; }
inline double dd(const Dual& a, int i) { return i < a.n ? a.d[i] : 0.0; }
inline Dual operator+(const Dual& a, const Dual& b) {
    Dual r; r.v = a.v + b.v; r.n = dn(a, b);
    for (int i = 0; i < r.n; i++) r.d[i] = dd(a, i) + dd(b, i);
    return r;
}
inline Dual operator-(const Dual& a, const Dual& b) {
    Dual r; r.v = a.v - b.v; r.n = dn(a, b);
    for (int i = 0; i < r.n; i++) r.d[i] = dd(a, i) - dd(b, i);
    return r;
}
inline Dual operator-(const Dual& a) {
    Dual r; r.v = -a.v; r.n = a.n;
    for (int i = 0; i < r.n; i++) r.d[i] = -a.d[i];
    return r;
}
inline Dual operator*(const Dual& a, const Dual& b) {
    Dual r; r.v = a.v * b.v; r.n = dn(a, b);
    for (int i = 0; i < r.n; i++) r.d[i] = dd(a, i) * b.v + a.v * dd(b, i);
    return r;
}
inline Dual operator/(const Dual& a, const Dual& b) {
    Dual r; r.v = a.v / b.v; r.n = dn(a, b);
    for (int i = 0; i < r.n; i++) r.d[i] = (dd(a, i) - r.v * dd(b, i)) / b.v;
    return r;
}
inline Dual sin(const Dual& a) {
    Dual r; r.v = std::sin(a.v); r.n = a.n; double c = std::cos(a.v);
    for (int i = 0; i < r.n; i++) r.d[i] = c * a.d[i];
    return r;
}
inline Dual cos(const Dual& a) {
    Dual r; r.v = std::cos(a.v); r.n = a.n; double s = -std::sin(a.v);
    for (int i = 0; i < r.n; i++) r.d[i] = s * a.d[i];
    return r;
}
inline Dual sqrt(const Dual& a) {
    Dual r; r.v = std::sqrt(a.v); r.n = a.n;
    for (int i = 0; i < r.n; i++) r.d[i] = 0.5 * a.d[i] / r.v;
    return r;
}
/* CasADi's derivative convention for fmin / fmax: on a tie the first argument carries the derivative
 * (casadi OP_FMIN: d0 = x<=y, OP_FMAX: d0 = x>=y) [EXT-UNVERIFIED] */
inline Dual fmin(const Dual& a, const Dual& b) { return a.v <= b.v ? a : b; }
inline Dual fmax(const Dual& a, const Dual& b) { return a.v >= b.v ? a : b; }
inline double val(double x) { return x; }
inline double val(const Dual& x) { return x.v; }

template <class T> inline T t_sin(const T& x) { return sin(x); }
template <> inline double t_sin<double>(const double& x) { return std::sin(x); }
template <class T> inline T t_cos(const T& x) { return cos(x); }
template <> inline double t_cos<double>(const double& x) { return std::cos(x); }
template <class T> inline T t_sqrt(const T& x) { return sqrt(x); }
template <> inline double t_sqrt<double>(const double& x) { return std::sqrt(x); }
template <class T> inline T t_min(const T& a, const T& b) { return fmin(a, b); }
template <> inline double t_min<double>(const double& a, const double& b) { return a <= b ? a : b; }
template <class T> inline T t_max(const T& a, const T& b) { return fmax(a, b); }
template <> inline double t_max<double>(const double& a, const double& b) { return a >= b ? a : b; }

/* ------------------------------------------------------------------------------------------------------------------
 * 3-vectors / 3x3 matrices over a scalar type
 * ---------------------------------------------------------------------------------------------------------------- */
template <class T> struct V3 {
    T x, y, z;
    V3() : x(0.0), y(0.0), z(0.0) {}
    V3(T a, T b, T c) : x(a), y(b), z(c) {}
    T& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    const T& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <class T> V3<T> operator+(const V3<T>& a, const V3<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> V3<T> operator-(const V3<T>& a, const V3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> V3<T> operator*(const V3<T>& a, const T& s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T> T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> V3<T> cross(const V3<T>& a, const V3<T>& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class T> struct M3 {
    T m[3][3];
    M3() { for (auto& r : m) for (auto& e : r) e = T(0.0); }
    static M3 eye() { M3 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = T(1.0); return r; }
};
template <class T> M3<T> operator*(const M3<T>& a, const M3<T>& b) {
    M3<T> r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        T s(0.0);
        for (int k = 0; k < 3; k++) s = s + a.m[i][k] * b.m[k][j];
        r.m[i][j] = s;
    }
    return r;
}
template <class T> V3<T> operator*(const M3<T>& a, const V3<T>& v) {
    return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
template <class T> V3<T> tmul(const M3<T>& a, const V3<T>& v) {  // a^T v
    return {a.m[0][0] * v.x + a.m[1][0] * v.y + a.m[2][0] * v.z, a.m[0][1] * v.x + a.m[1][1] * v.y + a.m[2][1] * v.z,
            a.m[0][2] * v.x + a.m[1][2] * v.y + a.m[2][2] * v.z};
}
template <class T> V3<T> cv3(const double* p) { return {T(p[0]), T(p[1]), T(p[2])}; }

/* Rodrigues rotation about a unit axis */
template <class T> M3<T> axis_rot(const double* a, const T& th) {
    T c = t_cos(th), s = t_sin(th), one(1.0);
    T v = one - c;
    M3<T> R;
    double ax = a[0], ay = a[1], az = a[2];
    R.m[0][0] = c + v * T(ax * ax);      R.m[0][1] = v * T(ax * ay) - s * T(az); R.m[0][2] = v * T(ax * az) + s * T(ay);
    R.m[1][0] = v * T(ay * ax) + s * T(az); R.m[1][1] = c + v * T(ay * ay);      R.m[1][2] = v * T(ay * az) - s * T(ax);
    R.m[2][0] = v * T(az * ax) - s * T(ay); R.m[2][1] = v * T(az * ay) + s * T(ax); R.m[2][2] = c + v * T(az * az);
    return R;
}
template <class T> M3<T> cm3(const double* p) {
    M3<T> R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = T(p[3 * i + j]);
    return R;
}

/* ------------------------------------------------------------------------------------------------------------------
 * rigid-body algorithms on a serial chain of revolute joints
 * ---------------------------------------------------------------------------------------------------------------- */

/* child-frame orientation relative to the parent link frame: R0 * Rot(axis, q) */
template <class T> M3<T> joint_rot(const smpc_joint& J, const T& q) { return cm3<T>(J.R0) * axis_rot<T>(J.axis, q); }

/* world poses of every actuated link frame */
template <class T> void fk_links(const smpc_joint* J, int nq, const T* q, M3<T>* Rw, V3<T>* pw) {
    M3<T> R = M3<T>::eye();
    V3<T> p;
    for (int i = 0; i < nq; i++) {
        p = p + R * cv3<T>(J[i].p0);
        R = R * joint_rot<T>(J[i], q[i]);
        Rw[i] = R;
        pw[i] = p;
    }
}

template <class T> V3<T> point_world(const smpc_point& P, const M3<T>* Rw, const V3<T>* pw) {
    if (P.link < 0) return cv3<T>(P.local);
    return pw[P.link] + Rw[P.link] * cv3<T>(P.local);
}

/* Recursive Newton-Euler in link frames (Luh-Walker-Paul): tau = M(q) qdd + h(q, qd), gravity included.
 * Same quantity as adam's mass_matrix[6:,6:] @ u + bias_force[6:] for a fixed base (env_model.py:80-83). */
template <class T>
void rnea(const smpc_joint* J, int nq, const double* grav, const T* q, const T* qd, const T* qdd, T* tau) {
    M3<T> R[MAXQ];
    V3<T> w[MAXQ], wd[MAXQ], vd[MAXQ], F[MAXQ], Nn[MAXQ];
    V3<T> w_p, wd_p, vd_p(T(-grav[0]), T(-grav[1]), T(-grav[2]));
    for (int i = 0; i < nq; i++) {
        R[i] = joint_rot<T>(J[i], q[i]);
        V3<T> a = cv3<T>(J[i].axis), p = cv3<T>(J[i].p0), c = cv3<T>(J[i].com);
        V3<T> wi = tmul(R[i], w_p) + a * qd[i];
        V3<T> wdi = tmul(R[i], wd_p) + a * qdd[i] + cross(wi, a * qd[i]);
        V3<T> vdi = tmul(R[i], vd_p + cross(wd_p, p) + cross(w_p, cross(w_p, p)));
        V3<T> vc = vdi + cross(wdi, c) + cross(wi, cross(wi, c));
        const double* I = J[i].inertia;
        M3<T> Im;
        Im.m[0][0] = T(I[0]); Im.m[0][1] = T(I[1]); Im.m[0][2] = T(I[2]);
        Im.m[1][0] = T(I[1]); Im.m[1][1] = T(I[3]); Im.m[1][2] = T(I[4]);
        Im.m[2][0] = T(I[2]); Im.m[2][1] = T(I[4]); Im.m[2][2] = T(I[5]);
        F[i] = vc * T(J[i].mass);
        Nn[i] = Im * wdi + cross(wi, Im * wi);
        w[i] = wi; wd[i] = wdi; vd[i] = vdi;
        w_p = wi; wd_p = wdi; vd_p = vdi;
    }
    V3<T> f_c, n_c;  // force / moment the child exerts back, expressed in the child frame
    for (int i = nq - 1; i >= 0; i--) {
        V3<T> c = cv3<T>(J[i].com);
        V3<T> f = F[i], n = Nn[i] + cross(c, F[i]);
        if (i + 1 < nq) {
            V3<T> fr = R[i + 1] * f_c;
            f = f + fr;
            n = n + R[i + 1] * n_c + cross(cv3<T>(J[i + 1].p0), fr);
        }
        tau[i] = dot(n, cv3<T>(J[i].axis));
        f_c = f; n_c = n;
    }
}

/* tau and its three Jacobians by forward-mode AD over rnea() */
void tau_and_jacobians(const smpc_problem_desc& D, const double* x, const double* u, double* tau, double* M,
                       double* dq, double* dv) {
    int nq = D.nq, nt = 3 * nq;
    Dual q[MAXQ], qd[MAXQ], qdd[MAXQ], t[MAXQ];
    for (int i = 0; i < nq; i++) {
        q[i] = Dual::var(x[i], i, nt);
        qd[i] = Dual::var(x[nq + i], nq + i, nt);
        qdd[i] = Dual::var(u[i], 2 * nq + i, nt);
    }
    rnea<Dual>(D.joints, nq, D.gravity, q, qd, qdd, t);
    for (int i = 0; i < nq; i++) {
        tau[i] = t[i].v;
        for (int j = 0; j < nq; j++) {
            dq[i * nq + j] = t[i].d[j];
            dv[i * nq + j] = t[i].d[nq + j];
            M[i * nq + j] = t[i].d[2 * nq + j];
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * collision rows (env_model.py:263-316, utils.py:94-124)
 * ---------------------------------------------------------------------------------------------------------------- */
template <class T> T segment_dist2(const V3<T>& A, const V3<T>& B, const V3<T>& Cc, const V3<T>& Dd) {
    /* utils.py:94-113, including the 1e-5 regulariser and the clamp order */
    V3<T> ab = B - A, cd = Dd - Cc, ac = Cc - A;
    T R = dot(ab, cd), S1 = dot(ab, ac), D1 = dot(ab, ab), S2 = dot(cd, ac), D2 = dot(cd, cd);
    T one(1.0), zero(0.0);
    T t = (S1 * D2 - S2 * R) / (D1 * D2 - (R * R + T(1e-5)));
    t = t_max(t_min(t, one), zero);
    T uu = (t * R - S2) / D2;
    uu = t_max(t_min(uu, one), zero);
    t = (uu * R + S1) / D1;
    t = t_max(t_min(t, one), zero);
    V3<T> w = ab * t - cd * uu - ac;
    return dot(w, w);
}
template <class T> T ball_segment_dist2(const V3<T>& A, const V3<T>& B, double len2, const V3<T>& P) {
    /* utils.py:115-118 */
    T one(1.0), zero(0.0);
    T t = t_min(t_max(dot(P - A, B - A) / T(len2), zero), one);
    V3<T> w = P - (A + (B - A) * t);
    return dot(w, w);
}
template <class T> T row_value(const smpc_problem_desc& D, const smpc_row& r, const M3<T>* Rw, const V3<T>* pw) {
    switch (r.kind) {
    case SMPC_ROW_SEG_FIXEDSEG:
        return segment_dist2<T>(point_world<T>(D.points[r.pa], Rw, pw), point_world<T>(D.points[r.pb], Rw, pw),
                                cv3<T>(r.C), cv3<T>(r.D));
    case SMPC_ROW_SEG_SEG:
        return segment_dist2<T>(point_world<T>(D.points[r.pa], Rw, pw), point_world<T>(D.points[r.pb], Rw, pw),
                                point_world<T>(D.points[r.pc], Rw, pw), point_world<T>(D.points[r.pd], Rw, pw));
    case SMPC_ROW_SEG_POINT:
        return ball_segment_dist2<T>(point_world<T>(D.points[r.pa], Rw, pw), point_world<T>(D.points[r.pb], Rw, pw),
                                     r.len2, cv3<T>(r.C));
    case SMPC_ROW_POINT_POINT: {
        V3<T> w = point_world<T>(D.points[r.pa], Rw, pw) - cv3<T>(r.C);
        return dot(w, w);
    }
    default: {  // SMPC_ROW_COORD
        V3<T> P = point_world<T>(D.points[r.pa], Rw, pw);
        return P[r.axis] - T(r.offset);
    }
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * MLP in float32 (safe_set.py:26-43; torch evaluates the network in fp32 through l4casadi, safe_set.py:89-94)
 * ---------------------------------------------------------------------------------------------------------------- */
struct Mlp {
    int act = 0;      // SMPC_ACT_* (parser.py:95-102)
    int nlayers = 0;  // number of Linear layers
    int dims[SMPC_MAX_LAYERS + 1] = {0};
    std::vector<float> W[SMPC_MAX_LAYERS], b[SMPC_MAX_LAYERS];
};
inline float gelu_tanh(float a, float* dgelu) {
    /* GELU(approximate='tanh') (parser.py:99) and its derivative */
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float inner = k0 * (a + k1 * a * a * a);
    float th = std::tanh(inner);
    *dgelu = 0.5f * (1.0f + th) + 0.5f * a * (1.0f - th * th) * k0 * (1.0f + 3.0f * k1 * a * a);
    return 0.5f * a * (1.0f + th);
}
/* the activations of parser.py:95-102 as torch.nn defines them: value and derivative */
inline float activation(int act, float a, float* d) {
    switch (act) {
    case SMPC_ACT_RELU: *d = a > 0.0f ? 1.0f : 0.0f; return a > 0.0f ? a : 0.0f;                       /* nn.ReLU */
    case SMPC_ACT_ELU: { float e = std::exp(a); *d = a > 0.0f ? 1.0f : e; return a > 0.0f ? a : e - 1.0f; }   /* nn.ELU, alpha = 1 */
    case SMPC_ACT_TANH: { float t = std::tanh(a); *d = 1.0f - t * t; return t; }                      /* nn.Tanh */
    case SMPC_ACT_SILU: { float s = 1.0f / (1.0f + std::exp(-a)); *d = s * (1.0f + a * (1.0f - s)); return a * s; }   /* nn.SiLU */
    default: return gelu_tanh(a, d);                                                                   /* nn.GELU('tanh') */
    }
}
/* y = net(s), grad = d y / d s; everything in fp32 like torch */
void mlp_eval(const Mlp& net, const float* s, float* y, float* grad) {
    int L = net.nlayers;
    std::vector<float> act[SMPC_MAX_LAYERS + 1], dact[SMPC_MAX_LAYERS];
    act[0].assign(s, s + net.dims[0]);
    for (int l = 0; l < L; l++) {
        int ni = net.dims[l], no = net.dims[l + 1];
        act[l + 1].resize(no);
        dact[l].assign(no, 1.0f);
        for (int o = 0; o < no; o++) {
            float acc = 0.0f;
            const float* w = &net.W[l][(size_t)o * ni];
            for (int i = 0; i < ni; i++) acc = std::fmaf(w[i], act[l][i], acc);
            acc += net.b[l][o];
            if (l + 1 < L) acc = activation(net.act, acc, &dact[l][o]);
            act[l + 1][o] = acc;
        }
    }
    *y = act[L][0];
    /* reverse sweep for the single output */
    std::vector<float> delta(1, 1.0f);
    for (int l = L - 1; l >= 0; l--) {
        int ni = net.dims[l], no = net.dims[l + 1];
        std::vector<float> dz(no), prev(ni, 0.0f);
        for (int o = 0; o < no; o++) dz[o] = delta[o] * dact[l][o];
        for (int o = 0; o < no; o++) {
            const float* w = &net.W[l][(size_t)o * ni];
            for (int i = 0; i < ni; i++) prev[i] = std::fmaf(w[i], dz[o], prev[i]);
        }
        delta.swap(prev);
    }
    for (int i = 0; i < net.dims[0]; i++) grad[i] = delta[i];
}

/* g(x, alpha) = nn(s) (100 - alpha)/100 - |v|  and dg/dx  (safe_set.py:82-94) */
void nn_row(const smpc_problem_desc& D, const Mlp& net, const double* x, double alpha, double* g, double* dgdx) {
    int n = D.nq, m = D.nn_dof;
    double v[MAXQ], vn2 = 0.0;
    for (int i = 0; i < m; i++) {
        v[i] = x[n + i] + (i == 0 ? D.nn_eps : 0.0);  // eps lands on the first velocity only (safe_set.py:83)
        vn2 += v[i] * v[i];
    }
    double vn = std::sqrt(vn2);
    float s[2 * MAXQ], y, gs[2 * MAXQ];
    for (int i = 0; i < m; i++) {
        s[i] = (float)((x[i] - D.nn_mean[i]) / D.nn_std[i]);
        s[m + i] = (float)(v[i] / vn);
    }
    mlp_eval(net, s, &y, gs);
    double kap = (100.0 - alpha) / 100.0;
    *g = (double)y * kap - vn;
    for (int i = 0; i < 2 * n; i++) dgdx[i] = 0.0;
    double gd_dot_v = 0.0;
    for (int j = 0; j < m; j++) gd_dot_v += (double)gs[m + j] * v[j];
    for (int i = 0; i < m; i++) {
        dgdx[i] = kap * (double)gs[i] / D.nn_std[i];
        dgdx[n + i] = kap * ((double)gs[m + i] / vn - v[i] * gd_dot_v / (vn * vn * vn)) - v[i] / vn;
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * one node: everything acados would evaluate through CasADi-generated functions
 * ---------------------------------------------------------------------------------------------------------------- */
bool nn_active(const smpc_problem_desc& D, int k, int N, const double* p) {
    /* casadi_if_else(p[4] > 0, g, (lb+ub)/2) (utils.py:207-210): a switched-off row sits mid-bounds = no row */
    if (D.nn_mode == SMPC_NN_NONE || k == 0) return false;
    if (D.nn_mode == SMPC_NN_TERMINAL && k != N) return false;
    return p[4] > 0.0;
}

void eval_node(const smpc_problem_desc& D, const Mlp* net, int k, int N, const double* x, const double* u,
               const double* p, smpc_node_eval* out) {
    int nq = D.nq;
    std::memset(out, 0, sizeof(*out));
    if (k < N) tau_and_jacobians(D, x, u, out->tau, out->M, out->dtau_dq, out->dtau_dv);

    /* kinematics with tangents wrt q */
    Dual q[MAXQ];
    for (int i = 0; i < nq; i++) q[i] = Dual::var(x[i], i, nq);
    M3<Dual> Rw[MAXQ];
    V3<Dual> pw[MAXQ];
    fk_links<Dual>(D.joints, nq, q, Rw, pw);

    V3<Dual> ee = point_world<Dual>(D.points[D.ee_point], Rw, pw);
    for (int c = 0; c < 3; c++) out->ee[c] = ee[c].v;
    if (D.cost_kind == SMPC_COST_REACH) {
        /* l = Q |ee - ref|^2 ; grad = 2 Q J^T delta ; GN Hessian 2 Q J^T J (cost_definition.py:69-96) */
        double del[3];
        for (int c = 0; c < 3; c++) del[c] = ee[c].v - p[c];
        for (int i = 0; i < nq; i++) {
            double g = 0.0;
            for (int c = 0; c < 3; c++) g += ee[c].d[i] * del[c];
            out->cost_grad_q[i] = 2.0 * D.Q * g;
            for (int j = 0; j < nq; j++) {
                double h = 0.0;
                for (int c = 0; c < 3; c++) h += ee[c].d[i] * ee[c].d[j];
                out->cost_hess_qq[i * nq + j] = 2.0 * D.Q * h;
            }
        }
        if (D.hessian == SMPC_HESS_EXACT) {
            /* + 2 Q sum_c delta_c d2 ee_c / dq_i dq_j : differentiate the gradient direction-wise by a second
             * dual sweep over the analytic point Jacobian J_j = z_j x (ee - p_j) */
            for (int j = 0; j < nq; j++) {
                V3<Dual> z = Rw[j] * cv3<Dual>(D.joints[j].axis);
                if (D.points[D.ee_point].link < j) continue;
                V3<Dual> Jj = cross(z, ee - pw[j]);
                for (int i = 0; i < nq; i++) {
                    double h = 0.0;
                    for (int c = 0; c < 3; c++) h += del[c] * dd(Jj[c], i);
                    out->cost_hess_qq[i * nq + j] += 2.0 * D.Q * h;
                }
            }
        }
    }
    for (int r = 0; r < D.n_rows; r++) {
        Dual v = row_value<Dual>(D, D.rows[r], Rw, pw);
        out->row_val[r] = v.v;
        for (int i = 0; i < nq; i++) out->row_grad[r * nq + i] = dd(v, i);
    }
    if (net && net->nlayers > 0 && nn_active(D, k, N, p)) nn_row(D, *net, x, p[3], &out->nn_val, out->nn_grad);
}

/* ------------------------------------------------------------------------------------------------------------------
 * stage QP + primal-dual interior point with a Riccati factorisation (what HPIPM does for acados, N5 in SURVEY 2)
 * ---------------------------------------------------------------------------------------------------------------- */
constexpr int MAXX = SMPC_MAX_NX, MAXU = SMPC_MAX_NQ, MAXZ = MAXX + MAXU;
constexpr int MAXR = MAXX + MAXU + SMPC_MAX_ROWS + 1;  // box | torque | collision | nn

struct Stage {
    int nu, nr;                 // nu = 0 at the terminal node
    double H[MAXZ][MAXZ];       // cost Hessian in z = [du; dx]
    double g[MAXZ];
    double b[MAXX];             // x_{k+1} = A x_k + B u_k + b
    double Cm[MAXR][MAXZ];      // rows: lo <= C z <= hi
    double lo[MAXR], hi[MAXR];
    bool has_lo[MAXR], has_hi[MAXR];
    double soft[MAXR];          // L1 weight on a lower row (< 0: hard)
    /* IPM state */
    double tl[MAXR], tu[MAXR], ll[MAXR], lu[MAXR], sl[MAXR];
    double dtl[MAXR], dtu[MAXR], dll[MAXR], dlu[MAXR], dsl[MAXR];
    /* Riccati storage */
    double L[MAXU][MAXU], W[MAXU][MAXX], Pb[MAXX], w[MAXU];
    double z[MAXZ], zn[MAXZ];
};

struct QpOpts {
    int max_iter;
    double tol, tol_res, mu0;   /* complementarity / linear-residual exit tolerances */
    int stall_iters;            /* > 0: give up after this many consecutive iterations with a step length below 1/2 */
};

/* returns 0 converged, 2 max-iter, 3 min-step, 4 breakdown (non-PD pivot / NaN), 5 stalled (QpOpts::stall_iters) */
int qp_ipm(std::vector<Stage>& S, int N, int nx, int nu, double dt, const double* dx0, const QpOpts& o, int* iters,
           double* res_out) {
    const int nq = nu;
    const double c = 0.5 * dt * dt;
    const double thr_hard = 3e-2, thr_soft = 1e-1, tau_ftb = 0.995, alpha_min = 1e-12;

    /* A x, A^T x, B u, B^T x for the double integrator (env_model.py:63-67) */
    auto Ax = [&](const double* x, double* y) {
        for (int i = 0; i < nq; i++) { y[i] = x[i] + dt * x[nq + i]; y[nq + i] = x[nq + i]; }
    };
    auto ATx = [&](const double* x, double* y) {
        for (int i = 0; i < nq; i++) { y[i] = x[i]; y[nq + i] = dt * x[i] + x[nq + i]; }
    };
    auto BTx = [&](const double* x, double* y) { for (int i = 0; i < nq; i++) y[i] = c * x[i] + dt * x[nq + i]; };

    /* ---- initial point: z = 0 (with dx_0 fixed), slacks from the current margins ---------------------------- */
    int m_comp = 0;
    double R0 = 0.0;
    for (int k = 0; k <= N; k++) {
        Stage& s = S[k];
        int nz = s.nu + nx;
        for (int i = 0; i < nz; i++) s.z[i] = 0.0;
        if (k == 0) for (int i = 0; i < nx; i++) s.z[s.nu + i] = dx0[i];
        for (int r = 0; r < s.nr; r++) {
            double cz = 0.0, cn = 0.0;
            for (int i = 0; i < nz; i++) { cz += s.Cm[r][i] * s.z[i]; cn = std::max(cn, std::fabs(s.Cm[r][i])); }
            /* Slack floor of the starting point.  A hard row starts at least thr_hard |c|_inf inside its bound: a distance to
             * the boundary measured in the variables, not in the row's own units.  With one absolute floor (0.1, round 1) a
             * steep row -- the safe-set row at zero velocity has |dg/dv| ~ 1e2..1e3 through the eps-regularised direction,
             * safe_set.py:83-87 -- whose margin of "50" is really 1e-4 away from active started as if far inactive, and its
             * multiplier was then walked down one row per iteration (7-DoF, N = 40, row on every node: 36 iterations, now
             * 14); a flat row (squared capsule distances, |c| ~ 0.1) sitting on its bound started 0.1 = three times its
             * own magnitude outside it (66 iterations on the degenerate 7-DoF start, now 33).  The soft row keeps the
             * absolute floor: its slack variable is in the row's units. */
            const double thr = s.soft[r] < 0.0 ? thr_hard * (cn > 0.0 ? cn : 1.0) : thr_soft;
            s.sl[r] = 0.0;
            if (s.has_lo[r]) {
                double sl0 = s.soft[r] >= 0.0 ? thr : 0.0;
                s.sl[r] = sl0;
                double slack = cz + sl0 - s.lo[r];
                s.tl[r] = std::max(slack, thr);
                s.ll[r] = o.mu0 / s.tl[r];
                if (s.soft[r] >= 0.0) s.ll[r] = std::min(s.ll[r], 0.5 * s.soft[r]);
                R0 = std::max(R0, std::fabs(slack - s.tl[r]));
                m_comp += s.soft[r] >= 0.0 ? 2 : 1;
            }
            if (s.has_hi[r]) {
                double slack = s.hi[r] - cz;
                s.tu[r] = std::max(slack, thr);
                s.lu[r] = o.mu0 / s.tu[r];
                R0 = std::max(R0, std::fabs(slack - s.tu[r]));
                m_comp += 1;
            }
        }
    }
    /* initial linear residuals: stationarity with pi = 0, dynamics */
    for (int k = 0; k <= N; k++) {
        Stage& s = S[k];
        int nz = s.nu + nx;
        for (int i = 0; i < nz; i++) {
            if (k == 0 && i >= s.nu) continue;  // dx_0 is data
            double r = s.g[i];
            for (int j = 0; j < nz; j++) r += s.H[i][j] * s.z[j];
            for (int q = 0; q < s.nr; q++) {
                if (s.has_lo[q]) r -= s.Cm[q][i] * s.ll[q];
                if (s.has_hi[q]) r += s.Cm[q][i] * s.lu[q];
            }
            R0 = std::max(R0, std::fabs(r));
        }
        if (k < N) {
            double ax[MAXX];
            Ax(&s.z[s.nu], ax);
            for (int i = 0; i < nx; i++) R0 = std::max(R0, std::fabs(ax[i] + s.b[i]));  // next dx starts at 0
        }
    }
    if (m_comp == 0) m_comp = 1;

    double rho = 1.0;  // product of (1 - alpha): every linear residual equals rho * (its initial value)
    int it = 0, status = 2;
    static thread_local std::vector<std::vector<double>> gh;
    if ((int)gh.size() < N + 1) gh.assign(N + 1, std::vector<double>(MAXZ));
    double P[MAXX][MAXX], pv[MAXX];

    auto mu_now = [&]() {
        double acc = 0.0;
        for (int k = 0; k <= N; k++) {
            Stage& s = S[k];
            for (int r = 0; r < s.nr; r++) {
                if (s.has_lo[r]) {
                    acc += s.ll[r] * s.tl[r];
                    if (s.soft[r] >= 0.0) acc += (s.soft[r] - s.ll[r]) * s.sl[r];
                }
                if (s.has_hi[r]) acc += s.lu[r] * s.tu[r];
            }
        }
        return acc / m_comp;
    };

    /* Riccati factorisation of H + C^T D C; fills L, W, Pb of every stage.  false on a non-positive pivot. */
    auto factorize = [&]() -> bool {
        for (int k = N; k >= 0; k--) {
            Stage& s = S[k];
            int nuk = s.nu, nz = nuk + nx;
            double Hh[MAXZ][MAXZ];
            for (int i = 0; i < nz; i++) for (int j = 0; j < nz; j++) Hh[i][j] = s.H[i][j];
            for (int r = 0; r < s.nr; r++) {
                double dsum = 0.0;
                if (s.has_lo[r]) {
                    if (s.soft[r] >= 0.0) dsum += 1.0 / (s.tl[r] / s.ll[r] + s.sl[r] / (s.soft[r] - s.ll[r]));
                    else dsum += s.ll[r] / s.tl[r];
                }
                if (s.has_hi[r]) dsum += s.lu[r] / s.tu[r];
                if (dsum == 0.0) continue;
                for (int i = 0; i < nz; i++) {
                    double ci = s.Cm[r][i] * dsum;
                    if (ci == 0.0) continue;
                    for (int j = 0; j < nz; j++) Hh[i][j] += ci * s.Cm[r][j];
                }
            }
            if (k == N) {
                for (int i = 0; i < nx; i++) for (int j = 0; j < nx; j++) P[i][j] = Hh[i][j];
                continue;
            }
            /* P b, B^T P B, B^T P A, A^T P A with the structured A, B */
            double PA[MAXX][MAXX], PB[MAXX][MAXU];
            for (int i = 0; i < nx; i++) {
                for (int j = 0; j < nq; j++) {
                    PA[i][j] = P[i][j];
                    PA[i][nq + j] = dt * P[i][j] + P[i][nq + j];
                    PB[i][j] = c * P[i][j] + dt * P[i][nq + j];
                }
            }
            for (int i = 0; i < nx; i++) {
                double acc = 0.0;
                for (int j = 0; j < nx; j++) acc += P[i][j] * s.b[j];
                s.Pb[i] = acc;
            }
            double Lam[MAXU][MAXU], G[MAXU][MAXX], APA[MAXX][MAXX];
            for (int i = 0; i < nq; i++) {
                for (int j = 0; j < nq; j++) Lam[i][j] = Hh[i][j] + c * PB[i][j] + dt * PB[nq + i][j];
                for (int j = 0; j < nx; j++) G[i][j] = Hh[i][nuk + j] + c * PA[i][j] + dt * PA[nq + i][j];
            }
            for (int i = 0; i < nq; i++) for (int j = 0; j < nx; j++) {
                APA[i][j] = PA[i][j];
                APA[nq + i][j] = dt * PA[i][j] + PA[nq + i][j];
            }
            /* Cholesky Lam = L L^T */
            for (int j = 0; j < nq; j++) {
                double d = Lam[j][j];
                for (int t = 0; t < j; t++) d -= s.L[j][t] * s.L[j][t];
                if (!(d > 0.0)) return false;
                double lj = std::sqrt(d);
                s.L[j][j] = lj;
                for (int i = j + 1; i < nq; i++) {
                    double v = Lam[i][j];
                    for (int t = 0; t < j; t++) v -= s.L[i][t] * s.L[j][t];
                    s.L[i][j] = v / lj;
                }
            }
            /* W = L^-1 G */
            for (int col = 0; col < nx; col++) {
                for (int i = 0; i < nq; i++) {
                    double v = G[i][col];
                    for (int t = 0; t < i; t++) v -= s.L[i][t] * s.W[t][col];
                    s.W[i][col] = v / s.L[i][i];
                }
            }
            if (k > 0) {
                double Pn[MAXX][MAXX];
                for (int i = 0; i < nx; i++) for (int j = 0; j < nx; j++) {
                    double v = Hh[nuk + i][nuk + j] + APA[i][j];
                    for (int t = 0; t < nq; t++) v -= s.W[t][i] * s.W[t][j];
                    Pn[i][j] = v;
                }
                for (int i = 0; i < nx; i++) for (int j = 0; j < nx; j++) P[i][j] = 0.5 * (Pn[i][j] + Pn[j][i]);
            }
        }
        return true;
    };

    /* The vector recursion needs P_{k+1} applied to nothing but b (kept as Pb) -- p_k only involves W, w. */
    double corr_w = 1.0;
    auto solve = [&](double sigma_mu, bool corrector) {
        /* gradient  g + C^T e */
        for (int k = 0; k <= N; k++) {
            Stage& s = S[k];
            int nz = s.nu + nx;
            for (int i = 0; i < nz; i++) gh[k][i] = s.g[i];
            for (int r = 0; r < s.nr; r++) {
                double e = 0.0;
                if (s.has_lo[r]) {
                    double ct = corrector ? corr_w * s.dtl[r] * s.dll[r] : 0.0;
                    if (s.soft[r] >= 0.0) {
                        double nu_ = s.soft[r] - s.ll[r];
                        double cs = corrector ? -corr_w * s.dsl[r] * s.dll[r] : 0.0;  // d_nu = -d_lambda
                        double deff = 1.0 / (s.tl[r] / s.ll[r] + s.sl[r] / nu_);
                        e += -s.ll[r] + deff * (-s.lo[r] + (sigma_mu - cs) / nu_ - (sigma_mu - ct) / s.ll[r]);
                    } else {
                        e += -s.ll[r] - (s.ll[r] / s.tl[r]) * s.lo[r] - (sigma_mu - ct) / s.tl[r];
                    }
                }
                if (s.has_hi[r]) {
                    double ct = corrector ? corr_w * s.dtu[r] * s.dlu[r] : 0.0;
                    e += s.lu[r] - (s.lu[r] / s.tu[r]) * s.hi[r] + (sigma_mu - ct) / s.tu[r];
                }
                if (e == 0.0) continue;
                for (int i = 0; i < nz; i++) gh[k][i] += s.Cm[r][i] * e;
            }
        }
        /* backward vector sweep */
        for (int i = 0; i < nx; i++) pv[i] = gh[N][i];
        for (int k = N - 1; k >= 0; k--) {
            Stage& s = S[k];
            double pt[MAXX], rho_u[MAXU], at[MAXX];
            for (int i = 0; i < nx; i++) pt[i] = pv[i] + s.Pb[i];
            BTx(pt, rho_u);
            for (int i = 0; i < nq; i++) rho_u[i] += gh[k][i];
            for (int i = 0; i < nq; i++) {
                double v = rho_u[i];
                for (int t = 0; t < i; t++) v -= s.L[i][t] * s.w[t];
                s.w[i] = v / s.L[i][i];
            }
            if (k > 0) {
                ATx(pt, at);
                for (int i = 0; i < nx; i++) {
                    double v = gh[k][s.nu + i] + at[i];
                    for (int t = 0; t < nq; t++) v -= s.W[t][i] * s.w[t];
                    pv[i] = v;
                }
            }
        }
        /* forward sweep */
        double xk[MAXX];
        for (int i = 0; i < nx; i++) xk[i] = dx0[i];
        for (int k = 0; k < N; k++) {
            Stage& s = S[k];
            double rhs[MAXU], uk[MAXU];
            for (int i = 0; i < nq; i++) {
                double v = s.w[i];
                for (int j = 0; j < nx; j++) v += s.W[i][j] * xk[j];
                rhs[i] = -v;
            }
            for (int i = nq - 1; i >= 0; i--) {
                double v = rhs[i];
                for (int t = i + 1; t < nq; t++) v -= s.L[t][i] * uk[t];
                uk[i] = v / s.L[i][i];
            }
            for (int i = 0; i < nq; i++) s.zn[i] = uk[i];
            for (int i = 0; i < nx; i++) s.zn[s.nu + i] = xk[i];
            double xn[MAXX];
            Ax(xk, xn);
            for (int i = 0; i < nq; i++) { xn[i] += c * uk[i] + s.b[i]; xn[nq + i] += dt * uk[i] + s.b[nq + i]; }
            for (int i = 0; i < nx; i++) xk[i] = xn[i];
        }
        for (int i = 0; i < nx; i++) S[N].zn[i] = xk[i];
        /* directions of slacks and multipliers */
        for (int k = 0; k <= N; k++) {
            Stage& s = S[k];
            int nz = s.nu + nx;
            for (int r = 0; r < s.nr; r++) {
                if (!s.has_lo[r] && !s.has_hi[r]) continue;
                double czn = 0.0;
                for (int i = 0; i < nz; i++) czn += s.Cm[r][i] * s.zn[i];
                if (s.has_lo[r]) {
                    double ct = corrector ? corr_w * s.dtl[r] * s.dll[r] : 0.0;
                    if (s.soft[r] >= 0.0) {
                        double nu_ = s.soft[r] - s.ll[r];
                        double cs = corrector ? -corr_w * s.dsl[r] * s.dll[r] : 0.0;
                        double deff = 1.0 / (s.tl[r] / s.ll[r] + s.sl[r] / nu_);
                        double dl = -deff * (czn - s.lo[r] + (sigma_mu - cs) / nu_ - (sigma_mu - ct) / s.ll[r]);
                        double dt_ = (sigma_mu - ct - s.tl[r] * dl) / s.ll[r] - s.tl[r];
                        double ds = (sigma_mu - cs + s.sl[r] * dl) / nu_ - s.sl[r];
                        s.dll[r] = dl; s.dtl[r] = dt_; s.dsl[r] = ds;
                    } else {
                        double dt_ = czn - s.lo[r] - s.tl[r];
                        s.dll[r] = (sigma_mu - ct - s.ll[r] * dt_) / s.tl[r] - s.ll[r];
                        s.dtl[r] = dt_;
                    }
                }
                if (s.has_hi[r]) {
                    double ct = corrector ? corr_w * s.dtu[r] * s.dlu[r] : 0.0;
                    double dt_ = s.hi[r] - czn - s.tu[r];
                    s.dlu[r] = (sigma_mu - ct - s.lu[r] * dt_) / s.tu[r] - s.lu[r];
                    s.dtu[r] = dt_;
                }
            }
        }
    };

    int blk_k = 0, blk_r = 0;   /* who blocks the step (SMPC_ORACLE_TRACE only) */
    auto max_step = [&]() {
        double a = 1e300;
        int cur_k = 0, cur_r = 0;
        auto lim = [&](double y, double dy) { if (dy < 0.0 && -y / dy < a) { a = -y / dy; blk_k = cur_k; blk_r = cur_r; } };
        for (int k = 0; k <= N; k++) {
            Stage& s = S[k];
            cur_k = k;
            for (int r = 0; r < s.nr; r++) {
                cur_r = r;
                if (s.has_lo[r]) {
                    lim(s.tl[r], s.dtl[r]);
                    lim(s.ll[r], s.dll[r]);
                    if (s.soft[r] >= 0.0) { lim(s.sl[r], s.dsl[r]); lim(s.soft[r] - s.ll[r], -s.dll[r]); }
                }
                if (s.has_hi[r]) { lim(s.tu[r], s.dtu[r]); lim(s.lu[r], s.dlu[r]); }
            }
        }
        return a;
    };

    double mu = mu_now();
    int stall = 0, stall_total = 0;
    for (it = 0; it < o.max_iter; it++) {
        if (mu <= o.tol && rho * R0 <= o.tol_res) { status = 0; break; }
        if (!factorize()) { status = 4; break; }
        /* predictor */
        solve(0.0, false);
        double a_aff = std::min(1.0, max_step());
        double acc = 0.0;
        for (int k = 0; k <= N; k++) {
            Stage& s = S[k];
            for (int r = 0; r < s.nr; r++) {
                if (s.has_lo[r]) {
                    acc += (s.ll[r] + a_aff * s.dll[r]) * (s.tl[r] + a_aff * s.dtl[r]);
                    if (s.soft[r] >= 0.0)
                        acc += (s.soft[r] - s.ll[r] - a_aff * s.dll[r]) * (s.sl[r] + a_aff * s.dsl[r]);
                }
                if (s.has_hi[r]) acc += (s.lu[r] + a_aff * s.dlu[r]) * (s.tu[r] + a_aff * s.dtu[r]);
            }
        }
        double mu_aff = acc / m_comp;
        double sigma = mu_aff / mu;
        sigma = sigma * sigma * sigma;
        /* safeguard against Mehrotra cycling: the second-order term models a FULL affine step; when the affine step is
         * blocked early (a_aff < 0.3) it is scaled by a_aff^2, which turns the iteration into a centring step */
        corr_w = a_aff >= 0.3 ? 1.0 : a_aff * a_aff;
        /* centring cap: Mehrotra's (mu_aff/mu)^3 asks for an almost pure centring step (sigma ~ 0.8-1) whenever the affine
         * step is blocked early; those steps stall on nearly-active rows.  Capping sigma at 0.3 keeps the mean iteration
         * count and halves the tail (max 21 -> 11 on the closed-loop 'st' workload) */
        sigma = std::min(sigma, 0.3);
        /* corrector */
        solve(sigma * mu, true);
        /* fraction to the boundary: 0.995 far from the solution, approaching 1 with the complementarity (capped at
         * 0.9999 so that zero-width boxes keep usable slacks) -- but only when the full Newton step is feasible; a step
         * that is blocked earlier keeps the classical 0.5% margin, otherwise the blocking pair lands far off-centre and
         * the next iterations cycle.  Saves the last 0.005-per-iteration crawl. */
        const double a_max = max_step();
        const double tau_k = a_max >= 0.99 ? std::min(0.9999, std::max(tau_ftb, 1.0 - mu)) : tau_ftb;
        double alpha = std::min(1.0, tau_k * a_max);
        if (!(alpha == alpha)) { status = 4; break; }
        if (alpha < alpha_min) { status = 3; break; }
        for (int k = 0; k <= N; k++) {
            Stage& s = S[k];
            int nz = s.nu + nx;
            for (int i = 0; i < nz; i++) s.z[i] += alpha * (s.zn[i] - s.z[i]);
            for (int r = 0; r < s.nr; r++) {
                if (s.has_lo[r]) {
                    s.tl[r] += alpha * s.dtl[r];
                    s.ll[r] += alpha * s.dll[r];
                    if (s.soft[r] >= 0.0) s.sl[r] += alpha * s.dsl[r];
                }
                if (s.has_hi[r]) { s.tu[r] += alpha * s.dtu[r]; s.lu[r] += alpha * s.dlu[r]; }
            }
        }
        rho *= (1.0 - alpha);
        const double mu_before = mu;
        mu = mu_now();
        /* a stalled iteration: a short step that did not halve the complementarity either (a degenerate but feasible QP
         * crawls with short steps while mu still falls); an iterate that meets the exit test is never reported as stalled */
        const bool stalled = alpha < 0.5 && !(mu < 0.5 * mu_before);
        stall = stalled ? stall + 1 : 0;
        stall_total += stalled ? 1 : 0;
        /* ... stall_iters in a row, or 7/6 of that in total (round 5: an infeasible QP whose complementarity falls in bursts resets
         * the run now and then and took up to 52 iterations where most give up after 27; with the cap on the total: 31) */
        if (o.stall_iters > 0 && (stall >= o.stall_iters || stall_total >= o.stall_iters + (o.stall_iters + 5) / 6) &&
            !(mu <= o.tol && rho * R0 <= o.tol_res)) { status = 5; it++; break; }
        if (std::getenv("SMPC_ORACLE_TRACE"))
            std::fprintf(stderr, "it %3d a_aff %.3e sigma %.3e alpha %.3e mu %.3e rho*R0 %.3e  blocked by stage %d row %d (tl %.2e ll %.2e tu %.2e lu %.2e)\n",
                         it, a_aff, sigma, alpha, mu, rho * R0, blk_k, blk_r, S[blk_k].tl[blk_r], S[blk_k].ll[blk_r], S[blk_k].tu[blk_r], S[blk_k].lu[blk_r]);
        if (!(mu == mu)) { status = 4; break; }
    }
    if (it == o.max_iter && status == 2 && mu <= o.tol && rho * R0 <= o.tol_res) status = 0;
    *iters = it;
    if (res_out) { res_out[0] = mu; res_out[1] = rho * R0; }
    return status;
}

/* build the stage QP of one instance from node evaluations */
bool build_qp(const smpc_problem_desc& D, int N, const double* lo_st, const double* hi_st, const double* zl_st,
              const std::vector<smpc_node_eval>& ev, const double* x0, const double* xg, const double* ug,
              const double* p, std::vector<Stage>& S, double* dx0) {
    int nq = D.nq, nx = 2 * nq, nu = nq;
    double dt = D.dt, c = 0.5 * dt * dt;
    for (int i = 0; i < nx; i++) dx0[i] = x0[i] - xg[i];
    /* collision rows at node 0 (kept by the reference when --noise == 0, controller.py:77-79): x_0 is pinned, so the
     * linearised rows are constants -- outside their bounds the QP is infeasible (reported as QP failure) */
    bool infeasible0 = false;
    if (D.rows_at_node0) {
        for (int i = 0; i < D.n_rows; i++) {
            double v = ev[0].row_val[i];
            for (int j = 0; j < nq; j++) v += ev[0].row_grad[i * nq + j] * dx0[j];
            if (std::fabs(D.rows[i].lb) < SMPC_INF && v < D.rows[i].lb - D.qp_tol) infeasible0 = true;
            if (std::fabs(D.rows[i].ub) < SMPC_INF && v > D.rows[i].ub + D.qp_tol) infeasible0 = true;
        }
    }
    for (int k = 0; k <= N; k++) {
        Stage& s = S[k];
        const smpc_node_eval& e = ev[k];
        const double* xk = xg + (size_t)k * nx;
        const double* pk = p + (size_t)k * SMPC_NP;
        std::memset((void*)&s, 0, sizeof(Stage));   /* (workspaces are reused across instances) */
        s.nu = k < N ? nu : 0;
        int o = s.nu, nz = o + nx;
        double cs = k < N ? D.cost_scale_stage : D.cost_scale_term;
        double lm = k < N ? D.lm_stage : D.lm_term;
        if (D.cost_kind == SMPC_COST_REACH) {
            if (k < N) {
                const double* uk = ug + (size_t)k * nu;
                for (int i = 0; i < nu; i++) { s.H[i][i] = cs * 2.0 * D.R; s.g[i] = cs * 2.0 * D.R * uk[i]; }
            }
            for (int i = 0; i < nq; i++) {
                s.g[o + i] = cs * e.cost_grad_q[i];
                for (int j = 0; j < nq; j++) s.H[o + i][o + j] = cs * e.cost_hess_qq[i * nq + j];
            }
        }
        for (int i = 0; i < nz; i++) s.H[i][i] += lm;
        if (k < N) {
            /* b = f(xg_k, ug_k) - xg_{k+1} */
            const double* uk = ug + (size_t)k * nu;
            const double* xn = xg + (size_t)(k + 1) * nx;
            for (int i = 0; i < nq; i++) {
                s.b[i] = xk[i] + dt * xk[nq + i] + c * uk[i] - xn[i];
                s.b[nq + i] = xk[nq + i] + dt * uk[i] - xn[nq + i];
            }
        }
        /* rows: box | torque | collision | nn */
        int r = 0;
        const double* lo_k = lo_st + (size_t)k * nx;
        const double* hi_k = hi_st + (size_t)k * nx;
        for (int i = 0; i < nx; i++, r++) {
            s.Cm[r][o + i] = 1.0;
            s.lo[r] = lo_k[i] - xk[i];
            s.hi[r] = hi_k[i] - xk[i];
            s.has_lo[r] = k >= 1 && std::fabs(lo_k[i]) < SMPC_INF;
            s.has_hi[r] = k >= 1 && std::fabs(hi_k[i]) < SMPC_INF;
            s.soft[r] = -1.0;
        }
        for (int i = 0; i < nq; i++, r++) {
            if (k < N) {
                for (int j = 0; j < nq; j++) {
                    s.Cm[r][j] = e.M[i * nq + j];
                    s.Cm[r][o + j] = e.dtau_dq[i * nq + j];
                    s.Cm[r][o + nq + j] = e.dtau_dv[i * nq + j];
                }
            }
            double tm = D.joints[i].tau_max;
            s.lo[r] = -tm - e.tau[i];
            s.hi[r] = tm - e.tau[i];
            s.has_lo[r] = s.has_hi[r] = k < N && tm < SMPC_INF;
            s.soft[r] = -1.0;
        }
        for (int i = 0; i < D.n_rows; i++, r++) {
            for (int j = 0; j < nq; j++) s.Cm[r][o + j] = e.row_grad[i * nq + j];
            s.lo[r] = D.rows[i].lb - e.row_val[i];
            s.hi[r] = D.rows[i].ub - e.row_val[i];
            s.has_lo[r] = k >= 1 && std::fabs(D.rows[i].lb) < SMPC_INF;
            s.has_hi[r] = k >= 1 && std::fabs(D.rows[i].ub) < SMPC_INF;
            s.soft[r] = -1.0;
        }
        {
            bool on = nn_active(D, k, N, pk);
            double w = k == N ? D.nn_soft_e : D.nn_soft_run;
            if (zl_st && w >= 0.0) w = zl_st[k];   /* cost_set(k,'zl',.) (controller.py:455-468) on a soft row */
            /* a zero slack weight leaves the row without effect on the optimum (the slack absorbs it at no cost): the reference
               only ever writes zl = 0 together with p[4] = -1 (row off, :455-460); treated as "row absent" in either case */
            if (zl_st && (k == N ? D.nn_soft_e : D.nn_soft_run) >= 0.0 && w == 0.0) on = false;
            for (int j = 0; j < nx; j++) s.Cm[r][o + j] = on ? e.nn_grad[j] : 0.0;
            s.lo[r] = 0.0 - e.nn_val;
            s.hi[r] = 1e6;
            s.has_lo[r] = on;
            s.has_hi[r] = false;
            s.soft[r] = on ? w : -1.0;
            r++;
        }
        s.nr = r;
    }
    return infeasible0;
}

struct Oracle {
    smpc_problem_desc D;
    Mlp net;
    int N;
    std::vector<double> lo_st, hi_st;  // [N+1][nx]
    std::vector<double> lo_b, hi_b;    // [B][N+1][nx] per-instance bounds (RealReceding), empty = none
    std::vector<double> zl;            // [N+1] run-time slack weights of soft rows, empty = descriptor's
    int inst_B = 0;
    void reset_bounds() {
        int nx = 2 * D.nq;
        lo_st.assign((size_t)(N + 1) * nx, 0.0);
        hi_st.assign((size_t)(N + 1) * nx, 0.0);
        for (int k = 0; k <= N; k++) for (int i = 0; i < nx; i++) {
            lo_st[(size_t)k * nx + i] = k == N ? D.x_lo_e[i] : D.x_lo[i];
            hi_st[(size_t)k * nx + i] = k == N ? D.x_hi_e[i] : D.x_hi[i];
        }
    }
};

}  // namespace

/* ==================================================================================================================
 * C entry points (loaded with ctypes by tests/ and bench.py only)
 * ================================================================================================================ */
extern "C" {

void* orc_create(const smpc_problem_desc* d) {
    if (!d || d->abi_version != SMPC_ABI_VERSION) return nullptr;
    Oracle* o = new Oracle();
    o->D = *d;
    o->N = d->N;
    o->reset_bounds();
    return o;
}
void orc_destroy(void* h) { delete (Oracle*)h; }

int orc_set_mlp(void* h, int nlayers, const int32_t* dims, const float* const* W, const float* const* b) {
    Oracle* o = (Oracle*)h;
    if (nlayers < 1 || nlayers > SMPC_MAX_LAYERS) return SMPC_EINVAL;
    o->net.nlayers = nlayers;
    for (int l = 0; l <= nlayers; l++) o->net.dims[l] = dims[l];
    for (int l = 0; l < nlayers; l++) {
        o->net.W[l].assign(W[l], W[l] + (size_t)dims[l] * dims[l + 1]);
        o->net.b[l].assign(b[l], b[l] + dims[l + 1]);
    }
    return 0;
}
int orc_set_mlp_activation(void* h, int act) {
    Oracle* o = (Oracle*)h;
    if (act < SMPC_ACT_GELU_TANH || act > SMPC_ACT_SILU) return SMPC_EINVAL;
    o->net.act = act;
    return 0;
}
int orc_set_horizon(void* h, int N) {
    Oracle* o = (Oracle*)h;
    if (N < 1 || N > SMPC_MAX_N) return SMPC_EINVAL;
    o->N = N;
    o->reset_bounds();
    return 0;
}
int orc_set_instance_bounds(void* h, int B, const double* lo, const double* hi) {
    Oracle* o = (Oracle*)h;
    if (!lo || !hi) { o->inst_B = 0; return 0; }
    size_t n = (size_t)B * (o->N + 1) * 2 * o->D.nq;
    o->lo_b.assign(lo, lo + n);
    o->hi_b.assign(hi, hi + n);
    o->inst_B = B;
    return 0;
}
int orc_set_stage_bounds(void* h, const double* lo, const double* hi) {
    Oracle* o = (Oracle*)h;
    if (!lo || !hi) { o->reset_bounds(); return 0; }
    size_t n = (size_t)(o->N + 1) * 2 * o->D.nq;
    o->lo_st.assign(lo, lo + n);
    o->hi_st.assign(hi, hi + n);
    return 0;
}

/* number of OpenMP threads used by the batch entry points (bench.py's cpu_baseline: all usable cores, then 1) */
int orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

int orc_set_slack_weights(void* h, const double* zl) {
    Oracle* o = (Oracle*)h;
    if (!zl) o->zl.clear();
    else o->zl.assign(zl, zl + o->N + 1);
    return 0;
}

/* ---- component-level entry points ------------------------------------------------------------------------------ */
void orc_rnea(void* h, const double* q, const double* qd, const double* qdd, double* tau) {
    Oracle* o = (Oracle*)h;
    rnea<double>(o->D.joints, o->D.nq, o->D.gravity, q, qd, qdd, tau);
}
void orc_rnea_joints(const smpc_joint* J, int nq, const double* grav, const double* q, const double* qd,
                     const double* qdd, double* tau) {
    rnea<double>(J, nq, grav, q, qd, qdd, tau);
}
void orc_fk(void* h, const double* q, double* R_out, double* p_out) {
    Oracle* o = (Oracle*)h;
    M3<double> R[MAXQ];
    V3<double> p[MAXQ];
    fk_links<double>(o->D.joints, o->D.nq, q, R, p);
    for (int i = 0; i < o->D.nq; i++) {
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) R_out[9 * i + 3 * a + b] = R[i].m[a][b];
        for (int a = 0; a < 3; a++) p_out[3 * i + a] = p[i][a];
    }
}
void orc_points(void* h, const double* q, double* out) {
    Oracle* o = (Oracle*)h;
    M3<double> R[MAXQ];
    V3<double> p[MAXQ];
    fk_links<double>(o->D.joints, o->D.nq, q, R, p);
    for (int i = 0; i < o->D.n_points; i++) {
        V3<double> w = point_world<double>(o->D.points[i], R, p);
        for (int a = 0; a < 3; a++) out[3 * i + a] = w[a];
    }
}
double orc_segment_dist2(const double* A, const double* B, const double* Cc, const double* Dd) {
    return segment_dist2<double>(cv3<double>(A), cv3<double>(B), cv3<double>(Cc), cv3<double>(Dd));
}
void orc_mlp(void* h, const float* s, float* y, float* grad) { mlp_eval(((Oracle*)h)->net, s, y, grad); }
void orc_nn_row(void* h, const double* x, double alpha, double* g, double* dgdx) {
    Oracle* o = (Oracle*)h;
    nn_row(o->D, o->net, x, alpha, g, dgdx);
}

int orc_eval_nodes(void* h, int B, const double* xg, const double* ug, const double* p, smpc_node_eval* out) {
    Oracle* o = (Oracle*)h;
    int N = o->N, nx = 2 * o->D.nq, nu = o->D.nq;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; b++) {
        for (int k = 0; k <= N; k++) {
            const double* x = xg + ((size_t)b * (N + 1) + k) * nx;
            const double* u = k < N ? ug + ((size_t)b * N + k) * nu : nullptr;
            const double* pk = p + ((size_t)b * (N + 1) + k) * SMPC_NP;
            eval_node(o->D, &o->net, k, N, x, u, pk, &out[(size_t)b * (N + 1) + k]);
        }
    }
    return 0;
}

/* ---- the hot path: one SQP-RTI solve per instance (controller.py:136-167) -------------------------------------- */
int orc_solve_batch(void* h, int B, const double* x0, const double* xg, const double* ug, const double* p,
                    double* x_out, double* u_out, int32_t* status, int32_t* qp_iter, double* qp_res) {
    Oracle* o = (Oracle*)h;
    const smpc_problem_desc& D = o->D;
    int N = o->N, nq = D.nq, nx = 2 * nq, nu = nq;
    QpOpts qo{D.qp_max_iter, D.qp_tol, D.qp_tol_res > 0.0 ? D.qp_tol_res : D.qp_tol, D.qp_mu0, D.qp_stall_iters};
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; b++) {
        const double* xb = xg + (size_t)b * (N + 1) * nx;
        const double* ub = ug + (size_t)b * N * nu;
        const double* pb = p + (size_t)b * (N + 1) * SMPC_NP;
        /* per-thread workspaces, reused across instances: 31 stages are ~0.4 MB, which glibc serves by mmap/munmap -- with
         * one allocation per instance the threads of a many-core host serialise on the kernel's address-space lock */
        static thread_local std::vector<smpc_node_eval> ev;
        static thread_local std::vector<Stage> S;
        ev.resize(N + 1);
        S.resize(N + 1);
        for (int k = 0; k <= N; k++)
            eval_node(D, &o->net, k, N, xb + (size_t)k * nx, k < N ? ub + (size_t)k * nu : nullptr,
                      pb + (size_t)k * SMPC_NP, &ev[k]);
        double dx0[MAXX];
        const size_t boff = (size_t)b * (N + 1) * nx;
        const double* blo = o->inst_B == B ? o->lo_b.data() + boff : o->lo_st.data();
        const double* bhi = o->inst_B == B ? o->hi_b.data() + boff : o->hi_st.data();
        const bool infeasible0 = build_qp(D, N, blo, bhi, o->zl.empty() ? nullptr : o->zl.data(), ev, x0 + (size_t)b * nx, xb, ub, pb, S, dx0);
        int it = 0;
        double res[2] = {0, 0};
        int qs = qp_ipm(S, N, nx, nu, D.dt, dx0, qo, &it, res);
        /* full step (FIXED_STEP, parser.py:139).  acados' RTI tolerates a QP that stopped at its iteration cap and
         * reports success [EXT-UNVERIFIED]; breakdown and min-step are ACADOS_QP_FAILURE. */
        int st = (qs == 0 || qs == 2) ? SMPC_STATUS_SUCCESS : SMPC_STATUS_QP_FAILURE;
        if (infeasible0) st = SMPC_STATUS_QP_FAILURE;
        double* xo = x_out + (size_t)b * (N + 1) * nx;
        double* uo = u_out + (size_t)b * N * nu;
        bool nan = false;
        for (int k = 0; k <= N; k++) {
            for (int i = 0; i < nx; i++) {
                double v = xb[(size_t)k * nx + i] + S[k].z[S[k].nu + i];
                xo[(size_t)k * nx + i] = v;
                nan |= !(v == v);
            }
            if (k < N) for (int i = 0; i < nu; i++) {
                double v = ub[(size_t)k * nu + i] + S[k].z[i];
                uo[(size_t)k * nu + i] = v;
                nan |= !(v == v);
            }
        }
        if (nan && st == SMPC_STATUS_SUCCESS) st = SMPC_STATUS_NAN;
        status[b] = st;
        if (qp_iter) qp_iter[b] = it;
        if (qp_res) { qp_res[2 * b] = res[0]; qp_res[2 * b + 1] = res[1]; }
    }
    return 0;
}

/* expose the assembled stage QP of one instance (for the scipy cross-check in tests/test_oracle_qp.py):
 * per stage k: H [MAXZ*MAXZ], g [MAXZ], b [MAXX], C [MAXR*MAXZ], lo, hi [MAXR], flags has_lo, has_hi, soft [MAXR] */
int orc_qp_dims(int* maxz, int* maxx, int* maxr) { *maxz = MAXZ; *maxx = MAXX; *maxr = MAXR; return 0; }
int orc_build_qp(void* h, const double* x0, const double* xg, const double* ug, const double* p, double* Hh,
                 double* g, double* bb, double* Cc, double* lo, double* hi, int32_t* has_lo, int32_t* has_hi,
                 double* soft, int32_t* nr, double* dx0) {
    Oracle* o = (Oracle*)h;
    const smpc_problem_desc& D = o->D;
    int N = o->N, nx = 2 * D.nq, nu = D.nq;
    std::vector<smpc_node_eval> ev(N + 1);
    for (int k = 0; k <= N; k++)
        eval_node(D, &o->net, k, N, xg + (size_t)k * nx, k < N ? ug + (size_t)k * nu : nullptr,
                  p + (size_t)k * SMPC_NP, &ev[k]);
    std::vector<Stage> S(N + 1);
    build_qp(D, N, o->lo_st.data(), o->hi_st.data(), o->zl.empty() ? nullptr : o->zl.data(), ev, x0, xg, ug, p, S, dx0);
    for (int k = 0; k <= N; k++) {
        std::memcpy(Hh + (size_t)k * MAXZ * MAXZ, S[k].H, sizeof(S[k].H));
        std::memcpy(g + (size_t)k * MAXZ, S[k].g, sizeof(S[k].g));
        std::memcpy(bb + (size_t)k * MAXX, S[k].b, sizeof(S[k].b));
        std::memcpy(Cc + (size_t)k * MAXR * MAXZ, S[k].Cm, sizeof(S[k].Cm));
        for (int r = 0; r < MAXR; r++) {
            bool in = r < S[k].nr;
            lo[(size_t)k * MAXR + r] = in ? S[k].lo[r] : 0.0;
            hi[(size_t)k * MAXR + r] = in ? S[k].hi[r] : 0.0;
            has_lo[(size_t)k * MAXR + r] = in && S[k].has_lo[r];
            has_hi[(size_t)k * MAXR + r] = in && S[k].has_hi[r];
            soft[(size_t)k * MAXR + r] = in ? S[k].soft[r] : -1.0;
        }
        nr[k] = S[k].nr;
    }
    return 0;
}

/* ---- callers around the solve ---------------------------------------------------------------------------------- */
int orc_guess_correction(void* h, int B, double* xg, const double* ug) {
    /* controller.py:226-231 */
    Oracle* o = (Oracle*)h;
    int N = o->N, nq = o->D.nq, nx = 2 * nq;
    double dt = o->D.dt, c = 0.5 * dt * dt;
    for (int b = 0; b < B; b++) {
        double* x = xg + (size_t)b * (N + 1) * nx;
        const double* u = ug + (size_t)b * N * nq;
        for (int k = 0; k < N; k++) for (int i = 0; i < nq; i++) {
            x[(k + 1) * nx + i] = x[k * nx + i] + dt * x[k * nx + nq + i] + c * u[k * nq + i];
            x[(k + 1) * nx + nq + i] = x[k * nx + nq + i] + dt * u[k * nq + i];
        }
    }
    return 0;
}
int orc_provide_control(void* h, int B, const int32_t* accept, const double* xt, const double* ut, double* xg,
                        double* ug, double* u_apply) {
    /* controller.py:169-184 */
    Oracle* o = (Oracle*)h;
    int N = o->N, nq = o->D.nq, nx = 2 * nq;
    for (int b = 0; b < B; b++) {
        double* x = xg + (size_t)b * (N + 1) * nx;
        double* u = ug + (size_t)b * N * nq;
        if (accept[b]) {
            std::memcpy(x, xt + (size_t)b * (N + 1) * nx, sizeof(double) * (N + 1) * nx);
            std::memcpy(u, ut + (size_t)b * N * nq, sizeof(double) * N * nq);
        }
        for (int i = 0; i < nq; i++) u_apply[(size_t)b * nq + i] = u[i];
        std::memmove(x, x + nx, sizeof(double) * N * nx);           // roll by -1 ...
        std::memcpy(x + (size_t)N * nx, x + (size_t)(N - 1) * nx, sizeof(double) * nx);  // ... last = second last
        if (N > 1) {
            std::memmove(u, u + nq, sizeof(double) * (N - 1) * nq);
            std::memcpy(u + (size_t)(N - 1) * nq, u + (size_t)(N - 2) * nq, sizeof(double) * nq);
        }
    }
    return 0;
}
int orc_check_trajectory(void* h, int B, int n_nodes, const double* x, const double* x_min, const double* x_max,
                         double tol_x, const double* row_lb, const double* row_ub, double alpha, double tol_safe,
                         int32_t* state_ok, int32_t* nn_ok) {
    /* env_model.py:170-173 (bounds and collisions), safe_set.py:61-68 */
    Oracle* o = (Oracle*)h;
    const smpc_problem_desc& D = o->D;
    int nq = D.nq, nx = 2 * nq;
    for (int b = 0; b < B; b++) {
        bool ok = true;
        for (int k = 0; k < n_nodes; k++) {
            const double* xk = x + ((size_t)b * n_nodes + k) * nx;
            for (int i = 0; i < nx; i++) ok &= (xk[i] >= x_min[i] - tol_x) && (xk[i] <= x_max[i] + tol_x);
            M3<double> R[MAXQ];
            V3<double> pw[MAXQ];
            fk_links<double>(D.joints, nq, xk, R, pw);
            for (int r = 0; r < D.n_rows; r++) {
                double v = row_value<double>(D, D.rows[r], R, pw);
                ok &= (row_lb[r] <= v) && (v <= row_ub[r]);
            }
            if (nn_ok) {
                double g = 0.0, dg[MAXX];
                if (o->net.nlayers > 0) nn_row(D, o->net, xk, alpha, &g, dg);
                nn_ok[(size_t)b * n_nodes + k] = (g >= -tol_safe) && (g <= 1e6 + tol_safe);
            }
        }
        state_ok[b] = ok;
    }
    return 0;
}
int orc_plant_step(void* h, int B, const double* x, const double* u, const smpc_joint* jn, const double* tau_noise,
                   double* x_next, double* u_eff) {
    /* env_model.py:192-206 */
    Oracle* o = (Oracle*)h;
    const smpc_problem_desc& D = o->D;
    int nq = D.nq, nx = 2 * nq;
    double dt = D.dt, c = 0.5 * dt * dt;
    for (int b = 0; b < B; b++) {
        const smpc_joint* J = jn ? jn + (size_t)b * nq : D.joints;
        const double* xb = x + (size_t)b * nx;
        const double* ub = u + (size_t)b * nq;
        double tau[MAXQ], hb[MAXQ], zero[MAXQ] = {0}, M[MAXQ][MAXQ];
        rnea<double>(J, nq, D.gravity, xb, xb + nq, ub, tau);
        rnea<double>(J, nq, D.gravity, xb, xb + nq, zero, hb);
        for (int j = 0; j < nq; j++) {
            double e[MAXQ] = {0}, col[MAXQ];
            e[j] = 1.0;
            rnea<double>(J, nq, D.gravity, xb, xb + nq, e, col);
            for (int i = 0; i < nq; i++) M[i][j] = col[i] - hb[i];
        }
        double rhs[MAXQ];
        for (int i = 0; i < nq; i++) {
            double t = tau[i] + (tau_noise ? tau_noise[(size_t)b * nq + i] : 0.0);
            double tm = J[i].tau_max;
            t = std::min(std::max(t, -tm), tm);
            rhs[i] = t - hb[i];
        }
        /* Gaussian elimination with partial pivoting (numpy.linalg.solve) */
        double A[MAXQ][MAXQ + 1];
        for (int i = 0; i < nq; i++) { for (int j = 0; j < nq; j++) A[i][j] = M[i][j]; A[i][nq] = rhs[i]; }
        for (int col = 0; col < nq; col++) {
            int piv = col;
            for (int i = col + 1; i < nq; i++) if (std::fabs(A[i][col]) > std::fabs(A[piv][col])) piv = i;
            if (piv != col) for (int j = 0; j <= nq; j++) std::swap(A[piv][j], A[col][j]);
            for (int i = col + 1; i < nq; i++) {
                double f = A[i][col] / A[col][col];
                for (int j = col; j <= nq; j++) A[i][j] -= f * A[col][j];
            }
        }
        double acc[MAXQ];
        for (int i = nq - 1; i >= 0; i--) {
            double v = A[i][nq];
            for (int j = i + 1; j < nq; j++) v -= A[i][j] * acc[j];
            acc[i] = v / A[i][i];
        }
        for (int i = 0; i < nq; i++) {
            x_next[(size_t)b * nx + i] = xb[i] + dt * xb[nq + i] + c * acc[i];
            x_next[(size_t)b * nx + nq + i] = xb[nq + i] + dt * acc[i];
            if (u_eff) u_eff[(size_t)b * nq + i] = acc[i];
        }
    }
    return 0;
}

}  // extern "C"
