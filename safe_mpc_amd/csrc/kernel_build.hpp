// kernel_build.hpp -- HOT LOOP A + the QP set-up as ONE lane-cooperative pass: linearisation of a shooting node (torque row and
// its three Jacobians, EE cost terms, collision rows; what acados evaluates through CasADi-generated C; reference
// env_model.py:80-83, 92-95, 131-163, 263-316, cost_definition.py:69-96) written STRAIGHT into the node's stage record of the QP
// workspace, together with the record's bounds, cost blocks and initial interior point (what k_qp_setup builds from the
// linearisation records of k_node_linearise).
//
// Why: in the three-stream closed loop the two thread-per-node kernels cost a stream's chain 0.65 ms of a 3.3 ms step for 0.2 ms of
// work (rounds 3-4, SMPC_DUP_KERNELS): k_node_linearise needs 512 registers per lane (+ 1.4 KB of scratch, 159 KB of straight-line
// code) and only starts on SIMDs that hold no QP wavefront; k_qp_setup's blocks need 23 KB of LDS next to the QP wavefronts' 148 KB;
// and the 2.6 KB linearisation record of every node makes a round trip through HBM between them.
//
// Mapping: SB_G = 8 lanes per node (a wavefront = 8 nodes), lane g < NQ owns joint g.  In program order:
//   K  kinematics lane g runs the forward kinematics up to ITS joint (sin / cos of every joint computed once, by its owner, and
//                 shared through LDS): R_g, p_g, S_g, v_g, a_g in world coordinates
//   R  dynamics   closed-form derivatives of rnea_deriv.hpp, distributed: lane g computes its body's spatial inertia Y_g, force f_g
//                 and momentum operator B_g (12 numbers, below); a suffix scan over the lanes (DPP moves) makes them the composites
//                 of bodies g..NQ-1; lane m then takes the pairs (m, k <= m) with S_k, psi_k, chi_k of the other joints from LDS,
//                 and in a second pass the pairs (m, k > m) from the three vectors the diagonal lanes left there: row m of M,
//                 dtau/dq, dtau/dqd ends up in lane m's registers
//   G  geometry   axis / origin of every joint and the world points riding on each link published through LDS; lane g: its entry of
//                 the cost gradient, its row of the (exact or Gauss-Newton) EE cost Hessian; collision rows g, g + 8: value
//                 forward, gradient in reverse mode (the clamps' tie rules of utils.py:94-118 as in device_model.hpp)
//   S  set-up     every row stays with the lane that computed its coefficients (torque row g and collision rows g, g + 8 on lane g,
//                 the network's row on lane 7, box rows i on lane i mod 8): bounds, initial slacks and multipliers, the stage's
//                 partial sums (initial residual, complementarity) -- the arithmetic of k_qp_setup; C^T e of the stationarity
//                 residual is a DPP butterfly sum over the 8 lanes
//   W  write-out  straight from the owner lanes' registers: the Jacobian is never staged in LDS (it was, as the image's 156 doubles:
//                 the block's 19 KB of LDS then needed a CU with at most six QP wavefronts; +8 KB of padding cost 3 % of a step)
// Registers: composites (28 doubles) + this lane's rows (18 + 6 per collision row): < 256 VGPRs, no scratch; LDS 1.2 KB per node.
//
// The network's row (value and gradient w.r.t. the state, kernels_mlp.hpp) is evaluated BEFORE this pass and read from a compact
// per-node buffer nn[node][1 + NX].
#pragma once
#include "kernel_qp.hpp"
#include "rnea_deriv.hpp"

namespace smpc {

constexpr int SB_G = 8;   // lanes per node
#ifndef SB_WAVES
#define SB_WAVES 2      // wavefronts per SIMD the builder is compiled for (register cap 512 / SB_WAVES)
#endif

// per-node LDS block (offsets in doubles): ONE exchange area, reused by the phases
template <int NQ, int MR_MAX> struct SbLds {
    static constexpr int NX = 2 * NQ;
    // start: sin / cos per joint
    static constexpr int X_SC = 0, XA_END = 2 * NQ;
    // phase R: per joint (S | psi, then c_kk | chi, then d_kk | Yc_k S_k)
    static constexpr int X_SPC = 0, SPC_D = 24, XR_END = SPC_D * NQ;
    // phase G: world points | (z, p, J_ee) per joint
    static constexpr int X_PT = 0, X_ZP = X_PT + 3 * SMPC_MAX_POINTS, XG_END = X_ZP + 9 * NQ;
    // phase S: e = -(lambda_l - lambda_u) of the box rows | z0 (the fixed dx_0 at node 0) | cost gradient w.r.t. q
    static constexpr int X_E = 0, X_Z0 = NX, X_GQ = X_Z0 + NX, XS_END = X_GQ + NQ;
    static constexpr int m1 = XA_END > XR_END ? XA_END : XR_END, m2 = XG_END > XS_END ? XG_END : XS_END;
    static constexpr int SIZE = qp_even_c(m1 > m2 ? m1 : m2);
};

// value of the lane OFF places up (lane i <- lane i + OFF) within its row of 16 lanes, as a DPP move: a VALU instruction with no
// round trip through the LDS crossbar (the composites' suffix scan moves 3 x 28 doubles this way; as ds_bpermute, and with the
// 36-entry B, it was latency-bound: 0.21 of the first version's 0.56 ms).  Lanes whose source lies outside the row read 0 (bound_ctrl).
template <int OFF> __device__ __forceinline__ double dpp_up(double x) {
    static_assert(OFF >= 1 && OFF <= 15, "row_shl:1..15");
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + OFF, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + OFF, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// butterfly over the 8 lanes of a node, all as DPP moves: xor 1, xor 2 (quad permutes), then the mirror of the half row (i <-> 7 - i)
template <int CTRL> __device__ __forceinline__ double dpp_mov(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double grp_sum(double v) {      // every lane of the group ends with the sum
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);     // row_half_mirror
    return v;
}
__device__ __forceinline__ double grp_max(double v) {
    v = fmax(v, dpp_mov<0xB1>(v));
    v = fmax(v, dpp_mov<0x4E>(v));
    v = fmax(v, dpp_mov<0x141>(v));
    return v;
}

// One node.  Every lane of the wavefront calls this (cross-lane steps inside); `valid` = this group has a node.  L = the group's
// LDS block.  ws = the stage's record.  g = lane within the group.
template <int NQ, int MRT>
__device__ __forceinline__ void stage_build(const smpc_problem_desc* __restrict__ D, const QpLayout<NQ>& Ly, const int N, const int k,
                                            const bool valid, const int g, double* __restrict__ L, const double* __restrict__ x0b,
                                            const double* __restrict__ xk, const double* __restrict__ uk, const double* __restrict__ pk,
                                            const double* __restrict__ lo_k, const double* __restrict__ hi_k,
                                            const double* __restrict__ zl_st, const double* __restrict__ nnk, double* __restrict__ w) {
    constexpr int NX = 2 * NQ, NU = NQ, NZ = 3 * NQ, NQP = QpLayout<NQ>::NQP, NZP = QpLayout<NQ>::NZP;
    constexpr int MR_MAX = MRT >= 0 ? MRT : SMPC_MAX_ROWS;
    constexpr int NSLOT = (MR_MAX + SB_G - 1) / SB_G;      // collision rows per lane (row g + 8 s on lane g)
    constexpr int NNL = SB_G - 1;                          // the lane that owns the network's row
    static_assert(NQ < SB_G, "lane SB_G - 1 carries the network's row, lanes 0 .. NQ - 1 a joint each");
    static_assert(MR_MAX <= 2 * SB_G && 2 * NQ <= 2 * SB_G, "at most two collision rows and two box rows per lane");
    using LD = SbLds<NQ, MR_MAX>;
    const int MR = Ly.MR, MRP = Ly.MRP;
    const int rT0 = NX, rC0 = NX + NQ, rNN = NX + NQ + MR;
    const bool last = (k == N);
    const bool jl = g < NQ;                       // this lane owns a joint
    const int gj = jl ? g : NQ - 1;               // (clamped: the other lanes shadow the last joint and write nothing)
    double* const sX = L;
    const double dt = D->dt, cB = 0.5 * dt * dt;

    // ---- inputs: every lane of the group reads the node's state and control (same addresses: one transaction per group) -----------
    double qd[NQ], qdd[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        qd[i] = xk[NQ + i];
        qdd[i] = last ? 0.0 : uk[i];
    }
    {
        double s, c;
        sincos(xk[gj], &s, &c);
        if (jl) { sX[LD::X_SC + 2 * g] = s; sX[LD::X_SC + 2 * g + 1] = c; }
    }
    lds_fence();

    // ---- forward kinematics up to this lane's joint (world frame, as rnea_deriv.hpp / device_model.hpp) ---------------------------
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    rd::V3 pj = rd::v3(0, 0, 0);
    rd::SV S{rd::v3(0, 0, 0), rd::v3(0, 0, 0)}, vv{rd::v3(0, 0, 0), rd::v3(0, 0, 0)},
        aa{rd::v3(0, 0, 0), rd::v3(-D->gravity[0], -D->gravity[1], -D->gravity[2])};
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        if (i <= gj) {
            const smpc_joint& Ji = D->joints[i];
            pj = pj + rd::rot(R, Ji.p0);
            const double s = sX[LD::X_SC + 2 * i], c = sX[LD::X_SC + 2 * i + 1], v = 1.0 - c;
            const double ax = Ji.axis[0], ay = Ji.axis[1], az = Ji.axis[2];
            const double Q[9] = {c + v * ax * ax,      v * ax * ay - s * az, v * ax * az + s * ay,
                                 v * ay * ax + s * az, c + v * ay * ay,      v * ay * az - s * ax,
                                 v * az * ax - s * ay, v * az * ay + s * ax, c + v * az * az};
            double A[9];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) A[3 * r + cc] = R[3 * r] * Ji.R0[cc] + R[3 * r + 1] * Ji.R0[3 + cc] + R[3 * r + 2] * Ji.R0[6 + cc];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) R[3 * r + cc] = A[3 * r] * Q[cc] + A[3 * r + 1] * Q[3 + cc] + A[3 * r + 2] * Q[6 + cc];
            const rd::V3 z = rd::rot(R, Ji.axis);
            S = rd::SV{z, rd::cross(pj, z)};
            const rd::SV jv = S * qd[i];
            vv = vv + jv;
            aa = aa + S * qdd[i] + rd::mxm(vv, jv);
        }
    }
    const rd::V3 zj = S.a;
    lds_fence();     // (sin / cos have been read: the exchange area goes to the dynamics)

    // ---- R: torque row and its Jacobians (rnea_deriv.hpp, distributed) -------------------------------------------------------------
    // The momentum operator of rnea_deriv.hpp, B m = Y (m x v) + m x* (Y v) + v x* (Y m), has a closed form with TWELVE numbers
    // instead of 36: with v = (w; u), h = Y v = (n; f) (f = m u + w x mc, the linear momentum) and m = (a; l),
    //     B m = (Baa a ; -2 f x a),   Baa = [w]x I - I [w]x - [n]x - (u mc^T + mc u^T - 2 (mc . u) 1)
    // -- the blocks acting on l cancel identically ([w]x[mc]x - [mc]x[w]x = [w x mc]x).  The composite over the bodies l >= m is
    // then Baa_c (9) and the composite linear momentum (3): a third of the registers, of the scan and of the products.
    rd::Inertia Yc;
    double Ba[9];
    rd::V3 Pl;
    rd::SV F;
    {
        const smpc_joint& Jg = D->joints[gj];
        const rd::V3 c = pj + rd::rot(R, Jg.com);
        const double* I = Jg.inertia;
        double T[9];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            T[3 * r + 0] = R[3 * r] * I[0] + R[3 * r + 1] * I[1] + R[3 * r + 2] * I[2];
            T[3 * r + 1] = R[3 * r] * I[1] + R[3 * r + 1] * I[3] + R[3 * r + 2] * I[4];
            T[3 * r + 2] = R[3 * r] * I[2] + R[3 * r + 1] * I[4] + R[3 * r + 2] * I[5];
        }
        const double m = jl ? Jg.mass : 0.0, cc = rd::dot(c, c);
        const double on = jl ? 1.0 : 0.0;
        Yc.m = m;
        Yc.mc = c * m;
        Yc.I[0] = on * (T[0] * R[0] + T[1] * R[1] + T[2] * R[2]) + m * (cc - c.x * c.x);
        Yc.I[1] = on * (T[0] * R[3] + T[1] * R[4] + T[2] * R[5]) - m * c.x * c.y;
        Yc.I[2] = on * (T[0] * R[6] + T[1] * R[7] + T[2] * R[8]) - m * c.x * c.z;
        Yc.I[3] = on * (T[3] * R[3] + T[4] * R[4] + T[5] * R[5]) + m * (cc - c.y * c.y);
        Yc.I[4] = on * (T[3] * R[6] + T[4] * R[7] + T[5] * R[8]) - m * c.y * c.z;
        Yc.I[5] = on * (T[6] * R[6] + T[7] * R[7] + T[8] * R[8]) + m * (cc - c.z * c.z);
        const rd::SV hm = rd::apply(Yc, vv);
        F = rd::apply(Yc, aa) + rd::mxf(vv, hm);
        Pl = hm.l;
        {
            const rd::V3 wv = vv.a, uv = vv.l, nv = hm.a, mc = Yc.mc;
            const double Im[9] = {Yc.I[0], Yc.I[1], Yc.I[2], Yc.I[1], Yc.I[3], Yc.I[4], Yc.I[2], Yc.I[4], Yc.I[5]};
            const double W[9] = {0.0, -wv.z, wv.y, wv.z, 0.0, -wv.x, -wv.y, wv.x, 0.0};      // [w]x
            const double uu[3] = {uv.x, uv.y, uv.z}, mm3[3] = {mc.x, mc.y, mc.z};
            const double Nx[9] = {0.0, -nv.z, nv.y, nv.z, 0.0, -nv.x, -nv.y, nv.x, 0.0};    // [n]x
            const double mcu2 = 2.0 * rd::dot(mc, uv);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int cix = 0; cix < 3; cix++) {
                    double sacc = -Nx[3 * r + cix] - (uu[r] * mm3[cix] + mm3[r] * uu[cix]) + (r == cix ? mcu2 : 0.0);
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        if (t != r) sacc += W[3 * r + t] * Im[3 * t + cix];       // (the diagonal of [w]x is zero)
                        if (t != cix) sacc -= Im[3 * r + t] * W[3 * t + cix];
                    }
                    Ba[3 * r + cix] = sacc;
                }
        }
        // (lanes without a joint carry a zero body: mass 0, inertia 0 -> Y, B, f all zero)
        // publish S, psi = v x S, chi = a x S - psi x v
        const rd::SV psi = rd::mxm(vv, S);
        const rd::SV chi = rd::mxm(aa, S) - rd::mxm(psi, vv);
        if (jl) {
            double* o = sX + LD::X_SPC + LD::SPC_D * g;
            o[0] = S.a.x; o[1] = S.a.y; o[2] = S.a.z; o[3] = S.l.x; o[4] = S.l.y; o[5] = S.l.z;
            o[6] = psi.a.x; o[7] = psi.a.y; o[8] = psi.a.z; o[9] = psi.l.x; o[10] = psi.l.y; o[11] = psi.l.z;
            o[12] = chi.a.x; o[13] = chi.a.y; o[14] = chi.a.z; o[15] = chi.l.x; o[16] = chi.l.y; o[17] = chi.l.z;
        }
    }
    // composites: suffix sums over the lanes of the group (lane g <- bodies g .. NQ-1)
    auto scan_step = [&](auto off_tag) {
        constexpr int OFF = decltype(off_tag)::value;
        // (the neighbour node's lanes share the row of 16: a select, not a multiplication by zero -- a NaN or Inf in the other
        //  node's composites must not reach this node's)
        const bool take = g + OFF < SB_G;
        auto acc = [&](double& x) { const double up = dpp_up<OFF>(x); x = take ? x + up : x; };
        acc(Yc.m); acc(Yc.mc.x); acc(Yc.mc.y); acc(Yc.mc.z);
#pragma unroll
        for (int i = 0; i < 6; i++) acc(Yc.I[i]);
#pragma unroll
        for (int i = 0; i < 9; i++) acc(Ba[i]);
        acc(Pl.x); acc(Pl.y); acc(Pl.z);
        acc(F.a.x); acc(F.a.y); acc(F.a.z); acc(F.l.x); acc(F.l.y); acc(F.l.z);
    };
    scan_step(std::integral_constant<int, 1>{});
    scan_step(std::integral_constant<int, 2>{});
    scan_step(std::integral_constant<int, 4>{});
    lds_fence();
    // this lane's torque row, in its registers: rowT[c] = d tau_g / d z_c, z = [u | q | v]
    double rowT[NZ];
#pragma unroll
    for (int c = 0; c < NZ; c++) rowT[c] = 0.0;
    double tau_g = 0.0;
    if (!last) {
        auto applyB = [&](const rd::SV& mv) {      // (Baa a ; -2 Pl x a)
            const rd::V3 a = mv.a;
            const rd::V3 ya = rd::v3(Ba[0] * a.x + Ba[1] * a.y + Ba[2] * a.z, Ba[3] * a.x + Ba[4] * a.y + Ba[5] * a.z,
                                     Ba[6] * a.x + Ba[7] * a.y + Ba[8] * a.z);
            return rd::SV{ya, rd::cross(Pl, a) * (-2.0)};
        };
        auto ld_sv = [&](const double* o) { return rd::SV{rd::v3(o[0], o[1], o[2]), rd::v3(o[3], o[4], o[5])}; };
        auto st_sv = [&](double* o, const rd::SV& x) { o[0] = x.a.x; o[1] = x.a.y; o[2] = x.a.z; o[3] = x.l.x; o[4] = x.l.y; o[5] = x.l.z; };
        // Pass 1, k = 0 .. NQ-1: every lane m >= k takes the pair (m, k): M[m][k] and row m's dtau_m/dq_k, dtau_m/dqd_k.  The lane on
        // the diagonal (m = k) leaves the three vectors the entries ABOVE the diagonal need -- c_kk, d_kk + S_k x* F_k, Yc_k S_k --
        // in LDS (over psi_k / chi_k, dead after this pass).  Pass 2, k = 1 .. NQ-1: every lane j < k: M[j][k] = S_j . Yc_k S_k,
        // dtau_j/dqd_k = S_j . c_kk, dtau_j/dq_k = S_j . d_kk.
        // (rolled loops: the row's entries are placed by select chains -- a register array cannot be indexed by the loop counter --
        //  which costs 36 selects per pass but keeps the code and the register pressure of ONE pair)
        auto put3 = [&](int kk, bool take, double m_, double dq_, double dv_) {
#pragma unroll
            for (int c = 0; c < NQ; c++) {
                const bool hit = take && c == kk;
                rowT[c] = hit ? m_ : rowT[c];
                rowT[NQ + c] = hit ? dq_ : rowT[NQ + c];
                rowT[2 * NQ + c] = hit ? dv_ : rowT[2 * NQ + c];
            }
        };
#pragma unroll 1
        for (int kk = 0; kk < NQ; kk++) {
            double* o = sX + LD::X_SPC + LD::SPC_D * kk;
            const rd::SV Sk = ld_sv(o), psi = ld_sv(o + 6), chi = ld_sv(o + 12);
            const rd::SV yS = rd::apply(Yc, Sk);
            const rd::SV cv = applyB(Sk) + rd::apply(Yc, psi) * 2.0;
            rd::SV dd = applyB(psi) + rd::apply(Yc, chi);
            if (kk == g) dd = dd + rd::mxf(Sk, F);
            put3(kk, kk <= g, rd::mdotf(S, yS), rd::mdotf(S, dd), rd::mdotf(S, cv));     // M[g][kk], dtau_g/dq_kk, dtau_g/dqd_kk
            if (kk == g) { st_sv(o + 6, cv); st_sv(o + 12, dd); st_sv(o + 18, yS); }
        }
        lds_fence();
#pragma unroll 1
        for (int kk = 1; kk < NQ; kk++) {
            const double* o = sX + LD::X_SPC + LD::SPC_D * kk;
            const rd::SV cv = ld_sv(o + 6), dd = ld_sv(o + 12), yS = ld_sv(o + 18);
            put3(kk, g < kk, rd::mdotf(S, yS), rd::mdotf(S, dd), rd::mdotf(S, cv));
        }
        tau_g = rd::mdotf(S, F);
        if (!jl) {
#pragma unroll
            for (int c = 0; c < NZ; c++) rowT[c] = 0.0;
        }
    }
    lds_fence();     // (the exchange area changes hands: dynamics -> geometry)

    // ---- G: geometry.  Publish axis, origin; the world points that ride on this link (world-fixed points: lane 0) --------------------
    if (jl) {
        double* zp = sX + LD::X_ZP + 9 * g;
        zp[0] = zj.x; zp[1] = zj.y; zp[2] = zj.z; zp[3] = pj.x; zp[4] = pj.y; zp[5] = pj.z;
    }
    for (int pt = 0; pt < D->n_points; pt++) {
        const smpc_point& P = D->points[pt];
        if (P.link == g && jl) {
            const rd::V3 wv = pj + rd::rot(R, P.local);
            sX[LD::X_PT + 3 * pt] = wv.x; sX[LD::X_PT + 3 * pt + 1] = wv.y; sX[LD::X_PT + 3 * pt + 2] = wv.z;
        } else if (P.link < 0 && g == 0) {
            sX[LD::X_PT + 3 * pt] = P.local[0]; sX[LD::X_PT + 3 * pt + 1] = P.local[1]; sX[LD::X_PT + 3 * pt + 2] = P.local[2];
        }
    }
    lds_fence();
    auto ld_pt = [&](int pt) { return rd::v3(sX[LD::X_PT + 3 * pt], sX[LD::X_PT + 3 * pt + 1], sX[LD::X_PT + 3 * pt + 2]); };
    const double cs = last ? D->cost_scale_term : D->cost_scale_stage;
    const double lm = last ? D->lm_term : D->lm_stage;
    const bool reach = D->cost_kind == SMPC_COST_REACH;
    // EE point, cost gradient and Hessian (cost_definition.py:69-96); lane g: entry g of the gradient, row g of the Hessian
    double gz_q = 0.0;                   // cost gradient w.r.t. q_g (scaled)
    {
        const rd::V3 ee = ld_pt(D->ee_point);
        const int link = D->points[D->ee_point].link;
        const bool on = jl && g <= link;
        const rd::V3 t = rd::cross(zj, ee - pj);         // dP/dq_g = z_g x (P - p_g) for g <= link
        const rd::V3 Jg = rd::v3(on ? t.x : 0.0, on ? t.y : 0.0, on ? t.z : 0.0);
        const double dx = ee.x - pk[0], dy = ee.y - pk[1], dz = ee.z - pk[2];
        if (jl) {
            double* zp = sX + LD::X_ZP + 9 * g;
            zp[6] = Jg.x; zp[7] = Jg.y; zp[8] = Jg.z;
        }
        lds_fence();
        const double Q2 = 2.0 * D->Q;
        if (reach) gz_q = cs * (Q2 * (Jg.x * dx + Jg.y * dy + Jg.z * dz));
        if (jl) {
            double* hq = w + Ly.oIMG + Ly.iHQQ + g * NQ;
            for (int j = 0; j < NQ; j++) {
                const double* zo = sX + LD::X_ZP + 9 * j;
                double hv = Jg.x * zo[6] + Jg.y * zo[7] + Jg.z * zo[8];
                if (D->hessian == SMPC_HESS_EXACT) {
                    // d2 P / dq_i dq_j = z_min x (z_max x (P - p_max)) for min(i,j) <= max(i,j) <= link
                    const int lo = g < j ? g : j, hi = g < j ? j : g;
                    if (hi <= link) {
                        const double* zh = sX + LD::X_ZP + 9 * hi;
                        const double* zl = sX + LD::X_ZP + 9 * lo;
                        const rd::V3 Jhi = rd::cross(rd::v3(zh[0], zh[1], zh[2]), rd::v3(ee.x - zh[3], ee.y - zh[4], ee.z - zh[5]));
                        hv += rd::dot(rd::v3(dx, dy, dz), rd::cross(rd::v3(zl[0], zl[1], zl[2]), Jhi));
                    }
                }
                if (valid) stnt_su((reach ? cs * (Q2 * hv) : 0.0) + (g == j ? lm : 0.0), hq + j);
            }
        }
        if (NQ * NQ < qp_even_c(NQ * NQ) && g == SB_G - 1 && valid) stnt_su(0.0, w + Ly.oIMG + Ly.iHQQ + NQ * NQ);
    }
    // Collision rows (env_model.py:263-316): row g + 8 s ON LANE g, value forward, gradient in reverse mode: the row's adjoints
    // w.r.t. its (up to four) moving points, then d row / d q_j = z_j . sum over the points riding on links >= j of
    // (P - p_j) x Pbar.  The clamps pass a derivative exactly where the forward-mode duals of device_model.hpp do (CasADi's tie
    // rules, utils.py:94-118): clamp01(t) for 0 <= t <= 1, the point-segment clamp likewise.  The row stays in this lane's registers.
    double growv[NSLOT > 0 ? NSLOT : 1], grow[NSLOT > 0 ? NSLOT : 1][NQ];
#pragma unroll
    for (int sl = 0; sl < NSLOT; sl++) {
        const int r = g + SB_G * sl;
        const bool have = r < MR;
        const smpc_row& row = D->rows[have ? r : 0];
        const rd::V3 Z0v = rd::v3(0, 0, 0);
        rd::V3 PA = Z0v, PB = Z0v, PC = Z0v, PD = Z0v, gA = Z0v, gB = Z0v, gC = Z0v, gD = Z0v;
        int lA = -1, lB = -1, lC = -1, lD = -1;      // links of the moving points (-1: fixed / unused)
        double val = 0.0;
        if (have) {
            if (row.kind == SMPC_ROW_SEG_FIXEDSEG || row.kind == SMPC_ROW_SEG_SEG) {
                PA = ld_pt(row.pa); PB = ld_pt(row.pb);
                lA = D->points[row.pa].link; lB = D->points[row.pb].link;
                if (row.kind == SMPC_ROW_SEG_SEG) {
                    PC = ld_pt(row.pc); PD = ld_pt(row.pd);
                    lC = D->points[row.pc].link; lD = D->points[row.pd].link;
                } else {
                    PC = rd::v3(row.C[0], row.C[1], row.C[2]); PD = rd::v3(row.D[0], row.D[1], row.D[2]);
                }
                // utils.py:94-113
                const rd::V3 ab = PB - PA, cd = PD - PC, ac = PC - PA;
                const double Rr = rd::dot(ab, cd), S1 = rd::dot(ab, ac), D1 = rd::dot(ab, ab), S2 = rd::dot(cd, ac), D2 = rd::dot(cd, cd);
                const double num = S1 * D2 - S2 * Rr, den = D1 * D2 - (Rr * Rr + 1e-5);
                const double t0 = num / den;
                const bool p1 = t0 <= 1.0 && t0 >= 0.0;
                const double t1 = t0 <= 1.0 ? (t0 >= 0.0 ? t0 : 0.0) : 1.0;
                const double u0 = (t1 * Rr - S2) / D2;
                const bool p2 = u0 <= 1.0 && u0 >= 0.0;
                const double u1 = u0 <= 1.0 ? (u0 >= 0.0 ? u0 : 0.0) : 1.0;
                const double t2 = (u1 * Rr + S1) / D1;
                const bool p3 = t2 <= 1.0 && t2 >= 0.0;
                const double t3 = t2 <= 1.0 ? (t2 >= 0.0 ? t2 : 0.0) : 1.0;
                const rd::V3 wv = ab * t3 - cd * u1 - ac;
                val = rd::dot(wv, wv);
                // reverse sweep
                const rd::V3 wb = wv * 2.0;
                rd::V3 abb = wb * t3, cdb = wb * (-u1), acb = wb * (-1.0);
                const double t3b = rd::dot(wb, ab);
                double u1b = -rd::dot(wb, cd);
                const double t2b = p3 ? t3b : 0.0;
                double Rb = t2b * u1 / D1, S1b = t2b / D1, D1b = -t2b * t2 / D1;
                u1b += t2b * Rr / D1;
                const double u0b = p2 ? u1b : 0.0;
                const double t1b = u0b * Rr / D2;
                Rb += u0b * t1 / D2;
                double S2b = -u0b / D2, D2b = -u0b * u0 / D2;
                const double t0b = p1 ? t1b : 0.0;
                const double numb = t0b / den, denb = -t0b * t0 / den;
                S1b += numb * D2; D2b += numb * S1 + denb * D1; S2b += -numb * Rr; Rb += -numb * S2 - 2.0 * denb * Rr; D1b += denb * D2;
                abb = abb + cd * Rb + ac * S1b + ab * (2.0 * D1b);
                cdb = cdb + ab * Rb + ac * S2b + cd * (2.0 * D2b);
                acb = acb + ab * S1b + cd * S2b;
                gA = Z0v - abb - acb; gB = abb; gC = acb - cdb; gD = cdb;
            } else if (row.kind == SMPC_ROW_SEG_POINT) {
                PA = ld_pt(row.pa); PB = ld_pt(row.pb);
                lA = D->points[row.pa].link; lB = D->points[row.pb].link;
                const rd::V3 Pp = rd::v3(row.C[0], row.C[1], row.C[2]);
                // utils.py:115-118 (fmin(fmax(., 0), 1))
                const rd::V3 pa_ = Pp - PA, ba = PB - PA;
                const double t0 = rd::dot(pa_, ba) / row.len2;
                const bool ps = t0 >= 0.0 && t0 <= 1.0;
                const double t = t0 >= 0.0 ? (t0 <= 1.0 ? t0 : 1.0) : 0.0;
                const rd::V3 wv = Pp - (PA + ba * t);
                val = rd::dot(wv, wv);
                const rd::V3 wb = wv * 2.0;
                const double tb = -rd::dot(wb, ba);
                const double t0b = (ps ? tb : 0.0) / row.len2;
                gA = wb * (t - 1.0) - (ba + pa_) * t0b;
                gB = wb * (-t) + pa_ * t0b;
            } else if (row.kind == SMPC_ROW_POINT_POINT) {
                PA = ld_pt(row.pa);
                lA = D->points[row.pa].link;
                const rd::V3 wv = PA - rd::v3(row.C[0], row.C[1], row.C[2]);
                val = rd::dot(wv, wv);
                gA = wv * 2.0;
            } else {
                PA = ld_pt(row.pa);
                lA = D->points[row.pa].link;
                val = (row.axis == 0 ? PA.x : (row.axis == 1 ? PA.y : PA.z)) - row.offset;
                gA = rd::v3(row.axis == 0, row.axis == 1, row.axis == 2);
            }
        }
        growv[sl] = val;
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            const double* zo = sX + LD::X_ZP + 9 * j;
            const rd::V3 zz = rd::v3(zo[0], zo[1], zo[2]), pp = rd::v3(zo[3], zo[4], zo[5]);
            rd::V3 acc = Z0v;
            acc = acc + rd::cross(PA - pp, gA) * (j <= lA ? 1.0 : 0.0);
            acc = acc + rd::cross(PB - pp, gB) * (j <= lB ? 1.0 : 0.0);
            acc = acc + rd::cross(PC - pp, gC) * (j <= lC ? 1.0 : 0.0);
            acc = acc + rd::cross(PD - pp, gD) * (j <= lD ? 1.0 : 0.0);
            grow[sl][j] = have ? rd::dot(zz, acc) : 0.0;
        }
    }
    lds_fence();     // (the exchange area changes hands: geometry -> set-up)

    // ---- S: the stage record (the arithmetic of k_qp_setup), every row on the lane that holds its coefficients --------------------
    bool nn_on = false;
    if (D->nn_mode != SMPC_NN_NONE && k >= 1 && (D->nn_mode == SMPC_NN_ALL || last)) nn_on = pk[4] > 0.0;
    double wsoft = nn_on ? (last ? D->nn_soft_e : D->nn_soft_run) : -1.0;
    if (zl_st && wsoft >= 0.0) {
        wsoft = zl_st[k];    // cost_set(k, 'zl', .) on a row the formulation made soft; zero weight = row absent (k_qp_setup)
        if (wsoft == 0.0) { nn_on = false; wsoft = -1.0; }
    }
    double* const sE = sX + LD::X_E;          // e of the box rows (they add to C^T e one column each)
    double* const sZ0 = sX + LD::X_Z0;
    double* const sGQ = sX + LD::X_GQ;
    for (int i = g; i < NX; i += SB_G) sZ0[i] = k == 0 ? x0b[i] - xk[i] : 0.0;
    if (jl) sGQ[g] = gz_q;
    lds_fence();
    // dynamics defect b; lane g: components g, g + 8
    double bmax = 0.0;
    auto defect = [&](int i) -> double {
        if (last) return 0.0;
        const double* xn = xk + NX;
        const int ii = i < NQ ? i : i - NQ;
        const double ukk = uk[ii];
        return i < NQ ? xk[ii] + dt * xk[NQ + ii] + cB * ukk - xn[ii] : xk[NQ + ii] + dt * ukk - xn[NQ + ii];
    };
    for (int i = g; i < NX; i += SB_G) {
        const double bb = defect(i);
        bmax = fmax(bmax, fabs(bb));
        if (valid) stnt_su(bb, w + Ly.oIMG + Ly.iB + i);
    }
    bmax = grp_max(bmax);
    const double bflag = bmax > 0.0 ? 1.0 : 0.0;
    // cost gradient: u part R u, q part from the EE point (owned by the joint lanes), v part 0
    auto gz_u = [&](int c) -> double { return (reach && !last) ? cs * 2.0 * D->R * uk[c] : 0.0; };
    for (int hz = g; hz < NZP; hz += SB_G) {
        if (hz >= NU && hz < NU + NQ) continue;
        if (valid) stnt_su(hz < NU ? gz_u(hz) : 0.0, w + Ly.oIMG + Ly.iGZ + hz);
    }
    if (jl && valid) stnt_su(gz_q, w + Ly.oIMG + Ly.iGZ + NU + g);
    if (g < 4 && valid) {
        const double huu = (reach && !last ? cs * 2.0 * D->R : 0.0) + lm;
        stnt_su(g == 0 ? huu : (g == 1 ? lm : (g == 2 ? wsoft : bflag)), w + Ly.oIMG + Ly.iSC + g);
        stnt_su(g == 0 ? wsoft : (g == 1 ? bflag : 0.0), w + Ly.oSL + g);
    }
    if (valid) stnt_su(0.0, w + Ly.oWC + g);

    double r0_loc = 0.0, mu_acc = 0.0, inf0 = 0.0;
    int cnt = 0;
    // initial slacks / multipliers of one row (k_qp_setup); returns e = -(lambda_l - lambda_u)
    auto init_row = [&](bool present, int r, double lo, double hi, double cz, double cn, bool soft) -> double {
        double tl = 1.0, ll = 0.0, tu = 1.0, lu = 0.0;
        if (present) {
            // slack floor of the starting point: a hard row starts at least QP_THR_HARD |c|_inf inside its bound
            const double thr = soft ? QP_THR : QP_THR_HARD * (cn > 0.0 ? cn : 1.0);
            if (lo > -QP_ABSENT) {
                const double s0 = soft ? QP_THR : 0.0;
                const double slack = cz + s0 - lo;
                tl = fmax(slack, thr);
                ll = D->qp_mu0 / tl;
                if (soft) ll = fmin(ll, 0.5 * wsoft);
                r0_loc = fmax(r0_loc, fabs(slack - tl));
                cnt += soft ? 2 : 1;
                mu_acc += ll * tl;
                if (soft) { mu_acc += (wsoft - ll) * s0; tu = s0; }   // the slack rides in the unused upper side
            }
            if (hi < QP_ABSENT) {
                const double slack = hi - cz;
                tu = fmax(slack, thr);
                lu = D->qp_mu0 / tu;
                r0_loc = fmax(r0_loc, fabs(slack - tu));
                cnt += 1;
                mu_acc += lu * tu;
            }
            if (valid) {
                stnt_su(dbl2{lo, hi}, reinterpret_cast<dbl2*>(w + Ly.oR0) + r);
                stnt_su(dbl2{tl, tu}, reinterpret_cast<dbl2*>(w + Ly.oR1) + r);
                stnt_su(dbl2{ll, lu}, reinterpret_cast<dbl2*>(w + Ly.oR2) + r);
            }
        }
        return -(ll - lu);
    };
    // box rows i = g, g + 8
    for (int i = g; i < NX; i += SB_G) {
        double lo = -QP_ABSENT, hi = QP_ABSENT;
        const double l = lo_k[i], h = hi_k[i];
        if (k >= 1 && fabs(l) < SMPC_INF) lo = l - xk[i];
        if (k >= 1 && fabs(h) < SMPC_INF) hi = h - xk[i];
        sE[i] = init_row(true, i, lo, hi, sZ0[i], 1.0, false);
    }
    // torque row g
    double eT = 0.0;
    {
        double lo = -QP_ABSENT, hi = QP_ABSENT, cz = 0.0, cn = 0.0;
        const double tm = D->joints[gj].tau_max;
        if (!last && tm < SMPC_INF) { lo = -tm - tau_g; hi = tm - tau_g; }
#pragma unroll
        for (int c = 0; c < NZ; c++) cn = fmax(cn, fabs(rowT[c]));
        if (k == 0) {
#pragma unroll
            for (int i = 0; i < NX; i++) cz = fma(rowT[NU + i], sZ0[i], cz);
        }
        eT = init_row(jl, rT0 + gj, lo, hi, cz, cn, false);
        if (!jl) eT = 0.0;
    }
    // collision rows g + 8 s
    double eC[NSLOT > 0 ? NSLOT : 1];
#pragma unroll
    for (int sl = 0; sl < NSLOT; sl++) {
        const int r = g + SB_G * sl;
        const bool have = r < MR;
        const smpc_row& row = D->rows[have ? r : 0];
        double lo = -QP_ABSENT, hi = QP_ABSENT, cz = 0.0, cn = 0.0;
        if (k >= 1 && fabs(row.lb) < SMPC_INF) lo = row.lb - growv[sl];
        if (k >= 1 && fabs(row.ub) < SMPC_INF) hi = row.ub - growv[sl];
#pragma unroll
        for (int j = 0; j < NQ; j++) cn = fmax(cn, fabs(grow[sl][j]));
        if (k == 0) {
#pragma unroll
            for (int j = 0; j < NQ; j++) cz = fma(grow[sl][j], sZ0[j], cz);
        }
        eC[sl] = init_row(have, rC0 + r, lo, hi, cz, cn, false);
        if (!have) eC[sl] = 0.0;
        // collision rows at node 0 (controller.py:77-79): constants of the QP; a violated one = QP infeasible
        if (have && k == 0 && D->rows_at_node0) {
            const double v = growv[sl] + cz;
            if ((fabs(row.lb) < SMPC_INF && v < row.lb - D->qp_tol) || (fabs(row.ub) < SMPC_INF && v > row.ub + D->qp_tol)) inf0 = 1.0;
        }
    }
    // the network's row: lane NNL
    double eN = 0.0;
    {
        const bool mine = g == NNL;
        double lo = -QP_ABSENT, cn = 0.0;
        if (nn_on && mine) {
            lo = 0.0 - nnk[0];
            for (int i = 0; i < NX; i++) cn = fmax(cn, fabs(nnk[1 + i]));
        }
        // (the row exists from node 1 on: dx_0 never enters it)
        eN = init_row(mine, rNN, lo, QP_ABSENT, 0.0, cn, wsoft >= 0.0);
        if (!mine) eN = 0.0;
    }
    for (int r = g; r < 32; r += SB_G)
        if (valid) { stnt_su(0.0, w + Ly.oCZA + r); stnt_su(0.0, w + Ly.oCZN + r); }
    for (int hz = g; hz < NZ; hz += SB_G) {
        const double z0 = hz >= NU ? sZ0[hz - NU] : 0.0;
        if (valid) { stnt_su(z0, w + Ly.oZ + hz); stnt_su(z0, w + Ly.oZN + hz); }
    }
    lds_fence();
    // stationarity residual at the initial point (pi = 0): g + C^T e; dx_0 does not enter.  This lane's share of C^T e, summed
    // over the group by a DPP butterfly; the box rows' e come in from LDS
    {
        double pc[NZ];
#pragma unroll
        for (int c = 0; c < NZ; c++) pc[c] = rowT[c] * eT;
#pragma unroll
        for (int sl = 0; sl < NSLOT; sl++)
#pragma unroll
            for (int j = 0; j < NQ; j++) pc[NU + j] = fma(grow[sl][j], eC[sl], pc[NU + j]);
        if (nn_on && g == NNL) {
#pragma unroll
            for (int i = 0; i < NX; i++) pc[NU + i] = fma(nnk[1 + i], eN, pc[NU + i]);
        }
#pragma unroll
        for (int c = 0; c < NZ; c++) {
            const double sc = grp_sum(pc[c]);
            const bool skip = (k == 0 && c >= NU) || (last && c < NU);
            const double gz = c < NU ? gz_u(c) : (c < NU + NQ ? sGQ[c - NU] : 0.0);
            const double a = gz + sc + (c >= NU ? sE[c - NU] : 0.0);
            r0_loc = fmax(r0_loc, skip ? 0.0 : fabs(a));
        }
    }
    if (!last) {
        for (int i = g; i < NX; i += SB_G) {
            double ax = 0.0;
            if (k == 0) ax = i < NQ ? sZ0[i] + dt * sZ0[NQ + i] : sZ0[i];
            r0_loc = fmax(r0_loc, fabs(ax + defect(i)));
        }
    }
    const double R0 = grp_max(r0_loc), musum = grp_sum(mu_acc), cntsum = grp_sum((double)cnt), infs = grp_max(inf0);
    if (g < 4 && valid) w[Ly.oPART + g] = g == 0 ? R0 : (g == 1 ? musum : (g == 2 ? cntsum : infs));

    // ---- W: the Jacobian part of the image, straight from the rows' owners: Tt[c][r] | Gt[ix][r] | gn[i] -----------------------------
    if (valid) {
        double* img = w + Ly.oIMG;
        if (g < NQP) {
#pragma unroll
            for (int c = 0; c < NZ; c++) stnt_su(rowT[c], img + Ly.iTT + c * NQP + g);      // (lanes NQ.. hold zeros: the pad rows)
        }
#pragma unroll
        for (int sl = 0; sl < NSLOT; sl++) {
            const int r = g + SB_G * sl;
            if (r < MRP) {
#pragma unroll
                for (int j = 0; j < NQ; j++) stnt_su(grow[sl][j], img + Ly.iGT + j * MRP + r);   // (zeros in the pad row)
            }
        }
        if (g == NNL)
            for (int i = 0; i < NX; i++) stnt_su(nn_on ? nnk[1 + i] : 0.0, img + Ly.iGN + i);
    }
    lds_fence();     // (the next node of this group reuses the block)
}

// Standalone launch: one wavefront = 8 nodes.  Node index t over (instance, stage); instances the policy layer masks out are skipped.
template <int NQ, int MRT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SB_WAVES, SB_WAVES))) void k_stage_build(const smpc_problem_desc* __restrict__ D, int B, int N, const double* __restrict__ x0,
                                                    const double* __restrict__ xg, const double* __restrict__ ug, const double* __restrict__ pp,
                                                    const double* __restrict__ lo_st, const double* __restrict__ hi_st,
                                                    const double* __restrict__ zl_st, const double* __restrict__ nn, double* __restrict__ ws_all,
                                                    long bnd_stride, const uint8_t* __restrict__ active, int32_t* __restrict__ zero_cnt) {
    constexpr int NX = 2 * NQ, NU = NQ;
    constexpr int MR_MAX = MRT >= 0 ? MRT : SMPC_MAX_ROWS;
    using LD = SbLds<NQ, MR_MAX>;
    // (the length of the network pass's compacted row list: the pass ran before this kernel on the same stream, so the counter can be
    //  handed back at zero for the next solve's k_nn_compact -- one memset launch fewer in the chain)
    if (zero_cnt && blockIdx.x == 0 && threadIdx.x == 0) *zero_cnt = 0;
    constexpr int NODES = 64 / SB_G;
    __shared__ __attribute__((aligned(16))) double smem[NODES * LD::SIZE];
    const int g = threadIdx.x & (SB_G - 1), grp = threadIdx.x / SB_G;
    const long total = (long)B * (N + 1);
    long t = (long)blockIdx.x * NODES + grp;
    bool valid = t < total;
    if (!valid) t = total - 1;
    const long b = t / (N + 1);
    const int k = (int)(t - b * (N + 1));
    if (active && !active[b]) valid = false;
    const QpLayout<NQ> Ly(MRT >= 0 ? MRT : D->n_rows);
    double* w = ws_all + (size_t)b * Ly.per_instance(N) + (size_t)k * Ly.stride;
    const double* xk = xg + ((size_t)b * (N + 1) + k) * NX;
    const double* uk = ug + ((size_t)b * N + (k < N ? k : N - 1)) * NU;
    const double* pk = pp + ((size_t)b * (N + 1) + k) * SMPC_NP;
    const size_t bo = (size_t)b * bnd_stride + (size_t)k * NX;
    stage_build<NQ, MRT>(D, Ly, N, k, valid, g, smem + grp * LD::SIZE, x0 + (size_t)b * NX, xk, uk, pk, lo_st + bo, hi_st + bo, zl_st,
                         nn ? nn + (D->nn_mode == SMPC_NN_TERMINAL ? (size_t)b : (size_t)b * (N + 1) + k) * (1 + NX) : nullptr, w);
}

}  // namespace smpc
