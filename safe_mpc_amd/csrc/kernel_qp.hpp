// kernel_qp.hpp -- HOT LOOP B: the stage-structured QP of one RTI step, solved by a Mehrotra primal-dual interior-point
// method whose Newton systems are factorised by a Riccati recursion (the role HPIPM plays for acados; N4/N5 in SURVEY
// section 2; options of reference controller.py:97-110, config.yaml:15-21).
//
// Mapping: ONE WAVEFRONT PER OCP INSTANCE (block = 64 threads).  The wave walks the horizon; the current stage's
// constraint Jacobians / IPM state / Riccati factors are staged in LDS (~11 KB per wave), lanes are spread over matrix
// elements (Hessian assembly, Riccati products) or over constraint rows (slack / multiplier updates, ratio tests).  The
// per-stage records live in a per-instance workspace in HBM, laid out so that every load of a record by the wave is one
// contiguous span; the double integrator's A, B are never stored (env_model.py:63-67: A = [[I, dt I],[0, I]],
// B = [[dt^2/2 I],[dt I]]) -- every product with them is expanded in closed form.
//
// The algorithm is the one restated in oracle/smpc_oracle.cpp::qp_ipm (same initial point, same Mehrotra rule, same
// step rule, same exit test) so that the two agree to rounding; the implementation shares nothing with it.
#pragma once
#include "device_model.hpp"

namespace smpc {

constexpr double QP_THR = 1e-1;
constexpr double QP_FTB = 0.995;
constexpr double QP_ALPHA_MIN = 1e-12;
constexpr double QP_ABSENT = 1e300;  // sentinel for a missing bound side inside the workspace

__host__ __device__ inline int qp_align2(int n) { return (n + 1) & ~1; }

// workspace layout of one stage, in doubles
template <int NQ> struct QpLayout {
    static constexpr int NX = 2 * NQ, NU = NQ, NZ = 3 * NQ;
    int MR, NRT;
    int oT, oGC, oGN, oLO, oHI, oHQQ, oGZ, oB, oSC, nStatic;  // static part
    int oTL, oTU, oLL, oLU, oSL, nState;                          // IPM state (offsets relative to state start)
    int oL, oW, oWV, oPB, nFactor;                                // Riccati factors (relative to factor start)
    int oZ, oZA, oZN, nIter;                                      // iterates (relative)
    int sState, sFactor, sIter, stride;                           // starts of the blocks inside a stage, stage stride
    __host__ __device__ explicit QpLayout(int n_rows) {
        MR = n_rows;
        NRT = NX + NQ + MR + 1;
        int o = 0;
        oT = o; o += NQ * NZ;
        oGC = o; o += MR * NQ;
        oGN = o; o += NX;
        oLO = o; o += NRT;
        oHI = o; o += NRT;
        oHQQ = o; o += NQ * NQ;
        oGZ = o; o += NZ;
        oB = o; o += NX;
        oSC = o; o += 4;
        nStatic = qp_align2(o);
        o = 0;
        oTL = o; o += NRT;
        oTU = o; o += NRT;
        oLL = o; o += NRT;
        oLU = o; o += NRT;
        oSL = o; o += 2;
        nState = qp_align2(o);
        o = 0;
        oL = o; o += NQ * NQ;
        oW = o; o += NQ * NX;
        oWV = o; o += NQ;
        oPB = o; o += NX;
        nFactor = qp_align2(o);
        o = 0;
        oZ = o; o += NZ;
        oZA = o; o += NZ;
        oZN = o; o += NZ;
        nIter = qp_align2(o);
        sState = nStatic;
        sFactor = sState + nState;
        sIter = sFactor + nFactor;
        stride = sIter + nIter;
    }
    __host__ __device__ size_t per_instance(int N) const { return (size_t)stride * (N + 1); }
};

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ void copy_in(double* dst, const double* __restrict__ src, int n, int lane) {
    for (int e = lane; e < n; e += 64) dst[e] = src[e];
}
__device__ __forceinline__ void copy_out(double* __restrict__ dst, const double* src, int n, int lane) {
    for (int e = lane; e < n; e += 64) dst[e] = src[e];
}

template <int NQ>
__global__ __launch_bounds__(64) void k_qp_ipm(const smpc_problem_desc* __restrict__ D, int B, int N,
                                               const double* __restrict__ x0, const double* __restrict__ xg,
                                               const double* __restrict__ ug, const double* __restrict__ pp,
                                               const double* __restrict__ lo_st, const double* __restrict__ hi_st,
                                               const smpc_node_eval* __restrict__ ev, double* __restrict__ ws_all,
                                               double* __restrict__ x_out, double* __restrict__ u_out,
                                               int32_t* __restrict__ status, int32_t* __restrict__ qp_iter) {
    constexpr int NX = 2 * NQ, NU = NQ, NZ = 3 * NQ;
    constexpr int MAXR = NX + NQ + SMPC_MAX_ROWS + 1;
    const int b = blockIdx.x;
    if (b >= B) return;
    const int lane = threadIdx.x;
    const QpLayout<NQ> Ly(D->n_rows);
    const int MR = Ly.MR, NRT = Ly.NRT;
    const int rT0 = NX, rC0 = NX + NQ, rNN = NX + NQ + MR;  // first torque row, first collision row, the nn row
    double* ws = ws_all + (size_t)b * Ly.per_instance(N);
    const double dt = D->dt, cB = 0.5 * dt * dt;

    // ---- LDS ---------------------------------------------------------------------------------------------------------
    __shared__ double sT[NQ * NZ], sGC[SMPC_MAX_ROWS * NQ], sGN[NX], sLO[MAXR], sHI[MAXR], sHQQ[NQ * NQ], sGZ[NZ],
        sB[NX], sSC[4];
    __shared__ double sTL[MAXR], sTU[MAXR], sLL[MAXR], sLU[MAXR], sSL[2];
    __shared__ double sD[MAXR], sE[MAXR], sTD[NQ * NZ];
    __shared__ double sH[NZ * NZ], sGH[NZ];
    __shared__ double sP[NX * NX], sPV[NX], sPT[NX], sPB[NX];
    __shared__ double sLam[NQ * NQ], sG[NQ * (NX + 1)], sL[NQ * NQ], sW[NQ * (NX + 1)];
    __shared__ double sZ[NZ], sZA[NZ], sZN[NZ], sX[NX], sXN[NX], sRHS[NQ];
    __shared__ unsigned char triI[NZ * (NZ + 1) / 2], triJ[NZ * (NZ + 1) / 2];
    __shared__ int sFlag;

    for (int e = lane; e < NZ * (NZ + 1) / 2; e += 64) {
        int i = 0, rem = e;
        while (rem >= NZ - i) { rem -= NZ - i; i++; }
        triI[e] = (unsigned char)i;
        triJ[e] = (unsigned char)(i + rem);
    }
    if (lane == 0) sFlag = 0;

    const double* xb = xg + (size_t)b * (N + 1) * NX;
    const double* ub = ug + (size_t)b * N * NU;
    const double* pb = pp + (size_t)b * (N + 1) * SMPC_NP;
    const smpc_node_eval* evb = ev + (size_t)b * (N + 1);

    if (lane < NX) sX[lane] = x0[(size_t)b * NX + lane] - xb[lane];  // dx0 (kept in sX until the loop starts)
    __syncthreads();
    double dx0_reg = lane < NX ? sX[lane] : 0.0;

    // row product  c_r . z   for the row owned by this lane (z in LDS, layout [u; q; v])
    auto row_dot = [&](int r, const double* z) -> double {
        if (r < rT0) return z[NU + r];
        if (r < rC0) {
            double a = 0.0;
            const double* t = &sT[(r - rT0) * NZ];
#pragma unroll
            for (int c = 0; c < NZ; c++) a = fma(t[c], z[c], a);
            return a;
        }
        if (r < rNN) {
            double a = 0.0;
            const double* g = &sGC[(r - rC0) * NQ];
#pragma unroll
            for (int c = 0; c < NQ; c++) a = fma(g[c], z[NU + c], a);
            return a;
        }
        double a = 0.0;
#pragma unroll
        for (int c = 0; c < NX; c++) a = fma(sGN[c], z[NU + c], a);
        return a;
    };
    // out[i] = sum_r C[r][i] e[r]   for the variable i owned by this lane (i < NZ)
    auto ct_dot = [&](int i, const double* e) -> double {
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < NQ; r++) a = fma(sT[r * NZ + i], e[rT0 + r], a);
        if (i >= NU) {
            const int ix = i - NU;
            a += e[ix];
            a = fma(sGN[ix], e[rNN], a);
            if (ix < NQ)
                for (int r = 0; r < MR; r++) a = fma(sGC[r * NQ + ix], e[rC0 + r], a);
        }
        return a;
    };

    // =====================================================================================================================
    // phase 0: assemble the stage records, initial point, initial residual norm
    // =====================================================================================================================
    double R0 = 0.0;
    int m_comp = 0;
    for (int k = 0; k <= N; k++) {
        const smpc_node_eval& e = evb[k];
        const double* xk = xb + (size_t)k * NX;
        const double* pk = pb + (size_t)k * SMPC_NP;
        const bool last = (k == N);
        const double cs = last ? D->cost_scale_term : D->cost_scale_stage;
        const double lm = last ? D->lm_term : D->lm_stage;
        const bool reach = D->cost_kind == SMPC_COST_REACH;
        bool nn_on = false;
        if (D->nn_mode != SMPC_NN_NONE && k >= 1 && (D->nn_mode == SMPC_NN_ALL || last)) nn_on = pk[4] > 0.0;

        for (int el = lane; el < NQ * NZ; el += 64) {
            const int r = el / NZ, c = el % NZ;
            double v = 0.0;
            if (!last) v = c < NQ ? e.M[r * NQ + c] : (c < 2 * NQ ? e.dtau_dq[r * NQ + c - NQ] : e.dtau_dv[r * NQ + c - 2 * NQ]);
            sT[el] = v;
        }
        for (int el = lane; el < MR * NQ; el += 64) sGC[el] = e.row_grad[el];
        if (lane < NX) sGN[lane] = nn_on ? e.nn_grad[lane] : 0.0;
        for (int el = lane; el < NQ * NQ; el += 64) {
            const int i = el / NQ, j = el % NQ;
            sHQQ[el] = (reach ? cs * e.cost_hess_qq[el] : 0.0) + (i == j ? lm : 0.0);
        }
        if (lane < NZ) {
            double g = 0.0;
            if (reach) {
                if (lane < NU) g = last ? 0.0 : cs * 2.0 * D->R * ub[(size_t)k * NU + lane];
                else if (lane < NU + NQ) g = cs * e.cost_grad_q[lane - NU];
            }
            sGZ[lane] = g;
        }
        if (lane < NX) {
            double bb = 0.0;
            if (!last) {
                const double* xn = xk + NX;
                const int i = lane < NQ ? lane : lane - NQ;
                const double uk = ub[(size_t)k * NU + i];
                bb = lane < NQ ? xk[i] + dt * xk[NQ + i] + cB * uk - xn[i] : xk[NQ + i] + dt * uk - xn[NQ + i];
            }
            sB[lane] = bb;
        }
        if (lane == 0) {
            sSC[0] = (reach && !last ? cs * 2.0 * D->R : 0.0) + lm;  // Huu diagonal
            sSC[1] = lm;                                              // LM on the velocity diagonal
            sSC[2] = nn_on ? (last ? D->nn_soft_e : D->nn_soft_run) : -1.0;
            sSC[3] = 0.0;
        }
        if (lane < NRT) {
            const int r = lane;
            double lo = -QP_ABSENT, hi = QP_ABSENT;
            if (r < rT0) {
                const double l = lo_st[(size_t)k * NX + r], h = hi_st[(size_t)k * NX + r];
                if (k >= 1 && fabs(l) < SMPC_INF) lo = l - xk[r];
                if (k >= 1 && fabs(h) < SMPC_INF) hi = h - xk[r];
            } else if (r < rC0) {
                const double tm = D->joints[r - rT0].tau_max;
                if (!last && tm < SMPC_INF) { lo = -tm - e.tau[r - rT0]; hi = tm - e.tau[r - rT0]; }
            } else if (r < rNN) {
                const smpc_row& row = D->rows[r - rC0];
                if (k >= 1 && fabs(row.lb) < SMPC_INF) lo = row.lb - e.row_val[r - rC0];
                if (k >= 1 && fabs(row.ub) < SMPC_INF) hi = row.ub - e.row_val[r - rC0];
            } else {
                if (nn_on) lo = 0.0 - e.nn_val;
            }
            sLO[r] = lo;
            sHI[r] = hi;
        }
        // z = 0, except the fixed dx_0
        if (lane < NZ) sZ[lane] = (k == 0 && lane >= NU) ? sX[lane - NU] : 0.0;
        __syncthreads();
        // initial slacks / multipliers
        double r0_loc = 0.0;
        int cnt = 0;
        if (lane < NRT) {
            const int r = lane;
            const double cz = row_dot(r, sZ);
            const bool soft = (r == rNN) && sSC[2] >= 0.0;
            double tl = 1.0, ll = 0.0, tu = 1.0, lu = 0.0;
            if (sLO[r] > -QP_ABSENT) {
                const double s0 = soft ? QP_THR : 0.0;
                const double slack = cz + s0 - sLO[r];
                tl = fmax(slack, QP_THR);
                ll = D->qp_mu0 / tl;
                if (soft) ll = fmin(ll, 0.5 * sSC[2]);
                r0_loc = fmax(r0_loc, fabs(slack - tl));
                cnt += soft ? 2 : 1;
                if (soft) sSL[0] = s0;
            }
            if (sHI[r] < QP_ABSENT) {
                const double slack = sHI[r] - cz;
                tu = fmax(slack, QP_THR);
                lu = D->qp_mu0 / tu;
                r0_loc = fmax(r0_loc, fabs(slack - tu));
                cnt += 1;
            }
            sTL[r] = tl; sLL[r] = ll; sTU[r] = tu; sLU[r] = lu;
            sE[r] = -(ll - lu);
        }
        if (lane == 0 && !(sSC[2] >= 0.0)) sSL[0] = 0.0;
        if (lane == 1) sSL[1] = 0.0;
        __syncthreads();
        // stationarity residual at the initial point (pi = 0): g + H z - C^T (ll - lu); the only non-zero z is dx_0, which
        // enters the u-rows of stage 0 through nothing (no cost cross term)
        if (lane < NZ && !(k == 0 && lane >= NU) && !(last && lane < NU)) {
            r0_loc = fmax(r0_loc, fabs(sGZ[lane] + ct_dot(lane, sE)));
        }
        if (!last && lane < NX) {
            // dynamics defect of the initial point: next dx is 0
            double ax = 0.0;
            if (k == 0) ax = lane < NQ ? sX[lane] + dt * sX[NQ + lane] : sX[lane];
            r0_loc = fmax(r0_loc, fabs(ax + sB[lane]));
        }
        R0 = fmax(R0, wave_max(r0_loc));
        m_comp += (int)wave_sum((double)cnt);
        // write the record
        double* w = ws + (size_t)k * Ly.stride;
        copy_out(w + Ly.oT, sT, NQ * NZ, lane);
        copy_out(w + Ly.oGC, sGC, MR * NQ, lane);
        copy_out(w + Ly.oGN, sGN, NX, lane);
        copy_out(w + Ly.oLO, sLO, NRT, lane);
        copy_out(w + Ly.oHI, sHI, NRT, lane);
        copy_out(w + Ly.oHQQ, sHQQ, NQ * NQ, lane);
        copy_out(w + Ly.oGZ, sGZ, NZ, lane);
        copy_out(w + Ly.oB, sB, NX, lane);
        copy_out(w + Ly.oSC, sSC, 4, lane);
        double* st = w + Ly.sState;
        copy_out(st + Ly.oTL, sTL, NRT, lane);
        copy_out(st + Ly.oTU, sTU, NRT, lane);
        copy_out(st + Ly.oLL, sLL, NRT, lane);
        copy_out(st + Ly.oLU, sLU, NRT, lane);
        copy_out(st + Ly.oSL, sSL, 2, lane);
        copy_out(w + Ly.sIter + Ly.oZ, sZ, NZ, lane);
        __syncthreads();
    }
    if (m_comp == 0) m_comp = 1;
    const double inv_m = 1.0 / (double)m_comp;

    auto load_static = [&](int k) {
        const double* w = ws + (size_t)k * Ly.stride;
        copy_in(sT, w + Ly.oT, NQ * NZ, lane);
        copy_in(sGC, w + Ly.oGC, MR * NQ, lane);
        copy_in(sGN, w + Ly.oGN, NX, lane);
        copy_in(sLO, w + Ly.oLO, NRT, lane);
        copy_in(sHI, w + Ly.oHI, NRT, lane);
        copy_in(sHQQ, w + Ly.oHQQ, NQ * NQ, lane);
        copy_in(sGZ, w + Ly.oGZ, NZ, lane);
        copy_in(sB, w + Ly.oB, NX, lane);
        copy_in(sSC, w + Ly.oSC, 4, lane);
    };
    auto load_state = [&](int k) {
        const double* st = ws + (size_t)k * Ly.stride + Ly.sState;
        copy_in(sTL, st + Ly.oTL, NRT, lane);
        copy_in(sTU, st + Ly.oTU, NRT, lane);
        copy_in(sLL, st + Ly.oLL, NRT, lane);
        copy_in(sLU, st + Ly.oLU, NRT, lane);
        copy_in(sSL, st + Ly.oSL, 2, lane);
    };
    auto store_state = [&](int k) {
        double* st = ws + (size_t)k * Ly.stride + Ly.sState;
        copy_out(st + Ly.oTL, sTL, NRT, lane);
        copy_out(st + Ly.oTU, sTU, NRT, lane);
        copy_out(st + Ly.oLL, sLL, NRT, lane);
        copy_out(st + Ly.oLU, sLU, NRT, lane);
        copy_out(st + Ly.oSL, sSL, 2, lane);
    };
    auto load_factor = [&](int k) {
        const double* f = ws + (size_t)k * Ly.stride + Ly.sFactor;
        copy_in(sL, f + Ly.oL, NQ * NQ, lane);
        for (int e = lane; e < NQ * NX; e += 64) sW[(e / NX) * (NX + 1) + e % NX] = f[Ly.oW + e];
        if (lane < NQ) sW[lane * (NX + 1) + NX] = f[Ly.oWV + lane];
        copy_in(sPB, f + Ly.oPB, NX, lane);
    };

    // slack / multiplier directions of the row owned by this lane for a trial point zt (LDS), given sigma*mu and the affine
    // trial point za (corrector term) -- returns through references; absent sides give zeros
    struct Dir { double dtl, dll, dtu, dlu, dsl; };
    auto row_dir = [&](int r, const double* zt, double sigmu, bool corr, const double* za) -> Dir {
        Dir o{0.0, 0.0, 0.0, 0.0, 0.0};
        const double czn = row_dot(r, zt);
        double cza = 0.0;
        if (corr) cza = row_dot(r, za);
        const bool soft = (r == rNN) && sSC[2] >= 0.0;
        if (sLO[r] > -QP_ABSENT) {
            const double tl = sTL[r], ll = sLL[r];
            if (soft) {
                const double wgt = sSC[2], sl = sSL[0], nu = wgt - ll;
                const double deff = 1.0 / (tl / ll + sl / nu);
                double ct = 0.0, cs2 = 0.0;
                if (corr) {
                    // affine directions of this row
                    const double dla = -deff * (cza - sLO[r] + 0.0 / nu - 0.0 / ll);
                    const double dta = (0.0 - tl * dla) / ll - tl;
                    const double dsa = (0.0 + sl * dla) / nu - sl;
                    ct = dta * dla;
                    cs2 = -dsa * dla;
                }
                const double dl = -deff * (czn - sLO[r] + (sigmu - cs2) / nu - (sigmu - ct) / ll);
                o.dll = dl;
                o.dtl = (sigmu - ct - tl * dl) / ll - tl;
                o.dsl = (sigmu - cs2 + sl * dl) / nu - sl;
            } else {
                double ct = 0.0;
                if (corr) {
                    const double dta = cza - sLO[r] - tl;
                    const double dla = (0.0 - ll * dta) / tl - ll;
                    ct = dta * dla;
                }
                const double dt_ = czn - sLO[r] - tl;
                o.dtl = dt_;
                o.dll = (sigmu - ct - ll * dt_) / tl - ll;
            }
        }
        if (sHI[r] < QP_ABSENT) {
            const double tu = sTU[r], lu = sLU[r];
            double ct = 0.0;
            if (corr) {
                const double dta = sHI[r] - cza - tu;
                const double dla = (0.0 - lu * dta) / tu - lu;
                ct = dta * dla;
            }
            const double dt_ = sHI[r] - czn - tu;
            o.dtu = dt_;
            o.dlu = (sigmu - ct - lu * dt_) / tu - lu;
        }
        return o;
    };
    // gradient coefficient e_r (and barrier weight D_r) of the row owned by this lane
    auto row_coeff = [&](int r, double sigmu, bool corr, const double* za, double* Dr) -> double {
        double e = 0.0, dsum = 0.0;
        double cza = 0.0;
        if (corr) cza = row_dot(r, za);
        const bool soft = (r == rNN) && sSC[2] >= 0.0;
        if (sLO[r] > -QP_ABSENT) {
            const double tl = sTL[r], ll = sLL[r];
            if (soft) {
                const double wgt = sSC[2], sl = sSL[0], nu = wgt - ll;
                const double deff = 1.0 / (tl / ll + sl / nu);
                double ct = 0.0, cs2 = 0.0;
                if (corr) {
                    const double dla = -deff * (cza - sLO[r]);
                    const double dta = (0.0 - tl * dla) / ll - tl;
                    const double dsa = (0.0 + sl * dla) / nu - sl;
                    ct = dta * dla;
                    cs2 = -dsa * dla;
                }
                e += -ll + deff * (-sLO[r] + (sigmu - cs2) / nu - (sigmu - ct) / ll);
                dsum += deff;
            } else {
                double ct = 0.0;
                if (corr) {
                    const double dta = cza - sLO[r] - tl;
                    const double dla = (0.0 - ll * dta) / tl - ll;
                    ct = dta * dla;
                }
                const double d = ll / tl;
                e += -ll - d * sLO[r] - (sigmu - ct) / tl;
                dsum += d;
            }
        }
        if (sHI[r] < QP_ABSENT) {
            const double tu = sTU[r], lu = sLU[r];
            double ct = 0.0;
            if (corr) {
                const double dta = sHI[r] - cza - tu;
                const double dla = (0.0 - lu * dta) / tu - lu;
                ct = dta * dla;
            }
            const double d = lu / tu;
            e += lu - d * sHI[r] + (sigmu - ct) / tu;
            dsum += d;
        }
        *Dr = dsum;
        return e;
    };
    // ratio test contribution of one row
    auto row_ratio = [&](int r, const Dir& d) -> double {
        double a = 1e300;
        const bool soft = (r == rNN) && sSC[2] >= 0.0;
        if (sLO[r] > -QP_ABSENT) {
            if (d.dtl < 0.0) a = fmin(a, -sTL[r] / d.dtl);
            if (d.dll < 0.0) a = fmin(a, -sLL[r] / d.dll);
            if (soft) {
                if (d.dsl < 0.0) a = fmin(a, -sSL[0] / d.dsl);
                if (-d.dll < 0.0) a = fmin(a, -(sSC[2] - sLL[r]) / (-d.dll));
            }
        }
        if (sHI[r] < QP_ABSENT) {
            if (d.dtu < 0.0) a = fmin(a, -sTU[r] / d.dtu);
            if (d.dlu < 0.0) a = fmin(a, -sLU[r] / d.dlu);
        }
        return a;
    };

    // Cholesky factor of sLam in registers (every lane redundantly); returns false on a non-positive pivot
    double Lr[NQ][NQ], Linv[NQ];
    auto chol_from_lds = [&](const double* A) -> bool {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            double dsum = A[j * NQ + j];
#pragma unroll
            for (int t = 0; t < j; t++) dsum = fma(-Lr[j][t], Lr[j][t], dsum);
            ok = ok && (dsum > 0.0);
            const double lj = sqrt(dsum);
            const double inv = 1.0 / lj;
            Lr[j][j] = lj;
            Linv[j] = inv;
#pragma unroll
            for (int i = j + 1; i < NQ; i++) {
                double v = A[i * NQ + j];
#pragma unroll
                for (int t = 0; t < j; t++) v = fma(-Lr[i][t], Lr[j][t], v);
                Lr[i][j] = v * inv;
            }
        }
        return ok;
    };
    auto load_L_regs = [&]() {
#pragma unroll
        for (int i = 0; i < NQ; i++) {
#pragma unroll
            for (int j = 0; j <= i; j++) Lr[i][j] = sL[i * NQ + j];
            Linv[i] = 1.0 / Lr[i][i];
        }
    };

    // backward vector step at stage k given sGH (gradient), sPV (p_{k+1}), sPB, sW, L in registers; updates sPV and the
    // w column of sW.  Must be called by all lanes.
    auto vector_back = [&](int k) {
        if (lane < NX) sPT[lane] = sPV[lane] + sPB[lane];
        __syncthreads();
        if (lane < NQ) sG[lane * (NX + 1) + NX] = sGH[lane] + cB * sPT[lane] + dt * sPT[NQ + lane];  // rho
        __syncthreads();
        {
            double wv[NQ];
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                double v = sG[i * (NX + 1) + NX];
#pragma unroll
                for (int t = 0; t < i; t++) v = fma(-Lr[i][t], wv[t], v);
                wv[i] = v * Linv[i];
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < NQ; i++) sW[i * (NX + 1) + NX] = wv[i];
            }
            __syncthreads();
            if (k > 0 && lane < NX) {
                // p_k = gh_x + A^T pt - W^T w
                const int i = lane;
                double v = sGH[NU + i] + (i < NQ ? sPT[i] : dt * sPT[i - NQ] + sPT[i]);
#pragma unroll
                for (int t = 0; t < NQ; t++) v = fma(-sW[t * (NX + 1) + i], wv[t], v);
                sXN[i] = v;
            }
        }
        __syncthreads();
        if (k > 0 && lane < NX) sPV[lane] = sXN[lane];
        __syncthreads();
    };
    // forward step: u = -L^-T (W x + w), with x in sX; writes the trial point [u; x] into zt and x_{k+1} into sX
    auto forward_step = [&](int k, double* zt) {
        if (lane < NQ) {
            double v = sW[lane * (NX + 1) + NX];
#pragma unroll
            for (int j = 0; j < NX; j++) v = fma(sW[lane * (NX + 1) + j], sX[j], v);
            sRHS[lane] = -v;
        }
        __syncthreads();
        double uk[NQ];
#pragma unroll
        for (int i = NQ - 1; i >= 0; i--) {
            double v = sRHS[i];
#pragma unroll
            for (int t = i + 1; t < NQ; t++) v = fma(-Lr[t][i], uk[t], v);
            uk[i] = v * Linv[i];
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NQ; i++) zt[i] = uk[i];
        }
        if (lane < NX) zt[NU + lane] = sX[lane];
        if (lane < NX) {
            const int i = lane < NQ ? lane : lane - NQ;
            double ui = 0.0;
#pragma unroll
            for (int t = 0; t < NQ; t++) ui = (t == i) ? uk[t] : ui;
            sXN[lane] = lane < NQ ? sX[i] + dt * sX[NQ + i] + cB * ui + sB[lane] : sX[NQ + i] + dt * ui + sB[lane];
        }
        __syncthreads();
    };

    // =====================================================================================================================
    // main loop
    // =====================================================================================================================
    // mu at the initial point
    double mu;
    {
        double acc = 0.0;
        for (int k = 0; k <= N; k++) {
            load_static(k);
            load_state(k);
            __syncthreads();
            if (lane < NRT) {
                const int r = lane;
                if (sLO[r] > -QP_ABSENT) {
                    acc += sLL[r] * sTL[r];
                    if (r == rNN && sSC[2] >= 0.0) acc += (sSC[2] - sLL[r]) * sSL[0];
                }
                if (sHI[r] < QP_ABSENT) acc += sLU[r] * sTU[r];
            }
            __syncthreads();
        }
        mu = wave_sum(acc) * inv_m;
    }
    double rho_lin = 1.0;
    int it = 0, st_code = 2;
    const double tol = D->qp_tol;
    const int max_iter = D->qp_max_iter;

    for (it = 0; it < max_iter; it++) {
        if (mu <= tol && rho_lin * R0 <= tol) { st_code = 0; break; }

        // ---------------- pass F: factorise H + C^T D C backwards and run the predictor's vector recursion --------------
        for (int k = N; k >= 0; k--) {
            const bool last = (k == N);
            load_static(k);
            load_state(k);
            __syncthreads();
            if (lane < NRT) {
                double Dr;
                sE[lane] = row_coeff(lane, 0.0, false, nullptr, &Dr);
                sD[lane] = Dr;
            }
            __syncthreads();
            for (int el = lane; el < NQ * NZ; el += 64) sTD[el] = sT[el] * sD[rT0 + el / NZ];
            __syncthreads();
            // Hessian (upper triangle, mirrored) and gradient
            for (int el = lane; el < NZ * (NZ + 1) / 2; el += 64) {
                const int i = triI[el], j = triJ[el];
                double a = 0.0;
#pragma unroll
                for (int r = 0; r < NQ; r++) a = fma(sTD[r * NZ + i], sT[r * NZ + j], a);
                if (i >= NU) {  // both in the x block (j >= i)
                    const int ix = i - NU, jx = j - NU;
                    a = fma(sGN[ix] * sD[rNN], sGN[jx], a);
                    if (jx < NQ) {
                        a += sHQQ[ix * NQ + jx];
                        for (int r = 0; r < MR; r++) a = fma(sGC[r * NQ + ix] * sD[rC0 + r], sGC[r * NQ + jx], a);
                    }
                    if (ix == jx) {
                        a += sD[ix];
                        if (ix >= NQ) a += sSC[1];
                    }
                } else if (i == j) {
                    a += sSC[0];
                }
                sH[i * NZ + j] = a;
                sH[j * NZ + i] = a;
            }
            if (lane < NZ) sGH[lane] = sGZ[lane] + ct_dot(lane, sE);
            __syncthreads();
            if (last) {
                for (int el = lane; el < NX * NX; el += 64) sP[el] = sH[(NU + el / NX) * NZ + NU + el % NX];
                if (lane < NX) sPV[lane] = sGH[NU + lane];
                __syncthreads();
                continue;
            }
            // P b, Lambda, G
            if (lane < NX) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < NX; j++) a = fma(sP[lane * NX + j], sB[j], a);
                sPB[lane] = a;
            }
            for (int el = lane; el < NQ * NQ; el += 64) {
                const int i = el / NQ, j = el % NQ;
                // B^T P B = c^2 P11 + c dt (P12 + P21) + dt^2 P22
                sLam[el] = sH[i * NZ + j] + cB * cB * sP[i * NX + j] +
                           cB * dt * (sP[i * NX + NQ + j] + sP[(NQ + i) * NX + j]) + dt * dt * sP[(NQ + i) * NX + NQ + j];
            }
            for (int el = lane; el < NQ * NX; el += 64) {
                const int i = el / NX, j = el % NX;
                // B^T P A: left block c P11 + dt P21 ; right block dt (c P11 + dt P21) + c P12 + dt P22
                double v;
                if (j < NQ) v = cB * sP[i * NX + j] + dt * sP[(NQ + i) * NX + j];
                else {
                    const int jj = j - NQ;
                    v = dt * (cB * sP[i * NX + jj] + dt * sP[(NQ + i) * NX + jj]) + cB * sP[i * NX + NQ + jj] +
                        dt * sP[(NQ + i) * NX + NQ + jj];
                }
                sG[i * (NX + 1) + j] = sH[i * NZ + NU + j] + v;
            }
            __syncthreads();
            if (!chol_from_lds(sLam)) {
                if (lane == 0) sFlag = 1;
            }
            // W = L^-1 G (one column per lane)
            if (lane < NX) {
                double col[NQ];
#pragma unroll
                for (int i = 0; i < NQ; i++) {
                    double v = sG[i * (NX + 1) + lane];
#pragma unroll
                    for (int t = 0; t < i; t++) v = fma(-Lr[i][t], col[t], v);
                    col[i] = v * Linv[i];
                    sW[i * (NX + 1) + lane] = col[i];
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < NQ; i++)
#pragma unroll
                    for (int j = 0; j < NQ; j++) sL[i * NQ + j] = j <= i ? Lr[i][j] : 0.0;
            }
            __syncthreads();
            // vector recursion with the predictor gradient (uses sPV = p_{k+1}); then the matrix recursion
            vector_back(k);
            if (k > 0) {
                // P_k = Hxx + A^T P A - W^T W, upper triangle then mirror
                for (int el = lane; el < NX * (NX + 1) / 2; el += 64) {
                    int i = 0, rem = el;
                    while (rem >= NX - i) { rem -= NX - i; i++; }
                    const int j = i + rem;
                    double a = sH[(NU + i) * NZ + NU + j];
                    // A^T P A, blocks: [P11, dt P11 + P12; dt P11 + P21, dt^2 P11 + dt (P12 + P21) + P22]
                    if (j < NQ) a += sP[i * NX + j];
                    else if (i < NQ) a += dt * sP[i * NX + (j - NQ)] + sP[i * NX + j];
                    else {
                        const int ii = i - NQ, jj = j - NQ;
                        a += dt * dt * sP[ii * NX + jj] + dt * (sP[ii * NX + j] + sP[i * NX + jj]) + sP[i * NX + j];
                    }
#pragma unroll
                    for (int t = 0; t < NQ; t++) a = fma(-sW[t * (NX + 1) + i], sW[t * (NX + 1) + j], a);
                    sTD[el] = a;  // staged in the (now free) sTD scratch: sH is still being read by other lanes
                }
                __syncthreads();
                for (int el = lane; el < NX * (NX + 1) / 2; el += 64) {
                    int i = 0, rem = el;
                    while (rem >= NX - i) { rem -= NX - i; i++; }
                    const int j = i + rem;
                    const double a = sTD[el];
                    sP[i * NX + j] = a;
                    sP[j * NX + i] = a;
                }
                __syncthreads();
            }
            // store factors
            {
                double* f = ws + (size_t)k * Ly.stride + Ly.sFactor;
                copy_out(f + Ly.oL, sL, NQ * NQ, lane);
                for (int e = lane; e < NQ * NX; e += 64) f[Ly.oW + e] = sW[(e / NX) * (NX + 1) + e % NX];
                if (lane < NQ) f[Ly.oWV + lane] = sW[lane * (NX + 1) + NX];
                copy_out(f + Ly.oPB, sPB, NX, lane);
            }
            __syncthreads();
        }
        if (sFlag) { st_code = 4; break; }

        // ---------------- pass A: predictor forward sweep, affine step length and centring parameter -------------------
        double amin = 1e300, S1 = 0.0, S2 = 0.0;
        if (lane < NX) sX[lane] = dx0_reg;
        __syncthreads();
        for (int k = 0; k <= N; k++) {
            load_static(k);
            load_state(k);
            if (k < N) load_factor(k);
            __syncthreads();
            if (k < N) {
                load_L_regs();
                forward_step(k, sZA);
            } else {
                if (lane < NZ) sZA[lane] = lane >= NU ? sX[lane - NU] : 0.0;
                __syncthreads();
            }
            if (lane < NRT) {
                const int r = lane;
                const Dir d = row_dir(r, sZA, 0.0, false, nullptr);
                amin = fmin(amin, row_ratio(r, d));
                if (sLO[r] > -QP_ABSENT) {
                    S1 += sLL[r] * d.dtl + sTL[r] * d.dll;
                    S2 += d.dll * d.dtl;
                    if (r == rNN && sSC[2] >= 0.0) {
                        S1 += (sSC[2] - sLL[r]) * d.dsl - sSL[0] * d.dll;
                        S2 += -d.dll * d.dsl;
                    }
                }
                if (sHI[r] < QP_ABSENT) {
                    S1 += sLU[r] * d.dtu + sTU[r] * d.dlu;
                    S2 += d.dlu * d.dtu;
                }
            }
            copy_out(ws + (size_t)k * Ly.stride + Ly.sIter + Ly.oZA, sZA, NZ, lane);
            __syncthreads();
            if (k < N && lane < NX) sX[lane] = sXN[lane];
            __syncthreads();
        }
        const double a_aff = fmin(1.0, wave_min(amin));
        S1 = wave_sum(S1);
        S2 = wave_sum(S2);
        const double mu_aff = (mu * (double)m_comp + a_aff * S1 + a_aff * a_aff * S2) * inv_m;
        double sigma = mu_aff / mu;
        sigma = sigma * sigma * sigma;
        const double sigmu = sigma * mu;

        // ---------------- pass C: corrector gradient and backward vector recursion -----------------------------------------
        for (int k = N; k >= 0; k--) {
            const bool last = (k == N);
            load_static(k);
            load_state(k);
            copy_in(sZA, ws + (size_t)k * Ly.stride + Ly.sIter + Ly.oZA, NZ, lane);
            if (!last) load_factor(k);
            __syncthreads();
            if (lane < NRT) {
                double Dr;
                sE[lane] = row_coeff(lane, sigmu, true, sZA, &Dr);
            }
            __syncthreads();
            if (lane < NZ) sGH[lane] = sGZ[lane] + ct_dot(lane, sE);
            __syncthreads();
            if (last) {
                if (lane < NX) sPV[lane] = sGH[NU + lane];
                __syncthreads();
                continue;
            }
            load_L_regs();
            vector_back(k);
            if (lane < NQ) ws[(size_t)k * Ly.stride + Ly.sFactor + Ly.oWV + lane] = sW[lane * (NX + 1) + NX];
            __syncthreads();
        }

        // ---------------- pass D: corrector forward sweep and step length --------------------------------------------------
        amin = 1e300;
        if (lane < NX) sX[lane] = dx0_reg;
        __syncthreads();
        for (int k = 0; k <= N; k++) {
            load_static(k);
            load_state(k);
            copy_in(sZA, ws + (size_t)k * Ly.stride + Ly.sIter + Ly.oZA, NZ, lane);
            if (k < N) load_factor(k);
            __syncthreads();
            if (k < N) {
                load_L_regs();
                forward_step(k, sZN);
            } else {
                if (lane < NZ) sZN[lane] = lane >= NU ? sX[lane - NU] : 0.0;
                __syncthreads();
            }
            if (lane < NRT) {
                const Dir d = row_dir(lane, sZN, sigmu, true, sZA);
                amin = fmin(amin, row_ratio(lane, d));
            }
            copy_out(ws + (size_t)k * Ly.stride + Ly.sIter + Ly.oZN, sZN, NZ, lane);
            __syncthreads();
            if (k < N && lane < NX) sX[lane] = sXN[lane];
            __syncthreads();
        }
        const double alpha = fmin(1.0, QP_FTB * wave_min(amin));
        if (!(alpha == alpha)) { st_code = 4; break; }
        if (alpha < QP_ALPHA_MIN) { st_code = 3; break; }

        // ---------------- pass U: take the step, new complementarity -------------------------------------------------------
        double acc = 0.0;
        for (int k = 0; k <= N; k++) {
            double* w = ws + (size_t)k * Ly.stride;
            load_static(k);
            load_state(k);
            copy_in(sZ, w + Ly.sIter + Ly.oZ, NZ, lane);
            copy_in(sZA, w + Ly.sIter + Ly.oZA, NZ, lane);
            copy_in(sZN, w + Ly.sIter + Ly.oZN, NZ, lane);
            __syncthreads();
            Dir d{0, 0, 0, 0, 0};
            if (lane < NRT) d = row_dir(lane, sZN, sigmu, true, sZA);
            __syncthreads();
            if (lane < NRT) {
                const int r = lane;
                if (sLO[r] > -QP_ABSENT) {
                    sTL[r] += alpha * d.dtl;
                    sLL[r] += alpha * d.dll;
                    if (r == rNN && sSC[2] >= 0.0) sSL[0] += alpha * d.dsl;
                }
                if (sHI[r] < QP_ABSENT) {
                    sTU[r] += alpha * d.dtu;
                    sLU[r] += alpha * d.dlu;
                }
            }
            if (lane < NZ) sZ[lane] += alpha * (sZN[lane] - sZ[lane]);
            __syncthreads();
            if (lane < NRT) {
                const int r = lane;
                if (sLO[r] > -QP_ABSENT) {
                    acc += sLL[r] * sTL[r];
                    if (r == rNN && sSC[2] >= 0.0) acc += (sSC[2] - sLL[r]) * sSL[0];
                }
                if (sHI[r] < QP_ABSENT) acc += sLU[r] * sTU[r];
            }
            store_state(k);
            copy_out(w + Ly.sIter + Ly.oZ, sZ, NZ, lane);
            __syncthreads();
        }
        rho_lin *= (1.0 - alpha);
        mu = wave_sum(acc) * inv_m;
        if (!(mu == mu)) { st_code = 4; break; }
    }
    if (it == max_iter && st_code == 2 && mu <= tol && rho_lin * R0 <= tol) st_code = 0;

    // ---- full step (FIXED_STEP, parser.py:139) and status mapping (see oracle: QP iteration cap is tolerated by RTI) ----
    bool bad = false;
    for (int k = 0; k <= N; k++) {
        const double* zk = ws + (size_t)k * Ly.stride + Ly.sIter + Ly.oZ;
        if (lane < NX) {
            const double v = xb[(size_t)k * NX + lane] + zk[NU + lane];
            x_out[((size_t)b * (N + 1) + k) * NX + lane] = v;
            bad |= !(v == v);
        }
        if (k < N && lane < NU) {
            const double v = ub[(size_t)k * NU + lane] + zk[lane];
            u_out[((size_t)b * N + k) * NU + lane] = v;
            bad |= !(v == v);
        }
    }
    const bool any_bad = wave_max(bad ? 1.0 : 0.0) > 0.0;
    if (lane == 0) {
        int stc = (st_code == 0 || st_code == 2) ? SMPC_STATUS_SUCCESS : SMPC_STATUS_QP_FAILURE;
        if (any_bad && stc == SMPC_STATUS_SUCCESS) stc = SMPC_STATUS_NAN;
        status[b] = stc;
        if (qp_iter) qp_iter[b] = it;
    }
}

}  // namespace smpc
