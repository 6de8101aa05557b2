// kernel_qp.hpp -- HOT LOOP B: the stage-structured QP of one RTI step, solved by a Mehrotra primal-dual interior-point
// method whose Newton systems are factorised by a Riccati recursion (the role HPIPM plays for acados; N4/N5 in SURVEY
// section 2; options of reference controller.py:97-110, config.yaml:15-21).
//
// Mapping: ONE WAVEFRONT PER OCP INSTANCE (block = 64 threads).  The wave walks the horizon four times per IPM iteration
//   B1  backward: apply the previous step, barrier weights, H + C^T D C, Riccati factorisation, predictor costate
//   F1  forward : predictor roll-out, affine step length, centring parameter
//   B2  backward: corrector gradient, costate recursion with the stored factors
//   F2  forward : corrector roll-out, step length
// Each stage is ONE contiguous record in a per-instance HBM workspace; the wave pulls a record with 16-byte loads into
// registers (one load instruction stream, issued a whole stage ahead of its use), commits it to LDS, works on it with
// lanes spread over matrix elements (Hessian, Riccati products) or constraint rows (slacks, multipliers, ratio tests), and
// writes back only the blocks that changed.  Cross-lane hand-offs inside the wave go through LDS with wave-scope fences
// (no s_barrier, no vmcnt drain), so the prefetch stays in flight under the arithmetic.
// The double integrator's A, B are never stored (env_model.py:63-67): every product with them is expanded in closed form.
//
// The algorithm is the one restated in oracle/smpc_oracle.cpp::qp_ipm (same initial point, Mehrotra rule, step rule and
// exit test), so the two agree to rounding; the implementation shares nothing with it.
#pragma once
#include "device_model.hpp"

namespace smpc {

constexpr double QP_THR = 1e-1;
constexpr double QP_FTB = 0.995;
constexpr double QP_ALPHA_MIN = 1e-12;
constexpr double QP_ABSENT = 1e300;  // sentinel for a missing bound side inside the workspace

__host__ __device__ inline int qp_even(int n) { return (n + 1) & ~1; }

// record layout of one stage, in doubles; every block starts on a 16-byte boundary
template <int NQ> struct QpLayout {
    static constexpr int NX = 2 * NQ, NU = NQ, NZ = 3 * NQ, WS = NX + 1;  // WS: row stride of W (last column = w)
    int MR, NRT;
    static constexpr int NZP = (NZ + 1) & ~1;                               // padded slot of one iterate vector
    int oT, oGC, oGN, oLO, oHI, oGZ, oB, oSC;   // C block (static)
    int oTL, oTU, oLL, oLU, oSL;                // state block
    int oZ, oZA, oZN;                           // iterate block
    int oL, oW, oPB;                            // factor block
    int oHQQ;                                   // H block (static, B1 only)
    int bState, bIter, bFac, bH;       // block starts
    int nPrefix, stride;               // doubles needed by F1/B2/F2, and by B1 (= record size)
    __host__ __device__ explicit QpLayout(int n_rows) {
        MR = n_rows;
        NRT = NX + NQ + MR + 1;
        int o = 0;
        oT = o; o += NQ * NZ;
        oGC = o; o += MR * NQ;
        oGN = o; o += NX;
        oLO = o; o += NRT;
        oHI = o; o += NRT;
        oGZ = o; o += NZ;
        oB = o; o += NX;
        oSC = o; o += 4;
        o = qp_even(o);
        bState = o;
        oTL = o; o += NRT;
        oTU = o; o += NRT;
        oLL = o; o += NRT;
        oLU = o; o += NRT;
        oSL = o; o += 2;
        o = qp_even(o);
        bIter = o;
        oZ = o; o += NZP;
        oZA = o; o += NZP;
        oZN = o; o += NZP;
        bFac = o;
        oL = o; o += NQ * NQ;
        oW = o; o += NQ * WS;
        oPB = o; o += NX;
        o = qp_even(o);
        bH = o;
        nPrefix = o;
        oHQQ = o; o += NQ * NQ;
        stride = qp_even(o);
    }
    __host__ __device__ size_t per_instance(int N) const { return (size_t)stride * (N + 1); }
};

typedef double dbl2 __attribute__((ext_vector_type(2)));

// wave-local hand-off through LDS: LDS operations of one wave execute in issue order, so all that is needed is that the
// compiler neither reorders nor caches them across this point
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}

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
// 1/x to (almost) full double precision: hardware seed + two Newton steps
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
// 1/sqrt(x), same recipe
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(fma(-hx * y, y, 0.5), y, y);
    y = fma(fma(-hx * y, y, 0.5), y, y);
    return y;
}

#ifndef QP_WAVES_PER_EU
#define QP_WAVES_PER_EU 3
#endif
constexpr int QP_PF = 5;  // 16-byte prefetch registers per lane: 5 * 64 * 2 = 640 doubles >= any record

template <int NQ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(QP_WAVES_PER_EU, QP_WAVES_PER_EU))) void k_qp_ipm(const smpc_problem_desc* __restrict__ D, int B, int N,
                                               const double* __restrict__ x0, const double* __restrict__ xg,
                                               const double* __restrict__ ug, const double* __restrict__ pp,
                                               const double* __restrict__ lo_st, const double* __restrict__ hi_st,
                                               const smpc_node_eval* __restrict__ ev, double* __restrict__ ws_all,
                                               double* __restrict__ x_out, double* __restrict__ u_out,
                                               int32_t* __restrict__ status, int32_t* __restrict__ qp_iter) {
    constexpr int NX = 2 * NQ, NU = NQ, NZ = 3 * NQ, WS = NX + 1;
    constexpr int MAXR = NX + NQ + SMPC_MAX_ROWS + 1;
    constexpr int NTRI_Z = NZ * (NZ + 1) / 2, NTRI_X = NX * (NX + 1) / 2;
    constexpr int REC_MAX = QP_PF * 128;
    constexpr int EV_D = (int)(sizeof(smpc_node_eval) / sizeof(double));
    const int b = blockIdx.x;
    if (b >= B) return;
    const int lane = threadIdx.x;
    const QpLayout<NQ> Ly(D->n_rows);
    const int MR = Ly.MR, NRT = Ly.NRT;
    const int rT0 = NX, rC0 = NX + NQ, rNN = NX + NQ + MR;  // first torque row, first collision row, the nn row
    double* ws = ws_all + (size_t)b * Ly.per_instance(N);
    const double dt = D->dt, cB = 0.5 * dt * dt;

    // ---- LDS ---------------------------------------------------------------------------------------------------------
    __shared__ __attribute__((aligned(16))) double rec[REC_MAX];   // the current stage record
    constexpr int EV_PAD = EV_D + (EV_D & 1);
    constexpr int SCR_A = NZ * NZ + NQ * NZ + NX * NX;                 // sH | sTD | sP, also the phase-0 staging area
    constexpr int SCR = SCR_A > EV_PAD ? SCR_A : EV_PAD;
    __shared__ __attribute__((aligned(16))) double scr[SCR];
    double* const sH = scr;
    double* const sTD = scr + NZ * NZ;
    double* const sP = scr + NZ * NZ + NQ * NZ;
    double* const sEV = scr;
    __shared__ double sD[MAXR], sE[MAXR], sGH[NZ], sPV[NX], sPT[NX];
    __shared__ double sLam[NQ * NQ], sG[NQ * WS];
    __shared__ double sX[NX], sXN[NX], sRHS[NQ], sU[NQ];
    __shared__ unsigned char triZi[NTRI_Z], triZj[NTRI_Z], triXi[NTRI_X], triXj[NTRI_X];

    double* const sT = rec + Ly.oT;
    double* const sGC = rec + Ly.oGC;
    double* const sGN = rec + Ly.oGN;
    double* const sLO = rec + Ly.oLO;
    double* const sHI = rec + Ly.oHI;
    double* const sGZ = rec + Ly.oGZ;
    double* const sTL = rec + Ly.oTL;
    double* const sTU = rec + Ly.oTU;
    double* const sLL = rec + Ly.oLL;
    double* const sLU = rec + Ly.oLU;
    double* const sSL = rec + Ly.oSL;
    double* const sZ = rec + Ly.oZ;
    double* const sZA = rec + Ly.oZA;
    double* const sZN = rec + Ly.oZN;
    double* const sL = rec + Ly.oL;
    double* const sW = rec + Ly.oW;
    double* const sPB = rec + Ly.oPB;
    double* const sHQQ = rec + Ly.oHQQ;
    double* const sB = rec + Ly.oB;
    double* const sSC = rec + Ly.oSC;

    for (int e = lane; e < NTRI_Z; e += 64) {
        int i = 0, rem = e;
        while (rem >= NZ - i) { rem -= NZ - i; i++; }
        triZi[e] = (unsigned char)i;
        triZj[e] = (unsigned char)(i + rem);
    }
    for (int e = lane; e < NTRI_X; e += 64) {
        int i = 0, rem = e;
        while (rem >= NX - i) { rem -= NX - i; i++; }
        triXi[e] = (unsigned char)i;
        triXj[e] = (unsigned char)(i + rem);
    }

    const double* xb = xg + (size_t)b * (N + 1) * NX;
    const double* ub = ug + (size_t)b * N * NU;
    const double* pb = pp + (size_t)b * (N + 1) * SMPC_NP;
    const smpc_node_eval* evb = ev + (size_t)b * (N + 1);

    const double dx0_reg = lane < NX ? x0[(size_t)b * NX + lane] - xb[lane] : 0.0;

    // ---- record movement ----------------------------------------------------------------------------------------------
    dbl2 pf[QP_PF];
    auto prefetch = [&](const double* src, int n) {  // n doubles (even), 16-byte aligned source
        const dbl2* s2 = reinterpret_cast<const dbl2*>(src);
        const int n2 = n >> 1;
#pragma unroll
        for (int j = 0; j < QP_PF; j++) {
            const int i = lane + 64 * j;
            if (i < n2) pf[j] = s2[i];
        }
    };
    auto commit = [&](double* dst, int n) {
        dbl2* d2 = reinterpret_cast<dbl2*>(dst);
        const int n2 = n >> 1;
#pragma unroll
        for (int j = 0; j < QP_PF; j++) {
            const int i = lane + 64 * j;
            if (i < n2) d2[i] = pf[j];
        }
    };
    auto write_back = [&](double* dst, const double* src_lds, int n) {  // n even, both 16-byte aligned
        dbl2* d2 = reinterpret_cast<dbl2*>(dst);
        const dbl2* s2 = reinterpret_cast<const dbl2*>(src_lds);
        for (int i = lane; i < (n >> 1); i += 64) d2[i] = s2[i];
    };

    // row product  c_r . z   for the row owned by this lane (z in LDS, layout [u; q; v])
    auto row_dot = [&](int r, const double* z) -> double {
        if (r < rT0) return z[NU + r];
        double a = 0.0;
        if (r < rC0) {
            const double* t = &sT[(r - rT0) * NZ];
#pragma unroll
            for (int c = 0; c < NZ; c++) a = fma(t[c], z[c], a);
        } else if (r < rNN) {
            const double* g = &sGC[(r - rC0) * NQ];
#pragma unroll
            for (int c = 0; c < NQ; c++) a = fma(g[c], z[NU + c], a);
        } else {
#pragma unroll
            for (int c = 0; c < NX; c++) a = fma(sGN[c], z[NU + c], a);
        }
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

    // ---- per-row IPM algebra (one lane = one two-sided row) --------------------------------------------------------------
    struct Dir { double dtl, dll, dtu, dlu, dsl; };
    // directions for the trial value czn = c.z+ ; corr: subtract the Mehrotra second-order term built from cza = c.z_aff
    auto row_dir = [&](int r, double czn, double sigmu, double cw, double cza) -> Dir {
        const bool corr = cw != 0.0;
        Dir o{0.0, 0.0, 0.0, 0.0, 0.0};
        const bool soft = (r == rNN) && sSC[2] >= 0.0;
        const double lo = sLO[r], hi = sHI[r];
        if (lo > -QP_ABSENT) {
            const double tl = sTL[r], ll = sLL[r];
            if (soft) {
                const double sl = sSL[0], nu = sSC[2] - ll;
                const double ill = fast_rcp(ll), inu = fast_rcp(nu);
                const double deff = fast_rcp(tl * ill + sl * inu);
                double ct = 0.0, cs2 = 0.0;
                if (corr) {
                    const double dla = -deff * (cza - lo);
                    const double dta = -tl * dla * ill - tl;
                    const double dsa = sl * dla * inu - sl;
                    ct = cw * dta * dla;
                    cs2 = -cw * dsa * dla;
                }
                const double dl = -deff * (czn - lo + (sigmu - cs2) * inu - (sigmu - ct) * ill);
                o.dll = dl;
                o.dtl = (sigmu - ct - tl * dl) * ill - tl;
                o.dsl = (sigmu - cs2 + sl * dl) * inu - sl;
            } else {
                const double itl = fast_rcp(tl);
                double ct = 0.0;
                if (corr) {
                    const double dta = cza - lo - tl;
                    const double dla = -ll * dta * itl - ll;
                    ct = cw * dta * dla;
                }
                const double dt_ = czn - lo - tl;
                o.dtl = dt_;
                o.dll = (sigmu - ct - ll * dt_) * itl - ll;
            }
        }
        if (hi < QP_ABSENT) {
            const double tu = sTU[r], lu = sLU[r];
            const double itu = fast_rcp(tu);
            double ct = 0.0;
            if (corr) {
                const double dta = hi - cza - tu;
                const double dla = -lu * dta * itu - lu;
                ct = cw * dta * dla;
            }
            const double dt_ = hi - czn - tu;
            o.dtu = dt_;
            o.dlu = (sigmu - ct - lu * dt_) * itu - lu;
        }
        return o;
    };
    // gradient coefficient e_r and barrier weight D_r
    auto row_coeff = [&](int r, double sigmu, double cw, double cza, double* Dr) -> double {
        const bool corr = cw != 0.0;
        double e = 0.0, dsum = 0.0;
        const bool soft = (r == rNN) && sSC[2] >= 0.0;
        const double lo = sLO[r], hi = sHI[r];
        if (lo > -QP_ABSENT) {
            const double tl = sTL[r], ll = sLL[r];
            if (soft) {
                const double sl = sSL[0], nu = sSC[2] - ll;
                const double ill = fast_rcp(ll), inu = fast_rcp(nu);
                const double deff = fast_rcp(tl * ill + sl * inu);
                double ct = 0.0, cs2 = 0.0;
                if (corr) {
                    const double dla = -deff * (cza - lo);
                    const double dta = -tl * dla * ill - tl;
                    const double dsa = sl * dla * inu - sl;
                    ct = cw * dta * dla;
                    cs2 = -cw * dsa * dla;
                }
                e += -ll + deff * (-lo + (sigmu - cs2) * inu - (sigmu - ct) * ill);
                dsum += deff;
            } else {
                const double itl = fast_rcp(tl);
                double ct = 0.0;
                if (corr) {
                    const double dta = cza - lo - tl;
                    const double dla = -ll * dta * itl - ll;
                    ct = cw * dta * dla;
                }
                const double d = ll * itl;
                e += -ll - d * lo - (sigmu - ct) * itl;
                dsum += d;
            }
        }
        if (hi < QP_ABSENT) {
            const double tu = sTU[r], lu = sLU[r];
            const double itu = fast_rcp(tu);
            double ct = 0.0;
            if (corr) {
                const double dta = hi - cza - tu;
                const double dla = -lu * dta * itu - lu;
                ct = cw * dta * dla;
            }
            const double d = lu * itu;
            e += lu - d * hi + (sigmu - ct) * itu;
            dsum += d;
        }
        *Dr = dsum;
        return e;
    };
    // largest step keeping this row's slacks / multipliers positive, and its terms of  sum(lambda t)(alpha)
    auto row_ratio = [&](int r, const Dir& d, double* S1, double* S2) -> double {
        double a = 1e300;
        const bool soft = (r == rNN) && sSC[2] >= 0.0;
        if (sLO[r] > -QP_ABSENT) {
            const double tl = sTL[r], ll = sLL[r];
            if (d.dtl < 0.0) a = fmin(a, -tl / d.dtl);
            if (d.dll < 0.0) a = fmin(a, -ll / d.dll);
            *S1 += ll * d.dtl + tl * d.dll;
            *S2 += d.dll * d.dtl;
            if (soft) {
                const double sl = sSL[0], nu = sSC[2] - ll;
                if (d.dsl < 0.0) a = fmin(a, -sl / d.dsl);
                if (d.dll > 0.0) a = fmin(a, nu / d.dll);
                *S1 += nu * d.dsl - sl * d.dll;
                *S2 += -d.dll * d.dsl;
            }
        }
        if (sHI[r] < QP_ABSENT) {
            const double tu = sTU[r], lu = sLU[r];
            if (d.dtu < 0.0) a = fmin(a, -tu / d.dtu);
            if (d.dlu < 0.0) a = fmin(a, -lu / d.dlu);
            *S1 += lu * d.dtu + tu * d.dlu;
            *S2 += d.dlu * d.dtu;
        }
        return a;
    };
    auto row_comp = [&](int r) -> double {  // lambda t (+ nu s) of this row
        double acc = 0.0;
        if (sLO[r] > -QP_ABSENT) {
            acc += sLL[r] * sTL[r];
            if (r == rNN && sSC[2] >= 0.0) acc += (sSC[2] - sLL[r]) * sSL[0];
        }
        if (sHI[r] < QP_ABSENT) acc += sLU[r] * sTU[r];
        return acc;
    };

    // ---- Cholesky factor in registers (every lane redundantly): strictly-lower entries + inverse diagonal -----------------
    double Lr[NQ][NQ], Linv[NQ];
    auto chol_from_lds = [&](const double* A) -> bool {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            double dsum = A[j * NQ + j];
#pragma unroll
            for (int t = 0; t < j; t++) dsum = fma(-Lr[j][t], Lr[j][t], dsum);
            ok = ok && (dsum > 0.0);
            const double inv = fast_rsqrt(dsum);
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
            for (int j = 0; j < i; j++) Lr[i][j] = sL[i * NQ + j];
            Linv[i] = sL[i * NQ + i];
        }
    };

    // backward costate step at stage k: given sGH, sPV (= p_{k+1}), sPB, sW, L in registers; updates sPV and W's w column
    auto vector_back = [&](int k) {
        if (lane < NX) sPT[lane] = sPV[lane] + sPB[lane];
        lds_fence();
        if (lane < NQ) sG[lane * WS + NX] = sGH[lane] + cB * sPT[lane] + dt * sPT[NQ + lane];  // rho
        lds_fence();
        double wv[NQ];
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            double v = sG[i * WS + NX];
#pragma unroll
            for (int t = 0; t < i; t++) v = fma(-Lr[i][t], wv[t], v);
            wv[i] = v * Linv[i];
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NQ; i++) sW[i * WS + NX] = wv[i];
        }
        if (k > 0 && lane < NX) {
            // p_k = gh_x + A^T pt - W^T w
            const int i = lane;
            double v = sGH[NU + i] + (i < NQ ? sPT[i] : dt * sPT[i - NQ] + sPT[i]);
#pragma unroll
            for (int t = 0; t < NQ; t++) v = fma(-sW[t * WS + i], wv[t], v);
            sXN[i] = v;
        }
        lds_fence();
        if (k > 0 && lane < NX) sPV[lane] = sXN[lane];
        lds_fence();
    };

    // =====================================================================================================================
    // phase 0: assemble the stage records, initial point, initial residual norm and complementarity
    // =====================================================================================================================
    double R0 = 0.0, mu_acc = 0.0;
    int m_comp = 0;
    if (lane < NX) sX[lane] = dx0_reg;
    prefetch(reinterpret_cast<const double*>(evb), EV_D);
    for (int k = 0; k <= N; k++) {
        commit(sEV, EV_D);
        lds_fence();
        if (k < N) prefetch(reinterpret_cast<const double*>(evb + k + 1), EV_D);
        const smpc_node_eval& e = *reinterpret_cast<const smpc_node_eval*>(sEV);
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
        // z = 0 except the fixed dx_0; z_aff, z+ start defined
        if (lane < NZ) {
            const double z0 = (k == 0 && lane >= NU) ? sX[lane - NU] : 0.0;
            sZ[lane] = z0; sZA[lane] = z0; sZN[lane] = z0;
        }
        for (int el = lane; el < Ly.bH - Ly.bFac; el += 64) rec[Ly.bFac + el] = 0.0;
        lds_fence();
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
                mu_acc += ll * tl;
                if (soft) { sSL[0] = s0; mu_acc += (sSC[2] - ll) * s0; }
            }
            if (sHI[r] < QP_ABSENT) {
                const double slack = sHI[r] - cz;
                tu = fmax(slack, QP_THR);
                lu = D->qp_mu0 / tu;
                r0_loc = fmax(r0_loc, fabs(slack - tu));
                cnt += 1;
                mu_acc += lu * tu;
            }
            sTL[r] = tl; sLL[r] = ll; sTU[r] = tu; sLU[r] = lu;
            sE[r] = -(ll - lu);
        }
        if (lane == 0 && !(sSC[2] >= 0.0)) sSL[0] = 0.0;
        if (lane == 1) sSL[1] = 0.0;
        lds_fence();
        // stationarity residual at the initial point (pi = 0): g - C^T (ll - lu); dx_0 does not enter (no cost cross term)
        if (lane < NZ && !(k == 0 && lane >= NU) && !(last && lane < NU))
            r0_loc = fmax(r0_loc, fabs(sGZ[lane] + ct_dot(lane, sE)));
        if (!last && lane < NX) {
            double ax = 0.0;  // dynamics defect of the initial point: next dx is 0
            if (k == 0) ax = lane < NQ ? sX[lane] + dt * sX[NQ + lane] : sX[lane];
            r0_loc = fmax(r0_loc, fabs(ax + sB[lane]));
        }
        R0 = fmax(R0, wave_max(r0_loc));
        m_comp += (int)wave_sum((double)cnt);
        write_back(ws + (size_t)k * Ly.stride, rec, Ly.stride);
        lds_fence();
    }
    if (m_comp == 0) m_comp = 1;
    const double inv_m = 1.0 / (double)m_comp;
    double mu = wave_sum(mu_acc) * inv_m;

    // =====================================================================================================================
    // main loop
    // =====================================================================================================================
    double rho_lin = 1.0, alpha = 0.0, sigmu = 0.0, corr_w = 1.0;
    bool pending = false;  // a step (alpha, z+, sigma mu) computed by F2 and not yet applied to state / z
    int it = 0, st_code = 2;
    const double tol = D->qp_tol;
    const int max_iter = D->qp_max_iter;
    bool broke = false;

    for (it = 0; it < max_iter; it++) {
        if (mu <= tol && rho_lin * R0 <= tol) { st_code = 0; break; }

        // ---------------- sweep B1: apply the pending step, factorise H + C^T D C, predictor costate ----------------------
        double mu_new = 0.0;
        prefetch(ws + (size_t)N * Ly.stride, Ly.stride);
        for (int k = N; k >= 0; k--) {
            const bool last = (k == N);
            commit(rec, Ly.stride);
            lds_fence();
            if (k > 0) prefetch(ws + (size_t)(k - 1) * Ly.stride, Ly.stride);
            if (pending) {
                Dir d{0, 0, 0, 0, 0};
                if (lane < NRT) d = row_dir(lane, row_dot(lane, sZN), sigmu, corr_w, row_dot(lane, sZA));
                lds_fence();
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
                lds_fence();
            }
            if (lane < NRT) {
                double Dr;
                sE[lane] = row_coeff(lane, 0.0, 0.0, 0.0, &Dr);
                sD[lane] = Dr;
                mu_new += row_comp(lane);
            }
            lds_fence();
            for (int el = lane; el < NQ * NZ; el += 64) sTD[el] = sT[el] * sD[rT0 + el / NZ];
            lds_fence();
            // Hessian (upper triangle, mirrored) and gradient
            for (int el = lane; el < NTRI_Z; el += 64) {
                const int i = triZi[el], j = triZj[el];
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
            lds_fence();
            if (last) {
                for (int el = lane; el < NX * NX; el += 64) sP[el] = sH[(NU + el / NX) * NZ + NU + el % NX];
                if (lane < NX) sPV[lane] = sGH[NU + lane];
                lds_fence();
            } else {
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
                    sG[i * WS + j] = sH[i * NZ + NU + j] + v;
                }
                lds_fence();
                if (!chol_from_lds(sLam)) broke = true;
                // W = L^-1 G (one column per lane)
                if (lane < NX) {
                    double col[NQ];
#pragma unroll
                    for (int i = 0; i < NQ; i++) {
                        double v = sG[i * WS + lane];
#pragma unroll
                        for (int t = 0; t < i; t++) v = fma(-Lr[i][t], col[t], v);
                        col[i] = v * Linv[i];
                        sW[i * WS + lane] = col[i];
                    }
                }
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < NQ; i++) {
#pragma unroll
                        for (int j = 0; j < NQ; j++) sL[i * NQ + j] = j < i ? Lr[i][j] : (j == i ? Linv[i] : 0.0);
                    }
                }
                lds_fence();
                vector_back(k);
                if (k > 0) {
                    // P_k = Hxx + A^T P A - W^T W: upper triangle into the (now free) sTD scratch, then mirrored into sP
                    for (int el = lane; el < NTRI_X; el += 64) {
                        const int i = triXi[el], j = triXj[el];
                        double a = sH[(NU + i) * NZ + NU + j];
                        // A^T P A, blocks: [P11, dt P11 + P12; dt P11 + P21, dt^2 P11 + dt (P12 + P21) + P22]
                        if (j < NQ) a += sP[i * NX + j];
                        else if (i < NQ) a += dt * sP[i * NX + (j - NQ)] + sP[i * NX + j];
                        else {
                            const int ii = i - NQ, jj = j - NQ;
                            a += dt * dt * sP[ii * NX + jj] + dt * (sP[ii * NX + j] + sP[i * NX + jj]) + sP[i * NX + j];
                        }
#pragma unroll
                        for (int t = 0; t < NQ; t++) a = fma(-sW[t * WS + i], sW[t * WS + j], a);
                        sTD[el] = a;
                    }
                    lds_fence();
                    for (int el = lane; el < NTRI_X; el += 64) {
                        const int i = triXi[el], j = triXj[el];
                        const double a = sTD[el];
                        sP[i * NX + j] = a;
                        sP[j * NX + i] = a;
                    }
                    lds_fence();
                }
            }
            // write back what changed: state + z (if a step was applied) and the factors
            double* w = ws + (size_t)k * Ly.stride;
            if (pending) write_back(w + Ly.bState, rec + Ly.bState, Ly.bFac - Ly.bState);
            if (!last) write_back(w + Ly.bFac, rec + Ly.bFac, Ly.bH - Ly.bFac);
            lds_fence();
        }
        if (wave_max(broke ? 1.0 : 0.0) > 0.0) { st_code = 4; pending = false; break; }
        if (pending) {
            mu = wave_sum(mu_new) * inv_m;
            pending = false;
            if (!(mu == mu)) { st_code = 4; break; }
        }

        // ---------------- sweep F1: predictor roll-out, affine step length, centring ---------------------------------------
        double amin = 1e300, S1 = 0.0, S2 = 0.0;
        if (lane < NX) sX[lane] = dx0_reg;
        prefetch(ws, Ly.nPrefix);
        for (int k = 0; k <= N; k++) {
            commit(rec, Ly.nPrefix);
            lds_fence();
            if (k < N) prefetch(ws + (size_t)(k + 1) * Ly.stride, Ly.nPrefix);
            if (k < N) {
                load_L_regs();
                if (lane < NQ) {
                    double v = sW[lane * WS + NX];
#pragma unroll
                    for (int j = 0; j < NX; j++) v = fma(sW[lane * WS + j], sX[j], v);
                    sRHS[lane] = -v;
                }
                lds_fence();
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
                    for (int i = 0; i < NQ; i++) { sZA[i] = uk[i]; sU[i] = uk[i]; }
                }
                if (lane < NX) sZA[NU + lane] = sX[lane];
                lds_fence();
                if (lane < NX) {
                    const int i = lane < NQ ? lane : lane - NQ;
                    const double bk = sB[lane];
                    sXN[lane] = lane < NQ ? sX[i] + dt * sX[NQ + i] + cB * sU[i] + bk : sX[NQ + i] + dt * sU[i] + bk;
                }
            } else {
                if (lane < NZ) sZA[lane] = lane >= NU ? sX[lane - NU] : 0.0;
            }
            lds_fence();
            if (lane < NRT) {
                const Dir d = row_dir(lane, row_dot(lane, sZA), 0.0, 0.0, 0.0);
                amin = fmin(amin, row_ratio(lane, d, &S1, &S2));
            }
            write_back(ws + (size_t)k * Ly.stride + Ly.oZA, sZA, QpLayout<NQ>::NZP);
            if (k < N && lane < NX) sX[lane] = sXN[lane];
            lds_fence();
        }
        const double a_aff = fmin(1.0, wave_min(amin));
        S1 = wave_sum(S1);
        S2 = wave_sum(S2);
        const double mu_aff = (mu * (double)m_comp + a_aff * S1 + a_aff * a_aff * S2) * inv_m;
        double sigma = mu_aff / mu;
        sigma = sigma * sigma * sigma;
        sigmu = sigma * mu;
        // safeguard against Mehrotra cycling (see oracle): damp the second-order term when the affine step is blocked early
        corr_w = a_aff >= 0.3 ? 1.0 : a_aff * a_aff;

        // ---------------- sweep B2: corrector gradient and costate recursion with the stored factors ------------------------
        prefetch(ws + (size_t)N * Ly.stride, Ly.nPrefix);
        for (int k = N; k >= 0; k--) {
            const bool last = (k == N);
            commit(rec, Ly.nPrefix);
            lds_fence();
            if (k > 0) prefetch(ws + (size_t)(k - 1) * Ly.stride, Ly.nPrefix);
            if (lane < NRT) {
                double Dr;
                sE[lane] = row_coeff(lane, sigmu, corr_w, row_dot(lane, sZA), &Dr);
            }
            lds_fence();
            if (lane < NZ) sGH[lane] = sGZ[lane] + ct_dot(lane, sE);
            lds_fence();
            if (last) {
                if (lane < NX) sPV[lane] = sGH[NU + lane];
                lds_fence();
            } else {
                load_L_regs();
                vector_back(k);
                if (lane < NQ) ws[(size_t)k * Ly.stride + Ly.oW + lane * WS + NX] = sW[lane * WS + NX];
                lds_fence();
            }
        }

        // ---------------- sweep F2: corrector roll-out and step length --------------------------------------------------------
        amin = 1e300;
        double S1c = 0.0, S2c = 0.0;
        if (lane < NX) sX[lane] = dx0_reg;
        prefetch(ws, Ly.nPrefix);
        for (int k = 0; k <= N; k++) {
            commit(rec, Ly.nPrefix);
            lds_fence();
            if (k < N) prefetch(ws + (size_t)(k + 1) * Ly.stride, Ly.nPrefix);
            if (k < N) {
                load_L_regs();
                if (lane < NQ) {
                    double v = sW[lane * WS + NX];
#pragma unroll
                    for (int j = 0; j < NX; j++) v = fma(sW[lane * WS + j], sX[j], v);
                    sRHS[lane] = -v;
                }
                lds_fence();
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
                    for (int i = 0; i < NQ; i++) { sZN[i] = uk[i]; sU[i] = uk[i]; }
                }
                if (lane < NX) sZN[NU + lane] = sX[lane];
                lds_fence();
                if (lane < NX) {
                    const int i = lane < NQ ? lane : lane - NQ;
                    const double bk = sB[lane];
                    sXN[lane] = lane < NQ ? sX[i] + dt * sX[NQ + i] + cB * sU[i] + bk : sX[NQ + i] + dt * sU[i] + bk;
                }
            } else {
                if (lane < NZ) sZN[lane] = lane >= NU ? sX[lane - NU] : 0.0;
            }
            lds_fence();
            if (lane < NRT) {
                const Dir d = row_dir(lane, row_dot(lane, sZN), sigmu, corr_w, row_dot(lane, sZA));
                amin = fmin(amin, row_ratio(lane, d, &S1c, &S2c));
            }
            write_back(ws + (size_t)k * Ly.stride + Ly.oZN, sZN, QpLayout<NQ>::NZP);
            if (k < N && lane < NX) sX[lane] = sXN[lane];
            lds_fence();
        }
        alpha = fmin(1.0, QP_FTB * wave_min(amin));
        if (!(alpha == alpha)) { st_code = 4; break; }
        if (alpha < QP_ALPHA_MIN) { st_code = 3; break; }
        pending = true;
        rho_lin *= (1.0 - alpha);
        // sum(lambda t) is a quadratic in the step length: the new complementarity is known before the step is applied
        mu = (mu * (double)m_comp + alpha * wave_sum(S1c) + alpha * alpha * wave_sum(S2c)) * inv_m;
        if (!(mu == mu)) { st_code = 4; pending = false; break; }
    }
    if (it == max_iter && st_code == 2 && mu <= tol && rho_lin * R0 <= tol) st_code = 0;

    // ---- full SQP step (FIXED_STEP, parser.py:139), applying the last IPM step if it is still pending -------------------
    bool bad = false;
    const double a_fin = pending ? alpha : 0.0;
    for (int k = 0; k <= N; k++) {
        const double* zk = ws + (size_t)k * Ly.stride + Ly.oZ;
        const double* zn = ws + (size_t)k * Ly.stride + Ly.oZN;
        if (lane < NX) {
            const double z = zk[NU + lane];
            const double v = xb[(size_t)k * NX + lane] + z + a_fin * (zn[NU + lane] - z);
            x_out[((size_t)b * (N + 1) + k) * NX + lane] = v;
            bad |= !(v == v);
        }
        if (k < N && lane < NU) {
            const double z = zk[lane];
            const double v = ub[(size_t)k * NU + lane] + z + a_fin * (zn[lane] - z);
            u_out[((size_t)b * N + k) * NU + lane] = v;
            bad |= !(v == v);
        }
    }
    const bool any_bad = wave_max(bad ? 1.0 : 0.0) > 0.0;
    if (lane == 0) {
        // acados' RTI tolerates a QP that stopped at its iteration cap (see oracle); breakdown / min-step are QP failures
        int stc = (st_code == 0 || st_code == 2) ? SMPC_STATUS_SUCCESS : SMPC_STATUS_QP_FAILURE;
        if (any_bad && stc == SMPC_STATUS_SUCCESS) stc = SMPC_STATUS_NAN;
        status[b] = stc;
        if (qp_iter) qp_iter[b] = it;
    }
}

}  // namespace smpc
