// kernel_qp.hpp -- HOT LOOP B: the stage-structured QP of one RTI step, solved by a Mehrotra primal-dual interior-point
// method whose Newton systems are factorised by a Riccati recursion (the role HPIPM plays for acados; N4/N5 in SURVEY
// section 2; options of reference controller.py:97-110, config.yaml:15-21).
//
// Mapping: ONE WAVEFRONT PER PAIR OF OCP INSTANCES (block = 64 threads): lanes 0-31 own one instance, lanes 32-63 another,
// each with its own LDS region, workspace and iteration state; the two halves never exchange data, so every branch
// below is half-uniform and the SIMT exec mask does the rest (a half that has converged simply idles until its twin is
// done; longest-first dispatch pairs instances with similar iteration counts).  Every per-stage phase of the algorithm is
// at most 32 wide except the two matrix assemblies, so a half-wave per instance nearly halves the instructions issued per
// instance compared with a full wave (v3, git history) while the LDS footprint per CU is the same (8 waves x 2 x 10 KB).
// Each half walks the horizon four times per IPM iteration
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
#include <type_traits>

#include "device_model.hpp"

namespace smpc {

constexpr double QP_THR = 1e-1;
constexpr double QP_FTB = 0.995;
constexpr double QP_ALPHA_MIN = 1e-12;
constexpr double QP_ABSENT = 1e300;  // sentinel for a missing bound side inside the workspace

__host__ __device__ inline int qp_even(int n) { return (n + 1) & ~1; }
constexpr int qp_even_c(int n) { return (n + 1) & ~1; }

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

__device__ __forceinline__ double half_min(double v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double half_max(double v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double half_sum(double v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
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

// diagnostic build (-DQP_PROFILE, scripts/qp_phase_profile.py): per-phase shader-clock sums over all waves
#ifdef QP_PROFILE
__device__ unsigned long long g_qp_prof[16];
#define QPT(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define QPT(i) do { } while (0)
#endif

#ifndef QP_WAVES_PER_EU
#define QP_WAVES_PER_EU 2
#endif
// 16-byte prefetch registers per lane: QpPf<NQ> * 32 lanes * 2 doubles must hold the largest record (checked in smpc_create)
template <int NQ> struct QpPf { static constexpr int value = NQ <= 6 ? 9 : 13; };

template <int NQ, int MRT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(QP_WAVES_PER_EU, QP_WAVES_PER_EU))) void k_qp_ipm(const smpc_problem_desc* __restrict__ D, int B, int N,
                                               const double* __restrict__ x0, const double* __restrict__ xg,
                                               const double* __restrict__ ug, const double* __restrict__ pp,
                                               const double* __restrict__ lo_st, const double* __restrict__ hi_st,
                                               const smpc_node_eval* __restrict__ ev, double* __restrict__ ws_all,
                                               double* __restrict__ x_out, double* __restrict__ u_out,
                                               int32_t* __restrict__ status, int32_t* __restrict__ qp_iter,
                                               const int32_t* __restrict__ order, int32_t* __restrict__ last_iter, long bnd_stride) {
    constexpr int NX = 2 * NQ, NU = NQ, NZ = 3 * NQ, WS = NX + 1;
    constexpr int MAXR = NX + NQ + SMPC_MAX_ROWS + 1;
    constexpr int NTRI_Z = NZ * (NZ + 1) / 2, NTRI_X = NX * (NX + 1) / 2;
    constexpr int QP_PF = QpPf<NQ>::value;
    constexpr int REC_MAX = QP_PF * 64;
    constexpr int EV_D = (int)(sizeof(smpc_node_eval) / sizeof(double));
#ifdef QP_PROFILE
    unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    const int hl = threadIdx.x & 31, half = threadIdx.x >> 5;
    const int slot = 2 * (int)blockIdx.x + half;
    if (slot >= B) return;  // odd batch: the last half-wave has no instance
    // longest-expected-first dispatch: slot i takes the instance with the i-th largest iteration count of the previous
    // call (instances are independent, so the order only changes the makespan, never a result)
    const int b = order ? order[slot] : slot;
    const QpLayout<NQ> Ly(MRT >= 0 ? MRT : D->n_rows);
    const int MR = MRT >= 0 ? MRT : Ly.MR;
    const int NRT = NX + NQ + MR + 1;
    const int rT0 = NX, rC0 = NX + NQ, rNN = NX + NQ + MR;  // first torque row, first collision row, the nn row
    double* ws = ws_all + (size_t)b * Ly.per_instance(N);
    const double dt = D->dt, cB = 0.5 * dt * dt;

    // ---- LDS ---------------------------------------------------------------------------------------------------------
    constexpr int EV_PAD = EV_D + (EV_D & 1);
    constexpr int SCR_A = 3 * NX * NX;                                  // Hxx | P ping | P pong; also the phase-0 staging area
    constexpr int SCR = SCR_A > EV_PAD ? SCR_A : EV_PAD;
    constexpr int MAXR_E = qp_even_c(MRT >= 0 ? NX + NQ + MRT + 1 : MAXR);
    // one region per half-wave: [record | scratch | D | E | gh | p ping | p pong | rho | dx0 | Lambda | G]
    constexpr int O_SCR = REC_MAX, O_D = O_SCR + SCR, O_E = O_D + MAXR_E, O_GH = O_E + MAXR_E, O_PVA = O_GH + qp_even_c(NZ),
                  O_PVB = O_PVA + NX, O_RHO = O_PVB + NX, O_X0 = O_RHO + qp_even_c(NQ), O_LAM = O_X0 + NX,
                  O_G = O_LAM + qp_even_c(NQ * NQ), HALF_D = O_G + qp_even_c(NQ * WS);
    __shared__ __attribute__((aligned(16))) double smem[2 * HALF_D];
    __shared__ unsigned char triZi[NTRI_Z], triZj[NTRI_Z], triXi[NTRI_X], triXj[NTRI_X];
    double* const rec = smem + half * HALF_D;   // the current stage record
    double* const scr = rec + O_SCR;
    double* const sHXX = scr;
    double* const sPa = scr + NX * NX;
    double* const sPb = scr + 2 * NX * NX;
    double* const sEV = scr;
    double* const sD = rec + O_D;
    double* const sE = rec + O_E;
    double* const sGH = rec + O_GH;
    double* const sPVa = rec + O_PVA;
    double* const sPVb = rec + O_PVB;
    double* const sRho = rec + O_RHO;
    double* const sX0 = rec + O_X0;
    double* const sLam = rec + O_LAM;
    double* const sG = rec + O_G;

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

    for (int e = hl; e < NTRI_Z; e += 32) {
        int i = 0, rem = e;
        while (rem >= NZ - i) { rem -= NZ - i; i++; }
        triZi[e] = (unsigned char)i;
        triZj[e] = (unsigned char)(i + rem);
    }
    for (int e = hl; e < NTRI_X; e += 32) {
        int i = 0, rem = e;
        while (rem >= NX - i) { rem -= NX - i; i++; }
        triXi[e] = (unsigned char)i;
        triXj[e] = (unsigned char)(i + rem);
    }

    const double* xb = xg + (size_t)b * (N + 1) * NX;
    const double* ub = ug + (size_t)b * N * NU;
    const double* pb = pp + (size_t)b * (N + 1) * SMPC_NP;
    const smpc_node_eval* evb = ev + (size_t)b * (N + 1);

    const double dx0_reg = hl < NX ? x0[(size_t)b * NX + hl] - xb[hl] : 0.0;

    // ---- record movement ----------------------------------------------------------------------------------------------
    dbl2 pf[QP_PF];
    auto prefetch = [&](const double* src, int n) {  // n doubles (even), 16-byte aligned source
        const dbl2* s2 = reinterpret_cast<const dbl2*>(src);
        const int n2 = n >> 1;
#pragma unroll
        for (int j = 0; j < QP_PF; j++) {
            const int i = hl + 32 * j;
            if (i < n2) pf[j] = s2[i];
        }
    };
    auto commit = [&](double* dst, int n) {
        dbl2* d2 = reinterpret_cast<dbl2*>(dst);
        const int n2 = n >> 1;
#pragma unroll
        for (int j = 0; j < QP_PF; j++) {
            const int i = hl + 32 * j;
            if (i < n2) d2[i] = pf[j];
        }
    };
    auto write_back = [&](double* dst, const double* src_lds, int n) {  // n even, both 16-byte aligned
        dbl2* d2 = reinterpret_cast<dbl2*>(dst);
        const dbl2* s2 = reinterpret_cast<const dbl2*>(src_lds);
        for (int i = hl; i < (n >> 1); i += 32) d2[i] = s2[i];
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

    // register-resident trial point -> value of row r (lane-dependent), z = [u; x] held by every lane
    auto sel = [&](const double* v, auto n_tag, int idx) -> double {
        constexpr int n = decltype(n_tag)::value;
        double o = v[0];
#pragma unroll
        for (int c = 1; c < n; c++) o = (idx == c) ? v[c] : o;
        return o;
    };
    using TagNX = std::integral_constant<int, NX>;
    using TagNZ = std::integral_constant<int, NZ>;
    using TagNQ = std::integral_constant<int, NQ>;
    auto row_dot_reg = [&](int r, const double* z) -> double {
        if (r < rT0) return sel(z + NU, TagNX{}, r);
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
    // forward step in registers: u = -L^-T (W x + w), z = [u; x], x <- A x + B u + b   (every lane holds x redundantly)
    auto roll_out = [&](bool has_u, double* xr, double* zr) {
        if (has_u) {
            double rhs[NQ], uk[NQ];
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                double v = sW[i * WS + NX];
#pragma unroll
                for (int jx = 0; jx < NX; jx++) v = fma(sW[i * WS + jx], xr[jx], v);
                rhs[i] = -v;
            }
#pragma unroll
            for (int i = NQ - 1; i >= 0; i--) {
                double v = rhs[i];
#pragma unroll
                for (int t = i + 1; t < NQ; t++) v = fma(-Lr[t][i], uk[t], v);
                uk[i] = v * Linv[i];
            }
#pragma unroll
            for (int i = 0; i < NQ; i++) zr[i] = uk[i];
#pragma unroll
            for (int i = 0; i < NX; i++) zr[NU + i] = xr[i];
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                const double q = xr[i], v = xr[NQ + i];
                xr[i] = q + dt * v + cB * uk[i] + sB[i];
                xr[NQ + i] = v + dt * uk[i] + sB[NQ + i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NQ; i++) zr[i] = 0.0;
#pragma unroll
            for (int i = 0; i < NX; i++) zr[NU + i] = xr[i];
        }
    };

    // =====================================================================================================================
    // phase 0: assemble the stage records, initial point, initial residual norm and complementarity
    // =====================================================================================================================
    double R0 = 0.0, mu_acc = 0.0;
    int m_comp = 0;
    if (hl < NX) sX0[hl] = dx0_reg;
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

        for (int el = hl; el < NQ * NZ; el += 32) {
            const int r = el / NZ, c = el % NZ;
            double v = 0.0;
            if (!last) v = c < NQ ? e.M[r * NQ + c] : (c < 2 * NQ ? e.dtau_dq[r * NQ + c - NQ] : e.dtau_dv[r * NQ + c - 2 * NQ]);
            sT[el] = v;
        }
        for (int el = hl; el < MR * NQ; el += 32) sGC[el] = e.row_grad[el];
        if (hl < NX) sGN[hl] = nn_on ? e.nn_grad[hl] : 0.0;
        for (int el = hl; el < NQ * NQ; el += 32) {
            const int i = el / NQ, j = el % NQ;
            sHQQ[el] = (reach ? cs * e.cost_hess_qq[el] : 0.0) + (i == j ? lm : 0.0);
        }
        if (hl < NZ) {
            double g = 0.0;
            if (reach) {
                if (hl < NU) g = last ? 0.0 : cs * 2.0 * D->R * ub[(size_t)k * NU + hl];
                else if (hl < NU + NQ) g = cs * e.cost_grad_q[hl - NU];
            }
            sGZ[hl] = g;
        }
        if (hl < NX) {
            double bb = 0.0;
            if (!last) {
                const double* xn = xk + NX;
                const int i = hl < NQ ? hl : hl - NQ;
                const double uk = ub[(size_t)k * NU + i];
                bb = hl < NQ ? xk[i] + dt * xk[NQ + i] + cB * uk - xn[i] : xk[NQ + i] + dt * uk - xn[NQ + i];
            }
            sB[hl] = bb;
        }
        if (hl == 0) {
            sSC[0] = (reach && !last ? cs * 2.0 * D->R : 0.0) + lm;  // Huu diagonal
            sSC[1] = lm;                                              // LM on the velocity diagonal
            sSC[2] = nn_on ? (last ? D->nn_soft_e : D->nn_soft_run) : -1.0;
            sSC[3] = 0.0;
        }
        for (int r = hl; r < NRT; r += 32) {
            double lo = -QP_ABSENT, hi = QP_ABSENT;
            if (r < rT0) {
                const size_t bo = (size_t)b * bnd_stride + (size_t)k * NX + r;  // bnd_stride = 0: bounds shared by all instances
                const double l = lo_st[bo], h = hi_st[bo];
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
        if (hl < NZ) {
            const double z0 = (k == 0 && hl >= NU) ? sX0[hl - NU] : 0.0;
            sZ[hl] = z0; sZA[hl] = z0; sZN[hl] = z0;
        }
        for (int el = hl; el < Ly.bH - Ly.bFac; el += 32) rec[Ly.bFac + el] = 0.0;
        lds_fence();
        // initial slacks / multipliers
        double r0_loc = 0.0;
        int cnt = 0;
        for (int r = hl; r < NRT; r += 32) {
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
        if (hl == 0 && !(sSC[2] >= 0.0)) sSL[0] = 0.0;
        if (hl == 1) sSL[1] = 0.0;
        lds_fence();
        // stationarity residual at the initial point (pi = 0): g - C^T (ll - lu); dx_0 does not enter (no cost cross term)
        if (hl < NZ && !(k == 0 && hl >= NU) && !(last && hl < NU))
            r0_loc = fmax(r0_loc, fabs(sGZ[hl] + ct_dot(hl, sE)));
        if (!last && hl < NX) {
            double ax = 0.0;  // dynamics defect of the initial point: next dx is 0
            if (k == 0) ax = hl < NQ ? sX0[hl] + dt * sX0[NQ + hl] : sX0[hl];
            r0_loc = fmax(r0_loc, fabs(ax + sB[hl]));
        }
        R0 = fmax(R0, half_max(r0_loc));
        m_comp += (int)half_sum((double)cnt);
        {
            const double bmax = half_max(hl < NX ? fabs(sB[hl]) : 0.0);
            if (hl == 0) sSC[3] = bmax > 0.0 ? 1.0 : 0.0;
            lds_fence();
        }
        write_back(ws + (size_t)k * Ly.stride, rec, Ly.stride);
        lds_fence();
    }
    QPT(13);
    if (m_comp == 0) m_comp = 1;
    const double inv_m = 1.0 / (double)m_comp;
    double mu = half_sum(mu_acc) * inv_m;

    // =====================================================================================================================
    // main loop
    // =====================================================================================================================
    double rho_lin = 1.0, alpha = 0.0, sigmu = 0.0, corr_w = 1.0;
    bool pending = false;  // a step (alpha, z+, sigma mu) computed by F2 and not yet applied to state / z
    int it = 0, st_code = 2;
    const double tol = D->qp_tol;
    const int max_iter = D->qp_max_iter;
    bool broke = false;
    double *Pc = sPa, *Pn = sPb;      // P_{k+1} (in use) / P_k (being built)
    double *pvc = sPVa, *pvn = sPVb;  // costate vectors, same ping-pong
    lds_fence();

    for (it = 0; it < max_iter; it++) {
        if (mu <= tol && rho_lin * R0 <= tol) { st_code = 0; break; }

        // ---------------- sweep B1: apply the pending step, factorise H + C^T D C, predictor costate ----------------------
        double mu_new = 0.0;
        prefetch(ws + (size_t)N * Ly.stride, Ly.stride);
        for (int k = N; k >= 0; k--) {
            asm volatile("; QPMARK B1_BEGIN");
            const bool last = (k == N);
            commit(rec, Ly.stride);
            lds_fence();
            QPT(0);
            if (k > 0) prefetch(ws + (size_t)(k - 1) * Ly.stride, Ly.stride);
            // -- rows: (apply step) + barrier weights + predictor coefficients; every lane touches only its own row
            for (int r = hl; r < NRT; r += 32) {
                if (pending) {
                    const Dir d = row_dir(r, row_dot(r, sZN), sigmu, corr_w, row_dot(r, sZA));
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
                double Dr;
                sE[r] = row_coeff(r, 0.0, 0.0, 0.0, &Dr);
                sD[r] = Dr;
                mu_new += row_comp(r);
            }
            if (pending && hl < NZ) sZ[hl] += alpha * (sZN[hl] - sZ[hl]);
            const bool bflag = !last && sSC[3] != 0.0;
            if (!last && hl < NX) {
                double a = 0.0;
                if (bflag) {
#pragma unroll
                    for (int jx = 0; jx < NX; jx++) a = fma(Pc[hl * NX + jx], sB[jx], a);
                }
                sPB[hl] = a;
            }
            lds_fence();
            QPT(1);
            // -- fused assembly: H + C^T D C, with B^T P B / B^T P A folded into the u-rows, gradient, rho
            for (int el = hl; el < NTRI_Z; el += 32) {
                const int i = triZi[el], j = triZj[el];
                double a = 0.0;
#pragma unroll
                for (int r = 0; r < NQ; r++) a = fma(sT[r * NZ + i] * sD[rT0 + r], sT[r * NZ + j], a);
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
                    if (last) { Pn[ix * NX + jx] = a; Pn[jx * NX + ix] = a; }
                    else sHXX[ix * NX + jx] = a;
                } else if (j < NU) {
                    if (i == j) a += sSC[0];
                    // B^T P B = c^2 P11 + c dt (P12 + P21) + dt^2 P22
                    a += cB * cB * Pc[i * NX + j] + cB * dt * (Pc[i * NX + NQ + j] + Pc[(NQ + i) * NX + j]) +
                         dt * dt * Pc[(NQ + i) * NX + NQ + j];
                    sLam[i * NQ + j] = a;
                    sLam[j * NQ + i] = a;
                } else {
                    // B^T P A: left block c P11 + dt P21 ; right block dt (c P11 + dt P21) + c P12 + dt P22
                    const int jx = j - NU;
                    if (jx < NQ) a += cB * Pc[i * NX + jx] + dt * Pc[(NQ + i) * NX + jx];
                    else {
                        const int jj = jx - NQ;
                        a += dt * (cB * Pc[i * NX + jj] + dt * Pc[(NQ + i) * NX + jj]) + cB * Pc[i * NX + NQ + jj] +
                             dt * Pc[(NQ + i) * NX + NQ + jj];
                    }
                    sG[i * WS + jx] = a;
                }
            }
            if (hl < NZ) {
                const double gh = sGZ[hl] + ct_dot(hl, sE);
                if (last) {
                    if (hl >= NU) pvn[hl - NU] = gh;
                } else if (hl < NU) {
                    // rho = gh_u + B^T (p_{k+1} + P b)
                    sG[hl * WS + NX] = gh + cB * (pvc[hl] + sPB[hl]) + dt * (pvc[NQ + hl] + sPB[NQ + hl]);
                } else {
                    sGH[hl] = gh;
                }
            }
            lds_fence();
            QPT(2);
            if (!last) {
                if (!chol_from_lds(sLam)) broke = true;
                // [W | w] = L^-1 [G | rho], one column per lane
                if (hl <= NX) {
                    double col[NQ];
#pragma unroll
                    for (int i = 0; i < NQ; i++) {
                        double v = sG[i * WS + hl];
#pragma unroll
                        for (int t = 0; t < i; t++) v = fma(-Lr[i][t], col[t], v);
                        col[i] = v * Linv[i];
                        sW[i * WS + hl] = col[i];
                    }
                }
                if (hl == 0) {
#pragma unroll
                    for (int i = 0; i < NQ; i++) {
#pragma unroll
                        for (int jx = 0; jx < NQ; jx++) sL[i * NQ + jx] = jx < i ? Lr[i][jx] : (jx == i ? Linv[i] : 0.0);
                    }
                }
                lds_fence();
                QPT(3);
                if (k > 0) {
                    // P_k = Hxx + A^T P A - W^T W (upper triangle, mirrored into the other buffer) and p_k
                    for (int el = hl; el < NTRI_X; el += 32) {
                        const int i = triXi[el], j = triXj[el];
                        double a = sHXX[i * NX + j];
                        // A^T P A, blocks: [P11, dt P11 + P12; dt P11 + P21, dt^2 P11 + dt (P12 + P21) + P22]
                        if (j < NQ) a += Pc[i * NX + j];
                        else if (i < NQ) a += dt * Pc[i * NX + (j - NQ)] + Pc[i * NX + j];
                        else {
                            const int ii = i - NQ, jj = j - NQ;
                            a += dt * dt * Pc[ii * NX + jj] + dt * (Pc[ii * NX + j] + Pc[i * NX + jj]) + Pc[i * NX + j];
                        }
#pragma unroll
                        for (int t = 0; t < NQ; t++) a = fma(-sW[t * WS + i], sW[t * WS + j], a);
                        Pn[i * NX + j] = a;
                        Pn[j * NX + i] = a;
                    }
                    if (hl < NX) {
                        const int i = hl;
                        // p_k = gh_x + A^T (p_{k+1} + P b) - W^T w
                        double v = sGH[NU + i] + (i < NQ ? pvc[i] + sPB[i]
                                                         : dt * (pvc[i - NQ] + sPB[i - NQ]) + pvc[i] + sPB[i]);
#pragma unroll
                        for (int t = 0; t < NQ; t++) v = fma(-sW[t * WS + i], sW[t * WS + NX], v);
                        pvn[i] = v;
                    }
                }
            }
            QPT(4);
            // write back what changed: state + z (if a step was applied) and the factors
            double* w = ws + (size_t)k * Ly.stride;
            if (pending) write_back(w + Ly.bState, rec + Ly.bState, Ly.bFac - Ly.bState);
            if (!last) write_back(w + Ly.bFac, rec + Ly.bFac, Ly.bH - Ly.bFac);
            lds_fence();
            if (last || k > 0) {
                double* t1 = Pc; Pc = Pn; Pn = t1;
                double* t2 = pvc; pvc = pvn; pvn = t2;
            }
            QPT(5);
            asm volatile("; QPMARK B1_END");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (half_max(broke ? 1.0 : 0.0) > 0.0) { st_code = 4; pending = false; break; }
        if (pending) {
            mu = half_sum(mu_new) * inv_m;
            pending = false;
            if (!(mu == mu)) { st_code = 4; break; }
        }

        // ---------------- sweep F1: predictor roll-out (state in registers), affine step length, centring ------------------
        double amin = 1e300, S1 = 0.0, S2 = 0.0;
        double xr[NX], zr[NZ];
#pragma unroll
        for (int i = 0; i < NX; i++) xr[i] = sX0[i];
        prefetch(ws, Ly.nPrefix);
        for (int k = 0; k <= N; k++) {
            asm volatile("; QPMARK F1_BEGIN");
            commit(rec, Ly.nPrefix);
            lds_fence();
            QPT(6);
            if (k < N) { prefetch(ws + (size_t)(k + 1) * Ly.stride, Ly.nPrefix); load_L_regs(); }
            roll_out(k < N, xr, zr);
            for (int r = hl; r < NRT; r += 32) {
                const Dir d = row_dir(r, row_dot_reg(r, zr), 0.0, 0.0, 0.0);
                amin = fmin(amin, row_ratio(r, d, &S1, &S2));
            }
            if (hl < NZ) ws[(size_t)k * Ly.stride + Ly.oZA + hl] = sel(zr, TagNZ{}, hl);
            lds_fence();
            QPT(7);
            asm volatile("; QPMARK F1_END");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const double a_aff = fmin(1.0, half_min(amin));
        S1 = half_sum(S1);
        S2 = half_sum(S2);
        const double mu_aff = (mu * (double)m_comp + a_aff * S1 + a_aff * a_aff * S2) * inv_m;
        double sigma = mu_aff / mu;
        sigma = fmin(sigma * sigma * sigma, 0.3);  // centring cap (see oracle): halves the iteration tail
        sigmu = sigma * mu;
        // safeguard against Mehrotra cycling (see oracle): damp the second-order term when the affine step is blocked early
        corr_w = a_aff >= 0.3 ? 1.0 : a_aff * a_aff;

        // ---------------- sweep B2: corrector gradient and costate recursion with the stored factors ------------------------
        prefetch(ws + (size_t)N * Ly.stride, Ly.nPrefix);
        for (int k = N; k >= 0; k--) {
            asm volatile("; QPMARK B2_BEGIN");
            const bool last = (k == N);
            commit(rec, Ly.nPrefix);
            lds_fence();
            QPT(8);
            if (k > 0) prefetch(ws + (size_t)(k - 1) * Ly.stride, Ly.nPrefix);
            for (int r = hl; r < NRT; r += 32) {
                double Dr;
                sE[r] = row_coeff(r, sigmu, corr_w, row_dot(r, sZA), &Dr);
            }
            lds_fence();
            if (hl < NZ) {
                const double gh = sGZ[hl] + ct_dot(hl, sE);
                if (last) {
                    if (hl >= NU) pvn[hl - NU] = gh;
                } else if (hl < NU) {
                    sRho[hl] = gh + cB * (pvc[hl] + sPB[hl]) + dt * (pvc[NQ + hl] + sPB[NQ + hl]);
                } else {
                    sGH[hl] = gh;
                }
            }
            lds_fence();
            if (!last) {
                load_L_regs();
                double wv[NQ];
#pragma unroll
                for (int i = 0; i < NQ; i++) {
                    double v = sRho[i];
#pragma unroll
                    for (int t = 0; t < i; t++) v = fma(-Lr[i][t], wv[t], v);
                    wv[i] = v * Linv[i];
                }
                if (k > 0 && hl < NX) {
                    const int i = hl;
                    double v = sGH[NU + i] + (i < NQ ? pvc[i] + sPB[i] : dt * (pvc[i - NQ] + sPB[i - NQ]) + pvc[i] + sPB[i]);
#pragma unroll
                    for (int t = 0; t < NQ; t++) v = fma(-sW[t * WS + i], wv[t], v);
                    pvn[i] = v;
                }
                if (hl < NQ) ws[(size_t)k * Ly.stride + Ly.oW + hl * WS + NX] = sel(wv, TagNQ{}, hl);
            }
            lds_fence();
            if (last || k > 0) { double* t2 = pvc; pvc = pvn; pvn = t2; }
            QPT(9);
            asm volatile("; QPMARK B2_END");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");

        // ---------------- sweep F2: corrector roll-out and step length --------------------------------------------------------
        amin = 1e300;
        double S1c = 0.0, S2c = 0.0;
#pragma unroll
        for (int i = 0; i < NX; i++) xr[i] = sX0[i];
        prefetch(ws, Ly.nPrefix);
        for (int k = 0; k <= N; k++) {
            asm volatile("; QPMARK F2_BEGIN");
            commit(rec, Ly.nPrefix);
            lds_fence();
            QPT(10);
            if (k < N) { prefetch(ws + (size_t)(k + 1) * Ly.stride, Ly.nPrefix); load_L_regs(); }
            roll_out(k < N, xr, zr);
            for (int r = hl; r < NRT; r += 32) {
                const Dir d = row_dir(r, row_dot_reg(r, zr), sigmu, corr_w, row_dot(r, sZA));
                amin = fmin(amin, row_ratio(r, d, &S1c, &S2c));
            }
            if (hl < NZ) ws[(size_t)k * Ly.stride + Ly.oZN + hl] = sel(zr, TagNZ{}, hl);
            lds_fence();
            QPT(11);
            asm volatile("; QPMARK F2_END");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        // fraction to the boundary (see oracle): 0.995, approaching 1 with the complementarity (cap 0.9999) when the step is
        // within 1% of the full Newton step; an earlier-blocked step keeps the classical margin to stay centred
        const double a_max = half_min(amin);
        const double tau_k = a_max >= 0.99 ? fmin(0.9999, fmax(QP_FTB, 1.0 - mu)) : QP_FTB;
        alpha = fmin(1.0, tau_k * a_max);
        if (!(alpha == alpha)) { st_code = 4; break; }
        if (alpha < QP_ALPHA_MIN) { st_code = 3; break; }
        pending = true;
        rho_lin *= (1.0 - alpha);
        // sum(lambda t) is a quadratic in the step length: the new complementarity is known before the step is applied
        mu = (mu * (double)m_comp + alpha * half_sum(S1c) + alpha * alpha * half_sum(S2c)) * inv_m;
        if (!(mu == mu)) { st_code = 4; pending = false; break; }
    }
    if (it == max_iter && st_code == 2 && mu <= tol && rho_lin * R0 <= tol) st_code = 0;

    // ---- full SQP step (FIXED_STEP, parser.py:139), applying the last IPM step if it is still pending -------------------
    bool bad = false;
    const double a_fin = pending ? alpha : 0.0;
    for (int k = 0; k <= N; k++) {
        const double* zk = ws + (size_t)k * Ly.stride + Ly.oZ;
        const double* zn = ws + (size_t)k * Ly.stride + Ly.oZN;
        if (hl < NX) {
            const double z = zk[NU + hl];
            const double v = xb[(size_t)k * NX + hl] + z + a_fin * (zn[NU + hl] - z);
            x_out[((size_t)b * (N + 1) + k) * NX + hl] = v;
            bad |= !(v == v);
        }
        if (k < N && hl < NU) {
            const double z = zk[hl];
            const double v = ub[(size_t)k * NU + hl] + z + a_fin * (zn[hl] - z);
            u_out[((size_t)b * N + k) * NU + hl] = v;
            bad |= !(v == v);
        }
    }
    const bool any_bad = half_max(bad ? 1.0 : 0.0) > 0.0;
#ifdef QP_PROFILE
    QPT(12);
    if (hl == 0) {
        for (int i = 0; i < 14; i++) atomicAdd(&g_qp_prof[i], tacc[i]);
        atomicAdd(&g_qp_prof[14], 1ull);
        atomicAdd(&g_qp_prof[15], (unsigned long long)it);
    }
#endif
    if (hl == 0) {
        // acados' RTI tolerates a QP that stopped at its iteration cap (see oracle); breakdown / min-step are QP failures
        int stc = (st_code == 0 || st_code == 2) ? SMPC_STATUS_SUCCESS : SMPC_STATUS_QP_FAILURE;
        if (any_bad && stc == SMPC_STATUS_SUCCESS) stc = SMPC_STATUS_NAN;
        status[b] = stc;
        if (qp_iter) qp_iter[b] = it;
        if (last_iter) last_iter[b] = it;
    }
}


// order[] = instance indices sorted by decreasing previous iteration count (counting sort, one block)
__global__ __launch_bounds__(1024) void k_order_by_iters(int B, const int32_t* __restrict__ last_iter, int32_t* __restrict__ order) {
    __shared__ int hist[256], offs[256];
    const int t = threadIdx.x;
    if (t < 256) hist[t] = 0;
    __syncthreads();
    for (int i = t; i < B; i += 1024) atomicAdd(&hist[min(max(last_iter[i], 0), 255)], 1);
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int v = 255; v >= 0; v--) { offs[v] = acc; acc += hist[v]; }
    }
    __syncthreads();
    for (int i = t; i < B; i += 1024) {
        const int v = min(max(last_iter[i], 0), 255);
        order[atomicAdd(&offs[v], 1)] = i;
    }
}

}  // namespace smpc
