// kernel_qp.hpp -- HOT LOOP B: the stage-structured QP of one RTI step, solved by a Mehrotra primal-dual interior-point
// method whose Newton systems are factorised by a Riccati recursion (the role HPIPM plays for acados; N4/N5 in SURVEY
// section 2; options of reference controller.py:97-110, config.yaml:15-21).
//
// Two kernels:
//   k_qp_setup : one half-wavefront per (instance, stage): builds the stage's constraint rows, bounds, cost blocks and the
//                initial interior point from the linearisation records (embarrassingly parallel, no horizon loop)
//   k_qp_ipm   : ONE WAVEFRONT PER PAIR OF OCP INSTANCES -- lanes 0-31 own one instance, lanes 32-63 another, each with its
//                own LDS region, workspace and iteration state.  The halves never exchange data, so every branch is
//                half-uniform and the SIMT exec mask does the rest (a converged half idles until its twin is done;
//                longest-first dispatch pairs instances with similar iteration counts).
// Per IPM iteration a half-wave walks the horizon four times
//   B1  backward: apply the previous step, barrier weights, the u rows of H + C^T D C, Cholesky, W = L^-1 [G | rho], L^-1,
//                 P_k (its x-x block of H + C^T D C assembled element by element inside the P update), predictor costate
//                 and gradient gh0
//   F1  forward : predictor roll-out u = -L^-T (W x + w), ratio test, centring; c.z_aff of every row and the vectors
//                 a1 = C^T e1, a2 = C^T e2 (the corrector gradient is gh0 + sigma mu a1 + cw a2)
//   B2  backward: costate recursion of the corrector with the stored factors (touches neither rows nor Jacobian)
//   F2  forward : corrector roll-out, ratio test, z+ and c.z+ of every row (B1 rebuilds the row steps from them)
// Data movement (v8, stores reorganised in v9, factor block packed in v10, the Jacobian stored once in v11; the git history and DESIGN.md
// section 4 have v1-v7 and what each taught):
//   * what a single lane owns -- the bounds, slacks and multipliers of "its" constraint row, its column of [G | rho | I] --
//     goes from the HBM workspace straight into that lane's registers, re-loaded for the next stage right after its
//     last use in this one;
//   * what many lanes read -- the general rows (a row per constraint lane, a column per variable lane), the factor block,
//     the B1 image -- arrives as whole 16-byte pieces (every lane useful: a narrow per-lane global load costs the address
//     unit a full 64-lane instruction) and is laid out in LDS over the buffers that only B1 uses (the factor block
//     double-buffered and committed a stage early in the forward sweeps, so the control chain starts without a hand-off);
//   * LDS otherwise holds the vectors of the recursions (x, u, costate, row coefficients) and B1's matrices (transposed
//     Jacobian image, scaled copy, P ping-pong, Lambda, G, W^T); cross-lane hand-offs are wave-scope LDS fences (no
//     s_barrier, no vmcnt drain);
//   * the corrector costate keeps the symmetric two-step form W^T (L^-1 rho) and u = -L^-T (W x + w): as accurate as
//     substitution; the closed-form Lambda^-1 rho loses the IPM's last iterations to cancellation;
//   * every vector-memory instruction of a stage is unconditional and straight-line (lanes beyond an array's length
//     duplicate its last entry -- same loads, same arithmetic, same stores to the same address -- stage indices are
//     clamped instead of branched on, the horizon's end stage is peeled off): gfx9 counts loads and stores in ONE
//     in-order vmcnt, and the compiler can only wait for "all but the n youngest" when n is the same on every path -- a
//     single conditional load otherwise turns each wait into vmcnt(0) and serialises the stage on the loads it has just
//     issued for the next one (same effect: a loaded register nobody reads, a spill reload in a loop preheader, a
//     pointer that went through inline asm -- see the comments at load_n, stage_ptr and before the forward loops);
//   * prefetch depth by sweep: B1 one stage, F1/F2 one stage for rows and two for the staged blocks, B2 five stages;
//   * LDS latency (point 6: a lone wavefront waits one round trip after the other): no branch around an
//     LDS read (every lane reads with clamped indices and selects), no LDS write between two passes' reads, and the operands
//     of a pass loaded into locals and "held" (hold_rows2 / hold_rows4 below) so that they go out as one batch.
// The double integrator's A, B are never stored (env_model.py:63-67): every product with them is expanded in closed form.
//
// The algorithm is the one restated in oracle/smpc_oracle.cpp::qp_ipm (same initial point, Mehrotra rule, step rule and
// exit test), so the two agree to rounding; the implementation shares nothing with it.
#pragma once
#include <type_traits>

#include "device_model.hpp"

namespace smpc {

constexpr double QP_THR = 1e-1;        // slack floor of a soft row and of its slack variable
constexpr double QP_THR_HARD = 3e-2;   // slack floor of a hard row per unit of its gradient's max-norm
constexpr double QP_FTB = 0.995;
constexpr double QP_ALPHA_MIN = 1e-12;
constexpr double QP_ABSENT = 1e300;  // sentinel for a missing bound side inside the workspace

constexpr int qp_even_c(int n) { return (n + 1) & ~1; }
constexpr int qp_al8(int n) { return (n + 7) & ~7; }   // block sizes in the stage record: multiples of 64 bytes

// Workspace of one stage, in doubles.  Every block starts on a 64-byte boundary and the record on a 128-byte one: stores are
// what this kernel's memory stream pays most for (scripts/mem_pipeline_probe.hip, profiles/r03_mem_pipeline_probe_stores.txt:
// a store costs 2-3 x its bytes in read-equivalents, an 8-byte store 16 bytes away from its neighbour a read-modify-write
// of the sectors it touches), so everything a sweep writes is contiguous, sector-aligned and as few instructions as possible.
//   rows r = 0..NRT-1: [x box (NX) | torque (NQ) | collision (MR) | safe-set (1)]; "general" rows are the last NRC
template <int NQ> struct QpLayout {
    static constexpr int NX = 2 * NQ, NU = NQ, NZ = 3 * NQ;
    static constexpr int NZP = qp_even_c(NZ), NQP = qp_even_c(NQ), WS2 = qp_even_c(NX + 1);
    static constexpr int LC0 = WS2;                    // factor rows: first column of L^-1
    static constexpr int KS = qp_even_c(LC0 + NQ);     // lanes that own a column of [G | rho | . | I] in the factorisation sweep
    // The factor block as stored, column by column, nothing padded: the NX + 1 columns of [W | w] (NQ doubles each), then the
    // columns of L^-1 from their diagonal entry down (column j at w_coff(j), NQ - j doubles) -- 100 doubles at NQ = 6 where the
    // rectangular NQ x KS image of v1-v9 took 120, and this block is written once and read three times per iteration.
    static constexpr int WR = NX + 1, LOFF = NQ * WR, NWP = qp_even_c(LOFF + NQ * (NQ + 1) / 2);
    __host__ __device__ static constexpr int w_coff(int j) { return LOFF + j * NQ - j * (j - 1) / 2; }
    int MR, MRP, NRC, NRT;
    // image of the factorisation sweep (copied verbatim into LDS), offsets relative to oIMG
    int iTT, iGT, iGN, iHQQ, iGZ, iB, iSC, nIMG;
    int nJ;                                                     // doubles of the image's Jacobian part [Tt | Gt | gn]
    int nF;                                                     // ... and of what the forward sweeps fetch: [Tt | Gt | gn | b | scalars]
    int oIMG, oW, oSL, oPART, oWC, oR0, oR1, oR2, oCZA, oCZN;
    int oZ, oZN, oA12;
    int stride;
    __host__ __device__ explicit QpLayout(int n_rows) {
        MR = n_rows;
        MRP = qp_even_c(MR);
        NRC = NQ + MR + 1;
        NRT = NX + NRC;
        int i = 0;
        iTT = i; i += NZ * NQP;            // torque rows transposed: Tt[c][r]
        iGT = i; i += NQ * MRP;            // collision rows transposed (q columns): Gt[ix][r]
        iGN = i; i += NX;                  // safe-set row
        nJ = i;                            // (even: NQP, MRP are, and NX = 2 NQ)
        // (v12: the defect and the stage scalars sit right behind the Jacobian, so that the forward sweeps get them with the pieces
        //  they fetch anyway -- the same number of load instructions -- instead of from a block of (b_i, soft weight) pairs of their
        //  own: 24 doubles per stage and forward sweep, one load instruction per stage, and the set-up's store of that block)
        iB = i; i += NX;                   // dynamics defect
        iSC = i; i += 4;                   // [Huu diagonal, LM on the v diagonal, soft weight, b != 0]
        nF = i;                            // (even)
        iHQQ = i; i += qp_even_c(NQ * NQ); // cost Hessian (q block) + LM
        iGZ = i; i += NZP;                 // cost gradient
        nIMG = i;                          // (even)
        int o = 0;
        // (v11: the constraint Jacobian is stored ONCE, in the image's transposed layout; v1-v10 kept a second, row-major copy
        //  for the forward sweeps -- 160 of a stage's 880 doubles at NQ = 6 -- which now transpose the image's pieces on their way
        //  into LDS instead)
        oIMG = o; o += qp_al8(nIMG);
        oW = o; o += qp_al8(NWP);         // factor block: columns of [W | w], then the packed columns of L^-1
        oSL = o; o += 16;                  // [soft weight, b != 0, ., . | setup partials: R0, sum lambda t, count, node-0 rows
        oPART = oSL + 4;                   //  infeasible | the corrector's w (B2 -> F2; one whole 64-byte sector)]
        oWC = oSL + 8;
        oR0 = o; o += qp_al8(NRT * 2);     // per row, as arrays of pairs: [lo, hi]
        oR1 = o; o += qp_al8(NRT * 2);     //   [t_l, t_u]        (the soft row has no upper side: its slack lives in t_u)
        oR2 = o; o += qp_al8(NRT * 2);     //   [lambda_l, lambda_u]
        oCZA = o; o += 32;                 // per row lane: c.z_aff (F1 -> F2, B1) and c.z+ (F2 -> B1), each its own contiguous array
        oCZN = o; o += 32;                 //   (as the two halves of a pair, each sweep's store was a read-modify-write of every sector)
        oZ = o; o += qp_al8(NZ);
        oZN = o; o += qp_al8(NZ);
        // (v12: no stored predictor gradient.  The corrector's costate recursion is linear in its gradient gh0 + sigma mu a1 + cw a2,
        //  and the part that belongs to gh0 is the predictor costate B1 has already computed -- its w is a column of the factor block.
        //  B2 propagates the DIFFERENCE, driven by sigma mu a1 + cw a2 alone, and F2 adds the two w: 18 doubles fewer written by B1
        //  and read by B2 per stage, no P b folded into anything.)
        oA12 = o; o += qp_al8(2 * NZ);     // pairs (C^T e1, C^T e2): the corrector's costate increment is driven by sigma mu a1 + cw a2 (F1 -> B2)
        static_assert(NQ <= 8, "the corrector's w shares a 16-double block with the stage scalars");
        stride = (o + 15) & ~15;
    }
    __host__ __device__ size_t per_instance(int N) const { return (size_t)stride * (N + 1); }
};

typedef double dbl2 __attribute__((ext_vector_type(2)));

// Cache policy of the stage-workspace accesses.  The workspace is a stream: every block is touched once per sweep by one
// wavefront and comes round again a millisecond later, long after the 4 MB L2 of its XCD has turned over.  Whether non-temporal
// (`nt`) accesses pay depends on the workspace's size against the 256 MB Infinity Cache, so k_qp_ipm is built both ways
// (template parameter NT: nt on its wide loads and on the factorisation sweep's wide stores) and the engine picks PER LAUNCH
// (engine.hip: qp_nt_threshold; measurements at the kernel, below).  SMPC_NT_MASK is the build-time override of rounds 2-3, kept
// for experiments on the remaining access classes (narrow 8-byte nt loads measured +6 %, k_qp_setup's stores no gain); its
// default 0 leaves everything but the NT template's accesses plain.
#ifndef SMPC_NT_MASK
#define SMPC_NT_MASK 0x0
#endif
template <int BIT, class T> __device__ __forceinline__ T ld_ws(const T* p) {
    if constexpr ((SMPC_NT_MASK >> BIT) & 1) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int BIT, class T> __device__ __forceinline__ void st_ws(T v, T* p) {
    if constexpr ((SMPC_NT_MASK >> BIT) & 1) __builtin_nontemporal_store(v, p);
    else *p = v;
}
// bit 0: wide loads of k_qp_ipm (pieces, row pairs)   bit 1: wide stores of the factorisation sweep (t, lambda, factor block)
// bit 2: narrow loads of k_qp_ipm                     bit 3: narrow stores of k_qp_ipm      bit 4: k_qp_setup (record load, stores)
template <class T> __device__ __forceinline__ T ldnt(const T* p) { return ld_ws<0>(p); }
template <class T> __device__ __forceinline__ void stnt_b1(T v, T* p) { st_ws<1>(v, p); }
template <class T> __device__ __forceinline__ T ldnt_s(const T* p) { return ld_ws<2>(p); }
template <class T> __device__ __forceinline__ void stnt_s(T v, T* p) { st_ws<3>(v, p); }
template <class T> __device__ __forceinline__ T ldnt_su(const T* p) { return ld_ws<4>(p); }
template <class T> __device__ __forceinline__ void stnt_su(T v, T* p) { st_ws<4>(v, p); }

// wave-local hand-off through LDS: LDS operations of one wave execute in issue order, so all that is needed is that the
// compiler neither reorders nor caches them across this point
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}

// "These values are needed HERE": an empty asm that takes loaded values as operands makes the compiler issue all their loads
// before this point and wait for them once.  Left to itself (and short of registers) it sinks every LDS read next to its
// use, and a lone wavefront then pays one round trip through LDS per read.
__device__ __forceinline__ void hold(double& a, double& b, double& c, double& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
template <int H> __device__ __forceinline__ void hold_rows2(dbl2 (&a)[H], dbl2 (&b)[H], double (&s)[5]) {
    static_assert(H == 3 || H == 4, "NQP / 2 of the built sizes");
    if constexpr (H == 3)
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(s[0]), "+v"(s[1]), "+v"(s[2]),
                          "+v"(s[3]), "+v"(s[4]));
    else
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(s[0]),
                          "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]));
}
// four padded rows (H 16-byte pieces each) and nine scalars in ONE statement: loads that are operands of different asm
// statements may be scheduled between them, and the point is a single wait
template <int H> __device__ __forceinline__ void hold_rows4(dbl2 (&a)[H], dbl2 (&b)[H], dbl2 (&c)[H], dbl2 (&d)[H], double (&s)[9]) {
    static_assert(H == 3 || H == 4, "NQP / 2 of the built sizes");
    if constexpr (H == 3)
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]),
                          "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]),
                          "+v"(s[6]), "+v"(s[7]), "+v"(s[8]));
    else
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(c[0]),
                          "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(s[0]), "+v"(s[1]),
                          "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]), "+v"(s[8]));
}
// reductions over the 32 lanes of one half-wave
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

// ---- per-row IPM algebra: one lane owns one two-sided row (lo <= c.z <= hi), everything in its registers -----------------
// Absent sides carry t = 1, lambda = 0 and are masked out by selects (no divergent branches: the lanes of a half-wave own
// rows of every kind).  The soft row (safe-set constraint relaxed by an L1-penalised slack, eliminated in closed form)
// has no upper side; its slack and the slack's direction use the upper side's slots (tu, dtu).
struct QpRow { double lo, hi, tl, tu, ll, lu; };
struct QpDir { double dtl, dll, dtu, dlu; };

// one side: gap = distance of the trial point to the bound, gap_a the same for z_aff; returns the largest -d/v in *rr.
// WANT_E: also the side's share of e1, e2, where the gradient coefficient of the row is e0 + sigma mu e1 + cw e2
template <bool WANT_E>
__device__ __forceinline__ void qp_side_dir(bool has, double sgn, double gap, double gap_a, double t, double l, double sigmu,
                                            double cw, double* dt_o, double* dl_o, double* rr, double* S1, double* S2,
                                            double* e1, double* e2) {
    const double ls = has ? l : 1.0;
    const double r = fast_rcp(t * ls);      // one reciprocal serves 1/t and 1/lambda
    const double it = ls * r, il = t * r;
    const double dta = gap_a - t;
    const double dla = -l * dta * it - l;
    const double ct = cw * dta * dla;       // Mehrotra second-order term (cw = 0: predictor)
    const double dtt = gap - t;
    const double dll = (sigmu - ct - l * dtt) * it - l;
    const double dt_ = has ? dtt : 0.0, dl_ = has ? dll : 0.0;
    *dt_o = dt_;
    *dl_o = dl_;
    *rr = fmax(*rr, fmax(-dt_ * it, -dl_ * il));
    *S1 += l * dt_ + t * dl_;
    *S2 += dl_ * dt_;
    if (WANT_E) {
        const double m = has ? sgn * it : 0.0;
        *e1 += m;
        *e2 -= dta * dla * m;
    }
}
// directions of one row for the trial value czn = c.z+, ratio-test term and the terms of sum(lambda t)(alpha)
template <bool WANT_E>
__device__ __forceinline__ QpDir qp_row_dir(const QpRow& s, bool soft, double wsoft, double czn, double sigmu, double cw,
                                            double cza, double* rr, double* S1, double* S2, double* e1, double* e2) {
    QpDir o;
    const bool hasl = s.lo > -QP_ABSENT, hasu = s.hi < QP_ABSENT;
    const double lo = hasl ? s.lo : 0.0, hi = hasu ? s.hi : 0.0;
    if (WANT_E) { *e1 = 0.0; *e2 = 0.0; }
    qp_side_dir<WANT_E>(hasl && !soft, -1.0, czn - lo, cza - lo, s.tl, s.ll, sigmu, cw, &o.dtl, &o.dll, rr, S1, S2, e1, e2);
    qp_side_dir<WANT_E>(hasu, 1.0, hi - czn, hi - cza, s.tu, s.lu, sigmu, cw, &o.dtu, &o.dlu, rr, S1, S2, e1, e2);
    if (soft) {
        const double tl = s.tl, ll = s.ll, sl = s.tu;
        const double nu = wsoft - ll;
        const double ill = fast_rcp(ll), inu = fast_rcp(nu);
        const double deff = fast_rcp(tl * ill + sl * inu);
        const double dla = -deff * (cza - lo);
        const double dta = -tl * dla * ill - tl;
        const double dsa = sl * dla * inu - sl;
        const double ct = cw * dta * dla;
        const double cs2 = -cw * dsa * dla;
        const double dl = -deff * (czn - lo + (sigmu - cs2) * inu - (sigmu - ct) * ill);
        const double dtl = (sigmu - ct - tl * dl) * ill - tl;
        const double dsl = (sigmu - cs2 + sl * dl) * inu - sl;
        o.dll = dl;
        o.dtl = dtl;
        o.dtu = dsl;
        *rr = fmax(*rr, fmax(fmax(-dtl / tl, -dl * ill), fmax(-dsl / sl, dl * inu)));
        *S1 += ll * dtl + tl * dl + nu * dsl - sl * dl;
        *S2 += dl * dtl - dl * dsl;
        if (WANT_E) {
            *e1 = deff * (inu - ill);
            *e2 = deff * dla * (dsa * inu + dta * ill);
        }
    }
    return o;
}
// gradient coefficient e_r (returned) and barrier weight D_r
__device__ __forceinline__ double qp_row_coeff(const QpRow& s, bool soft, double wsoft, double sigmu, double cw, double cza,
                                               double* Dr) {
    const bool hasl = s.lo > -QP_ABSENT, hasu = s.hi < QP_ABSENT;
    const double lo = hasl ? s.lo : 0.0, hi = hasu ? s.hi : 0.0;
    double e, dsum;
    {
        const double itl = fast_rcp(s.tl), itu = fast_rcp(s.tu);
        const double dtal = cza - lo - s.tl, dtau = hi - cza - s.tu;
        const double ctl = cw * dtal * (-s.ll * dtal * itl - s.ll);
        const double ctu = cw * dtau * (-s.lu * dtau * itu - s.lu);
        const double dl = s.ll * itl, du = s.lu * itu;
        const double el = -s.ll - dl * lo - (sigmu - ctl) * itl;
        const double eu = s.lu - du * hi + (sigmu - ctu) * itu;
        e = ((hasl && !soft) ? el : 0.0) + (hasu ? eu : 0.0);
        dsum = ((hasl && !soft) ? dl : 0.0) + (hasu ? du : 0.0);
    }
    if (soft) {
        const double tl = s.tl, ll = s.ll, sl = s.tu;
        const double nu = wsoft - ll;
        const double ill = fast_rcp(ll), inu = fast_rcp(nu);
        const double deff = fast_rcp(tl * ill + sl * inu);
        const double dla = -deff * (cza - lo);
        const double dta = -tl * dla * ill - tl;
        const double dsa = sl * dla * inu - sl;
        const double ct = cw * dta * dla;
        const double cs2 = -cw * dsa * dla;
        e = -ll + deff * (-lo + (sigmu - cs2) * inu - (sigmu - ct) * ill);
        dsum = deff;
    }
    *Dr = dsum;
    return e;
}
__device__ __forceinline__ double qp_row_comp(const QpRow& s, bool soft, double wsoft) {  // lambda t (+ nu s)
    double acc = s.ll * s.tl + s.lu * s.tu;          // absent sides hold lambda = 0
    if (soft) acc = s.ll * s.tl + (wsoft - s.ll) * s.tu;
    return acc;
}

// diagnostic build (-DQP_PROFILE, scripts/qp_phase_profile.py): per-phase shader-clock sums over all half-waves
#ifdef QP_PROFILE
__device__ unsigned long long g_qp_prof[16];
#define QPT(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define QPT(i) do { } while (0)
#endif


// =========================================================================================================================
// k_qp_setup: stage records + initial interior point, one half-wave per (instance, stage); a block = the EV_TILE nodes of one
// interleaved tile of linearisation records (device_model.hpp), loaded by all its threads and taken apart in LDS
// =========================================================================================================================
template <int NQ, int MRT>
__global__ __launch_bounds__(32 * EV_TILE) void k_qp_setup(const smpc_problem_desc* __restrict__ D, int B, int N,
                                                  const double* __restrict__ x0, const double* __restrict__ xg,
                                                  const double* __restrict__ ug, const double* __restrict__ pp,
                                                  const double* __restrict__ lo_st, const double* __restrict__ hi_st,
                                                  const double* __restrict__ zl_st, const double* __restrict__ ev, double* __restrict__ ws_all,
                                                  long bnd_stride, const uint8_t* __restrict__ active) {
    using LyT = QpLayout<NQ>;
    constexpr int NX = 2 * NQ, NU = NQ, NZ = 3 * NQ, NZP = LyT::NZP, NQP = LyT::NQP, NL = 32;
    constexpr int MAXRC = NQ + (MRT >= 0 ? MRT : SMPC_MAX_ROWS) + 1;
    constexpr int EV_PAD = qp_even_c(EV_D);
    const int hl = threadIdx.x & 31, half = threadIdx.x >> 5;
    constexpr int O_C = EV_PAD, O_LO = O_C + MAXRC * NZP, O_HI = O_LO + NL, O_E = O_HI + NL, O_Z0 = O_E + NL, O_GZ = O_Z0 + NL,
                  O_B = O_GZ + NZP, HALF_D = O_B + NX;
    __shared__ __attribute__((aligned(16))) double smem[EV_TILE * HALF_D];
    static_assert(EV_TILE % 2 == 0 && EV_TILE <= 32, "one half-wave per node of a tile");
    {
        // the tile: EV_TILE * EV_D doubles, contiguous; piece p holds element (2p) / EV_TILE of nodes (2p) % EV_TILE and the next one
        const dbl2* s2 = reinterpret_cast<const dbl2*>(ev + (size_t)blockIdx.x * (EV_TILE * EV_D));
        for (int p2 = threadIdx.x; p2 < EV_TILE * EV_D / 2; p2 += 32 * EV_TILE) {
            const dbl2 v = ldnt_su(s2 + p2);
            const int f = (2 * p2) / EV_TILE, n = (2 * p2) % EV_TILE;
            smem[n * HALF_D + f] = v.x;
            smem[(n + 1) * HALF_D + f] = v.y;
        }
    }
    __syncthreads();
    const long pi = (long)EV_TILE * blockIdx.x + half;
    if (pi >= (long)B * (N + 1)) return;
    const int b = (int)(pi / (N + 1)), k = (int)(pi - (long)b * (N + 1));
    if (active && !active[b]) return;   // (policy layer: this instance does not step its controller; k_qp_ipm skips it too)
    const LyT Ly(MRT >= 0 ? MRT : D->n_rows);
    const int MR = Ly.MR, MRP = Ly.MRP, NRC = Ly.NRC, NRT = Ly.NRT;
    const int rT0 = NX, rC0 = NX + NQ, rNN = NX + NQ + MR;
    double* w = ws_all + (size_t)b * Ly.per_instance(N) + (size_t)k * Ly.stride;
    const double dt = D->dt, cB = 0.5 * dt * dt;

    double* const sEV = smem + half * HALF_D;
    double* const sC = sEV + O_C;
    double* const sLO = sEV + O_LO;
    double* const sHI = sEV + O_HI;
    double* const sE = sEV + O_E;
    double* const sZ0 = sEV + O_Z0;
    double* const sGZ = sEV + O_GZ;
    double* const sB = sEV + O_B;

    const double* xk = xg + ((size_t)b * (N + 1) + k) * NX;
    const double* pk = pp + ((size_t)b * (N + 1) + k) * SMPC_NP;
    const bool last = (k == N);
    // z = 0 except the fixed dx_0
    sZ0[hl] = (k == 0 && hl >= NU && hl < NZ) ? x0[(size_t)b * NX + hl - NU] - xk[hl - NU] : 0.0;
    lds_fence();
    const smpc_node_eval& e = *reinterpret_cast<const smpc_node_eval*>(sEV);
    const double cs = last ? D->cost_scale_term : D->cost_scale_stage;
    const double lm = last ? D->lm_term : D->lm_stage;
    const bool reach = D->cost_kind == SMPC_COST_REACH;
    bool nn_on = false;
    if (D->nn_mode != SMPC_NN_NONE && k >= 1 && (D->nn_mode == SMPC_NN_ALL || last)) nn_on = pk[4] > 0.0;
    double wsoft = nn_on ? (last ? D->nn_soft_e : D->nn_soft_run) : -1.0;
    if (zl_st && wsoft >= 0.0) {
        wsoft = zl_st[k];    // cost_set(k, 'zl', .) on a row the formulation made soft
        // A zero weight leaves the row without effect on the optimum (its slack absorbs it at no cost) -- and would start the
        // interior point ON the boundary of the multiplier's box [0, 0].  The reference only writes zl = 0 together with
        // p[4] = -1 (controller.py:455-460); either way the row is absent.
        if (wsoft == 0.0) { nn_on = false; wsoft = -1.0; }
    }

    // general rows, row-major [torque | collision | safe-set] x [u q v]
    for (int el = hl; el < NRC * NZP; el += 32) {
        const int r = el / NZP, c = el - r * NZP;
        double v = 0.0;
        if (c < NZ) {
            if (r < NQ) {
                if (!last) v = c < NQ ? e.M[r * NQ + c] : (c < 2 * NQ ? e.dtau_dq[r * NQ + c - NQ] : e.dtau_dv[r * NQ + c - 2 * NQ]);
            } else if (r < NQ + MR) {
                if (c >= NU && c < NU + NQ) v = e.row_grad[(r - NQ) * NQ + c - NU];
            } else {
                if (c >= NU && nn_on) v = e.nn_grad[c - NU];
            }
        }
        sC[el] = v;
    }
    if (hl < NZP) {
        double g = 0.0;
        if (reach && hl < NZ) {
            if (hl < NU) g = last ? 0.0 : cs * 2.0 * D->R * ug[((size_t)b * N + k) * NU + hl];
            else if (hl < NU + NQ) g = cs * e.cost_grad_q[hl - NU];
        }
        sGZ[hl] = g;
    }
    if (hl < NX) {
        double bb = 0.0;
        if (!last) {
            const double* xn = xk + NX;
            const int i = hl < NQ ? hl : hl - NQ;
            const double uk = ug[((size_t)b * N + k) * NU + i];
            bb = hl < NQ ? xk[i] + dt * xk[NQ + i] + cB * uk - xn[i] : xk[NQ + i] + dt * uk - xn[NQ + i];
        }
        sB[hl] = bb;
    }
    {
        const int r = hl;   // rows >= NRT stay absent on both sides
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
        } else if (r == rNN) {
            if (nn_on) lo = 0.0 - e.nn_val;
        }
        sLO[r] = lo;
        sHI[r] = hi;
    }
    lds_fence();
    const double bmax = half_max(hl < NX ? fabs(sB[hl]) : 0.0);
    const double bflag = bmax > 0.0 ? 1.0 : 0.0;

    // ---- static blocks out ---------------------------------------------------------------------------------------------
    if (hl < NRT) stnt_su(dbl2{sLO[hl], sHI[hl]}, reinterpret_cast<dbl2*>(w + Ly.oR0) + hl);
    double* img = w + Ly.oIMG;
    for (int el = hl; el < NZ * NQP; el += 32) {
        const int c = el / NQP, r = el - c * NQP;
        stnt_su(r < NQ ? sC[r * NZP + c] : 0.0, img + Ly.iTT + el);
    }
    for (int el = hl; el < NQ * MRP; el += 32) {
        const int ix = el / MRP, r = el - ix * MRP;
        stnt_su(r < MR ? sC[(NQ + r) * NZP + NU + ix] : 0.0, img + Ly.iGT + el);
    }
    if (hl < NX) stnt_su(sC[(NQ + MR) * NZP + NU + hl], img + Ly.iGN + hl);
    for (int el = hl; el < qp_even_c(NQ * NQ); el += 32) {
        const int i = el / NQ, j = el - i * NQ;
        stnt_su(el < NQ * NQ ? (reach ? cs * e.cost_hess_qq[el] : 0.0) + (i == j ? lm : 0.0) : 0.0, img + Ly.iHQQ + el);
    }
    if (hl < NZP) stnt_su(sGZ[hl], img + Ly.iGZ + hl);
    if (hl < NX) stnt_su(sB[hl], img + Ly.iB + hl);
    if (hl < 4) {
        const double huu = (reach && !last ? cs * 2.0 * D->R : 0.0) + lm;
        stnt_su(hl == 0 ? huu : (hl == 1 ? lm : (hl == 2 ? wsoft : bflag)), img + Ly.iSC + hl);
    }

    // ---- initial slacks / multipliers ----------------------------------------------------------------------------------
    double r0_loc = 0.0, mu_acc = 0.0;
    int cnt = 0;
    {
        const int r = hl;
        double tl = 1.0, ll = 0.0, tu = 1.0, lu = 0.0;
        if (r < NRT) {
            double cz, cn = 1.0;
            if (r < NX) cz = sZ0[NU + r];
            else {
                cz = 0.0;
                cn = 0.0;
                for (int c = 0; c < NZ; c++) {
                    const double cv = sC[(r - NX) * NZP + c];
                    cz = fma(cv, sZ0[c], cz);
                    cn = fmax(cn, fabs(cv));
                }
            }
            const bool soft = (r == rNN) && wsoft >= 0.0;
            // slack floor of the starting point: a hard row starts at least QP_THR_HARD |c|_inf inside its bound (a distance to
            // the boundary measured in the variables, not in the row's units; see the oracle's qp_ipm for what steep and flat
            // rows do under one absolute floor); the soft row keeps the absolute floor of its slack variable
            const double thr = soft ? QP_THR : QP_THR_HARD * (cn > 0.0 ? cn : 1.0);
            if (sLO[r] > -QP_ABSENT) {
                const double s0 = soft ? QP_THR : 0.0;
                const double slack = cz + s0 - sLO[r];
                tl = fmax(slack, thr);
                ll = D->qp_mu0 / tl;
                if (soft) ll = fmin(ll, 0.5 * wsoft);
                r0_loc = fmax(r0_loc, fabs(slack - tl));
                cnt += soft ? 2 : 1;
                mu_acc += ll * tl;
                if (soft) { mu_acc += (wsoft - ll) * s0; tu = s0; }   // the slack rides in the unused upper side
            }
            if (sHI[r] < QP_ABSENT) {
                const double slack = sHI[r] - cz;
                tu = fmax(slack, thr);
                lu = D->qp_mu0 / tu;
                r0_loc = fmax(r0_loc, fabs(slack - tu));
                cnt += 1;
                mu_acc += lu * tu;
            }
        }
        if (r < NRT) {
            stnt_su(dbl2{tl, tu}, reinterpret_cast<dbl2*>(w + Ly.oR1) + r);
            stnt_su(dbl2{ll, lu}, reinterpret_cast<dbl2*>(w + Ly.oR2) + r);
        }
        stnt_su(0.0, w + Ly.oCZA + hl);   // (32 entries each: one per lane)
        stnt_su(0.0, w + Ly.oCZN + hl);
        sE[r] = -(ll - lu);
    }
    if (hl < 4) stnt_su(hl == 0 ? wsoft : (hl == 1 ? bflag : 0.0), w + Ly.oSL + hl);
    if (hl < 8) stnt_su(0.0, w + Ly.oWC + hl);
    if (hl < NZ) { stnt_su(sZ0[hl], w + Ly.oZ + hl); stnt_su(sZ0[hl], w + Ly.oZN + hl); }
    lds_fence();
    // stationarity residual at the initial point (pi = 0): g - C^T (ll - lu); dx_0 does not enter (no cost cross term)
    if (hl < NZ && !(k == 0 && hl >= NU) && !(last && hl < NU)) {
        double a = sGZ[hl];
        for (int r = 0; r < NRC; r++) a = fma(sC[r * NZP + hl], sE[NX + r], a);
        if (hl >= NU) a += sE[hl - NU];
        r0_loc = fmax(r0_loc, fabs(a));
    }
    if (!last && hl < NX) {
        double ax = 0.0;  // dynamics defect of the initial point: next dx is 0
        if (k == 0) ax = hl < NQ ? sZ0[NU + hl] + dt * sZ0[NU + NQ + hl] : sZ0[NU + hl];
        r0_loc = fmax(r0_loc, fabs(ax + sB[hl]));
    }
    // collision rows at node 0 (kept by the reference when --noise == 0, controller.py:77-79): x_0 is pinned, so the linearised
    // rows are constants of the QP -- one outside its bounds makes the QP infeasible (the instance reports QP failure)
    double inf0 = 0.0;
    if (k == 0 && D->rows_at_node0 && hl >= rC0 && hl < rNN) {
        const smpc_row& row = D->rows[hl - rC0];
        double v = e.row_val[hl - rC0];
        for (int c = 0; c < NZ; c++) v = fma(sC[(hl - NX) * NZP + c], sZ0[c], v);
        if ((fabs(row.lb) < SMPC_INF && v < row.lb - D->qp_tol) || (fabs(row.ub) < SMPC_INF && v > row.ub + D->qp_tol)) inf0 = 1.0;
    }
    const double R0 = half_max(r0_loc), musum = half_sum(mu_acc), cntsum = half_sum((double)cnt), infs = half_max(inf0);
    if (hl < 4) w[Ly.oPART + hl] = hl == 0 ? R0 : (hl == 1 ? musum : (hl == 2 ? cntsum : infs));
}

#ifndef QP_WAVES_PER_EU
#define QP_WAVES_PER_EU 2
#endif

// =========================================================================================================================
// k_qp_ipm
// =========================================================================================================================
// NT: the wide accesses of the stage workspace -- 16-byte loads of every sweep, the factorisation sweep's wide stores -- are
// non-temporal.  The workspace is a stream (every block is touched once per sweep by one wavefront and comes round again long after
// its XCD's 4 MB L2 has turned over), so for a workspace far larger than the 256 MB Infinity Cache `nt` is a plain win: C1's loop at
// 8192 / 16384 / 32768 / 65536 instances -5 / -12 / -14 / -16 % per step (round 4, A/B in one session).  At the headline 4096 (three
// sub-batch workspaces of 244 MB) it LOSES 2-3 %: part of a sub-batch's workspace survives in the Infinity Cache between a
// stage builder and the first sweeps, and between sweeps, and nt gives that up.  The engine picks by the handle's workspace size
// (engine.hip: qp_nt_threshold); results do not depend on it.
template <int NQ, int MRT, bool NT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(QP_WAVES_PER_EU, QP_WAVES_PER_EU))) void k_qp_ipm(
    const smpc_problem_desc* __restrict__ D, int B, int N, const double* __restrict__ x0, const double* __restrict__ xg,
    const double* __restrict__ ug, double* __restrict__ ws_all, double* __restrict__ x_out, double* __restrict__ u_out,
    int32_t* __restrict__ status, int32_t* __restrict__ qp_iter, const int32_t* __restrict__ order,
    int32_t* __restrict__ last_iter, unsigned long long* __restrict__ wstat, const uint8_t* __restrict__ active,
    int32_t* __restrict__ it_hist) {
    using LyT = QpLayout<NQ>;
    constexpr int NX = 2 * NQ, NU = NQ, NZ = 3 * NQ, NZP = LyT::NZP, NQP = LyT::NQP, WS2 = LyT::WS2, NL = 32,
                  LC0 = LyT::LC0, KS = LyT::KS, NWP = LyT::NWP;
    // (the wide workspace accesses of this kernel: these two shadow the build-time-selectable helpers of the same names)
    auto ldnt = [](auto* p) {
        if constexpr (NT || (SMPC_NT_MASK & 1)) return __builtin_nontemporal_load(p);
        else return *p;
    };
    auto stnt_b1 = [](auto v, auto* p) {
        if constexpr (NT || (SMPC_NT_MASK & 2)) __builtin_nontemporal_store(v, p);
        else *p = v;
    };
    // load-balance probe (smpc_get_qp_wave_stats): two reads of the constant 100 MHz clock per half-wave; the first one is
    // parked in LDS (the kernel has no register to spare)
    __shared__ unsigned long long s_tbegin[2];
    if (wstat && (threadIdx.x & 31) == 0) s_tbegin[threadIdx.x >> 5] = __builtin_amdgcn_s_memrealtime();
    constexpr int MR_MAX = MRT >= 0 ? MRT : SMPC_MAX_ROWS, MRP_MAX = qp_even_c(MR_MAX);
    constexpr int NRC_MAX = NQ + MR_MAX + 1;
    static_assert(MRT < 0 || NX + NRC_MAX <= NL, "one lane per constraint row");
    static_assert(KS <= NL, "one lane per column of [G | rho | I]");
#ifndef QP_P_INPLACE
#define QP_P_INPLACE 0      // (measured, round 6: the eighth 7-DoF wavefront it buys makes C4 SLOWER, 20.82 -> 21.31 ms per step)
#endif
    constexpr bool P_INPLACE = QP_P_INPLACE != 0;
    constexpr int NTRI_U = NQ * (NQ + 1) / 2, NTRI_X = NX * (NX + 1) / 2;
    // first index of the q-q corner's elements in the x-x index table (below): they need the cost Hessian and the collision rows
    constexpr int TRI_Q0 = P_INPLACE ? NTRI_X - NTRI_U : 0;
    constexpr int IMG_MAX = NZ * NQP + NQ * MRP_MAX + NX + qp_even_c(NQ * NQ) + NZP + NX + 4;
    constexpr int IMG_PF = (IMG_MAX / 2 + 31) / 32;
    constexpr int CST_MAX = NRC_MAX * NZP;                                       // row-major image of the general rows in LDS
    constexpr int NF_MAX = NZ * NQP + NQ * MRP_MAX + NX + NX + 4, CST_PF = (NF_MAX / 2 + 31) / 32;   // ... fetched as the image's [Tt | Gt | gn | b | scalars]
    constexpr int W_N2 = NWP / 2, WST_PF = (W_N2 + 31) / 32;                 // ... and the factor block
    // (staged in LDS by the other sweeps, in the buffers that only the factorisation sweep uses)
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
    if (active && !active[b]) {
        // the policy layer masks out instances that do not step their controller (dead, or following a backup trajectory):
        // their state can be anywhere, and an infeasible QP running to its iteration cap would set the launch time
        if (hl == 0) {
            status[b] = SMPC_STATUS_SUCCESS;
            if (qp_iter) qp_iter[b] = 0;
            if (last_iter) last_iter[b] = 0;
            if (it_hist) atomicAdd(&it_hist[0], 1);
        }
        return;
    }
    const LyT Ly(MRT >= 0 ? MRT : D->n_rows);
    const int MR = Ly.MR, MRP = Ly.MRP, NRC = Ly.NRC, NRT = Ly.NRT;
    const int rT0 = NX, rC0 = NX + NQ, rNN = NX + NQ + MR;
    double* const ws = ws_all + (size_t)b * Ly.per_instance(N);
    const double dt = D->dt, cB = 0.5 * dt * dt;
    const int img_n2 = Ly.nIMG >> 1;   // 16-byte pieces of the image
    const int c_n2 = Ly.nF >> 1;            // ... and of its part [Tt | Gt | gn | b | scalars] (what the forward sweeps fetch)

    // ---- LDS: one region per half-wave ---------------------------------------------------------------------------------
    // [image | D | E | -- buffers of the factorisation sweep only: TD GD Lambda G Wt P P -- | vectors]
    constexpr int O_D = IMG_MAX, O_E = O_D + NL, O_TD = O_E + NL, O_GD = O_TD + NZ * NQP, O_LAM = O_GD + NQ * MRP_MAX,
                  O_G = O_LAM + qp_even_c(NQ * NQ), O_WT = O_G + NQ * WS2, O_PA = O_WT + (NX + 1) * NQP,
                  O_PB2 = O_PA + (P_INPLACE ? 0 : NX * NX),      // (P_k is built over P_{k+1}: see the index table below)
                  O_PEND = O_PB2 + NX * NX, O_STG = O_TD + CST_MAX + 2 * NWP,   // (the forward sweeps stage their blocks over the B1-only buffers)
                  O_PVA = O_PEND > O_STG ? O_PEND : O_STG, O_PVB = O_PVA + NX, O_PB = O_PVB + NX,
                  O_ZU = O_PB + NX, O_XB = O_ZU + NQP, O_RHO = O_XB + 2 * NX, O_WV = O_RHO + NQP, HALF_D = O_WV + NQP;
    __shared__ __attribute__((aligned(16))) double smem[2 * HALF_D];
    __shared__ unsigned char triUi[NTRI_U], triUj[NTRI_U], triXi[NTRI_X], triXj[NTRI_X];
    double* const sIMG = smem + half * HALF_D;
    double* const sTT = sIMG + Ly.iTT;
    double* const sGT = sIMG + Ly.iGT;
    double* const sGN = sIMG + Ly.iGN;
    double* const sHQQ = sIMG + Ly.iHQQ;
    double* const sGZ = sIMG + Ly.iGZ;
    double* const sB = sIMG + Ly.iB;
    double* const sSC = sIMG + Ly.iSC;
    double* const sTD = sIMG + O_TD;
    double* const sGD = sIMG + O_GD;
    double* const sD = sIMG + O_D;
    double* const sE = sIMG + O_E;
    double* const sLam = sIMG + O_LAM;
    double* const sG = sIMG + O_G;
    double* const sWT = sIMG + O_WT;
    // B1 lays the factor block out as it is stored before storing it -- over Lambda and G, which are dead by then: the Cholesky
    // factor sits in registers and every lane has read its column of [G | rho] (round 5: a buffer of NX * NX doubles of its own until
    // then; without it the 7-DoF block is 21.6 KB instead of 24.7 and seven wavefronts fit a CU instead of six)
    double* const sWI = sIMG + O_LAM;
    static_assert(CST_MAX + 2 * NWP <= O_PVA - O_TD, "staging area of the forward sweeps");
    static_assert(NWP + NQ <= qp_even_c(NQ * NQ) + NQ * WS2, "the stored image of the factor block fits over Lambda and G");
    double* const sCst = sIMG + O_TD;                      // forward sweeps only: general rows ...
    double* const sWstA = sIMG + O_TD + CST_MAX;           // ... and the factor block, double-buffered (B2: single)
    double* const sPB = sIMG + O_PB;
    double* const sZU = sIMG + O_ZU;
    double* const sRho = sIMG + O_RHO;
    double* const sWv = sIMG + O_WV;

    // (both halves write the same table values: a lone half must not depend on its twin)
    for (int e = hl; e < NTRI_U; e += 32) {
        int i = 0, rem = e;
        while (rem >= NQ - i) { rem -= NQ - i; i++; }
        triUi[e] = (unsigned char)i;
        triUj[e] = (unsigned char)(i + rem);
    }
    // Upper triangle of the x-x block.  Default (QP_P_INPLACE=0): the q-q corner first, P ping-ponged between two buffers.  With
    // -DQP_P_INPLACE=1 the table is ordered so that P_k can be built OVER P_{k+1} (one P buffer instead of two: 1.5 KB of LDS per
    // half-wave at 7-DoF, 18.4 instead of 21.6 KB per block, i.e. the eighth wavefront per CU -- built and measured in round 6, parity
    // green, and slower in C4's loop, so not the default): element (i, j) of P_k reads entries (i, j), (i, j - NQ), (i - NQ, j),
    // (i - NQ, j - NQ) of P_{k+1} (A^T P A in closed form), i.e. its own and entries of blocks "below" its own in the order
    // v-v > q-v > q-q.  With the v-v block first, then q-v, then the q-q corner, a pass only ever overwrites entries that no later pass
    // reads; inside a pass every read is issued and waited for before the first write (hold_rows4).  The corner's elements -- the only
    // ones that need the cost Hessian and the collision rows -- are the table's last NTRI_U.
    static_assert(NTRI_U <= 32, "the v-v block (same size as the q-q corner) fits the first pass");
    for (int e = hl; e < NTRI_X; e += 32) {
        int i = 0, j;
        if (P_INPLACE) {
            if (e < NTRI_U) {                       // v-v, rows NQ .. NX-1
                int rem = e;
                while (rem >= NQ - i) { rem -= NQ - i; i++; }
                j = NQ + i + rem;
                i += NQ;
            } else if (e < NTRI_U + NQ * NQ) {      // q-v: the full NQ x NQ block
                const int t = e - NTRI_U;
                i = t / NQ;
                j = NQ + t - i * NQ;
            } else {                                // q-q corner
                int rem = e - NTRI_U - NQ * NQ;
                while (rem >= NQ - i) { rem -= NQ - i; i++; }
                j = i + rem;
            }
        } else if (e < NTRI_U) {
            int rem = e;
            while (rem >= NQ - i) { rem -= NQ - i; i++; }
            j = i + rem;
        } else {
            int rem = e - NTRI_U;
            while (rem >= (i < NQ ? NQ : NX - i)) { rem -= (i < NQ ? NQ : NX - i); i++; }
            j = (i < NQ ? NQ : i) + rem;
        }
        triXi[e] = (unsigned char)i;
        triXj[e] = (unsigned char)j;
    }

    const double* xb0 = xg + (size_t)b * (N + 1) * NX;
    const double* ub0 = ug + (size_t)b * N * NU;
    const double dx0_reg = hl < NX ? x0[(size_t)b * NX + hl] - xb0[hl] : 0.0;
    // lane roles (clamped indices keep every load unconditional)
    const int hl_u = hl < NQ ? hl : NQ - 1;                          // control owned / aliased by this lane
    const int hl_x = hl < NX ? hl : NX - 1;                          // state component
    const int hr = hl < NRT ? hl : NRT - 1;                          // constraint row (lanes >= NRT duplicate the last one)
    const bool row_live = hl < NRT;                                  // ... and stay out of the sums
    const int hl_c = hr >= NX ? hr - NX : 0;                         // general row
    const int hz = hl < NZ ? hl : NZ - 1;                            // variable [u; x]
    const int hl_px = hz >= NU ? hz - NU : 0;                        // state component of the lanes NU..
    const int hc = hl < KS ? hl : KS - 1;                            // column of [G | rho | . | I]
    const bool soft_lane = (hr == rNN);
    // where this lane's column of the factor block starts in the stored image (entry i of the column at wbase + i): the
    // lanes 0..NX own the columns of [W | w]; lane LC0 + j owns column j of L^-1, whose entries above the diagonal are zeros
    // that land on the tail of the columns before it and are overwritten there by the (later) stores of those entries; the
    // lanes in between write zeros behind the block (and its pad entry, if it has one)
    const int wbase = hc <= NX ? hc * NQ
                               : ((hc < LC0 || hc - LC0 >= NQ) ? NWP - (LyT::LOFF + NQ * (NQ + 1) / 2) % 2
                                                                : LyT::w_coff(hc - LC0) - (hc - LC0));
    const int lbase = LyT::w_coff(hl_u) - hl_u;                  // L^-1[j][hl_u] (j >= hl_u) sits at lbase + j

    // Base of stage k's record.  Opaque to the optimiser on purpose: otherwise loop-invariant code motion precomputes one
    // 64-bit pointer per (block, lane role) pair -- some forty register pairs that live across the whole kernel and spill.
    // With the barrier every address is  stage base + small per-lane offset + immediate, formed where it is used.
    // (The barrier sits on the stage index, not on the pointer: a pointer that went through inline asm loses its
    // address space and turns every global_load into a flat_load, which also counts in lgkmcnt.)
    auto stage_ptr = [&](int k) -> double* {
        asm volatile("" : "+v"(k));
        return ws + (size_t)k * Ly.stride;
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
    // sum_r a[r] b[r] over one padded row pair (pads are zero).  Two accumulators: a lone wavefront pays the full latency of
    // every dependent FMA, and one chain over all terms is twice as long as it has to be.
    auto pdot = [&](const double* a, const double* bb, int n2) -> double {
        const dbl2* a2 = reinterpret_cast<const dbl2*>(a);
        const dbl2* b2 = reinterpret_cast<const dbl2*>(bb);
        double s0 = 0.0, s1 = 0.0;
        for (int h = 0; h < n2; h++) {
            const dbl2 x = a2[h], y = b2[h];
            s0 = fma(x.x, y.x, s0);
            s1 = fma(x.y, y.y, s1);
        }
        return s0 + s1;
    };

    // ---- initial residual norm and complementarity from the setup partials -----------------------------------------------
    double R0 = 0.0, mu;
    int m_comp;
    {
        double ms = 0.0, cn = 0.0;
        for (int k = hl; k <= N; k += 32) {
            const double* pt = ws + (size_t)k * Ly.stride + Ly.oPART;
            R0 = fmax(R0, pt[0]);
            ms += pt[1];
            cn += pt[2];
        }
        R0 = half_max(R0);
        m_comp = (int)half_sum(cn);
        if (m_comp == 0) m_comp = 1;
        mu = half_sum(ms) / (double)m_comp;
    }
    const double inv_m = 1.0 / (double)m_comp;
    QPT(13);

    // =====================================================================================================================
    // main loop
    // =====================================================================================================================
    double rho_lin = 1.0, alpha = 0.0, sigmu = 0.0, corr_w = 1.0;
    bool pending = false;  // a step (alpha, directions, z+) computed by F2 and not yet applied to state / z
    int it = 0, st_code = 2;
    const double tol = D->qp_tol;
    const double tol_r = D->qp_tol_res > 0.0 ? D->qp_tol_res : D->qp_tol;
    const int max_iter = D->qp_max_iter;
    const int stall_max = D->qp_stall_iters;     // > 0: give up after this many consecutive iterations with alpha < 1/2 (smpc.h)
    int stall = 0;
    bool broke = false;
    double *Pc = sIMG + O_PA, *Pn = sIMG + O_PB2;    // P_{k+1} (in use) / P_k (being built)
    double *pvc = sIMG + O_PVA, *pvn = sIMG + O_PVB;  // costate vectors, same ping-pong
    lds_fence();

    for (it = 0; it < max_iter; it++) {
        if (mu <= tol && rho_lin * R0 <= tol_r) { st_code = 0; break; }

        // ---------------- sweep B1: apply the pending step, factorise H + C^T D C, predictor costate ----------------------
        // (alpha = 0 and zero directions before the first step: the update is applied unconditionally)
        double mu_new = 0.0;
        {
            dbl2 img[IMG_PF];
            QpRow rs;
            dbl2 slb;
            double zc, znc, czar, cznr;
            auto load_b1 = [&](int k) {
                const double* w = stage_ptr(k);
                const dbl2* s2 = reinterpret_cast<const dbl2*>(w + Ly.oIMG);
#pragma unroll
                for (int j = 0; j < IMG_PF; j++) img[j] = ldnt(s2 + min(hl + 32 * j, img_n2 - 1));
                const dbl2 r0 = ldnt(reinterpret_cast<const dbl2*>(w + Ly.oR0) + hr), r1 = ldnt(reinterpret_cast<const dbl2*>(w + Ly.oR1) + hr),
                           r2 = ldnt(reinterpret_cast<const dbl2*>(w + Ly.oR2) + hr);
                rs.lo = r0.x; rs.hi = r0.y; rs.tl = r1.x; rs.tu = r1.y; rs.ll = r2.x; rs.lu = r2.y;
                czar = ldnt_s(w + Ly.oCZA + hr);
                cznr = ldnt_s(w + Ly.oCZN + hr);
                slb = ldnt_s(reinterpret_cast<const dbl2*>(w + Ly.oSL));
                zc = ldnt_s(w + Ly.oZ + hz);
                znc = ldnt_s(w + Ly.oZN + hz);
            };
            auto stage_b1 = [&](int k, auto last_tag) {
                constexpr bool last = decltype(last_tag)::value;
                asm volatile("; QPMARK B1_BEGIN");
                double* w = stage_ptr(k);
                // -- image -> LDS; rows: apply the step, barrier weights, predictor coefficients -- all in the owner's registers
                {
                    dbl2* d2 = reinterpret_cast<dbl2*>(sIMG);
#pragma unroll
                    for (int j = 0; j < IMG_PF; j++) d2[min(hl + 32 * j, img_n2 - 1)] = img[j];
                }
                const double wsoft = slb.x;
                const bool bflag = !last && slb.y != 0.0;
                {
                    const bool soft = soft_lane && wsoft >= 0.0;
                    // the step of this row, recomputed exactly as F2 did from the values it left behind
                    double rr_ = 0.0, s1_ = 0.0, s2_ = 0.0;
                    const QpDir rd = qp_row_dir<false>(rs, soft, wsoft, cznr, sigmu, corr_w, czar, &rr_, &s1_, &s2_, nullptr, nullptr);
                    rs.tl += alpha * rd.dtl; rs.ll += alpha * rd.dll;
                    rs.tu += alpha * rd.dtu; rs.lu += alpha * rd.dlu;
                    stnt_b1(dbl2{rs.tl, rs.tu}, reinterpret_cast<dbl2*>(w + Ly.oR1) + hr);
                    stnt_b1(dbl2{rs.ll, rs.lu}, reinterpret_cast<dbl2*>(w + Ly.oR2) + hr);
                    double Dr;
                    sE[hr] = qp_row_coeff(rs, soft, wsoft, 0.0, 0.0, 0.0, &Dr);
                    sD[hr] = Dr;
                    mu_new += row_live ? qp_row_comp(rs, soft, wsoft) : 0.0;
                }
                zc += alpha * (znc - zc);
                stnt_s(zc, w + Ly.oZ + hz);
                lds_fence();
                QPT(0);
                load_b1(k > 0 ? k - 1 : 0);
                // -- P b, and the rows scaled by their barrier weights
                {
                    // (whole 16-byte reads: as twelve scalar ones the compiler, short of registers, waits for each in turn)
                    const double a = bflag ? pdot(Pc + hl_x * NX, sB, NX / 2) : 0.0;
                    sPB[hl_x] = a;   // (reaches the corrector's backward sweep folded into the stored gradient, below)
                }
                // (fixed trip counts with clamped indices instead of data-dependent loop bounds: the passes of one loop are
                //  independent, and only a fully unrolled loop lets the scheduler overlap their LDS round trips)
                if constexpr (NQ % 2 == 0) {
                    // Two entries per lane and pass (a row of Tt has NQP = NQ entries, a row of Gt MRP: a pair never straddles
                    // two rows, and rT0, rC0 are even), and every read before the first write: the compiler cannot tell the
                    // arrays apart, so a write between two passes' reads makes each pass a round trip through LDS of its own.
                    constexpr int TD_P = (NZ * NQP / 2 + 31) / 32, GD_P = (NQ * MRP_MAX / 2 + 31) / 32;
                    const dbl2* tt2 = reinterpret_cast<const dbl2*>(sTT);
                    const dbl2* gt2 = reinterpret_cast<const dbl2*>(sGT);
                    dbl2 ta[TD_P], td[TD_P], ga[GD_P], gd[GD_P];
#pragma unroll
                    for (int t = 0; t < TD_P; t++) {
                        const int e2 = min(hl + 32 * t, NZ * NQP / 2 - 1);
                        ta[t] = tt2[e2];
                        td[t] = reinterpret_cast<const dbl2*>(sD + rT0)[e2 % (NQP / 2)];
                    }
#pragma unroll
                    for (int t = 0; t < GD_P; t++) {
                        const int e2 = max(min(hl + 32 * t, NQ * (MRP >> 1) - 1), 0);   // (no collision rows: nothing is written)
                        const int r2 = e2 % max(MRP >> 1, 1);
                        ga[t] = gt2[e2];
                        gd[t] = reinterpret_cast<const dbl2*>(sD + rC0)[r2];
                        if (2 * r2 + 1 >= MR) gd[t].y = 0.0;   // (the pad column of Gt; its neighbour in D is the safe-set row)
                    }
#pragma unroll
                    for (int t = 0; t < TD_P; t++) reinterpret_cast<dbl2*>(sTD)[min(hl + 32 * t, NZ * NQP / 2 - 1)] = ta[t] * td[t];
#pragma unroll
                    for (int t = 0; t < GD_P; t++)
                        if (MR > 0) reinterpret_cast<dbl2*>(sGD)[max(min(hl + 32 * t, NQ * (MRP >> 1) - 1), 0)] = ga[t] * gd[t];
                } else {
#pragma unroll
                    for (int t = 0; t < (NZ * NQP + 31) / 32; t++) {
                        const int el = min(hl + 32 * t, NZ * NQP - 1);
                        const int r = el % NQP;
                        sTD[el] = sTT[el] * sD[rT0 + (r < NQ ? r : 0)];   // pad entries of Tt are zero
                    }
#pragma unroll
                    for (int t = 0; t < (NQ * MRP_MAX + 31) / 32; t++) {
                        const int el = hl + 32 * t;
                        if (el < NQ * MRP) {
                            const int r = el % MRP;
                            sGD[el] = sGT[el] * sD[rC0 + (r < MR ? r : 0)];
                        }
                    }
                }
                lds_fence();
                QPT(1);
                // -- H + C^T D C by blocks, with B^T P B / B^T P A folded into the u rows
                if (!last) {
#pragma unroll
                    for (int t = 0; t < (NTRI_U + 31) / 32; t++) {
                        const int el = min(hl + 32 * t, NTRI_U - 1);   // (lanes past the end repeat the last element)
                        const int i = triUi[el], j = triUj[el];
                        // (operands loaded, then held: one batch of reads and one wait per pass, here and in the next loop)
                        constexpr int H = NQP / 2;
                        dbl2 ti[H], tj[H];
                        double sc[5];
#pragma unroll
                        for (int h = 0; h < H; h++) {
                            ti[h] = reinterpret_cast<const dbl2*>(sTT + i * NQP)[h];
                            tj[h] = reinterpret_cast<const dbl2*>(sTD + j * NQP)[h];
                        }
                        sc[0] = sSC[0]; sc[1] = Pc[i * NX + j]; sc[2] = Pc[i * NX + NQ + j]; sc[3] = Pc[(NQ + i) * NX + j];
                        sc[4] = Pc[(NQ + i) * NX + NQ + j];
                        hold_rows2<H>(ti, tj, sc);
                        double s0 = 0.0, s1 = 0.0;
#pragma unroll
                        for (int h = 0; h < H; h++) { s0 = fma(ti[h].x, tj[h].x, s0); s1 = fma(ti[h].y, tj[h].y, s1); }
                        double a = s0 + s1;
                        a += i == j ? sc[0] : 0.0;
                        // B^T P B = c^2 P11 + c dt (P12 + P21) + dt^2 P22
                        a += cB * cB * sc[1] + cB * dt * (sc[2] + sc[3]) + dt * dt * sc[4];
                        sLam[i * NQ + j] = a;
                        sLam[j * NQ + i] = a;
                    }
#pragma unroll
                    for (int t = 0; t < (NQ * NX + 31) / 32; t++) {
                        const int el = min(hl + 32 * t, NQ * NX - 1);
                        const int i = el / NX, jx = el - i * NX;
                        // B^T P A: left block c P11 + dt P21 ; right block dt (c P11 + dt P21) + c P12 + dt P22
                        // (all four entries read whatever the block, as in the P update below)
                        constexpr int H = NQP / 2;
                        const bool right = jx >= NQ;
                        const int jj = right ? jx - NQ : jx;
                        dbl2 ti[H], tj[H];
                        double sc[5];
#pragma unroll
                        for (int h = 0; h < H; h++) {
                            ti[h] = reinterpret_cast<const dbl2*>(sTT + i * NQP)[h];
                            tj[h] = reinterpret_cast<const dbl2*>(sTD + (NU + jx) * NQP)[h];
                        }
                        sc[0] = Pc[i * NX + jj]; sc[1] = Pc[(NQ + i) * NX + jj]; sc[2] = Pc[i * NX + jx]; sc[3] = Pc[(NQ + i) * NX + jx];
                        sc[4] = 0.0;
                        hold_rows2<H>(ti, tj, sc);
                        double s0 = 0.0, s1 = 0.0;
#pragma unroll
                        for (int h = 0; h < H; h++) { s0 = fma(ti[h].x, tj[h].x, s0); s1 = fma(ti[h].y, tj[h].y, s1); }
                        double a = s0 + s1;
                        {
                            const double left = cB * sc[0] + dt * sc[1];
                            a += right ? dt * left + cB * sc[2] + dt * sc[3] : left;
                        }
                        sG[i * WS2 + jx] = a;
                    }
                }
                // element (ix, jx), ix <= jx, of the x-x block of H + C^T D C.  (The diagonal's extras and the q-q block are read
                // with everything else and selected: behind a branch each of them is a round trip through LDS of its own.)
                auto hxx_elem = [&](int ix, int jx, bool corner_pass) -> double {
                    const double dii = sD[ix], lmv = sSC[1], dnn = sD[rNN];
                    double a = pdot(sTT + (NU + ix) * NQP, sTD + (NU + jx) * NQP, NQP / 2);
                    a = fma(sGN[ix] * dnn, sGN[jx], a);
                    if (corner_pass) {   // (compile-time per pass: the index table puts the whole q-q corner into the first one)
                        const int iq = min(ix, NQ - 1), jq = min(jx, NQ - 1);   // (read by every lane of the pass, selected)
                        const double qq = sHQQ[iq * NQ + jq] + pdot(sGT + iq * MRP, sGD + jq * MRP, MRP >> 1);
                        a += jx < NQ ? qq : 0.0;
                    }
                    a += ix == jx ? dii + (ix >= NQ ? lmv : 0.0) : 0.0;
                    return a;
                };
                if (last) {   // P_N = the x-x block itself; at the other stages it is assembled inside the P update below
#pragma unroll
                    for (int t = 0; t < (NTRI_X + 31) / 32; t++) {
                        const int el = min(hl + 32 * t, NTRI_X - 1);
                        const int ix = triXi[el], jx = triXj[el];
                        const double a = hxx_elem(ix, jx, 32 * (t + 1) > TRI_Q0 && 32 * t < TRI_Q0 + NTRI_U);
                        Pn[ix * NX + jx] = a;
                        Pn[jx * NX + ix] = a;
                    }
                }
                // -- gradient: g + C^T e; lanes NU.. keep the x part in a register for the costate update
                // (every operand is read by every lane and the lane's kind selects: behind branches each read was a round trip
                //  through LDS of its own -- eight of them in this block)
                double ghx = 0.0;
                {
                    const int ix = hl_px, iq = min(hl_px, NQ - 1);   // state component of the lanes NU.. (0 on the control lanes)
                    const bool ctl = hz < NU;
                    // (all scalar operands up front, in one batch with the rows below)
                    double g0 = sGZ[hz], eb = sE[ix], gn_i = sGN[ix], e_nn = sE[rNN];
                    double pb1 = sPB[ctl ? hz : ix], pb2 = sPB[ctl ? NQ + hz : (ix >= NQ ? ix - NQ : 0)];
                    double pv1 = pvc[hl_u], pv2 = pvc[NQ + hl_u];
                    hold(g0, eb, gn_i, e_nn);
                    hold(pb1, pb2, pv1, pv2);
                    const double gnn = gn_i * e_nn;
                    double tq = 0.0, cq = 0.0;
                    if constexpr (NQ % 2 == 0) {
                        tq = pdot(sTT + hz * NQP, sE + rT0, NQP / 2);
                        cq = pdot(sGT + iq * MRP, sE + rC0, MRP >> 1);   // (pad column of Gt: zero, times the safe-set row's e)
                    } else {
#pragma unroll
                        for (int r = 0; r < NQ; r++) tq = fma(sTT[hz * NQP + r], sE[rT0 + r], tq);
                        for (int r = 0; r < MR; r++) cq = fma(sGT[iq * MRP + r], sE[rC0 + r], cq);
                    }
                    const double gh = g0 + tq + (hz >= NU ? eb + gnn + (ix < NQ ? cq : 0.0) : 0.0);
                    // (nothing of this gradient is stored: the corrector's backward sweep only propagates what sigma mu a1 + cw a2 add
                    //  to this recursion, see the layout)
                    if (last) {
                        if (hz >= NU) pvn[hz - NU] = gh;
                    } else {
                        // rho = gh_u + B^T (p_{k+1} + P b)  (on the control lanes pb1, pb2 are entries hl, NQ + hl of P b)
                        const double rho = gh + cB * (pv1 + pb1) + dt * (pv2 + pb2);
                        if (hl < NU) sG[hl * WS2 + NX] = rho;
                        else ghx = gh;
                    }
                }
                lds_fence();
                QPT(2);
                if (!last) {
                    if (!chol_from_lds(sLam)) broke = true;
                    // one column per lane: lanes 0..NX solve L [W | w] = [G | rho], lanes LC0.. solve L y = e_j (the rows of
                    // the factor block then end with the rows of L^-1); every lane stores its column
                    {
                        double col[NQ];
                        const int cg = hc <= NX ? hc : NX;
#pragma unroll
                        for (int i = 0; i < NQ; i++) col[i] = hc <= NX ? sG[i * WS2 + cg] : (i == hc - LC0 ? 1.0 : 0.0);
#pragma unroll
                        for (int i = 0; i < NQ; i++) {
                            double v = col[i];
#pragma unroll
                            for (int t = 0; t < i; t++) v = fma(-Lr[i][t], col[t], v);
                            col[i] = v * Linv[i];
                        }
                        if (hc <= NX) {
#pragma unroll
                            for (int i = 0; i < NQ; i++) sWT[hc * NQP + i] = col[i];
                            if (NQP > NQ) sWT[hc * NQP + NQ] = 0.0;   // (the other sweeps stage their blocks over this buffer)
                        }
                        // the block goes out as whole 16-byte pieces (WST_PF store instructions instead of NQ narrow ones: this
                        // stream pays per store), laid out in a buffer of its own
#pragma unroll
                        for (int i = 0; i < NQ; i++) sWI[wbase + i] = col[i];   // (in this order: see wbase)
                    }
                    lds_fence();
                    {
                        const dbl2* s2 = reinterpret_cast<const dbl2*>(sWI);
                        dbl2* d2 = reinterpret_cast<dbl2*>(w + Ly.oW);
#pragma unroll
                        for (int j = 0; j < WST_PF; j++) {
                            const int pc = min(hl + 32 * j, W_N2 - 1);
                            stnt_b1(s2[pc], d2 + pc);
                        }
                    }
                    QPT(3);
                    if (k > 0) {
                        // P_k = Hxx + A^T P A - W^T W (upper triangle, mirrored into the other buffer) and p_k.  The x-x block of
                        // H + C^T D C is assembled here, element by element, not in a pass of its own before the factorisation
                        // (which needs only the u-u and u-x blocks): three passes, their index look-ups and a buffer fewer
#pragma unroll
                        for (int t = 0; t < (NTRI_X + 31) / 32; t++) {
                            const int el = min(hl + 32 * t, NTRI_X - 1);
                            const int i = triXi[el], j = triXj[el];
                            // every operand of the element -- rows i, j of the scaled Jacobian and of W, the safe-set row's entries,
                            // the diagonal extras, the four entries of P_{k+1} (A^T P A by blocks: [P11, dt P11 + P12; dt P11 + P21,
                            // dt^2 P11 + dt (P12 + P21) + P22], i <= j, clamped indices and zero weights outside a block) -- is
                            // loaded, then held: one batch of reads, one wait
                            constexpr int H = NQP / 2;
                            dbl2 ti[H], tj[H], wi[H], wj[H];
                            double sc[9];
                            {
                                const dbl2 *pti = reinterpret_cast<const dbl2*>(sTT + (NU + i) * NQP),
                                           *ptj = reinterpret_cast<const dbl2*>(sTD + (NU + j) * NQP),
                                           *pwi = reinterpret_cast<const dbl2*>(sWT + i * NQP), *pwj = reinterpret_cast<const dbl2*>(sWT + j * NQP);
#pragma unroll
                                for (int h = 0; h < H; h++) { ti[h] = pti[h]; tj[h] = ptj[h]; wi[h] = pwi[h]; wj[h] = pwj[h]; }
                                const int ii = i >= NQ ? i - NQ : i, jj = j >= NQ ? j - NQ : j;
                                sc[0] = sGN[i]; sc[1] = sGN[j]; sc[2] = sD[rNN]; sc[3] = sD[i]; sc[4] = sSC[1];
                                sc[5] = Pc[i * NX + j]; sc[6] = Pc[i * NX + jj]; sc[7] = Pc[ii * NX + j]; sc[8] = Pc[ii * NX + jj];
                            }
                            hold_rows4<H>(ti, tj, wi, wj, sc);
                            double a;
                            {
                                double s0 = 0.0, s1 = 0.0, w0 = 0.0, w1 = 0.0;
#pragma unroll
                                for (int h = 0; h < H; h++) {
                                    s0 = fma(ti[h].x, tj[h].x, s0); s1 = fma(ti[h].y, tj[h].y, s1);
                                    w0 = fma(wi[h].x, wj[h].x, w0); w1 = fma(wi[h].y, wj[h].y, w1);
                                }
                                a = (s0 + s1) - (w0 + w1);
                            }
                            a = fma(sc[0] * sc[2], sc[1], a);
                            if (32 * (t + 1) > TRI_Q0 && 32 * t < TRI_Q0 + NTRI_U) {   // (the passes that hold elements of the q-q corner, see the index table)
                                const int iq = min(i, NQ - 1), jq = min(j, NQ - 1);
                                const double qq = sHQQ[iq * NQ + jq] + pdot(sGT + iq * MRP, sGD + jq * MRP, MRP >> 1);
                                a += j < NQ ? qq : 0.0;
                            }
                            a += i == j ? sc[3] + (i >= NQ ? sc[4] : 0.0) : 0.0;
                            {
                                const double cj = j >= NQ ? dt : 0.0, ci = i >= NQ ? dt : 0.0;
                                a += sc[5] + cj * sc[6] + ci * (sc[7] + dt * sc[8]);
                            }
                            Pn[i * NX + j] = a;
                            Pn[j * NX + i] = a;
                        }
                        {
                            // p_k = gh_x + A^T (p_{k+1} + P b) - W^T w  (every lane reads, the state lanes write)
                            const int i = hl_px, i2 = i >= NQ ? i - NQ : 0;
                            const double q1 = pvc[i] + sPB[i], q2 = pvc[i2] + sPB[i2];
                            double v = ghx + q1 + (i >= NQ ? dt * q2 : 0.0);
                            v -= pdot(sWT + i * NQP, sWT + NX * NQP, NQP / 2);
                            if (hl >= NU && hl < NZ) pvn[i] = v;
                        }
                    }
                }
                QPT(4);
                lds_fence();
                if (last || k > 0) {
                    if (!P_INPLACE) { double* t1 = Pc; Pc = Pn; Pn = t1; }
                    double* t2 = pvc; pvc = pvn; pvn = t2;
                }
                QPT(5);
                asm volatile("; QPMARK B1_END");
            };
            load_b1(N);
            stage_b1(N, std::true_type{});
#pragma unroll 1
            for (int k = N - 1; k >= 0; k--) stage_b1(k, std::false_type{});
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (half_max(broke ? 1.0 : 0.0) > 0.0) { st_code = 4; pending = false; break; }
        if (pending) {
            mu = half_sum(mu_new) * inv_m;
            pending = false;
            if (!(mu == mu)) { st_code = 4; break; }
        }

        // ---------------- forward sweeps: roll-out through the stored factors, ratio tests -------------------------------------
        // F1 (predictor): stores c.z_aff of every row (for F2, B1) and the two vectors a1 = C^T e1, a2 = C^T e2 that make the
        //                 corrector gradient gh0 + sigma mu a1 + cw a2 -- B2 then never touches the rows or the Jacobian.
        // F2 (corrector): stores z+ and c.z+ of every row.
        // Returns the largest ratio -d/v over all slacks and multipliers (its reciprocal is the step to the boundary).
        // The general rows and the factor block come in as whole 16-byte pieces (every lane useful) and are laid out in LDS,
        // where the lanes that need a row or a column of them pick it up: narrow per-lane global loads cost the address
        // unit a full 64-lane instruction each.  The factor block is double-buffered in LDS and committed a stage early, so
        // the control chain of a stage starts without a hand-off.
        auto sweep_f = [&](auto corr_tag, double* rr_out, double* S1_out, double* S2_out) {
            constexpr bool CORR = decltype(corr_tag)::value;
            double rr = 0.0, S1 = 0.0, S2 = 0.0;
            // (ping-pong buffers as offsets, not pointers: an offset can go through an optimisation barrier without the LDS
            //  address space being lost)
            int o_xb = O_XB, o_xn = O_XB + NX;
            int o_wc = O_TD + CST_MAX, o_wn = O_TD + CST_MAX + NWP;    // factor block of this stage / of the next one
            dbl2 Cs[CST_PF], Ws[WST_PF];
            // the rows of a stage (bounds, slacks, multipliers, c.z pair, (b_i, soft weight)) in TWO register sets: the loads of
            // stage k + 1 go out at the top of stage k, a whole stage before their use.  (With one set they could only be
            // issued after stage k's rows had been consumed, ~0.6 of a stage ahead -- under full load less than the memory
            // latency, and the row phase of every forward stage stalled for the difference: 1.0 M of the 5.2 M clocks of a
            // half-wave, profiles/r02_qp_phase_profile.txt.)
            struct RSet { dbl2 r0, r1, r2; double cza, wc; };
            RSet RA, RB;
            // Where the two entries of piece j of this lane go in the row-major LDS image (offsets in doubles from sIMG): the pieces
            // come in the image's transposed layout -- Tt[c][r], Gt[ix][r], gn[i] -- and are transposed on their way into LDS (two
            // 8-byte LDS stores per piece instead of one 16-byte one; the sweeps are bound by the global stream, not by LDS).  Pad
            // entries of the image go to a cell nobody reads (the image's own LDS area is idle during the forward sweeps).
            // (both offsets of a piece packed into one register: the kernel has none to spare)
            int cdst[CST_PF];
            {
                auto dst_of = [&](int e) -> int {
                    if (e < NZ * NQP) {
                        const int c = e / NQP, r = e - c * NQP;
                        return r < NQ ? O_TD + r * NZP + c : 0;
                    }
                    if (e < NZ * NQP + NQ * MRP) {
                        const int t = e - NZ * NQP, ix = t / max(MRP, 1), r = t - ix * MRP;
                        return r < MR ? O_TD + (NQ + r) * NZP + NU + ix : 0;
                    }
                if (e < Ly.nJ) return O_TD + (NQ + MR) * NZP + NU + (e - NZ * NQP - NQ * MRP);
                return e;                  // the defect and the stage scalars: their own cells of the (idle) image area -- sB, sSC
                };
#pragma unroll
                for (int j = 0; j < CST_PF; j++) {
                    const int e = 2 * min(hl + 32 * j, c_n2 - 1);
                    cdst[j] = dst_of(e) | (dst_of(e + 1) << 16);
                }
            }
            // structural zeros of the image (the factorisation sweep used this LDS for other things): u and v columns of the
            // collision rows, u columns of the safe-set row -- written once per sweep, the stages only refresh the rest
            for (int el = hl; el < NRC * NZP; el += 32) {
                const int r = el / NZP, c = el - r * NZP;
                const bool keep = c < NZ && (r < NQ || (c >= NU && (c < NU + NQ || r == NQ + MR)));   // (what the stages' pieces refresh)
                if (!keep) sCst[el] = 0.0;
            }
            auto load_w = [&](int k) {
                const dbl2* s2 = reinterpret_cast<const dbl2*>(stage_ptr(k) + Ly.oW);
#pragma unroll
                for (int j = 0; j < WST_PF; j++) Ws[j] = ldnt(s2 + min(hl + 32 * j, W_N2 - 1));
            };
            auto commit_w = [&](double* dst) {
                dbl2* d2 = reinterpret_cast<dbl2*>(dst);
#pragma unroll
                for (int j = 0; j < WST_PF; j++) d2[min(hl + 32 * j, W_N2 - 1)] = Ws[j];
            };
            auto load_c = [&](int k) {
                const double* w = stage_ptr(k);
                const dbl2* s2 = reinterpret_cast<const dbl2*>(w + Ly.oIMG);
#pragma unroll
                for (int j = 0; j < CST_PF; j++) Cs[j] = ldnt(s2 + min(hl + 32 * j, c_n2 - 1));
            };
            auto load_r = [&](RSet& R, int k) {
                const double* w = stage_ptr(k);
                R.r0 = ldnt(reinterpret_cast<const dbl2*>(w + Ly.oR0) + hr);
                R.r1 = ldnt(reinterpret_cast<const dbl2*>(w + Ly.oR1) + hr);
                R.r2 = ldnt(reinterpret_cast<const dbl2*>(w + Ly.oR2) + hr);
                if (CORR) {
                    R.cza = ldnt_s(w + Ly.oCZA + hr);
                    R.wc = ldnt_s(w + Ly.oWC + hl_u);   // the corrector's w (B2)
                }
            };
            if (hl < NX) sIMG[o_xb + hl] = dx0_reg;
            load_w(0);
            load_c(0);
            load_r(RA, 0);
            commit_w(sIMG + o_wc);
            load_w(N > 1 ? 1 : 0);
            lds_fence();
            // Everything in flight lands before the loop starts (stage 0 needs it at once anyway).  Without this, a register
            // that is still "pending" on loop entry -- a spill reload in the preheader is enough -- makes the compiler put
            // its wait INSIDE the loop body, where it then drains the prefetch queue on every stage.
            asm volatile("" : "+v"(o_wc), "+v"(o_wn), "+v"(o_xb), "+v"(o_xn));   // (reloaded here, not in the preheader)
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            auto stage_f = [&](int k, RSet& cur, RSet& nxt) {
                if (CORR) asm volatile("; QPMARK F2_BEGIN"); else asm volatile("; QPMARK F1_BEGIN");
                const bool last = (k == N);
                const int kn = k < N ? k + 1 : N;          // past the end the loads repeat the end stage: they stay unconditional
                const int kk = k + 2 < N ? k + 2 : N - 1;  // (there are no factors at the end stage)
                double* w = stage_ptr(k);
                double *const xb = sIMG + o_xb, *const xn = sIMG + o_xn, *const wc_ = sIMG + o_wc, *const wn_ = sIMG + o_wn;
                // general rows -> LDS (rows for the constraint lanes, columns for a1 / a2)
                {
#pragma unroll
                    for (int j = 0; j < CST_PF; j++) {
                        sIMG[cdst[j] & 0xffff] = Cs[j].x;
                        sIMG[cdst[j] >> 16] = Cs[j].y;
                    }
                }
                load_c(kn);
                load_r(nxt, kn);
                // u = -L^-T (W x + w): one lane per control, two lane-parallel products with an LDS hand-off in between
                {
                    const double* wr_ = wc_ + hl_u;          // row hl_u of [W | w]: one entry per column
                    // (q and v halves in accumulators of their own: this chain is on the path from one stage's state to the next)
                    double a = wr_[NX * NQ] + (CORR ? cur.wc : 0.0), a_v = 0.0;   // (the predictor's w + the corrector's increment, B2)
#pragma unroll
                    for (int j = 0; j < NQ; j++) {
                        a = fma(wr_[j * NQ], xb[j], a);
                        a_v = fma(wr_[(NQ + j) * NQ], xb[NQ + j], a_v);
                    }
                    a += a_v;
                    if (hl < NQ) sRho[hl] = a;
                }
                lds_fence();
                {
                    double a = 0.0;
#pragma unroll
                    for (int j = 0; j < NQ; j++) a = fma(hl_u <= j ? wc_[lbase + j] : 0.0, sRho[j], a);   // L^-T rho
                    if (hl < NQ) sZU[hl] = last ? 0.0 : -a;
                }
                lds_fence();
                QPT(CORR ? 10 : 6);
                // x+ = A x + B u + b
                if (!last && hl < NX) {
                    const int i = hl < NQ ? hl : hl - NQ;
                    const double u = sZU[i];
                    const double bi = sB[hl];      // (this stage's defect: it came in with the Jacobian pieces)
                    xn[hl] = hl < NQ ? xb[hl] + dt * xb[NQ + hl] + cB * u + bi : xb[hl] + dt * u + bi;
                }
                // rows: c.z for the trial point, directions, ratio test
                {
                    const double* cr = sCst + hl_c * NZP;
                    double a = 0.0, a_q = 0.0, a_v = 0.0;   // (three chains of NQ terms instead of one of 3 NQ)
#pragma unroll
                    for (int c = 0; c < NQ; c++) {
                        a = fma(cr[c], sZU[c], a);
                        a_q = fma(cr[NU + c], xb[c], a_q);
                        a_v = fma(cr[NU + NQ + c], xb[NQ + c], a_v);
                    }
                    a += a_q + a_v;
                    const double cz = hr < NX ? xb[hl_x] : a;
                    const QpRow rs{cur.r0.x, cur.r0.y, cur.r1.x, cur.r1.y, cur.r2.x, cur.r2.y};
                    const double wsoft = sSC[2];
                    const bool soft = soft_lane && wsoft >= 0.0;
                    double s1_ = 0.0, s2_ = 0.0;
                    if (!CORR) {
                        double e1, e2;
                        qp_row_dir<true>(rs, soft, wsoft, cz, 0.0, 0.0, cz, &rr, &s1_, &s2_, &e1, &e2);
                        stnt_s(cz, w + Ly.oCZA + hr);
                        sD[hr] = e1;     // (B1's D / E arrays are free during the forward sweeps)
                        sE[hr] = e2;
                    } else {
                        qp_row_dir<false>(rs, soft, wsoft, cz, sigmu, corr_w, cur.cza, &rr, &s1_, &s2_, nullptr, nullptr);
                        stnt_s(cz, w + Ly.oCZN + hr);
                        stnt_s(hz < NU ? sZU[hl_u] : xb[hl_px], w + Ly.oZN + hz);
                    }
                    S1 += row_live ? s1_ : 0.0;
                    S2 += row_live ? s2_ : 0.0;
                }
                // next stage's factor block -> its LDS buffer; the one after that -> registers
                commit_w(wn_);
                load_w(kk);
                lds_fence();
                if (!CORR) {
                    // (all reads of a half of the rows first, then its products: read by read the compiler waits for each in turn)
                    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
                    {
                        constexpr int RH = (NRC_MAX + 1) / 2;
                        double cc[RH], d1[RH], d2[RH];
#pragma unroll
                        for (int hh = 0; hh < 2; hh++) {
#pragma unroll
                            for (int r = 0; r < RH; r++) {
                                const int rr_ = min(hh * RH + r, NRC_MAX - 1);
                                cc[r] = sCst[rr_ * NZP + hz];
                                d1[r] = sD[NX + rr_];
                                d2[r] = sE[NX + rr_];
                            }
#pragma unroll
                            for (int r = 0; r < RH; r++) {
                                const bool on = hh * RH + r < NRC && hh * RH + r < NRC_MAX;
                                if (hh == 0) { a1 = on ? fma(cc[r], d1[r], a1) : a1; a2 = on ? fma(cc[r], d2[r], a2) : a2; }
                                else { b1 = on ? fma(cc[r], d1[r], b1) : b1; b2 = on ? fma(cc[r], d2[r], b2) : b2; }
                            }
                        }
                    }
                    a1 += b1;
                    a2 += b2;
                    if (hz >= NU) { a1 += sD[hz - NU]; a2 += sE[hz - NU]; }
                    stnt_s(dbl2{a1, a2}, reinterpret_cast<dbl2*>(w + Ly.oA12) + hz);
                    lds_fence();   // (the next stage overwrites the staged rows)
                }
                { const int t = o_xb; o_xb = o_xn; o_xn = t; }
                { const int t = o_wc; o_wc = o_wn; o_wn = t; }
                QPT(CORR ? 11 : 7);
                if (CORR) asm volatile("; QPMARK F2_END"); else asm volatile("; QPMARK F1_END");
            };
            {
                int k = 0;
#pragma unroll 1
                for (; k + 1 <= N; k += 2) {
                    stage_f(k, RA, RB);
                    stage_f(k + 1, RB, RA);
                }
                if (k <= N) stage_f(k, RA, RB);
            }
            *rr_out = half_max(rr);
            *S1_out = half_sum(S1);
            *S2_out = half_sum(S2);
        };

        double rr_aff, S1, S2;
        sweep_f(std::false_type{}, &rr_aff, &S1, &S2);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const double a_aff = rr_aff > 1.0 ? 1.0 / rr_aff : 1.0;
        const double mu_aff = (mu * (double)m_comp + a_aff * S1 + a_aff * a_aff * S2) * inv_m;
        double sigma = mu_aff / mu;
        sigma = fmin(sigma * sigma * sigma, 0.3);  // centring cap (see oracle): halves the iteration tail
        sigmu = sigma * mu;
        // safeguard against Mehrotra cycling (see oracle): damp the second-order term when the affine step is blocked early
        corr_w = a_aff >= 0.3 ? 1.0 : a_aff * a_aff;

        // ---------------- sweep B2: costate recursion of the corrector with the stored factors -------------------------------
        // ~100 instructions per stage: loads run B2D stages ahead (as many register sets, the loop unrolled by as much)
        {
#ifndef QP_B2_DEPTH
#define QP_B2_DEPTH 5
#endif
            // (NQ = 7: a register set is 16 registers -- three pieces of the factor block instead of two -- and five of them pushed the
            //  kernel into spill reloads inside every stage of this sweep; three sets, as rounds 1-3 had for every size)
            constexpr int B2D = NQ >= 7 ? 3 : QP_B2_DEPTH;
            struct BSet { dbl2 a12; dbl2 Ws[WST_PF]; };
            // the two entries of P b this lane needs: controls (i, NQ+i), states (ix, ix-NQ)
            const int ip1 = hl < NU ? hl : hl_px;
            const int ip2 = hl < NU ? NQ + hl : (hl_px >= NQ ? hl_px - NQ : 0);
            auto load_b = [&](BSet& S, int k) {
                const double* w = stage_ptr(k);
                S.a12 = ldnt(reinterpret_cast<const dbl2*>(w + Ly.oA12) + hz);
                const int kf = k < N ? k : N - 1;   // (there are no factors at the end stage)
                const dbl2* s2 = reinterpret_cast<const dbl2*>(stage_ptr(kf) + Ly.oW);
#pragma unroll
                for (int j = 0; j < WST_PF; j++) S.Ws[j] = ldnt(s2 + min(hl + 32 * j, W_N2 - 1));
            };
            auto stage_b2 = [&](BSet& S, int k, auto last_tag) {
                constexpr bool last = decltype(last_tag)::value;
                asm volatile("; QPMARK B2_BEGIN");
                double* w = stage_ptr(k);
                const int kp = k > B2D ? k - B2D : 0;
                const double gh = sigmu * S.a12.x + corr_w * S.a12.y;   // (the increment's gradient: the predictor's part is in B1's w)
                if (last) {
                    if (hz >= NU) pvn[hz - NU] = gh;
                    // (the factor pieces of this set are never used: consume them so that their registers stay reserved)
                    double sink = 0.0;
#pragma unroll
                    for (int j = 0; j < WST_PF; j++) sink = fma(0.0, S.Ws[j].x + S.Ws[j].y, sink);
                    if (hl == 63 && sink != 0.0) sRho[0] = sink;
                    load_b(S, kp);
                } else {
                    // factor block -> LDS;  rho = gh_u + B^T q,  q = p_{k+1} + P b ;  wv = L^-1 rho ;  p_k = gh_x + A^T q - W^T wv
                    {
                        dbl2* d2 = reinterpret_cast<dbl2*>(sWstA);
#pragma unroll
                        for (int j = 0; j < WST_PF; j++) d2[min(hl + 32 * j, W_N2 - 1)] = S.Ws[j];
                    }
                    int i1 = ip1, i2 = ip2;
                    if constexpr (NQ >= 7) {
                        // (7-DoF: the two indices were spilled across the other sweeps and reloaded from scratch inside every stage of this
                        //  one -- a scratch load in a stage body turns the counted waits into vmcnt(0); formed here from the lane index instead)
                        int hh = hl;
                        asm volatile("" : "+v"(hh));
                        const int px = hh < NZ ? (hh >= NU ? hh - NU : 0) : NZ - 1 - NU;
                        i1 = hh < NU ? hh : px;
                        i2 = hh < NU ? NQ + hh : (px >= NQ ? px - NQ : 0);
                    }
                    const double q1 = pvc[i1], q2 = pvc[i2];
                    if (hl < NU) sRho[hl] = gh + cB * q1 + dt * q2;
                    load_b(S, kp);
                    lds_fence();
                    QPT(8);
                    {
                        double v = 0.0;
#pragma unroll
                        for (int j = 0; j < NQ; j++)   // L^-1 rho: row hl_u, entries j <= hl_u
                            v = fma(j <= hl_u ? sWstA[LyT::w_coff(j) - j + hl_u] : 0.0, sRho[j], v);
                        if (hl < NQ) sWv[hl] = v;
                        // the corrector's w: one 64-byte sector of its own (inside the factor block it was NQ separate
                        // read-modify-writes); lanes NQ.. hold the last entry again and fill the sector
                        stnt_s(v, w + Ly.oWC + (hl < 8 ? hl : NQ - 1));
                    }
                    lds_fence();
                    if (k > 0 && hz >= NU) {
                        double v = gh + (hl_px < NQ ? q1 : dt * q2 + q1);
#pragma unroll
                        for (int t = 0; t < NQ; t++) v = fma(-sWstA[hl_px * NQ + t], sWv[t], v);
                        pvn[hl_px] = v;
                    }
                }
                lds_fence();
                if (last || k > 0) { double* t2 = pvc; pvc = pvn; pvn = t2; }
                QPT(9);
                asm volatile("; QPMARK B2_END");
            };
            // stage N - m uses set m % B2D
            BSet S[B2D];
#pragma unroll
            for (int j = 0; j < B2D; j++) load_b(S[j], N >= j ? N - j : 0);
            stage_b2(S[0], N, std::true_type{});
            int k = N - 1;
#pragma unroll 1
            for (; k - (B2D - 1) >= 0; k -= B2D) {
#pragma unroll
                for (int j = 0; j < B2D; j++) stage_b2(S[(j + 1) % B2D], k - j, std::false_type{});
            }
#pragma unroll
            for (int j = 0; j < B2D - 1; j++)
                if (k - j >= 0) stage_b2(S[(j + 1) % B2D], k - j, std::false_type{});
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");

        // ---------------- sweep F2: corrector roll-out and step length --------------------------------------------------------
        double rr_max, S1c, S2c;
        sweep_f(std::true_type{}, &rr_max, &S1c, &S2c);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        // fraction to the boundary (see oracle): 0.995, approaching 1 with the complementarity (cap 0.9999) when the step is
        // within 1% of the full Newton step; an earlier-blocked step keeps the classical margin to stay centred
        const double a_max = rr_max > 0.0 ? 1.0 / rr_max : 1e300;
        const double tau_k = a_max >= 0.99 ? fmin(0.9999, fmax(QP_FTB, 1.0 - mu)) : QP_FTB;
        alpha = fmin(1.0, tau_k * a_max);
        if (!(alpha == alpha)) { st_code = 4; break; }
        if (alpha < QP_ALPHA_MIN) { st_code = 3; break; }
        pending = true;
        rho_lin *= (1.0 - alpha);
        // sum(lambda t) is a quadratic in the step length: the new complementarity is known before the step is applied
        const double mu_before = mu;
        mu = (mu * (double)m_comp + alpha * S1c + alpha * alpha * S2c) * inv_m;
        if (!(mu == mu)) { st_code = 4; pending = false; break; }
        // a stalled iteration: a short step that did not halve the complementarity either (a degenerate but feasible QP
        // crawls with short steps while mu still falls); an iterate that meets the exit test is never reported as stalled
        // ... stall_max of them in a row, or 7/6 of that in total (an infeasible QP whose complementarity falls in bursts resets the run
        // now and then).  Both counts in one register: the run in the low half, the total in the high half.
        {
            const bool stalled = alpha < 0.5 && !(mu < 0.5 * mu_before);
            stall = stalled ? stall + 0x10001 : (stall & ~0xffff);
        }
        if (stall_max > 0 && ((stall & 0xffff) >= stall_max || (stall >> 16) >= stall_max + (stall_max + 5) / 6) &&
            !(mu <= tol && rho_lin * R0 <= tol_r)) { st_code = 5; it++; break; }    // (the step just computed is still applied)
    }
    if (it == max_iter && st_code == 2 && mu <= tol && rho_lin * R0 <= tol_r) st_code = 0;

    // ---- full SQP step (FIXED_STEP, parser.py:139), applying the last IPM step if it is still pending -------------------
    bool bad = false;
    const double a_fin = pending ? alpha : 0.0;
    // (chunks of 8 stages: all loads of a chunk are issued before its first -- conditional -- store, so the chunk costs one
    //  memory latency instead of eight)
    constexpr int EP_CH = 8;
    for (int k0 = 0; k0 <= N; k0 += EP_CH) {
        double zz[EP_CH], zzn[EP_CH], bb[EP_CH];
#pragma unroll
        for (int j = 0; j < EP_CH; j++) {
            const int k = k0 + j <= N ? k0 + j : N;
            const double* w = ws + (size_t)k * Ly.stride;
            zz[j] = ldnt_s(w + Ly.oZ + hz);
            zzn[j] = ldnt_s(w + Ly.oZN + hz);
            const int ku = k < N ? k : N - 1;
            bb[j] = hl < NU ? ub0[(size_t)ku * NU + hl_u] : xb0[(size_t)k * NX + hl_px];
        }
#pragma unroll
        for (int j = 0; j < EP_CH; j++) {
            const int k = k0 + j;
            const double v = bb[j] + zz[j] + a_fin * (zzn[j] - zz[j]);
            if (k <= N) {
                if (hl < NU) {
                    if (k < N) { u_out[((size_t)b * N + k) * NU + hl] = v; bad |= !(v == v); }
                } else if (hl < NZ) {
                    x_out[((size_t)b * (N + 1) + k) * NX + hl - NU] = v;
                    bad |= !(v == v);
                }
            }
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
    if (wstat && hl == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime(), t_begin = s_tbegin[half];
        atomicAdd(&wstat[0], t_end - t_begin);   // busy ticks summed over half-waves
        atomicMin(&wstat[1], t_begin);           // first start
        atomicMax(&wstat[2], t_end);             // last end
        atomicAdd(&wstat[3], 1ull);
    }
    if (hl == 0) {
        // acados' RTI tolerates a QP that stopped at its iteration cap (see oracle); breakdown / min-step are QP failures
        int stc = (st_code == 0 || st_code == 2) ? SMPC_STATUS_SUCCESS : SMPC_STATUS_QP_FAILURE;
        if (ws[Ly.oPART + 3] != 0.0) stc = SMPC_STATUS_QP_FAILURE;   // a violated constant row at node 0 (k_qp_setup, rows_at_node0)
        if (any_bad && stc == SMPC_STATUS_SUCCESS) stc = SMPC_STATUS_NAN;
        status[b] = stc;
        if (qp_iter) qp_iter[b] = it;
        if (last_iter) last_iter[b] = it;
        if (it_hist) atomicAdd(&it_hist[min(it, 255)], 1);   // next solve's longest-first order (k_order_by_iters)
    }
}


// order[] = instance indices sorted by decreasing previous iteration count: a counting sort over ONE-WAVE blocks (a block of
// another stream's kernel only starts where a wavefront slot is free; the single 1024-thread block of rounds 1-4 needed a whole CU
// without QP wavefronts and waited 3.4 ms on average, 30 ms at worst, at 65 536 instances: profiles/r04_c2_kernel_summary_by_grid.txt).
// The histogram hist[256] of the iteration counts is accumulated by k_qp_ipm itself as its half-waves retire; every block here
// turns it into bin offsets (256 adds), takes positions inside a bin from the global cursors cur[256], and the last block to
// finish clears hist, cur and the ticket for the next solve (no parity, nothing baked into a captured graph).  The order inside
// a bin is arbitrary (as before): instances are independent, the order only changes the makespan.
constexpr int ORD_PER_BLOCK = 256;
__global__ __launch_bounds__(64) void k_order_by_iters(int B, const int32_t* __restrict__ last_iter, int32_t* __restrict__ order,
                                                       int32_t* __restrict__ hist, int32_t* __restrict__ cur, int32_t* __restrict__ ticket) {
    __shared__ int offs[256];
    const int t = threadIdx.x;
    {
        // offs[v] = number of instances with more than v iterations: a suffix sum over 256 bins, four bins per lane
        int h4[4], run = 0;
#pragma unroll
        for (int j = 3; j >= 0; j--) { h4[j] = hist[4 * t + j]; }
        int mine = h4[0] + h4[1] + h4[2] + h4[3];
        // inclusive suffix scan of `mine` over the 64 lanes (lane t needs the sum over lanes > t)
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_down(incl, o);
            if (t + o < 64) incl += v;
        }
        run = incl - mine;                        // bins of the lanes above
#pragma unroll
        for (int j = 3; j >= 0; j--) { offs[4 * t + j] = run; run += h4[j]; }
    }
    __syncthreads();
    const int i0 = blockIdx.x * ORD_PER_BLOCK;
#pragma unroll
    for (int j = 0; j < ORD_PER_BLOCK / 64; j++) {
        const int i = i0 + 64 * j + t;
        if (i < B) {
            const int v = min(max(last_iter[i], 0), 255);
            const int pos = offs[v] + atomicAdd(&cur[v], 1);
            if (pos < B) order[pos] = i;          // (always true when hist holds exactly the last solve of this batch)
        }
    }
    __threadfence();
    __syncthreads();
    if (t == 0) offs[0] = atomicAdd(ticket, 1);
    __syncthreads();
    if (offs[0] == (int)gridDim.x - 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) { hist[4 * t + j] = 0; cur[4 * t + j] = 0; }
        if (t == 0) *ticket = 0;
    }
}

}  // namespace smpc
