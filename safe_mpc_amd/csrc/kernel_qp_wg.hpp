// kernel_qp_wg.hpp -- k_qp_ipm_wg: the LATENCY form of the stage QP's interior-point solve (round 6).
//
// k_qp_ipm (kernel_qp.hpp) gives an instance half a wavefront and walks the horizon four times per iteration with it: whatever
// the batch, an iteration of a lone instance costs 0.19 ms (Z1, N = 30), which is what the tail of every launch and a small batch
// (one GPU's share of the 4096-instance headline on 8 GPUs: 512) pay.  About half of an iteration's work does not belong to the
// recursions at all: it is the same for every stage and needs nothing of the stage before or after --
//     rows: apply the step, barrier weights D, e;  H + C^T D C and g + C^T e of the stage;  c.z, ratio test, C^T e1, C^T e2.
// Here ONE WORKGROUP (NHW half-wavefronts) owns an instance.  Per iteration:
//   A  stage-parallel (half-wave h takes stages h, h + NHW, ...): rows + the stage's P-independent blocks -> "H record" (HBM / L2)
//   B  wavefront 0: Riccati recursion  Lambda = Huu + B^T P B, G, Cholesky, [W | w], P_k, p_k; then the gains K = L^-T W, k = L^-T w
//      and Lambda^-1 (a back-substitution in registers) -> factor blocks (LDS)
//   C  wavefront 0: predictor roll-out u = -(K x + k) (LDS -> z in LDS; one hand-off per stage: the state)
//   D  stage-parallel: c.z_aff, ratio test, e1 / e2, a1 = C^T e1, a2 = C^T e2 (LDS)
//   E  all threads: the corrector's gradient of every stage; wavefront 0: corrector costate (touches only LDS; one hand-off per stage)
//   F  wavefront 0: corrector roll-out
//   G  stage-parallel: c.z+, ratio test, z+ -> workspace
// with a workgroup barrier between phases and the step-length logic replicated on every lane.  A and B overlap by chunks of eight
// stages (wavefront 0 factorises a chunk while the other half-waves assemble the one below it), D runs beside C and G beside F by
// chunks of six (the other half-waves take the rows of stages that are rolled out already).  The sequential phases keep the factor
// blocks, the defects, a1 / a2 and the roll-out in LDS: no HBM round trip sits on their chains any more.
// SAME algorithm, SAME workspace records (written by k_stage_build / k_qp_setup, QpLayout) and the same per-row arithmetic
// (qp_row_dir / qp_row_coeff of kernel_qp.hpp) as k_qp_ipm; only the order of a few sums differs (the x-x block is assembled before
// the recursion instead of inside the P update, [W | w]^T [W | w] comes from the f64 MFMA, the control is -(K x + k) with the gain
// formed explicitly instead of -L^-T (W x + w), complementarity sums are taken in eight groups of stages), so the two kernels agree to rounding, iterate for iterate (tests/test_gpu_parity.py::test_qp_kernels_agree).
// What bounds ONE wavefront is its instruction count -- one instruction per ~5 clocks whatever it is, an LDS hand-off 83 clocks, a
// dependent f64 MFMA 80 (scripts/lat_probe.hip, profiles/r06_lat_probe.txt) -- so the recursions are written for few instructions:
// lane-dependent addresses formed once, reads batched behind one wait (hold_n), in-place updates, zeros stored instead of masks.
// The engine picks per launch (engine.hip: qp_wg_choice; smpc_set_qp_mode): this form up to 1024 instances, k_qp_ipm for throughput.
// DESIGN.md section 4c has the measurements (143 k clocks per IPM iteration of a lone instance against k_qp_ipm's 447 k).
// Reference: controller.py:97-110, 136-167 (the QP HPIPM solves inside acados' RTI step).
#pragma once
#include "kernel_qp.hpp"

namespace smpc {

// P-independent blocks of one stage's KKT system, written by phase A and read by phase B of the same workgroup.  Laid out for the
// reader: wavefront 0 takes every element straight into the register of the lane that uses it (no LDS staging on the recursion's
// chain) -- element e of [triangle of Huu | Hux row-major] by lane e mod 64, the x-x block as full rows of 16 so that lane (g, c) of the
// f64 MFMA's result layout finds rows g, g + 4, ... of column c, the gradient by the lane of its variable.
template <int NQ> struct HRecLayout {
    static constexpr int NX = 2 * NQ, NZ = 3 * NQ;
    static constexpr int NTRI_U = NQ * (NQ + 1) / 2, NTRI_X = NX * (NX + 1) / 2;
    static constexpr int NE = NTRI_U + NQ * NX;                // elements of [triangle of Huu + (C^T D C)_uu | (C^T D C)_ux]
    static constexpr int oA = 0;                               // ... element e at oA + e
    static constexpr int oXX = 128;                            // x-x block, full, row i at oXX + 16 i
    static constexpr int oGH = oXX + 16 * NX;                  // g + C^T e  [u | x]
    static constexpr int SIZE = qp_al8(oGH + NZ);
    static_assert(NE <= 128 && NX + 1 <= 16, "two elements per lane; [W | w] fits one 16 x 16 tile");
};

template <int NQ> struct WgLds {
    using LyT = QpLayout<NQ>;
    static constexpr int NX = 2 * NQ, NZ = 3 * NQ, NZP = LyT::NZP, NQP = LyT::NQP, WS2 = LyT::WS2;
    // factor block of a stage as this kernel keeps it in LDS: NQ rows of [K | k | pad | Lambda^-1], row-major (K = L^-T W, k = L^-T w,
    // Lambda^-1 = L^-T L^-1: what the Riccati lanes' columns of [W | w | L^-1] become under the back-substitution) -- those lanes
    // store with one base address and immediate offsets, the roll-outs read row i of [K | k] and the corrector row i of Lambda^-1
    // as whole 16-byte pieces
    static constexpr int LCOL = NX + 2, FS = qp_even_c(NX + 2 + NQ), NFW = NQ * FS;
    static constexpr int NZS = NZP + 2;        // roll-out record of a stage: [u | x | two cells that absorb the stores of idle lanes]
    static constexpr int ACC = 8 * 32 * 2;     // accumulator cells of the row phases (aliased over the assembling half-waves' scratch)
    int MRP, NRC, SCR_A, SCR_D, CST;
    int o_fac, o_a12, o_scrd, o_scra2, o_z, o_wc, o_b, o_bf, o_seq;
    int s_P, s_pv, s_lam, s_G, s_red, s_part, s_flag;
    int total;
    // scratch of a half-wave in phase A: [image | D | E | TD | GD]; in phases D / G: [D | E | b, scalars | row-major general rows | dump]
    int a_D, a_E, a_TD, a_GD;
    int d_D, d_E, d_BS, d_CST, d_DUMP;
    __host__ __device__ WgLds(int N, int MR, int NHW) {
        MRP = qp_even_c(MR);
        NRC = NQ + MR + 1;
        const int IMG = NZ * NQP + NQ * MRP + NX + qp_even_c(NQ * NQ) + NZP + NX + 4;
        a_D = qp_even_c(IMG); a_E = a_D + 32; a_TD = a_E + 32; a_GD = a_TD + NZ * NQP;
        SCR_A = qp_even_c(a_GD + NQ * MRP);
        CST = NRC * NZP;
        d_D = 0; d_E = 32; d_BS = 64; d_CST = d_BS + qp_even_c(NX + 4); d_DUMP = d_CST + qp_even_c(CST);
        SCR_D = d_DUMP + 2;
        const int n1 = N + 1;
        const int r0 = n1 * (NFW + 2 * NZP), ra = (NHW < 2 ? NHW : 2) * SCR_A;     // (half-waves 0 and 1: phase-A scratch over the dead factor blocks)
        o_fac = 0;
        o_a12 = n1 * NFW;
        o_scrd = r0 > ra ? r0 : ra;
        o_scra2 = o_scrd + NHW * SCR_D;            // phase-A scratch of the half-waves 2 .. NHW-1, which assemble a chunk of stages WHILE
        {                                          // wavefront 0 factorises the chunk above it (so not over the factor blocks); in the
            const int na = (NHW > 2 ? NHW - 2 : 0) * SCR_A;     // row phases the same cells hold the sums of the stage groups: 8 groups x
            o_z = o_scra2 + (na > ACC ? na : ACC);              // 32 lanes x (S1, S2)
        }
        o_wc = o_z + (n1 + 1) * NZS;
        o_b = o_wc + n1 * NQP;
        o_bf = o_b + n1 * NX;
        o_seq = o_bf + qp_even_c(n1);
        int s = o_seq;
        s_P = s; s += NX * NX;
        s_pv = s; s += NX + 2;               // (+ a cell for the stores of idle lanes)
        s_lam = s; s += qp_even_c(NQ * NQ);
        s_G = s; s += NQ * WS2;
        s_part = s; s += qp_even_c(n1);      // complementarity sum of every stage (phase A)
        s_red = s; s += 4 * 8;               // (eight groups whatever the workgroup size: block_reduce)
        s_flag = s; s += 2;
        total = s;
    }
};

#ifdef QP_PROFILE
__device__ unsigned long long g_wg_prof[16];
#define WGT(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tprev; tprev = t_; } } while (0)
#else
#define WGT(i) do { } while (0)
#endif


#ifndef QP_WG_CHUNK
#define QP_WG_CHUNK 8      // stages per step of the assembly / factorisation overlap (phases A and B)
#endif
#ifndef QP_WG_RCH4
#define QP_WG_RCH4 6       // stages per step of the roll-out / rows overlap (phases C and D, F and G) in a 4-half-wave workgroup
#endif
#ifndef QP_WG_RCH8
#define QP_WG_RCH8 6       // ... in an 8-half-wave workgroup
#endif
#ifndef QP_WG_SEQ_LANES
#define QP_WG_SEQ_LANES 32 // lanes of wavefront 0 that run the roll-outs and the corrector's costate (written for 32; 64: the upper half mirrors)
#endif
#ifndef QP_WG_RND4
#define QP_WG_RND4 2       // rounds of rows the two free half-waves of a 4-half-wave workgroup take per step of the roll-out
#endif
#ifndef QP_WG_RND8
#define QP_WG_RND8 1       // the same for the six free half-waves of an 8-half-wave workgroup
#endif
typedef double v4d __attribute__((ext_vector_type(4)));
// "These N values are needed HERE": one empty asm that lists them all -- the compiler then issues every load behind them before this
// point and waits once (kernel_qp.hpp: hold_rows2 / hold_rows4; left alone it sinks each LDS read next to its use, and a lone wavefront
// pays a round trip through LDS per read: profiles/r06_wg_isa_before_hold.txt)
template <int N> __device__ __forceinline__ void hold_n(double (&a)[N]) {
    static_assert(N >= 1 && N <= 28, "one asm statement takes at most 30 operands");
    if constexpr (N == 1) asm volatile("" : "+v"(a[0]));
    else if constexpr (N == 2) asm volatile("" : "+v"(a[0]), "+v"(a[1]));
    else if constexpr (N == 3) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
    else if constexpr (N == 4) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
    else if constexpr (N == 5) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]));
    else if constexpr (N == 6) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));
    else if constexpr (N == 7) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]));
    else if constexpr (N == 8) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    else if constexpr (N == 9) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]));
    else if constexpr (N == 10) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]));
    else if constexpr (N == 11) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]));
    else if constexpr (N == 12) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]));
    else if constexpr (N == 13) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]));
    else if constexpr (N == 14) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]));
    else if constexpr (N == 15) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]));
    else if constexpr (N == 16) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
    else if constexpr (N == 17) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]));
    else if constexpr (N == 18) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]));
    else if constexpr (N == 19) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]));
    else if constexpr (N == 20) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]));
    else if constexpr (N == 21) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]), "+v"(a[20]));
    else if constexpr (N == 22) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]), "+v"(a[20]), "+v"(a[21]));
    else if constexpr (N == 23) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]), "+v"(a[20]), "+v"(a[21]), "+v"(a[22]));
    else if constexpr (N == 24) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]), "+v"(a[20]), "+v"(a[21]), "+v"(a[22]), "+v"(a[23]));
    else if constexpr (N == 25) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]), "+v"(a[20]), "+v"(a[21]), "+v"(a[22]), "+v"(a[23]), "+v"(a[24]));
    else if constexpr (N == 26) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]), "+v"(a[20]), "+v"(a[21]), "+v"(a[22]), "+v"(a[23]), "+v"(a[24]), "+v"(a[25]));
    else if constexpr (N == 27) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]), "+v"(a[20]), "+v"(a[21]), "+v"(a[22]), "+v"(a[23]), "+v"(a[24]), "+v"(a[25]), "+v"(a[26]));
    else if constexpr (N == 28) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]), "+v"(a[20]), "+v"(a[21]), "+v"(a[22]), "+v"(a[23]), "+v"(a[24]), "+v"(a[25]), "+v"(a[26]), "+v"(a[27]));
}
// One workgroup of NHW half-wavefronts per instance: block b solves instance b.
#ifndef QP_WG_WAVES_PER_EU
#define QP_WG_WAVES_PER_EU 1
#endif
template <int NQ, int MRT, int NHW>
__global__ __launch_bounds__(32 * NHW) __attribute__((amdgpu_waves_per_eu(QP_WG_WAVES_PER_EU, QP_WG_WAVES_PER_EU))) void k_qp_ipm_wg(
    const smpc_problem_desc* __restrict__ D, int B, int N, const double* __restrict__ x0, const double* __restrict__ xg,
    const double* __restrict__ ug, double* __restrict__ ws_all, double* __restrict__ hrec_all, double* __restrict__ x_out,
    double* __restrict__ u_out, int32_t* __restrict__ status, int32_t* __restrict__ qp_iter, int32_t* __restrict__ last_iter,
    const uint8_t* __restrict__ active, int32_t* __restrict__ it_hist) {
    using LyT = QpLayout<NQ>;
    using HR = HRecLayout<NQ>;
    constexpr int NX = 2 * NQ, NU = NQ, NZ = 3 * NQ, NZP = LyT::NZP, NQP = LyT::NQP, WS2 = LyT::WS2, LC0 = LyT::LC0, KS = LyT::KS;
    constexpr int MR_MAX = MRT >= 0 ? MRT : SMPC_MAX_ROWS, MRP_MAX = qp_even_c(MR_MAX), NRC_MAX = NQ + MR_MAX + 1;
    constexpr int NTRI_U = HR::NTRI_U, NTRI_X = HR::NTRI_X;
    constexpr int IMG_MAX = NZ * NQP + NQ * MRP_MAX + NX + qp_even_c(NQ * NQ) + NZP + NX + 4, IMG_PF = (IMG_MAX / 2 + 31) / 32;
    constexpr int NF_MAX = NZ * NQP + NQ * MRP_MAX + NX + NX + 4, CST_PF = (NF_MAX / 2 + 31) / 32;
    constexpr int NT = 32 * NHW;
    static_assert(MRT < 0 || NX + NRC_MAX <= 32, "one lane per constraint row");
    static_assert(KS <= 32, "one lane per column of [G | rho | I]");
    static_assert(NTRI_U <= 32, "the q-q corner fits the first pass");
    extern __shared__ __attribute__((aligned(16))) double dsm[];
    __shared__ unsigned char triUi[NTRI_U], triUj[NTRI_U], triXi[NTRI_X], triXj[NTRI_X];
#ifdef QP_PROFILE
    unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    const int tid = threadIdx.x, hl = tid & 31, hw = tid >> 5;
    const bool seq = tid < 64;            // wavefront 0 runs the sequential phases (its upper half mirrors the lower one wherever a
                                          // phase is written for 32 lanes: same values to the same addresses)
    const int ln = tid & 63, lg = ln >> 4, lc = ln & 15;      // lane of the wavefront; (group, column) in the f64 MFMA's result layout
    const int b = blockIdx.x;
    if (b >= B) return;
    if (active && !active[b]) {
        if (tid == 0) {
            status[b] = SMPC_STATUS_SUCCESS;
            if (qp_iter) qp_iter[b] = 0;
            if (last_iter) last_iter[b] = 0;
            if (it_hist) atomicAdd(&it_hist[0], 1);
        }
        return;
    }
    const LyT Ly(MRT >= 0 ? MRT : D->n_rows);
    const WgLds<NQ> Ls(N, Ly.MR, NHW);
    const int MR = Ly.MR, MRP = Ly.MRP, NRC = Ly.NRC, NRT = Ly.NRT;
    const int rT0 = NX, rC0 = NX + NQ, rNN = NX + NQ + MR;
    double* const ws = ws_all + (size_t)b * Ly.per_instance(N);
    double* const hrec = hrec_all + (size_t)b * (N + 1) * HR::SIZE;
    const double dt = D->dt, cB = 0.5 * dt * dt;
    const int img_n2 = Ly.nIMG >> 1, c_n2 = Ly.nF >> 1;

    // index tables (as in k_qp_ipm)
    for (int e = tid; e < NTRI_U; e += NT) {
        int i = 0, rem = e;
        while (rem >= NQ - i) { rem -= NQ - i; i++; }
        triUi[e] = (unsigned char)i;
        triUj[e] = (unsigned char)(i + rem);
    }
    for (int e = tid; e < NTRI_X; e += NT) {
        int i = 0, j;
        if (e < NTRI_U) {
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
    // lane roles inside a half-wave (clamped: lanes past an array's end repeat its last entry)
    const int hl_u = hl < NQ ? hl : NQ - 1;
    const int hl_x = hl < NX ? hl : NX - 1;
    const int hr = hl < NRT ? hl : NRT - 1;
    const bool row_live = hl < NRT;
    const int hl_c = hr >= NX ? hr - NX : 0;
    const int hz = hl < NZ ? hl : NZ - 1;
    const int hl_px = hz >= NU ? hz - NU : 0;
    const int hc = hl < KS ? hl : KS - 1;
    const bool soft_lane = (hr == rNN);

    double* const sFac = dsm + Ls.o_fac;
    double* const sA12 = dsm + Ls.o_a12;
    double* const sZ = dsm + Ls.o_z;
    double* const sWC = dsm + Ls.o_wc;
    double* const sBk = dsm + Ls.o_b;
    double* const sBf = dsm + Ls.o_bf;
    double* const sLam = dsm + Ls.s_lam;
    double* const sG = dsm + Ls.s_G;
    double* const sRed = dsm + Ls.s_red;
    double* const sFlag = dsm + Ls.s_flag;
    double* const sPart = dsm + Ls.s_part;

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
    // max / sum / sum over the whole workgroup; every thread gets the result.  The sums are taken in an order that does not depend on
    // the workgroup's size: stage k belongs to group k mod 8 (a lane adds its stages of a group in increasing order, then the 32
    // lanes, then the groups 0 .. 7), so a workgroup of 4 half-waves, which owns two groups per half-wave, gets the bits of one of 8.
    constexpr int NVG = 8, GPH = NVG / NHW;        // groups, groups per half-wave
    static_assert(NVG % NHW == 0 && NHW <= NVG && NHW >= 4, "half-waves per workgroup: 8 or 4 (wavefront 0 runs the recursions, the others assemble beside it)");
    auto block_reduce = [&](double& vmax, double (&s1)[GPH], double (&s2)[GPH]) {
        vmax = half_max(vmax);
#pragma unroll
        for (int g = 0; g < GPH; g++) {
            const double a_ = half_sum(s1[g]), c_ = half_sum(s2[g]);
            if (hl == 0) { sRed[4 * (hw + g * NHW) + 1] = a_; sRed[4 * (hw + g * NHW) + 2] = c_; }
        }
        if (hl == 0) sRed[4 * hw] = vmax;
        __syncthreads();
        double m = sRed[0], a = sRed[1], c = sRed[2];
#pragma unroll
        for (int j = 1; j < NVG; j++) { if (j < NHW) m = fmax(m, sRed[4 * j]); a += sRed[4 * j + 1]; c += sRed[4 * j + 2]; }
        vmax = m; s1[0] = a; s2[0] = c;
        __syncthreads();
    };

    // where piece j of a stage's [Tt | Gt | gn | b | scalars] goes in the row-major scratch of the row phases (as in k_qp_ipm's
    // forward sweeps: transposed on the way into LDS; pads go to a dump cell)
    int cdst[CST_PF];
    {
        auto dst_of = [&](int e) -> int {
            if (e < NZ * NQP) {
                const int c = e / NQP, r = e - c * NQP;
                return r < NQ ? Ls.d_CST + r * NZP + c : Ls.d_DUMP;
            }
            if (e < NZ * NQP + NQ * MRP) {
                const int t = e - NZ * NQP, ix = t / max(MRP, 1), r = t - ix * MRP;
                return r < MR ? Ls.d_CST + (NQ + r) * NZP + NU + ix : Ls.d_DUMP;
            }
            if (e < Ly.nJ) return Ls.d_CST + (NQ + MR) * NZP + NU + (e - NZ * NQP - NQ * MRP);
            return Ls.d_BS + (e - Ly.nJ);      // the defect (NX) and the stage scalars (4)
        };
#pragma unroll
        for (int j = 0; j < CST_PF; j++) {
            const int e = 2 * min(hl + 32 * j, c_n2 - 1);
            cdst[j] = dst_of(e) | (dst_of(e + 1) << 16);
        }
    }
    {
        // structural zeros of the row-major image (this scratch belongs to the row phases for the whole kernel)
        double* scr = dsm + Ls.o_scrd + hw * Ls.SCR_D;
        for (int el = hl; el < NRC * NZP; el += 32) scr[Ls.d_CST + el] = 0.0;
    }

    // ---- per-lane constants of the sequential phases (wavefront 0; cheap, so every wavefront computes them) -------------------
    // One wavefront alone issues an instruction every ~5 clocks whatever it is (scripts/lat_probe.hip, profiles/r06_lat_probe.txt):
    // the recursions are bound by their instruction COUNT.  So every lane-dependent LDS address is formed once, here.
    constexpr int XR = (NX + 3) / 4;          // rows of the x-x tile per lane: lane (g, c) owns rows g + 4 r of column c
    constexpr int XR_W = NX / 4;              // ... and row NX (W^T w) sits in register XR_W of the lanes of group NX % 4
    constexpr int FS = WgLds<NQ>::FS, LCOL = WgLds<NQ>::LCOL, NFW = WgLds<NQ>::NFW, NZS = WgLds<NQ>::NZS;
    double* const sP = dsm + Ls.s_P;
    double* const sPv = dsm + Ls.s_pv;
    // [Lambda | G]: element e = ln + 64 t is  H_e + sum of four entries of P_{k+1} with fixed coefficients (B^T P B, B^T P A in closed form)
    int lgo[2][4], lgd[2][2];
    double lgc[2][4];
    {
        const int dump = Ls.s_flag + 1;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int e = ln + 64 * t;
            lgo[t][0] = lgo[t][1] = lgo[t][2] = lgo[t][3] = 0;
            lgd[t][0] = lgd[t][1] = dump;
            lgc[t][0] = lgc[t][1] = lgc[t][2] = lgc[t][3] = 0.0;
            if (e < NTRI_U) {
                int i = 0, rem = e;
                while (rem >= NQ - i) { rem -= NQ - i; i++; }
                const int j = i + rem;
                lgo[t][0] = i * NX + j; lgo[t][1] = i * NX + NQ + j; lgo[t][2] = (NQ + i) * NX + j; lgo[t][3] = (NQ + i) * NX + NQ + j;
                lgc[t][0] = cB * cB; lgc[t][1] = cB * dt; lgc[t][2] = cB * dt; lgc[t][3] = dt * dt;
                lgd[t][0] = Ls.s_lam + i * NQ + j; lgd[t][1] = Ls.s_lam + j * NQ + i;
            } else if (e < HR::NE) {
                const int el = e - NTRI_U, i = el / NX, jx = el - i * NX;
                const bool right = jx >= NQ;
                const int jj = right ? jx - NQ : jx;
                lgo[t][0] = i * NX + jj; lgo[t][1] = (NQ + i) * NX + jj; lgo[t][2] = i * NX + jx; lgo[t][3] = (NQ + i) * NX + jx;
                if (right) { lgc[t][0] = dt * cB; lgc[t][1] = dt * dt; lgc[t][2] = cB; lgc[t][3] = dt; }
                else { lgc[t][0] = cB; lgc[t][1] = dt; }
                lgd[t][0] = lgd[t][1] = Ls.s_G + i * WS2 + jx;
            }
        }
    }
    // P update: entry (i, c) = (g + 4 r, c) through the expression of its upper-triangle twin (lo, hi), so that P stays exactly symmetric;
    // P is updated IN PLACE (every read of a stage is issued, and waited for, before its first write: one wavefront, program order)
    int lpo[XR][4], lpd[XR];
    double lpc[XR][2];
#pragma unroll
    for (int r = 0; r < XR; r++) {
        const int i = min(lg + 4 * r, NX - 1), c = min(lc, NX - 1);
        const int lo = min(i, c), hi = max(i, c);
        const int ii = lo >= NQ ? lo - NQ : lo, jj = hi >= NQ ? hi - NQ : hi;
        lpo[r][0] = lo * NX + hi; lpo[r][1] = lo * NX + jj; lpo[r][2] = ii * NX + hi; lpo[r][3] = ii * NX + jj;
        lpc[r][0] = hi >= NQ ? dt : 0.0;
        lpc[r][1] = lo >= NQ ? dt : 0.0;
        lpd[r] = (lg + 4 * r < NX && lc < NX) ? Ls.s_P + (lg + 4 * r) * NX + lc : Ls.s_flag + 1;
    }
    const int rho_dst = hl < NU ? Ls.s_G + hl * WS2 + NX : Ls.s_flag + 1;                  // rho of the Riccati phase, column NX of [G | rho]
    const int pv_dst_b = (lg == NX % 4 && lc < NX) ? Ls.s_pv + lc : Ls.s_pv + NX;          // p_k from the lanes that hold W^T w
    const double mfa0 = (hl <= NX && hl < 16 && lg == 0) ? 1.0 : 0.0, mfa1 = (hl <= NX && hl < 16 && lg == 2) ? 1.0 : 0.0;   // MFMA operand masks
    const int fac_col = hc <= NX ? hc : (hc >= LC0 && hc - LC0 < NQ ? LCOL + hc - LC0 : NX + 1);                        // this lane's column of the factor block (idle lanes: the pad)
    const int iu = hl_x < NQ ? hl_x : hl_x - NQ;                                            // control of the roll-out's lanes 0 .. NX-1
    const int z_dst_u = hl < NQ ? hl : NZP, z_dst_x = hl < NX ? NZS + NU + hl : NZP + 1;      // roll-out stores (idle lanes: the record's spare cells)
    const int pv_dst_e = (hl >= NU && hl < NZ) ? hl - NU : NX;                               // corrector costate

    // ---- initial residual norm and complementarity from the setup partials ------------------------------------------------
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
    const double dx0_reg = hl < NX ? x0[(size_t)b * NX + hl] - xb0[hl] : 0.0;

    double rho_lin = 1.0, alpha = 0.0, sigmu = 0.0, corr_w = 1.0;
    bool pending = false;
    int it = 0, st_code = 2;
    const double tol = D->qp_tol;
    const double tol_r = D->qp_tol_res > 0.0 ? D->qp_tol_res : D->qp_tol;
    const int max_iter = D->qp_max_iter;
    const int stall_max = D->qp_stall_iters;
    int stall = 0;
    if (tid == 0) sFlag[0] = 0.0;
    __syncthreads();
    WGT(13);

    for (it = 0; it < max_iter; it++) {
        if (mu <= tol && rho_lin * R0 <= tol_r) { st_code = 0; break; }

        // =============== phases A and B, overlapped by chunks of stages =====================================================
        // A (rows + the stage's P-independent blocks, any half-wave, any order) feeds B (the Riccati recursion on wavefront 0, stage N
        // down to 0).  All half-waves assemble the top chunk of QP_WG_CHUNK stages; from then on wavefront 0 factorises a chunk WHILE the
        // half-waves 2 .. NHW-1 assemble the one below it, a workgroup barrier between steps (no flags, no spinning).  With two
        // wavefronts per workgroup the two run at the same rate (8 stages of B ~ 20 k clocks, 8 stages of A on two half-waves ~ 20 k) and
        // the assembly all but disappears behind the recursion: A + B 118 k -> ~90 k clocks; with four wavefronts 98 k -> ~85 k.
        auto phaseA_stage = [&](int k, double* const scr) {
            double* const sIMG = scr;
            double* const sTT = sIMG + Ly.iTT;
            double* const sGT = sIMG + Ly.iGT;
            double* const sGN = sIMG + Ly.iGN;
            double* const sHQQ = sIMG + Ly.iHQQ;
            double* const sGZ = sIMG + Ly.iGZ;
            double* const sB = sIMG + Ly.iB;
            double* const sSC = sIMG + Ly.iSC;
            double* const sD = scr + Ls.a_D;
            double* const sE = scr + Ls.a_E;
            double* const sTD = scr + Ls.a_TD;
            double* const sGD = scr + Ls.a_GD;
            double mu_part = 0.0;
            {
                const bool last = (k == N);
                double* w = ws + (size_t)k * Ly.stride;
                double* hk = hrec + (size_t)k * HR::SIZE;
                dbl2 img[IMG_PF];
                {
                    const dbl2* s2 = reinterpret_cast<const dbl2*>(w + Ly.oIMG);
#pragma unroll
                    for (int j = 0; j < IMG_PF; j++) img[j] = s2[min(hl + 32 * j, img_n2 - 1)];
                }
                QpRow rs;
                {
                    const dbl2 r0 = reinterpret_cast<const dbl2*>(w + Ly.oR0)[hr], r1 = reinterpret_cast<const dbl2*>(w + Ly.oR1)[hr],
                               r2 = reinterpret_cast<const dbl2*>(w + Ly.oR2)[hr];
                    rs.lo = r0.x; rs.hi = r0.y; rs.tl = r1.x; rs.tu = r1.y; rs.ll = r2.x; rs.lu = r2.y;
                }
                const double czar = w[Ly.oCZA + hr], cznr = w[Ly.oCZN + hr], znc = w[Ly.oZN + hz];
                const dbl2 slb = *reinterpret_cast<const dbl2*>(w + Ly.oSL);
                double zc = w[Ly.oZ + hz];
                {
                    dbl2* d2 = reinterpret_cast<dbl2*>(sIMG);
#pragma unroll
                    for (int j = 0; j < IMG_PF; j++) d2[min(hl + 32 * j, img_n2 - 1)] = img[j];
                }
                const double wsoft = slb.x;
                {
                    const bool soft = soft_lane && wsoft >= 0.0;
                    double rr_ = 0.0, s1_ = 0.0, s2_ = 0.0;
                    const QpDir rd = qp_row_dir<false>(rs, soft, wsoft, cznr, sigmu, corr_w, czar, &rr_, &s1_, &s2_, nullptr, nullptr);
                    rs.tl += alpha * rd.dtl; rs.ll += alpha * rd.dll;
                    rs.tu += alpha * rd.dtu; rs.lu += alpha * rd.dlu;
                    reinterpret_cast<dbl2*>(w + Ly.oR1)[hr] = dbl2{rs.tl, rs.tu};
                    reinterpret_cast<dbl2*>(w + Ly.oR2)[hr] = dbl2{rs.ll, rs.lu};
                    double Dr;
                    sE[hr] = qp_row_coeff(rs, soft, wsoft, 0.0, 0.0, 0.0, &Dr);
                    sD[hr] = Dr;
                    mu_part = row_live ? qp_row_comp(rs, soft, wsoft) : 0.0;
                }
                zc += alpha * (znc - zc);
                w[Ly.oZ + hz] = zc;
                lds_fence();
                // the defect and its flag stay in LDS for the recursions
                if (hl < NX) sBk[k * NX + hl] = sB[hl];
                if (hl == 0) sBf[k] = (!last && slb.y != 0.0) ? 1.0 : 0.0;
                // rows scaled by their barrier weights
                if constexpr (NQ % 2 == 0) {
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
                        const int e2 = max(min(hl + 32 * t, NQ * (MRP >> 1) - 1), 0);
                        const int r2 = e2 % max(MRP >> 1, 1);
                        ga[t] = gt2[e2];
                        gd[t] = reinterpret_cast<const dbl2*>(sD + rC0)[r2];
                        if (2 * r2 + 1 >= MR) gd[t].y = 0.0;
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
                        sTD[el] = sTT[el] * sD[rT0 + (r < NQ ? r : 0)];
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
                // u-u triangle and u-x block of H + C^T D C (no torque rows at the end stage: the blocks are not used there)
                if (!last) {
#pragma unroll
                    for (int t = 0; t < (NTRI_U + 31) / 32; t++) {
                        const int el = min(hl + 32 * t, NTRI_U - 1);
                        const int i = triUi[el], j = triUj[el];
                        double a = pdot(sTT + i * NQP, sTD + j * NQP, NQP / 2);
                        a += i == j ? sSC[0] : 0.0;
                        hk[HR::oA + el] = a;
                    }
#pragma unroll
                    for (int t = 0; t < (NQ * NX + 31) / 32; t++) {
                        const int el = min(hl + 32 * t, NQ * NX - 1);
                        const int i = el / NX, jx = el - i * NX;
                        hk[HR::oA + NTRI_U + el] = pdot(sTT + i * NQP, sTD + (NU + jx) * NQP, NQP / 2);
                    }
                }
                // x-x triangle
#pragma unroll
                for (int t = 0; t < (NTRI_X + 31) / 32; t++) {
                    const int el = min(hl + 32 * t, NTRI_X - 1);
                    const int ix = triXi[el], jx = triXj[el];
                    const double dii = sD[ix], lmv = sSC[1], dnn = sD[rNN];
                    double a = pdot(sTT + (NU + ix) * NQP, sTD + (NU + jx) * NQP, NQP / 2);
                    a = fma(sGN[ix] * dnn, sGN[jx], a);
                    if (32 * t < NTRI_U) {
                        const int iq = min(ix, NQ - 1), jq = min(jx, NQ - 1);
                        const double qq = sHQQ[iq * NQ + jq] + pdot(sGT + iq * MRP, sGD + jq * MRP, MRP >> 1);
                        a += jx < NQ ? qq : 0.0;
                    }
                    a += ix == jx ? dii + (ix >= NQ ? lmv : 0.0) : 0.0;
                    hk[HR::oXX + 16 * ix + jx] = a;
                    hk[HR::oXX + 16 * jx + ix] = a;
                }
                // gradient g + C^T e
                {
                    const int ix = hl_px, iq = min(hl_px, NQ - 1);
                    const double g0 = sGZ[hz], eb = sE[ix], gn_i = sGN[ix], e_nn = sE[rNN];
                    const double gnn = gn_i * e_nn;
                    double tq = 0.0, cq = 0.0;
                    if constexpr (NQ % 2 == 0) {
                        tq = pdot(sTT + hz * NQP, sE + rT0, NQP / 2);
                        cq = pdot(sGT + iq * MRP, sE + rC0, MRP >> 1);
                    } else {
#pragma unroll
                        for (int r = 0; r < NQ; r++) tq = fma(sTT[hz * NQP + r], sE[rT0 + r], tq);
                        for (int r = 0; r < MR; r++) cq = fma(sGT[iq * MRP + r], sE[rC0 + r], cq);
                    }
                    const double gh = g0 + tq + (hz >= NU ? eb + gnn + (ix < NQ ? cq : 0.0) : 0.0);
                    hk[HR::oGH + hz] = gh;
                }
                lds_fence();      // (the next stage of this half-wave overwrites the scratch)
            }
            // the stage's share of sum(lambda t): summed over its rows here, over the stages in stage order at the end (the same bits
            // whatever half-wave took the stage)
            mu_part = half_sum(mu_part);
            if (hl == 0) sPart[k] = mu_part;
        };
        double dmin = 1.0;
        auto riccati_chunk = [&](int k_hi, int k_lo) {
            double Lr[NQ][NQ], Linv[NQ];
            double hA[2], hX[XR], hG, hGx;
            auto load_h = [&](int k) {
                const double* hk = hrec + (size_t)k * HR::SIZE;
                hA[0] = hk[HR::oA + ln];
                hA[1] = hk[HR::oA + 64 + ln];
#pragma unroll
                for (int r = 0; r < XR; r++) hX[r] = hk[HR::oXX + 16 * min(lg + 4 * r, NX - 1) + min(lc, NX - 1)];
                hG = hk[HR::oGH + hz];
                hGx = hk[HR::oGH + NU + min(lc, NX - 1)];
            };
            int k = k_hi;
            load_h(k);
            if (k == N) {
                // P_N = the x-x block, p_N = the x part of the gradient
#pragma unroll
                for (int r = 0; r < XR; r++) dsm[lpd[r]] = hX[r];
                dsm[Ls.s_pv + pv_dst_e] = hG;
                lds_fence();
                k = N - 1;
                if (k >= k_lo) load_h(k);
            }
            for (; k >= k_lo; k--) {
                double* const fk = sFac + k * NFW;
                const double a0 = hA[0], a1 = hA[1], gh = hG, ghx = hGx;
                double hx[XR];
#pragma unroll
                for (int r = 0; r < XR; r++) hx[r] = hX[r];
                if (k > k_lo) load_h(k - 1);      // (the chunk below is still being assembled: its first record is fetched after the barrier)
                const bool bfl = sBf[k] != 0.0;
                asm volatile("; WGMARK B_LG_BEGIN");
                // Lambda = Huu + B^T P B (triangle, mirrored) and G = Hux + B^T P A: one formula, four entries of P with this lane's
                // coefficients; rho = gh_u + B^T (p + P b)
                {
                    double pe[10];
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int q = 0; q < 4; q++) pe[4 * t + q] = sP[lgo[t][q]];
                    pe[8] = sPv[hl_u]; pe[9] = sPv[NQ + hl_u];
                    hold_n(pe);
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        double a = t == 0 ? a0 : a1;
#pragma unroll
                        for (int q = 0; q < 4; q++) a = fma(lgc[t][q], pe[4 * t + q], a);
                        dsm[lgd[t][0]] = a;
                        dsm[lgd[t][1]] = a;
                    }
                    double pb1 = 0.0, pb2 = 0.0;
                    if (bfl) {
                        pb1 = pdot(sP + hl_u * NX, sBk + k * NX, NX / 2);
                        pb2 = pdot(sP + (NQ + hl_u) * NX, sBk + k * NX, NX / 2);
                    }
                    dsm[rho_dst] = gh + cB * (pe[8] + pb1) + dt * (pe[9] + pb2);
                }
                lds_fence();
                asm volatile("; WGMARK B_CHOL_BEGIN");
                WGT(8);
                // Cholesky of Lambda in registers (every lane), one column of [W | w | L^-1] per lane
                double col[NQ];
                {
                    double A[NQ][NQ], cv[NQ];
                    const int cg = hc <= NX ? hc : NX;
#pragma unroll
                    for (int j = 0; j < NQ; j++)
#pragma unroll
                        for (int i = j; i < NQ; i++) A[i][j] = sLam[i * NQ + j];
#pragma unroll
                    for (int i = 0; i < NQ; i++) cv[i] = sG[i * WS2 + cg];
                    hold_n(cv);
#pragma unroll
                    for (int i = 0; i < NQ; i++) col[i] = hc <= NX ? cv[i] : (i == hc - LC0 ? 1.0 : 0.0);
#pragma unroll
                    for (int j = 0; j < NQ; j++) {
                        double dsum = A[j][j];
#pragma unroll
                        for (int t = 0; t < j; t++) dsum = fma(-Lr[j][t], Lr[j][t], dsum);
                        dmin = fmin(dmin, dsum);          // (a pivot <= 0 or NaN is reported once, after the sweep)
                        const double inv = fast_rsqrt(dsum);
                        Linv[j] = inv;
#pragma unroll
                        for (int i = j + 1; i < NQ; i++) {
                            double v = A[i][j];
#pragma unroll
                            for (int t = 0; t < j; t++) v = fma(-Lr[i][t], Lr[j][t], v);
                            Lr[i][j] = v * inv;
                        }
                        // (the column solve advances with the factor: its step j needs row j of L only)
                        double v = col[j];
#pragma unroll
                        for (int t = 0; t < j; t++) v = fma(-Lr[j][t], col[t], v);
                        col[j] = v * inv;
                    }
                }
                asm volatile("; WGMARK B_PUPD_BEGIN");
                WGT(9);
                double pr[4 * XR + 2];
                const int i_ = min(lc, NX - 1), i2_ = i_ >= NQ ? i_ - NQ : 0;
                v4d ww = {0.0, 0.0, 0.0, 0.0};
                double kc[NQ];
                {
                    // (at stage 0 too, where nothing uses them: a branch here would put the wait for the MFMAs in front of the
                    //  back-substitution below)
#pragma unroll
                    for (int r = 0; r < XR; r++)
#pragma unroll
                        for (int q = 0; q < 4; q++) pr[4 * r + q] = sP[lpo[r][q]];
                    pr[4 * XR] = sPv[i_]; pr[4 * XR + 1] = sPv[i2_];
                    hold_n(pr);
                    // [W | w]^T [W | w] as a 16 x 16 tile: lane (g, c) of the result holds rows g, g + 4, ... of column c.  A = B: the lanes of
                    // groups 0 and 2 (which hold the same columns, c = their lane in the group) feed two k-slices per instruction
#pragma unroll
                    for (int m = 0; m < (NQ + 1) / 2; m++) {
                        const double av = 2 * m + 1 < NQ ? fma(col[2 * m + 1 < NQ ? 2 * m + 1 : 0], mfa1, col[2 * m] * mfa0) : col[2 * m] * mfa0;
                        ww = __builtin_amdgcn_mfma_f64_16x16x4f64(av, av, ww, 0, 0, 0);
                    }
                }
                // What the roll-outs and the corrector's costate read is L^-T times this lane's column: the feedback gain K = L^-T W,
                // k = L^-T w, and, from the columns of L^-1, Lambda^-1 -- a back-substitution in registers, column-oriented (every
                // finished entry is taken out of all rows above it at once: a chain of 2 NQ - 1 dependent operations instead of
                // NQ (NQ + 1) / 2; nothing hides it -- the f64 MFMA runs on the FP64 pipeline of the vector unit, 78.6 TFLOP/s either way).
                // With it a roll-out's stage is u = -(K x + k), x+ = A x + B u + b with ONE hand-off (the state): lane i needs row
                // i mod NQ of K only; the costate's stage needs column i of K and a row of Lambda^-1 (the factored form
                // L^-T (W x + w) of k_qp_ipm costs a hand-off more per stage in each of the three recursions).
                {
                    double v[NQ];
#pragma unroll
                    for (int i = 0; i < NQ; i++) v[i] = col[i];
#pragma unroll
                    for (int t = NQ - 1; t >= 0; t--) {
                        kc[t] = v[t] * Linv[t];
#pragma unroll
                        for (int i = 0; i < t; i++) v[i] = fma(-Lr[t][i], kc[t], v[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < NQ; i++) fk[i * FS + fac_col] = kc[i];
                if (k > 0) {
                    // P_k = Hxx + A^T P A - W^T W,  p_k = gh_x + A^T (p + P b) - W^T w
                    double pb1 = 0.0, pb2 = 0.0;
                    if (bfl) {
                        pb1 = pdot(sP + i_ * NX, sBk + k * NX, NX / 2);
                        pb2 = pdot(sP + i2_ * NX, sBk + k * NX, NX / 2);
                    }
#pragma unroll
                    for (int r = 0; r < XR; r++) {
                        double a = hx[r] - ww[r];
                        a += pr[4 * r] + lpc[r][0] * pr[4 * r + 1] + lpc[r][1] * (pr[4 * r + 2] + dt * pr[4 * r + 3]);
                        dsm[lpd[r]] = a;
                    }
                    {
                        const double q1 = pr[4 * XR] + pb1, q2 = pr[4 * XR + 1] + pb2;
                        double v = ghx + q1 + (i_ >= NQ ? dt * q2 : 0.0);
                        v -= ww[XR_W];
                        dsm[pv_dst_b] = v;
                    }
                    lds_fence();
                }
                asm volatile("; WGMARK B_PUPD_END");
                WGT(10);
            }
            if (!(half_min(dmin) > 0.0) && hl == 0) sFlag[0] = 1.0;
        };
        {
            constexpr int CH = QP_WG_CHUNK;
            double* const scr_a = hw < 2 ? dsm + hw * Ls.SCR_A : dsm + Ls.o_scra2 + (hw - 2) * Ls.SCR_A;
            int k_hi = N, k_lo = max(N - CH + 1, 0);
            for (int k = k_hi - hw; k >= k_lo; k -= NHW) phaseA_stage(k, scr_a);
            __syncthreads();
            WGT(0);
            for (;;) {
                const int n_hi = k_lo - 1, n_lo = max(n_hi - CH + 1, 0);
                if (seq) riccati_chunk(k_hi, k_lo);
                else if (n_hi >= 0)
                    for (int k = n_hi - (hw - 2); k >= n_lo; k -= NHW - 2) phaseA_stage(k, scr_a);
                __syncthreads();
                if (n_hi < 0) break;
                k_hi = n_hi; k_lo = n_lo;
            }
        }
        WGT(1);
        if (pending) {
            double acc = 0.0;
            for (int k = 0; k <= N; k++) acc += sPart[k];
            mu = acc * inv_m;
            pending = false;
            if (!(mu == mu)) { st_code = 4; break; }
        }
        if (sFlag[0] != 0.0) { st_code = 4; pending = false; break; }

        // roll-out through the gains: z_k = [u_k | x_k] for every stage (wavefront 0).  Every operand of a stage is read in one batch at
        // its top; lane i (state entry i) forms the control of its joint, u = -(K x + k [+ Lambda^-1 r of the corrector]), from row
        // i mod NQ of [K | k] by itself, so the only hand-off of a stage is the state.
        auto rollout = [&](auto corr_tag, int k_lo, int k_hi) {
            constexpr bool CORR = decltype(corr_tag)::value;
            asm volatile("; WGMARK ROLL_BEGIN");
            if (k_lo == 0) {
                if (hl < NX) sZ[NU + hl] = dx0_reg;
                lds_fence();
            }
            for (int k = k_lo; k < k_hi; k++) {
                const double* fk = sFac + k * NFW;
                double* zk = sZ + k * NZS;
                double krow[NX + 2], xs[NX], ex[4];
#pragma unroll
                for (int j = 0; j < (NX + 2) / 2; j++) {
                    const dbl2 v = reinterpret_cast<const dbl2*>(fk + iu * FS)[j];       // row iu of [K | k | pad]
                    krow[2 * j] = v.x; krow[2 * j + 1] = v.y;
                }
                if constexpr (NQ % 2 == 0) {
#pragma unroll
                    for (int j = 0; j < NX / 2; j++) {
                        const dbl2 v = reinterpret_cast<const dbl2*>(zk + NU)[j];
                        xs[2 * j] = v.x; xs[2 * j + 1] = v.y;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NX; j++) xs[j] = zk[NU + j];
                }
                ex[0] = zk[NU + hl_x]; ex[1] = zk[NU + NQ + iu];
                ex[2] = sBk[k * NX + hl_x];
                ex[3] = CORR ? sWC[k * NQP + iu] : 0.0;
                hold_n(krow); hold_n(xs); hold_n(ex);
                double a = krow[NX] + ex[3], a_v = 0.0;
#pragma unroll
                for (int j = 0; j < NQ; j++) {
                    a = fma(krow[j], xs[j], a);
                    a_v = fma(krow[NQ + j], xs[NQ + j], a_v);
                }
                const double u = -(a + a_v);
                zk[z_dst_u] = u;
                zk[z_dst_x] = hl < NQ ? ex[0] + dt * ex[1] + cB * u + ex[2] : ex[0] + dt * u + ex[2];
                lds_fence();
            }
            if (k_hi == N && hl < NQ) sZ[N * NZS + hl] = 0.0;      // no control at the end stage
            asm volatile("; WGMARK ROLL_END");
        };
        // Roll-out (wavefront 0) and the rows of every stage for it (ratio test, sums; predictor: c.z_aff, a1 / a2; corrector: c.z+, z+),
        // overlapped the way phases A and B are: the roll-out advances by chunks of QP_WG_RCH stages with a workgroup barrier after
        // each, and WHILE wavefront 0 rolls out a chunk the half-waves 2 .. NHW-1 take the rows of stages that are complete, RND
        // rounds per step, in increasing stage order; what is left when the roll-out is through is shared by all half-waves.  Which
        // half-wave takes which stage is a function of (N, NHW) alone.  The sums of a lane over the stages of a group (k mod 8) live
        // in LDS cells, added in increasing k whoever processes the stage -- so they do not depend on the schedule or on NHW.
        auto rows_phase = [&](auto corr_tag, double* rr_out, double* S1_out, double* S2_out) {
            constexpr bool CORR = decltype(corr_tag)::value;
            constexpr int NHELP = NHW - 2, RND = NHW == 4 ? QP_WG_RND4 : QP_WG_RND8, RCH = NHW == 4 ? QP_WG_RCH4 : QP_WG_RCH8;
            static_assert(RND * NHELP <= NVG && RCH >= 1, "a step's stages must lie in different groups (their cells are shared)");
            double rr = 0.0;
            double* const scr = dsm + Ls.o_scrd + hw * Ls.SCR_D;
            double* const sD = scr + Ls.d_D;
            double* const sE = scr + Ls.d_E;
            double* const sBS = scr + Ls.d_BS;
            double* const sCst = scr + Ls.d_CST;
            dbl2* const sAcc = reinterpret_cast<dbl2*>(dsm + Ls.o_scra2);
            dbl2 Cs[CST_PF], qr0, qr1, qr2;
            double qcza = 0.0;
            int k_pref = -1;                       // the stage whose records sit in (Cs, qr*, qcza)
            auto load_r = [&](int k) {
                const double* w = ws + (size_t)k * Ly.stride;
                const dbl2* s2 = reinterpret_cast<const dbl2*>(w + Ly.oIMG);
#pragma unroll
                for (int j = 0; j < CST_PF; j++) Cs[j] = s2[min(hl + 32 * j, c_n2 - 1)];
                qr0 = reinterpret_cast<const dbl2*>(w + Ly.oR0)[hr]; qr1 = reinterpret_cast<const dbl2*>(w + Ly.oR1)[hr];
                qr2 = reinterpret_cast<const dbl2*>(w + Ly.oR2)[hr];
                if (CORR) qcza = w[Ly.oCZA + hr];
                k_pref = k;
            };
            // the rows of stage k; k_next: the stage this half-wave expects to take after it (requested now, a whole stage ahead)
            auto do_stage = [&](int k, int k_next) {
                if (k_pref != k) load_r(k);        // (an irregular step of the schedule: the request was for another stage)
                double* w = ws + (size_t)k * Ly.stride;
                const double* zk = sZ + k * NZS;
#pragma unroll
                for (int j = 0; j < CST_PF; j++) {
                    scr[cdst[j] & 0xffff] = Cs[j].x;
                    scr[cdst[j] >> 16] = Cs[j].y;
                }
                const dbl2 r0 = qr0, r1 = qr1, r2 = qr2;
                const double cza = qcza;
                load_r(k_next);
                lds_fence();
                {
                    // (every LDS operand of the row's value in ONE batch: the row of the Jacobian image, the stage's z, the soft weight, this
                    //  lane's own state entry, its sum cell -- left alone the compiler makes three or four round trips of them)
                    const double* cr = sCst + hl_c * NZP;
                    dbl2* const cell = sAcc + (k & (NVG - 1)) * 32 + hl;
                    double cv[NZ], zv[NZ], ex[5];
#pragma unroll
                    for (int j = 0; j < NZ / 2; j++) {
                        const dbl2 v = reinterpret_cast<const dbl2*>(cr)[j], z2 = reinterpret_cast<const dbl2*>(zk)[j];
                        cv[2 * j] = v.x; cv[2 * j + 1] = v.y;
                        zv[2 * j] = z2.x; zv[2 * j + 1] = z2.y;
                    }
                    if constexpr (NZ % 2) { cv[NZ - 1] = cr[NZ - 1]; zv[NZ - 1] = zk[NZ - 1]; }
                    {
                        const dbl2 acc0 = *cell;
                        ex[0] = sBS[NX + 2]; ex[1] = zk[NU + hl_x]; ex[2] = acc0.x; ex[3] = acc0.y; ex[4] = zk[hz];
                    }
                    hold_n(cv); hold_n(zv); hold_n(ex);
                    double a = 0.0, a_q = 0.0, a_v = 0.0;
#pragma unroll
                    for (int c = 0; c < NQ; c++) {
                        a = fma(cv[c], zv[c], a);
                        a_q = fma(cv[NU + c], zv[NU + c], a_q);
                        a_v = fma(cv[NU + NQ + c], zv[NU + NQ + c], a_v);
                    }
                    a += a_q + a_v;
                    const double cz = hr < NX ? ex[1] : a;
                    const QpRow rs{r0.x, r0.y, r1.x, r1.y, r2.x, r2.y};
                    const double wsoft = ex[0];
                    const bool soft = soft_lane && wsoft >= 0.0;
                    double s1_ = 0.0, s2_ = 0.0;
                    if (!CORR) {
                        double e1, e2;
                        qp_row_dir<true>(rs, soft, wsoft, cz, 0.0, 0.0, cz, &rr, &s1_, &s2_, &e1, &e2);
                        w[Ly.oCZA + hr] = cz;
                        sD[hr] = e1;
                        sE[hr] = e2;
                    } else {
                        qp_row_dir<false>(rs, soft, wsoft, cz, sigmu, corr_w, cza, &rr, &s1_, &s2_, nullptr, nullptr);
                        w[Ly.oCZN + hr] = cz;
                        w[Ly.oZN + hz] = ex[4];
                    }
                    dbl2 acc = dbl2{ex[2], ex[3]};
                    if (k < NVG) acc = dbl2{0.0, 0.0};       // (the first stage of its group: nothing to zero beforehand)
                    acc.x += row_live ? s1_ : 0.0;
                    acc.y += row_live ? s2_ : 0.0;
                    *cell = acc;
                }
                if (!CORR) {
                    lds_fence();
                    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0, xd[2];
                    {
                        constexpr int RH = (NRC_MAX + 1) / 2;
                        double cc[RH], d1[RH], d2[RH];
                        const int ixd = hz >= NU ? hz - NU : 0;
                        xd[0] = sD[ixd]; xd[1] = sE[ixd];
#pragma unroll
                        for (int hh = 0; hh < 2; hh++) {
#pragma unroll
                            for (int r = 0; r < RH; r++) {
                                const int rr_ = min(min(hh * RH + r, NRC_MAX - 1), NRC - 1);
                                cc[r] = sCst[rr_ * NZP + hz];
                                d1[r] = sD[NX + rr_];
                                d2[r] = sE[NX + rr_];
                            }
                            hold_n(cc); hold_n(d1); hold_n(d2);
                            if (hh == 0) hold_n(xd);
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
                    if (hz >= NU) { a1 += xd[0]; a2 += xd[1]; }
                    if (hl < NZ) reinterpret_cast<dbl2*>(sA12 + k * 2 * NZP)[hz] = dbl2{a1, a2};
                }
                lds_fence();
            };
            // (two loops with the same number of barriers behind a branch the hardware takes as a whole wavefront: the registers that
            //  carry a half-wave's requested records from step to step are then not alive inside the roll-out)
            int hdone = 0;                         // stages whose rows are done (always the stages 0 .. hdone-1)
            if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
                for (int k_lo = 0; k_lo < N; k_lo += RCH) {
                    if (tid < QP_WG_SEQ_LANES) rollout(corr_tag, k_lo, min(k_lo + RCH, N));
                    hdone += min(k_lo - hdone, RND * NHELP);
                    __syncthreads();
                }
                load_r(min(hdone + hw, N));        // this half-wave's first stage of what is left
            } else {
                const int h = hw - 2;
                load_r(min(h, N));
                for (int k_lo = 0; k_lo < N; k_lo += RCH) {
                    const bool last = k_lo + RCH >= N;
                    const int n = min(k_lo - hdone, RND * NHELP);      // (stages below k_lo are rolled out)
#pragma unroll
                    for (int i = 0; i < RND; i++) {
                        const int k = hdone + h + i * NHELP;
                        if (k < hdone + n) {
                            int kn = k + NHELP;
                            if (i == RND - 1 || kn >= hdone + n) kn = hdone + n + (last ? hw : h);
                            do_stage(k, min(kn, N));
                        }
                    }
                    hdone += n;
                    __syncthreads();
                }
            }
            WGT(CORR ? 5 : 2);
            for (int k = hdone + hw; k <= N; k += NHW) do_stage(k, min(k + NHW, N));
            __syncthreads();
            double S1[GPH], S2[GPH];
#pragma unroll
            for (int g = 0; g < GPH; g++) {
                const int grp = hw + g * NHW;
                const dbl2 v = sAcc[grp * 32 + hl];
                S1[g] = grp <= N ? v.x : 0.0;
                S2[g] = grp <= N ? v.y : 0.0;
            }
            block_reduce(rr, S1, S2);
            *rr_out = rr; *S1_out = S1[0]; *S2_out = S2[0];
        };

        // =============== phases C, D: predictor ============================================================================
        double rr_aff, S1, S2;
        rows_phase(std::false_type{}, &rr_aff, &S1, &S2);
        WGT(3);
        const double a_aff = rr_aff > 1.0 ? 1.0 / rr_aff : 1.0;
        const double mu_aff = (mu * (double)m_comp + a_aff * S1 + a_aff * a_aff * S2) * inv_m;
        double sigma = mu_aff / mu;
        sigma = fmin(sigma * sigma * sigma, 0.3);
        sigmu = sigma * mu;
        corr_w = a_aff >= 0.3 ? 1.0 : a_aff * a_aff;

        // =============== phase E: corrector costate (wavefront 0) =========================================================
        // Its gradient sigma mu a1 + cw a2 first, for every stage at once (all threads): the control part into the corrector's cells,
        // the state part over a1.  Then the recursion p_k = gh_x + A^T p - K^T r, r = gh_u + B^T p, with the control offset of the
        // corrector's roll-out, Lambda^-1 r, on the way: every lane forms r by itself from the costate it has read anyway, so the one
        // hand-off of a stage is the costate (in place).
        for (int idx = tid; idx < (N + 1) * NZ; idx += NT) {
            const int k = idx / NZ, c = idx - k * NZ;
            const dbl2 a12 = reinterpret_cast<const dbl2*>(sA12 + k * 2 * NZP)[c];
            const double gh = sigmu * a12.x + corr_w * a12.y;
            if (c < NU) { if (k < N) sWC[k * NQP + c] = gh; }
            else sA12[(k * NZP + c) * 2] = gh;
        }
        __syncthreads();
        if (tid < QP_WG_SEQ_LANES) {
            asm volatile("; WGMARK E_BEGIN");
            const int ip1 = hl < NU ? hl : hl_px;
            const int ip2 = hl < NU ? NQ + hl : (hl_px >= NQ ? hl_px - NQ : 0);
            double* const wc_dst = hl < NQ ? sWC + hl : sPv + NX + 1;        // (idle lanes: the costate's spare cell)
            const int wc_str = hl < NQ ? NQP : 0;
            sPv[pv_dst_e] = sA12[(N * NZP + hz) * 2];
            lds_fence();
            for (int k = N - 1; k >= 0; k--) {
                const double* fk = sFac + k * NFW;
                double pv[NX], gu[NQP], kcl[NQ], lam[NQ], ex[3];
#pragma unroll
                for (int j = 0; j < NX / 2; j++) {
                    const dbl2 v = reinterpret_cast<const dbl2*>(sPv)[j];
                    pv[2 * j] = v.x; pv[2 * j + 1] = v.y;
                }
#pragma unroll
                for (int j = 0; j < NQP / 2; j++) {
                    const dbl2 v = reinterpret_cast<const dbl2*>(sWC + k * NQP)[j];
                    gu[2 * j] = v.x; gu[2 * j + 1] = v.y;
                }
#pragma unroll
                for (int t = 0; t < NQ; t++) kcl[t] = fk[t * FS + hl_px];                           // column hl_px of K
                if constexpr (NQ % 2 == 0) {
#pragma unroll
                    for (int j = 0; j < NQ / 2; j++) {
                        const dbl2 v = reinterpret_cast<const dbl2*>(fk + hl_u * FS + LCOL)[j];      // row hl_u of Lambda^-1
                        lam[2 * j] = v.x; lam[2 * j + 1] = v.y;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NQ; j++) lam[j] = fk[hl_u * FS + LCOL + j];
                }
                ex[0] = sA12[(k * NZP + hz) * 2]; ex[1] = sPv[ip1]; ex[2] = sPv[ip2];
                hold_n(pv); hold_n(gu); hold_n(kcl); hold_n(lam); hold_n(ex);
                double kr = 0.0, lr = 0.0;
#pragma unroll
                for (int j = 0; j < NQ; j++) {
                    const double r = gu[j] + cB * pv[j] + dt * pv[NQ + j];
                    kr = fma(kcl[j], r, kr);
                    lr = fma(lam[j], r, lr);
                }
                wc_dst[k * wc_str] = lr;
                if (k > 0) {
                    sPv[pv_dst_e] = ex[0] + (hl_px < NQ ? ex[1] : dt * ex[2] + ex[1]) - kr;
                    lds_fence();
                }
            }
            asm volatile("; WGMARK E_END");
        }
        __syncthreads();
        WGT(4);
        // =============== phases F, G: corrector =============================================================================
        double rr_max, S1c, S2c;
        rows_phase(std::true_type{}, &rr_max, &S1c, &S2c);
        WGT(6);
        const double a_max = rr_max > 0.0 ? 1.0 / rr_max : 1e300;
        const double tau_k = a_max >= 0.99 ? fmin(0.9999, fmax(QP_FTB, 1.0 - mu)) : QP_FTB;
        alpha = fmin(1.0, tau_k * a_max);
        if (!(alpha == alpha)) { st_code = 4; break; }
        if (alpha < QP_ALPHA_MIN) { st_code = 3; break; }
        pending = true;
        rho_lin *= (1.0 - alpha);
        const double mu_before = mu;
        mu = (mu * (double)m_comp + alpha * S1c + alpha * alpha * S2c) * inv_m;
        if (!(mu == mu)) { st_code = 4; pending = false; break; }
        {
            const bool stalled = alpha < 0.5 && !(mu < 0.5 * mu_before);
            stall = stalled ? stall + 0x10001 : (stall & ~0xffff);
        }
        if (stall_max > 0 && ((stall & 0xffff) >= stall_max || (stall >> 16) >= stall_max + (stall_max + 5) / 6) &&
            !(mu <= tol && rho_lin * R0 <= tol_r)) { st_code = 5; it++; break; }
    }
    if (it == max_iter && st_code == 2 && mu <= tol && rho_lin * R0 <= tol_r) st_code = 0;

    // ---- full SQP step, applying the last IPM step if it is still pending (stage-parallel) --------------------------------
    __syncthreads();
    double bad = 0.0;
    const double a_fin = pending ? alpha : 0.0;
    for (int idx = tid; idx < (N + 1) * NZ; idx += NT) {
        const int k = idx / NZ, c = idx - k * NZ;
        const double* w = ws + (size_t)k * Ly.stride;
        const double zz = w[Ly.oZ + c], zzn = w[Ly.oZN + c];
        if (c < NU) {
            if (k < N) {
                const double v = ub0[(size_t)k * NU + c] + zz + a_fin * (zzn - zz);
                u_out[((size_t)b * N + k) * NU + c] = v;
                bad = !(v == v) ? 1.0 : bad;
            }
        } else {
            const double v = xb0[(size_t)k * NX + c - NU] + zz + a_fin * (zzn - zz);
            x_out[((size_t)b * (N + 1) + k) * NX + c - NU] = v;
            bad = !(v == v) ? 1.0 : bad;
        }
    }
    {
        double d1[GPH] = {}, d2[GPH] = {};
        block_reduce(bad, d1, d2);
    }
    WGT(12);
#ifdef QP_PROFILE
    if (tid == 0) {
        for (int i = 0; i < 14; i++) atomicAdd(&g_wg_prof[i], tacc[i]);
        atomicAdd(&g_wg_prof[14], 1ull);
        atomicAdd(&g_wg_prof[15], (unsigned long long)it);
    }
#endif
    if (tid == 0) {
        int stc = (st_code == 0 || st_code == 2) ? SMPC_STATUS_SUCCESS : SMPC_STATUS_QP_FAILURE;
        if (ws[Ly.oPART + 3] != 0.0) stc = SMPC_STATUS_QP_FAILURE;
        if (bad > 0.0 && stc == SMPC_STATUS_SUCCESS) stc = SMPC_STATUS_NAN;
        status[b] = stc;
        if (qp_iter) qp_iter[b] = it;
        if (last_iter) last_iter[b] = it;
        if (it_hist) atomicAdd(&it_hist[min(it, 255)], 1);
    }
}

}  // namespace smpc
