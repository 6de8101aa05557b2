// engine.hip -- C ABI (include/smpc.h) of the MI355X batched safe-MPC engine: handle, device buffers, kernel launches.
// Built for gfx950 only:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC engine.hip -o libsmpc_hip.so
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/smpc.h"
#include "kernel_qp.hpp"
#include "kernel_qp_wg.hpp"
#include "kernel_build.hpp"
#include "kernels_callers.hpp"
#include "kernels_policy.hpp"
#include "kernels_mlp.hpp"
#include "kernels_nodes.hpp"

using namespace smpc;

namespace {
thread_local char g_create_err[256] = "";
}

struct smpc_handle {
    smpc_problem_desc desc;
    int device = 0;
    int N = 0;
    hipStream_t stream = nullptr;
    smpc_problem_desc* d_desc = nullptr;
    double *d_lo = nullptr, *d_hi = nullptr;  // [N+1][nx] stage bounds
    double* d_zl = nullptr;                   // [N+1] run-time slack weights of the soft safe-set rows (cost_set), or null
    double *d_lo_b = nullptr, *d_hi_b = nullptr;  // [B][N+1][nx] per-instance bounds (RealReceding), valid for inst_B
    int inst_B = 0;
    size_t inst_cap = 0;   // doubles allocated in d_lo_b / d_hi_b
    // network
    int nlayers = 0;
    int act = SMPC_ACT_GELU_TANH;
    int dims[SMPC_MAX_LAYERS + 1] = {0};
    int H = 0;
    float* d_Wfwd[SMPC_MAX_LAYERS] = {nullptr};  // [K][N] = W^T (layer 0 padded to MLP_KPAD rows)
    float* d_Wbwd[SMPC_MAX_LAYERS] = {nullptr};  // [out][in] as given (layer 0 padded to MLP_NPAD columns)
    float* d_bias[SMPC_MAX_LAYERS] = {nullptr};
    // per-batch scratch, grown on demand
    int capB = 0;
    double* d_ev = nullptr;       // linearisation records of the last call, interleaved tiles of EV_TILE nodes (device_model.hpp)
    double* d_nn = nullptr;       // value and gradient of the network's row (read by the stage builder): [B][N+1][1 + nx] with the row
                                  // on every node, [B][1 + nx] with the row on the end node only, not allocated without a network row
    bool ev_new_order[64] = {};   // per slot of the event ring: the solve ran MLP -> stage builder (ev1, ev2 swap their meaning)
    double* d_ws = nullptr;
    size_t ws_bytes = 0;
    int qp_mode = -2;             // smpc_set_qp_mode: SMPC_QP_AUTO / _THROUGHPUT / _LATENCY; -2 = not set (the process default, SMPC_QP_WG)
    double* d_hrec = nullptr;     // k_qp_ipm_wg only: the stages' P-independent blocks, [B][N+1][HRecLayout::SIZE] (allocated on first use)
    size_t hrec_doubles = 0;
    size_t wg_lds_set[2] = {0, 0};    // dynamic-LDS limit already raised for this handle's instantiation of k_qp_ipm_wg (8 / 4 half-waves)
    int32_t *d_order = nullptr, *d_last_it = nullptr;  // longest-first dispatch order from the previous call's iterations
    int order_B = 0;                                   // batch size d_last_it is valid for (0 = none yet)
    int32_t* d_ord_hist = nullptr;                     // [256] histogram of d_last_it (k_qp_ipm) | [256] bin cursors | ticket (k_order_by_iters)
    // staging for host-pointer calls
    int capIO = 0;
    double *d_x0 = nullptr, *d_xg = nullptr, *d_ug = nullptr, *d_p = nullptr, *d_xo = nullptr, *d_uo = nullptr;
    int32_t *d_st = nullptr, *d_it = nullptr;
    // MLP activations
    size_t capM = 0;
    float *d_S = nullptr, *d_y = nullptr, *d_GS = nullptr, *d_dA = nullptr, *d_dB = nullptr;
    int32_t *d_nn_idx = nullptr, *d_nn_cnt = nullptr;   // compacted list of the nodes whose safe-set row is on + its length
                                // INVARIANT: *d_nn_cnt is zero whenever no chain of kernels is using it.  Every chain that fills it ends in
                                // something that hands it back at zero -- on the solve path a kernel that runs anyway (k_stage_build after the
                                // network pass, k_policy_post after the safe-set test), elsewhere a memset -- so the hot path has no memset
                                // launch of its own, and a captured step can be replayed whatever ran in between
    size_t nn_idx_cap = 0;
    float* d_act[SMPC_MAX_LAYERS] = {nullptr};
    float* d_dg[SMPC_MAX_LAYERS] = {nullptr};
    // generic scratch for the caller entry points
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0;
    double* d_chk = nullptr;    // check bounds of smpc_check_trajectory [x_min | x_max | row_lb | row_ub], uploaded on change
    std::vector<double> chk_cache;
    char* d_roll = nullptr;     // staging of smpc_rollout_batch's host-pointer path, grown on demand
    size_t roll_bytes = 0;
    // timing
    int timing = 0;
    int timed = 0;              // a solve has been timed since timing was enabled
    // a ring of timing-event sets, one per solve: a loop that enqueues far ahead of the GPU reads the per-kernel times of its
    // last EV_RING solves afterwards (smpc_get_timing_history), without a synchronisation inside the loop
    static constexpr int EV_RING = 64;
    hipEvent_t ev_sets[EV_RING][5] = {};
    bool ev_complete[EV_RING] = {};   // all five events of the slot were recorded by ONE solve (cleared when the slot is reused)
    hipEvent_t* ev_t = ev_sets[0];
    int ev_cur = 0;
    bool timing_now = false;    // this solve records its events (timing on and the stream is not being captured into a graph)
    long timed_count = 0;       // solves timed since timing was enabled
    // sub-batch workers of smpc_rollout_batch: full handles on their own streams that borrow this handle's network weights
    std::vector<smpc_handle*> kids;
    bool borrowed_mlp = false;            // (a worker: the weight buffers belong to its parent)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    unsigned long long* d_wstat = nullptr;   // [4] load-balance probe of k_qp_ipm (timing builds of a call only)
    const uint8_t* d_active = nullptr;   // smpc_policy_step only: instances the QP kernels skip (borrowed for the call)
    int32_t* d_polw = nullptr;  // scratch of the policy entry points: verdicts, masks, the next state of the plant
    size_t polw_bytes = 0;
    int32_t* d_pol = nullptr;   // fails / accept counters of smpc_rollout_batch, [2][pol_B]
    int pol_B = 0;
    float last_ms[4] = {0, 0, 0, 0};
    long mlp_rows_whole = 0;    // > 0 (a worker of smpc_rollout_batch): network rows of the WHOLE call, which selects the GEMM kernel
    long mlp_rows_hint = 0;     // > 0 (smpc_policy_step of the receding policies): the rows EXPECTED to be live in a compacted list -- one or
                                // two nodes per instance, where the list's capacity is every node -- which selects the network kernel
                                // (the count itself is only known on the device; any kernel is correct for any count)
    char err[256] = "";
};

namespace {

int fail(smpc_handle* h, int code, const char* fmt, ...) {
    char* dst = h ? h->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 256, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(h, expr)                                                                                       \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return fail(h, SMPC_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));        \
    } while (0)

template <class T> int dev_alloc(smpc_handle* h, T** p, size_t count) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (count == 0) return SMPC_OK;
    hipError_t e = hipMalloc((void**)p, count * sizeof(T));
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(h, SMPC_ENOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    }
    return SMPC_OK;
}

size_t ws_doubles_per_instance(const smpc_problem_desc& d, int N) {
    switch (d.nq) {
    case 5: return QpLayout<5>(d.n_rows).per_instance(N);
    case 6: return QpLayout<6>(d.n_rows).per_instance(N);
    default: return QpLayout<7>(d.n_rows).per_instance(N);
    }
}

int upload_bounds(smpc_handle* h, const double* lo, const double* hi) {
    const int nx = 2 * h->desc.nq;
    std::vector<double> l((size_t)(h->N + 1) * nx), u((size_t)(h->N + 1) * nx);
    for (int k = 0; k <= h->N; k++)
        for (int i = 0; i < nx; i++) {
            l[(size_t)k * nx + i] = lo ? lo[(size_t)k * nx + i] : (k == h->N ? h->desc.x_lo_e[i] : h->desc.x_lo[i]);
            u[(size_t)k * nx + i] = hi ? hi[(size_t)k * nx + i] : (k == h->N ? h->desc.x_hi_e[i] : h->desc.x_hi[i]);
        }
    int rc;
    if ((rc = dev_alloc(h, &h->d_lo, l.size()))) return rc;
    if ((rc = dev_alloc(h, &h->d_hi, u.size()))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_lo, l.data(), l.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_hi, u.data(), u.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SMPC_OK;
}

int ensure_batch(smpc_handle* h, int B) {
    const size_t per = ws_doubles_per_instance(h->desc, h->N);
    const size_t need = per * (size_t)B * sizeof(double);
    if (B > h->capB || need > h->ws_bytes) {
        int rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if ((rc = dev_alloc(h, &h->d_ev, ev_tiles((size_t)B * (h->N + 1)) * EV_TILE * EV_D))) return rc;
        if ((rc = dev_alloc(h, &h->d_ws, per * (size_t)B))) return rc;
        {
            const size_t nn_nodes = h->desc.nn_mode == SMPC_NN_NONE ? 0 : (h->desc.nn_mode == SMPC_NN_TERMINAL ? (size_t)B : (size_t)B * (h->N + 1));
            if ((rc = dev_alloc(h, &h->d_nn, nn_nodes * (1 + 2 * h->desc.nq)))) return rc;
            // (entries beyond n_dof_safe_set are never written and must read as zero)
            if (nn_nodes) HIPCHK(h, hipMemsetAsync(h->d_nn, 0, sizeof(double) * nn_nodes * (1 + 2 * h->desc.nq), h->stream));
        }
        if ((rc = dev_alloc(h, &h->d_order, (size_t)B))) return rc;
        if ((rc = dev_alloc(h, &h->d_last_it, (size_t)B))) return rc;
        if ((rc = dev_alloc(h, &h->d_ord_hist, (size_t)520))) return rc;
        HIPCHK(h, hipMemsetAsync(h->d_ord_hist, 0, 520 * sizeof(int32_t), h->stream));
        h->order_B = 0;
        h->ws_bytes = need;
        h->capB = B;
        h->capIO = 0;
    }
    return SMPC_OK;
}

int ensure_io(smpc_handle* h, int B) {
    if (B <= h->capIO) return SMPC_OK;
    const int nx = 2 * h->desc.nq, nu = h->desc.nq, NN = h->N;   // (smpc_set_horizon resets capIO)
    int rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((rc = dev_alloc(h, &h->d_x0, (size_t)B * nx))) return rc;
    if ((rc = dev_alloc(h, &h->d_xg, (size_t)B * (NN + 1) * nx))) return rc;
    if ((rc = dev_alloc(h, &h->d_ug, (size_t)B * NN * nu))) return rc;
    if ((rc = dev_alloc(h, &h->d_p, (size_t)B * (NN + 1) * SMPC_NP))) return rc;
    if ((rc = dev_alloc(h, &h->d_xo, (size_t)B * (NN + 1) * nx))) return rc;
    if ((rc = dev_alloc(h, &h->d_uo, (size_t)B * NN * nu))) return rc;
    if ((rc = dev_alloc(h, &h->d_st, (size_t)B))) return rc;
    if ((rc = dev_alloc(h, &h->d_it, (size_t)B))) return rc;
    h->capIO = B;
    return SMPC_OK;
}

int ensure_tmp(smpc_handle* h, size_t bytes) {
    if (bytes <= h->tmp_bytes) return SMPC_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_tmp) { (void)hipFree(h->d_tmp); h->d_tmp = nullptr; }
    hipError_t e = hipMalloc(&h->d_tmp, bytes);
    if (e != hipSuccess) return fail(h, SMPC_ENOMEM, "hipMalloc(%zu) failed", bytes);
    h->tmp_bytes = bytes;
    return SMPC_OK;
}

int ensure_mlp(smpc_handle* h, size_t M) {
    const size_t Mp = (M + 127) / 128 * 128;
    if (Mp <= h->capM) return SMPC_OK;
    int rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t H = h->H;
    if ((rc = dev_alloc(h, &h->d_S, Mp * MLP_KPAD))) return rc;
    if ((rc = dev_alloc(h, &h->d_y, Mp))) return rc;
    if ((rc = dev_alloc(h, &h->d_GS, Mp * MLP_NPAD))) return rc;
    if ((rc = dev_alloc(h, &h->d_dA, Mp * H))) return rc;
    if ((rc = dev_alloc(h, &h->d_dB, Mp * H))) return rc;
    for (int l = 0; l + 1 < h->nlayers; l++) {
        if ((rc = dev_alloc(h, &h->d_act[l], Mp * H))) return rc;
        if ((rc = dev_alloc(h, &h->d_dg[l], Mp * H))) return rc;
    }
    h->capM = Mp;
    return SMPC_OK;
}

// list of live network rows (mode 3 of run_mlp) and its device-side length
int ensure_nn_idx(smpc_handle* h, size_t M) {
    if (M <= h->nn_idx_cap) return SMPC_OK;
    int rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((rc = dev_alloc(h, &h->d_nn_idx, M + 1))) return rc;
    h->nn_idx_cap = M;
    h->d_nn_cnt = h->d_nn_idx + M;
    HIPCHK(h, hipMemsetAsync(h->d_nn_cnt, 0, sizeof(int32_t), h->stream));
    return SMPC_OK;
}

// forward (and optionally backward) pass of the network over M rows whose states are found through (mode, N) in x.
// mode 3: the rows are the compacted list h->d_nn_idx of live nodes; their count is only known on the device (h->d_nn_cnt), so
// the grids cover all M candidate rows and the blocks past the count return at once.
// d_p / d_ev: when given and the pass ran as the fused kernel, the chain rule to the node records is done as well and *chained
// is set (the caller then skips k_nn_chain).
template <int NQ> int run_mlp(smpc_handle* h, int M, int mode, int N, const double* d_x, bool backward, const double* d_p = nullptr,
                              double* d_ev = nullptr, bool* chained = nullptr, int compact = 0) {
    int rc;
    if ((rc = ensure_mlp(h, (size_t)M))) return rc;
    const int Mp = (M + 127) / 128 * 128, H = h->H, L = h->nlayers;
    hipStream_t s = h->stream;
    const int32_t* idx = mode == 3 ? h->d_nn_idx : nullptr;
    const int32_t* live = mode == 3 ? h->d_nn_cnt : nullptr;
    if (chained) *chained = false;
    // Few rows (the terminal row: M = B): the whole pass as ONE kernel, activations in LDS / registers (kernels_mlp.hpp).  The
    // choice follows the rows of the whole call (mlp_rows_whole), like the tiled kernel's below.
    {
        const long rows_all = h->mlp_rows_whole > 0 ? (h->mlp_rows_whole + 127) / 128 * 128
                                                    : (h->mlp_rows_hint > 0 && mode == 3 ? (h->mlp_rows_hint + 127) / 128 * 128 : (long)Mp);
        static const bool no_fused = getenv("SMPC_MLP_UNFUSED") != nullptr;     // (A/B knob)
        static const long fused_max = [] { const char* e = getenv("SMPC_MLP_FUSED_MAX"); return e ? atol(e) : 8192L; }();   // (A/B knob)
        if (!no_fused && rows_all < fused_max && H == MLPF_H && L == 4 && (!backward || (d_p && d_ev))) {
            MlpWeights Wt;
            for (int l = 0; l < SMPC_MAX_LAYERS; l++) { Wt.wf[l] = h->d_Wfwd[l]; Wt.wb[l] = h->d_Wbwd[l]; Wt.bias[l] = h->d_bias[l]; }
            const dim3 grd((M + MLPF_ROWS - 1) / MLPF_ROWS), blk(256);
            if (backward)
                hipLaunchKernelGGL((k_mlp_fused<NQ, true>), grd, blk, 0, s, h->d_desc, M, N, mode, h->act, Wt, d_x, d_p, idx, live, h->d_y, d_ev,
                                   compact);
            else
                hipLaunchKernelGGL((k_mlp_fused<NQ, false>), grd, blk, 0, s, h->d_desc, M, N, mode, h->act, Wt, d_x, d_p, idx, live, h->d_y,
                                   (double*)nullptr, 0);
            HIPCHK(h, hipGetLastError());
            if (chained) *chained = backward;
            return SMPC_OK;
        }
        // Many rows (the row on every node): the same pass as ONE-WAVE blocks (k_mlp_wave, round 5) -- a block fits where a single QP
        // wavefront has retired, and only the activation derivatives leave the chip.  SMPC_MLP_LARGE=chain brings the layer-by-layer
        // GEMMs back (A/B runs; they also serve any network that is not 256 wide with three hidden layers).
        static const bool large_chain = [] { const char* e = getenv("SMPC_MLP_LARGE"); return e && !strcmp(e, "chain"); }();
        if (!no_fused && !large_chain && rows_all >= fused_max && H == MLPF_H && L == 4 && (!backward || (d_p && d_ev))) {
            MlpWeights Wt;
            for (int l = 0; l < SMPC_MAX_LAYERS; l++) { Wt.wf[l] = h->d_Wfwd[l]; Wt.wb[l] = h->d_Wbwd[l]; Wt.bias[l] = h->d_bias[l]; }
            const dim3 grd((M + MLPF_ROWS - 1) / MLPF_ROWS), blk(64);
            if (backward)
                hipLaunchKernelGGL((k_mlp_wave<NQ, true>), grd, blk, 0, s, h->d_desc, M, N, mode, h->act, Wt, d_x, d_p, idx, live, h->d_y, d_ev,
                                   compact, h->d_dg[0], h->d_dg[1], h->d_dg[2]);
            else
                hipLaunchKernelGGL((k_mlp_wave<NQ, false>), grd, blk, 0, s, h->d_desc, M, N, mode, h->act, Wt, d_x, d_p, idx, live, h->d_y,
                                   (double*)nullptr, 0, (float*)nullptr, (float*)nullptr, (float*)nullptr);
            HIPCHK(h, hipGetLastError());
            if (chained) *chained = backward;
            return SMPC_OK;
        }
    }
    hipLaunchKernelGGL((k_nn_features<NQ>), dim3((Mp + 63) / 64), dim3(64), 0, s, h->d_desc, M, Mp, N, mode, d_x,
                       h->d_S, idx, live);
    // The layer-by-layer GEMMs.  Default (round 4): k_gemm_f32 as ONE-WAVE blocks -- a wavefront of it is self-contained (32 x 64
    // tile, operands from L2, 90-130 registers, no LDS), so its blocks start on any SIMD with one free slot.  The 128 x 128
    // LDS-tiled kernel (256-thread blocks: 200 registers per lane on all four SIMDs of one CU at once + 37 KB of LDS) is 12 % faster
    // alone on the GPU (90 vs 79 TFLOP/s) but in C4's loop its blocks wait for CUs that QP wavefronts keep refilling: 12-14 ms per
    // solve for 1.9 ms of work (profiles/r04_c4_kernel_summary_by_grid.txt).  SMPC_MLP_GEMM=tiled brings it back (A/B runs); the
    // choice then follows the rows of the whole call as before (the two kernels sum K in different orders).
    static const bool use_tiled_env = [] { const char* e = getenv("SMPC_MLP_GEMM"); return e && !strcmp(e, "tiled"); }();
    const long rows_sel = h->mlp_rows_whole > 0 ? (h->mlp_rows_whole + 127) / 128 * 128 : (long)Mp;
    const bool tiled = use_tiled_env && rows_sel >= 8192 && H % 128 == 0;
    const dim3 blk(tiled ? 256 : 64), blk_t(256);
    const dim3 grd(tiled ? Mp / 128 : Mp / 32, H / 64), grd_t(Mp / 128, H / 128);
    hipLaunchKernelGGL((k_gemm_f32<EPI_BIAS_GELU>), grd, blk, 0, s, Mp, H, MLP_KPAD, h->d_S, h->d_Wfwd[0], h->d_bias[0],
                       (const float*)nullptr, h->d_act[0], h->d_dg[0], live, h->act);
    for (int l = 1; l + 1 < L; l++) {
        if (tiled)
            hipLaunchKernelGGL((k_gemm_f32_tiled<EPI_BIAS_GELU>), grd_t, blk_t, 0, s, Mp, H, H, h->d_act[l - 1], h->d_Wfwd[l],
                               h->d_bias[l], (const float*)nullptr, h->d_act[l], h->d_dg[l], live, h->act);
        else
            hipLaunchKernelGGL((k_gemm_f32<EPI_BIAS_GELU>), grd, blk, 0, s, Mp, H, H, h->d_act[l - 1], h->d_Wfwd[l],
                               h->d_bias[l], (const float*)nullptr, h->d_act[l], h->d_dg[l], live, h->act);
    }
    hipLaunchKernelGGL(k_nn_output, dim3((Mp + 3) / 4), dim3(256), 0, s, Mp, H, h->d_act[L - 2], h->d_dg[L - 2],
                       h->d_Wbwd[L - 1], h->d_bias[L - 1], h->d_y, h->d_dA, live);
    if (backward) {
        float *cur = h->d_dA, *nxt = h->d_dB;
        for (int l = L - 2; l >= 1; l--) {
            if (tiled)
                hipLaunchKernelGGL((k_gemm_f32_tiled<EPI_MUL>), grd_t, blk_t, 0, s, Mp, H, H, cur, h->d_Wbwd[l],
                                   (const float*)nullptr, h->d_dg[l - 1], nxt, (float*)nullptr, live, h->act);
            else
                hipLaunchKernelGGL((k_gemm_f32<EPI_MUL>), grd, blk, 0, s, Mp, H, H, cur, h->d_Wbwd[l], (const float*)nullptr,
                                   h->d_dg[l - 1], nxt, (float*)nullptr, live, h->act);
            float* t = cur; cur = nxt; nxt = t;
        }
        hipLaunchKernelGGL((k_gemm_f32<EPI_PLAIN>), dim3(tiled ? Mp / 128 : Mp / 32, MLP_NPAD / 64), blk, 0, s, Mp, MLP_NPAD, H, cur,
                           h->d_Wbwd[0], (const float*)nullptr, (const float*)nullptr, h->d_GS, (float*)nullptr, live, h->act);
    }
    HIPCHK(h, hipGetLastError());
    return SMPC_OK;
}

// The network's row (value + gradient w.r.t. the state, chain rule and per-node switch included) of every node that carries it:
// into the nodes' linearisation records (compact = 0) or into nn[node][1 + nx] (compact = 1, what the stage builder reads).
template <int NQ>
int launch_nn(smpc_handle* h, int B, const double* d_xg, const double* d_p, double* d_out, int compact) {
    if (h->desc.nn_mode == SMPC_NN_NONE) return SMPC_OK;
    const int N = h->N;
    hipStream_t s = h->stream;
    if (h->nlayers == 0) return fail(h, SMPC_ESTATE, "nn_mode != NONE but smpc_set_mlp was not called");
    // row on every node: only the nodes whose per-node switch is on are evaluated (compacted list, mode 3)
    const int mode = h->desc.nn_mode == SMPC_NN_TERMINAL ? 1 : 3;
    const int M = mode == 1 ? B : B * N;
    int rc;
    if (mode == 3) {
        if ((rc = ensure_nn_idx(h, (size_t)M))) return rc;
        hipLaunchKernelGGL(k_nn_compact, dim3((M + 63) / 64), dim3(64), 0, s, M, N, d_p, h->d_nn_idx, h->d_nn_cnt);
    }
    bool chained = false;
    if ((rc = run_mlp<NQ>(h, M, mode, N, d_xg, true, d_p, d_out, &chained, compact))) return rc;
    if (!chained)
        hipLaunchKernelGGL((k_nn_chain<NQ>), dim3((M + 63) / 64), dim3(64), 0, s, h->d_desc, M, N, mode, d_xg, d_p,
                           h->d_y, h->d_GS, d_out, mode == 3 ? h->d_nn_idx : (const int32_t*)nullptr,
                           mode == 3 ? h->d_nn_cnt : (const int32_t*)nullptr, compact);
    HIPCHK(h, hipGetLastError());
    return SMPC_OK;
}

template <int NQ>
int launch_eval(smpc_handle* h, int B, const double* d_xg, const double* d_ug, const double* d_p, double* d_ev, bool timed = false) {
    // (timed: called from launch_solve, which owns the current slot of the event ring; smpc_eval_nodes on its own records nothing --
    //  it would re-record two events of the last solve's slot and leave that slot's durations meaningless)
    const int N = h->N;
    hipStream_t s = h->stream;
    const long n1 = (long)B * (N + 1);
    hipLaunchKernelGGL((k_node_linearise<NQ>), dim3((unsigned)((n1 + 63) / 64)), dim3(64), 0, s, h->d_desc, B, N, d_xg, d_ug,
                       d_p, d_ev);
    HIPCHK(h, hipGetLastError());
    if (timed) HIPCHK(h, hipEventRecord(h->ev_t[1], s));
    int rc;
    if ((rc = launch_nn<NQ>(h, B, d_xg, d_p, d_ev, 0))) return rc;
    // (this path has no stage builder behind the network pass to hand the row list's counter back at zero: see d_nn_cnt)
    if (h->desc.nn_mode == SMPC_NN_ALL && h->d_nn_cnt) HIPCHK(h, hipMemsetAsync(h->d_nn_cnt, 0, sizeof(int32_t), s));
    if (timed) HIPCHK(h, hipEventRecord(h->ev_t[2], s));
    return SMPC_OK;
}

// experiment knob: extra dynamic LDS per block of k_qp_ipm limits how many wavefronts are resident per CU (the rest queue
// behind the longest-first order and backfill)
static size_t qp_pad_lds() {
#ifdef SMPC_EXPERIMENTS     // (experiment builds only: make FLAGS+=-DSMPC_EXPERIMENTS; the shipped library reads no such knob on its solve path)
    static const size_t v = [] { const char* e = getenv("SMPC_QP_PAD_LDS"); return e ? (size_t)atol(e) : (size_t)0; }();
    return v;
#else
    return 0;
#endif
}

// k_qp_ipm's non-temporal variant: -1 (default) by workspace size, 0 / 1 forced (SMPC_QP_NT, A/B runs).  The threshold sits between
// what was measured to lose (a 244 MB sub-batch workspace, three of them in flight) and to win (489 MB, three in flight).
constexpr size_t qp_nt_threshold = (size_t)384 << 20;
static int qp_nt_mode() {
    static const int v = [] { const char* e = getenv("SMPC_QP_NT"); return e ? atoi(e) : -1; }();
    return v;
}

// 1: the lane-cooperative stage builder (kernel_build.hpp: MLP -> k_stage_build -> k_qp_ipm), the default; 0: the thread-per-node
// kernels of rounds 1-3 (k_node_linearise -> MLP -> k_qp_setup -> k_qp_ipm), kept for A/B runs and as what smpc_eval_nodes uses
static int stage_build_mode() {
    static const int v = [] { const char* e = getenv("SMPC_STAGE_BUILD"); return e ? atoi(e) : 1; }();
    return v;
}

// everything of a solve before the interior point: the stage records of the QP workspace, by either path
template <int NQ>
int launch_stage_records(smpc_handle* h, int B, const double* x0, const double* xg, const double* ug, const double* p, bool timed, int path) {
    int rc;
    const bool per_inst = h->inst_B == B;
    const double* blo = per_inst ? h->d_lo_b : h->d_lo;
    const double* bhi = per_inst ? h->d_hi_b : h->d_hi;
    const long bstride = per_inst ? (long)(h->N + 1) * 2 * h->desc.nq : 0L;
    // experiment knob: the linearisation (+ network pass) and / or the set-up launched once more, to measure what
    // a stream's chain pays for them inside the loop (bit 0: linearisation, bit 1: set-up; old path only)
#ifdef SMPC_EXPERIMENTS
    static const int dup = [] { const char* e = getenv("SMPC_DUP_KERNELS"); return e ? atoi(e) : 0; }();
#else
    constexpr int dup = 0;
#endif
    if (path == 1) {
        if ((rc = launch_nn<NQ>(h, B, xg, p, h->d_nn, h->desc.nn_mode == SMPC_NN_TERMINAL ? 2 : 1))) return rc;
        if (timed) HIPCHK(h, hipEventRecord(h->ev_t[1], h->stream));
        const long nodes = (long)B * (h->N + 1);
        const dim3 grd((unsigned)((nodes + 64 / SB_G - 1) / (64 / SB_G))), blk(64);
        const double* nn = h->desc.nn_mode != SMPC_NN_NONE ? h->d_nn : nullptr;
        int32_t* const zero_cnt = h->desc.nn_mode == SMPC_NN_ALL ? h->d_nn_cnt : nullptr;     // (the builder hands the list's counter back at zero)
        // (experiment knob: extra dynamic LDS per block of the builder -- how much its start depends on LDS room next to QP wavefronts)
#ifdef SMPC_EXPERIMENTS
        static const size_t sb_pad = [] { const char* e = getenv("SMPC_SB_PAD_LDS"); return e ? (size_t)atol(e) : (size_t)0; }();
#else
        constexpr size_t sb_pad = 0;
#endif
#define SMPC_SB_LAUNCH(MR_)                                                                                                       \
        hipLaunchKernelGGL((k_stage_build<NQ, MR_>), grd, blk, sb_pad, h->stream, h->d_desc, B, h->N, x0, xg, ug, p, blo, bhi, h->d_zl, nn,     \
                           h->d_ws, bstride, h->d_active, zero_cnt)
        switch (h->desc.n_rows) {
        case 6: SMPC_SB_LAUNCH(6); break;
        case 4: SMPC_SB_LAUNCH(4); break;
        default: SMPC_SB_LAUNCH(-1); break;
        }
#undef SMPC_SB_LAUNCH
        HIPCHK(h, hipGetLastError());
        if (timed) { HIPCHK(h, hipEventRecord(h->ev_t[2], h->stream)); HIPCHK(h, hipEventRecord(h->ev_t[4], h->stream)); }
        return SMPC_OK;
    }
    if ((rc = launch_eval<NQ>(h, B, xg, ug, p, h->d_ev, timed))) return rc;
    if (dup & 1) { if ((rc = launch_eval<NQ>(h, B, xg, ug, p, h->d_ev, false))) return rc; }
    const int tiles = (int)ev_tiles((size_t)B * (h->N + 1));
    // the row counts of the shipped geometries are compile-time constants of the kernels (6: the reference's six capsule
    // pairs, config.yaml:205-216; 4: config_fr7.yaml); any other count takes the runtime-row-count instantiation
#define SMPC_SETUP_LAUNCH(MR_)                                                                                                     \
    do {                                                                                                                           \
        hipLaunchKernelGGL((k_qp_setup<NQ, MR_>), dim3(tiles), dim3(32 * EV_TILE), 0, h->stream, h->d_desc, B, h->N, x0, xg, ug, p, blo,     \
                           bhi, h->d_zl, h->d_ev, h->d_ws, bstride, h->d_active);                                                  \
        if (dup & 2)                                                                                                               \
            hipLaunchKernelGGL((k_qp_setup<NQ, MR_>), dim3(tiles), dim3(32 * EV_TILE), 0, h->stream, h->d_desc, B, h->N, x0, xg, ug, p, blo, \
                               bhi, h->d_zl, h->d_ev, h->d_ws, bstride, h->d_active);                                              \
    } while (0)
    switch (h->desc.n_rows) {
    case 6: SMPC_SETUP_LAUNCH(6); break;
    case 4: SMPC_SETUP_LAUNCH(4); break;
    default: SMPC_SETUP_LAUNCH(-1); break;
    }
#undef SMPC_SETUP_LAUNCH
    HIPCHK(h, hipGetLastError());
    if (timed) HIPCHK(h, hipEventRecord(h->ev_t[4], h->stream));
    return SMPC_OK;
}

// k_qp_ipm_wg (kernel_qp_wg.hpp), the latency form of the interior-point solve -- one workgroup per instance: -1 (default) chosen by
// batch size, 0 never, 1 whenever its LDS fits (SMPC_QP_WG; smpc_set_qp_mode per handle).  Built with 8 half-wavefronts per
// workgroup (four wavefronts, one per SIMD: one workgroup per CU -- up to qp_wg_full_batch instances run in one round) and with 4
// (two wavefronts: two workgroups per CU, the stage-parallel phases take twice the rounds -- 512 instances in one round, up to
// qp_wg_max_batch = 1024 in one launch whose later workgroups start as the first ones retire).  Above that k_qp_ipm's two instances per
// wavefront use the chip better (per step, latency form against k_qp_ipm: 0.99 ms against 2.16 at 512 instances, 1.37 against 2.27 at
// 1024 as two sub-batches of 512, 2.06 against 2.46 at 1536 as three; from 2048 on k_qp_ipm wins -- DESIGN.md section 4c).  A handle
// only sees its own launch: a caller that spreads ONE batch over several handles should pick the form from the total (bench.py does:
// the latency form up to 1536 instances per GPU in sub-batches of at most 512).
#ifndef QP_WG_FULL_BATCH
#define QP_WG_FULL_BATCH 256
#endif
#ifndef QP_WG_MAX_BATCH
#define QP_WG_MAX_BATCH 1024
#endif
static int qp_wg_mode() {
    static const int v = [] { const char* e = getenv("SMPC_QP_WG"); return e ? atoi(e) : -1; }();
    return v;
}
static int qp_wg_max_batch() {
    static const int v = [] { const char* e = getenv("SMPC_QP_WG_MAX_BATCH"); return e ? atoi(e) : QP_WG_MAX_BATCH; }();
    return v;
}
static int qp_wg_full_batch() {
    static const int v = [] { const char* e = getenv("SMPC_QP_WG_FULL_BATCH"); return e ? atoi(e) : QP_WG_FULL_BATCH; }();
    return v;
}
constexpr size_t QP_WG_LDS_LIMIT = 160 * 1024 - 512;     // one CU's LDS less the kernel's static tables

// half-wavefronts per workgroup for a launch of B instances (0: the latency form is not to be used)
template <int NQ> int qp_wg_choice(const smpc_handle* h, int B) {
    const int mode = h->qp_mode >= -1 ? h->qp_mode : qp_wg_mode();
    if (mode == 0) return 0;
    if (mode < 0 && B > qp_wg_max_batch()) return 0;
    const auto fits = [&](int nhw) { return (size_t)WgLds<NQ>(h->N, h->desc.n_rows, nhw).total * sizeof(double) <= QP_WG_LDS_LIMIT; };
    if (B <= qp_wg_full_batch() && fits(8)) return 8;
    return fits(4) ? 4 : 0;
}

template <int NQ, int NHW>
int launch_qp_wg(smpc_handle* h, int B, const double* x0, const double* xg, const double* ug, double* xo, double* uo, int32_t* st,
                 int32_t* it) {
    const size_t need = (size_t)B * (h->N + 1) * HRecLayout<NQ>::SIZE;
    if (need > h->hrec_doubles) {
        int rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if ((rc = dev_alloc(h, &h->d_hrec, need))) return rc;
        h->hrec_doubles = need;
    }
    const size_t lds = (size_t)WgLds<NQ>(h->N, h->desc.n_rows, NHW).total * sizeof(double);
#define SMPC_WG_LAUNCH(MR_)                                                                                                        \
    do {                                                                                                                           \
        if (!h->wg_lds_set[NHW == 8 ? 0 : 1]) {                                                                                   \
            /* (once per handle = per device, nq and row count: the kernel may use a whole CU's LDS -- always the same value, so a    \
             *  handle with a short horizon never lowers the limit under one with a long horizon) */                               \
            HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_qp_ipm_wg<NQ, MR_, NHW>),                               \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)QP_WG_LDS_LIMIT));                      \
            h->wg_lds_set[NHW == 8 ? 0 : 1] = QP_WG_LDS_LIMIT;                                                                     \
        }                                                                                                                          \
        hipLaunchKernelGGL((k_qp_ipm_wg<NQ, MR_, NHW>), dim3(B), dim3(32 * NHW), lds, h->stream, h->d_desc, B, h->N, x0, xg,         \
                           ug, h->d_ws, h->d_hrec, xo, uo, st, it, h->d_last_it, h->d_active, h->d_ord_hist);                      \
    } while (0)
    switch (h->desc.n_rows) {
    case 6: SMPC_WG_LAUNCH(6); break;
    case 4: SMPC_WG_LAUNCH(4); break;
    default: SMPC_WG_LAUNCH(-1); break;
    }
#undef SMPC_WG_LAUNCH
    return SMPC_OK;
}

template <int NQ>
int launch_solve(smpc_handle* h, int B, const double* x0, const double* xg, const double* ug, const double* p,
                 double* xo, double* uo, int32_t* st, int32_t* it) {
    int rc;
    h->timing_now = false;
    if (h->timing) {
        // a solve that is being captured into a hipGraph records nothing: every replay would re-record the one slot it captured
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        h->timing_now = cs == hipStreamCaptureStatusNone;
    }
    const bool timed = h->timing_now;
    const int path = stage_build_mode();
    if (timed) {
        h->ev_cur = (h->ev_cur + 1) % smpc_handle::EV_RING;
        h->ev_t = h->ev_sets[h->ev_cur];
        h->ev_complete[h->ev_cur] = false;      // (an error return below leaves the slot invalid, not stale)
        h->ev_new_order[h->ev_cur] = path == 1;
        h->timed_count++;
        HIPCHK(h, hipEventRecord(h->ev_t[0], h->stream));
    }
    if ((rc = launch_stage_records<NQ>(h, B, x0, xg, ug, p, timed, path))) return rc;
    const int wg_nhw = qp_wg_choice<NQ>(h, B);
    const bool wg = wg_nhw > 0;
    const int32_t* order = nullptr;
    if (h->order_B == B && B > 1 && !wg) {
        hipLaunchKernelGGL(k_order_by_iters, dim3((B + ORD_PER_BLOCK - 1) / ORD_PER_BLOCK), dim3(64), 0, h->stream, B, h->d_last_it,
                           h->d_order, h->d_ord_hist, h->d_ord_hist + 256, h->d_ord_hist + 512);
        order = h->d_order;
    } else {
        // (no order this time: the histogram k_qp_ipm adds to must hold this solve alone when the next one sorts by it)
        HIPCHK(h, hipMemsetAsync(h->d_ord_hist, 0, 520 * sizeof(int32_t), h->stream));
    }
    unsigned long long* wstat = nullptr;
    if (timed && h->timing == 1) {      // (timing mode 2: events only, no in-kernel load-balance probe)
        if (!h->d_wstat) HIPCHK(h, hipMalloc((void**)&h->d_wstat, 4 * sizeof(unsigned long long)));
        const unsigned long long init[4] = {0ull, ~0ull, 0ull, 0ull};
        HIPCHK(h, hipMemcpyAsync(h->d_wstat, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
        wstat = h->d_wstat;
    }
    // non-temporal workspace accesses once this launch's workspace is well beyond the Infinity Cache (kernel_qp.hpp, k_qp_ipm)
    if (wg) {
        if ((rc = wg_nhw == 8 ? launch_qp_wg<NQ, 8>(h, B, x0, xg, ug, xo, uo, st, it) : launch_qp_wg<NQ, 4>(h, B, x0, xg, ug, xo, uo, st, it))) return rc;
    } else {
    const bool nt = qp_nt_mode() < 0 ? ws_doubles_per_instance(h->desc, h->N) * sizeof(double) * (size_t)B >= qp_nt_threshold : qp_nt_mode() > 0;
#define SMPC_QP_LAUNCH(MR_, NT_)                                                                                                   \
    hipLaunchKernelGGL((k_qp_ipm<NQ, MR_, NT_>), dim3((B + 1) / 2), dim3(64), qp_pad_lds(), h->stream, h->d_desc, B, h->N, x0, xg,  \
                       ug, h->d_ws, xo, uo, st, it, order, h->d_last_it, wstat, h->d_active, h->d_ord_hist)
    switch (h->desc.n_rows) {
    case 6: if (nt) SMPC_QP_LAUNCH(6, true); else SMPC_QP_LAUNCH(6, false); break;
    case 4: if (nt) SMPC_QP_LAUNCH(4, true); else SMPC_QP_LAUNCH(4, false); break;
    default: SMPC_QP_LAUNCH(-1, false); break;      // (the runtime-row-count instantiation is not built twice)
    }
#undef SMPC_QP_LAUNCH
    }
    h->order_B = B;
    HIPCHK(h, hipGetLastError());
    if (timed) {
        HIPCHK(h, hipEventRecord(h->ev_t[3], h->stream));
        h->ev_complete[h->ev_cur] = true;
        h->timed = 1;
    }
    return SMPC_OK;
}

#define DISPATCH_NQ(h, call)                                                      \
    switch ((h)->desc.nq) {                                                       \
    case 5: { constexpr int NQ_ = 5; rc = call; } break;                          \
    case 6: { constexpr int NQ_ = 6; rc = call; } break;                          \
    case 7: { constexpr int NQ_ = 7; rc = call; } break;                          \
    default: rc = fail(h, SMPC_EINVAL, "nq=%d not built (5, 6, 7)", (h)->desc.nq); \
    }


// The check bounds of the state tests live in a small device block of their own and are uploaded only when they change: a
// copy from pageable host memory waits for the stream to drain, which would turn every per-step call of a device-resident
// loop into a host synchronisation.
int upload_check_bounds(smpc_handle* h, const double* x_min, const double* x_max, const double* row_lb_chk,
                        const double* row_ub_chk) {
    const int nx = 2 * h->desc.nq, nr = h->desc.n_rows;
    std::vector<double> cur((size_t)2 * nx + 2 * SMPC_MAX_ROWS, 0.0);
    memcpy(cur.data(), x_min, sizeof(double) * nx);
    memcpy(cur.data() + nx, x_max, sizeof(double) * nx);
    if (nr > 0) {
        memcpy(cur.data() + 2 * nx, row_lb_chk, sizeof(double) * nr);
        memcpy(cur.data() + 2 * nx + SMPC_MAX_ROWS, row_ub_chk, sizeof(double) * nr);
    }
    if (!h->d_chk) HIPCHK(h, hipMalloc((void**)&h->d_chk, cur.size() * sizeof(double)));
    if (cur != h->chk_cache) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipMemcpy(h->d_chk, cur.data(), cur.size() * sizeof(double), hipMemcpyHostToDevice));
        h->chk_cache.swap(cur);
    }
    return SMPC_OK;
}

// state test (+ safe-set test if d_nn) of B trajectories of n_nodes nodes on the device, against the uploaded check bounds;
// collision rows on the leading coll_nodes nodes only
int check_nodes_dev(smpc_handle* h, int B, int n_nodes, const double* d_x, double tol_x, int coll_nodes, double alpha,
                    double tol_safe, int32_t* d_ok, int32_t* d_nn, bool nn_listed = false, bool ok_prefilled = false) {
    const int nq = h->desc.nq, nx = 2 * nq;
    hipStream_t s = h->stream;
    const size_t M = (size_t)B * n_nodes;
    double* d_min = h->d_chk;
    double* d_max = d_min + nx;
    double* d_rlb = d_max + nx;
    double* d_rub = d_rlb + SMPC_MAX_ROWS;
    // verdicts start at "ok", stream-ordered (smpc_policy_step: an earlier kernel of the step has done it)
    if (!ok_prefilled) HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)d_ok, 1, (size_t)B, s));
    // (one wavefront per block throughout the small kernels: a multi-wave block needs room on several SIMDs of ONE CU at the
    //  same moment, and next to resident QP wavefronts -- 256 registers each, two fill a SIMD -- it waited for that up to a
    //  millisecond: k_policy_post, six blocks of four waves, averaged 131 us in the three-stream loop; rocprofv3, round 3)
    const dim3 grd((unsigned)((M + 63) / 64)), blk(64);
    switch (nq) {
    case 5: hipLaunchKernelGGL((k_check_nodes<5>), grd, blk, 0, s, h->d_desc, B, n_nodes, d_x, d_min, d_max, tol_x, d_rlb, d_rub, d_ok, coll_nodes); break;
    case 6: hipLaunchKernelGGL((k_check_nodes<6>), grd, blk, 0, s, h->d_desc, B, n_nodes, d_x, d_min, d_max, tol_x, d_rlb, d_rub, d_ok, coll_nodes); break;
    default: hipLaunchKernelGGL((k_check_nodes<7>), grd, blk, 0, s, h->d_desc, B, n_nodes, d_x, d_min, d_max, tol_x, d_rlb, d_rub, d_ok, coll_nodes); break;
    }
    HIPCHK(h, hipGetLastError());
    if (d_nn) {
        int rc;
        // nn_listed: only the nodes in h->d_nn_idx (length on the device, h->d_nn_cnt) are evaluated; their verdicts land at the
        // nodes' own positions of d_nn, the rest of d_nn is left as it is
        DISPATCH_NQ(h, (run_mlp<NQ_>(h, (int)M, nn_listed ? 3 : 0, 0, d_x, false)));
        if (rc) return rc;
        const dim3 g2((unsigned)((M + 63) / 64)), b2(64);
        const int32_t* li = nn_listed ? h->d_nn_idx : nullptr;
        const int32_t* lc = nn_listed ? h->d_nn_cnt : nullptr;
        switch (nq) {
        case 5: hipLaunchKernelGGL((k_check_nn<5>), g2, b2, 0, s, h->d_desc, (int)M, d_x, alpha, tol_safe, h->d_y, d_nn, li, lc); break;
        case 6: hipLaunchKernelGGL((k_check_nn<6>), g2, b2, 0, s, h->d_desc, (int)M, d_x, alpha, tol_safe, h->d_y, d_nn, li, lc); break;
        default: hipLaunchKernelGGL((k_check_nn<7>), g2, b2, 0, s, h->d_desc, (int)M, d_x, alpha, tol_safe, h->d_y, d_nn, li, lc); break;
        }
        HIPCHK(h, hipGetLastError());
    }
    return SMPC_OK;
}

// scratch of the policy entry points (grown on demand: the first, eager steps of a loop; never inside a graph capture)
int ensure_polw(smpc_handle* h, size_t bytes) {
    if (bytes <= h->polw_bytes) return SMPC_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_polw) { (void)hipFree(h->d_polw); h->d_polw = nullptr; h->polw_bytes = 0; }
    if (hipMalloc((void**)&h->d_polw, bytes) != hipSuccess) return fail(h, SMPC_ENOMEM, "hipMalloc(%zu) failed", bytes);
    h->polw_bytes = bytes;
    return SMPC_OK;
}

// Worker handles of smpc_rollout_batch: same problem, own stream and workspaces, the parent's network weights (borrowed).
int rollout_workers(smpc_handle* h, int n) {
    while ((int)h->kids.size() < n) {
        smpc_handle* k = nullptr;
        int rc = smpc_create(&h->desc, h->device, &k);
        if (rc) return fail(h, rc, "rollout worker: %s", smpc_last_error(nullptr));
        k->borrowed_mlp = true;
        h->kids.push_back(k);
    }
    for (int i = 0; i < n; i++) {
        smpc_handle* k = h->kids[i];
        int rc;
        if (k->N != h->N && (rc = smpc_set_horizon(k, h->N))) return fail(h, rc, "rollout worker: %s", k->err);
        k->nlayers = h->nlayers;
        k->act = h->act;
        k->qp_mode = h->qp_mode;
        k->H = h->H;
        for (int l = 0; l <= SMPC_MAX_LAYERS; l++) k->dims[l] = h->dims[l];
        for (int l = 0; l < SMPC_MAX_LAYERS; l++) { k->d_Wfwd[l] = h->d_Wfwd[l]; k->d_Wbwd[l] = h->d_Wbwd[l]; k->d_bias[l] = h->d_bias[l]; }
        // stage bounds and slack weights follow the parent (small; stream-ordered on the worker's stream)
        const size_t nb = (size_t)(h->N + 1) * 2 * h->desc.nq;
        HIPCHK(h, hipMemcpyAsync(k->d_lo, h->d_lo, nb * sizeof(double), hipMemcpyDeviceToDevice, k->stream));
        HIPCHK(h, hipMemcpyAsync(k->d_hi, h->d_hi, nb * sizeof(double), hipMemcpyDeviceToDevice, k->stream));
        if (h->d_zl) {
            if (!k->d_zl && (rc = dev_alloc(h, &k->d_zl, (size_t)h->N + 1))) return rc;
            HIPCHK(h, hipMemcpyAsync(k->d_zl, h->d_zl, sizeof(double) * (h->N + 1), hipMemcpyDeviceToDevice, k->stream));
        } else if (k->d_zl) {
            HIPCHK(h, hipStreamSynchronize(k->stream));
            (void)hipFree(k->d_zl);
            k->d_zl = nullptr;
        }
    }
    return SMPC_OK;
}

}  // namespace

extern "C" {

int smpc_abi_version(void) { return SMPC_ABI_VERSION; }

const char* smpc_last_error(const smpc_handle* h) { return h ? h->err : g_create_err; }

int smpc_create(const smpc_problem_desc* desc, int device, smpc_handle** out) {
    if (!desc || !out) return fail(nullptr, SMPC_EINVAL, "null argument");
    *out = nullptr;
    if (desc->abi_version != SMPC_ABI_VERSION)
        return fail(nullptr, SMPC_EINVAL, "descriptor ABI %d, library ABI %d", desc->abi_version, SMPC_ABI_VERSION);
    if (desc->nq < 5 || desc->nq > SMPC_MAX_NQ) return fail(nullptr, SMPC_EINVAL, "nq=%d unsupported", desc->nq);
    if (desc->N < 1 || desc->N > SMPC_MAX_N) return fail(nullptr, SMPC_EINVAL, "N=%d outside 1..%d", desc->N, SMPC_MAX_N);
    if (desc->n_rows < 0 || desc->n_rows > SMPC_MAX_ROWS || desc->n_points < 1 || desc->n_points > SMPC_MAX_POINTS)
        return fail(nullptr, SMPC_EINVAL, "n_rows / n_points out of range");
    for (int i = 0; i < desc->n_points; i++)
        if (desc->points[i].link >= desc->nq) return fail(nullptr, SMPC_EINVAL, "point %d rides on link %d >= nq", i, desc->points[i].link);
    for (int r = 0; r < desc->n_rows; r++) {
        const smpc_row& row = desc->rows[r];
        if (row.kind < 0 || row.kind > SMPC_ROW_COORD || row.pa < 0 || row.pa >= desc->n_points)
            return fail(nullptr, SMPC_EINVAL, "row %d malformed", r);
    }
    // k_qp_ipm gives every two-sided constraint row (x box, torque, collision, safe-set) its own lane of a half-wavefront
    if (3 * desc->nq + desc->n_rows + 1 > 32)
        return fail(nullptr, SMPC_EINVAL, "3*nq + n_rows + 1 = %d constraint rows per stage exceed the 32 lanes of a half-wavefront",
                    3 * desc->nq + desc->n_rows + 1);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, SMPC_EHIP, "no HIP device visible: the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, SMPC_EINVAL, "device %d of %d", device, ndev);
    smpc_handle* h = new (std::nothrow) smpc_handle();
    if (!h) return fail(nullptr, SMPC_ENOMEM, "out of host memory");
    h->desc = *desc;
    h->device = device;
    h->N = desc->N;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_desc, sizeof(smpc_problem_desc));
    if (e == hipSuccess) e = hipMemcpy(h->d_desc, desc, sizeof(smpc_problem_desc), hipMemcpyHostToDevice);
    for (int i = 0; i < 5 * smpc_handle::EV_RING && e == hipSuccess; i++) e = hipEventCreate(&h->ev_sets[i / 5][i % 5]);
    if (e != hipSuccess) {
        fail(nullptr, SMPC_EHIP, "device setup failed: %s", hipGetErrorString(e));
        smpc_destroy(h);
        return SMPC_EHIP;
    }
    int rc = upload_bounds(h, nullptr, nullptr);
    if (rc) {
        snprintf(g_create_err, sizeof(g_create_err), "%s", h->err);
        smpc_destroy(h);
        return rc;
    }
    *out = h;
    return SMPC_OK;
}

void smpc_destroy(smpc_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (smpc_handle* k : h->kids) smpc_destroy(k);
    h->kids.clear();
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    void* ptrs[] = {h->d_desc, h->d_zl, h->d_lo, h->d_hi, h->d_lo_b, h->d_hi_b, h->d_ev, h->d_nn, h->d_ws, h->d_hrec, h->d_order, h->d_last_it, h->d_ord_hist, h->d_x0, h->d_xg, h->d_ug, h->d_p, h->d_xo, h->d_uo,
                    h->d_st, h->d_it, h->d_S, h->d_y, h->d_GS, h->d_dA, h->d_dB, h->d_tmp};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int l = 0; l < SMPC_MAX_LAYERS; l++) {
        if (h->d_Wfwd[l] && !h->borrowed_mlp) (void)hipFree(h->d_Wfwd[l]);
        if (h->d_Wbwd[l] && !h->borrowed_mlp) (void)hipFree(h->d_Wbwd[l]);
        if (h->d_bias[l] && !h->borrowed_mlp) (void)hipFree(h->d_bias[l]);
        if (h->d_act[l]) (void)hipFree(h->d_act[l]);
        if (h->d_dg[l]) (void)hipFree(h->d_dg[l]);
    }
    for (auto& set : h->ev_sets) for (auto& e : set) if (e) (void)hipEventDestroy(e);
    if (h->d_pol) (void)hipFree(h->d_pol);
    if (h->d_polw) (void)hipFree(h->d_polw);
    if (h->d_nn_idx) (void)hipFree(h->d_nn_idx);
    if (h->d_roll) (void)hipFree(h->d_roll);
    if (h->d_chk) (void)hipFree(h->d_chk);
    if (h->d_wstat) (void)hipFree(h->d_wstat);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int smpc_set_mlp(smpc_handle* h, int nlayers, const int32_t* dims, const float* const* W, const float* const* b,
                 int on_device) {
    if (!h || !dims || !W || !b) return fail(h, SMPC_EINVAL, "null argument");
    if (nlayers < 2 || nlayers > SMPC_MAX_LAYERS) return fail(h, SMPC_EINVAL, "nlayers=%d outside 2..%d", nlayers, SMPC_MAX_LAYERS);
    const int in = dims[0], H = dims[1];
    if (in != 2 * h->desc.nn_dof || in > MLP_KPAD) return fail(h, SMPC_EINVAL, "input width %d != 2*n_dof_safe_set", in);
    if (H % 64 != 0 || dims[nlayers] != 1) return fail(h, SMPC_EINVAL, "hidden width must be a multiple of 64 and output 1");
    for (int l = 1; l < nlayers; l++)
        if (dims[l] != H) return fail(h, SMPC_EINVAL, "hidden layers must share one width");
    (void)hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (smpc_handle* k : h->kids) smpc_destroy(k);      // (workers borrow the weight buffers replaced below)
    h->kids.clear();
    h->capM = 0;
    for (int l = 0; l < nlayers; l++) {
        const int ni = dims[l], no = dims[l + 1];
        std::vector<float> w((size_t)ni * no), bb(no);
        HIPCHK(h, hipMemcpy(w.data(), W[l], w.size() * sizeof(float), on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
        HIPCHK(h, hipMemcpy(bb.data(), b[l], bb.size() * sizeof(float), on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
        const int kf = l == 0 ? MLP_KPAD : ni;            // rows of the forward operand W^T
        const int nb = l == 0 ? MLP_NPAD : ni;            // columns of the backward operand W
        std::vector<float> wf((size_t)kf * no, 0.0f), wb((size_t)no * nb, 0.0f);
        for (int o = 0; o < no; o++)
            for (int i = 0; i < ni; i++) {
                wf[(size_t)i * no + o] = w[(size_t)o * ni + i];
                wb[(size_t)o * nb + i] = w[(size_t)o * ni + i];
            }
        int rc;
        if ((rc = dev_alloc(h, &h->d_Wfwd[l], wf.size()))) return rc;
        if ((rc = dev_alloc(h, &h->d_Wbwd[l], wb.size()))) return rc;
        if ((rc = dev_alloc(h, &h->d_bias[l], bb.size()))) return rc;
        HIPCHK(h, hipMemcpy(h->d_Wfwd[l], wf.data(), wf.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->d_Wbwd[l], wb.data(), wb.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->d_bias[l], bb.data(), bb.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    h->nlayers = nlayers;
    for (int l = 0; l <= nlayers; l++) h->dims[l] = dims[l];
    h->H = H;
    return SMPC_OK;
}

int smpc_set_mlp_activation(smpc_handle* h, int act) {
    if (!h) return SMPC_EINVAL;
    if (act < SMPC_ACT_GELU_TANH || act > SMPC_ACT_SILU) return fail(h, SMPC_EINVAL, "unknown activation %d", act);
    h->act = act;
    for (smpc_handle* k : h->kids) k->act = act;
    return SMPC_OK;
}

int smpc_set_qp_mode(smpc_handle* h, int mode) {
    if (!h) return SMPC_EINVAL;
    if (mode < SMPC_QP_AUTO || mode > SMPC_QP_LATENCY) return fail(h, SMPC_EINVAL, "qp mode %d (SMPC_QP_AUTO, _THROUGHPUT, _LATENCY)", mode);
    h->qp_mode = mode;
    for (smpc_handle* k : h->kids) k->qp_mode = mode;
    return SMPC_OK;
}

int smpc_set_horizon(smpc_handle* h, int N) {
    if (!h) return SMPC_EINVAL;
    if (N < 1 || N > SMPC_MAX_N) return fail(h, SMPC_EINVAL, "N=%d outside 1..%d", N, SMPC_MAX_N);
    (void)hipSetDevice(h->device);
    h->N = N;
    h->inst_B = 0;
    if (h->d_zl) { (void)hipFree(h->d_zl); h->d_zl = nullptr; }   // per-node weights belong to the old horizon
    h->ws_bytes = 0;  // workspace layout, linearisation records and IO staging are sized by N
    h->capIO = 0;
    return upload_bounds(h, nullptr, nullptr);
}

int smpc_set_stage_bounds(smpc_handle* h, const double* lo, const double* hi) {
    if (!h) return SMPC_EINVAL;
    if ((lo == nullptr) != (hi == nullptr)) return fail(h, SMPC_EINVAL, "lo and hi must both be given or both be NULL");
    (void)hipSetDevice(h->device);
    return upload_bounds(h, lo, hi);
}

int smpc_set_slack_weights(smpc_handle* h, const double* zl) {
    if (!h) return SMPC_EINVAL;
    (void)hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!zl) {
        if (h->d_zl) { (void)hipFree(h->d_zl); h->d_zl = nullptr; }
        return SMPC_OK;
    }
    for (int k = 1; k <= h->N; k++)
        if (!(zl[k] >= 0.0)) return fail(h, SMPC_EINVAL, "slack weight of node %d is negative or NaN", k);
    int rc;
    if ((rc = dev_alloc(h, &h->d_zl, (size_t)h->N + 1))) return rc;
    HIPCHK(h, hipMemcpy(h->d_zl, zl, sizeof(double) * (h->N + 1), hipMemcpyHostToDevice));
    return SMPC_OK;
}

int smpc_set_instance_bounds(smpc_handle* h, int B, const double* lo, const double* hi, int on_device) {
    if (!h) return SMPC_EINVAL;
    if ((lo == nullptr) != (hi == nullptr)) return fail(h, SMPC_EINVAL, "lo and hi must both be given or both be NULL");
    (void)hipSetDevice(h->device);
    if (!lo) { h->inst_B = 0; return SMPC_OK; }
    if (B <= 0) return fail(h, SMPC_EINVAL, "bad batch size");
    const size_t n = (size_t)B * (h->N + 1) * 2 * h->desc.nq;
    int rc;
    if (n > h->inst_cap) {   // (re)allocate only when the tube grows: a per-step caller pays two stream-ordered copies
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if ((rc = dev_alloc(h, &h->d_lo_b, n))) return rc;
        if ((rc = dev_alloc(h, &h->d_hi_b, n))) return rc;
        h->inst_cap = n;
    }
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    HIPCHK(h, hipMemcpyAsync(h->d_lo_b, lo, n * sizeof(double), kind, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_hi_b, hi, n * sizeof(double), kind, h->stream));
    if (!on_device) HIPCHK(h, hipStreamSynchronize(h->stream));   // host buffers may be reused by the caller on return
    h->inst_B = B;
    return SMPC_OK;
}

int smpc_solve_batch(smpc_handle* h, int B, const double* x0, const double* xg, const double* ug, const double* p,
                     double* x_out, double* u_out, int32_t* status, int32_t* qp_iter, int on_device) {
    if (!h) return SMPC_EINVAL;
    if (B <= 0 || !x0 || !xg || !ug || !p || !x_out || !u_out || !status) return fail(h, SMPC_EINVAL, "bad argument");
    (void)hipSetDevice(h->device);
    int rc;
    if ((rc = ensure_batch(h, B))) return rc;
    const int N = h->N, nx = 2 * h->desc.nq, nu = h->desc.nq;
    if (on_device) {
        DISPATCH_NQ(h, (launch_solve<NQ_>(h, B, x0, xg, ug, p, x_out, u_out, status, qp_iter)));
        return rc;
    }
    if ((rc = ensure_io(h, B))) return rc;
    hipStream_t s = h->stream;
    HIPCHK(h, hipMemcpyAsync(h->d_x0, x0, sizeof(double) * B * nx, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->d_xg, xg, sizeof(double) * B * (N + 1) * nx, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->d_ug, ug, sizeof(double) * B * N * nu, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->d_p, p, sizeof(double) * B * (N + 1) * SMPC_NP, hipMemcpyHostToDevice, s));
    DISPATCH_NQ(h, (launch_solve<NQ_>(h, B, h->d_x0, h->d_xg, h->d_ug, h->d_p, h->d_xo, h->d_uo, h->d_st, h->d_it)));
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(x_out, h->d_xo, sizeof(double) * B * (N + 1) * nx, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(u_out, h->d_uo, sizeof(double) * B * N * nu, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(status, h->d_st, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
    if (qp_iter) HIPCHK(h, hipMemcpyAsync(qp_iter, h->d_it, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return SMPC_OK;
}

int smpc_eval_nodes(smpc_handle* h, int B, const double* xg, const double* ug, const double* p, smpc_node_eval* out,
                    int on_device) {
    if (!h) return SMPC_EINVAL;
    if (B <= 0 || !xg || !ug || !p || !out) return fail(h, SMPC_EINVAL, "bad argument");
    (void)hipSetDevice(h->device);
    int rc;
    if ((rc = ensure_batch(h, B))) return rc;
    const int N = h->N, nx = 2 * h->desc.nq, nu = h->desc.nq;
    hipStream_t s = h->stream;
    const long nodes = (long)B * (N + 1);
    const double *dxg = xg, *dug = ug, *dp = p;
    double* dout = reinterpret_cast<double*>(out);
    if (!on_device) {
        if ((rc = ensure_io(h, B))) return rc;
        if ((rc = ensure_tmp(h, sizeof(smpc_node_eval) * (size_t)nodes))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->d_xg, xg, sizeof(double) * B * (N + 1) * nx, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->d_ug, ug, sizeof(double) * B * N * nu, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->d_p, p, sizeof(double) * B * (N + 1) * SMPC_NP, hipMemcpyHostToDevice, s));
        dxg = h->d_xg; dug = h->d_ug; dp = h->d_p;
        dout = (double*)h->d_tmp;
    }
    // (entries no kernel writes -- the unused tails of the MAX_NQ / MAX_ROWS arrays -- read as zero)
    HIPCHK(h, hipMemsetAsync(h->d_ev, 0, sizeof(double) * ev_tiles((size_t)nodes) * EV_TILE * EV_D, s));
    DISPATCH_NQ(h, (launch_eval<NQ_>(h, B, dxg, dug, dp, h->d_ev)));
    if (rc) return rc;
    hipLaunchKernelGGL(k_ev_untile, dim3((unsigned)((nodes * EV_D + 255) / 256)), dim3(256), 0, s, nodes, h->d_ev, dout);
    HIPCHK(h, hipGetLastError());
    if (!on_device) {
        HIPCHK(h, hipMemcpyAsync(out, dout, sizeof(smpc_node_eval) * (size_t)nodes, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
    }
    return SMPC_OK;
}

int smpc_guess_correction(smpc_handle* h, int B, double* xg, const double* ug, int on_device) {
    if (!h) return SMPC_EINVAL;
    if (B <= 0 || !xg || !ug) return fail(h, SMPC_EINVAL, "bad argument");
    (void)hipSetDevice(h->device);
    const int N = h->N, nq = h->desc.nq, nx = 2 * nq;
    hipStream_t s = h->stream;
    double *dx = xg;
    const double* du = ug;
    if (!on_device) {
        int rc;
        if ((rc = ensure_tmp(h, sizeof(double) * B * ((size_t)(N + 1) * nx + (size_t)N * nq)))) return rc;
        dx = (double*)h->d_tmp;
        double* duw = dx + (size_t)B * (N + 1) * nx;
        HIPCHK(h, hipMemcpyAsync(dx, xg, sizeof(double) * B * (N + 1) * nx, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(duw, ug, sizeof(double) * B * N * nq, hipMemcpyHostToDevice, s));
        du = duw;
    }
    hipLaunchKernelGGL(k_guess_correction, dim3((B * nq + 63) / 64), dim3(64), 0, s, B, N, nq, h->desc.dt, dx, du, (const uint8_t*)nullptr,
                       (int32_t*)nullptr, (int32_t*)nullptr);
    HIPCHK(h, hipGetLastError());
    if (!on_device) {
        HIPCHK(h, hipMemcpyAsync(xg, dx, sizeof(double) * B * (N + 1) * nx, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
    }
    return SMPC_OK;
}

int smpc_provide_control(smpc_handle* h, int B, const int32_t* accept, const double* x_temp, const double* u_temp,
                         double* xg, double* ug, double* u_apply, int on_device) {
    if (!h) return SMPC_EINVAL;
    if (B <= 0 || !accept || !x_temp || !u_temp || !xg || !ug || !u_apply) return fail(h, SMPC_EINVAL, "bad argument");
    (void)hipSetDevice(h->device);
    const int N = h->N, nq = h->desc.nq, nx = 2 * nq;
    hipStream_t s = h->stream;
    const size_t nX = (size_t)B * (N + 1) * nx, nU = (size_t)B * N * nq;
    if (on_device) {
        hipLaunchKernelGGL(k_provide_control, dim3((B * (nx + nq) + 63) / 64), dim3(64), 0, s, B, N, nq, accept, x_temp,
                           u_temp, xg, ug, u_apply, (const uint8_t*)nullptr, (const uint8_t*)nullptr, (const double*)nullptr);
        HIPCHK(h, hipGetLastError());
        return SMPC_OK;
    }
    int rc;
    if ((rc = ensure_tmp(h, sizeof(double) * (2 * nX + 2 * nU + (size_t)B * nq) + sizeof(int32_t) * B + 64))) return rc;
    double* d_xt = (double*)h->d_tmp;
    double* d_ut = d_xt + nX;
    double* d_xg = d_ut + nU;
    double* d_ug = d_xg + nX;
    double* d_ua = d_ug + nU;
    int32_t* d_acc = (int32_t*)(d_ua + (size_t)B * nq);
    HIPCHK(h, hipMemcpyAsync(d_xt, x_temp, sizeof(double) * nX, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(d_ut, u_temp, sizeof(double) * nU, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(d_xg, xg, sizeof(double) * nX, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(d_ug, ug, sizeof(double) * nU, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(d_acc, accept, sizeof(int32_t) * B, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_provide_control, dim3((B * (nx + nq) + 63) / 64), dim3(64), 0, s, B, N, nq, d_acc, d_xt, d_ut,
                       d_xg, d_ug, d_ua, (const uint8_t*)nullptr, (const uint8_t*)nullptr, (const double*)nullptr);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(xg, d_xg, sizeof(double) * nX, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(ug, d_ug, sizeof(double) * nU, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(u_apply, d_ua, sizeof(double) * B * nq, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return SMPC_OK;
}

int smpc_check_trajectory(smpc_handle* h, int B, int n_nodes, const double* x, const double* x_min, const double* x_max,
                          double tol_x, const double* row_lb_chk, const double* row_ub_chk, double alpha, double tol_safe,
                          int32_t* state_ok, int32_t* nn_ok, int on_device) {
    if (!h) return SMPC_EINVAL;
    if (B <= 0 || n_nodes <= 0 || !x || !x_min || !x_max || !state_ok) return fail(h, SMPC_EINVAL, "bad argument");
    if (h->desc.n_rows > 0 && (!row_lb_chk || !row_ub_chk)) return fail(h, SMPC_EINVAL, "row check bounds missing");
    if (nn_ok && h->nlayers == 0) return fail(h, SMPC_ESTATE, "nn_ok requested but smpc_set_mlp was not called");
    (void)hipSetDevice(h->device);
    const int nq = h->desc.nq, nx = 2 * nq;
    hipStream_t s = h->stream;
    const size_t M = (size_t)B * n_nodes;
    // (x_min / x_max / row bounds are host pointers on both paths: small, constant per caller)
    int rc;
    const size_t big = on_device ? 0 : sizeof(double) * M * nx + sizeof(int32_t) * (B + M) + 64;
    if ((rc = ensure_tmp(h, big))) return rc;
    if ((rc = upload_check_bounds(h, x_min, x_max, row_lb_chk, row_ub_chk))) return rc;
    const double* d_x = x;
    int32_t *d_ok = state_ok, *d_nn = nn_ok;
    if (!on_device) {
        double* dx = (double*)h->d_tmp;
        HIPCHK(h, hipMemcpyAsync(dx, x, sizeof(double) * M * nx, hipMemcpyHostToDevice, s));
        d_x = dx;
        d_ok = (int32_t*)(dx + M * nx);
        d_nn = d_ok + B;
    }
    if ((rc = check_nodes_dev(h, B, n_nodes, d_x, tol_x, n_nodes, alpha, tol_safe, d_ok, nn_ok ? d_nn : nullptr))) return rc;
    if (!on_device) {
        HIPCHK(h, hipMemcpyAsync(state_ok, d_ok, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
        if (nn_ok) HIPCHK(h, hipMemcpyAsync(nn_ok, d_nn, sizeof(int32_t) * M, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
    }
    return SMPC_OK;
}

int smpc_plant_step(smpc_handle* h, int B, const double* x, const double* u, const smpc_joint* joints_noisy,
                    const double* tau_noise, double* x_next, double* u_eff, int on_device) {
    if (!h) return SMPC_EINVAL;
    if (B <= 0 || !x || !u || !x_next) return fail(h, SMPC_EINVAL, "bad argument");
    (void)hipSetDevice(h->device);
    const int nq = h->desc.nq, nx = 2 * nq;
    hipStream_t s = h->stream;
    const double *dx = x, *du = u, *dn = tau_noise;
    const smpc_joint* dj = joints_noisy;
    double *dxn = x_next, *due = u_eff;
    if (!on_device) {
        int rc;
        const size_t bytes = sizeof(double) * B * (2 * (size_t)nx + 3 * (size_t)nq) + sizeof(smpc_joint) * (size_t)B * nq + 64;
        if ((rc = ensure_tmp(h, bytes))) return rc;
        double* w = (double*)h->d_tmp;
        double* wx = w; w += (size_t)B * nx;
        double* wu = w; w += (size_t)B * nq;
        double* wn = w; w += (size_t)B * nq;
        dxn = w; w += (size_t)B * nx;
        due = w; w += (size_t)B * nq;
        smpc_joint* wj = (smpc_joint*)w;
        HIPCHK(h, hipMemcpyAsync(wx, x, sizeof(double) * B * nx, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(wu, u, sizeof(double) * B * nq, hipMemcpyHostToDevice, s));
        dx = wx; du = wu;
        if (tau_noise) { HIPCHK(h, hipMemcpyAsync(wn, tau_noise, sizeof(double) * B * nq, hipMemcpyHostToDevice, s)); dn = wn; }
        if (joints_noisy) { HIPCHK(h, hipMemcpyAsync(wj, joints_noisy, sizeof(smpc_joint) * (size_t)B * nq, hipMemcpyHostToDevice, s)); dj = wj; }
    }
    switch (nq) {
    case 5: hipLaunchKernelGGL((k_plant_step<5>), dim3((B + 8) / 9), dim3(64), 0, s, h->d_desc, B, dx, du, dj, dn, dxn, due); break;
    case 6: hipLaunchKernelGGL((k_plant_step<6>), dim3((B + 7) / 8), dim3(64), 0, s, h->d_desc, B, dx, du, dj, dn, dxn, due); break;
    default: hipLaunchKernelGGL((k_plant_step<7>), dim3((B + 6) / 7), dim3(64), 0, s, h->d_desc, B, dx, du, dj, dn, dxn, due); break;
    }
    HIPCHK(h, hipGetLastError());
    if (!on_device) {
        HIPCHK(h, hipMemcpyAsync(x_next, dxn, sizeof(double) * B * nx, hipMemcpyDeviceToHost, s));
        if (u_eff) HIPCHK(h, hipMemcpyAsync(u_eff, due, sizeof(double) * B * nq, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
    }
    return SMPC_OK;
}

int smpc_rollout_batch(smpc_handle* h, int B, int n_steps, const double* x0, double* x_guess, double* u_guess,
                       const double* p, const smpc_joint* joints_noisy, const double* tau_noise, double* x_traj,
                       double* u_traj, int32_t* status_traj, int32_t* iter_traj, int on_device) {
    if (!h) return SMPC_EINVAL;
    if (B <= 0 || n_steps <= 0 || !x0 || !x_guess || !u_guess || !p || !x_traj || !u_traj || !status_traj)
        return fail(h, SMPC_EINVAL, "bad argument");
    (void)hipSetDevice(h->device);
    const int N = h->N, nq = h->desc.nq, nx = 2 * nq;
    hipStream_t s = h->stream;
    int rc;
    const size_t nX = (size_t)B * (N + 1) * nx, nU = (size_t)B * N * nq, nP = (size_t)B * (N + 1) * SMPC_NP;
    const size_t sx = (size_t)B * nx, su = (size_t)B * nq;
    // device views of the arguments (host pointers: one staging allocation for the call)
    char* stage = nullptr;
    const double *dx0 = x0, *dp = p, *dnoise = tau_noise;
    const smpc_joint* dj = joints_noisy;
    double *dxg = x_guess, *dug = u_guess, *dxt = x_traj, *dut = u_traj;
    int32_t *dst = status_traj, *dit = iter_traj;
    size_t o_x0 = 0, o_xg, o_ug, o_p, o_xt, o_ut, o_j, o_n, o_st, o_it, o_end;
    if (!on_device) {
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        o_xg = al(o_x0 + sizeof(double) * sx);
        o_ug = al(o_xg + sizeof(double) * nX);
        o_p = al(o_ug + sizeof(double) * nU);
        o_xt = al(o_p + sizeof(double) * nP);
        o_ut = al(o_xt + sizeof(double) * sx * (n_steps + 1));
        o_j = al(o_ut + sizeof(double) * su * n_steps);
        o_n = al(o_j + (joints_noisy ? sizeof(smpc_joint) * (size_t)B * nq : 0));
        o_st = al(o_n + (tau_noise ? sizeof(double) * su * n_steps : 0));
        o_it = al(o_st + sizeof(int32_t) * (size_t)B * n_steps);
        o_end = al(o_it + sizeof(int32_t) * (size_t)B * n_steps);
        if (o_end > h->roll_bytes) {
            HIPCHK(h, hipStreamSynchronize(s));
            if (h->d_roll) { (void)hipFree(h->d_roll); h->d_roll = nullptr; h->roll_bytes = 0; }
            if (hipMalloc((void**)&h->d_roll, o_end) != hipSuccess) return fail(h, SMPC_ENOMEM, "hipMalloc(%zu) failed", o_end);
            h->roll_bytes = o_end;
        }
        stage = h->d_roll;
        dxg = (double*)(stage + o_xg); dug = (double*)(stage + o_ug); dxt = (double*)(stage + o_xt); dut = (double*)(stage + o_ut);
        dst = (int32_t*)(stage + o_st); dit = (int32_t*)(stage + o_it);
        hipError_t e = hipMemcpyAsync(stage + o_x0, x0, sizeof(double) * sx, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(dxg, x_guess, sizeof(double) * nX, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(dug, u_guess, sizeof(double) * nU, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(stage + o_p, p, sizeof(double) * nP, hipMemcpyHostToDevice, s);
        if (e == hipSuccess && joints_noisy)
            e = hipMemcpyAsync(stage + o_j, joints_noisy, sizeof(smpc_joint) * (size_t)B * nq, hipMemcpyHostToDevice, s);
        if (e == hipSuccess && tau_noise)
            e = hipMemcpyAsync(stage + o_n, tau_noise, sizeof(double) * su * n_steps, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return fail(h, SMPC_EHIP, "staging copy failed: %s", hipGetErrorString(e));
        dx0 = (const double*)(stage + o_x0); dp = (const double*)(stage + o_p);
        dj = joints_noisy ? (const smpc_joint*)(stage + o_j) : nullptr;
        dnoise = tau_noise ? (const double*)(stage + o_n) : nullptr;
    } else if (!dit) {
        dit = h->d_it;
    }
    // ---- the steps.  Instances are independent, so a large batch is split into sub-batches that advance on their own
    // streams (worker handles): while one sub-batch's QP launch waits for its slowest instances the others' kernels fill the
    // chip (scripts/rollout_bench.py, B = 4096, round 4: 3.53 / 3.06 / 3.04 / 3.07 ms per step with 1 / 2 / 3 / 4 sub-batches,
    // twice on one box; the Python-driven three-stream loop of bench.py: 2.9 over the same 40 steps).  Results are the same bits
    // whatever the split.
    int n_sub = B >= 3072 ? 3 : (B >= 1024 ? 2 : 1);
    if (const char* ev = getenv("SMPC_ROLLOUT_STREAMS")) n_sub = atoi(ev);
    if (n_sub < 1) n_sub = 1;
    if (n_sub > B) n_sub = B;
    if (h->inst_B == B) n_sub = 1;                       // per-instance stage bounds are held by this handle only
    // (workspaces are allocated by whoever solves: the workers below when the batch is split -- the parent then holds none of
    //  the QP workspace / linearisation records of the full batch -- otherwise this handle)
    rc = SMPC_OK;
    hipError_t e = hipMemcpyAsync(dxt, dx0, sizeof(double) * sx, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return fail(h, SMPC_EHIP, "rollout init failed: %s", hipGetErrorString(e));
    if (n_sub > 1) {
        h->timed = 0;      // (the solves run on the workers: smpc_get_timing on this handle reports "nothing timed", not stale numbers)
        if ((rc = rollout_workers(h, n_sub))) return rc;
        if (!h->ev_fork) HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIPCHK(h, hipEventRecord(h->ev_fork, s));        // inputs (staging copies, x_traj[0]) are ordered before the workers
    }
    struct Slice { smpc_handle* w; int lo, n; };
    std::vector<Slice> slices;
    for (int k = 0; k < n_sub; k++) {
        const int base = B / n_sub, rem = B % n_sub;
        const int lo = k * base + (k < rem ? k : rem), n = base + (k < rem ? 1 : 0);
        smpc_handle* w = n_sub > 1 ? h->kids[k] : h;
        if (n_sub > 1) HIPCHK(h, hipStreamWaitEvent(w->stream, h->ev_fork, 0));
        const int nn_mode = h->desc.nn_mode;
        w->mlp_rows_whole = n_sub > 1 ? (nn_mode == SMPC_NN_TERMINAL ? (long)B : (long)B * N) : 0;
        int rw;
        if ((rw = ensure_batch(w, n)) || (rw = ensure_io(w, n))) return fail(h, rw, "worker %d: %s", k, w->err);
        if (w->pol_B < n) {     // fails / accept counters of the policy (kept in the handle: the device path does not synchronise)
            (void)hipStreamSynchronize(w->stream);
            if (w->d_pol) (void)hipFree(w->d_pol);
            w->d_pol = nullptr;
            w->pol_B = 0;
            if (hipMalloc((void**)&w->d_pol, sizeof(int32_t) * 2 * (size_t)n) != hipSuccess) return fail(h, SMPC_ENOMEM, "hipMalloc failed");
            w->pol_B = n;
        }
        if (hipMemsetAsync(w->d_pol, 0, sizeof(int32_t) * n, w->stream) != hipSuccess) return fail(h, SMPC_EHIP, "rollout init failed");
        slices.push_back({w, lo, n});
    }
    for (int t = 0; t < n_steps && rc == SMPC_OK; t++) {
        for (const Slice& sl : slices) {
            smpc_handle* w = sl.w;
            const int lo = sl.lo, n = sl.n;
            double* xt = dxt + (size_t)t * sx + (size_t)lo * nx;
            double* ut = dut + (size_t)t * su + (size_t)lo * nq;
            double* xg_w = dxg + (size_t)lo * (N + 1) * nx;
            double* ug_w = dug + (size_t)lo * N * nq;
            const double* p_w = dp + (size_t)lo * (N + 1) * SMPC_NP;
            int32_t* st_w = dst + (size_t)t * B + lo;
            int32_t* it_w = (on_device && !iter_traj) ? w->d_it : dit + (size_t)t * B + lo;
            int32_t *d_fails = w->d_pol, *d_accept = w->d_pol + n;
            if ((rc = smpc_guess_correction(w, n, xg_w, ug_w, 1))) break;
            if ((rc = smpc_solve_batch(w, n, xt, xg_w, ug_w, p_w, w->d_xo, w->d_uo, st_w, it_w, 1))) break;
            hipLaunchKernelGGL(k_accept, dim3((n + 63) / 64), dim3(64), 0, w->stream, n, st_w, d_fails, d_accept);
            if ((rc = smpc_provide_control(w, n, d_accept, w->d_xo, w->d_uo, xg_w, ug_w, ut, 1))) break;
            rc = smpc_plant_step(w, n, xt, ut, dj ? dj + (size_t)lo * nq : nullptr,
                                 dnoise ? dnoise + (size_t)t * su + (size_t)lo * nq : nullptr, xt + sx, nullptr, 1);
            if (rc) break;
        }
        if (rc && n_sub > 1) for (const Slice& sl : slices) if (sl.w->err[0]) snprintf(h->err, sizeof(h->err), "%s", sl.w->err);
    }
    if (n_sub > 1) {      // join: this handle's stream (copy-back, the caller's smpc_sync) comes after every worker
        if (!h->ev_join) HIPCHK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        for (const Slice& sl : slices) {
            HIPCHK(h, hipEventRecord(h->ev_join, sl.w->stream));
            HIPCHK(h, hipStreamWaitEvent(s, h->ev_join, 0));
        }
    }
    if (rc == SMPC_OK && !on_device) {
        e = hipMemcpyAsync(x_traj, dxt, sizeof(double) * sx * (n_steps + 1), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(u_traj, dut, sizeof(double) * su * n_steps, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(status_traj, dst, sizeof(int32_t) * (size_t)B * n_steps, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && iter_traj) e = hipMemcpyAsync(iter_traj, dit, sizeof(int32_t) * (size_t)B * n_steps, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(x_guess, dxg, sizeof(double) * nX, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(u_guess, dug, sizeof(double) * nU, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) rc = fail(h, SMPC_EHIP, "copy back failed: %s", hipGetErrorString(e));
    }
    if (stage) {   // host path: results are on the host when the call returns
        if (hipStreamSynchronize(s) != hipSuccess && rc == SMPC_OK) rc = fail(h, SMPC_EHIP, "stream synchronisation failed");
    }
    return rc;
}

// ---- the policy layer on the device ------------------------------------------------------------------------------------------
int smpc_policy_step(smpc_handle* h, int B, const smpc_policy_params* par, const smpc_policy_state* st, const double* x,
                     const uint8_t* stepping, const double* u_other, double* u_out, uint8_t* abort_out, int32_t* any_abort) {
    if (!h) return SMPC_EINVAL;
    if (B <= 0 || !par || !st || !x || !u_out || !abort_out || !any_abort) return fail(h, SMPC_EINVAL, "bad argument");
    if (!st->x_guess || !st->u_guess || !st->x_temp || !st->u_temp || !st->p || !st->x_viable || !st->fails ||
        !st->current_step || !st->status || !st->qp_iter)
        return fail(h, SMPC_EINVAL, "policy state incomplete");
    const int kind = par->kind;
    if (kind < SMPC_POLICY_NAIVE || kind > SMPC_POLICY_REAL_RECEDING) return fail(h, SMPC_EINVAL, "unknown policy kind %d", kind);
    const bool receding = kind == SMPC_POLICY_RECEDING || kind == SMPC_POLICY_REAL_RECEDING;
    if (receding && !st->r) return fail(h, SMPC_EINVAL, "receding policy without r");
    if (kind == SMPC_POLICY_REAL_RECEDING && (!par->stage_lo || !par->stage_hi)) return fail(h, SMPC_EINVAL, "stage_lo / stage_hi missing");
    if (stepping && !u_other) return fail(h, SMPC_EINVAL, "u_other missing");
    if (kind != SMPC_POLICY_NAIVE && (!par->x_min || !par->x_max || (h->desc.n_rows > 0 && (!par->row_lb_chk || !par->row_ub_chk))))
        return fail(h, SMPC_EINVAL, "check bounds missing");
    if (receding && h->nlayers == 0) return fail(h, SMPC_ESTATE, "receding policy but smpc_set_mlp was not called");
    (void)hipSetDevice(h->device);
    const int N = h->N, nq = h->desc.nq, nx = 2 * nq;
    hipStream_t s = h->stream;
    int rc;
    if ((rc = ensure_batch(h, B))) return rc;
    // scratch: state_ok [B] | safe [B][N+1] | accept [B] | active [B] bytes
    const size_t nI = (size_t)B * (N + 3);
    if ((rc = ensure_polw(h, sizeof(int32_t) * nI + (size_t)B + 64))) return rc;
    int32_t* d_ok = h->d_polw;
    int32_t* d_safe = d_ok + B;
    int32_t* d_acc = d_safe + (size_t)B * (N + 1);
    uint8_t* d_act = (uint8_t*)(d_acc + B);
    if (kind != SMPC_POLICY_NAIVE && (rc = upload_check_bounds(h, par->x_min, par->x_max, par->row_lb_chk, par->row_ub_chk))) return rc;
    // guessCorrection (not RealReceding, controller.py:524-565); the launch also resets *any_abort.  Every kind launches exactly
    // one of the two kernels that do so -- k_guess_correction here, k_policy_pre (RealReceding) below -- and both grids are
    // non-empty (B > 0, nq > 0), so thread 0 of block 0 always exists.
    if (kind != SMPC_POLICY_REAL_RECEDING)
        hipLaunchKernelGGL(k_guess_correction, dim3((B * nq + 63) / 64), dim3(64), 0, s, B, N, nq, h->desc.dt, st->x_guess,
                           st->u_guess, stepping, any_abort, d_ok);
    if (receding) {
        if (kind == SMPC_POLICY_REAL_RECEDING) {
            const size_t n = (size_t)B * (N + 1) * nx;
            if (n > h->inst_cap) {
                HIPCHK(h, hipStreamSynchronize(s));
                if ((rc = dev_alloc(h, &h->d_lo_b, n))) return rc;
                if ((rc = dev_alloc(h, &h->d_hi_b, n))) return rc;
                h->inst_cap = n;
                // (instances that never step keep valid bounds)
                HIPCHK(h, hipMemsetAsync(h->d_lo_b, 0, n * sizeof(double), s));
                HIPCHK(h, hipMemsetAsync(h->d_hi_b, 0, n * sizeof(double), s));
            }
            h->inst_B = B;
        }
        hipLaunchKernelGGL(k_policy_pre, dim3((unsigned)(((size_t)B * (N + 1) + 63) / 64)), dim3(64), 0, s, B, N, nx, kind,
                           stepping, st->r, st->p, st->x_guess, par->stage_lo, par->stage_hi, par->tube, h->d_lo_b, h->d_hi_b,
                           kind == SMPC_POLICY_REAL_RECEDING ? any_abort : (int32_t*)nullptr,
                           kind == SMPC_POLICY_REAL_RECEDING ? d_ok : (int32_t*)nullptr);
    }
    if (st->traj) {      // controller.py:153-156: the nodes' reference points follow the step counter
        if (st->traj_len < 1) return fail(h, SMPC_EINVAL, "traj_len must be >= 1");
        hipLaunchKernelGGL(k_policy_traj, dim3((unsigned)(((size_t)B * (N + 1) + 63) / 64)), dim3(64), 0, s, B, N, stepping,
                           st->current_step, st->traj, (long)st->traj_len, st->p);
    }
    HIPCHK(h, hipGetLastError());
    h->d_active = stepping;
    // (the receding policies carry the row at node r and at the end node, and test nodes r + 2 .. N afterwards -- in steady state one or
    //  two per instance: their compacted lists are short, whatever their capacity)
    h->mlp_rows_hint = receding ? 2L * B : 0L;
    DISPATCH_NQ(h, (launch_solve<NQ_>(h, B, x, st->x_guess, st->u_guess, st->p, st->x_temp, st->u_temp, st->status, st->qp_iter)));
    h->d_active = nullptr;
    if (rc) { h->mlp_rows_hint = 0; return rc; }
    if (kind != SMPC_POLICY_NAIVE) {
        // checkStateConstraints(x_temp) (+ checkSafeConstraints(x_temp) on every node for the receding policies)
        const int coll = par->collision_first_node ? 1 : N + 1;
        if (receding) {
            // the safe-set test is only ever read at nodes r + 2 .. N of the stepping instances (k_policy_post): list them
            const size_t M = (size_t)B * (N + 1);
            if ((rc = ensure_nn_idx(h, M))) return rc;
            hipLaunchKernelGGL(k_policy_safe_list, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, s, B, N, par->abort_flag, stepping, st->r,
                               h->d_nn_idx, h->d_nn_cnt);
        }
        if ((rc = check_nodes_dev(h, B, N + 1, st->x_temp, par->tol_x, coll, par->alpha, par->tol_safe, d_ok, receding ? d_safe : nullptr,
                                  receding, true))) {
            h->mlp_rows_hint = 0;
            return rc;
        }
    }
    h->mlp_rows_hint = 0;
    hipLaunchKernelGGL(k_policy_post, dim3((B + 63) / 64), dim3(64), 0, s, B, N, nx, kind, par->abort_flag, stepping, st->status,
                       d_ok, d_safe, st->x_guess, st->fails, st->current_step, st->r, st->x_viable, d_acc, d_act, abort_out, any_abort,
                       (receding && kind != SMPC_POLICY_NAIVE) ? h->d_nn_cnt : (int32_t*)nullptr);

    hipLaunchKernelGGL(k_provide_control, dim3((B * (nx + nq) + 63) / 64), dim3(64), 0, s, B, N, nq, d_acc, st->x_temp, st->u_temp,
                       st->x_guess, st->u_guess, u_out, stepping, d_act, u_other);
    HIPCHK(h, hipGetLastError());
    return SMPC_OK;
}

int smpc_loop_pre(smpc_handle* h, int B, int Nb, const smpc_loop_state* ls, const int64_t* r, const uint8_t* pending, double* u_other,
                  uint8_t* stepping) {
    if (!h) return SMPC_EINVAL;
    if (B <= 0 || Nb <= 0 || !ls || !u_other || !stepping) return fail(h, SMPC_EINVAL, "bad argument");
    if (!ls->x_cur || !ls->alive || !ls->sa || !ls->ja || !ls->x_abort || !ls->u_abort || !ls->step)
        return fail(h, SMPC_EINVAL, "loop state incomplete");
    (void)hipSetDevice(h->device);
    hipLaunchKernelGGL(k_loop_pre, dim3((B + 63) / 64), dim3(64), 0, h->stream, B, h->desc.nq, Nb, ls->x_cur, ls->alive, ls->sa,
                       ls->ja, ls->x_abort, ls->u_abort, r, ls->step, ls->r_log, u_other, stepping, pending, ls->resumed);
    HIPCHK(h, hipGetLastError());
    return SMPC_OK;
}

int smpc_loop_classify_aborts(smpc_handle* h, int B, const smpc_loop_state* ls, int reference_quirks, uint8_t* abort,
                              int32_t* any_event) {
    if (!h) return SMPC_EINVAL;
    if (B <= 0 || !ls || !abort || !any_event) return fail(h, SMPC_EINVAL, "bad argument");
    if (!ls->sa) return fail(h, SMPC_EINVAL, "loop state incomplete");
    (void)hipSetDevice(h->device);
    HIPCHK(h, hipMemsetAsync(any_event, 0, sizeof(int32_t), h->stream));
    hipLaunchKernelGGL(k_loop_classify_aborts, dim3((B + 63) / 64), dim3(64), 0, h->stream, B, reference_quirks, ls->resumed,
                       ls->sa, abort, any_event);
    HIPCHK(h, hipGetLastError());
    return SMPC_OK;
}

int smpc_loop_apply_backup(smpc_handle* h, int B, int Nb, const smpc_loop_state* ls, int n_c, const int64_t* rows,
                           const int32_t* status_c, const double* x_c, const double* u_c, uint8_t* viable, double* u,
                           uint8_t* pending) {
    if (!h) return SMPC_EINVAL;
    if (B <= 0 || Nb <= 0 || n_c < 0 || n_c > B || !ls || !viable || !u || !pending) return fail(h, SMPC_EINVAL, "bad argument");
    if (n_c == 0) return SMPC_OK;
    if (!rows || !status_c || !x_c || !u_c) return fail(h, SMPC_EINVAL, "bad argument");
    if (!ls->x_cur || !ls->alive || !ls->sa || !ls->collided || !ls->ja || !ls->last_x || !ls->last_u || !ls->x_abort ||
        !ls->u_abort || !ls->step)
        return fail(h, SMPC_EINVAL, "loop state incomplete");
    (void)hipSetDevice(h->device);
    hipLaunchKernelGGL(k_loop_apply_backup, dim3((n_c + 63) / 64), dim3(64), 0, h->stream, n_c, h->desc.nq, Nb, rows, status_c, x_c, u_c,
                       ls->x_cur, ls->step, ls->alive, ls->sa, ls->collided, viable, ls->ja, ls->last_x, ls->last_u, ls->x_abort,
                       ls->u_abort, u, pending);
    HIPCHK(h, hipGetLastError());
    return SMPC_OK;
}

int smpc_loop_post(smpc_handle* h, int B, const smpc_policy_params* par, const smpc_loop_state* ls, const double* u,
                   const smpc_joint* joints_noisy, const double* tau_noise) {
    if (!h) return SMPC_EINVAL;
    if (B <= 0 || !par || !ls || !u) return fail(h, SMPC_EINVAL, "bad argument");
    if (!ls->x_cur || !ls->alive || !ls->collided || !ls->last_x || !ls->last_u || !ls->step || !ls->x_log || !ls->u_log)
        return fail(h, SMPC_EINVAL, "loop state incomplete");
    if (!par->x_min || !par->x_max || (h->desc.n_rows > 0 && (!par->row_lb_chk || !par->row_ub_chk)))
        return fail(h, SMPC_EINVAL, "check bounds missing");
    (void)hipSetDevice(h->device);
    const int nq = h->desc.nq, nx = 2 * nq;
    hipStream_t s = h->stream;
    int rc;
    // scratch behind smpc_policy_step's (same handle, stream-ordered): x_next [B][nx] | ok [B]
    const size_t off = (sizeof(int32_t) * (size_t)B * (h->N + 3) + (size_t)B + 64 + 15) & ~(size_t)15;
    if ((rc = ensure_polw(h, off + sizeof(double) * (size_t)B * nx + sizeof(int32_t) * (size_t)B))) return rc;
    double* d_xn = (double*)((char*)h->d_polw + off);
    int32_t* d_okn = (int32_t*)(d_xn + (size_t)B * nx);
    if ((rc = upload_check_bounds(h, par->x_min, par->x_max, par->row_lb_chk, par->row_ub_chk))) return rc;
    if ((rc = smpc_plant_step(h, B, ls->x_cur, u, joints_noisy, tau_noise, d_xn, nullptr, 1))) return rc;
    // one node per instance: the model bounds widened by tol_x and the rows against their check bounds = checkStateConstraints
    if ((rc = check_nodes_dev(h, B, 1, d_xn, par->tol_x, 1, 0.0, 0.0, d_okn, nullptr))) return rc;
    hipLaunchKernelGGL(k_loop_post, dim3((B + 63) / 64), dim3(64), 0, s, B, nq, u, d_xn, d_okn, ls->step, ls->x_log, ls->u_log,
                       ls->alive, ls->collided, ls->last_x, ls->last_u, ls->x_cur);
    hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(1), 0, s, ls->step);
    HIPCHK(h, hipGetLastError());
    return SMPC_OK;
}

int smpc_sync(smpc_handle* h) {
    if (!h) return SMPC_EINVAL;
    (void)hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SMPC_OK;
}

void* smpc_stream(smpc_handle* h) { return h ? (void*)h->stream : nullptr; }

int smpc_enable_timing(smpc_handle* h, int on) {
    if (!h) return SMPC_EINVAL;
    h->timing = on == 2 ? 2 : (on ? 1 : 0);
    h->timed = 0;
    h->timed_count = 0;
    for (bool& c : h->ev_complete) c = false;
    return SMPC_OK;
}

int smpc_get_timing(smpc_handle* h, float* ms4) {
    if (!h || !ms4) return SMPC_EINVAL;
    if (!h->timing) return fail(h, SMPC_ESTATE, "timing not enabled");
    if (!h->timed || !h->ev_complete[h->ev_cur]) return fail(h, SMPC_ESTATE, "no solve has been timed since smpc_enable_timing");
    (void)hipSetDevice(h->device);
    HIPCHK(h, hipEventSynchronize(h->ev_t[3]));
    // (stage-builder path: the network pass comes first, ev0 -> ev1, then linearisation + set-up in one kernel, ev1 -> ev2)
    const bool nw = h->ev_new_order[h->ev_cur];
    HIPCHK(h, hipEventElapsedTime(&ms4[nw ? 1 : 0], h->ev_t[0], h->ev_t[1]));
    HIPCHK(h, hipEventElapsedTime(&ms4[nw ? 0 : 1], h->ev_t[1], h->ev_t[2]));
    HIPCHK(h, hipEventElapsedTime(&ms4[2], h->ev_t[2], h->ev_t[3]));
    HIPCHK(h, hipEventElapsedTime(&ms4[3], h->ev_t[0], h->ev_t[3]));
    return SMPC_OK;
}

int smpc_get_qp_timing(smpc_handle* h, float* ms2) {
    if (!h || !ms2) return SMPC_EINVAL;
    if (!h->timing) return fail(h, SMPC_ESTATE, "timing not enabled");
    if (!h->timed || !h->ev_complete[h->ev_cur]) return fail(h, SMPC_ESTATE, "no solve has been timed since smpc_enable_timing");
    (void)hipSetDevice(h->device);
    HIPCHK(h, hipEventSynchronize(h->ev_t[3]));
    HIPCHK(h, hipEventElapsedTime(&ms2[0], h->ev_t[2], h->ev_t[4]));
    HIPCHK(h, hipEventElapsedTime(&ms2[1], h->ev_t[4], h->ev_t[3]));
    return SMPC_OK;
}

int smpc_get_timing_history(smpc_handle* h, int back, float* ms6) {
    if (!h || !ms6 || back < 0) return SMPC_EINVAL;
    for (int i = 0; i < 6; i++) ms6[i] = 0.0f;
    if (!h->timing || back >= smpc_handle::EV_RING || (long)back >= h->timed_count) return SMPC_OK;
    (void)hipSetDevice(h->device);
    const int slot = (h->ev_cur - back + 2 * smpc_handle::EV_RING) % smpc_handle::EV_RING;
    if (!h->ev_complete[slot]) return SMPC_OK;     // the solve that owns the slot returned early: valid stays 0
    hipEvent_t* ev = h->ev_sets[slot];
    if (hipEventQuery(ev[3]) != hipSuccess) { (void)hipGetLastError(); return SMPC_OK; }   // not finished yet: valid stays 0
    const bool nw = h->ev_new_order[slot];
    HIPCHK(h, hipEventElapsedTime(&ms6[nw ? 1 : 0], ev[0], ev[1]));
    HIPCHK(h, hipEventElapsedTime(&ms6[nw ? 0 : 1], ev[1], ev[2]));
    HIPCHK(h, hipEventElapsedTime(&ms6[2], ev[2], ev[4]));
    HIPCHK(h, hipEventElapsedTime(&ms6[3], ev[4], ev[3]));
    HIPCHK(h, hipEventElapsedTime(&ms6[4], ev[0], ev[3]));
    ms6[5] = 1.0f;
    return SMPC_OK;
}

int smpc_accumulate_stats(smpc_handle* h, int B, const int32_t* status, const int32_t* qp_iter, unsigned long long* acc3) {
    if (!h) return SMPC_EINVAL;
    if (B < 0 || !status || !acc3) return fail(h, SMPC_EINVAL, "bad argument");
    if (B == 0) return SMPC_OK;
    (void)hipSetDevice(h->device);
    hipLaunchKernelGGL(k_accumulate_stats, dim3((B + 63) / 64), dim3(64), 0, h->stream, B, status, qp_iter, acc3);
    HIPCHK(h, hipGetLastError());
    return SMPC_OK;
}

int smpc_get_qp_wave_stats(smpc_handle* h, double* out3) {
    if (!h || !out3) return SMPC_EINVAL;
    if (h->timing != 1 || !h->timed || !h->d_wstat) return fail(h, SMPC_ESTATE, "no solve has been timed since smpc_enable_timing(1)");
    (void)hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    unsigned long long w[4];
    HIPCHK(h, hipMemcpy(w, h->d_wstat, sizeof(w), hipMemcpyDeviceToHost));
    const double tick_us = 1e-2;   // s_memrealtime: constant 100 MHz
    out3[0] = w[3] ? (double)w[0] / (double)w[3] * tick_us : 0.0;   // mean busy time of a half-wave (one instance), us
    out3[1] = w[3] ? (double)(w[2] - w[1]) * tick_us : 0.0;          // first start -> last end, us
    out3[2] = (double)w[3];
    return SMPC_OK;
}

}  // extern "C"

// Test hook (not part of include/smpc.h): the stage records of the QP workspace as either path builds them (path 1: MLP ->
// k_stage_build; path 0: k_node_linearise -> MLP -> k_qp_setup), copied to the host, and the layout's offsets -- so that
// tests/test_gpu_parity.py can compare the two builders block by block.  Host pointers.  layout[24] = {stride, nIMG, oIMG, oSL,
// nF, oR0, oR1, oR2, oCZA, oCZN, oZ, oZN, NRT, nJ, doubles per instance, record version, iTT, iGT, iGN, iB, iSC, iHQQ, iGZ, 0}.
extern "C" int smpc_debug_stage_records(smpc_handle* h, int B, const double* x0, const double* xg, const double* ug, const double* p,
                                        int path, double* ws_out, int32_t* layout) {
    if (!h || B <= 0 || !x0 || !xg || !ug || !p || !ws_out || !layout) return SMPC_EINVAL;
    (void)hipSetDevice(h->device);
    int rc;
    if ((rc = ensure_batch(h, B)) || (rc = ensure_io(h, B))) return rc;
    const int N = h->N, nx = 2 * h->desc.nq, nu = h->desc.nq;
    hipStream_t s = h->stream;
    HIPCHK(h, hipMemcpyAsync(h->d_x0, x0, sizeof(double) * B * nx, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->d_xg, xg, sizeof(double) * B * (N + 1) * nx, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->d_ug, ug, sizeof(double) * B * N * nu, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->d_p, p, sizeof(double) * B * (N + 1) * SMPC_NP, hipMemcpyHostToDevice, s));
    const size_t per = ws_doubles_per_instance(h->desc, N);
    HIPCHK(h, hipMemsetAsync(h->d_ws, 0, per * (size_t)B * sizeof(double), s));
    DISPATCH_NQ(h, (launch_stage_records<NQ_>(h, B, h->d_x0, h->d_xg, h->d_ug, h->d_p, false, path)));
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(ws_out, h->d_ws, per * (size_t)B * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    auto fill = [&](auto Ly) {
        const int v[24] = {Ly.stride, Ly.nIMG, Ly.oIMG, Ly.oSL, Ly.nF, Ly.oR0, Ly.oR1, Ly.oR2, Ly.oCZA, Ly.oCZN, Ly.oZ, Ly.oZN, Ly.NRT, Ly.nJ,
                           (int)per, 12, Ly.iTT, Ly.iGT, Ly.iGN, Ly.iB, Ly.iSC, Ly.iHQQ, Ly.iGZ, 0};
        for (int i = 0; i < 24; i++) layout[i] = v[i];
    };
    switch (h->desc.nq) {
    case 5: fill(QpLayout<5>(h->desc.n_rows)); break;
    case 6: fill(QpLayout<6>(h->desc.n_rows)); break;
    default: fill(QpLayout<7>(h->desc.n_rows)); break;
    }
    return SMPC_OK;
}

#ifdef QP_PROFILE
// diagnostic builds only (not part of include/smpc.h): per-phase shader-clock sums of k_qp_ipm since the last call
extern "C" int smpc_debug_qp_profile(unsigned long long* out16) {
    unsigned long long zero[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(smpc::g_qp_prof), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(smpc::g_qp_prof), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
extern "C" int smpc_debug_qp_wg_profile(unsigned long long* out16) {
    unsigned long long zero[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(smpc::g_wg_prof), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(smpc::g_wg_prof), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif
