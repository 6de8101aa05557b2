// kernels_mlp.hpp -- the learned safe-set row: batched MLP forward + input-gradient on the matrix cores.
//
// Replaces l4casadi + libtorch (reference src/safe_mpc/safe_set.py:89-94, N3 in SURVEY section 2), which evaluates the
// network one sample at a time on the CPU.  Here all (instance, node) pairs that carry the row are gathered into one
// [M x in] matrix and pushed through  in -> H -> H -> ... -> 1  as dense GEMMs.  The reference's torch model is fp32
// (safe_set.py:75-78), so the GEMMs use the exact-fp32 MFMA  v_mfma_f32_32x32x2_f32  (a k-ordered fmaf chain, no
// reduced-precision path): one wave owns a 32 x 64 tile of C, operands come straight from L2-resident global memory
// (A: one dwordx4 per lane per 8-deep K chunk, B: eight coalesced dword rows), no LDS.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_model.hpp"

namespace smpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MLP_KPAD = 16;   // padded input width (2*nn_dof <= 14)
constexpr int MLP_NPAD = 64;   // padded width of the last backward GEMM (tile width)

__device__ __forceinline__ float gelu_tanh_f(float a, float* dg) {
    // GELU(approximate='tanh') (parser.py:99) and its derivative, fp32
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float a2 = a * a;
    const float inner = k0 * (a + k1 * a * a2);
    const float th = tanhf(inner);
    *dg = 0.5f * (1.0f + th) + 0.5f * a * (1.0f - th * th) * k0 * (1.0f + 3.0f * k1 * a2);
    return 0.5f * a * (1.0f + th);
}

// the activations the reference's parser offers (parser.py:95-102): value and derivative, fp32
__device__ __forceinline__ float act_f(int act, float a, float* dg) {
    switch (act) {
    case SMPC_ACT_RELU: *dg = a > 0.0f ? 1.0f : 0.0f; return a > 0.0f ? a : 0.0f;
    case SMPC_ACT_ELU: { const float e = expf(a); *dg = a > 0.0f ? 1.0f : e; return a > 0.0f ? a : e - 1.0f; }     // alpha = 1 (torch default)
    case SMPC_ACT_TANH: { const float t = tanhf(a); *dg = 1.0f - t * t; return t; }
    case SMPC_ACT_SILU: { const float sg = 1.0f / (1.0f + expf(-a)); *dg = sg * (1.0f + a * (1.0f - sg)); return a * sg; }
    default: return gelu_tanh_f(a, dg);
    }
}

enum { EPI_BIAS_GELU = 0, EPI_MUL = 1, EPI_PLAIN = 2 };   // (EPI_BIAS_GELU: bias + the configured activation)

// C[M x N] = A[M x K] * Bm[K x N]  (row-major, M % 32 == 0, N % 64 == 0, K % 8 == 0)
//   EPI_BIAS_GELU: out1 = gelu(acc + bias[n]), out2 = gelu'(acc + bias[n])
//   EPI_MUL:       out1 = acc * aux[m][n]
//   EPI_PLAIN:     out1 = acc
template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_f32(int M, int N, int K, const float* __restrict__ A,
                                                  const float* __restrict__ Bm, const float* __restrict__ bias,
                                                  const float* __restrict__ aux, float* __restrict__ out1,
                                                  float* __restrict__ out2, const int32_t* __restrict__ m_live, int act) {
    // (the wavefronts of a block are independent -- no LDS, no barrier -- so the block size is the launch's choice: 256 threads, or
    //  64 where the kernel has to find room beside resident QP wavefronts, engine.hip run_mlp)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m0 = (blockIdx.x * (int)(blockDim.x >> 6) + wave) * 32;
    const int n0 = blockIdx.y * 64;
    if (m0 >= M) return;
    if (m_live && m0 >= *m_live) return;   // (rows past the live count of a compacted list: nothing to compute)
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
    const float* arow = A + (size_t)(m0 + li) * K + 4 * lh;
    const float* bcol = Bm + (size_t)(4 * lh) * N + n0 + li;
    // One k-step = 8 MFMAs, shorter than an L2 round trip for its operands: the main loop takes four k-steps at a time and
    // issues all their loads (4 x (1 + 8)) before the 32 MFMAs that consume them.
    auto kstep = [&](const f32x4 av, const float* b0, const float* b1) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], b0[r], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], b1[r], acc1, 0, 0, 0);
        }
    };
    int kc = 0;
    for (; kc + 32 <= K; kc += 32) {
        f32x4 av[4];
        float b0[4][4], b1[4][4];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            av[s4] = *reinterpret_cast<const f32x4*>(arow + kc + 8 * s4);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                b0[s4][r] = bcol[(size_t)(kc + 8 * s4 + r) * N];
                b1[s4][r] = bcol[(size_t)(kc + 8 * s4 + r) * N + 32];
            }
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) kstep(av[s4], b0[s4], b1[s4]);
    }
    for (; kc < K; kc += 8) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(arow + kc);
        float b0[4], b1[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            b0[r] = bcol[(size_t)(kc + r) * N];
            b1[r] = bcol[(size_t)(kc + r) * N + 32];
        }
        kstep(av, b0, b1);
    }
    // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const int row = m0 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int col = n0 + 32 * half + li;
            const float a = half ? acc1[reg] : acc0[reg];
            const size_t o = (size_t)row * N + col;
            if (EPI == EPI_BIAS_GELU) {
                float dg;
                out1[o] = act_f(act, a + bias[col], &dg);
                out2[o] = dg;
            } else if (EPI == EPI_MUL) {
                out1[o] = a * aux[o];
            } else {
                out1[o] = a;
            }
        }
    }
}

// The same product for large M (the safe-set row on every node: M = B * N rows): a 256-thread block owns a 128 x 128 tile
// of C, each of its four wavefronts a 64 x 64 quarter (2 x 2 MFMA tiles, 64 accumulator registers); A and B are staged
// through LDS in k-chunks of 16 (register prefetch of the next chunk while the current one feeds 32 MFMAs per wave), so an
// operand is read from L2 once per block instead of once per wavefront.  M % 128 == 0, N % 128 == 0, K % 16 == 0.
template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_f32_tiled(int M, int N, int K, const float* __restrict__ A,
                                                        const float* __restrict__ Bm, const float* __restrict__ bias,
                                                        const float* __restrict__ aux, float* __restrict__ out1,
                                                        float* __restrict__ out2, const int32_t* __restrict__ m_live, int act) {
    constexpr int TM = 128, TN = 128, TK = 16, AS = TK + 1, BS = TN + 32;   // row strides: conflict-free column / row reads
    __shared__ float As[2][TM * AS];
    __shared__ __attribute__((aligned(16))) float Bs[2][TK * BS];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    if (m_live && m0 >= *m_live) return;   // (whole block: uniform, before the first barrier)
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
    // global -> register staging: A 128 x 16 = 512 float4 (2 per thread), B 16 x 128 = 512 float4 (2 per thread)
    const int ar = t >> 2, ac = (t & 3) * 4;          // A: rows ar, ar + 64; columns ac .. ac+3
    const int bk = t >> 5, bn = (t & 31) * 4;         // B: rows bk, bk + 8;  columns bn .. bn+3
    const float* ap = A + (size_t)(m0 + ar) * K + ac;
    const float* bp = Bm + (size_t)bk * N + n0 + bn;
    f32x4 ra0, ra1, rb0, rb1;
    auto gload = [&](int k0) {
        ra0 = *reinterpret_cast<const f32x4*>(ap + k0);
        ra1 = *reinterpret_cast<const f32x4*>(ap + (size_t)64 * K + k0);
        rb0 = *reinterpret_cast<const f32x4*>(bp + (size_t)k0 * N);
        rb1 = *reinterpret_cast<const f32x4*>(bp + (size_t)(k0 + 8) * N);
    };
    auto sstore = [&](int buf) {
        float* a = As[buf] + ar * AS + ac;
#pragma unroll
        for (int c = 0; c < 4; c++) { a[c] = ra0[c]; a[64 * AS + c] = ra1[c]; }
        *reinterpret_cast<f32x4*>(Bs[buf] + bk * BS + bn) = rb0;
        *reinterpret_cast<f32x4*>(Bs[buf] + (bk + 8) * BS + bn) = rb1;
    };
    gload(0);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += TK) {
        const bool more = k0 + TK < K;
        if (more) gload(k0 + TK);
        const float* a = As[buf] + (wm + li) * AS + lh;
        const float* b = Bs[buf] + lh * BS + wn + li;
#pragma unroll
        for (int s2 = 0; s2 < TK / 2; s2++) {
            const float a0 = a[2 * s2], a1 = a[32 * AS + 2 * s2];
            const float b0 = b[2 * s2 * BS], b1 = b[2 * s2 * BS + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // C/D map of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = m0 + wm + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                const int col = n0 + wn + 32 * j + li;
                const float v = acc[i][j][reg];
                const size_t o = (size_t)row * N + col;
                if (EPI == EPI_BIAS_GELU) {
                    float dg;
                    out1[o] = act_f(act, v + bias[col], &dg);
                    out2[o] = dg;
                } else if (EPI == EPI_MUL) {
                    out1[o] = v * aux[o];
                } else {
                    out1[o] = v;
                }
            }
}

// features of the safe-set network (safe_set.py:82-87): s = [(q - mean)/std ; v/|v|], v = qd with eps on v_0.
// row m <-> node:  mode 0: node = m (plain list of states);  mode 1 (terminal): node = m (N+1) + N;
//                  mode 2 (all nodes but the first): b = m / N, k = 1 + m % N;
//                  mode 3: node = idx[m] for m < *m_live -- the nodes of mode 2 whose row is switched on (k_nn_compact).
__device__ __forceinline__ long nn_row_to_node(int mode, int N, int m, const int32_t* __restrict__ idx = nullptr) {
    if (mode == 0) return m;
    if (mode == 1) return (long)m * (N + 1) + N;
    if (mode == 3) return idx[m];
    return (long)(m / N) * (N + 1) + 1 + m % N;
}

// The per-node switch of the safe-set row (utils.py:207-210, p[4] of the node): the receding policy carries the row at two
// nodes per instance (controller.py:452-469), so the network is evaluated on the list of live nodes only.  The order of the
// list is whatever the atomics give; every row's result is independent of its position.
__global__ void k_nn_compact(int M, int N, const double* __restrict__ p, int32_t* __restrict__ idx, int32_t* __restrict__ m_live) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const long node = nn_row_to_node(2, N, m);
    if (p[node * SMPC_NP + 4] > 0.0) idx[atomicAdd(m_live, 1)] = (int32_t)node;
}
template <int NQ>
__global__ void k_nn_features(const smpc_problem_desc* __restrict__ D, int M, int Mpad, int N, int mode,
                              const double* __restrict__ xg, float* __restrict__ S, const int32_t* __restrict__ idx,
                              const int32_t* __restrict__ m_live) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= Mpad) return;
    const int live = m_live ? *m_live : M;
    if (m >= ((live + 127) & ~127)) return;      // (only the tiles that will be computed need defined rows)
    float* s = S + (size_t)m * MLP_KPAD;
#pragma unroll
    for (int i = 0; i < MLP_KPAD; i++) s[i] = 0.0f;
    if (m >= live) return;
    const double* x = xg + nn_row_to_node(mode, N, m, idx) * (2 * NQ);
    const int nd = D->nn_dof;
    double v[NQ], vn2 = 0.0;
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        v[i] = i < nd ? x[NQ + i] + (i == 0 ? D->nn_eps : 0.0) : 0.0;
        vn2 += v[i] * v[i];
    }
    const double inv = 1.0 / sqrt(vn2);
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        if (i < nd) {
            s[i] = (float)((x[i] - D->nn_mean[i]) / D->nn_std[i]);
            s[nd + i] = (float)(v[i] * inv);
        }
    }
}

// output layer y = a . w + b  and  delta = w (.) gelu'(z) of the last hidden layer; one wave per row
__global__ __launch_bounds__(256) void k_nn_output(int M, int H, const float* __restrict__ A, const float* __restrict__ Dg,
                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ y, float* __restrict__ delta,
                                                   const int32_t* __restrict__ m_live) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    if (m_live && row >= ((*m_live + 127) & ~127)) return;
    float acc = 0.0f;
    for (int c = lane; c < H; c += 64) {
        const size_t o = (size_t)row * H + c;
        acc = fmaf(A[o], w[c], acc);
        delta[o] = w[c] * Dg[o];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) y[row] = acc + bias[0];
}

// chain rule back to the state (safe_set.py:82-94) and the per-node switch (utils.py:207-210): writes nn_val, nn_grad
template <int NQ>
__global__ void k_nn_chain(const smpc_problem_desc* __restrict__ D, int M, int N, int mode,
                           const double* __restrict__ xg, const double* __restrict__ p, const float* __restrict__ y,
                           const float* __restrict__ GS, double* __restrict__ out, const int32_t* __restrict__ idx,
                           const int32_t* __restrict__ m_live, int compact) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    if (m_live && m >= *m_live) return;
    const long node = nn_row_to_node(mode, N, m, idx);
    const double* x = xg + node * (2 * NQ);
    const double* pk = p + node * SMPC_NP;
    // compact: out is nn[node][1 + 2 NQ] = (value, gradient) -- what k_stage_build / the QP's stage builder read; otherwise the
    // node's linearisation record (interleaved tile: element f at o[f * EV_TILE])
    // (compact = 2, the row on the end node only: nn[instance][1 + 2 NQ], a buffer (N + 1) times smaller)
    double* const o = compact ? out + (compact == 2 ? node / (N + 1) : node) * (1 + 2 * NQ) : ev_node(out, node);
    const long es = compact ? 1 : EV_TILE;
    const int o_val = compact ? 0 : SMPC_EV_OFF(nn_val), o_grad = compact ? 1 : SMPC_EV_OFF(nn_grad);
    if (!(pk[4] > 0.0)) return;  // switched off: row sits mid-bounds, leave (0, 0)
    const int nd = D->nn_dof;
    const float* gs = GS + (size_t)m * MLP_NPAD;
    double v[NQ], vn2 = 0.0, gdv = 0.0;
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        v[i] = i < nd ? x[NQ + i] + (i == 0 ? D->nn_eps : 0.0) : 0.0;
        vn2 += v[i] * v[i];
    }
    const double vn = sqrt(vn2);
#pragma unroll
    for (int i = 0; i < NQ; i++)
        if (i < nd) gdv += (double)gs[nd + i] * v[i];
    const double kap = (100.0 - pk[3]) / 100.0;
    o[o_val * es] = (double)y[m] * kap - vn;
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        if (i < nd) {
            o[(o_grad + i) * es] = kap * (double)gs[i] / D->nn_std[i];
            o[(o_grad + NQ + i) * es] = kap * ((double)gs[nd + i] / vn - v[i] * gdv / (vn * vn * vn)) - v[i] / vn;
        }
    }
}

// =========================================================================================================================
// k_mlp_fused: the WHOLE network pass for small row counts (the terminal safe-set row: M = B rows) in one kernel -- features,
// three hidden layers forward, output, three layers backward, input gradient, chain rule -- instead of nine launches whose
// activations make a round trip through L2 between layers (188 us at M = 4096, 7 % of the fp32 MFMA peak: launch latency).
// A 256-thread block owns 16 rows.  Activations live in two LDS buffers (ping-pong); the activation derivatives stay in the
// registers of the wavefront that produced them, in the accumulator layout the backward product's epilogue needs them in.
// v_mfma_f32_16x16x4_f32: wave w owns the 64 columns 64 w .. 64 w + 63 of every 256-wide layer as four 16-column tiles taken
// with stride 4 (tile t = columns 64 w + 4 j + t), so that one 16-byte load per lane fetches the B operands of all four tiles
// (16 lanes x 16 B = 256 contiguous bytes of a weight row) and one 16-byte LDS write stores a lane's four outputs.
// Weights stream from L2 (1 MB per block).  H = 256, three hidden layers (the reference's NeuralNetwork, safe_set.py:26-43).
struct MlpWeights {
    const float* wf[SMPC_MAX_LAYERS];     // forward operands W^T [K][N] (layer 0: K padded to MLP_KPAD)
    const float* wb[SMPC_MAX_LAYERS];     // backward operands W [out][in] (layer 0: in padded to MLP_NPAD)
    const float* bias[SMPC_MAX_LAYERS];
};
constexpr int MLPF_ROWS = 16, MLPF_H = 256, MLPF_LD = MLPF_H + 4;

template <int NQ, bool BWD>
__global__ __launch_bounds__(256) void k_mlp_fused(const smpc_problem_desc* __restrict__ D, int M, int N, int mode, int act,
                                                   MlpWeights Wt, const double* __restrict__ xg, const double* __restrict__ p,
                                                   const int32_t* __restrict__ idx, const int32_t* __restrict__ m_live,
                                                   float* __restrict__ y_out, double* __restrict__ ev_out, int compact) {
    constexpr int H = MLPF_H, LD = MLPF_LD, R = MLPF_ROWS;
    __shared__ __attribute__((aligned(16))) float bufA[R * LD];
    __shared__ __attribute__((aligned(16))) float bufB[R * LD];
    __shared__ __attribute__((aligned(16))) float sfeat[R * MLP_KPAD];
    __shared__ float sy[R];
    __shared__ float sgs[4 * R * 16];
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, j = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * R;
    const int live = m_live ? *m_live : M;
    if (m0 >= live) return;                                  // (uniform: before the first barrier)
    // ---- features (safe_set.py:82-87), one thread per row; rows past the live count are zero
    if (t < R) {
        const int m = m0 + t;
        float* sf = sfeat + t * MLP_KPAD;
#pragma unroll
        for (int i = 0; i < MLP_KPAD; i++) sf[i] = 0.0f;
        if (m < live) {
            const double* x = xg + nn_row_to_node(mode, N, m, idx) * (2 * NQ);
            const int nd = D->nn_dof;
            double v[NQ], vn2 = 0.0;
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                v[i] = i < nd ? x[NQ + i] + (i == 0 ? D->nn_eps : 0.0) : 0.0;
                vn2 += v[i] * v[i];
            }
            const double inv = 1.0 / sqrt(vn2);
#pragma unroll
            for (int i = 0; i < NQ; i++)
                if (i < nd) {
                    sf[i] = (float)((x[i] - D->nn_mean[i]) / D->nn_std[i]);
                    sf[nd + i] = (float)(v[i] * inv);
                }
        }
    }
    __syncthreads();
    const int c0 = 64 * w + 4 * j;                           // this lane's four columns c0 .. c0 + 3 (one per tile)
    f32x4 acc[4];
    float dg[3][4][4];                                       // [hidden layer][tile][row register]
    auto zero_acc = [&]() {
#pragma unroll
        for (int tt = 0; tt < 4; tt++) acc[tt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    // one 16-deep K chunk: 4 MFMA steps x 4 tiles.  Step r pairs A[.][16 c + 4 kq + r] with B[16 c + 4 kq + r][.]
    auto chunk = [&](const f32x4 av, const f32x4* bv) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int tt = 0; tt < 4; tt++) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r][tt], acc[tt], 0, 0, 0);
        }
    };
    // acc += In[16 x K] * Bm[K x 256] over this wave's columns; In in LDS (row stride ld_in), Bm row-major with 256 columns
    auto gemm = [&](const float* In, int ld_in, const float* __restrict__ Bm, int K) {
        const float* arow = In + j * ld_in + 4 * kq;         // (A operand: row j, four consecutive k per lane)
        const float* brow = Bm + (size_t)(4 * kq) * H + c0;
        // the B operands of four chunks: one in use, three in flight (a chunk's 16 MFMAs last 512 cycles, an L2 round trip
        // under 1 024 wavefronts streaming the same weights rather longer)
        f32x4 bq[4][4];
        auto bload = [&](f32x4* bv, int c) {
#pragma unroll
            for (int r = 0; r < 4; r++) bv[r] = *reinterpret_cast<const f32x4*>(brow + (size_t)(16 * c + r) * H);
        };
        bload(bq[0], 0);
        if (K == 16) {
            chunk(*reinterpret_cast<const f32x4*>(arow), bq[0]);
            return;
        }
        const int nch = K >> 4;                              // (a multiple of 4: K is a multiple of 64)
        bload(bq[1], 1);
        bload(bq[2], 2);
        bload(bq[3], 3);
        // (sched_barrier: left alone, the scheduler sinks every load to just before its first use -- fewer live registers, and
        //  one load in flight instead of twelve)
        __builtin_amdgcn_sched_barrier(0);
        for (int c = 0; c < nch; c += 4) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                chunk(*reinterpret_cast<const f32x4*>(arow + 16 * (c + q)), bq[q]);
                __builtin_amdgcn_sched_barrier(0);
                if (c + q + 4 < nch) bload(bq[q], c + q + 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // C/D map of a 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + register
    auto store_rows = [&](float* Out, const float v[4][4]) {   // v[tile][reg] -> Out[4 kq + reg][c0 + tile]
#pragma unroll
        for (int r = 0; r < 4; r++)
            *reinterpret_cast<f32x4*>(Out + (4 * kq + r) * LD + c0) = f32x4{v[0][r], v[1][r], v[2][r], v[3][r]};
    };
    auto forward_epilogue = [&](auto ltag, float* Out) {
        constexpr int l = decltype(ltag)::value;
        const f32x4 bs = *reinterpret_cast<const f32x4*>(Wt.bias[l] + c0);
        float v[4][4];
#pragma unroll
        for (int tt = 0; tt < 4; tt++)
#pragma unroll
            for (int r = 0; r < 4; r++) v[tt][r] = act_f(act, acc[tt][r] + bs[tt], &dg[l][tt][r]);
        store_rows(Out, v);
    };
    // ---- forward: layer 0 (K = MLP_KPAD), layers 1, 2
    zero_acc();
    gemm(sfeat, MLP_KPAD, Wt.wf[0], MLP_KPAD);
    forward_epilogue(std::integral_constant<int, 0>{}, bufA);
    __syncthreads();
    zero_acc();
    gemm(bufA, LD, Wt.wf[1], H);
    forward_epilogue(std::integral_constant<int, 1>{}, bufB);
    __syncthreads();
    zero_acc();
    gemm(bufB, LD, Wt.wf[2], H);
    forward_epilogue(std::integral_constant<int, 2>{}, bufA);
    __syncthreads();
    // ---- output layer: y = a . w3 + b3 (16 threads per row, 16 columns each)
    {
        const int row = t >> 4, part = t & 15;
        const float* a = bufA + row * LD + 16 * part;
        const float* w3 = Wt.wb[3] + 16 * part;             // (the last layer's W is [1][H])
        float sacc = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; c++) sacc = fmaf(a[c], w3[c], sacc);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
        if (part == 0) {
            const float yv = sacc + Wt.bias[3][0];
            sy[row] = yv;
            if (y_out && m0 + row < live) y_out[m0 + row] = yv;
        }
    }
    if (!BWD) return;
    __syncthreads();                                         // (every wave has read bufA)
    // ---- backward: delta2 = w3 (.) act'(z2); delta1 = (delta2 W2) (.) act'(z1); delta0 = (delta1 W1) (.) act'(z0)
    {
        const f32x4 w3 = *reinterpret_cast<const f32x4*>(Wt.wb[3] + c0);
        float v[4][4];
#pragma unroll
        for (int tt = 0; tt < 4; tt++)
#pragma unroll
            for (int r = 0; r < 4; r++) v[tt][r] = w3[tt] * dg[2][tt][r];
        store_rows(bufB, v);
    }
    __syncthreads();
    auto backward_epilogue = [&](auto ltag, float* Out) {
        constexpr int l = decltype(ltag)::value;
        float v[4][4];
#pragma unroll
        for (int tt = 0; tt < 4; tt++)
#pragma unroll
            for (int r = 0; r < 4; r++) v[tt][r] = acc[tt][r] * dg[l][tt][r];
        store_rows(Out, v);
    };
    zero_acc();
    gemm(bufB, LD, Wt.wb[2], H);
    backward_epilogue(std::integral_constant<int, 1>{}, bufA);
    __syncthreads();
    zero_acc();
    gemm(bufA, LD, Wt.wb[1], H);
    backward_epilogue(std::integral_constant<int, 0>{}, bufB);
    __syncthreads();
    // ---- input gradient: GS[16 x 16] = delta0 [16 x 256] * W0 [256 x MLP_NPAD], columns 0..15; wave w takes k = 64 w .. 64 w + 63
    {
        f32x4 g = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float* arow = bufB + j * LD + 64 * w + 4 * kq;
        const float* bcol = Wt.wb[0] + (size_t)(64 * w + 4 * kq) * MLP_NPAD + j;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * c);
#pragma unroll
            for (int r = 0; r < 4; r++) g = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bcol[(size_t)(16 * c + r) * MLP_NPAD], g, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) sgs[(w * R + 4 * kq + r) * 16 + j] = g[r];
    }
    __syncthreads();
    // ---- chain rule back to the state (safe_set.py:82-94) and the per-node switch (utils.py:207-210): nn_val, nn_grad
    if (t < R && m0 + t < live) {
        const int m = m0 + t;
        const long node = nn_row_to_node(mode, N, m, idx);
        const double* x = xg + node * (2 * NQ);
        const double* pk = p + node * SMPC_NP;
        if (pk[4] > 0.0) {                                   // (switched off: the row sits mid-bounds, (0, 0) is left)
            double* const o = compact ? ev_out + (compact == 2 ? node / (N + 1) : node) * (1 + 2 * NQ) : ev_node(ev_out, node);      // (see k_nn_chain)
            const long es = compact ? 1 : EV_TILE;
            const int o_val = compact ? 0 : SMPC_EV_OFF(nn_val), o_grad = compact ? 1 : SMPC_EV_OFF(nn_grad);
            const int nd = D->nn_dof;
            float gs[MLP_KPAD];
#pragma unroll
            for (int i = 0; i < MLP_KPAD; i++) gs[i] = ((sgs[(0 * R + t) * 16 + i] + sgs[(1 * R + t) * 16 + i]) + sgs[(2 * R + t) * 16 + i]) + sgs[(3 * R + t) * 16 + i];
            double v[NQ], vn2 = 0.0, gdv = 0.0;
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                v[i] = i < nd ? x[NQ + i] + (i == 0 ? D->nn_eps : 0.0) : 0.0;
                vn2 += v[i] * v[i];
            }
            const double vn = sqrt(vn2);
#pragma unroll
            for (int i = 0; i < NQ; i++)
                if (i < nd) gdv += (double)gs[nd + i] * v[i];
            const double kap = (100.0 - pk[3]) / 100.0;
            o[o_val * es] = (double)sy[t] * kap - vn;
#pragma unroll
            for (int i = 0; i < NQ; i++)
                if (i < nd) {
                    o[(o_grad + i) * es] = kap * (double)gs[i] / D->nn_std[i];
                    o[(o_grad + NQ + i) * es] = kap * ((double)gs[nd + i] / vn - v[i] * gdv / (vn * vn * vn)) - v[i] / vn;
                }
        }
    }
}

// =========================================================================================================================
// k_mlp_wave: the whole network pass as ONE-WAVE blocks, for LARGE row counts (the safe-set row on every node: M = B * N rows,
// BASELINE config 4 and 'constraint_everywhere').  Round 5.  What it replaces there: the layer-by-layer
// GEMM chain, whose activations, derivatives and deltas make six round trips of M x 256 floats through HBM per pass (7.8 GB of
// the 36 GB a 5 461-instance sub-batch of config 4 moves per solve), and the four-wave k_mlp_fused above, which is 1.7 x faster
// than that chain alone on the GPU (1.32 vs 2.20 ms over 218 440 rows) but whose blocks -- four wavefronts and 38 KB of LDS at
// once on one CU -- wait behind resident QP wavefronts inside the closed loop (A/B in one session: C4 24.4 -> 23.9 ms per step,
// 'constraint_everywhere' on C1 3.96 -> 4.03).  A block here is ONE wavefront that owns 16 rows through the whole pass and fits
// the hole one retired QP wavefront leaves: < 256 registers, 16.7 KB of LDS.
//   * v_mfma_f32_16x16x4_f32, the tiling of k_mlp_fused (four 16-column tiles taken with stride 4, so that one 16-byte load per
//     lane fetches the B operands of all four): the wavefront walks the four 64-column groups of a 256-wide layer one after the
//     other; the A operand (the layer's input, 16 x 256) sits in ONE LDS buffer, the layer's output stays in 64 registers until
//     the last group has read the input, then overwrites it.
//   * the activation derivatives of the three hidden layers (16 x 256 each) go through the per-layer buffers the GEMM chain
//     already had ([M][256] floats, written in the forward half and read back by the same wavefront some 100 us later: L2 /
//     Infinity Cache traffic mostly, 3 KB per row instead of the chain's 18 KB);
//   * weights stream from L2, three 16-deep K chunks in flight (a chunk's 16 MFMAs last 512 cycles).
// Same arithmetic and summation order as k_mlp_fused; H = 256, three hidden layers (the reference's NeuralNetwork,
// safe_set.py:26-43).
template <int NQ, bool BWD>
__global__ __launch_bounds__(64) void k_mlp_wave(const smpc_problem_desc* __restrict__ D, int M, int N, int mode, int act,
                                                 MlpWeights Wt, const double* __restrict__ xg, const double* __restrict__ p,
                                                 const int32_t* __restrict__ idx, const int32_t* __restrict__ m_live,
                                                 float* __restrict__ y_out, double* __restrict__ ev_out, int compact,
                                                 float* __restrict__ dg0, float* __restrict__ dg1, float* __restrict__ dg2) {
    constexpr int H = MLPF_H, LD = MLPF_LD, R = MLPF_ROWS;
    __shared__ __attribute__((aligned(16))) float buf[R * LD];       // the current layer's input (features / activations / deltas)
    __shared__ float sy[R];
    const int lane = threadIdx.x, j = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * R;
    const int live = m_live ? *m_live : M;
    if (m0 >= live) return;
    auto wave_sync = [&]() {       // LDS hand-off inside one wavefront: its LDS operations execute in order, the compiler must not reorder them
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
    };
    // ---- features (safe_set.py:82-87), one lane per row, into the head of the buffer with row stride MLP_KPAD
    if (lane < R) {
        const int m = m0 + lane;
        float* sf = buf + lane * MLP_KPAD;
#pragma unroll
        for (int i = 0; i < MLP_KPAD; i++) sf[i] = 0.0f;
        if (m < live) {
            const double* x = xg + nn_row_to_node(mode, N, m, idx) * (2 * NQ);
            const int nd = D->nn_dof;
            double v[NQ], vn2 = 0.0;
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                v[i] = i < nd ? x[NQ + i] + (i == 0 ? D->nn_eps : 0.0) : 0.0;
                vn2 += v[i] * v[i];
            }
            const double inv = 1.0 / sqrt(vn2);
#pragma unroll
            for (int i = 0; i < NQ; i++)
                if (i < nd) {
                    sf[i] = (float)((x[i] - D->nn_mean[i]) / D->nn_std[i]);
                    sf[nd + i] = (float)(v[i] * inv);
                }
        }
    }
    wave_sync();
    f32x4 acc[4];
    float outH[4][4][4];                                     // [column group][tile][row register]: the layer's output
    auto zero_acc = [&]() {
#pragma unroll
        for (int tt = 0; tt < 4; tt++) acc[tt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    auto chunk = [&](const f32x4 av, const f32x4* bv) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int tt = 0; tt < 4; tt++) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r][tt], acc[tt], 0, 0, 0);
        }
    };
    // acc += In[16 x K] * Bm[K x 256] over the columns c0 .. c0 + 3 of this lane (one per tile); In in LDS, Bm row-major, 256 columns
    auto gemm = [&](int ld_in, const float* __restrict__ Bm, int K, int c0) {
        const float* arow = buf + j * ld_in + 4 * kq;
        const float* brow = Bm + (size_t)(4 * kq) * H + c0;
        f32x4 bq[4][4];
        auto bload = [&](f32x4* bv, int c) {
#pragma unroll
            for (int r = 0; r < 4; r++) bv[r] = *reinterpret_cast<const f32x4*>(brow + (size_t)(16 * c + r) * H);
        };
        bload(bq[0], 0);
        if (K == 16) {
            chunk(*reinterpret_cast<const f32x4*>(arow), bq[0]);
            return;
        }
        const int nch = K >> 4;
        bload(bq[1], 1);
        bload(bq[2], 2);
        bload(bq[3], 3);
        __builtin_amdgcn_sched_barrier(0);
        for (int c = 0; c < nch; c += 4) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                chunk(*reinterpret_cast<const f32x4*>(arow + 16 * (c + q)), bq[q]);
                __builtin_amdgcn_sched_barrier(0);
                if (c + q + 4 < nch) bload(bq[q], c + q + 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // C/D map of a 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + register.  A lane's four tiles of a group are the four
    // consecutive columns c0 .. c0 + 3: one 16-byte access per row.
    auto rows_to_lds = [&]() {                                  // outH -> buf[row][column], every group
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                *reinterpret_cast<f32x4*>(buf + (4 * kq + r) * LD + 64 * g + 4 * j) = f32x4{outH[g][0][r], outH[g][1][r], outH[g][2][r], outH[g][3][r]};
    };
    auto dg_ptr = [&](float* base, int g, int r) -> f32x4* {    // derivative of (row 4 kq + r, columns of group g) in its layer's buffer
        return reinterpret_cast<f32x4*>(base + (size_t)(m0 + 4 * kq + r) * H + 64 * g + 4 * j);
    };
    // ---- forward: three hidden layers
    auto forward_layer = [&](auto ltag, int ld_in, int K, float* dgl) {
        constexpr int l = decltype(ltag)::value;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int c0 = 64 * g + 4 * j;
            zero_acc();
            gemm(ld_in, Wt.wf[l], K, c0);
            const f32x4 bs = *reinterpret_cast<const f32x4*>(Wt.bias[l] + c0);
            float dv[4][4];
#pragma unroll
            for (int tt = 0; tt < 4; tt++)
#pragma unroll
                for (int r = 0; r < 4; r++) outH[g][tt][r] = act_f(act, acc[tt][r] + bs[tt], &dv[tt][r]);
            if (BWD) {
#pragma unroll
                for (int r = 0; r < 4; r++) *dg_ptr(dgl, g, r) = f32x4{dv[0][r], dv[1][r], dv[2][r], dv[3][r]};
            }
        }
        wave_sync();            // (every group has read the layer's input)
        rows_to_lds();
        wave_sync();
    };
    forward_layer(std::integral_constant<int, 0>{}, MLP_KPAD, MLP_KPAD, dg0);
    forward_layer(std::integral_constant<int, 1>{}, LD, H, dg1);
    forward_layer(std::integral_constant<int, 2>{}, LD, H, dg2);
    // ---- output layer: y = a . w3 + b3 (four lanes per row, 64 columns each)
    {
        const int row = lane >> 2, part = lane & 3;
        const float* a = buf + row * LD + 64 * part;
        const float* w3 = Wt.wb[3] + 64 * part;              // (the last layer's W is [1][H])
        float sacc = 0.0f;
#pragma unroll 8
        for (int c = 0; c < 64; c++) sacc = fmaf(a[c], w3[c], sacc);
        sacc += __shfl_xor(sacc, 1);
        sacc += __shfl_xor(sacc, 2);
        if (part == 0) {
            const float yv = sacc + Wt.bias[3][0];
            sy[row] = yv;
            if (y_out && m0 + row < live) y_out[m0 + row] = yv;
        }
    }
    if (!BWD) return;
    wave_sync();
    // ---- backward: delta2 = w3 (.) act'(z2); delta1 = (delta2 W2) (.) act'(z1); delta0 = (delta1 W1) (.) act'(z0)
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const f32x4 w3 = *reinterpret_cast<const f32x4*>(Wt.wb[3] + 64 * g + 4 * j);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const f32x4 dv = *dg_ptr(dg2, g, r);
#pragma unroll
            for (int tt = 0; tt < 4; tt++) outH[g][tt][r] = w3[tt] * dv[tt];
        }
    }
    rows_to_lds();
    wave_sync();
    auto backward_layer = [&](const float* __restrict__ Wb, float* dgl) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            f32x4 dv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) dv[r] = *dg_ptr(dgl, g, r);      // (issued before the product: it has 256 MFMAs to arrive in)
            zero_acc();
            gemm(LD, Wb, H, 64 * g + 4 * j);
#pragma unroll
            for (int tt = 0; tt < 4; tt++)
#pragma unroll
                for (int r = 0; r < 4; r++) outH[g][tt][r] = acc[tt][r] * dv[r][tt];
        }
        wave_sync();
        rows_to_lds();
        wave_sync();
    };
    backward_layer(Wt.wb[2], dg1);
    backward_layer(Wt.wb[1], dg0);
    // ---- input gradient: GS[16 x 16] = delta0 [16 x 256] * W0 [256 x MLP_NPAD], columns 0 .. 15
    f32x4 gsv = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    {
        const float* arow = buf + j * LD + 4 * kq;
        const float* bcol = Wt.wb[0] + (size_t)(4 * kq) * MLP_NPAD + j;
#pragma unroll 4
        for (int c = 0; c < H / 16; c++) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * c);
#pragma unroll
            for (int r = 0; r < 4; r++) gsv = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bcol[(size_t)(16 * c + r) * MLP_NPAD], gsv, 0, 0, 0);
        }
    }
    wave_sync();                                             // (the last product has read the buffer: its head now takes GS[16][16])
#pragma unroll
    for (int r = 0; r < 4; r++) buf[(4 * kq + r) * 16 + j] = gsv[r];
    wave_sync();
    // ---- chain rule back to the state (safe_set.py:82-94) and the per-node switch (utils.py:207-210): nn_val, nn_grad
    if (lane < R && m0 + lane < live) {
        const int m = m0 + lane;
        const long node = nn_row_to_node(mode, N, m, idx);
        const double* x = xg + node * (2 * NQ);
        const double* pk = p + node * SMPC_NP;
        if (pk[4] > 0.0) {                                   // (switched off: the row sits mid-bounds, (0, 0) is left)
            double* const o = compact ? ev_out + (compact == 2 ? node / (N + 1) : node) * (1 + 2 * NQ) : ev_node(ev_out, node);      // (see k_nn_chain)
            const long es = compact ? 1 : EV_TILE;
            const int o_val = compact ? 0 : SMPC_EV_OFF(nn_val), o_grad = compact ? 1 : SMPC_EV_OFF(nn_grad);
            const int nd = D->nn_dof;
            float gs[MLP_KPAD];
#pragma unroll
            for (int i = 0; i < MLP_KPAD; i++) gs[i] = buf[lane * 16 + i];
            double v[NQ], vn2 = 0.0, gdv = 0.0;
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                v[i] = i < nd ? x[NQ + i] + (i == 0 ? D->nn_eps : 0.0) : 0.0;
                vn2 += v[i] * v[i];
            }
            const double vn = sqrt(vn2);
#pragma unroll
            for (int i = 0; i < NQ; i++)
                if (i < nd) gdv += (double)gs[nd + i] * v[i];
            const double kap = (100.0 - pk[3]) / 100.0;
            o[o_val * es] = (double)sy[lane] * kap - vn;
#pragma unroll
            for (int i = 0; i < NQ; i++)
                if (i < nd) {
                    o[(o_grad + i) * es] = kap * (double)gs[i] / D->nn_std[i];
                    o[(o_grad + NQ + i) * es] = kap * ((double)gs[nd + i] / vn - v[i] * gdv / (vn * vn * vn)) - v[i] / vn;
                }
        }
    }
}

}  // namespace smpc
