// What bandwidth can the QP kernel's ACCESS PATTERN reach at its occupancy, with no arithmetic at all?
// 2048 wavefronts (8 per CU, as k_qp_ipm), each half-wave walking its own 204 KB region stage by stage (31 stages of 6.7 KB),
// reading ~3.3 KB per stage as 16-byte-per-lane pieces of 512 B and writing ~0.9 KB, `DEPTH` stages of loads in flight; the
// sweep direction alternates like the kernel's.  Prints GB/s per depth.  Not part of the product (measurement tool).
//   build: hipcc --offload-arch=gfx950 -O3 -o hbm_pattern scripts/hbm_pattern.hip
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <vector>
typedef double dbl2 __attribute__((ext_vector_type(2)));
constexpr int STRIDE = 842, NST = 31;            // doubles per stage record, stages per instance (Z1, N = 30)
constexpr int RD_PIECES = 7, WR_PIECES = 2;      // 512-byte pieces read / written per stage visit and half-wave (3.5 KB / 1 KB)

template <int DEPTH>
__global__ __launch_bounds__(64) void k_walk(double* __restrict__ ws, int B, int sweeps, double* out) {
    const int hl = threadIdx.x & 31, half = threadIdx.x >> 5;
    const int b = 2 * blockIdx.x + half;
    if (b >= B) return;
    double* base = ws + (size_t)b * STRIDE * NST;
    dbl2 buf[DEPTH][RD_PIECES];
    double acc = 0.0;
    for (int s = 0; s < sweeps; s++) {
        const bool fwd = s & 1;
        auto stage = [&](int i) { return fwd ? i : NST - 1 - i; };
        auto load = [&](int slot, int i) {
            const dbl2* p = reinterpret_cast<const dbl2*>(base + (size_t)stage(i < NST ? i : NST - 1) * STRIDE);
#pragma unroll
            for (int j = 0; j < RD_PIECES; j++) buf[slot][j] = p[hl + 32 * j];
        };
#pragma unroll
        for (int d = 0; d < DEPTH; d++) load(d, d);
#pragma unroll 1
        for (int i0 = 0; i0 < NST; i0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                const int i = i0 + d;
                if (i < NST) {
                    double a = 0.0;
#pragma unroll
                    for (int j = 0; j < RD_PIECES; j++) a += buf[d][j].x + buf[d][j].y;
                    acc += a;
                    dbl2* q = reinterpret_cast<dbl2*>(base + (size_t)stage(i) * STRIDE + 2 * 32 * RD_PIECES);
#pragma unroll
                    for (int j = 0; j < WR_PIECES; j++) q[hl + 32 * j] = dbl2{a, acc};
                    load(d, i + DEPTH);
                }
            }
        }
    }
    if (acc == 1.2345e-300) out[0] = acc;
}

template <int DEPTH> void run(double* ws, int B, double* out) {
    const int sweeps = 20;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_walk<DEPTH>, dim3((B + 1) / 2), dim3(64), 0, 0, ws, B, 2, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_walk<DEPTH>, dim3((B + 1) / 2), dim3(64), 0, 0, ws, B, sweeps, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)B * NST * sweeps * (RD_PIECES + WR_PIECES) * 512.0;
    printf("B %5d  loads in flight: %d stage(s) = %4.1f KB per wave   %7.3f ms   %7.1f GB/s (read %4.0f %%)\n", B, DEPTH,
           2.0 * DEPTH * RD_PIECES * 0.5, ms, bytes / ms / 1e6, 100.0 * RD_PIECES / (RD_PIECES + WR_PIECES));
}

int main() {
    const int Bmax = 8192;
    double *ws, *out;
    hipMalloc(&ws, sizeof(double) * (size_t)Bmax * STRIDE * NST);
    hipMalloc(&out, 8);
    hipMemset(ws, 0, sizeof(double) * (size_t)Bmax * STRIDE * NST);
    for (int B : {4096, 8192, 2048}) {
        run<1>(ws, B, out);
        run<2>(ws, B, out);
        run<3>(ws, B, out);
        run<4>(ws, B, out);
    }
    return 0;
}
