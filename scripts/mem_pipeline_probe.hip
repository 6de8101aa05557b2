// How should the stage workspace of k_qp_ipm travel?  A model of ONE horizon sweep of the QP kernel with the arithmetic
// replaced by a dependent FP64 chain of selectable length: 2048 wavefronts (8 per CU), each half-wave walking its own
// 204 KB region stage by stage, with the loads of `DEPTH` stages in flight.  Per stage and half-wave:
//   NG pieces of 512 B by LDS-DMA (global_load_lds_dwordx4: no VGPR destination; ring of DEPTH slots in LDS)
//   NV pieces of 512 B into VGPRs (16 B per lane, as the kernel's "whole pieces")
//   NN narrow loads (8 B per lane, 18 useful lanes: the kernel's per-variable vectors)
//   SW wide stores (16 B per lane), SN narrow stores (8 B per lane)
// Prints microseconds per stage visit and GB/s at B = 4096 (whole batch resident) and B = 256 (a wave alone on its CU).
// Measurement tool, not part of the product.
//   build: hipcc --offload-arch=gfx950 -O3 -o mem_pipeline_probe scripts/mem_pipeline_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double dbl2 __attribute__((ext_vector_type(2)));
constexpr int STRIDE = 848, NST = 31;   // (848 doubles = 53 x 128 B: stage records start on a cache line)
__constant__ int c_nl = 18, c_off = 2;   // narrow stores: lanes per store, offset of the store area from a 128-byte boundary (doubles)

#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void st16_asm(void* p, dbl2 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st8_asm(void* p, double v) {
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// MODE 0: every load into VGPRs (compiler-counted).  MODE 1: NG pieces by LDS-DMA + manual vmcnt, no VGPR loads (NV = NN = 0),
// stores in asm so that the count is exact.  MODE 2: mixed -- LDS-DMA pieces issued BEFORE the stage's VGPR loads; the
// compiler's wait for those VGPR loads covers them (vmcnt retires in order).
template <int MODE, int DEPTH, int NG, int NV, int NN, int SW, int SN>
__global__ __launch_bounds__(64) void k_probe(double* __restrict__ ws, int B, int sweeps, int work, double* out, int alternate) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x, hl = lane & 31, half = lane >> 5;
    const int b = 2 * blockIdx.x + half;
    if (b >= B) return;
    double* base = ws + (size_t)b * STRIDE * NST;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_AS void*)smem);
    constexpr int SLOT_D = NG * 128;     // doubles per ring slot (both halves)
    dbl2 bw[DEPTH][NV > 0 ? NV : 1];
    double bn[DEPTH][NN > 0 ? NN : 1];
    double acc = 0.0;
    const int hz = hl < 18 ? hl : 17;
    for (int s = 0; s < sweeps; s++) {
        const bool fwd = (alternate & 1) ? (s & 1) : true;   // alternate = 0: every sweep forward, no turn-around reuse in L2 / Infinity Cache
        auto stage_of = [&](int i) { i = i < NST ? i : NST - 1; return fwd ? i : NST - 1 - i; };
        auto issue = [&](int slot, int i) {
            const double* p = base + (size_t)stage_of(i) * STRIDE;
            if constexpr (NG > 0) {
#pragma unroll
                for (int j = 0; j < NG; j++)
                    glds16(reinterpret_cast<const dbl2*>(p) + hl + 32 * j, lds0 + (unsigned)((slot * SLOT_D + j * 128) * 8));
            }
            if constexpr (NV > 0) {
#pragma unroll
                for (int j = 0; j < NV; j++) bw[slot][j] = (reinterpret_cast<const dbl2*>(p) + 32 * NG)[hl + 32 * j];
            }
            if constexpr (NN > 0) {
#pragma unroll
                for (int j = 0; j < NN; j++) bn[slot][j] = p[64 * (NG + NV) + 18 * j + hz];
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; d++) issue(d, d);
        if constexpr (MODE == 1) wait_vm<0>();
#pragma unroll 1
        for (int i0 = 0; i0 < NST; i0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                const int i = i0 + d;
                if (i < NST) {
                    double a = 0.0;
                    if constexpr (NV > 0) {
#pragma unroll
                        for (int j = 0; j < NV; j++) a += bw[d][j].x + bw[d][j].y;
                    }
                    if constexpr (NN > 0) {
#pragma unroll
                        for (int j = 0; j < NN; j++) a += bn[d][j];
                    }
                    if constexpr (NG > 0) {
                        if constexpr (MODE == 1) wait_vm<(DEPTH - 1) * NG + DEPTH * (SW + SN)>();
                        else asm volatile("" : "+v"(a) :: "memory");
#pragma unroll
                        for (int j = 0; j < NG; j++) {
                            const dbl2 v = *reinterpret_cast<const dbl2*>(smem + d * SLOT_D + j * 128 + 2 * lane);
                            a += v.x + v.y;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    issue(d, i + DEPTH);
#pragma unroll 1
                    for (int w = 0; w < work; w++) a = fma(a, 0.999999, 1e-9);
                    acc += a;
                    double* q = base + (size_t)stage_of(i) * STRIDE + 64 * (NG + NV) + 32 * ((18 * NN + 31) / 32);
                    if constexpr (MODE == 1) {
#pragma unroll
                        for (int j = 0; j < SW; j++) st16_asm(reinterpret_cast<dbl2*>(q) + hl + 32 * j, dbl2{a, acc});
#pragma unroll
                        for (int j = 0; j < SN; j++) st8_asm(q + 64 * SW + 18 * j + hz, a);
                    } else {
#pragma unroll
                        for (int j = 0; j < SW; j++) (reinterpret_cast<dbl2*>(q))[hl + 32 * j] = dbl2{a, acc};
#pragma unroll
                        for (int j = 0; j < SN; j++) q[64 * SW + c_off + ((alternate & 2) ? 36 * j + 2 * hz : c_nl * j + (hl < c_nl ? hl : c_nl - 1))] = a;   // bit 1: 8-byte stores 16 bytes apart (the kernel's c.z columns)
                    }
                }
            }
        }
        if constexpr (MODE != 0) wait_vm<0>();
    }
    if (acc == 1.2345e-300) out[0] = acc;
}

static int g_alternate = 1;
template <int MODE, int DEPTH, int NG, int NV, int NN, int SW, int SN> void run(double* ws, double* out, int work, const char* what) {
    static_assert(64 * (NG + NV) + 32 * ((18 * NN + 31) / 32) + 64 * SW + 2 + 36 * SN <= STRIDE, "stage record");
    const int sweeps = 16;
    size_t lds = (size_t)DEPTH * NG * 1024;
    if (lds < 19 * 1024) lds = 19 * 1024;   // never more than 8 wavefronts per CU, like k_qp_ipm
    auto kern = k_probe<MODE, DEPTH, NG, NV, NN, SW, SN>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int B : {4096, 2048, 256}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3((B + 1) / 2), dim3(64), lds, 0, ws, B, 2, work, out, g_alternate);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3((B + 1) / 2), dim3(64), lds, 0, ws, B, sweeps, work, out, g_alternate);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double rd = 512.0 * (NG + NV) + 144.0 * NN, wr = 512.0 * SW + 144.0 * SN;
        const double bytes = (double)B * NST * sweeps * (rd + wr);
        printf("alt %d %-34s mode %d depth %d work %4d  B %4d  LDS %5zu  read %4.0f B wr %4.0f B  vmem %2d  %6.3f us/stage  %7.1f GB/s\n", g_alternate, what,
               MODE, DEPTH, work, B, lds, rd, wr, NG + NV + NN + SW + SN, 1e3 * ms / (NST * sweeps), bytes / ms / 1e6);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    double *ws, *out;
    const int Bmax = 4096;
    hipMalloc(&ws, sizeof(double) * (size_t)Bmax * STRIDE * NST);
    hipMalloc(&out, 8);
    hipMemset(ws, 0, sizeof(double) * (size_t)Bmax * STRIDE * NST);
    g_alternate = 0;
    const int cfg[5][2] = {{18, 2}, {18, 0}, {20, 0}, {24, 0}, {32, 0}};
    for (int c = 0; c < 5; c++) {
        hipMemcpyToSymbol(HIP_SYMBOL(c_nl), &cfg[c][0], sizeof(int));
        hipMemcpyToSymbol(HIP_SYMBOL(c_off), &cfg[c][1], sizeof(int));
        printf("--- narrow stores: %d lanes (%d B) each, store area %d B past a 128-byte boundary\n", cfg[c][0], 8 * cfg[c][0], 8 * cfg[c][1]);
        for (int work : {0, 70}) {
            run<0, 1, 0, 6, 0, 0, 0>(ws, out, work, "vgpr 6w, no stores");
            run<0, 1, 0, 6, 0, 0, 6>(ws, out, work, "vgpr 6w, st 6n");
        }
    }
    return 0;
}
