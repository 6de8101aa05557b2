// lat_probe.hip -- what ONE wavefront pays on gfx950 for the primitives k_qp_ipm_wg's sequential phases are chains of (round 6):
// dependent FP64 FMA, v_rsq_f64 + two Newton steps, an LDS hand-off (ds_write -> wave fence -> ds_read), a dependent
// v_mfma_f64_16x16x4_f64 accumulation, a v_readlane_b32 pair feeding an FMA, a dependent ds_read chain.
// Clocks are s_memtime ticks (100 MHz constant clock is NOT used: __builtin_readcyclecounter = shader clock on gfx9) per operation, lone
// wavefront on an idle chip.   hipcc --offload-arch=gfx950 -O3 -o lat_probe scripts/lat_probe.hip && ./lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ double rdlane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
constexpr int REP = 256;
__global__ __launch_bounds__(64) void k_probe(double* out, unsigned long long* clk, double seed) {
    __shared__ double lds[1024];
    const int l = threadIdx.x;
    lds[l] = seed + l; lds[64 + l] = 0.5; lds[128 + l] = (l * 7 + 3) % 64;
    __syncthreads();
    double a = seed + 1e-3 * l, b = 1.0 + 1e-9 * l, c = 1e-7;
    unsigned long long t0, t1;
    // 0: dependent FMA chain
    t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < REP; i++) a = fma(a, b, c);
    t1 = __builtin_readcyclecounter();
    if (l == 0) clk[0] = t1 - t0;
    // 1: 4 independent FMA chains (issue rate)
    double a1 = a, a2 = a + 1, a3 = a + 2, a4 = a + 3;
    t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < REP; i++) { a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); a4 = fma(a4, b, c); }
    t1 = __builtin_readcyclecounter();
    if (l == 0) clk[1] = t1 - t0;
    a = a1 + a2 + a3 + a4;
    // 2: rsq + two Newton steps, dependent
    double x = fabs(a) + 2.0;
    t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int i = 0; i < REP; i++) {
        double y = __builtin_amdgcn_rsq(x);
        const double hx = 0.5 * x;
        y = fma(fma(-hx * y, y, 0.5), y, y);
        y = fma(fma(-hx * y, y, 0.5), y, y);
        x = y + 2.0;
    }
    t1 = __builtin_readcyclecounter();
    if (l == 0) clk[2] = t1 - t0;
    a += x;
    // 3: LDS hand-off: write, wave fence, read another lane's cell
    double v = a;
    t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int i = 0; i < REP; i++) {
        lds[256 + l] = v;
        lds_fence();
        v = lds[256 + ((l + 1) & 63)] + 1.0;
        lds_fence();
    }
    t1 = __builtin_readcyclecounter();
    if (l == 0) clk[3] = t1 - t0;
    a += v;
    // 4: dependent MFMA f64 16x16x4 accumulation
    v4d d = {a, a, a, a};
    t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < REP; i++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, d, 0, 0, 0);
    t1 = __builtin_readcyclecounter();
    if (l == 0) clk[4] = t1 - t0;
    a = d[0] + d[1] + d[2] + d[3];
    // 5: MFMA whose operand depends on the previous result (VALU reads the accumulator, feeds A)
    double opa = b;
    t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int i = 0; i < REP; i++) {
        v4d z = {0, 0, 0, 0};
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(opa, opa, z, 0, 0, 0);
        opa = z[0] * 1e-3 + 1.0;
    }
    t1 = __builtin_readcyclecounter();
    if (l == 0) clk[5] = t1 - t0;
    a += opa;
    // 6: readlane pair -> FMA, dependent
    double r = a;
    t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < REP; i++) r = fma(r, 0.999, rdlane(r, 3));
    t1 = __builtin_readcyclecounter();
    if (l == 0) clk[6] = t1 - t0;
    a += r;
    // 7: dependent LDS read chain (pointer chase)
    int idx = l;
    t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < REP; i++) idx = (int)lds[128 + (idx & 63)];
    t1 = __builtin_readcyclecounter();
    if (l == 0) clk[7] = t1 - t0;
    a += idx;
    // 8: 12 independent LDS reads batched, then one wait (per batch)
    double acc = 0.0;
    t0 = __builtin_readcyclecounter();
#pragma unroll 2
    for (int i = 0; i < REP; i++) {
        double q[12];
#pragma unroll
        for (int j = 0; j < 12; j++) q[j] = lds[((l + j * 5 + i) & 63)];
        asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]), "+v"(q[9]), "+v"(q[10]), "+v"(q[11]));
#pragma unroll
        for (int j = 0; j < 12; j++) acc += q[j];
        lds_fence();
    }
    t1 = __builtin_readcyclecounter();
    if (l == 0) clk[8] = t1 - t0;
    a += acc;
    // accuracy of v_rsq_f64 and of one / two Newton steps on it (max relative error against 1 / sqrt over 64 x 256 arguments)
    {
        double e0 = 0.0, e1 = 0.0, e2 = 0.0;
        for (int i = 0; i < 256; i++) {
            const double xx = (1.0 + l * 0.0371 + i * 0.00113) * (i % 3 == 0 ? 1e-6 : (i % 3 == 1 ? 1.0 : 3.7e5));
            const double ref = 1.0 / sqrt(xx);
            double y = __builtin_amdgcn_rsq(xx);
            e0 = fmax(e0, fabs(y - ref) / ref);
            const double hx = 0.5 * xx;
            y = fma(fma(-hx * y, y, 0.5), y, y);
            e1 = fmax(e1, fabs(y - ref) / ref);
            y = fma(fma(-hx * y, y, 0.5), y, y);
            e2 = fmax(e2, fabs(y - ref) / ref);
        }
        for (int o = 32; o > 0; o >>= 1) { e0 = fmax(e0, __shfl_xor(e0, o)); e1 = fmax(e1, __shfl_xor(e1, o)); e2 = fmax(e2, __shfl_xor(e2, o)); }
        if (l == 0) { out[64] = e0; out[65] = e1; out[66] = e2; }
    }
    // 9: empty timer pair
    t0 = __builtin_readcyclecounter();
    t1 = __builtin_readcyclecounter();
    if (l == 0) clk[9] = t1 - t0;
    out[l] = a;
}
int main() {
    double* out; unsigned long long* clk;
    hipMalloc(&out, 72 * 8); hipMalloc(&clk, 16 * 8);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out, clk, 1.25);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"dependent v_fma_f64", "v_fma_f64, 4 independent chains (per FMA)", "rsq + 2 Newton steps + add (dependent)",
                           "LDS hand-off: write, fence, read, fence", "dependent v_mfma_f64_16x16x4 (accumulator chain)",
                           "v_mfma_f64_16x16x4 -> VALU -> operand of the next", "v_readlane x2 -> v_fma (dependent)", "dependent ds_read_b64 (pointer chase)",
                           "12 batched ds_read_b64 + 12 adds + fence", "timer pair"};
    const int per[] = {REP, 4 * REP, REP, REP, REP, REP, REP, REP, REP, 1};
    printf("# scripts/lat_probe.hip: one wavefront alone on the chip, shader clocks per operation (%d repetitions)\n", REP);
    for (int i = 0; i < 10; i++) printf("%-58s %8.1f clocks\n", names[i], (double)h[i] / per[i]);
    double acc3[3];
    hipMemcpy(acc3, out + 64, sizeof(acc3), hipMemcpyDeviceToHost);
    printf("v_rsq_f64 max relative error %.2e; after one Newton step %.2e; after two %.2e\n", acc3[0], acc3[1], acc3[2]);
    return 0;
}
