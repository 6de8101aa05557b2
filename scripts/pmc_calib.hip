// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths k_qp_ipm uses (MI355X_MICROARCH.md, section HBM:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Each kernel moves exactly BYTES bytes of a 1 GiB buffer (larger than the 256 MiB Infinity Cache), once.
//   build: hipcc --offload-arch=gfx950 -O3 -o pmc_calib scripts/pmc_calib.hip ; run under rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl2 __attribute__((ext_vector_type(2)));
// 8 bytes per lane, contiguous per wave (the row arrays of the QP workspace)
__global__ void k_read8(const double* __restrict__ p, size_t n, double* out) {
    double a = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a == 1.2345e-300) out[0] = a;
}
// 16 bytes per lane (the image / gain rows)
__global__ void k_read16(const dbl2* __restrict__ p, size_t n, double* out) {
    double a = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { dbl2 v = p[i]; a += v.x + v.y; }
    if (a == 1.2345e-300) out[0] = a;
}
// 208-byte pieces read by 26 of 32 lanes, 8 bytes per lane, pieces 5.9 KB apart (one row array per stage record)
__global__ void k_read_rows(const double* __restrict__ p, size_t n_rec, size_t rec, double* out) {
    double a = 0.0;
    const int hl = threadIdx.x & 31;
    for (size_t r = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) / 32; r < n_rec; r += (size_t)gridDim.x * blockDim.x / 32)
        a += p[r * rec + (hl < 26 ? hl : 25)];
    if (a == 1.2345e-300) out[0] = a;
}
__global__ void k_write8(double* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}
int main() {
    const size_t bytes = 1ull << 30, n = bytes / 8;
    double *p, *out;
    hipMalloc(&p, bytes); hipMalloc(&out, 8);
    hipMemset(p, 0, bytes);
    hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, 0, p, n, out);
    hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const dbl2*)p, n / 2, out);
    const size_t rec = 752, n_rec = n / rec;
    hipLaunchKernelGGL(k_read_rows, dim3(4096), dim3(256), 0, 0, p, n_rec, rec, out);
    hipLaunchKernelGGL(k_write8, dim3(4096), dim3(256), 0, 0, p, n);
    hipDeviceSynchronize();
    printf("k_read8 %zu bytes; k_read16 %zu bytes; k_read_rows %zu pieces x 208 B = %zu bytes (lines touched: 2-3 x 128 B each); k_write8 %zu bytes\n",
           bytes, bytes, n_rec, n_rec * 208, bytes);
    return 0;
}
