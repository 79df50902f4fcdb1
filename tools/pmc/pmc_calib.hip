// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the
// solve kernels use (MI355X_MICROARCH.md §HBM: "FETCH_SIZE reports exactly 1/2 of the bytes
// of a wide coalesced streaming read (16 B/lane) ... other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").
// Streams a buffer far larger than the 256 MiB Infinity Cache with 8 B/lane and 16 B/lane loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void read8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc == 12345.678) out[threadIdx.x] = acc;
}
__global__ void read16(const double2* __restrict__ in, double* __restrict__ out, size_t n2) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { double2 v = in[i]; acc += v.x + v.y; }
    if (acc == 12345.678) out[threadIdx.x] = acc;
}
// 29-wide rows read the way ik_kernel reads its Jacobians: 32-lane groups, lanes 0..28 read
// consecutive doubles of a 232-byte row, rows of one instance are contiguous
__global__ void read_rows29(const double* __restrict__ in, double* __restrict__ out, size_t rows) {
    const int i = threadIdx.x & 31;
    double acc = 0;
    for (size_t r = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 5; r < rows; r += ((size_t)gridDim.x * blockDim.x) >> 5)
        if (i < 29) acc += in[r * 29 + i];
    if (acc == 12345.678) out[threadIdx.x] = acc;
}
__global__ void write8(double* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = 1.0;
}
int main() {
    const size_t n = (size_t)1 << 26;           // 64 Mi doubles = 512 MiB
    double *a, *o;
    if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&o, 4096) != hipSuccess) return 1;
    hipMemset(a, 0, n * 8);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        read8<<<2048, 256>>>(a, o, n);
        read16<<<2048, 256>>>((const double2*)a, o, n / 2);
        read_rows29<<<2048, 256>>>(a, o, n / 29);
        write8<<<2048, 256>>>(a, n);
    }
    hipDeviceSynchronize();
    std::printf("bytes per launch: read8 %zu read16 %zu read_rows29 %zu write8 %zu\n", n * 8, n * 8, (n / 29) * 29 * 8, n * 8);
    return 0;
}
