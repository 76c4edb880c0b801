// hbm_calib.hip -- calibration kernels for rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950.
// MI355X_MICROARCH.md (HBM): FETCH_SIZE reads exactly half the bytes of a 16-B-per-lane streaming read and
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".
// The integrator's state traffic is 8 B per lane, SoA-coalesced (512 B per wave instruction), so this
// program streams a known number of bytes with exactly that pattern: each lane loads NCOMP doubles at stride B
// and stores NCOMP doubles at stride B, like lane_load / lane_store in rk_core.h.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int NCOMP = 16;

__global__ __launch_bounds__(64) void calib_soa_rw8(const double *__restrict__ in, double *__restrict__ out, size_t B)
{
    const size_t j = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= B) return;
    double v[NCOMP];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) v[c] = in[c * B + j];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) out[c * B + j] = v[c] * 1.0000001 + 1.0;
}

int main(int argc, char **argv)
{
    const size_t B = argc > 1 ? strtoull(argv[1], nullptr, 10) : (size_t)(1u << 24);  // 16M lanes x 16 x 8 B = 2 GiB each way
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    double *in, *out;
    const size_t bytes = B * NCOMP * sizeof(double);
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 0x11, bytes);
    hipMemset(out, 0, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(calib_soa_rw8, dim3((B + 63) / 64), dim3(64), 0, 0, in, out, B);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("calib_soa_rw8: read %zu bytes, wrote %zu bytes in %.3f ms = %.1f GB/s (r+w)\n", bytes, bytes, ms, 2.0 * bytes / ms / 1e6);
    }
    return 0;
}
