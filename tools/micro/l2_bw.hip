// Achievable L2 -> CU read bandwidth on this chip: every workgroup streams 16-byte loads over
// a buffer small enough to stay resident in each XCD's 4 MB L2 (default 2 MB), the way the
// convolution kernel's operand loads do.  Build: hipcc --offload-arch=gfx950 -O3 l2_bw.hip -o l2_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4 *gptr;

__global__ __launch_bounds__(256) void read_kernel(const float *buf, size_t quads, int reps, float *sink) {
    f32x4 acc = 0.f;
    // quads is a power of two (checked by the host): every index is masked into the buffer
    const size_t mask = quads - 1;
    // a workgroup streams 128 KB of consecutive addresses per repetition (more than a CU's L1),
    // starting at a block- and repetition-dependent offset
    for (int r = 0; r < reps; ++r) {
        size_t i = (((size_t)blockIdx.x * 37 + (size_t)r * 4099) * 8192 + threadIdx.x) & mask;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            acc += *(gptr)(buf + 4 * i);
            i = (i + 256) & mask;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 2;
    const int grid = argc > 2 ? atoi(argv[2]) : 2048;
    const int reps = argc > 3 ? atoi(argv[3]) : 64;
    const size_t bytes = mb << 20, quads = bytes / 16;
    if (mb == 0 || (mb & (mb - 1)) != 0 || grid <= 0 || reps <= 0) {
        fprintf(stderr, "buffer size must be a power of two (MB), grid and reps positive\n");
        return 2;
    }
    float *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    if (hipMemset(buf, 0, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return 1;
    if (hipEventCreate(&e1) != hipSuccess) return 1;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, buf, quads, reps, sink);
    if (hipEventRecord(e0) != hipSuccess) return 1;
    const int launches = 10;
    for (int w = 0; w < launches; ++w) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, buf, quads, reps, sink);
    if (hipEventRecord(e1) != hipSuccess) return 1;
    if (hipEventSynchronize(e1) != hipSuccess) return 1;
    float ms = 0;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return 1;
    const double total = (double)launches * grid * 256 * reps * 32 * 16;
    printf("buffer %zu MB, grid %d, %d x 32 loads per thread: %.2f TB/s (%.1f us per launch)\n", mb, grid, reps,
           total / (ms * 1e-3) / 1e12, 1e3 * ms / launches);
    return hipGetLastError() != hipSuccess;
}
