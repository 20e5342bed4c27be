// Vector-quantisation step that follows the PSO run in the VQ-VAE workflow: a codebook whose
// rows are seeded from final particle positions, nearest-code lookup and row gather.
//
//   distance[i][c] = (|code_c|^2 + |x_i|^2) - 2 * <x_i, code_c>        (fp32)
//   index[i]       = arg min_c distance[i][c]      (lowest c among equal distances)
//
// The work is a skinny GEMM (n rows x K codes x D) followed by a min-loc per row; at the
// sizes of the workflow (n <= a few thousand, K = 256, D = 100) it is latency/HBM bound, so
// it stays on the vector ALUs: the codebook is transposed once per call so that lanes read
// consecutive codes, inputs are staged in LDS and broadcast, and the min-loc reuses the
// wavefront-shuffle reduction of the swarm kernels.
#include "common.h"

#include <math.h>

#include <mutex>

namespace gdpso {
namespace {

constexpr int VQ_ROWS = 4;       // input rows per workgroup
constexpr int VQ_THREADS = 256;  // 4 wavefronts
constexpr int VQ_MAX_D = 4096;   // VQ_ROWS * VQ_MAX_D floats of LDS = 64 KiB
static_assert(VQ_ROWS == VQ_THREADS / 64, "one wavefront per row computes |x|^2");

// codebook [k][d] -> transposed [d][kp] (kp = k rounded up to 64) and squared norms [kp].
// One thread per code; the squared norm is an in-order fp32 sum like a row-wise reduction.
__global__ __launch_bounds__(256) void vq_prepare_kernel(const float *__restrict__ cb, int k, int d, int kp,
                                                         float *__restrict__ cbt, float *__restrict__ csq) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= kp) return;
    float s = 0.f;
    for (int j = 0; j < d; ++j) {
        const float v = c < k ? cb[(size_t)c * d + j] : 0.f;
        cbt[(size_t)j * kp + c] = v;
        s = fmaf(v, v, s);
    }
    csq[c] = s;
}

// NaN-aware "a at index ia beats b at index ib": a NaN distance wins over any number (the
// reference's torch.min propagates NaN), equal distances go to the lower index.
__device__ inline bool vq_better(float a, int ia, float b, int ib) {
    const bool na = a != a, nb = b != b;
    if (na || nb) return na && (!nb || ia < ib);
    return a < b || (a == b && ia < ib);
}

__global__ __launch_bounds__(VQ_THREADS) void vq_nearest_kernel(const float *__restrict__ x, int n, int d, int k, int kp,
                                                                const float *__restrict__ cbt, const float *__restrict__ csq,
                                                                const float *__restrict__ cb, long long *__restrict__ indices,
                                                                float *__restrict__ codes, float *__restrict__ best_dist) {
    extern __shared__ float smem[];
    float *s_x = smem;  // [VQ_ROWS][d]
    __shared__ float s_xsq[VQ_ROWS];
    __shared__ float s_val[VQ_ROWS][VQ_THREADS / 64];
    __shared__ int s_idx[VQ_ROWS][VQ_THREADS / 64];
    __shared__ int s_win[VQ_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * VQ_ROWS;
    const int rows = min(VQ_ROWS, n - row0);
    for (int i = tid; i < VQ_ROWS * d; i += VQ_THREADS) {
        const int r = i / d;
        s_x[i] = r < rows ? x[(size_t)(row0 + r) * d + (i - r * d)] : 0.f;
    }
    __syncthreads();
    // |x|^2: one wavefront per row
    {
        float s = 0.f;
        for (int j = lane; j < d; j += 64) s = fmaf(s_x[wave * d + j], s_x[wave * d + j], s);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) s_xsq[wave] = s;
    }
    __syncthreads();
    float bv[VQ_ROWS];
    int bi[VQ_ROWS];
#pragma unroll
    for (int r = 0; r < VQ_ROWS; ++r) { bv[r] = INFINITY; bi[r] = 0x7fffffff; }
    for (int c = tid; c < kp; c += VQ_THREADS) {
        float acc[VQ_ROWS];
#pragma unroll
        for (int r = 0; r < VQ_ROWS; ++r) acc[r] = 0.f;
        const float *col = cbt + c;
        for (int j = 0; j < d; ++j) {
            const float w = col[(size_t)j * kp];
#pragma unroll
            for (int r = 0; r < VQ_ROWS; ++r) acc[r] = fmaf(s_x[r * d + j], w, acc[r]);
        }
        if (c < k) {
            const float cs = csq[c];
#pragma unroll
            for (int r = 0; r < VQ_ROWS; ++r) {
                const float dist = fmaf(-2.f, acc[r], cs + s_xsq[r]);
                if (vq_better(dist, c, bv[r], bi[r])) { bv[r] = dist; bi[r] = c; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < VQ_ROWS; ++r) {
        float v = bv[r];
        int i = bi[r];
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(i, o, 64);
            if (vq_better(ov, oi, v, i)) { v = ov; i = oi; }
        }
        if (lane == 0) { s_val[r][wave] = v; s_idx[r][wave] = i; }
    }
    __syncthreads();
    if (tid < rows) {
        float v = s_val[tid][0];
        int i = s_idx[tid][0];
        for (int w = 1; w < VQ_THREADS / 64; ++w)
            if (vq_better(s_val[tid][w], s_idx[tid][w], v, i)) { v = s_val[tid][w]; i = s_idx[tid][w]; }
        s_win[tid] = i;
        indices[row0 + tid] = i;
        if (best_dist) best_dist[row0 + tid] = v;
    }
    if (codes) {
        __syncthreads();
        for (int e = tid; e < rows * d; e += VQ_THREADS) {
            const int r = e / d, j = e - r * d;
            codes[(size_t)(row0 + r) * d + j] = cb[(size_t)s_win[r] * d + j];
        }
    }
}

// codes[i] = codebook[indices[i]]; an index outside [0, k) yields a row of NaN (the host side
// of the reference interface checks the range first and raises like torch's embedding).
__global__ __launch_bounds__(256) void vq_lookup_kernel(const float *__restrict__ cb, int k, int d,
                                                        const long long *__restrict__ indices, long total,
                                                        float *__restrict__ codes) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long r = e / d;
    const int j = (int)(e - r * d);
    const long long c = indices[r];
    codes[e] = (c >= 0 && c < k) ? cb[(size_t)c * d + j] : NAN;
}

// codebook[row0 + i][j] = (float) pos[i][j]
__global__ __launch_bounds__(256) void vq_seed_kernel(float *__restrict__ cb, int d, int row0, const double *__restrict__ pos,
                                                      long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    cb[(size_t)row0 * d + e] = (float)pos[e];
}

inline int round_up64(int k) { return (k + 63) / 64 * 64; }
inline long cdiv(long a, long b) { return (a + b - 1) / b; }

}  // namespace
}  // namespace gdpso

using namespace gdpso;

extern "C" size_t gdpso_vq_workspace_bytes(int k, int d) {
    if (k <= 0 || d <= 0) return 0;
    return ((size_t)d + 1) * round_up64(k) * sizeof(float);
}

extern "C" int gdpso_vq_nearest(void *stream_, int n, int d, int k, const float *inputs, const float *codebook,
                                long long *indices, float *codes, float *best_dist, void *workspace, size_t workspace_bytes) {
    GDPSO_REQUIRE(n >= 0 && k > 0 && d > 0, "vq_nearest: n >= 0, k > 0 and d > 0 are required");
    GDPSO_REQUIRE(n == 0 || (inputs && codebook && indices && workspace), "vq_nearest: null argument");
    GDPSO_REQUIRE(d <= VQ_MAX_D, "vq_nearest: embedding size above 4096");
    if (workspace_bytes < gdpso_vq_workspace_bytes(k, d)) return fail(GDPSO_ERR_WORKSPACE, "vq_nearest: workspace too small");
    if (n == 0) return GDPSO_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const int kp = round_up64(k);
    float *cbt = (float *)workspace, *csq = cbt + (size_t)d * kp;
    hipLaunchKernelGGL(vq_prepare_kernel, dim3(cdiv(kp, 256)), dim3(256), 0, stream, codebook, k, d, kp, cbt, csq);
    hipLaunchKernelGGL(vq_nearest_kernel, dim3(cdiv(n, VQ_ROWS)), dim3(VQ_THREADS), (size_t)VQ_ROWS * d * sizeof(float), stream,
                       inputs, n, d, k, kp, cbt, csq, codebook, indices, codes, best_dist);
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

// ---------------------------------------------------------------------------------
// The same search for callers that cannot allocate (HIP graph capture, tight serving loops):
// the transposed codebook and the squared norms go to a file-scope device array instead of a
// caller workspace.  Every GPU the library runs on holds its own copy of the array (1 MiB) at
// its own address until the process ends; in host code the array's NAME is not a device address,
// so the address is asked for once per device (hipGetSymbolAddress, with that device current) and
// only that pointer is handed to kernels.
namespace gdpso {
namespace {
constexpr size_t VQ_STATIC_FLOATS = GDPSO_VQ_STATIC_MAX_BYTES / sizeof(float);  // (d + 1) * round_up64(k) floats at most
static __device__ __align__(256) float scratch[VQ_STATIC_FLOATS];

constexpr int VQ_MAX_DEVICES = 64;
float *g_scratch_ptr[VQ_MAX_DEVICES] = {};
std::mutex g_scratch_mutex;

// Device address of `scratch` on the current device (looked up on the first call per device).
int vq_static_scratch(float **out) {
    int dev = -1;
    GDPSO_HIP(hipGetDevice(&dev));
    GDPSO_REQUIRE(dev >= 0 && dev < VQ_MAX_DEVICES, "vq_nearest_static: device ordinal above 63");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        GDPSO_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        GDPSO_REQUIRE(p != nullptr, "vq_nearest_static: no device address for the scratch array");
        g_scratch_ptr[dev] = (float *)p;
    }
    *out = g_scratch_ptr[dev];
    return GDPSO_OK;
}
}  // namespace
}  // namespace gdpso

extern "C" int gdpso_vq_nearest_static(void *stream_, int n, int d, int k, const float *inputs, const float *codebook,
                                       long long *indices, float *codes, float *best_dist) {
    GDPSO_REQUIRE(n >= 0 && k > 0 && d > 0, "vq_nearest_static: n >= 0, k > 0 and d > 0 are required");
    GDPSO_REQUIRE(n == 0 || (inputs && codebook && indices), "vq_nearest_static: null argument");
    GDPSO_REQUIRE(d <= VQ_MAX_D, "vq_nearest_static: embedding size above 4096");
    if (gdpso_vq_workspace_bytes(k, d) > VQ_STATIC_FLOATS * sizeof(float))
        return fail(GDPSO_ERR_WORKSPACE, "vq_nearest_static: (d + 1) * round_up(k, 64) floats exceed the 1 MiB static scratch; "
                                         "use gdpso_vq_nearest with a caller workspace");
    if (n == 0) return GDPSO_OK;
    float *ws = nullptr;
    int rc = vq_static_scratch(&ws);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const int kp = round_up64(k);
    // both kernels take the looked-up pointer; the prepare kernel rewrites every word the search
    // reads (d * kp transposed codes, kp norms), so nothing of an earlier call's contents survives
    float *cbt = ws, *csq = cbt + (size_t)d * kp;
    hipLaunchKernelGGL(vq_prepare_kernel, dim3(cdiv(kp, 256)), dim3(256), 0, stream, codebook, k, d, kp, cbt, csq);
    hipLaunchKernelGGL(vq_nearest_kernel, dim3(cdiv(n, VQ_ROWS)), dim3(VQ_THREADS), (size_t)VQ_ROWS * d * sizeof(float), stream,
                       inputs, n, d, k, kp, cbt, csq, codebook, indices, codes, best_dist);
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

extern "C" int gdpso_vq_lookup(void *stream_, int n, int d, int k, const float *codebook, const long long *indices, float *codes) {
    GDPSO_REQUIRE(n >= 0 && k > 0 && d > 0, "vq_lookup: n >= 0, k > 0 and d > 0 are required");
    GDPSO_REQUIRE(n == 0 || (codebook && indices && codes), "vq_lookup: null argument");
    if (n == 0) return GDPSO_OK;
    const long total = (long)n * d;
    hipLaunchKernelGGL(vq_lookup_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream_, codebook, k, d, indices, total,
                       codes);
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

extern "C" int gdpso_vq_seed_codebook(void *stream_, float *codebook, int k, int d, int row0, const double *positions, int n) {
    GDPSO_REQUIRE(k > 0 && d > 0 && n >= 0 && row0 >= 0, "vq_seed_codebook: bad size");
    GDPSO_REQUIRE(n == 0 || (codebook && positions), "vq_seed_codebook: null argument");
    GDPSO_REQUIRE((long)row0 + n <= k, "vq_seed_codebook: rows beyond the codebook");
    if (n == 0) return GDPSO_OK;
    const long total = (long)n * d;
    hipLaunchKernelGGL(vq_seed_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream_, codebook, d, row0, positions, total);
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}
