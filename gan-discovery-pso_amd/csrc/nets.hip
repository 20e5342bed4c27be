// Generator, assessor and encoder forward passes: the non-GEMM kernels (Cin = 1 or
// Cout = 1 layers, pooling, per-image reductions, classifier + fitness) and the plans
// that chain them with the implicit-GEMM convolution kernel.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

#include "common.h"

namespace gdpso {

// ---------------------------------------------------------------------------------
// Latents: float64 (or float32) positions -> float32 rows zero-padded to the GEMM's K.
// The float64 -> float32 cast is the one at reference src/pso/util_discovery.py:41-43.
template <typename E>
__global__ __launch_bounds__(256) void latent_prep_kernel(const void *__restrict__ z, int z_is_f64, int n, int zd,
                                                          int kpad, E *__restrict__ out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)n * kpad) return;
    const int row = (int)(e / kpad), col = (int)(e - (long)row * kpad);
    float v = 0.f;
    if (col < zd) v = z_is_f64 ? (float)((const double *)z)[(long)row * zd + col] : ((const float *)z)[(long)row * zd + col];
    out[e] = (E)v;
}

// First generator layer for short latents (z_dim <= GF_MAX_Z): ConvTranspose2d(Z -> 2F, k, s1, p0) on a 1x1
// input + BatchNorm + ReLU (reference src/utils/util_dcgan.py:135,143-145) straight from the swarm's float64
// positions.  With Z = 10 the layer is 10 multiply-adds per output: as a GEMM it needs its K padded to a
// K-tile of 32 (two thirds of the matrix work on zeros) behind a kernel of its own that casts and pads the
// latents.  Here a thread holds the Z weights of four consecutive outputs (position-major columns (ky, kx, co),
// so the row it writes is the NHWC map) in registers and walks a few images; the latents of those images
// come through LDS.  The sum runs over k in the order in which the matrix-core path adds the products of a
// 32-wide K-tile (pairs (8 g + s, 8 g + 4 + s)), each step one fused multiply-add, so both paths round alike.
constexpr int GF_MAX_Z = 16, GF_IMAGES = 4;
template <typename E>
__global__ __launch_bounds__(256) void gen_first_kernel(const void *__restrict__ z, int z_is_f64, int n, int zd, int ncol,
                                                        const float *__restrict__ wt, const float *__restrict__ scale,
                                                        const float *__restrict__ shift, E *__restrict__ out) {
    __shared__ float s_z[GF_IMAGES * GF_MAX_Z];
    const int tid = threadIdx.x, img0 = blockIdx.y * GF_IMAGES;
    if (tid < GF_IMAGES * GF_MAX_Z) {
        const int i = tid / GF_MAX_Z, k = tid % GF_MAX_Z, img = img0 + i;
        float v = 0.f;
        if (img < n && k < zd) v = z_is_f64 ? (float)((const double *)z)[(long)img * zd + k] : ((const float *)z)[(long)img * zd + k];
        s_z[tid] = (float)(E)v;  // (16-bit storage: the latent rounded as the matrix-core path rounds its A operand)
    }
    __syncthreads();
    const int col = (blockIdx.x * 256 + tid) * 4;
    if (col >= ncol) return;
    f32x4 w[GF_MAX_Z];
#pragma unroll
    for (int k = 0; k < GF_MAX_Z; ++k) {
        w[k] = 0.f;
        if (k < zd) w[k] = *(const f32x4 *)(wt + (size_t)k * ncol + col);
    }
    const f32x4 sc = *(const f32x4 *)(scale + col), sh = *(const f32x4 *)(shift + col);
#pragma unroll
    for (int i = 0; i < GF_IMAGES; ++i) {
        const int img = img0 + i;
        if (img >= n) break;
        f32x4 acc = 0.f;
#pragma unroll
        for (int j = 0; j < GF_MAX_Z; ++j) {
            const int g = j >> 3, h = j & 1, st = (j >> 1) & 3, k = 8 * g + 4 * h + st;  // 0, 4, 1, 5, 2, 6, 3, 7, 8, 12, ...
            if (k < zd) {
                const float zk = s_z[i * GF_MAX_Z + k];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(zk, w[k][q], acc[q]);
            }
        }
        f32x4 v = acc * sc + sh;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        store4<E>(out + (size_t)img * ncol + col, v);
    }
}

// ---------------------------------------------------------------------------------
// Last generator layer, fused: ConvTranspose2d(F -> 1, k4, s2, p1) + bias + tanh
// (reference src/utils/util_dcgan.py:137-138), per-image min/max rescale to [0,1]
// (src/utils/util_data.py:129-135 on a batch of one) and, optionally, the pixel MSE
// against a target image (src/pso/util_discovery.py:97).
// One workgroup per image.  The NHWC input image is staged in LDS once (when it fits);
// the four output phases (oy&1, ox&1) are processed one after the other so that the
// 2x2 kernel taps of a phase are wave-uniform and live in registers.  16 lanes share one
// output pixel (4 channels each, one 16-byte read per tap) and reduce by shuffles.
// E: element type of the incoming activations (the arithmetic is fp32 whatever it is).
template <bool STAGED, typename E>
__global__ __launch_bounds__(512) void gen_last_kernel(const E *__restrict__ a2, const float *__restrict__ w5,
                                                       float bias, int hin, int cin, float *__restrict__ img,
                                                       float *__restrict__ img01, const float *__restrict__ target,
                                                       float *__restrict__ mse) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int side = 2 * hin, npix = side * side, nin = hin * hin;
    float *s_img = smem;            // [npix]
    float *s_red = s_img + npix;    // [3][8]: min, max, squared error of the 8 waves
    float *s_a2 = s_red + 32;       // [nin][cin] when STAGED
    const int tid = threadIdx.x, n = blockIdx.x;
    const E *src = a2 + (size_t)n * nin * cin;
    if (STAGED) {
        for (int i = tid * 4; i < nin * cin; i += 2048) *(f32x4 *)&s_a2[i] = load4<E>(&src[i]);
        __syncthreads();
    }
    // four channels of one input pixel: from the staged fp32 copy or straight from global memory
    auto act4 = [&](size_t idx) -> f32x4 {
        if constexpr (STAGED)
            return *(const f32x4 *)&s_a2[idx];
        else
            return load4<E>(src + idx);
    };
    const int grp = tid >> 4, sub = tid & 15;
    for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
        // taps of this phase: (input offset, kernel index) = (0, 1), (-1, 3) for even rows; (0, 2), (+1, 0) for odd
        const int dy1 = py ? 1 : -1, ky0 = py ? 2 : 1, ky1 = py ? 0 : 3;
        const int dx1 = px ? 1 : -1, kx0 = px ? 2 : 1, kx1 = px ? 0 : 3;
        for (int c0 = 0; c0 < cin; c0 += 64) {
            const int c = c0 + sub * 4;
            const bool cok = c < cin;
            f32x4 w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
            if (cok) {
                w00 = *(const f32x4 *)&w5[(ky0 * 4 + kx0) * cin + c];
                w01 = *(const f32x4 *)&w5[(ky0 * 4 + kx1) * cin + c];
                w10 = *(const f32x4 *)&w5[(ky1 * 4 + kx0) * cin + c];
                w11 = *(const f32x4 *)&w5[(ky1 * 4 + kx1) * cin + c];
            }
            for (int q0 = 0; q0 < nin; q0 += 32) {
                const int q = q0 + grp;
                float acc = 0.f;
                if (q < nin && cok) {
                    const int my = q / hin, mx = q - my * hin;
                    const int iy1 = my + dy1, ix1 = mx + dx1;
                    const bool yok = (unsigned)iy1 < (unsigned)hin, xok = (unsigned)ix1 < (unsigned)hin;
                    const f32x4 z = 0.f;
                    const f32x4 v00 = act4((size_t)(my * hin + mx) * cin + c);
                    const f32x4 v01 = xok ? act4((size_t)(my * hin + ix1) * cin + c) : z;
                    const f32x4 v10 = yok ? act4((size_t)(iy1 * hin + mx) * cin + c) : z;
                    const f32x4 v11 = (yok && xok) ? act4((size_t)(iy1 * hin + ix1) * cin + c) : z;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc = fmaf(v00[k], w00[k], acc);
                        acc = fmaf(v01[k], w01[k], acc);
                        acc = fmaf(v10[k], w10[k], acc);
                        acc = fmaf(v11[k], w11[k], acc);
                    }
                }
                acc += __shfl_xor(acc, 8, 16);
                acc += __shfl_xor(acc, 4, 16);
                acc += __shfl_xor(acc, 2, 16);
                acc += __shfl_xor(acc, 1, 16);
                if (sub == 0 && q < nin) {
                    const int my = q / hin, mx = q - my * hin;
                    const int o = (2 * my + py) * side + 2 * mx + px;
                    s_img[o] = c0 == 0 ? acc : s_img[o] + acc;
                }
            }
        }
    }
    __syncthreads();
    // bias + tanh, then the per-image min / max: wavefront shuffle reduction, then across the 4 waves
    float mn = INFINITY, mx = -INFINITY, se = 0.f;
    float *dst = img ? img + (size_t)n * npix : nullptr;
    const float *tg = target ? target + (size_t)n * npix : nullptr;
    for (int i = tid; i < npix; i += 512) {
        const float v = tanhf(s_img[i] + bias);
        s_img[i] = v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
        if (dst) dst[i] = v;
        if (tg) {
            const float dlt = tg[i] - v;
            se = fmaf(dlt, dlt, se);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
        se += __shfl_xor(se, o);
    }
    if ((tid & 63) == 0) {
        s_red[tid >> 6] = mn;
        s_red[8 + (tid >> 6)] = mx;
        s_red[16 + (tid >> 6)] = se;
    }
    __syncthreads();
    mn = s_red[0];
    mx = s_red[8];
    se = s_red[16];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        mn = fminf(mn, s_red[k]);
        mx = fmaxf(mx, s_red[8 + k]);
        se += s_red[16 + k];
    }
    const float range = mx - mn;
    float *d01 = img01 + (size_t)n * npix;
    for (int i = tid; i < npix; i += 512) d01[i] = (s_img[i] - mn) / range;  // each thread rescales the pixels it wrote
    if (mse && tid == 0) mse[n] = se / (float)npix;
}

// ---------------------------------------------------------------------------------
// The same layer on the matrix cores.  Per image, C[input pixel][kernel position (ky,kx)] =
// A[pixel][channel] * W[channel][(ky,kx)] is a 196 x 16 x F product (v_mfma_f32_16x16x4_f32,
// exact fp32); an output pixel is then the sum of the (at most) four entries of C whose input
// pixel and kernel position land on it.  A lives in LDS with padded rows (conflict-free 16-byte
// fragment reads); lane l of a wave holds A[row l&15][k] and W[k][col l&15] for the four
// k = 16j + 4(l>>4) + s of step j, so one 16-byte read feeds four MFMAs.  Needs cin % 16 == 0.
using f32x4_t = __attribute__((ext_vector_type(4))) float;
constexpr int GL_MAX_KSTEPS = 8;  // cin <= 128

template <typename E>
__global__ __launch_bounds__(512) void gen_last_mfma_kernel(const E *__restrict__ a2, const float *__restrict__ w5,
                                                            float bias, int hin, int cin, int band, float *__restrict__ img,
                                                            float *__restrict__ img01, const float *__restrict__ target,
                                                            float *__restrict__ mse, FastDiv div_c4, FastDiv div_side) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int side = 2 * hin, npix = side * side;
    // The input map is walked in bands of `band` rows (all of it when it fits: the 14x14 map of the reference's
    // generator; the 32x32 map of the 64x64 one in bands): a band's rows plus one row either side are staged,
    // multiplied, and the output rows 2 r0 .. 2 r1 - 1 of the band gathered into the image.
    const int brows = band >= hin ? hin : band + 2, nblk = (brows * hin + 15) / 16, lda = cin + 4;
    // n / d by multiply-shift (FastDiv): the chip has no integer divide
    auto fdiv = [](int v, FastDiv f) { return (int)(((unsigned long long)(unsigned)v * f.mul) >> f.shift); };
    float *s_img = smem;                    // [npix]
    float *s_red = s_img + npix;            // [3][8]
    float *s_c = s_red + 32;                // [nblk * 16][16]
    float *s_a2 = s_c + nblk * 16 * 16;     // [nblk * 16][lda]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.x;
    const E *src = a2 + (size_t)n * hin * hin * cin;
    const int c4 = cin / 4;
    // W fragments: column l&15 is the kernel position ky*4+kx, stored as w5[position][channel]
    const int lrow = lane & 15, lk = lane >> 4, ksteps = cin / 16;
    f32x4_t bfrag[GL_MAX_KSTEPS];
#pragma unroll
    for (int j = 0; j < GL_MAX_KSTEPS; ++j)
        bfrag[j] = j < ksteps ? *(const f32x4_t *)&w5[(size_t)lrow * cin + 16 * j + 4 * lk] : (f32x4_t)0.f;
    for (int r0 = 0; r0 < hin; r0 += band) {
        const int r1 = min(r0 + band, hin);
        const int lo = max(r0 - 1, 0), hi = min(r1 + 1, hin), npx = (hi - lo) * hin;  // staged rows [lo, hi)
        if (r0 > 0) __syncthreads();  // the previous band's gather has read s_c
        for (int i = tid; i < nblk * 16 * c4; i += 512) {
            const int r = fdiv(i, div_c4), c = (i - r * c4) * 4;
            f32x4_t v = 0.f;
            if (r < npx) v = load4<E>(&src[(size_t)(lo * hin + r) * cin + c]);
            *(f32x4_t *)&s_a2[r * lda + c] = v;
        }
        __syncthreads();
        for (int b = wave; b * 16 < npx; b += 8) {
            f32x4_t acc = 0.f;
            const float *pa = s_a2 + (b * 16 + lrow) * lda + 4 * lk;
#pragma unroll
            for (int j = 0; j < GL_MAX_KSTEPS; ++j) {
                if (j < ksteps) {
                    const f32x4_t av = *(const f32x4_t *)(pa + 16 * j);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bfrag[j][s], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s_c[(b * 16 + 4 * lk + r) * 16 + lrow] = acc[r];
        }
        __syncthreads();
        // the four contributions of every output pixel of the band's rows
        for (int i = 2 * r0 * side + tid; i < 2 * r1 * side; i += 512) {
            const int oy = fdiv(i, div_side), ox = i - oy * side;
            const int py = oy & 1, my = oy >> 1, px = ox & 1, mxx = ox >> 1;
            // (input offset, kernel index) = (0, 1), (-1, 3) for even rows; (0, 2), (+1, 0) for odd; same along x
            const int iy1 = my + (py ? 1 : -1), ky0 = py ? 2 : 1, ky1 = py ? 0 : 3;
            const int ix1 = mxx + (px ? 1 : -1), kx0 = px ? 2 : 1, kx1 = px ? 0 : 3;
            const bool yok = (unsigned)iy1 < (unsigned)hin, xok = (unsigned)ix1 < (unsigned)hin;
            const float t00 = s_c[((my - lo) * hin + mxx) * 16 + ky0 * 4 + kx0];
            const float t01 = xok ? s_c[((my - lo) * hin + ix1) * 16 + ky0 * 4 + kx1] : 0.f;
            const float t10 = yok ? s_c[((iy1 - lo) * hin + mxx) * 16 + ky1 * 4 + kx0] : 0.f;
            const float t11 = (yok && xok) ? s_c[((iy1 - lo) * hin + ix1) * 16 + ky1 * 4 + kx1] : 0.f;
            s_img[i] = (t00 + t01) + (t10 + t11);
        }
    }
    __syncthreads();
    // bias + tanh, per-image min / max
    float mn = INFINITY, mx = -INFINITY, se = 0.f;
    float *dst = img ? img + (size_t)n * npix : nullptr;
    const float *tg = target ? target + (size_t)n * npix : nullptr;
    for (int i = tid; i < npix; i += 512) {
        const float v = tanhf(s_img[i] + bias);
        s_img[i] = v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
        if (dst) dst[i] = v;
        if (tg) {
            const float dlt = tg[i] - v;
            se = fmaf(dlt, dlt, se);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
        se += __shfl_xor(se, o);
    }
    if (lane == 0) {
        s_red[wave] = mn;
        s_red[8 + wave] = mx;
        s_red[16 + wave] = se;
    }
    __syncthreads();
    mn = s_red[0];
    mx = s_red[8];
    se = s_red[16];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        mn = fminf(mn, s_red[k]);
        mx = fmaxf(mx, s_red[8 + k]);
        se += s_red[16 + k];
    }
    const float range = mx - mn;
    float *d01 = img01 + (size_t)n * npix;
    for (int i = tid; i < npix; i += 512) d01[i] = (s_img[i] - mn) / range;  // each thread rescales the pixels it wrote
    if (mse && tid == 0) mse[n] = se / (float)npix;
}

// ---------------------------------------------------------------------------------
// Assessor stem, fused: Conv2d(1 -> 64, k7, s2, p3, no bias) + BatchNorm + ReLU +
// MaxPool2d(3, s2, p1) (reference src/pso/util_cnn.py:85-88, 103-106).  NHWC output.
// Grid (image, band of pooled rows).  Lane = output channel (its 49 weights live in
// registers); the eight waves take PAIRS of horizontally adjacent conv pixels round-robin:
// the two 7-wide windows of a pair span 9 input pixels, read from the zero-padded image
// in LDS as three wave-wide broadcast 16-byte reads per kernel row (row stride and window
// start are multiples of 4 floats).  Conv rows of the band stay in LDS for the pooling.
// The stem is the first kernel of an assessor pass: it also clears the control block of the chained
// launch that follows later on the same stream (16-byte words, any number of workgroups), which saves
// that launch a memset of its own.
__device__ __forceinline__ void stem_zero(f32x4 *__restrict__ zero, int zero_quads, int block, int blocks) {
    if (zero == nullptr) return;
    for (int i = block * 512 + (int)threadIdx.x; i < zero_quads; i += blocks * 512) zero[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <typename E>
__global__ __launch_bounds__(512) void stem_kernel(const float *__restrict__ img01, const float *__restrict__ w,
                                                   const float *__restrict__ scale, const float *__restrict__ shift,
                                                   int side, int ho, int hp, int band, int ps, E *__restrict__ out,
                                                   f32x4 *__restrict__ zero, int zero_quads) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stem_zero(zero, zero_quads, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    const int prows = side + 6;
    float *s_img = smem;                // [side+6][ps], zero border of 3 on the left/top
    float *s_conv = smem + prows * ps;  // [(2*band+1)][ho][64]
    const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
    const int n = blockIdx.x, pr0 = blockIdx.y * band;
    const int pr1 = min(hp, pr0 + band);
    const int cr0 = 2 * pr0 - 1, cr1 = 2 * (pr1 - 1) + 1;  // conv rows needed (may stick out)
    const float *src = img01 + (size_t)n * side * side;
    for (int row = tid >> 6; row < prows; row += 8)
        for (int col = tid & 63; col < ps; col += 64) {
            const int y = row - 3, x = col - 3;
            s_img[row * ps + col] = ((unsigned)y < (unsigned)side && (unsigned)x < (unsigned)side) ? src[y * side + x] : 0.f;
        }
    float wreg[49];
#pragma unroll
    for (int t = 0; t < 49; ++t) wreg[t] = w[t * 64 + c];
    const float sc = scale[c], sh = shift[c];
    __syncthreads();
    const int rows = cr1 - cr0 + 1, pairs = (ho + 1) >> 1;
    for (int p = q; p < rows * pairs; p += 8) {
        const int r = p / pairs, cx = 2 * (p - r * pairs), cy = cr0 + r;
        float v0 = 0.f, v1 = 0.f;
        if ((unsigned)cy < (unsigned)ho) {
            const float *base = s_img + (2 * cy) * ps + 2 * cx;  // padded coordinates: +3 - 3
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const f32x4 x0 = *(const f32x4 *)(base + ky * ps);
                const f32x4 x1 = *(const f32x4 *)(base + ky * ps + 4);
                const f32x4 x2 = *(const f32x4 *)(base + ky * ps + 8);
                const float xs[12] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], x2[0], x2[1], x2[2], x2[3]};
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    a0 = fmaf(xs[kx], wreg[ky * 7 + kx], a0);
                    a1 = fmaf(xs[kx + 2], wreg[ky * 7 + kx], a1);
                }
            }
            v0 = fmaxf(a0 * sc + sh, 0.f);
            v1 = fmaxf(a1 * sc + sh, 0.f);
        }
        s_conv[(r * ho + cx) * 64 + c] = v0;
        if (cx + 1 < ho) s_conv[(r * ho + cx + 1) * 64 + c] = v1;
    }
    __syncthreads();
    for (int p = q; p < (pr1 - pr0) * hp; p += 8) {
        const int pr = pr0 + p / hp, pc = p % hp;
        float m = -INFINITY;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int cy = 2 * pr + dy;
            if ((unsigned)cy >= (unsigned)ho) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int cx = 2 * pc + dx;
                if ((unsigned)cx >= (unsigned)ho) continue;
                m = fmaxf(m, s_conv[((cy - cr0) * ho + cx) * 64 + c]);
            }
        }
        out[(((size_t)n * hp + pr) * hp + pc) * 64 + c] = (E)m;
    }
}

// ---------------------------------------------------------------------------------
// The stem on the matrix cores, one workgroup per image (used when the whole convolution
// output of an image fits LDS, i.e. the 28x28 inputs of the reference).  Per image the 7x7/2
// convolution is a (ho*ho) x 64 x 49 product: A[pixel][k = ky*7+kx] is gathered from the
// zero-padded image in LDS, B[k][channel] are the weights (K padded to 52 with zero weights),
// v_mfma_f32_16x16x4_f32 with lane l holding A[row l&15][k = 4t + (l>>4)] and B[that k][col l&15]
// at step t.  A wave takes 16-pixel row blocks round-robin and all four 16-channel column blocks
// of each (its 52 weight values stay in registers).  BatchNorm + ReLU are applied on the way to
// LDS, then the 3x3/2 max pool reads the conv image from LDS.
constexpr int STEM_KSTEPS = 13;  // ceil(49 / 4)

template <typename E>
__global__ __launch_bounds__(512) void stem_mfma_kernel(const float *__restrict__ img01, const float *__restrict__ w,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        int side, int ho, int hp, int ps, int band, FastDiv div_ho, FastDiv div_hp,
                                                        E *__restrict__ out, f32x4 *__restrict__ zero, int zero_quads) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stem_zero(zero, zero_quads, (int)blockIdx.x, (int)gridDim.x);
    // The pooled rows are walked in bands of `band` (all of them when the whole convolution image fits: 28x28
    // inputs; larger inputs in bands): a band's convolution rows 2 pr0 - 1 .. 2 pr1 - 1 are computed into LDS, pooled
    // and written; the rows two neighbouring bands both need are computed twice.
    const int prows = side + 6;
    // n / d by multiply-shift (FastDiv): the chip has no integer divide
    auto fdiv = [](int v, FastDiv f) { return (int)(((unsigned long long)(unsigned)v * f.mul) >> f.shift); };
    constexpr int LDC = 64 + 4;
    float *s_img = smem;                 // [prows][ps], zero border of 3
    float *s_conv = smem + prows * ps;   // [nblk * 16][LDC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.x;
    const float *src = img01 + (size_t)n * side * side;
    for (int row = tid >> 6; row < prows; row += 8)
        for (int col = tid & 63; col < ps; col += 64) {
            const int y = row - 3, x = col - 3;
            s_img[row * ps + col] = ((unsigned)y < (unsigned)side && (unsigned)x < (unsigned)side) ? src[y * side + x] : 0.f;
        }
    // weights [49][64] -> LDS rows of 80 floats (the conv buffer is free until the products start),
    // read with 16-byte loads; the fragment reads below are then conflict-free
    constexpr int LDW = 80;
    for (int i = tid; i < 49 * 16; i += 512) {
        const int k = i >> 4, c = (i & 15) * 4;
        *(f32x4_t *)&s_conv[k * LDW + c] = *(const f32x4_t *)&w[k * 64 + c];
    }
    const int lrow = lane & 15, lk = lane >> 4;
    float sc[4], sh[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        sc[cb] = scale[cb * 16 + lrow];
        sh[cb] = shift[cb * 16 + lrow];
    }
    __syncthreads();
    // this lane's k per step, as an offset into the padded image, and its weights
    int koff[STEM_KSTEPS];
    float bw[STEM_KSTEPS][4];
#pragma unroll
    for (int t = 0; t < STEM_KSTEPS; ++t) {
        const int k = 4 * t + lk;
        const int kk = k < 49 ? k : 0;
        koff[t] = (kk / 7) * ps + kk % 7;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) bw[t][cb] = k < 49 ? s_conv[k * LDW + cb * 16 + lrow] : 0.f;
    }
    for (int pr0 = 0; pr0 < hp; pr0 += band) {
        const int pr1 = min(pr0 + band, hp);
        const int cr0 = max(2 * pr0 - 1, 0), cr1 = min(2 * pr1 - 1, ho - 1), npx = (cr1 - cr0 + 1) * ho;  // conv rows [cr0, cr1]
        __syncthreads();  // the weights have left s_conv / the previous band's pooling has read it
        for (int b = wave; b * 16 < npx; b += 8) {
            const int px = min(b * 16 + lrow, npx - 1);  // rows past the band repeat its last pixel; they are never read back
            const int yl = fdiv(px, div_ho), x = px - yl * ho, y = cr0 + yl;
            const float *base = s_img + (2 * y) * ps + 2 * x;  // padded coordinates: +3 - 3
            float av[STEM_KSTEPS];
#pragma unroll
            for (int t = 0; t < STEM_KSTEPS; ++t) av[t] = base[koff[t]];
            f32x4_t acc[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = 0.f;
#pragma unroll
            for (int t = 0; t < STEM_KSTEPS; ++t)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bw[t][cb], acc[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s_conv[(b * 16 + 4 * lk + r) * LDC + cb * 16 + lrow] = fmaxf(acc[cb][r] * sc[cb] + sh[cb], 0.f);
        }
        __syncthreads();
        // MaxPool2d(3, s2, p1) of the band's pooled rows; consecutive threads take consecutive channels
        for (int e = tid; e < (pr1 - pr0) * hp * 64; e += 512) {
            const int c = e & 63, pl = e >> 6, prl = fdiv(pl, div_hp), pc = pl - prl * hp, pr = pr0 + prl;
            float m = -INFINITY;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int cy = 2 * pr + dy;
                if ((unsigned)cy >= (unsigned)ho) continue;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int cx = 2 * pc + dx;
                    if ((unsigned)cx >= (unsigned)ho) continue;
                    m = fmaxf(m, s_conv[((cy - cr0) * ho + cx) * LDC + c]);
                }
            }
            out[((size_t)n * hp * hp + (size_t)pr * hp + pc) * 64 + c] = (E)m;
        }
    }
}

// ---------------------------------------------------------------------------------
// Assessor head, fused: AdaptiveMaxPool2d(1) over the last feature map, Linear(C -> n_class)
// (reference src/pso/util_cnn.py:113-115), softmax + class pick + clamp + mode + eps
// (src/pso/util_discovery.py:69-80) and the inverter variant's weighted sum (:86-101).
// One workgroup per image; per-class dot products reduced by wavefront shuffles.
constexpr int MAX_CLASSES = 64;

template <typename E>
__global__ __launch_bounds__(256) void head_kernel(const E *__restrict__ feat, int hw, int c, const float *__restrict__ fc_w,
                                                   const float *__restrict__ fc_b, int n_class, float *__restrict__ logits,
                                                   float *__restrict__ fitness, int class_index,
                                                   const int *__restrict__ class_per_image, int mode, float threshold,
                                                   float eps, const float *__restrict__ mse, float w_ass, float w_rec) {
    __shared__ float s_part[4][MAX_CLASSES];
    __shared__ float s_logit[MAX_CLASSES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.x;
    const E *src = feat + (size_t)n * hw * c;
    // spatial max of this thread's channels: 4 consecutive channels per 1024-channel chunk (c <= 2048)
    float4 pooled[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ch = tid * 4 + u * 1024;
        pooled[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch < c) {
            f32x4 v = load4<E>(src + ch);
            for (int p = 1; p < hw; ++p) {
                const f32x4 t = load4<E>(src + (size_t)p * c + ch);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], t[q]);
            }
            pooled[u] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    // four classes at a time: their eight weight loads are issued together (one class per trip
    // of the loop cost a load round trip each); per class the sum runs in the same order as before
    for (int k0 = 0; k0 < n_class; k0 += 4) {
        float4 w[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ch = tid * 4 + u * 1024;
                w[j][u] = (k0 + j < n_class && ch < c) ? *(const float4 *)(fc_w + (size_t)(k0 + j) * c + ch)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k0 + j >= n_class) break;
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (tid * 4 + u * 1024 < c) {
                    acc = fmaf(pooled[u].x, w[j][u].x, acc);
                    acc = fmaf(pooled[u].y, w[j][u].y, acc);
                    acc = fmaf(pooled[u].z, w[j][u].z, acc);
                    acc = fmaf(pooled[u].w, w[j][u].w, acc);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0) s_part[wave][k0 + j] = acc;
        }
    }
    __syncthreads();
    if (tid < n_class) {
        const float l = ((s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid])) + fc_b[tid];
        s_logit[tid] = l;
        if (logits) logits[(size_t)n * n_class + tid] = l;
    }
    __syncthreads();
    if (tid == 0 && fitness) {
        float m = -INFINITY;
        for (int k = 0; k < n_class; ++k) m = fmaxf(m, s_logit[k]);
        float sum = 0.f;
        for (int k = 0; k < n_class; ++k) sum += expf(s_logit[k] - m);
        // several swarms in one batch: every image carries the class its swarm is after
        const int sel = n_class > 2 ? (class_per_image ? class_per_image[n] : class_index) : 1;
        const float p = expf(s_logit[sel] - m) / sum;
        const float pc = fminf(p + threshold, 1.0f);
        float f = (mode == GDPSO_MODE_IN_TRAINING ? pc : (1.0f - pc)) + eps;
        if (mse) f = (w_ass * f + w_rec * mse[n]) + eps;
        fitness[n] = f;
    }
}

// ---------------------------------------------------------------------------------
// Encoder first layer: Conv2d(1 -> F, k4, s2, p1) + bias + LeakyReLU(0.2)
// (reference src/inverter/utils_ae/util_inverter.py:170-171).  Same mapping as the stem.
__global__ __launch_bounds__(256) void enc_first_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                        const float *__restrict__ bias, int side, int ho, int cout,
                                                        float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int pside = side + 2;
    const int tid = threadIdx.x, n = blockIdx.x;
    const float *src = img + (size_t)n * side * side;
    for (int i = tid; i < pside * pside; i += 256) {
        const int y = i / pside - 1, x = i % pside - 1;
        smem[i] = ((unsigned)y < (unsigned)side && (unsigned)x < (unsigned)side) ? src[y * side + x] : 0.f;
    }
    __syncthreads();
    for (int c = tid & 63; c < cout; c += 64) {
        float wreg[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) wreg[t] = w[t * cout + c];
        const float b = bias[c];
        for (int p = tid >> 6; p < ho * ho; p += 4) {
            const int cy = p / ho, cx = p - cy * ho;
            const float *base = smem + (2 * cy) * pside + 2 * cx;
            float acc = 0.f;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) acc = fmaf(base[ky * pside + kx], wreg[ky * 4 + kx], acc);
            acc += b;
            out[((size_t)n * ho * ho + p) * cout + c] = acc > 0.f ? acc : 0.2f * acc;
        }
    }
}

// ---------------------------------------------------------------------------------
// First layer of the small "AlexNet" assessor, fused: Conv2d(1 -> C, k x k, padding p, bias) +
// activation + MaxPool2d(2) (reference src/pso/util_cnn.py:197, 216-219).  One workgroup per
// image: the zero-padded image sits in LDS, lane = output channel (its k*k weights in registers),
// the waves take pooled pixels round-robin; each pooled pixel is the maximum of its four
// activated convolution outputs (the pool drops an odd last row / column like torch's floor mode).
constexpr int ALEX_MAX_K = 7;
template <typename E>
__global__ __launch_bounds__(256) void alex_first_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                         const float *__restrict__ bias, int side, int k, int pad, int cout,
                                                         float slope, E *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int pside = side + 2 * pad, hc = pside - k + 1, hp = hc / 2;
    const int tid = threadIdx.x, n = blockIdx.x;
    const float *src = img + (size_t)n * side * side;
    for (int i = tid; i < pside * pside; i += 256) {
        const int y = i / pside - pad, x = i % pside - pad;
        smem[i] = ((unsigned)y < (unsigned)side && (unsigned)x < (unsigned)side) ? src[y * side + x] : 0.f;
    }
    __syncthreads();
    const int lanes = cout < 64 ? cout : 64;  // channels handled side by side (cout is a multiple of 32)
    const int groups = 256 / lanes;
    for (int c = tid % lanes; c < cout; c += lanes) {
        float wreg[ALEX_MAX_K * ALEX_MAX_K];
#pragma unroll
        for (int t = 0; t < ALEX_MAX_K * ALEX_MAX_K; ++t) wreg[t] = t < k * k ? w[t * cout + c] : 0.f;
        const float b = bias[c];
        for (int p = tid / lanes; p < hp * hp; p += groups) {
            const int py = p / hp, px = p - py * hp;
            float m = -INFINITY;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float *base = smem + (2 * py + (q >> 1)) * pside + 2 * px + (q & 1);
                float acc = 0.f;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) acc = fmaf(base[ky * pside + kx], wreg[ky * k + kx], acc);
                acc += b;
                m = fmaxf(m, fmaxf(acc, 0.f) + slope * fminf(acc, 0.f));
            }
            out[((size_t)n * hp * hp + p) * cout + c] = (E)m;
        }
    }
}

// MaxPool2d(2) on an NHWC map (floor mode), four channels per thread.
template <typename E>
__global__ __launch_bounds__(256) void pool2_kernel(const E *__restrict__ in, int n, int h, int c, E *__restrict__ out) {
    const int hp = h / 2, c4 = c / 4;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)n * hp * hp * c4) return;
    const int cc = (int)(e % c4) * 4;
    const long pix = e / c4;
    const int px = (int)(pix % hp), py = (int)((pix / hp) % hp), img = (int)(pix / ((long)hp * hp));
    const E *base = in + (((size_t)img * h + 2 * py) * h + 2 * px) * c + cc;
    f32x4 m = load4<E>(base);
    const f32x4 b = load4<E>(base + c), d = load4<E>(base + (size_t)h * c), f = load4<E>(base + (size_t)h * c + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = fmaxf(fmaxf(m[q], b[q]), fmaxf(d[q], f[q]));
    store4<E>(out + (size_t)pix * c + cc, m);
}

__global__ __launch_bounds__(256) void slice_cols_kernel(const float *__restrict__ in, int ld, int n, int cols,
                                                         float *__restrict__ out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)n * cols) return;
    const int r = (int)(e / cols), c = (int)(e - (long)r * cols);
    out[e] = in[(size_t)r * ld + c];
}

static int device_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
}

static size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

}  // namespace gdpso

using namespace gdpso;

// ============================================================================ generator
struct GenLayout;
struct gdpso_generator {
    // launch plans by batch size, made on first use (planning every layer again on every forward
    // pass was a measurable part of the host's enqueue time)
    mutable std::mutex plan_mutex;
    mutable std::map<int, std::shared_ptr<const GenLayout>> plans;
    int dtype;  // DType of the activations between the layers and of the convolution weights
    int z_dim, kpad, features, k1, side;
    ConvLayer l1, l2;
    const float *w5;
    float b5;
    const float *w1t = nullptr;  // first layer's weights as [z_dim][k * k * 2F] fp32 (rounded to `dtype`), for gen_first_kernel
    bool first_direct = false;   // the first layer runs in gen_first_kernel (short latents)
    int num_cus;
    bool last_mfma;            // the last layer runs on the matrix cores (features % 16 == 0, <= 128)
    size_t last_mfma_lds_max;  // dynamic LDS the MFMA kernel may use on this device
    // the whole generator as one launch with its maps in LDS (stage_resident.hip), or null: the reference's shape only
    // (first kernel 7, 64 features: 28x28 images)
    ResidentGenerator *resident = nullptr;
    ~gdpso_generator() { resident_generator_free(resident); }
    DeviceArena arena;
};

struct GenLayout {
    size_t zpad, a1, a2, splitk, total;
    ConvLayer l1, l2;  // the handle's layers with the launch plan for this batch size
};
namespace {
constexpr size_t MAX_CACHED_PLANS = 16;
std::shared_ptr<const GenLayout> gen_layout(const gdpso_generator *g, int n) {
    std::lock_guard<std::mutex> lock(g->plan_mutex);
    auto it = g->plans.find(n);
    if (it != g->plans.end()) return it->second;
    auto Lp = std::make_shared<GenLayout>();
    GenLayout &L = *Lp;
    ConvLayer &l1 = L.l1, &l2 = L.l2;
    l1 = g->l1;
    l2 = g->l2;
    plan_conv(l1, n, g->num_cus);
    plan_conv(l2, n, g->num_cus);
    size_t off = 0;
    const size_t es = dtype_size(g->dtype);
    L.zpad = off; off += align_up((size_t)n * g->kpad * es);
    L.a1 = off;   off += align_up((size_t)n * g->k1 * g->k1 * 2 * g->features * es);
    L.a2 = off;   off += align_up((size_t)n * 4 * g->k1 * g->k1 * g->features * es);
    L.splitk = off; off += align_up(std::max(conv_splitk_bytes(l1, n), conv_splitk_bytes(l2, n)));
    L.total = off;
    if (g->plans.size() >= MAX_CACHED_PLANS) g->plans.clear();
    g->plans[n] = Lp;
    return Lp;
}
}  // namespace

extern "C" int gdpso_generator_create(gdpso_generator **out, int z_dim, int features_g, int first_kernel,
                                      const float *const *t, int n_tensors) {
    return gdpso_generator_create_typed(out, z_dim, features_g, first_kernel, t, n_tensors, GDPSO_DTYPE_F32);
}

extern "C" int gdpso_generator_create_typed(gdpso_generator **out, int z_dim, int features_g, int first_kernel,
                                            const float *const *t, int n_tensors, int dtype) {
    GDPSO_REQUIRE(out && t, "generator_create: null argument");
    GDPSO_REQUIRE(dtype_ok(dtype), "generator_create: dtype must be GDPSO_DTYPE_F32, _BF16 or _F16");
    GDPSO_REQUIRE(dtype == DT_F32 || features_g % 64 == 0, "generator_create: 16-bit types need features_g to be a multiple of 64");
    GDPSO_REQUIRE(n_tensors == 14, "generator_create: expected 14 tensors");
    GDPSO_REQUIRE(z_dim > 0 && features_g >= 32 && features_g % 32 == 0, "generator_create: features_g must be a multiple of 32");
    GDPSO_REQUIRE(first_kernel >= 1 && first_kernel * first_kernel * 2 * features_g % 64 == 0, "generator_create: bad first_kernel");
    auto g = new gdpso_generator();
    g->dtype = dtype;
    g->z_dim = z_dim;
    g->kpad = (z_dim + dtype_kquant(dtype) - 1) / dtype_kquant(dtype) * dtype_kquant(dtype);
    g->features = features_g;
    g->k1 = first_kernel;
    g->side = 4 * first_kernel;
    g->num_cus = device_cus();
    const int c1 = 2 * features_g, c2 = features_g, k = first_kernel;
    int rc;
    {
        // ConvTranspose2d(Z -> 2F, k, s1, p0) on a 1x1 input is a GEMM [n,Z] x [Z, k*k*2F]; its
        // columns are ordered (ky, kx, co) so the result is already the NHWC map [n,k,k,2F].
        std::vector<float> w((size_t)z_dim * k * k * c1), sc, sh, sc1, sh1;
        // view as Conv2d weight [cout' = k*k*c1][cin = Z][1][1]
        for (int ci = 0; ci < z_dim; ++ci)
            for (int co = 0; co < c1; ++co)
                for (int p = 0; p < k * k; ++p) w[((size_t)p * c1 + co) * z_dim + ci] = t[0][((size_t)ci * c1 + co) * k * k + p];
        std::vector<float> packed;
        make_conv(g->l1, w.data(), k * k * c1, z_dim, 1, 1, 1, 0, 1, 1, ACT_RELU, packed, g->kpad, 0);
        g->l1.flops_nominal = 2.0 * z_dim * c1 * k * k;
        fold_bn(c1, t[1], t[2], t[3], t[4], t[5], sc1, sh1);
        sc.resize((size_t)k * k * c1);
        sh.resize(sc.size());
        for (int p = 0; p < k * k; ++p)
            for (int co = 0; co < c1; ++co) {
                sc[(size_t)p * c1 + co] = sc1[co];
                sh[(size_t)p * c1 + co] = sh1[co];
            }
        if ((rc = upload_conv(g->arena, g->l1, packed, sc, sh, dtype))) { delete g; return rc; }
        const char *e = getenv("GDPSO_GEN_FIRST_DIRECT");
        g->first_direct = (e ? e[0] != '0' : true) && z_dim <= GF_MAX_Z;
        if (g->first_direct) {
            const size_t ncol = (size_t)k * k * c1;
            std::vector<float> wt((size_t)z_dim * ncol);
            for (int ci = 0; ci < z_dim; ++ci)
                for (size_t col = 0; col < ncol; ++col) {
                    float v = w[col * z_dim + ci];
                    if (dtype == DT_BF16) {
                        const uint32_t u = (uint32_t)f32_to_bf16_bits(v) << 16;
                        std::memcpy(&v, &u, 4);
                    } else if (dtype == DT_F16) {
                        v = (float)(_Float16)v;
                    }
                    wt[(size_t)ci * ncol + col] = v;
                }
            if ((rc = g->arena.upload(wt, &g->w1t))) { delete g; return rc; }
        }
    }
    {
        std::vector<float> packed, sc, sh;
        make_convt_k4s2p1(g->l2, t[6], c1, c2, k, k, ACT_RELU, packed);
        fold_bn(c2, t[7], t[8], t[9], t[10], t[11], sc, sh);
        if ((rc = upload_conv(g->arena, g->l2, packed, sc, sh, dtype))) { delete g; return rc; }
    }
    {
        std::vector<float> w5((size_t)16 * c2);
        for (int ci = 0; ci < c2; ++ci)
            for (int p = 0; p < 16; ++p) w5[(size_t)p * c2 + ci] = t[12][(size_t)ci * 16 + p];
        if ((rc = g->arena.upload(w5, &g->w5))) { delete g; return rc; }
        g->b5 = t[13][0];
    }
    {
        const char *e = getenv("GDPSO_GEN_LAST_MFMA");
        g->last_mfma = (e ? e[0] != '0' : true) && features_g % 16 == 0 && features_g <= 16 * GL_MAX_KSTEPS;
        g->last_mfma_lds_max = 64 * 1024;
        if (g->last_mfma) {
            hipError_t e2 = hipErrorUnknown;
            GDPSO_WITH_DTYPE(dtype, e2 = hipFuncSetAttribute((const void *)gen_last_mfma_kernel<E>,
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
            if (e2 == hipSuccess) g->last_mfma_lds_max = 152 * 1024;
        }
    }
    if (first_kernel == 7 && features_g == 64 && resident_generator_enabled()) {
        if ((rc = resident_generator_new(g->arena, dtype, z_dim, g->first_direct ? g->w1t : nullptr, g->l1.scale, g->l1.shift, t[6], &g->l2,
                                         g->w5, g->b5, &g->resident))) { delete g; return rc; }
    }
    *out = g;
    return GDPSO_OK;
}

extern "C" void gdpso_generator_destroy(gdpso_generator *g) { delete g; }
extern "C" int gdpso_generator_image_side(const gdpso_generator *g) { return g ? g->side : 0; }

extern "C" size_t gdpso_generator_workspace_bytes(const gdpso_generator *g, int n) {
    if (!g || n <= 0) return 0;
    return gen_layout(g, n)->total;
}

extern "C" int gdpso_generator_forward(const gdpso_generator *g, void *stream_, int n, const void *z, int z_is_f64,
                                       float *img, float *img01, const float *target, float *mse, void *workspace,
                                       size_t workspace_bytes) {
    GDPSO_REQUIRE(g && z && img01 && workspace, "generator_forward: null argument");
    GDPSO_REQUIRE(n > 0, "generator_forward: n must be positive");
    GDPSO_REQUIRE((target == nullptr) == (mse == nullptr), "generator_forward: target and mse go together");
    hipStream_t stream = (hipStream_t)stream_;
    {
        int rc0 = chain_note_stream(stream, nullptr);  // (the assessor runs chained launches only beside a single stream)
        if (rc0) return rc0;
    }
    const std::shared_ptr<const GenLayout> plan = gen_layout(g, n);
    const GenLayout &L = *plan;
    const ConvLayer &l1 = L.l1, &l2 = L.l2;
    if (workspace_bytes < L.total) return fail(GDPSO_ERR_WORKSPACE, "generator_forward: workspace too small");
    char *ws = (char *)workspace;
    void *zpad = ws + L.zpad, *a1 = ws + L.a1, *a2 = ws + L.a2;
    float *sk = (float *)(ws + L.splitk);
    int rc;
    if ((rc = conv_splitk_prepare(stream, sk))) return rc;
    // the whole generator in one launch with its maps in LDS, at the batch sizes the resident launches serve (the
    // first layer too when the latents are short; else it stays a GEMM on the matrix cores and the launch reads its map)
    const int resident_images = g->resident ? resident_generator_images(plan_batch(n), g->num_cus, g->dtype) : 0;
    if (resident_images > 0) {
        if (g->first_direct)
            return launch_resident_generator(g->resident, stream, n, resident_images, g->num_cus, z, z_is_f64, nullptr, img, img01, target, mse);
        const long ne = (long)n * g->kpad;
        GDPSO_WITH_DTYPE(g->dtype, hipLaunchKernelGGL(latent_prep_kernel<E>, dim3(ceil_div(ne, 256)), dim3(256), 0, stream, z, z_is_f64, n,
                                                      g->z_dim, g->kpad, (E *)zpad));
        if ((rc = launch_conv(l1, stream, n, zpad, nullptr, a1, sk))) return rc;
        return launch_resident_generator(g->resident, stream, n, resident_images, g->num_cus, nullptr, 0, a1, img, img01, target, mse);
    }
    if (g->first_direct) {
        const int ncol = g->k1 * g->k1 * 2 * g->features;
        GDPSO_WITH_DTYPE(g->dtype, hipLaunchKernelGGL(gen_first_kernel<E>, dim3(ceil_div(ncol / 4, 256), ceil_div(n, GF_IMAGES)), dim3(256), 0,
                                                      stream, z, z_is_f64, n, g->z_dim, ncol, g->w1t, l1.scale, l1.shift, (E *)a1));
    } else {
        const long ne = (long)n * g->kpad;
        GDPSO_WITH_DTYPE(g->dtype, hipLaunchKernelGGL(latent_prep_kernel<E>, dim3(ceil_div(ne, 256)), dim3(256), 0, stream, z, z_is_f64, n,
                                                      g->z_dim, g->kpad, (E *)zpad));
        if ((rc = launch_conv(l1, stream, n, zpad, nullptr, a1, sk))) return rc;
    }
    if ((rc = launch_conv(l2, stream, n, a1, nullptr, a2, sk))) return rc;
    const int hin = 2 * g->k1;
    const size_t base = ((size_t)g->side * g->side + 32) * sizeof(float);
    const size_t staged = base + (size_t)hin * hin * g->features * sizeof(float);
    // the MFMA form stages bands of the input map: the whole map when it fits half a CU's LDS (two workgroups a CU),
    // else the fewest bands that do; a band's rows and one either side, as 16-row blocks of fp32 rows padded by 4
    auto mfma_lds_of = [&](int band) {
        const int nblk = ((band >= hin ? hin : band + 2) * hin + 15) / 16;
        return base + ((size_t)nblk * 16 * 16 + (size_t)nblk * 16 * (g->features + 4)) * sizeof(float);
    };
    int band = hin;
    for (int parts = 1; parts <= hin && mfma_lds_of(band) > std::min<size_t>(g->last_mfma_lds_max, 78 * 1024); ++parts) band = ceil_div(hin, parts);
    const size_t mfma_lds = mfma_lds_of(band);
    if (g->last_mfma && mfma_lds <= g->last_mfma_lds_max)
        GDPSO_WITH_DTYPE(g->dtype, hipLaunchKernelGGL(gen_last_mfma_kernel<E>, dim3(n), dim3(512), mfma_lds, stream, (const E *)a2, g->w5,
                                                      g->b5, hin, g->features, band, img, img01, target, mse,
                                                      fast_div_for((unsigned)(g->features / 4)), fast_div_for((unsigned)(2 * hin))));
    else if (staged <= 64 * 1024)
        GDPSO_WITH_DTYPE(g->dtype, hipLaunchKernelGGL((gen_last_kernel<true, E>), dim3(n), dim3(512), staged, stream, (const E *)a2, g->w5,
                                                      g->b5, hin, g->features, img, img01, target, mse));
    else
        GDPSO_WITH_DTYPE(g->dtype, hipLaunchKernelGGL((gen_last_kernel<false, E>), dim3(n), dim3(512), base, stream, (const E *)a2, g->w5,
                                                      g->b5, hin, g->features, img, img01, target, mse));
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

// ============================================================================= assessor
struct Bottleneck {
    ConvLayer c1, c2, c3, ds;
    bool has_ds;
};

struct AssLayout;
struct gdpso_assessor {
    mutable std::mutex plan_mutex;  // launch plans by batch size, made on first use
    mutable std::map<int, std::shared_ptr<const AssLayout>> plans;
    int dtype;  // DType of the feature maps and of the convolution weights
    int kind = 0;  // 0: bottleneck ResNet, 1: the 4-convolution "AlexNet"
    // AlexNet: first layer (Cin = 1) on the vector ALUs, three convolutions, two hidden Linear layers as
    // 1x1 convolutions; the last Linear + softmax run in head_kernel (fc_w / fc_b, feat_c = 256, feat_hw = 1)
    const float *alex_w1 = nullptr, *alex_b1 = nullptr;
    int alex_k = 0, alex_pad = 0, alex_c1 = 32, alex_h[5] = {0, 0, 0, 0, 0};  // map side after each convolution's pool
    float alex_slope = 0.f;
    ConvLayer alex_conv[3], alex_fc[2];
    int n_class, side, ho, hp, stem_band, stem_ps;
    const float *stem_w, *stem_scale, *stem_shift;
    std::vector<Bottleneck> blocks;
    int chain_first = 0;  // first block behind the first stage
    // stem and first stage as one launch with the activations in LDS (stage_resident.hip), or null: 28x28 images only
    ResidentStage *resident = nullptr;
    int resident_blocks = 0;  // blocks its launch covers: the first stage's, or the first two stages'
    ~gdpso_assessor() { resident_stage_free(resident); }
    const float *fc_w, *fc_b;
    int feat_c, feat_hw;
    double flops;
    int num_cus;
    DeviceArena arena;
};

struct AssLayout {
    size_t x, y, t1, t2, sc, splitk, chain_ctl, chain_z, chain_out, total;
    std::vector<Bottleneck> blocks;
    ConvLayer alex_conv[3], alex_fc[2];  // AlexNet: the handle's layers with this batch size's launch plan
    // blocks [chain_first, end) as one persistent launch, when every layer of theirs can be chained at
    // this batch size (null otherwise: a launch per layer)
    ConvChain *chain = nullptr;
    int chain_first = 0;
    bool resident = false;  // blocks [0, chain_first) through the handle's resident stage at this batch size
    int resident_images = 0;  // ... with this many images a workgroup
    bool resident_second = false;  // ... and the second stage in the same launch
    AssLayout() = default;
    AssLayout(const AssLayout &) = delete;
    AssLayout &operator=(const AssLayout &) = delete;
    ~AssLayout() { chain_free(chain); }
};
namespace {
std::shared_ptr<const AssLayout> ass_layout(const gdpso_assessor *a, int n) {
    std::lock_guard<std::mutex> lock(a->plan_mutex);
    auto it = a->plans.find(n);
    if (it != a->plans.end()) return it->second;
    auto Lp = std::make_shared<AssLayout>();
    AssLayout &L = *Lp;
    L.blocks = a->blocks;
    // (buffers sized for the batch rounded up to a multiple of 16 images: a chained launch gives each of its 8 or
    // 16 shares room for ceil(n / shares) images)
    const size_t n_pad = ((size_t)n + 15) / 16 * 16;
    size_t xy = n_pad * a->hp * a->hp * 64, t1 = 0, t2 = 0, sc = 0, sk = 0;
    auto out_elems = [&](const ConvLayer &c) { return n_pad * c.hout * c.wout * c.cout; };
    if (a->kind == 1) {
        xy = (size_t)n * a->alex_h[0] * a->alex_h[0] * a->alex_c1;
        for (int i = 0; i < 3; ++i) {
            L.alex_conv[i] = a->alex_conv[i];
            plan_conv(L.alex_conv[i], n, a->num_cus);
            xy = std::max(xy, out_elems(L.alex_conv[i]));
            sk = std::max(sk, conv_splitk_bytes(L.alex_conv[i], n));
        }
        for (int i = 0; i < 2; ++i) {
            L.alex_fc[i] = a->alex_fc[i];
            plan_conv(L.alex_fc[i], n, a->num_cus);
            xy = std::max(xy, out_elems(L.alex_fc[i]));
            sk = std::max(sk, conv_splitk_bytes(L.alex_fc[i], n));
        }
    }
    for (auto &b : L.blocks) {
        plan_conv(b.c1, n, a->num_cus);
        plan_conv(b.c2, n, a->num_cus);
        plan_conv(b.c3, n, a->num_cus);
        t1 = std::max(t1, out_elems(b.c1));
        t2 = std::max(t2, out_elems(b.c2));
        xy = std::max(xy, out_elems(b.c3));
        sk = std::max({sk, conv_splitk_bytes(b.c1, n), conv_splitk_bytes(b.c2, n), conv_splitk_bytes(b.c3, n)});
        if (b.has_ds) {
            plan_conv(b.ds, n, a->num_cus);
            sc = std::max(sc, out_elems(b.ds));
            sk = std::max(sk, conv_splitk_bytes(b.ds, n));
        }
    }
    size_t off = 0;
    const size_t es = dtype_size(a->dtype);
    L.x = off; off += align_up(xy * es);
    L.y = off; off += align_up(xy * es);
    L.t1 = off; off += align_up(t1 * es);
    L.t2 = off; off += align_up(t2 * es);
    L.sc = off; off += align_up(sc * es);
    L.splitk = off; off += align_up(sk);
    L.chain_ctl = L.chain_z = L.chain_out = 0;
    // Every stage behind the first in one launch (the first stage's layers are long enough for their
    // launches not to matter, and two of its three layers per block already share one).  The chain reads
    // the first stage's output where it lies, keeps its intermediates in per-share slices of y-or-x / z /
    // t1 / t2 / sc and writes the last block's output to a buffer of its own.
    L.resident_images = a->resident != nullptr ? resident_stage_images(plan_batch(n), a->num_cus, a->dtype) : 0;
    L.resident = L.resident_images > 0;
    // (a chain picks up behind the resident launch: behind the first stage, or behind the second)
    L.resident_second = L.resident && resident_stage_second(a->resident, plan_batch(n));
    L.chain_first = L.resident ? (L.resident_second ? a->resident_blocks : a->chain_first) : a->chain_first;
    // (large batches: the chain's steps hold many tiles per workgroup, and the launches per layer, whose cost per layer no
    // longer matters there, deal them better: behind the resident launch +4 % at 2304 ... 3072 images in fp32, +8 % with
    // 16-bit storage, even at 2048.  GDPSO_CHAIN_MAX_IMAGES, by the planned batch)
    static const int chain_max_images = [] {
        const char *e = getenv("GDPSO_CHAIN_MAX_IMAGES");
        return e ? atoi(e) : 2048;
    }();
    if (a->kind == 0 && chain_enabled() && L.chain_first > 0 && L.chain_first < (int)L.blocks.size() && plan_batch(n) <= chain_max_images) {
        size_t zmax = 0;
        for (size_t i = L.chain_first; i < L.blocks.size(); ++i) zmax = std::max(zmax, out_elems(L.blocks[i].c3));
        int nteams = 0;
        for (int t = chain_teams(); t >= 1 && !nteams; --t) {  // two teams per XCD need whole tiles in a sixteenth of the images
            bool ok = true;
            for (size_t i = L.chain_first; i < L.blocks.size() && ok; ++i) {
                const Bottleneck &b = L.blocks[i];
                ok = chain_layer_ok(b.c1, n, t) && chain_layer_ok(b.c2, n, t) && chain_layer_ok(b.c3, n, t) && (!b.has_ds || chain_layer_ok(b.ds, n, t));
            }
            if (ok) nteams = t;
        }
        if (nteams) {
            const size_t xy_bytes = align_up(xy * es), z_bytes = align_up(zmax * es);
            L.chain_ctl = off; off += align_up(chain_ctl_bytes());
            L.chain_z = off; off += z_bytes;
            L.chain_out = off; off += align_up(out_elems(L.blocks.back().c3) * es);
            ConvChain *c = chain_new(n, nteams);
            const bool even = L.chain_first % 2 == 0;  // the first stage's output is in x after an even number of blocks
            const ChainBuf input{even ? L.x : L.y, 0}, ping{even ? L.y : L.x, xy_bytes}, pong{L.chain_z, z_bytes};
            const ChainBuf bt1{L.t1, align_up(t1 * es)}, bt2{L.t2, align_up(t2 * es)}, bsc{L.sc, align_up(sc * es)};
            const ChainBuf fin{L.chain_out, 0};
            ChainBuf cur = input;
            int rc = GDPSO_OK;
            for (size_t i = L.chain_first; i < L.blocks.size() && !rc; ++i) {
                const Bottleneck &b = L.blocks[i];
                const bool last = i + 1 == L.blocks.size();
                const ChainBuf next = last ? fin : ((i - L.chain_first) % 2 == 0 ? ping : pong);
                ChainBuf res = cur;
                if (b.has_ds) {
                    rc = chain_add(c, &b.ds, cur, nullptr, bsc, &b.c1, &bt1);
                    res = bsc;
                } else {
                    rc = chain_add(c, &b.c1, cur, nullptr, bt1);
                }
                if (!rc) rc = chain_add(c, &b.c2, bt1, nullptr, bt2);
                if (!rc) rc = chain_add(c, &b.c3, bt2, &res, next);
                cur = next;
            }
            if (!rc) rc = chain_finish(c, L.chain_ctl);
            if (rc)
                chain_free(c);  // the per-layer launches remain
            else
                L.chain = c;
        }
    }
    L.total = off;
    if (a->plans.size() >= MAX_CACHED_PLANS) a->plans.clear();
    a->plans[n] = Lp;
    return Lp;
}
}  // namespace

extern "C" int gdpso_assessor_create_resnet(gdpso_assessor **out, const int depths[4], int n_class, int img_side,
                                            const float *const *t, int n_tensors) {
    return gdpso_assessor_create_resnet_typed(out, depths, n_class, img_side, t, n_tensors, GDPSO_DTYPE_F32);
}

extern "C" int gdpso_assessor_create_resnet_typed(gdpso_assessor **out, const int depths[4], int n_class, int img_side,
                                                  const float *const *t, int n_tensors, int dtype) {
    GDPSO_REQUIRE(out && depths && t, "assessor_create: null argument");
    GDPSO_REQUIRE(dtype_ok(dtype), "assessor_create: dtype must be GDPSO_DTYPE_F32, _BF16 or _F16");
    GDPSO_REQUIRE(n_class >= 2 && n_class <= MAX_CLASSES, "assessor_create: n_class must be in [2, 64]");
    GDPSO_REQUIRE(img_side >= 8 && img_side <= 64, "assessor_create: img_side must be in [8, 64]");
    int expect = 5 + 2;
    for (int s = 0; s < 4; ++s) {
        GDPSO_REQUIRE(depths[s] >= 1, "assessor_create: every stage needs at least one block");
        expect += depths[s] * 15 + 5;
    }
    GDPSO_REQUIRE(n_tensors == expect, "assessor_create: wrong number of tensors for these depths");
    auto a = new gdpso_assessor();
    a->dtype = dtype;
    a->n_class = n_class;
    a->side = img_side;
    a->ho = (img_side + 6 - 7) / 2 + 1;
    a->hp = (a->ho + 2 - 3) / 2 + 1;
    a->num_cus = device_cus();
    // pooled rows per stem workgroup: conv rows of the band must fit 64 KiB of LDS
    {
        const size_t row_bytes = (size_t)a->ho * 64 * 4;
        a->stem_ps = (2 * a->ho + 12 + 3) / 4 * 4;  // padded row: the last pair's window reads up to 2*ho + 9
        const size_t avail = (size_t)64 * 1024 - (size_t)(img_side + 6) * a->stem_ps * 4;
        const int max_rows = (int)std::max<size_t>(3, avail / row_bytes);
        a->stem_band = std::max(1, std::min(a->hp, (max_rows - 1) / 2));
    }
    int rc, ti = 0;
    {
        std::vector<float> w((size_t)49 * 64), sc, sh;
        for (int co = 0; co < 64; ++co)
            for (int p = 0; p < 49; ++p) w[(size_t)p * 64 + co] = t[0][(size_t)co * 49 + p];
        fold_bn(64, nullptr, t[1], t[2], t[3], t[4], sc, sh);
        if ((rc = a->arena.upload(w, &a->stem_w)) || (rc = a->arena.upload(sc, &a->stem_scale)) ||
            (rc = a->arena.upload(sh, &a->stem_shift))) { delete a; return rc; }
        ti = 5;
        a->flops = 2.0 * a->ho * a->ho * 64 * 49;
    }
    int cin = 64, h = a->hp;
    std::vector<const float *> s1w, s2w;  // the first two stages' torch weights: conv1, conv2, conv3 per block
    const float *s1w_ds = nullptr, *s2w_ds = nullptr;
    for (int s = 0; s < 4; ++s) {
        const int mid = 64 << s;
        for (int b = 0; b < depths[s]; ++b) {
            const int stride = (b == 0 && s > 0) ? 2 : 1;
            Bottleneck blk;
            std::memset(&blk, 0, sizeof(blk));
            std::vector<float> packed, sc, sh;
            make_conv(blk.c1, t[ti], mid, cin, 1, 1, 1, 0, h, h, ACT_RELU, packed);
            fold_bn(mid, nullptr, t[ti + 1], t[ti + 2], t[ti + 3], t[ti + 4], sc, sh);
            if ((rc = upload_conv(a->arena, blk.c1, packed, sc, sh, dtype))) { delete a; return rc; }
            make_conv(blk.c2, t[ti + 5], mid, mid, 3, 3, stride, 1, h, h, ACT_RELU, packed);
            fold_bn(mid, nullptr, t[ti + 6], t[ti + 7], t[ti + 8], t[ti + 9], sc, sh);
            if ((rc = upload_conv(a->arena, blk.c2, packed, sc, sh, dtype))) { delete a; return rc; }
            const int h2 = blk.c2.hout;
            make_conv(blk.c3, t[ti + 10], mid * 4, mid, 1, 1, 1, 0, h2, h2, ACT_RELU, packed);  // ReLU after the residual add
            fold_bn(mid * 4, nullptr, t[ti + 11], t[ti + 12], t[ti + 13], t[ti + 14], sc, sh);
            if ((rc = upload_conv(a->arena, blk.c3, packed, sc, sh, dtype))) { delete a; return rc; }
            if (s <= 1) {
                auto &sw = s == 0 ? s1w : s2w;
                sw.push_back(t[ti]);
                sw.push_back(t[ti + 5]);
                sw.push_back(t[ti + 10]);
                if (b == 0) (s == 0 ? s1w_ds : s2w_ds) = t[ti + 15];
            }
            ti += 15;
            blk.has_ds = (b == 0);
            a->flops += blk.c1.flops_nominal + blk.c2.flops_nominal + blk.c3.flops_nominal;
            if (blk.has_ds) {
                make_conv(blk.ds, t[ti], mid * 4, cin, 1, 1, stride, 0, h, h, ACT_NONE, packed);
                fold_bn(mid * 4, nullptr, t[ti + 1], t[ti + 2], t[ti + 3], t[ti + 4], sc, sh);
                if ((rc = upload_conv(a->arena, blk.ds, packed, sc, sh, dtype))) { delete a; return rc; }
                ti += 5;
                a->flops += blk.ds.flops_nominal;
            }
            a->blocks.push_back(blk);
            cin = mid * 4;
            h = h2;
        }
    }
    a->feat_c = cin;
    a->feat_hw = h * h;
    a->chain_first = depths[0];
    if (img_side == 28 && depths[0] <= RS_MAX_BLOCKS && resident_stage_enabled()) {
        std::vector<const ConvLayer *> c1, c2, c3, d1, d2, d3;
        for (int b = 0; b < depths[0] + depths[1]; ++b) {
            (b < depths[0] ? c1 : d1).push_back(&a->blocks[b].c1);
            (b < depths[0] ? c2 : d2).push_back(&a->blocks[b].c2);
            (b < depths[0] ? c3 : d3).push_back(&a->blocks[b].c3);
        }
        if ((rc = resident_stage_new(a->arena, dtype, depths[0], s1w.data(), s1w_ds, c1.data(), c2.data(), c3.data(), &a->blocks[0].ds,
                                     a->stem_w, a->stem_scale, a->stem_shift, depths[1], s2w.data(), s2w_ds, d1.data(), d2.data(),
                                     d3.data(), &a->blocks[depths[0]].ds, &a->resident))) { delete a; return rc; }
        a->resident_blocks = depths[0] + (resident_stage_has_second(a->resident) ? depths[1] : 0);
    }
    {
        std::vector<float> w(t[ti], t[ti] + (size_t)n_class * cin), b(t[ti + 1], t[ti + 1] + n_class);
        if ((rc = a->arena.upload(w, &a->fc_w)) || (rc = a->arena.upload(b, &a->fc_b))) { delete a; return rc; }
        a->flops += 2.0 * cin * n_class;
    }
    *out = a;
    return GDPSO_OK;
}

// The small "AlexNet" assessor of reference src/pso/util_cnn.py:193-249: four k x k convolutions
// (1 -> 32 -> 64 -> 128 -> 256, bias, activation, MaxPool2d(2) after each), flatten, Linear(-> 256),
// Linear(256 -> 256), Linear(256 -> n_class), the activation after the first two (Dropout is inactive
// in eval mode).
extern "C" int gdpso_assessor_create_alexnet(gdpso_assessor **out, int n_class, int img_side, int ksize, int pad_same,
                                             int activation, const float *const *t, int n_tensors, int dtype) {
    GDPSO_REQUIRE(out && t, "assessor_create_alexnet: null argument");
    GDPSO_REQUIRE(dtype_ok(dtype), "assessor_create_alexnet: dtype must be GDPSO_DTYPE_F32, _BF16 or _F16");
    GDPSO_REQUIRE(n_tensors == 14, "assessor_create_alexnet: expected 14 tensors (4 convolutions and 3 Linear layers, weight and bias)");
    GDPSO_REQUIRE(n_class >= 2 && n_class <= MAX_CLASSES, "assessor_create_alexnet: n_class must be in [2, 64]");
    GDPSO_REQUIRE(ksize >= 1 && ksize <= ALEX_MAX_K && (ksize % 2 == 1 || !pad_same),
                  "assessor_create_alexnet: kernel size must be in [1, 7], odd with 'same' padding");
    GDPSO_REQUIRE(img_side >= 8 && img_side <= 128, "assessor_create_alexnet: img_side must be in [8, 128]");
    GDPSO_REQUIRE(activation == ACT_RELU || activation == ACT_LEAKY02, "assessor_create_alexnet: activation must be ReLU or LeakyReLU(0.2)");
    const int pad = pad_same ? (ksize - 1) / 2 : 0;
    int h = img_side, hs[5];
    for (int i = 0; i < 4; ++i) {
        const int hc = h + 2 * pad - ksize + 1;
        // the error torch raises from the reference's constructor (its dummy forward pass, util_cnn.py:207-209)
        GDPSO_REQUIRE(hc >= 1, "assessor_create_alexnet: kernel size can't be greater than actual input size (a convolution's input "
                               "is smaller than its window: with padding 'valid' the image is too small for four convolutions)");
        h = hc / 2;
        GDPSO_REQUIRE(h >= 1, "assessor_create_alexnet: a pooled map is empty (image too small)");
        hs[i] = h;
    }
    const int to_linear = 256 * h * h;
    GDPSO_REQUIRE(to_linear <= 4096, "assessor_create_alexnet: more than 4096 flattened features");
    auto a = new gdpso_assessor();
    a->dtype = dtype;
    a->kind = 1;
    a->n_class = n_class;
    a->side = img_side;
    a->ho = a->hp = a->stem_band = a->stem_ps = 0;
    a->num_cus = device_cus();
    a->alex_k = ksize;
    a->alex_pad = pad;
    a->alex_slope = activation == ACT_RELU ? 0.f : 0.2f;
    for (int i = 0; i < 4; ++i) a->alex_h[i] = hs[i];
    int rc;
    {
        // the 16-bit convolution kernels take channels in multiples of 64: the first layer then writes 32
        // more channels of zeros (zero weights, zero bias), which the second layer's zero-padded weights ignore
        const int c1 = a->alex_c1 = dtype == DT_F32 ? 32 : 64;
        std::vector<float> w1((size_t)ksize * ksize * c1, 0.f), b1(c1, 0.f);
        std::copy(t[1], t[1] + 32, b1.begin());
        for (int co = 0; co < 32; ++co)
            for (int p = 0; p < ksize * ksize; ++p) w1[(size_t)p * c1 + co] = t[0][(size_t)co * ksize * ksize + p];
        if ((rc = a->arena.upload(w1, &a->alex_w1)) || (rc = a->arena.upload(b1, &a->alex_b1))) { delete a; return rc; }
        const int hc = img_side + 2 * pad - ksize + 1;
        a->flops = 2.0 * hc * hc * 32 * ksize * ksize;
    }
    const int chans[4] = {32, 64, 128, 256};
    std::vector<float> packed, ones, bias;
    for (int i = 0; i < 3; ++i) {
        const int cin = chans[i], cout = chans[i + 1], hin = hs[i];
        make_conv(a->alex_conv[i], t[2 + 2 * i], cout, cin, ksize, ksize, 1, pad, hin, hin, activation, packed,
                  dtype == DT_F32 ? 0 : std::max(cin, 64), 0);
        ones.assign(cout, 1.f);
        bias.assign(t[3 + 2 * i], t[3 + 2 * i] + cout);
        if ((rc = upload_conv(a->arena, a->alex_conv[i], packed, ones, bias, dtype))) { delete a; return rc; }
        a->flops += a->alex_conv[i].flops_nominal;
    }
    {
        // flatten: torch's order is (channel, y, x); the NHWC map is (y, x, channel)
        std::vector<float> w((size_t)256 * to_linear);
        for (int o = 0; o < 256; ++o)
            for (int c = 0; c < 256; ++c)
                for (int p = 0; p < h * h; ++p) w[(size_t)o * to_linear + (size_t)p * 256 + c] = t[8][(size_t)o * to_linear + (size_t)c * h * h + p];
        make_conv(a->alex_fc[0], w.data(), 256, to_linear, 1, 1, 1, 0, 1, 1, activation, packed);
        ones.assign(256, 1.f);
        bias.assign(t[9], t[9] + 256);
        if ((rc = upload_conv(a->arena, a->alex_fc[0], packed, ones, bias, dtype))) { delete a; return rc; }
        make_conv(a->alex_fc[1], t[10], 256, 256, 1, 1, 1, 0, 1, 1, activation, packed);
        bias.assign(t[11], t[11] + 256);
        if ((rc = upload_conv(a->arena, a->alex_fc[1], packed, ones, bias, dtype))) { delete a; return rc; }
        a->flops += a->alex_fc[0].flops_nominal + a->alex_fc[1].flops_nominal;
    }
    a->feat_c = 256;
    a->feat_hw = 1;
    {
        std::vector<float> w(t[12], t[12] + (size_t)n_class * 256), b(t[13], t[13] + n_class);
        if ((rc = a->arena.upload(w, &a->fc_w)) || (rc = a->arena.upload(b, &a->fc_b))) { delete a; return rc; }
        a->flops += 2.0 * 256 * n_class;
    }
    *out = a;
    return GDPSO_OK;
}

extern "C" void gdpso_assessor_destroy(gdpso_assessor *a) { delete a; }
extern "C" int gdpso_assessor_n_class(const gdpso_assessor *a) { return a ? a->n_class : 0; }
extern "C" double gdpso_assessor_flops_per_image(const gdpso_assessor *a) { return a ? a->flops : 0.0; }
extern "C" size_t gdpso_assessor_workspace_bytes(const gdpso_assessor *a, int n) {
    if (!a || n <= 0) return 0;
    return ass_layout(a, n)->total;
}

static int assessor_forward_impl(const gdpso_assessor *a, void *stream_, int n, const float *img01, float *logits,
                                 float *fitness, int class_index, const int *class_per_image, int mode, float threshold,
                                 float eps, const float *mse, float w_ass, float w_rec, void *workspace,
                                 size_t workspace_bytes) {
    GDPSO_REQUIRE(a && img01 && workspace, "assessor_forward: null argument");
    GDPSO_REQUIRE(n > 0, "assessor_forward: n must be positive");
    GDPSO_REQUIRE(logits || fitness, "assessor_forward: nothing to compute");
    GDPSO_REQUIRE(mode == GDPSO_MODE_IN_TRAINING || mode == GDPSO_MODE_OUT_TRAINING, "assessor_forward: unknown fitness mode");
    GDPSO_REQUIRE(!fitness || a->n_class == 2 || class_per_image || (class_index >= 0 && class_index < a->n_class),
                  "assessor_forward: class_index out of range");
    hipStream_t stream = (hipStream_t)stream_;
    bool single_stream = true;
    {
        int rc0 = chain_note_stream(stream, &single_stream);
        if (rc0) return rc0;
    }
    const std::shared_ptr<const AssLayout> plan = ass_layout(a, n);
    const AssLayout &L = *plan;
    if (workspace_bytes < L.total) return fail(GDPSO_ERR_WORKSPACE, "assessor_forward: workspace too small");
    char *ws = (char *)workspace;
    void *x = ws + L.x, *y = ws + L.y, *t1 = ws + L.t1, *t2 = ws + L.t2, *sc = ws + L.sc;
    float *sk = (float *)(ws + L.splitk);
    if (a->kind == 1) {
        int rc;
        const int pside = a->side + 2 * a->alex_pad;
        GDPSO_WITH_DTYPE(a->dtype, hipLaunchKernelGGL(alex_first_kernel<E>, dim3(n), dim3(256), (size_t)pside * pside * sizeof(float), stream,
                                                      img01, a->alex_w1, a->alex_b1, a->side, a->alex_k, a->alex_pad, a->alex_c1, a->alex_slope,
                                                      (E *)x));
        if ((rc = conv_splitk_prepare(stream, sk))) return rc;
        for (int i = 0; i < 3; ++i) {
            const ConvLayer &c = L.alex_conv[i];
            if ((rc = launch_conv(c, stream, n, x, nullptr, y, sk))) return rc;
            const long quads = (long)n * (c.hout / 2) * (c.hout / 2) * (c.cout / 4);
            GDPSO_WITH_DTYPE(a->dtype, hipLaunchKernelGGL(pool2_kernel<E>, dim3(ceil_div(quads, 256)), dim3(256), 0, stream, (const E *)y, n,
                                                          c.hout, c.cout, (E *)x));
        }
        if ((rc = launch_conv(L.alex_fc[0], stream, n, x, nullptr, y, sk))) return rc;
        if ((rc = launch_conv(L.alex_fc[1], stream, n, y, nullptr, x, sk))) return rc;
        GDPSO_WITH_DTYPE(a->dtype, hipLaunchKernelGGL(head_kernel<E>, dim3(n), dim3(256), 0, stream, (const E *)x, 1, 256, a->fc_w, a->fc_b,
                                                      a->n_class, logits, fitness, class_index, class_per_image, mode, threshold, eps,
                                                      mse, w_ass, w_rec));
        GDPSO_HIP(hipGetLastError());
        return GDPSO_OK;
    }
    static const bool stem_mfma = [] {
        const char *e = getenv("GDPSO_STEM_MFMA");
        return e ? e[0] != '0' : true;
    }();
    const bool use_chain = L.chain != nullptr && single_stream;
    f32x4 *const ctl_zero = use_chain ? (f32x4 *)(ws + L.chain_ctl) : nullptr;
    const int ctl_quads = (int)(chain_ctl_bytes() / 16);
    const int mps = (a->side + 6 + 3) / 4 * 4;
    // the MFMA stem keeps a band of pooled rows' convolution rows in LDS: all rows when they fit beside the padded
    // image (28x28 inputs: 57 KB), else the largest band that fits GDPSO_STEM_LDS_KB
    auto stem_lds_of = [&](int band) {
        const int crows = band >= a->hp ? a->ho : 2 * band + 1;
        return ((size_t)(a->side + 6) * mps + (size_t)((crows * a->ho + 15) / 16 * 16) * 68) * sizeof(float);
    };
    static const size_t stem_lds_max = [] {
        const char *e = getenv("GDPSO_STEM_LDS_KB");
        size_t kb = e ? (size_t)atoi(e) : 100;  // (64x64 inputs: bands of 4 pooled rows; measured 64 / 82 / 100 / 152: 0.719 / 0.717 / 0.715 / 0.714 ms per iteration of 256)
        if (kb > 64) {
            hipError_t e1 = hipFuncSetAttribute((const void *)stem_mfma_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            hipError_t e2 = hipFuncSetAttribute((const void *)stem_mfma_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            hipError_t e3 = hipFuncSetAttribute((const void *)stem_mfma_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) kb = 64;
            (void)hipGetLastError();
        }
        return std::min<size_t>(kb, 152) * 1024;
    }();
    int mfma_band = a->hp;
    while (mfma_band > 1 && stem_lds_of(mfma_band) > stem_lds_max) --mfma_band;
    const size_t mfma_smem = stem_lds_of(mfma_band);
    int rc;
    if (L.resident) {
        // stem and first stages in one launch; its output lies where the tiled kernels would have left it (in x
        // after an even number of blocks, else in y)
        if (L.chain_first % 2 == 1) std::swap(x, y);
        if ((rc = launch_resident_stage(a->resident, stream, n, L.resident_images, a->num_cus, L.resident_second, img01, x, ctl_zero, ctl_quads))) return rc;
    } else if (stem_mfma && mfma_smem <= stem_lds_max) {
        GDPSO_WITH_DTYPE(a->dtype, hipLaunchKernelGGL(stem_mfma_kernel<E>, dim3(n), dim3(512), mfma_smem, stream, img01, a->stem_w,
                                                      a->stem_scale, a->stem_shift, a->side, a->ho, a->hp, mps, mfma_band,
                                                      fast_div_for((unsigned)a->ho), fast_div_for((unsigned)a->hp), (E *)x, ctl_zero, ctl_quads));
    } else {
        const int rows = 2 * a->stem_band + 1;
        const size_t smem = ((size_t)(a->side + 6) * a->stem_ps + (size_t)rows * a->ho * 64) * sizeof(float);
        dim3 grid(n, ceil_div(a->hp, a->stem_band));
        GDPSO_WITH_DTYPE(a->dtype, hipLaunchKernelGGL(stem_kernel<E>, grid, dim3(512), smem, stream, img01, a->stem_w, a->stem_scale,
                                                      a->stem_shift, a->side, a->ho, a->hp, a->stem_band, a->stem_ps, (E *)x, ctl_zero, ctl_quads));
    }
    if ((rc = conv_splitk_prepare(stream, sk))) return rc;
    for (size_t bi = 0; bi < L.blocks.size(); ++bi) {
        const Bottleneck &b = L.blocks[bi];
        if (L.resident && (int)bi < L.chain_first) continue;
        if (use_chain && (int)bi >= L.chain_first) {
            if ((int)bi == L.chain_first && (rc = launch_chain(L.chain, stream, workspace, /*ctl_cleared=*/true))) return rc;
            x = ws + L.chain_out;  // (the remaining blocks are all part of the chain)
            continue;
        }
        const void *res = x;
        if (b.has_ds) {  // identity_downsample and conv1 read the same input: one launch when their shapes allow
            if ((rc = launch_conv_pair(b.ds, b.c1, stream, n, x, sc, t1, sk))) return rc;
            res = sc;
        } else {
            if ((rc = launch_conv(b.c1, stream, n, x, nullptr, t1, sk))) return rc;
        }
        if ((rc = launch_conv_then_1x1(b.c2, b.c3, stream, n, t1, t2, res, y, sk))) return rc;
        std::swap(x, y);
    }
    GDPSO_WITH_DTYPE(a->dtype, hipLaunchKernelGGL(head_kernel<E>, dim3(n), dim3(256), 0, stream, (const E *)x, a->feat_hw, a->feat_c,
                                                  a->fc_w, a->fc_b, a->n_class, logits, fitness, class_index, class_per_image, mode,
                                                  threshold, eps, mse, w_ass, w_rec));
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

extern "C" int gdpso_assessor_forward(const gdpso_assessor *a, void *stream, int n, const float *img01, float *logits,
                                      float *fitness, int class_index, int mode, float threshold, float eps,
                                      const float *mse, float w_ass, float w_rec, void *workspace, size_t workspace_bytes) {
    return assessor_forward_impl(a, stream, n, img01, logits, fitness, class_index, nullptr, mode, threshold, eps, mse, w_ass, w_rec,
                                 workspace, workspace_bytes);
}

extern "C" int gdpso_assessor_forward_classes(const gdpso_assessor *a, void *stream, int n, const float *img01, float *logits,
                                              float *fitness, const int *class_per_image, int mode, float threshold, float eps,
                                              const float *mse, float w_ass, float w_rec, void *workspace,
                                              size_t workspace_bytes) {
    GDPSO_REQUIRE(class_per_image, "assessor_forward_classes: null class table");
    return assessor_forward_impl(a, stream, n, img01, logits, fitness, 0, class_per_image, mode, threshold, eps, mse, w_ass, w_rec,
                                 workspace, workspace_bytes);
}

// ============================================================================== encoder
struct gdpso_encoder {
    int enc_dim, features, out_pad;
    const float *w1, *b1;
    ConvLayer c2, c3;
    int num_cus;
    DeviceArena arena;
};

namespace {
struct EncLayout {
    size_t a1, a2, lat, splitk, total;
};
EncLayout enc_layout(const gdpso_encoder *e, int n, ConvLayer &c2, ConvLayer &c3) {
    c2 = e->c2;
    c3 = e->c3;
    plan_conv(c2, n, e->num_cus);
    plan_conv(c3, n, e->num_cus);
    EncLayout L;
    size_t off = 0;
    L.a1 = off; off += align_up((size_t)n * 14 * 14 * e->features * 4);
    L.a2 = off; off += align_up((size_t)n * 7 * 7 * 2 * e->features * 4);
    L.lat = off; off += align_up((size_t)n * e->out_pad * 4);
    L.splitk = off; off += align_up(std::max(conv_splitk_bytes(c2, n), conv_splitk_bytes(c3, n)));
    L.total = off;
    return L;
}
}  // namespace

extern "C" int gdpso_encoder_create(gdpso_encoder **out, int enc_dim, int features_e, const float *const *t, int n_tensors) {
    GDPSO_REQUIRE(out && t, "encoder_create: null argument");
    GDPSO_REQUIRE(n_tensors == 6 || n_tensors == 10, "encoder_create: expected 6 tensors (or 10 with BatchNorm)");
    GDPSO_REQUIRE(enc_dim > 0 && features_e >= 32 && features_e % 32 == 0, "encoder_create: features_e must be a multiple of 32");
    auto e = new gdpso_encoder();
    e->enc_dim = enc_dim;
    e->features = features_e;
    e->out_pad = (enc_dim + 63) / 64 * 64;
    e->num_cus = device_cus();
    const int c1 = features_e, c2 = 2 * features_e;
    int rc;
    std::vector<float> w1((size_t)16 * c1), b1(t[1], t[1] + c1);
    for (int co = 0; co < c1; ++co)
        for (int p = 0; p < 16; ++p) w1[(size_t)p * c1 + co] = t[0][(size_t)co * 16 + p];
    if ((rc = e->arena.upload(w1, &e->w1)) || (rc = e->arena.upload(b1, &e->b1))) { delete e; return rc; }
    std::vector<float> packed, ones, bias;
    make_conv(e->c2, t[2], c2, c1, 4, 4, 2, 1, 14, 14, ACT_LEAKY02, packed);
    if (n_tensors == 10) {
        fold_bn(c2, t[3], t[6], t[7], t[8], t[9], ones, bias);  // eval-mode BatchNorm after the second convolution
    } else {
        ones.assign(c2, 1.f);
        bias.assign(t[3], t[3] + c2);
    }
    if ((rc = upload_conv(e->arena, e->c2, packed, ones, bias))) { delete e; return rc; }
    make_conv(e->c3, t[4], enc_dim, c2, 7, 7, 2, 0, 7, 7, ACT_NONE, packed, 0, e->out_pad);
    ones.assign(e->out_pad, 1.f);
    bias.assign(e->out_pad, 0.f);
    std::copy(t[5], t[5] + enc_dim, bias.begin());
    if ((rc = upload_conv(e->arena, e->c3, packed, ones, bias))) { delete e; return rc; }
    *out = e;
    return GDPSO_OK;
}

extern "C" void gdpso_encoder_destroy(gdpso_encoder *e) { delete e; }
extern "C" size_t gdpso_encoder_workspace_bytes(const gdpso_encoder *e, int n) {
    if (!e || n <= 0) return 0;
    ConvLayer a, b;
    return enc_layout(e, n, a, b).total;
}

extern "C" int gdpso_encoder_forward(const gdpso_encoder *e, void *stream_, int n, const float *img, float *latent,
                                     void *workspace, size_t workspace_bytes) {
    GDPSO_REQUIRE(e && img && latent && workspace, "encoder_forward: null argument");
    GDPSO_REQUIRE(n > 0, "encoder_forward: n must be positive");
    hipStream_t stream = (hipStream_t)stream_;
    {
        int rc0 = chain_note_stream(stream, nullptr);  // (the assessor runs chained launches only beside a single stream)
        if (rc0) return rc0;
    }
    ConvLayer c2, c3;
    const EncLayout L = enc_layout(e, n, c2, c3);
    if (workspace_bytes < L.total) return fail(GDPSO_ERR_WORKSPACE, "encoder_forward: workspace too small");
    char *ws = (char *)workspace;
    float *a1 = (float *)(ws + L.a1), *a2 = (float *)(ws + L.a2), *lat = (float *)(ws + L.lat), *sk = (float *)(ws + L.splitk);
    hipLaunchKernelGGL(enc_first_kernel, dim3(n), dim3(256), 30 * 30 * sizeof(float), stream, img, e->w1, e->b1, 28, 14, e->features, a1);
    int rc;
    if ((rc = conv_splitk_prepare(stream, sk))) return rc;
    if ((rc = launch_conv(c2, stream, n, a1, nullptr, a2, sk))) return rc;
    if ((rc = launch_conv(c3, stream, n, a2, nullptr, lat, sk))) return rc;
    hipLaunchKernelGGL(slice_cols_kernel, dim3(ceil_div((long)n * e->enc_dim, 256)), dim3(256), 0, stream, lat, e->out_pad, n,
                       e->enc_dim, latent);
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}
