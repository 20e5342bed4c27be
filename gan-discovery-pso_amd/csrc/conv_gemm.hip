// Implicit-GEMM convolution on NHWC fp32 activations with fp32-input MFMA (gfx950).
//
// Every convolution of the generator (reference src/utils/util_dcgan.py:133-139), the
// assessor's bottleneck blocks (src/pso/util_cnn.py:160-190) and the encoder
// (src/inverter/utils_ae/util_inverter.py:168-174) runs through this one kernel:
//   C[m][co] = sum_k A[m][k] * Wt[co][k],  m = (image, my, mx),  k = (tap, ci)
// A rows are gathered on the fly from the NHWC input (no im2col buffer): a K-tile of 32
// channels lies inside one tap because every Cin on the path is a multiple of 32, so a
// row's K-tile is 128 contiguous bytes or, when the tap falls in the padding, zeros.
//
// Arithmetic is v_mfma_f32_32x32x2_f32: exact fp32 products and fp32 accumulation (an
// fmaf chain), which is what keeps the results within 1e-4 of the fp32 CPU reference
// through ~50 layers.  The instruction is 16x slower than bf16 MFMA, so the kernel is
// MFMA-issue bound: one ds_read_b128 per operand feeds four MFMAs (256 cycles).
//
// Tile: BM x BN outputs per 256-thread workgroup (4 waves as 2x2, each (BM/2)x(BN/2)),
// BK = 32, two LDS buffers, register prefetch of the next K-tile, one barrier per K-tile.
// Small-M layers (late ResNet stages at 1x1..4x4) use 64x64 tiles plus split-K over
// blockIdx.z with fp32 partial slabs and a fused reduce+epilogue kernel.
#include "common.h"

namespace gdpso {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;
constexpr int LDA = BK + 4;  // 144-byte rows: ds_read_b128 of 16 consecutive rows is conflict-free
constexpr int PHASE_TABLE_INTS = 4 + MAX_TAPS;  // {ntaps, off_y, off_x, wt_offset, taps...}

struct ConvKArgs {
    const float *in;
    const float *wt;
    const float *scale;
    const float *shift;
    const float *res;
    float *out;
    float *part;       // split-K slabs [splits][pixels][cout], or null
    const int *table;  // [nphase][PHASE_TABLE_INTS]
    int M, hm, wm;
    int hin, win, cin;
    int cout, hout, wout;
    int in_stride, out_mul;
    int act, splits, ksplit;
    int tiles_n;
    long slab;  // elements per split-K slab
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_LEAKY02) return v > 0.f ? v : 0.2f * v;
    return v;
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvKArgs a) {
    constexpr int WM = BM / 64, WN = BN / 64;  // 32x32 MFMA blocks per wave, per dimension
    constexpr int PA = BM / 32, PB = BN / 32;  // loader passes (32 rows x 32 floats per pass)
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDA];
    float *sA = lds;
    float *sB = lds + 2 * BM * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    const int tm = blockIdx.x / a.tiles_n, tn = blockIdx.x - tm * a.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int ph = blockIdx.z / a.splits, sp = blockIdx.z - ph * a.splits;
    const int *tab = a.table + ph * PHASE_TABLE_INTS;
    const int ntaps = tab[0], off_y = tab[1], off_x = tab[2];
    const float *wt = a.wt + tab[3];
    const int ktot = ntaps * a.cin;
    const int kbeg = sp * a.ksplit;
    const int kend = min(ktot, kbeg + a.ksplit);

    // loader coordinates: 8 threads x 16 B cover one 32-float row
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;
    int a_iy0[PA], a_ix0[PA], a_img[PA];
    bool a_ok[PA];
    const int hw = a.hm * a.wm;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m0 + lrow + 32 * p;
        a_ok[p] = m < a.M;
        const int mm = a_ok[p] ? m : 0;
        const int n = mm / hw, r = mm - n * hw;
        const int my = r / a.wm, mx = r - my * a.wm;
        a_iy0[p] = my * a.in_stride;
        a_ix0[p] = mx * a.in_stride;
        a_img[p] = n * a.hin * a.win;
    }
    const float *wrow[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) wrow[p] = wt + (size_t)(n0 + lrow + 32 * p) * ktot + lcol;

    float4 ra[PA], rb[PB];
    auto gload = [&](int kt) {
        const int tap = kt / a.cin, ci0 = kt - tap * a.cin;
        const int packed = tab[4 + tap];
        const int dy = (int)(short)(packed & 0xffff), dx = packed >> 16;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int iy = a_iy0[p] + dy, ix = a_ix0[p] + dx;
            const bool ok = a_ok[p] && (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
            ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) ra[p] = *(const float4 *)(a.in + (size_t)(a_img[p] + iy * a.win + ix) * a.cin + ci0 + lcol);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) rb[p] = *(const float4 *)(wrow[p] + kt);
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PA; ++p) *(float4 *)&sA[buf * BM * LDA + (lrow + 32 * p) * LDA + lcol] = ra[p];
#pragma unroll
        for (int p = 0; p < PB; ++p) *(float4 *)&sB[buf * BN * LDA + (lrow + 32 * p) * LDA + lcol] = rb[p];
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kbeg < kend) {
        gload(kbeg);
        sstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (int kt = kbeg; kt < kend; kt += BK) {
        const bool more = kt + BK < kend;
        if (more) gload(kt + BK);
        const float *pa = sA + buf * BM * LDA + (wr * (BM / 2) + l31) * LDA + half * 4;
        const float *pb = sB + buf * BN * LDA + (wc * (BN / 2) + l31) * LDA + half * 4;
        // Within a K-tile the 32 k's are consumed in a permuted order: MFMA step s of group g
        // takes k = 8g + 4*half + s from both operands, so each lane's four steps are one
        // 16-byte LDS read.  The sum over k is the same set of products.
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 av[WM], bv[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) av[i] = *(const float4 *)(pa + i * 32 * LDA + g * 8);
#pragma unroll
            for (int j = 0; j < WN; ++j) bv[j] = *(const float4 *)(pb + j * 32 * LDA + g * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        const float x = s == 0 ? av[i].x : s == 1 ? av[i].y : s == 2 ? av[i].z : av[i].w;
                        const float y = s == 0 ? bv[j].x : s == 1 ? bv[j].y : s == 2 ? bv[j].z : bv[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i][j], 0, 0, 0);
                    }
            }
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // Epilogue.  Accumulator register r of lane l is C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]:
    // the channel sits on the lane, so each store instruction writes two 128-byte row segments.
    const bool direct = a.splits == 1;
    float *dst = direct ? a.out : a.part + (size_t)sp * a.slab;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = n0 + wc * (BN / 2) + j * 32 + l31;
        const float sc = a.scale[col], sh = a.shift[col];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= a.M) continue;
                size_t pix = (size_t)m;
                if (a.out_mul != 1 || a.hm != a.hout || a.wm != a.wout) {
                    const int n = m / hw, rr = m - n * hw;
                    const int my = rr / a.wm, mx = rr - my * a.wm;
                    pix = ((size_t)n * a.hout + my * a.out_mul + off_y) * a.wout + mx * a.out_mul + off_x;
                }
                const size_t o = pix * a.cout + col;
                float v = acc[i][j][r];
                if (direct) {
                    v = v * sc + sh;
                    if (a.res) v += a.res[o];
                    v = apply_act(v, a.act);
                }
                dst[o] = v;
            }
        }
    }
}

// Sums the split-K slabs and applies the epilogue.  One float4 (4 channels) per thread.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ part, int splits, long slab,
                                                            int cout, const float *__restrict__ scale,
                                                            const float *__restrict__ shift,
                                                            const float *__restrict__ res, float *__restrict__ out,
                                                            int act) {
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= slab) return;
    float4 s = *(const float4 *)(part + e);
    for (int k = 1; k < splits; ++k) {
        const float4 t = *(const float4 *)(part + (size_t)k * slab + e);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const int c = (int)(e % cout);
    const float4 sc = *(const float4 *)(scale + c), sh = *(const float4 *)(shift + c);
    float4 v = make_float4(s.x * sc.x + sh.x, s.y * sc.y + sh.y, s.z * sc.z + sh.z, s.w * sc.w + sh.w);
    if (res) {
        const float4 r = *(const float4 *)(res + e);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    v.x = apply_act(v.x, act); v.y = apply_act(v.y, act); v.z = apply_act(v.z, act); v.w = apply_act(v.w, act);
    *(float4 *)(out + e) = v;
}

// --------------------------------------------------------------------------- host side
// Optional per-launch timing of the convolution kernel with HIP events recorded on the
// launch stream (used by the benchmark to price the kernel against its roofline).
namespace {
struct ConvTiming {
    bool on = false;
    std::vector<hipEvent_t> events;
    size_t used = 0;
    double flops = 0.0;
} g_timing;
}  // namespace

static long max_ktot(const ConvLayer &L) {
    long k = 0;
    for (int p = 0; p < L.nphase; ++p) k = std::max<long>(k, (long)L.phase[p].ntaps * L.cin);
    return k;
}

// Chooses tile and split-K so that the launch fills the chip (2 workgroups per CU).
int plan_conv(ConvLayer &L, int n_images, int num_cus) {
    const long M = (long)n_images * L.hm * L.wm;
    const long target = 2L * num_cus;
    L.tile = 64;
    if (L.cout % 128 == 0) {
        const long t128 = (long)ceil_div(M, 128) * (L.cout / 128) * L.nphase;
        if (t128 >= (3 * target) / 4) L.tile = 128;
    }
    if (L.cout % L.tile != 0) return fail(GDPSO_ERR_ARG, "conv: cout must be a multiple of 64");
    const long tiles = (long)ceil_div(M, L.tile) * (L.cout / L.tile) * L.nphase;
    const long ktot = max_ktot(L);
    int splits = 1;
    if (tiles < target) {
        splits = (int)std::min<long>(ceil_div(target, tiles), 16);
        // keep at least 128 k per split so the prologue/epilogue stays amortised
        splits = (int)std::max<long>(1, std::min<long>(splits, ktot / 128));
    }
    L.splits = splits;
    return GDPSO_OK;
}

size_t conv_splitk_bytes(const ConvLayer &L, int n_images) {
    if (L.splits <= 1) return 0;
    return (size_t)L.splits * n_images * L.hout * L.wout * L.cout * sizeof(float);
}

int launch_conv(const ConvLayer &L, hipStream_t stream, int n_images, const float *in, const float *res,
                float *out, float *splitk_ws) {
    ConvKArgs a;
    a.in = in; a.wt = L.wt; a.scale = L.scale; a.shift = L.shift; a.res = res; a.out = out;
    a.part = L.splits > 1 ? splitk_ws : nullptr;
    a.table = L.table;
    a.M = n_images * L.hm * L.wm; a.hm = L.hm; a.wm = L.wm;
    a.hin = L.hin; a.win = L.win; a.cin = L.cin;
    a.cout = L.cout; a.hout = L.hout; a.wout = L.wout;
    a.in_stride = L.in_stride; a.out_mul = L.out_mul;
    a.act = L.act; a.splits = L.splits;
    const long ktot = max_ktot(L);
    a.ksplit = ceil_div(ceil_div(ktot, BK), L.splits) * BK;
    a.tiles_n = L.cout / L.tile;
    a.slab = (long)n_images * L.hout * L.wout * L.cout;
    if (L.cin % BK != 0) return fail(GDPSO_ERR_ARG, "conv: cin must be a multiple of 32");
    if (L.splits > 1 && !splitk_ws) return fail(GDPSO_ERR_WORKSPACE, "conv: split-K workspace missing");
    const int tiles_m = ceil_div(a.M, L.tile);
    dim3 grid(tiles_m * a.tiles_n, 1, L.nphase * L.splits);
    const bool timed = g_timing.on && g_timing.used + 2 <= g_timing.events.size();
    if (timed) GDPSO_HIP(hipEventRecord(g_timing.events[g_timing.used], stream));
    if (L.tile == 128)
        hipLaunchKernelGGL((conv_gemm_kernel<128, 128>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((conv_gemm_kernel<64, 64>), grid, dim3(256), 0, stream, a);
    if (timed) {
        GDPSO_HIP(hipEventRecord(g_timing.events[g_timing.used + 1], stream));
        g_timing.used += 2;
        g_timing.flops += L.flops_nominal * n_images;
    }
    if (L.splits > 1) {
        const long quads = a.slab / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ceil_div(quads, 256)), dim3(256), 0, stream, a.part, L.splits,
                           a.slab, L.cout, L.scale, L.shift, res, out, L.act);
    }
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

}  // namespace gdpso

extern "C" int gdpso_conv_timing_begin(int max_launches) {
    using namespace gdpso;
    GDPSO_REQUIRE(max_launches > 0, "conv_timing_begin: max_launches must be positive");
    while (g_timing.events.size() < (size_t)2 * max_launches) {
        hipEvent_t e;
        GDPSO_HIP(hipEventCreate(&e));
        g_timing.events.push_back(e);
    }
    g_timing.used = 0;
    g_timing.flops = 0.0;
    g_timing.on = true;
    return GDPSO_OK;
}

extern "C" int gdpso_conv_timing_end(double *total_ms, int *launches, double *flops) {
    using namespace gdpso;
    g_timing.on = false;
    double ms = 0.0;
    for (size_t i = 0; i + 1 < g_timing.used; i += 2) {
        GDPSO_HIP(hipEventSynchronize(g_timing.events[i + 1]));
        float t = 0.f;
        GDPSO_HIP(hipEventElapsedTime(&t, g_timing.events[i], g_timing.events[i + 1]));
        ms += t;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = (int)(g_timing.used / 2);
    if (flops) *flops = g_timing.flops;
    return GDPSO_OK;
}
