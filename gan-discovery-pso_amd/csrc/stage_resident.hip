// The assessor's stem and first two ResNet stages with their activations resident in LDS (gfx950, 28x28 images).
//
// The stem (reference src/pso/util_cnn.py:85-88, 103-106) and the bottleneck blocks of the first two stages
// (:94-95, 147-157, 160-190: 1x1 -> 3x3 -> 1x1 convolutions with BatchNorm, the projection shortcut of a stage's
// first block, residual add, ReLU) work on maps of 14x14, 7x7 and 4x4 pixels when the assessor sees the
// reference's 28x28 images.  Rows of different images never interact and an image's maps are small: 49 x 256
// floats for a block's output in the first stage, 16 x 512 in the second.  One workgroup therefore takes TWO
// images (98 rows = 7 blocks of 16 on the 7x7 maps, 32 rows = 2 blocks on the 4x4; 150 KB of the CU's 160 KB of
// LDS), walks the stem and every layer of both stages for them and writes only the second stage's output: no
// intermediate touches HBM, nothing is handed between workgroups, only workgroup barriers stand between
// layers, and the launch is n / 2 workgroups (one per CU at the 512 particles of the headline configuration)
// where the tiled kernels need a launch per layer (tiles of 64 rows: 1.5 per CU in the first stage) or the
// chained launch a hand-off between workgroups per layer.  Smaller batches give a workgroup ONE image (4 and 1 row
// blocks) so that every image still has a CU of its own; see resident_stage_images.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation).  The A operand of every
// layer is read from LDS where the previous layer's epilogue left it (rows padded by 16 bytes: conflict-free
// 16-byte fragment reads); the B operand comes straight from global memory (the two stages' weights are 5.7 MB,
// resident in every L2) in fragment order: the host packs a layer as [column block of 16][k group of 16][lane][4],
// so a wave's fragment load is one contiguous KiB, prefetched three k groups ahead (six where a group is only
// eight MFMAs).  Lane l of a wave holds A[row l & 15][k] and B[k][column l & 15] for the four
// k = 16 j + 4 (l >> 4) + s of group j: one 16-byte read of each feeds four MFMAs.  A 3x3 convolution reads its
// nine taps as nine row offsets into the same LDS map; a tap that leaves the image reads a row of zeros; the
// stride-2 layers of the second stage's first block pick their rows the same way.
//
// Eight waves.  First stage: a 64-column layer gives wave w column block w & 3 and every second row block (4 or 3
// of the 7); a 256-column layer gives it column blocks 2w, 2w + 1 and all seven row blocks.  Second stage: a
// 128-column layer gives it column block w, a 512-column layer blocks 4w .. 4w + 3, both row blocks each.  The sums
// run over k in another order than the tiled kernels' (groups of 16 interleaved by 4), so the two paths agree
// to fp32 rounding, not bit for bit; a row's result does not depend on the batch it arrives in nor on its place
// in the workgroup.  What bounds the scheme is the weights' way into the CU: a workgroup reads every weight of a
// layer for its few rows, 33 GB/s per CU at the second stage's 32 rows; the third stage (2x2 maps: 8 rows) would
// need 134 GB/s per CU and stays with the chained launch.
//
// 16-bit storage (bf16 / fp16): the same kernel with v_mfma_f32_16x16x32_bf16 / _f16; a 16-byte fragment is then
// eight consecutive k of ONE instruction (a k group is 32 wide), LDS rows are padded by 8 elements, accumulation,
// BatchNorm, residual add and ReLU stay fp32 and a layer's result is rounded once on its way to LDS, as the
// tiled kernels round it on its way to memory.
#include <hip/hip_ext.h>

#include <cstdlib>
#include <cstring>

#include "common.h"

namespace gdpso {

constexpr int RS_PIX = 49, RS_SIDE = 7, RS_IMAGES = 2, RS_ROWS = RS_PIX * RS_IMAGES;  // 98 rows = 7 blocks of 16
constexpr int RS_ZERO_ROW = RS_ROWS;  // row 98 of the first intermediate: zeros (taps outside the image)
constexpr int RS2_SIDE = 4, RS2_PIX = 16, RS2_ROWS = RS2_PIX * RS_IMAGES;  // second stage: 4x4 maps, 32 rows = 2 blocks of 16
// The stem's scratch, floats whatever the element type: convolution images [2][208][68], padded images [2][34][36],
// staged weights [49][80].
constexpr int ST_SIDE = 28, ST_HO = 14, ST_NPX = ST_HO * ST_HO, ST_NBLK = (ST_NPX + 15) / 16, ST_PROWS = ST_SIDE + 6, ST_PS = 36;
constexpr int ST_LDC = 68, ST_LDW = 80, ST_KSTEPS = 13, ST_IMG_FLOATS = ST_PROWS * ST_PS;
constexpr size_t ST_CONV_BYTES = (size_t)RS_IMAGES * ST_NBLK * 16 * ST_LDC * 4, ST_IN_BYTES = (size_t)(RS_IMAGES * ST_IMG_FLOATS + 49 * ST_LDW) * 4;

// LDS layout of one element type.  X: the block's input / output [98][256 + pad]; T1: conv1's output [99][64 + pad]
// (row 98 zeros); T2: the pooled stem output, then conv2's output [99][64 + pad].  Rows are padded by 16 bytes:
// the 16-byte fragment reads of eight consecutive rows then fall into different banks.  The stem's convolution
// images lie over X and T1 (and must end ahead of T2, which the pooling writes while it reads them); its padded
// images and staged weights lie in T2 where T2 is large enough (fp32: the pooling overwrites them when they are
// dead) and behind T2 otherwise.
template <typename E>
struct RsLds {
    static constexpr int VEC = 16 / (int)sizeof(E);     // elements of a 16-byte fragment
    static constexpr int KG = is_f32<E>() ? 16 : 32;     // k of one group: four fp32 MFMAs of 4, or one 16-bit MFMA of 32
    static constexpr int LDX = 256 + VEC, LDT = 64 + VEC;
    static constexpr size_t X_BYTES = (size_t)RS_ROWS * LDX * sizeof(E), T_BYTES = (size_t)(RS_ROWS + 1) * LDT * sizeof(E);
    static constexpr size_t T1_OFF = X_BYTES;
    static constexpr size_t T2_OFF = X_BYTES + T_BYTES > ST_CONV_BYTES ? X_BYTES + T_BYTES : ST_CONV_BYTES;
    static constexpr size_t ST_IN_OFF = T_BYTES >= ST_IN_BYTES ? T2_OFF : T2_OFF + T_BYTES;
    static constexpr size_t BYTES1 = ST_IN_OFF + ST_IN_BYTES > T2_OFF + T_BYTES ? ST_IN_OFF + ST_IN_BYTES : T2_OFF + T_BYTES;
    // Second stage (4x4 maps, 32 rows): U1 = conv1's output [99][128 + pad] behind X (98 rows of the 7x7 maps in the
    // first block, 32 afterwards; row 98 zeros); once the first block's conv1 and shortcut have read X, the block
    // output Y [32][512 + pad] and conv2's output U2 [32][128 + pad] lie where X was.
    static constexpr int LDU = 128 + VEC, LDY = 512 + VEC;
    static constexpr size_t U1_OFF = X_BYTES, U_BYTES = (size_t)(RS_ROWS + 1) * LDU * sizeof(E);
    static constexpr size_t Y_BYTES = (size_t)RS2_ROWS * LDY * sizeof(E), U2_OFF = Y_BYTES, U2_BYTES = (size_t)RS2_ROWS * LDU * sizeof(E);
    static constexpr size_t BYTES2 = U1_OFF + U_BYTES;
    static_assert(U2_OFF + U2_BYTES <= X_BYTES && U2_OFF % 16 == 0, "second-stage buffers exceed the first stage's map");
    static constexpr size_t BYTES = BYTES1 > BYTES2 ? BYTES1 : BYTES2;
    static_assert(ST_CONV_BYTES <= T2_OFF && ST_CONV_BYTES <= ST_IN_OFF, "the stem's convolution images overlap what is written while they are read");
    static_assert(T1_OFF % 16 == 0 && T2_OFF % 16 == 0 && ST_IN_OFF % 16 == 0 && BYTES <= 160 * 1024, "LDS layout");
};

struct RsLayer {
    // weights in fragment order, elements of the network's type: [cout / 16][K / KG][64 lanes][VEC],
    // K = taps * cin in (tap, channel) order, lane l = column l & 15, k = KG * group + VEC * (l >> 4) + (0 .. VEC - 1)
    const void *w;
    const float *scale, *shift;
};
struct RsArgs {
    const float *img01;  // [n][28][28]: the images the assessor sees
    void *out;           // [n][49][256], elements of the network's type
    int n, nblocks;
    const float *stem_w, *stem_scale, *stem_shift;  // [49][64] (tap-major), [64], [64]
    f32x4 *zero;         // 16-byte words this launch clears on its way (the control block of the chained launch behind it), or null
    int zero_quads;
    RsLayer ds, c1[RS_MAX_BLOCKS], c2[RS_MAX_BLOCKS], c3[RS_MAX_BLOCKS];
    // second stage (kernel instantiations with S2): its shortcut and blocks; `out` is then [n][16][512]
    int nblocks2;
    RsLayer ds2, d1[RS2_MAX_BLOCKS], d2[RS2_MAX_BLOCKS], d3[RS2_MAX_BLOCKS];
};

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
// One k group of one 16x16 block: a, b are the lane's 16-byte fragments.
template <typename E>
__device__ __forceinline__ f32x4 rs_mfma(const f32x4 &a, const f32x4 &b, f32x4 c) {
    if constexpr (is_f32<E>()) {
#pragma unroll
        for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], c, 0, 0, 0);
        return c;
    } else if constexpr (__is_same(E, bf16_t)) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
}

using global_f32x4 = const __attribute__((address_space(1))) f32x4;
using global_f32 = const __attribute__((address_space(1))) float;

// The matrix product of one layer for one wave: NRB row blocks (rb0, rb0 + RBSTEP, ...) x NCB column blocks
// (cb0, cb0 + 1, ...) of 16 x 16 outputs, CIN input channels per tap, TAPS = 1 (1x1) or 9 (3x3, padding 1), input
// maps of S_IN x S_IN pixels, output maps of S_OUT x S_OUT, stride STRIDE; two images a workgroup, rows in
// (image, y, x) order.  A: LDS rows of lda elements, row RS_ZERO_ROW all zeros when TAPS = 9 (taps outside the map
// read it).  ROWS_OUT: the output rows that exist (lanes of rows past them repeat the last one; their sums are
// dropped by the epilogue).  PF: k groups of B fragments in flight ahead of the one being multiplied.
template <typename E, int NRB, int RBSTEP, int NCB, int CIN, int TAPS, int S_IN, int S_OUT, int STRIDE, int ROWS_OUT, int PF>
__device__ __forceinline__ void rs_gemm(const E *__restrict__ A, const int lda, const void *w, const int rb0, const int cb0,
                                        const int lane, f32x4 (&acc)[NRB][NCB]) {
    constexpr int VEC = RsLds<E>::VEC, KG = RsLds<E>::KG, GPT = CIN / KG, J = TAPS * GPT;
    constexpr bool PLAIN = TAPS == 1 && STRIDE == 1;  // output row r reads input row r
    const int lrow = lane & 15, lk = lane >> 4;
    global_f32x4 *bp = (global_f32x4 *)w + (size_t)cb0 * J * 64 + lane;
    f32x4 bq[PF][NCB];
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < J) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) bq[p][c] = bp[(c * J + p) * 64];
        }
    // this lane's A row per row block: the input row of the window's centre, and the centre's coordinates
    int arow[NRB], ay[NRB], ax[NRB];
#pragma unroll
    for (int i = 0; i < NRB; ++i) {
        const int row = ROWS_OUT % 16 == 0 ? (rb0 + i * RBSTEP) * 16 + lrow : min((rb0 + i * RBSTEP) * 16 + lrow, ROWS_OUT - 1);
        if (PLAIN) {
            arow[i] = row;
            ay[i] = ax[i] = 0;
        } else {
            constexpr int PO = S_OUT * S_OUT;
            const int im = row >= PO ? 1 : 0, p = row - im * PO;
            const int oy = S_OUT == 4 ? p >> 2 : (p * 37) >> 8;  // p / S_OUT (p / 7 for p < 49)
            const int ox = p - oy * S_OUT;
            ay[i] = oy * STRIDE;
            ax[i] = ox * STRIDE;
            arow[i] = im * S_IN * S_IN + ay[i] * S_IN + ax[i];
        }
    }
    auto a_load = [&](int i, int j) -> f32x4 {
        const int tap = j / GPT, c0 = (j - tap * GPT) * KG;
        int row = arow[i];
        if (TAPS == 9) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const bool ok = (unsigned)(ay[i] + dy) < (unsigned)S_IN && (unsigned)(ax[i] + dx) < (unsigned)S_IN;
            row = ok ? row + dy * S_IN + dx : RS_ZERO_ROW;
        }
        return *(const f32x4 *)(A + row * lda + c0 + VEC * lk);
    };
#pragma unroll
    for (int i = 0; i < NRB; ++i)
#pragma unroll
        for (int c = 0; c < NCB; ++c) acc[i][c] = 0.f;
    f32x4 aq[2][NRB];
#pragma unroll
    for (int i = 0; i < NRB; ++i) aq[0][i] = a_load(i, 0);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        // (the scheduler may not move anything across a group's border: left alone it sinks the fragment loads
        // to just ahead of their first use, which undoes the prefetch distance)
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < J) {
#pragma unroll
            for (int i = 0; i < NRB; ++i) aq[(j + 1) & 1][i] = a_load(i, j + 1);
        }
        f32x4 bcur[NCB];
#pragma unroll
        for (int c = 0; c < NCB; ++c) bcur[c] = bq[j % PF][c];
        if (j + PF < J) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) bq[j % PF][c] = bp[(c * J + j + PF) * 64];
        }
        if constexpr (is_f32<E>()) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < NRB; ++i)
#pragma unroll
                    for (int c = 0; c < NCB; ++c)
                        acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[j & 1][i][s], bcur[c][s], acc[i][c], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < NRB; ++i)
#pragma unroll
                for (int c = 0; c < NCB; ++c) acc[i][c] = rs_mfma<E>(aq[j & 1][i], bcur[c], acc[i][c]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// BatchNorm scale / shift of a wave's column blocks (lane: column l & 15); requested ahead of the product.
template <int NCB>
__device__ __forceinline__ void rs_affine(const RsLayer &L, int cb0, int lane, float (&sc)[NCB], float (&sh)[NCB]) {
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        sc[c] = ((global_f32 *)L.scale)[(cb0 + c) * 16 + (lane & 15)];
        sh[c] = ((global_f32 *)L.shift)[(cb0 + c) * 16 + (lane & 15)];
    }
}

// Epilogue: scale / shift, residual, ReLU, rounded to E on the way to LDS (O: rows of ldo elements).  A lane holds
// rows 4 (l >> 4) + r of column l & 15 of each block.  RES: 0 none, 1 the elements of O it overwrites, 2 `keep`
// (what rs_keep left in registers, same blocks).
template <typename E, int NRB, int RBSTEP, int NCB, int ROWS_OUT, int RES, bool RELU>
__device__ __forceinline__ void rs_store(const f32x4 (&acc)[NRB][NCB], const float (&sc)[NCB], const float (&sh)[NCB], E *O, const int ldo,
                                         const int rb0, const int cb0, const int lane, const f32x4 (&keep)[NRB][NCB]) {
    const int lrow = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int i = 0; i < NRB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (rb0 + i * RBSTEP) * 16 + 4 * lk + r;
            if (ROWS_OUT % 16 == 0 || row < ROWS_OUT) {
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    E *o = O + row * ldo + (cb0 + c) * 16 + lrow;
                    float v = acc[i][c][r] * sc[c] + sh[c];
                    if (RES == 1) v += (float)*o;
                    if (RES == 2) v += keep[i][c][r];
                    if (RELU) v = fmaxf(v, 0.f);
                    *o = (E)v;
                }
            }
        }
}

// A layer's result kept in registers for a later layer's residual (the projection shortcut of the second stage, whose
// blocks are those of the 1x1 layer that adds it): scale / shift, rounded to E as a stored result would be.
template <typename E, int NRB, int NCB>
__device__ __forceinline__ void rs_keep(const f32x4 (&acc)[NRB][NCB], const float (&sc)[NCB], const float (&sh)[NCB], f32x4 (&keep)[NRB][NCB]) {
#pragma unroll
    for (int i = 0; i < NRB; ++i)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) keep[i][c][r] = (float)(E)(acc[i][c][r] * sc[c] + sh[c]);
}

// One layer of the first stage for one wave (7x7 maps, 49 rows an image): product + epilogue.
template <typename E, int IMG, int NRB, int RBSTEP, int NCB, int CIN, int TAPS, bool RES, bool RELU>
__device__ __forceinline__ void rs_layer(const E *__restrict__ A, const int lda, const RsLayer L, const int rb0, const int cb0,
                                         E *O, const int ldo, const int lane) {
    float sc[NCB], sh[NCB];
    rs_affine<NCB>(L, cb0, lane, sc, sh);
    f32x4 acc[NRB][NCB];
    rs_gemm<E, NRB, RBSTEP, NCB, CIN, TAPS, RS_SIDE, RS_SIDE, 1, IMG * RS_PIX, 3>(A, lda, L.w, rb0, cb0, lane, acc);
    rs_store<E, NRB, RBSTEP, NCB, IMG * RS_PIX, RES ? 1 : 0, RELU>(acc, sc, sh, O, ldo, rb0, cb0, lane, acc);
}

// A layer of 64 output channels over the whole workgroup (8 waves): wave w takes column block w & 3 and half of the
// row blocks: of the seven of two images the even (w < 4: four) or odd ones (three), of the four of one image the
// first or the last two.
template <typename E, int IMG, int CIN, int TAPS>
__device__ __forceinline__ void rs_layer64(const E *A, int lda, const RsLayer L, E *O, int wave, int lane) {
    if constexpr (IMG == 2) {
        if (wave < 4)
            rs_layer<E, IMG, 4, 2, 1, CIN, TAPS, false, true>(A, lda, L, 0, wave & 3, O, RsLds<E>::LDT, lane);
        else
            rs_layer<E, IMG, 3, 2, 1, CIN, TAPS, false, true>(A, lda, L, 1, wave & 3, O, RsLds<E>::LDT, lane);
    } else {
        rs_layer<E, IMG, 2, 1, 1, CIN, TAPS, false, true>(A, lda, L, 2 * (wave >> 2), wave & 3, O, RsLds<E>::LDT, lane);
    }
}

// A 1x1 layer of 64 -> 256 channels: wave w takes column blocks 2w, 2w + 1 and every row block.
template <typename E, int IMG, bool RES, bool RELU>
__device__ __forceinline__ void rs_layer256(const E *A, const RsLayer L, E *X, int wave, int lane) {
    rs_layer<E, IMG, (IMG * RS_PIX + 15) / 16, 1, 2, 64, 1, RES, RELU>(A, RsLds<E>::LDT, L, 0, 2 * wave, X, RsLds<E>::LDX, lane);
}

// The stem for the workgroup's two images: Conv2d(1 -> 64, k7, s2, p3, no bias) + BatchNorm + ReLU + MaxPool2d(3, s2, p1)
// (reference src/pso/util_cnn.py:85-88, 103-106), 28x28 -> 14x14 -> 7x7.  Per image the convolution is a 196 x 64 x 49
// product on v_mfma_f32_16x16x4_f32 with the operand order of stem_mfma_kernel (nets.hip; bit-identical to it): A
// gathered from the zero-padded image in LDS, lane l holding A[row l & 15][k = 4 t + (l >> 4)] and B[that k][column
// l & 15] at step t, K padded to 52 with zero weights.  The convolution images (2 x 208 rows x 68 floats) lie where
// the 256-channel map and the first intermediate will be; padded images and staged weights where the pooled result
// goes once they are dead (see RsLds).
template <typename E, int IMG>
__device__ __forceinline__ void rs_stem(const RsArgs &a, char *smem, E *T2, int img0, int nimg, int tid) {
    const int lane = tid & 63, wave = tid >> 6, lrow = lane & 15, lk = lane >> 4;
    float *s_conv = (float *)smem;                            // [2][208][68]
    float *s_img = (float *)(smem + RsLds<E>::ST_IN_OFF);     // [2][34][36], zero border of 3
    float *s_w = s_img + RS_IMAGES * ST_IMG_FLOATS;           // [49][80]
    for (int i = tid; i < IMG * ST_IMG_FLOATS; i += 512) {
        const int im = i / ST_IMG_FLOATS, q = i - im * ST_IMG_FLOATS, row = q / ST_PS, col = q - row * ST_PS;
        const int y = row - 3, x = col - 3;
        float v = 0.f;
        if (im < nimg && (unsigned)y < (unsigned)ST_SIDE && (unsigned)x < (unsigned)ST_SIDE)
            v = ((global_f32 *)a.img01)[(size_t)(img0 + im) * ST_SIDE * ST_SIDE + y * ST_SIDE + x];
        s_img[i] = v;
    }
    for (int i = tid; i < 49 * 16; i += 512) {
        const int k = i >> 4, c = (i & 15) * 4;
        *(f32x4 *)&s_w[k * ST_LDW + c] = *(global_f32x4 *)&a.stem_w[k * 64 + c];
    }
    float sc[4], sh[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        sc[cb] = ((global_f32 *)a.stem_scale)[cb * 16 + lrow];
        sh[cb] = ((global_f32 *)a.stem_shift)[cb * 16 + lrow];
    }
    __syncthreads();
    int koff[ST_KSTEPS];
    float bw[ST_KSTEPS][4];
#pragma unroll
    for (int t = 0; t < ST_KSTEPS; ++t) {
        const int k = 4 * t + lk;
        const int kk = k < 49 ? k : 0;
        koff[t] = (kk / 7) * ST_PS + kk % 7;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) bw[t][cb] = k < 49 ? s_w[k * ST_LDW + cb * 16 + lrow] : 0.f;
    }
    for (int b = wave; b < IMG * ST_NBLK; b += 8) {
        const int im = b >= ST_NBLK ? 1 : 0, bb = b - im * ST_NBLK;
        const int px = min(bb * 16 + lrow, ST_NPX - 1);  // rows past the image repeat the last pixel; they are never read back
        const int y = px / ST_HO, x = px - y * ST_HO;
        const float *base = s_img + im * ST_IMG_FLOATS + (2 * y) * ST_PS + 2 * x;  // padded coordinates: +3 - 3
        float av[ST_KSTEPS];
#pragma unroll
        for (int t = 0; t < ST_KSTEPS; ++t) av[t] = base[koff[t]];
        f32x4 acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = 0.f;
#pragma unroll
        for (int t = 0; t < ST_KSTEPS; ++t)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bw[t][cb], acc[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s_conv[(b * 16 + 4 * lk + r) * ST_LDC + cb * 16 + lrow] = fmaxf(acc[cb][r] * sc[cb] + sh[cb], 0.f);
    }
    __syncthreads();
    // MaxPool2d(3, s2, p1) into the rows of the first block's input; consecutive threads take consecutive channels
    for (int e = tid; e < IMG * RS_PIX * 64; e += 512) {
        const int c = e & 63, pp = e >> 6, im = pp >= RS_PIX ? 1 : 0, p = pp - im * RS_PIX, pr = (p * 37) >> 8, pc = p - pr * RS_SIDE;
        float m = -INFINITY;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int cy = 2 * pr + dy;
            if ((unsigned)cy >= (unsigned)ST_HO) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int cx = 2 * pc + dx;
                if ((unsigned)cx >= (unsigned)ST_HO) continue;
                m = fmaxf(m, s_conv[(im * ST_NBLK * 16 + cy * ST_HO + cx) * ST_LDC + c]);
            }
        }
        T2[pp * RsLds<E>::LDT + c] = (E)m;
    }
}

// The second stage for the workgroup's two images, from the first stage's output in X (reference
// src/pso/util_cnn.py:95, 147-157: bottlenecks of 128 -> 128 -> 512 channels, the first with stride 2 in its 3x3
// convolution and a strided 1x1 projection shortcut; 7x7 -> 4x4 maps).  32 rows = two row blocks: a wave takes one
// column block of a 128-column layer (k groups six ahead: its groups are short) or four of a 512-column layer, both
// row blocks each.  The shortcut's blocks are those its wave owns in conv3, so its result waits in registers.
template <typename E, int IMG>
__device__ __forceinline__ void rs_stage2(const RsArgs &a, char *smem, int wave, int lane, int tid) {
    using LY = RsLds<E>;
    constexpr int LDX = LY::LDX, LDU = LY::LDU, LDY = LY::LDY;
    constexpr int ROWS1 = IMG * RS_PIX, NRB1 = (ROWS1 + 15) / 16, ROWS2 = IMG * RS2_PIX, NRB2 = ROWS2 / 16;
    E *X = (E *)smem, *U1 = (E *)(smem + LY::U1_OFF), *Y = (E *)smem, *U2 = (E *)(smem + LY::U2_OFF);
    if (tid < LDU * (int)sizeof(E) / 16) *(f32x4 *)((char *)(U1 + RS_ZERO_ROW * LDU) + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 keep[NRB2][4];
    {
        float sc[1], sh[1];
        rs_affine<1>(a.d1[0], wave, lane, sc, sh);
        f32x4 acc[NRB1][1];
        rs_gemm<E, NRB1, 1, 1, 256, 1, RS_SIDE, RS_SIDE, 1, ROWS1, 3>(X, LDX, a.d1[0].w, 0, wave, lane, acc);
        rs_store<E, NRB1, 1, 1, ROWS1, 0, true>(acc, sc, sh, U1, LDU, 0, wave, lane, acc);
    }
    {
        float sc[4], sh[4];
        rs_affine<4>(a.ds2, 4 * wave, lane, sc, sh);
        f32x4 acc[NRB2][4];
        rs_gemm<E, NRB2, 1, 4, 256, 1, RS_SIDE, RS2_SIDE, 2, ROWS2, 3>(X, LDX, a.ds2.w, 0, 4 * wave, lane, acc);
        rs_keep<E, NRB2, 4>(acc, sc, sh, keep);
    }
    __syncthreads();
    for (int b = 0; b < a.nblocks2; ++b) {
        if (b > 0) {
            float sc[1], sh[1];
            rs_affine<1>(a.d1[b], wave, lane, sc, sh);
            f32x4 acc[NRB2][1];
            rs_gemm<E, NRB2, 1, 1, 512, 1, RS2_SIDE, RS2_SIDE, 1, ROWS2, 6>(Y, LDY, a.d1[b].w, 0, wave, lane, acc);
            rs_store<E, NRB2, 1, 1, ROWS2, 0, true>(acc, sc, sh, U1, LDU, 0, wave, lane, acc);
            __syncthreads();
        }
        {
            float sc[1], sh[1];
            rs_affine<1>(a.d2[b], wave, lane, sc, sh);
            f32x4 acc[NRB2][1];
            if (b == 0)
                rs_gemm<E, NRB2, 1, 1, 128, 9, RS_SIDE, RS2_SIDE, 2, ROWS2, 6>(U1, LDU, a.d2[b].w, 0, wave, lane, acc);
            else
                rs_gemm<E, NRB2, 1, 1, 128, 9, RS2_SIDE, RS2_SIDE, 1, ROWS2, 6>(U1, LDU, a.d2[b].w, 0, wave, lane, acc);
            rs_store<E, NRB2, 1, 1, ROWS2, 0, true>(acc, sc, sh, U2, LDU, 0, wave, lane, acc);
        }
        __syncthreads();
        {
            float sc[4], sh[4];
            rs_affine<4>(a.d3[b], 4 * wave, lane, sc, sh);
            f32x4 acc[NRB2][4];
            rs_gemm<E, NRB2, 1, 4, 128, 1, RS2_SIDE, RS2_SIDE, 1, ROWS2, 3>(U2, LDU, a.d3[b].w, 0, 4 * wave, lane, acc);
            if (b == 0)
                rs_store<E, NRB2, 1, 4, ROWS2, 2, true>(acc, sc, sh, Y, LDY, 0, 4 * wave, lane, keep);
            else
                rs_store<E, NRB2, 1, 4, ROWS2, 1, true>(acc, sc, sh, Y, LDY, 0, 4 * wave, lane, keep);
        }
        __syncthreads();
    }
}

// S2: the second stage behind the first, `out` = its output [n][16][512]; else `out` = the first stage's [n][49][256].
// IMG: images a workgroup takes, 2 or 1 (batches that leave half the CUs without a workgroup otherwise: a
// workgroup's time goes with its row blocks, 7 -> 4 on the 7x7 maps, 2 -> 1 on the 4x4; same LDS layout, same bits).
template <typename E, bool S2, int IMG>
__global__ __launch_bounds__(512) void stage_resident_kernel(const RsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using LY = RsLds<E>;
    constexpr int LDX = LY::LDX, LDT = LY::LDT;
    E *X = (E *)smem;                    // [98][LDX]: the block's input / output (256 channels)
    E *T1 = (E *)(smem + LY::T1_OFF);    // [99][LDT]: conv1's output, row 98 zeros
    E *T2 = (E *)(smem + LY::T2_OFF);    // [99][LDT]: the stem's (pooled) output, then conv2's output
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img0 = blockIdx.x * IMG;
    const int nimg = min(IMG, a.n - img0);
    if (a.zero != nullptr)
        for (int i = blockIdx.x * 512 + tid; i < a.zero_quads; i += gridDim.x * 512) a.zero[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    rs_stem<E, IMG>(a, smem, T2, img0, nimg, tid);
    __syncthreads();
    // (the zero row lies where the stem's convolution images were)
    if (tid < LDT * (int)sizeof(E) / 16) *(f32x4 *)((char *)(T1 + RS_ZERO_ROW * LDT) + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    // first block: conv1 and the projection shortcut read the same input
    rs_layer64<E, IMG, 64, 1>(T2, LDT, a.c1[0], T1, wave, lane);
    rs_layer256<E, IMG, false, false>(T2, a.ds, X, wave, lane);
    __syncthreads();
    rs_layer64<E, IMG, 64, 9>(T1, LDT, a.c2[0], T2, wave, lane);
    __syncthreads();
    rs_layer256<E, IMG, true, true>(T2, a.c3[0], X, wave, lane);
    __syncthreads();
    for (int b = 1; b < a.nblocks; ++b) {
        rs_layer64<E, IMG, 256, 1>(X, LDX, a.c1[b], T1, wave, lane);
        __syncthreads();
        rs_layer64<E, IMG, 64, 9>(T1, LDT, a.c2[b], T2, wave, lane);
        __syncthreads();
        rs_layer256<E, IMG, true, true>(T2, a.c3[b], X, wave, lane);
        __syncthreads();
    }
    constexpr int VEC = LY::VEC;
    if constexpr (S2) {
        rs_stage2<E, IMG>(a, smem, wave, lane, tid);
        constexpr int QPR = 512 / VEC;  // 16-byte words per row
        f32x4 *dst = (f32x4 *)((E *)a.out + (size_t)img0 * RS2_PIX * 512);
        for (int i = tid; i < nimg * RS2_PIX * QPR; i += 512) {
            const int r = i / QPR, c = (i - r * QPR) * VEC;
            dst[i] = *(const f32x4 *)((E *)smem + r * LY::LDY + c);
        }
    } else {
        constexpr int QPR = 256 / VEC;
        f32x4 *dst = (f32x4 *)((E *)a.out + (size_t)img0 * RS_PIX * 256);
        for (int i = tid; i < nimg * RS_PIX * QPR; i += 512) {
            const int r = i / QPR, c = (i - r * QPR) * VEC;
            dst[i] = *(const f32x4 *)(X + r * LDX + c);
        }
    }
}

// ------------------------------------------------------------------------------------ host side
static bool g_rs_enabled = [] {
    const char *e = getenv("GDPSO_STAGE1_RESIDENT");
    return e ? e[0] != '0' : true;
}();
// Largest batch (as planned: plan_batch) that takes the resident kernel.  Its workgroups are one per CU
// and run for the whole stage, so batches of several rounds are better served by the tiled kernels,
// whose large grids reach a higher fraction of the matrix peak.
static int g_rs_max_images = [] {
    const char *e = getenv("GDPSO_STAGE1_RESIDENT_MAX");
    return e ? atoi(e) : 1024;
}();
// Smallest batch that takes it: below, a handful of workgroups walk some sixty layers on a few CUs while the tiled
// kernels spread every layer over the device.
static int g_rs_min_images = [] {
    const char *e = getenv("GDPSO_STAGE1_RESIDENT_MIN");
    return e ? atoi(e) : 64;
}();

struct ResidentStage {
    int dtype;
    bool stage2;            // the second stage rides along (args.nblocks2 blocks)
    RsArgs args;            // layers filled in, in / out / n per launch
    // per image: nominal FLOPs, FLOPs issued to the matrix cores with two images (executed) or one (executed1) a
    // workgroup (row blocks of 16: 7 for 98 rows, 4 for 49); weight bytes once per launch; of the stem and the first
    // stage, and of both stages
    struct Work {
        double flops, executed, executed1, weight_bytes;
    } first, both;
};

// The second stage in the same launch (0: the resident launch ends with the first stage).
static bool g_rs_stage2 = [] {
    const char *e = getenv("GDPSO_STAGE2_RESIDENT");
    return e ? e[0] != '0' : true;
}();

static int g_rs2_min_images = [] {
    const char *e = getenv("GDPSO_STAGE2_RESIDENT_MIN");
    return e ? atoi(e) : 160;
}();
static int g_rs_force_images = 0;  // test aid (gdpso_resident_force_images)

bool resident_stage_enabled() { return g_rs_enabled; }
bool resident_stage_has_second(const ResidentStage *s) { return s != nullptr && s->stage2; }
void resident_stage_free(ResidentStage *s) { delete s; }

// Packs a torch [cout][cin][kh][kw] weight in the kernel's fragment order, rounded to `dtype` (nearest even).
static void rs_pack(const float *w, int cout, int cin, int taps, int dtype, std::vector<uint8_t> &out) {
    const int K = taps * cin, es = (int)dtype_size(dtype), vec = 16 / es, kg = dtype == DT_F32 ? 16 : 32, J = K / kg;
    out.resize((size_t)cout * K * es);
    for (int cb = 0; cb < cout / 16; ++cb)
        for (int j = 0; j < J; ++j)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < vec; ++s) {
                    const int co = cb * 16 + (lane & 15), k = kg * j + vec * (lane >> 4) + s;
                    const int tap = k / cin, ci = k - tap * cin;
                    const float v = w[((size_t)co * cin + ci) * taps + tap];
                    const size_t at = (((size_t)cb * J + j) * 64 + lane) * vec + s;
                    if (dtype == DT_F32) {
                        std::memcpy(&out[at * 4], &v, 4);
                    } else {
                        const uint16_t h = dtype == DT_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
                        std::memcpy(&out[at * 2], &h, 2);
                    }
                }
}

// weights: per block conv1, conv2, conv3 torch weights (w[3 b + i]) and the first block's shortcut;
// layers: the blocks' ConvLayers (their device scale / shift arrays are shared).
int resident_stage_new(DeviceArena &arena, int dtype, int nblocks, const float *const *w, const float *w_ds, const ConvLayer *const *c1,
                       const ConvLayer *const *c2, const ConvLayer *const *c3, const ConvLayer *ds, const float *stem_w,
                       const float *stem_scale, const float *stem_shift, int nblocks2, const float *const *w2, const float *w2_ds,
                       const ConvLayer *const *d1, const ConvLayer *const *d2, const ConvLayer *const *d3, const ConvLayer *ds2,
                       ResidentStage **out) {
    *out = nullptr;
    if (nblocks < 1 || nblocks > RS_MAX_BLOCKS) return GDPSO_OK;  // (not an error: the tiled kernels remain)
    if (nblocks2 > RS2_MAX_BLOCKS || !g_rs_stage2) nblocks2 = 0;
    hipError_t e = hipSuccess;
#define GDPSO_RS_ATTR(S2, IMG)                                                                                                         \
    if (e == hipSuccess)                                                                                                               \
        GDPSO_WITH_DTYPE(dtype, e = hipFuncSetAttribute((const void *)stage_resident_kernel<E, S2, IMG>,                                \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)RsLds<E>::BYTES));
    GDPSO_RS_ATTR(false, 1) GDPSO_RS_ATTR(false, 2) GDPSO_RS_ATTR(true, 1) GDPSO_RS_ATTR(true, 2)
#undef GDPSO_RS_ATTR
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return GDPSO_OK;
    }
    auto s = new ResidentStage();
    std::memset(&s->args, 0, sizeof(s->args));
    s->dtype = dtype;
    s->args.nblocks = nblocks;
    s->args.stem_w = stem_w;
    s->args.stem_scale = stem_scale;
    s->args.stem_shift = stem_shift;
    // the stem: 196 pixels x 64 channels x 49 taps per image (13 row blocks of 16, K padded to 52, on the matrix cores)
    ResidentStage::Work wk;
    wk.flops = 2.0 * ST_NPX * 64 * 49;
    wk.executed = wk.executed1 = 2.0 * ST_NBLK * 16 * 64 * 4 * ST_KSTEPS;
    wk.weight_bytes = 4.0 * 49 * 64;
    std::vector<uint8_t> packed;
    int rc = GDPSO_OK;
    auto add = [&](RsLayer &dst, const float *wt, const ConvLayer *L, int cin, int cout, int taps, int rows = 112, int rows1 = 64) {
        if (rc) return;
        rs_pack(wt, cout, cin, taps, dtype, packed);
        rc = arena.upload(packed.data(), packed.size(), &dst.w);
        dst.scale = L->scale;
        dst.shift = L->shift;
        wk.flops += L->flops_nominal;
        wk.executed += 2.0 * rows / 2 * cout * cin * taps;  // row blocks of 16 for two images: seven on the 7x7 maps, two on the 4x4
        wk.executed1 += 2.0 * rows1 * cout * cin * taps;   // ... for one image: four and one
        wk.weight_bytes += (double)dtype_size(dtype) * cout * cin * taps;
    };
    add(s->args.ds, w_ds, ds, 64, 256, 1);
    for (int b = 0; b < nblocks; ++b) {
        add(s->args.c1[b], w[3 * b], c1[b], b == 0 ? 64 : 256, 64, 1);
        add(s->args.c2[b], w[3 * b + 1], c2[b], 64, 64, 9);
        add(s->args.c3[b], w[3 * b + 2], c3[b], 64, 256, 1);
    }
    s->first = wk;
    s->stage2 = nblocks2 > 0;
    s->args.nblocks2 = nblocks2;
    if (s->stage2) {
        add(s->args.ds2, w2_ds, ds2, 256, 512, 1, 32, 16);
        for (int b = 0; b < nblocks2; ++b) {
            add(s->args.d1[b], w2[3 * b], d1[b], b == 0 ? 256 : 512, 128, 1, b == 0 ? 112 : 32, b == 0 ? 64 : 16);
            add(s->args.d2[b], w2[3 * b + 1], d2[b], 128, 128, 9, 32, 16);
            add(s->args.d3[b], w2[3 * b + 2], d3[b], 128, 512, 1, 32, 16);
        }
    }
    s->both = wk;
    if (rc) {
        delete s;
        return rc;
    }
    *out = s;
    return GDPSO_OK;
}

// How many images a workgroup of the resident launch should take for a batch planned as n_images on a device of
// num_cus CUs: 0 (the tiled kernels serve the batch better), 1 or 2.  A workgroup holds its CU for the whole run of
// layers, so the launch takes rounds x (time of one workgroup): one image a workgroup while that still leaves every
// image a CU of its own, two up to two images per CU; beyond, whole rounds of 2 x num_cus images, worth it only
// when the last round is nearly full (measured on MI355X, 256 CUs, fp32: 64 ... 256 images one a workgroup, ... 512
// two; 768 images = 1.5 rounds lose to the tiled kernels, 1024 = 2 rounds win).
int resident_stage_images(int n_images, int num_cus) {
    if (n_images < g_rs_min_images || n_images > g_rs_max_images) return 0;
    if (g_rs_force_images > 0) return g_rs_force_images;
    if (n_images <= num_cus) return 1;
    if (n_images <= 2 * num_cus) return 2;
    const long capacity = (long)ceil_div(ceil_div(n_images, 2), num_cus) * 2 * num_cus;
    return (long)n_images * 100 >= capacity * 85 ? 2 : 0;
}

// Whether the launch should go on through the second stage for a batch planned as n_images: with few workgroups
// its long layers on a handful of CUs lose to the chained launch, which spreads them over the device (measured:
// even at 128 images, +6 % at 192, +5 % at 256).
bool resident_stage_second(const ResidentStage *s, int n_images) { return s != nullptr && s->stage2 && n_images >= g_rs2_min_images; }

int launch_resident_stage(const ResidentStage *s, hipStream_t stream, int n_images, int images_per_wg, bool second, const float *img01,
                          void *out, void *zero, int zero_quads) {
    if (images_per_wg != 1 && images_per_wg != 2) return fail(GDPSO_ERR_ARG, "resident stage: one or two images a workgroup");
    if (second && !s->stage2) return fail(GDPSO_ERR_ARG, "resident stage: the handle holds no second stage");
    RsArgs a = s->args;
    a.img01 = img01;
    a.out = out;
    a.n = n_images;
    a.zero = (f32x4 *)zero;
    a.zero_quads = zero_quads;
    const dim3 grid((unsigned)ceil_div(n_images, images_per_wg));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const double out_elems = second ? (double)RS2_PIX * 512 : (double)RS_PIX * 256;
    const bool timed = conv_timing_slot(&e0, &e1);
    if (timed) {
        const ResidentStage::Work &w = second ? s->both : s->first;
        conv_timing_account(w.flops * n_images, (images_per_wg == 2 ? w.executed : w.executed1) * n_images,
                            w.weight_bytes + (double)n_images * (4.0 * ST_SIDE * ST_SIDE + (double)dtype_size(s->dtype) * out_elems));
    }
#define GDPSO_RS_LAUNCH(S2, IMG)                                                                                                            \
    if (timed) {                                                                                                                            \
        GDPSO_WITH_DTYPE(s->dtype, hipExtLaunchKernelGGL((stage_resident_kernel<E, S2, IMG>), grid, dim3(512), RsLds<E>::BYTES, stream, e0, \
                                                         e1, 0, a));                                                                        \
    } else {                                                                                                                                \
        GDPSO_WITH_DTYPE(s->dtype, hipLaunchKernelGGL((stage_resident_kernel<E, S2, IMG>), grid, dim3(512), RsLds<E>::BYTES, stream, a));    \
    }
    if (second) {
        if (images_per_wg == 2) {
            GDPSO_RS_LAUNCH(true, 2)
        } else {
            GDPSO_RS_LAUNCH(true, 1)
        }
    } else {
        if (images_per_wg == 2) {
            GDPSO_RS_LAUNCH(false, 2)
        } else {
            GDPSO_RS_LAUNCH(false, 1)
        }
    }
#undef GDPSO_RS_LAUNCH
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

}  // namespace gdpso

extern "C" int gdpso_resident_options(int stages, int max_images, int min_images) {
    using namespace gdpso;
    GDPSO_REQUIRE(stages >= 0 && stages <= 2, "resident_options: stages must be 0, 1 or 2");
    GDPSO_REQUIRE(max_images >= 0 && min_images >= 0, "resident_options: max_images / min_images must be 0 (keep) or a batch size");
    if (min_images > 0) g_rs_min_images = min_images;
    g_rs_enabled = stages != 0;
    g_rs_stage2 = stages == 2;
    if (max_images > 0) g_rs_max_images = max_images;
    return GDPSO_OK;
}

extern "C" int gdpso_resident_force_images(int images_per_workgroup) {
    using namespace gdpso;
    GDPSO_REQUIRE(images_per_workgroup >= 0 && images_per_workgroup <= 2, "resident_force_images: 0 (planner's choice), 1 or 2");
    g_rs_force_images = images_per_workgroup;
    return GDPSO_OK;
}
