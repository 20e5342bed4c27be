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
// blocks) so that every image still has a CU of its own; larger ones run in rounds of workgroups, the last round, when
// it holds at most one image per CU, as a launch of its own with one image a workgroup; see resident_stage_images.
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

#include <cmath>
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
// (cb0, cb0 + 1, ...) of 16 x 16 outputs, CIN input channels per tap, TAPS = 1 (1x1), 9 (3x3, padding 1) or 4 (one
// sub-pixel phase of a transposed convolution k4 s2 p1: the taps (0, 0), (0, sx), (sy, 0), (sy, sx), sy / sx = -1 for the
// even output rows / columns of the phase, +1 for the odd ones), input
// maps of S_IN x S_IN pixels, output maps of S_OUT x S_OUT, stride STRIDE; two images a workgroup, rows in
// (image, y, x) order.  A: LDS rows of lda elements, row RS_ZERO_ROW all zeros when TAPS = 9 (taps outside the map
// read it).  ROWS_OUT: the output rows that exist (lanes of rows past them repeat the last one; their sums are
// dropped by the epilogue).  PF: k groups of B fragments in flight ahead of the one being multiplied.
// `rows` (the generator's second layer): the input row of every output row of the wave's blocks, for row blocks made of
// rows of one border class (-1: a padding row); with GSKIP the last two row blocks hold the rows whose second tap
// along y, resp. along x, leaves the map, and the k groups of those taps are left out for them.
template <typename E, int NRB, int RBSTEP, int NCB, int CIN, int TAPS, int S_IN, int S_OUT, int STRIDE, int ROWS_OUT, int PF, bool GSKIP = false>
__device__ __forceinline__ void rs_gemm(const E *__restrict__ A, const int lda, const void *w, const int rb0, const int cb0,
                                        const int lane, f32x4 (&acc)[NRB][NCB], const int sy = 0, const int sx = 0,
                                        const short *__restrict__ rows = nullptr) {
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
        int row = ROWS_OUT % 16 == 0 ? (rb0 + i * RBSTEP) * 16 + lrow : min((rb0 + i * RBSTEP) * 16 + lrow, ROWS_OUT - 1);
        if (rows != nullptr) {
            const int pr = rows[(rb0 + i * RBSTEP) * 16 + lrow];
            row = pr < 0 ? ROWS_OUT - 1 : pr;
        }
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
        if (TAPS == 4) {
            const int dy = (tap >> 1) ? sy : 0, dx = (tap & 1) ? sx : 0;
            const bool ok = (unsigned)(ay[i] + dy) < (unsigned)S_IN && (unsigned)(ax[i] + dx) < (unsigned)S_IN;
            row = ok ? row + dy * S_IN + dx : RS_ZERO_ROW;
        }
        return *(const f32x4 *)(A + row * lda + c0 + VEC * lk);
    };
#pragma unroll
    for (int i = 0; i < NRB; ++i)
#pragma unroll
        for (int c = 0; c < NCB; ++c) acc[i][c] = 0.f;
    // (GSKIP) whether row block i takes part in k group j
    auto live = [&](int i, int j) -> bool {
        if (!GSKIP) return true;
        const int tap = j / GPT;
        return !((i == NRB - 2 && (tap >> 1) == 1) || (i == NRB - 1 && (tap & 1) == 1));
    };
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
            for (int i = 0; i < NRB; ++i)
                if (live(i, j + 1)) aq[(j + 1) & 1][i] = a_load(i, j + 1);
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
                    if (live(i, j)) {
#pragma unroll
                        for (int c = 0; c < NCB; ++c)
                            acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[j & 1][i][s], bcur[c][s], acc[i][c], 0, 0, 0);
                    }
        } else {
#pragma unroll
            for (int i = 0; i < NRB; ++i)
                if (live(i, j)) {
#pragma unroll
                    for (int c = 0; c < NCB; ++c) acc[i][c] = rs_mfma<E>(aq[j & 1][i], bcur[c], acc[i][c]);
                }
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

// ------------------------------------------------------------------------------------ the generator
// The reference's generator (src/utils/util_dcgan.py:128-149: ConvTranspose2d(Z -> 128, k7) + BN + ReLU,
// ConvTranspose2d(128 -> 64, k4, s2, p1) + BN + ReLU, ConvTranspose2d(64 -> 1, k4, s2, p1) + tanh) and the per-image
// rescale (src/utils/util_data.py:129-135) the same way: a workgroup takes two images (or one), keeps the 7x7x128 map of
// the first layer (98 rows) and the 14x14x64 map of the second (392 rows) in LDS and writes only the images.  The
// second layer is four sub-pixel phases, each a 2x2-tap convolution of 98 rows x 64 columns with K = 512: wave w takes
// phase w >> 1 and column blocks 2 (w & 1), 2 (w & 1) + 1 with all seven row blocks, through rs_gemm (the A operand in LDS,
// weights from global memory in fragment order); its epilogue scatters a row to pixel (2 my + py, 2 mx + px) of the
// 14x14 map.  The first layer (latents of up to 16 dimensions) is the arithmetic of gen_first_kernel, the last layer,
// tanh, min / max and rescale that of gen_last_mfma_kernel (nets.hip; same operand order, same bits given the same
// 14x14 map); longer latents leave the first layer to the matrix-core GEMM and the kernel reads its output.
constexpr int GR_C1 = 128, GR_C2 = 64, GR_HIN = 14, GR_NIN = GR_HIN * GR_HIN, GR_SIDE = 28, GR_NPIX = GR_SIDE * GR_SIDE;
constexpr int GR_NCOL = RS_PIX * GR_C1, GR_MAX_Z = 16, GR_NBLK = (GR_NIN + 15) / 16;

template <typename E>
struct GrLds {
    static constexpr int VEC = 16 / (int)sizeof(E);
    static constexpr int LDA1 = GR_C1 + VEC, LDA2 = GR_C2 + VEC;
    static constexpr size_t A1_BYTES = (size_t)(RS_ROWS + 1) * LDA1 * sizeof(E), A2_OFF = A1_BYTES;
    static constexpr size_t A2_BYTES = (size_t)RS_IMAGES * GR_NIN * LDA2 * sizeof(E), ROWS_OFF = A2_OFF + A2_BYTES;
    static constexpr size_t BYTES = ROWS_OFF + 4 * 112 * sizeof(short);  // (behind the maps: the phases' row orders, see GrArgs::rows)
    // the last layer's products [2 x 208][16] floats, the two images [2][784] floats and the reduction slots lie where the
    // first map was (and, with 16-bit storage, run on into the second map, which is dead by the time they are written)
    static constexpr size_t SC_OFF = 0, SC_BYTES = (size_t)RS_IMAGES * GR_NBLK * 16 * 16 * 4, SIMG_OFF = SC_BYTES;
    static constexpr size_t SRED_OFF = SIMG_OFF + (size_t)RS_IMAGES * GR_NPIX * 4;
    static_assert(SC_BYTES <= A2_OFF && SRED_OFF + RS_IMAGES * 24 * 4 <= ROWS_OFF && BYTES <= 160 * 1024, "LDS layout");
};

struct GrArgs {
    const void *z;            // latents [n][zd], float64 or float32 (zd <= GR_MAX_Z), or null:
    const void *a1;           // ... the first layer's output [n][49][128], elements of the network's type
    int z_is_f64, zd, n;
    const float *w1t, *sc1, *sh1;  // first layer: [zd][6272] (column = pixel * 128 + channel), scale / shift [6272]
    const void *w3;                // second layer in fragment order: [4 phases][4 column blocks][512 / KG][64 lanes][VEC]
    const float *sc3, *sh3;        // [64]
    const short *rows;             // [4 phases][112]: the rows of a phase ordered by border class (two images a workgroup), -1 = padding
    const float *w5;               // last layer [16 kernel positions][64]
    float b5;
    float *img, *img01;            // [n][28][28]: tanh output (may be null), rescaled image
    const float *target;           // inverter variant: target images and the pixel MSE against them, or null
    float *mse;
};

template <typename E, int IMG>
__global__ __launch_bounds__(512) void gen_resident_kernel(const GrArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using LY = GrLds<E>;
    constexpr int VEC = LY::VEC, LDA1 = LY::LDA1, LDA2 = LY::LDA2, ROWS = IMG * RS_PIX, NRB = (ROWS + 15) / 16;
    E *A1 = (E *)smem, *A2 = (E *)(smem + LY::A2_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 15, lk = lane >> 4;
    const int img0 = blockIdx.x * IMG, nimg = min(IMG, a.n - img0);
    if (tid < LDA1 * (int)sizeof(E) / 16) *(f32x4 *)((char *)(A1 + RS_ZERO_ROW * LDA1) + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    short *s_rows = (short *)(smem + LY::ROWS_OFF);
    if (IMG == 2 && tid < 4 * 112) s_rows[tid] = a.rows[tid];
    // ---- first layer
    if (a.z != nullptr) {
        float *s_z = (float *)A2;  // [IMG][GR_MAX_Z]
        if (tid < IMG * GR_MAX_Z) {
            const int i = tid / GR_MAX_Z, k = tid % GR_MAX_Z, im = img0 + i;
            float v = 0.f;
            if (i < nimg && k < a.zd) v = a.z_is_f64 ? (float)((const double *)a.z)[(long)im * a.zd + k] : ((const float *)a.z)[(long)im * a.zd + k];
            s_z[tid] = (float)(E)v;  // (16-bit storage: the latent rounded as the matrix-core path rounds its A operand)
        }
        __syncthreads();
        constexpr int NQ = GR_NCOL / 4;  // 1568 quads of columns: three passes of the workgroup and 32 quads more
#pragma unroll
        for (int pass = 0; pass < (NQ + 511) / 512; ++pass) {
            const int q = tid + 512 * pass;
            if (512 * (pass + 1) > NQ && q >= NQ) break;
            const int col = q * 4;
            f32x4 w[GR_MAX_Z];
#pragma unroll
            for (int k = 0; k < GR_MAX_Z; ++k) {
                w[k] = 0.f;
                if (k < a.zd) w[k] = *(global_f32x4 *)(a.w1t + (size_t)k * GR_NCOL + col);
            }
            const f32x4 sc = *(global_f32x4 *)(a.sc1 + col), sh = *(global_f32x4 *)(a.sh1 + col);
#pragma unroll
            for (int i = 0; i < IMG; ++i) {
                f32x4 acc = 0.f;
#pragma unroll
                for (int j = 0; j < GR_MAX_Z; ++j) {
                    const int g = j >> 3, h = j & 1, st = (j >> 1) & 3, k = 8 * g + 4 * h + st;  // 0, 4, 1, 5, 2, 6, 3, 7, 8, 12, ...
                    if (k < a.zd) {
                        const float zk = s_z[i * GR_MAX_Z + k];
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = fmaf(zk, w[k][c], acc[c]);
                    }
                }
                f32x4 v = acc * sc + sh;
                E *o = A1 + (i * RS_PIX + col / GR_C1) * LDA1 + col % GR_C1;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (E)fmaxf(v[c], 0.f);
            }
        }
    } else {
        constexpr int QPR = GR_C1 / VEC;  // 16-byte words per row
        global_f32x4 *src = (global_f32x4 *)((const E *)a.a1 + (size_t)img0 * RS_PIX * GR_C1);
        for (int i = tid; i < ROWS * QPR; i += 512) {
            const int r = i / QPR, c = (i - r * QPR) * VEC;
            f32x4 v = 0.f;
            if (r < nimg * RS_PIX) v = src[i];
            *(f32x4 *)(A1 + r * LDA1 + c) = v;
        }
    }
    __syncthreads();
    // ---- second layer: four phases of 98 rows x 64 columns x K = 512
    {
        constexpr int J3 = 4 * GR_C1 / RsLds<E>::KG;
        const int ph = wave >> 1, py = ph >> 1, px = ph & 1, cb0 = 2 * (wave & 1);
        float sc[2], sh[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            sc[c] = ((global_f32 *)a.sc3)[(cb0 + c) * 16 + lrow];
            sh[c] = ((global_f32 *)a.sh3)[(cb0 + c) * 16 + lrow];
        }
        f32x4 acc[NRB][2];
        // two images: the rows of a phase in border classes (a.rows: interior rows in five blocks, then the rows whose
        // second tap along y leaves the map, then those whose second tap along x does), so that the last two blocks skip
        // those taps: 24 tap-blocks instead of 28.  One image has too few rows for that (4 blocks either way).
        const short *rows = IMG == 2 ? s_rows + ph * 16 * NRB : nullptr;
        rs_gemm<E, NRB, 1, 2, GR_C1, 4, RS_SIDE, RS_SIDE, 1, ROWS, 3, IMG == 2>(A1, LDA1, (const f32x4 *)a.w3 + (size_t)ph * 4 * J3 * 64, 0, cb0,
                                                                               lane, acc, py ? 1 : -1, px ? 1 : -1, rows);
#pragma unroll
        for (int i = 0; i < NRB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rows != nullptr ? rows[i * 16 + 4 * lk + r] : i * 16 + 4 * lk + r;
                if (row >= 0 && row < ROWS) {
                    const int im = row >= RS_PIX ? 1 : 0, p = row - im * RS_PIX, my = (p * 37) >> 8, mx = p - my * RS_SIDE;
                    E *o = A2 + (im * GR_NIN + (2 * my + py) * GR_HIN + 2 * mx + px) * LDA2 + cb0 * 16 + lrow;
#pragma unroll
                    for (int c = 0; c < 2; ++c) o[c * 16] = (E)fmaxf(acc[i][c][r] * sc[c] + sh[c], 0.f);
                }
            }
    }
    __syncthreads();
    // ---- last layer: C[input pixel][kernel position] = A2[pixel][channel] * W5[channel][position] per image
    float *s_c = (float *)(smem + LY::SC_OFF), *s_img = (float *)(smem + LY::SIMG_OFF), *s_red = (float *)(smem + LY::SRED_OFF);
    {
        f32x4 bfrag[GR_C2 / 16];
#pragma unroll
        for (int j = 0; j < GR_C2 / 16; ++j) bfrag[j] = *(global_f32x4 *)&a.w5[(size_t)lrow * GR_C2 + 16 * j + 4 * lk];
        f32x4 accs[(IMG * GR_NBLK + 7) / 8];
#pragma unroll
        for (int t = 0; t < (IMG * GR_NBLK + 7) / 8; ++t) {
            const int b = wave + 8 * t;
            accs[t] = 0.f;
            if (b < IMG * GR_NBLK) {
                const int im = b >= GR_NBLK ? 1 : 0, bb = b - im * GR_NBLK;
                const E *pa = A2 + (im * GR_NIN + min(bb * 16 + lrow, GR_NIN - 1)) * LDA2 + 4 * lk;
#pragma unroll
                for (int j = 0; j < GR_C2 / 16; ++j) {
                    f32x4 av;
                    if constexpr (is_f32<E>()) {
                        av = *(const f32x4 *)(pa + 16 * j);
                    } else {
                        using V = __attribute__((ext_vector_type(4))) E;
                        const V v = *(const V *)(pa + 16 * j);
                        av = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) accs[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bfrag[j][s], accs[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // every wave has read the second map: the products may overwrite the first map's place (and beyond)
#pragma unroll
        for (int t = 0; t < (IMG * GR_NBLK + 7) / 8; ++t) {
            const int b = wave + 8 * t;
            if (b < IMG * GR_NBLK) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_c[(b * 16 + 4 * lk + r) * 16 + lrow] = accs[t][r];
            }
        }
    }
    __syncthreads();
    // ---- gather the four contributions of every output pixel, bias + tanh, min / max, rescale: thread t takes pixels
    // t and t + 512 of both images (an image's sums run in the same order whichever place it has in the workgroup)
    {
        float mn[IMG], mx[IMG], se[IMG];
#pragma unroll
        for (int im = 0; im < IMG; ++im) {
            mn[im] = INFINITY;
            mx[im] = -INFINITY;
            se[im] = 0.f;
        }
        for (int i = tid; i < GR_NPIX; i += 512) {
            const int oy = i / GR_SIDE, ox = i - oy * GR_SIDE;
            const int py = oy & 1, my = oy >> 1, px = ox & 1, mxx = ox >> 1;
            // (input offset, kernel index) = (0, 1), (-1, 3) for even rows; (0, 2), (+1, 0) for odd; same along x
            const int iy1 = my + (py ? 1 : -1), ky0 = py ? 2 : 1, ky1 = py ? 0 : 3;
            const int ix1 = mxx + (px ? 1 : -1), kx0 = px ? 2 : 1, kx1 = px ? 0 : 3;
            const bool yok = (unsigned)iy1 < (unsigned)GR_HIN, xok = (unsigned)ix1 < (unsigned)GR_HIN;
            const int o00 = (my * GR_HIN + mxx) * 16 + ky0 * 4 + kx0, o01 = (my * GR_HIN + ix1) * 16 + ky0 * 4 + kx1;
            const int o10 = (iy1 * GR_HIN + mxx) * 16 + ky1 * 4 + kx0, o11 = (iy1 * GR_HIN + ix1) * 16 + ky1 * 4 + kx1;
#pragma unroll
            for (int im = 0; im < IMG; ++im) {
                if (im < nimg) {
                    const float *sc_im = s_c + im * GR_NBLK * 16 * 16;
                    const float t00 = sc_im[o00];
                    const float t01 = xok ? sc_im[o01] : 0.f;
                    const float t10 = yok ? sc_im[o10] : 0.f;
                    const float t11 = (yok && xok) ? sc_im[o11] : 0.f;
                    const float v = tanhf(((t00 + t01) + (t10 + t11)) + a.b5);
                    s_img[im * GR_NPIX + i] = v;
                    mn[im] = fminf(mn[im], v);
                    mx[im] = fmaxf(mx[im], v);
                    if (a.img) a.img[(size_t)(img0 + im) * GR_NPIX + i] = v;
                    if (a.target) {
                        const float dlt = a.target[(size_t)(img0 + im) * GR_NPIX + i] - v;
                        se[im] = fmaf(dlt, dlt, se[im]);
                    }
                }
            }
        }
#pragma unroll
        for (int im = 0; im < IMG; ++im) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mn[im] = fminf(mn[im], __shfl_xor(mn[im], o));
                mx[im] = fmaxf(mx[im], __shfl_xor(mx[im], o));
                se[im] += __shfl_xor(se[im], o);
            }
            if (lane == 0) {
                s_red[im * 24 + wave] = mn[im];
                s_red[im * 24 + 8 + wave] = mx[im];
                s_red[im * 24 + 16 + wave] = se[im];
            }
        }
        __syncthreads();
#pragma unroll
        for (int im = 0; im < IMG; ++im) {
            if (im < nimg) {
                const float *rd = s_red + im * 24;
                float lo = rd[0], hi = rd[8], sum = rd[16];
#pragma unroll
                for (int k = 1; k < 8; ++k) {
                    lo = fminf(lo, rd[k]);
                    hi = fmaxf(hi, rd[8 + k]);
                    sum += rd[16 + k];
                }
                const float range = hi - lo;
                float *d01 = a.img01 + (size_t)(img0 + im) * GR_NPIX;
                for (int i = tid; i < GR_NPIX; i += 512) d01[i] = (s_img[im * GR_NPIX + i] - lo) / range;  // each thread rescales the pixels it wrote
                if (a.mse && tid == 0) a.mse[img0 + im] = sum / (float)GR_NPIX;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ host side
static bool g_rs_enabled = [] {
    const char *e = getenv("GDPSO_STAGE1_RESIDENT");
    return e ? e[0] != '0' : true;
}();
// Largest batch (as planned: plan_batch) that takes the resident kernel: none by default.  Its workgroups are one
// per CU and run for the whole stage, so a large batch is served in rounds; measured against the tiled kernels (whose
// large grids reach a higher fraction of the matrix peak, but in some fifty launches with every map through memory)
// the resident launches still come out ahead at whole rounds: +7 % at 2048 particles, +2 % at 4096 ... 16384.
constexpr int RS_NO_LIMIT = 1 << 30;
static int g_rs_max_images = [] {
    const char *e = getenv("GDPSO_STAGE1_RESIDENT_MAX");
    return e ? atoi(e) : RS_NO_LIMIT;
}();
// Smallest batch that takes it, 0: by the element type.  fp32, 64: below, a handful of workgroups walk some sixty
// layers on a few CUs, bound by their matrix pipes, while the tiled kernels spread every layer over the device.
// 16-bit storage, 1: there the tiled launches are bound by their fixed costs, and one launch for some fifteen wins at
// every size (32 images: 114 k -> 148 k particle-evals/s).
static int g_rs_min_images = [] {
    const char *e = getenv("GDPSO_STAGE1_RESIDENT_MIN");
    return e ? atoi(e) : 0;
}();
static int rs_min_images(int dtype) { return g_rs_min_images > 0 ? g_rs_min_images : dtype == DT_F32 ? 64 : 1; }

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

// Smallest batch whose launch goes on through the second stage, 0: by the element type (fp32: 160, 16-bit: 1).
static int g_rs2_min_images = [] {
    const char *e = getenv("GDPSO_STAGE2_RESIDENT_MIN");
    return e ? atoi(e) : 0;
}();
static int rs2_min_images(int dtype) { return g_rs2_min_images > 0 ? g_rs2_min_images : dtype == DT_F32 ? 160 : 1; }
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

// A last round of at most one image per CU runs with ONE image a workgroup (GDPSO_RESIDENT_TAIL=0: with two like the rest).
static bool g_rs_tail_single = [] {
    const char *e = getenv("GDPSO_RESIDENT_TAIL");
    return e ? e[0] != '0' : true;
}();

// Of a batch served two images a workgroup, how many images (the last ones) form a last round that one image a
// workgroup serves faster: the remainder behind whole rounds of 2 x num_cus images, when it is at most num_cus.
static int resident_tail_images(int n_images, int images_per_wg, int num_cus) {
    if (!g_rs_tail_single || g_rs_force_images > 0 || images_per_wg != 2 || num_cus <= 0 || n_images <= 2 * num_cus) return 0;
    const int r = n_images % (2 * num_cus);
    return r <= num_cus ? r : 0;
}

// How many images a workgroup of the resident launch should take for a batch planned as n_images on a device of
// num_cus CUs: 0 (the tiled kernels serve the batch better), 1 or 2.  A workgroup holds its CU for the whole run of
// layers, so the launch takes rounds x (time of one workgroup): one image a workgroup while that still leaves every
// image a CU of its own, two up to two images per CU; beyond, whole rounds of 2 x num_cus images and a last round
// of one or two images a workgroup (resident_tail_images; a one-image round takes 0.62 of a two-image one).  Measured
// on MI355X (256 CUs, fp32) against the tiled kernels at 600 ... 16384 images: ahead at every size, by 0.5 % (600
// images: a last round of 88 workgroups) to 9 % (768).  Without the one-image last round (GDPSO_RESIDENT_TAIL=0)
// only batches whose rounds come out at least 85 % full take the launch (768 images: 6 % behind the tiled kernels).
int resident_stage_images(int n_images, int num_cus, int dtype) {
    if (n_images < rs_min_images(dtype) || n_images > g_rs_max_images) return 0;
    if (g_rs_force_images > 0) return g_rs_force_images;
    if (n_images <= num_cus) return 1;
    if (n_images <= 2 * num_cus || g_rs_tail_single) return 2;
    const long capacity = (long)ceil_div(ceil_div(n_images, 2), num_cus) * 2 * num_cus;
    return (long)n_images * 100 >= capacity * 85 ? 2 : 0;
}

// Whether the launch should go on through the second stage for a batch planned as n_images: in fp32, with few
// workgroups, its long layers on a handful of CUs lose to the chained launch, which spreads them over the device
// (measured: even at 128 images, +6 % at 192, +5 % at 256); with 16-bit storage it pays at every size (+8 % at 64 and 128).
bool resident_stage_second(const ResidentStage *s, int n_images) {
    return s != nullptr && s->stage2 && n_images >= rs2_min_images(s->dtype);
}

static int launch_resident_stage_part(const ResidentStage *s, hipStream_t stream, int n_images, int images_per_wg, bool second,
                                      const float *img01, void *out, void *zero, int zero_quads) {
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

int launch_resident_stage(const ResidentStage *s, hipStream_t stream, int n_images, int images_per_wg, int num_cus, bool second,
                          const float *img01, void *out, void *zero, int zero_quads) {
    if (images_per_wg != 1 && images_per_wg != 2) return fail(GDPSO_ERR_ARG, "resident stage: one or two images a workgroup");
    if (second && !s->stage2) return fail(GDPSO_ERR_ARG, "resident stage: the handle holds no second stage");
    // whole rounds two images a workgroup, a short last round one image a workgroup in a launch of its own behind it
    const int tail = resident_tail_images(n_images, images_per_wg, num_cus), head = n_images - tail;
    int rc = launch_resident_stage_part(s, stream, head, images_per_wg, second, img01, out, zero, zero_quads);
    if (rc || tail == 0) return rc;
    const size_t out_bytes = (size_t)(second ? RS2_PIX * 512 : RS_PIX * 256) * dtype_size(s->dtype);
    return launch_resident_stage_part(s, stream, tail, 1, second, img01 + (size_t)head * ST_SIDE * ST_SIDE, (char *)out + (size_t)head * out_bytes,
                                      nullptr, 0);
}

// ---- the resident generator (host side)
// 0: never; 1: where the launch is expected to beat the tiled kernels (resident_generator_images); 2: wherever the
// resident launches' batch limits allow.
static int g_gr_mode = [] {
    const char *e = getenv("GDPSO_GEN_RESIDENT");
    return e ? (e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1) : 1;
}();

struct ResidentGenerator {
    int dtype;
    GrArgs args;  // weights filled in; latents / first map, outputs and n per launch
    // per image: nominal FLOPs of the second and last layer (and of the first, when the launch computes it: first_flops),
    // FLOPs issued to the matrix cores with two / one image a workgroup (second and last layer; the first is vector ALU work)
    double flops, first_flops, executed2, executed1, weight_bytes;
};

bool resident_generator_enabled() { return g_gr_mode != 0; }

// Images a workgroup of the generator's launch takes for a batch planned as n_images (0: the tiled kernels).  Within
// the batch limits of the resident launches; with 16-bit storage always (the tiled launches are bound by their fixed
// costs there: +2 ... 3.5 % at 64 ... 512 images); in fp32 the launch is bound by the matrix pipe of the CUs that got
// a workgroup, so it pays only when (nearly) every CU has one: priced in units of a two-image round (a one-image round
// 0.55) against the tiled kernels' 1.11 per 2 x num_cus images (measured: 64 images -4 %, 128 -2.5 %, 192 -0.8 %,
// 256 +0.6 %, 384 -0.8 %, 512 +0.9 %, 768 and more ahead).
int resident_generator_images(int n_images, int num_cus, int dtype) {
    const int images = g_gr_mode == 0 ? 0 : resident_stage_images(n_images, num_cus, dtype);
    if (images == 0 || g_gr_mode == 2 || dtype != DT_F32 || g_rs_force_images > 0) return images;
    const int round = 2 * num_cus, full = n_images / round, r = n_images % round;
    const double last = r == 0 ? 0.0 : r <= num_cus ? 0.55 : 1.0;
    return full + last <= 1.11 * n_images / round ? images : 0;
}
void resident_generator_free(ResidentGenerator *g) { delete g; }

// w3: the second layer's torch weight [128][64][4][4] (ConvTranspose2d layout [cin][cout][ky][kx]).
int resident_generator_new(DeviceArena &arena, int dtype, int z_dim, const float *w1t, const float *sc1, const float *sh1, const float *w3,
                           const ConvLayer *l2, const float *w5, float b5, ResidentGenerator **out) {
    *out = nullptr;
    hipError_t e = hipSuccess;
#define GDPSO_GR_ATTR(IMG)                                                                                                    \
    if (e == hipSuccess)                                                                                                      \
        GDPSO_WITH_DTYPE(dtype, e = hipFuncSetAttribute((const void *)gen_resident_kernel<E, IMG>,                             \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)GrLds<E>::BYTES));
    GDPSO_GR_ATTR(1) GDPSO_GR_ATTR(2)
#undef GDPSO_GR_ATTR
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return GDPSO_OK;  // (not an error: the tiled kernels remain)
    }
    auto g = new ResidentGenerator();
    std::memset(&g->args, 0, sizeof(g->args));
    g->dtype = dtype;
    g->args.zd = z_dim;
    g->args.w1t = w1t;
    g->args.sc1 = sc1;
    g->args.sh1 = sh1;
    g->args.sc3 = l2->scale;
    g->args.sh3 = l2->shift;
    g->args.w5 = w5;
    g->args.b5 = b5;
    // fragment order per phase: B[k = tap * 128 + ci][co] = W[ci][co][ky][kx], tap = 2 ty + tx, ky = 1 | 3 (even output
    // rows: offsets 0 | -1) or 2 | 0 (odd rows: 0 | +1), the same along x
    const int es = (int)dtype_size(dtype), vec = 16 / es, kg = dtype == DT_F32 ? 16 : 32, K = 4 * GR_C1, J = K / kg;
    std::vector<uint8_t> packed((size_t)4 * GR_C2 * K * es);
    for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
        for (int cb = 0; cb < GR_C2 / 16; ++cb)
            for (int j = 0; j < J; ++j)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < vec; ++s) {
                        const int co = cb * 16 + (lane & 15), k = kg * j + vec * (lane >> 4) + s;
                        const int tap = k / GR_C1, ci = k - tap * GR_C1, ty = tap >> 1, tx = tap & 1;
                        const int ky = ty == 0 ? (py ? 2 : 1) : (py ? 0 : 3), kx = tx == 0 ? (px ? 2 : 1) : (px ? 0 : 3);
                        const float v = w3[(((size_t)ci * GR_C2 + co) * 4 + ky) * 4 + kx];
                        const size_t at = ((((size_t)ph * (GR_C2 / 16) + cb) * J + j) * 64 + lane) * vec + s;
                        if (dtype == DT_F32) {
                            std::memcpy(&packed[at * 4], &v, 4);
                        } else {
                            const uint16_t h = dtype == DT_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
                            std::memcpy(&packed[at * 2], &h, 2);
                        }
                    }
    }
    int rc = arena.upload(packed.data(), packed.size(), &g->args.w3);
    // rows of a phase by border class: a phase's second tap along y reads row my + 1 (odd output rows) or my - 1 (even),
    // so it leaves the map for my = 6 resp. 0; the same along x
    std::vector<int16_t> rows(4 * 112, -1);
    for (int ph = 0; ph < 4; ++ph) {
        const int ybord = (ph >> 1) ? RS_SIDE - 1 : 0, xbord = (ph & 1) ? RS_SIDE - 1 : 0;
        int16_t *t = rows.data() + ph * 112;
        int n0 = 0, ny = 80, nx = 96;  // interior rows from 0 (72 of them), y-border rows from 80 (14), x-border rows from 96 (12)
        for (int r = 0; r < RS_ROWS; ++r) {
            const int p = r % RS_PIX, my = p / RS_SIDE, mx = p % RS_SIDE;
            if (my == ybord)
                t[ny++] = (int16_t)r;  // (a corner row's x tap reads the row of zeros)
            else if (mx == xbord)
                t[nx++] = (int16_t)r;
            else
                t[n0++] = (int16_t)r;
        }
    }
    if (!rc) rc = arena.upload(rows.data(), rows.size() * sizeof(int16_t), (const void **)&g->args.rows);
    if (rc) {
        delete g;
        return rc;
    }
    const double last_nominal = 2.0 * GR_NPIX * 4 * GR_C2, last_executed = 2.0 * GR_NBLK * 16 * 16 * GR_C2;  // 4 taps a pixel; 13 blocks x 16 positions
    g->flops = l2->flops_nominal + last_nominal;
    g->first_flops = 2.0 * z_dim * GR_NCOL;
    g->executed2 = 2.0 * 4 * 16 * (5 + 0.5 + 0.5) / 2 * GR_C2 * K + last_executed;  // 5 row blocks with four taps, 2 with two
    g->executed1 = 2.0 * 4 * 64 * GR_C2 * K + last_executed;
    g->weight_bytes = (double)es * 4 * GR_C2 * K;
    *out = g;
    return GDPSO_OK;
}

// z (latents, z_dim <= 16 only) or a1 (the first layer's output) is given, the other null.
static int launch_resident_generator_part(const ResidentGenerator *g, hipStream_t stream, int n_images, int images_per_wg, const void *z,
                                          int z_is_f64, const void *a1, float *img, float *img01, const float *target, float *mse) {
    GrArgs a = g->args;
    a.z = z;
    a.z_is_f64 = z_is_f64;
    a.a1 = a1;
    a.n = n_images;
    a.img = img;
    a.img01 = img01;
    a.target = target;
    a.mse = mse;
    const dim3 grid((unsigned)ceil_div(n_images, images_per_wg));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = conv_timing_slot(&e0, &e1);
    if (timed)
        conv_timing_account((g->flops + (z ? g->first_flops : 0.0)) * n_images, (images_per_wg == 2 ? g->executed2 : g->executed1) * n_images,
                            g->weight_bytes + (double)n_images * (dtype_size(g->dtype) * (a1 ? (double)RS_PIX * GR_C1 : 0.0) + 8.0 * GR_NPIX));
#define GDPSO_GR_LAUNCH(IMG)                                                                                                          \
    if (timed) {                                                                                                                      \
        GDPSO_WITH_DTYPE(g->dtype, hipExtLaunchKernelGGL((gen_resident_kernel<E, IMG>), grid, dim3(512), GrLds<E>::BYTES, stream, e0, e1, 0, a)); \
    } else {                                                                                                                          \
        GDPSO_WITH_DTYPE(g->dtype, hipLaunchKernelGGL((gen_resident_kernel<E, IMG>), grid, dim3(512), GrLds<E>::BYTES, stream, a));    \
    }
    if (images_per_wg == 2) {
        GDPSO_GR_LAUNCH(2)
    } else {
        GDPSO_GR_LAUNCH(1)
    }
#undef GDPSO_GR_LAUNCH
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

int launch_resident_generator(const ResidentGenerator *g, hipStream_t stream, int n_images, int images_per_wg, int num_cus, const void *z,
                              int z_is_f64, const void *a1, float *img, float *img01, const float *target, float *mse) {
    if (images_per_wg != 1 && images_per_wg != 2) return fail(GDPSO_ERR_ARG, "resident generator: one or two images a workgroup");
    if ((z == nullptr) == (a1 == nullptr)) return fail(GDPSO_ERR_ARG, "resident generator: latents or the first map, not both");
    if (z != nullptr && g->args.zd > GR_MAX_Z) return fail(GDPSO_ERR_ARG, "resident generator: latents of more than 16 dimensions");
    const int tail = resident_tail_images(n_images, images_per_wg, num_cus), head = n_images - tail;
    int rc = launch_resident_generator_part(g, stream, head, images_per_wg, z, z_is_f64, a1, img, img01, target, mse);
    if (rc || tail == 0) return rc;
    const size_t zb = (size_t)head * g->args.zd * (z_is_f64 ? 8 : 4), ab = (size_t)head * RS_PIX * GR_C1 * dtype_size(g->dtype);
    const size_t px = (size_t)head * GR_NPIX;
    return launch_resident_generator_part(g, stream, tail, 1, z ? (const char *)z + zb : nullptr, z_is_f64, a1 ? (const char *)a1 + ab : nullptr,
                                          img ? img + px : nullptr, img01 + px, target ? target + px : nullptr, mse ? mse + head : nullptr);
}

}  // namespace gdpso

extern "C" int gdpso_resident_options(int stages, int max_images, int min_images) {
    using namespace gdpso;
    GDPSO_REQUIRE(stages >= 0 && stages <= 2, "resident_options: stages must be 0, 1 or 2");
    GDPSO_REQUIRE(max_images >= -1 && min_images >= -1, "resident_options: max_images / min_images must be 0 (keep), -1 (default) or a batch size");
    if (min_images != 0) g_rs_min_images = min_images > 0 ? min_images : 0;  // (0 inside: by the element type)
    g_rs_enabled = stages != 0;
    g_rs_stage2 = stages == 2;
    if (max_images != 0) g_rs_max_images = max_images > 0 ? max_images : RS_NO_LIMIT;
    return GDPSO_OK;
}

extern "C" int gdpso_resident_generator_option(int mode) {
    using namespace gdpso;
    GDPSO_REQUIRE(mode >= 0 && mode <= 2, "resident_generator_option: 0 (off), 1 (where it pays) or 2 (wherever the batch limits allow)");
    g_gr_mode = mode;
    return GDPSO_OK;
}

extern "C" int gdpso_resident_describe(int n_images, int num_cus, int dtype, int *stage_images, int *tail_images, int *second_stage,
                                       int *generator_images) {
    using namespace gdpso;
    GDPSO_REQUIRE(n_images > 0 && num_cus > 0, "resident_describe: n_images and num_cus must be positive");
    GDPSO_REQUIRE(dtype == DT_F32 || dtype == DT_BF16 || dtype == DT_F16, "resident_describe: unknown element type");
    const int images = g_rs_enabled ? resident_stage_images(n_images, num_cus, dtype) : 0;
    if (stage_images) *stage_images = images;
    if (tail_images) *tail_images = resident_tail_images(n_images, images, num_cus);
    if (second_stage) *second_stage = images > 0 && g_rs_stage2 && n_images >= rs2_min_images(dtype) ? 1 : 0;
    if (generator_images) *generator_images = resident_generator_images(n_images, num_cus, dtype);
    return GDPSO_OK;
}

extern "C" int gdpso_resident_force_images(int images_per_workgroup) {
    using namespace gdpso;
    GDPSO_REQUIRE(images_per_workgroup >= 0 && images_per_workgroup <= 2, "resident_force_images: 0 (planner's choice), 1 or 2");
    g_rs_force_images = images_per_workgroup;
    return GDPSO_OK;
}
