// Internal declarations shared by the translation units of libgdpso.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/gdpso.h"

namespace gdpso {

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define GDPSO_HIP(expr)                                                                         \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return ::gdpso::fail(GDPSO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define GDPSO_REQUIRE(cond, msg)                                      \
    do {                                                              \
        if (!(cond)) return ::gdpso::fail(GDPSO_ERR_ARG, (msg));      \
    } while (0)

constexpr int MAX_TAPS = 64;   // 7x7 = 49 is the largest window on the path
constexpr int MAX_PHASES = 16; // 3x3 border classes of a padded convolution, or 4 sub-pixel phases x 2x2 classes
constexpr int PHASE_HDR = 12;  // ints ahead of the taps in a phase's device table

// Storage / matrix-core input type of a network's activations and weights.  Accumulation, BatchNorm
// scale / shift, residual adds and activations are always fp32.
enum DType : int { DT_F32 = GDPSO_DTYPE_F32, DT_BF16 = GDPSO_DTYPE_BF16, DT_F16 = GDPSO_DTYPE_F16 };
inline bool dtype_ok(int d) { return d == DT_F32 || d == DT_BF16 || d == DT_F16; }
inline size_t dtype_size(int d) { return d == DT_F32 ? 4 : 2; }
// K elements one wave consumes per K-tile (eight 16-byte fragments' worth along K per lane pair):
// 32 floats or 64 half-width elements; channel counts must be multiples of it.
inline int dtype_kquant(int d) { return d == DT_F32 ? 32 : 64; }

// Element types of the reduced-precision instantiations, and four consecutive elements <-> four
// floats (epilogues: residual in, result out; fp32 kernels that read or write 16-bit activations).
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16_t = __bf16;
using f16_t = _Float16;
template <typename E>
constexpr int vec_of() { return 16 / (int)sizeof(E); }  // elements per 16-byte vector
template <typename E>
constexpr bool is_f32() { return sizeof(E) == 4; }
template <typename E>
__device__ __forceinline__ f32x4 load4(const E *p) {  // p: global memory
    if constexpr (is_f32<E>()) {
        return *(const __attribute__((address_space(1))) f32x4 *)p;
    } else {
        using V = __attribute__((ext_vector_type(4))) E;
        const V v = *(const __attribute__((address_space(1))) V *)p;
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
}
template <typename E>
__device__ __forceinline__ void store4(E *p, f32x4 v) {
    if constexpr (is_f32<E>()) {
        *(f32x4 *)p = v;
    } else {
        using V = __attribute__((ext_vector_type(4))) E;
        *(V *)p = V{(E)v[0], (E)v[1], (E)v[2], (E)v[3]};
    }
}
// Runs STMT with the type alias E bound to the element type of DTYPE.
#define GDPSO_WITH_DTYPE(DTYPE, STMT)       \
    do {                                    \
        if ((DTYPE) == ::gdpso::DT_F32) {   \
            using E = float;                \
            STMT;                           \
        } else if ((DTYPE) == ::gdpso::DT_BF16) { \
            using E = ::gdpso::bf16_t;      \
            STMT;                           \
        } else {                            \
            using E = ::gdpso::f16_t;       \
            STMT;                           \
        }                                   \
    } while (0)

// Activation applied in a convolution's epilogue.
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY02 = 2 };

// One convolution (or one transposed convolution split into sub-pixel phases) expressed as
// an implicit GEMM over NHWC activations.  The output is covered by `nphase` phases; phase p
// owns the output pixels (my*out_mul + off_y, mx*out_mul + off_x), (my, mx) over its own
// hm x wm grid, and computes
//   out[pix, co] = act( scale[co] * sum_{t,ci} in[n, my*in_stride + in_y0 + dy[t], mx*in_stride + in_x0 + dx[t], ci]
//                                             * wt[wt_offset + co*wstride + wtap[t]*cin + ci]  + shift[co] (+ res[pix, co]) )
// Taps that fall outside the input read zeros.  A padded convolution is split into border
// classes (runs of output rows / columns with the same set of in-range taps), each a phase
// whose tap list holds only taps that are in range for all of its pixels; the classes share
// one weight matrix.
struct ConvPhase {
    int ntaps;
    int off_y, off_x;      // output offset of this phase
    int hm, wm;            // output sub-grid of this phase
    int in_y0, in_x0;      // input origin of this phase
    long wt_offset;        // element offset of this phase's [cout][wstride] weight matrix
    int wstride;           // floats per output channel in that matrix
    int8_t dy[MAX_TAPS];
    int8_t dx[MAX_TAPS];
    int16_t wtap[MAX_TAPS];  // where the tap's cin weights start in a weight row, in units of cin
};

struct ConvLayer {
    int cin, cout;
    int hin, win;
    int hout, wout;
    int in_stride, out_mul;
    int nphase;
    int act;
    ConvPhase phase[MAX_PHASES];
    int dtype;           // DType of the weights at `wt` and of the activations the layer reads and writes
    const void *wt;      // device, elements of `dtype`
    // The same weights once more in chunk-major order, for the tile shapes whose waves load their B fragments
    // straight from global memory: every [cout][wstride] matrix as [cout / 32][wstride / VEC][32][VEC], VEC = the
    // elements of a 16-byte chunk (null when cout is not a multiple of 32 or a row not a whole number of chunks).
    const void *wtq;
    const float *scale;  // device [cout] (never null)
    const float *shift;  // device [cout]
    const int *table;    // device [nphase][PHASE_HDR + MAX_TAPS], see upload_conv
    size_t weight_floats;  // elements of the packed weights
    // launch plan (chosen once per batch size by plan_conv)
    int tile;            // 128 or 64
    int splits;          // split-K factor
    double flops_nominal;  // per image: 2 * out_pixels * cout * cin * kh * kw (padding taps included)
};

// conv_gemm.hip
int plan_conv(ConvLayer &L, int n_images, int num_cus);
size_t conv_splitk_bytes(const ConvLayer &L, int n_images);
int conv_splitk_prepare(hipStream_t stream, float *splitk_ws);
// in / res / out are NHWC arrays of the layer's dtype; splitk_ws holds fp32 partial sums.
int launch_conv(const ConvLayer &L, hipStream_t stream, int n_images, const void *in, const void *res,
                void *out, float *splitk_ws);

int launch_conv_pair(const ConvLayer &La, const ConvLayer &Lb, hipStream_t stream, int n_images, const void *in,
                     void *out_a, void *out_b, float *splitk_ws);

int launch_conv_then_1x1(const ConvLayer &La, const ConvLayer &Lb, hipStream_t stream, int n_images, const void *in,
                         void *mid, const void *res, void *out, float *splitk_ws);

// A run of dependent layers in one persistent launch (conv_gemm.hip: conv_chain_kernel).  Offsets are
// bytes into the workspace the chain is launched on.
struct ConvChain;
// A buffer a chain reads or writes: byte offset in the workspace and, for buffers the chain writes,
// their size: every share of the images keeps a fixed slice (size / shares) of it, whatever the layer,
// which must hold a share's bytes of every layer stored there.  bytes 0 = standard NHWC layout (the
// chain's input, read only, and its final output, written once).
struct ChainBuf {
    size_t offset, bytes;
};
bool chain_enabled();
int chain_teams();         // teams per XCD a new chain should have (1 or 2)
size_t chain_ctl_bytes();  // control block inside the workspace (zeroed by launch_chain)
ConvChain *chain_new(int n_images, int nteams);
void chain_free(ConvChain *c);
int chain_steps(const ConvChain *c);
bool chain_layer_ok(const ConvLayer &L, int n_images, int nteams);
int chain_add(ConvChain *c, const ConvLayer *La, const ChainBuf &in, const ChainBuf *res, const ChainBuf &out_a,
              const ConvLayer *Lb = nullptr, const ChainBuf *out_b = nullptr);
int chain_finish(ConvChain *c, size_t ctl_off);
// ctl_cleared: an earlier kernel on the same stream has zeroed the control block (chain_ctl_bytes() at the
// offset given to chain_finish) since the chain last ran on this workspace; else a memset is enqueued first.
int launch_chain(const ConvChain *c, hipStream_t stream, void *workspace, bool ctl_cleared = false);
// Called by every entry point that launches a network: tells whether `stream` is (still) the only
// stream this library has launched on on the current device; chains run only while that holds.
int chain_note_stream(hipStream_t stream, bool *single);

// The batch size launch plans are made for: n_images, or the pinned size of gdpso_plan_options.
int plan_batch(int n_images);
// Per-launch timing of convolution kernels (gdpso_conv_timing_begin / _end) for launches made outside
// conv_gemm.hip: conv_timing_slot hands out the start / stop events of the next timed launch (false: timing
// is off), conv_timing_account adds what that launch computes and moves.
bool conv_timing_slot(hipEvent_t *start, hipEvent_t *stop);
void conv_timing_account(double flops_nominal, double flops_executed, double bytes);

// stage_resident.hip: the stem and the first ResNet stage in one launch with the activations resident in LDS
// (28x28 images, 7x7 maps, 64 -> 64 -> 256 channel bottlenecks; any of the three element types).
constexpr int RS_MAX_BLOCKS = 4, RS2_MAX_BLOCKS = 8;
struct ResidentStage;
struct DeviceArena;
struct ConvLayer;
bool resident_stage_enabled();
// Images a workgroup of the resident launch takes for a batch planned as n_images (1 or 2), or 0: the tiled kernels.
int resident_stage_images(int n_images, int num_cus, int dtype);
// *out stays null (and GDPSO_OK is returned) when the device cannot run the kernel.
int resident_stage_new(DeviceArena &arena, int dtype, int nblocks, const float *const *w, const float *w_ds, const ConvLayer *const *c1,
                       const ConvLayer *const *c2, const ConvLayer *const *c3, const ConvLayer *ds, const float *stem_w,
                       const float *stem_scale, const float *stem_shift, int nblocks2, const float *const *w2, const float *w2_ds,
                       const ConvLayer *const *d1, const ConvLayer *const *d2, const ConvLayer *const *d3, const ConvLayer *ds2,
                       ResidentStage **out);
// Whether the handle's launch also runs the second stage (its output is then that stage's).
bool resident_stage_has_second(const ResidentStage *s);
void resident_stage_free(ResidentStage *s);
// img01: the [n][28][28] images; out: the stage's NHWC output; zero / zero_quads: 16-byte words the launch clears on its way.
// second: go on through the second stage (out: that stage's output); only with resident_stage_has_second.
int launch_resident_stage(const ResidentStage *s, hipStream_t stream, int n_images, int images_per_wg, int num_cus, bool second, const float *img01,
                          void *out, void *zero, int zero_quads);
// Whether a batch planned as n_images should take the second stage along.
bool resident_stage_second(const ResidentStage *s, int n_images);

// The reference's generator (Z -> 128 x 7 x 7 -> 64 x 14 x 14 -> 28 x 28) in one launch with its maps resident in LDS.
struct ResidentGenerator;
bool resident_generator_enabled();
// Images a workgroup of the generator's launch takes for a batch planned as n_images: 0 (tiled kernels), 1 or 2.
int resident_generator_images(int n_images, int num_cus, int dtype);
// w1t / sc1 / sh1: the first layer as gen_first_kernel takes it (device); w3: the second layer's torch weight (host);
// l2: that layer (its device scale / shift are shared); w5: the last layer [16][64] (device).  *out stays null (and
// GDPSO_OK is returned) when the device cannot run the kernel.
int resident_generator_new(DeviceArena &arena, int dtype, int z_dim, const float *w1t, const float *sc1, const float *sh1, const float *w3,
                           const ConvLayer *l2, const float *w5, float b5, ResidentGenerator **out);
void resident_generator_free(ResidentGenerator *g);
// z (latents, z_dim <= 16) or a1 (the first layer's output, elements of the network's type): one of them, the other null.
int launch_resident_generator(const ResidentGenerator *g, hipStream_t stream, int n_images, int images_per_wg, int num_cus, const void *z, int z_is_f64,
                              const void *a1, float *img, float *img01, const float *target, float *mse);

// host-side helpers (pack.cpp)
struct DeviceArena {
    std::vector<void *> blocks;
    ~DeviceArena();
    int upload(const void *host, size_t bytes, const void **dev);
    int upload(const std::vector<float> &host, const float **dev) {
        return upload(host.data(), host.size() * sizeof(float), (const void **)dev);
    }
};

// Uploads a layer's packed weights (rounded to `dtype`, round-to-nearest-even), scale/shift and tap
// table; fills the device pointers and L.dtype.
int upload_conv(DeviceArena &arena, ConvLayer &L, const std::vector<float> &packed,
                const std::vector<float> &scale, const std::vector<float> &shift, int dtype = DT_F32);

// The PHASE_HDR + 1 ints a workgroup reads about its phase: the header of the phase's row of the device
// table (upload_conv) followed by the first packed tap.
void conv_phase_header(const ConvPhase &P, int *out);

// fp32 -> bf16 / fp16 bit patterns, round-to-nearest-even (host).
uint16_t f32_to_bf16_bits(float v);
uint16_t f32_to_f16_bits(float v);

// Folds an eval-mode BatchNorm (and an optional preceding bias) into y = acc*scale + shift.
void fold_bn(int c, const float *bias, const float *gamma, const float *beta, const float *mean,
             const float *var, std::vector<float> &scale, std::vector<float> &shift);

// Builds an ordinary convolution layer from a PyTorch [cout][cin][kh][kw] weight.
// Returns the packed weights through `packed` (the caller uploads them).  cin is padded
// with zero weights to `cin_pad` and cout to `cout_pad` (both multiples the kernel needs).
void make_conv(ConvLayer &L, const float *w, int cout, int cin, int kh, int kw, int stride, int pad,
               int hin, int win, int act, std::vector<float> &packed, int cin_pad = 0, int cout_pad = 0);

// Whether a kh x kw window with this padding fits a ConvPhase: at most MAX_TAPS taps per phase, tap
// offsets (k - pad) representable in the int8 dy / dx fields.  make_conv does not check it.
inline bool conv_window_fits(int ksize, int pad) {
    return ksize >= 1 && (long)ksize * ksize <= MAX_TAPS && pad >= 0 && pad <= 127 && ksize - 1 - pad <= 127;
}

// Builds the 4-phase form of ConvTranspose2d(k4, s2, p1) from a PyTorch [cin][cout][4][4] weight.
void make_convt_k4s2p1(ConvLayer &L, const float *w, int cin, int cout, int hin, int win, int act,
                       std::vector<float> &packed);

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Division by a run-time constant d >= 1 without a divide: for 0 <= n < 2^31,
// n / d == (uint64(n) * mul) >> shift with shift = 31 + ceil(log2 d), mul = ceil(2^shift / d)
// (mul * d - 2^shift < d <= 2^(shift - 31), so the error term n * that / 2^shift stays below 1 / d).
// The GPU has no integer divide: the compiler's expansion is some thirty dependent instructions.
struct FastDiv {
    unsigned mul, shift;
};
inline FastDiv fast_div_for(unsigned d) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    FastDiv f;
    f.shift = 31 + s;
    f.mul = (unsigned)(((1ull << f.shift) + d - 1) / d);
    return f;
}

}  // namespace gdpso
