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
// Tile: BM x BN outputs per 256-thread workgroup, two LDS buffers, register prefetch of the
// next K-tile, one barrier per K-tile, LDS-staged 16-byte epilogue.  Four shapes: 128x128 and
// 64x64 (four waves as 2x2, K-tiles of 32), 32x64 (two waves side by side, the other two take
// the second half of every 64-wide K-tile) and 32x32 (each wave a quarter of every 128-wide
// K-tile); the K parts are added through LDS in a fixed order.  Each step down doubles the
// workgroups of the small-M layers of the late ResNet stages without split-K slabs.  Layers that still have fewer
// workgroups than CUs split K over blockIdx.z into fp32 slabs summed by splitk_reduce_kernel.
//
// A layer is a list of phases, each an output sub-grid with its own tap list.  Padded
// convolutions are split into border classes (first / middle / last output rows x columns):
// inside a class every tap is in range for every row, so the taps that would only multiply
// padding zeros are not in its list at all.  On the small maps of the late ResNet stages
// that is most of the nominal work (a 3x3 convolution on a 2x2 map has 4 live taps of 9).
//
// Two launch-saving forms share the tile code: two layers that read the same input in one
// grid (identity_downsample + conv1), and a 64-channel layer whose activated tile stays in LDS
// as the A operand of the 1x1 layer behind it (conv2 + conv3 of the first stage).
#include <hip/hip_ext.h>

#include <cstdlib>
#include <cstring>
#include <deque>
#include <atomic>
#include <map>
#include <mutex>

#include "common.h"

#ifndef GDPSO_BDIRECT
#define GDPSO_BDIRECT 1  // 0: weights staged through LDS in every tile shape (A/B builds)
#endif
#ifndef GDPSO_PF
#define GDPSO_PF 2  // K-tiles the one-block-per-wave tile shapes keep in flight in registers (2 or 3)
#endif

namespace gdpso {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
// Operand loads name the global address space: a generic (flat) load also counts against
// lgkmcnt, so the LDS waits ahead of each barrier would drain the prefetch as well.
using global_f32x4_ptr = const __attribute__((address_space(1))) f32x4 *;

// (bf16_t, f16_t, vec_of, is_f32, load4, store4: common.h.)  With v_mfma_f32_32x32x16_bf16 / _f16 a
// lane's 16-byte fragment is the eight consecutive k of one instruction, so the tile code below is
// the fp32 one with K counted in 16-byte vectors: VEC elements each, 4 floats or 8 half-width.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int PHASE_TABLE_INTS = PHASE_HDR + MAX_TAPS;  // header (see upload_conv) + packed taps

struct ConvKArgs {
    const void *in;    // NHWC activations, elements of the instantiation's type
    const void *wt;    // packed weights, same type
    const void *wtq;   // the same weights in chunk-major layout (see BD in conv_tile), or null
    const float *scale;
    const float *shift;
    const void *res;   // residual, same type as the activations, or null
    void *out;
    float *part;       // split-K slabs [splits][pixels][cout], or null
    int *counters;     // split-K arrival counters, one per output tile (zero between launches), or null
    const int *table;  // [nphase][PHASE_TABLE_INTS]
    int n_images;
    int hin, win, cin;
    int cout, hout, wout;
    int in_stride, out_mul;
    int act, splits;
    int tiles_n;
    long slab;  // elements per split-K slab
    // Per phase: (running sum over phases of ceil(tiles / 8), the phase's tiles being M-tiles x column
    // tiles) | (8 * ceil(tiles / 8) - tiles) << 24; 0xffffff past the last phase.
    unsigned phase_word[MAX_PHASES];
    unsigned tiles_n_mul, tiles_n_shift;  // FastDiv by tiles_n
    int per_xcd;                  // workgroups of this layer per XCD (= the last phase's running end)
    int share_images;             // chains: images per share (the last shares of a batch that does not divide evenly hold fewer, or none)
    int snake;                    // CUs per XCD when the second round of workgroups is dealt in reverse, else 0
};

// A 1x1 convolution over the same pixels, applied to this layer's output tile before it leaves
// the workgroup (this layer has exactly 64 output channels = the whole K of the next one).
struct FuseArgs {  // fp32 instantiation only
    const float *wt;     // [cout][64]
    const float *scale;
    const float *shift;
    const float *res;    // residual added before the activation, or null
    float *out;
    int cout, act;
};

// What a tile of a chain waits for before it reads activations: the first `count` words of `flags`
// (one per member of the team, null: nothing) all having reached `target`.  `status` is the chain's
// abort word: a wait that runs out of patience sets it and every later wait of the launch returns at
// once, so a broken chain ends with wrong results and a raised flag instead of hanging the GPU.
struct ChainWait {
    const unsigned *flags = nullptr;
    unsigned target = 0;
    int count = 0;
    unsigned *status = nullptr;
    int peek = 1;
#ifdef GDPSO_STAMPS
    unsigned long long *stamp = nullptr;  // diagnostic build: where to note the time the wait ended
    unsigned long long *tile_stamps = nullptr;  // ... and where the tile code's own stamps go (null: not this tile)
#endif
};
// How long a wait may last, in s_memrealtime ticks (100 MHz).  The registration of all workgroups can
// legitimately take long when kernels of other streams or processes hold compute units (the chain's
// workgroups trickle in as those finish, and the ones already there keep their slots); a step's flags
// come from workgroups that are all running.  Past these limits something is broken.
constexpr unsigned long long CHAIN_REGISTER_TICKS = 2000000000ull;  // 20 s
constexpr unsigned long long CHAIN_STEP_TICKS = 200000000ull;       // 2 s
__device__ unsigned g_chain_aborts = 0;         // chains that gave up waiting since the library was loaded
__device__ unsigned g_chain_abort_bits = 0;     // what they were waiting for: 1 a step's flags, 2 the registration of all workgroups
__device__ unsigned g_chain_abort_arrived = 0;  // workgroups that had registered when a registration wait gave up

__device__ __forceinline__ void chain_abort(unsigned *status, unsigned bit) {
    if (__hip_atomic_fetch_or(status, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __hip_atomic_fetch_add(&g_chain_aborts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_or(&g_chain_abort_bits, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The first wave polls, one member's word per lane (agent-scope atomic loads: never served from the
// CU's L1), then the workgroup meets at a barrier.  No read-modify-write anywhere: arrivals at one
// counter serialise in the L2 (64 of them cost 1.6 us, measured by tools/micro/xcd_barrier.hip), plain
// stores to words of their own do not.  The memory clobber keeps the activation loads behind the wait.
// `early`: what chain_peek read from the lane's flag word at the top of the tile; the first check uses it.
__device__ __forceinline__ void chain_wait(const ChainWait &w, int tid, unsigned early) {
    if (w.flags == nullptr) return;
    if (tid < 64) {
        unsigned spins = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int base = 0; base < w.count; base += 64) {
            bool fresh = base == 0;
            for (;;) {
                const int m = base + tid;
                const unsigned f = fresh ? early : m < w.count ? __hip_atomic_load(w.flags + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                fresh = false;
                if (__all((int)(f >= w.target))) break;
                ++spins;
                if ((spins & 255u) == 0) {
                    const unsigned st = __hip_atomic_load(w.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__any((int)(st != 0 || __builtin_amdgcn_s_memrealtime() - t0 > CHAIN_STEP_TICKS))) {
                        if (tid == 0) chain_abort(w.status, 1u);
                        base = w.count;
                        break;
                    }
                }
            }
        }
    }
    __syncthreads();
    asm volatile("" ::: "memory");
#ifdef GDPSO_STAMPS
    if (w.stamp && tid == 0) *w.stamp = __builtin_amdgcn_s_memrealtime();
#endif
}

// The first look at the flags, taken at the very top of a tile so that the load's round trip to the L2
// (~0.7 us) runs under the tile's prologue: the workgroup that finished the previous step last comes to
// its wait when everybody else has long published, and would otherwise pay that round trip in full on
// what is the critical path of the step.
__device__ __forceinline__ unsigned chain_peek(const ChainWait &w, int tid) {
    if (w.flags == nullptr || tid >= 64) return 0u;
    if (!w.peek) return 0u;
    return tid < w.count ? __hip_atomic_load(w.flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
}

// How the tile code reads a layer's arguments: `a->field`, and a.in() / a.res() / a.out() for the three buffers.
// ArgsDirect: the kernel's own parameter (loads from the kernel-argument segment, placed where the values are
// first used or demanded).  ArgsChain: a record of a chain's step list in constant memory, read the same way,
// field by field where needed; copied whole into registers at the top of a step, its ~56 words stayed live
// across the tile and the compiler kept moving them between SGPRs and VGPR lanes (500 lane moves in the
// chained kernel, none in the others).  in / res / out hold byte offsets into the chain's workspace there;
// `gaps` = gap_in[2], gap_res[2], gap_out[2] of the step, `which` the layer of the step.
struct ArgsDirect {
    const ConvKArgs &r;
    __device__ __forceinline__ const ConvKArgs *operator->() const { return &r; }
    __device__ __forceinline__ const void *in() const { return r.in; }
    __device__ __forceinline__ const void *res() const { return r.res; }
    __device__ __forceinline__ void *out() const { return r.out; }
};
struct ArgsChain {
    typedef const __attribute__((address_space(4))) ConvKArgs *args_ptr;
    typedef const __attribute__((address_space(4))) unsigned *gaps_ptr;
    args_ptr p;
    gaps_ptr gaps;
    char *ws;
    int share, which;
    __device__ __forceinline__ args_ptr operator->() const { return p; }
    __device__ __forceinline__ const void *in() const { return ws + (size_t)p->in + (size_t)share * gaps[which]; }
    __device__ __forceinline__ const void *res() const {
        const void *const r = p->res;
        return r == (const void *)~(uintptr_t)0 ? nullptr : (const void *)(ws + (size_t)r + (size_t)share * gaps[2 + which]);
    }
    __device__ __forceinline__ void *out() const { return ws + (size_t)p->out + (size_t)share * gaps[4 + which]; }
};

// What a workgroup needs to know about its phase (all wave-uniform).
struct PhaseInfo {
    int M, hm, wm;      // rows of this phase = n_images * hm * wm
    int off_y, off_x;   // output pixel = (my * out_mul + off_y, mx * out_mul + off_x)
    unsigned hw_mul, hw_shift, wm_mul, wm_shift;  // FastDiv by hm * wm and by wm
};

// n / d for 0 <= n < 2^31 (see FastDiv)
__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned shift) {
    return (int)(((unsigned long long)(unsigned)n * mul) >> shift);
}

// Taps that fall into the padding (and tail rows past M) read from here instead of
// branching around the load: a conditional load would make the compiler wait for the whole
// prefetch at the branch join, i.e. before the MFMAs it is supposed to overlap with.
constexpr int MAX_CIN = 4096;
__device__ __attribute__((aligned(128))) float g_zero_page[MAX_CIN + 32] = {};

#ifdef GDPSO_STAMPS
// Diagnostic build only (tools/stamp_probe.py): per-workgroup time stamps of the tile code.
__device__ unsigned long long *g_stamps = nullptr;

#define GDPSO_STAMP(SLOT)                                                                   \
    if (g_stamps && threadIdx.x == 0) {                                                     \
        g_stamps[(size_t)blockIdx.x * 16 + (SLOT)] = __builtin_amdgcn_s_memtime();           \
        if ((SLOT) == 0) g_stamps[(size_t)blockIdx.x * 16 + 7] = __builtin_amdgcn_s_memrealtime(); \
        if ((SLOT) == 0)                                                                    \
            g_stamps[(size_t)blockIdx.x * 16 + 5] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | /* HW_REG_HW_ID */ \
                                                   ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); /* HW_REG_XCC_ID */ \
        if ((SLOT) == 3) g_stamps[(size_t)blockIdx.x * 16 + 6] = __builtin_amdgcn_s_memrealtime(); \
    }
#else
#define GDPSO_STAMP(SLOT)
#endif

// Activations as one branch-free formula, max(v, 0) + slope * min(v, 0) with slope 1 (none), 0 (ReLU)
// or 0.2 (LeakyReLU): the same values as the three definitions (one of the two terms is always a
// zero), and the store passes need no branch on a run-time activation code per pass.
__device__ __forceinline__ float act_slope(int act) { return act == ACT_RELU ? 0.f : (act == ACT_LEAKY02 ? 0.2f : 1.f); }
__device__ __forceinline__ float apply_slope(float v, float slope) { return fmaxf(v, 0.f) + slope * fminf(v, 0.f); }
__device__ __forceinline__ float apply_act(float v, int act) { return apply_slope(v, act_slope(act)); }

// What a thread needs for its part of the store pass, gathered BEFORE the accumulators go to
// LDS: the loads of scale, shift and (unsplit layers) the residual are then in flight across the
// LDS round trip and its barrier instead of starting behind them.
template <int BM, int BN>
struct StorePlan {
    static constexpr int C4 = BN / 4;     // float4 per tile row
    static constexpr int RPP = 256 / C4;  // tile rows per pass
    static constexpr int PASSES = BM / RPP;
    static constexpr bool PREFETCH_RES = PASSES <= 8;  // 4 registers a pass
    int off[PASSES];  // element offset of this thread's 4 channels in each of its rows (-1: row past M)
    f32x4 sc, sh;
    f32x4 rres[PREFETCH_RES ? PASSES : 1];
};

template <typename E, int BM, int BN, bool CHAIN = false, typename ARGS>
__device__ __forceinline__ void conv_store_plan(const ARGS &a, int tid, int m0, int n0, const PhaseInfo &P,
                                                StorePlan<BM, BN> &sp) {
    using Plan = StorePlan<BM, BN>;
    const int hw = P.hm * P.wm;
    const int c4 = tid % Plan::C4, r0 = tid / Plan::C4;
    const int col = n0 + c4 * 4;
    sp.sc = *(global_f32x4_ptr)(a->scale + col);
    sp.sh = *(global_f32x4_ptr)(a->shift + col);
    const bool remap = a->out_mul != 1 || P.hm != a->hout || P.wm != a->wout;
    // 32-bit offsets are enough, the host refuses outputs of 2 GiB and more
#pragma unroll
    for (int pass = 0; pass < Plan::PASSES; ++pass) {
        const int m = m0 + pass * Plan::RPP + r0;
        // straight-line (a branch per pass made the compiler re-fetch its arguments behind every one)
        const int n = fast_div(m, P.hw_mul, P.hw_shift), rr = m - n * hw;
        const int my = fast_div(rr, P.wm_mul, P.wm_shift), mx = rr - my * P.wm;
        const int pix = remap ? (n * a->hout + my * a->out_mul + P.off_y) * a->wout + mx * a->out_mul + P.off_x : m;
        sp.off[pass] = m < P.M ? pix * a->cout + col : -1;
    }
    if constexpr (Plan::PREFETCH_RES) {
        // rows past M read the first row's residual (any valid address; the value is not used)
        const bool want = a->splits == 1 && a.res() != nullptr;
        if constexpr (CHAIN) {
            // written earlier in this launch by a workgroup of the same XCD: read past the L1 (sc1)
            const __amdgpu_buffer_rsrc_t rsrc_res = __builtin_amdgcn_make_buffer_rsrc((void *)a.res(), 0, (int)(a->slab * sizeof(E)), 0x00020000);
#pragma unroll
            for (int pass = 0; pass < Plan::PASSES; ++pass) {
                sp.rres[pass] = 0.f;
                if (want) {
                    const int boff = (sp.off[pass] < 0 ? col : sp.off[pass]) * (int)sizeof(E);
                    if constexpr (is_f32<E>()) {
                        sp.rres[pass] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, boff, 0, 16 /* sc1 */));
                    } else {  // four 16-bit elements
                        using V = __attribute__((ext_vector_type(4))) E;
                        using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
                        const V v = __builtin_bit_cast(V, (u32x2)__builtin_amdgcn_raw_buffer_load_b64(rsrc_res, boff, 0, 16 /* sc1 */));
                        sp.rres[pass] = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
                    }
                }
            }
        } else {
#pragma unroll
            for (int pass = 0; pass < Plan::PASSES; ++pass) {
                sp.rres[pass] = 0.f;
                if (want) sp.rres[pass] = load4<E>((const E *)a.res() + (sp.off[pass] < 0 ? col : sp.off[pass]));
            }
        }
    }
}

// Epilogue shared by the convolution kernels.  `lds` is the workgroup's whole LDS array
// (lds_floats long); every operand read from it must have been retired by a barrier.
// The tile has been laid out row-major in LDS ([BM][BN + 4], starting at `lds`) and a barrier
// has passed; it is written with 16-byte stores: one wave instruction covers whole 256/512-byte
// row segments, and the residual is read the same way.
// KP > 1: the tile arrives as KP partial tiles (regions of BM x (BN + 4) floats, one per K part
// of the workgroup), which are added in the order 0, 1, ... as they are read.
template <typename E, int BM, int BN, int KP = 1, typename ARGS>
__device__ __forceinline__ void conv_store_tile(const ARGS &a, float *lds, int lds_floats, int tid, int sp, int tile,
                                                const StorePlan<BM, BN> &plan) {
    E *const out = (E *)a.out();
    const E *const res = (const E *)a.res();
    using Plan = StorePlan<BM, BN>;
    constexpr int LDC = BN + 4;
    float *sC = lds;
    auto tile_quad = [&](int row) {
        f32x4 v = *(const f32x4 *)&sC[row * LDC + (tid % (BN / 4)) * 4];
#pragma unroll
        for (int q = 1; q < KP; ++q) v += *(const f32x4 *)&sC[q * BM * LDC + row * LDC + (tid % (BN / 4)) * 4];
        return v;
    };
    constexpr int RPP = Plan::RPP, PASSES = Plan::PASSES;
    const int r0 = tid / Plan::C4;
    const bool direct = a->splits == 1;
    const f32x4 sc = plan.sc, sh = plan.sh;
    const int(&off)[PASSES] = plan.off;
    if (direct) {
        const float slope = act_slope(a->act);
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            if constexpr (Plan::PREFETCH_RES) {
                // every row computed, only the store masked: no branch ahead of the LDS read
                f32x4 v = tile_quad(pass * RPP + r0);
                v = v * sc + sh + plan.rres[pass];  // zeros without a residual
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = apply_slope(v[q], slope);
                if (off[pass] >= 0) store4<E>(out + off[pass], v);
            } else {
                if (off[pass] < 0) continue;
                f32x4 v = tile_quad(pass * RPP + r0);
                v = v * sc + sh;
                if (res) v += load4<E>(res + off[pass]);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = apply_slope(v[q], slope);
                store4<E>(out + off[pass], v);
            }
        }
        return;
    }

    // Split-K: every split writes its partial tile to its slab; the workgroup that arrives
    // last at the tile's counter sums the slabs in split order (so the result does not
    // depend on who is last) and applies the epilogue.  Hand-off: write-through (sc1) 16-byte
    // slab stores, which need no L2 write-back; every wave drains its stores; workgroup
    // barrier; one lane adds to the counter.  The last arriver takes one agent-scope acquire
    // (drops this CU's stale L1 lines) before anyone loads.
    float *slab = a->part + (size_t)sp * a->slab;
    if (a->counters == nullptr) {  // reduction by the separate kernel
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass)
            if (off[pass] >= 0) *(f32x4 *)(slab + off[pass]) = tile_quad(pass * RPP + r0);
        return;
    }
    {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)(a->slab * 4), 0x00020000);
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass)
            if (off[pass] >= 0)
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(u32x4, tile_quad(pass * RPP + r0)), rsrc,
                    off[pass] * 4, 0, 16 /* sc1 */);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *s_last = (int *)&lds[lds_floats - 4];  // past the output tile
    if (tid == 0) {
        int *cnt = a->counters + tile;
        const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == a->splits - 1;
        if (last) {
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *s_last = last;
    }
    __syncthreads();
    if (!*s_last) return;
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        if (off[pass] < 0) continue;
        f32x4 v = *(const f32x4 *)(a->part + off[pass]);
#pragma unroll 4
        for (int k = 1; k < a->splits; ++k) v += *(const f32x4 *)(a->part + (size_t)k * a->slab + off[pass]);
        v = v * sc + sh;
        if (res) v += load4<E>(res + off[pass]);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = apply_act(v[q], a->act);
        store4<E>(out + off[pass], v);
    }
}

// One 32x32x16 matrix instruction on 16-bit operands: a lane's 16 bytes are its eight k
// (k = 8 * (lane >> 5) + j) of row / column lane & 31.
template <typename E>
__device__ __forceinline__ f32x16 mfma16(const f32x4 &a, const f32x4 &b, const f32x16 &c) {
    if constexpr (sizeof(E) == 4) {
        return c;  // not used by the fp32 instantiation
    } else if constexpr (__is_same(E, bf16_t)) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
}

// BM x BN output tile, K-tiles of BKT elements.  KW = 1: the four waves form a 2 x 2 grid over the
// tile.  KW = 4 (32 x 32): every wave takes a quarter of each K-tile of the one 32 x 32 block.
// KW = 2 (BM = 32): two waves side by side cover the tile and the other two take the
// second half of every K-tile; the two halves are added through LDS before the epilogue.  That
// shape doubles the number of workgroups of a small-M layer without split-K slabs.
// E: element type of activations and weights (float, bf16_t, f16_t); BKT: elements per K-tile.
// CHAIN: the tile is one of many that a persistent workgroup of conv_chain_kernel walks through.  `xcd` is
// then the share of the images the workgroup's team owns and `slot` the tile's index inside it; the
// activations (A operand, residual) were written earlier in the same launch by workgroups on the same
// XCD and are read with sc1 loads, which miss the CU's L1 and hit the XCD's L2; the tile's first
// activation load waits for `wait` (weights, tables and arguments are fetched ahead of the wait).
template <typename E, int BM, int BN, int BKT, int KW, bool FUSE = false, bool CHAIN = false, typename ARGS>
__device__ __forceinline__ void conv_tile(const ARGS &a, const int xcd, int slot, float *lds, const FuseArgs *fuse = nullptr,
                                          const ChainWait wait = ChainWait{}) {
    constexpr int VEC = vec_of<E>();  // elements per 16-byte vector
    constexpr bool F32 = is_f32<E>();
    static_assert(!FUSE || (BM == 64 && BN == 64 && KW == 1 && F32), "the fused tail is written for the fp32 64x64 tile");
    static_assert(BKT / KW == 8 * VEC, "every wave consumes eight 16-byte vectors of k per K-tile (32 floats, 64 half-width elements)");
    static_assert(KW == 1 || BM == 32, "the K-split layout is for 32-row tiles");
    constexpr int LDA = BKT + VEC;  // 144- or 272-byte rows: ds_read_b128 of consecutive rows is conflict-free
    static_assert(KW == 1 || BN * KW == 128, "K-split layouts: 32x64 over two wave pairs, 32x32 over four waves");
    constexpr int WM = KW == 1 ? BM / 64 : 1, WN = KW == 1 ? BN / 64 : 1;  // 32x32 MFMA blocks per wave
    constexpr int RP = 256 / (BKT / VEC);      // rows covered by one loader pass
    constexpr int PA = BM / RP, PB = BN / RP;  // loader passes
    // BD: the B operand (weights) never passes through LDS.  In every tile shape with one MFMA block per wave a
    // wave's B fragments are its own (the 32 columns of its block over its part of the K-tile), or shared with
    // one other wave only (64x64), so LDS staging buys no reuse and costs the staging stores, which are what
    // bounds a K loop that is alone on its CU (ds_write_b128 moves its VGPRs to the LDS at ~13 cycles a wave
    // instruction on a path the four SIMDs share: measured, 16 K-steps of the 32x32 tile take 12.6 us with the
    // stores and 8.3 us without, against 7.7 us of MFMA time).  The weights are kept a second time in
    // chunk-major order, [32-column block][16-byte chunk of K][32 columns][VEC elements], in which the 64 lanes
    // of a wave's fragment load (lane = column, lane half = odd / even chunk) read 1 KiB contiguously, and go
    // from global memory straight into the registers the MFMAs read.
    constexpr bool BD = GDPSO_BDIRECT && WM * WN == 1 && KW > 1;
    constexpr int LDS_FLOATS = 2 * (BM + BN) * LDA * (int)sizeof(E) / 4;
    static_assert(LDS_FLOATS >= BM * (BN + 4) + 4, "the output tile is staged in the operand buffers");
    E *sA = (E *)lds;
    E *sB = sA + 2 * BM * LDA;
    const E *const in = (const E *)a.in();

#ifdef GDPSO_STAMPS
    // a chain stamps the tiles of one chosen step only; `g_stamps` below is then this pointer, already offset
    // so that adding the workgroup's slot lands in its row
    unsigned long long *const g_stamps = CHAIN ? (wait.tile_stamps ? wait.tile_stamps - (size_t)blockIdx.x * 16 : nullptr) : gdpso::g_stamps;
#endif
    GDPSO_STAMP(0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned flags_early = 0u;
    if constexpr (CHAIN) flags_early = chain_peek(wait, tid);
    const int wr = KW == 1 ? wave >> 1 : 0, wc = KW == 4 ? 0 : wave & 1, kh = KW == 1 ? 0 : (KW == 2 ? wave >> 1 : wave);
    const int l31 = lane & 31, half = lane >> 5;

    // Workgroups b and b+8 share an XCD (and its L2).  Each XCD gets a contiguous eighth of the
    // tiles of EVERY phase, phase after phase: contiguous, so that tiles that share input rows or
    // weight columns hit the same L2; of every phase, because phases differ in cost (a border
    // class has fewer taps than the interior) and an XCD holding only interior tiles would finish
    // last.  A phase's tile count is rounded up to a multiple of 8; the surplus workgroups exit.
    // Everything the prologue reads from the kernel arguments is demanded here, in one statement
    // (30 operands at most): fetched where first used, the arguments cost the prologue a chain of
    // scalar-load round trips of ~0.25 us each ahead of the first operand load.
#define GDPSO_W(I) "s"(a->phase_word[I])
    static_assert(MAX_PHASES == 16, "the operand list names 16 phases");
    asm volatile("" ::"s"(a.in()), "s"(a->wt), "s"(a->wtq), "s"(a->table), "s"(a->n_images), "s"(a->hin), "s"(a->win), "s"(a->cin),
                 "s"(a->in_stride), "s"(a->splits), "s"(a->tiles_n), "s"(a->per_xcd), "s"(a->snake), "s"(a->tiles_n_mul),
                 "s"(a->tiles_n_shift), GDPSO_W(0), GDPSO_W(1), GDPSO_W(2), GDPSO_W(3), GDPSO_W(4), GDPSO_W(5), GDPSO_W(6),
                 GDPSO_W(7), GDPSO_W(8), GDPSO_W(9), GDPSO_W(10), GDPSO_W(11), GDPSO_W(12), GDPSO_W(13), GDPSO_W(14),
                 GDPSO_W(15));
#undef GDPSO_W
    const int tile = slot * 8 + xcd;  // the launch's workgroup index (split-K arrival counter)
    // An XCD deals its workgroups round-robin to its CUs: with two per CU and phases in order of
    // decreasing cost, CU i would get the i-th heaviest and the (i + CUs)-th heaviest.  Dealing the
    // second round from the light end pairs heavy with light.
    if (a->snake && slot >= a->snake) slot = a->per_xcd - 1 - (slot - a->snake);
    // The phase of this slot: the first whose running end exceeds it.  The ends increase, so
    // `past` holds for p < ph exactly and the xor chain telescopes to phase_word[ph] (a chain of
    // selects over the array would be turned into a load indexed by ph, from a scratch copy).
    int ph = 0, base = 0;
    unsigned sel = a->phase_word[0];
    if (a->phase_word[1] != 0xffffffu) {  // single-phase layers (every 1x1 convolution) have nothing to search
#pragma unroll
        for (int p = 0; p < MAX_PHASES - 1; ++p) {
            const int end_p = (int)(a->phase_word[p] & 0xffffffu);
            // integer forms throughout, so that the chain stays on the scalar unit
            const int past = (int)((unsigned)(end_p - 1 - slot) >> 31);  // slot >= end_p
            ph += past;
            base = max(base, end_p & -past);
            sel ^= (a->phase_word[p + 1] ^ a->phase_word[p]) & (unsigned)-past;
        }
    }
    const int per_xcd = (int)(sel & 0xffffffu) - base;
    // a chain counts a phase's tiles per share (its slot is the tile's index inside the share's rows of the phase)
    const int idx = CHAIN ? slot - base : xcd * per_xcd + (slot - base);
    if (!CHAIN && idx >= 8 * per_xcd - (int)(sel >> 24)) return;
    const int tml = fast_div(idx, a->tiles_n_mul, a->tiles_n_shift), tn = idx - tml * a->tiles_n;
    int m0 = tml * BM;
    const int n0 = tn * BN;
    const int sp = CHAIN ? 0 : blockIdx.z;
    // The table is read through the constant address space: that keeps its loads scalar whatever
    // else the kernel does (behind an asm statement or a store the compiler assumes that global
    // memory may have changed, and falls back to vector loads).
    typedef const __attribute__((address_space(4))) int *const_int_ptr;
    const const_int_ptr tab = (const_int_ptr)(uintptr_t)(a->table + ph * PHASE_TABLE_INTS);
    int hdr[PHASE_HDR + 1];
#pragma unroll
    for (int i = 0; i <= PHASE_HDR; ++i) hdr[i] = tab[i];  // (the first tap comes with the header: one scalar-load round trip fewer)
    const int ntaps = hdr[0];
    const int tap0_packed = hdr[PHASE_HDR];  // read with the header: one scalar-load round trip fewer
    PhaseInfo P;
    P.off_y = hdr[1];
    P.off_x = hdr[2];
    P.hm = hdr[4];
    P.wm = hdr[5];
    P.M = a->n_images * P.hm * P.wm;
    P.hw_mul = (unsigned)hdr[9];
    P.wm_mul = (unsigned)hdr[10];
    P.hw_shift = (unsigned)hdr[11] & 0xff;
    P.wm_shift = ((unsigned)hdr[11] >> 8) & 0xff;
    const int in_y0 = hdr[6], in_x0 = hdr[7], wstride = hdr[8];
    if constexpr (CHAIN) {
        // the share's rows of this phase: tiles start at its first row, the last one may be partly (or, in
        // the last shares of a batch that does not divide by the number of shares, wholly) past its end
        const int share_rows = a->share_images * P.hm * P.wm;
        m0 += xcd * share_rows;
        P.M = min(P.M, (xcd + 1) * share_rows);
        if (m0 >= P.M) {
            // An empty tile (past the share's rows of this phase) still takes the step's wait when it is the
            // workgroup's first of the step: the caller hands the wait to the first tile only, and a later
            // tile of the same workgroup may lie in the non-empty head of the next phase.
            chain_wait(wait, tid, flags_early);
            return;
        }
    }
#ifdef GDPSO_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(wstride), "s"(ntaps) : "memory");  // the phase header has arrived
    GDPSO_STAMP(8)
#endif
    const E *wt = (const E *)a->wt + hdr[3];
    const int ktot = ntaps * a->cin;
    // this phase's K range per split (the divide only where K is split at all)
    const int ksplit = a->splits == 1 ? ktot : ((ktot / BKT + a->splits - 1) / a->splits) * BKT;
    const int kbeg = sp * ksplit;
    const int kend = min(ktot, kbeg + ksplit);

    // loader coordinates: BKT/VEC threads x 16 B cover one row of a K-tile
    const int lrow = tid / (BKT / VEC), lcol = (tid % (BKT / VEC)) * VEC;
    int a_iy0[PA], a_ix0[PA], a_img[PA];
    bool a_ok[PA];
    const int hw = P.hm * P.wm;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m0 + lrow + RP * p;
        a_ok[p] = m < P.M;
        const int mm = a_ok[p] ? m : 0;
        const int n = fast_div(mm, P.hw_mul, P.hw_shift), r = mm - n * hw;
        const int my = fast_div(r, P.wm_mul, P.wm_shift), mx = r - my * P.wm;
        a_iy0[p] = my * a->in_stride + in_y0;
        a_ix0[p] = mx * a->in_stride + in_x0;
        a_img[p] = n * a->hin * a->win;
    }
    // Gather pointers advance by one K-tile (32 channels) per load and are recomputed only
    // when the K range crosses into the next tap.  Rows whose tap falls in the padding (only
    // in layers too irregular to be split into border classes) and tail rows past M point
    // into the zero page and advance like the others.  A tap entry is
    // dy | dx << 8 | (position of the tap in the weight rows) << 16.
    const E *aptr[PA], *bptr[PB];
    const E *brow[PB];
    // CHAIN: the A operand goes through buffer loads (byte offsets from a.in(); the sc1 bit makes them miss
    // the L1); rows in the padding point past the end of the buffer, where a buffer load returns zeros
    unsigned aoff[PA];
    constexpr unsigned A_OUTSIDE = 0xC0000000u;  // stays out of range however far the K loop advances it
    __amdgpu_buffer_rsrc_t rsrc_in;
    if constexpr (CHAIN)
        rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, a->n_images * a->hin * a->win * a->cin * (int)sizeof(E), 0x00020000);
#pragma unroll
    for (int p = 0; p < PB; ++p) brow[p] = wt + ((n0 + lrow + RP * p) * wstride + lcol);  // 32-bit offsets: see conv_args
    // BD: this lane's chunk 0 of the K-tile at weight-row position 0: column block (n0 + wc * BN / 2) / 32, column
    // l31, chunk kh * 8 + half of the K-tile; a chunk of K is 32 * VEC elements further on, a lane's four
    // fragments of a K-tile (chunks 2 g + half) 64 * VEC elements apart
    const E *bq0 = nullptr, *bqp = nullptr;
    if constexpr (BD)
        bq0 = (const E *)a->wtq + hdr[3] + ((size_t)((n0 + wc * (BN / 2)) >> 5) * (wstride / VEC) * 32 + l31 + (kh * 8 + half) * 32) * VEC;
    int tap = a->splits == 1 ? 0 : kbeg / a->cin, ci0 = kbeg - tap * a->cin;
#define GDPSO_SET_TAP(CI, PACKED)                                                                               \
    {                                                                                                           \
        const int packed_ = (PACKED);                                                                           \
        const int dy_ = (int)(signed char)(packed_ & 0xff), dx_ = (int)(signed char)((packed_ >> 8) & 0xff);    \
        const int wk_ = (packed_ >> 16) * a->cin + (CI);                                                         \
        _Pragma("unroll") for (int p = 0; p < PA; ++p) {                                                        \
            const int iy = a_iy0[p] + dy_, ix = a_ix0[p] + dx_;                                                 \
            const bool ok = a_ok[p] && (unsigned)iy < (unsigned)a->hin && (unsigned)ix < (unsigned)a->win;        \
            if constexpr (CHAIN)                                                                                \
                aoff[p] = ok ? (unsigned)((a_img[p] + iy * a->win + ix) * a->cin + (CI) + lcol) * (unsigned)sizeof(E) : A_OUTSIDE; \
            else                                                                                                \
                aptr[p] = ok ? in + ((a_img[p] + iy * a->win + ix) * a->cin + (CI) + lcol)                        \
                             : (const E *)g_zero_page + (CI) + lcol;                                            \
        }                                                                                                       \
        if constexpr (BD)                                                                                       \
            bqp = bq0 + wk_ * 32;                                                                               \
        else                                                                                                    \
            _Pragma("unroll") for (int p = 0; p < PB; ++p) bptr[p] = brow[p] + wk_;                             \
    }
    // Prefetch registers for the next K-tile.  Native vector types over fully unrolled loops:
    // with HIP's float4 struct these arrays were not promoted to registers (they went to
    // scratch, which put a wait behind every prefetch).
    // Tiles of one MFMA block per wave spend too little time in a K-step (16 MFMAs, ~0.5 us) to
    // cover a load's latency: they keep TWO K-tiles in flight, in two register sets that are
    // always named by literal index (a run-time index would put the arrays in scratch).
    constexpr bool PF2 = WM * WN == 1;
    constexpr int NSET = PF2 ? GDPSO_PF : 1;  // K-tiles in flight in registers
    static_assert(GDPSO_PF == 2 || GDPSO_PF == 3, "two or three register sets");
    f32x4 ra[NSET][PA], rb[NSET][BD ? 4 : PB];  // BD: rb[set][g] is the wave's B fragment of MFMA group g
    // A K-tile's loads (straight-line, so that a K-step can interleave them with its MFMAs) and,
    // separately, the move of the gather pointers to the next K-tile (which branches at a tap change).
#define GDPSO_GISSUE_A(S)                                                                                       \
    {                                                                                                           \
        _Pragma("unroll") for (int p = 0; p < PA; ++p) {                                                        \
            if constexpr (CHAIN)                                                                                \
                ra[S][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, (int)aoff[p], 0, 16 /* sc1 */)); \
            else                                                                                                \
                ra[S][p] = *(global_f32x4_ptr)aptr[p];                                                          \
        }                                                                                                       \
    }
#define GDPSO_GISSUE_B(S)                                                                                       \
    {                                                                                                           \
        if constexpr (BD) {                                                                                     \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) rb[S][g] = *(global_f32x4_ptr)(bqp + g * 64 * VEC);    \
        } else {                                                                                                \
            _Pragma("unroll") for (int p = 0; p < PB; ++p) rb[S][p] = *(global_f32x4_ptr)bptr[p];                \
        }                                                                                                       \
    }
    // BD: fragment G of the K-tile two ahead, into the registers the MFMAs of group G have just read
#define GDPSO_GISSUE_BG(S, G) rb[S][G] = *(global_f32x4_ptr)(bqp + (G) * 64 * VEC);
#define GDPSO_GISSUE(S)                                                                                         \
    {                                                                                                           \
        GDPSO_GISSUE_A(S)                                                                                       \
        GDPSO_GISSUE_B(S)                                                                                       \
    }
#define GDPSO_GADVANCE()                                                                                        \
    {                                                                                                           \
        _Pragma("unroll") for (int p = 0; p < PA; ++p) {                                                        \
            if constexpr (CHAIN)                                                                                \
                aoff[p] += BKT * (unsigned)sizeof(E);                                                           \
            else                                                                                                \
                aptr[p] += BKT;                                                                                 \
        }                                                                                                       \
        if constexpr (BD)                                                                                       \
            bqp += BKT * 32;                                                                                    \
        else                                                                                                    \
            _Pragma("unroll") for (int p = 0; p < PB; ++p) bptr[p] += BKT;                                      \
        ci0 += BKT;                                                                                             \
        if (ci0 == a->cin) {                                                                                     \
            ci0 = 0;                                                                                            \
            if (++tap < ntaps) GDPSO_SET_TAP(0, tab[PHASE_HDR + tap])                                           \
        }                                                                                                       \
    }
#define GDPSO_GLOAD(S)                                                                                          \
    {                                                                                                           \
        GDPSO_GISSUE(S)                                                                                         \
        GDPSO_GADVANCE()                                                                                        \
    }
#define GDPSO_SSTORE(BUF, S)                                                                                    \
    {                                                                                                           \
        _Pragma("unroll") for (int p = 0; p < PA; ++p)                                                          \
            *(f32x4 *)(void *)&sA[(BUF) * BM * LDA + (lrow + RP * p) * LDA + lcol] = ra[S][p];                  \
        if constexpr (!BD) {                                                                                    \
            _Pragma("unroll") for (int p = 0; p < PB; ++p)                                                      \
                *(f32x4 *)(void *)&sB[(BUF) * BN * LDA + (lrow + RP * p) * LDA + lcol] = rb[S][p];              \
        }                                                                                                       \
    }
    // Operand fragments of MFMA group G (two 16-byte vectors of k: one per lane half) into register set SET.
#define GDPSO_LOADFRAG(SET, G)                                                                                  \
    {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < WM; ++i) av[SET][i] = *(const f32x4 *)(const void *)(pa + i * 32 * LDA + (G) * 2 * VEC); \
        if constexpr (!BD) {                                                                                    \
            _Pragma("unroll") for (int j = 0; j < WN; ++j) bv[SET][j] = *(const f32x4 *)(const void *)(pb + j * 32 * LDA + (G) * 2 * VEC); \
        }                                                                                                       \
    }

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = kbeg < kend ? (kend - kbeg) / BKT : 0;  // K-tiles of this workgroup
    if (nk > 0) {
        GDPSO_SET_TAP(ci0, tap == 0 ? tap0_packed : tab[PHASE_HDR + tap])
        if constexpr (CHAIN) {
            // weights first; the activations only once the layer that writes them has finished in this share
            GDPSO_GISSUE_B(0)
            GDPSO_STAMP(11)  // weights requested, the wait begins
            chain_wait(wait, tid, flags_early);
            GDPSO_STAMP(12)
            GDPSO_GISSUE_A(0)
            GDPSO_GADVANCE()
        } else {
            GDPSO_GLOAD(0)
        }
#ifdef GDPSO_STAMPS
        GDPSO_STAMP(9)  // first loads issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GDPSO_STAMP(10)  // ... and returned
#endif
        GDPSO_SSTORE(0, 0)
        // the second K-tile only now: issued ahead of the first one's LDS stores it would share the
        // CU's 64 B/clk load path with it and delay the first MFMA
        if constexpr (PF2) {
            if (nk > 1) GDPSO_GLOAD(1)
            if constexpr (NSET == 3) {
                if (nk > 2) GDPSO_GLOAD(2)
            }
        }
    }
    // What the epilogue reads from the kernel arguments is demanded here, so that it arrives during
    // the K loop: fetched where first used, each argument costs the epilogue a scalar-load round trip.
    if constexpr (!FUSE)
        asm volatile("" ::"s"(a->scale), "s"(a->shift), "s"(a.res()), "s"(a.out()), "s"(a->part), "s"(a->counters), "s"(a->cout),
                     "s"(a->hout), "s"(a->wout), "s"(a->out_mul), "s"(a->act), "s"(a->slab));
    __syncthreads();
    GDPSO_STAMP(1)
    // One K-tile.  Within it the 32 k's of a wave are consumed in a permuted order: MFMA step s of
    // group g takes k = 8g + 4*half + s from both operands, so each lane's four steps are one
    // 16-byte LDS read; the sum over k is the same set of products.  The fragments of group g+1
    // are read while the MFMAs of group g issue, and (MORE) the prefetched next K-tile moves from
    // registers to the other LDS buffer ahead of the last group, in the shadow of the MFMAs that
    // are still executing, instead of after them.  The sched_group_barrier sequence pins that
    // order (0x100 = DS read, 0x200 = DS write, 0x008 = MFMA).
    // Diagnostic builds may compile parts of the K loop out (-DGDPSO_SKIP_BITS=n; bit 0: its MFMAs,
    // bit 1: its global loads, bit 2: its LDS stores): what is left shows what the skipped part
    // costs; results are garbage.
#ifdef GDPSO_SKIP_BITS
#define GDPSO_DBG(BIT) (!((GDPSO_SKIP_BITS) & (BIT)))
#else
#define GDPSO_DBG(BIT) true
#endif
    // Waves with one MFMA block (16 MFMAs a K-step) take the pinning further: every other
    // instruction of the step sits between two MFMAs of the dependent chain, where it issues while
    // the previous MFMA executes: the K-tile's global loads behind MFMAs 1-3 and 5-7, the fragment
    // reads behind 0, 4 and 8, the LDS stores behind 9-12, so that they have drained when the
    // last MFMA issues and the barrier follows it directly.  With a single wave per SIMD (one
    // workgroup per CU) nothing else would cover them: measured there, the loads, the stores and the
    // MFMAs of a K-step simply added up.
#define GDPSO_SGB(MASK, N) \
    if ((N) > 0) __builtin_amdgcn_sched_group_barrier(MASK, (N) > 0 ? (N) : 1, 0);
#define GDPSO_SGB4(MA, NA, MB, NB) \
    GDPSO_SGB(MA, NA) GDPSO_SGB(MB, NB) GDPSO_SGB(MA, NA) GDPSO_SGB(MB, NB) GDPSO_SGB(MA, NA) GDPSO_SGB(MB, NB) GDPSO_SGB(MA, NA) GDPSO_SGB(MB, NB)
#define GDPSO_KSTEP(LOAD, LS, MORE, SS, BS)                                                                     \
    {                                                                                                           \
        if ((LOAD) && GDPSO_DBG(2)) {                                                                           \
            if constexpr (BD) GDPSO_GISSUE_A(LS) else GDPSO_GISSUE(LS)                                          \
        }                                                                                                       \
        const E *pa = sA + buf * BM * LDA + (wr * (BM / 2) + l31) * LDA + half * VEC + kh * 8 * VEC;           \
        const E *pb = sB + buf * BN * LDA + (wc * (BN / 2) + l31) * LDA + half * VEC + kh * 8 * VEC;           \
        f32x4 av[2][WM], bv[2][WN];                                                                             \
        GDPSO_LOADFRAG(0, 0)                                                                                    \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                         \
            if (g < 3) GDPSO_LOADFRAG((g + 1) & 1, g + 1)                                                       \
            if (g == 3 && (MORE) && GDPSO_DBG(4)) GDPSO_SSTORE(buf ^ 1, SS)                                     \
            if constexpr (BD) {                                                                                 \
                /* B: fragment g of this K-tile straight from the registers its load filled (set BS) */        \
                if constexpr (F32) {                                                                            \
                    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                               \
                        if (GDPSO_DBG(1))                                                                       \
                            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][0][s], rb[BS][g][s], acc[0][0], 0, 0, 0); \
                } else {                                                                                        \
                    acc[0][0] = mfma16<E>(av[g & 1][0], rb[BS][g], acc[0][0]);                                  \
                }                                                                                               \
                if ((LOAD) && GDPSO_DBG(2)) GDPSO_GISSUE_BG(BS, g)                                              \
            } else if constexpr (F32) {                                                                         \
                _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                   \
                    _Pragma("unroll") for (int i = 0; i < WM; ++i)                                              \
                        _Pragma("unroll") for (int j = 0; j < WN; ++j)                                          \
                            if (GDPSO_DBG(1))                                                                   \
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][i][s], bv[g & 1][j][s], acc[i][j], 0, 0, 0); \
            } else {                                                                                            \
                _Pragma("unroll") for (int i = 0; i < WM; ++i)                                                  \
                    _Pragma("unroll") for (int j = 0; j < WN; ++j)                                              \
                        acc[i][j] = mfma16<E>(av[g & 1][i], bv[g & 1][j], acc[i][j]);                           \
            }                                                                                                   \
        }                                                                                                       \
        if constexpr (!F32) {                                                                                   \
            /* 16-bit: a K-step is 4 (or 16) short MFMAs; the compiler's own schedule is kept */                 \
        } else if constexpr (BD) {                                                                              \
            /* one A fragment read per MFMA group, the A loads of the K-tile two ahead behind MFMAs 1-3, each  */ \
            /* B fragment reloaded behind the group that has just read it, the A stores behind MFMAs 9-12      */ \
            constexpr int NA = (LOAD) ? PA : 0, NB = (LOAD) ? 1 : 0, NW = (MORE) ? PA : 0;                      \
            GDPSO_SGB(0x100, 1)                                                                                 \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 1)                                                             \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, (NA + 2) / 3)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, (NA + 1) / 3)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, NA / 3)                                                        \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 1) GDPSO_SGB(0x020, NB)                                        \
            GDPSO_SGB(0x008, 3)                                                                                 \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 1) GDPSO_SGB(0x020, NB)                                        \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, (NW + 3) / 4)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, (NW + 2) / 4)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, (NW + 1) / 4)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, NW / 4) GDPSO_SGB(0x020, NB)                                   \
            GDPSO_SGB(0x008, 3)                                                                                 \
            GDPSO_SGB(0x020, NB)                                                                                \
        } else if constexpr (WM * WN == 1) {                                                                    \
            constexpr int NV = (LOAD) ? PA + PB : 0, NW = (MORE) ? PA + PB : 0;                                 \
            GDPSO_SGB(0x100, 2)                                                                                 \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 2)                                                             \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, (NV + 5) / 6)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, (NV + 4) / 6)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, (NV + 3) / 6)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 2)                                                             \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, (NV + 2) / 6)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, (NV + 1) / 6)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, NV / 6)                                                        \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 2)                                                             \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, (NW + 3) / 4)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, (NW + 2) / 4)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, (NW + 1) / 4)                                                  \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, NW / 4)                                                        \
            GDPSO_SGB(0x008, 3)                                                                                 \
        } else if constexpr (WM * WN == 2) {                                                                    \
            /* 1 x 2 blocks (64x128): 32 MFMAs, 8 a group; the next group's three fragment reads behind the   */ \
            /* group's first three, then the K-tile's six loads (groups 0, 1) or LDS stores (groups 2, 3)     */ \
            static_assert(WM == 1 && WN == 2 && PA + PB == 6, "pinned for the 64x128 tile");                    \
            constexpr int NV = (LOAD) ? 1 : 0, NW = (MORE) ? 1 : 0;                                             \
            GDPSO_SGB(0x100, 3)                                                                                 \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 1) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 1) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 1) \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, NV) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, NV) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, NV) \
            GDPSO_SGB(0x008, 2)                                                                                 \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 1) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 1) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 1) \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, NV) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, NV) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x020, NV) \
            GDPSO_SGB(0x008, 2)                                                                                 \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 1) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 1) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x100, 1) \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, NW) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, NW) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, NW) \
            GDPSO_SGB(0x008, 2)                                                                                 \
            GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, NW) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, NW) GDPSO_SGB(0x008, 1) GDPSO_SGB(0x200, NW) \
            GDPSO_SGB(0x008, 5)                                                                                 \
        } else {                                                                                                \
            /* 2 x 2 blocks: 64 MFMAs in pairs; per group of 16: the next group's four fragment reads    */    \
            /* behind the first four pairs, then a load (groups 0, 1) or two LDS stores (groups 2, 3)     */    \
            /* behind each of the other four                                                              */    \
            static_assert(WM == 2 && WN == 2 && PA + PB == 8, "pinned for the 128x128 tile");                   \
            constexpr int NV = (LOAD) ? 1 : 0, NW = (MORE) ? 1 : 0;                                             \
            GDPSO_SGB(0x100, 4)                                                                                 \
            GDPSO_SGB4(0x008, 2, 0x100, 1) GDPSO_SGB4(0x008, 2, 0x020, NV)                                      \
            GDPSO_SGB4(0x008, 2, 0x100, 1) GDPSO_SGB4(0x008, 2, 0x020, NV)                                      \
            GDPSO_SGB4(0x008, 2, 0x100, 1) GDPSO_SGB4(0x008, 2, 0x200, NW)                                      \
            GDPSO_SGB4(0x008, 2, 0x200, NW) GDPSO_SGB(0x008, 8)                                                 \
        }                                                                                                       \
        __syncthreads();                                                                                        \
        buf ^= 1;                                                                                               \
        if (LOAD) GDPSO_GADVANCE()                                                                              \
    }
    int buf = 0;
    if constexpr (PF2 && NSET == 3) {
        // K-tile t travels through register set t % 3: step t issues the loads of tile t + 3 into the set that
        // tile t left (its A half went to LDS a step ago, its B fragments are reloaded group by group as the
        // MFMAs read them), and moves the A half of tile t + 1, loaded two steps ago, into LDS.
        int t = 0;
        for (; t + 5 < nk; t += 3) {
            GDPSO_KSTEP(true, 0, true, 1, 0)
            GDPSO_KSTEP(true, 1, true, 2, 1)
            GDPSO_KSTEP(true, 2, true, 0, 2)
        }
        const int left = nk - t;  // 0 (empty K range) to 5; t is a multiple of 3
#define GDPSO_TAIL(I)                                                                         \
    if (left > (I)) {                                                                         \
        if ((I) + 3 < left) GDPSO_KSTEP(true, (I) % 3, true, ((I) + 1) % 3, (I) % 3)           \
        else if ((I) + 1 < left) GDPSO_KSTEP(false, 0, true, ((I) + 1) % 3, (I) % 3)           \
        else GDPSO_KSTEP(false, 0, false, 0, (I) % 3)                                         \
    }
        GDPSO_TAIL(0)
        GDPSO_TAIL(1)
        GDPSO_TAIL(2)
        GDPSO_TAIL(3)
        GDPSO_TAIL(4)
#undef GDPSO_TAIL
    } else if constexpr (PF2) {
        // K-tile t travels through register set t & 1: step t issues the loads of tile t + 2 into
        // the set that tile t left, and moves tile t + 1 (loaded a whole step ago) into LDS.
        // (BD: the B half of a set is not moved anywhere: step t's MFMAs read it where it is, last argument.)
        int t = 0;
        for (; t + 3 < nk; t += 2) {
            GDPSO_KSTEP(true, 0, true, 1, 0)
            GDPSO_KSTEP(true, 1, true, 0, 1)
        }
        const int left = nk - t;  // 0 (empty K range), 1, 2 or 3; t is even
        if (left == 3) {
            GDPSO_KSTEP(true, 0, true, 1, 0)
            GDPSO_KSTEP(false, 0, true, 0, 1)
            GDPSO_KSTEP(false, 0, false, 0, 0)
        } else if (left == 2) {
            GDPSO_KSTEP(false, 0, true, 1, 0)
            GDPSO_KSTEP(false, 0, false, 0, 1)
        } else if (left == 1) {
            GDPSO_KSTEP(false, 0, false, 0, 0)
        }
    } else {
        for (int t = 0; t + 1 < nk; ++t) GDPSO_KSTEP(true, 0, true, 0, 0)
        if (nk > 0) GDPSO_KSTEP(false, 0, false, 0, 0)
    }
#undef GDPSO_KSTEP
#undef GDPSO_SGB
#undef GDPSO_SGB4
#undef GDPSO_GISSUE
#undef GDPSO_GISSUE_A
#undef GDPSO_GISSUE_B
#undef GDPSO_GISSUE_BG
#undef GDPSO_GADVANCE
#undef GDPSO_DBG
#undef GDPSO_SET_TAP
#undef GDPSO_GLOAD
#undef GDPSO_SSTORE
#undef GDPSO_LOADFRAG
    GDPSO_STAMP(2)

    if constexpr (FUSE) {
        // This layer's 64 x 64 tile, activated, becomes the A operand (K = 64) of the next layer;
        // its weights come through LDS one 64-channel column block at a time, prefetched a
        // block ahead; every block ends in the usual LDS-staged epilogue with its residual.
        const FuseArgs &f = *fuse;
        constexpr int LD2 = 64 + 4;
        float *sA2 = lds, *sB2 = lds + 64 * LD2, *sC2 = lds + 2 * 64 * LD2;
        {
            const int col = wc * 32 + l31;
            const float sc1 = a->scale[n0 + col], sh1 = a->shift[n0 + col];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sA2[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LD2 + col] = apply_act(acc[0][0][r] * sc1 + sh1, a->act);
        }
        const int c4 = tid & 15, r0 = tid >> 4;  // 16 threads x 16 B per 64-float row, 16 rows per pass
        const int hw2 = P.hm * P.wm;
        const bool remap = a->out_mul != 1 || P.hm != a->hout || P.wm != a->wout;
        int off[4];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int m = m0 + pass * 16 + r0;
            const int n = fast_div(m, P.hw_mul, P.hw_shift), rr = m - n * hw2;
            const int my = fast_div(rr, P.wm_mul, P.wm_shift), mx = rr - my * P.wm;
            const int pix = remap ? (n * a->hout + my * a->out_mul + P.off_y) * a->wout + mx * a->out_mul + P.off_x : m;
            off[pass] = m < P.M ? pix * f.cout + c4 * 4 : -1;
        }
        const int nblocks = f.cout / 64;
        f32x4 rw[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) rw[p] = *(global_f32x4_ptr)(f.wt + (size_t)(r0 + 16 * p) * 64 + c4 * 4);
        for (int cb = 0; cb < nblocks; ++cb) {
#pragma unroll
            for (int p = 0; p < 4; ++p) *(f32x4 *)&sB2[(r0 + 16 * p) * LD2 + c4 * 4] = rw[p];
            if (cb + 1 < nblocks) {
#pragma unroll
                for (int p = 0; p < 4; ++p) rw[p] = *(global_f32x4_ptr)(f.wt + (size_t)((cb + 1) * 64 + r0 + 16 * p) * 64 + c4 * 4);
            }
            f32x4 rres[4];
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) rres[pass] = 0.f;
            if (f.res) {  // straight-line: rows past M read the first row's residual, which is not used
#pragma unroll
                for (int pass = 0; pass < 4; ++pass)
                    rres[pass] = *(global_f32x4_ptr)(f.res + (off[pass] < 0 ? c4 * 4 : off[pass]) + cb * 64);
            }
            __syncthreads();
            f32x16 acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
            const float *pa2 = sA2 + (wr * 32 + l31) * LD2 + half * 4;
            const float *pb2 = sB2 + (wc * 32 + l31) * LD2 + half * 4;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 av2 = *(const f32x4 *)(pa2 + g * 8), bv2 = *(const f32x4 *)(pb2 + g * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av2[s], bv2[s], acc2, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sC2[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LD2 + wc * 32 + l31] = acc2[r];
            __syncthreads();
            const f32x4 sc2 = *(const f32x4 *)(f.scale + cb * 64 + c4 * 4), sh2 = *(const f32x4 *)(f.shift + cb * 64 + c4 * 4);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                if (off[pass] < 0) continue;
                f32x4 v = *(const f32x4 *)&sC2[(pass * 16 + r0) * LD2 + c4 * 4];
                v = v * sc2 + sh2 + rres[pass];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = apply_act(v[q], f.act);
                *(f32x4 *)(f.out + off[pass] + cb * 64) = v;
            }
        }
        return;
    }
    // Epilogue through LDS.  Accumulator register r of lane l is
    // C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]; the tile is first laid out row-major
    // in LDS (the K loop's last barrier has retired every operand read).
    constexpr int LDC = BN + 4;
    float *sC = lds;
    StorePlan<BM, BN> plan;
    static_assert(!CHAIN || StorePlan<BM, BN>::PREFETCH_RES, "a chain's residual is read by conv_store_plan only");
    conv_store_plan<E, BM, BN, CHAIN>(a, tid, m0, n0, P, plan);
    if (KW == 1) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sC[(wr * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDC + wc * (BN / 2) + j * 32 + l31] = acc[i][j][r];
    } else {
        // K-split layouts: the wave of K part q leaves its 32x32 block in region q (BM x LDC floats
        // each); the store pass adds the regions as it reads them
        constexpr int REGION = BM * LDC;
        static_assert(LDS_FLOATS >= KW * REGION + 4, "room for the partial tiles");
#pragma unroll
        for (int r = 0; r < 16; ++r)
            sC[kh * REGION + ((r & 3) + 8 * (r >> 2) + 4 * half) * LDC + wc * 32 + l31] = acc[0][0][r];
    }
    __syncthreads();
    conv_store_tile<E, BM, BN, KW>(a, lds, LDS_FLOATS, tid, sp, tile, plan);
#ifdef GDPSO_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have left the CU
    GDPSO_STAMP(3)
#endif
}

// BKB: bytes of K per K-tile row (128, 256 or 512); the K-tile holds BKB / sizeof(E) elements.
template <typename E, int BM, int BN, int BKB, int KW>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvKArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * (BKB / 4 + 4)];
    conv_tile<E, BM, BN, BKB / (int)sizeof(E), KW>(ArgsDirect{a}, blockIdx.x & 7, blockIdx.x >> 3, lds);
}

// A convolution with 64 output channels followed by a 1x1 convolution, one launch: conv2 + conv3
// of the first-stage bottleneck blocks.  The intermediate never leaves the workgroup.
__global__ __launch_bounds__(256, 2) void conv_gemm_fused_kernel(const ConvKArgs a, const FuseArgs f) {
    __shared__ __attribute__((aligned(16))) float lds[3 * 64 * 68];
    conv_tile<float, 64, 64, 32, 1, true>(ArgsDirect{a}, blockIdx.x & 7, blockIdx.x >> 3, lds, &f);
}

// Two independent layers in one grid (same tile shape, no split-K): the first tiles_a workgroups
// (a multiple of 8, so that both layers see workgroup index & 7 == XCD) belong to layer a, the
// rest to layer b.  Used for identity_downsample + conv1 of a bottleneck block, which read the
// same input: one launch, one fill and drain, and the tail of one layer overlaps the other.
template <typename E, int BM, int BN, int BKB, int KW>
__global__ __launch_bounds__(256, 2) void conv_gemm_pair_kernel(const ConvKArgs a, const ConvKArgs b, int tiles_a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * (BKB / 4 + 4)];
    // two copies of the tile code, each reading its own kernel arguments straight from SGPRs
    if ((int)blockIdx.x < tiles_a)
        conv_tile<E, BM, BN, BKB / (int)sizeof(E), KW>(ArgsDirect{a}, blockIdx.x & 7, blockIdx.x >> 3, lds);
    else
        conv_tile<E, BM, BN, BKB / (int)sizeof(E), KW>(ArgsDirect{b}, (blockIdx.x - tiles_a) & 7, (blockIdx.x - tiles_a) >> 3, lds);
}

// ----------------------------------------------------------------------------- layer chains
// A run of dependent layers in ONE launch.  Rows of different images never interact, so the images
// are cut into 8 shares (16 with two teams per XCD) and every XCD walks the whole run for its share:
// producer and consumer of every intermediate sit on the same XCD, and the hand-off between two
// layers is a store drain, a plain store of the step number into the workgroup's own word of a flag
// array and, on the other side, one wave polling that array and activation loads that miss the L1
// and hit the XCD's L2 (about 1.5 us a round, measured by tools/micro/xcd_barrier.hip; a hand-off
// that crosses XCDs needs an L2 write-back and costs twenty-five times that).  The launch gaps between
// the layers and the skew of their fill and drain disappear.
//
// The grid is resident as a whole (the host sizes it from the occupancy query).  Workgroups find out
// where they run from the hardware's XCC_ID and CU id, register, and wait until everybody has: after
// that each knows how many workgroups every XCD got.  Populated XCDs take the shares in turn (XCD of
// rank r among P takes shares r, r + P, ...: with the usual even deal, one each), the members of an XCD
// split each step's tiles statically (member j takes tiles j, j + members, ...), publish the step in
// their flag word after their last tile and wait for the previous step's flags ahead of their first.
// Nothing depends on how the dispatcher deals workgroups to XCDs except the speed.
constexpr int CHAIN_MAX_STEPS = 192;  // ResNet152 behind its first stage: 47 blocks x 3
constexpr int CHAIN_MAX_SHARES = 16;  // 8 XCDs x up to two teams
constexpr int CHAIN_MAX_MEMBERS = 1024;  // workgroups of one launch (all of them may land on one XCD)

// An address must not be written by two XCDs in one launch (two L2s would each hold a dirty copy and
// the order of their write-backs decides what memory ends up with), and nothing a slower XCD still
// has to read may be overwritten by a faster one.  So every buffer a chain writes is cut into fixed
// slices, one per share, whatever the layer: share s keeps its rows of every intermediate at
// slice s (`gap` = slice size - the layer's bytes per share, added to the base once per share), the
// chain's input is only read, and its final output has a buffer of its own.
// Tile shape code of a chain step only: 64 rows x 128 columns (2 x 2 waves, two MFMA blocks side by side per
// wave).  Every output is the same sum in the same order as with the 64x64 tile (each wave walks the whole K of
// its rows), so a layer the planner gave 64x64 tiles may run 64x128 as a step of a chain without its bits
// changing; a chain takes it for steps that would otherwise hand a workgroup several 64x64 tiles (each tile
// pays its prologue, operand fetch and epilogue again, ~3 us that no K loop of the step hides).
constexpr int TILE_64x128 = 129;
struct ChainStep {
    ConvKArgs a[2];  // one layer, or two that read the same input; in / res / out hold byte offsets into the workspace
    int tiles[2];    // tiles per share of each (0: absent)
    int shape[2];    // tile shape code of each (64, 32, TILE_32x32, TILE_64x128)
    unsigned gap_in[2], gap_res[2], gap_out[2];  // bytes per share by which the buffer's slices exceed the layer's rows (0: standard layout)
};

struct ChainCtl {  // zeroed ahead of every launch
    unsigned arrived, status, pad0[30];
    unsigned members[8][32];                        // per XCD (one 128-byte line each): workgroups that were [0] the first, [1] the second, [2] a further one on their CU
    unsigned cu_count[8][256];                      // per XCD and CU (HW_ID bits 8-15): workgroups registered
    unsigned cu_first[8][256];                      // 1 + the ticket of the CU's first workgroup
    unsigned flags[CHAIN_MAX_SHARES][CHAIN_MAX_MEMBERS];  // per share and member: steps finished (1 + the last step's index)
};

#ifdef GDPSO_STAMPS
// Diagnostic build only (tools/chain_probe.py): per workgroup and step, s_memrealtime (100 MHz) when
// the step was entered, when its wait ended, when its tiles were done and when it was signalled.
__device__ unsigned long long *g_chain_stamps = nullptr;
__device__ int g_chain_stamp_step = -1;  // the step whose tiles also leave the tile code's own stamps in g_stamps
#endif

template <typename E>
__global__ __launch_bounds__(256, 2) void conv_chain_kernel(const ChainStep *steps, int nsteps, char *ws, ChainCtl *ctl,
                                                             unsigned drop_xcc_mask, int nteams, unsigned stagger_ticks, int peek) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 64 * 132];  // the largest of the three tile shapes
    __shared__ int s_reg[12];
    const int tid = threadIdx.x;
    if (tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(63508) & 7u;  // HW_REG_XCC_ID
        const bool dropped = (drop_xcc_mask >> xcc) & 1u;            // test aid: behave as if this XCD had got no workgroup
        // Arrival order on the CU (0: first, 1: second, 2: any further), then a ticket among the XCD's
        // workgroups of that order; the first of a CU leaves its ticket for the second to find.  Every
        // returned value is consumed before the arrival is counted, so whoever sees the full count sees these.
        const unsigned cu = (__builtin_amdgcn_s_getreg(63492) >> 8) & 0xffu;  // HW_REG_HW_ID: CU, shader array, shader engine
        unsigned k = 0, j = 0;
        if (!dropped) {
            k = min(2u, __hip_atomic_fetch_add(&ctl->cu_count[xcc][cu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            j = __hip_atomic_fetch_add(&ctl->members[xcc][k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k == 0) __hip_atomic_store(&ctl->cu_first[xcc][cu], j + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(j) : "memory");
        __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        bool ok = true;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(&ctl->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            ++spins;
            if ((spins & 255u) == 0 && (__builtin_amdgcn_s_memrealtime() - t0 > CHAIN_REGISTER_TICKS ||
                                        __hip_atomic_load(&ctl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                chain_abort(&ctl->status, 2u);
                __hip_atomic_fetch_max(&g_chain_abort_arrived, __hip_atomic_load(&ctl->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        int populated = 0, rank = 0;
        unsigned m0 = 0, m1 = 0, m2 = 0;
        for (unsigned x = 0; x < 8; ++x) {
            const unsigned f = __hip_atomic_load(&ctl->members[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            populated += f != 0;
            rank += (f != 0 && x < xcc);
            if (x == xcc) {
                m0 = f;
                m1 = __hip_atomic_load(&ctl->members[x][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                m2 = __hip_atomic_load(&ctl->members[x][2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // The usual deal puts exactly two workgroups on every CU.  Then the second of a CU is numbered
        // from the far end of where its first stands: with one team per XCD (members 0 .. 2 m0 - 1) a CU
        // holds members i and 2 m0 - 1 - i, so a step with up to m0 tiles runs one tile per CU and a step whose
        // tiles are sorted heaviest first pairs heavy with light; with two teams per XCD (the firsts and the
        // seconds, each 0 .. m0 - 1) a CU holds member i of one team and m0 - 1 - i of the other.
        // Any other deal: one team of all the XCD's workgroups in ticket order.
        const bool paired = m1 == m0 && m2 == 0;
        int team = 0, local_teams = 1, member, members;
        if (paired) {
            const unsigned j0 = k == 0 ? j : __hip_atomic_load(&ctl->cu_first[xcc][cu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u;
            if (nteams == 2) {
                team = (int)k;
                local_teams = 2;
                member = (int)(k == 0 ? j0 : m0 - 1u - j0);
                members = (int)m0;
            } else {
                member = (int)(k == 0 ? j0 : 2u * m0 - 1u - j0);
                members = (int)(2u * m0);
            }
        } else {
            member = (int)(k == 0 ? j : k == 1 ? m0 + j : m0 + m1 + j);
            members = (int)(m0 + m1 + m2);
        }
        s_reg[0] = ok && !dropped ? 1 : 0;
        s_reg[1] = member;
        s_reg[2] = members;
        s_reg[3] = rank;
        s_reg[4] = populated;
        s_reg[5] = team;
        s_reg[6] = local_teams;
        if (team == 1 && stagger_ticks) {  // the second team starts late, so that the two teams' hand-offs do not coincide
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < stagger_ticks) __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (!s_reg[0]) return;
    const int member = s_reg[1], members = s_reg[2], rank = s_reg[3], populated = s_reg[4], team = s_reg[5], local_teams = s_reg[6];
    // populated XCDs take the 8 groups of shares in turn, an XCD's teams the shares of a group
    for (int sidx = rank * nteams + team; sidx < 8 * nteams;) {
        const int share = sidx;
        int prev_tiles = 0;
        for (int s = 0; s < nsteps; ++s) {
            typedef const __attribute__((address_space(4))) int *const_int_ptr;
            const const_int_ptr meta = (const_int_ptr)(uintptr_t)steps[s].tiles;  // tiles[2], shape[2]
            const int ta = meta[0], ntile = ta + meta[1];
            bool first = true;
#ifdef GDPSO_STAMPS
            unsigned long long *const stamps = g_chain_stamps ? g_chain_stamps + ((size_t)blockIdx.x * CHAIN_MAX_STEPS + s) * 4 : nullptr;
            if (stamps && tid == 0 && member < ntile) stamps[0] = __builtin_amdgcn_s_memrealtime();
            if (stamps && tid == 0 && s == 0)  // last step slot: share + 1, member
                g_chain_stamps[((size_t)blockIdx.x * CHAIN_MAX_STEPS + CHAIN_MAX_STEPS - 1) * 4] = (unsigned long long)(share + 1) | ((unsigned long long)member << 8);
#endif
            for (int t = member; t < ntile; t += members) {
                const int which = t >= ta;
                const int slot = t - (which ? ta : 0);
                const int shape = meta[2 + which];
                ChainWait w;
                w.peek = peek;
                if (first && s > 0) {
                    w.flags = ctl->flags[share];
                    w.target = (unsigned)s;  // step s - 1 finished
                    w.count = min(members, prev_tiles);
                    w.status = &ctl->status;
#ifdef GDPSO_STAMPS
                    w.stamp = stamps ? stamps + 1 : nullptr;
                    w.tile_stamps = (g_stamps && s == g_chain_stamp_step) ? g_stamps + (size_t)blockIdx.x * 16 : nullptr;
#endif
                }
                if (!first) __syncthreads();  // the previous tile's store pass has left the LDS
                const ArgsChain a{(ArgsChain::args_ptr)(uintptr_t)&steps[s].a[which], (ArgsChain::gaps_ptr)(uintptr_t)steps[s].gap_in, ws, share, which};
                if (shape == 64)
                    conv_tile<E, 64, 64, 128 / (int)sizeof(E), 1, false, true>(a, share, slot, lds, nullptr, w);
                else if (shape == TILE_64x128)
                    conv_tile<E, 64, 128, 128 / (int)sizeof(E), 1, false, true>(a, share, slot, lds, nullptr, w);
                else if (shape == 32)
                    conv_tile<E, 32, 64, 256 / (int)sizeof(E), 2, false, true>(a, share, slot, lds, nullptr, w);
                else
                    conv_tile<E, 32, 32, 512 / (int)sizeof(E), 4, false, true>(a, share, slot, lds, nullptr, w);
                first = false;
            }
            if (!first) {
#ifdef GDPSO_STAMPS
                if (stamps && tid == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();
#endif
                // the step's results are in the L2 once the stores have been acknowledged
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(&ctl->flags[share][member], (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef GDPSO_STAMPS
                if (stamps && tid == 0) stamps[3] = __builtin_amdgcn_s_memrealtime();
#endif
            }
            prev_tiles = ntile;
        }
        const int u = sidx % nteams;
        sidx += u + local_teams < nteams ? local_teams : populated * nteams - u + team;
    }
}

// Sums the split-K slabs and applies the epilogue.  Four channels per thread.
template <typename E>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ part, int splits, long slab,
                                                            int cout, const float *__restrict__ scale,
                                                            const float *__restrict__ shift,
                                                            const E *__restrict__ res, E *__restrict__ out,
                                                            int act) {
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= slab) return;
    f32x4 s = *(const f32x4 *)(part + e);
    for (int k = 1; k < splits; ++k) s += *(const f32x4 *)(part + (size_t)k * slab + e);
    const int c = (int)(e % cout);
    const f32x4 sc = *(const f32x4 *)(scale + c), sh = *(const f32x4 *)(shift + c);
    f32x4 v = s * sc + sh;
    if (res) v += load4<E>(res + e);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = apply_act(v[q], act);
    store4<E>(out + e, v);
}

// --------------------------------------------------------------------------- host side
// Optional per-launch timing of the convolution kernel with HIP events recorded on the
// launch stream (used by the benchmark to price the kernel against its roofline).
namespace {
struct ConvTiming {
    bool on = false;
    std::vector<hipEvent_t> events;
    size_t used = 0;
    double flops = 0.0;     // nominal: every tap of every output pixel, padding included
    double executed = 0.0;  // what the launches actually multiplied (padding taps of border classes skipped)
    double bytes = 0.0;  // input + weights + output (+ residual), each counted once per launch
} g_timing;
}  // namespace

// Development switch (A/B measurements on one device).  Measured at 512 images on MI355X:
// the separate split-K reduce kernel is 1-3 % faster end to end than the in-kernel
// (last-arriver) reduction, which only pays on much longer K loops / larger grids.
static bool g_allow_tile32 = [] {
    const char *e = getenv("GDPSO_CONV_TILE32");
    return e ? e[0] != '0' : true;
}();

static bool g_snake = [] {
    const char *e = getenv("GDPSO_CONV_SNAKE");
    return e ? e[0] != '0' : true;
}();
static int g_cus_per_xcd = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return std::max(1, cus / 8);
}();

static bool g_fuse_1x1 = [] {
    const char *e = getenv("GDPSO_CONV_FUSE_1X1");
    return e ? e[0] != '0' : true;
}();

static bool g_pair_launch = [] {
    const char *e = getenv("GDPSO_CONV_PAIR");
    return e ? e[0] != '0' : true;
}();

static int g_fused_max_splits = [] {
    const char *e = getenv("GDPSO_SPLITK_FUSED_MAX");
    return e ? atoi(e) : 0;
}();

static long max_ktot(const ConvLayer &L) {
    long k = 0;
    for (int p = 0; p < L.nphase; ++p) k = std::max<long>(k, (long)L.phase[p].ntaps * L.cin);
    return k;
}

// Chooses tile and split-K.  Measured on MI355X over the ResNet50 / generator shapes at 512
// images: 64x64 tiles beat 128x128 unless the grid is several workgroups per CU deep, and
// split-K only pays when the unsplit grid has fewer workgroups than CUs; then just enough
// splits to give every CU one workgroup.
// L.tile names the kernel shape: 128 -> 128x128, 64 -> 64x64, 32 -> 32 rows x 64 columns with the
// K-tile split over wave pairs, 16 -> 32x32 with the K-tile split over the four waves.
constexpr int TILE_32x32 = 16;
static int tile_bm(int tile) { return tile == TILE_32x32 ? 32 : tile == TILE_64x128 ? 64 : tile; }
static int tile_bn(int tile) { return tile == 32 ? 64 : tile == TILE_32x32 ? 32 : tile == TILE_64x128 ? 128 : tile; }

static long m_tiles(const ConvLayer &L, int n_images, int tile) {
    long t = 0;
    for (int p = 0; p < L.nphase; ++p) t += ceil_div((long)n_images * L.phase[p].hm * L.phase[p].wm, tile_bm(tile));
    return t;
}

// Launch-plan options (gdpso_plan_options).  The tile shape decides the order in which a row's K parts
// are added (the 32-row shapes split every K-tile over wave pairs or over the four waves, split-K adds
// slabs), so two batch sizes planned with different shapes agree to fp32 rounding only.  A plan made
// for a FIXED batch size gives every image the same bits whatever batch it arrives in: that is what
// makes a sharded run reproduce an unsharded one bit for bit.
static int g_plan_images = [] {
    const char *e = getenv("GDPSO_PLAN_IMAGES");
    return e && atoi(e) > 0 ? atoi(e) : 0;
}();
static bool g_plan_splitk = [] {
    const char *e = getenv("GDPSO_PLAN_SPLITK");
    return e ? e[0] != '0' : true;
}();

int plan_batch(int n_images) { return g_plan_images > 0 ? g_plan_images : n_images; }

bool conv_timing_slot(hipEvent_t *start, hipEvent_t *stop) {
    if (!g_timing.on || g_timing.used + 2 > g_timing.events.size()) return false;
    *start = g_timing.events[g_timing.used];
    *stop = g_timing.events[g_timing.used + 1];
    g_timing.used += 2;
    return true;
}
void conv_timing_account(double flops_nominal, double flops_executed, double bytes) {
    g_timing.flops += flops_nominal;
    g_timing.executed += flops_executed;
    g_timing.bytes += bytes;
}

int plan_conv(ConvLayer &L, int n_images, int num_cus) {
    n_images = plan_batch(n_images);  // shapes as for the pinned batch size, whatever this one is
    L.tile = 64;
    if (L.cout % 128 == 0) {
        const long t128 = m_tiles(L, n_images, 128) * (L.cout / 128);
        if (t128 >= 4L * num_cus) L.tile = 128;
    }
    if (L.cout % L.tile != 0) return fail(GDPSO_ERR_ARG, "conv: cout must be a multiple of 64");
    // at most one 64x64 tile per CU: 32-row tiles double the workgroups without slabs (measured:
    // a gain up to 256 tiles, a loss at 392)
    const int kq = dtype_kquant(L.dtype);  // k per wave per K-tile: the K-split tiles need 2 and 4 of them per tap
    if (L.tile == 64 && g_allow_tile32 && L.cin % (2 * kq) == 0 && m_tiles(L, n_images, 64) * (L.cout / 64) <= (long)num_cus) L.tile = 32;
    // still at most one workgroup per CU: 32x32 tiles with the K-tile split four ways double them again
    if (L.tile == 32 && L.cin % (4 * kq) == 0 && m_tiles(L, n_images, 32) * (L.cout / 64) <= (long)num_cus) L.tile = TILE_32x32;
    const long tiles = m_tiles(L, n_images, L.tile) * (L.cout / tile_bn(L.tile));
    const long ktot = max_ktot(L);
    int splits = 1;
    if (tiles < num_cus) {
        splits = (int)std::min<long>(ceil_div((long)num_cus, tiles), 16);
        // keep at least 128 k per split so the prologue/epilogue stays amortised
        splits = (int)std::max<long>(1, std::min<long>(splits, ktot / 128));
    }
    if (!g_plan_splitk) splits = 1;
    L.splits = splits;
    return GDPSO_OK;
}

// Split-K workspace: [SPLITK_COUNTER_BYTES of arrival counters][splits slabs].
constexpr size_t SPLITK_COUNTER_BYTES = 16384;
size_t conv_splitk_bytes(const ConvLayer &L, int n_images) {
    if (L.splits <= 1) return SPLITK_COUNTER_BYTES;
    return SPLITK_COUNTER_BYTES + (size_t)L.splits * n_images * L.hout * L.wout * L.cout * sizeof(float);
}

// Zeroes the arrival counters; once per forward pass, ahead of the first convolution.  A kernel, not
// hipMemsetAsync: a memset node of a captured graph did not clear them on replay (the captured loop then left layers
// unreduced), a kernel node does.
__global__ __launch_bounds__(256) void clear_quads_kernel(f32x4 *__restrict__ p, int quads) {
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i < quads) p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

int conv_splitk_prepare(hipStream_t stream, float *splitk_ws) {
    if (g_fused_max_splits <= 0) return GDPSO_OK;  // the counters are only read by the in-kernel reduction
    constexpr int quads = (int)(SPLITK_COUNTER_BYTES / 16);
    hipLaunchKernelGGL(clear_quads_kernel, dim3(ceil_div(quads, 256)), dim3(256), 0, stream, (f32x4 *)splitk_ws, quads);
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

// Fills the kernel arguments of one layer; returns its number of workgroups per split through *tiles.
// nshare > 0: for a chain over that many shares of the images; a phase's tiles are then counted per share
// (ceil(images / nshare) images each) and *grid_x is nshare times a share's tiles.
static int conv_args(const ConvLayer &L, int n_images, const void *in, const void *res, void *out, float *splitk_ws,
                     ConvKArgs &a, long *grid_x, bool *fused_reduce, int nshare = 0) {
    if (!dtype_ok(L.dtype)) return fail(GDPSO_ERR_ARG, "conv: unknown element type");
    a.in = in; a.wt = L.wt; a.wtq = L.wtq; a.scale = L.scale; a.shift = L.shift; a.res = res; a.out = out;
    const long tiles_m = m_tiles(L, n_images, L.tile);
    const long tiles_total = tiles_m * (L.cout / tile_bn(L.tile));
    a.table = L.table;
    a.n_images = n_images;
    long per_xcd = 0;
    for (int p = 0; p < MAX_PHASES; ++p) {
        if (p >= L.nphase) {
            a.phase_word[p] = 0xffffffu;
            continue;
        }
        if (nshare > 0) {
            per_xcd += (long)ceil_div((long)ceil_div(n_images, nshare) * L.phase[p].hm * L.phase[p].wm, tile_bm(L.tile)) * (L.cout / tile_bn(L.tile));
            a.phase_word[p] = (unsigned)per_xcd;
            continue;
        }
        const long tiles = (long)ceil_div((long)n_images * L.phase[p].hm * L.phase[p].wm, tile_bm(L.tile)) * (L.cout / tile_bn(L.tile));
        const long eighth = ceil_div(tiles, 8);  // an XCD's part
        per_xcd += eighth;
        a.phase_word[p] = (unsigned)per_xcd | ((unsigned)(8 * eighth - tiles) << 24);
    }
    if (per_xcd >= 0xffffff) return fail(GDPSO_ERR_ARG, "conv: too many tiles");
    *grid_x = (nshare > 0 ? nshare : 8) * per_xcd;
    a.per_xcd = (int)per_xcd;
    a.share_images = nshare > 0 ? ceil_div(n_images, nshare) : 0;
    a.snake = (g_snake && L.nphase > 1 && per_xcd > g_cus_per_xcd && per_xcd <= 2 * g_cus_per_xcd) ? g_cus_per_xcd : 0;
    // in-kernel (last-arriver) reduction up to GDPSO_SPLITK_FUSED_MAX splits (default: off); one counter per workgroup index
    *fused_reduce = L.splits > 1 && L.splits <= g_fused_max_splits && (size_t)*grid_x * sizeof(int) <= SPLITK_COUNTER_BYTES;
    a.part = L.splits > 1 ? splitk_ws + SPLITK_COUNTER_BYTES / sizeof(float) : nullptr;
    a.counters = *fused_reduce ? (int *)splitk_ws : nullptr;
    a.hin = L.hin; a.win = L.win; a.cin = L.cin;
    a.cout = L.cout; a.hout = L.hout; a.wout = L.wout;
    a.in_stride = L.in_stride; a.out_mul = L.out_mul;
    a.act = L.act; a.splits = L.splits;
    a.tiles_n = L.cout / tile_bn(L.tile);
    const FastDiv dn = fast_div_for((unsigned)a.tiles_n);
    a.tiles_n_mul = dn.mul;
    a.tiles_n_shift = dn.shift;
    a.slab = (long)n_images * L.hout * L.wout * L.cout;
    const int kq = dtype_kquant(L.dtype);
    if (L.cin % kq != 0 || L.cin > MAX_CIN) return fail(GDPSO_ERR_ARG, "conv: cin must be a multiple of 32 (fp32) or 64 (16-bit), at most 4096");
    if (L.splits > 1 && !splitk_ws) return fail(GDPSO_ERR_WORKSPACE, "conv: split-K workspace missing");
    if (a.slab * 4 >= (1L << 31)) return fail(GDPSO_ERR_ARG, "conv: output larger than 2 GiB");
    // the kernel forms element offsets into the input and the weights in 32 bits
    if ((long)n_images * L.hin * L.win * L.cin * 4 >= (1L << 31)) return fail(GDPSO_ERR_ARG, "conv: input larger than 2 GiB");
    if ((long)L.weight_floats * 4 >= (1L << 31)) return fail(GDPSO_ERR_ARG, "conv: weights larger than 2 GiB");
    if (tiles_total >= (1L << 30) || (long)n_images * L.hout * L.wout >= (1L << 31)) return fail(GDPSO_ERR_ARG, "conv: too many rows");
    if (GDPSO_BDIRECT && L.tile != 128 && !L.wtq) return fail(GDPSO_ERR_ARG, "conv: the layer has no chunk-major weights (cout must be a multiple of 32)");
    if (L.tile == 32 && L.cin % (2 * kq) != 0) return fail(GDPSO_ERR_ARG, "conv: the 32-row tile needs cin to be a multiple of 64 (fp32) or 128 (16-bit)");
    if (L.tile == TILE_32x32 && L.cin % (4 * kq) != 0) return fail(GDPSO_ERR_ARG, "conv: the 32x32 tile needs cin to be a multiple of 128 (fp32) or 256 (16-bit)");
    return GDPSO_OK;
}

static void conv_account(const ConvLayer &L, int n_images, const ConvKArgs &a, bool has_res) {
    g_timing.flops += L.flops_nominal * n_images;
    for (int p = 0; p < L.nphase; ++p)
        g_timing.executed += 2.0 * n_images * L.phase[p].hm * L.phase[p].wm * (double)L.phase[p].ntaps * L.cin * L.cout;
    g_timing.bytes += (double)dtype_size(L.dtype) * ((double)n_images * L.hin * L.win * L.cin + (double)L.weight_floats + (double)a.slab * (has_res ? 2 : 1));
}

int launch_conv(const ConvLayer &L, hipStream_t stream, int n_images, const void *in, const void *res,
                void *out, float *splitk_ws) {
    ConvKArgs a;
    long tiles_total = 0;
    bool fused_reduce = false;
    int rc = conv_args(L, n_images, in, res, out, splitk_ws, a, &tiles_total, &fused_reduce);
    if (rc) return rc;
    dim3 grid((unsigned)tiles_total, 1, L.splits);
    // With timing on, the kernel's own begin/end timestamps are attached to two events by the
    // launch itself (no extra markers on the stream, so nothing but the kernel is timed).
    const bool timed = g_timing.on && g_timing.used + 2 <= g_timing.events.size();
    hipEvent_t e0 = timed ? g_timing.events[g_timing.used] : nullptr;
    hipEvent_t e1 = timed ? g_timing.events[g_timing.used + 1] : nullptr;
#define GDPSO_LAUNCH(KERNEL)                                                                 \
    if (timed)                                                                               \
        hipExtLaunchKernelGGL((KERNEL), grid, dim3(256), 0, stream, e0, e1, 0, a);           \
    else                                                                                     \
        hipLaunchKernelGGL((KERNEL), grid, dim3(256), 0, stream, a);
#define GDPSO_LAUNCH_TILES(E)                                       \
    if (L.tile == 128) {                                            \
        GDPSO_LAUNCH((conv_gemm_kernel<E, 128, 128, 128, 1>))       \
    } else if (L.tile == 64) {                                      \
        GDPSO_LAUNCH((conv_gemm_kernel<E, 64, 64, 128, 1>))         \
    } else if (L.tile == 32) {                                      \
        GDPSO_LAUNCH((conv_gemm_kernel<E, 32, 64, 256, 2>))         \
    } else {                                                        \
        GDPSO_LAUNCH((conv_gemm_kernel<E, 32, 32, 512, 4>))         \
    }
    if (L.dtype == DT_F32) {
        GDPSO_LAUNCH_TILES(float)
    } else if (L.dtype == DT_BF16) {
        GDPSO_LAUNCH_TILES(bf16_t)
    } else {
        GDPSO_LAUNCH_TILES(f16_t)
    }
#undef GDPSO_LAUNCH_TILES
#undef GDPSO_LAUNCH
    if (timed) {
        g_timing.used += 2;
        conv_account(L, n_images, a, res != nullptr);
    }
    if (L.splits > 1 && !fused_reduce) {
        const long quads = a.slab / 4;
        const dim3 rgrid(ceil_div(quads, 256));
        if (L.dtype == DT_F32)
            hipLaunchKernelGGL(splitk_reduce_kernel<float>, rgrid, dim3(256), 0, stream, a.part, L.splits, a.slab, L.cout, L.scale,
                               L.shift, (const float *)res, (float *)out, L.act);
        else if (L.dtype == DT_BF16)
            hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, rgrid, dim3(256), 0, stream, a.part, L.splits, a.slab, L.cout, L.scale,
                               L.shift, (const bf16_t *)res, (bf16_t *)out, L.act);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<f16_t>, rgrid, dim3(256), 0, stream, a.part, L.splits, a.slab, L.cout, L.scale,
                               L.shift, (const f16_t *)res, (f16_t *)out, L.act);
    }
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

// La (64 output channels, 64x64 tiles, unsplit) followed by the 1x1 convolution Lb over the same
// pixels, in one launch when the shapes allow; `mid` receives La's output only on the fallback path.
int launch_conv_then_1x1(const ConvLayer &La, const ConvLayer &Lb, hipStream_t stream, int n_images, const void *in,
                         void *mid, const void *res, void *out, float *splitk_ws) {
    const bool ok = g_fuse_1x1 && La.dtype == DT_F32 && Lb.dtype == DT_F32 && La.tile == 64 && La.splits == 1 && La.cout == 64 && Lb.cin == 64 && Lb.nphase == 1 &&
                    Lb.phase[0].ntaps == 1 && Lb.in_stride == 1 && Lb.out_mul == 1 && Lb.hout == La.hout && Lb.wout == La.wout &&
                    Lb.cout % 64 == 0 && Lb.phase[0].wstride == 64;
    if (!ok) {
        int rc = launch_conv(La, stream, n_images, in, nullptr, mid, splitk_ws);
        if (rc) return rc;
        return launch_conv(Lb, stream, n_images, mid, res, out, splitk_ws);
    }
    ConvKArgs a;
    long tiles = 0;
    bool fr = false;
    int rc = conv_args(La, n_images, in, nullptr, nullptr, splitk_ws, a, &tiles, &fr);
    if (rc) return rc;
    FuseArgs f;
    f.wt = (const float *)Lb.wt; f.scale = Lb.scale; f.shift = Lb.shift; f.res = (const float *)res; f.out = (float *)out;
    f.cout = Lb.cout; f.act = Lb.act;
    if ((long)n_images * Lb.hout * Lb.wout * Lb.cout * 4 >= (1L << 31)) return fail(GDPSO_ERR_ARG, "conv: output larger than 2 GiB");
    dim3 grid((unsigned)tiles, 1, 1);
    const bool timed = g_timing.on && g_timing.used + 2 <= g_timing.events.size();
    if (timed) {
        hipExtLaunchKernelGGL(conv_gemm_fused_kernel, grid, dim3(256), 0, stream, g_timing.events[g_timing.used],
                              g_timing.events[g_timing.used + 1], 0, a, f);
        g_timing.used += 2;
        conv_account(La, n_images, a, false);
        ConvKArgs b;
        b.slab = (long)n_images * Lb.hout * Lb.wout * Lb.cout;
        conv_account(Lb, n_images, b, res != nullptr);
        g_timing.bytes -= 2 * 4.0 * n_images * Lb.hin * Lb.win * Lb.cin;  // the intermediate is neither written nor read
    } else {
        hipLaunchKernelGGL(conv_gemm_fused_kernel, grid, dim3(256), 0, stream, a, f);
    }
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

// Two layers that read the same input, in one launch (see conv_gemm_pair_kernel).  Falls back
// to two launches when the planner gave them different shapes or either is split.
int launch_conv_pair(const ConvLayer &La, const ConvLayer &Lb, hipStream_t stream, int n_images, const void *in,
                     void *out_a, void *out_b, float *splitk_ws) {
    if (!g_pair_launch || La.dtype != Lb.dtype || La.tile != Lb.tile || La.tile == TILE_32x32 || La.splits != 1 || Lb.splits != 1) {
        int rc = launch_conv(La, stream, n_images, in, nullptr, out_a, splitk_ws);
        if (rc) return rc;
        return launch_conv(Lb, stream, n_images, in, nullptr, out_b, splitk_ws);
    }
    ConvKArgs a, b;
    long ta = 0, tb = 0;
    bool fa = false, fb = false;
    int rc;
    if ((rc = conv_args(La, n_images, in, nullptr, out_a, splitk_ws, a, &ta, &fa))) return rc;
    if ((rc = conv_args(Lb, n_images, in, nullptr, out_b, splitk_ws, b, &tb, &fb))) return rc;
    dim3 grid((unsigned)(ta + tb), 1, 1);
    const bool timed = g_timing.on && g_timing.used + 2 <= g_timing.events.size();
    hipEvent_t e0 = timed ? g_timing.events[g_timing.used] : nullptr;
    hipEvent_t e1 = timed ? g_timing.events[g_timing.used + 1] : nullptr;
#define GDPSO_LAUNCH(KERNEL)                                                                          \
    if (timed)                                                                                        \
        hipExtLaunchKernelGGL((KERNEL), grid, dim3(256), 0, stream, e0, e1, 0, a, b, (int)ta);         \
    else                                                                                              \
        hipLaunchKernelGGL((KERNEL), grid, dim3(256), 0, stream, a, b, (int)ta);
#define GDPSO_LAUNCH_TILES(E)                                       \
    if (La.tile == 128) {                                           \
        GDPSO_LAUNCH((conv_gemm_pair_kernel<E, 128, 128, 128, 1>))  \
    } else if (La.tile == 64) {                                     \
        GDPSO_LAUNCH((conv_gemm_pair_kernel<E, 64, 64, 128, 1>))    \
    } else {                                                        \
        GDPSO_LAUNCH((conv_gemm_pair_kernel<E, 32, 64, 256, 2>))    \
    }
    if (La.dtype == DT_F32) {
        GDPSO_LAUNCH_TILES(float)
    } else if (La.dtype == DT_BF16) {
        GDPSO_LAUNCH_TILES(bf16_t)
    } else {
        GDPSO_LAUNCH_TILES(f16_t)
    }
#undef GDPSO_LAUNCH_TILES
#undef GDPSO_LAUNCH
    if (timed) {
        g_timing.used += 2;
        conv_account(La, n_images, a, false);
        conv_account(Lb, n_images, b, false);
        g_timing.bytes -= (double)dtype_size(La.dtype) * n_images * La.hin * La.win * La.cin;  // the shared input is read once
    }
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

// ---- chains (see conv_chain_kernel)
static bool g_chain_enabled = [] {
    const char *e = getenv("GDPSO_CONV_CHAIN");
    return e ? e[0] != '0' : true;
}();
static unsigned g_chain_drop_mask = 0;  // test aid, see gdpso_chain_debug
static std::atomic<unsigned> g_chain_launches{0};
// Teams per XCD (GDPSO_CHAIN_TEAMS=1|2) and, with two, how long the second starts after the first
// (GDPSO_CHAIN_STAGGER_US), so that one team computes while the other hands over.
static int g_chain_teams = [] {
    const char *e = getenv("GDPSO_CHAIN_TEAMS");
    return e && atoi(e) == 2 ? 2 : 1;
}();
static unsigned g_chain_stagger = [] {
    const char *e = getenv("GDPSO_CHAIN_STAGGER_US");
    return (unsigned)(100.0 * (e ? atof(e) : 6.0));  // s_memrealtime ticks (100 MHz)
}();
static int g_chain_peek = [] {
    const char *e = getenv("GDPSO_CHAIN_PEEK");
    return e ? atoi(e) : 1;
}();
int chain_teams() { return g_chain_teams; }

struct ConvChain {
    std::vector<ChainStep> steps;
    std::deque<ConvLayer> layers;  // the chain's own copies (its tile choice), in step order
    std::vector<char> layer_has_res;
    int n_images = 0;
    int nteams = 1;   // teams per XCD: 8 * nteams shares
    int dtype = -1;   // of every layer (set by the first)
    int nshare() const { return 8 * nteams; }
    void *dev_steps = nullptr;
    size_t ctl_off = 0;
    int grid = 0;
    ~ConvChain() {
        if (dev_steps) (void)hipFree(dev_steps);
    }
};

bool chain_enabled() { return g_chain_enabled; }
size_t chain_ctl_bytes() { return sizeof(ChainCtl); }
ConvChain *chain_new(int n_images, int nteams) {
    auto c = new ConvChain();
    c->n_images = n_images;
    c->nteams = nteams == 2 ? 2 : 1;
    return c;
}
void chain_free(ConvChain *c) { delete c; }
int chain_steps(const ConvChain *c) { return c ? (int)c->steps.size() : 0; }

// The tile shape a layer runs with as a step of a chain over `nshare` shares of the images (0: it cannot
// be a step): the planner's shape, if it is one of the three small ones.  A share holds
// ceil(images / nshare) images (the last shares fewer, or none); where its rows of a phase do not fill
// whole tiles the last tile is partly empty.  Measured on MI355X, the chain pays at every batch size down
// to 8 images (one a share: 0.409 ms per swarm iteration against 0.453 with a launch per layer, where
// the host can barely enqueue the 70 launches as fast as the device runs them), so the threshold
// (GDPSO_CHAIN_MIN_SHARE, images a share) defaults to 1.  A chain never splits K over workgroups (the
// planner does when a layer alone has fewer tiles than the device has CUs).
static int g_chain_min_share_images = [] {
    const char *e = getenv("GDPSO_CHAIN_MIN_SHARE");
    return e && atoi(e) > 0 ? atoi(e) : 1;
}();
static bool g_chain_wide = [] {
    const char *e = getenv("GDPSO_CHAIN_WIDE");  // 0: chain steps keep the planner's 64x64 tiles however many a workgroup gets
    return e ? e[0] != '0' : true;
}();
static int chain_cus() {
    static int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n;
    }();
    return cus;
}
int chain_tile(const ConvLayer &L, int n_images, int nshare) {
    if (!dtype_ok(L.dtype) || ceil_div(n_images, nshare) < g_chain_min_share_images) return 0;
    // (a layer planned with 128x128 tiles is long enough to stand alone)
    return (L.tile == 64 || L.tile == 32 || L.tile == TILE_32x32) ? L.tile : 0;
}
bool chain_layer_ok(const ConvLayer &L, int n_images, int nteams) { return chain_tile(L, n_images, 8 * (nteams == 2 ? 2 : 1)) != 0; }

// slice size - bytes of one share's rows (pixels per image x channels); 0 for a buffer in standard layout
static int chain_gap(const ConvChain &c, const ChainBuf &b, long pixels, int channels, unsigned *gap) {
    *gap = 0;
    if (b.bytes == 0) return GDPSO_OK;
    const size_t slice = b.bytes / c.nshare() / 16 * 16;
    const size_t share_bytes = (size_t)ceil_div(c.n_images, c.nshare()) * pixels * channels * dtype_size(c.dtype);
    if (slice < share_bytes || slice - share_bytes > 0xffffffffu) return fail(GDPSO_ERR_ARG, "chain: buffer too small for a layer's share");
    *gap = (unsigned)(slice - share_bytes);
    return GDPSO_OK;
}

static int chain_fill(const ConvChain &c, const ConvLayer &L, const ChainBuf &in, const ChainBuf *res, const ChainBuf &out,
                      ChainStep &st, int w) {
    long grid_x = 0;
    bool fr = false;
    ConvKArgs &a = st.a[w];
    int rc = conv_args(L, c.n_images, (const void *)in.offset, res ? (const void *)(uintptr_t)1 : nullptr, (void *)out.offset, nullptr, a,
                       &grid_x, &fr, c.nshare());
    if (rc) return rc;
    a.res = res ? (const void *)res->offset : (const void *)~(uintptr_t)0;
    a.snake = 0;
    st.tiles[w] = (int)(grid_x / c.nshare());
    st.shape[w] = L.tile;
    if ((rc = chain_gap(c, in, (long)L.hin * L.win, L.cin, &st.gap_in[w]))) return rc;
    if ((rc = chain_gap(c, out, (long)L.hout * L.wout, L.cout, &st.gap_out[w]))) return rc;
    if (res && (rc = chain_gap(c, *res, (long)L.hout * L.wout, L.cout, &st.gap_res[w]))) return rc;
    return GDPSO_OK;
}

// Appends a step: layer La (input at byte offset `in` of the workspace, optional residual, output), and
// optionally Lb reading the same input (no residual).
int chain_add(ConvChain *c, const ConvLayer *La, const ChainBuf &in, const ChainBuf *res, const ChainBuf &out_a, const ConvLayer *Lb,
              const ChainBuf *out_b) {
    if ((int)c->steps.size() >= CHAIN_MAX_STEPS) return fail(GDPSO_ERR_ARG, "chain: too many steps");
    const int ta = chain_tile(*La, c->n_images, c->nshare()), tb = Lb ? chain_tile(*Lb, c->n_images, c->nshare()) : 1;
    if (!ta || !tb) return fail(GDPSO_ERR_ARG, "chain: layer cannot be chained at this batch size");
    if (c->dtype < 0) c->dtype = La->dtype;
    if (La->dtype != c->dtype || (Lb && Lb->dtype != c->dtype)) return fail(GDPSO_ERR_ARG, "chain: layers of different element types");
    ChainStep st;
    std::memset(&st, 0, sizeof(st));
    int ta_use = ta, tb_use = tb;
    if (g_chain_wide) {
        // more 64x64 tiles than a team has members: the 64-row layers of the step whose width allows it take 64x128
        const long members = std::max(1, chain_cus() * 2 / 8 / c->nteams);
        const int share_images = ceil_div(c->n_images, c->nshare());
        long tiles = m_tiles(*La, share_images, ta) * (La->cout / tile_bn(ta));
        if (Lb) tiles += m_tiles(*Lb, share_images, tb) * (Lb->cout / tile_bn(tb));
        if (tiles > members) {
            if (ta == 64 && La->cout % 128 == 0) ta_use = TILE_64x128;
            if (Lb && tb == 64 && Lb->cout % 128 == 0) tb_use = TILE_64x128;
        }
    }
    c->layers.push_back(*La);
    c->layers.back().tile = ta_use;
    c->layers.back().splits = 1;
    int rc = chain_fill(*c, c->layers.back(), in, res, out_a, st, 0);
    if (rc) return rc;
    c->layer_has_res.push_back(res != nullptr);
    if (Lb) {
        if (!out_b) return fail(GDPSO_ERR_ARG, "chain: second layer without an output");
        c->layers.push_back(*Lb);
        c->layers.back().tile = tb_use;
        c->layers.back().splits = 1;
        if ((rc = chain_fill(*c, c->layers.back(), in, nullptr, *out_b, st, 1))) return rc;
        c->layer_has_res.push_back(0);
    }
    c->steps.push_back(st);
    return GDPSO_OK;
}

// Uploads the step table and sizes the grid: every workgroup must be resident at once.
int chain_finish(ConvChain *c, size_t ctl_off) {
    if (c->steps.empty()) return fail(GDPSO_ERR_ARG, "chain: no steps");
    int dev = 0, cus = 0, occ = 0;
    GDPSO_HIP(hipGetDevice(&dev));
    GDPSO_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    GDPSO_WITH_DTYPE(c->dtype, GDPSO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv_chain_kernel<E>, 256, 0)));
    if (occ < 1) return fail(GDPSO_ERR_HIP, "chain: the kernel does not fit a compute unit");
    c->grid = cus * std::min(occ, 2);
    if (c->grid > CHAIN_MAX_MEMBERS) return fail(GDPSO_ERR_ARG, "chain: more workgroups than the control block holds");
    c->ctl_off = ctl_off;
    GDPSO_HIP(hipMalloc(&c->dev_steps, c->steps.size() * sizeof(ChainStep)));
    GDPSO_HIP(hipMemcpy(c->dev_steps, c->steps.data(), c->steps.size() * sizeof(ChainStep), hipMemcpyHostToDevice));
    return GDPSO_OK;
}

// A chain's workgroups wait for each other, so its grid must be resident as a whole, and the
// workgroups that are already there hold their compute units while they wait for the rest.  Beside
// kernels of another stream that is a bad neighbour: the others crawl through the few free slots, the
// chain's last workgroups wait behind them (measured: half a second for what takes a millisecond), and
// two chains side by side could each hold half of the device.  So a device runs chains only while
// this library has seen a single stream on it: the entry points that launch networks report their
// stream here, and once a second stream shows up the first is drained (once) and every later forward
// pass on that device runs one launch per layer.  Kernels the caller launches on other streams
// without the library's knowledge only slow a chain down.
namespace {
struct ChainStreams {
    hipStream_t first = nullptr;
    bool any = false, several = false;
};
std::mutex g_chain_streams_mutex;
std::map<int, ChainStreams> g_chain_streams;  // by device
}  // namespace

int chain_note_stream(hipStream_t stream, bool *single) {
    int dev = 0;
    GDPSO_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_chain_streams_mutex);
    ChainStreams &cs = g_chain_streams[dev];
    if (!cs.any) {
        cs.any = true;
        cs.first = stream;
    } else if (!cs.several && cs.first != stream) {
        cs.several = true;
        // (a stream the caller has destroyed in the meantime has nothing left to wait for)
        if (hipStreamSynchronize(cs.first) != hipSuccess) (void)hipGetLastError();
    }
    if (single) *single = !cs.several;
    return GDPSO_OK;
}

int launch_chain(const ConvChain *c, hipStream_t stream, void *workspace, bool ctl_cleared) {
    char *ws = (char *)workspace;
    ChainCtl *ctl = (ChainCtl *)(ws + c->ctl_off);
    static_assert(sizeof(ChainCtl) % 16 == 0, "the control block is cleared in 16-byte words");
    if (!ctl_cleared) {  // (by a kernel, like the split-K counters: memset nodes of a captured graph are not relied on)
        static_assert(sizeof(ChainCtl) % 16 == 0, "the control block is cleared in 16-byte words");
        constexpr int quads = (int)(sizeof(ChainCtl) / 16);
        hipLaunchKernelGGL(clear_quads_kernel, dim3(ceil_div(quads, 256)), dim3(256), 0, stream, (f32x4 *)ctl, quads);
        GDPSO_HIP(hipGetLastError());
    }
    const bool timed = g_timing.on && g_timing.used + 2 <= g_timing.events.size();
    const dim3 grid((unsigned)c->grid);
    if (timed) {
        GDPSO_WITH_DTYPE(c->dtype, hipExtLaunchKernelGGL(conv_chain_kernel<E>, grid, dim3(256), 0, stream, g_timing.events[g_timing.used],
                                                         g_timing.events[g_timing.used + 1], 0, (const ChainStep *)c->dev_steps,
                                                         (int)c->steps.size(), ws, ctl, g_chain_drop_mask, c->nteams, g_chain_stagger, g_chain_peek));
        g_timing.used += 2;
        size_t li = 0;
        for (const ChainStep &st : c->steps)
            for (int w = 0; w < 2; ++w)
                if (st.tiles[w]) {
                    conv_account(c->layers[li], c->n_images, st.a[w], c->layer_has_res[li] != 0);
                    ++li;
                }
    } else {
        GDPSO_WITH_DTYPE(c->dtype, hipLaunchKernelGGL(conv_chain_kernel<E>, grid, dim3(256), 0, stream, (const ChainStep *)c->dev_steps,
                                                      (int)c->steps.size(), ws, ctl, g_chain_drop_mask, c->nteams, g_chain_stagger, g_chain_peek));
    }
    GDPSO_HIP(hipGetLastError());
    ++g_chain_launches;
    return GDPSO_OK;
}

}  // namespace gdpso

// Development / test switches of the chained launches: `enabled` 0 runs every layer as a launch of its
// own (plans made afterwards); `drop_xcc_mask` makes the workgroups on those XCDs stand aside, so that
// the other XCDs take their shares as well (the path for an uneven deal of workgroups to XCDs);
// `teams` (0: unchanged) sets the teams per XCD of plans made afterwards.
extern "C" int gdpso_chain_debug(int enabled, unsigned drop_xcc_mask, int teams) {
    using namespace gdpso;
    GDPSO_REQUIRE((drop_xcc_mask & 0xffu) != 0xffu, "chain_debug: at least one XCD must keep its workgroups");
    GDPSO_REQUIRE(teams >= 0 && teams <= 2, "chain_debug: teams must be 0 (keep), 1 or 2");
    g_chain_enabled = enabled != 0;
    g_chain_drop_mask = drop_xcc_mask & 0xffu;
    if (teams) g_chain_teams = teams;
    {
        std::lock_guard<std::mutex> lock(g_chain_streams_mutex);
        g_chain_streams.clear();  // forget the streams seen so far (a test that used several on purpose)
    }
    return GDPSO_OK;
}

// Options of launch plans made from now on (handles keep the plans they have already made for a batch
// size).  plan_images > 0: tile shapes and split counts are chosen as if every batch held that many
// images (0: each batch by its own size); splitk = 0: K is never split over workgroups.
extern "C" int gdpso_plan_options(int plan_images, int splitk) {
    using namespace gdpso;
    GDPSO_REQUIRE(plan_images >= 0, "plan_options: plan_images must be 0 (off) or a batch size");
    g_plan_images = plan_images;
    g_plan_splitk = splitk != 0;
    return GDPSO_OK;
}

// Chained launches since the library was loaded, and how many of them ran out of patience waiting for
// their workgroups or for a step (0 on a healthy device).  Synchronises the device.
extern "C" int gdpso_chain_stats(unsigned *launches, unsigned *count) {
    using namespace gdpso;
    GDPSO_REQUIRE(launches && count, "chain_stats: null pointer");
    *launches = g_chain_launches.load();
    GDPSO_HIP(hipMemcpyFromSymbol(count, HIP_SYMBOL(g_chain_aborts), sizeof(unsigned)));
    unsigned bits = 0;
    GDPSO_HIP(hipMemcpyFromSymbol(&bits, HIP_SYMBOL(g_chain_abort_bits), sizeof(unsigned)));
    unsigned arrived = 0;
    GDPSO_HIP(hipMemcpyFromSymbol(&arrived, HIP_SYMBOL(g_chain_abort_arrived), sizeof(unsigned)));
    if (*count) set_error("chain aborts: waited in vain for " + std::string(bits & 2 ? "the registration of all workgroups (" + std::to_string(arrived) + " had registered) " : "") +
                          std::string(bits & 1 ? "a step's flags" : ""));
    return GDPSO_OK;
}

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
    g_timing.bytes = 0.0;
    g_timing.executed = 0.0;
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

#ifdef GDPSO_STAMPS
extern "C" int gdpso_conv_stamps_set(void *device_buffer) {
    GDPSO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gdpso::g_stamps), &device_buffer, sizeof(void *)));
    return GDPSO_OK;
}
// [workgroups][CHAIN_MAX_STEPS = 192][4] 64-bit stamps
extern "C" int gdpso_chain_stamps_set(void *device_buffer, int tile_step) {
    GDPSO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gdpso::g_chain_stamps), &device_buffer, sizeof(void *)));
    GDPSO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gdpso::g_chain_stamp_step), &tile_step, sizeof(int)));
    return GDPSO_OK;
}

#endif

extern "C" double gdpso_conv_timing_bytes(void) { return gdpso::g_timing.bytes; }
extern "C" double gdpso_conv_timing_executed_flops(void) { return gdpso::g_timing.executed; }

extern "C" int gdpso_conv_plan_describe(int n_images, int hin, int cin, int cout, int ksize, int stride, int pad,
                                        int transposed, int num_cus, int *out, int max_ints) {
    using namespace gdpso;
    if (!(n_images > 0 && hin > 0 && cin > 0 && cout > 0 && num_cus > 0 && out && max_ints > 0)) {
        set_error("conv_plan_describe: bad argument");
        return -GDPSO_ERR_ARG;
    }
    if (!transposed && !(ksize > 0 && stride > 0 && pad >= 0 && hin + 2 * pad >= ksize)) {
        set_error("conv_plan_describe: bad window");
        return -GDPSO_ERR_ARG;
    }
    if (!transposed && !conv_window_fits(ksize, pad)) {
        set_error("conv_plan_describe: window too large (at most 64 taps, offsets within +-127)");
        return -GDPSO_ERR_ARG;
    }
    ConvLayer L;
    std::vector<float> packed;
    if (transposed) {
        std::vector<float> w((size_t)cin * cout * 16, 0.f);
        make_convt_k4s2p1(L, w.data(), cin, cout, hin, hin, ACT_NONE, packed);
    } else {
        std::vector<float> w((size_t)cout * cin * ksize * ksize, 0.f);
        make_conv(L, w.data(), cout, cin, ksize, ksize, stride, pad, hin, hin, ACT_NONE, packed);
    }
    if (plan_conv(L, n_images, num_cus)) return -GDPSO_ERR_ARG;
    std::vector<int> v = {L.nphase, L.tile, L.splits, L.hout, L.wout};
    for (int p = 0; p < L.nphase; ++p) {
        const ConvPhase &P = L.phase[p];
        for (int x : {P.ntaps, P.off_y, P.off_x, P.hm, P.wm, P.in_y0, P.in_x0}) v.push_back(x);
        for (int t = 0; t < P.ntaps; ++t) {
            v.push_back(P.dy[t]);
            v.push_back(P.dx[t]);
            v.push_back(P.wtap[t]);
        }
    }
    if ((int)v.size() > max_ints) {
        set_error("conv_plan_describe: output buffer too small");
        return -GDPSO_ERR_ARG;
    }
    std::copy(v.begin(), v.end(), out);
    return (int)v.size();
}

// The tile a layer would run with as a step of a chained launch at this batch size (0: it cannot be a
// step), after the planner's own choice for the layer alone.  Host only.
extern "C" int gdpso_chain_tile_describe(int n_images, int hin, int cin, int cout, int ksize, int stride, int pad, int num_cus,
                                         int teams, int dtype) {
    using namespace gdpso;
    if (!(n_images > 0 && hin > 0 && cin > 0 && cout > 0 && num_cus > 0 && ksize > 0 && stride > 0 && pad >= 0 && hin + 2 * pad >= ksize &&
          conv_window_fits(ksize, pad) && (teams == 1 || teams == 2) && dtype_ok(dtype))) {
        set_error("chain_tile_describe: bad argument");
        return -GDPSO_ERR_ARG;
    }
    ConvLayer L;
    std::vector<float> packed, w((size_t)cout * cin * ksize * ksize, 0.f);
    make_conv(L, w.data(), cout, cin, ksize, ksize, stride, pad, hin, hin, ACT_NONE, packed);
    L.dtype = dtype;
    if (plan_conv(L, n_images, num_cus)) return -GDPSO_ERR_ARG;
    return chain_tile(L, n_images, 8 * teams);
}

// Times one convolution layer in isolation (development / tuning aid): random weights and
// inputs, `reps` back-to-back launches bracketed by HIP events.  tile = 0 / splits = 0 keep
// the planner's choice.
extern "C" int gdpso_conv_probe(int n_images, int hin, int cin, int cout, int ksize, int stride, int pad, int tile,
                                int splits, int with_res, int reps, double *avg_us, int *tile_used, int *splits_used) {
    return gdpso_conv_probe_typed(n_images, hin, cin, cout, ksize, stride, pad, tile, splits, with_res, reps, avg_us, tile_used,
                                  splits_used, GDPSO_DTYPE_F32);
}

extern "C" int gdpso_conv_probe_typed(int n_images, int hin, int cin, int cout, int ksize, int stride, int pad, int tile,
                                      int splits, int with_res, int reps, double *avg_us, int *tile_used, int *splits_used,
                                      int dtype) {
    using namespace gdpso;
    GDPSO_REQUIRE(dtype_ok(dtype), "conv_probe: unknown dtype");
    const int kq = dtype_kquant(dtype);
    GDPSO_REQUIRE(n_images > 0 && hin > 0 && cin % kq == 0 && cout % 64 == 0 && reps > 0 && avg_us, "conv_probe: bad argument");
    GDPSO_REQUIRE(ksize > 0 && stride > 0 && pad >= 0 && hin + 2 * pad >= ksize && conv_window_fits(ksize, pad),
                  "conv_probe: bad window (at most 64 taps, offsets within +-127)");
    GDPSO_REQUIRE(tile == 0 || tile == 64 || (tile == 32 && cin % (2 * kq) == 0) || (tile == TILE_32x32 && cin % (4 * kq) == 0) ||
                      (tile == 128 && cout % 128 == 0),
                  "conv_probe: bad tile");
    std::vector<float> w((size_t)cout * cin * ksize * ksize), packed, sc(cout, 1.f), sh(cout, 0.1f);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto &v : w) v = rnd() * 0.1f;
    ConvLayer L;
    make_conv(L, w.data(), cout, cin, ksize, ksize, stride, pad, hin, hin, ACT_RELU, packed);
    DeviceArena arena;
    int rc = upload_conv(arena, L, packed, sc, sh, dtype);
    if (rc) return rc;
    int dev = 0, cus = 256;
    GDPSO_HIP(hipGetDevice(&dev));
    GDPSO_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if ((rc = plan_conv(L, n_images, cus))) return rc;
    if (tile) L.tile = tile;
    if (splits) L.splits = splits;
    const size_t in_elems = (size_t)n_images * hin * hin * cin, out_elems = (size_t)n_images * L.hout * L.wout * cout;
    const void *d_in = nullptr;
    if (dtype == DT_F32) {
        std::vector<float> hin_v(in_elems);
        for (auto &v : hin_v) v = rnd();
        if ((rc = arena.upload(hin_v.data(), in_elems * 4, &d_in))) return rc;
    } else {
        std::vector<uint16_t> hin_v(in_elems);
        for (auto &v : hin_v) v = dtype == DT_BF16 ? f32_to_bf16_bits(rnd()) : f32_to_f16_bits(rnd());
        if ((rc = arena.upload(hin_v.data(), in_elems * 2, &d_in))) return rc;
    }
    void *d_out = nullptr, *d_res = nullptr;
    float *d_sk = nullptr;
    GDPSO_HIP(hipMalloc(&d_out, out_elems * 4));
    arena.blocks.push_back(d_out);
    GDPSO_HIP(hipMalloc(&d_res, out_elems * 4));
    arena.blocks.push_back(d_res);
    GDPSO_HIP(hipMemset(d_res, 0, out_elems * 4));
    GDPSO_HIP(hipMalloc(&d_sk, std::max<size_t>(4, conv_splitk_bytes(L, n_images))));
    arena.blocks.push_back(d_sk);
    if ((rc = conv_splitk_prepare(nullptr, d_sk))) return rc;
    hipEvent_t e0, e1;
    GDPSO_HIP(hipEventCreate(&e0));
    GDPSO_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        if ((rc = launch_conv(L, nullptr, n_images, d_in, with_res ? d_res : nullptr, d_out, d_sk))) return rc;
    GDPSO_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; ++i)
        if ((rc = launch_conv(L, nullptr, n_images, d_in, with_res ? d_res : nullptr, d_out, d_sk))) return rc;
    GDPSO_HIP(hipEventRecord(e1, nullptr));
    GDPSO_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    GDPSO_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = 1e3 * ms / reps;
    if (tile_used) *tile_used = L.tile;
    if (splits_used) *splits_used = L.splits;
    return GDPSO_OK;
}

// Element-wise casts between fp32 arrays and the 16-bit storage types (round to nearest even).
template <typename E>
__global__ __launch_bounds__(256) void cast_from_f32_kernel(const float *__restrict__ in, E *__restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (E)in[i];
}
template <typename E>
__global__ __launch_bounds__(256) void cast_to_f32_kernel(const E *__restrict__ in, float *__restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// Runs ONE convolution layer on the caller's data (test aid: lets a test drive every tile
// shape and split count on arbitrary layer shapes, which whole-network tests reach only by
// way of the planner).  weight/scale/shift are host arrays (weight in PyTorch layout:
// [cout][cin][k][k], or [cin][cout][4][4] when `transposed`, which means ConvTranspose2d
// k4 s2 p1); in/res/out are device NHWC fp32 arrays.  tile = 0 / splits = 0 keep the planner's
// choice.  With a 16-bit dtype the layer runs on copies of in / res rounded to that type and
// its output is widened back into `out`.  Allocates and frees its own device copies;
// synchronises before returning.
extern "C" int gdpso_conv_layer_run_typed(int n_images, int hin, int cin, int cout, int ksize, int stride, int pad,
                                          int transposed, int act, int tile, int splits, const float *weight,
                                          const float *scale, const float *shift, const float *in, const float *res,
                                          float *out, int *tile_used, int *splits_used, int dtype) {
    using namespace gdpso;
    GDPSO_REQUIRE(dtype_ok(dtype), "conv_layer_run: unknown dtype");
    const int kq = dtype_kquant(dtype);
    GDPSO_REQUIRE(n_images > 0 && hin > 0 && cin > 0 && cin % kq == 0 && cout > 0 && cout % 64 == 0, "conv_layer_run: bad shape");
    GDPSO_REQUIRE(weight && scale && shift && in && out, "conv_layer_run: null pointer");
    GDPSO_REQUIRE(act == ACT_NONE || act == ACT_RELU || act == ACT_LEAKY02, "conv_layer_run: bad activation");
    GDPSO_REQUIRE(transposed || (ksize > 0 && stride > 0 && pad >= 0 && hin + 2 * pad >= ksize), "conv_layer_run: bad window");
    GDPSO_REQUIRE(transposed || conv_window_fits(ksize, pad), "conv_layer_run: window too large (at most 64 taps, offsets within +-127)");
    GDPSO_REQUIRE(tile == 0 || tile == 64 || (tile == 32 && cin % (2 * kq) == 0) || (tile == TILE_32x32 && cin % (4 * kq) == 0) ||
                      (tile == 128 && cout % 128 == 0),
                  "conv_layer_run: bad tile");
    GDPSO_REQUIRE(splits >= 0 && splits <= 16, "conv_layer_run: bad split count");
    ConvLayer L;
    std::vector<float> packed, sc(scale, scale + cout), sh(shift, shift + cout);
    if (transposed)
        make_convt_k4s2p1(L, weight, cin, cout, hin, hin, (Act)act, packed);
    else
        make_conv(L, weight, cout, cin, ksize, ksize, stride, pad, hin, hin, (Act)act, packed);
    DeviceArena arena;
    int rc = upload_conv(arena, L, packed, sc, sh, dtype);
    if (rc) return rc;
    int dev = 0, cus = 256;
    GDPSO_HIP(hipGetDevice(&dev));
    GDPSO_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if ((rc = plan_conv(L, n_images, cus))) return rc;
    if (tile) L.tile = tile;
    if (splits) L.splits = splits;
    float *d_sk = nullptr;
    GDPSO_HIP(hipMalloc(&d_sk, conv_splitk_bytes(L, n_images)));
    arena.blocks.push_back(d_sk);
    GDPSO_HIP(hipMemset(d_sk, 0, SPLITK_COUNTER_BYTES));
    const long n_in = (long)n_images * hin * hin * cin, n_out = (long)n_images * L.hout * L.wout * cout;
    const void *k_in = in, *k_res = res;
    void *k_out = out;
    if (dtype != DT_F32) {
        void *p_in = nullptr, *p_res = nullptr, *p_out = nullptr;
        GDPSO_HIP(hipMalloc(&p_in, n_in * 2));
        arena.blocks.push_back(p_in);
        GDPSO_HIP(hipMalloc(&p_out, n_out * 2));
        arena.blocks.push_back(p_out);
        GDPSO_WITH_DTYPE(dtype, hipLaunchKernelGGL(cast_from_f32_kernel<E>, dim3(ceil_div(n_in, 256)), dim3(256), 0, nullptr, in, (E *)p_in, n_in));
        if (res) {
            GDPSO_HIP(hipMalloc(&p_res, n_out * 2));
            arena.blocks.push_back(p_res);
            GDPSO_WITH_DTYPE(dtype, hipLaunchKernelGGL(cast_from_f32_kernel<E>, dim3(ceil_div(n_out, 256)), dim3(256), 0, nullptr, res, (E *)p_res, n_out));
        }
        k_in = p_in; k_res = p_res; k_out = p_out;
    }
    if ((rc = launch_conv(L, nullptr, n_images, k_in, k_res, k_out, d_sk))) return rc;
    if (dtype != DT_F32)
        GDPSO_WITH_DTYPE(dtype, hipLaunchKernelGGL(cast_to_f32_kernel<E>, dim3(ceil_div(n_out, 256)), dim3(256), 0, nullptr, (const E *)k_out, out, n_out));
    GDPSO_HIP(hipGetLastError());
    GDPSO_HIP(hipStreamSynchronize(nullptr));
    if (tile_used) *tile_used = L.tile;
    if (splits_used) *splits_used = L.splits;
    return GDPSO_OK;
}

extern "C" int gdpso_conv_layer_run(int n_images, int hin, int cin, int cout, int ksize, int stride, int pad,
                                    int transposed, int act, int tile, int splits, const float *weight,
                                    const float *scale, const float *shift, const float *in, const float *res,
                                    float *out, int *tile_used, int *splits_used) {
    return gdpso_conv_layer_run_typed(n_images, hin, cin, cout, ksize, stride, pad, transposed, act, tile, splits, weight, scale,
                                      shift, in, res, out, tile_used, splits_used, GDPSO_DTYPE_F32);
}
