// Swarm arithmetic of reference src/pso/util_pso.py:24-193 on structure-of-arrays state.
//
// Positions, velocities and personal-best positions are float64 like the reference's
// numpy arrays; fitness values are float32 like its torch results.  This file is built
// with -ffp-contract=off and every float64 expression below is written in the
// reference's evaluation order, so positions and velocities are bit-identical to numpy's
// for identical fitness decisions.  These kernels are HBM/latency bound (a few hundred
// bytes per particle): elements are mapped to consecutive lanes for coalesced access.
#include <algorithm>
#include <cstring>

#include "common.h"

namespace gdpso {

constexpr int PB = 256;  // particles per workgroup in the personal-best kernel

struct BestRec {
    float val;
    int idx;
};

// min-loc combine with the reference's scan order: strict '<' in increasing particle index
// (util_pso.py:139-142) == smallest value, lowest index among equals; NaN never wins.
__device__ __forceinline__ BestRec better_of(BestRec a, BestRec b) {
    if (b.val < a.val || (b.val == a.val && b.idx < a.idx)) return b;
    return a;
}

__device__ __forceinline__ BestRec wave_best(BestRec r) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        BestRec t;
        t.val = __shfl_xor(r.val, o);
        t.idx = __shfl_xor(r.idx, o);
        r = better_of(r, t);
    }
    return r;
}

__device__ __forceinline__ bool running(const double *ctrl, int iteration) {
    const int stop = (int)ctrl[GDPSO_CTRL_STOP_ITER];
    return stop == 0 || iteration <= stop;
}

// Swarm.__init__ / Particle.__init__ bookkeeping.
__global__ __launch_bounds__(256) void reset_kernel(gdpso_swarm s, double w_inertia) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long nd = (long)s.n * s.d;
    if (e < nd) {
        s.pbest_pos[e] = s.pos[e];
        if (s.hist_pos) s.hist_pos[e] = s.pos[e];
        if (s.hist_vel) s.hist_vel[e] = s.vel[e];
    }
    if (e < s.n) s.pbest_val[e] = INFINITY;
    if (e < s.d) s.gbest_pos[e] = 0.0;
    if (e <= s.max_iter) s.g_best_val[e] = INFINITY;
    if (e < s.max_iter) {
        s.g_best_val_dummy[e] = INFINITY;
        if (s.mean_dist) s.mean_dist[e] = 0.0;
    }
    if (e < GDPSO_CTRL_DOUBLES) {
        double v = 0.0;
        if (e == GDPSO_CTRL_W) v = w_inertia;
        if (e == GDPSO_CTRL_GBEST_VAL) v = (double)INFINITY;
        if (e == GDPSO_CTRL_GBEST_IDX) v = -1.0;
        if (e == GDPSO_CTRL_N_IMPROVE) v = 1.0;  // g_best_val starts as [inf]
        s.ctrl[e] = v;
    }
}

// update_personal_best: "if fitness < p_best_val: p_best_val, p_best_pos = fitness, pos"
// (util_pso.py:122-124) for PB particles per workgroup, then the workgroup's min-loc over
// the updated personal bests -> partial[blockIdx.x].
__global__ __launch_bounds__(256) void pbest_kernel(gdpso_swarm s, int iteration, BestRec *__restrict__ partial) {
    __shared__ unsigned char s_better[PB];
    __shared__ BestRec s_best[4];
    if (!running(s.ctrl, iteration)) return;
    const int tid = threadIdx.x, p0 = blockIdx.x * PB, p = p0 + tid;
    BestRec r;
    r.val = INFINITY;
    r.idx = 0x7fffffff;
    bool better = false;
    if (p < s.n) {
        const float f = s.fitness[p];
        float pb = s.pbest_val[p];
        better = f < pb;
        if (better) {
            pb = f;
            s.pbest_val[p] = f;
        }
        r.val = pb;
        r.idx = s.first_index + p;
    }
    s_better[tid] = better;
    r = wave_best(r);
    if ((tid & 63) == 0) s_best[tid >> 6] = r;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = better_of(better_of(s_best[0], s_best[1]), better_of(s_best[2], s_best[3]));
    const int cnt = min(PB, s.n - p0);
    const long base = (long)p0 * s.d;
    for (int e = tid; e < cnt * s.d; e += 256)
        if (s_better[e / s.d]) s.pbest_pos[base + e] = s.pos[base + e];
}

// Reduces the per-workgroup partials to this rank's candidate record
// {value, global index, personal-best position}.
__global__ __launch_bounds__(256) void candidate_kernel(gdpso_swarm s, int iteration, const BestRec *__restrict__ partial,
                                                        int n_partial, int rank_slot) {
    __shared__ BestRec s_best[4];
    if (!running(s.ctrl, iteration)) return;
    const int tid = threadIdx.x;
    BestRec r;
    r.val = INFINITY;
    r.idx = 0x7fffffff;
    for (int i = tid; i < n_partial; i += 256) r = better_of(r, partial[i]);
    r = wave_best(r);
    if ((tid & 63) == 0) s_best[tid >> 6] = r;
    __syncthreads();
    r = better_of(better_of(s_best[0], s_best[1]), better_of(s_best[2], s_best[3]));
    double *rec = s.cand + (size_t)rank_slot * (s.d + 2);
    const bool have = r.idx != 0x7fffffff && r.val < INFINITY;
    if (tid == 0) {
        rec[0] = have ? (double)r.val : (double)INFINITY;
        rec[1] = have ? (double)r.idx : -1.0;
    }
    for (int k = tid; k < s.d; k += 256) rec[2 + k] = have ? s.pbest_pos[(size_t)(r.idx - s.first_index) * s.d + k] : 0.0;
}

// update_global_best (util_pso.py:135-151) over the candidate records of all ranks, the
// inertia decay (:178-179) and the early-stopping test (:186-188).  One workgroup; every
// rank runs it on identical inputs and reaches identical decisions.
// One thread: the decision part of update_global_best.  Returns the winning slot or -1.
__device__ int decide_global_best(const gdpso_swarm &s, int iteration, int schedule_inertia, int early_stopping,
                                  double tolerance) {
    float cur = (float)s.ctrl[GDPSO_CTRL_GBEST_VAL];
    float best = cur;
    int best_slot = -1, best_idx = 0x7fffffff;
    for (int k = 0; k < s.n_ranks; ++k) {
        const double *rec = s.cand + (size_t)k * (s.d + 2);
        const float v = (float)rec[0];
        const int idx = (int)rec[1];
        if (idx < 0) continue;
        // strict improvement over the stored best; among ranks, lowest global index wins ties
        if (v < best || (best_slot >= 0 && v == best && idx < best_idx)) {
            best = v;
            best_slot = k;
            best_idx = idx;
        }
    }
    int n_imp = (int)s.ctrl[GDPSO_CTRL_N_IMPROVE];
    if (best_slot >= 0) {
        if (isinf(cur) && cur > 0.f)
            s.g_best_val[n_imp - 1] = best;  // the first improvement overwrites the inf
        else
            s.g_best_val[n_imp++] = best;
        s.ctrl[GDPSO_CTRL_GBEST_VAL] = (double)best;
        s.ctrl[GDPSO_CTRL_GBEST_IDX] = (double)best_idx;
        s.ctrl[GDPSO_CTRL_N_IMPROVE] = (double)n_imp;
    }
    s.g_best_val_dummy[iteration - 1] = best;
    if (iteration > 1 && schedule_inertia) s.ctrl[GDPSO_CTRL_W] = 0.99 * s.ctrl[GDPSO_CTRL_W];
    if (iteration > 2 && n_imp > 2 && early_stopping) {
        const float diff = fabsf(s.g_best_val[n_imp - 1] - s.g_best_val[n_imp - 2]);
        if (diff < (float)tolerance) s.ctrl[GDPSO_CTRL_STOP_ITER] = (double)iteration;
    }
    s.ctrl[GDPSO_CTRL_ITERS_DONE] = (double)iteration;
    return best_slot;
}

__global__ __launch_bounds__(64) void global_best_kernel(gdpso_swarm s, int iteration, int schedule_inertia,
                                                         int early_stopping, double tolerance) {
    __shared__ int s_take;
    if (!running(s.ctrl, iteration)) return;
    const int tid = threadIdx.x;
    if (tid == 0) s_take = decide_global_best(s, iteration, schedule_inertia, early_stopping, tolerance);
    __syncthreads();
    const int slot = s_take;
    if (slot >= 0) {
        const double *rec = s.cand + (size_t)slot * (s.d + 2);
        for (int k = tid; k < s.d; k += 64) s.gbest_pos[k] = rec[2 + k];
    }
}

// Philox4x32-10 (Salmon et al. 2011), counter = (particle, iteration), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    // same construction as numpy's random(): 53 random bits / 2^53, in [0, 1)
    return (double)((((uint64_t)(hi >> 5)) << 26) | (uint64_t)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

// move_particles: update_velocity then update_position (util_pso.py:43-50, 39-41):
//   inertia   = w * vel
//   best_cogn = (w_cogn * r1) * (g_best_pos - pos)      (r1 pairs with the GLOBAL best)
//   best_soci = (w_soci * r2) * (p_best_pos - pos)      (r2 pairs with the PERSONAL best)
//   vel = (inertia + best_soci) + best_cogn ;  pos = pos + vel
// r1, r2 are scalars per particle.  One thread per (particle, dimension) element.
__device__ __forceinline__ void move_element(const gdpso_swarm &s, int iteration, long e, double w, double w_cogn,
                                             double w_soci, const double *__restrict__ rnd, uint64_t seed, int f32_terms) {
    const int p = (int)(e / s.d), k = (int)(e - (long)p * s.d);
    double r1, r2;
    if (rnd) {
        r1 = rnd[2 * (size_t)p];
        r2 = rnd[2 * (size_t)p + 1];
    } else {
        uint32_t o[4];
        philox4x32_10((uint32_t)(s.first_index + p), 0u, (uint32_t)iteration, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        r1 = u53(o[0], o[1]);
        r2 = u53(o[2], o[3]);
    }
    const double x = s.pos[e], v = s.vel[e];
    const double inertia = w * v;
    const double c1 = w_cogn * r1;
    const double c2 = w_soci * r2;
    double best_cogn, best_soci;
    if (f32_terms) {
        // float32 positions (first move of the inverter variant): numpy keeps float32 for
        // array - array and for python-float * float32-array
        const float xf = (float)x;
        best_cogn = (double)((float)c1 * ((float)s.gbest_pos[k] - xf));
        best_soci = (double)((float)c2 * ((float)s.pbest_pos[e] - xf));
    } else {
        best_cogn = c1 * (s.gbest_pos[k] - x);
        best_soci = c2 * (s.pbest_pos[e] - x);
    }
    const double nv = (inertia + best_soci) + best_cogn;
    const double nx = x + nv;
    s.vel[e] = nv;
    s.pos[e] = nx;
    const size_t h = (size_t)iteration * s.n * s.d + e;
    if (s.hist_pos) s.hist_pos[h] = nx;
    if (s.hist_vel) s.hist_vel[h] = nv;
}

__global__ __launch_bounds__(256) void move_kernel(gdpso_swarm s, int iteration, double w_cogn, double w_soci,
                                                   const double *__restrict__ rnd, uint64_t seed, int f32_terms) {
    if (!running(s.ctrl, iteration)) return;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)s.n * s.d) return;
    move_element(s, iteration, e, s.ctrl[GDPSO_CTRL_W], w_cogn, w_soci, rnd, seed, f32_terms);
}

// The four phases above for a swarm small enough for one workgroup (single process): personal
// bests, this swarm's candidate, the global-best decision and the move, separated by
// workgroup barriers instead of kernel boundaries.  Same device functions, same results.
constexpr int SMALL_THREADS = 1024;
constexpr int SMALL_MAX_N = 4096;
// ... and of at most 16 state elements per thread: beyond that the move phase of a single workgroup
// (Philox and float64 arithmetic per element) takes longer than the four launches it replaces
// (measured at 512 particles x 100 dimensions: 82 us against 25 us)
constexpr long SMALL_MAX_ELEMS = 16L * 1024;
// Personal bests and this rank's candidate record (slot rank_slot of s.cand), one workgroup.
__device__ __forceinline__ void small_local_phase(const gdpso_swarm &s, int rank_slot, unsigned char *s_better, BestRec *s_best) {
    const int tid = threadIdx.x;
    // update_personal_best
    BestRec r;
    r.val = INFINITY;
    r.idx = 0x7fffffff;
    for (int p = tid; p < s.n; p += SMALL_THREADS) {
        const float f = s.fitness[p];
        float pb = s.pbest_val[p];
        const bool better = f < pb;
        if (better) {
            pb = f;
            s.pbest_val[p] = f;
        }
        s_better[p] = better;
        BestRec c;
        c.val = pb;
        c.idx = s.first_index + p;
        r = better_of(r, c);
    }
    r = wave_best(r);
    if ((tid & 63) == 0) s_best[tid >> 6] = r;
    __syncthreads();
    const long nd = (long)s.n * s.d;
    for (long e = tid; e < nd; e += SMALL_THREADS)
        if (s_better[e / s.d]) s.pbest_pos[e] = s.pos[e];
    r = s_best[0];
#pragma unroll
    for (int w = 1; w < SMALL_THREADS / 64; ++w) r = better_of(r, s_best[w]);
    __threadfence_block();
    __syncthreads();
    // this swarm's candidate record
    double *rec = s.cand + (size_t)rank_slot * (s.d + 2);
    const bool have = r.idx != 0x7fffffff && r.val < INFINITY;
    if (tid == 0) {
        rec[0] = have ? (double)r.val : (double)INFINITY;
        rec[1] = have ? (double)r.idx : -1.0;
    }
    for (int k = tid; k < s.d; k += SMALL_THREADS) rec[2 + k] = have ? s.pbest_pos[(size_t)(r.idx - s.first_index) * s.d + k] : 0.0;
}

// One whole PSO step of one swarm by one workgroup (every thread of the workgroup calls it).
__device__ __forceinline__ void small_step(const gdpso_swarm &s, int iteration, int schedule_inertia, int early_stopping,
                                           double tolerance, double w_cogn, double w_soci, const double *__restrict__ rnd,
                                           uint64_t seed, int f32_terms, unsigned char *s_better, BestRec *s_best, int *s_take_p) {
    int &s_take = *s_take_p;
    const int tid = threadIdx.x;
    small_local_phase(s, 0, s_better, s_best);
    const double *rec = s.cand;
    __threadfence_block();
    __syncthreads();
    // update_global_best, inertia decay, early-stopping test
    if (tid == 0) s_take = decide_global_best(s, iteration, schedule_inertia, early_stopping, tolerance);
    __syncthreads();
    if (s_take >= 0)
        for (int k = tid; k < s.d; k += SMALL_THREADS) s.gbest_pos[k] = rec[2 + k];
    __threadfence_block();
    __syncthreads();
    // move_particles
    const long nd = (long)s.n * s.d;
    const double w = s.ctrl[GDPSO_CTRL_W];
    for (long e = tid; e < nd; e += SMALL_THREADS) move_element(s, iteration, e, w, w_cogn, w_soci, rnd, seed, f32_terms);
}

__global__ __launch_bounds__(SMALL_THREADS) void swarm_step_small_kernel(gdpso_swarm s, int iteration, int schedule_inertia,
                                                                         int early_stopping, double tolerance, double w_cogn,
                                                                         double w_soci, const double *__restrict__ rnd,
                                                                         uint64_t seed, int f32_terms) {
    __shared__ unsigned char s_better[SMALL_MAX_N];
    __shared__ BestRec s_best[SMALL_THREADS / 64];
    __shared__ int s_take;
    if (!running(s.ctrl, iteration)) return;
    small_step(s, iteration, schedule_inertia, early_stopping, tolerance, w_cogn, w_soci, rnd, seed, f32_terms, s_better, s_best,
               &s_take);
}

// Several independent swarms that share one evaluation batch (the reference walks its classes one
// after the other, src/training/pso_discovery.py:174-218; each has its own global best, inertia,
// draws and stopping state): workgroup b steps swarm b.  Descriptors travel as kernel arguments.
constexpr int MULTI_MAX = 8;  // swarms per launch (more are stepped by further launches)
struct SwarmBatch {
    gdpso_swarm s[MULTI_MAX];
    const double *rnd[MULTI_MAX];  // this iteration's (r1, r2) rows, or null (device RNG)
    unsigned long long seed[MULTI_MAX];
    double w_cogn[MULTI_MAX], w_soci[MULTI_MAX], tolerance[MULTI_MAX];
    int schedule_inertia[MULTI_MAX], early_stopping[MULTI_MAX], f32_terms[MULTI_MAX];
    int dist_tile[MULTI_MAX], dist_tiles[MULTI_MAX];  // mean-distance tiling of each swarm
    double *dist_partial[MULTI_MAX];
};
static_assert(sizeof(SwarmBatch) <= 3072, "kernel arguments are limited to 4 KiB");

// a swarm takes part in `iteration` while it has not stopped and the iteration is within its own budget
__device__ __forceinline__ bool running_multi(const gdpso_swarm &s, int iteration) {
    return iteration <= s.max_iter && running(s.ctrl, iteration);
}

__global__ __launch_bounds__(SMALL_THREADS) void swarm_step_small_multi_kernel(SwarmBatch b, int iteration) {
    __shared__ unsigned char s_better[SMALL_MAX_N];
    __shared__ BestRec s_best[SMALL_THREADS / 64];
    __shared__ int s_take;
    const int k = blockIdx.x;
    const gdpso_swarm &s = b.s[k];
    if (!running_multi(s, iteration)) return;
    small_step(s, iteration, b.schedule_inertia[k], b.early_stopping[k], b.tolerance[k], b.w_cogn[k], b.w_soci[k], b.rnd[k],
               b.seed[k], b.f32_terms[k], s_better, s_best, &s_take);
}

// The local half alone, for the sharded loop (the exchange of the candidate records follows it).
__global__ __launch_bounds__(SMALL_THREADS) void swarm_local_small_kernel(gdpso_swarm s, int iteration, int rank_slot) {
    __shared__ unsigned char s_better[SMALL_MAX_N];
    __shared__ BestRec s_best[SMALL_THREADS / 64];
    if (!running(s.ctrl, iteration)) return;
    small_local_phase(s, rank_slot, s_better, s_best);
}

// Swarm.mse (util_pso.py:76-86): mean Euclidean distance over unordered pairs.  Tile of
// 64 x 64 pairs per workgroup (upper triangle only), float64 throughout; per-workgroup sums
// are combined in a fixed order by the finishing kernel, so the result is reproducible.
// DT particles per tile side: a power of two in [8, 64] whose two tiles fit 48 KiB, smaller
// for small swarms (the tile side fixes the summation order, so it depends on n and d only).
constexpr int DT_MIN = 8;
static int dist_tile_side(int d, int n) {
    int dt = 64;
    while (dt > DT_MIN && (size_t)2 * dt * d * sizeof(double) > 48 * 1024) dt >>= 1;
    // small swarms: smaller tiles until there are enough workgroups to spread over the chip
    while (dt > 16) {
        const long t = (n + dt - 1) / dt;
        if (t * (t + 1) / 2 >= 128) break;
        dt >>= 1;
    }
    return dt;
}
__device__ __forceinline__ void pair_dist_tile(const double *__restrict__ pos, int n, int d, int DT, int tiles, int block,
                                                double *__restrict__ partial, double *sd, double *s_sum) {
    // linear block id -> (ti <= tj) upper-triangular tile
    int b = block, ti = 0;
    while (b >= tiles - ti) {
        b -= tiles - ti;
        ++ti;
    }
    const int tj = ti + b;
    const int tid = threadIdx.x;
    double *si = sd, *sj = sd + DT * d;
    for (int e = tid; e < DT * d; e += 256) {
        const int r = e / d;
        si[e] = (ti * DT + r) < n ? pos[(size_t)ti * DT * d + e] : 0.0;
        sj[e] = (tj * DT + r) < n ? pos[(size_t)tj * DT * d + e] : 0.0;
    }
    __syncthreads();
    double sum = 0.0;
    for (int q = tid; q < DT * DT; q += 256) {
        const int a = q / DT, c = q - a * DT;
        const int gi = ti * DT + a, gj = tj * DT + c;
        if (gi < gj && gj < n) {
            double acc = 0.0;
            for (int k = 0; k < d; ++k) {
                const double df = si[a * d + k] - sj[c * d + k];
                acc += df * df;
            }
            sum += sqrt(acc);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((tid & 63) == 0) s_sum[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) partial[block] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
}

__global__ __launch_bounds__(256) void pair_dist_kernel(const double *__restrict__ pos, int n, int d, int DT, int tiles,
                                                        const double *ctrl, int iteration, double *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) double sd[];  // [2][DT][d]
    __shared__ double s_sum[4];
    if (!running(ctrl, iteration)) return;
    pair_dist_tile(pos, n, d, DT, tiles, blockIdx.x, partial, sd, s_sum);
}

// blockIdx.y = swarm of the batch, blockIdx.x = its tile (surplus blocks of smaller swarms exit)
__global__ __launch_bounds__(256) void pair_dist_multi_kernel(SwarmBatch b, int iteration) {
    extern __shared__ __attribute__((aligned(16))) double sd[];
    __shared__ double s_sum[4];
    const int k = blockIdx.y;
    const gdpso_swarm &s = b.s[k];
    const int tiles = b.dist_tiles[k];
    if (!s.mean_dist || s.n < 2 || (int)blockIdx.x >= tiles * (tiles + 1) / 2 || !running_multi(s, iteration)) return;
    pair_dist_tile(s.pos, s.n, s.d, b.dist_tile[k], tiles, blockIdx.x, b.dist_partial[k], sd, s_sum);
}

__device__ __forceinline__ void pair_dist_finish(const double *__restrict__ partial, int n_partial, int n, int iteration,
                                                 double *__restrict__ out, double *s_sum) {
    const int tid = threadIdx.x;
    double sum = 0.0;
    for (int i = tid; i < n_partial; i += 256) sum += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((tid & 63) == 0) s_sum[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) out[iteration - 1] = ((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3])) / (0.5 * (double)n * (double)(n - 1));
}

__global__ __launch_bounds__(256) void pair_dist_finish_kernel(const double *__restrict__ partial, int n_partial, int n,
                                                               const double *ctrl, int iteration, double *__restrict__ out) {
    __shared__ double s_sum[4];
    if (!running(ctrl, iteration)) return;
    pair_dist_finish(partial, n_partial, n, iteration, out, s_sum);
}

__global__ __launch_bounds__(256) void pair_dist_finish_multi_kernel(SwarmBatch b, int iteration) {
    __shared__ double s_sum[4];
    const int k = blockIdx.x;
    const gdpso_swarm &s = b.s[k];
    if (!s.mean_dist || s.n < 2 || !running_multi(s, iteration)) return;
    const int tiles = b.dist_tiles[k];
    pair_dist_finish(b.dist_partial[k], tiles * (tiles + 1) / 2, s.n, iteration, s.mean_dist, s_sum);
}

static int check_swarm(const gdpso_swarm *s) {
    GDPSO_REQUIRE(s, "swarm: null descriptor");
    GDPSO_REQUIRE(s->n > 0 && s->d > 0 && s->n_ranks >= 1 && s->max_iter >= 1, "swarm: n, d, n_ranks, max_iter must be positive");
    GDPSO_REQUIRE(s->pos && s->vel && s->pbest_pos && s->pbest_val && s->fitness && s->gbest_pos && s->ctrl && s->g_best_val &&
                      s->g_best_val_dummy && s->cand && s->scratch,
                  "swarm: null state buffer");
    return GDPSO_OK;
}

static size_t dist_tiles(int n, int dt) {
    const size_t t = (size_t)ceil_div(n, dt);
    return t * (t + 1) / 2;
}

}  // namespace gdpso

using namespace gdpso;

extern "C" size_t gdpso_swarm_scratch_bytes(int n) {
    if (n <= 0) return 0;
    const size_t partials = (size_t)ceil_div(n, PB) * sizeof(BestRec);
    // distance partials sized for the smallest tile side, whatever dim_space turns out to be
    return (partials + 255) / 256 * 256 + dist_tiles(n, DT_MIN) * sizeof(double) + 256;
}

extern "C" int gdpso_swarm_reset(const gdpso_swarm *s, void *stream, double w_inertia) {
    int rc = check_swarm(s);
    if (rc) return rc;
    const long m = std::max<long>(std::max<long>((long)s->n * s->d, s->max_iter + 1), GDPSO_CTRL_DOUBLES);
    hipLaunchKernelGGL(reset_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, (hipStream_t)stream, *s, w_inertia);
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

extern "C" int gdpso_swarm_local_best(const gdpso_swarm *s, void *stream, int iteration, int rank_slot) {
    int rc = check_swarm(s);
    if (rc) return rc;
    GDPSO_REQUIRE(iteration >= 1 && iteration <= s->max_iter, "swarm_local_best: iteration out of range");
    GDPSO_REQUIRE(rank_slot >= 0 && rank_slot < s->n_ranks, "swarm_local_best: rank_slot out of range");
    GDPSO_REQUIRE(gdpso_swarm_scratch_bytes(s->n) <= s->scratch_bytes, "swarm_local_best: scratch too small");
    if (s->n <= SMALL_MAX_N && (long)s->n * s->d <= SMALL_MAX_ELEMS) {  // one workgroup does both steps
        hipLaunchKernelGGL(swarm_local_small_kernel, dim3(1), dim3(SMALL_THREADS), 0, (hipStream_t)stream, *s, iteration, rank_slot);
        GDPSO_HIP(hipGetLastError());
        return GDPSO_OK;
    }
    BestRec *partial = (BestRec *)s->scratch;
    const int nb = ceil_div(s->n, PB);
    hipLaunchKernelGGL(pbest_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *s, iteration, partial);
    hipLaunchKernelGGL(candidate_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *s, iteration, partial, nb, rank_slot);
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

extern "C" int gdpso_swarm_global_best(const gdpso_swarm *s, void *stream, int iteration, int schedule_inertia,
                                       int early_stopping, double tolerance) {
    int rc = check_swarm(s);
    if (rc) return rc;
    GDPSO_REQUIRE(iteration >= 1 && iteration <= s->max_iter, "swarm_global_best: iteration out of range");
    hipLaunchKernelGGL(global_best_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *s, iteration, schedule_inertia,
                       early_stopping, tolerance);
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

extern "C" int gdpso_swarm_move(const gdpso_swarm *s, void *stream, int iteration, double w_cogn, double w_soci,
                                const double *rnd, uint64_t seed, int f32_first_move) {
    int rc = check_swarm(s);
    if (rc) return rc;
    GDPSO_REQUIRE(iteration >= 1 && iteration <= s->max_iter, "swarm_move: iteration out of range");
    const long nd = (long)s->n * s->d;
    hipLaunchKernelGGL(move_kernel, dim3(ceil_div(nd, 256)), dim3(256), 0, (hipStream_t)stream, *s, iteration, w_cogn, w_soci,
                       rnd, seed, (f32_first_move && iteration == 1) ? 1 : 0);
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

extern "C" int gdpso_swarm_mean_distance(const gdpso_swarm *s, void *stream, int iteration, const double *pos_all, int n_all) {
    int rc = check_swarm(s);
    if (rc) return rc;
    GDPSO_REQUIRE(s->mean_dist && pos_all, "swarm_mean_distance: null buffer");
    GDPSO_REQUIRE(iteration >= 1 && iteration <= s->max_iter, "swarm_mean_distance: iteration out of range");
    GDPSO_REQUIRE(n_all >= 2, "swarm_mean_distance: needs at least two particles");
    const int DT = dist_tile_side(s->d, n_all);
    GDPSO_REQUIRE((size_t)2 * DT * s->d * sizeof(double) <= 64 * 1024, "swarm_mean_distance: dim_space too large");
    GDPSO_REQUIRE(gdpso_swarm_scratch_bytes(std::max(n_all, s->n)) <= s->scratch_bytes, "swarm_mean_distance: scratch too small");
    const int tiles = ceil_div(n_all, DT);
    const int nb = (int)dist_tiles(n_all, DT);
    double *partial = (double *)((char *)s->scratch + ((size_t)ceil_div(s->n, PB) * sizeof(BestRec) + 255) / 256 * 256);
    hipLaunchKernelGGL(pair_dist_kernel, dim3(nb), dim3(256), (size_t)2 * DT * s->d * sizeof(double), (hipStream_t)stream, pos_all,
                       n_all, s->d, DT, tiles, s->ctrl, iteration, partial);
    hipLaunchKernelGGL(pair_dist_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nb, n_all, s->ctrl, iteration,
                       s->mean_dist);
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

extern "C" int gdpso_swarm_iterate(const gdpso_swarm *s, const gdpso_generator *g, const gdpso_assessor *a, const gdpso_run *r,
                                   void *stream, int first_iteration, int n_iterations) {
    int rc = check_swarm(s);
    if (rc) return rc;
    GDPSO_REQUIRE(g && a && r, "swarm_iterate: null argument");
    GDPSO_REQUIRE(s->n_ranks == 1, "swarm_iterate: single-process loop; with several ranks call the phases separately");
    GDPSO_REQUIRE(first_iteration >= 1 && n_iterations >= 0 && first_iteration + n_iterations - 1 <= s->max_iter,
                  "swarm_iterate: iteration range exceeds max_iter");
    GDPSO_REQUIRE(r->img01 && r->workspace, "swarm_iterate: null scratch buffer");
    GDPSO_REQUIRE(!r->target || (r->mse && r->img), "swarm_iterate: the inverter variant needs img and mse scratch");
    for (int i = first_iteration; i < first_iteration + n_iterations; ++i) {
        if ((rc = gdpso_generator_forward(g, stream, s->n, s->pos, 1, r->img, r->img01, r->target, r->target ? r->mse : nullptr,
                                          r->workspace, r->workspace_bytes)))
            return rc;
        if ((rc = gdpso_assessor_forward(a, stream, s->n, r->img01, nullptr, s->fitness, r->class_index, r->mode, r->threshold,
                                         r->eps, r->target ? r->mse : nullptr, r->w_ass, r->w_rec, r->workspace,
                                         r->workspace_bytes)))
            return rc;
        const double *rnd = r->rand ? r->rand + (size_t)(i - 1) * s->n * 2 : nullptr;
        if (s->n <= SMALL_MAX_N && (long)s->n * s->d <= SMALL_MAX_ELEMS) {
            hipLaunchKernelGGL(swarm_step_small_kernel, dim3(1), dim3(SMALL_THREADS), 0, (hipStream_t)stream, *s, i,
                               r->schedule_inertia, r->early_stopping, r->tolerance, r->w_cogn, r->w_soci, rnd, r->seed,
                               (r->f32_first_move && i == 1) ? 1 : 0);
            GDPSO_HIP(hipGetLastError());
        } else {
            if ((rc = gdpso_swarm_local_best(s, stream, i, 0))) return rc;
            if ((rc = gdpso_swarm_global_best(s, stream, i, r->schedule_inertia, r->early_stopping, r->tolerance))) return rc;
            if ((rc = gdpso_swarm_move(s, stream, i, r->w_cogn, r->w_soci, rnd, r->seed, r->f32_first_move))) return rc;
        }
        if (r->diagnostics && s->mean_dist && s->n >= 2)
            if ((rc = gdpso_swarm_mean_distance(s, stream, i, s->pos, s->n))) return rc;
    }
    return GDPSO_OK;
}

// Several independent swarms through ONE generator + assessor batch per iteration.
extern "C" int gdpso_swarm_iterate_multi(const gdpso_swarm *swarms, const gdpso_run *runs, int n_swarms, const gdpso_generator *g,
                                         const gdpso_assessor *a, const int *class_per_image, void *stream, int first_iteration,
                                         int n_iterations) {
    GDPSO_REQUIRE(swarms && runs && g && a && n_swarms >= 1, "swarm_iterate_multi: null argument");
    GDPSO_REQUIRE(first_iteration >= 1 && n_iterations >= 0, "swarm_iterate_multi: bad iteration range");
    long total = 0;
    int rc;
    for (int k = 0; k < n_swarms; ++k) {
        const gdpso_swarm &s = swarms[k];
        if ((rc = check_swarm(&s))) return rc;
        GDPSO_REQUIRE(s.n_ranks == 1, "swarm_iterate_multi: single-process loop");
        GDPSO_REQUIRE(s.d == swarms[0].d, "swarm_iterate_multi: every swarm needs the same dim_space");
        // one evaluation batch: positions and fitness values of the swarms lie back to back
        GDPSO_REQUIRE(s.pos == swarms[0].pos + total * s.d && s.fitness == swarms[0].fitness + total,
                      "swarm_iterate_multi: pos / fitness of the swarms must be consecutive slices of one buffer");
        GDPSO_REQUIRE(!runs[k].target && runs[k].mode == runs[0].mode && runs[k].threshold == runs[0].threshold &&
                          runs[k].eps == runs[0].eps,
                      "swarm_iterate_multi: the swarms share mode, threshold and eps; the inverter variant is not batched");
        GDPSO_REQUIRE(gdpso_swarm_scratch_bytes(s.n) <= s.scratch_bytes, "swarm_iterate_multi: scratch too small");
        total += s.n;
    }
    GDPSO_REQUIRE(total < (1L << 31), "swarm_iterate_multi: too many particles");
    const gdpso_run &r0 = runs[0];
    GDPSO_REQUIRE(r0.img01 && r0.workspace, "swarm_iterate_multi: runs[0] carries the shared image and workspace buffers");
    GDPSO_REQUIRE(n_swarms == 1 || class_per_image || gdpso_assessor_n_class(a) == 2,
                  "swarm_iterate_multi: a per-image class table is needed when the swarms are after different classes");
    for (int i = first_iteration; i < first_iteration + n_iterations; ++i) {
        bool any = false;
        for (int k = 0; k < n_swarms; ++k) any |= i <= swarms[k].max_iter;
        if (!any) break;
        if ((rc = gdpso_generator_forward(g, stream, (int)total, swarms[0].pos, 1, r0.img, r0.img01, nullptr, nullptr, r0.workspace,
                                          r0.workspace_bytes)))
            return rc;
        if (class_per_image)
            rc = gdpso_assessor_forward_classes(a, stream, (int)total, r0.img01, nullptr, swarms[0].fitness, class_per_image, r0.mode,
                                                r0.threshold, r0.eps, nullptr, r0.w_ass, r0.w_rec, r0.workspace, r0.workspace_bytes);
        else
            rc = gdpso_assessor_forward(a, stream, (int)total, r0.img01, nullptr, swarms[0].fitness, r0.class_index, r0.mode,
                                        r0.threshold, r0.eps, nullptr, r0.w_ass, r0.w_rec, r0.workspace, r0.workspace_bytes);
        if (rc) return rc;
        for (int k0 = 0; k0 < n_swarms; k0 += MULTI_MAX) {
            const int cnt = std::min(MULTI_MAX, n_swarms - k0);
            SwarmBatch b;
            std::memset((void *)&b, 0, sizeof(b));
            bool small = true, dist = false;
            int max_blocks = 1;
            size_t lds = 0;
            for (int j = 0; j < cnt; ++j) {
                const gdpso_swarm &s = swarms[k0 + j];
                const gdpso_run &r = runs[k0 + j];
                b.s[j] = s;
                b.rnd[j] = (r.rand && i <= s.max_iter) ? r.rand + (size_t)(i - 1) * s.n * 2 : nullptr;
                b.seed[j] = r.seed;
                b.w_cogn[j] = r.w_cogn;
                b.w_soci[j] = r.w_soci;
                b.tolerance[j] = r.tolerance;
                b.schedule_inertia[j] = r.schedule_inertia;
                b.early_stopping[j] = r.early_stopping;
                b.f32_terms[j] = 0;
                small &= s.n <= SMALL_MAX_N && (long)s.n * s.d <= SMALL_MAX_ELEMS;
                if (r.diagnostics && s.mean_dist && s.n >= 2) {
                    const int DT = dist_tile_side(s.d, s.n);
                    GDPSO_REQUIRE((size_t)2 * DT * s.d * sizeof(double) <= 64 * 1024, "swarm_iterate_multi: dim_space too large");
                    b.dist_tile[j] = DT;
                    b.dist_tiles[j] = ceil_div(s.n, DT);
                    b.dist_partial[j] = (double *)((char *)s.scratch + ((size_t)ceil_div(s.n, PB) * sizeof(BestRec) + 255) / 256 * 256);
                    max_blocks = std::max(max_blocks, (int)dist_tiles(s.n, DT));
                    lds = std::max(lds, (size_t)2 * DT * s.d * sizeof(double));
                    dist = true;
                } else {
                    b.s[j].mean_dist = nullptr;  // the distance kernels skip this swarm
                }
            }
            if (small) {
                hipLaunchKernelGGL(swarm_step_small_multi_kernel, dim3(cnt), dim3(SMALL_THREADS), 0, (hipStream_t)stream, b, i);
            } else {  // large swarms: the multi-workgroup kernels, swarm by swarm
                for (int j = 0; j < cnt; ++j) {
                    const gdpso_swarm &s = swarms[k0 + j];
                    const gdpso_run &r = runs[k0 + j];
                    if (i > s.max_iter) continue;
                    if ((rc = gdpso_swarm_local_best(&s, stream, i, 0))) return rc;
                    if ((rc = gdpso_swarm_global_best(&s, stream, i, r.schedule_inertia, r.early_stopping, r.tolerance))) return rc;
                    if ((rc = gdpso_swarm_move(&s, stream, i, r.w_cogn, r.w_soci, b.rnd[j], r.seed, 0))) return rc;
                }
            }
            if (dist) {
                hipLaunchKernelGGL(pair_dist_multi_kernel, dim3(max_blocks, cnt), dim3(256), lds, (hipStream_t)stream, b, i);
                hipLaunchKernelGGL(pair_dist_finish_multi_kernel, dim3(cnt), dim3(256), 0, (hipStream_t)stream, b, i);
            }
            GDPSO_HIP(hipGetLastError());
        }
    }
    return GDPSO_OK;
}
