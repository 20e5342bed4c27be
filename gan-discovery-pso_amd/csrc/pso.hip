// Swarm arithmetic of reference src/pso/util_pso.py:24-193 on structure-of-arrays state.
//
// Positions, velocities and personal-best positions are float64 like the reference's
// numpy arrays; fitness values are float32 like its torch results.  This file is built
// with -ffp-contract=off and every float64 expression below is written in the
// reference's evaluation order, so positions and velocities are bit-identical to numpy's
// for identical fitness decisions.  These kernels are HBM/latency bound (a few hundred
// bytes per particle): elements are mapped to consecutive lanes for coalesced access.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace gdpso {

// Particles per workgroup in the personal-best kernel: PB at least (what the scratch buffer's layout is
// sized by), up to PB_MAX once that still leaves eight workgroups per compute unit (pbest_particles).  Few for
// small swarms, because a workgroup also copies the improved particles' positions (d doubles each): at 512
// particles x 100 dimensions two workgroups of 256 particles took 16.7 us, sixteen of 32 take 4.
constexpr int PB = 32, PB_MAX = 256;
static int pbest_particles(int n) {
    int pb = PB_MAX;
    while (pb > PB && (n + pb - 1) / pb < 2048) pb >>= 1;
    return pb;
}

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

// The distance part of a swarm's scratch buffer (the tile sums): behind the personal-best partials, 256-byte aligned.
__host__ __device__ __forceinline__ double *dist_partials(const gdpso_swarm &s) {
    return (double *)((char *)s.scratch + ((size_t)((s.n + PB - 1) / PB) * sizeof(BestRec) + 255) / 256 * 256);
}

// Record r of the candidate table, and where in a record the distance sum of an iteration lies
// (two slots, by the iteration's parity: the sums of iteration i are written while those of i - 1 are
// still being exchanged).
__host__ __device__ __forceinline__ double *cand_row(const gdpso_swarm &s, int r) {
    return s.cand + (size_t)r * (s.cand_stride > 0 ? s.cand_stride : GDPSO_CAND_DOUBLES(s.d));
}
__host__ __device__ __forceinline__ int dist_slot(int d, int iteration) { return d + 2 + (iteration & 1); }

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
// (util_pso.py:122-124) for pb particles per workgroup, then the workgroup's min-loc over
// the updated personal bests -> partial[blockIdx.x].
__global__ __launch_bounds__(256) void pbest_kernel(gdpso_swarm s, int iteration, BestRec *__restrict__ partial, int pb) {
    __shared__ unsigned char s_better[PB_MAX];
    __shared__ BestRec s_best[4];
    if (!running(s.ctrl, iteration)) return;
    const int tid = threadIdx.x, p0 = blockIdx.x * pb, p = p0 + tid;
    BestRec r;
    r.val = INFINITY;
    r.idx = 0x7fffffff;
    bool better = false;
    if (tid < pb && p < s.n) {
        const float f = s.fitness[p];
        float best = s.pbest_val[p];
        better = f < best;
        if (better) {
            best = f;
            s.pbest_val[p] = f;
        }
        r.val = best;
        r.idx = s.first_index + p;
    }
    if (tid < pb) s_better[tid] = better;
    r = wave_best(r);
    if ((tid & 63) == 0) s_best[tid >> 6] = r;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = better_of(better_of(s_best[0], s_best[1]), better_of(s_best[2], s_best[3]));
    const int cnt = min(pb, s.n - p0);
    const long base = (long)p0 * s.d;
    for (int e = tid; e < cnt * s.d; e += 256)
        if (s_better[e / s.d]) s.pbest_pos[base + e] = s.pos[base + e];
}

// Reduces the per-workgroup partials to this rank's candidate record
// {value, global index, personal-best position} (every thread of a 256-thread workgroup calls it).
__device__ __forceinline__ void write_candidate(const gdpso_swarm &s, const BestRec *__restrict__ partial, int n_partial, int rank_slot,
                                                BestRec *s_best) {
    const int tid = threadIdx.x;
    BestRec r;
    r.val = INFINITY;
    r.idx = 0x7fffffff;
    for (int i = tid; i < n_partial; i += 256) r = better_of(r, partial[i]);
    r = wave_best(r);
    if ((tid & 63) == 0) s_best[tid >> 6] = r;
    __syncthreads();
    r = better_of(better_of(s_best[0], s_best[1]), better_of(s_best[2], s_best[3]));
    double *rec = cand_row(s, rank_slot);
    const bool have = r.idx != 0x7fffffff && r.val < INFINITY;
    if (tid == 0) {
        rec[0] = have ? (double)r.val : (double)INFINITY;
        rec[1] = have ? (double)r.idx : -1.0;
    }
    for (int k = tid; k < s.d; k += 256) rec[2 + k] = have ? s.pbest_pos[(size_t)(r.idx - s.first_index) * s.d + k] : 0.0;
}

__global__ __launch_bounds__(256) void candidate_kernel(gdpso_swarm s, int iteration, const BestRec *__restrict__ partial,
                                                        int n_partial, int rank_slot) {
    __shared__ BestRec s_best[4];
    if (!running(s.ctrl, iteration)) return;
    write_candidate(s, partial, n_partial, rank_slot, s_best);
}

// update_global_best (util_pso.py:135-151) over the candidate records of all ranks, the
// inertia decay (:178-179) and the early-stopping test (:186-188).  One workgroup; every
// rank runs it on identical inputs and reaches identical decisions.
// One thread: the decision part of update_global_best.  Returns the winning slot or -1.
// Bookkeeping of a decision: `best` is the global best value after it, best_slot >= 0 when a candidate took over
// (its global index best_idx), cur / n_imp / w the control block's values before it.  Returns the inertia weight
// the move of this iteration uses.
__device__ __forceinline__ double record_global_best(const gdpso_swarm &s, int iteration, int schedule_inertia, int early_stopping,
                                                     double tolerance, float cur, float best, int best_slot, int best_idx, int n_imp,
                                                     double w) {
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
    if (iteration > 1 && schedule_inertia) {
        w = 0.99 * w;
        s.ctrl[GDPSO_CTRL_W] = w;
    }
    if (iteration > 2 && n_imp > 2 && early_stopping) {
        const float diff = fabsf(s.g_best_val[n_imp - 1] - s.g_best_val[n_imp - 2]);
        if (diff < (float)tolerance) s.ctrl[GDPSO_CTRL_STOP_ITER] = (double)iteration;
    }
    s.ctrl[GDPSO_CTRL_ITERS_DONE] = (double)iteration;
    return w;
}

__device__ int decide_global_best(const gdpso_swarm &s, int iteration, int schedule_inertia, int early_stopping,
                                  double tolerance) {
    float cur = (float)s.ctrl[GDPSO_CTRL_GBEST_VAL];
    float best = cur;
    int best_slot = -1, best_idx = 0x7fffffff;
    for (int k = 0; k < s.n_ranks; ++k) {
        const double *rec = cand_row(s, k);
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
    record_global_best(s, iteration, schedule_inertia, early_stopping, tolerance, cur, best, best_slot, best_idx,
                       (int)s.ctrl[GDPSO_CTRL_N_IMPROVE], s.ctrl[GDPSO_CTRL_W]);
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
        const double *rec = cand_row(s, slot);
        for (int k = tid; k < s.d; k += 64) s.gbest_pos[k] = rec[2 + k];
    }
}

// Single process: the candidate record and the global-best update it feeds in one launch.
__global__ __launch_bounds__(256) void candidate_global_kernel(gdpso_swarm s, int iteration, const BestRec *__restrict__ partial,
                                                               int n_partial, int schedule_inertia, int early_stopping,
                                                               double tolerance) {
    __shared__ BestRec s_best[4];
    __shared__ int s_take;
    if (!running(s.ctrl, iteration)) return;
    write_candidate(s, partial, n_partial, 0, s_best);
    __threadfence_block();
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid == 0) s_take = decide_global_best(s, iteration, schedule_inertia, early_stopping, tolerance);
    __syncthreads();
    if (s_take >= 0) {
        const double *rec = cand_row(s, s_take);
        for (int k = tid; k < s.d; k += 256) s.gbest_pos[k] = rec[2 + k];
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
// The draws of particle p at `iteration`: the host's (rnd) or Philox's.
__device__ __forceinline__ void move_draws(const gdpso_swarm &s, int iteration, int p, const double *__restrict__ rnd, uint64_t seed,
                                           double &r1, double &r2) {
    if (rnd) {
        r1 = rnd[2 * (size_t)p];
        r2 = rnd[2 * (size_t)p + 1];
    } else {
        uint32_t o[4];
        philox4x32_10((uint32_t)(s.first_index + p), 0u, (uint32_t)iteration, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        r1 = u53(o[0], o[1]);
        r2 = u53(o[2], o[3]);
    }
}

// One element's new velocity; the new position is x + it.
__device__ __forceinline__ double move_velocity(double x, double v, double pbest, double gbest, double r1, double r2, double w,
                                                double w_cogn, double w_soci, int f32_terms) {
    const double inertia = w * v;
    const double c1 = w_cogn * r1;
    const double c2 = w_soci * r2;
    double best_cogn, best_soci;
    if (f32_terms) {
        // float32 positions (first move of the inverter variant): numpy keeps float32 for
        // array - array and for python-float * float32-array
        const float xf = (float)x;
        best_cogn = (double)((float)c1 * ((float)gbest - xf));
        best_soci = (double)((float)c2 * ((float)pbest - xf));
    } else {
        best_cogn = c1 * (gbest - x);
        best_soci = c2 * (pbest - x);
    }
    return (inertia + best_soci) + best_cogn;
}

__device__ __forceinline__ void move_element(const gdpso_swarm &s, int iteration, long e, double w, double w_cogn,
                                             double w_soci, const double *__restrict__ rnd, uint64_t seed, int f32_terms) {
    const int p = (int)(e / s.d), k = (int)(e - (long)p * s.d);
    double r1, r2;
    move_draws(s, iteration, p, rnd, seed, r1, r2);
    const double x = s.pos[e];
    const double nv = move_velocity(x, s.vel[e], s.pbest_pos[e], s.gbest_pos[k], r1, r2, w, w_cogn, w_soci, f32_terms);
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
    double *rec = cand_row(s, rank_slot);
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

// The same step for a single process and at most FAST_ELEMS state elements per thread, arranged so that it has ONE
// round trip to memory on its critical path instead of six: everything the step reads is requested at entry (fitness
// and personal-best values, the control block, the old global best, and the position / velocity / personal-best
// elements the thread will move), the candidate record, the decision and the new global best pass through LDS (they
// are still written to memory for the host), and the move works on the registers.  Same arithmetic, same results.
constexpr int FAST_ELEMS = 8, FAST_MAX_D = 1024;
static bool g_small_fast = [] {
    const char *e = getenv("GDPSO_SWARM_FAST_STEP");
    return e ? e[0] != '0' : true;
}();
static bool small_step_fast_ok(const gdpso_swarm &s) {
    return g_small_fast && s.n_ranks == 1 && (long)s.n * s.d <= (long)FAST_ELEMS * SMALL_THREADS && s.d <= FAST_MAX_D;
}

__device__ __forceinline__ void small_step_fast(const gdpso_swarm &s, int iteration, int schedule_inertia, int early_stopping,
                                                double tolerance, double w_cogn, double w_soci, const double *__restrict__ rnd,
                                                uint64_t seed, int f32_terms, unsigned char *s_better, BestRec *s_best, bool in_budget) {
    __shared__ double s_gbest[FAST_MAX_D];
    __shared__ double s_w;
    __shared__ int s_take, s_best_p;
    const int tid = threadIdx.x, n = s.n, d = s.d;
    const long nd = (long)n * d;
    // ---- everything the step reads, requested at once
    const double stop_at = s.ctrl[GDPSO_CTRL_STOP_ITER];
    double c_gbest = 0.0, c_nimp = 0.0, c_w = 0.0;
    if (tid == 0) {
        c_gbest = s.ctrl[GDPSO_CTRL_GBEST_VAL];
        c_nimp = s.ctrl[GDPSO_CTRL_N_IMPROVE];
        c_w = s.ctrl[GDPSO_CTRL_W];
    }
    float fit[SMALL_MAX_N / SMALL_THREADS], pbv[SMALL_MAX_N / SMALL_THREADS];
#pragma unroll
    for (int i = 0; i < SMALL_MAX_N / SMALL_THREADS; ++i) {
        const int p = tid + i * SMALL_THREADS;
        fit[i] = p < n ? s.fitness[p] : 0.f;
        pbv[i] = p < n ? s.pbest_val[p] : 0.f;
    }
    double x[FAST_ELEMS], v[FAST_ELEMS], pb[FAST_ELEMS];
#pragma unroll
    for (int i = 0; i < FAST_ELEMS; ++i) {
        const long e = tid + (long)i * SMALL_THREADS;
        x[i] = e < nd ? s.pos[e] : 0.0;
        v[i] = e < nd ? s.vel[e] : 0.0;
        pb[i] = e < nd ? s.pbest_pos[e] : 0.0;
    }
    for (int k = tid; k < d; k += SMALL_THREADS) s_gbest[k] = s.gbest_pos[k];
    const int stop = (int)stop_at;
    if (!in_budget || !(stop == 0 || iteration <= stop)) return;  // (uniform over the workgroup)
    // ---- update_personal_best
    BestRec r;
    r.val = INFINITY;
    r.idx = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < SMALL_MAX_N / SMALL_THREADS; ++i) {
        const int p = tid + i * SMALL_THREADS;
        if (p < n) {
            const bool better = fit[i] < pbv[i];
            if (better) {
                pbv[i] = fit[i];
                s.pbest_val[p] = fit[i];
            }
            s_better[p] = better;
            BestRec c;
            c.val = pbv[i];
            c.idx = s.first_index + p;
            r = better_of(r, c);
        }
    }
    r = wave_best(r);
    if ((tid & 63) == 0) s_best[tid >> 6] = r;
    __syncthreads();
    r = s_best[0];
#pragma unroll
    for (int w = 1; w < SMALL_THREADS / 64; ++w) r = better_of(r, s_best[w]);
    const bool have = r.idx != 0x7fffffff && r.val < INFINITY;
    // ---- personal-best positions, and the candidate = the best particle's (updated) personal best
    if (tid == 0) {
        // update_global_best, inertia decay, early-stopping test: one candidate, this swarm's
        const float cur = (float)c_gbest;
        const bool take = have && r.val < cur;
        s_take = take ? 1 : 0;
        s_best_p = have ? r.idx - s.first_index : -1;
        double *rec = cand_row(s, 0);
        rec[0] = have ? (double)r.val : (double)INFINITY;
        rec[1] = have ? (double)r.idx : -1.0;
        s_w = record_global_best(s, iteration, schedule_inertia, early_stopping, tolerance, cur, take ? r.val : cur, take ? 0 : -1,
                                 take ? r.idx : 0x7fffffff, (int)c_nimp, c_w);
    }
    __syncthreads();
    const int best_p = s_best_p;
    const bool take = s_take != 0;
    double *rec = cand_row(s, 0);
#pragma unroll
    for (int i = 0; i < FAST_ELEMS; ++i) {
        const long e = tid + (long)i * SMALL_THREADS;
        if (e < nd) {
            const int p = (int)(e / d), k = (int)(e - (long)p * d);
            if (s_better[p]) {
                pb[i] = x[i];
                s.pbest_pos[e] = x[i];
            }
            if (p == best_p) {
                rec[2 + k] = pb[i];
                if (take) {
                    s_gbest[k] = pb[i];
                    s.gbest_pos[k] = pb[i];
                }
            }
        }
    }
    if (best_p < 0)
        for (int k = tid; k < d; k += SMALL_THREADS) rec[2 + k] = 0.0;
    __syncthreads();
    // ---- move_particles
    const double w = s_w;
#pragma unroll
    for (int i = 0; i < FAST_ELEMS; ++i) {
        const long e = tid + (long)i * SMALL_THREADS;
        if (e < nd) {
            const int p = (int)(e / d), k = (int)(e - (long)p * d);
            double r1, r2;
            move_draws(s, iteration, p, rnd, seed, r1, r2);
            const double nv = move_velocity(x[i], v[i], pb[i], s_gbest[k], r1, r2, w, w_cogn, w_soci, f32_terms);
            const double nx = x[i] + nv;
            s.vel[e] = nv;
            s.pos[e] = nx;
            const size_t h = (size_t)iteration * n * d + e;
            if (s.hist_pos) s.hist_pos[h] = nx;
            if (s.hist_vel) s.hist_vel[h] = nv;
        }
    }
}

template <bool FAST>
__global__ __launch_bounds__(SMALL_THREADS) void swarm_step_small_kernel(gdpso_swarm s, int iteration, int schedule_inertia,
                                                                         int early_stopping, double tolerance, double w_cogn,
                                                                         double w_soci, const double *__restrict__ rnd,
                                                                         uint64_t seed, int f32_terms) {
    __shared__ unsigned char s_better[SMALL_MAX_N];
    __shared__ BestRec s_best[SMALL_THREADS / 64];
    __shared__ int s_take;
    if constexpr (FAST) {
        small_step_fast(s, iteration, schedule_inertia, early_stopping, tolerance, w_cogn, w_soci, rnd, seed, f32_terms, s_better, s_best, true);
        return;
    }
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
    double *dist_partial[MULTI_MAX];  // this iteration's tile sums ...
    double *dist_prev[MULTI_MAX];     // ... and, when the launch also finishes the previous iteration's mean, that one's (else null)
};
static_assert(sizeof(SwarmBatch) <= 3072, "kernel arguments are limited to 4 KiB");

// a swarm takes part in `iteration` while it has not stopped and the iteration is within its own budget
__device__ __forceinline__ bool running_multi(const gdpso_swarm &s, int iteration) {
    return iteration <= s.max_iter && running(s.ctrl, iteration);
}

template <bool FAST>
__global__ __launch_bounds__(SMALL_THREADS) void swarm_step_small_multi_kernel(SwarmBatch b, int iteration) {
    __shared__ unsigned char s_better[SMALL_MAX_N];
    __shared__ BestRec s_best[SMALL_THREADS / 64];
    __shared__ int s_take;
    const int k = blockIdx.x;
    const gdpso_swarm &s = b.s[k];
    if constexpr (FAST) {
        small_step_fast(s, iteration, b.schedule_inertia[k], b.early_stopping[k], b.tolerance[k], b.w_cogn[k], b.w_soci[k], b.rnd[k],
                        b.seed[k], b.f32_terms[k], s_better, s_best, iteration <= s.max_iter);
        return;
    }
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

// Swarm.mse (util_pso.py:76-86): mean Euclidean distance over unordered pairs, float64 throughout.
// A workgroup of 16 x 16 threads takes a tile of TS x TS pairs (upper triangle of tiles only) and
// walks the coordinates in chunks of DIST_KC through LDS; a thread holds (TS/16)^2 running sums
// of squares in registers, so every coordinate read from LDS feeds TS/16 pairs (one pair per two LDS
// reads made the kernel LDS-bound: 512 us at 4096 particles x 100 coordinates, 43 us this way).
// Rows are DIST_KC + 2 doubles apart: 16-byte aligned, so two coordinates come with one LDS read, and the
// 16 lanes that differ in the column read 16 different groups of four banks.  A chunk is always DIST_KC
// coordinates long (past the last one zeros, which add nothing to a sum of squares), so the loop over it
// unrolls completely.  Per-tile sums are combined in a fixed order by the finishing kernel, so the result
// is reproducible.  The tile side depends on n and the number of ranks only (it fixes the summation order).
// Several ranks: the tiles are dealt round-robin (tile b belongs to part b % n_parts), every rank
// sums its own, and the sums are added in rank order once they have been exchanged.
constexpr int DIST_KC = 32, DIST_KCP = DIST_KC + 2;
static size_t dist_lds_bytes(int ts) { return (size_t)2 * ts * DIST_KCP * sizeof(double); }
static int dist_tile_side(int n, int n_parts) {
    // 64: four pairs per LDS read pair, worth it once every part has several tiles per compute unit (a
    // tile of 64 x 64 pairs x 100 coordinates keeps a compute unit busy for ~15 us, so a launch of fewer
    // tiles than compute units takes that long however few they are); 32: the same with 1/4 the work
    // per tile; 16 (one pair per thread, LDS-bound) only for swarms too small to give 32 tiles of 32.
    auto per_part = [&](int ts) {
        const long t = (n + ts - 1) / ts;
        return t * (t + 1) / 2 / n_parts;
    };
    if (per_part(64) >= 1024) return 64;
    if (per_part(32) >= 32) return 32;
    return 16;
}

// Particle g's coordinates start at pos + (g / blk_rows) * blk_stride + (g % blk_rows) * d: blocks of blk_rows
// particles, blk_stride doubles apart (one block of n particles for positions that lie back to back).
template <int TS>
__device__ __forceinline__ void pair_dist_tile(const double *__restrict__ pos, int n, int d, int blk_rows, size_t blk_stride, int tiles,
                                                int block, double *__restrict__ out, double *sd, double *s_sum) {
    constexpr int R = TS / 16;
    // linear tile id -> (ti <= tj) upper-triangular tile
    int b = block, ti = 0;
    while (b >= tiles - ti) {
        b -= tiles - ti;
        ++ti;
    }
    const int tj = ti + b;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    double *si = sd, *sj = sd + TS * DIST_KCP;
    double acc[R][R];
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
        for (int v = 0; v < R; ++v) acc[u][v] = 0.0;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    // a chunk of both tiles travels through registers: requested a chunk ahead, stored once the
    // previous chunk has been consumed
    constexpr int EPT = TS * DIST_KC / 256;  // elements per thread and tile
    double ri[EPT], rj[EPT];
    const double *pi[EPT], *pj[EPT];  // where this thread's rows of the two tiles start (null: past the last particle)
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int r = (tid + 256 * q) / DIST_KC;
        const int gi = ti * TS + r, gj = tj * TS + r;
        pi[q] = gi < n ? pos + (size_t)(gi / blk_rows) * blk_stride + (size_t)(gi % blk_rows) * d : nullptr;
        pj[q] = gj < n ? pos + (size_t)(gj / blk_rows) * blk_stride + (size_t)(gj % blk_rows) * d : nullptr;
    }
    auto request = [&](int k0) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int k = k0 + (tid + 256 * q) % DIST_KC;
            ri[q] = k < d && pi[q] ? pi[q][k] : 0.0;
            rj[q] = k < d && pj[q] ? pj[q][k] : 0.0;
        }
    };
    request(0);
    for (int k0 = 0; k0 < d; k0 += DIST_KC) {
        if (k0) __syncthreads();  // the previous chunk has been read
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + 256 * q, r = e / DIST_KC, kk = e % DIST_KC;
            si[r * DIST_KCP + kk] = ri[q];
            sj[r * DIST_KCP + kk] = rj[q];
        }
        __syncthreads();
        if (k0 + DIST_KC < d) request(k0 + DIST_KC);
#pragma unroll
        for (int kk = 0; kk < DIST_KC; kk += 2) {
            f64x2 a[R], c[R];
#pragma unroll
            for (int u = 0; u < R; ++u) a[u] = *(const f64x2 *)&si[(ty + 16 * u) * DIST_KCP + kk];
#pragma unroll
            for (int v = 0; v < R; ++v) c[v] = *(const f64x2 *)&sj[(tx + 16 * v) * DIST_KCP + kk];
#pragma unroll
            for (int u = 0; u < R; ++u)
#pragma unroll
                for (int v = 0; v < R; ++v) {
                    const double d0 = a[u][0] - c[v][0], d1 = a[u][1] - c[v][1];
                    acc[u][v] = __builtin_fma(d0, d0, acc[u][v]);
                    acc[u][v] = __builtin_fma(d1, d1, acc[u][v]);
                }
        }
    }
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
        for (int v = 0; v < R; ++v) {
            const int gi = ti * TS + ty + 16 * u, gj = tj * TS + tx + 16 * v;
            if (gi < gj && gj < n) sum += sqrt(acc[u][v]);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((tid & 63) == 0) s_sum[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) *out = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
}

__device__ __forceinline__ void pair_dist_tile_any(int ts, const double *__restrict__ pos, int n, int d, int tiles, int block,
                                                    double *__restrict__ out, double *sd, double *s_sum) {
    if (ts == 64)
        pair_dist_tile<64>(pos, n, d, n, 0, tiles, block, out, sd, s_sum);
    else if (ts == 32)
        pair_dist_tile<32>(pos, n, d, n, 0, tiles, block, out, sd, s_sum);
    else
        pair_dist_tile<16>(pos, n, d, n, 0, tiles, block, out, sd, s_sum);
}

// workgroup q of part `part` takes tile part + q * n_parts -> partial[q].  One kernel per tile side: the
// 64 x 64 tile's 16 running sums per thread cost it 176 registers (two workgroups per compute unit),
// which the smaller tiles must not inherit.
// A launch may carry one more workgroup behind its tiles (blockIdx.x == n_blocks) that finishes the PREVIOUS iteration's
// mean from that iteration's tile sums (pair_dist_finish: the same additions in the same order as the finishing kernel).
struct DistTail {
    const double *partial;  // null: no such workgroup
    int n_partial, iteration;
    double pairs;
    double *out;
};

__device__ __forceinline__ void pair_dist_finish(const double *__restrict__ partial, int n_partial, double pairs, double *__restrict__ out,
                                                 double *s_sum);

template <int TS>
__global__ __launch_bounds__(256) void pair_dist_kernel(const double *__restrict__ pos, int n, int d, int blk_rows, size_t blk_stride,
                                                        int tiles, int part, int n_parts, const double *ctrl, int iteration,
                                                        double *__restrict__ partial, int n_blocks, DistTail tail) {
    __shared__ __attribute__((aligned(16))) double sd[2 * TS * DIST_KCP];
    __shared__ double s_sum[4];
    if ((int)blockIdx.x >= n_blocks) {
        if (running(ctrl, tail.iteration)) pair_dist_finish(tail.partial, tail.n_partial, tail.pairs, tail.out, s_sum);
        return;
    }
    if (!running(ctrl, iteration)) return;
    pair_dist_tile<TS>(pos, n, d, blk_rows, blk_stride, tiles, part + (int)blockIdx.x * n_parts, partial + blockIdx.x, sd, s_sum);
}

// blockIdx.y = swarm of the batch, blockIdx.x = its tile (surplus blocks of smaller swarms exit)
__global__ __launch_bounds__(256) void pair_dist_multi_kernel(SwarmBatch b, int iteration) {
    extern __shared__ __attribute__((aligned(16))) double sd[];  // [2][largest tile side of the batch][DIST_KCP]
    __shared__ double s_sum[4];
    const int k = blockIdx.y;
    const gdpso_swarm &s = b.s[k];
    const int tiles = b.dist_tiles[k];
    if (blockIdx.x + 1 == gridDim.x && b.dist_prev[k] != nullptr) {  // (a column of workgroups behind the tiles: the previous iteration's mean)
        if (s.mean_dist && s.n >= 2 && running_multi(s, iteration - 1))
            pair_dist_finish(b.dist_prev[k], tiles * (tiles + 1) / 2, 0.5 * (double)s.n * (double)(s.n - 1), s.mean_dist + (iteration - 2), s_sum);
        return;
    }
    if (!s.mean_dist || s.n < 2 || (int)blockIdx.x >= tiles * (tiles + 1) / 2 || !running_multi(s, iteration)) return;
    pair_dist_tile_any(b.dist_tile[k], s.pos, s.n, s.d, tiles, blockIdx.x, b.dist_partial[k] + blockIdx.x, sd, s_sum);
}

// The tile sums in a fixed order, behind the launch that made them (folded into the tile kernel, with the
// workgroup that finishes last adding up, it cost more than a launch: every workgroup then pays a
// device-scope fence): a launch of its own, or a workgroup of the NEXT iteration's tile launch (the fused loops:
// the tile sums alternate between two buffers by the iteration's parity).
// pairs = 0: the sum itself (a rank's share) -> out, else the mean over `pairs` pairs.
__device__ __forceinline__ void pair_dist_finish(const double *__restrict__ partial, int n_partial, double pairs, double *__restrict__ out,
                                                 double *s_sum) {
    const int tid = threadIdx.x;
    double sum = 0.0;
    for (int i = tid; i < n_partial; i += 256) sum += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((tid & 63) == 0) s_sum[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        const double total = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        *out = pairs == 0.0 ? total : total / pairs;
    }
}

__global__ __launch_bounds__(256) void pair_dist_finish_kernel(const double *__restrict__ partial, int n_partial, double pairs,
                                                               const double *ctrl, int iteration, double *__restrict__ out) {
    __shared__ double s_sum[4];
    if (!running(ctrl, iteration)) return;
    pair_dist_finish(partial, n_partial, pairs, out, s_sum);
}

__global__ __launch_bounds__(256) void pair_dist_finish_multi_kernel(SwarmBatch b, int iteration) {
    __shared__ double s_sum[4];
    const int k = blockIdx.x;
    const gdpso_swarm &s = b.s[k];
    if (!s.mean_dist || s.n < 2 || !running_multi(s, iteration)) return;
    const int tiles = b.dist_tiles[k];
    pair_dist_finish(b.dist_partial[k], tiles * (tiles + 1) / 2, 0.5 * (double)s.n * (double)(s.n - 1), s.mean_dist + (iteration - 1), s_sum);
}

// The ranks' sums (last slot of every candidate record), added in rank order by one thread -> the
// mean over `pairs` pairs: the same bits on every rank.
__device__ __forceinline__ void dist_mean(const gdpso_swarm &s, int iteration, double pairs) {
    if (!running(s.ctrl, iteration)) return;
    double total = 0.0;
    for (int r = 0; r < s.n_ranks; ++r) total += cand_row(s, r)[dist_slot(s.d, iteration)];
    s.mean_dist[iteration - 1] = total / pairs;
}

__global__ void dist_mean_kernel(gdpso_swarm s, int iteration, double pairs) {
    if (threadIdx.x == 0) dist_mean(s, iteration, pairs);
}

// What follows the exchange of the candidate records, for a swarm small enough for one workgroup:
// the mean of the deferred diagnostic (diag_iteration, 0: none), update_global_best and the move.
__global__ __launch_bounds__(SMALL_THREADS) void swarm_after_exchange_small_kernel(gdpso_swarm s, int iteration, int schedule_inertia,
                                                                                   int early_stopping, double tolerance, double w_cogn,
                                                                                   double w_soci, const double *__restrict__ rnd,
                                                                                   uint64_t seed, int f32_terms, int diag_iteration,
                                                                                   double pairs) {
    __shared__ int s_take;
    const int tid = threadIdx.x;
    if (tid == 0) {
        if (diag_iteration > 0) dist_mean(s, diag_iteration, pairs);
        s_take = running(s.ctrl, iteration) ? decide_global_best(s, iteration, schedule_inertia, early_stopping, tolerance) : -2;
    }
    __syncthreads();
    if (s_take == -2) return;  // stopped before this iteration
    if (s_take >= 0) {
        const double *rec = cand_row(s, s_take);
        for (int k = tid; k < s.d; k += SMALL_THREADS) s.gbest_pos[k] = rec[2 + k];
    }
    __threadfence_block();
    __syncthreads();
    const long nd = (long)s.n * s.d;
    const double w = s.ctrl[GDPSO_CTRL_W];
    for (long e = tid; e < nd; e += SMALL_THREADS) move_element(s, iteration, e, w, w_cogn, w_soci, rnd, seed, f32_terms);
}

static int check_swarm(const gdpso_swarm *s) {
    GDPSO_REQUIRE(s, "swarm: null descriptor");
    GDPSO_REQUIRE(s->n > 0 && s->d > 0 && s->n_ranks >= 1 && s->max_iter >= 1, "swarm: n, d, n_ranks, max_iter must be positive");
    GDPSO_REQUIRE(s->cand_stride == 0 || s->cand_stride >= GDPSO_CAND_DOUBLES(s->d), "swarm: cand_stride shorter than a record");
    GDPSO_REQUIRE(s->pos && s->vel && s->pbest_pos && s->pbest_val && s->fitness && s->gbest_pos && s->ctrl && s->g_best_val &&
                      s->g_best_val_dummy && s->cand && s->scratch,
                  "swarm: null state buffer");
    return GDPSO_OK;
}

static size_t dist_tiles(int n, int dt) {
    const size_t t = (size_t)ceil_div(n, dt);
    return t * (t + 1) / 2;
}
// doubles between the two sets of tile sums of a swarm's scratch (set = iteration & 1 in the fused loops, else 0)
static size_t dist_set_stride(size_t n_tiles) { return (n_tiles + 31) / 32 * 32; }

static bool g_dist_defer = [] {
    const char *e = getenv("GDPSO_DIST_DEFER");
    return e ? e[0] != '0' : true;
}();

}  // namespace gdpso

using namespace gdpso;

extern "C" size_t gdpso_swarm_scratch_bytes(int n) {
    if (n <= 0) return 0;
    const size_t partials = (size_t)ceil_div(n, PB) * sizeof(BestRec);
    // distance partials sized for the smallest tile side, twice (the fused loops alternate between two sets)
    return (partials + 255) / 256 * 256 + 2 * dist_set_stride(dist_tiles(n, 16)) * sizeof(double) + 256;
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
    const int pb = pbest_particles(s->n), nb = ceil_div(s->n, pb);
    hipLaunchKernelGGL(pbest_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *s, iteration, partial, pb);
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

// tiles of part `part` of `n_parts` -> their sum in *out (pairs == 0) or the mean over `pairs` pairs.
// set: which of the scratch's two sets of tile sums; tiles / finish: which of the two steps to enqueue; tail_iteration
// > 0 (with tiles): the tile launch carries a workgroup that finishes that iteration's mean from the OTHER set.
static int launch_distance(const gdpso_swarm *s, void *stream, int iteration, const double *pos_all, int n_all, int part, int n_parts,
                           double pairs, double *out, int blk_rows = 0, size_t blk_stride = 0, int set = 0, bool do_tiles = true,
                           bool do_finish = true, int tail_iteration = 0) {
    if (blk_rows <= 0) blk_rows = n_all;  // positions back to back
    const int ts = dist_tile_side(n_all, n_parts);
    const int tiles = ceil_div(n_all, ts);
    const int nb_all = (int)dist_tiles(n_all, ts);
    const int nb = part < nb_all ? ceil_div(nb_all - part, n_parts) : 0;  // tiles part, part + n_parts, ...
    const size_t stride = dist_set_stride((size_t)nb_all);
    double *partial = dist_partials(*s) + (size_t)set * stride;
    if (do_tiles && (nb > 0 || tail_iteration > 0)) {
        DistTail tail;
        tail.partial = tail_iteration > 0 ? dist_partials(*s) + (size_t)(1 - set) * stride : nullptr;
        tail.n_partial = nb;
        tail.iteration = tail_iteration;
        tail.pairs = pairs;
        tail.out = tail_iteration > 0 ? s->mean_dist + (tail_iteration - 1) : nullptr;
        auto kernel = ts == 64 ? pair_dist_kernel<64> : ts == 32 ? pair_dist_kernel<32> : pair_dist_kernel<16>;
        hipLaunchKernelGGL(kernel, dim3(nb + (tail_iteration > 0 ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, pos_all, n_all, s->d, blk_rows,
                           blk_stride, tiles, part, n_parts, s->ctrl, iteration, partial, nb, tail);
    }
    if (do_finish)
        hipLaunchKernelGGL(pair_dist_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nb, pairs, s->ctrl, iteration, out);
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

static int check_distance(const gdpso_swarm *s, int iteration, const double *pos_all, int n_all) {
    int rc = check_swarm(s);
    if (rc) return rc;
    GDPSO_REQUIRE(pos_all, "swarm distance: null buffer");
    GDPSO_REQUIRE(iteration >= 1 && iteration <= s->max_iter, "swarm distance: iteration out of range");
    GDPSO_REQUIRE(n_all >= 2, "swarm distance: needs at least two particles");
    GDPSO_REQUIRE(gdpso_swarm_scratch_bytes(std::max(n_all, s->n)) <= s->scratch_bytes, "swarm distance: scratch too small");
    return GDPSO_OK;
}

extern "C" int gdpso_swarm_mean_distance(const gdpso_swarm *s, void *stream, int iteration, const double *pos_all, int n_all) {
    int rc = check_distance(s, iteration, pos_all, n_all);
    if (rc) return rc;
    GDPSO_REQUIRE(s->mean_dist, "swarm_mean_distance: the swarm keeps no diagnostic");
    return launch_distance(s, stream, iteration, pos_all, n_all, 0, 1, 0.5 * (double)n_all * (double)(n_all - 1), s->mean_dist + (iteration - 1));
}

extern "C" int gdpso_swarm_distance_part(const gdpso_swarm *s, void *stream, int iteration, const double *pos_all, int n_all,
                                         int rank_slot) {
    int rc = check_distance(s, iteration, pos_all, n_all);
    if (rc) return rc;
    GDPSO_REQUIRE(rank_slot >= 0 && rank_slot < s->n_ranks, "swarm_distance_part: rank_slot out of range");
    double *slot = cand_row(*s, rank_slot) + dist_slot(s->d, iteration);
    return launch_distance(s, stream, iteration, pos_all, n_all, rank_slot, s->n_ranks, 0.0, slot);
}

extern "C" int gdpso_swarm_distance_part_table(const gdpso_swarm *s, void *stream, int iteration, int n_all, int rank_slot) {
    int rc = check_distance(s, iteration, s ? s->cand : nullptr, n_all);
    if (rc) return rc;
    GDPSO_REQUIRE(rank_slot >= 0 && rank_slot < s->n_ranks, "swarm_distance_part_table: rank_slot out of range");
    GDPSO_REQUIRE(n_all % s->n_ranks == 0, "swarm_distance_part_table: the ranks must hold equally many particles");
    const int blk_rows = n_all / s->n_ranks;
    GDPSO_REQUIRE((size_t)s->cand_stride >= GDPSO_CAND_DOUBLES(s->d) + (size_t)blk_rows * s->d,
                  "swarm_distance_part_table: cand_stride leaves no room for a rank's positions behind its record");
    double *slot = cand_row(*s, rank_slot) + dist_slot(s->d, iteration);
    return launch_distance(s, stream, iteration, s->cand + GDPSO_CAND_DOUBLES(s->d), n_all, rank_slot, s->n_ranks, 0.0, slot, blk_rows,
                           (size_t)s->cand_stride);
}

extern "C" int gdpso_swarm_distance_finish(const gdpso_swarm *s, void *stream, int iteration, int n_all) {
    int rc = check_swarm(s);
    if (rc) return rc;
    GDPSO_REQUIRE(s->mean_dist, "swarm_distance_finish: the swarm keeps no diagnostic");
    GDPSO_REQUIRE(iteration >= 1 && iteration <= s->max_iter, "swarm_distance_finish: iteration out of range");
    GDPSO_REQUIRE(n_all >= 2, "swarm_distance_finish: needs at least two particles");
    hipLaunchKernelGGL(dist_mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *s, iteration, 0.5 * (double)n_all * (double)(n_all - 1));
    GDPSO_HIP(hipGetLastError());
    return GDPSO_OK;
}

extern "C" int gdpso_swarm_after_exchange(const gdpso_swarm *s, void *stream, int iteration, int schedule_inertia, int early_stopping,
                                          double tolerance, double w_cogn, double w_soci, const double *rnd, uint64_t seed,
                                          int f32_first_move, int diag_iteration, int n_all) {
    int rc = check_swarm(s);
    if (rc) return rc;
    GDPSO_REQUIRE(iteration >= 1 && iteration <= s->max_iter, "swarm_after_exchange: iteration out of range");
    GDPSO_REQUIRE(diag_iteration >= 0 && diag_iteration <= s->max_iter, "swarm_after_exchange: diagnostic iteration out of range");
    GDPSO_REQUIRE(diag_iteration == 0 || (s->mean_dist && n_all >= 2), "swarm_after_exchange: the diagnostic needs mean_dist and two particles");
    if (s->n <= SMALL_MAX_N && (long)s->n * s->d <= SMALL_MAX_ELEMS) {
        hipLaunchKernelGGL(swarm_after_exchange_small_kernel, dim3(1), dim3(SMALL_THREADS), 0, (hipStream_t)stream, *s, iteration,
                           schedule_inertia, early_stopping, tolerance, w_cogn, w_soci, rnd, seed, (f32_first_move && iteration == 1) ? 1 : 0,
                           diag_iteration, 0.5 * (double)n_all * (double)(n_all - 1));
        GDPSO_HIP(hipGetLastError());
        return GDPSO_OK;
    }
    if (diag_iteration > 0 && (rc = gdpso_swarm_distance_finish(s, stream, diag_iteration, n_all))) return rc;
    if ((rc = gdpso_swarm_global_best(s, stream, iteration, schedule_inertia, early_stopping, tolerance))) return rc;
    return gdpso_swarm_move(s, stream, iteration, w_cogn, w_soci, rnd, seed, f32_first_move);
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
    const bool diag = r->diagnostics && s->mean_dist && s->n >= 2, defer = g_dist_defer;
    const double pairs = 0.5 * (double)s->n * (double)(s->n - 1);
    if (diag && n_iterations > 0 && (rc = check_distance(s, first_iteration, s->pos, s->n))) return rc;
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
            if (small_step_fast_ok(*s))
                hipLaunchKernelGGL(swarm_step_small_kernel<true>, dim3(1), dim3(SMALL_THREADS), 0, (hipStream_t)stream, *s, i,
                                   r->schedule_inertia, r->early_stopping, r->tolerance, r->w_cogn, r->w_soci, rnd, r->seed,
                                   (r->f32_first_move && i == 1) ? 1 : 0);
            else
                hipLaunchKernelGGL(swarm_step_small_kernel<false>, dim3(1), dim3(SMALL_THREADS), 0, (hipStream_t)stream, *s, i,
                                   r->schedule_inertia, r->early_stopping, r->tolerance, r->w_cogn, r->w_soci, rnd, r->seed,
                                   (r->f32_first_move && i == 1) ? 1 : 0);
            GDPSO_HIP(hipGetLastError());
        } else {
            GDPSO_REQUIRE(gdpso_swarm_scratch_bytes(s->n) <= s->scratch_bytes, "swarm_iterate: scratch too small");
            const int pb = pbest_particles(s->n), nb = ceil_div(s->n, pb);
            hipLaunchKernelGGL(pbest_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *s, i, (BestRec *)s->scratch, pb);
            hipLaunchKernelGGL(candidate_global_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *s, i, (const BestRec *)s->scratch, nb,
                               r->schedule_inertia, r->early_stopping, r->tolerance);
            GDPSO_HIP(hipGetLastError());
            if ((rc = gdpso_swarm_move(s, stream, i, r->w_cogn, r->w_soci, rnd, r->seed, r->f32_first_move))) return rc;
        }
        // the diagnostic: the tiles of iteration i now; their sums are added up by a workgroup of iteration i + 1's tile
        // launch (one launch fewer per iteration), the last iteration's by the finishing kernel
        if (diag) {
            const bool last = i == first_iteration + n_iterations - 1;
            if ((rc = launch_distance(s, stream, i, s->pos, s->n, 0, 1, pairs, s->mean_dist + (i - 1), 0, 0, defer ? (i & 1) : 0, true,
                                      !defer || last, defer && i > first_iteration ? i - 1 : 0)))
                return rc;
        }
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
    const bool defer = g_dist_defer;
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
            int max_blocks = 1, max_ts = 16;
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
                    const int DT = dist_tile_side(s.n, 1);
                    b.dist_tile[j] = DT;
                    b.dist_tiles[j] = ceil_div(s.n, DT);
                    const size_t set_stride = dist_set_stride(dist_tiles(s.n, DT));
                    b.dist_partial[j] = dist_partials(s) + (defer ? (size_t)(i & 1) * set_stride : 0);
                    b.dist_prev[j] = defer && i > first_iteration ? dist_partials(s) + (size_t)(1 - (i & 1)) * set_stride : nullptr;
                    max_blocks = std::max(max_blocks, (int)dist_tiles(s.n, DT));
                    max_ts = std::max(max_ts, DT);
                    dist = true;
                } else {
                    b.s[j].mean_dist = nullptr;  // the distance kernels skip this swarm
                }
            }
            if (small) {
                bool fast = true;
                for (int j = 0; j < cnt; ++j) fast = fast && small_step_fast_ok(swarms[k0 + j]);
                if (fast)
                    hipLaunchKernelGGL(swarm_step_small_multi_kernel<true>, dim3(cnt), dim3(SMALL_THREADS), 0, (hipStream_t)stream, b, i);
                else
                    hipLaunchKernelGGL(swarm_step_small_multi_kernel<false>, dim3(cnt), dim3(SMALL_THREADS), 0, (hipStream_t)stream, b, i);
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
                // (with the finish deferred: a last column of workgroups adds up the previous iteration's tile sums; the last
                // iteration's are added up by the finishing kernel)
                const bool tail = defer && i > first_iteration;
                hipLaunchKernelGGL(pair_dist_multi_kernel, dim3(max_blocks + (tail ? 1 : 0), cnt), dim3(256), dist_lds_bytes(max_ts),
                                   (hipStream_t)stream, b, i);
                bool last = i == first_iteration + n_iterations - 1;
                if (!last) {  // (or no swarm has an iteration i + 1)
                    last = true;
                    for (int k = 0; k < n_swarms; ++k) last = last && i + 1 > swarms[k].max_iter;
                }
                if (!defer || last) hipLaunchKernelGGL(pair_dist_finish_multi_kernel, dim3(cnt), dim3(256), 0, (hipStream_t)stream, b, i);
            }
            GDPSO_HIP(hipGetLastError());
        }
    }
    return GDPSO_OK;
}
