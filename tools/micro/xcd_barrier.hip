// What a hand-off between dependent layers costs inside ONE persistent launch when producer and
// consumer workgroups sit on the same XCD (share one L2), against a hand-off across the whole chip
// and against a kernel boundary.  Every round each workgroup writes a block of floats, passes a
// barrier with the other workgroups of its team and reads (and checks) what the team wrote.
//   mode 0  team = the workgroups of one XCD (by the hardware's XCC_ID); plain stores, drained with
//           s_waitcnt, arrival counter updated by an L2 atomic; consumer reads with sc1 loads (miss
//           the CU's L1, hit the XCD's L2)
//   mode 1  as 0, but the consumer invalidates its L1 once (agent-scope acquire fence) and reads
//           with ordinary loads
//   mode 2  team = whole grid; agent-scope release (L2 write-back) / acquire around the counter
//   mode 3  one kernel launch per round (no barrier inside)
//   mode 4  as 0, but the counter is polled with an atomic read-modify-write (add 0), which executes in the L2
//   mode 5  as 0, but one poller per team member reads its own copy of a "go" word that the last arriver
//           (found by the value its arrival returns) writes once per 8 members
//   mode 6  no atomics: every member stores the round number into its own word of a flag array; wave 0 of each
//           member reads the whole array with one load per lane (sc1) until every word has reached the round
// Also reports how the hardware dealt workgroups to XCDs (blockIdx.x & 7 == XCC_ID ?).
// Build: hipcc --offload-arch=gfx950 -O3 xcd_barrier.hip -o xcd_barrier ; run: ./xcd_barrier [grid] [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BLOCK_FLOATS = 2048;       // written per workgroup per round (8 KB)
constexpr unsigned SPIN_LIMIT = 20000;  // bounded wait: a broken barrier ends the kernel instead of hanging it

struct Ctl {
    unsigned arrive[8][32];   // per-XCD arrival counters, one 128-byte line each
    unsigned members[8][32];  // per-XCD membership tickets
    unsigned all[32];         // whole-grid arrival counter
    unsigned errors, timeouts, mismatch_xcc, pad;
    unsigned team_size[8];
    unsigned go[8][8][32];    // mode 5: per XCD, 8 copies of the round number on separate lines
    unsigned flags[8][64];    // mode 6: per XCD, one word per member
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(63508) & 15u; }  // HW_REG_XCC_ID

// A read-modify-write that the compiler cannot turn into a plain load (it does that to "add 0"): returns the
// counter as the L2 holds it.
__device__ __forceinline__ unsigned l2_atomic_read(unsigned *p) {
    unsigned v, zero = 0;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory");
    return v;
}

__device__ __forceinline__ float value_of(int round, int team, int member, int i) {
    return (float)((round * 131 + team * 17 + member * 3 + i) & 0xffff);
}

template <int MODE>
__global__ __launch_bounds__(256) void rounds_kernel(Ctl *ctl, float *buf, int rounds, int team_size, int round0) {
    __shared__ unsigned s_member, s_team;
    const int tid = threadIdx.x;
    if (tid == 0) {
        if (MODE == 2 || MODE == 3) {
            s_team = 0;
            s_member = blockIdx.x;
        } else {
            const unsigned x = xcc_id();
            s_team = x;
            s_member = __hip_atomic_fetch_add(&ctl->members[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (x != (blockIdx.x & 7u)) __hip_atomic_fetch_add(&ctl->mismatch_xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    const int team = (int)s_team, member = (int)s_member;
    if (MODE < 2 && member >= team_size) return;  // more workgroups on this XCD than the team holds: not part of the test
    unsigned *cnt = (MODE == 2) ? &ctl->all[0] : &ctl->arrive[team][0];
    const size_t team_floats = (size_t)team_size * BLOCK_FLOATS;
    unsigned bad = 0;
    for (int r = round0; r < round0 + rounds; ++r) {
        float *mine = buf + ((size_t)(r & 1) * 8 + team) * team_floats + (size_t)member * BLOCK_FLOATS;
        for (int i = tid * 4; i < BLOCK_FLOATS; i += 1024) {
            f32x4 v = {value_of(r, team, member, i), value_of(r, team, member, i + 1), value_of(r, team, member, i + 2), value_of(r, team, member, i + 3)};
            *(f32x4 *)(mine + i) = v;
        }
        if (MODE == 6) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&ctl->flags[team][member], (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (tid < 64) {
                unsigned spins = 0;
                for (;;) {
                    const unsigned f = tid < team_size ? __hip_atomic_load(&ctl->flags[team][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                    if (__all((int)(f >= (unsigned)(r + 1)))) break;
                    if (++spins > SPIN_LIMIT) {
                        if (tid == 0) __hip_atomic_fetch_add(&ctl->timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __syncthreads();
        } else if (MODE != 3) {
            if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned want = (unsigned)(r + 1) * (unsigned)team_size;
                unsigned ticket = 0;
                if (MODE == 2)
                    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (MODE == 5)
                    ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else
                    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                unsigned spins = 0;
                if (MODE == 5 && ticket == want - 1) {
                    for (int c = 0; c < 8; ++c) __hip_atomic_store(&ctl->go[team][c][0], (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // the counter is read with an atomic at agent scope: never served from this CU's L1
                while ((MODE == 4 ? l2_atomic_read(cnt)
                        : MODE == 5 ? (__hip_atomic_load(&ctl->go[team][member & 7][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(r + 1) ? want : 0u)
                                    : __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < want) {
                    if (++spins > SPIN_LIMIT) {
                        __hip_atomic_fetch_add(&ctl->timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (MODE == 1 || MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
            return;  // mode 3 checks in the next launch
        }
        // read four 16-byte pieces written by four other members
        const float *teambuf = buf + ((size_t)(r & 1) * 8 + team) * team_floats;
        for (int k = 0; k < 4; ++k) {
            const int m = (member + 1 + k * 7) % team_size;
            const int i = (tid * 4 + k * 1024) % BLOCK_FLOATS;
            f32x4 v;
            if (MODE == 0 || MODE >= 4) {
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)teambuf, 0, (int)(team_floats * 4), 0x00020000);
                v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((size_t)m * BLOCK_FLOATS + i) * 4), 0, 16 /* sc1 */));
            } else {
                v = *(const f32x4 *)(teambuf + (size_t)m * BLOCK_FLOATS + i);
            }
            for (int q = 0; q < 4; ++q) bad += v[q] != value_of(r, team, m, i + q);
        }
    }
    if (bad) __hip_atomic_fetch_add(&ctl->errors, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            return 1;                                                             \
        }                                                                         \
    } while (0)

template <int MODE>
static int run(const char *name, Ctl *ctl, float *buf, int grid, int rounds) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int team_size = MODE == 2 || MODE == 3 ? grid : grid / 8;
    float best = 1e30f;
    Ctl h;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctl, 0, sizeof(Ctl)));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (MODE == 3) {
            for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(rounds_kernel<MODE>, dim3(grid), dim3(256), 0, 0, ctl, buf, 1, team_size, r);
        } else {
            hipLaunchKernelGGL(rounds_kernel<MODE>, dim3(grid), dim3(256), 0, 0, ctl, buf, rounds, team_size, 0);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    }
    fflush(stdout);
    printf("%-44s %7.2f us/round  errors %u timeouts %u xcc!=bid&7 %u  members", name, 1e3 * best / rounds, h.errors, h.timeouts, h.mismatch_xcc);
    for (int x = 0; x < 8; ++x) printf(" %u", h.members[x][0]);
    printf("\n");
    return 0;
}

int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    const int rounds = argc > 2 ? atoi(argv[2]) : 200;
    if (grid <= 0 || grid % 8 != 0 || grid > 2048 || rounds <= 0) {
        fprintf(stderr, "grid: a multiple of 8 up to 2048 (every workgroup must be resident at once)\n");
        return 2;
    }
    Ctl *ctl;
    float *buf;
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMalloc(&buf, (size_t)2 * 8 * grid * BLOCK_FLOATS * sizeof(float)));
    CK(hipMemset(buf, 0, (size_t)2 * 8 * grid * BLOCK_FLOATS * sizeof(float)));
    printf("grid %d, %d rounds, %d B written per workgroup per round\n", grid, rounds, BLOCK_FLOATS * 4);
    if (run<0>("XCD team, L2 atomic, sc1 loads", ctl, buf, grid, rounds)) return 1;
    if (run<1>("XCD team, L2 atomic, L1 invalidate", ctl, buf, grid, rounds)) return 1;
    if (run<2>("whole grid, agent release/acquire", ctl, buf, grid, rounds)) return 1;
    if (run<3>("one launch per round", ctl, buf, grid, rounds)) return 1;
    if (run<4>("XCD team, poll with L2 atomic add 0", ctl, buf, grid, rounds)) return 1;
    if (run<5>("XCD team, last arriver writes go words", ctl, buf, grid, rounds)) return 1;
    if (grid <= 512 && run<6>("XCD team, flag array, no atomics", ctl, buf, grid, rounds)) return 1;
    return 0;
}
