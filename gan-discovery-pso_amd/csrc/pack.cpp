// Host-side weight repacking: PyTorch layouts -> the layouts the kernels read.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace gdpso {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
const char *last_error_cstr() { return g_error.c_str(); }

DeviceArena::~DeviceArena() {
    for (void *p : blocks) (void)hipFree(p);
}

int DeviceArena::upload(const void *host, size_t bytes, const void **dev) {
    void *p = nullptr;
    GDPSO_HIP(hipMalloc(&p, bytes ? bytes : 4));
    blocks.push_back(p);
    if (bytes) GDPSO_HIP(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
    *dev = p;
    return GDPSO_OK;
}

// BatchNorm2d in eval mode: y = (x - mean) / sqrt(var + 1e-5) * gamma + beta, with an
// optional bias added to x first (ConvTranspose2d layers of the generator have one).
void fold_bn(int c, const float *bias, const float *gamma, const float *beta, const float *mean,
             const float *var, std::vector<float> &scale, std::vector<float> &shift) {
    scale.resize(c);
    shift.resize(c);
    for (int i = 0; i < c; ++i) {
        const double a = (double)gamma[i] / std::sqrt((double)var[i] + 1e-5);
        const double b = bias ? (double)bias[i] : 0.0;
        scale[i] = (float)a;
        shift[i] = (float)((b - (double)mean[i]) * a + (double)beta[i]);
    }
}

// A run of consecutive output coordinates along one axis that see the same set of in-range
// kernel offsets (the first rows of a padded convolution miss the top taps, and so on).
struct AxisRun {
    int o0, len;
    std::vector<int> ks;   // kernel indices in range for every coordinate of the run
    std::vector<int> ds;   // the input offset each of them adds
};

// `offset_of(k)` is the input coordinate minus o*stride for kernel index k.
template <class F>
static std::vector<AxisRun> axis_runs(int nk, int stride, int n_out, int n_in, F offset_of, bool split) {
    std::vector<AxisRun> runs;
    unsigned long prev = ~0ul;
    for (int o = 0; o < n_out; ++o) {
        unsigned long mask = 0;
        for (int k = 0; k < nk; ++k) {
            const int i = o * stride + offset_of(k);
            if (i >= 0 && i < n_in) mask |= 1ul << k;
        }
        if (runs.empty() || mask != prev) {
            AxisRun r;
            r.o0 = o;
            r.len = 0;
            for (int k = 0; k < nk; ++k)
                if (mask >> k & 1) {
                    r.ks.push_back(k);
                    r.ds.push_back(offset_of(k));
                }
            runs.push_back(r);
            prev = mask;
        }
        runs.back().len++;
    }
    if (!split && runs.size() > 1) {  // one run with every offset that is ever in range (zeros are read for the rest)
        AxisRun all;
        all.o0 = 0;
        all.len = n_out;
        for (int k = 0; k < nk; ++k) {
            bool any = false;
            for (const AxisRun &r : runs)
                for (int kk : r.ks) any |= kk == k;
            if (any) {
                all.ks.push_back(k);
                all.ds.push_back(offset_of(k));
            }
        }
        runs.assign(1, all);
    }
    return runs;
}

// GDPSO_CONV_BORDER_CLASSES=0 keeps every padded convolution in one phase (A/B measurements).
static bool border_classes_enabled() {
    const char *e = getenv("GDPSO_CONV_BORDER_CLASSES");
    return e ? e[0] != '0' : true;
}

static void sort_phases_heaviest_first(ConvLayer &L) {
    // workgroups are dispatched in phase order: long K loops first, short ones fill the tail
    std::stable_sort(L.phase, L.phase + L.nphase, [](const ConvPhase &a, const ConvPhase &b) { return a.ntaps > b.ntaps; });
}

void make_conv(ConvLayer &L, const float *w, int cout, int cin, int kh, int kw, int stride, int pad,
               int hin, int win, int act, std::vector<float> &packed, int cin_pad, int cout_pad) {
    if (cin_pad < cin) cin_pad = cin;
    if (cout_pad < cout) cout_pad = cout;
    std::memset(&L, 0, sizeof(L));
    L.cin = cin_pad;
    L.cout = cout_pad;
    L.hin = hin;
    L.win = win;
    L.hout = (hin + 2 * pad - kh) / stride + 1;
    L.wout = (win + 2 * pad - kw) / stride + 1;
    L.in_stride = stride;
    L.out_mul = 1;
    L.act = act;
    L.flops_nominal = 2.0 * L.hout * L.wout * cout * cin * kh * kw;
    bool split = border_classes_enabled();
    auto rows = axis_runs(kh, stride, L.hout, hin, [&](int k) { return k - pad; }, split);
    auto cols = axis_runs(kw, stride, L.wout, win, [&](int k) { return k - pad; }, split);
    if (rows.size() * cols.size() > (size_t)MAX_PHASES) {
        rows = axis_runs(kh, stride, L.hout, hin, [&](int k) { return k - pad; }, false);
        cols = axis_runs(kw, stride, L.wout, win, [&](int k) { return k - pad; }, false);
    }
    // weight rows hold the taps that are in range for at least one output pixel
    std::vector<int> slot(kh * kw, -1);
    std::vector<int> live;
    for (const AxisRun &r : rows)
        for (const AxisRun &c : cols)
            for (int ky : r.ks)
                for (int kx : c.ks)
                    if (slot[ky * kw + kx] < 0) slot[ky * kw + kx] = 0;
    for (int t = 0; t < kh * kw; ++t)
        if (slot[t] == 0) {
            slot[t] = (int)live.size();
            live.push_back(t);
        }
    const size_t ktot = live.size() * (size_t)cin_pad;
    packed.assign((size_t)cout_pad * ktot, 0.f);
    for (int co = 0; co < cout; ++co)
        for (size_t t = 0; t < live.size(); ++t)
            for (int ci = 0; ci < cin; ++ci)
                packed[(size_t)co * ktot + t * cin_pad + ci] = w[((size_t)co * cin + ci) * kh * kw + live[t]];
    L.weight_floats = packed.size();
    L.nphase = 0;
    for (const AxisRun &r : rows)
        for (const AxisRun &c : cols) {
            ConvPhase &P = L.phase[L.nphase++];
            P.off_y = r.o0;
            P.off_x = c.o0;
            P.hm = r.len;
            P.wm = c.len;
            P.in_y0 = r.o0 * stride;
            P.in_x0 = c.o0 * stride;
            P.wt_offset = 0;
            P.wstride = (int)ktot;
            P.ntaps = 0;
            for (size_t a = 0; a < r.ks.size(); ++a)
                for (size_t b = 0; b < c.ks.size(); ++b) {
                    P.dy[P.ntaps] = (int8_t)r.ds[a];
                    P.dx[P.ntaps] = (int8_t)c.ds[b];
                    P.wtap[P.ntaps] = (int16_t)slot[r.ks[a] * kw + c.ks[b]];
                    P.ntaps++;
                }
        }
    sort_phases_heaviest_first(L);
}

// ConvTranspose2d(k=4, s=2, p=1): oy = 2*iy - 1 + ky.  Output row 2m takes (iy, ky) in
// {(m,1), (m-1,3)}; row 2m+1 takes {(m,2), (m+1,0)}; likewise along x.  Each of the four
// output (sub-pixel) phases is therefore a 2x2-tap convolution over the input, and each is
// split further into border classes where m-1 or m+1 leaves the input.
void make_convt_k4s2p1(ConvLayer &L, const float *w, int cin, int cout, int hin, int win, int act,
                       std::vector<float> &packed) {
    std::memset(&L, 0, sizeof(L));
    L.cin = cin;
    L.cout = cout;
    L.hin = hin;
    L.win = win;
    L.hout = 2 * hin;
    L.wout = 2 * win;
    L.in_stride = 1;
    L.out_mul = 2;
    L.act = act;
    L.flops_nominal = 2.0 * L.hout * L.wout * cout * cin * 4;
    static const int d_of[2][2] = {{0, -1}, {0, 1}};
    static const int k_of[2][2] = {{1, 3}, {2, 0}};
    const size_t ktot = (size_t)4 * cin;
    packed.assign((size_t)4 * cout * ktot, 0.f);
    L.weight_floats = packed.size();
    const bool split = border_classes_enabled();
    L.nphase = 0;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const long wt_offset = (long)(py * 2 + px) * cout * ktot;
            for (int ty = 0; ty < 2; ++ty)
                for (int tx = 0; tx < 2; ++tx) {
                    const int t = ty * 2 + tx;
                    const int ky = k_of[py][ty], kx = k_of[px][tx];
                    for (int co = 0; co < cout; ++co)
                        for (int ci = 0; ci < cin; ++ci)
                            packed[wt_offset + (size_t)co * ktot + (size_t)t * cin + ci] =
                                w[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx];
                }
            const auto rows = axis_runs(2, 1, hin, hin, [&](int t) { return d_of[py][t]; }, split);
            const auto cols = axis_runs(2, 1, win, win, [&](int t) { return d_of[px][t]; }, split);
            for (const AxisRun &r : rows)
                for (const AxisRun &c : cols) {
                    ConvPhase &P = L.phase[L.nphase++];
                    P.off_y = 2 * r.o0 + py;
                    P.off_x = 2 * c.o0 + px;
                    P.hm = r.len;
                    P.wm = c.len;
                    P.in_y0 = r.o0;
                    P.in_x0 = c.o0;
                    P.wt_offset = wt_offset;
                    P.wstride = (int)ktot;
                    P.ntaps = 0;
                    for (size_t a = 0; a < r.ks.size(); ++a)
                        for (size_t b = 0; b < c.ks.size(); ++b) {
                            P.dy[P.ntaps] = (int8_t)r.ds[a];
                            P.dx[P.ntaps] = (int8_t)c.ds[b];
                            P.wtap[P.ntaps] = (int16_t)(r.ks[a] * 2 + c.ks[b]);
                            P.ntaps++;
                        }
                }
        }
    sort_phases_heaviest_first(L);
}

uint16_t f32_to_bf16_bits(float v) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  // NaN stays a (quiet) NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

uint16_t f32_to_f16_bits(float v) {
    const _Float16 h = (_Float16)v;  // the compiler's conversion rounds to nearest even
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}

// Device table of a layer: per phase PHASE_HDR ints {ntaps, off_y, off_x, wt_offset, hm, wm,
// in_y0, in_x0, wstride, mul(hm * wm), mul(wm), shift(hm * wm) | shift(wm) << 8} (mul / shift: FastDiv) followed by MAX_TAPS packed taps (dy & 0xff) | (dx & 0xff) << 8
// | wtap << 16.
static int packed_tap(const ConvPhase &P, int k) {
    return ((int)P.dy[k] & 0xff) | (((int)P.dx[k] & 0xff) << 8) | ((int)P.wtap[k] << 16);
}

void conv_phase_header(const ConvPhase &P, int *t) {
    t[0] = P.ntaps;
    t[1] = P.off_y;
    t[2] = P.off_x;
    t[3] = (int)P.wt_offset;
    t[4] = P.hm;
    t[5] = P.wm;
    t[6] = P.in_y0;
    t[7] = P.in_x0;
    t[8] = P.wstride;
    const FastDiv dhw = fast_div_for((unsigned)(P.hm * P.wm)), dw = fast_div_for((unsigned)P.wm);
    t[9] = (int)dhw.mul;
    t[10] = (int)dw.mul;
    t[11] = (int)(dhw.shift | (dw.shift << 8));
    t[PHASE_HDR] = P.ntaps > 0 ? packed_tap(P, 0) : 0;
}

int upload_conv(DeviceArena &arena, ConvLayer &L, const std::vector<float> &packed,
                const std::vector<float> &scale, const std::vector<float> &shift, int dtype) {
    L.dtype = dtype;
    std::vector<int> table((size_t)L.nphase * (PHASE_HDR + MAX_TAPS), 0);
    for (int p = 0; p < L.nphase; ++p) {
        const ConvPhase &P = L.phase[p];
        int *t = table.data() + (size_t)p * (PHASE_HDR + MAX_TAPS);
        conv_phase_header(P, t);
        for (int k = 0; k < P.ntaps; ++k) t[PHASE_HDR + k] = packed_tap(P, k);
    }
    int rc;
    // chunk-major copy: element (co, k) of the matrix at wt_offset goes to
    // wt_offset + (((co / 32) * (wstride / vec) + k / vec) * 32 + co % 32) * vec + k % vec
    const int vec = dtype == DT_F32 ? 4 : 8;
    bool quad_ok = L.cout % 32 == 0;
    for (int p = 0; p < L.nphase; ++p) quad_ok = quad_ok && L.phase[p].wstride % vec == 0;
    std::vector<float> quad;
    if (quad_ok) {
        quad.assign(packed.size(), 0.f);
        std::vector<long> done;
        for (int p = 0; p < L.nphase; ++p) {
            const long off = L.phase[p].wt_offset;
            if (std::find(done.begin(), done.end(), off) != done.end()) continue;  // phases share matrices
            done.push_back(off);
            const size_t ws = (size_t)L.phase[p].wstride, cpr = ws / vec;
            for (int co = 0; co < L.cout; ++co)
                for (size_t k = 0; k < ws; ++k)
                    quad[off + ((((size_t)co / 32) * cpr + k / vec) * 32 + co % 32) * vec + k % vec] = packed[off + (size_t)co * ws + k];
        }
    }
    L.wtq = nullptr;
    if (dtype == DT_F32) {
        if ((rc = arena.upload(packed.data(), packed.size() * sizeof(float), &L.wt))) return rc;
        if (quad_ok && (rc = arena.upload(quad.data(), quad.size() * sizeof(float), &L.wtq))) return rc;
    } else {
        std::vector<uint16_t> half(packed.size());
        for (size_t i = 0; i < packed.size(); ++i) half[i] = dtype == DT_BF16 ? f32_to_bf16_bits(packed[i]) : f32_to_f16_bits(packed[i]);
        if ((rc = arena.upload(half.data(), half.size() * sizeof(uint16_t), &L.wt))) return rc;
        if (quad_ok) {
            for (size_t i = 0; i < quad.size(); ++i) half[i] = dtype == DT_BF16 ? f32_to_bf16_bits(quad[i]) : f32_to_f16_bits(quad[i]);
            if ((rc = arena.upload(half.data(), half.size() * sizeof(uint16_t), &L.wtq))) return rc;
        }
    }
    if ((rc = arena.upload(scale, &L.scale))) return rc;
    if ((rc = arena.upload(shift, &L.shift))) return rc;
    if ((rc = arena.upload(table.data(), table.size() * sizeof(int), (const void **)&L.table))) return rc;
    return GDPSO_OK;
}

}  // namespace gdpso

extern "C" {
int gdpso_abi_version(void) { return GDPSO_ABI_VERSION; }
const char *gdpso_last_error(void) { return gdpso::last_error_cstr(); }
size_t gdpso_struct_size(int which) { return which == 0 ? sizeof(gdpso_swarm) : which == 1 ? sizeof(gdpso_run) : 0; }
}

extern "C" unsigned gdpso_fast_div_check(unsigned n, unsigned d) {
    const gdpso::FastDiv f = gdpso::fast_div_for(d);
    return (unsigned)(((unsigned long long)n * f.mul) >> f.shift);
}
